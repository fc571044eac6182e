"""RNN-T loss with torchaudio's call signature, backed by the HIP kernels.

Drop-in for ``torchaudio.functional.rnnt_loss`` as the reference calls it:
  wenet/transducer/transducer.py:142-147  (reduction="mean", blank=self.blank)
  wenet/transducer/transducer.py:296-301  (reduction='none', rescoring)

Semantics kept from torchaudio 0.10 (the version the reference pins,
README.md:64): logits (B, T, U+1, V) float (or half), targets (B, U) int32,
lengths int32, ``blank=-1`` means the last class, ``clamp`` clips the gradient,
reduction in {"none", "mean", "sum"} where "mean" is a plain batch mean.  The
same argument checks raise ``RuntimeError``.

Difference in mechanism (not in results): torchaudio computes the gradient in
the forward call and multiplies it by ``grad_output`` in backward (two more
passes over a logits-sized tensor).  Here forward does pass 1 + the lattice
sweeps only and backward writes ``grad_output[b] * d cost_b/d logits`` directly,
so the logits-sized tensor is touched three times in total.
"""
from __future__ import annotations

import os

import torch

from . import _lib


def _validate(logits, targets, logit_lengths, target_lengths, blank):
    # Conditions of torchaudio's rnnt_loss (compute.cpp), same error type.
    def req(cond, msg):
        if not cond:
            raise RuntimeError(msg)

    req(logits.device == targets.device == logit_lengths.device == target_lengths.device,
        "logits, targets, logit_lengths and target_lengths must be on the same device")
    req(logits.dtype in (torch.float32, torch.float16, torch.bfloat16), "logits must be float32 or float16 type")
    req(targets.dtype == torch.int32, "targets must be int32 type")
    req(logit_lengths.dtype == torch.int32, "logit_lengths must be int32 type")
    req(target_lengths.dtype == torch.int32, "target_lengths must be int32 type")
    req(logits.is_contiguous(), "logits must be contiguous")
    req(targets.is_contiguous(), "targets must be contiguous")
    req(logit_lengths.is_contiguous(), "logit_lengths must be contiguous")
    req(target_lengths.is_contiguous(), "target_lengths must be contiguous")
    req(logits.dim() == 4, "logits must be 4-D (batch, time, target, class)")
    req(targets.dim() == 2, "targets must be 2-D (batch, max target length)")
    req(logit_lengths.dim() == 1, "logit_lengths must be 1-D")
    req(target_lengths.dim() == 1, "target_lengths must be 1-D")
    B = logits.size(0)
    req(logit_lengths.size(0) == B, "batch dimension mismatch between logits and logit_lengths")
    req(target_lengths.size(0) == B, "batch dimension mismatch between logits and target_lengths")
    req(targets.size(0) == B, "batch dimension mismatch between logits and targets")
    req(0 <= blank < logits.size(-1), "blank must be within [0, logits.shape[-1])")
    lens = torch.stack([logit_lengths, target_lengths]).cpu()     # the one host sync, as torchaudio does
    req(int(lens[0].max()) == logits.size(1), "input length mismatch")
    req(int(lens[1].max()) + 1 == logits.size(2), "output length mismatch")
    req(targets.size(1) == logits.size(2) - 1, "output length mismatch")
    req(int(lens.min()) >= 0, "lengths must be non-negative")


class _RNNTLossFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")          # fp16/bf16 logits are handled natively (fp32 arithmetic inside)
    def forward(ctx, logits, targets, logit_lengths, target_lengths, blank, clamp, inplace_grad):
        if not logits.is_cuda:
            raise RuntimeError("wenet_celoss_amd.rnnt_loss: logits must live on a HIP device "
                               "(this package has no CPU path)")
        lib = _lib.load()
        B, T, U1, V = logits.shape
        dev = logits.device
        ws_bytes = lib.wr_rnnt_workspace_bytes(B, T, U1)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        costs = torch.empty(B, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.wr_rnnt_loss_fwd(_lib.ptr(logits), _lib.dtype_code(logits.dtype), _lib.ptr(targets),
                                      _lib.ptr(logit_lengths), _lib.ptr(target_lengths), B, T, U1, V, blank,
                                      _lib.ptr(costs), _lib.ptr(ws), ws_bytes, _lib.current_stream(dev))
        _lib.check(rc, "wr_rnnt_loss_fwd")
        ctx.save_for_backward(logits, targets, logit_lengths, target_lengths, ws)
        ctx.blank, ctx.clamp, ctx.inplace_grad = blank, clamp, inplace_grad
        return costs.to(logits.dtype) if logits.dtype != torch.float32 else costs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_costs):
        logits, targets, logit_lengths, target_lengths, ws = ctx.saved_tensors
        lib = _lib.load()
        B, T, U1, V = logits.shape
        dev = logits.device
        grads = logits if ctx.inplace_grad else torch.empty_like(logits)
        gc = grad_costs.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            rc = lib.wr_rnnt_loss_bwd(_lib.ptr(logits), _lib.dtype_code(logits.dtype), _lib.ptr(targets),
                                      _lib.ptr(logit_lengths), _lib.ptr(target_lengths), B, T, U1, V, ctx.blank,
                                      float(ctx.clamp), _lib.ptr(gc), _lib.ptr(grads), _lib.ptr(ws), ws.numel(),
                                      _lib.current_stream(dev))
        _lib.check(rc, "wr_rnnt_loss_bwd")
        return grads, None, None, None, None, None, None


def rnnt_loss(logits: torch.Tensor, targets: torch.Tensor, logit_lengths: torch.Tensor,
              target_lengths: torch.Tensor, blank: int = -1, clamp: float = -1, reduction: str = "mean",
              inplace_grad: bool | None = None) -> torch.Tensor:
    """torchaudio.functional.rnnt_loss(logits, targets, logit_lengths,
    target_lengths, blank=-1, clamp=-1, reduction="mean").

    ``inplace_grad`` (extension; default from env WR_RNNT_INPLACE_GRAD, off):
    write the gradient over the logits storage in backward -- halves the
    footprint of the (B,T,U+1,V) tensors when nothing else needs the logits
    (true for the reference's forward, transducer.py:132-147).
    """
    if reduction not in ("none", "mean", "sum"):
        raise ValueError("reduction should be one of 'none', 'mean', or 'sum'")
    if blank < 0:
        blank = logits.shape[-1] + blank
    _validate(logits, targets, logit_lengths, target_lengths, blank)
    if inplace_grad is None:
        inplace_grad = os.environ.get("WR_RNNT_INPLACE_GRAD", "0") == "1"
    costs = _RNNTLossFn.apply(logits, targets, logit_lengths, target_lengths, int(blank), float(clamp),
                              bool(inplace_grad))
    if reduction == "mean":
        return costs.mean()
    if reduction == "sum":
        return costs.sum()
    return costs


class RNNTLoss(torch.nn.Module):
    """torchaudio.transforms.RNNTLoss counterpart."""

    def __init__(self, blank: int = -1, clamp: float = -1.0, reduction: str = "mean"):
        super().__init__()
        self.blank, self.clamp, self.reduction = blank, clamp, reduction

    def forward(self, logits, targets, logit_lengths, target_lengths):
        return rnnt_loss(logits, targets, logit_lengths, target_lengths, self.blank, self.clamp, self.reduction)


def rnnt_lattice(logits, targets, logit_lengths, target_lengths, blank=0):
    """Diagnostics for tests: (costs, alpha, beta) with alpha/beta as plain (B,T,U+1) tensors."""
    lib = _lib.load()
    B, T, U1, V = logits.shape
    dev = logits.device
    ws_bytes = lib.wr_rnnt_workspace_bytes(B, T, U1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    costs = torch.empty(B, dtype=torch.float32, device=dev)
    alpha = torch.empty(B, T, U1, dtype=torch.float32, device=dev)
    beta = torch.empty_like(alpha)
    with torch.cuda.device(dev):
        st = _lib.current_stream(dev)
        _lib.check(lib.wr_rnnt_loss_fwd(_lib.ptr(logits), _lib.dtype_code(logits.dtype), _lib.ptr(targets),
                                        _lib.ptr(logit_lengths), _lib.ptr(target_lengths), B, T, U1, V, blank,
                                        _lib.ptr(costs), _lib.ptr(ws), ws_bytes, st), "wr_rnnt_loss_fwd")
        _lib.check(lib.wr_rnnt_export_lattice(_lib.ptr(ws), ws_bytes, _lib.ptr(logit_lengths),
                                              _lib.ptr(target_lengths), B, T, U1, _lib.ptr(alpha), _lib.ptr(beta),
                                              st), "wr_rnnt_export_lattice")
    return costs, alpha, beta
