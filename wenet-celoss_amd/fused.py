"""Joiner + RNN-T loss as ONE autograd node: the loss block of the reference forward
(wenet/transducer/transducer.py:131-147: `self.joint(...)` followed by `torchaudio.functional.rnnt_loss(...)`).

Why one node.  The unfused pair moves the (B, T, U+1, V) logits tensor through HBM five times (joiner write, loss
pass 1 read, gradient pass read + write, joiner backward read).  A joiner forward workgroup owns every column of
its 64 lattice cells, so it produces the loss's row statistics (denom, skip / emit log-probabilities) in its
epilogue (`wr_joint_fwd_lse`); the loss then only runs its lattice sweeps (`wr_rnnt_loss_fwd_from_lse`) -- pass 1,
one full read of the logits (16.6 ms of the 50 ms loss step at B=32, T=1000, U=150, V=5000), is gone.  And because
the logits are internal to the node, the gradient pass can write over them (done above 16 GiB of logits, where the
footprint matters: one logits-sized tensor instead of two).

Results: costs and every gradient agree with the unfused path to fp32 rounding of the row log-sum-exp (the
statistics are merged in a different order); tests/test_fused_gpu.py states the tolerance (1e-6 relative on costs).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib
from .joint import _PRECISIONS, _resolve_precision, joint_backward


class _JointRnntFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, ep, pp, w, b, targets, llens, tlens, blank, clamp, terms):
        if not ep.is_cuda:
            raise RuntimeError("wenet_celoss_amd.joint_rnnt_loss: tensors must live on a HIP device "
                               "(this package has no CPU path)")
        lib = _lib.load()
        B, T, J = ep.shape
        U1 = pp.shape[1]
        V = w.shape[0]
        dev = ep.device
        ep, pp, w, b = ep.contiguous(), pp.contiguous(), w.contiguous(), b.contiguous()
        logits = torch.empty(B, T, U1, V, dtype=torch.float32, device=dev)
        rws_bytes = lib.wr_rnnt_workspace_bytes(B, T, U1)
        rws = torch.empty(rws_bytes, dtype=torch.uint8, device=dev)
        costs = torch.empty(B, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            if terms == 0:
                ws_bytes = lib.wr_joint_workspace_bytes(J, V)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                rc = lib.wr_joint_fwd_lse(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens),
                                          _lib.ptr(tlens), _lib.ptr(targets), B, T, U1, J, V, blank, _lib.ptr(logits),
                                          _lib.ptr(ws), ws_bytes, _lib.ptr(rws), rws_bytes, st)
                _lib.check(rc, "wr_joint_fwd_lse")
            else:
                ws_bytes = lib.wr_joint_split_workspace_bytes(J, V)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                rc = lib.wr_joint_fwd_split_lse(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens),
                                                _lib.ptr(tlens), _lib.ptr(targets), B, T, U1, J, V, blank, terms,
                                                _lib.ptr(logits), _lib.ptr(ws), ws_bytes, _lib.ptr(rws), rws_bytes, st)
                _lib.check(rc, "wr_joint_fwd_split_lse")
            rc = lib.wr_rnnt_loss_fwd_from_lse(_lib.ptr(logits), _lib.ptr(targets), _lib.ptr(llens), _lib.ptr(tlens), B, T, U1,
                                               V, blank, _lib.ptr(costs), _lib.ptr(rws), rws_bytes, st)
            _lib.check(rc, "wr_rnnt_loss_fwd_from_lse")
        ctx.save_for_backward(ep, pp, w, targets, llens, tlens, logits, rws)
        ctx.blank, ctx.clamp, ctx.terms = blank, clamp, terms
        return costs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_costs):
        ep, pp, w, targets, llens, tlens, logits, rws = ctx.saved_tensors
        lib = _lib.load()
        B, T, U1, V = logits.shape
        dev = logits.device
        gc = grad_costs.to(torch.float32).contiguous()
        # Nothing else holds the logits, so the gradient may overwrite them (one logits-sized tensor instead of two).
        # Rewriting a line microseconds after reading it costs the gradient pass ~11 % (5.2 against 5.85 TB/s, DESIGN.md
        # section 4), so this is done only where the footprint matters: above WR_FUSED_INPLACE_BYTES (default 16 GiB).
        inplace = logits.numel() * logits.element_size() > int(os.environ.get("WR_FUSED_INPLACE_BYTES", str(16 << 30)))
        grads = logits if inplace else torch.empty_like(logits)
        with torch.cuda.device(dev):
            rc = lib.wr_rnnt_loss_bwd(_lib.ptr(logits), _lib.WR_F32, _lib.ptr(targets), _lib.ptr(llens), _lib.ptr(tlens),
                                      B, T, U1, V, ctx.blank, float(ctx.clamp), _lib.ptr(gc), _lib.ptr(grads),
                                      _lib.ptr(rws), rws.numel(), _lib.current_stream(dev))
        _lib.check(rc, "wr_rnnt_loss_bwd")
        d_ep, d_pp, d_w, d_b = joint_backward(grads, ep, pp, w, llens, tlens, ctx.terms, ctx.needs_input_grad[2],
                                              ctx.needs_input_grad[3], gout_zero_in_padding=True)
        return d_ep, d_pp, d_w, d_b, None, None, None, None, None, None


def joint_rnnt_loss(ep: torch.Tensor, pp: torch.Tensor, w_out: torch.Tensor, b_out: torch.Tensor,
                    targets: torch.Tensor, logit_lengths: torch.Tensor, target_lengths: torch.Tensor, blank: int = 0,
                    clamp: float = -1.0, reduction: str = "mean", precision: Optional[str] = None) -> torch.Tensor:
    """rnnt_loss(ffn_out(tanh(ep[:, :, None] + pp[:, None])), targets, logit_lengths, target_lengths) without the
    logits ever leaving the node.  ep (B, T, J) = enc_ffn(encoder_out), pp (B, U+1, J) = pred_ffn(predictor_out);
    targets (B, U) int32 with padding already mapped to a valid class; lengths (B,) int32; requires
    max(logit_lengths) == T and max(target_lengths) + 1 == U+1 like torchaudio's rnnt_loss.
    ``precision``: "fp32" (exact MFMA, default) or "bf16x3" (split precision, joint.py); reduction as rnnt_loss."""
    if reduction not in ("none", "mean", "sum"):
        raise ValueError("reduction should be one of 'none', 'mean', or 'sum'")
    precision = _resolve_precision(precision)
    if precision == "bf16":
        raise ValueError("joint_rnnt_loss: the AMP single-term mode keeps 16-bit logits; use TransducerJoint + rnnt_loss")
    V = w_out.shape[0]
    if blank < 0:
        blank = V + blank
    if not 0 <= blank < V:
        raise RuntimeError("blank must be within [0, logits.shape[-1])")
    dev = ep.device
    tg = targets.to(device=dev, dtype=torch.int32).contiguous()
    ll = logit_lengths.to(device=dev, dtype=torch.int32).contiguous()
    tl = target_lengths.to(device=dev, dtype=torch.int32).contiguous()
    B, T = ep.shape[0], ep.shape[1]
    U1 = pp.shape[1]
    if not (tg.dim() == 2 and tg.shape == (B, U1 - 1) and ll.shape == (B,) and tl.shape == (B,)):
        raise RuntimeError("joint_rnnt_loss: targets must be (B, U) and lengths (B,) for ep (B,T,J), pp (B,U+1,J)")
    lens = torch.stack([ll, tl]).cpu()                      # the one host sync, as in rnnt_loss
    if int(lens[0].max()) != T:
        raise RuntimeError("input length mismatch")
    if int(lens[1].max()) + 1 != U1:
        raise RuntimeError("output length mismatch")
    if int(lens.min()) < 0:
        raise RuntimeError("lengths must be non-negative")
    costs = _JointRnntFn.apply(ep, pp, w_out, b_out, tg, ll, tl, int(blank), float(clamp), _PRECISIONS[precision])
    if reduction == "mean":
        return costs.mean()
    if reduction == "sum":
        return costs.sum()
    return costs
