"""CTC module with the reference's interface, backed by the HIP kernels.

Mirror of ``wenet/transformer/ctc.py:21-84``: same constructor arguments, same
parameter names (``ctc_lo.weight/bias`` so reference checkpoints load), same
``forward(hs_pad, hlens, ys_pad, ys_lens)`` result
(``CTCLoss(reduction='sum')(log_softmax(ctc_lo(dropout(hs_pad)))) / B``), same
``log_softmax`` / ``argmax`` helpers.  The ``ctc_lo`` projection is a plain
library GEMM (torch.nn.Linear on rocBLAS); log-softmax, the alpha/beta lattice
and the gradient w.r.t. the projection output are the fused HIP path
(``wr_ctc_loss_fwd/bwd``).  Quirk kept: ``F.dropout`` is called with its default
``training=True`` (ctc.py:57), so a non-zero ``dropout_rate`` applies in eval too.

CPU tensors (BASELINE config 1: "torch.nn.CTCLoss on CPU, plumbing, runs without a
GPU"): ``CTC.forward`` then executes the reference's own statement sequence --
``log_softmax(2)`` + stock ``torch.nn.CTCLoss`` (ctc.py:44,58-63) -- on PyTorch's CPU
kernels.  That is the reference's call, not a port of the HIP path; the functional
``ctc_loss`` and every other entry point of this package keep raising on CPU tensors.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib


class _CTCLossFn(torch.autograd.Function):
    # Under AMP ctc_lo produces fp16/bf16; the reference's log_softmax autocasts to fp32 (ctc.py:60), so do we.
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits, targets, input_lengths, target_lengths, blank):
        if not logits.is_cuda:
            raise RuntimeError("wenet_celoss_amd.ctc_loss: logits must live on a HIP device "
                               "(this package has no CPU path)")
        lib = _lib.load()
        B, T, V = logits.shape
        S = targets.shape[1]
        dev = logits.device
        ws_bytes = lib.wr_ctc_workspace_bytes(B, T, S)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.wr_ctc_loss_fwd(_lib.ptr(logits), _lib.dtype_code(logits.dtype), _lib.ptr(targets),
                                     _lib.ptr(input_lengths), _lib.ptr(target_lengths), B, T, S, V, blank,
                                     _lib.ptr(nll), _lib.ptr(ws), ws_bytes, _lib.current_stream(dev))
        _lib.check(rc, "wr_ctc_loss_fwd")
        ctx.save_for_backward(logits, targets, input_lengths, target_lengths, ws)
        ctx.blank = blank
        return nll

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_nll):
        logits, targets, input_lengths, target_lengths, ws = ctx.saved_tensors
        lib = _lib.load()
        B, T, V = logits.shape
        S = targets.shape[1]
        dev = logits.device
        grads = torch.empty_like(logits)
        g = grad_nll.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            rc = lib.wr_ctc_loss_bwd(_lib.ptr(logits), _lib.dtype_code(logits.dtype), _lib.ptr(targets),
                                     _lib.ptr(input_lengths), _lib.ptr(target_lengths), B, T, S, V, ctx.blank,
                                     _lib.ptr(g), _lib.ptr(grads), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev))
        _lib.check(rc, "wr_ctc_loss_bwd")
        return grads, None, None, None, None


def ctc_loss(logits: torch.Tensor, targets: torch.Tensor, input_lengths: torch.Tensor,
             target_lengths: torch.Tensor, blank: int = 0, reduction: str = "sum") -> torch.Tensor:
    """Fused log-softmax + CTC loss on batch-major pre-softmax activations.

    logits (B, T, V) float32; targets (B, S) any integer dtype, padded with
    anything (``IGNORE_ID`` = -1 in the reference, processor.py:722-724);
    lengths (B,).  reduction: 'none' | 'sum' | 'mean' (mean as torch.nn.CTCLoss:
    nll / target_length, then batch mean).
    """
    if reduction not in ("none", "sum", "mean"):
        raise ValueError(f"{reduction} is not a valid value for reduction")
    if logits.dim() != 3:
        raise RuntimeError("ctc_loss: logits must be (batch, time, vocab)")
    B = logits.size(0)
    if targets.dim() == 1:
        raise RuntimeError("ctc_loss: concatenated 1-D targets are not supported; pass (batch, max_len)")
    if not (input_lengths.numel() == B and target_lengths.numel() == B and targets.size(0) == B):
        raise RuntimeError("ctc_loss: batch size mismatch between logits, targets and lengths")
    dev = logits.device
    tg = targets.to(device=dev, dtype=torch.int32)
    tg = torch.where(tg < 0, torch.zeros_like(tg), tg).contiguous()
    if tg.size(1) == 0:
        tg = torch.zeros(B, 1, dtype=torch.int32, device=dev)
    il = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
    tl = target_lengths.to(device=dev, dtype=torch.int32).contiguous()
    if logits.dtype != torch.float32:
        logits = logits.float()             # fp16/bf16 activations (AMP or not): the CTC kernels take fp32; the
                                            # cast is differentiable, so the gradient returns in the input dtype
    nll = _CTCLossFn.apply(logits.contiguous(), tg, il, tl, int(blank))
    if reduction == "sum":
        return nll.sum()
    if reduction == "mean":
        return (nll / tl.clamp(min=1).to(nll.dtype)).mean()
    return nll


def ctc_greedy_search(logits: torch.Tensor, lens: torch.Tensor, blank: int = 0, eos: int = -1):
    """ASRModel.ctc_greedy_search (wenet/transformer/asr_model.py:281-324) from the ctc_lo output (B, T, V) on:
    returns (hyps: List[List[int]], scores (B,)).  eos defaults to V-1 (asr_model.py:52-53)."""
    if not logits.is_cuda:
        raise RuntimeError("wenet_celoss_amd.ctc_greedy_search: logits must live on a HIP device (no CPU path)")
    lib = _lib.load()
    x = logits.detach().float().contiguous()
    B, T, V = x.shape
    dev = x.device
    ln = lens.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
    wsb = lib.wr_ctc_decode_workspace_bytes(B, T, 1)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    hyps = torch.empty(B, T, dtype=torch.int32, device=dev)
    hl = torch.empty(B, dtype=torch.int32, device=dev)
    sc = torch.empty(B, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.wr_ctc_greedy_search(_lib.ptr(x), _lib.ptr(ln), B, T, V, int(blank), int(eos if eos >= 0 else V - 1),
                                      _lib.ptr(hyps), _lib.ptr(hl), _lib.ptr(sc), _lib.ptr(ws), wsb, _lib.current_stream(dev))
    _lib.check(rc, "wr_ctc_greedy_search")
    hc, lc = hyps.cpu(), hl.cpu().tolist()
    return [hc[b, :lc[b]].tolist() for b in range(B)], sc


def ctc_prefix_beam_search(logits: torch.Tensor, lens: torch.Tensor, beam_size: int, blank: int = 0):
    """ASRModel._ctc_prefix_beam_search (asr_model.py:326-409) from the ctc_lo output (B, T, V) on.
    Per utterance: [(prefix tuple, score)] best first (the reference's `hyps` for batch size 1)."""
    if not logits.is_cuda:
        raise RuntimeError("wenet_celoss_amd.ctc_prefix_beam_search: logits must live on a HIP device (no CPU path)")
    lib = _lib.load()
    x = logits.detach().float().contiguous()
    B, T, V = x.shape
    dev = x.device
    ln = lens.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
    wsb = lib.wr_ctc_decode_workspace_bytes(B, T, beam_size)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    hyps = torch.empty(B, beam_size, T, dtype=torch.int32, device=dev)
    hl = torch.empty(B, beam_size, dtype=torch.int32, device=dev)
    sc = torch.empty(B, beam_size, dtype=torch.float64, device=dev)
    nh = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.wr_ctc_prefix_beam_search(_lib.ptr(x), _lib.ptr(ln), B, T, V, int(beam_size), int(blank), _lib.ptr(hyps),
                                           _lib.ptr(hl), _lib.ptr(sc), _lib.ptr(nh), _lib.ptr(ws), wsb,
                                           _lib.current_stream(dev))
    _lib.check(rc, "wr_ctc_prefix_beam_search")
    hc, lc, scc, nc = hyps.cpu(), hl.cpu().tolist(), sc.cpu().tolist(), nh.cpu().tolist()
    return [[(tuple(hc[b, e, :lc[b][e]].tolist()), scc[b][e]) for e in range(nc[b])] for b in range(B)]


def forced_align(ctc_probs: torch.Tensor, y: torch.Tensor, blank_id: int = 0) -> list:
    """wenet/utils/ctc_util.py:27-83: ctc_probs (T, D) log-posteriors, y (L,) label ids -> per-frame token list."""
    return forced_align_batch(ctc_probs.unsqueeze(0), y.reshape(1, -1), torch.tensor([ctc_probs.size(0)]),
                              torch.tensor([y.numel()]), blank_id=blank_id, normalized=True)[0]


def forced_align_batch(logits: torch.Tensor, targets: torch.Tensor, input_lengths: torch.Tensor,
                       target_lengths: torch.Tensor, blank_id: int = 0, normalized: bool = False) -> list:
    """Extension: B utterances at once.  logits (B, T, V) pre-softmax (or log-posteriors with normalized=True)."""
    if not logits.is_cuda:
        raise RuntimeError("wenet_celoss_amd.forced_align: tensors must live on a HIP device (no CPU path)")
    lib = _lib.load()
    x = logits.detach().float().contiguous()
    B, T, V = x.shape
    dev = x.device
    tg = targets.to(device=dev, dtype=torch.int32)
    tg = torch.where(tg < 0, torch.zeros_like(tg), tg).contiguous()
    S = tg.shape[1]
    il = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
    tl = target_lengths.to(device=dev, dtype=torch.int32).contiguous()
    wsb = lib.wr_ctc_align_workspace_bytes(B, T, S)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ali = torch.empty(B, T, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.wr_ctc_forced_align(_lib.ptr(x), int(normalized), _lib.ptr(tg), _lib.ptr(il), _lib.ptr(tl), B, T, S, V,
                                     int(blank_id), _lib.ptr(ali), _lib.ptr(ws), wsb, _lib.current_stream(dev))
    _lib.check(rc, "wr_ctc_forced_align")
    ac, ilc = ali.cpu(), il.cpu().tolist()
    return [ac[b, :ilc[b]].tolist() for b in range(B)]


class CTC(torch.nn.Module):
    """CTC module (wenet/transformer/ctc.py:21-84)."""

    def __init__(self, odim: int, encoder_output_size: int, dropout_rate: float = 0.0, reduce: bool = True):
        super().__init__()
        eprojs = encoder_output_size
        self.dropout_rate = dropout_rate
        self.ctc_lo = torch.nn.Linear(eprojs, odim)
        self.reduction_type = "sum" if reduce else "none"
        self.ctc_loss = torch.nn.CTCLoss(reduction=self.reduction_type)     # ctc.py:44; used for CPU tensors only

    @torch.jit.unused      # backed by a ctypes autograd Function: opaque to TorchScript (train.py:203-205 smoke export)
    def forward(self, hs_pad: torch.Tensor, hlens: torch.Tensor, ys_pad: torch.Tensor,
                ys_lens: torch.Tensor) -> torch.Tensor:
        """hs_pad (B, Tmax, D), hlens (B), ys_pad (B, Lmax) padded with -1, ys_lens (B)."""
        ys_hat = self.ctc_lo(F.dropout(hs_pad, p=self.dropout_rate))      # (B, T, V); ctc.py:57
        if not ys_hat.is_cuda:
            # config 1 (CPU plumbing): exactly the reference's statements, on stock PyTorch (ctc.py:58-63)
            ys_hat = ys_hat.transpose(0, 1).log_softmax(2)
            return self.ctc_loss(ys_hat, ys_pad, hlens, ys_lens) / ys_hat.size(1)
        loss = ctc_loss(ys_hat, ys_pad, hlens, ys_lens, blank=0, reduction=self.reduction_type)
        return loss / ys_hat.size(0)                                        # batch-size average; ctc.py:63

    def log_softmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        return F.log_softmax(self.ctc_lo(hs_pad), dim=2)

    def argmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        return torch.argmax(self.ctc_lo(hs_pad), dim=2)
