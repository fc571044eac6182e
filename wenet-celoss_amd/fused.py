"""Joiner + RNN-T loss as ONE autograd node: the loss block of the reference forward
(wenet/transducer/transducer.py:131-147: `self.joint(...)` followed by `torchaudio.functional.rnnt_loss(...)`).

Why one node.  The unfused pair moves the (B, T, U+1, V) logits tensor through HBM five times (joiner write, loss
pass 1 read, gradient pass read + write, joiner backward read).  A joiner forward workgroup owns every column of
its 64 lattice cells, so it can produce the loss's row statistics (denom, skip / emit log-probabilities) in its
epilogue (`wr_joint_fwd_lse`); the loss then only runs its lattice sweeps (`wr_rnnt_loss_fwd_from_lse`) and pass 1,
one full read of the logits, is gone.  Whether that pays is a measurement (see `forward` below): the split-precision
forward takes the epilogue, the exact-fp32 forward runs the loss's own row pass (14 ms per 32 utterances at the BASELINE
shape since the round-2 launch-shape change).  And because
the logits are internal to the node, the gradient pass writes over them: one logits-sized tensor instead of two.

Results: costs and every gradient agree with the unfused path to fp32 rounding of the row log-sum-exp (the
statistics are merged in a different order); tests/test_fused_gpu.py states the tolerance (1e-6 relative on costs).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import _lib
from .joint import _PRECISIONS, _resolve_precision, activation_code, joint_backward


class _JointRnntFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, ep, pp, w, b, targets, llens, tlens, blank, clamp, terms, act=0):
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
            # Row statistics of the loss: as the joiner forward's epilogue (wr_joint_fwd*_lse) or as the loss's own row pass.
            # Round 2 measured both ways per 8 utterances at the BASELINE shape: the split-precision forward pays 1.7 ms for
            # the epilogue against 3.5 ms for the row pass (epilogue wins); the exact-fp32 forward, since its fragment-layout
            # rewrite, pays 5.6 ms (54.8 against 49.2 ms: the epilogue's vector work does not hide behind the wave's own
            # MFMAs) against the same 3.5 ms (the row pass wins).  WR_FUSED_LSE_EPILOGUE=1 / 0 forces either.
            epi = os.environ.get("WR_FUSED_LSE_EPILOGUE")
            epilogue = (terms != 0) if epi is None else (epi == "1")
            if terms == 0 and not epilogue:
                ws_bytes = lib.wr_joint_workspace_bytes(J, V)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                rc = lib.wr_joint_fwd(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens), _lib.ptr(tlens),
                                      B, T, U1, J, V, act, _lib.ptr(logits), _lib.ptr(ws), ws_bytes, st)
                _lib.check(rc, "wr_joint_fwd")
            elif not epilogue:
                ws_bytes = lib.wr_joint_split_workspace_bytes(J, V)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                rc = lib.wr_joint_fwd_split(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens),
                                            _lib.ptr(tlens), B, T, U1, J, V, act, terms, _lib.ptr(logits), _lib.WR_F32,
                                            _lib.ptr(ws), ws_bytes, st)
                _lib.check(rc, "wr_joint_fwd_split")
            elif terms == 0:
                ws_bytes = lib.wr_joint_workspace_bytes(J, V)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                rc = lib.wr_joint_fwd_lse(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens),
                                          _lib.ptr(tlens), _lib.ptr(targets), B, T, U1, J, V, act, blank, _lib.ptr(logits),
                                          _lib.ptr(ws), ws_bytes, _lib.ptr(rws), rws_bytes, st)
                _lib.check(rc, "wr_joint_fwd_lse")
            else:
                ws_bytes = lib.wr_joint_split_workspace_bytes(J, V)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                rc = lib.wr_joint_fwd_split_lse(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens),
                                                _lib.ptr(tlens), _lib.ptr(targets), B, T, U1, J, V, act, blank, terms,
                                                _lib.ptr(logits), _lib.ptr(ws), ws_bytes, _lib.ptr(rws), rws_bytes, st)
                _lib.check(rc, "wr_joint_fwd_split_lse")
            if epilogue:
                rc = lib.wr_rnnt_loss_fwd_from_lse(_lib.ptr(logits), _lib.ptr(targets), _lib.ptr(llens), _lib.ptr(tlens), B, T,
                                                   U1, V, blank, _lib.ptr(costs), _lib.ptr(rws), rws_bytes, st)
                _lib.check(rc, "wr_rnnt_loss_fwd_from_lse")
            else:
                rc = lib.wr_rnnt_loss_fwd(_lib.ptr(logits), _lib.WR_F32, _lib.ptr(targets), _lib.ptr(llens), _lib.ptr(tlens), B,
                                          T, U1, V, blank, _lib.ptr(costs), _lib.ptr(rws), rws_bytes, st)
                _lib.check(rc, "wr_rnnt_loss_fwd")
        ctx.save_for_backward(ep, pp, w, b, targets, llens, tlens, logits, rws)
        ctx.blank, ctx.clamp, ctx.terms, ctx.act = blank, clamp, terms, act
        ctx.logits_hold_gradient = False
        return costs

    @staticmethod
    def _recompute_logits(ctx, lib, ep, pp, w, b, llens, tlens, logits):
        """A second backward through a retained graph (retain_graph=True, per-loss torch.autograd.grad) finds the
        gradient of the first one where the logits were: the write went through a raw pointer, which autograd's
        version counter never sees.  The joiner forward is deterministic and its plain / epilogue variants give
        bit-identical logits, so the node rebuilds them in the same buffer (one forward's time) instead of failing
        or -- what it did before -- differentiating gradients-as-logits.  The lattice in `rws` is still the first
        forward's (the gradient pass only reads it)."""
        B, T, U1, V = logits.shape
        J = ep.shape[2]
        dev = logits.device
        st = _lib.current_stream(dev)
        if ctx.terms == 0:
            ws_bytes = lib.wr_joint_workspace_bytes(J, V)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            rc = lib.wr_joint_fwd(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens), _lib.ptr(tlens),
                                  B, T, U1, J, V, ctx.act, _lib.ptr(logits), _lib.ptr(ws), ws_bytes, st)
            _lib.check(rc, "wr_joint_fwd")
        else:
            ws_bytes = lib.wr_joint_split_workspace_bytes(J, V)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            rc = lib.wr_joint_fwd_split(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens),
                                        _lib.ptr(tlens), B, T, U1, J, V, ctx.act, ctx.terms, _lib.ptr(logits), _lib.WR_F32,
                                        _lib.ptr(ws), ws_bytes, st)
            _lib.check(rc, "wr_joint_fwd_split")

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_costs):
        ep, pp, w, b, targets, llens, tlens, logits, rws = ctx.saved_tensors
        lib = _lib.load()
        B, T, U1, V = logits.shape
        dev = logits.device
        gc = grad_costs.to(torch.float32).contiguous()
        if ctx.logits_hold_gradient:
            with torch.cuda.device(dev):
                _JointRnntFn._recompute_logits(ctx, lib, ep, pp, w, b, llens, tlens, logits)
            ctx.logits_hold_gradient = False
        # Nothing else holds the logits, so the gradient overwrites them (one logits-sized tensor instead of two).  With
        # round 1's plain loads that cost the gradient pass ~11 % (a line rewritten microseconds after it was read); with
        # the non-temporal loads of round 2 it costs nothing measurable (46.97 / 46.94 against 47.03 / 47.40 ms per
        # 32-utterance step).  WR_FUSED_INPLACE_BYTES=n keeps a separate buffer below n bytes.
        inplace = logits.numel() * logits.element_size() >= int(os.environ.get("WR_FUSED_INPLACE_BYTES", "0"))
        grads = logits if inplace else torch.empty_like(logits)
        with torch.cuda.device(dev):
            rc = lib.wr_rnnt_loss_bwd(_lib.ptr(logits), _lib.WR_F32, _lib.ptr(targets), _lib.ptr(llens), _lib.ptr(tlens),
                                      B, T, U1, V, ctx.blank, float(ctx.clamp), _lib.ptr(gc), _lib.ptr(grads),
                                      _lib.ptr(rws), rws.numel(), _lib.current_stream(dev))
        _lib.check(rc, "wr_rnnt_loss_bwd")
        ctx.logits_hold_gradient = inplace
        d_ep, d_pp, d_w, d_b = joint_backward(grads, ep, pp, w, llens, tlens, ctx.terms, ctx.needs_input_grad[2],
                                              ctx.needs_input_grad[3], gout_zero_in_padding=True, act=ctx.act)
        return d_ep, d_pp, d_w, d_b, None, None, None, None, None, None, None


def plan_buckets(t_lens, u_lens, max_buckets: int = 4, min_gain: float = 0.08, min_cells: int = 20000):
    """Group utterances by label length so that each group is padded to its own maxima.

    The joiner kernels skip 64- / 256-cell tiles that lie wholly in padding, which removes the frames beyond an
    utterance's length (runs of U+1 cells) but not the label positions beyond its label count (short runs inside every
    frame): their cost is B * T * (Umax + 1) whatever the individual label counts are.  Sorting the batch by label count
    and cutting it into a few groups, each a joiner + loss call of its own padded to ITS longest label sequence (and
    frame count), removes most of that padding; per-utterance costs and gradients do not change (utterances are
    independent; the weight gradient is the sum over the groups).

    Returns a list of index lists (ascending label length), or None when one call is best: dynamic programming over the
    sorted order with cost sum_g n_g * maxT_g * (maxU_g + 1), at most `max_buckets` groups, and the split must save at
    least `min_gain` of the cells; batches of fewer than `min_cells` lattice cells (a couple of milliseconds of joiner
    work, e.g. the n-best list of a rescoring call) are not worth the extra launches."""
    n = len(t_lens)
    order = sorted(range(n), key=lambda i: (u_lens[i], t_lens[i]))
    whole = n * max(t_lens) * (max(u_lens) + 1)
    if n < 2 or whole < max(min_cells, 1):
        return None
    us = np.asarray([u_lens[i] for i in order], dtype=np.int64)
    ts = np.asarray([t_lens[i] for i in order], dtype=np.int64)
    # cost[a, b] of the group of sorted positions [a, b) = (b - a) * max(ts[a:b]) * (us[b - 1] + 1): the running
    # maximum per start index makes it one vector operation per row (the host runs this every training step; the
    # first version recomputed max(ts[a:b]) inside a triple Python loop, 1 s at 512 utterances).
    INF = np.iinfo(np.int64).max // 4
    cost = np.full((n + 1, n + 1), INF, dtype=np.int64)
    width = us + 1
    for a in range(n):
        cnt = np.arange(1, n - a + 1, dtype=np.int64)
        cost[a, a + 1:] = cnt * np.maximum.accumulate(ts[a:]) * width[a:]
    best = np.full((max_buckets + 1, n + 1), INF, dtype=np.int64)
    cut = np.zeros((max_buckets + 1, n + 1), dtype=np.int64)
    best[0, 0] = 0
    for g in range(1, max_buckets + 1):
        cand = np.minimum(best[g - 1][:, None] + cost, INF)       # [a, b]; INF rows / entries stay INF
        cut[g] = cand.argmin(axis=0)                                # first minimum = smallest a, as the scalar loop did
        best[g] = cand[cut[g], np.arange(n + 1)]
    g_best = min(range(1, max_buckets + 1), key=lambda g: int(best[g][n]))
    if g_best == 1 or best[g_best][n] > (1.0 - min_gain) * whole:
        return None
    groups, b = [], n
    for g in range(g_best, 0, -1):
        a = int(cut[g][b])
        groups.append(order[a:b])
        b = a
    return groups[::-1]


def joint_rnnt_loss(ep: torch.Tensor, pp: torch.Tensor, w_out: torch.Tensor, b_out: torch.Tensor,
                    targets: torch.Tensor, logit_lengths: torch.Tensor, target_lengths: torch.Tensor, blank: int = 0,
                    clamp: float = -1.0, reduction: str = "mean", precision: Optional[str] = None,
                    buckets: Optional[int] = None, activation: str = "tanh") -> torch.Tensor:
    """rnnt_loss(ffn_out(act(ep[:, :, None] + pp[:, None])), targets, logit_lengths, target_lengths) without the
    logits ever leaving the node.  ep (B, T, J) = enc_ffn(encoder_out), pp (B, U+1, J) = pred_ffn(predictor_out);
    targets (B, U) int32 with padding already mapped to a valid class; lengths (B,) int32; requires
    max(logit_lengths) == T and max(target_lengths) + 1 == U+1 like torchaudio's rnnt_loss.
    ``precision``: "fp32" (exact MFMA, default) or "bf16x3" (split precision, joint.py); reduction as rnnt_loss.
    ``buckets``: at most this many groups by label length, each padded to its own maxima (`plan_buckets`; default from
    WR_FUSED_BUCKETS, 4; 1 = one call for the whole batch).  Costs come back in the caller's utterance order."""
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
    if buckets is None:
        buckets = int(os.environ.get("WR_FUSED_BUCKETS", "4"))
    groups = plan_buckets(lens[0].tolist(), lens[1].tolist(), max_buckets=buckets) if buckets > 1 else None
    terms = _PRECISIONS[precision]
    act = activation_code(activation)
    if groups is None:
        costs = _JointRnntFn.apply(ep, pp, w_out, b_out, tg, ll, tl, int(blank), float(clamp), terms, act)
    else:
        costs = torch.empty(B, dtype=torch.float32, device=dev)
        parts, index = [], []
        for g in groups:
            idx = torch.tensor(g, device=dev)
            tg_max, ug_max = int(lens[0][g].max()), int(lens[1][g].max())
            parts.append(_JointRnntFn.apply(ep[idx, :tg_max], pp[idx, :ug_max + 1], w_out, b_out,
                                            tg[idx, :ug_max].contiguous(), ll[idx].contiguous(), tl[idx].contiguous(),
                                            int(blank), float(clamp), terms, act))
            index.append(idx)
        costs = torch.cat(parts)[torch.argsort(torch.cat(index))]      # back to the caller's order (differentiable)
    if reduction == "mean":
        return costs.mean()
    if reduction == "sum":
        return costs.sum()
    return costs
