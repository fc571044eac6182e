"""TransducerJoint with the reference's interface, backed by the MFMA kernels.

Mirror of ``wenet/transducer/joint.py:9-70``: same constructor arguments and
parameter names (``enc_ffn``, ``pred_ffn``, ``ffn_out`` -- reference checkpoints
load unchanged), same ``forward(enc_out, pred_out) -> (B, T, U, V)``.

The two pre-join projections are plain library GEMMs on small tensors
(torch.nn.Linear on rocBLAS).  Everything after them -- broadcast add, tanh,
the 512 -> V contraction and its backward w.r.t. the activations -- is the
fused HIP path (``wr_joint_fwd`` / ``wr_joint_bwd_dz`` / ``wr_joint_bwd_dw``); only the
two reductions of ``dZ`` over u / t are library sum calls.

Precision (extension, default exact fp32): ``precision="bf16x3"`` (or env ``WR_JOINT_PRECISION=bf16x3``) runs the
forward contraction on the bf16 matrix cores with every fp32 operand split in two (three MFMA terms, fp32
accumulation): logits within 1e-4 of the fp32 result relative to their scale, 2.7x faster.
``precision="bf16"`` is the single-term AMP mode (the reference under ``--use_amp`` runs this Linear in fp16):
under autocast the logits come out in the autocast dtype.  In both modes the activation gradient ``dZ = dY W``
and the weight gradient ``dW = dY^T H`` use the same split (``wr_joint_bwd_dz_split``, ``wr_joint_bwd_dw_split``,
when V is a multiple of 4; otherwise the exact kernels); the bias gradient is summed in fp32.

Supported configurations: ``joint_mode='add'`` (the only mode the reference accepts, joint.py:22); every
``activation`` of ``get_activation`` (common.py:228-242: tanh -- shipped --, relu, hardtanh, selu, swish, gelu; value
and derivative are evaluated inside the kernels from the pre-activation), ``prejoin_linear`` on (shipped, conf/encoder_bias_conformer_rnnt_*.yaml:21-26) or off,
``postjoin_linear`` off (shipped) or on (training forward; distributed over the two addends, see
``pre_activation``).
"""
from __future__ import annotations

from typing import Optional

import os

import torch
from torch import nn

from . import _lib

_PRECISIONS = {"fp32": 0, "bf16x3": 3, "bf16": 1}


def activation_code(activation: str) -> int:
    """wr_activation code of a get_activation name (wenet/utils/common.py:228-242)."""
    if activation not in _lib.ACTIVATIONS:
        raise KeyError(f"joint activation must be one of {sorted(_lib.ACTIVATIONS)}, got {activation!r}")
    return _lib.ACTIVATIONS[activation]


_TORCH_ACTIVATIONS = {"tanh": nn.Tanh, "relu": nn.ReLU, "hardtanh": nn.Hardtanh, "selu": nn.SELU, "swish": nn.SiLU,
                      "gelu": nn.GELU}


def _resolve_precision(precision: Optional[str]) -> str:
    if precision is None:
        precision = os.environ.get("WR_JOINT_PRECISION", "fp32")
    if precision not in _PRECISIONS:
        raise ValueError(f"joint precision must be one of {sorted(_PRECISIONS)}, got {precision!r}")
    return precision


class _JointFn(torch.autograd.Function):
    # Under AMP (executor.py:91 wraps the forward in autocast) the pre-join Linear layers hand over fp16/bf16
    # activations; the MFMA kernels are exact-fp32, so inputs are cast up and the logits come out fp32.
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, ep, pp, w, b, llens, tlens, terms=0, out_dtype=torch.float32, act=0):
        if not ep.is_cuda:
            raise RuntimeError("wenet_celoss_amd.TransducerJoint: tensors must live on a HIP device "
                               "(this package has no CPU path)")
        lib = _lib.load()
        B, T, J = ep.shape
        U1 = pp.shape[1]
        V = w.shape[0]
        dev = ep.device
        ep, pp, w, b = ep.contiguous(), pp.contiguous(), w.contiguous(), b.contiguous()
        if terms == 0:
            out = torch.empty(B, T, U1, V, dtype=torch.float32, device=dev)
            ws_bytes = lib.wr_joint_workspace_bytes(J, V)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = lib.wr_joint_fwd(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens),
                                      _lib.ptr(tlens), B, T, U1, J, V, act, _lib.ptr(out), _lib.ptr(ws), ws_bytes,
                                      _lib.current_stream(dev))
            _lib.check(rc, "wr_joint_fwd")
        else:
            out = torch.empty(B, T, U1, V, dtype=out_dtype, device=dev)
            ws_bytes = lib.wr_joint_split_workspace_bytes(J, V)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = lib.wr_joint_fwd_split(_lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(b), _lib.ptr(llens),
                                            _lib.ptr(tlens), B, T, U1, J, V, act, terms, _lib.ptr(out),
                                            _lib.dtype_code(out_dtype), _lib.ptr(ws), ws_bytes,
                                            _lib.current_stream(dev))
            _lib.check(rc, "wr_joint_fwd_split")
        ctx.save_for_backward(ep, pp, w, llens, tlens)
        ctx.terms = terms
        ctx.act = act
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gout):
        ep, pp, w, llens, tlens = ctx.saved_tensors
        d_ep, d_pp, d_w, d_b = joint_backward(gout, ep, pp, w, llens, tlens, ctx.terms, ctx.needs_input_grad[2],
                                              ctx.needs_input_grad[3], act=ctx.act)
        return d_ep, d_pp, d_w, d_b, None, None, None, None, None


_MM_OUT_DTYPE = None


def _mm_takes_out_dtype() -> bool:
    """torch.mm(bf16, bf16, out_dtype=float32) -- fp32 results of a bf16 GEMM without a rounding to bf16 in between -- is what
    the library form of the AMP backward needs; a PyTorch without it keeps the backward on this package's kernels."""
    global _MM_OUT_DTYPE
    if _MM_OUT_DTYPE is None:
        try:
            a = torch.zeros(8, 8, dtype=torch.bfloat16, device="cuda")
            _MM_OUT_DTYPE = torch.mm(a, a, out_dtype=torch.float32).dtype == torch.float32
        except (TypeError, RuntimeError):
            _MM_OUT_DTYPE = False
    return _MM_OUT_DTYPE


def _amp_backward_library(lib, gout, ep, pp, w, llens, tlens, need_w, need_b, gout_zero_in_padding, act):
    """Single-term (AMP) backward with a 16-bit (bfloat16 or float16) logits gradient: the two contractions dH = dY W and [dW | db] = dY^T [H | 1]
    are plain 16-bit GEMMs with fp32 accumulation and fp32 results -- they go to the vendor GEMM library (measured at the
    B = 16 BASELINE slice: 11.9 + 15.8 ms against 23.0 + 34.5 ms for this package's single-term kernels, which stay
    reachable with WR_AMP_BACKWARD=kernels); what is fused around them stays here: ``wr_joint_dz_act`` applies the
    activation's derivative to dH in place and writes H in bf16 (zero in padded cells), ``wr_joint_db_bf16`` sums the
    gradient's columns for the bias; the gradient tensor is never widened."""
    B, T, J = ep.shape
    U1 = pp.shape[1]
    V = w.shape[0]
    dev = ep.device
    M = B * T * U1
    g2 = gout.view(M, V)
    dt = gout.dtype                                 # bfloat16, or float16 (autocast's default dtype: the reference's --use_amp)
    dz = torch.mm(g2, w.to(dt), out_dtype=torch.float32).view(B, T, U1, J)
    hb = torch.empty(M, J, dtype=dt, device=dev) if need_w else None
    with torch.cuda.device(dev):
        rc = lib.wr_joint_dz_act(_lib.ptr(dz), _lib.ptr(ep), _lib.ptr(pp), _lib.ptr(llens), _lib.ptr(tlens), B, T, U1, J, act,
                                 _lib.ptr(hb), _lib.dtype_code(dt), J, _lib.current_stream(dev))
    _lib.check(rc, "wr_joint_dz_act")
    d_ep = dz.sum(dim=2)
    d_pp = dz.sum(dim=1)
    d_w = d_b = None
    if need_w:                                      # H is zero in padded cells: they contribute nothing
        d_w = torch.mm(g2.t(), hb, out_dtype=torch.float32)
    if need_b:                                      # (as one more column of that GEMM, N = J + 8, the library padded to
        d_b = torch.empty(V, dtype=torch.float32, device=dev)       # its next tile: 23.3 ms against 15.8 + 3)
        wsb = lib.wr_joint_db_workspace_bytes(B, T, U1, V)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        if gout_zero_in_padding:
            llens = tlens = None
        with torch.cuda.device(dev):
            fn = lib.wr_joint_db_bf16 if dt == torch.bfloat16 else lib.wr_joint_db_f16
            rc = fn(_lib.ptr(g2), _lib.ptr(llens), _lib.ptr(tlens), B, T, U1, V, _lib.ptr(d_b), _lib.ptr(ws),
                    wsb, _lib.current_stream(dev))
        _lib.check(rc, "wr_joint_db_bf16")
    return d_ep, d_pp, d_w, d_b


def joint_backward(gout, ep, pp, w, llens, tlens, terms: int, need_w: bool, need_b: bool,
                   gout_zero_in_padding: bool = False, act: int = 0):
    """Backward of the joiner from the logits gradient `gout` (B,T,U1,V): returns (d_ep, d_pp, d_w, d_b).
    Shared by the joiner's autograd Function and by the fused joiner + RNN-T loss Function (fused.py).
    `gout_zero_in_padding`: the caller guarantees gout == 0 outside [0,T_b) x [0,U_b] (the RNN-T gradient pass
    zero-fills there).  The activation gradient still gets the lengths (it skips padded tiles and writes H = 0 in
    padded cells), but the weight-gradient reduction then needs no per-row mask: padded rows contribute 0 * 0."""
    lib = _lib.load()
    B, T, J = ep.shape
    U1 = pp.shape[1]
    V = w.shape[0]
    dev = ep.device
    # AMP step: the loss hands back a bf16 gradient for bf16 logits; the split kernels take it as it is (bf16 values are
    # their own hi parts) -- no widening pass over the logits-sized tensor, half the gradient bytes in dZ and dW
    ok16 = terms != 0 and V % 8 == 0 and V >= 32 and J % 4 == 0
    if (ok16 and terms == 1 and gout.dtype in (torch.bfloat16, torch.float16)
            and os.environ.get("WR_AMP_BACKWARD", "library") != "kernels" and _mm_takes_out_dtype()):
        return _amp_backward_library(lib, gout.contiguous(), ep, pp, w, llens, tlens, need_w, need_b, gout_zero_in_padding, act)
    g16 = gout.dtype == torch.bfloat16 and ok16
    gout = gout.contiguous() if g16 else gout.float().contiguous()
    dz = torch.empty(B, T, U1, J, dtype=torch.float32, device=dev)
    h = torch.empty_like(dz) if need_w else None
    if terms != 0 and V % 4 == 0 and V >= 32:       # same split as the forward (gradient rows 16-byte aligned)
        wsb = lib.wr_joint_dz_split_workspace_bytes(J, V)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        fn = lib.wr_joint_bwd_dz_split_bf16 if g16 else lib.wr_joint_bwd_dz_split
        with torch.cuda.device(dev):
            rc = fn(_lib.ptr(gout), _lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(llens),
                    _lib.ptr(tlens), B, T, U1, J, V, act, terms, _lib.ptr(dz), _lib.ptr(h),
                    _lib.ptr(ws), wsb, _lib.current_stream(dev))
        _lib.check(rc, "wr_joint_bwd_dz_split")
    else:
        with torch.cuda.device(dev):
            rc = lib.wr_joint_bwd_dz(_lib.ptr(gout), _lib.ptr(ep), _lib.ptr(pp), _lib.ptr(w), _lib.ptr(llens),
                                     _lib.ptr(tlens), B, T, U1, J, V, act, _lib.ptr(dz), _lib.ptr(h),
                                     _lib.current_stream(dev))
        _lib.check(rc, "wr_joint_bwd_dz")
    d_ep = dz.sum(dim=2)
    d_pp = dz.sum(dim=1)
    d_w = d_b = None
    if gout_zero_in_padding:
        llens = tlens = None
    if need_w:
        d_w = torch.empty(V, J, dtype=torch.float32, device=dev)
        d_b = torch.empty(V, dtype=torch.float32, device=dev)
        if terms != 0 and V % 4 == 0 and J % 4 == 0:
            wsb = lib.wr_joint_dw_split_workspace_bytes(B, T, U1, J, V)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            fn = lib.wr_joint_bwd_dw_split_bf16 if g16 else lib.wr_joint_bwd_dw_split
            with torch.cuda.device(dev):
                rc = fn(_lib.ptr(gout), _lib.ptr(h), _lib.ptr(llens), _lib.ptr(tlens), B, T, U1,
                        J, V, terms, _lib.ptr(d_w), _lib.ptr(d_b), _lib.ptr(ws), wsb,
                        _lib.current_stream(dev))
            _lib.check(rc, "wr_joint_bwd_dw_split")
        else:
            wsb = lib.wr_joint_dw_workspace_bytes(J, V)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = lib.wr_joint_bwd_dw(_lib.ptr(gout), _lib.ptr(h), _lib.ptr(llens), _lib.ptr(tlens), B, T, U1, J, V,
                                         _lib.ptr(d_w), _lib.ptr(d_b), _lib.ptr(ws), wsb, _lib.current_stream(dev))
            _lib.check(rc, "wr_joint_bwd_dw")
    elif need_b:
        g2 = gout.float().view(-1, V)
        if llens is not None:
            tt = torch.arange(T, device=dev)[None, :, None] < llens[:, None, None]
            uu = torch.arange(U1, device=dev)[None, None, :] <= tlens[:, None, None]
            g2 = torch.where((tt & uu).view(-1, 1), g2, torch.zeros((), device=dev))
        d_b = g2.sum(0)
    if not need_b:
        d_b = None
    return d_ep, d_pp, d_w, d_b


def joint_logits(ep: torch.Tensor, pp: torch.Tensor, w_out: torch.Tensor, b_out: torch.Tensor,
                 logit_lengths: Optional[torch.Tensor] = None,
                 target_lengths: Optional[torch.Tensor] = None, precision: Optional[str] = None,
                 activation: str = "tanh") -> torch.Tensor:
    """ffn_out(act(ep[:, :, None] + pp[:, None])) -> (B, T, U1, V); `activation` one of _lib.ACTIVATIONS (default tanh).  With lengths, cells in the
    padded region are left unwritten (they are never read by the RNN-T loss).  ``precision``: see the
    module docstring ("fp32" exact, "bf16x3" split, "bf16" AMP)."""
    precision = _resolve_precision(precision)
    terms = _PRECISIONS[precision]
    out_dtype = torch.float32
    if precision == "bf16" and torch.is_autocast_enabled("cuda"):
        out_dtype = torch.get_autocast_dtype("cuda")
    if (logit_lengths is None) != (target_lengths is None):
        raise RuntimeError("joint_logits: pass both length tensors or neither")
    if logit_lengths is not None:
        logit_lengths = logit_lengths.to(device=ep.device, dtype=torch.int32).contiguous()
        target_lengths = target_lengths.to(device=ep.device, dtype=torch.int32).contiguous()
    return _JointFn.apply(ep, pp, w_out, b_out, logit_lengths, target_lengths, terms, out_dtype, activation_code(activation))


class TransducerJoint(nn.Module):
    """wenet/transducer/joint.py:9-70."""

    def __init__(self, voca_size: int, enc_output_size: int, pred_output_size: int, join_dim: int,
                 prejoin_linear: bool = True, postjoin_linear: bool = False, joint_mode: str = "add",
                 activation: str = "tanh", precision: Optional[str] = None):
        assert joint_mode in ["add"]
        super().__init__()
        self.activation = activation
        self.act_code = activation_code(activation)      # KeyError for an unknown name, as get_activation's lookup
        self.activatoin = _TORCH_ACTIVATIONS[activation]()   # the reference's attribute (sic); used by the export body only
        self.precision = precision                 # None: WR_JOINT_PRECISION or exact fp32
        self.prejoin_linear = prejoin_linear
        self.postjoin_linear = postjoin_linear
        self.joint_mode = joint_mode
        if not self.prejoin_linear and not self.postjoin_linear:
            assert enc_output_size == pred_output_size == join_dim
        self.enc_ffn: Optional[nn.Linear] = None
        self.pred_ffn: Optional[nn.Linear] = None
        if self.prejoin_linear:
            self.enc_ffn = nn.Linear(enc_output_size, join_dim)
            self.pred_ffn = nn.Linear(pred_output_size, join_dim)
        self.post_ffn: Optional[nn.Linear] = None
        if self.postjoin_linear:
            self.post_ffn = nn.Linear(enc_output_size, join_dim)          # joint.py:39-41 (applied to a join_dim tensor)
        self.ffn_out = nn.Linear(join_dim, voca_size)

    def pre_activation(self, enc_out: torch.Tensor, pred_out: torch.Tensor):
        """The two addends whose broadcast sum enters the activation: (B, T, J), (B, U, J).
        prejoin_linear (joint.py:55-58): enc_ffn / pred_ffn.  postjoin_linear (:66-67) applies a Linear to the 4-D sum
        enc[:, :, None] + pred[:, None]; a Linear distributes over the sum, post(e + p) = (W e + b) + W p, so it is
        applied to the two small addends instead of the (B, T, U, J) tensor (same value up to fp32 rounding of one
        addition per element; the 4-D tensor is never formed)."""
        if self.prejoin_linear and self.enc_ffn is not None and self.pred_ffn is not None:
            enc_out = self.enc_ffn(enc_out)
            pred_out = self.pred_ffn(pred_out)
        if self.postjoin_linear and self.post_ffn is not None:
            enc_out = self.post_ffn(enc_out)
            pred_out = torch.nn.functional.linear(pred_out, self.post_ffn.weight)
        return enc_out, pred_out

    def _export_forward(self, enc_out: torch.Tensor, pred_out: torch.Tensor) -> torch.Tensor:
        """TorchScript-export body of the joiner for the step export `forward_joint_step` (transducer.py:619-622):
        a scripted artefact cannot reach the ctypes library, so it carries the reference's module graph
        (joint.py:55-69).  Never executed in eager mode -- `forward` below is the only eager entry."""
        enc = enc_out
        pred = pred_out
        if self.enc_ffn is not None and self.pred_ffn is not None:
            enc = self.enc_ffn(enc_out)
            pred = self.pred_ffn(pred_out)
        out = enc.unsqueeze(2) + pred.unsqueeze(1)
        if self.post_ffn is not None:
            out = self.post_ffn(out)
        return self.ffn_out(self.activatoin(out))

    @torch.jit.unused      # backed by a ctypes autograd Function: opaque to TorchScript (train.py:203-205 smoke export)
    def forward(self, enc_out: torch.Tensor, pred_out: torch.Tensor,
                logit_lengths: Optional[torch.Tensor] = None,
                target_lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """enc_out (B, T, E), pred_out (B, U, P) -> (B, T, U, V).  The optional lengths are an
        extension (skip cells the loss never reads); the reference call passes none."""
        enc_out, pred_out = self.pre_activation(enc_out, pred_out)
        return joint_logits(enc_out, pred_out, self.ffn_out.weight, self.ffn_out.bias, logit_lengths, target_lengths,
                            self.precision, self.activation)
