"""Hot-word greedy search with the gate inside the device step: the fork's default decode
path, wenet/transducer/search/greedy_search.py:297-430 (`basic_greedy_search_both`), for a hot-word module with the
structure of the reference's wenet/transformer/context_bias.py::ContextBias.

Split of the work (the reference's own split, :327-336 before its loop vs :340-425 inside it):
  before the loop, on the caller's module (stock PyTorch, out of scope -- SURVEY.md section 2 row 9):
      bias_hidden of the hot-word list and of the empty list, the two biased encoder outputs and the encoder-side bias
      feature (forward_bias_hidden, forward_encoder_bias);
  the loop itself -- predictor step, predictor biasing (multi-head attention over the list + LayerNorm + Linear +
      LayerNorm), the hot-word gate, the gate / go-back state machine, joiner, log-softmax, argmax -- on the device
      under hipGraph replay (`wr_greedy_search_hotword`, csrc/decode.hip), one host read-back per replay.

`device_capable(cb)` says whether a module has that structure (attribute names and layer types of the reference class,
'linear' / MultiHeadedAttention configuration as shipped in conf/*.yaml); any other hot-word module keeps the
host-driven loop of search/greedy_search.py.
"""
from __future__ import annotations

import ctypes
from typing import List, Tuple

import torch

from . import _lib
from .decoder import DecoderCache, DeviceDecoder, _f32

_LINEARS = ("linear_q", "linear_k", "linear_v", "linear_out")


def device_capable(cb) -> bool:
    """True if `cb` looks like the reference ContextBias in the configuration the device kernels implement."""
    try:
        pb, hb = cb.predictor_bias, cb.hw_bias
        ok = all(isinstance(getattr(pb, n), torch.nn.Linear) and isinstance(getattr(hb, n), torch.nn.Linear) for n in _LINEARS)
        ok = ok and all(isinstance(getattr(cb, n), torch.nn.LayerNorm) for n in
                        ("predictor_bias_bias_norm", "predictor_bias_out_norm", "hw_bias_norm"))
        ok = ok and all(isinstance(getattr(cb, n), torch.nn.Linear) for n in
                        ("predictor_bias_combine", "hw_output_layer", "hw_output_layer_enc", "hw_output_layer_dec"))
        ok = ok and all(callable(getattr(cb, n)) for n in ("forward_bias_hidden", "forward_encoder_bias"))
        D = pb.linear_q.weight.shape[0]
        ok = ok and isinstance(pb.h, int) and D % pb.h == 0 and cb.predictor_bias_combine.weight.shape == (D, 2 * D)
        ok = ok and all(abs(getattr(cb, n).eps - 1e-5) < 1e-12 for n in
                        ("predictor_bias_bias_norm", "predictor_bias_out_norm", "hw_bias_norm"))
        ok = ok and D <= 512 and cb.hw_output_layer_enc.weight.shape[0] <= 256 and cb.hw_output_layer.weight.shape[0] <= 8
        return bool(ok)
    except AttributeError:
        return False


def list_fits(cb, n_ctx: int) -> bool:
    """The bias kernel keeps its vectors and one score per (head, list entry) in LDS: hw_check in csrc/decode.hip admits
    (21 * dim + heads * max_ctx) * 4 bytes <= 60 KB, i.e. about 2 400 entries at dim 256 with 4 heads, 1 000 at dim 512.
    Longer hot-word lists take the host-driven loop (search/greedy_search.py) instead of failing in the attach call."""
    D = cb.predictor_bias.linear_q.weight.shape[0]
    return (21 * D + int(cb.predictor_bias.h) * max(int(n_ctx), 8)) * 4 <= 60 * 1024


class HotwordDecoder(DeviceDecoder):
    """A decoder handle with the hot-word module attached."""

    def __init__(self, predictor, joint, cb, max_lanes: int, tmax: int, max_hyp: int, max_ctx: int):
        super().__init__(predictor, joint, max_lanes, max_lanes, tmax, max_hyp, 1)
        hold = []

        def ptr(t):
            t = _f32(t)
            hold.append(t)
            return t.data_ptr()
        pb, hb = cb.predictor_bias, cb.hw_bias
        w = _lib.HotwordWeights()
        w.dim = pb.linear_q.weight.shape[0]
        w.heads = int(pb.h)
        w.hw_dim = cb.hw_output_layer_enc.weight.shape[0]
        w.n_labels = cb.hw_output_layer.weight.shape[0]
        w.q_w, w.q_b = ptr(pb.linear_q.weight), ptr(pb.linear_q.bias)
        w.k_w, w.k_b = ptr(pb.linear_k.weight), ptr(pb.linear_k.bias)
        w.v_w, w.v_b = ptr(pb.linear_v.weight), ptr(pb.linear_v.bias)
        w.o_w, w.o_b = ptr(pb.linear_out.weight), ptr(pb.linear_out.bias)
        w.bias_norm_w, w.bias_norm_b = ptr(cb.predictor_bias_bias_norm.weight), ptr(cb.predictor_bias_bias_norm.bias)
        w.combine_w, w.combine_b = ptr(cb.predictor_bias_combine.weight), ptr(cb.predictor_bias_combine.bias)
        w.out_norm_w, w.out_norm_b = ptr(cb.predictor_bias_out_norm.weight), ptr(cb.predictor_bias_out_norm.bias)
        w.hw_enc_w, w.hw_enc_b = ptr(cb.hw_output_layer_enc.weight), ptr(cb.hw_output_layer_enc.bias)
        w.hw_v_w, w.hw_v_b = ptr(hb.linear_v.weight), ptr(hb.linear_v.bias)
        w.hw_o_w, w.hw_o_b = ptr(hb.linear_out.weight), ptr(hb.linear_out.bias)
        w.hw_norm_w, w.hw_norm_b = ptr(cb.hw_bias_norm.weight), ptr(cb.hw_bias_norm.bias)
        w.hw_out_w, w.hw_out_b = ptr(cb.hw_output_layer.weight), ptr(cb.hw_output_layer.bias)
        self._hw, self._hw_keep, self.max_ctx = w, hold, max_ctx
        nbytes = self._lib.wr_hotword_workspace_bytes(self._h, ctypes.byref(w), max_ctx)
        if nbytes == 0:
            raise RuntimeError("wr_hotword_workspace_bytes rejected the configuration")
        self._hw_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.wr_decoder_attach_hotword(self._h, ctypes.byref(w), max_ctx, _lib.ptr(self._hw_ws), nbytes,
                                                     _lib.current_stream(self.device))
        self._check(rc, "wr_decoder_attach_hotword")

    def greedy_hotword(self, enc_hot, enc_cold, enc_feat, enc_lens, hidden_hot, hidden_cold, n_steps: int = 64,
                       blank: int = 0, filter_on: bool = False) -> Tuple[List[List[int]], List[List[int]]]:
        """enc_* (N, T, D); hidden_* (n_ctx, D) -> (token lists, gate traces), one per stream."""
        eh, ec, ef = _f32(enc_hot), _f32(enc_cold), _f32(enc_feat)
        hh, hc = _f32(hidden_hot), _f32(hidden_cold)
        N, T, _ = eh.shape
        lens = enc_lens.to(device=self.device, dtype=torch.int32).reshape(-1).contiguous()
        cap = max(self.max_hyp, 1) + 1
        hyps = torch.empty(N, max(self.max_hyp, 1), dtype=torch.int32, device=self.device)
        hl = torch.empty(N, dtype=torch.int32, device=self.device)
        trace = torch.empty(N, cap, dtype=torch.int32, device=self.device)
        tl = torch.empty(N, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.wr_greedy_search_hotword(self._h, _lib.ptr(eh), _lib.ptr(ec), _lib.ptr(ef), _lib.ptr(lens),
                                                    _lib.ptr(hh), hh.shape[0], _lib.ptr(hc), hc.shape[0], N, T, int(n_steps),
                                                    int(blank), int(bool(filter_on)), _lib.ptr(hyps), _lib.ptr(hl),
                                                    _lib.ptr(trace), cap, _lib.ptr(tl), _lib.current_stream(self.device))
        self._check(rc, "wr_greedy_search_hotword")
        hl_c, tl_c, hy_c, tr_c = hl.cpu().tolist(), tl.cpu().tolist(), hyps.cpu(), trace.cpu()
        if max(hl_c, default=0) > self.max_hyp:
            raise RuntimeError(f"greedy search produced {max(hl_c)} tokens but the decoder was sized for {self.max_hyp}")
        return ([hy_c[i, :hl_c[i]].tolist() for i in range(N)], [tr_c[i, :tl_c[i]].tolist() for i in range(N)])


class HotwordDecoderCache(DecoderCache):
    """DecoderCache whose key also covers the hot-word module's weights."""

    def get_hw(self, predictor, joint, cb, lanes: int, tmax: int, max_hyp: int, n_ctx: int) -> HotwordDecoder:
        from .decoder import _fingerprint, _weights_key
        ps = list(predictor.parameters()) + list(joint.parameters()) + list(cb.parameters())
        key = _weights_key(ps)
        d = self._dec
        fp = _fingerprint(ps) if self._check_content else None
        same = (d is not None and key == self._key and not d.poisoned and
                (fp is None or (self._fp is not None and self._fp.device == fp.device and torch.equal(fp, self._fp))))
        if not same or lanes > d.max_lanes or tmax > d.tmax or max_hyp > d.max_hyp or n_ctx > d.max_ctx:
            grow = (lambda new, old: max(new, old)) if same else (lambda new, old: new)
            caps = (grow(lanes, d.max_lanes if d else 0), grow(tmax, d.tmax if d else 0),
                    grow(max_hyp, d.max_hyp if d else 0), grow(max(n_ctx, 8), d.max_ctx if d else 0))
            self._dec = None
            self._dec = HotwordDecoder(predictor, joint, cb, *caps)
            self._key, self._fp = key, fp
        return self._dec

    def __deepcopy__(self, memo):
        return HotwordDecoderCache(self._check_content)

    def __reduce__(self):
        return (HotwordDecoderCache, (self._check_content,))


def greedy_search_both_device(model, encoder_out: torch.Tensor, encoder_out_lens: torch.Tensor, context_list: torch.Tensor,
                              context_lengths: torch.Tensor, n_steps: int = 64, filter_on: bool = False
                              ) -> Tuple[List[List[int]], List[List[int]]]:
    """The loop of basic_greedy_search_both on the device for N streams sharing one hot-word list: encoder_out
    (N, T, D) -> (token lists, gate traces)."""
    cb = model.context_bias
    with torch.no_grad():
        hidden = cb.forward_bias_hidden(context_list, context_lengths)                                    # :327
        hidden_empty = cb.forward_bias_hidden(torch.zeros((1, 1), dtype=torch.int), context_lengths[0].unsqueeze(0))   # :328-333
        enc_plain = encoder_out.clone()
        enc_hot, enc_feat = cb.forward_encoder_bias(hidden, encoder_out)                                  # :335
        enc_cold, _ = cb.forward_encoder_bias(hidden_empty, enc_plain)                                    # :336
    N, T, _ = enc_hot.shape
    lens = torch.as_tensor(encoder_out_lens).reshape(-1)
    if lens.numel() == 1 and N > 1:
        lens = lens.expand(N)
    cache = getattr(model, "_hw_decoder_cache", None)
    if cache is None:
        cache = HotwordDecoderCache()
        try:
            model._hw_decoder_cache = cache
        except Exception:
            pass
    hh = hidden.reshape(-1, hidden.shape[-1])
    hc = hidden_empty.reshape(-1, hidden_empty.shape[-1])
    # a go-back lowers the per-frame emission counter by one (greedy_search.py:378), so a frame can carry more than n_steps
    # tokens: room for one extra token per frame and rewind
    dec = cache.get_hw(model.predictor, model.joint, cb, lanes=N, tmax=T, max_hyp=T * (n_steps + 2), n_ctx=max(hh.shape[0], hc.shape[0]))
    return dec.greedy_hotword(enc_hot, enc_cold, enc_feat, lens, hh, hc, n_steps=n_steps, blank=model.blank, filter_on=filter_on)
