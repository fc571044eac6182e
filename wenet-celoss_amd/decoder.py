"""Python owner of a `wr_decoder` handle (include/wr_api.h): packs the predictor /
joiner weights in the reference modules' layouts into `wr_transducer_weights`,
allocates the workspace through torch, and exposes greedy search, prefix beam
search and the predictor step."""
from __future__ import annotations

import ctypes
import os
from typing import List, Tuple

import torch

from . import _lib


def _f32(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


def _joint_activation_code(joint, default: str = "tanh") -> int:
    """wr_activation of a joiner (or stateless predictor) module: ours carries `act_code`; the reference's modules
    (joint.py:25, predictor.py:237,387) only the instantiated activation under their attribute `activatoin`."""
    code = getattr(joint, "act_code", None)
    if code is not None:
        return int(code)
    act = getattr(joint, "activatoin", None)
    if act is None:
        return _lib.ACTIVATIONS[default]
    by_type = {"Tanh": "tanh", "ReLU": "relu", "Hardtanh": "hardtanh", "SELU": "selu", "SiLU": "swish", "Swish": "swish",
               "GELU": "gelu"}
    name = by_type.get(type(act).__name__)
    if name is None or (name == "gelu" and getattr(act, "approximate", "none") != "none") or \
            (name == "hardtanh" and (act.min_val, act.max_val) != (-1.0, 1.0)):
        raise NotImplementedError(f"wenet_celoss_amd decoding: joiner activation {act!r} is not one of get_activation's")
    return _lib.ACTIVATIONS[name]


class DeviceDecoder:
    """One handle per (predictor, joint) pair and capacity.  Not thread-safe."""

    def __init__(self, predictor, joint, max_lanes: int, max_utt: int, tmax: int, max_hyp: int = 0, max_beam: int = 1):
        lib = _lib.load()
        rnn = getattr(predictor, "rnn", None)
        kind = 0 if rnn is not None else 1 if hasattr(predictor, "pos_embed") else 2 if hasattr(predictor, "conv") else -1
        if kind < 0 or (kind == 0 and not isinstance(rnn, torch.nn.LSTM)):
            raise NotImplementedError("wenet_celoss_amd decoding implements RNNPredictor with an LSTM (the shipped "
                                      "configuration), EmbeddingPredictor and ConvPredictor; got "
                                      f"{type(predictor).__name__}" + (f" / {type(rnn).__name__}" if rnn is not None else ""))
        dev = joint.ffn_out.weight.device
        if joint.enc_ffn is None or joint.pred_ffn is None:
            # prejoin_linear=False (joint.py:30-31 then requires enc == pred == join width): the step kernels multiply by
            # identity weights -- x * 1 + 0 * others is exact in fp32, so the joiner sees enc / pred unchanged
            J = joint.ffn_out.weight.shape[1]
            eye, zero = torch.eye(J, device=dev), torch.zeros(J, device=dev)
            enc_w = pred_w = eye
            enc_b = pred_b = zero
        else:
            enc_w, enc_b = joint.enc_ffn.weight, joint.enc_ffn.bias
            pred_w, pred_b = joint.pred_ffn.weight, joint.pred_ffn.bias
        post = getattr(joint, "post_ffn", None)
        if post is not None:
            # postjoin_linear (joint.py:66-67): post(e + p) = (W e + b) + W p -- a Linear distributes over the sum, so
            # it is folded into the two pre-join projections once (products formed in float64, rounded to fp32 once)
            with torch.no_grad():
                pw64 = post.weight.double()
                enc_b = (pw64 @ enc_b.double() + post.bias.double()).float()
                enc_w = (pw64 @ enc_w.double()).float()
                pred_b = (pw64 @ pred_b.double()).float()
                pred_w = (pw64 @ pred_w.double()).float()
        if dev.type != "cuda":
            raise RuntimeError("wenet_celoss_amd decoding: modules must live on a HIP device (this package has no CPU path)")
        self.device = dev
        self._keep: List[torch.Tensor] = []

        def hold(t):
            t = _f32(t)
            self._keep.append(t)
            return t.data_ptr()

        w = _lib.TransducerWeights()
        w.vocab_size = joint.ffn_out.weight.shape[0]
        w.enc_dim = enc_w.shape[1]
        w.embed_dim = predictor.embed.weight.shape[1]
        w.join_dim = joint.ffn_out.weight.shape[1]
        w.activation = _joint_activation_code(joint)
        w.embed = hold(predictor.embed.weight)
        w.embed_rows = predictor.embed.weight.shape[0]
        w.predictor_type = kind
        if kind == 0:
            w.pred_dim = predictor.projection.weight.shape[0]
            w.hidden = rnn.hidden_size
            w.n_layers = rnn.num_layers
            if rnn.num_layers > 4:
                raise NotImplementedError("at most 4 LSTM layers")
            for l in range(rnn.num_layers):
                w.w_ih[l] = hold(getattr(rnn, f"weight_ih_l{l}"))
                w.w_hh[l] = hold(getattr(rnn, f"weight_hh_l{l}"))
                if rnn.bias:
                    w.b_ih[l] = hold(getattr(rnn, f"bias_ih_l{l}"))
                    w.b_hh[l] = hold(getattr(rnn, f"bias_hh_l{l}"))
                else:                                        # RNNPredictor(bias=False): zero biases
                    w.b_ih[l] = w.b_hh[l] = hold(torch.zeros(4 * rnn.hidden_size, device=dev))
            w.proj_w, w.proj_b = hold(predictor.projection.weight), hold(predictor.projection.bias)
        else:
            # stateless predictors (predictor.py:203-481): the history of context_size - 1 token embeddings takes the
            # place of the LSTM state: that many "layers" of width embed_dim
            ctx = int(predictor.context_size)
            if not 2 <= ctx <= 5:
                raise NotImplementedError(f"history_size must be 1..4 (got {ctx - 1}): the token history travels in the "
                                          "decoder's four state slots")
            w.pred_dim = w.hidden = w.embed_dim
            w.n_layers = ctx - 1
            w.context_size = ctx
            w.pred_activation = _joint_activation_code(predictor, "swish" if kind == 1 else "relu")
            w.ln_eps = float(predictor.norm.eps)
            w.norm_w, w.norm_b = hold(predictor.norm.weight), hold(predictor.norm.bias)
            if kind == 1:
                w.n_head = int(predictor.num_heads)
                w.pos_w = hold(predictor.pos_embed.weight)
                w.ffn_w, w.ffn_b = hold(predictor.ffn.weight), hold(predictor.ffn.bias)
            else:
                w.conv_w = hold(predictor.conv.weight)
                w.conv_b = hold(predictor.conv.bias) if predictor.conv.bias is not None else None
        w.enc_ffn_w, w.enc_ffn_b = hold(enc_w), hold(enc_b)
        w.pred_ffn_w, w.pred_ffn_b = hold(pred_w), hold(pred_b)
        w.out_w, w.out_b = hold(joint.ffn_out.weight), hold(joint.ffn_out.bias)
        self._w = w
        self.dims = dict(V=w.vocab_size, E=w.enc_dim, P=w.pred_dim, D=w.embed_dim, H=w.hidden, L=w.n_layers, J=w.join_dim)
        self.max_lanes, self.max_utt, self.tmax, self.max_hyp, self.max_beam = max_lanes, max_utt, tmax, max_hyp, max_beam
        nbytes = lib.wr_decoder_workspace_bytes(ctypes.byref(w), max_lanes, max_utt, tmax, max_hyp, max_beam)
        if nbytes == 0:
            raise RuntimeError("wr_decoder_workspace_bytes rejected the configuration")
        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        handle = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = lib.wr_decoder_create(ctypes.byref(w), max_lanes, max_utt, tmax, max_hyp, max_beam, _lib.ptr(self._ws),
                                       nbytes, _lib.current_stream(dev), ctypes.byref(handle))
        _lib.check(rc, "wr_decoder_create")
        self._h = handle
        self._lib = lib
        look = os.environ.get("WR_GREEDY_LOOKAHEAD")      # unset: the library default (adaptive)
        if look is not None:
            self.set_lookahead(int(look))

    def _check(self, rc: int, what: str) -> None:
        """A failed search leaves the handle's lane state undefined: mark it so that DecoderCache rebuilds it."""
        if rc != 0:
            self.poisoned = True
        _lib.check(rc, what)

    poisoned = False

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.wr_decoder_destroy(h)
            except Exception:
                pass

    def set_graph(self, enable: bool) -> None:
        _lib.check(self._lib.wr_decoder_set_graph(self._h, int(enable)), "wr_decoder_set_graph")

    def set_lookahead(self, frames: int) -> None:
        """Greedy search: encoder frames evaluated per micro-step (1..4, 0 = adaptive); token sequences do not
        depend on it."""
        _lib.check(self._lib.wr_decoder_set_lookahead(self._h, int(frames)), "wr_decoder_set_lookahead")

    def greedy(self, encoder_out: torch.Tensor, encoder_out_lens: torch.Tensor, n_steps: int = 64, blank: int = 0
               ) -> List[List[int]]:
        """encoder_out (N, T, E); returns one token list per stream."""
        enc = _f32(encoder_out)
        N, T, _ = enc.shape
        lens = encoder_out_lens.to(device=self.device, dtype=torch.int32).reshape(-1).contiguous()
        hyps = torch.empty(N, max(self.max_hyp, 1), dtype=torch.int32, device=self.device)
        hl = torch.empty(N, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.wr_greedy_search(self._h, _lib.ptr(enc), _lib.ptr(lens), N, T, int(n_steps), int(blank),
                                            _lib.ptr(hyps), _lib.ptr(hl), _lib.current_stream(self.device))
        self._check(rc, "wr_greedy_search")
        hl_c, hy_c = hl.cpu().tolist(), hyps.cpu()
        if max(hl_c, default=0) > self.max_hyp:
            raise RuntimeError(f"greedy search produced {max(hl_c)} tokens but the decoder was sized for {self.max_hyp}")
        return [hy_c[i, :hl_c[i]].tolist() for i in range(N)]

    def greedy_chunk(self, encoder_chunk: torch.Tensor, chunk_lens: torch.Tensor, n_steps: int = 64, blank: int = 0,
                     reset: bool = False, reference_new_cache: bool = True) -> List[List[int]]:
        """Streaming greedy search: tokens emitted in this chunk for each stream (state carries over)."""
        enc = _f32(encoder_chunk)
        N, T, _ = enc.shape
        lens = chunk_lens.to(device=self.device, dtype=torch.int32).reshape(-1).contiguous()
        hyps = torch.empty(N, max(self.max_hyp, 1), dtype=torch.int32, device=self.device)
        hl = torch.empty(N, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.wr_greedy_search_chunk(self._h, _lib.ptr(enc), _lib.ptr(lens), N, T, int(n_steps), int(blank),
                                                  int(reset), int(reference_new_cache), _lib.ptr(hyps), _lib.ptr(hl),
                                                  _lib.current_stream(self.device))
        self._check(rc, "wr_greedy_search_chunk")
        hl_c, hy_c = hl.cpu().tolist(), hyps.cpu()
        if max(hl_c, default=0) > self.max_hyp:
            raise RuntimeError(f"greedy search produced {max(hl_c)} tokens but the decoder was sized for {self.max_hyp}")
        return [hy_c[i, :hl_c[i]].tolist() for i in range(N)]

    def prefix_beam(self, encoder_out: torch.Tensor, encoder_out_lens: torch.Tensor, ctc_logp: torch.Tensor,
                    beam_size: int, ctc_weight: float, transducer_weight: float, blank: int = 0
                    ) -> List[List[Tuple[List[int], float]]]:
        """encoder_out (B, T, E), ctc_logp (B, T, V).  Per utterance: [(hyp incl. seed blank, score)] best first."""
        enc = _f32(encoder_out)
        ctc = _f32(ctc_logp)
        B, T, _ = enc.shape
        lens = encoder_out_lens.to(device=self.device, dtype=torch.int32).reshape(-1).contiguous()
        lmax = self.tmax + 1
        hyps = torch.empty(B, beam_size, lmax, dtype=torch.int32, device=self.device)
        hl = torch.empty(B, beam_size, dtype=torch.int32, device=self.device)
        sc = torch.empty(B, beam_size, dtype=torch.float64, device=self.device)
        nh = torch.empty(B, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.wr_prefix_beam_search(self._h, _lib.ptr(enc), _lib.ptr(lens), _lib.ptr(ctc), B, T, int(beam_size),
                                                 float(ctc_weight), float(transducer_weight), int(blank), _lib.ptr(hyps),
                                                 _lib.ptr(hl), _lib.ptr(sc), _lib.ptr(nh), _lib.current_stream(self.device))
        self._check(rc, "wr_prefix_beam_search")
        hyps, hl, sc, nh = hyps.cpu(), hl.cpu().tolist(), sc.cpu().tolist(), nh.cpu().tolist()
        out = []
        for b in range(B):
            out.append([(hyps[b, e, :hl[b][e]].tolist(), sc[b][e]) for e in range(nh[b])])
        return out

    def predictor_step(self, tokens: torch.Tensor, cache_h: torch.Tensor, cache_c: torch.Tensor):
        """tokens (N,), cache (L, N, H) -> out (N, P), new_h, new_c (L, N, H)."""
        N = tokens.numel()
        tok = tokens.to(device=self.device, dtype=torch.int32).reshape(-1).contiguous()
        ch, cc = _f32(cache_h), _f32(cache_c)
        out = torch.empty(N, self.dims["P"], dtype=torch.float32, device=self.device)
        nh, nc = torch.empty_like(ch), torch.empty_like(cc)
        with torch.cuda.device(self.device):
            rc = self._lib.wr_predictor_step(self._h, _lib.ptr(tok), _lib.ptr(ch), _lib.ptr(cc), N, _lib.ptr(out),
                                             _lib.ptr(nh), _lib.ptr(nc), _lib.current_stream(self.device))
        self._check(rc, "wr_predictor_step")
        return out, nh, nc


def _params(predictor, joint):
    return list(predictor.parameters()) + list(joint.parameters())


def _weights_key(ps):
    return tuple((p.data_ptr(), p._version) for p in ps)


def _fingerprint(ps):
    """Content fingerprint of the weights (L2 norm and plain sum of every parameter, computed on the device):
    `p.data.mul_()`-style edits (EMA, weight averaging) leave data_ptr and _version unchanged, and the handle holds
    re-laid copies of the weights, so identity alone would let it decode with stale weights."""
    with torch.no_grad():
        flat = [p.detach() for p in ps]
        return torch.stack(list(torch._foreach_norm(flat)) + [t.sum(dtype=torch.float32) for t in flat])


class DecoderCache:
    """Keeps a DeviceDecoder alive across calls and rebuilds it when the modules' weights change (identity, version
    counter or -- with `check_content` -- the content fingerprint above) or a call needs more capacity.
    `invalidate()` forces a rebuild.  Copies and pickles of a module that owns a cache get an empty cache (the
    native handle is rebuilt lazily)."""

    def __init__(self, check_content: bool = True):
        self._dec = None
        self._key = None
        self._fp = None
        self._check_content = check_content

    def invalidate(self) -> None:
        self._dec, self._key, self._fp = None, None, None

    def __deepcopy__(self, memo):
        return DecoderCache(self._check_content)

    def __reduce__(self):
        return (DecoderCache, (self._check_content,))

    def discard(self, dec) -> None:
        """Drop `dec` if it is the cached handle (called when a search on it raised: its lane state is undefined)."""
        if self._dec is dec:
            self.invalidate()

    def get(self, predictor, joint, lanes: int, utts: int, tmax: int, max_hyp: int, beam: int) -> DeviceDecoder:
        ps = _params(predictor, joint)
        key = _weights_key(ps)
        d = self._dec
        same = d is not None and key == self._key and not d.poisoned
        fp = None
        if same and self._check_content:
            fp = _fingerprint(ps)
            same = self._fp is not None and self._fp.device == fp.device and torch.equal(fp, self._fp)
        if not same or lanes > d.max_lanes or utts > d.max_utt or tmax > d.tmax or max_hyp > d.max_hyp or beam > d.max_beam:
            grow = (lambda new, old: max(new, old)) if same else (lambda new, old: new)
            caps = (grow(lanes, d.max_lanes if d else 0), grow(utts, d.max_utt if d else 0), grow(tmax, d.tmax if d else 0),
                    grow(max_hyp, d.max_hyp if d else 0), grow(beam, d.max_beam if d else 1))
            self._dec = None            # release the old handle before building the new one
            self._dec = DeviceDecoder(predictor, joint, *caps)
            self._key = key
            self._fp = (fp if fp is not None else _fingerprint(ps)) if self._check_content else None
        return self._dec
