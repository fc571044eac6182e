"""The predictors with the reference's interface (wenet/transducer/predictor.py): RNNPredictor (:58-200, the
shipped one), EmbeddingPredictor (:203-372) and ConvPredictor (:375-481) -- same constructors, same parameter
names (reference checkpoints load), same `forward`, `forward_step`, `init_state`, `cache_to_batch`,
`batch_to_cache`.

Training-time `forward` over a whole label sequence is the library LSTM
(MIOpen through torch.nn.LSTM), as SURVEY.md section 8a (row a8) scopes it.
`forward_step` -- the call the decoders make once per emitted token -- runs on
the HIP step kernels (`wr_predictor_step`).  The two stateless predictors keep the embeddings of the last
`history_size` tokens as their state; on the device that history travels in the slots of the LSTM state
(include/wr_api.h, `predictor_type`), so greedy / beam search and the step API take them unchanged."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn

from .decoder import DecoderCache


def ApplyPadding(input, padding, pad_value) -> torch.Tensor:
    """predictor.py:9-15"""
    return padding * pad_value + input * (1 - padding)


class _StepJoint(nn.Module):
    """Minimal joiner stand-in so that a predictor can own a decoder handle on its own."""

    def __init__(self, p_dim, device):
        super().__init__()
        self.enc_ffn = nn.Linear(1, 4).to(device)
        self.pred_ffn = nn.Linear(p_dim, 4).to(device)
        self.ffn_out = nn.Linear(4, 2).to(device)


class PredictorBase(nn.Module):
    """Interface of the predictors (predictor.py:16-55): state handling for the decoders plus the two forwards."""

    def _abstract(self, *_):
        raise NotImplementedError("this is a base predictor")

    init_state = batch_to_cache = cache_to_batch = forward = forward_step = _abstract


class RNNPredictor(PredictorBase):
    def __init__(self, voca_size: int, embed_size: int, output_size: int, embed_dropout: float, hidden_size: int,
                 num_layers: int, bias: bool = True, rnn_type: str = "lstm", dropout: float = 0.1) -> None:
        super().__init__()
        if rnn_type != "lstm":
            # the reference's own forward / forward_step unpack an (m, c) state tuple (predictor.py:113,190), which
            # torch.nn.GRU / RNN do not return: only 'lstm' runs there either
            raise NotImplementedError("wenet_celoss_amd.RNNPredictor implements rnn_type='lstm' (the reference's forward "
                                      "unpacks an LSTM state tuple, so 'gru' / 'rnn' do not run there either)")
        self.n_layers = num_layers
        self.hidden_size = hidden_size
        self.embed = nn.Embedding(voca_size, embed_size)
        self.dropout = nn.Dropout(embed_dropout)
        self.rnn = nn.LSTM(input_size=embed_size, hidden_size=hidden_size, num_layers=num_layers, bias=bias,
                           batch_first=True, dropout=dropout)
        self.projection = nn.Linear(hidden_size, output_size)
        self._step_cache = DecoderCache(check_content=False)   # per-token calls: identity + version only;
                                                               # after editing weights through .data call invalidate()
        self._step_joint = None

    def forward(self, input: torch.Tensor, cache: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """input (B, U) -> (B, U, output_size)   (predictor.py:88-121)"""
        embed = self.dropout(self.embed(input))
        if cache is None:
            state = self.init_state(batch_size=input.size(0), device=input.device)
            states = (state[0], state[1])
        else:
            assert len(cache) == 2
            states = (cache[0], cache[1])
        out, _ = self.rnn(embed, states)
        return self.projection(out)

    def batch_to_cache(self, cache: List[torch.Tensor]) -> List[List[torch.Tensor]]:
        """[state_m (L, bs, H), state_c] -> [[m_1, c_1], [m_2, c_2], ...]   (predictor.py:123-143)"""
        assert len(cache) == 2
        state_ms, state_cs = cache
        assert state_ms.size(1) == state_cs.size(1)
        return [[m, c] for m, c in zip(torch.split(state_ms, 1, dim=1), torch.split(state_cs, 1, dim=1))]

    def cache_to_batch(self, cache: List[List[torch.Tensor]]) -> List[torch.Tensor]:
        """predictor.py:145-158"""
        return [torch.cat([s[0] for s in cache], dim=1), torch.cat([s[1] for s in cache], dim=1)]

    def init_state(self, batch_size: int, device: torch.device, method: str = "zero") -> List[torch.Tensor]:
        assert batch_size > 0
        _ = method
        return [torch.zeros(self.n_layers, batch_size, self.hidden_size, device=device),
                torch.zeros(self.n_layers, batch_size, self.hidden_size, device=device)]

    def forward_step(self, input: torch.Tensor, padding: torch.Tensor, cache: List[torch.Tensor]
                     ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """input (N, 1) tokens, padding (N, 1) (1 keeps the old state), cache [m (L,N,H), c (L,N,H)]
        -> (out (N, 1, output_size), [m, c])   (predictor.py:179-200; eval semantics: dropout off).
        Eager calls always run the HIP step kernels.  Only inside a TorchScript artefact (torch.jit.script(model),
        wenet/bin/train.py:203-205, export_jit.py) -- which cannot reach a ctypes library -- the step is the plain
        module graph `_export_step`, so that the exported file keeps the reference's runtime contract."""
        if torch.jit.is_scripting():
            return self._export_step(input, padding, cache)
        else:
            return self._hip_step(input, padding, cache)

    def _export_step(self, input: torch.Tensor, padding: torch.Tensor, cache: List[torch.Tensor]
                     ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """TorchScript-export body of forward_step (never executed in eager mode): the reference's statements,
        predictor.py:179-200."""
        assert len(cache) == 2
        state_m, state_c = cache[0], cache[1]
        embed = self.dropout(self.embed(input))
        out, (m, c) = self.rnn(embed, (state_m, state_c))
        out = self.projection(out)
        m = ApplyPadding(m, padding.unsqueeze(0), state_m)
        c = ApplyPadding(c, padding.unsqueeze(0), state_c)
        return out, [m, c]

    @torch.jit.unused
    def _hip_step(self, input: torch.Tensor, padding: torch.Tensor, cache: List[torch.Tensor]
                  ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        assert len(cache) == 2
        state_m, state_c = cache
        N = input.size(0)
        if self._step_joint is None or self._step_joint[0].ffn_out.weight.device != self.embed.weight.device:
            self._step_joint = [_StepJoint(self.projection.weight.shape[0], self.embed.weight.device)]
        dec = self._step_cache.get(self, self._step_joint[0], lanes=N, utts=1, tmax=1, max_hyp=0, beam=1)
        out, m, c = dec.predictor_step(input.reshape(-1), state_m, state_c)
        pad = padding.to(out.device).reshape(1, N, 1)
        m = ApplyPadding(m, pad, state_m)
        c = ApplyPadding(c, pad, state_c)
        return out.unsqueeze(1), [m, c]


_ACTIVATIONS = {"hardtanh": nn.Hardtanh, "tanh": nn.Tanh, "relu": nn.ReLU, "selu": nn.SELU, "swish": nn.SiLU, "gelu": nn.GELU}


class _HistoryPredictor(PredictorBase):
    """What EmbeddingPredictor and ConvPredictor share: the state is [history (N, context_size - 1, embed)], the
    training forward slides a window of context_size embeddings over the zero-prefixed label sequence, and
    `forward_step` is one window on the device."""

    def _init_common(self, voca_size, embed_size, embed_dropout, history_size, activation, layer_norm_epsilon):
        from . import _lib
        assert history_size >= 0
        self.embed_size = embed_size
        self.context_size = history_size + 1
        self.embed = nn.Embedding(voca_size, embed_size)
        self.embed_dropout = nn.Dropout(p=embed_dropout)
        self.norm = nn.LayerNorm(embed_size, eps=layer_norm_epsilon)
        self.activatoin = _ACTIVATIONS[activation]()                  # the reference's attribute name (sic)
        self.act_code = _lib.ACTIVATIONS[activation]
        self._step_cache = DecoderCache(check_content=False)
        self._step_joint = None

    def init_state(self, batch_size: int, device: torch.device, method: str = "zero") -> List[torch.Tensor]:
        assert batch_size > 0
        _ = method
        return [torch.zeros(batch_size, self.context_size - 1, self.embed_size, device=device)]

    def batch_to_cache(self, cache: List[torch.Tensor]) -> List[List[torch.Tensor]]:
        """[history (bs, ...)] -> [[history_1], [history_2], ...]   (predictor.py:254-268, :405-419)"""
        assert len(cache) == 1
        return [[h] for h in torch.split(cache[0], 1, dim=0)]

    def cache_to_batch(self, cache: List[List[torch.Tensor]]) -> List[torch.Tensor]:
        return [torch.cat([h[0] for h in cache], dim=0)]

    def _windows(self, input: torch.Tensor, cache: Optional[List[torch.Tensor]]) -> torch.Tensor:
        """(B, U) tokens -> (B, U, context_size, embed): window u holds the embeddings of positions u - ctx + 1 .. u."""
        x = self.embed_dropout(self.embed(input))
        if cache is None:
            prefix = self.init_state(x.size(0), device=x.device)[0]
        else:
            assert len(cache) == 1
            prefix = cache[0]
        x = torch.cat((prefix, x), dim=1)
        return x.unfold(1, self.context_size, 1).transpose(2, 3)

    def _combine(self, win: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def forward(self, input: torch.Tensor, cache: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """Training-time forward over a label sequence (library ops; predictor.py:283-323, :430-453)."""
        return self.activatoin(self.norm(self._combine(self._windows(input, cache))))

    def forward_step(self, input: torch.Tensor, padding: torch.Tensor, cache: List[torch.Tensor]
                     ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """input (N, 1) tokens, cache [history (N, ctx - 1, embed)] -> (out (N, 1, embed), [new history]).  The reference
        does not apply `padding` to the new history (its TODO at predictor.py:370, :480); neither does this."""
        if torch.jit.is_scripting():
            return self._export_step(input, padding, cache)
        else:
            return self._hip_step(input, padding, cache)

    def _export_step(self, input: torch.Tensor, padding: torch.Tensor, cache: List[torch.Tensor]
                     ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """TorchScript-export body (a scripted artefact cannot reach the ctypes library); never run in eager mode."""
        assert input.size(1) == 1 and len(cache) == 1
        ctx = torch.cat((cache[0], self.embed_dropout(self.embed(input))), dim=1)
        out = self.activatoin(self.norm(self._combine(ctx.unsqueeze(1))))
        return out, [ctx[:, 1:, :]]

    @torch.jit.unused
    def _hip_step(self, input: torch.Tensor, padding: torch.Tensor, cache: List[torch.Tensor]
                  ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        assert input.size(1) == 1 and len(cache) == 1
        history = cache[0]
        assert history.size(1) == self.context_size - 1
        N = input.size(0)
        dev = self.embed.weight.device
        if self._step_joint is None or self._step_joint[0].ffn_out.weight.device != dev:
            self._step_joint = [_StepJoint(self.embed_size, dev)]
        dec = self._step_cache.get(self, self._step_joint[0], lanes=N, utts=1, tmax=1, max_hyp=0, beam=1)
        slots = history.transpose(0, 1).contiguous()               # the device keeps one "layer" per history slot
        out, new_slots, _ = dec.predictor_step(input.reshape(-1), slots, torch.zeros_like(slots))
        return out.unsqueeze(1), [new_slots.transpose(0, 1).contiguous()]


class EmbeddingPredictor(_HistoryPredictor):
    """predictor.py:203-372 (https://arxiv.org/pdf/2109.07513.pdf): embed -> multi-head positional weighting of the
    last context_size embeddings -> ffn -> LayerNorm -> activation."""

    def __init__(self, voca_size: int, embed_size: int, embed_dropout: float, n_head: int, history_size: int = 2,
                 activation: str = "swish", bias: bool = False, layer_norm_epsilon: float = 1e-5) -> None:
        super().__init__()
        self.num_heads = n_head
        self._init_common(voca_size, embed_size, embed_dropout, history_size, activation, layer_norm_epsilon)
        self.pos_embed = nn.Linear(embed_size * self.context_size, self.num_heads, bias=bias)   # only .weight is used
        self.ffn = nn.Linear(embed_size, embed_size)

    def _combine(self, win: torch.Tensor) -> torch.Tensor:
        # pos[h, e, c]; weight[.., h, c] = <window[c], pos[h, :, c]>; output = mean over (h, c) of weight * window[c]
        pos = self.pos_embed.weight.view(self.num_heads, self.embed_size, self.context_size)
        weight = torch.einsum("buce,hec->buhc", win, pos)
        out = torch.einsum("buhc,buce->bue", weight, win) / (self.num_heads * self.context_size)
        return self.ffn(out)


class ConvPredictor(_HistoryPredictor):
    """predictor.py:375-481: embed -> depthwise Conv1d over the last context_size embeddings -> LayerNorm -> activation."""

    def __init__(self, voca_size: int, embed_size: int, embed_dropout: float, history_size: int = 2,
                 activation: str = "relu", bias: bool = False, layer_norm_epsilon: float = 1e-5) -> None:
        super().__init__()
        self._init_common(voca_size, embed_size, embed_dropout, history_size, activation, layer_norm_epsilon)
        self.conv = nn.Conv1d(in_channels=embed_size, out_channels=embed_size, kernel_size=self.context_size, padding=0,
                              groups=embed_size, bias=bias)

    def init_state(self, batch_size: int, device: torch.device, method: str = "zero") -> List[torch.Tensor]:
        assert batch_size > 0
        assert method == "zero"                     # predictor.py:393 (the embedding predictor ignores `method`)
        return [torch.zeros(batch_size, self.context_size - 1, self.embed_size, device=device)]

    def _combine(self, win: torch.Tensor) -> torch.Tensor:
        out = torch.einsum("buce,ec->bue", win, self.conv.weight[:, 0, :])      # one filter tap per (channel, position)
        bias = self.conv.bias
        if bias is not None:
            out = out + bias
        return out
