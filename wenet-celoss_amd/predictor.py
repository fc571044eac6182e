"""RNNPredictor with the reference's interface (wenet/transducer/predictor.py:58-200):
same constructor, same parameter names (embed / rnn / projection: reference
checkpoints load), same `forward`, `forward_step`, `init_state`,
`cache_to_batch`, `batch_to_cache`.

Training-time `forward` over a whole label sequence is the library LSTM
(MIOpen through torch.nn.LSTM), as SURVEY.md section 8a (row a8) scopes it.
`forward_step` -- the call the decoders make once per emitted token -- runs on
the HIP step kernels (`wr_predictor_step`)."""
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


class RNNPredictor(nn.Module):
    def __init__(self, voca_size: int, embed_size: int, output_size: int, embed_dropout: float, hidden_size: int,
                 num_layers: int, bias: bool = True, rnn_type: str = "lstm", dropout: float = 0.1) -> None:
        super().__init__()
        if rnn_type != "lstm" or not bias:
            raise NotImplementedError("wenet_celoss_amd.RNNPredictor implements the shipped configuration "
                                      "(rnn_type='lstm', bias=True)")
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
