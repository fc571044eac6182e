"""Transducer greedy search with the upstream core loop's semantics
(wenet/transducer/search/greedy_search copy.py:6-63),
for any number of independent streams at once, with the whole loop on the
device (`wr_greedy_search`).

The fork's hot-word variants (greedy_search.py:34-430: context gating and
"go-back" re-decoding around ContextBias) are not accelerated in this round
(SURVEY.md section 8f, item 3); with no hot words they reduce to this loop."""
from __future__ import annotations

from typing import List

import torch


def basic_greedy_search(model: torch.nn.Module, encoder_out: torch.Tensor, encoder_out_lens: torch.Tensor,
                        context_list: torch.Tensor = None, context_lengths: torch.Tensor = None,
                        n_steps: int = 64) -> List[List[int]]:
    """encoder_out (N, T, E), encoder_out_lens scalar or (N,) -> one token list per stream.

    `model` needs `.blank`, `.predictor` (RNNPredictor), `.joint` (TransducerJoint) and a
    `_decoder_cache` (wenet_celoss_amd.decoder.DecoderCache); Transducer provides all of them.
    context_* are accepted for signature compatibility and ignored (no hot words)."""
    N, T, _ = encoder_out.shape
    lens = torch.as_tensor(encoder_out_lens).reshape(-1)
    if lens.numel() == 1 and N > 1:
        lens = lens.expand(N)
    cache = getattr(model, "_decoder_cache", None)
    if cache is None:
        from ..decoder import DecoderCache
        cache = DecoderCache()
        try:
            model._decoder_cache = cache
        except Exception:
            pass
    dec = cache.get(model.predictor, model.joint, lanes=N, utts=N, tmax=T, max_hyp=T * n_steps, beam=1)
    return dec.greedy(encoder_out, lens, n_steps=n_steps, blank=model.blank)
