"""Transducer prefix beam search with the reference's interface
(wenet/transducer/search/prefix_beam_search.py:7-148): `Sequence`,
`PrefixBeamSearch(encoder, predictor, joint, ctc, blank)` and
`prefix_beam_search(speech, speech_lengths, decoding_chunk_size, beam_size,
num_decoding_left_chunks, simulate_streaming, ctc_weight, transducer_weight)
-> (beam: List[Sequence], encoder_out)`.

One expansion per hypothesis per frame, CTC/transducer score mixture, top-k,
prefix fusion with float64 log_add and the stable prune all run on the device
(`wr_prefix_beam_search`); unlike the reference (batch 1 only, :54-58) any
number of utterances can be searched together via `prefix_beam_search_batch`."""
from __future__ import annotations

from typing import List

import torch

from ..decoder import DecoderCache


class Sequence:
    __slots__ = {"hyp", "score", "cache"}

    def __init__(self, hyp, score, cache):
        self.hyp = hyp
        self.score = score
        self.cache = cache


class PrefixBeamSearch:
    def __init__(self, encoder, predictor, joint, ctc, blank):
        self.encoder = encoder
        self.predictor = predictor
        self.joint = joint
        self.ctc = ctc
        self.blank = blank
        self._decoder_cache = DecoderCache()

    def search_encoded(self, encoder_out: torch.Tensor, encoder_out_lens: torch.Tensor, beam_size: int = 5,
                       ctc_weight: float = 0.3, transducer_weight: float = 0.7) -> List[List[Sequence]]:
        """encoder_out (B, T, E) already computed -> per utterance the pruned beam (best first)."""
        B, T, _ = encoder_out.shape
        with torch.no_grad():
            ctc_logp = self.ctc.log_softmax(encoder_out)                 # (B, T, V), ctc.py:66-75
        dec = self._decoder_cache.get(self.predictor, self.joint, lanes=B * beam_size, utts=B, tmax=T, max_hyp=0,
                                      beam=beam_size)
        res = dec.prefix_beam(encoder_out, encoder_out_lens, ctc_logp, beam_size, ctc_weight, transducer_weight, self.blank)
        return [[Sequence(hyp=h, score=s, cache=None) for h, s in utt] for utt in res]

    def prefix_beam_search(self, speech: torch.Tensor, speech_lengths: torch.Tensor, decoding_chunk_size: int = -1,
                           beam_size: int = 5, num_decoding_left_chunks: int = -1, simulate_streaming: bool = False,
                           ctc_weight: float = 0.3, transducer_weight: float = 0.7):
        assert speech.shape[0] == speech_lengths.shape[0]
        assert decoding_chunk_size != 0
        assert speech.shape[0] == 1
        encoder_out, _ = self.encoder(speech, speech_lengths, decoding_chunk_size, num_decoding_left_chunks)
        lens = torch.tensor([encoder_out.size(1)], dtype=torch.int32)
        beam = self.search_encoded(encoder_out, lens, beam_size, ctc_weight, transducer_weight)[0]
        return beam, encoder_out

    def prefix_beam_search_batch(self, speech, speech_lengths, decoding_chunk_size=-1, beam_size=5,
                                 num_decoding_left_chunks=-1, ctc_weight=0.3, transducer_weight=0.7):
        """Extension: B utterances at once.  Returns (List[List[Sequence]], encoder_out)."""
        encoder_out, mask = self.encoder(speech, speech_lengths, decoding_chunk_size, num_decoding_left_chunks)
        lens = mask.squeeze(1).sum(1).to(torch.int32)
        return self.search_encoded(encoder_out, lens, beam_size, ctc_weight, transducer_weight), encoder_out
