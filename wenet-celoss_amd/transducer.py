"""Transducer model head with the reference's interface
(wenet/transducer/transducer.py:20-629): same constructor keywords, same
`forward(speech, speech_lengths, text, text_lengths, context_list,
context_lengths, hw_label)` returning the dict
{loss, loss_att, loss_ctc, loss_rnnt, hw_loss}, same `greedy_search`,
`beam_search`, `transducer_attention_rescoring`, `_cal_transducer_score` and
step exports -- so wenet/bin/train.py and wenet/bin/recognize.py can drive it.

What runs where
  joiner logits      wenet_celoss_amd.TransducerJoint  (MFMA HIP kernels)
  RNN-T loss + grad  wenet_celoss_amd.rnnt_loss        (HIP, replaces torchaudio, :142-147 / :296-301)
  CTC loss + grad    wenet_celoss_amd.CTC              (HIP, replaces nn.CTCLoss)
  greedy / beam      wenet_celoss_amd.search.*         (HIP decode kernels under hipGraph)
  encoder, attention decoder, ContextBias: whatever modules the caller passes
  (stock PyTorch-ROCm; out of scope).

`context_bias` may be None (no hot words); when a ContextBias module is given it
is called exactly where the reference calls it in `forward`, and `greedy_search`
dispatches on `loss_mode` to the fork's hot-word variants (host-driven control flow
over the HIP step kernels).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn
from torch.nn.utils.rnn import pad_sequence

from .common import IGNORE_ID, LabelSmoothingLoss, add_blank, add_sos_eos, end_blank, reverse_pad_list
from .decoder import DecoderCache
from .fused import joint_rnnt_loss, plan_buckets
from .joint import TransducerJoint, _resolve_precision
from .rnnt_loss import rnnt_loss
from .search.greedy_search import basic_greedy_search, basic_greedy_search_both, basic_greedy_search_hw
from .search.prefix_beam_search import PrefixBeamSearch


def check_limits(text: torch.Tensor, text_lengths: torch.Tensor, with_ctc: bool) -> None:
    """Shape limits of the loss kernels (INTEGRATION.md "Shape limits"), checked on the padded label width before
    anything is launched, with a message that names the longest utterance.  No device sync unless a limit is hit."""
    width = int(text.shape[1])
    limit = 511 if with_ctc else 1023
    if width <= limit:
        return
    i = int(text_lengths.argmax())
    what = ("the CTC loss kernels take label sequences padded to at most 511" if with_ctc else
            "the RNN-T loss kernels take at most 1023 labels (U + 1 <= 1024 lattice columns)")
    raise RuntimeError(f"wenet_celoss_amd.Transducer: the label batch is padded to {width} (longest: utterance {i} with "
                       f"{int(text_lengths[i])} labels); {what} -- lower filter_conf.token_max_length")


class Transducer(nn.Module):
    """Transducer-ctc-attention hybrid Encoder-Predictor-Decoder model (transducer.py:20)."""

    def __init__(self, vocab_size: int, blank: int, encoder: nn.Module, predictor: nn.Module, joint: nn.Module,
                 attention_decoder: Optional[nn.Module] = None, ctc: Optional[nn.Module] = None,
                 context_bias: Optional[nn.Module] = None, ctc_weight: float = 0, ignore_id: int = IGNORE_ID,
                 reverse_weight: float = 0.0, lsm_weight: float = 0.0, length_normalized_loss: bool = False,
                 transducer_weight: float = 1.0, attention_weight: float = 0.0, hw_weight: float = 0.4,
                 loss_mode: str = "both") -> None:
        assert attention_weight + ctc_weight + transducer_weight == 1.0          # transducer.py:46 (kept as is)
        super().__init__()
        # ASRModel part (asr_model.py:38-70): sos/eos are the last class
        self.sos = vocab_size - 1
        self.eos = vocab_size - 1
        self.vocab_size = vocab_size
        self.ignore_id = ignore_id
        self.ctc_weight = ctc_weight
        self.reverse_weight = reverse_weight
        self.encoder = encoder
        self.decoder = attention_decoder
        self.ctc = ctc

        self.blank = blank
        self.transducer_weight = transducer_weight
        self.attention_decoder_weight = 1 - self.transducer_weight - self.ctc_weight
        self.context_bias = context_bias
        self.predictor = predictor
        self.joint = joint
        self.bs = None
        self.hw_weight = hw_weight
        self.loss_mode = loss_mode
        self.hw_criterion = nn.CrossEntropyLoss()
        if attention_decoder is not None:
            self.criterion_att = LabelSmoothingLoss(size=vocab_size, padding_idx=ignore_id, smoothing=lsm_weight,
                                                    normalize_length=length_normalized_loss)
        # encoder.embed.{subsampling_rate, right_context} as the exports below report them (asr_model.py:542-554); -1 when
        # the encoder module has no `embed` (the reference would fail to script such a model)
        embed = getattr(encoder, "embed", None)
        self._subsampling_rate: int = int(getattr(embed, "subsampling_rate", -1))
        self._right_context: int = int(getattr(embed, "right_context", -1))
        self._decoder_cache = DecoderCache()
        # joiner + RNN-T loss as one autograd node (fused.py): pass 1 of the loss rides on the joiner's epilogue and the
        # logits never leave the node (half the footprint).  WR_FUSED_LOSS=0 (or this attribute) selects the two separate ops.
        self.fused_loss = os.environ.get("WR_FUSED_LOSS", "1") != "0"

    # ------------------------------------------------------------- training --
    def compute_loss(self, encoder_out: torch.Tensor, encoder_out_lens: torch.Tensor, predictor_out: torch.Tensor,
                     text: torch.Tensor, text_lengths: torch.Tensor, skip_padding: bool = False
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
        """The loss block of the reference forward (transducer.py:131-147): joiner -> int32 label prep -> RNN-T loss.
        Returns (joint_out, loss_rnnt).  With `fused_loss` the two run as one autograd node and joint_out is
        None (the logits never leave it); otherwise `skip_padding=True` (extension) lets the joiner skip lattice cells
        that the loss never reads."""
        rnnt_text = text.to(torch.int64)
        rnnt_text = torch.where(rnnt_text == self.ignore_id, 0, rnnt_text).to(torch.int32)
        rnnt_text_lengths = text_lengths.to(torch.int32)
        encoder_out_lens = encoder_out_lens.to(torch.int32)
        if self._can_fuse_loss():
            jt = self.joint
            ep, pp = jt.pre_activation(encoder_out, predictor_out)
            loss = joint_rnnt_loss(ep, pp, jt.ffn_out.weight, jt.ffn_out.bias,
                                   rnnt_text, encoder_out_lens, rnnt_text_lengths, blank=self.blank, reduction="mean",
                                   precision=jt.precision, activation=jt.activation)
            return None, loss
        if self.fused_loss and isinstance(self.joint, TransducerJoint):
            # the AMP single-term joiner keeps 16-bit logits (two ops), but a ragged batch is still cut into
            # label-length groups, each padded to its own maxima (fused.plan_buckets): same costs, fewer padded cells
            lens = torch.stack([encoder_out_lens, rnnt_text_lengths]).cpu()
            groups = plan_buckets(lens[0].tolist(), lens[1].tolist(), max_buckets=int(os.environ.get("WR_FUSED_BUCKETS", "4")))
            if groups is not None:
                dev = encoder_out.device
                parts, index = [], []
                for g in groups:
                    idx = torch.tensor(g, device=dev)
                    tg, ug = int(lens[0][g].max()), int(lens[1][g].max())
                    ll, tl = encoder_out_lens[idx].contiguous(), rnnt_text_lengths[idx].contiguous()
                    logits = self.joint(encoder_out[idx, :tg], predictor_out[idx, :ug + 1], ll, tl)
                    parts.append(rnnt_loss(logits, rnnt_text[idx, :ug].contiguous(), ll, tl, blank=self.blank, reduction="none",
                                           inplace_grad=True))
                    index.append(idx)
                return None, torch.cat(parts).float().mean()
        if skip_padding:
            joint_out = self.joint(encoder_out, predictor_out, encoder_out_lens, rnnt_text_lengths)
        else:
            joint_out = self.joint(encoder_out, predictor_out)
        loss = rnnt_loss(joint_out, rnnt_text.contiguous(), encoder_out_lens.contiguous(),
                         rnnt_text_lengths.contiguous(), blank=self.blank, reduction="mean")
        return joint_out, loss

    def _can_fuse_loss(self) -> bool:
        jt = self.joint
        return (self.fused_loss and isinstance(jt, TransducerJoint)
                and _resolve_precision(jt.precision) != "bf16")        # the AMP single-term mode keeps 16-bit logits

    @torch.jit.unused      # wenet/bin/train.py:203-205 scripts the model as an export smoke test; the HIP-backed forward
    def forward(self, speech: torch.Tensor, speech_lengths: torch.Tensor, text: torch.Tensor,  # is opaque to TorchScript
                text_lengths: torch.Tensor, context_list: torch.Tensor = torch.IntTensor([0]),
                context_lengths: torch.Tensor = torch.IntTensor([0]), hw_label=torch.IntTensor([0])
                ) -> Dict[str, Optional[torch.Tensor]]:
        """Frontend + Encoder + predictor + joint + loss (transducer.py:79-270)."""
        assert text_lengths.dim() == 1, text_lengths.shape
        assert (speech.shape[0] == speech_lengths.shape[0] == text.shape[0] == text_lengths.shape[0]), \
            (speech.shape, speech_lengths.shape, text.shape, text_lengths.shape)
        check_limits(text, text_lengths, with_ctc=self.ctc_weight != 0.0 and self.ctc is not None)
        cb = self.context_bias
        bias_hidden = cb.forward_bias_hidden(context_list, context_lengths) if cb is not None else None

        encoder_out, encoder_mask = self.encoder(speech, speech_lengths)
        encoder_out_lens = encoder_mask.squeeze(1).sum(1)
        encoder_out_bias = None
        if cb is not None:
            encoder_out, encoder_out_bias = cb.forward_encoder_bias(bias_hidden, encoder_out)
        ys_in_pad = add_blank(text, self.blank, self.ignore_id)
        predictor_out = self.predictor(ys_in_pad)
        predictor_out_bias = None
        if cb is not None:
            predictor_out, predictor_out_bias = cb.forward_predictor_bias(bias_hidden, predictor_out)
        predictor_out_unbiased = predictor_out.clone()

        _, loss_rnnt = self.compute_loss(encoder_out, encoder_out_lens, predictor_out, text, text_lengths)
        loss = self.transducer_weight * loss_rnnt

        loss_att: Optional[torch.Tensor] = None
        if self.attention_decoder_weight != 0.0 and self.decoder is not None:
            loss_att, _ = self._calc_att_loss(encoder_out, encoder_mask, text, text_lengths)
        loss_ctc: Optional[torch.Tensor] = None
        if self.ctc_weight != 0.0 and self.ctc is not None:
            loss_ctc = self.ctc(encoder_out, encoder_out_lens, text, text_lengths)
        if loss_ctc is not None:
            loss = loss + self.ctc_weight * loss_ctc.sum()
        if loss_att is not None:
            loss = loss + self.attention_decoder_weight * loss_att.sum()

        hw_loss: Optional[torch.Tensor] = None
        if self.hw_weight != 0.0 and cb is not None:
            if self.loss_mode == "pred":
                hw_output = cb.forward_hw_pred(bias_hidden, predictor_out_unbiased).permute(0, 2, 1)
                hw_label_pad = end_blank(hw_label, self.blank, self.ignore_id)[..., :-1]
                hw_loss = self.hw_criterion(hw_output[..., :-1], hw_label_pad)
            elif self.loss_mode == "both":
                hw_output = cb.forward_hw_pred_both(encoder_out_bias, predictor_out_bias).permute(0, 2, 1)
                hw_label_pad = end_blank(hw_label, self.blank, self.ignore_id)[..., :-1]
                hw_loss = self.hw_criterion(hw_output[..., :-1], hw_label_pad)
            else:
                _, hw_output_dec = cb.forward_hw_pred_both_sep(encoder_out_bias, predictor_out_bias)
                hw_label_pad = add_blank(hw_label, self.blank, self.ignore_id)
                hw_loss = self.hw_criterion(hw_output_dec.permute(0, 2, 1), hw_label_pad)
            loss = loss + self.hw_weight * hw_loss
        return {"loss": loss, "loss_att": loss_att, "loss_ctc": loss_ctc, "loss_rnnt": loss_rnnt, "hw_loss": hw_loss}

    def _calc_att_loss(self, encoder_out, encoder_mask, ys_pad, ys_pad_lens):
        """asr_model.py:115-148 (attention decoder is whatever module the caller attached)."""
        ys_in_pad, ys_out_pad = add_sos_eos(ys_pad, self.sos, self.eos, self.ignore_id)
        ys_in_lens = ys_pad_lens + 1
        r_ys_pad = reverse_pad_list(ys_pad, ys_pad_lens, float(self.ignore_id))
        r_ys_in_pad, r_ys_out_pad = add_sos_eos(r_ys_pad, self.sos, self.eos, self.ignore_id)
        decoder_out, r_decoder_out, _ = self.decoder(encoder_out, encoder_mask, ys_in_pad, ys_in_lens, r_ys_in_pad,
                                                     self.reverse_weight)
        loss_att = self.criterion_att(decoder_out, ys_out_pad)
        r_loss_att = torch.tensor(0.0, device=loss_att.device)
        if self.reverse_weight > 0.0:
            r_loss_att = self.criterion_att(r_decoder_out, r_ys_out_pad)
        loss_att = loss_att * (1 - self.reverse_weight) + r_loss_att * self.reverse_weight
        pred = decoder_out.view(-1, self.vocab_size).argmax(1)
        mask = ys_out_pad.view(-1) != self.ignore_id
        acc = float((pred[mask] == ys_out_pad.view(-1)[mask]).sum()) / max(int(mask.sum()), 1)
        return loss_att, acc

    # -------------------------------------------------------------- decoding --
    def init_bs(self):
        if self.bs is None:
            self.bs = PrefixBeamSearch(self.encoder, self.predictor, self.joint, self.ctc, self.blank)

    def _cal_transducer_score(self, encoder_out: torch.Tensor, encoder_mask: torch.Tensor, hyps_lens: torch.Tensor,
                              hyps_pad: torch.Tensor):
        """-rnnt_loss(reduction='none') per hypothesis (transducer.py:277-302)."""
        hyps_pad_blank = add_blank(hyps_pad, self.blank, self.ignore_id)
        xs_in_lens = encoder_mask.squeeze(1).sum(1).int()
        predictor_out = self.predictor(hyps_pad_blank)
        rnnt_text = hyps_pad.to(torch.int64)
        rnnt_text = torch.where(rnnt_text == self.ignore_id, 0, rnnt_text).to(torch.int32)
        if self._can_fuse_loss():                   # same node as the training loss block: no pass 1 over the logits
            ep, pp = self.joint.pre_activation(encoder_out, predictor_out)
            loss_td = joint_rnnt_loss(ep, pp, self.joint.ffn_out.weight, self.joint.ffn_out.bias, rnnt_text,
                                      xs_in_lens, hyps_lens.int(), blank=self.blank, reduction="none",
                                      precision=self.joint.precision, activation=self.joint.activation)
            return loss_td * -1
        joint_out = self.joint(encoder_out, predictor_out)
        loss_td = rnnt_loss(joint_out, rnnt_text.contiguous(), xs_in_lens.contiguous(), hyps_lens.int().contiguous(),
                            blank=self.blank, reduction="none")
        return loss_td * -1

    def _cal_attn_score(self, encoder_out, encoder_mask, hyps_pad, hyps_lens, reverse_weight: Optional[float] = None):
        """transducer.py:304-330 (decoder called with self.reverse_weight) and the same block inside
        asr_model.py:485-502 (called with the rescoring call's reverse_weight)."""
        if reverse_weight is None:
            reverse_weight = self.reverse_weight
        ori_hyps_pad = hyps_pad
        hyps_pad, _ = add_sos_eos(hyps_pad, self.sos, self.eos, self.ignore_id)
        hyps_lens = hyps_lens + 1
        r_hyps_pad = reverse_pad_list(ori_hyps_pad, hyps_lens, self.ignore_id)
        r_hyps_pad, _ = add_sos_eos(r_hyps_pad, self.sos, self.eos, self.ignore_id)
        decoder_out, r_decoder_out, _ = self.decoder(encoder_out, encoder_mask, hyps_pad, hyps_lens, r_hyps_pad,
                                                     reverse_weight)
        decoder_out = torch.nn.functional.log_softmax(decoder_out, dim=-1).cpu().numpy()
        r_decoder_out = torch.nn.functional.log_softmax(r_decoder_out, dim=-1).cpu().numpy()
        return decoder_out, r_decoder_out

    def _attention_scores(self, hyps, decoder_out, r_decoder_out, reverse_weight: float):
        """Per hypothesis: sum of the attention decoder's log-probabilities of its tokens plus <eos>, mixed with the
        right-to-left decoder's when reverse_weight > 0 (the inner loop shared by asr_model.py:511-525 and
        transducer.py:485-503; numpy fp32 rows summed left to right as there)."""
        out = []
        for i, hyp in enumerate(hyps):
            n = len(hyp)
            score = 0.0
            for j, w in enumerate(hyp):
                score += decoder_out[i][j][w]
            score += decoder_out[i][n][self.eos]
            if reverse_weight > 0:
                r_score = 0.0
                for j, w in enumerate(hyp):
                    r_score += r_decoder_out[i][n - j - 1][w]
                r_score += r_decoder_out[i][n][self.eos]
                score = score * (1 - reverse_weight) + r_score * reverse_weight
            out.append(score)
        return out

    # CTC decode modes inherited from ASRModel in the reference (asr_model.py:281-440)
    def _forward_encoder(self, speech, speech_lengths, decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1,
                         simulate_streaming: bool = False):
        if simulate_streaming and decoding_chunk_size > 0:
            return self.encoder.forward_chunk_by_chunk(speech, decoding_chunk_size=decoding_chunk_size,
                                                       num_decoding_left_chunks=num_decoding_left_chunks)
        return self.encoder(speech, speech_lengths, decoding_chunk_size, num_decoding_left_chunks)

    def ctc_greedy_search(self, speech: torch.Tensor, speech_lengths: torch.Tensor, decoding_chunk_size: int = -1,
                          num_decoding_left_chunks: int = -1, simulate_streaming: bool = False):
        """asr_model.py:281-324 -> (hyps, scores)."""
        from .ctc import ctc_greedy_search
        assert speech.shape[0] == speech_lengths.shape[0]
        assert decoding_chunk_size != 0
        encoder_out, encoder_mask = self._forward_encoder(speech, speech_lengths, decoding_chunk_size,
                                                          num_decoding_left_chunks, simulate_streaming)
        encoder_out_lens = encoder_mask.squeeze(1).sum(1)
        with torch.no_grad():
            logits = self.ctc.ctc_lo(encoder_out)
        return ctc_greedy_search(logits, encoder_out_lens, blank=0, eos=self.eos)

    def _ctc_prefix_beam_search(self, speech: torch.Tensor, speech_lengths: torch.Tensor, beam_size: int,
                                decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1,
                                simulate_streaming: bool = False):
        """asr_model.py:326-409 -> (hyps [(prefix, score)], encoder_out)."""
        from .ctc import ctc_prefix_beam_search
        assert speech.shape[0] == speech_lengths.shape[0]
        assert decoding_chunk_size != 0
        assert speech.shape[0] == 1
        encoder_out, encoder_mask = self._forward_encoder(speech, speech_lengths, decoding_chunk_size,
                                                          num_decoding_left_chunks, simulate_streaming)
        with torch.no_grad():
            logits = self.ctc.ctc_lo(encoder_out)
        lens = torch.tensor([encoder_out.size(1)], dtype=torch.int32)
        return ctc_prefix_beam_search(logits, lens, beam_size)[0], encoder_out

    def ctc_prefix_beam_search(self, speech: torch.Tensor, speech_lengths: torch.Tensor, beam_size: int,
                               decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1,
                               simulate_streaming: bool = False):
        """asr_model.py:411-440"""
        hyps, _ = self._ctc_prefix_beam_search(speech, speech_lengths, beam_size, decoding_chunk_size,
                                               num_decoding_left_chunks, simulate_streaming)
        return hyps[0]

    # ------------------------------------------- ASRModel surface the reference class inherits (asr_model.py) --
    # The reference's Transducer IS an ASRModel (transducer.py:20), so wenet/bin/recognize.py may call `recognize`
    # (--mode attention, recognize.py:259) and `attention_rescoring` (--mode attention_rescoring, :351) on it, and
    # the C++ runtime calls the exported helpers below.  Plain host logic over the caller's attention decoder; the
    # n-best list of attention_rescoring comes from the HIP CTC prefix beam search.
    def recognize(self, speech: torch.Tensor, speech_lengths: torch.Tensor, beam_size: int = 10,
                  decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1, simulate_streaming: bool = False):
        """Beam search on the attention decoder, batched (asr_model.py:175-279) -> (best_hyps (B, L), best_scores (B,)).
        Every step keeps `beam_size` prefixes per utterance: per-prefix top-k, finished prefixes carry one zero-cost
        <eos> branch, top-k again over the beam_size^2 candidates; no length normalisation."""
        assert speech.shape[0] == speech_lengths.shape[0]
        assert decoding_chunk_size != 0
        dev = speech.device
        B, N = speech.shape[0], beam_size
        memory, memory_mask = self._forward_encoder(speech, speech_lengths, decoding_chunk_size, num_decoding_left_chunks,
                                                    simulate_streaming)
        L = memory.size(1)
        memory = memory.repeat_interleave(N, dim=0)                              # utterance-major, beam-minor rows
        memory_mask = memory_mask.repeat_interleave(N, dim=0)
        hyps = torch.full((B * N, 1), self.sos, dtype=torch.long, device=dev)
        scores = torch.full((B, N), -float("inf"), device=dev)
        scores[:, 0] = 0.0                                                       # one live prefix per utterance at start
        scores = scores.view(-1, 1)
        done = torch.zeros(B * N, 1, dtype=torch.bool, device=dev)
        cache: Optional[List[torch.Tensor]] = None
        first_of_utt = torch.arange(B, device=dev).unsqueeze(1) * N            # (B, 1)
        for i in range(1, L + 1):
            if bool(done.all()):
                break
            causal = torch.ones(i, i, dtype=torch.bool, device=dev).tril().unsqueeze(0).expand(B * N, i, i)
            logp, cache = self.decoder.forward_one_step(memory, memory_mask, hyps, causal, cache)
            cand_logp, cand_tok = logp.topk(N)                                   # (B*N, N)
            # a finished prefix keeps exactly one continuation: <eos> at no cost
            cand_logp = cand_logp.masked_fill(done, -float("inf"))
            cand_logp[:, 0] = torch.where(done.squeeze(1), torch.zeros_like(cand_logp[:, 0]), cand_logp[:, 0])
            cand_tok = cand_tok.masked_fill(done, self.eos)
            scores, pick = (scores + cand_logp).view(B, N * N).topk(N)           # (B, N) each
            scores = scores.view(-1, 1)
            parent = (first_of_utt + pick // N).view(-1)                         # row of the prefix each survivor extends
            tok = cand_tok.view(B, N * N).gather(1, pick).view(-1, 1)
            hyps = torch.cat([hyps.index_select(0, parent), tok], dim=1)
            done = tok == self.eos
        best_scores, best = scores.view(B, N).max(dim=-1)
        rows = best + torch.arange(B, device=dev) * N
        return hyps.index_select(0, rows)[:, 1:], best_scores

    def attention_rescoring(self, speech: torch.Tensor, speech_lengths: torch.Tensor, beam_size: int,
                            decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1, ctc_weight: float = 0.0,
                            simulate_streaming: bool = False, reverse_weight: float = 0.0):
        """CTC prefix beam search n-best rescored by the attention decoder (asr_model.py:443-540) ->
        (best prefix tuple, score)."""
        assert speech.shape[0] == speech_lengths.shape[0]
        assert decoding_chunk_size != 0
        if reverse_weight > 0.0:
            assert hasattr(self.decoder, "right_decoder")
        device = speech.device
        assert speech.shape[0] == 1
        hyps, encoder_out = self._ctc_prefix_beam_search(speech, speech_lengths, beam_size, decoding_chunk_size,
                                                         num_decoding_left_chunks, simulate_streaming)
        assert len(hyps) == beam_size
        prefixes = [h[0] for h in hyps]
        hyps_pad = pad_sequence([torch.tensor(h, device=device, dtype=torch.long) for h in prefixes], True, self.ignore_id)
        hyps_lens = torch.tensor([len(h) for h in prefixes], device=device, dtype=torch.long)
        encoder_out = encoder_out.repeat(beam_size, 1, 1)
        encoder_mask = torch.ones(beam_size, 1, encoder_out.size(1), dtype=torch.bool, device=device)
        decoder_out, r_decoder_out = self._cal_attn_score(encoder_out, encoder_mask, hyps_pad, hyps_lens,
                                                          reverse_weight=reverse_weight)
        att = self._attention_scores(prefixes, decoder_out, r_decoder_out, reverse_weight)
        best_score, best_index = -float("inf"), 0
        for i in range(len(hyps)):
            score = att[i] + hyps[i][1] * ctc_weight
            if score > best_score:
                best_score, best_index = score, i
        return hyps[best_index][0], best_score

    def beam_search(self, speech: torch.Tensor, speech_lengths: torch.Tensor, decoding_chunk_size: int = -1,
                    beam_size: int = 5, num_decoding_left_chunks: int = -1, simulate_streaming: bool = False,
                    ctc_weight: float = 0.3, transducer_weight: float = 0.7, **_ignored):
        """transducer.py:332-377.  Extra keyword arguments (recognize.py:303-304 passes context_list /
        context_lengths, which the reference signature lacks) are accepted and ignored."""
        self.init_bs()
        beam, _ = self.bs.prefix_beam_search(speech, speech_lengths, decoding_chunk_size, beam_size,
                                             num_decoding_left_chunks, simulate_streaming, ctc_weight,
                                             transducer_weight)
        return beam[0].hyp[1:], beam[0].score

    def transducer_attention_rescoring(self, speech: torch.Tensor, speech_lengths: torch.Tensor, beam_size: int,
                                       decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1,
                                       simulate_streaming: bool = False, reverse_weight: float = 0.0,
                                       ctc_weight: float = 0.0, attn_weight: float = 0.0,
                                       transducer_weight: float = 0.0, search_ctc_weight: float = 1.0,
                                       search_transducer_weight: float = 0.0, beam_search_type: str = "transducer"):
        """transducer.py:379-513"""
        assert speech.shape[0] == speech_lengths.shape[0]
        assert decoding_chunk_size != 0
        if reverse_weight > 0.0:
            assert hasattr(self.decoder, "right_decoder")
        device = speech.device
        assert speech.shape[0] == 1
        self.init_bs()
        if beam_search_type == "transducer":
            beam, encoder_out = self.bs.prefix_beam_search(speech, speech_lengths, decoding_chunk_size=decoding_chunk_size,
                                                           beam_size=beam_size,
                                                           num_decoding_left_chunks=num_decoding_left_chunks,
                                                           ctc_weight=search_ctc_weight,
                                                           transducer_weight=search_transducer_weight)
            beam_score = [s.score for s in beam]
            hyps = [s.hyp[1:] for s in beam]
        elif beam_search_type == "ctc":                                     # transducer.py:447-456
            hyps, encoder_out = self._ctc_prefix_beam_search(speech, speech_lengths, beam_size=beam_size,
                                                             decoding_chunk_size=decoding_chunk_size,
                                                             num_decoding_left_chunks=num_decoding_left_chunks,
                                                             simulate_streaming=simulate_streaming)
            beam_score = [hyp[1] for hyp in hyps]
            hyps = [hyp[0] for hyp in hyps]                                 # prefix tuples, as the reference keeps them
        else:
            raise ValueError(f"unknown beam_search_type {beam_search_type!r}")
        assert len(hyps) == beam_size
        hyps_pad = pad_sequence([torch.tensor(h, device=device, dtype=torch.long) for h in hyps], True, self.ignore_id)
        hyps_lens = torch.tensor([len(h) for h in hyps], device=device, dtype=torch.long)
        encoder_out = encoder_out.repeat(beam_size, 1, 1)
        encoder_mask = torch.ones(beam_size, 1, encoder_out.size(1), dtype=torch.bool, device=device)
        td_score = self._cal_transducer_score(encoder_out, encoder_mask, hyps_lens, hyps_pad)
        decoder_out, r_decoder_out = self._cal_attn_score(encoder_out, encoder_mask, hyps_pad, hyps_lens)
        att = self._attention_scores(hyps, decoder_out, r_decoder_out, reverse_weight)
        best_score, best_index = -float("inf"), 0
        for i in range(len(hyps)):
            score = att[i] * attn_weight + beam_score[i] * ctc_weight + td_score[i] * transducer_weight
            if score > best_score:
                best_score, best_index = score, i
        return hyps[best_index], best_score

    def greedy_search(self, speech: torch.Tensor, speech_lengths: torch.Tensor, decoding_chunk_size: int = -1,
                      num_decoding_left_chunks: int = -1, simulate_streaming: bool = False, n_steps: int = 64,
                      context_list: torch.Tensor = torch.IntTensor([0]),
                      context_lengths: torch.Tensor = torch.IntTensor([0]), context_filter_state: str = "on",
                      context_decoder_labels_padded: torch.Tensor = torch.IntTensor([0])):
        """transducer.py:515-598 -> (hyps: List[List[int]], dist).  With no hot-word module the fork's
        variants reduce to the upstream loop and `dist` (an edit distance over the hot-word gate trace) is 0."""
        assert speech.size(0) == 1
        assert speech.shape[0] == speech_lengths.shape[0]
        assert decoding_chunk_size != 0
        _ = simulate_streaming
        encoder_out, encoder_mask = self.encoder(speech, speech_lengths, decoding_chunk_size, num_decoding_left_chunks)
        encoder_out_lens = encoder_mask.squeeze(1).sum()
        if self.context_bias is None:                       # no hot-word module: the variants reduce to the core loop
            return basic_greedy_search(self, encoder_out, encoder_out_lens, n_steps=n_steps), 0
        if self.loss_mode == "pred":                        # transducer.py:559-567
            hyps, dist, _ = basic_greedy_search_hw(self, encoder_out, encoder_out_lens, context_list, context_lengths,
                                                   n_steps=n_steps, context_filter_state=context_filter_state,
                                                   context_decoder_labels_padded=context_decoder_labels_padded)
            return hyps, dist
        # loss_mode 'both' (the default) -- and the reference's final `else` branch, which calls the same function
        # but assigns its (hyps, dist) tuple to `hyps` (transducer.py:589-597, a latent bug); we return (hyps, dist)
        return basic_greedy_search_both(self, encoder_out, encoder_out_lens, context_list, context_lengths,
                                        n_steps=n_steps, context_filter_state=context_filter_state,
                                        context_decoder_labels_padded=context_decoder_labels_padded)

    def greedy_search_batch(self, speech: torch.Tensor, speech_lengths: torch.Tensor, decoding_chunk_size: int = -1,
                            num_decoding_left_chunks: int = -1, n_steps: int = 64) -> List[List[int]]:
        """Extension: N independent streams decoded together (BASELINE config 3)."""
        encoder_out, encoder_mask = self.encoder(speech, speech_lengths, decoding_chunk_size, num_decoding_left_chunks)
        return basic_greedy_search(self, encoder_out, encoder_mask.squeeze(1).sum(1), n_steps=n_steps)

    # ------------------------------------- streaming greedy ("transducer ref.py":541-606) --
    def reset_cache(self, n_streams: int = 1, chunk_frames: int = 64, n_steps: int = 64) -> None:
        """Start `n_streams` fresh streams (the reference's reset_cache is the n_streams=1 case)."""
        self._stream = dict(n=n_streams, fresh=True, tmax=chunk_frames, n_steps=n_steps)

    def forward_greedy_search(self, encoder_out: torch.Tensor, encoder_out_lens: torch.Tensor, n_steps: int = 64,
                              reference_new_cache: bool = True):
        """Decode the next chunk of encoder frames of every stream: encoder_out (N, Tchunk, E) -> tokens of this
        chunk (List[int] for one stream as in the reference, List[List[int]] for several)."""
        st = getattr(self, "_stream", None)
        if st is None:
            self.reset_cache(encoder_out.size(0))
            st = self._stream
        N, T, _ = encoder_out.shape
        assert N == st["n"], "call reset_cache(n_streams) before changing the number of streams"
        lens = torch.as_tensor(encoder_out_lens).reshape(-1)
        if lens.numel() == 1 and N > 1:
            lens = lens.expand(N)
        dec = self._decoder_cache.get(self.predictor, self.joint, lanes=N, utts=N, tmax=max(T, st["tmax"]),
                                      max_hyp=max(T, st["tmax"]) * n_steps, beam=1)
        if not st["fresh"] and getattr(self, "_stream_dec", None) is not dec:
            raise RuntimeError("the decoder handle was rebuilt (weights changed or capacity grew) in the middle of a "
                               "stream: call reset_cache() with the largest chunk size first")
        hyps = dec.greedy_chunk(encoder_out, lens, n_steps=n_steps, blank=self.blank, reset=st["fresh"],
                                reference_new_cache=reference_new_cache)
        st["fresh"] = False
        self._stream_dec = dec
        return hyps[0] if N == 1 else hyps

    # ----------------------------------------------------- step exports (:600-629) --
    # @torch.jit.export as in the reference, so that torch.jit.script(model) (train.py:203-205, export_jit.py) yields an
    # artefact with the four step methods the C++ runtime calls (runtime/core/decoder/torch_asr_model.cc).  Eager calls
    # run the HIP kernels; inside the scripted artefact -- which cannot reach a ctypes library -- the predictor step and
    # the joiner are their plain module graphs (`_export_step`, `_export_forward`).
    @torch.jit.export
    def forward_encoder_chunk(self, xs: torch.Tensor, offset: int, required_cache_size: int,
                              att_cache: torch.Tensor = torch.zeros(0, 0, 0, 0),
                              cnn_cache: torch.Tensor = torch.zeros(0, 0, 0, 0)
                              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return self.encoder.forward_chunk(xs, offset, required_cache_size, att_cache, cnn_cache)

    # The ASRModel exports the C++ runtime calls on every model (asr_model.py:542-634; torch_asr_model.cc reads
    # subsampling_rate / right_context / sos_symbol / eos_symbol / is_bidirectional_decoder at load time).
    @torch.jit.export
    def subsampling_rate(self) -> int:
        if torch.jit.is_scripting():
            return self._subsampling_rate        # captured at construction: a scripted body cannot probe for `embed`
        else:
            return self.encoder.embed.subsampling_rate

    @torch.jit.export
    def right_context(self) -> int:
        if torch.jit.is_scripting():
            return self._right_context
        else:
            return self.encoder.embed.right_context

    @torch.jit.export
    def sos_symbol(self) -> int:
        return self.sos

    @torch.jit.export
    def eos_symbol(self) -> int:
        return self.eos

    @torch.jit.export
    def ctc_activation(self, xs: torch.Tensor) -> torch.Tensor:
        """Linear + log-softmax in front of the CTC search (asr_model.py:597-607)."""
        if self.ctc is not None:
            return self.ctc.log_softmax(xs)
        else:
            raise RuntimeError("ctc_activation: the model was built without a CTC head")

    @torch.jit.export
    def is_bidirectional_decoder(self) -> bool:
        if self.decoder is not None:
            if hasattr(self.decoder, "right_decoder"):
                return True
            else:
                return False
        else:
            return False

    @torch.jit.export
    def forward_attention_decoder(self, hyps: torch.Tensor, hyps_lens: torch.Tensor, encoder_out: torch.Tensor,
                                  reverse_weight: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor]:
        """Score an n-best list (sos-prefixed `hyps` (N, L), lengths counting the sos) on the attention decoder against
        ONE encoder output (asr_model.py:620-710) -> (log-probs (N, L, V), right-to-left log-probs or a zero tensor).
        The right-to-left input is built without pad_sequence (ONNX-friendly in the reference): the label part of every
        row reversed in place, positions beyond its length filled with eos, the sos column kept."""
        if self.decoder is not None:
            assert encoder_out.size(0) == 1
            n = hyps.size(0)
            assert hyps_lens.size(0) == n
            memory = encoder_out.repeat(n, 1, 1)
            memory_mask = torch.ones(n, 1, memory.size(1), dtype=torch.bool, device=memory.device)
            lab_lens = (hyps_lens - 1).unsqueeze(1)                                   # without the sos
            labels = hyps[:, 1:]
            pos = torch.arange(0, int(torch.max(lab_lens)), 1, device=memory.device)
            valid = lab_lens > pos                                                    # (N, Lmax)
            src = (lab_lens - 1 - pos) * valid                                        # mirrored index, 0 where invalid
            flipped = torch.where(valid, torch.gather(labels, 1, src), self.eos)
            r_hyps = torch.cat([hyps[:, 0:1], flipped], dim=1)
            decoder_out, r_decoder_out, _ = self.decoder(memory, memory_mask, hyps, hyps_lens, r_hyps, reverse_weight)
            decoder_out = torch.nn.functional.log_softmax(decoder_out, dim=-1)
            r_decoder_out = torch.nn.functional.log_softmax(r_decoder_out, dim=-1)
            return decoder_out, r_decoder_out
        else:
            raise RuntimeError("forward_attention_decoder: the model was built without an attention decoder")

    @torch.jit.export
    def forward_predictor_step(self, xs: torch.Tensor, cache: List[torch.Tensor]
                               ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        assert len(cache) == 2
        padding = torch.zeros(1, 1, device=xs.device)
        return self.predictor.forward_step(xs, padding, cache)

    @torch.jit.export
    def forward_joint_step(self, enc_out: torch.Tensor, pred_out: torch.Tensor) -> torch.Tensor:
        if torch.jit.is_scripting():
            return self.joint._export_forward(enc_out, pred_out)
        else:
            return self.joint(enc_out, pred_out)

    @torch.jit.export
    def forward_predictor_init_state(self) -> List[torch.Tensor]:
        return self.predictor.init_state(1, device=self.joint.ffn_out.weight.device)
