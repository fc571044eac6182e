"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Plain numpy restatements of the reference's transducer decoders and of the
small modules they call, one utterance at a time, following the reference
line by line in behaviour (not in code):

  predictor step   /root/reference/wenet/transducer/predictor.py:160-200
                   (RNNPredictor.forward_step: embed -> LSTM layers -> projection,
                   ApplyPadding :9-15; gate order i,f,g,o as torch.nn.LSTM)
  joiner           /root/reference/wenet/transducer/joint.py:45-70
                   (ffn_out(tanh(enc_ffn(enc) + pred_ffn(pred))))
  greedy search    /root/reference/wenet/transducer/search/greedy_search copy.py:6-63
                   (the upstream core loop; semantics in SURVEY.md App. A.3)
  prefix beam      /root/reference/wenet/transducer/search/prefix_beam_search.py:42-148
                    with log_add from wenet/utils/common.py:268-276
  ctc log-softmax  /root/reference/wenet/transformer/ctc.py:66-75

PINNED: tests/test_oracle_decode.py replays every fixture under tests/golden/
(greedy_core_*, prefix_beam_*, predictor_step_*, joint_ref_*), which
tests/golden/make_golden.py produced by running the reference's own modules.
Weights are passed as dicts of numpy arrays keyed like the reference modules'
state_dict (embed.weight, rnn.weight_ih_l0, ..., projection.weight;
enc_ffn/pred_ffn/ffn_out .weight/.bias; ctc_lo.weight/.bias).
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F)


def log_softmax(x):
    x = x.astype(F)
    m = x.max(-1, keepdims=True)
    e = np.exp((x - m).astype(F))
    return (x - m - np.log(e.sum(-1, keepdims=True, dtype=F))).astype(F)


def log_add(args):
    """wenet/utils/common.py:268-276 (Python floats = float64)."""
    if all(a == -float("inf") for a in args):
        return -float("inf")
    a_max = max(args)
    return a_max + math.log(sum(math.exp(a - a_max) for a in args))


class Predictor:
    """RNNPredictor step API (predictor.py:123-200).  cache = [h (L,N,H), c (L,N,H)]."""

    def __init__(self, w, n_layers):
        self.w = {k: np.asarray(v, F) for k, v in w.items()}
        self.L = int(n_layers)
        self.H = self.w["rnn.weight_hh_l0"].shape[1]

    def init_state(self, n):
        return [np.zeros((self.L, n, self.H), F), np.zeros((self.L, n, self.H), F)]

    def forward_step(self, tokens, padding, cache):
        """tokens (N,) int, padding (N,1) f32 (1 keeps the old state), cache -> (out (N,O), new cache)."""
        h0, c0 = cache
        x = self.w["embed.weight"][np.asarray(tokens, np.int64)]
        hs, cs = [], []
        for l in range(self.L):
            g = (x @ self.w[f"rnn.weight_ih_l{l}"].T + self.w[f"rnn.bias_ih_l{l}"]
                 + h0[l] @ self.w[f"rnn.weight_hh_l{l}"].T + self.w[f"rnn.bias_hh_l{l}"]).astype(F)
            H = self.H
            i, f, gg, o = _sigmoid(g[:, :H]), _sigmoid(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), _sigmoid(g[:, 3 * H:])
            c = (f * c0[l] + i * gg).astype(F)
            h = (o * np.tanh(c)).astype(F)
            hs.append(h); cs.append(c)
            x = h
        out = (x @ self.w["projection.weight"].T + self.w["projection.bias"]).astype(F)
        m, c = np.stack(hs), np.stack(cs)
        p = np.asarray(padding, F).reshape(1, -1, 1)
        m = p * h0 + m * (1 - p)            # ApplyPadding, predictor.py:9-15
        c = p * c0 + c * (1 - p)
        return out, [m.astype(F), c.astype(F)]


class Joint:
    """TransducerJoint.forward for step shapes (joint.py:45-70)."""

    def __init__(self, w):
        self.w = {k: np.asarray(v, F) for k, v in w.items()}

    def __call__(self, enc, pred):
        """enc (..., E), pred (..., P) broadcastable leading dims -> (..., V)."""
        e = enc @ self.w["enc_ffn.weight"].T + self.w["enc_ffn.bias"]
        p = pred @ self.w["pred_ffn.weight"].T + self.w["pred_ffn.bias"]
        h = np.tanh((e + p).astype(F))
        return (h @ self.w["ffn_out.weight"].T + self.w["ffn_out.bias"]).astype(F)

    def full(self, enc, pred):
        """(B,T,E),(B,U1,P) -> (B,T,U1,V)"""
        e = enc @ self.w["enc_ffn.weight"].T + self.w["enc_ffn.bias"]
        p = pred @ self.w["pred_ffn.weight"].T + self.w["pred_ffn.bias"]
        h = np.tanh((e[:, :, None, :] + p[:, None, :, :]).astype(F))
        return (h @ self.w["ffn_out.weight"].T + self.w["ffn_out.bias"]).astype(F)


def greedy_search(pred: Predictor, joint: Joint, enc, T, blank=0, n_steps=64, return_margin=False):
    """'greedy_search copy.py':14-63 for one utterance.  enc (T,E)."""
    cache = pred.init_state(1)
    tok = np.array([blank])
    padding = np.zeros((1, 1), F)
    t, hyps, prev_nblk, per_frame = 0, [], True, 0
    out, new_cache = None, None
    min_margin = float("inf")
    while t < T:
        if prev_nblk:
            out, new_cache = pred.forward_step(tok, padding, cache)
        lp = log_softmax(joint(enc[t][None, :], out))[0]
        k = int(lp.argmax())
        if return_margin:
            srt = np.sort(lp)
            min_margin = min(min_margin, float(srt[-1] - srt[-2]))
        if k != blank:
            hyps.append(k)
            prev_nblk = True
            per_frame += 1
            tok = np.array([k])
            cache = new_cache
        if k == blank or per_frame >= n_steps:
            if k == blank:
                prev_nblk = False
            t += 1
            per_frame = 0
    return (hyps, min_margin) if return_margin else hyps


class StreamingGreedy:
    """reset_cache / forward_greedy_search of "wenet/transducer/transducer ref.py":541-606 for one stream.
    PINNED: tests/golden/greedy_stream_*.npz hold what those two methods of the reference returned chunk by
    chunk (tests/golden/make_golden.py::gen_greedy_stream loads the file with empty stubs for its unused
    module-level imports of torchaudio / k2), and tests/test_oracle_decode.py::test_streaming_greedy replays
    them.  `reference_new_cache=True` is the reference's behaviour: `new_cache = self.cache` at the top of every
    chunk (:571) drops the predictor state that was computed but not yet committed in the previous chunk;
    False keeps it, which makes a chunked decode equal the offline decode of the concatenated frames."""

    def __init__(self, pred: Predictor, joint: Joint, blank=0, clear_margin=1e-4):
        self.pred, self.joint, self.blank = pred, joint, blank
        self.clear_margin = clear_margin
        self.reset_cache()

    def reset_cache(self):
        self.cache = self.pred.init_state(1)
        self.pending = self.cache
        self.out = None
        self.tok = np.array([self.blank])
        self.per_frame = 0
        self.prev_nblk = True
        self.min_margin = float("inf")      # smallest top-1 / top-2 log-prob gap over all decisions so far
        self.n_tokens = 0                   # tokens emitted since reset_cache
        self.clear_tokens = None            # tokens emitted before the first decision whose gap was < clear_margin

    def forward_greedy_search(self, enc, T, n_steps=64, reference_new_cache=True):
        padding = np.zeros((1, 1), F)
        new_cache = self.cache if reference_new_cache else self.pending
        hyps, t = [], 0
        while t < T:
            if self.prev_nblk:
                self.out, new_cache = self.pred.forward_step(self.tok, padding, self.cache)
            lp = log_softmax(self.joint(enc[t][None, :], self.out))[0]
            k = int(lp.argmax())
            top2 = np.partition(lp, -2)[-2:]
            self.min_margin = min(self.min_margin, float(top2[1] - top2[0]))
            if self.clear_tokens is None and float(top2[1] - top2[0]) < self.clear_margin:
                self.clear_tokens = self.n_tokens
            if k != self.blank:
                self.n_tokens += 1
                hyps.append(k)
                self.prev_nblk = True
                self.per_frame += 1
                self.tok = np.array([k])
                self.cache = new_cache
            if k == self.blank or self.per_frame >= n_steps:
                if k == self.blank:
                    self.prev_nblk = False
                t += 1
                self.per_frame = 0
        self.pending = new_cache
        return hyps


# ---------------------------------------------- hot-word greedy search, the fork's default --
def _layer_norm(x, w, b, eps=1e-5):
    x = x.astype(F)
    mu = x.mean(-1, keepdims=True, dtype=F)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=F)
    return ((x - mu) / np.sqrt(var + F(eps)) * w + b).astype(F)


class ContextBiasNP:
    """The per-step part of /root/reference/wenet/transformer/context_bias.py::ContextBias that the greedy loops call
    (weights keyed like its state_dict):
      forward_predictor_bias  :375-381  MultiHeadedAttention(query = predictor step, key = value = bias_hidden)
                              (attention.py:47-113,153-186) -> predictor_bias_bias_norm -> cat -> predictor_bias_combine
                              -> predictor_bias_out_norm; returns (biased output, bias feature)
      forward_hw_pred_both    :388-394  hw_output_layer_enc / _dec -> hw_bias attention -> hw_bias_norm -> hw_output_layer
    The loop-invariant parts (forward_bias_hidden, forward_encoder_bias) are taken as data."""

    def __init__(self, w, heads, hw_heads):
        self.w = {k: np.asarray(v, F) for k, v in w.items()}
        self.h, self.hw_h = int(heads), int(hw_heads)

    def _lin(self, name, x):
        return (x @ self.w[name + ".weight"].T + self.w[name + ".bias"]).astype(F)

    def _mha(self, name, heads, query, memory):
        """query (Tq, D), memory (Tk, D) -> (Tq, D); no mask, dropout 0 (attention.py:153-186)."""
        q, k, v = self._lin(name + ".linear_q", query), self._lin(name + ".linear_k", memory), self._lin(name + ".linear_v", memory)
        D = q.shape[-1]
        dk = D // heads
        out = np.zeros_like(q)
        for h in range(heads):
            sl = slice(h * dk, (h + 1) * dk)
            sc = (q[:, sl] @ k[:, sl].T / F(math.sqrt(dk))).astype(F)
            sc = sc - sc.max(-1, keepdims=True)
            e = np.exp(sc).astype(F)
            att = (e / e.sum(-1, keepdims=True, dtype=F)).astype(F)
            out[:, sl] = att @ v[:, sl]
        return self._lin(name + ".linear_out", out)

    def forward_predictor_bias(self, hidden, pred):
        """hidden (Nc, D), pred (1, D) -> (biased (1, D), bias feature (1, D))"""
        pb = self._mha("predictor_bias", self.h, pred, hidden)
        pb = _layer_norm(pb, self.w["predictor_bias_bias_norm.weight"], self.w["predictor_bias_bias_norm.bias"])
        cat = np.concatenate([pred, pb], -1)
        out = _layer_norm(self._lin("predictor_bias_combine", cat), self.w["predictor_bias_out_norm.weight"],
                          self.w["predictor_bias_out_norm.bias"])
        return out, pb

    def forward_hw_pred_both(self, enc_bias_step, pred_bias_step):
        """(1, D), (1, D) -> gate logits (n_labels,)"""
        e = self._lin("hw_output_layer_enc", enc_bias_step)
        d = self._lin("hw_output_layer_dec", pred_bias_step)
        hb = self._mha("hw_bias", self.hw_h, d, e)
        hb = _layer_norm(hb, self.w["hw_bias_norm.weight"], self.w["hw_bias_norm.bias"])
        return self._lin("hw_output_layer", hb)[0]


def greedy_search_both(pred: Predictor, joint: Joint, cb: ContextBiasNP, hidden, hidden_empty, enc_hot, enc_hot_feat,
                       enc_cold, T, labels, blank=0, n_steps=64, filter_on=False, return_go_backs=False):
    """/root/reference/wenet/transducer/search/greedy_search.py:297-430 (`basic_greedy_search_both`) for one
    utterance, from the loop-invariant tensors on: hidden (Nc, D) / hidden_empty (1, D) = forward_bias_hidden of the
    hot-word list / of the empty list (:327-333), enc_hot / enc_hot_feat / enc_cold (T, D) = forward_encoder_bias
    (:335-336).  Returns (hyps, dist, gate trace, number of joiner decisions[, number of go-backs])."""
    cache = pred.init_state(1)
    new_cache = cache
    tok = np.array([blank])
    padding = np.zeros((1, 1), F)
    t, hyps, result = 0, [], []
    prev_nblk, per_frame = True, 0
    go_back, go_back_end, last_t = False, -1, 0
    steps, caches, inputs = [], [], []
    out, decisions, n_back = None, 0, 0
    while t < T:
        if prev_nblk:
            raw, new_cache = pred.forward_step(tok, padding, cache)
            steps.append(raw); caches.append(cache); inputs.append(tok)
            out, feat = cb.forward_predictor_bias(hidden, raw)
            gate = int(np.argmax(cb.forward_hw_pred_both(enc_hot_feat[t][None, :], feat)))   # topk(1): first maximum
            if filter_on:
                if not go_back:
                    if gate == 0:
                        result.append(0)
                        last_t = t
                    else:
                        if result and result[-1] == 0:
                            go_back_end, t, go_back = t, last_t, True
                            n_back += 1
                            result.pop(); hyps.pop(); inputs.pop()
                            per_frame -= 1
                            steps.pop(); caches.pop()
                            out, cache, tok = steps[-1], caches[-1], inputs[-1]
                            continue
                        result.append(1)
                else:
                    result.append(1)
                    if t >= go_back_end:
                        go_back = False
            else:
                result.append(1)
            if result[-1] == 0:
                out, _ = cb.forward_predictor_bias(hidden_empty, raw)
        e = (enc_hot if result[-1] == 1 else enc_cold)[t][None, :]
        lp = log_softmax(joint(e, out))[0]
        k = int(lp.argmax())
        decisions += 1
        if k != blank:
            hyps.append(k)
            prev_nblk = True
            per_frame += 1
            tok = np.array([k])
            cache = new_cache
        if k == blank or per_frame >= n_steps:
            if k == blank:
                prev_nblk = False
            t += 1
            per_frame = 0
    res = (hyps, edit_distance(list(labels), result), result, decisions)
    return res + (n_back,) if return_go_backs else res


def edit_distance(a, b):
    """greedy_search.py:6-32"""
    m, n = len(a), len(b)
    prev = list(range(n + 1))
    for i in range(1, m + 1):
        cur = [i] + [0] * n
        for j in range(1, n + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (0 if int(a[i - 1]) == int(b[j - 1]) else 1))
        prev = cur
    return float(prev[n])


def ctc_log_softmax(w, enc):
    """ctc.py:66-75: log_softmax(ctc_lo(hs)).  enc (T,E) -> (T,V)"""
    return log_softmax(enc @ np.asarray(w["ctc_lo.weight"], F).T + np.asarray(w["ctc_lo.bias"], F))


def prefix_beam_search(pred: Predictor, joint: Joint, ctc_w, enc, T, beam_size=5, ctc_weight=0.3,
                       transducer_weight=0.7, blank=0, return_margin=False, stop_below=None):
    """prefix_beam_search.py:42-148 for one utterance.  Returns the pruned beam as a
    list of dicts {hyp (with the seed blank), score (float64), cache}.
    return_margin: also return the smallest score gap that decided anything visible in the result -- between
    neighbours of the sorted fused candidates down to the first pruned one, over all frames.  An implementation
    whose per-frame log-probs differ in the last fp32 bits can only produce a different beam when this is tiny.
    stop_below: stop in front of the first frame whose own margin is below this value and return
    (beam after the clear prefix, smallest margin inside the prefix, number of frames decoded) -- the long-utterance
    comparison of BASELINE config 5: everything up to the first ambiguous decision must agree."""
    ctc_probs = ctc_log_softmax(ctc_w, enc[:T])
    beam = [dict(hyp=[blank], score=0.0, cache=pred.init_state(1))]
    min_margin = float("inf")
    for i in range(T):
        n = len(beam)
        toks = np.array([s["hyp"][-1] for s in beam])
        cache = [np.concatenate([s["cache"][0] for s in beam], 1), np.concatenate([s["cache"][1] for s in beam], 1)]
        scores = np.array([s["score"] for s in beam]).astype(F)          # fp32 tensor from Python floats (:86)
        out, new_cache = pred.forward_step(toks, np.zeros((n, 1), F), cache)
        logp = log_softmax(joint(np.broadcast_to(enc[i], (n, enc.shape[1])), out))
        # (:99-101) log(tw*exp(logp) + cw*exp(ctc[i])) in fp32
        logp = np.log((F(transducer_weight) * np.exp(logp) + F(ctc_weight) * np.exp(ctc_probs[i])[None, :]).astype(F)).astype(F)
        order = np.argsort(-logp, axis=1, kind="stable")[:, :beam_size]   # topk: larger first, lower index on ties
        topv = np.take_along_axis(logp, order, 1)
        cand = (scores[:, None] + topv).astype(F)
        beam_a = []
        for j in range(n):
            base = beam[j]
            for t in range(beam_size):
                k = int(order[j, t])
                if k == blank:
                    beam_a.append(dict(hyp=list(base["hyp"]), score=float(cand[j, t]), cache=base["cache"]))
                else:
                    beam_a.append(dict(hyp=base["hyp"] + [k], score=float(cand[j, t]),
                                       cache=[new_cache[0][:, j:j + 1], new_cache[1][:, j:j + 1]]))
        fusion = [beam_a[0]]
        for s1 in beam_a[1:]:
            for s0 in fusion:
                if s1["hyp"] == s0["hyp"]:
                    s0["score"] = log_add([s0["score"], s1["score"]])
                    break
            else:
                fusion.append(s1)
        fusion.sort(key=lambda s: s["score"], reverse=True)                 # stable, like list.sort
        if return_margin or stop_below is not None:
            sc = [s["score"] for s in fusion[:beam_size + 1]]
            frame_margin = float("inf")
            for a, b in zip(sc[:-1], sc[1:]):
                if math.isfinite(a) and math.isfinite(b):
                    frame_margin = min(frame_margin, a - b)
            if stop_below is not None and frame_margin < stop_below:
                return beam, min_margin, i
            min_margin = min(min_margin, frame_margin)
        beam = fusion[:beam_size]
    if stop_below is not None:
        return beam, min_margin, T
    return (beam, min_margin) if return_margin else beam


# ---------------------------------------------------------------- CTC decode modes --
def ctc_greedy_search(logits, lens, eos):
    """ASRModel.ctc_greedy_search, wenet/transformer/asr_model.py:281-324, from the ctc_lo output on.
    logits (B,T,V).  Returns (hyps, scores): frames past an utterance's length are filled with `eos` BEFORE
    duplicates/blanks are removed (so shorter utterances of a batch end in one eos), and the score is the
    maximum over ALL T frames of the best log-probability -- both exactly as the reference does."""
    lp = log_softmax(np.asarray(logits, F))
    best = lp.argmax(-1)
    top = lp.max(-1)
    B, T = best.shape
    hyps = []
    for b in range(B):
        seq = [int(best[b, t]) if t < lens[b] else eos for t in range(T)]
        out, cur = [], 0
        while cur < len(seq):                      # remove_duplicates_and_blank, common.py:256-265
            if seq[cur] != 0:
                out.append(seq[cur])
            prev = cur
            while cur < len(seq) and seq[cur] == seq[prev]:
                cur += 1
        hyps.append(out)
    return hyps, top.max(1)


def ctc_prefix_beam_search(logp, T, beam_size):
    """ASRModel._ctc_prefix_beam_search, asr_model.py:326-409, for one utterance from its CTC log-probs (T,V).
    Returns [(prefix tuple, score)] best first."""
    from collections import defaultdict
    logp = np.asarray(logp, F)
    cur_hyps = [(tuple(), (0.0, -float("inf")))]
    for t in range(T):
        row = logp[t]
        next_hyps = defaultdict(lambda: (-float("inf"), -float("inf")))
        order = np.argsort(-row, kind="stable")[:beam_size]
        for s in order:
            s = int(s)
            ps = float(row[s])
            for prefix, (pb, pnb) in cur_hyps:
                last = prefix[-1] if len(prefix) > 0 else None
                if s == 0:
                    n_pb, n_pnb = next_hyps[prefix]
                    next_hyps[prefix] = (log_add([n_pb, pb + ps, pnb + ps]), n_pnb)
                elif s == last:
                    n_pb, n_pnb = next_hyps[prefix]
                    next_hyps[prefix] = (n_pb, log_add([n_pnb, pnb + ps]))
                    n_prefix = prefix + (s,)
                    n_pb, n_pnb = next_hyps[n_prefix]
                    next_hyps[n_prefix] = (n_pb, log_add([n_pnb, pb + ps]))
                else:
                    n_prefix = prefix + (s,)
                    n_pb, n_pnb = next_hyps[n_prefix]
                    next_hyps[n_prefix] = (n_pb, log_add([n_pnb, pb + ps, pnb + ps]))
        nh = sorted(next_hyps.items(), key=lambda x: log_add(list(x[1])), reverse=True)
        cur_hyps = nh[:beam_size]
    return [(y[0], log_add([y[1][0], y[1][1]])) for y in cur_hyps]


def forced_align(ctc_probs, y, blank_id=0):
    """wenet/utils/ctc_util.py:27-83 restated (fp32 scores, torch.argmax = first maximum, and the reference's
    negative-index quirk: for s = 0 the candidate log_alpha[t-1, s-1] is the LAST state)."""
    lp = np.asarray(ctc_probs, F)
    T = lp.shape[0]
    ext = [blank_id]
    for tok in y:
        ext += [int(tok), blank_id]
    NS = len(ext)
    alpha = np.full((T, NS), -np.inf, F)
    path = np.full((T, NS), -1, np.int64)
    alpha[0, 0] = lp[0, ext[0]]
    alpha[0, 1] = lp[0, ext[1]]
    for t in range(1, T):
        for s in range(NS):
            if ext[s] == blank_id or s < 2 or ext[s] == ext[s - 2]:
                prev = [s, s - 1]
            else:
                prev = [s, s - 1, s - 2]
            cands = np.array([alpha[t - 1, p] for p in prev], F)       # p = -1 wraps, as in Python
            k = int(np.argmax(cands))
            alpha[t, s] = cands[k] + lp[t, ext[s]]
            path[t, s] = prev[k]
    st = [NS - 1, NS - 2][int(np.argmax(np.array([alpha[-1, NS - 1], alpha[-1, NS - 2]], F)))]
    seq = [0] * T
    seq[-1] = st
    for t in range(T - 2, -1, -1):
        seq[t] = path[t + 1, seq[t + 1]]
    return [ext[s] for s in seq]
