"""Pins oracle/decode_oracle.py (and the C CTC oracle at module level) against
the fixtures that tests/golden/make_golden.py produced by running the
reference's own modules: predictor step, joiner, greedy search, prefix beam
search, CTC module."""
import glob
import math
import os

import numpy as np
import pytest

import oracle
from oracle import decode_oracle as do
from conftest import GOLDEN


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def names(pattern):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, pattern)))


def test_fixture_inventory():
    assert len(names("greedy_core_*.npz")) >= 6
    assert len(names("prefix_beam_*.npz")) >= 6
    assert len(names("ctc_ref_*.npz")) >= 5
    assert len(names("joint_ref_*.npz")) >= 2
    assert len(names("predictor_step_*.npz")) >= 2
    assert len(names("greedy_stream_*.npz")) >= 5
    assert len(names("greedy_both_real_*.npz")) >= 6


def test_common_helpers():
    d = load("common_ref.npz")
    for pair, ref in zip(d["log_add_in"], d["log_add_out"]):
        assert do.log_add(list(pair)) == pytest.approx(ref, rel=1e-15)
    assert do.log_add([-float("inf")] * 2) == -float("inf")


@pytest.mark.parametrize("name", names("predictor_step_*.npz"))
def test_predictor_step(name):
    d = load(name)
    p = do.Predictor(sub(d, "w_"), int(d["n_layers"]))
    cache = p.init_state(d["toks"].shape[1])
    for s in range(d["toks"].shape[0]):
        out, cache = p.forward_step(d["toks"][s], d["padding"], cache)
        np.testing.assert_allclose(out, d["outs"][s][:, 0, :], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cache[0], d["m"][s], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cache[1], d["c"][s], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", names("joint_ref_*.npz"))
def test_joint(name):
    d = load(name)
    j = do.Joint(sub(d, "w_"))
    np.testing.assert_allclose(j.full(d["enc"], d["pred"]), d["out"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", names("greedy_core_*.npz"))
def test_greedy(name):
    d = load(name)
    p = do.Predictor(sub(d, "pred_"), int(d["n_layers"]))
    j = do.Joint(sub(d, "joint_"))
    hyp, margin = do.greedy_search(p, j, d["enc"][0], int(d["T"]), blank=0, n_steps=int(d["n_steps"]),
                                   return_margin=True)
    assert hyp == list(d["hyp"])                       # token sequences identical
    assert margin == pytest.approx(float(d["min_margin"]), abs=1e-4)


@pytest.mark.parametrize("name", names("greedy_stream_*.npz"))
def test_streaming_greedy(name):
    """The restated reset_cache / forward_greedy_search pair returns, chunk by chunk, what the reference's own
    methods ("transducer ref.py":541-606) returned for the same chunking."""
    d = load(name)
    p = do.Predictor(sub(d, "pred_"), int(d["n_layers"]))
    j = do.Joint(sub(d, "joint_"))
    st = do.StreamingGreedy(p, j)
    a, k = 0, 0
    for n, cnt in zip(d["chunk_sizes"], d["chunk_token_counts"]):
        got = st.forward_greedy_search(d["enc"][0, a:a + n], int(n), n_steps=int(d["n_steps"]), reference_new_cache=True)
        assert got == list(d["chunk_tokens"][k:k + cnt]), (name, a, n)
        a += n
        k += cnt


@pytest.mark.parametrize("name", names("greedy_both_real_*.npz"))
def test_greedy_both_with_real_context_bias(name):
    """The restated hot-word loop + ContextBias step arithmetic reproduce what the reference's
    basic_greedy_search_both produced with its real ContextBias module: tokens, edit distance, gate trace and even the
    number of joiner decisions (go-back re-decoding included)."""
    d = load(name)
    p = do.Predictor(sub(d, "pred_"), int(d["n_layers"]))
    j = do.Joint(sub(d, "joint_"))
    cb = do.ContextBiasNP(sub(d, "cb_"), int(d["heads"]), int(d["hw_heads"]))
    hyps, dist, trace, n_dec = do.greedy_search_both(p, j, cb, d["hidden"][0], d["hidden_empty"][0], d["enc_hot"][0],
                                                     d["enc_hot_feat"][0], d["enc_cold"][0], int(d["T"]), d["labels"][0],
                                                     n_steps=int(d["n_steps"]), filter_on=str(d["filt"]) == "on")
    assert hyps == list(d["hyp"])
    assert trace == list(d["trace"])
    assert dist == float(d["dist"])
    assert n_dec == int(d["n_decisions"])
    # the gate depends on the frame only: with a single key the attention weight is exactly 1
    for t in range(int(d["T"])):
        g = cb.forward_hw_pred_both(d["enc_hot_feat"][0][t][None, :], np.zeros((1, d["enc_hot_feat"].shape[-1]), np.float32))
        np.testing.assert_allclose(g, d["gate_logits"][t], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", names("prefix_beam_*.npz"))
def test_prefix_beam(name):
    d = load(name)
    p = do.Predictor(sub(d, "pred_"), int(d["n_layers"]))
    j = do.Joint(sub(d, "joint_"))
    beam = do.prefix_beam_search(p, j, sub(d, "ctc_"), d["enc"][0], int(d["T"]), beam_size=int(d["beam"]),
                                 ctc_weight=float(d["ctc_weight"]), transducer_weight=float(d["transducer_weight"]))
    assert len(beam) == len(d["scores"])
    for k, s in enumerate(beam):                         # exact hypotheses in exact order
        assert s["hyp"] == list(d["hyps"][k][: d["hyp_lens"][k]])
        assert s["score"] == pytest.approx(d["scores"][k], rel=1e-5)
        np.testing.assert_allclose(s["cache"][0], d["cache_m"][k], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", names("ctc_ref_*.npz"))
def test_ctc_module_fixture(name):
    """Reference CTC.forward (ctc.py:46-64) == Linear + fused-log-softmax CTC oracle, /B."""
    d = load(name)
    logits = d["hs"] @ d["w_ctc_lo.weight"].T + d["w_ctc_lo.bias"]
    ys = np.where(d["ys"] < 0, 0, d["ys"]).astype(np.int32)
    nll, grad = oracle.ctc_loss_f64(logits.astype(np.float32), ys, d["hlens"], d["ys_lens"])
    B = logits.shape[0]
    if np.isfinite(d["loss"]):
        assert nll.sum() / B == pytest.approx(float(d["loss"]), rel=1e-5)
        grad_hs = (grad / B) @ d["w_ctc_lo.weight"]
        np.testing.assert_allclose(grad_hs, d["grad_hs"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(np.einsum("btv,btd->vd", grad / B, d["hs"]), d["grad_w"], rtol=1e-4, atol=1e-4)  # fp32 reference sums B*T terms
    else:
        assert np.isinf(nll.sum())
    lp = do.log_softmax(logits)
    np.testing.assert_allclose(lp, d["log_softmax"], rtol=1e-5, atol=1e-5)
    assert (logits.argmax(-1) == d["argmax"]).all()


@pytest.mark.parametrize("name", names("ctc_decode_*.npz"))
def test_ctc_decode_modes(name):
    """oracle restatements of ASRModel.ctc_greedy_search / _ctc_prefix_beam_search vs the reference's outputs."""
    d = load(name)
    V = d["logits"].shape[-1]
    hyps, scores = do.ctc_greedy_search(d["logits"], d["lens"], eos=V - 1)
    for b, h in enumerate(hyps):
        assert h == list(d["greedy"][b][: d["greedy_lens"][b]])
    np.testing.assert_allclose(scores, d["greedy_scores"], rtol=1e-5, atol=1e-6)
    lp = do.log_softmax(d["logits"])
    for b in range(d["logits"].shape[0]):
        nb = do.ctc_prefix_beam_search(lp[b], int(d["lens"][b]), int(d["beam"]))
        assert len(nb) == int(d["nbest_n"][b])
        for k, (pref, sc) in enumerate(nb):
            assert list(pref) == list(d["nbest"][b, k][: d["nbest_lens"][b, k]])
            assert sc == pytest.approx(d["nbest_scores"][b, k], rel=1e-6)


def test_ctc_prefix_known_answer_from_reference_gtest():
    """runtime/core/test/ctc_prefix_beam_search_test.cc:30-73: n-best [2,1], [1,2], [1] with likelihoods
    0.2185 / 0.1550 / 0.1525."""
    d = load("ctc_prefix_kat.npz")
    nb = do.ctc_prefix_beam_search(np.log(d["probs"]), 3, int(d["beam"]))
    for k, (pref, sc) in enumerate(nb):
        assert list(pref) == list(d["nbest"][k][: d["nbest_lens"][k]])
        assert math.exp(sc) == pytest.approx(float(d["likelihood"][k]), rel=1e-4)


@pytest.mark.parametrize("name", names("ctc_align_*.npz"))
def test_forced_align(name):
    d = load(name)
    assert do.forced_align(d["ctc_probs"], d["y"]) == list(d["alignment"])
