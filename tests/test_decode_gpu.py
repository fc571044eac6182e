"""Decode parity on the GPU: token sequences identical to what the reference's
own decoders produced (fixtures tests/golden/greedy_core_*, prefix_beam_*,
predictor_step_*), batched execution identical to one-stream-at-a-time, hipGraph
replay identical to plain launches, and BASELINE-shape runs checked against the
numpy oracle."""
import glob
import os
import types

import numpy as np
import pytest
import torch

from oracle import decode_oracle as do
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def names(pattern):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def sub(d, prefix):
    return {k[len(prefix):]: d[k] for k in d.files if k.startswith(prefix)}


def build_modules(d, with_ctc=False):
    import wenet_celoss_amd as w
    pw, jw = sub(d, "pred_"), sub(d, "joint_")
    V, D = pw["embed.weight"].shape
    H = pw["rnn.weight_hh_l0"].shape[1]
    L = int(d["n_layers"])
    P = pw["projection.weight"].shape[0]
    J, E = jw["enc_ffn.weight"].shape
    pred = w.RNNPredictor(V, D, P, 0.1, H, L).to(DEV).eval()
    pred.load_state_dict({k: torch.tensor(v) for k, v in pw.items()})
    joint = w.TransducerJoint(V, E, P, J).to(DEV).eval()
    joint.load_state_dict({k: torch.tensor(v) for k, v in jw.items()})
    ctc = None
    if with_ctc:
        ctc = w.CTC(V, E).to(DEV).eval()
        ctc.load_state_dict({k[4:]: torch.tensor(d[k]) for k in d.files if k.startswith("ctc_ctc_lo")})
    return pred, joint, ctc


@pytest.mark.parametrize("path", names("predictor_step_*.npz"))
def test_predictor_step_matches_reference(path):
    import wenet_celoss_amd as w
    d = np.load(path)
    pw = sub(d, "w_")
    V, D = pw["embed.weight"].shape
    H = pw["rnn.weight_hh_l0"].shape[1]
    P = pw["projection.weight"].shape[0]
    pred = w.RNNPredictor(V, D, P, 0.1, H, int(d["n_layers"])).to(DEV).eval()
    pred.load_state_dict({k: torch.tensor(v) for k, v in pw.items()})
    N = d["toks"].shape[1]
    cache = pred.init_state(N, device=torch.device(DEV))
    padding = torch.tensor(d["padding"], device=DEV)
    for s in range(d["toks"].shape[0]):
        out, cache = pred.forward_step(torch.tensor(d["toks"][s], device=DEV).reshape(N, 1), padding, cache)
        np.testing.assert_allclose(out.cpu().numpy(), d["outs"][s], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(cache[0].cpu().numpy(), d["m"][s], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(cache[1].cpu().numpy(), d["c"][s], rtol=1e-4, atol=1e-5)
    # cache_to_batch / batch_to_cache layout (predictor.py:123-158)
    back = pred.cache_to_batch(pred.batch_to_cache(cache))
    assert torch.equal(back[0], cache[0]) and torch.equal(back[1], cache[1])
    # training-time forward (library LSTM) agrees with the reference too
    full = pred(torch.tensor(d["toks"][:, :1].T.copy(), device=DEV))
    np.testing.assert_allclose(full.detach().cpu().numpy(), d["full_first_lane"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("path", names("greedy_core_*.npz"))
def test_greedy_matches_reference_tokens(path, use_graph):
    import wenet_celoss_amd as w
    d = np.load(path)
    pred, joint, _ = build_modules(d)
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    enc = torch.tensor(d["enc"], device=DEV)
    T, n_steps = int(d["T"]), int(d["n_steps"])
    from wenet_celoss_amd.decoder import DecoderCache
    model._decoder_cache = DecoderCache()
    dec = model._decoder_cache.get(pred, joint, lanes=1, utts=1, tmax=T, max_hyp=T * n_steps, beam=1)
    dec.set_graph(use_graph)
    hyps = w.basic_greedy_search(model, enc, torch.tensor(T), n_steps=n_steps)
    assert hyps == [list(d["hyp"])]


def test_greedy_batched_streams_equal_single_streams():
    """All fixtures that share weights' shapes cannot be batched (different weights), so batch one
    fixture's utterance with truncated / shifted copies of itself and compare with the oracle per stream."""
    import wenet_celoss_amd as w
    d = np.load(names("greedy_core_1.npz")[0])
    pred, joint, _ = build_modules(d)
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    enc0 = d["enc"][0]
    T = enc0.shape[0]
    rng = np.random.default_rng(0)
    lens = [T, T // 2, 1, T - 7, 13, T]
    encs = np.zeros((len(lens), T, enc0.shape[1]), np.float32)
    for i, l in enumerate(lens):
        shift = int(rng.integers(0, T - l + 1))
        encs[i, :l] = enc0[shift:shift + l]
    encs[-1] = enc0[::-1]
    hyps = w.basic_greedy_search(model, torch.tensor(encs, device=DEV), torch.tensor(lens), n_steps=int(d["n_steps"]))
    p = do.Predictor(sub(d, "pred_"), int(d["n_layers"])); j = do.Joint(sub(d, "joint_"))
    for i, l in enumerate(lens):
        ref, margin = do.greedy_search(p, j, encs[i], l, n_steps=int(d["n_steps"]), return_margin=True)
        if margin > 1e-3:
            assert hyps[i] == ref, i
        else:                                   # an unlucky near-tie in a derived stream: compare the common prefix only
            assert hyps[i][:3] == ref[:3]


def test_lstm_predictor_without_biases_steps_like_the_module_graph():
    """RNNPredictor(bias=False) (predictor.py:66): the HIP step against the plain module graph on the same weights."""
    import wenet_celoss_amd as w
    torch.manual_seed(3)
    pred = w.RNNPredictor(30, 12, 10, 0.0, 16, 2, bias=False, dropout=0.0).to(DEV).eval()
    assert not any("bias" in n for n, _ in pred.rnn.named_parameters())
    N = 4
    cache = pred.init_state(N, device=torch.device(DEV))
    ref_cache = [c.clone() for c in cache]
    pad = torch.zeros(N, 1, device=DEV)
    for s in range(5):
        tok = torch.randint(0, 30, (N, 1), device=DEV)
        out, cache = pred.forward_step(tok, pad, cache)
        with torch.no_grad():
            ro, ref_cache = pred._export_step(tok, pad, ref_cache)
        torch.testing.assert_close(out, ro, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(cache[0], ref_cache[0], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(cache[1], ref_cache[1], rtol=1e-4, atol=1e-5)


def build_history_predictor(d, prefix, kind, history, n_head=4, act=None, bias=None):
    import wenet_celoss_amd as w
    pw = sub(d, prefix)
    V, D = pw["embed.weight"].shape
    if kind == "embedding":
        pred = w.EmbeddingPredictor(V, D, 0.1, n_head, history, act or "swish", "pos_embed.bias" in pw)
    else:
        pred = w.ConvPredictor(V, D, 0.1, history, act or "relu", "conv.bias" in pw)
    pred.load_state_dict({k: torch.tensor(v) for k, v in pw.items()})
    return pred.to(DEV).eval()


@pytest.mark.parametrize("path", names("predictor_var_*.npz"))
def test_stateless_predictor_step_matches_reference(path):
    """EmbeddingPredictor / ConvPredictor forward_step on the device (history in the decoder's state slots) against the
    reference modules' step outputs and caches (predictor.py:325-372, :455-481), plus the training forward."""
    d = np.load(path)
    pred = build_history_predictor(d, "w_", str(d["kind"]), int(d["history"]), int(d["n_head"]), str(d["act"]))
    steps, N = d["toks"].shape
    cache = pred.init_state(N, device=torch.device(DEV))
    assert cache[0].shape == (N, int(d["history"]), pred.embed_size)
    for s in range(steps):
        out, cache = pred.forward_step(torch.tensor(d["toks"][s], device=DEV).reshape(N, 1), torch.zeros(N, 1, device=DEV), cache)
        np.testing.assert_allclose(out.cpu().numpy(), d["outs"][s], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(cache[0].cpu().numpy(), d["hist"][s], rtol=0, atol=0)     # embeddings, copied
    back = pred.cache_to_batch(pred.batch_to_cache(cache))
    assert torch.equal(back[0], cache[0])
    full = pred(torch.tensor(d["toks"].T.copy(), device=DEV))
    np.testing.assert_allclose(full.detach().cpu().numpy(), d["full"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("kind,D,history,heads", [("embedding", 300, 2, 4), ("embedding", 520, 4, 8), ("conv", 300, 3, 0),
                                                  ("conv", 1024, 1, 0), ("embedding", 8, 1, 1)])
def test_stateless_predictor_step_wide_embeddings_against_the_module_graph(kind, D, history, heads):
    """Widths beyond one element per thread (D > 256), the longest history (4 slots) and the smallest sizes: the HIP
    step against the plain module graph (`_export_step`, library ops) on the same weights, several lanes at once."""
    import wenet_celoss_amd as w
    torch.manual_seed(D + history)
    V, N = 50, 5
    pred = (w.EmbeddingPredictor(V, D, 0.0, heads, history, "swish") if kind == "embedding"
            else w.ConvPredictor(V, D, 0.0, history, "relu", True)).to(DEV).eval()
    with torch.no_grad():
        pred.norm.weight.add_(torch.randn(D, device=DEV) * 0.2)
        pred.norm.bias.add_(torch.randn(D, device=DEV) * 0.2)
    cache = pred.init_state(N, device=torch.device(DEV))
    ref = [c.clone() for c in cache]
    pad = torch.zeros(N, 1, device=DEV)
    for s_ in range(history + 3):
        tok = torch.randint(0, V, (N, 1), device=DEV)
        out, cache = pred.forward_step(tok, pad, cache)
        with torch.no_grad():
            ro, ref = pred._export_step(tok, pad, ref)
        torch.testing.assert_close(out, ro, rtol=1e-4, atol=2e-5)
        assert torch.equal(cache[0], ref[0])


def test_stateless_predictor_streams_batched_equal_single():
    """Several streams with different lengths through the greedy kernels with a ConvPredictor: every stream decodes as it
    does alone (lanes that wait for a blank keep their pending history; predication per lane)."""
    import wenet_celoss_amd as w
    d = np.load(names("decode_var_2.npz")[0])
    pred = build_history_predictor(d, "pred_", str(d["kind"]), 2)
    jw = {k: v for k, v in sub(d, "joint_").items() if k != "act"}
    J, E = jw["enc_ffn.weight"].shape
    V = jw["ffn_out.weight"].shape[0]
    joint = w.TransducerJoint(V, E, pred.embed_size, J, activation=str(d["joint_act"])).to(DEV).eval()
    joint.load_state_dict({k: torch.tensor(v) for k, v in jw.items()})
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    enc0 = d["enc"][0]
    T = enc0.shape[0]
    lens = [T, T // 2, 3, T - 4, T]
    encs = np.zeros((len(lens), T, enc0.shape[1]), np.float32)
    for i, l in enumerate(lens):
        encs[i, :l] = enc0[:l] if i % 2 == 0 else enc0[T - l:]
    encs[-1] = enc0[::-1]
    n_steps = int(d["n_steps"])
    batch = w.basic_greedy_search(model, torch.tensor(encs, device=DEV), torch.tensor(lens), n_steps=n_steps)
    assert batch[0] == list(d["hyp"])
    for i, l in enumerate(lens):
        single = w.basic_greedy_search(model, torch.tensor(encs[i:i + 1, :l], device=DEV), torch.tensor([l]), n_steps=n_steps)
        assert single[0] == batch[i], i


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("path", names("decode_var_*.npz"))
def test_stateless_predictors_and_other_joiner_activations_decode_like_the_reference(path, use_graph):
    """Greedy loop and PrefixBeamSearch of the reference over EmbeddingPredictor / ConvPredictor with relu / swish / tanh
    joiners: same tokens, same n-best order and scores."""
    import wenet_celoss_amd as w
    d = np.load(path)
    jw = {k: v for k, v in sub(d, "joint_").items() if k != "act"}          # "joint_act" is the activation's name
    J, E = jw["enc_ffn.weight"].shape
    V = jw["ffn_out.weight"].shape[0]
    if str(d["kind"]) == "lstm_nobias":                  # RNNPredictor(bias=False) + post-join Linear + hardtanh
        pw = sub(d, "pred_")
        P = pw["projection.weight"].shape[0]
        pred = w.RNNPredictor(V, pw["embed.weight"].shape[1], P, 0.1, pw["rnn.weight_hh_l0"].shape[1], 2, bias=False)
        pred.load_state_dict({k: torch.tensor(v) for k, v in pw.items()})
        pred = pred.to(DEV).eval()
    else:
        pred = build_history_predictor(d, "pred_", str(d["kind"]), 2)
        P = pred.embed_size
    joint = w.TransducerJoint(V, E, P, J, activation=str(d["joint_act"]), postjoin_linear=bool(d["postjoin"])).to(DEV).eval()
    joint.load_state_dict({k: torch.tensor(v) for k, v in jw.items()})
    ctc = w.CTC(V, E).to(DEV).eval()
    ctc.load_state_dict({k[4:]: torch.tensor(d[k]) for k in d.files if k.startswith("ctc_ctc_lo")})
    enc = torch.tensor(d["enc"], device=DEV)
    T, n_steps = int(d["T"]), int(d["n_steps"])
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    from wenet_celoss_amd.decoder import DecoderCache
    model._decoder_cache = DecoderCache()
    dec = model._decoder_cache.get(pred, joint, lanes=1, utts=1, tmax=T, max_hyp=T * n_steps, beam=1)
    dec.set_graph(use_graph)
    hyps = w.basic_greedy_search(model, enc, torch.tensor(T), n_steps=n_steps)
    assert hyps == [list(d["hyp"])]
    bs = w.PrefixBeamSearch(None, pred, joint, ctc, 0)
    beam = bs.search_encoded(enc, torch.tensor([T], dtype=torch.int32), beam_size=int(d["beam"]))[0]
    assert len(beam) == len(d["beam_scores"])
    for k, sq in enumerate(beam):
        assert sq.hyp == list(d["beam_hyps"][k][: d["beam_lens"][k]]), k
        assert sq.score == pytest.approx(d["beam_scores"][k], rel=1e-5)


@pytest.mark.parametrize("path", names("prefix_beam_*.npz"))
def test_prefix_beam_matches_reference(path):
    import wenet_celoss_amd as w
    d = np.load(path)
    pred, joint, ctc = build_modules(d, with_ctc=True)

    class Enc(torch.nn.Module):
        def forward(self, speech, lens, a=-1, b=-1):
            return torch.tensor(d["enc"], device=DEV), torch.ones(1, 1, int(d["T"]), dtype=torch.bool, device=DEV)

    bs = w.PrefixBeamSearch(Enc(), pred, joint, ctc, 0)
    beam, enc_out = bs.prefix_beam_search(torch.zeros(1, int(d["T"]), 80, device=DEV), torch.tensor([int(d["T"])]),
                                          beam_size=int(d["beam"]), ctc_weight=float(d["ctc_weight"]),
                                          transducer_weight=float(d["transducer_weight"]))
    assert len(beam) == len(d["scores"])
    for k, s in enumerate(beam):                # exact hypotheses, exact order
        assert s.hyp == list(d["hyps"][k][: d["hyp_lens"][k]]), k
        assert s.score == pytest.approx(d["scores"][k], rel=1e-5)


def test_prefix_beam_batched_equals_single():
    import wenet_celoss_amd as w
    d = np.load(names("prefix_beam_1.npz")[0])
    pred, joint, ctc = build_modules(d, with_ctc=True)
    enc0 = torch.tensor(d["enc"], device=DEV)
    T = int(d["T"])
    encs = torch.cat([enc0, enc0.flip(1), enc0 * 0.5], 0)
    lens = torch.tensor([T, T - 5, 9], dtype=torch.int32)
    bs = w.PrefixBeamSearch(None, pred, joint, ctc, 0)
    batch = bs.search_encoded(encs, lens, beam_size=4)
    for b in range(3):
        single = bs.search_encoded(encs[b:b + 1, :int(lens[b])].contiguous(), lens[b:b + 1], beam_size=4)[0]
        assert [s.hyp for s in single] == [s.hyp for s in batch[b]]
        np.testing.assert_allclose([s.score for s in single], [s.score for s in batch[b]], rtol=1e-6)
    p = do.Predictor(sub(d, "pred_"), int(d["n_layers"])); j = do.Joint(sub(d, "joint_"))
    ref = do.prefix_beam_search(p, j, {k[4:]: d[k] for k in d.files if k.startswith("ctc_ctc_lo")},
                                encs[1].cpu().numpy(), T - 5, beam_size=4)
    assert [s["hyp"] for s in ref] == [s.hyp for s in batch[1]]


def test_config3_shape_streams_match_oracle():
    """BASELINE config 3 shape: 64 streams, V=5000, E=P=256, J=512, LSTM 2x256, n_steps=64, T=32 (two chunks).
    Every stream's tokens are compared with the numpy oracle; streams whose smallest top-1/top-2 margin is
    below 1e-4 (ambiguous under fp32 summation order) are reported, not compared."""
    import wenet_celoss_amd as w
    torch.manual_seed(5)
    V, E, P, J, H, L, N, T = 5000, 256, 256, 512, 256, 2, 64, 32
    pred = w.RNNPredictor(V, P, P, 0.1, H, L).to(DEV).eval()
    joint = w.TransducerJoint(V, E, P, J).to(DEV).eval()
    with torch.no_grad():                        # spread the logits (clear top-1/top-2 margins) and favour blank
        joint.ffn_out.weight *= 10
        joint.ffn_out.bias[0] += 13.0
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    enc = torch.randn(N, T, E, device=DEV)
    lens = torch.randint(8, T + 1, (N,)); lens[0] = T
    hyps = w.basic_greedy_search(model, enc, lens, n_steps=64)
    p = do.Predictor({k: v.detach().cpu().numpy() for k, v in pred.state_dict().items()}, L)
    j = do.Joint({k: v.detach().cpu().numpy() for k, v in joint.state_dict().items()})
    compared = 0
    for i in range(0, N, 4):
        ref, margin = do.greedy_search(p, j, enc[i].cpu().numpy(), int(lens[i]), n_steps=64, return_margin=True)
        if margin > 1e-4:
            assert hyps[i] == ref, (i, margin)
            compared += 1
    assert compared >= 8
    assert sum(len(h) for h in hyps) > N         # something was emitted


def test_config5_full_shape_prefix_beam_matches_oracle():
    """BASELINE config 5 at full shape: one call with B=16 utterances, T=1500 encoder frames, V=5000, E=P=256,
    J=512, LSTM 2x256, beam 8, ctc/transducer weights (0.3, 0.7) -- prefix_beam_search.py:42-148 for every
    utterance.  The ten utterances of at most 150 frames are re-decoded whole by the numpy oracle, the five long ones
    (400 .. 1500 frames) over their clear prefix (below); n-best hypotheses and
    their order must be identical and scores agree to 1e-5 relative for every one whose decisions were clear:
    random weights give near-degenerate beams (hypotheses whose scores differ by less than an fp32 ulp of the
    score), so an utterance is compared only if the smallest gap between neighbouring candidates around the prune
    boundary, over all its frames, exceeds 1e-4 (`return_margin`); at least four must qualify (seven do for this
    seed on the CPU).  The long utterances are checked for shape, order and against a single-utterance call."""
    import wenet_celoss_amd as w
    torch.manual_seed(0)
    V, E, P, J, H, L, B, T, beam = 5000, 256, 256, 512, 256, 2, 16, 1500, 8
    pred = w.RNNPredictor(V, P, P, 0.1, H, L).eval()
    joint = w.TransducerJoint(V, E, P, J).eval()
    ctc = w.CTC(V, E).eval()
    with torch.no_grad():
        joint.ffn_out.weight *= 10
        joint.ffn_out.bias[0] += 13
        ctc.ctc_lo.weight *= 10
        ctc.ctc_lo.bias[0] += 13
    enc = torch.randn(B, T, E)
    lens = [1500, 1500, 1000, 700, 400, 250, 150, 120, 100, 90, 80, 70, 60, 50, 40, 30]
    p = do.Predictor({k: v.detach().numpy() for k, v in pred.state_dict().items()}, L)
    j = do.Joint({k: v.detach().numpy() for k, v in joint.state_dict().items()})
    cw = {k: v.detach().numpy() for k, v in ctc.state_dict().items()}
    pred, joint, ctc = pred.to(DEV), joint.to(DEV), ctc.to(DEV)
    bs = w.PrefixBeamSearch(torch.nn.Identity(), pred, joint, ctc, 0)
    enc_d = enc.to(DEV)
    res = bs.search_encoded(enc_d, torch.tensor(lens), beam, 0.3, 0.7)
    assert len(res) == B
    for b in range(B):
        assert len(res[b]) == beam
        sc = [s.score for s in res[b]]
        assert sc == sorted(sc, reverse=True) and all(s.hyp[0] == 0 and len(s.hyp) <= lens[b] + 1 for s in res[b])
    compared, margins = 0, []
    for i in range(6, B):
        ref, margin = do.prefix_beam_search(p, j, cw, enc[i].numpy(), lens[i], beam_size=beam, ctc_weight=0.3,
                                            transducer_weight=0.7, return_margin=True)
        margins.append(margin)
        if margin > 1e-4:
            assert [s["hyp"] for s in ref] == [s.hyp for s in res[i]], (i, margin)
            np.testing.assert_allclose([s.score for s in res[i]], [s["score"] for s in ref], rtol=1e-5)
            compared += 1
    assert compared >= 4, margins
    # The long utterances (T = 1500, 1500, 1000, 700, 400 -- the config's defining length).  Random weights give beams
    # whose entries sit within 1e-2 of each other, so a decision with a gap below 1e-4 turns up every ~100 frames and
    # from there on two fp32 implementations may legitimately keep different tails.  Up to that frame everything must
    # agree: the oracle stops in front of the first frame whose prune margin is < 1e-4 (`stop_below`), the HIP search
    # decodes the same frames, and n-best, order and scores are compared as above.
    clear = []
    for i in range(5):
        ref, margin, f = do.prefix_beam_search(p, j, cw, enc[i].numpy(), lens[i], beam_size=beam, ctc_weight=0.3,
                                               transducer_weight=0.7, stop_below=1e-4)
        clear.append(f)
        assert f >= 30, (i, f)
        got = bs.search_encoded(enc_d[i:i + 1, :f].contiguous(), torch.tensor([f]), beam, 0.3, 0.7)[0]
        assert [s["hyp"] for s in ref] == [s.hyp for s in got], (i, f, margin)
        np.testing.assert_allclose([s.score for s in got], [s["score"] for s in ref], rtol=1e-5)
    assert sum(clear) >= 500, clear                   # 42 + 60 + 67 + 277 + 149 frames for this seed on the CPU
    # a full-length utterance decoded alone gives what it gave inside the batch
    single = bs.search_encoded(enc_d[1:2].contiguous(), torch.tensor([lens[1]]), beam, 0.3, 0.7)[0]
    assert [s.hyp for s in single] == [s.hyp for s in res[1]]
    np.testing.assert_allclose([s.score for s in single], [s.score for s in res[1]], rtol=1e-6)


def test_config3_chunked_streams_match_reference_loop():
    """BASELINE config 3 in its deployment form: reset_cache(64) + forward_greedy_search over chunks of 16 encoder
    frames (67 fbank frames per chunk, asr_model.py:580-581), V=5000, E=P=256, J=512, LSTM 2x256, n_steps=64.
    EVERY stream is compared, chunk by chunk, with the numpy restatement of the reference's streaming loop
    ("transducer ref.py":541-606; pinned to the reference by the greedy_stream_* fixtures), and -- with the pending
    predictor state kept -- with the offline loop on the concatenated frames.  Streams whose smallest top-1 / top-2
    log-prob gap is below 1e-4 (ambiguous under fp32 summation order) are counted, not compared."""
    import wenet_celoss_amd as w
    torch.manual_seed(7)
    V, E, P, J, H, L, N, C, n_chunks = 5000, 256, 256, 512, 256, 2, 64, 16, 3
    pred = w.RNNPredictor(V, P, P, 0.1, H, L).to(DEV).eval()
    joint = w.TransducerJoint(V, E, P, J).to(DEV).eval()
    with torch.no_grad():                        # spread the logits (clear margins) and favour blank
        joint.ffn_out.weight *= 10
        joint.ffn_out.bias[0] += 12.0
    m = w.Transducer(V, 0, torch.nn.Identity(), pred, joint, ctc_weight=0.0, transducer_weight=1.0, hw_weight=0.0)
    enc = torch.randn(N, C * n_chunks, E, device=DEV)
    chunk_lens = [torch.full((N,), C), torch.full((N,), C), torch.randint(5, C + 1, (N,))]
    p = do.Predictor({k: v.detach().cpu().numpy() for k, v in pred.state_dict().items()}, L)
    j = do.Joint({k: v.detach().cpu().numpy() for k, v in joint.state_dict().items()})
    enc_np = enc.cpu().numpy()
    for quirk in (True, False):
        m.reset_cache(N, chunk_frames=C, n_steps=64)
        got = [[] for _ in range(N)]
        for c in range(n_chunks):
            res = m.forward_greedy_search(enc[:, c * C:(c + 1) * C].contiguous(), chunk_lens[c], n_steps=64,
                                          reference_new_cache=quirk)
            for i in range(N):
                got[i].append(res[i])
        compared = 0
        emitted = 0
        for i in range(N):
            ref = do.StreamingGreedy(p, j)
            want = [ref.forward_greedy_search(enc_np[i, c * C:(c + 1) * C], int(chunk_lens[c][i]), n_steps=64,
                                              reference_new_cache=quirk) for c in range(n_chunks)]
            if ref.min_margin > 1e-4:
                assert got[i] == want, (quirk, i, ref.min_margin)
                compared += 1
                emitted += sum(len(x) for x in want)
            if not quirk and ref.min_margin > 1e-4:  # chunked with the pending state kept == offline on the valid frames
                cat = np.concatenate([enc_np[i, c * C:c * C + int(chunk_lens[c][i])] for c in range(n_chunks)])
                off = do.greedy_search(p, j, cat, cat.shape[0], n_steps=64)
                assert [t for ch in got[i] for t in ch] == off, i
        assert compared >= 48, compared                # at most a quarter of the streams may be ambiguous
        assert emitted > compared                      # the scenario emits tokens


def test_config3_chunked_streams_unscaled_weights_prefix_rule():
    """Config 3 again WITHOUT the engineered weights of the test above (no x10 on ffn_out, no blank bias): plain randomly
    initialised modules, whose logits are nearly flat (top-1 / top-2 gaps of a few 1e-2), so every decision emits and
    every frame runs into the n_steps cap.  The rule is the prefix rule: a stream's tokens are compared with the
    reference loop up to the first decision whose top-1 / top-2 log-prob gap is below 1e-4 (from there on two fp32
    implementations may differ); at least 24 of the 32 compared streams must be clear for 20 tokens or more, and at least
    8 to the end."""
    import wenet_celoss_amd as w
    torch.manual_seed(17)
    V, E, P, J, H, L, N, C, n_chunks, n_steps = 5000, 256, 256, 512, 256, 2, 64, 16, 3, 4
    pred = w.RNNPredictor(V, P, P, 0.1, H, L).to(DEV).eval()
    joint = w.TransducerJoint(V, E, P, J).to(DEV).eval()
    m = w.Transducer(V, 0, torch.nn.Identity(), pred, joint, ctc_weight=0.0, transducer_weight=1.0, hw_weight=0.0)
    enc = torch.randn(N, C * n_chunks, E, device=DEV)
    chunk_lens = [torch.full((N,), C), torch.randint(5, C + 1, (N,)), torch.full((N,), C)]
    p = do.Predictor({k: v.detach().cpu().numpy() for k, v in pred.state_dict().items()}, L)
    j = do.Joint({k: v.detach().cpu().numpy() for k, v in joint.state_dict().items()})
    enc_np = enc.cpu().numpy()
    m.reset_cache(N, chunk_frames=C, n_steps=n_steps)
    got = [[] for _ in range(N)]
    for c in range(n_chunks):
        res = m.forward_greedy_search(enc[:, c * C:(c + 1) * C].contiguous(), chunk_lens[c], n_steps=n_steps)
        for i in range(N):
            got[i] += res[i]
    compared_tokens, fully_clear, long_prefix = 0, 0, 0
    for i in range(0, N, 2):
        ref = do.StreamingGreedy(p, j)
        want = []
        for c in range(n_chunks):
            want += ref.forward_greedy_search(enc_np[i, c * C:(c + 1) * C], int(chunk_lens[c][i]), n_steps=n_steps)
        k = len(want) if ref.clear_tokens is None else ref.clear_tokens
        assert got[i][:k] == want[:k], (i, k, ref.min_margin)
        compared_tokens += k
        long_prefix += k >= 20
        if ref.clear_tokens is None:
            assert got[i] == want, i
            fully_clear += 1
    print('unscaled config 3: tokens compared', compared_tokens, 'streams clear to the end', fully_clear, 'with >= 20 clear tokens', long_prefix)
    assert compared_tokens >= 32 * 40 and fully_clear >= 8 and long_prefix >= 24, (compared_tokens, fully_clear, long_prefix)


def test_streaming_chunks_equal_offline_and_reference_quirk():
    """Chunk-synchronous greedy ("transducer ref.py":541-606): with the pending predictor state kept, decoding
    chunk by chunk equals decoding the concatenated frames; with `reference_new_cache=True` it equals the
    restated reference loop (which resets new_cache to the committed cache at every chunk start)."""
    import wenet_celoss_amd as w
    d = np.load(names("greedy_core_1.npz")[0])
    pred, joint, _ = build_modules(d)
    m = w.Transducer(64, 0, torch.nn.Identity(), pred, joint, ctc_weight=0.0, transducer_weight=1.0, hw_weight=0.0)
    enc = torch.tensor(d["enc"], device=DEV)                # (1, 60, 16)
    T, n_steps = int(d["T"]), int(d["n_steps"])
    chunks = [(0, 16), (16, 32), (32, 37), (37, 60)]
    m.reset_cache(1, chunk_frames=32, n_steps=n_steps)
    got = []
    for a, b in chunks:
        got += m.forward_greedy_search(enc[:, a:b].contiguous(), torch.tensor([b - a]), n_steps=n_steps,
                                       reference_new_cache=False)
    assert got == list(d["hyp"])                            # == the reference's offline tokens for the whole utterance
    p = do.Predictor(sub(d, "pred_"), int(d["n_layers"])); j = do.Joint(sub(d, "joint_"))
    for quirk in (True, False):
        ref = do.StreamingGreedy(p, j)
        m.reset_cache(1, chunk_frames=32, n_steps=n_steps)
        for a, b in chunks:
            r = ref.forward_greedy_search(d["enc"][0, a:b], b - a, n_steps=n_steps, reference_new_cache=quirk)
            g = m.forward_greedy_search(enc[:, a:b].contiguous(), torch.tensor([b - a]), n_steps=n_steps,
                                        reference_new_cache=quirk)
            assert g == r, (quirk, a, b)
    # several streams at once, different chunk lengths per stream
    m.reset_cache(3, chunk_frames=32, n_steps=n_steps)
    encs = torch.cat([enc, enc.flip(1), enc * 0.5], 0)
    outs = [[], [], []]
    for a, b in chunks:
        lens = torch.tensor([b - a, max(b - a - 3, 1), b - a])
        res = m.forward_greedy_search(encs[:, a:b].contiguous(), lens, n_steps=n_steps, reference_new_cache=False)
        for i in range(3):
            outs[i] += res[i]
    assert outs[0] == list(d["hyp"])


@pytest.mark.parametrize("path", names("greedy_stream_*.npz"))
def test_streaming_greedy_matches_reference_chunks(path):
    """reset_cache() + forward_greedy_search(chunk) return, chunk by chunk, exactly the tokens that the reference's
    own TorchScript exports ("wenet/transducer/transducer ref.py":541-606) returned for the same chunking
    (fixtures from tests/golden/make_golden.py::gen_greedy_stream) -- including its `new_cache = self.cache` reset
    at every chunk start; chunk boundaries right after an emission and the n_steps cap are among the cases."""
    import wenet_celoss_amd as w
    d = np.load(path)
    pred, joint, _ = build_modules(d)
    m = w.Transducer(64, 0, torch.nn.Identity(), pred, joint, ctc_weight=0.0, transducer_weight=1.0, hw_weight=0.0)
    enc = torch.tensor(d["enc"], device=DEV)
    n_steps = int(d["n_steps"])
    for graph in (True, False):
        m.reset_cache(1, chunk_frames=int(max(d["chunk_sizes"])), n_steps=n_steps)
        a, k = 0, 0
        for n, cnt in zip(d["chunk_sizes"].tolist(), d["chunk_token_counts"].tolist()):
            got = m.forward_greedy_search(enc[:, a:a + n].contiguous(), torch.tensor([n]), n_steps=n_steps)
            if a == 0:
                m._stream_dec.set_graph(graph)
            assert got == d["chunk_tokens"][k:k + cnt].tolist(), (path, a, n, graph)
            a += n
            k += cnt
    # the same utterance as one of several streams (other streams must not disturb it)
    m.reset_cache(3, chunk_frames=int(max(d["chunk_sizes"])), n_steps=n_steps)
    encs = torch.cat([enc.flip(1), enc, enc * 0.5], 0)
    a, k = 0, 0
    for n, cnt in zip(d["chunk_sizes"].tolist(), d["chunk_token_counts"].tolist()):
        res = m.forward_greedy_search(encs[:, a:a + n].contiguous(), torch.tensor([n, n, n]), n_steps=n_steps)
        assert res[1] == d["chunk_tokens"][k:k + cnt].tolist(), (path, a, n)
        a += n
        k += cnt


@pytest.mark.parametrize("path", names("greedy_fork_*.npz"))
def test_fork_hotword_greedy_matches_reference(path):
    """The fork's greedy variants (greedy_search.py:34-176 'pred', :297-430 'both'; context filter on/off):
    tokens, the edit distance and (for 'pred') the gate trace equal what the reference's loops produced with the
    same stand-in hot-word module (tests/bias_stub.py)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bias_stub import TinyBias
    import wenet_celoss_amd as w
    d = np.load(path)
    pred, joint, _ = build_modules(d)
    bias = TinyBias(64, 16, 16, seed=int(d["seed"]), gate_bias=float(d["gate_bias"])).to(DEV).eval()
    mode = str(d["mode"])
    m = w.Transducer(64, 0, torch.nn.Identity(), pred, joint, context_bias=bias, ctc_weight=0.0, transducer_weight=1.0,
                     loss_mode=mode)
    enc = torch.tensor(d["enc"], device=DEV)
    ctx, ctx_len, labels = torch.tensor(d["ctx"]), torch.tensor(d["ctx_len"]), torch.tensor(d["labels"])
    fn = w.basic_greedy_search_both if mode == "both" else w.basic_greedy_search_hw
    out = fn(m, enc, torch.tensor(int(d["T"])), ctx, ctx_len, n_steps=64, context_filter_state=str(d["filt"]),
             context_decoder_labels_padded=labels)
    assert out[0] == [list(d["hyp"])]
    assert out[1] == float(d["dist"])
    if mode == "pred":
        assert list(out[2]) == list(d["trace"])
    # through the model entry point (transducer.py:515-598)

    class Enc(torch.nn.Module):
        def forward(self, speech, lens, a=-1, b=-1):
            return enc, torch.ones(1, 1, int(d["T"]), dtype=torch.bool, device=DEV)
    m.encoder = Enc()
    hyps, dist = m.greedy_search(torch.zeros(1, int(d["T"]), 8, device=DEV), torch.tensor([int(d["T"])]), context_list=ctx,
                                 context_lengths=ctx_len, context_filter_state=str(d["filt"]),
                                 context_decoder_labels_padded=labels)
    assert hyps == [list(d["hyp"])] and dist == float(d["dist"])


def test_stream_state_is_invalidated_by_other_searches_on_the_handle():
    """A chunk continuation after the handle's lanes were reused (predictor step / beam search) must be refused,
    not decoded from clobbered caches."""
    from wenet_celoss_amd.decoder import DeviceDecoder
    d = np.load(names("greedy_core_1.npz")[0])
    pred, joint, _ = build_modules(d)
    dec = DeviceDecoder(pred, joint, max_lanes=4, max_utt=2, tmax=32, max_hyp=256, max_beam=2)
    enc = torch.tensor(d["enc"], device=DEV)[:, :16].contiguous()
    lens = torch.tensor([16])
    first = dec.greedy_chunk(enc, lens, n_steps=4, reset=True)
    again = dec.greedy_chunk(enc, lens, n_steps=4, reset=False)          # a legal continuation
    assert isinstance(first, list) and isinstance(again, list)
    L, H = dec.dims["L"], dec.dims["H"]
    dec.predictor_step(torch.tensor([1]), torch.zeros(L, 1, H, device=DEV), torch.zeros(L, 1, H, device=DEV))
    with pytest.raises(RuntimeError, match="no stream state"):
        dec.greedy_chunk(enc, lens, n_steps=4, reset=False)
    dec.greedy_chunk(enc, lens, n_steps=4, reset=True)
    V = dec.dims["V"]
    dec.prefix_beam(enc, lens, torch.log_softmax(torch.randn(1, 16, V, device=DEV), -1), 2, 0.3, 0.7)
    with pytest.raises(RuntimeError, match="no stream state"):
        dec.greedy_chunk(enc, lens, n_steps=4, reset=False)


@pytest.mark.parametrize("look", [0, 2, 3, 4])        # 0: chosen per replay from the share of blank decisions
@pytest.mark.parametrize("path", names("greedy_core_*.npz"))
def test_greedy_lookahead_keeps_reference_tokens(path, look):
    """wr_decoder_set_lookahead: several encoder frames per micro-step, same tokens as the reference's loop (the
    fixtures include n_steps limits that force frame advances after emissions)."""
    import wenet_celoss_amd as w
    d = np.load(path)
    pred, joint, _ = build_modules(d)
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    enc = torch.tensor(d["enc"], device=DEV)
    T, n_steps = int(d["T"]), int(d["n_steps"])
    from wenet_celoss_amd.decoder import DecoderCache
    model._decoder_cache = DecoderCache()
    dec = model._decoder_cache.get(pred, joint, lanes=1, utts=1, tmax=T, max_hyp=T * n_steps, beam=1)
    dec.set_lookahead(look)
    assert w.basic_greedy_search(model, enc, torch.tensor(T), n_steps=n_steps) == [list(d["hyp"])]
    with pytest.raises(RuntimeError, match="frames"):
        dec.set_lookahead(5)


@pytest.mark.parametrize("look", [0, 2, 4])
def test_greedy_lookahead_batched_and_streaming(look):
    """Look-ahead with ragged streams and across chunk boundaries: identical to look-ahead 1."""
    import wenet_celoss_amd as w
    from wenet_celoss_amd.decoder import DeviceDecoder
    d = np.load(names("greedy_core_1.npz")[0])
    pred, joint, _ = build_modules(d)
    enc0 = torch.tensor(d["enc"], device=DEV)[0]
    T = enc0.shape[0]
    lens = torch.tensor([T, T // 2, 1, T - 7, 13])
    encs = torch.stack([enc0, enc0.flip(0), enc0 * 0.5, enc0.roll(5, 0), enc0 * 2.0])
    n_steps = int(d["n_steps"])
    dec = DeviceDecoder(pred, joint, max_lanes=5, max_utt=5, tmax=T, max_hyp=T * n_steps)
    base = dec.greedy(encs, lens, n_steps=n_steps)
    chunks = [(0, 16), (16, 21), (21, T)]
    base_chunks = []
    for a, b in chunks:
        cl = (lens - a).clamp(min=0, max=b - a)
        base_chunks.append(dec.greedy_chunk(encs[:, a:b].contiguous(), cl.clamp(min=1), n_steps=n_steps, reset=a == 0,
                                            reference_new_cache=False))
    dec.set_lookahead(look)
    assert dec.greedy(encs, lens, n_steps=n_steps) == base
    for i, (a, b) in enumerate(chunks):
        cl = (lens - a).clamp(min=0, max=b - a)
        got = dec.greedy_chunk(encs[:, a:b].contiguous(), cl.clamp(min=1), n_steps=n_steps, reset=a == 0,
                               reference_new_cache=False)
        assert got == base_chunks[i], (a, b)


def test_greedy_many_streams_equal_small_batches():
    """300 streams in one call (lane tiles beyond the first 128) give each stream the tokens it gets in a small
    batch -- streams are independent."""
    from wenet_celoss_amd.decoder import DeviceDecoder
    d = np.load(names("greedy_core_1.npz")[0])
    pred, joint, _ = build_modules(d)
    enc0 = torch.tensor(d["enc"], device=DEV)[0]
    T, n_steps = enc0.shape[0], int(d["n_steps"])
    g = torch.Generator(device=DEV).manual_seed(3)
    N = 300
    scale = 0.5 + torch.rand(N, 1, 1, device=DEV, generator=g)
    encs = enc0[None] * scale
    encs[1::3] = encs[1::3].flip(1)
    lens = torch.randint(1, T + 1, (N,), generator=torch.Generator().manual_seed(4))
    big = DeviceDecoder(pred, joint, max_lanes=N, max_utt=N, tmax=T, max_hyp=T * n_steps)
    small = DeviceDecoder(pred, joint, max_lanes=20, max_utt=20, tmax=T, max_hyp=T * n_steps)
    got = big.greedy(encs, lens, n_steps=n_steps)
    for a in range(0, N, 100):
        ref = small.greedy(encs[a:a + 20].contiguous(), lens[a:a + 20], n_steps=n_steps)
        assert got[a:a + 20] == ref, a
    with pytest.raises(RuntimeError, match="max_lanes"):
        DeviceDecoder(pred, joint, max_lanes=2000, max_utt=1, tmax=T)


def test_prefix_beam_many_utterances_equal_single():
    """40 utterances x beam 5 = 200 lanes in one call: every utterance gets the hypotheses of its own single run."""
    import wenet_celoss_amd as w
    d = np.load(names("prefix_beam_1.npz")[0])
    pred, joint, ctc = build_modules(d, with_ctc=True)
    enc0 = torch.tensor(d["enc"], device=DEV)
    T = int(d["T"])
    g = torch.Generator(device=DEV).manual_seed(12)
    B = 40
    encs = enc0 * (0.5 + torch.rand(B, 1, 1, device=DEV, generator=g))
    encs[::2] = encs[::2].flip(1)
    lens = torch.randint(3, T + 1, (B,), generator=torch.Generator().manual_seed(13)).to(torch.int32)
    bs = w.PrefixBeamSearch(None, pred, joint, ctc, 0)
    batch = bs.search_encoded(encs, lens, beam_size=5)
    for b in (0, 7, 25, 39):
        single = bs.search_encoded(encs[b:b + 1, :int(lens[b])].contiguous(), lens[b:b + 1], beam_size=5)[0]
        assert [s.hyp for s in single] == [s.hyp for s in batch[b]], b
        np.testing.assert_allclose([s.score for s in single], [s.score for s in batch[b]], rtol=1e-6)


def test_decoder_cache_sees_data_edits_and_survives_copies():
    """The handle holds re-laid copies of the weights.  An in-place edit through `.data` (EMA, weight averaging)
    changes neither data_ptr nor the version counter; the cache's content fingerprint must still rebuild the handle.
    A model that has decoded can be deep-copied and pickled (the native handle is dropped, rebuilt lazily)."""
    import copy
    import io
    import wenet_celoss_amd as w
    d = np.load(names("greedy_core_0.npz")[0])
    pred, joint, _ = build_modules(d)
    m = w.Transducer(64, 0, torch.nn.Identity(), pred, joint, ctc_weight=0.0, transducer_weight=1.0, hw_weight=0.0)
    enc = torch.tensor(d["enc"], device=DEV)
    T, n_steps = int(d["T"]), int(d["n_steps"])
    first = w.basic_greedy_search(m, enc, torch.tensor(T), n_steps=n_steps)
    assert first == [list(d["hyp"])]
    dec0 = m._decoder_cache._dec
    assert w.basic_greedy_search(m, enc, torch.tensor(T), n_steps=n_steps) == first
    assert m._decoder_cache._dec is dec0                    # unchanged weights: the handle is reused
    ver = joint.ffn_out.bias._version
    joint.ffn_out.bias.data[0] -= 40.0                      # blank can no longer win
    assert joint.ffn_out.bias._version == ver               # ... and nothing but the content says so
    second = w.basic_greedy_search(m, enc, torch.tensor(T), n_steps=n_steps)
    assert m._decoder_cache._dec is not dec0
    p = do.Predictor(sub(d, "pred_"), int(d["n_layers"]))
    jw = sub(d, "joint_"); jw["ffn_out.bias"] = jw["ffn_out.bias"].copy(); jw["ffn_out.bias"][0] -= 40.0
    assert second == [do.greedy_search(p, do.Joint(jw), d["enc"][0], T, n_steps=n_steps)]
    assert second != first
    m2 = copy.deepcopy(m)
    assert w.basic_greedy_search(m2, enc, torch.tensor(T), n_steps=n_steps) == second
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)                # a file this test wrote itself
    assert w.basic_greedy_search(m3, enc, torch.tensor(T), n_steps=n_steps) == second
    m._decoder_cache.invalidate()
    assert m._decoder_cache._dec is None


def test_ctc_loss_takes_half_precision_logits_outside_autocast():
    """ctc.py's custom_fwd casts only under autocast; fp16/bf16 activations handed over directly are cast up too and
    the gradient returns in the input dtype."""
    import wenet_celoss_amd as w
    torch.manual_seed(3)
    x = torch.randn(2, 12, 9, device=DEV)
    y = torch.tensor([[1, 2, 3], [4, 4, -1]], device=DEV)
    il, tl = torch.tensor([12, 9], device=DEV), torch.tensor([3, 2], device=DEV)
    ref = w.ctc_loss(x.clone().requires_grad_(True), y, il, tl)
    for dt in (torch.float16, torch.bfloat16):
        xh = x.to(dt).requires_grad_(True)
        loss = w.ctc_loss(xh, y, il, tl)
        loss.backward()
        assert xh.grad.dtype == dt and torch.isfinite(xh.grad).all()
        assert loss.item() == pytest.approx(ref.item(), rel=2e-2)


def test_greedy_without_prejoin_linear_matches_oracle():
    """prejoin_linear=False (joint.py:30-31: encoder, predictor and join widths equal): the step kernels run with
    identity pre-join weights, which is exact in fp32; tokens equal the oracle's with the same identities."""
    import wenet_celoss_amd as w
    d = np.load(names("greedy_core_0.npz")[0])
    pw, jw = sub(d, "pred_"), sub(d, "joint_")
    V, D = pw["embed.weight"].shape
    H, L = pw["rnn.weight_hh_l0"].shape[1], int(d["n_layers"])
    P = pw["projection.weight"].shape[0]
    pred = w.RNNPredictor(V, D, P, 0.1, H, L).to(DEV).eval()
    pred.load_state_dict({k: torch.tensor(v) for k, v in pw.items()})
    joint = w.TransducerJoint(V, P, P, P, prejoin_linear=False).to(DEV).eval()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        joint.ffn_out.weight.copy_((torch.randint(-16, 17, joint.ffn_out.weight.shape, generator=g).float() / 8).to(DEV))
        joint.ffn_out.bias.zero_()
        joint.ffn_out.bias[0] += 3.0
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    T = int(d["T"])
    enc = torch.tensor(d["enc"][:, :, :P], device=DEV).contiguous()
    hyps = w.basic_greedy_search(model, enc, torch.tensor(T), n_steps=4)
    jd = {"enc_ffn.weight": np.eye(P, dtype=np.float32), "enc_ffn.bias": np.zeros(P, np.float32),
          "pred_ffn.weight": np.eye(P, dtype=np.float32), "pred_ffn.bias": np.zeros(P, np.float32),
          "ffn_out.weight": joint.ffn_out.weight.detach().cpu().numpy(), "ffn_out.bias": joint.ffn_out.bias.detach().cpu().numpy()}
    ref, margin = do.greedy_search(do.Predictor(pw, L), do.Joint(jd), enc[0].cpu().numpy(), T, n_steps=4, return_margin=True)
    assert margin > 1e-4 and len(ref) > 3
    assert hyps == [ref]


@pytest.mark.parametrize("where", ["same_block", "other_block", "with_blank"])
def test_greedy_exact_ties_go_to_the_first_index(where):
    """Two vocabulary entries with identical joiner rows tie exactly on every step; log_softmax + argmax give the lower
    index (torch.argmax, numpy argmax).  The device resolves rows from per-block records: the pair shares a 32-column
    block (runner-up equals the block maximum: the block is re-read and scanned), sits in two blocks, or involves blank.
    The duplicated entry is the fixture's most frequent token, so the tie is at the top of many decisions."""
    import wenet_celoss_amd as w
    d = np.load(names("greedy_core_6.npz")[0])
    pred, joint, _ = build_modules(d)
    toks, counts = np.unique(d["hyp"], return_counts=True)
    hot = int(toks[counts.argmax()])                      # wins the argmax most often
    if where == "same_block":
        a, b = hot, hot + 1 if hot % 32 != 31 else hot - 1
    elif where == "other_block":
        a, b = hot, (hot + 32) % 64 or 1
    else:
        a, b = 0, hot                                     # the token gets blank's row: blank (index 0) wins their ties
    lo, hi = min(a, b), max(a, b)
    with torch.no_grad():
        src = a if where == "with_blank" else hot
        dst = b if src == a else a
        joint.ffn_out.weight[dst] = joint.ffn_out.weight[src]
        joint.ffn_out.bias[dst] = joint.ffn_out.bias[src]
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    T, n_steps = int(d["T"]), int(d["n_steps"])
    enc = torch.tensor(d["enc"], device=DEV)
    hyps = w.basic_greedy_search(model, enc, torch.tensor(T), n_steps=n_steps)
    jw = {k: v.detach().cpu().numpy() for k, v in joint.state_dict().items()}
    ref = do.greedy_search(do.Predictor(sub(d, "pred_"), int(d["n_layers"])), do.Joint(jw), d["enc"][0], T, n_steps=n_steps)
    assert hyps == [ref]
    assert hi not in ref                                  # the higher index of a tied pair never wins
    if where != "with_blank":
        assert lo in ref                                  # ... and the tie was at the top: the lower index was emitted
