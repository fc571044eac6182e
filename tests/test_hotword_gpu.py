"""Hot-word greedy search with the gate inside the device step against fixtures produced by
the reference's basic_greedy_search_both (wenet/transducer/search/greedy_search.py:297-430) with its REAL ContextBias
module (tests/golden/make_golden.py::gen_greedy_both_real): tokens, edit distance and gate trace identical, context
filter on and off, go-back re-decoding included; hipGraph replay == plain launches; several streams == one at a time;
the device path == the host-driven loop."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PATHS = sorted(glob.glob(os.path.join(GOLDEN, "greedy_both_real_*.npz")))


def build(d):
    import wenet_celoss_amd as w
    from context_bias_mirror import from_fixture
    from test_decode_gpu import build_modules
    pred, joint, _ = build_modules(d)
    cb = from_fixture(d, DEV)
    m = w.Transducer(64, 0, torch.nn.Identity(), pred, joint, context_bias=cb, ctc_weight=0.0, transducer_weight=1.0,
                     loss_mode="both")
    return m


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("graph", [True, False])
def test_device_hotword_greedy_matches_reference(path, graph):
    import wenet_celoss_amd as w
    from wenet_celoss_amd.hotword import greedy_search_both_device
    d = np.load(path)
    m = build(d)
    enc = torch.tensor(d["enc"], device=DEV)
    ctx, ctx_len, labels = torch.tensor(d["ctx"]), torch.tensor(d["ctx_len"]), torch.tensor(d["labels"])
    T, n_steps, filt = int(d["T"]), int(d["n_steps"]), str(d["filt"])
    hyps, traces = greedy_search_both_device(m, enc, torch.tensor(T), ctx, ctx_len, n_steps=n_steps, filter_on=filt == "on")
    m._hw_decoder_cache._dec.set_graph(graph)
    hyps, traces = greedy_search_both_device(m, enc, torch.tensor(T), ctx, ctx_len, n_steps=n_steps, filter_on=filt == "on")
    assert hyps[0] == d["hyp"].tolist()
    assert traces[0] == d["trace"].tolist()
    out = w.basic_greedy_search_both(m, enc, torch.tensor(T), ctx, ctx_len, n_steps=n_steps, context_filter_state=filt,
                                     context_decoder_labels_padded=labels)
    assert out[0] == [d["hyp"].tolist()] and out[1] == float(d["dist"])

    class Enc(torch.nn.Module):
        def forward(self, speech, lens, a=-1, b=-1):
            return enc, torch.ones(1, 1, T, dtype=torch.bool, device=DEV)
    m.encoder = Enc()
    hy, dist = m.greedy_search(torch.zeros(1, T, 8, device=DEV), torch.tensor([T]), n_steps=n_steps, context_list=ctx,
                               context_lengths=ctx_len, context_filter_state=filt, context_decoder_labels_padded=labels)
    assert hy == [d["hyp"].tolist()] and dist == float(d["dist"])


@pytest.mark.parametrize("path", PATHS[:3])
def test_host_driven_loop_agrees(path, monkeypatch):
    """The host-driven loop of round 1 (still used for hot-word modules of another structure) gives the same answer
    with the same module."""
    import wenet_celoss_amd as w
    d = np.load(path)
    m = build(d)
    enc = torch.tensor(d["enc"], device=DEV)
    ctx, ctx_len, labels = torch.tensor(d["ctx"]), torch.tensor(d["ctx_len"]), torch.tensor(d["labels"])
    monkeypatch.setenv("WR_HOTWORD_HOST", "1")
    out = w.basic_greedy_search_both(m, enc, torch.tensor(int(d["T"])), ctx, ctx_len, n_steps=int(d["n_steps"]),
                                     context_filter_state=str(d["filt"]), context_decoder_labels_padded=labels)
    assert out[0] == [d["hyp"].tolist()] and out[1] == float(d["dist"])


def test_streams_decoded_together_equal_single_streams():
    """Extension: N utterances sharing one hot-word list advance together; every stream equals its single run
    (one of them is the fixture utterance itself)."""
    from wenet_celoss_amd.hotword import greedy_search_both_device
    d = np.load(PATHS[4])
    m = build(d)
    enc = torch.tensor(d["enc"], device=DEV)
    T = int(d["T"])
    g = torch.Generator().manual_seed(3)
    others = (torch.randint(-16, 17, (3, T, enc.shape[2]), generator=g).float() / 8).to(DEV)
    encs = torch.cat([others[:1], enc, others[1:]], 0)
    lens = torch.tensor([T - 7, T, T, T - 20])
    ctx, ctx_len = torch.tensor(d["ctx"]), torch.tensor(d["ctx_len"])
    hyps, traces = greedy_search_both_device(m, encs, lens, ctx, ctx_len, n_steps=int(d["n_steps"]), filter_on=True)
    assert hyps[1] == d["hyp"].tolist() and traces[1] == d["trace"].tolist()
    for i in (0, 2, 3):
        h1, t1 = greedy_search_both_device(m, encs[i:i + 1, :int(lens[i])].contiguous(), lens[i:i + 1], ctx, ctx_len,
                                           n_steps=int(d["n_steps"]), filter_on=True)
        assert hyps[i] == h1[0] and traces[i] == t1[0], i


def test_shipped_shape_against_the_numpy_oracle():
    """The shipped dimensions (conf/encoder_bias_conformer_rnnt_*.yaml: D = 256, 4 heads, unified_hw_odim 100, V = 5000,
    J = 512, LSTM 2 x 256) with random weights: tokens and gate trace equal the numpy restatement (pinned to the
    reference by the fixtures above) on every utterance whose decisions were clear."""
    from context_bias_mirror import ContextBiasMirror
    from oracle import decode_oracle as do
    import wenet_celoss_amd as w
    from wenet_celoss_amd.hotword import greedy_search_both_device
    torch.manual_seed(12)
    V, D, J, H, L, HW, T, N = 5000, 256, 512, 256, 2, 100, 24, 6
    pred = w.RNNPredictor(V, D, D, 0.1, H, L).eval()
    joint = w.TransducerJoint(V, D, D, J).eval()
    cb = ContextBiasMirror(V, D, layers=1, heads=4, hw_dim=HW, hw_heads=4).eval()
    with torch.no_grad():
        joint.ffn_out.weight *= 10
        joint.ffn_out.bias[0] += 9.0
        cb.hw_output_layer_enc.weight.mul_(6.0)
        cb.hw_output_layer.weight.mul_(4.0)
    enc = torch.randn(N, T, D)
    ctx = torch.randint(1, V, (5, 4)); ctx_len = torch.tensor([1, 4, 3, 2, 4], dtype=torch.int32); ctx[0, 0] = 0
    for r in range(5):
        ctx[r, ctx_len[r]:] = -1
    with torch.no_grad():
        hidden = cb.forward_bias_hidden(ctx, ctx_len)
        hidden_empty = cb.forward_bias_hidden(torch.zeros((1, 1), dtype=torch.int), ctx_len[0].unsqueeze(0))
        enc_hot, feat = cb.forward_encoder_bias(hidden, enc)
        enc_cold, _ = cb.forward_encoder_bias(hidden_empty, enc.clone())
        gl = cb.forward_hw_pred_both(feat.reshape(N * T, 1, D), torch.zeros(N * T, 1, D))[:, 0, :]
        dlt = (gl[:, 0] - gl[:, 1]).sort().values
        cb.hw_output_layer.bias[1] += float((dlt[N * T // 2 - 1] + dlt[N * T // 2]) / 2)      # about half the frames gate 1
        gl = cb.forward_hw_pred_both(feat.reshape(N * T, 1, D), torch.zeros(N * T, 1, D))[:, 0, :].reshape(N, T, -1)
    p = do.Predictor({k: v.detach().numpy() for k, v in pred.state_dict().items()}, L)
    j = do.Joint({k: v.detach().numpy() for k, v in joint.state_dict().items()})
    cbn = do.ContextBiasNP({k: v.detach().numpy() for k, v in cb.state_dict().items()}, 4, 4)
    m = w.Transducer(V, 0, torch.nn.Identity(), pred.to(DEV), joint.to(DEV), context_bias=cb.to(DEV), ctc_weight=0.0,
                     transducer_weight=1.0, loss_mode="both")
    hyps, traces = greedy_search_both_device(m, enc.to(DEV), torch.full((N,), T), ctx, ctx_len, n_steps=4, filter_on=True)
    compared = 0
    for i in range(N):
        gate_margin = float((gl[i, :, 0] - gl[i, :, 1]).abs().min())
        mj = MarginTracker(j)
        ref = do.greedy_search_both(p, mj, cbn, hidden[0].numpy(), hidden_empty[0].numpy(), enc_hot[i].numpy(),
                                    feat[i].numpy(), enc_cold[i].numpy(), T, [0], n_steps=4, filter_on=True)
        if gate_margin > 1e-3 and mj.min_margin > 1e-3:
            assert hyps[i] == ref[0], i
            assert traces[i] == ref[2], i
            compared += 1
    assert compared >= 3


@pytest.mark.parametrize("n_steps,n_ctx", [(1, 4), (2, 4), (3, 4), (64, 4), (2, 12)])
def test_random_models_against_the_numpy_oracle(n_steps, n_ctx):
    """Many small random models (V = 48, D = 16) with a gate that flips often: go-backs at the emission cap, right at
    frame 0, several in a row.  Tokens, gate trace and edit distance equal the numpy restatement (pinned to the
    reference by the fixtures) whenever every joiner decision and every gate was clear.  Lists of 4 entries take the
    folded attention of the bias kernel (heads * entries <= D), the list of 12 its general form."""
    from context_bias_mirror import ContextBiasMirror
    from oracle import decode_oracle as do
    import wenet_celoss_amd as w
    from wenet_celoss_amd.hotword import greedy_search_both_device
    V, D, J, H, L, HW, T = 48, 16, 32, 16, 2, 8, 36
    compared = go_backs = 0
    for seed in range(12):
        torch.manual_seed(1000 * n_steps + seed + 17 * (n_ctx - 4))
        pred = w.RNNPredictor(V, D, D, 0.1, H, L).eval()
        joint = w.TransducerJoint(V, D, D, J).eval()
        cb = ContextBiasMirror(V, D, layers=1, heads=2, hw_dim=HW, hw_heads=2).eval()
        with torch.no_grad():
            for prm in list(pred.parameters()) + list(joint.parameters()):
                prm.mul_(3.0)
            joint.ffn_out.bias[0] += 2.5
            cb.hw_output_layer_enc.weight.mul_(6.0)
            cb.hw_output_layer.weight.mul_(4.0)
        enc = torch.randn(1, T, D)
        ctx = torch.randint(1, V, (n_ctx, 3)); ctx[0, 0] = 0
        ctx_len = torch.tensor(([1, 3, 2, 3] * 3)[:n_ctx], dtype=torch.int32)
        for r in range(n_ctx):
            ctx[r, ctx_len[r]:] = -1
        with torch.no_grad():
            hidden = cb.forward_bias_hidden(ctx, ctx_len)
            hidden_empty = cb.forward_bias_hidden(torch.zeros((1, 1), dtype=torch.int), ctx_len[0].unsqueeze(0))
            enc_hot, feat = cb.forward_encoder_bias(hidden, enc)
            enc_cold, _ = cb.forward_encoder_bias(hidden_empty, enc.clone())
            gl = cb.forward_hw_pred_both(feat.transpose(0, 1), torch.zeros(T, 1, D))[:, 0, :]
            dlt = (gl[:, 0] - gl[:, 1]).sort().values
            cb.hw_output_layer.bias[1] += float((dlt[T // 2 - 1] + dlt[T // 2]) / 2)
            gl = cb.forward_hw_pred_both(feat.transpose(0, 1), torch.zeros(T, 1, D))[:, 0, :]
        labels = [0, 1, 1, 0, 1]
        mj = MarginTracker(do.Joint({k: v.detach().numpy() for k, v in joint.state_dict().items()}))
        ref = do.greedy_search_both(do.Predictor({k: v.detach().numpy() for k, v in pred.state_dict().items()}, L), mj,
                                    do.ContextBiasNP({k: v.detach().numpy() for k, v in cb.state_dict().items()}, 2, 2),
                                    hidden[0].numpy(), hidden_empty[0].numpy(), enc_hot[0].numpy(), feat[0].numpy(),
                                    enc_cold[0].numpy(), T, labels, n_steps=n_steps, filter_on=True, return_go_backs=True)
        if float((gl[:, 0] - gl[:, 1]).abs().min()) < 1e-3 or mj.min_margin < 1e-3:
            continue
        m = w.Transducer(V, 0, torch.nn.Identity(), pred.to(DEV), joint.to(DEV), context_bias=cb.to(DEV), ctc_weight=0.0,
                         transducer_weight=1.0, loss_mode="both")
        out = w.basic_greedy_search_both(m, enc.to(DEV), torch.tensor(T), ctx, ctx_len, n_steps=n_steps,
                                         context_filter_state="on", context_decoder_labels_padded=torch.tensor([labels]))
        _, traces = greedy_search_both_device(m, enc.to(DEV), torch.tensor(T), ctx, ctx_len, n_steps=n_steps, filter_on=True)
        assert out[0] == [ref[0]], (seed, n_steps)
        assert traces[0] == ref[2], (seed, n_steps)
        assert out[1] == ref[1]
        compared += 1
        go_backs += ref[4]
    assert compared >= (2 if n_steps == 64 else 6), compared      # long emission runs leave fewer clear cases
    assert go_backs > 0 or n_steps == 64           # with 64 tokens per frame these models rarely cross a gate flip


class MarginTracker:
    """Joiner wrapper for the numpy oracle: smallest top-1 / top-2 logit gap over all decisions."""

    def __init__(self, joint):
        self.joint, self.min_margin = joint, float("inf")

    def __call__(self, enc, pred):
        out = self.joint(enc, pred)
        top = np.partition(out[0], -2)[-2:]
        self.min_margin = min(self.min_margin, float(top[1] - top[0]))
        return out
