"""CPU tests of the host-side logic: helpers pinned against the reference's own
outputs (tests/golden/common_ref.npz), shard arithmetic, and the N>1 path with
a world-size-2 gloo group."""
import glob
import os
import sys
from typing import List, Tuple

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def test_add_blank_and_log_add_match_reference():
    from wenet_celoss_amd.common import add_blank, log_add, end_blank
    d = np.load(os.path.join(GOLDEN, "common_ref.npz"))
    out = add_blank(torch.tensor(d["ys"]), 0, -1)
    assert (out.numpy() == d["add_blank"]).all()
    for pair, ref in zip(d["log_add_in"], d["log_add_out"]):
        assert log_add(list(pair)) == ref
    assert log_add([-float("inf"), -float("inf")]) == float(d["log_add_all_inf"])
    eb = end_blank(torch.tensor([[1, 2, -1]]), 0, -1)
    assert eb.tolist() == [[1, 2, 0, 0]]


def test_shard_arithmetic():
    from wenet_celoss_amd.dist import balanced_shards, shard_bounds
    for n in (1, 7, 32, 33):
        for ws in (1, 2, 3, 8):
            b = [shard_bounds(n, ws, r) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    costs = [1000 * 151, 500 * 60, 900 * 100, 700 * 151, 650 * 80, 999 * 150, 510 * 51, 800 * 120]
    sh = balanced_shards(costs, 4)
    assert sorted(i for s in sh for i in s) == list(range(8))
    loads = [sum(costs[i] for i in s) for s in sh]
    assert max(loads) <= 1.35 * (sum(costs) / 4)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wenet_celoss_amd.dist import global_mean_of_rank_means, global_utterance_mean, shard_bounds
    costs = torch.arange(1.0, 8.0)                       # 7 utterances, uneven shards (4 + 3)
    lo, hi = shard_bounds(7, world, rank)
    local = costs[lo:hi]
    a = global_mean_of_rank_means(local.mean())
    b = global_utterance_mean(local)
    q.put((rank, float(a), float(b), lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # rank shards: [1,2,3,4] and [5,6,7]
    assert (res[0][3], res[0][4], res[1][3], res[1][4]) == (0, 4, 4, 7)
    want_rank_means = (2.5 + 6.0) / 2                     # what DDP-style averaging yields (reference behaviour)
    want_true_mean = 4.0
    for r in res:
        assert r[1] == pytest.approx(want_rank_means)
        assert r[2] == pytest.approx(want_true_mean)


def _ddp_ctc_worker(rank, world, port, q):
    """One rank of a CPU data-parallel step over the host logic that has a CPU form: balanced utterance shards
    (dist.balanced_shards) -> wenet_celoss_amd.CTC (config 1's CPU path) under DistributedDataParallel with
    join() / no_sync(), as wenet/utils/executor.py:48-53,81-86 drives it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wenet_celoss_amd as w
    from wenet_celoss_amd.dist import balanced_shards, global_mean_of_rank_means
    torch.manual_seed(3)
    model = torch.nn.Sequential()
    model.enc = torch.nn.Linear(6, 8)
    model.ctc = w.CTC(11, 8)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.enc, self.ctc = model.enc, model.ctc

        def forward(self, x, xl, y, yl):
            return self.ctc(torch.tanh(self.enc(x)), xl, y, yl)
    net = Net()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(5, 14, 6, generator=g)
    xl = torch.tensor([14, 9, 12, 7, 14])
    y = torch.randint(1, 11, (5, 3), generator=g)
    yl = torch.tensor([3, 2, 3, 1, 2])
    mine = balanced_shards([int(a) * (2 * int(b) + 1) for a, b in zip(xl, yl)], world)[rank]
    ddp = torch.nn.parallel.DistributedDataParallel(net, find_unused_parameters=True)
    with ddp.join():
        with ddp.no_sync():
            ddp(x[mine], xl[mine], y[mine], yl[mine]).backward()
        loss = ddp(x[mine] * 0.5, xl[mine], y[mine], yl[mine])
        loss.backward()
    grads = {n: p.grad.clone() for n, p in net.named_parameters()}
    q.put((rank, mine, {k: v.numpy() for k, v in grads.items()}, float(global_mean_of_rank_means(loss.detach()))))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_ddp_step_over_balanced_shards():
    """N > 1 on CPU: two gloo ranks, uneven shards dealt by balanced_shards, DDP join()/no_sync() around the CTC
    module; every rank ends with the mean over ranks of the accumulated per-rank gradients."""
    import wenet_celoss_amd as w
    from wenet_celoss_amd.dist import balanced_shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_ctc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0][1] + res[1][1]) == [0, 1, 2, 3, 4] and len(res[0][1]) != len(res[1][1])
    # single-process reference
    torch.manual_seed(3)
    enc = torch.nn.Linear(6, 8)
    ctc = w.CTC(11, 8)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(5, 14, 6, generator=g)
    xl = torch.tensor([14, 9, 12, 7, 14])
    y = torch.randint(1, 11, (5, 3), generator=g)
    yl = torch.tensor([3, 2, 3, 1, 2])
    shards = balanced_shards([int(a) * (2 * int(b) + 1) for a, b in zip(xl, yl)], 2)
    want, losses = None, []
    for mine in shards:
        for p in list(enc.parameters()) + list(ctc.parameters()):
            p.grad = None
        ctc(torch.tanh(enc(x[mine])), xl[mine], y[mine], yl[mine]).backward()
        l2 = ctc(torch.tanh(enc(x[mine] * 0.5)), xl[mine], y[mine], yl[mine])
        l2.backward()
        losses.append(float(l2))
        gr = {"enc.weight": enc.weight.grad, "enc.bias": enc.bias.grad, "ctc.ctc_lo.weight": ctc.ctc_lo.weight.grad,
              "ctc.ctc_lo.bias": ctc.ctc_lo.bias.grad}
        want = {k: v.clone() / 2 for k, v in gr.items()} if want is None else {k: want[k] + v / 2 for k, v in gr.items()}
    for r in res:
        for k, v in want.items():
            np.testing.assert_allclose(r[2][k], v.numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
        assert r[3] == pytest.approx(sum(losses) / 2, rel=1e-6)


def test_context_bias_mirror_matches_reference_intermediates():
    """tests/context_bias_mirror.py (the stand-in that carries the reference ContextBias weights on the GPU box) reproduces
    the tensors the reference module computed before its greedy loop, and its per-step methods the recorded gate."""
    import glob
    from context_bias_mirror import from_fixture
    paths = sorted(glob.glob(os.path.join(GOLDEN, "greedy_both_real_*.npz")))
    assert len(paths) >= 6
    for path in paths:
        d = np.load(path)
        cb = from_fixture(d)
        ctx, ctx_len, enc = torch.tensor(d["ctx"]), torch.tensor(d["ctx_len"]), torch.tensor(d["enc"])
        with torch.no_grad():
            hidden = cb.forward_bias_hidden(ctx, ctx_len)
            hidden_empty = cb.forward_bias_hidden(torch.zeros((1, 1), dtype=torch.int), ctx_len[0].unsqueeze(0))
            enc_hot, feat = cb.forward_encoder_bias(hidden, enc)
            enc_cold, _ = cb.forward_encoder_bias(hidden_empty, enc.clone())
            gl = cb.forward_hw_pred_both(feat.transpose(0, 1), torch.zeros(enc.shape[1], 1, enc.shape[2]))[:, 0, :]
        for got, key in ((hidden, "hidden"), (hidden_empty, "hidden_empty"), (enc_hot, "enc_hot"), (feat, "enc_hot_feat"),
                         (enc_cold, "enc_cold"), (gl, "gate_logits")):
            np.testing.assert_allclose(got.numpy(), d[key], rtol=1e-5, atol=1e-6, err_msg=f"{os.path.basename(path)}:{key}")
        from wenet_celoss_amd.hotword import device_capable
        assert device_capable(cb)
    from bias_stub import TinyBias
    assert not device_capable(TinyBias(64, 16, 16))        # other hot-word modules keep the host-driven loop


def test_label_width_limits_are_reported_before_launch():
    """The loss kernels' label-width limits are checked in Transducer.forward on the padded batch, naming the longest
    utterance -- on the CPU, before any kernel (or the missing GPU) is touched."""
    from wenet_celoss_amd.transducer import check_limits
    text = torch.zeros(3, 600, dtype=torch.long)
    lens = torch.tensor([10, 600, 40])
    check_limits(text[:, :511], lens.clamp(max=511), with_ctc=True)
    check_limits(text, lens, with_ctc=False)
    with pytest.raises(RuntimeError, match=r"utterance 1 with 600 labels.*511"):
        check_limits(text, lens, with_ctc=True)
    with pytest.raises(RuntimeError, match=r"1023 labels"):
        check_limits(torch.zeros(2, 1100, dtype=torch.long), torch.tensor([1100, 3]), with_ctc=False)


def test_joiner_and_predictor_variants_accepted_or_refused_loudly():
    import wenet_celoss_amd as w
    j = w.TransducerJoint(10, 8, 8, 8, prejoin_linear=False)
    assert j.enc_ffn is None and j.pred_ffn is None and j.post_ffn is None
    j2 = w.TransducerJoint(10, 8, 8, 8, postjoin_linear=True)
    assert set(dict(j2.named_parameters())) >= {"post_ffn.weight", "post_ffn.bias", "enc_ffn.weight", "ffn_out.weight"}
    with pytest.raises(AssertionError):
        w.TransducerJoint(10, 8, 6, 8, prejoin_linear=False)          # joint.py:30-31: widths must agree
    for act in ("tanh", "relu", "hardtanh", "selu", "swish", "gelu"):            # get_activation's table, common.py:233-240
        assert w.TransducerJoint(10, 8, 8, 8, activation=act).act_code == w._lib.ACTIVATIONS[act]
    with pytest.raises(KeyError):
        w.TransducerJoint(10, 8, 8, 8, activation="sigmoid")
    with pytest.raises(NotImplementedError, match="lstm"):
        w.RNNPredictor(10, 8, 8, 0.1, 8, 1, rnn_type="gru")
    assert w.RNNPredictor(10, 8, 8, 0.1, 8, 1, bias=False).rnn.bias is False
    # the stateless predictors keep the reference's parameter names and state layout (predictor.py:203-481)
    e = w.EmbeddingPredictor(10, 8, 0.1, 4, history_size=2)
    c = w.ConvPredictor(10, 8, 0.1, history_size=3, bias=True)
    assert set(e.state_dict()) == {"pos_embed.weight", "embed.weight", "ffn.weight", "ffn.bias", "norm.weight", "norm.bias"}
    assert set(c.state_dict()) == {"embed.weight", "conv.weight", "conv.bias", "norm.weight", "norm.bias"}
    assert e.init_state(3, torch.device("cpu"))[0].shape == (3, 2, 8) and c.init_state(2, torch.device("cpu"))[0].shape == (2, 3, 8)
    x = torch.randint(0, 10, (2, 6))
    assert e.eval()(x).shape == (2, 6, 8) and c.eval()(x).shape == (2, 6, 8)
    assert len(c.batch_to_cache(c.init_state(2, torch.device("cpu")))) == 2
    with pytest.raises(RuntimeError, match="HIP device|no CPU path"):          # forward_step is the device path
        e.forward_step(x[:, :1], torch.zeros(2, 1), e.init_state(2, torch.device("cpu")))
    with pytest.raises(NotImplementedError, match="base predictor"):
        w.PredictorBase().init_state(1, torch.device("cpu"))
    e, pp = j2.pre_activation(torch.randn(2, 3, 8), torch.randn(2, 4, 8))
    assert e.shape == (2, 3, 8) and pp.shape == (2, 4, 8)


def test_bench_self_launch_names_the_missing_devices():
    """`python bench.py --gpus N` without a launcher environment starts its own ranks; with fewer than N devices
    visible it must stop before touching a GPU, naming what is missing (here: no GPU at all)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible")
    assert r.returncode == 2 and "needs 2 HIP devices" in r.stderr and "missing: " in r.stderr and r.stdout == ""
    # a launcher environment that disagrees with --gpus is refused as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_plan_buckets_properties():
    """fused.plan_buckets: a partition of the batch in ascending label length that saves at least the stated share of
    lattice cells, or None (uniform batches, tiny batches, savings below the bar)."""
    import random
    from wenet_celoss_amd.fused import plan_buckets
    assert plan_buckets([1000] * 32, [150] * 32) is None            # BASELINE config: nothing to gain
    assert plan_buckets([10], [5]) is None and plan_buckets([0, 0], [0, 0]) is None
    assert plan_buckets([30, 30, 30], [2, 9, 20]) is None           # a rescoring n-best: too small to split
    assert plan_buckets([30, 30, 30], [2, 9, 20], min_cells=0) is not None
    rnd = random.Random(4)
    for _ in range(20):
        n = rnd.randint(2, 40)
        t = [rnd.randint(50, 400) for _ in range(n)]
        u = [rnd.randint(0, 120) for _ in range(n)]
        g = plan_buckets(t, u, max_buckets=4, min_gain=0.08, min_cells=0)
        whole = n * max(t) * (max(u) + 1)
        if g is None:
            continue
        assert 2 <= len(g) <= 4 and sorted(i for x in g for i in x) == list(range(n))
        cells = sum(len(x) * max(t[i] for i in x) * (max(u[i] for i in x) + 1) for x in g)
        assert cells <= 0.92 * whole
        tops = [max(u[i] for i in x) for x in g]
        assert tops == sorted(tops)                                  # groups in ascending label length
        assert all(max(u[i] for i in g[k]) <= min(u[i] for i in g[k + 1]) for k in range(len(g) - 1))


def test_transducer_constructor_contract():
    """Same keyword surface and weight-sum assertion as the reference (transducer.py:23-46)."""
    import wenet_celoss_amd as w
    enc = torch.nn.Identity()
    p = w.RNNPredictor(10, 4, 4, 0.1, 4, 2)
    j = w.TransducerJoint(10, 4, 4, 8)
    m = w.Transducer(10, 0, enc, p, j, ctc=w.CTC(10, 4), ctc_weight=0.25, transducer_weight=0.75)
    assert m.attention_decoder_weight == 0.0 and m.sos == 9 and m.eos == 9
    with pytest.raises(AssertionError):
        w.Transducer(10, 0, enc, p, j, ctc_weight=0.3, transducer_weight=0.3)
    keys = set(m.state_dict().keys())
    for k in ("joint.enc_ffn.weight", "joint.pred_ffn.bias", "joint.ffn_out.weight", "ctc.ctc_lo.weight",
              "predictor.embed.weight", "predictor.rnn.weight_ih_l0", "predictor.projection.bias"):
        assert k in keys                                   # reference checkpoints' parameter names


class _ScriptableEncoder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.proj = torch.nn.Linear(8, 12)

    def forward(self, xs: torch.Tensor, xs_lens: torch.Tensor, decoding_chunk_size: int = 0,
                num_decoding_left_chunks: int = -1):
        mask = (torch.arange(xs.size(1))[None, :] < xs_lens[:, None]).unsqueeze(1)
        return torch.tanh(self.proj(xs)), mask

    @torch.jit.export
    def forward_chunk(self, xs: torch.Tensor, offset: int, required_cache_size: int, att_cache: torch.Tensor,
                      cnn_cache: torch.Tensor):
        return torch.tanh(self.proj(xs)), att_cache, cnn_cache


def test_model_survives_train_py_script_export(tmp_path):
    """wenet/bin/train.py:203-205 does `torch.jit.script(model).save(init.zip)` on rank 0 before training; the
    HIP-backed forwards are marked @torch.jit.unused (as the reference's own k2 variant does,
    transducer_k2_loss.py:80) so that the unchanged train.py gets past this smoke export."""
    import wenet_celoss_amd as w
    m = w.Transducer(23, 0, _ScriptableEncoder(), w.RNNPredictor(23, 10, 10, 0.0, 14, 2, dropout=0.0),
                     w.TransducerJoint(23, 12, 10, 16), ctc=w.CTC(23, 12), ctc_weight=0.3, transducer_weight=0.7,
                     hw_weight=0.0)
    sm = torch.jit.script(m)
    sm.save(str(tmp_path / "init.zip"))
    assert (tmp_path / "init.zip").stat().st_size > 0
    # the artefact exposes the reference's four step exports (transducer.py:600-629) and they compute the reference's
    # module graphs (a scripted file cannot call the HIP library); checked against the modules' own weights
    ld = torch.jit.load(str(tmp_path / "init.zip")).eval()
    m.eval()
    cache = ld.forward_predictor_init_state()
    assert [tuple(c.shape) for c in cache] == [(2, 1, 14), (2, 1, 14)]
    tok = torch.tensor([[5]])
    out, new_cache = ld.forward_predictor_step(tok, cache)
    with torch.no_grad():
        ref_out, (rm, rc) = m.predictor.rnn(m.predictor.embed(tok), (cache[0], cache[1]))
        ref_out = m.predictor.projection(ref_out)
    torch.testing.assert_close(out, ref_out)
    torch.testing.assert_close(new_cache[0], rm)
    enc = torch.randn(1, 1, 12)
    js = ld.forward_joint_step(enc, out)
    with torch.no_grad():
        ref = m.joint.ffn_out(torch.tanh(m.joint.enc_ffn(enc).unsqueeze(2) + m.joint.pred_ffn(out).unsqueeze(1)))
    torch.testing.assert_close(js, ref)
    y, _, _ = ld.forward_encoder_chunk(torch.randn(1, 4, 8), 0, -1)
    assert y.shape == (1, 4, 12)
    # eager mode never takes the export bodies: CPU tensors are refused by the HIP path
    with pytest.raises(RuntimeError, match="HIP device"):
        m.forward_joint_step(enc, out)


def test_joint_precision_selection(monkeypatch):
    """precision= / WR_JOINT_PRECISION select the joiner mode; unknown names are rejected; every mode refuses CPU
    tensors (there is no CPU path in the product)."""
    import pytest
    import torch
    import wenet_celoss_amd as w
    from wenet_celoss_amd import joint as jm
    monkeypatch.delenv("WR_JOINT_PRECISION", raising=False)
    assert jm._resolve_precision(None) == "fp32"
    monkeypatch.setenv("WR_JOINT_PRECISION", "bf16x3")
    assert jm._resolve_precision(None) == "bf16x3"
    assert jm._resolve_precision("bf16") == "bf16"          # the argument wins over the environment
    with pytest.raises(ValueError, match="precision"):
        jm._resolve_precision("tf32")
    monkeypatch.setenv("WR_JOINT_PRECISION", "fast")
    with pytest.raises(ValueError, match="precision"):
        jm._resolve_precision(None)
    monkeypatch.delenv("WR_JOINT_PRECISION")
    for prec in (None, "bf16x3", "bf16"):
        m = w.TransducerJoint(12, 4, 4, 8, precision=prec)
        assert m.precision == prec
        with pytest.raises(RuntimeError, match="HIP device"):
            m(torch.zeros(1, 2, 4), torch.zeros(1, 3, 4))


class _StepExport(torch.nn.Module):
    """What Transducer.forward_predictor_step does for the runtime: an exported method that calls forward_step."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.p(x)

    @torch.jit.export
    def step(self, x: torch.Tensor, pad: torch.Tensor, cache: List[torch.Tensor]) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        return self.p.forward_step(x, pad, cache)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "predictor_var_*.npz"))))
def test_stateless_predictors_script_and_reproduce_the_reference_steps(path):
    """torch.jit.script (train.py:203-205, export_jit.py) over EmbeddingPredictor / ConvPredictor: the exported step body
    and the training forward reproduce the reference modules' outputs stored in the fixtures (CPU, library ops)."""
    import wenet_celoss_amd as w
    d = np.load(path)
    pw = {k[2:]: torch.tensor(d[k]) for k in d.files if k.startswith("w_")}
    V, D = pw["embed.weight"].shape
    if str(d["kind"]) == "embedding":
        m = w.EmbeddingPredictor(V, D, 0.1, int(d["n_head"]), int(d["history"]), str(d["act"]), "pos_embed.bias" in pw)
    else:
        m = w.ConvPredictor(V, D, 0.1, int(d["history"]), str(d["act"]), "conv.bias" in pw)
    m.load_state_dict(pw)
    m.eval()
    sm = torch.jit.script(_StepExport(m))
    steps, N = d["toks"].shape
    cache = m.init_state(N, torch.device("cpu"))
    for s_ in range(steps):
        out, cache = sm.step(torch.tensor(d["toks"][s_]).reshape(N, 1), torch.zeros(N, 1), cache)
        np.testing.assert_allclose(out.detach().numpy(), d["outs"][s_], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(cache[0].detach().numpy(), d["hist"][s_])
    np.testing.assert_allclose(sm(torch.tensor(d["toks"].T.copy())).detach().numpy(), d["full"], rtol=1e-5, atol=1e-6)
