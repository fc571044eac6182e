"""Joiner + RNN-T loss as one node (wenet_celoss_amd.joint_rnnt_loss: wr_joint_fwd_lse -> wr_rnnt_loss_fwd_from_lse ->
wr_rnnt_loss_bwd in place -> joiner backward) against the two separate ops and against the float64 oracle.

Tolerances: the fused epilogue merges the row log-sum-exp statistics in a different order than rnnt_lse_kernel, so
costs may differ in the last fp32 bits: 1e-6 relative (VERDICT r1 item 6 asks <= 1e-6); gradients 1e-5 of the
tensor's largest entry."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["epilogue", "row-pass"])
def lse_mode(request, monkeypatch):
    """Every test of this file runs with the loss's row statistics coming from the joiner forward's epilogue and from the
    loss's own row pass (the node picks per precision; WR_FUSED_LSE_EPILOGUE forces one)."""
    monkeypatch.setenv("WR_FUSED_LSE_EPILOGUE", "1" if request.param == "epilogue" else "0")
    return request.param

CASES = [  # B, T, U, J, V, ragged
    (2, 9, 4, 16, 50, False),
    (3, 70, 11, 32, 257, True),      # several 64-cell tiles, odd V (scalar tails, column padding)
    (2, 33, 6, 64, 5000, True),      # the shipped vocabulary: 20 column chunks
    (1, 5, 0, 8, 7, False),          # U = 0: blank-only lattice
    (4, 130, 37, 512, 1024, True),   # the shipped join_dim
]


def make(B, T, U, J, V, ragged, seed=0):
    g = torch.Generator().manual_seed(seed)
    ep = torch.randn(B, T, J, generator=g).to(DEV)
    pp = torch.randn(B, U + 1, J, generator=g).to(DEV)
    w = (torch.randn(V, J, generator=g) * (2.0 / J ** 0.5)).to(DEV)
    b = torch.randn(V, generator=g).to(DEV)
    y = torch.randint(1, V, (B, max(U, 1)), generator=g, dtype=torch.int32)[:, :U].contiguous().to(DEV)
    if ragged:
        tl = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g)
        ul = torch.randint(0, U + 1, (B,), generator=g)
        tl[0], ul[-1] = T, U
    else:
        tl, ul = torch.full((B,), T), torch.full((B,), U)
    return ep, pp, w, b, y, tl.to(torch.int32).to(DEV), ul.to(torch.int32).to(DEV)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_fused_equals_unfused_and_oracle(case, precision):
    import wenet_celoss_amd as w_
    B, T, U, J, V, ragged = case
    ep, pp, w, b, y, tl, ul = make(*case)
    leaves = [t.clone().requires_grad_(True) for t in (ep, pp, w, b)]
    gc = torch.linspace(0.5, 1.5, B, device=DEV)
    costs = w_.joint_rnnt_loss(*leaves, y, tl, ul, blank=0, reduction="none", precision=precision)
    (costs * gc).sum().backward()
    ref_leaves = [t.clone().requires_grad_(True) for t in (ep, pp, w, b)]
    logits = w_.joint_logits(*ref_leaves, tl, ul, precision=precision)
    ref_costs = w_.rnnt_loss(logits, y, tl, ul, blank=0, reduction="none")
    (ref_costs * gc).sum().backward()
    torch.testing.assert_close(costs, ref_costs, rtol=1e-6, atol=1e-6)
    for name, a, r in zip(("ep", "pp", "w", "b"), leaves, ref_leaves):
        scale = float(r.grad.abs().max())
        assert float((a.grad - r.grad).abs().max()) <= 1e-5 * scale + 1e-9, name
    if precision == "fp32":      # float64 oracle on the exact logits
        lg = (torch.tanh(ep.double()[:, :, None] + pp.double()[:, None]) @ w.double().T + b.double()).float().cpu().numpy()
        oc, _ = oracle.rnnt_loss_f64(lg, y.cpu().numpy().reshape(B, U), tl.cpu().numpy(), ul.cpu().numpy())
        np.testing.assert_allclose(costs.detach().cpu().numpy(), oc, rtol=2e-5)


def test_fused_inplace_and_separate_gradient_buffers_agree(monkeypatch):
    import wenet_celoss_amd as w_
    args = make(3, 70, 11, 32, 257, True)
    res = []
    for thr in ("0", str(1 << 40)):                          # always in place / never
        monkeypatch.setenv("WR_FUSED_INPLACE_BYTES", thr)
        leaves = [t.clone().requires_grad_(True) for t in args[:4]]
        w_.joint_rnnt_loss(*leaves, *args[4:], blank=0, reduction="sum").backward()
        res.append([t.grad.clone() for t in leaves])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_bucketing_by_label_length_keeps_costs_and_gradients():
    """`buckets`: the batch cut into groups by label length, each its own joiner + loss call padded to its own maxima
    (fused.plan_buckets).  Utterances are independent, so per-utterance costs are bit-identical and come back in the
    caller's order; gradients agree to summation order."""
    import wenet_celoss_amd as w_
    from wenet_celoss_amd.fused import plan_buckets
    B, T, U, J, V = 9, 60, 40, 32, 300
    ep, pp, w, b, y, tl, ul = make(B, T, U, J, V, False, seed=21)
    tl = torch.tensor([60, 33, 58, 41, 60, 25, 47, 52, 39], dtype=torch.int32, device=DEV)
    ul = torch.tensor([40, 3, 17, 38, 9, 22, 40, 5, 30], dtype=torch.int32, device=DEV)
    groups = plan_buckets(tl.tolist(), ul.tolist())
    assert 9 * 60 * 41 >= 20000 and groups is not None and len(groups) >= 2 and sorted(i for g in groups for i in g) == list(range(B))
    res = []
    gc_w = torch.linspace(0.5, 1.5, B, device=DEV)
    for nb in (1, 4):
        leaves = [t.clone().requires_grad_(True) for t in (ep, pp, w, b)]
        costs = w_.joint_rnnt_loss(*leaves, y, tl, ul, blank=0, reduction="none", buckets=nb)
        (costs * gc_w).sum().backward()
        res.append((costs.detach(), [t.grad for t in leaves]))
    assert torch.equal(res[0][0], res[1][0])
    for name, a, r in zip(("ep", "pp", "w", "b"), res[1][1], res[0][1]):
        assert float((a - r).abs().max()) <= 1e-5 * float(r.abs().max()) + 1e-9, name


def test_fused_workspace_matches_pass1():
    """The lattice the sweeps build from the epilogue's statistics equals the one built from rnnt_lse_kernel's."""
    import wenet_celoss_amd as w_
    from wenet_celoss_amd import _lib
    from wenet_celoss_amd.rnnt_loss import rnnt_lattice
    lib = _lib.load()
    B, T, U, J, V = 3, 41, 9, 32, 300
    ep, pp, w, b, y, tl, ul = make(B, T, U, J, V, True, seed=5)
    U1 = U + 1
    logits = torch.empty(B, T, U1, V, device=DEV)
    rwsb = lib.wr_rnnt_workspace_bytes(B, T, U1)
    rws = torch.zeros(rwsb, dtype=torch.uint8, device=DEV)
    jwsb = lib.wr_joint_workspace_bytes(J, V)
    jws = torch.empty(jwsb, dtype=torch.uint8, device=DEV)
    costs = torch.empty(B, device=DEV)
    st = _lib.current_stream(torch.device(DEV))
    P = _lib.ptr
    _lib.check(lib.wr_joint_fwd_lse(P(ep), P(pp), P(w), P(b), P(tl), P(ul), P(y), B, T, U1, J, V, 0, 0, P(logits), P(jws), jwsb,
                                    P(rws), rwsb, st), "fwd_lse")
    _lib.check(lib.wr_rnnt_loss_fwd_from_lse(P(logits), P(y), P(tl), P(ul), B, T, U1, V, 0, P(costs), P(rws), rwsb, st),
               "from_lse")
    alpha = torch.empty(B, T, U1, device=DEV); beta = torch.empty_like(alpha)
    _lib.check(lib.wr_rnnt_export_lattice(P(rws), rwsb, P(tl), P(ul), B, T, U1, P(alpha), P(beta), st), "export")
    ref_logits = w_.joint_logits(ep, pp, w, b, tl, ul)
    c2, a2, b2 = rnnt_lattice(ref_logits, y, tl, ul, blank=0)
    # the logits the fused kernel wrote are the unfused kernel's, bit for bit, wherever the loss reads them
    tt = torch.arange(T, device=DEV)[None, :, None] < tl[:, None, None]
    uu = torch.arange(U1, device=DEV)[None, None, :] <= ul[:, None, None]
    m = (tt & uu)
    assert torch.equal(logits[m], ref_logits[m])
    torch.testing.assert_close(costs, c2, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(alpha, a2, rtol=1e-6, atol=1e-5)
    torch.testing.assert_close(beta, b2, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_fused_repair_pass_on_extreme_spread(precision):
    """A row spreading over more than 88 nats overflows the epilogue's partial sums (they are taken against the
    first logit a lane sees, not a running maximum): the workgroup raises the workspace flag and the stand-alone
    pass 1 recomputes the statistics.  Costs and gradients still equal the separate ops."""
    import wenet_celoss_amd as w_
    B, T, U, J, V = 2, 20, 5, 32, 700
    ep, pp, w, b, y, tl, ul = make(B, T, U, J, V, True, seed=9)
    b = b.clone()
    b[300] += 400.0                                          # column 300 lives in the second 256-column chunk
    b[650] -= 400.0
    leaves = [t.clone().requires_grad_(True) for t in (ep, pp, w, b)]
    costs = w_.joint_rnnt_loss(*leaves, y, tl, ul, blank=0, reduction="none", precision=precision)
    costs.sum().backward()
    ref_leaves = [t.clone().requires_grad_(True) for t in (ep, pp, w, b)]
    ref_costs = w_.rnnt_loss(w_.joint_logits(*ref_leaves, tl, ul, precision=precision), y, tl, ul, blank=0, reduction="none")
    ref_costs.sum().backward()
    assert torch.isfinite(costs).all()
    torch.testing.assert_close(costs, ref_costs, rtol=1e-6, atol=1e-5)
    for a, r in zip(leaves, ref_leaves):
        assert float((a.grad - r.grad).abs().max()) <= 1e-5 * float(r.grad.abs().max()) + 1e-9


def test_transducer_forward_fused_equals_unfused():
    from test_transducer_gpu import build
    m = build()
    g = torch.Generator().manual_seed(2)
    speech = torch.randn(3, 11, 8, generator=g).to(DEV)
    slen = torch.tensor([11, 7, 9], dtype=torch.int32, device=DEV)
    text = torch.tensor([[3, 5, 2, 9], [4, 4, -1, -1], [7, 1, 6, -1]], device=DEV)
    tlen = torch.tensor([4, 2, 3], dtype=torch.int32, device=DEV)
    res = {}
    for fused in (True, False):
        m.zero_grad()
        m.fused_loss = fused
        out = m(speech, slen, text, tlen)
        out["loss"].backward()
        res[fused] = (out["loss"].item(), out["loss_rnnt"].item(), {n: p.grad.clone() for n, p in m.named_parameters()})
    assert res[True][0] == pytest.approx(res[False][0], rel=1e-6)
    assert res[True][1] == pytest.approx(res[False][1], rel=1e-6)
    for n, gr in res[False][2].items():
        assert float((res[True][2][n] - gr).abs().max()) <= 1e-5 * float(gr.abs().max()) + 1e-9, n


def test_fused_argument_checks():
    import wenet_celoss_amd as w_
    ep, pp, w, b, y, tl, ul = make(2, 9, 4, 16, 50, False)
    with pytest.raises(RuntimeError, match="input length mismatch"):
        w_.joint_rnnt_loss(ep, pp, w, b, y, tl - 1, ul)
    with pytest.raises(RuntimeError, match="output length mismatch"):
        w_.joint_rnnt_loss(ep, pp, w, b, y, tl, ul - 1)
    with pytest.raises(ValueError):
        w_.joint_rnnt_loss(ep, pp, w, b, y, tl, ul, precision="bf16")
    with pytest.raises(RuntimeError, match="HIP device"):
        w_.joint_rnnt_loss(ep.cpu(), pp.cpu(), w.cpu(), b.cpu(), y.cpu(), tl.cpu(), ul.cpu())


def test_full_baseline_shape_loss_block():
    """The loss block of Transducer.forward at the BASELINE shape (B=32, T=1000, U=150, V=5000, join_dim 512, ragged
    lengths): the fused node against the two separate ops -- costs to 1e-6, every gradient to 1e-5 of its largest entry --
    and 64 sampled logit rows of the separate ops' output against a float64 evaluation of joint.py:60-69 (their
    log-sum-exp against the fused node's through the costs).  One logits-sized tensor (96.6 GB) lives at a time."""
    import gc
    import wenet_celoss_amd as w_
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * 2 ** 30:
        pytest.skip("needs 150 GB of free HBM")
    B, T, U, J, V = 32, 1000, 150, 512, 5000
    g = torch.Generator().manual_seed(17)
    ep = (torch.randn(B, T, J, generator=g) * 0.7).to(DEV)
    pp = (torch.randn(B, U + 1, J, generator=g) * 0.7).to(DEV)
    w = (torch.randn(V, J, generator=g) * 0.06).to(DEV)
    b = (torch.randn(V, generator=g) * 0.5).to(DEV)
    y = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32).to(DEV)
    tl = torch.randint(T // 2, T + 1, (B,), generator=g); tl[0] = T
    ul = torch.randint(U // 3, U + 1, (B,), generator=g); ul[-1] = U
    tl, ul = tl.to(torch.int32).to(DEV), ul.to(torch.int32).to(DEV)
    gc_w = torch.linspace(0.5, 1.5, B, device=DEV)

    def run(fused):
        leaves = [t.clone().requires_grad_(True) for t in (ep, pp, w, b)]
        if fused:
            costs = w_.joint_rnnt_loss(*leaves, y, tl, ul, blank=0, reduction="none")
        else:
            logits = w_.joint_logits(*leaves, tl, ul)
            costs = w_.rnnt_loss(logits, y, tl, ul, blank=0, reduction="none", inplace_grad=True)
            del logits
        (costs * gc_w).sum().backward()
        out = (costs.detach().clone(), [t.grad.clone() for t in leaves])
        del costs, leaves
        gc.collect(); torch.cuda.empty_cache()
        return out
    c1, g1 = run(True)
    c0, g0 = run(False)
    assert torch.isfinite(c1).all()
    torch.testing.assert_close(c1, c0, rtol=1e-6, atol=1e-3)
    for name, a, r in zip(("ep", "pp", "w", "b"), g1, g0):
        assert float((a - r).abs().max()) <= 1e-5 * float(r.abs().max()) + 1e-12, name
    # sampled rows of the logits against float64
    logits = w_.joint_logits(ep, pp, w, b, tl, ul)
    rs = np.random.default_rng(3)
    wd, bd = w.double().cpu(), b.double().cpu()
    for _ in range(64):
        bi = int(rs.integers(B)); t = int(rs.integers(int(tl[bi]))); u = int(rs.integers(int(ul[bi]) + 1))
        ref = torch.tanh(ep[bi, t].double().cpu() + pp[bi, u].double().cpu()) @ wd.T + bd
        np.testing.assert_allclose(logits[bi, t, u].cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_second_backward_through_a_retained_graph(precision):
    """The node writes the gradient over its saved logits through a raw pointer (invisible to autograd's version
    counter).  A second backward through a retained graph -- retain_graph=True, or per-loss torch.autograd.grad --
    must not differentiate gradients-as-logits: the node rebuilds the logits first.  Both passes give the gradients
    of a fresh node, bit for bit (the kernels are deterministic)."""
    import wenet_celoss_amd as w_
    args = make(3, 70, 11, 32, 257, True)
    leaves = [t.clone().requires_grad_(True) for t in args[:4]]
    costs = w_.joint_rnnt_loss(*leaves, *args[4:], blank=0, reduction="none", precision=precision, buckets=1)
    g1 = torch.autograd.grad(costs.sum(), leaves, retain_graph=True)
    g2 = torch.autograd.grad(costs.sum(), leaves, retain_graph=True)
    g3 = torch.autograd.grad((costs * torch.tensor([2.0, 0.0, -1.0], device=DEV)).sum(), leaves)
    fresh = [t.clone().requires_grad_(True) for t in args[:4]]
    c2 = w_.joint_rnnt_loss(*fresh, *args[4:], blank=0, reduction="none", precision=precision, buckets=1)
    r3 = torch.autograd.grad((c2 * torch.tensor([2.0, 0.0, -1.0], device=DEV)).sum(), fresh)
    for a, b_, c, d in zip(g1, g2, g3, r3):
        assert torch.equal(a, b_)
        assert torch.equal(c, d)
