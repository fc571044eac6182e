"""Parity of the HIP CTC loss (C-ABI wr_ctc_loss_fwd/bwd) with the CPU oracle,
with torch.nn.CTCLoss run on the CPU, and with the fixtures produced by the
reference's own CTC module (tests/golden/ctc_ref_*.npz).
Tolerance: nll rtol 1e-5; gradient atol 1e-5 + rtol 1e-4 (north-star bar: 1e-4 rel)."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run_hip(logits, targets, ilens, tlens, grad_out=None):
    import wenet_celoss_amd as w
    x = torch.tensor(logits, device=DEV, requires_grad=True)
    nll = w.ctc_loss(x, torch.tensor(targets, device=DEV), torch.tensor(ilens, device=DEV),
                     torch.tensor(tlens, device=DEV), reduction="none")
    if grad_out is None:
        nll.sum().backward()
    else:
        nll.backward(torch.tensor(grad_out, device=DEV))
    return nll.detach().cpu().numpy(), x.grad.cpu().numpy()


def make(rng, B, T, S, V, repeat=False, full=False):
    logits = (rng.normal(size=(B, T, V)) * 2).astype(np.float32)
    targets = rng.integers(1, V, size=(B, max(S, 1))).astype(np.int64)
    if repeat and S > 1:
        m = rng.random((B, S - 1)) < 0.5
        targets[:, 1:S] = np.where(m, targets[:, :S - 1], targets[:, 1:S])
    if full:
        ilens = np.full(B, T, np.int32); tlens = np.full(B, S, np.int32)
    else:
        tlens = rng.integers(0, S + 1, size=B).astype(np.int32); tlens[0] = S
        ilens = np.array([rng.integers(min(2 * s + 1, T), T + 1) for s in tlens], np.int32); ilens[-1] = T
    for b in range(B):
        targets[b, tlens[b]:] = -1
    return logits, targets[:, :max(S, 1)], ilens, tlens


@pytest.mark.parametrize("B,T,S,V,repeat", [
    (1, 1, 0, 3, False), (2, 6, 0, 5, False), (3, 10, 1, 4, False), (4, 30, 7, 20, True),
    (3, 50, 20, 33, True),      # 41 states, KS=1
    (2, 90, 40, 64, True),      # 81 states, KS=2
    (2, 200, 80, 50, False),    # KS=3
    (2, 330, 150, 40, True),    # 301 states, KS=5: the BASELINE label length
    (1, 520, 255, 30, False),   # KS=8: round 1's maximum
    (2, 1100, 400, 25, True),   # 801 states, 13 waves
    (1, 1030, 511, 12, False),  # 1023 states: the supported maximum (16 waves)
    (5, 77, 12, 1000, True),
    (1, 9, 3, 16384, False),    # the widest vocabulary the gradient kernel's LDS row takes
    (2, 12, 4, 4099, True),     # rows alternate between the four 16-byte alignments
])
def test_parity_vs_oracle(B, T, S, V, repeat):
    rng = np.random.default_rng(B * 7 + T + S * 3 + V)
    logits, targets, ilens, tlens = make(rng, B, T, S, V, repeat)
    nll, grad = run_hip(logits, targets, ilens, tlens)
    on, og = oracle.ctc_loss_f64(logits, targets, ilens, tlens)
    np.testing.assert_allclose(nll, on, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(grad, og, rtol=1e-4, atol=1e-5)
    for b in range(B):
        assert not grad[b, ilens[b]:].any()


def test_logits_at_an_odd_storage_offset():
    """Logits whose storage starts 4 bytes past a 16-byte boundary (a contiguous view into a larger buffer): the gradient
    rows no longer share the logits rows' alignment, which takes the kernels' scalar row path."""
    import wenet_celoss_amd as w
    rng = np.random.default_rng(5)
    B, T, S, V = 2, 14, 5, 36
    logits, targets, ilens, tlens = make(rng, B, T, S, V, True)
    buf = torch.zeros(B * T * V + 1, device=DEV)
    buf[1:] = torch.tensor(logits, device=DEV).reshape(-1)
    x = buf[1:].view(B, T, V).requires_grad_(True)
    assert x.data_ptr() % 16 == 4 and x.is_contiguous()
    nll = w.ctc_loss(x, torch.tensor(targets, device=DEV), torch.tensor(ilens, device=DEV), torch.tensor(tlens, device=DEV),
                     reduction="none")
    g, = torch.autograd.grad(nll.sum(), x)
    on, og = oracle.ctc_loss_f64(logits, targets, ilens, tlens)
    np.testing.assert_allclose(nll.detach().cpu().numpy(), on, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g.cpu().numpy(), og, rtol=1e-4, atol=1e-5)


def test_infeasible_is_inf_and_grad_scaling():
    rng = np.random.default_rng(1)
    logits, targets, ilens, tlens = make(rng, 3, 9, 4, 8, full=True)
    targets[0] = 3                          # "3 3 3 3" needs 7 frames ...
    ilens[0] = 5                            # ... but only 5 are given
    nll, _ = run_hip(logits, targets, ilens, tlens)
    assert np.isinf(nll[0]) and np.isfinite(nll[1:]).all()
    logits, targets, ilens, tlens = make(rng, 3, 20, 4, 8)
    go = np.array([2.0, -0.5, 0.0], np.float32)
    _, g = run_hip(logits, targets, ilens, tlens, grad_out=go)
    _, og = oracle.ctc_loss_f64(logits, targets, ilens, tlens)
    np.testing.assert_allclose(g, og * go[:, None, None], rtol=1e-4, atol=1e-5)


def test_vs_torch_ctcloss_cpu_medium():
    """Same call the reference makes (ctc.py:60-61) on the CPU, at T=300,B=8,S=60,V=500."""
    rng = np.random.default_rng(2)
    logits, targets, ilens, tlens = make(rng, 8, 300, 60, 500, repeat=True)
    nll, grad = run_hip(logits, targets, ilens, tlens)
    x = torch.tensor(logits, dtype=torch.float64, requires_grad=True)   # ATen's own fp32 lattice is too noisy to check against
    lp = x.transpose(0, 1).log_softmax(2)
    ref = torch.nn.CTCLoss(reduction="none")(lp, torch.tensor(np.where(targets < 0, 0, targets)),
                                             torch.tensor(ilens.astype(np.int64)), torch.tensor(tlens.astype(np.int64)))
    ref.sum().backward()
    np.testing.assert_allclose(nll, ref.detach().numpy(), rtol=1e-5)
    np.testing.assert_allclose(grad, x.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "ctc_ref_*.npz"))))
def test_module_matches_reference_fixture(path):
    """wenet_celoss_amd.CTC.forward == the reference CTC.forward outputs (loss, d/d hs_pad, d/d ctc_lo)."""
    import wenet_celoss_amd as w
    d = np.load(path)
    V, D = d["w_ctc_lo.weight"].shape
    ctc = w.CTC(V, D).to(DEV)
    ctc.load_state_dict({"ctc_lo.weight": torch.tensor(d["w_ctc_lo.weight"]), "ctc_lo.bias": torch.tensor(d["w_ctc_lo.bias"])})
    hs = torch.tensor(d["hs"], device=DEV, requires_grad=True)
    loss = ctc(hs, torch.tensor(d["hlens"], device=DEV), torch.tensor(d["ys"], device=DEV),
               torch.tensor(d["ys_lens"], device=DEV))
    if np.isfinite(d["loss"]):
        assert loss.item() == pytest.approx(float(d["loss"]), rel=1e-5)
        loss.backward()
        np.testing.assert_allclose(hs.grad.cpu().numpy(), d["grad_hs"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ctc.ctc_lo.weight.grad.cpu().numpy(), d["grad_w"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(ctc.ctc_lo.bias.grad.cpu().numpy(), d["grad_b"], rtol=1e-4, atol=1e-4)
    else:
        assert np.isinf(loss.item())
    np.testing.assert_allclose(ctc.log_softmax(hs.detach()).detach().cpu().numpy(), d["log_softmax"], rtol=1e-5, atol=1e-5)
    assert (ctc.argmax(hs.detach()).cpu().numpy() == d["argmax"]).all()


def test_full_size_properties():
    """BASELINE CTC shape (T=1000,B=32,S=150,V=5000): nll AND the whole gradient of every utterance against the
    float64 oracle at the north-star tolerance (nll rtol 1e-5; gradient rtol 1e-4 / atol 1e-5 -- T = 1000
    dependent steps, lattice values ~1e4); nll also against torch.nn.CTCLoss on the CPU (the reference's own call,
    ctc.py:60-61); every gradient row sums to ~0, padding frames are zero."""
    import wenet_celoss_amd as w
    torch.manual_seed(0)
    B, T, S, V = 32, 1000, 150, 5000
    x = torch.randn(B, T, V, device=DEV, requires_grad=True)
    y = torch.randint(1, V, (B, S), device=DEV)
    il = torch.randint(600, T + 1, (B,), dtype=torch.int32, device=DEV); il[0] = T
    tl = torch.randint(50, S + 1, (B,), dtype=torch.int32, device=DEV); tl[0] = S
    nll = w.ctc_loss(x, y, il, tl, reduction="none")
    nll.sum().backward()
    g = x.grad
    assert g.sum(-1).abs().max().item() < 1e-4
    for b in range(0, B, 5):
        assert not g[b, il[b]:].any()
    lp = x.detach().cpu().transpose(0, 1).log_softmax(2)
    ref = torch.nn.CTCLoss(reduction="none")(lp, y.cpu(), il.cpu().long(), tl.cpu().long())
    np.testing.assert_allclose(nll.detach().cpu().numpy(), ref.numpy(), rtol=2e-5)
    onll, og = oracle.ctc_loss_f64(x.detach().cpu().numpy(), y.cpu().numpy().astype(np.int32),
                                   il.cpu().numpy(), tl.cpu().numpy())
    np.testing.assert_allclose(nll.detach().cpu().numpy(), onll, rtol=1e-5)
    np.testing.assert_allclose(g.cpu().numpy(), og, rtol=1e-4, atol=1e-5)
