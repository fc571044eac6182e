"""Pins the CPU oracle for the fused log-softmax + CTC loss (oracle/ctc_oracle.c)
against torch.nn.CTCLoss run live, exactly as the reference calls it
(/root/reference/wenet/transformer/ctc.py:57-63: log_softmax(2) on (T,B,V),
CTCLoss(reduction='sum'), blank=0, zero_infinity=False)."""
import numpy as np
import pytest
import torch

import oracle


def torch_ctc(logits, targets, ilens, tlens):
    x = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    lp = x.transpose(0, 1).log_softmax(2)
    loss = torch.nn.CTCLoss(reduction="none")(lp, torch.tensor(targets, dtype=torch.long),
                                              torch.tensor(ilens, dtype=torch.long),
                                              torch.tensor(tlens, dtype=torch.long))
    loss.sum().backward()
    return loss.detach().numpy(), x.grad.numpy()


@pytest.mark.parametrize("seed", range(8))
def test_ctc_oracle_vs_torch(seed):
    rng = np.random.default_rng(seed)
    B, T, S, V = 4, int(rng.integers(8, 40)), int(rng.integers(1, 8)), int(rng.integers(4, 32))
    logits = rng.normal(size=(B, T, V)).astype(np.float32) * 2
    targets = rng.integers(1, V, size=(B, S)).astype(np.int32)
    if seed % 2 == 0:           # force repeated labels (the s-2 transition must be blocked)
        targets[:, 1:] = np.where(rng.random((B, S - 1)) < 0.5, targets[:, :-1], targets[:, 1:]) if S > 1 else targets[:, 1:]
    ilens = np.array([T] + list(rng.integers(2 * S + 1, T + 1, size=B - 1)), dtype=np.int32)
    ilens = np.minimum(ilens, T)
    tlens = np.array([S] + list(rng.integers(0, S + 1, size=B - 1)), dtype=np.int32)
    nll, grad = oracle.ctc_loss_f64(logits, targets, ilens, tlens)
    rn, rg = torch_ctc(logits, targets, ilens, tlens)
    finite = np.isfinite(rn)
    np.testing.assert_allclose(nll[finite], rn[finite], rtol=1e-10)
    assert (np.isinf(nll) == np.isinf(rn)).all()
    for b in range(B):
        if finite[b]:
            np.testing.assert_allclose(grad[b], rg[b], atol=2e-7)
            assert not grad[b, ilens[b]:].any()


def test_ctc_infeasible_is_inf():
    rng = np.random.default_rng(3)
    logits = rng.normal(size=(1, 3, 5)).astype(np.float32)
    targets = np.array([[1, 1, 2]], dtype=np.int32)   # needs >= 4 frames
    nll, _ = oracle.ctc_loss_f64(logits, targets, [3], [3], want_grad=False)
    rn, _ = torch_ctc(logits, targets, [3], [3])
    assert np.isinf(nll[0]) and np.isinf(rn[0])
