"""Pins the CPU oracle for the RNN-T loss (oracle/rnnt_oracle.c).

The reference holds no fixture for torchaudio.functional.rnnt_loss
(/root/reference/wenet/transducer/transducer.py:142-147), so the oracle is
pinned by: the public warp-transducer/torchaudio known-answer vector
(SURVEY.md App. A.5), brute-force enumeration of all alignments, and an
independent float64 autograd of the alpha recursion.
"""
import itertools
import math

import numpy as np
import pytest
import torch

import oracle

KAT_LOGITS = np.array(
    [[[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1], [0.1, 0.1, 0.2, 0.8, 0.1]],
      [[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.1, 0.1], [0.7, 0.1, 0.2, 0.1, 0.1]]]], dtype=np.float32)
KAT_GRAD = np.array(
    [[[[-0.13116686, -0.39992680, 0.17703122, 0.17703122, 0.17703122],
       [-0.18572753, 0.12247054, -0.18168408, 0.12247054, 0.12247054],
       [-0.32091246, 0.06269139, 0.06928471, 0.12624497, 0.06269139]],
      [[0.05456068, -0.21824272, 0.05456068, 0.05456068, 0.05456068],
       [0.12073957, 0.12073957, -0.48295828, 0.12073957, 0.12073957],
       [-0.69258820, 0.16871117, 0.18645468, 0.16871117, 0.16871117]]]], dtype=np.float32)
KAT_COST = 4.495666773770733


def brute_force_cost(logits, y, T, U, blank):
    """-log sum over every monotone path of prod of softmax probs (float64)."""
    lp = logits.astype(np.float64)
    lp = lp - np.log(np.exp(lp - lp.max(-1, keepdims=True)).sum(-1, keepdims=True)) - lp.max(-1, keepdims=True)
    total = -math.inf
    # a path = an interleaving of (T-1) blanks-before-last... enumerate positions of the U emits
    # among T+U-1 moves followed by the final blank at (T-1,U).
    for emits in itertools.combinations(range(T + U - 1), U):
        t = u = 0
        s = 0.0
        es = set(emits)
        for k in range(T + U - 1):
            if k in es:
                s += lp[t, u, y[u]]
                u += 1
            else:
                s += lp[t, u, blank]
                t += 1
        assert t == T - 1 and u == U
        s += lp[T - 1, U, blank]
        total = np.logaddexp(total, s)
    return -total


def torch_autograd(logits, targets, llens, tlens, blank):
    """Independent float64 alpha recursion + autograd (costs, grad of sum of costs)."""
    x = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(x, dim=-1)
    costs = []
    for b in range(x.shape[0]):
        T, U = int(llens[b]), int(tlens[b])
        alpha = [[None] * (U + 1) for _ in range(T)]
        for t in range(T):
            for u in range(U + 1):
                if t == 0 and u == 0:
                    alpha[t][u] = torch.zeros((), dtype=torch.float64)
                    continue
                terms = []
                if t > 0:
                    terms.append(alpha[t - 1][u] + lp[b, t - 1, u, blank])
                if u > 0:
                    terms.append(alpha[t][u - 1] + lp[b, t, u - 1, int(targets[b, u - 1])])
                alpha[t][u] = torch.logsumexp(torch.stack(terms), 0)
        costs.append(-(alpha[T - 1][U] + lp[b, T - 1, U, blank]))
    costs = torch.stack(costs)
    costs.sum().backward()
    return costs.detach().numpy(), x.grad.numpy()


def test_kat_public_vector():
    costs, grad = oracle.rnnt_loss_f64(KAT_LOGITS, np.array([[1, 2]]), [2], [2], blank=0)
    # KAT_COST was derived from float64 logits; the oracle takes them rounded to float32
    assert abs(costs[0] - KAT_COST) < 5e-8
    np.testing.assert_allclose(grad, KAT_GRAD, atol=2e-7)
    # the KAT cost itself, re-derived by enumeration
    assert abs(brute_force_cost(KAT_LOGITS[0], [1, 2], 2, 2, 0) - KAT_COST) < 5e-8


@pytest.mark.parametrize("seed", range(20))
def test_brute_force_small_lattices(seed):
    rng = np.random.default_rng(seed)
    T = int(rng.integers(1, 5)); U = int(rng.integers(0, 4)); V = int(rng.integers(2, 7))
    blank = int(rng.integers(0, V)) if seed % 3 == 0 else 0
    logits = rng.normal(size=(1, T, U + 1, V)).astype(np.float32) * 2
    y = rng.integers(0, V, size=(1, max(U, 1))).astype(np.int32)
    if U > 0:
        y[y == blank] = (blank + 1) % V
    costs, _ = oracle.rnnt_loss_f64(logits, y[:, :U] if U else np.zeros((1, 0), np.int32), [T], [U], blank=blank)
    ref = brute_force_cost(logits[0], list(y[0]), T, U, blank)
    assert abs(costs[0] - ref) <= 1e-9 * max(1.0, abs(ref))


@pytest.mark.parametrize("seed", range(6))
def test_autograd_ragged(seed):
    rng = np.random.default_rng(100 + seed)
    B, T, U, V = 3, int(rng.integers(4, 12)), int(rng.integers(2, 6)), int(rng.integers(4, 16))
    logits = rng.normal(size=(B, T, U + 1, V)).astype(np.float32) * 1.5
    targets = rng.integers(1, V, size=(B, U)).astype(np.int32)
    llens = np.array([T] + list(rng.integers(1, T + 1, size=B - 1)), dtype=np.int32)
    tlens = np.array([int(rng.integers(0, U + 1)), U] + list(rng.integers(0, U + 1, size=B - 2)), dtype=np.int32)
    costs, grad = oracle.rnnt_loss_f64(logits, targets, llens, tlens, blank=0)
    rc, rg = torch_autograd(logits, targets, llens, tlens, 0)
    np.testing.assert_allclose(costs, rc, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(grad, rg, rtol=0, atol=2e-7)
    # gradient is exactly zero in the padding region
    for b in range(B):
        assert not grad[b, llens[b]:].any()
        assert not grad[b, :, tlens[b] + 1:].any()


def test_clamp():
    rng = np.random.default_rng(7)
    logits = rng.normal(size=(2, 5, 4, 6)).astype(np.float32) * 3
    targets = rng.integers(1, 6, size=(2, 3)).astype(np.int32)
    _, g0 = oracle.rnnt_loss_f64(logits, targets, [5, 4], [3, 2], clamp=-1)
    _, g1 = oracle.rnnt_loss_f64(logits, targets, [5, 4], [3, 2], clamp=0.05)
    np.testing.assert_allclose(g1, np.clip(g0, -0.05, 0.05), atol=1e-7)


def test_f32_baseline_port_matches_f64():
    rng = np.random.default_rng(11)
    B, T, U, V = 4, 23, 9, 64
    logits = rng.normal(size=(B, T, U + 1, V)).astype(np.float32)
    targets = rng.integers(1, V, size=(B, U)).astype(np.int32)
    llens = np.array([T, T - 3, 7, 1], dtype=np.int32)
    tlens = np.array([U, 4, U, 0], dtype=np.int32)
    c64, g64 = oracle.rnnt_loss_f64(logits, targets, llens, tlens)
    c32, g32 = oracle.rnnt_loss_f32(logits, targets, llens, tlens, nthreads=4)
    np.testing.assert_allclose(c32, c64, rtol=1e-5)
    np.testing.assert_allclose(g32, g64, rtol=1e-4, atol=2e-5)
    # label == blank corner of the case chain
    targets[0, 2] = 0
    c64, g64 = oracle.rnnt_loss_f64(logits, targets, llens, tlens)
    c32, g32 = oracle.rnnt_loss_f32(logits, targets, llens, tlens, nthreads=2)
    np.testing.assert_allclose(c32, c64, rtol=1e-5)
    np.testing.assert_allclose(g32, g64, rtol=1e-4, atol=2e-5)
