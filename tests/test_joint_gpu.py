"""Parity of the fused MFMA joiner (wr_joint_fwd / wr_joint_bwd_dz) with the
fixtures produced by the reference's TransducerJoint (tests/golden/joint_ref_*.npz)
and with a float64 torch evaluation of the same formula at larger shapes.
Tolerance 1e-4 relative (north-star bar); exact-fp32 MFMA keeps us well inside."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref64(enc, pred, sd, gout=None):
    """float64 torch evaluation of joint.py:55-69 (+ autograd)."""
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd.items()}
    e = torch.tensor(enc, dtype=torch.float64, requires_grad=True)
    p = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    ep = e @ t["enc_ffn.weight"].T + t["enc_ffn.bias"]
    pp = p @ t["pred_ffn.weight"].T + t["pred_ffn.bias"]
    out = torch.tanh(ep[:, :, None] + pp[:, None]) @ t["ffn_out.weight"].T + t["ffn_out.bias"]
    grads = None
    if gout is not None:
        out.backward(torch.tensor(gout, dtype=torch.float64))
        grads = dict(enc=e.grad.numpy(), pred=p.grad.numpy(), **{k: v.grad.numpy() for k, v in t.items()})
    return out.detach().numpy(), grads


def build(sd, V, E, P, J):
    import wenet_celoss_amd as w
    m = w.TransducerJoint(V, E, P, J).to(DEV)
    m.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    return m


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "joint_ref_*.npz"))))
def test_matches_reference_fixture(path):
    d = np.load(path)
    sd = {k[2:]: d[k] for k in d.files if k.startswith("w_")}
    V, J = sd["ffn_out.weight"].shape
    E, P = sd["enc_ffn.weight"].shape[1], sd["pred_ffn.weight"].shape[1]
    m = build(sd, V, E, P, J)
    enc = torch.tensor(d["enc"], device=DEV, requires_grad=True)
    pred = torch.tensor(d["pred"], device=DEV, requires_grad=True)
    out = m(enc, pred)
    np.testing.assert_allclose(out.detach().cpu().numpy(), d["out"], rtol=1e-4, atol=1e-5)
    out.backward(torch.tensor(d["gout"], device=DEV))
    np.testing.assert_allclose(enc.grad.cpu().numpy(), d["grad_enc"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pred.grad.cpu().numpy(), d["grad_pred"], rtol=1e-4, atol=1e-4)
    for name, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), d["g_" + name], rtol=1e-4, atol=1e-4, err_msg=name)


@pytest.mark.parametrize("B,T,U1,E,P,J,V", [
    (2, 9, 5, 32, 24, 128, 300),      # V not a multiple of 256, one MFMA column tile per wave in backward
    (1, 70, 3, 16, 16, 256, 1000),    # M = 210: a partial last 64-cell tile
    (3, 11, 7, 64, 64, 512, 517),     # the shipped join_dim; V odd -> scalar dY staging path
    (2, 13, 4, 8, 8, 36, 64),         # J not a multiple of 8/128 (zero-padded k depth)
])
def test_parity_float64(B, T, U1, E, P, J, V):
    g = torch.Generator().manual_seed(B * 100 + T + J + V)
    import wenet_celoss_amd as w
    m = w.TransducerJoint(V, E, P, J)
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    enc = torch.randn(B, T, E, generator=g); pred = torch.randn(B, U1, P, generator=g)
    gout = torch.randn(B, T, U1, V, generator=g)
    ro, rg = ref64(enc.numpy(), pred.numpy(), sd, gout.numpy())
    e = enc.to(DEV).requires_grad_(True); p = pred.to(DEV).requires_grad_(True)
    out = m(e, p)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ro, rtol=1e-4, atol=2e-5)
    out.backward(gout.to(DEV))
    scale = lambda a: 1e-4 * max(1.0, float(np.abs(a).max()))
    np.testing.assert_allclose(e.grad.cpu().numpy(), rg["enc"], rtol=1e-4, atol=scale(rg["enc"]))
    np.testing.assert_allclose(p.grad.cpu().numpy(), rg["pred"], rtol=1e-4, atol=scale(rg["pred"]))
    for name, prm in m.named_parameters():
        np.testing.assert_allclose(prm.grad.cpu().numpy(), rg[name], rtol=1e-4, atol=scale(rg[name]), err_msg=name)


def test_lengths_skip_padding_and_feed_rnnt_loss():
    """With lengths, cells in the padded region are not computed; the loss and all
    gradients are unchanged because the loss never reads them."""
    import wenet_celoss_amd as w
    torch.manual_seed(4)
    B, T, U, E, P, J, V = 3, 80, 70, 16, 16, 128, 64
    m = w.TransducerJoint(V, E, P, J).to(DEV)
    enc = torch.randn(B, T, E, device=DEV); pred = torch.randn(B, U + 1, P, device=DEV)
    y = torch.randint(1, V, (B, U), dtype=torch.int32, device=DEV)
    ll = torch.tensor([80, 20, 5], dtype=torch.int32, device=DEV)
    tl = torch.tensor([10, 70, 3], dtype=torch.int32, device=DEV)
    res = []
    for lens in (False, True):
        m.zero_grad()
        e = enc.clone().requires_grad_(True); p = pred.clone().requires_grad_(True)
        logits = m(e, p, ll, tl) if lens else m(e, p)
        loss = w.rnnt_loss(logits, y, ll, tl, blank=0, reduction="mean")
        loss.backward()
        res.append((loss.item(), e.grad.clone(), p.grad.clone(), m.ffn_out.weight.grad.clone()))
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-6)
    for a, b in zip(res[0][1:], res[1][1:]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "joint_var_*.npz"))))
def test_constructor_variants_match_reference_fixture(path, precision):
    """Every activation of get_activation, prejoin_linear off, postjoin_linear on: outputs and all gradients of the
    reference's TransducerJoint (joint.py:16-70) built with the same options.  Exact kernels at the 1e-4 bar; the split
    kernels relative to each tensor's scale (their operand rounding is 2^-17)."""
    import wenet_celoss_amd as w
    d = np.load(path)
    sd = {k[2:]: d[k] for k in d.files if k.startswith("w_")}
    V, J = sd["ffn_out.weight"].shape
    E, P = d["enc"].shape[-1], d["pred"].shape[-1]
    m = w.TransducerJoint(V, E, P, J, prejoin_linear=bool(d["prejoin"]), postjoin_linear=bool(d["postjoin"]),
                          activation=str(d["activation"]), precision=precision).to(DEV)
    m.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    enc = torch.tensor(d["enc"], device=DEV, requires_grad=True)
    pred = torch.tensor(d["pred"], device=DEV, requires_grad=True)
    out = m(enc, pred)
    rel = 1e-4 if precision == "fp32" else 2e-4

    def close(got, ref, name):
        ref = np.asarray(ref)
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=rel, atol=rel * max(1.0, float(np.abs(ref).max())),
                                   err_msg=name)
    close(out, d["out"], "out")
    out.backward(torch.tensor(d["gout"], device=DEV))
    close(enc.grad, d["grad_enc"], "enc")
    close(pred.grad, d["grad_pred"], "pred")
    for name, p in m.named_parameters():
        close(p.grad, d["g_" + name], name)


@pytest.mark.parametrize("activation", ["relu", "hardtanh", "selu", "swish", "gelu"])
def test_activation_through_the_fused_loss_node(activation):
    """joint_rnnt_loss with a non-tanh activation: same costs and gradients as joiner module + rnnt_loss."""
    import wenet_celoss_amd as w
    torch.manual_seed(11)
    B, T, U, J, V = 3, 21, 6, 64, 96
    m = w.TransducerJoint(V, 16, 16, J, activation=activation).to(DEV)
    enc = torch.randn(B, T, 16, device=DEV); pred = torch.randn(B, U + 1, 16, device=DEV)
    y = torch.randint(1, V, (B, U), dtype=torch.int32, device=DEV)
    ll = torch.tensor([T, 9, 15], dtype=torch.int32, device=DEV); tl = torch.tensor([U, 2, 4], dtype=torch.int32, device=DEV)
    res = []
    for fused in (False, True):
        m.zero_grad()
        e = enc.clone().requires_grad_(True); p = pred.clone().requires_grad_(True)
        if fused:
            ep, pp = m.pre_activation(e, p)
            loss = w.joint_rnnt_loss(ep, pp, m.ffn_out.weight, m.ffn_out.bias, y, ll, tl, activation=activation)
        else:
            loss = w.rnnt_loss(m(e, p, ll, tl), y, ll, tl, blank=0, reduction="mean")
        loss.backward()
        res.append((loss.item(), e.grad.clone(), p.grad.clone(), m.ffn_out.weight.grad.clone(), m.enc_ffn.weight.grad.clone()))
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-5)
    for a, b in zip(res[0][1:], res[1][1:]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(a.abs().max()))


def test_unsupported_configuration_raises():
    import wenet_celoss_amd as w
    with pytest.raises(KeyError):                    # get_activation's dictionary lookup (common.py:242)
        w.TransducerJoint(10, 8, 8, 8, activation="sigmoid")
    m = w.TransducerJoint(10, 8, 8, 516).to(DEV)
    with pytest.raises(RuntimeError, match="join_dim"):
        m(torch.zeros(1, 2, 8, device=DEV), torch.zeros(1, 2, 8, device=DEV))


# ---- split-precision forward on the bf16 matrix cores (wr_joint_fwd_split) ---------------------------------
# Tolerances, relative to the r.m.s. of the logits: "bf16x3" (three split terms, fp32 accumulation) 1e-4 -- the
# north-star bar of the fp32 path; "bf16" (single term, the AMP mode) 3e-2, the rounding of bf16 operands.
@pytest.mark.parametrize("precision,tol", [("bf16x3", 1e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("B,T,U1,E,P,J,V", [
    (2, 9, 5, 32, 24, 128, 300),      # V not a multiple of 64: a partial last column pair
    (1, 70, 3, 16, 16, 256, 1000),    # M = 210: a partial last 64-cell tile
    (3, 11, 7, 64, 64, 512, 517),     # the shipped join_dim; odd V
    (2, 13, 4, 8, 8, 36, 64),         # J not a multiple of 64 (zero-padded k depth), one column pair
    (1, 40, 9, 16, 16, 512, 5000),    # the shipped vocabulary: 79 column pairs over 4 waves
    (1, 1, 1, 4, 4, 4, 32),           # a single lattice cell, the smallest join_dim
    (1, 3, 2, 8, 8, 508, 36),         # join_dim just under the limit
    (2, 5, 3, 8, 8, 20, 4100),        # 65 column pairs: an odd last round
])
def test_split_forward_parity_float64(precision, tol, B, T, U1, E, P, J, V):
    g = torch.Generator().manual_seed(B * 100 + T + J + V)
    import wenet_celoss_amd as w
    m = w.TransducerJoint(V, E, P, J, precision=precision)
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    enc = torch.randn(B, T, E, generator=g); pred = torch.randn(B, U1, P, generator=g)
    gout = torch.randn(B, T, U1, V, generator=g)
    ro, rg = ref64(enc.numpy(), pred.numpy(), sd, gout.numpy())
    e = enc.to(DEV).requires_grad_(True); p = pred.to(DEV).requires_grad_(True)
    out = m(e, p)
    assert out.dtype == torch.float32
    rms = float(np.sqrt((ro ** 2).mean()))
    err = float(np.abs(out.detach().cpu().numpy() - ro).max())
    assert err <= tol * rms, (err, rms)
    # backward: dZ = dY W and dW = dY^T H run with the same split when V % 4 == 0 (else exact fp32); db is fp32
    out.backward(gout.to(DEV))
    for got, want in ((e.grad, rg["enc"]), (p.grad, rg["pred"]), (m.enc_ffn.weight.grad, rg["enc_ffn.weight"]),
                      (m.ffn_out.weight.grad, rg["ffn_out.weight"])):
        want_rms = float(np.sqrt((want ** 2).mean()))
        assert float(np.abs(got.cpu().numpy() - want).max()) <= tol * want_rms
    scale = lambda a: 1e-4 * max(1.0, float(np.abs(a).max()))
    np.testing.assert_allclose(m.ffn_out.bias.grad.cpu().numpy(), rg["ffn_out.bias"], rtol=1e-4,
                               atol=scale(rg["ffn_out.bias"]))


@pytest.mark.parametrize("terms,tol", [(3, 1e-4), (1, 3e-2)])
@pytest.mark.parametrize("B,T,U1,J,V", [(2, 9, 5, 128, 300), (1, 70, 3, 256, 1000), (2, 13, 4, 36, 64), (1, 20, 9, 512, 5000),
                                        (1, 33, 2, 64, 32), (1, 7, 3, 96, 44)])
def test_split_dz_matches_exact_kernel(terms, tol, B, T, U1, J, V):
    """wr_joint_bwd_dz_split against wr_joint_bwd_dz on the same inputs (row tails V % 32 != 0, partial cell tiles,
    join_dim not a multiple of 32, lengths): error relative to the r.m.s. of dZ; H and the zeros of padded cells
    are identical."""
    from wenet_celoss_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(terms * 1000 + T + J + V)
    ep = torch.randn(B, T, J, generator=g).to(DEV); pp = torch.randn(B, U1, J, generator=g).to(DEV)
    w = (torch.randn(V, J, generator=g) * 0.1).to(DEV)
    gout = torch.randn(B, T, U1, V, generator=g).to(DEV)
    ll = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32); ll[0] = T
    tl = torch.randint(0, U1, (B,), generator=g).to(torch.int32); tl[0] = U1 - 1
    ll, tl = ll.to(DEV), tl.to(DEV)
    st, P = _lib.current_stream(torch.device(DEV)), _lib.ptr
    wsb = lib.wr_joint_dz_split_workspace_bytes(J, V)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    for lens in ((None, None), (ll, tl)):
        dz0 = torch.empty(B, T, U1, J, device=DEV); h0 = torch.empty_like(dz0)
        dz1 = torch.full_like(dz0, float("nan")); h1 = torch.full_like(dz0, float("nan"))
        _lib.check(lib.wr_joint_bwd_dz(P(gout), P(ep), P(pp), P(w), P(lens[0]), P(lens[1]), B, T, U1, J, V, 0, P(dz0), P(h0), st))
        _lib.check(lib.wr_joint_bwd_dz_split(P(gout), P(ep), P(pp), P(w), P(lens[0]), P(lens[1]), B, T, U1, J, V, 0, terms,
                                             P(dz1), P(h1), P(ws), wsb, st))
        assert torch.equal(h0, h1)
        assert torch.equal(dz0 == 0, dz1 == 0) or lens[0] is None
        rms = float(dz0.pow(2).mean().sqrt())
        assert float((dz0 - dz1).abs().max()) <= tol * rms
    assert lib.wr_joint_bwd_dz_split(P(gout), P(ep), P(pp), P(w), None, None, B, T, U1, J, 30, 0, terms, P(dz1), None,
                                     P(ws), wsb, st) != 0           # V not a multiple of 4


def test_split_forward_lengths_and_loss():
    """bf16x3 logits fed to the RNN-T loss: loss within 1e-5 relative of the exact-fp32 joiner's, padded tiles
    skipped when lengths are given."""
    import wenet_celoss_amd as w
    torch.manual_seed(5)
    B, T, U, E, P, J, V = 3, 80, 70, 16, 16, 128, 200
    m = w.TransducerJoint(V, E, P, J).to(DEV)
    enc = torch.randn(B, T, E, device=DEV); pred = torch.randn(B, U + 1, P, device=DEV)
    y = torch.randint(1, V, (B, U), dtype=torch.int32, device=DEV)
    ll = torch.tensor([80, 20, 5], dtype=torch.int32, device=DEV)
    tl = torch.tensor([10, 70, 3], dtype=torch.int32, device=DEV)
    losses = {}
    for prec in ("fp32", "bf16x3"):
        m.precision = prec
        for lens in (False, True):
            logits = m(enc, pred, ll, tl) if lens else m(enc, pred)
            losses[prec, lens] = w.rnnt_loss(logits, y, ll, tl, blank=0, reduction="none").cpu()
    torch.testing.assert_close(losses["bf16x3", False], losses["fp32", False], rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(losses["bf16x3", True], losses["bf16x3", False], rtol=1e-6, atol=1e-5)


def test_split_forward_amp_dtype_and_errors():
    import wenet_celoss_amd as w
    torch.manual_seed(6)
    m = w.TransducerJoint(96, 16, 16, 64, precision="bf16").to(DEV)
    enc = torch.randn(2, 7, 16, device=DEV); pred = torch.randn(2, 4, 16, device=DEV)
    ref = w.joint_logits(m.enc_ffn(enc), m.pred_ffn(pred), m.ffn_out.weight, m.ffn_out.bias)
    for dt in (torch.float16, torch.bfloat16):
        with torch.autocast("cuda", dtype=dt):
            out = m(enc, pred)
        assert out.dtype == dt                     # as the reference's Linear under autocast
        assert float((out.detach().float() - ref.detach()).abs().max()) <= 5e-2 * float(ref.detach().std())
    with pytest.raises(ValueError, match="precision"):
        w.joint_logits(enc, pred, m.ffn_out.weight, m.ffn_out.bias, precision="fp8")
    from wenet_celoss_amd import _lib
    lib = _lib.load()
    assert lib.wr_joint_fwd_split(None, None, None, None, None, None, 1, 1, 1, 64, 10, 0, 2, None, 0, None, 0, None) != 0
    assert b"terms" in lib.wr_last_error()


@pytest.mark.parametrize("terms,tol", [(3, 1e-4), (1, 3e-2)])
@pytest.mark.parametrize("B,T,U1,J,V", [(2, 9, 5, 128, 300), (1, 70, 3, 256, 1000), (2, 13, 4, 36, 64), (1, 20, 9, 512, 5000),
                                        (1, 33, 2, 64, 32), (3, 50, 7, 260, 520)])
def test_split_dw_matches_float64(terms, tol, B, T, U1, J, V):
    """wr_joint_bwd_dw_split against a float64 evaluation of dW = dY^T H, db = sum dY (with and without lengths:
    padded cells must not contribute); partial 256-blocks in v and j, cell ranges that do not fill a 16-cell step."""
    from wenet_celoss_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(terms * 777 + T + J + V)
    gout = torch.randn(B, T, U1, V, generator=g).to(DEV)
    h = torch.tanh(torch.randn(B, T, U1, J, generator=g)).to(DEV)
    ll = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32); ll[0] = T
    tl = torch.randint(0, U1, (B,), generator=g).to(torch.int32); tl[0] = U1 - 1
    ll, tl = ll.to(DEV), tl.to(DEV)
    st, P = _lib.current_stream(torch.device(DEV)), _lib.ptr
    wsb = lib.wr_joint_dw_split_workspace_bytes(B, T, U1, J, V)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    for lens in ((None, None), (ll, tl)):
        g64, h64 = gout.double(), h.double()
        if lens[0] is not None:
            tt = torch.arange(T, device=DEV)[None, :, None] < ll[:, None, None]
            uu = torch.arange(U1, device=DEV)[None, None, :] <= tl[:, None, None]
            g64 = g64 * (tt & uu)[..., None]
        dw_ref = g64.reshape(-1, V).T @ h64.reshape(-1, J)
        db_ref = g64.reshape(-1, V).sum(0)
        dw = torch.full((V, J), float("nan"), device=DEV); db = torch.full((V,), float("nan"), device=DEV)
        _lib.check(lib.wr_joint_bwd_dw_split(P(gout), P(h), P(lens[0]), P(lens[1]), B, T, U1, J, V, terms, P(dw), P(db),
                                             P(ws), wsb, st))
        rms = float(dw_ref.pow(2).mean().sqrt())
        assert float((dw.double() - dw_ref).abs().max()) <= tol * rms
        torch.testing.assert_close(db.double(), db_ref, rtol=1e-5, atol=1e-4)
    assert lib.wr_joint_bwd_dw_split(P(gout), P(h), None, None, B, T, U1, J, 30, terms, P(dw), P(db), P(ws), wsb, st) != 0


@pytest.mark.parametrize("terms", [1, 3])
@pytest.mark.parametrize("B,T,U1,J,V", [(2, 9, 5, 128, 304), (1, 70, 3, 256, 1000), (2, 13, 4, 36, 64), (1, 20, 9, 512, 5000),
                                        (1, 33, 2, 64, 32), (3, 50, 7, 260, 520)])
def test_split_backward_takes_a_bf16_gradient_as_it_is(terms, B, T, U1, J, V):
    """wr_joint_bwd_dz_split_bf16 / wr_joint_bwd_dw_split_bf16 (the AMP step's gradient dtype) against the fp32 entry
    points on the same values widened to fp32: bf16 values are their own hi parts, so dZ, H, dW and db are
    bit-identical; row tails V % 32 != 0, partial tiles, with and without lengths.  V % 8 != 0 is refused."""
    from wenet_celoss_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(terms * 31 + T + J + V)
    ep = torch.randn(B, T, J, generator=g).to(DEV); pp = torch.randn(B, U1, J, generator=g).to(DEV)
    w = (torch.randn(V, J, generator=g) * 0.1).to(DEV)
    g16 = torch.randn(B, T, U1, V, generator=g).to(DEV).to(torch.bfloat16)
    g32 = g16.float()
    ll = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32); ll[0] = T
    tl = torch.randint(0, U1, (B,), generator=g).to(torch.int32); tl[0] = U1 - 1
    ll, tl = ll.to(DEV), tl.to(DEV)
    st, P = _lib.current_stream(torch.device(DEV)), _lib.ptr
    wsz = lib.wr_joint_dz_split_workspace_bytes(J, V); wz = torch.empty(wsz, dtype=torch.uint8, device=DEV)
    wsw = lib.wr_joint_dw_split_workspace_bytes(B, T, U1, J, V); ww = torch.empty(wsw, dtype=torch.uint8, device=DEV)
    for lens in ((None, None), (ll, tl)):
        dz0 = torch.empty(B, T, U1, J, device=DEV); h0 = torch.empty_like(dz0)
        dz1 = torch.full_like(dz0, float("nan")); h1 = torch.full_like(dz0, float("nan"))
        _lib.check(lib.wr_joint_bwd_dz_split(P(g32), P(ep), P(pp), P(w), P(lens[0]), P(lens[1]), B, T, U1, J, V, 0, terms,
                                             P(dz0), P(h0), P(wz), wsz, st))
        _lib.check(lib.wr_joint_bwd_dz_split_bf16(P(g16), P(ep), P(pp), P(w), P(lens[0]), P(lens[1]), B, T, U1, J, V, 0, terms,
                                                  P(dz1), P(h1), P(wz), wsz, st))
        assert torch.equal(dz0, dz1) and torch.equal(h0, h1)
        if J % 4 == 0:
            dw0 = torch.empty(V, J, device=DEV); db0 = torch.empty(V, device=DEV)
            dw1 = torch.full_like(dw0, float("nan")); db1 = torch.full_like(db0, float("nan"))
            _lib.check(lib.wr_joint_bwd_dw_split(P(g32), P(h0), P(lens[0]), P(lens[1]), B, T, U1, J, V, terms, P(dw0), P(db0),
                                                 P(ww), wsw, st))
            _lib.check(lib.wr_joint_bwd_dw_split_bf16(P(g16), P(h0), P(lens[0]), P(lens[1]), B, T, U1, J, V, terms, P(dw1),
                                                      P(db1), P(ww), wsw, st))
            assert torch.equal(dw0, dw1) and torch.equal(db0, db1)
    assert lib.wr_joint_bwd_dz_split_bf16(P(g16), P(ep), P(pp), P(w), None, None, B, T, U1, J, 36, 0, terms, P(dz1), None,
                                          P(wz), wsz, st) != 0      # V not a multiple of 8


def test_amp_step_backward_uses_the_bf16_gradient(monkeypatch):
    """Under autocast(bfloat16) with precision="bf16" the loss returns a bf16 gradient; the joiner's backward feeds it
    to the split kernels unwidened.  Gradients equal those of the same backward run on the widened gradient."""
    import wenet_celoss_amd as w
    from wenet_celoss_amd.joint import joint_backward
    torch.manual_seed(11)
    B, T, U, E, P, J, V = 2, 40, 12, 16, 16, 128, 200
    m = w.TransducerJoint(V, E, P, J, precision="bf16").to(DEV)
    enc = torch.randn(B, T, E, device=DEV, requires_grad=True); pred = torch.randn(B, U + 1, P, device=DEV, requires_grad=True)
    y = torch.randint(1, V, (B, U), dtype=torch.int32, device=DEV)
    ll = torch.tensor([40, 17], dtype=torch.int32, device=DEV); tl = torch.tensor([12, 5], dtype=torch.int32, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(enc, pred)
        assert logits.dtype == torch.bfloat16
        loss = w.rnnt_loss(logits, y, ll, tl, blank=0, reduction="mean")
    (gl,) = torch.autograd.grad(loss, logits, retain_graph=True)
    assert gl.dtype == torch.bfloat16
    loss.backward()
    ep = m.enc_ffn(enc.detach()).float(); pp = m.pred_ffn(pred.detach()).float()
    monkeypatch.setenv("WR_AMP_BACKWARD", "kernels")        # the kernels' path: bit-identical on the widened gradient
    want = joint_backward(gl.float(), ep, pp, m.ffn_out.weight.detach(), None, None, 1, True, True)
    got = joint_backward(gl, ep, pp, m.ffn_out.weight.detach(), None, None, 1, True, True)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert m.ffn_out.weight.grad is not None and torch.isfinite(m.ffn_out.weight.grad).all()
    assert enc.grad is not None and torch.isfinite(enc.grad).all() and torch.isfinite(pred.grad).all()


@pytest.mark.parametrize("act", ["tanh", "relu"])
@pytest.mark.parametrize("B,T,U1,J,V", [(2, 9, 5, 128, 304), (1, 70, 3, 256, 1000), (3, 50, 7, 260, 520)])
def test_amp_backward_library_gemms_match_the_single_term_kernels(monkeypatch, act, B, T, U1, J, V):
    """The AMP backward's default (dH and [dW | db] as vendor-library bf16 GEMMs around wr_joint_dz_act) against this
    package's single-term kernels (WR_AMP_BACKWARD=kernels) on the same bf16 gradient: same products, another order
    of the fp32 sums -- 2e-3 of each result's r.m.s.; padded cells contribute nothing, their dZ rows are exactly zero."""
    from wenet_celoss_amd.joint import joint_backward, activation_code
    g = torch.Generator().manual_seed(T + J + V)
    ep = torch.randn(B, T, J, generator=g).to(DEV); pp = torch.randn(B, U1, J, generator=g).to(DEV)
    w = (torch.randn(V, J, generator=g) * 0.1).to(DEV)
    gout = torch.randn(B, T, U1, V, generator=g).to(DEV).to(torch.bfloat16)
    ll = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32); ll[0] = T
    tl = torch.randint(0, U1, (B,), generator=g).to(torch.int32); tl[0] = U1 - 1
    ll, tl = ll.to(DEV), tl.to(DEV)
    code = activation_code(act)
    for lens in ((None, None), (ll, tl)):
        monkeypatch.setenv("WR_AMP_BACKWARD", "kernels")
        want = joint_backward(gout, ep, pp, w, lens[0], lens[1], 1, True, True, act=code)
        monkeypatch.setenv("WR_AMP_BACKWARD", "library")
        got = joint_backward(gout, ep, pp, w, lens[0], lens[1], 1, True, True, act=code)
        for a, b_ in zip(got, want):
            assert a.shape == b_.shape and a.dtype == b_.dtype == torch.float32
            assert float((a - b_).abs().max()) <= 2e-3 * float(b_.pow(2).mean().sqrt()) + 1e-6
        if lens[0] is not None:
            tt = torch.arange(T, device=DEV)[None, :, None] < ll[:, None, None]
            uu = torch.arange(U1, device=DEV)[None, None, :] <= tl[:, None, None]
            pad = ~(tt & uu)
            # d_ep sums dZ over u: a frame past the utterance's length has only padded cells
            assert float(got[0][pad.all(dim=2)].abs().max() if pad.all(dim=2).any() else 0.0) == 0.0


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_amp_backward_library_gemms_match_float64(dt):
    """The library form of the AMP backward for both autocast dtypes (float16 is torch.cuda.amp.autocast's default, the
    reference's --use_amp) against a float64 evaluation with the operands rounded as the step rounds them (W and H to the
    gradient's dtype): 1e-3 of each result's r.m.s.; lengths given, padded cells excluded."""
    from wenet_celoss_amd.joint import joint_backward
    B, T, U1, J, V = 2, 21, 5, 96, 520
    g = torch.Generator().manual_seed(5)
    ep = torch.randn(B, T, J, generator=g).to(DEV); pp = torch.randn(B, U1, J, generator=g).to(DEV)
    w = (torch.randn(V, J, generator=g) * 0.1).to(DEV)
    gout = (torch.randn(B, T, U1, V, generator=g) * 0.3).to(DEV).to(dt)
    ll = torch.tensor([21, 9], dtype=torch.int32, device=DEV); tl = torch.tensor([4, 2], dtype=torch.int32, device=DEV)
    tt = torch.arange(T, device=DEV)[None, :, None] < ll[:, None, None]
    uu = torch.arange(U1, device=DEV)[None, None, :] <= tl[:, None, None]
    ok = (tt & uu)[..., None].double()
    d_ep, d_pp, d_w, d_b = joint_backward(gout, ep, pp, w, ll, tl, 1, True, True)
    g64 = gout.double() * ok
    h = torch.tanh(ep[:, :, None, :] + pp[:, None, :, :]).double()
    dz = (g64 @ w.to(dt).double()) * (1 - h * h) * ok
    want = (dz.sum(2), dz.sum(1), g64.reshape(-1, V).T @ (h.to(dt).double() * ok).reshape(-1, J), g64.sum((0, 1, 2)))
    for a_, b_ in zip((d_ep, d_pp, d_w, d_b), want):
        assert a_.dtype == torch.float32
        assert float((a_.double() - b_).abs().max()) <= 1e-3 * float(b_.pow(2).mean().sqrt()) + 1e-6


@pytest.mark.parametrize("h_dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", ["tanh", "swish"])
def test_dz_act_and_db_entry_points(h_dtype, act):
    """wr_joint_dz_act (activation gradient in place + the activation copy, fp32 or bf16, with the ones column when the row
    stride exceeds J) and wr_joint_db_bf16 (column sums of a bf16 gradient over the valid cells) against PyTorch."""
    from wenet_celoss_amd import _lib
    from wenet_celoss_amd.joint import activation_code, _TORCH_ACTIVATIONS
    lib = _lib.load()
    B, T, U1, J, V = 2, 11, 4, 36, 72
    g = torch.Generator().manual_seed(17)
    ep = torch.randn(B, T, J, generator=g).to(DEV); pp = torch.randn(B, U1, J, generator=g).to(DEV)
    dh = torch.randn(B, T, U1, J, generator=g).to(DEV)
    ll = torch.tensor([11, 6], dtype=torch.int32, device=DEV); tl = torch.tensor([3, 1], dtype=torch.int32, device=DEV)
    tt = torch.arange(T, device=DEV)[None, :, None] < ll[:, None, None]
    uu = torch.arange(U1, device=DEV)[None, None, :] <= tl[:, None, None]
    valid = (tt & uu)
    st, P = _lib.current_stream(torch.device(DEV)), _lib.ptr
    z = (ep[:, :, None, :] + pp[:, None, :, :]).double().requires_grad_(True)
    hh = _TORCH_ACTIVATIONS[act]()(z)
    (dact,) = torch.autograd.grad(hh.sum(), z)
    for lens, ok in (((None, None), torch.ones_like(valid)), ((ll, tl), valid)):
        for ld in (J, J + 8):
            dz = dh.clone()
            h = torch.full((B, T, U1, ld), float("nan"), dtype=h_dtype, device=DEV)
            _lib.check(lib.wr_joint_dz_act(P(dz), P(ep), P(pp), P(lens[0]), P(lens[1]), B, T, U1, J, activation_code(act), P(h),
                                           _lib.dtype_code(h_dtype), ld, st))
            want_dz = (dh.double() * dact) * ok[..., None]
            torch.testing.assert_close(dz.double(), want_dz, rtol=2e-5, atol=2e-6)
            want_h = (hh.detach() * ok[..., None])
            tol = dict(rtol=1e-5, atol=1e-6) if h_dtype == torch.float32 else dict(rtol=8e-3, atol=1e-6) \
                if h_dtype == torch.bfloat16 else dict(rtol=1e-3, atol=1e-6)
            torch.testing.assert_close(h[..., :J].double(), want_h, **tol)
            if ld > J:
                assert torch.equal(h[..., J].float(), ok.float())
                assert float(h[..., J + 1:].float().abs().max()) == 0.0
    # bias gradient
    gout = torch.randn(B, T, U1, V, generator=g).to(DEV).to(torch.bfloat16)
    gout[1, 8:] = float("nan")                                       # padded frames may hold anything
    wsb = lib.wr_joint_db_workspace_bytes(B, T, U1, V); ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    db = torch.full((V,), float("nan"), device=DEV)
    _lib.check(lib.wr_joint_db_bf16(P(gout), P(ll), P(tl), B, T, U1, V, P(db), P(ws), wsb, st))
    want = torch.where(valid[..., None], gout.float(), torch.zeros((), device=DEV)).double().sum(dim=(0, 1, 2))
    torch.testing.assert_close(db.double(), want, rtol=1e-5, atol=1e-5)
    gout = torch.nan_to_num(gout.float()).to(torch.bfloat16)
    _lib.check(lib.wr_joint_db_bf16(P(gout), None, None, B, T, U1, V, P(db), P(ws), wsb, st))
    torch.testing.assert_close(db.double(), gout.double().sum(dim=(0, 1, 2)), rtol=1e-5, atol=1e-5)
    assert lib.wr_joint_db_bf16(P(gout), None, None, B, T, U1, 36, P(db), P(ws), wsb, st) != 0      # V % 8
    g16 = gout.to(torch.float16)
    _lib.check(lib.wr_joint_db_f16(P(g16), P(ll), P(tl), B, T, U1, V, P(db), P(ws), wsb, st))
    want = torch.where(valid[..., None], g16.float(), torch.zeros((), device=DEV)).double().sum(dim=(0, 1, 2))
    torch.testing.assert_close(db.double(), want, rtol=1e-5, atol=1e-5)


def test_split_training_step_with_lengths_matches_exact(monkeypatch):
    """Joiner + RNN-T loss, forward and backward, ragged lengths: every gradient of the bf16x3 mode against the exact
    mode's (tolerances at the assertion); the environment switch selects the same path as the argument."""
    import wenet_celoss_amd as w
    torch.manual_seed(8)
    B, T, U, E, P, J, V = 3, 60, 40, 16, 16, 128, 200
    m = w.TransducerJoint(V, E, P, J).to(DEV)
    enc = torch.randn(B, T, E, device=DEV); pred = torch.randn(B, U + 1, P, device=DEV)
    y = torch.randint(1, V, (B, U), dtype=torch.int32, device=DEV)
    ll = torch.tensor([60, 33, 7], dtype=torch.int32, device=DEV)
    tl = torch.tensor([12, 40, 5], dtype=torch.int32, device=DEV)

    def run(lens):
        m.zero_grad()
        e = enc.clone().requires_grad_(True); p = pred.clone().requires_grad_(True)
        logits = m(e, p, ll, tl) if lens else m(e, p)
        loss = w.rnnt_loss(logits, y, ll, tl, blank=0, reduction="mean")
        loss.backward()
        return [loss.detach(), e.grad, p.grad] + [prm.grad.clone() for prm in m.parameters()]

    ref = run(False)
    m.precision = "bf16x3"
    for lens in (False, True):
        got = run(lens)
        assert float(got[0]) == pytest.approx(float(ref[0]), rel=1e-6)
        for a, b in zip(got[1:], ref[1:]):
            # end to end the error of single elements follows sum |dY H| rather than the cancelled sums: the bar is
            # 1e-4 of the tensor's largest magnitude for the worst element and 2e-5 of its r.m.s. in the r.m.s. sense
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-9
            assert float((a - b).pow(2).mean().sqrt()) <= 2e-5 * float(b.pow(2).mean().sqrt()) + 1e-9
    m.precision = None
    monkeypatch.setenv("WR_JOINT_PRECISION", "bf16x3")
    env = run(True)
    m.precision = "bf16x3"
    arg = run(True)
    for a, b in zip(env, arg):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_amp_training_step_tracks_fp32(dt):
    """The --use_amp configuration (executor.py:91): precision="bf16" under autocast hands 16-bit logits to the
    RNN-T loss kernels; loss and gradients stay within the 16-bit operand rounding of the exact step."""
    import wenet_celoss_amd as w
    torch.manual_seed(9)
    B, T, U, E, P, J, V = 2, 40, 12, 16, 16, 64, 128
    m = w.TransducerJoint(V, E, P, J).to(DEV)
    enc = torch.randn(B, T, E, device=DEV); pred = torch.randn(B, U + 1, P, device=DEV)
    y = torch.randint(1, V, (B, U), dtype=torch.int32, device=DEV)
    ll = torch.tensor([40, 31], dtype=torch.int32, device=DEV); tl = torch.tensor([12, 7], dtype=torch.int32, device=DEV)

    def run(amp):
        m.zero_grad()
        m.precision = "bf16" if amp else "fp32"
        e = enc.clone().requires_grad_(True); p = pred.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=dt, enabled=amp):
            logits = m(e, p)
            assert logits.dtype == (dt if amp else torch.float32)
            loss = w.rnnt_loss(logits, y, ll, tl, blank=0, reduction="mean")
        loss.float().backward()
        return [loss.detach().float(), e.grad, p.grad, m.ffn_out.weight.grad.clone()]

    ref, got = run(False), run(True)
    assert float(got[0]) == pytest.approx(float(ref[0]), rel=2e-2)
    for a, b in zip(got[1:], ref[1:]):
        assert torch.isfinite(a).all()
        assert float((a.float() - b).pow(2).mean().sqrt()) <= 5e-2 * float(b.pow(2).mean().sqrt())


@pytest.mark.parametrize("B,T,U1,J,V", [(2, 9, 5, 128, 300), (1, 70, 3, 256, 1000), (3, 11, 7, 512, 516), (2, 13, 4, 36, 64),
                                        (1, 300, 2, 260, 20)])
def test_exact_dz_block_tiling_matches_cell_tiling(B, T, U1, J, V):
    """wr_joint_bwd_dz's shipped tiling (256 x 256 blocks, dY transposed while staged) against the 64-cell kernel of
    joint.hip (tuning knob 10 = 1): both are exact-fp32 MFMA sums over v in ascending order (agreement to 1e-5 of the
    r.m.s. is asserted; measured: bit-identical); H and the zeros of padded cells identical; row tails V % 16, partial
    blocks in cells and columns."""
    from wenet_celoss_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(T + J + V)
    ep = torch.randn(B, T, J, generator=g).to(DEV); pp = torch.randn(B, U1, J, generator=g).to(DEV)
    w = (torch.randn(V, J, generator=g) * 0.1).to(DEV)
    gout = torch.randn(B, T, U1, V, generator=g).to(DEV)
    ll = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32); ll[0] = T
    tl = torch.randint(0, U1, (B,), generator=g).to(torch.int32); tl[0] = U1 - 1
    ll, tl = ll.to(DEV), tl.to(DEV)
    st, P = _lib.current_stream(torch.device(DEV)), _lib.ptr
    try:
        for lens in ((None, None), (ll, tl)):
            res = []
            for knob in (0, 1):
                lib.wr_tune_set(10, knob)
                dz = torch.full((B, T, U1, J), float("nan"), device=DEV); h = torch.full_like(dz, float("nan"))
                _lib.check(lib.wr_joint_bwd_dz(P(gout), P(ep), P(pp), P(w), P(lens[0]), P(lens[1]), B, T, U1, J, V, 0, P(dz), P(h), st))
                res.append((dz, h))
            assert torch.equal(res[0][1], res[1][1])
            assert torch.equal(res[0][0] == 0, res[1][0] == 0) or lens[0] is None
            rms = float(res[0][0].pow(2).mean().sqrt())
            assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-5 * rms
    finally:
        lib.wr_tune_set(10, 0)


@pytest.mark.parametrize("prejoin,postjoin", [(False, False), (True, True), (False, True)])
def test_joiner_variants_match_the_reference_formula(prejoin, postjoin):
    """joint.py:45-70 with prejoin_linear off and / or postjoin_linear on, evaluated literally in float64 (the
    post-join Linear applied to the 4-D sum), against the module (which distributes that Linear over the two addends
    and never forms the 4-D tensor): logits and every gradient."""
    import wenet_celoss_amd as w
    torch.manual_seed(5)
    B, T, U1, D, V = 2, 9, 4, 24, 37
    m = w.TransducerJoint(V, D, D, D, prejoin_linear=prejoin, postjoin_linear=postjoin).to(DEV)
    enc = torch.randn(B, T, D, device=DEV, requires_grad=True)
    pred = torch.randn(B, U1, D, device=DEV, requires_grad=True)
    out = m(enc, pred)
    gout = torch.randn_like(out)
    out.backward(gout)
    prm = {k: v.detach().double().cpu().requires_grad_(True) for k, v in m.state_dict().items()}
    e, p = enc.detach().double().cpu().requires_grad_(True), pred.detach().double().cpu().requires_grad_(True)
    e2, p2 = e, p
    if prejoin:
        e2 = e @ prm["enc_ffn.weight"].T + prm["enc_ffn.bias"]
        p2 = p @ prm["pred_ffn.weight"].T + prm["pred_ffn.bias"]
    x = e2[:, :, None] + p2[:, None]
    if postjoin:
        x = x @ prm["post_ffn.weight"].T + prm["post_ffn.bias"]
    ref = torch.tanh(x) @ prm["ffn_out.weight"].T + prm["ffn_out.bias"]
    ref.backward(gout.double().cpu())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    for name, got, want in [("enc", enc.grad, e.grad), ("pred", pred.grad, p.grad)] + \
            [(k, dict(m.named_parameters())[k].grad, prm[k].grad) for k in prm]:
        scale = float(want.abs().max())
        assert float((got.double().cpu() - want).abs().max()) <= 1e-4 * scale + 1e-9, name
