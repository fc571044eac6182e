#!/usr/bin/env python3
"""Secondary measurements for bench.py's `"extra"` object (SURVEY.md section 8d "Secondary": CTC loss+grad utt/s, joiner
TFLOP/s, greedy steps/s and utt/s, beam frames/s) -- the numbers `tools/bench_extra.py` prints in development, in a
bounded form (about 20-40 s on one MI355X) so that the driver-run bench line carries them.

Every leg is guarded on its own: a failure is recorded as {"error": ...} and the other legs still run; the headline
fields of bench.py never depend on anything here.  All legs go through the product's public entry points (the C-ABI via
wenet_celoss_amd), with synthetic inputs of the BASELINE.json shapes:
  joiner        B=8 slice of configs[1] (T=1000, U=150, J=512, V=5000): exact-fp32 forward / dZ / dW, TFLOP/s and fraction
                of the 157.3 TFLOP/s fp32-matrix peak; single-term bf16 forward (the --use_amp arithmetic), fraction of the
                2.5 PFLOP/s dense bf16 peak; three-term split forward in fp32-equivalent TFLOP/s
  loss_block    joiner + RNN-T loss, forward + backward through autograd (the fused node, exact fp32), B=16, ms per step
  ctc           loss + gradient at (T=1000, B=32, S=150, V=5000), utt/s, with torch.nn.CTCLoss on the host cores beside it
  greedy        configs[2]: 64 streams x 64 encoder frames, V=5000, n_steps=64: utt/s, us per micro-step
  beam          configs[4]: beam 8, B=16, T=1500, V=5000: frames/s, us per frame
  hotword       the fork's default decode path at the shipped dimensions, one stream: us per joiner decision
"""
from __future__ import annotations

import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MATRIX_PEAK_TF = 157.3      # MI355X_MICROARCH.md: dense fp32 MFMA
BF16_DENSE_PEAK_TF = 2500.0     # dense bf16 MFMA (no sparsity)


def _median_ms(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def _wall_ms(fn, steps, warmup=1):
    for _ in range(warmup):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def leg_joiner(dev, steps=3):
    from wenet_celoss_amd import _lib
    lib = _lib.load()
    B, T, U1, J, V = 8, 1000, 151, 512, 5000
    g = torch.Generator(device=dev).manual_seed(1)
    ep = torch.randn(B, T, J, device=dev, generator=g); pp = torch.randn(B, U1, J, device=dev, generator=g)
    w = torch.randn(V, J, device=dev, generator=g) * 0.05; b = torch.randn(V, device=dev, generator=g)
    out = torch.empty(B, T, U1, V, device=dev)
    st = _lib.current_stream(dev); P = _lib.ptr
    flops = 2.0 * B * T * U1 * J * V
    res = {"shape": {"B": B, "T": T, "U1": U1, "J": J, "V": V}, "flop_per_call": flops}
    wsb = lib.wr_joint_workspace_bytes(J, V)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ms = _median_ms(lambda: _lib.check(lib.wr_joint_fwd(P(ep), P(pp), P(w), P(b), None, None, B, T, U1, J, V, 0, P(out),
                                                        P(ws), wsb, st)), steps)
    res["fwd_f32"] = {"ms": round(ms, 3), "TFLOPs": round(flops / ms / 1e9, 1), "frac": round(flops / ms / 1e9 / F32_MATRIX_PEAK_TF, 4)}
    # split-precision forwards on the bf16 matrix cores
    wss = lib.wr_joint_split_workspace_bytes(J, V)
    ws_s = torch.empty(wss, dtype=torch.uint8, device=dev)
    out16 = torch.empty(B, T, U1, V, dtype=torch.bfloat16, device=dev)
    ms = _median_ms(lambda: _lib.check(lib.wr_joint_fwd_split(P(ep), P(pp), P(w), P(b), None, None, B, T, U1, J, V, 0, 1,
                                                              P(out16), _lib.WR_BF16, P(ws_s), wss, st)), steps)
    res["fwd_bf16_single_term"] = {"ms": round(ms, 3), "TFLOPs": round(flops / ms / 1e9, 1),
                                   "frac": round(flops / ms / 1e9 / BF16_DENSE_PEAK_TF, 4), "logits": "bf16"}
    # reference point: the vendor GEMM library on the BARE contraction of the same shape (H given in bf16, no tanh, no bias)
    try:
        hb = torch.tanh(torch.randn(B * T * U1, J, device=dev, generator=g)).to(torch.bfloat16)
        wb = w.to(torch.bfloat16)
        o2 = out16.view(B * T * U1, V)
        ms = _median_ms(lambda: torch.mm(hb, wb.t(), out=o2), steps)
        res["vendor_gemm_bf16_bare_contraction"] = {"ms": round(ms, 3), "TFLOPs": round(flops / ms / 1e9, 1),
                                                    "frac": round(flops / ms / 1e9 / BF16_DENSE_PEAK_TF, 4)}
        del hb, wb, o2
    except Exception as e:
        res["vendor_gemm_bf16_bare_contraction"] = {"error": str(e)[:120]}
    del out16
    out3 = torch.empty(B, T, U1, V, device=dev)
    ms = _median_ms(lambda: _lib.check(lib.wr_joint_fwd_split(P(ep), P(pp), P(w), P(b), None, None, B, T, U1, J, V, 0, 3,
                                                              P(out3), _lib.WR_F32, P(ws_s), wss, st)), steps)
    res["fwd_bf16x3"] = {"ms": round(ms, 3), "TFLOPs_fp32_equiv": round(flops / ms / 1e9, 1),
                         "bf16_TFLOPs": round(3 * flops / ms / 1e9, 1), "frac": round(3 * flops / ms / 1e9 / BF16_DENSE_PEAK_TF, 4),
                         "max_abs_err_vs_f32": float((out3[0, :4] - out[0, :4]).abs().max())}
    del out3
    dz = torch.empty(B, T, U1, J, device=dev); h = torch.empty_like(dz)
    ms = _median_ms(lambda: _lib.check(lib.wr_joint_bwd_dz(P(out), P(ep), P(pp), P(w), None, None, B, T, U1, J, V, 0, P(dz),
                                                           P(h), st)), steps)
    res["dz_f32"] = {"ms": round(ms, 3), "TFLOPs": round(flops / ms / 1e9, 1), "frac": round(flops / ms / 1e9 / F32_MATRIX_PEAK_TF, 4)}
    wsb2 = lib.wr_joint_dw_workspace_bytes(J, V)
    ws2 = torch.empty(wsb2, dtype=torch.uint8, device=dev)
    dwt = torch.empty(V, J, device=dev); dbt = torch.empty(V, device=dev)
    ms = _median_ms(lambda: _lib.check(lib.wr_joint_bwd_dw(P(out), P(h), None, None, B, T, U1, J, V, P(dwt), P(dbt), P(ws2),
                                                           wsb2, st)), steps)
    res["dw_f32"] = {"ms": round(ms, 3), "TFLOPs": round(flops / ms / 1e9, 1), "frac": round(flops / ms / 1e9 / F32_MATRIX_PEAK_TF, 4)}
    res["peaks_TFLOPs"] = {"f32_matrix": F32_MATRIX_PEAK_TF, "bf16_dense": BF16_DENSE_PEAK_TF}
    return res


def leg_loss_block(dev, steps=2):
    import wenet_celoss_amd as w
    torch.manual_seed(3)
    B, T, U, V, E, Pd, J = 16, 1000, 150, 5000, 256, 256, 512
    enc = torch.randn(B, T, E, device=dev, requires_grad=True)
    pred = torch.randn(B, U + 1, Pd, device=dev, requires_grad=True)
    y = torch.randint(1, V, (B, U), dtype=torch.int32, device=dev)
    ll = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), U, dtype=torch.int32, device=dev)
    joint = w.TransducerJoint(V, E, Pd, J, precision="fp32").to(dev)
    with torch.no_grad():
        for prm in joint.parameters():
            prm.copy_(torch.randn_like(prm) * 0.05)

    def step():
        joint.zero_grad(set_to_none=True); enc.grad = None; pred.grad = None
        loss = w.joint_rnnt_loss(joint.enc_ffn(enc), joint.pred_ffn(pred), joint.ffn_out.weight, joint.ffn_out.bias, y, ll, tl,
                                 blank=0, reduction="mean", precision="fp32")
        loss.backward()
        return loss
    ms = _median_ms(step, steps)
    return {"what": "pre-join projections -> joiner -> RNN-T loss -> backward to encoder/predictor outputs and joiner weights "
                    "(fused node, exact fp32)", "B": B, "T": T, "U": U, "V": V, "J": J, "ms_per_step": round(ms, 2),
            "utt_per_s": round(B / ms * 1e3, 2)}


def leg_amp_block(dev, steps=2):
    """The loss block as the reference runs it under --use_amp (executor.py:91 autocast): 16-bit logits from the single-term
    joiner forward, rnnt_loss on them, backward with the bf16 gradient (library GEMMs around wr_joint_dz_act / _db_bf16)."""
    import wenet_celoss_amd as w
    torch.manual_seed(3)
    torch.cuda.reset_peak_memory_stats()
    B, T, U, V, E, Pd, J = 16, 1000, 150, 5000, 256, 256, 512
    enc = torch.randn(B, T, E, device=dev, requires_grad=True)
    pred = torch.randn(B, U + 1, Pd, device=dev, requires_grad=True)
    y = torch.randint(1, V, (B, U), dtype=torch.int32, device=dev)
    ll = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), U, dtype=torch.int32, device=dev)
    joint = w.TransducerJoint(V, E, Pd, J, precision="bf16").to(dev)
    with torch.no_grad():
        for prm in joint.parameters():
            prm.copy_(torch.randn_like(prm) * 0.05)

    def step():
        joint.zero_grad(set_to_none=True); enc.grad = None; pred.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = w.rnnt_loss(joint(enc, pred), y, ll, tl, blank=0, reduction="mean")
        loss.backward()
        return loss
    ms = _median_ms(step, steps)
    return {"what": "AMP loss block: pre-join projections -> single-term bf16 joiner (bf16 logits) -> RNN-T loss -> backward",
            "B": B, "T": T, "U": U, "V": V, "J": J, "ms_per_step": round(ms, 2), "utt_per_s": round(B / ms * 1e3, 2),
            "max_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}


def leg_ctc(dev, steps=20):
    from wenet_celoss_amd import _lib
    lib = _lib.load()
    B, T, S, V = 32, 1000, 150, 5000
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(B, T, V, device=dev, generator=g)
    y = torch.randint(1, V, (B, S), dtype=torch.int32, device=dev, generator=g)
    il = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), S, dtype=torch.int32, device=dev)
    wsb = lib.wr_ctc_workspace_bytes(B, T, S)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    nll = torch.empty(B, device=dev); gr = torch.empty_like(x); go = torch.full((B,), 1.0 / B, device=dev)
    st = _lib.current_stream(dev); P = _lib.ptr

    def step():
        _lib.check(lib.wr_ctc_loss_fwd(P(x), 0, P(y), P(il), P(tl), B, T, S, V, 0, P(nll), P(ws), wsb, st))
        _lib.check(lib.wr_ctc_loss_bwd(P(x), 0, P(y), P(il), P(tl), B, T, S, V, 0, P(go), P(gr), P(ws), wsb, st))
    ms = _median_ms(step, steps, warmup=3)
    res = {"shape": {"T": T, "B": B, "S": S, "V": V}, "ms_per_step": round(ms, 4), "utt_per_s": round(B / ms * 1e3, 1),
           "algorithmic_GBps(3*4*T*B*V)": round(3 * 4.0 * T * B * V / ms / 1e6, 1), "dependent_steps_per_s": round(T / ms * 1e3)}
    # the reference's own call (ctc.py:60-61) on the host cores, one warm-up + one timed run
    xc = x.cpu().requires_grad_(True)
    yc, ilc, tlc = y.cpu().long(), il.cpu().long(), tl.cpu().long()
    threads = torch.get_num_threads()
    for k in range(2):
        t0 = time.perf_counter()
        lp = xc.transpose(0, 1).log_softmax(2)
        loss = torch.nn.CTCLoss(reduction="sum")(lp, yc, ilc, tlc) / B
        loss.backward()
        cpu_ms = (time.perf_counter() - t0) * 1e3
    res["cpu_torch_ctcloss"] = {"ms_per_step": round(cpu_ms, 1), "utt_per_s": round(B / cpu_ms * 1e3, 1), "threads": threads,
                                "nll_rel_diff_vs_gpu": abs(float(loss) - float(nll.sum() / B)) / abs(float(loss))}
    return res


def _decode_modules(dev, seed):
    import wenet_celoss_amd as w
    torch.manual_seed(seed)
    V, E, Pd, J, H, L = 5000, 256, 256, 512, 256, 2
    pred = w.RNNPredictor(V, Pd, Pd, 0.1, H, L).to(dev).eval()
    joint = w.TransducerJoint(V, E, Pd, J).to(dev).eval()
    with torch.no_grad():
        joint.ffn_out.weight *= 10
        joint.ffn_out.bias[0] += 16.0
    return pred, joint


def leg_greedy(dev, steps=5):
    import wenet_celoss_amd as w
    pred, joint = _decode_modules(dev, 5)
    N, T, E = 64, 64, 256
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    enc = torch.randn(N, T, E, device=dev)
    lens = torch.full((N,), T)
    ms, hyps = _wall_ms(lambda: w.basic_greedy_search(model, enc, lens, n_steps=64), steps, warmup=2)
    ntok = sum(len(h) for h in hyps)
    micro = max(len(h) for h in hyps) + T               # lanes advance in lock step: a micro-step per token or frame of the slowest
    res = {"config": "64 streams x 64 encoder frames (4 chunks of 16), V=5000, LSTM 2x256, J=512, n_steps=64, random weights",
           "ms_per_call": round(ms, 3), "utt_per_s": round(N / ms * 1e3, 1), "tokens": ntok, "micro_steps": micro,
           "us_per_micro_step": round(ms * 1e3 / micro, 2), "lane_steps_per_s": round((ntok + N * T) / ms * 1e3)}
    # a speech-shaped stream (most frames blank) with the adaptive look-ahead
    gq = torch.Generator(device=dev).manual_seed(11)
    quiet = torch.randn(N, T, E, device=dev, generator=gq) * 0.1
    loud = torch.randn(N, T, E, device=dev, generator=gq) * 2.5
    spikes = torch.rand(N, T, device=dev, generator=gq) < 0.25
    e2 = torch.where(spikes[..., None], loud, quiet)
    with torch.no_grad():
        joint.ffn_out.bias[0] += 1.0
    m2 = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    ms2, h2 = _wall_ms(lambda: w.basic_greedy_search(m2, e2, lens, n_steps=2), steps, warmup=2)
    res["speech_shaped"] = {"ms_per_call": round(ms2, 3), "utt_per_s": round(N / ms2 * 1e3, 1), "tokens": sum(len(h) for h in h2)}
    return res


def leg_beam(dev, steps=3):
    import wenet_celoss_amd as w
    pred, joint = _decode_modules(dev, 6)
    B, T, beam = 16, 1500, 8
    ctc = w.CTC(5000, 256).to(dev).eval()
    bs = w.PrefixBeamSearch(None, pred, joint, ctc, 0)
    enc = torch.randn(B, T, 256, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32)
    ms, out = _wall_ms(lambda: bs.search_encoded(enc, lens, beam_size=beam), steps, warmup=2)
    return {"config": "beam 8, B=16, T=1500, V=5000, ctc_weight 0.3 / transducer_weight 0.7", "ms_per_call": round(ms, 2),
            "frames_per_s": round(B * T / ms * 1e3), "utt_per_s": round(B / ms * 1e3, 2), "us_per_frame": round(ms * 1e3 / T, 2),
            "best_len": len(out[0][0].hyp)}


def leg_hotword(dev, steps=3):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from context_bias_mirror import ContextBiasMirror
    import wenet_celoss_amd as w
    from wenet_celoss_amd.hotword import greedy_search_both_device
    torch.manual_seed(12)
    V, D, J, H, L, HW, T = 5000, 256, 512, 256, 2, 100, 200
    pred = w.RNNPredictor(V, D, D, 0.1, H, L).eval()
    joint = w.TransducerJoint(V, D, D, J).eval()
    cb = ContextBiasMirror(V, D, layers=1, heads=4, hw_dim=HW, hw_heads=4).eval()
    with torch.no_grad():
        joint.ffn_out.weight *= 10
        joint.ffn_out.bias[0] += 10.0
        cb.hw_output_layer_enc.weight.mul_(6.0)
        cb.hw_output_layer.weight.mul_(4.0)
    n_ctx = 50
    ctx = torch.randint(1, V, (n_ctx, 6)); ctx_len = torch.randint(2, 7, (n_ctx,)).to(torch.int32); ctx[0, 0] = 0; ctx_len[0] = 1
    m = w.Transducer(V, 0, torch.nn.Identity(), pred.to(dev), joint.to(dev), context_bias=cb.to(dev), ctc_weight=0.0,
                     transducer_weight=1.0, loss_mode="both")
    enc = torch.randn(1, T, D, device=dev)
    lens = torch.full((1,), T)
    run = lambda: greedy_search_both_device(m, enc, lens, ctx, ctx_len, n_steps=64, filter_on=True)
    ms, (hyps, traces) = _wall_ms(run, steps, warmup=2)
    decisions = len(hyps[0]) + T                        # lower bound: a go-back re-decodes frames
    res = {"config": "hot-word greedy ('both', filter on), 1 stream, T=200, V=5000, D=256, 4 heads, hw_odim 100, 50 hot words",
           "ms_per_call": round(ms, 2), "tokens": len(hyps[0]), "gate_ones": sum(t.count(1) for t in traces),
           "gate_zeros": sum(t.count(0) for t in traces), "us_per_decision": round(ms * 1e3 / decisions, 1)}
    # filter off: every predictor step is biased with the hot-word list (all gates 1) -- the cost of a gate-1 decision
    run1 = lambda: greedy_search_both_device(m, enc, lens, ctx, ctx_len, n_steps=64, filter_on=False)
    ms1, (h1, t1) = _wall_ms(run1, steps, warmup=1)
    res["filter_off_all_gate_1"] = {"ms_per_call": round(ms1, 2), "tokens": len(h1[0]),
                                    "us_per_decision": round(ms1 * 1e3 / (len(h1[0]) + T), 1)}
    return res


LEGS = (("joiner", leg_joiner), ("loss_block", leg_loss_block), ("amp_block", leg_amp_block), ("ctc", leg_ctc),
        ("greedy", leg_greedy), ("beam", leg_beam), ("hotword", leg_hotword))


def collect(dev, budget_s: float = 90.0, only=None):
    """Run the legs in order until the time budget is used up; each leg in its own try block."""
    out = {}
    t_start = time.perf_counter()
    for name, fn in LEGS:
        if only and name not in only:
            continue
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"skipped": f"time budget of {budget_s:.0f} s used up"}
            continue
        t0 = time.perf_counter()
        try:
            with torch.cuda.device(dev):
                out[name] = fn(dev)
        except Exception as e:                       # a broken leg must not take the headline line down
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        out[name]["leg_seconds"] = round(time.perf_counter() - t0, 1)
    out["seconds"] = round(time.perf_counter() - t_start, 1)
    return out


if __name__ == "__main__":
    import json
    names = sys.argv[1:] or None
    print(json.dumps(collect(torch.device("cuda:0"), budget_s=600.0, only=names), indent=1))
