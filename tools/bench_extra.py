#!/usr/bin/env python3
"""Secondary measurements (SURVEY.md 8d "Secondary"): joiner TFLOP/s, CTC loss+grad utt/s.
Not the headline metric (bench.py is); prints one JSON line per measurement."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def bench_joint(args):
    from wenet_celoss_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, T, U1, J, V = args.B, args.T, args.U + 1, 512, args.V
    ep = torch.randn(B, T, J, device=dev); pp = torch.randn(B, U1, J, device=dev)
    w = torch.randn(V, J, device=dev) * 0.05; b = torch.randn(V, device=dev)
    out = torch.empty(B, T, U1, V, device=dev)
    ws_bytes = lib.wr_joint_workspace_bytes(J, V)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    st = _lib.current_stream(dev); P = _lib.ptr
    f = lambda: _lib.check(lib.wr_joint_fwd(P(ep), P(pp), P(w), P(b), None, None, B, T, U1, J, V, 0, P(out), P(ws), ws_bytes, st))
    flops = 2.0 * B * T * U1 * J * V
    ms = timeit(f, args.steps)
    print(json.dumps({"what": "joint_fwd", "shape": [B, T, U1, J, V], "ms": round(ms, 3),
                      "TFLOPs": round(flops / ms / 1e9, 2), "peak_f32_mfma": 157.3,
                      "frac": round(flops / ms / 1e9 / 157.3, 4), "probe": float(out[0, 0, 0, :8].abs().sum())}), flush=True)
    if args.fwd_only:
        return
    # split-precision variants on the bf16 matrix cores, checked against the exact-fp32 logits just computed
    wsb_s = lib.wr_joint_split_workspace_bytes(J, V)
    ws_s = torch.empty(wsb_s, dtype=torch.uint8, device=dev)
    scale = float(out[0, :8].std())
    for terms, odt, code, parts in ((3, torch.float32, 0, 1), (3, torch.float32, 0, 2), (3, torch.float32, 0, 4),
                                    (1, torch.float32, 0, 1), (1, torch.float32, 0, 2), (1, torch.bfloat16, 2, 1), (1, torch.bfloat16, 2, 2)):
        lib.wr_tune_set(7, parts)
        out_s = torch.empty(B, T, U1, V, dtype=odt, device=dev)
        fs = lambda: _lib.check(lib.wr_joint_fwd_split(P(ep), P(pp), P(w), P(b), None, None, B, T, U1, J, V, 0, terms,
                                                       P(out_s), code, P(ws_s), wsb_s, st))
        ms = timeit(fs, args.steps)
        err = float((out_s[:1].float() - out[:1]).abs().max())
        print(json.dumps({"what": "joint_fwd_split", "terms": terms, "parts": parts, "out": str(odt).split(".")[-1],
                          "shape": [B, T, U1, J, V], "ms": round(ms, 3), "TFLOPs_fp32_equiv": round(flops / ms / 1e9, 2),
                          "bf16_TFLOPs": round(terms * flops / ms / 1e9, 1), "frac_of_2500": round(terms * flops / ms / 1e9 / 2500, 4),
                          "max_abs_err_vs_fp32": err, "logit_std": scale, "rel_to_scale": err / scale}), flush=True)
        del out_s
    lib.wr_tune_set(7, 0)
    dz = torch.empty(B, T, U1, J, device=dev); h = torch.empty_like(dz)
    g = lambda: _lib.check(lib.wr_joint_bwd_dz(P(out), P(ep), P(pp), P(w), None, None, B, T, U1, J, V, 0, P(dz), P(h), st))
    dz_blocks = None
    for blocks in (1, 0):                              # knob 10: 0 = 256 x 256 block tiling (default), 1 = 64-cell tiling
        lib.wr_tune_set(10, 0 if blocks else 1)
        ms = timeit(g, args.steps)
        rec = {"what": "joint_bwd_dz", "tiling": "blocks" if blocks else "cells64", "shape": [B, T, U1, J, V], "ms": round(ms, 3),
               "TFLOPs": round(flops / ms / 1e9, 2), "frac": round(flops / ms / 1e9 / 157.3, 4)}
        if blocks:
            dz_blocks = dz[:1].clone()
        else:
            rec["max_abs_diff_blocks_vs_cells64"] = float((dz[:1] - dz_blocks).abs().max())
        print(json.dumps(rec), flush=True)
    lib.wr_tune_set(10, 0)
    wsz = lib.wr_joint_dz_split_workspace_bytes(J, V)
    wz = torch.empty(wsz, dtype=torch.uint8, device=dev)
    dz_ref = dz.clone()
    for terms in (3, 1):
        gs = lambda: _lib.check(lib.wr_joint_bwd_dz_split(P(out), P(ep), P(pp), P(w), None, None, B, T, U1, J, V, 0, terms,
                                                           P(dz), P(h), P(wz), wsz, st))
        ms = timeit(gs, args.steps)
        err = float((dz[:1] - dz_ref[:1]).abs().max())
        print(json.dumps({"what": "joint_bwd_dz_split", "terms": terms, "shape": [B, T, U1, J, V], "ms": round(ms, 3),
                          "TFLOPs_fp32_equiv": round(flops / ms / 1e9, 2), "max_abs_err_vs_fp32": err,
                          "dz_rms": float(dz_ref[:1].pow(2).mean().sqrt())}), flush=True)
    g()
    wsb2 = lib.wr_joint_dw_workspace_bytes(J, V)
    ws2 = torch.empty(wsb2, dtype=torch.uint8, device=dev)
    dwt = torch.empty(V, J, device=dev); dbt = torch.empty(V, device=dev)
    kdw = lambda: _lib.check(lib.wr_joint_bwd_dw(P(out), P(h), None, None, B, T, U1, J, V, P(dwt), P(dbt), P(ws2), wsb2, st))
    for slabs in (1, 0):                               # knob 9: 1 = first tiling (128-row slabs), 0 = 256 x 256 blocks (default)
        lib.wr_tune_set(9, slabs)
        ms = timeit(kdw, args.steps)
        print(json.dumps({"what": "joint_bwd_dw", "tiling": "slabs" if slabs else "blocks", "shape": [B, T, U1, J, V],
                          "ms": round(ms, 3), "TFLOPs": round(flops / ms / 1e9, 2),
                          "frac": round(flops / ms / 1e9 / 157.3, 4)}), flush=True)
    dw_ref, db_ref = dwt.clone(), dbt.clone()
    wsb3 = lib.wr_joint_dw_split_workspace_bytes(B, T, U1, J, V)
    ws3 = torch.empty(wsb3, dtype=torch.uint8, device=dev)
    for terms in (3, 1):
        ks = lambda: _lib.check(lib.wr_joint_bwd_dw_split(P(out), P(h), None, None, B, T, U1, J, V, terms, P(dwt), P(dbt),
                                                           P(ws3), wsb3, st))
        ms = timeit(ks, args.steps)
        rms = float(dw_ref.pow(2).mean().sqrt())
        print(json.dumps({"what": "joint_bwd_dw_split", "terms": terms, "shape": [B, T, U1, J, V], "ms": round(ms, 3),
                          "TFLOPs_fp32_equiv": round(flops / ms / 1e9, 2),
                          "max_err_over_rms": float((dwt - dw_ref).abs().max()) / rms,
                          "db_max_err_over_rms": float((dbt - db_ref).abs().max() / db_ref.pow(2).mean().sqrt())}), flush=True)
    if args.dw:
        # yardstick, not product: the library's fp32 GEMM (hipBLASLt / rocBLAS through torch) on the three contractions
        g2, h2 = out.view(-1, V), h.view(-1, J)
        dz2 = dz.view(-1, J)
        for name, k in (("forward  H @ W^T + b", lambda: torch.addmm(b, h2, w.t(), out=g2)),
                        ("dZ       dY @ W", lambda: torch.mm(g2, w, out=dz2)),
                        ("dW       dY^T @ H", lambda: g2.t().mm(h2))):
            ms = timeit(k, max(1, args.steps // 2))
            print(json.dumps({"what": "library fp32 GEMM (yardstick)", "contraction": name, "ms": round(ms, 3),
                              "TFLOPs": round(flops / ms / 1e9, 2), "frac": round(flops / ms / 1e9 / 157.3, 4)}), flush=True)


def bench_ctc(args):
    import wenet_celoss_amd as wc
    from wenet_celoss_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, T, S, V = 32, 1000, 150, 5000
    x = torch.randn(B, T, V, device=dev)
    y = torch.randint(1, V, (B, S), dtype=torch.int32, device=dev)
    il = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), S, dtype=torch.int32, device=dev)
    ws_bytes = lib.wr_ctc_workspace_bytes(B, T, S)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    nll = torch.empty(B, device=dev); g = torch.empty_like(x); go = torch.full((B,), 1.0 / B, device=dev)
    st = _lib.current_stream(dev); P = _lib.ptr

    def step():
        _lib.check(lib.wr_ctc_loss_fwd(P(x), 0, P(y), P(il), P(tl), B, T, S, V, 0, P(nll), P(ws), ws_bytes, st))
        _lib.check(lib.wr_ctc_loss_bwd(P(x), 0, P(y), P(il), P(tl), B, T, S, V, 0, P(go), P(g), P(ws), ws_bytes, st))
    ms = timeit(step, 20, warmup=3)
    # CPU reference, exactly the reference's call (ctc.py:60-61), on the host cores
    import time
    xc = x.cpu().requires_grad_(True)
    torch.set_num_threads(min(os.cpu_count(), 16))
    t0 = time.perf_counter()
    for _ in range(3):
        lp = xc.transpose(0, 1).log_softmax(2)
        l = torch.nn.CTCLoss(reduction="sum")(lp, y.cpu().long(), il.cpu().long(), tl.cpu().long()) / B
        l.backward()
    cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"what": "ctc_loss+grad", "shape": [B, T, S, V], "ms": round(ms, 4), "utt_per_s": round(B / ms * 1e3, 1),
                      "GBps_algorithmic(3*4*T*B*V)": round(3 * 4.0 * T * B * V / ms / 1e6, 1),
                      "steps_per_s": round(T / ms * 1e3), "cpu_torch_ctcloss_ms": round(cpu_ms, 1),
                      "cpu_utt_per_s": round(B / cpu_ms * 1e3, 1), "cpu_threads": torch.get_num_threads()}), flush=True)


def bench_ctcdec(args):
    """CTC decode modes and forced alignment from the ctc_lo output on: B=16, T=1500, V=5000."""
    import time
    import wenet_celoss_amd as wc
    from oracle import decode_oracle as do
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    B, T, V, S = (args.B if args.B != 32 else 16), (args.T if args.T != 1000 else 1500), 5000, 150
    x = torch.randn(B, T, V, device=dev) * 3
    x[..., 0] += 14.0                                  # blank-heavy, like a trained CTC head (about one label per 12 frames)
    lens = torch.full((B,), T)
    y = torch.randint(1, V, (B, S))

    def timed(fn, n):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, r

    dt, (hyps, _) = timed(lambda: wc.ctc_greedy_search(x, lens), args.steps)
    print(json.dumps({"what": "ctc_greedy_search", "B": B, "T": T, "V": V, "ms_per_call": round(dt * 1e3, 3),
                      "utt_per_s": round(B / dt, 1), "frames_per_s": round(B * T / dt), "tokens": sum(len(h) for h in hyps)}), flush=True)
    dt, res = timed(lambda: wc.ctc_prefix_beam_search(x, lens, 10), args.steps)
    print(json.dumps({"what": "ctc_prefix_beam_search", "beam": 10, "B": B, "T": T, "V": V, "ms_per_call": round(dt * 1e3, 3),
                      "utt_per_s": round(B / dt, 1), "frames_per_s": round(B * T / dt), "best_len": len(res[0][0][0])}), flush=True)
    dt, ali = timed(lambda: wc.forced_align_batch(x, y, lens, torch.full((B,), S)), args.steps)
    print(json.dumps({"what": "ctc_forced_align", "B": B, "T": T, "S": S, "ms_per_call": round(dt * 1e3, 3),
                      "utt_per_s": round(B / dt, 1), "frames_per_s": round(B * T / dt)}), flush=True)
    # CPU: the numpy restatement of the reference's prefix beam search on one utterance of 200 frames
    lp = do.log_softmax(x[0, :200].cpu().numpy())
    t0 = time.perf_counter()
    ref = do.ctc_prefix_beam_search(lp, 200, 10)
    cdt = time.perf_counter() - t0
    print(json.dumps({"what": "ctc_prefix_beam_search_cpu_oracle", "frames": 200, "s": round(cdt, 3),
                      "frames_per_s": round(200 / cdt)}), flush=True)


def bench_greedy(args):
    """BASELINE config 3: B=64 streams, chunks of 16 encoder frames, V=5000, LSTM 2x256, J=512, n_steps=64."""
    import time
    import types
    import wenet_celoss_amd as w
    from oracle import decode_oracle as do
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    V, E, P, J, H, L, N = 5000, 256, 256, 512, 256, 2, args.streams
    T = 16 * args.chunks
    pred = w.RNNPredictor(V, P, P, 0.1, H, L).to(dev).eval()
    joint = w.TransducerJoint(V, E, P, J).to(dev).eval()
    with torch.no_grad():
        joint.ffn_out.weight *= 10
        joint.ffn_out.bias[0] += 16.0
    model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
    enc = torch.randn(N, T, E, device=dev)
    lens = torch.full((N,), T)
    for use_graph in (True, False):
        hyps = w.basic_greedy_search(model, enc, lens, n_steps=64)          # builds the handle / graph
        model._decoder_cache._dec.set_graph(use_graph)
        w.basic_greedy_search(model, enc, lens, n_steps=64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hyps = w.basic_greedy_search(model, enc, lens, n_steps=64)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ntok = sum(len(h) for h in hyps)
        micro = ntok + N * T                                     # joiner evaluations actually needed
        print(json.dumps({"what": "greedy_search", "hipGraph": use_graph, "streams": N, "frames": T, "tokens": ntok,
                          "ms_per_call": round(dt * 1e3, 3), "utt_per_s": round(N / dt, 1),
                          "lane_steps_per_s": round(micro / dt)}), flush=True)
    # look-ahead: frames per micro-step (token sequences must not change).  Second scenario shaped like speech: most
    # frames quiet (blank), a quarter of them loud, at most 2 symbols per frame -> about one token per five frames
    # (the dense scenario above keeps emitting until n_steps stops it, which no trained model does).
    scenarios = [("dense", enc, 64, 0.0)]
    gq = torch.Generator(device=dev).manual_seed(11)
    quiet = torch.randn(N, T, E, device=dev, generator=gq) * 0.1
    loud = torch.randn(N, T, E, device=dev, generator=gq) * 2.5
    spikes = torch.rand(N, T, device=dev, generator=gq) < 0.25
    scenarios.append(("speech-like", torch.where(spikes[..., None], loud, quiet), 2, 1.0))
    for name, e_s, n_steps_s, extra_blank in scenarios:
        with torch.no_grad():
            joint.ffn_out.bias[0] += extra_blank
        m_s = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
        ref_hyps = w.basic_greedy_search(m_s, e_s, lens, n_steps=n_steps_s)
        for look in (1, 2, 4, 0):                      # 0: adaptive
            dec = m_s._decoder_cache._dec
            dec.set_lookahead(look)
            got = w.basic_greedy_search(m_s, e_s, lens, n_steps=n_steps_s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                got = w.basic_greedy_search(m_s, e_s, lens, n_steps=n_steps_s)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            print(json.dumps({"what": "greedy_search_lookahead", "scenario": name, "frames_per_micro_step": look,
                              "streams": N, "frames": T, "n_steps": n_steps_s, "tokens": sum(len(h) for h in got),
                              "max_tokens_per_stream": max(len(h) for h in got), "ms_per_call": round(dt * 1e3, 3),
                              "utt_per_s": round(N / dt, 1), "tokens_identical": got == ref_hyps}), flush=True)
        with torch.no_grad():
            joint.ffn_out.bias[0] -= extra_blank
        m_s._decoder_cache._dec.set_lookahead(0)
    # CPU reference: the reference's loop restated in numpy (oracle), one stream, scaled to 64
    p = do.Predictor({k: v.detach().cpu().numpy() for k, v in pred.state_dict().items()}, L)
    j = do.Joint({k: v.detach().cpu().numpy() for k, v in joint.state_dict().items()})
    t0 = time.perf_counter()
    ref = do.greedy_search(p, j, enc[0].cpu().numpy(), T, n_steps=64)
    cdt = time.perf_counter() - t0
    print(json.dumps({"what": "greedy_search_cpu_oracle", "one_stream_s": round(cdt, 3), "utt_per_s": round(1 / cdt, 2),
                      "tokens_match": ref == hyps[0]}), flush=True)


def bench_beam(args):
    """BASELINE config 5: beam=8, B=16, T=1500, V=5000."""
    import time
    import wenet_celoss_amd as w
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    V, E, P, J, H, L, B, T, beam = 5000, 256, 256, 512, 256, 2, args.B if args.B != 32 else 16, args.T if args.T != 1000 else 1500, 8
    pred = w.RNNPredictor(V, P, P, 0.1, H, L).to(dev).eval()
    joint = w.TransducerJoint(V, E, P, J).to(dev).eval()
    ctc = w.CTC(V, E).to(dev).eval()
    with torch.no_grad():
        joint.ffn_out.weight *= 10
        joint.ffn_out.bias[0] += 16.0
    bs = w.PrefixBeamSearch(None, pred, joint, ctc, 0)
    enc = torch.randn(B, T, E, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32)
    for use_graph in (True, False):
        out = bs.search_encoded(enc, lens, beam_size=beam)
        bs._decoder_cache._dec.set_graph(use_graph)
        bs.search_encoded(enc, lens, beam_size=beam)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = bs.search_encoded(enc, lens, beam_size=beam)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"what": "prefix_beam_search", "hipGraph": use_graph, "B": B, "T": T, "beam": beam,
                          "ms_per_call": round(dt * 1e3, 2), "frames_per_s": round(B * T / dt),
                          "utt_per_s": round(B / dt, 2), "best_len": len(out[0][0].hyp)}), flush=True)


def bench_hotword(args):
    """Hot-word greedy search (the fork's default decode path, greedy_search.py:297-430) at the shipped dimensions:
    the loop on the device (gate table + fused predictor biasing + state machine in the update kernel, hipGraph) next
    to the host-driven loop of round 1, one utterance as the reference decodes, and several streams together."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from context_bias_mirror import ContextBiasMirror
    import wenet_celoss_amd as w
    from wenet_celoss_amd.hotword import greedy_search_both_device
    dev = torch.device("cuda:0")
    torch.manual_seed(12)
    V, D, J, H, L, HW, T = args.V, 256, 512, 256, 2, 100, args.T
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
    for N in (1, args.streams):
        enc = torch.randn(N, T, D, device=dev)
        lens = torch.full((N,), T)
        run = lambda: greedy_search_both_device(m, enc, lens, ctx, ctx_len, n_steps=args.n_steps, filter_on=True)
        hyps, traces = run()
        ms = timeit(run, args.steps)
        ntok = sum(len(h) for h in hyps)
        decisions = ntok + N * T                       # lower bound: go-back re-decodes more
        print(json.dumps({"what": "hotword_greedy device", "streams": N, "T": T, "V": V, "n_ctx": n_ctx, "ms_per_call": round(ms, 2),
                          "tokens": ntok, "gate_zeros": sum(t.count(0) for t in traces), "gate_ones": sum(t.count(1) for t in traces),
                          "us_per_decision": round(ms * 1e3 / (decisions / N), 1), "utt_per_s": round(N / ms * 1e3, 1)}), flush=True)
    os.environ["WR_HOTWORD_HOST"] = "1"
    enc = torch.randn(1, T, D, device=dev)
    run = lambda: w.basic_greedy_search_both(m, enc, torch.tensor(T), ctx, ctx_len, n_steps=args.n_steps, context_filter_state="on",
                                             context_decoder_labels_padded=torch.zeros(1, 4, dtype=torch.long))
    out = run()
    ms = timeit(run, max(1, args.steps // 2))
    print(json.dumps({"what": "hotword_greedy host-driven (round 1)", "streams": 1, "T": T, "ms_per_call": round(ms, 2),
                      "tokens": len(out[0][0]), "us_per_decision": round(ms * 1e3 / (len(out[0][0]) + T), 1)}), flush=True)
    os.environ.pop("WR_HOTWORD_HOST")


def bench_step(args):
    """The loss block of Transducer.forward (transducer.py:131-147) through the autograd path: pre-join
    projections -> joiner -> RNN-T loss -> backward to enc/pred outputs and all joiner weights."""
    import wenet_celoss_amd as w
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, T, U, V, E, P, J = args.B, args.T, args.U, args.V, 256, 256, 512
    enc = torch.randn(B, T, E, device=dev, requires_grad=True)
    pred = torch.randn(B, U + 1, P, device=dev, requires_grad=True)
    y = torch.randint(1, V, (B, U), dtype=torch.int32, device=dev)
    ll = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), U, dtype=torch.int32, device=dev)
    if args.ragged:                                    # utterances sorted by frames (processor.py:704), label counts spread
        gcpu = torch.Generator().manual_seed(5)
        ll = torch.sort(torch.randint(int(0.8 * T), T + 1, (B,), generator=gcpu), descending=True).values.to(torch.int32).to(dev)
        tl = torch.randint(U // 3, U + 1, (B,), generator=gcpu).to(torch.int32).to(dev)
        ll[0], tl[-1] = T, U
    base = None
    precs = ("fp32", "fp32-fused", "bf16x3", "bf16x3-fused", "bf16-autocast")
    if args.only:
        precs = tuple(x for x in precs if x in args.only.split(","))
    for prec in precs:
        amp = prec == "bf16-autocast"                  # the --use_amp configuration (executor.py:91)
        fused = prec.endswith("-fused")                # joiner + loss as one node (fused.py): no pass 1, gradient in place
        jprec = "bf16" if amp else prec.replace("-fused", "")
        joint = w.TransducerJoint(V, E, P, J, precision=jprec).to(dev)
        torch.manual_seed(4)
        with torch.no_grad():
            for prm in joint.parameters():
                prm.copy_(torch.randn_like(prm) * 0.05)

        def step():
            joint.zero_grad(set_to_none=True); enc.grad = None; pred.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                if fused:
                    loss = w.joint_rnnt_loss(joint.enc_ffn(enc), joint.pred_ffn(pred), joint.ffn_out.weight,
                                             joint.ffn_out.bias, y, ll, tl, blank=0, reduction="mean", precision=jprec,
                                             buckets=args.buckets)
                else:
                    logits = joint(enc, pred, ll, tl) if args.ragged else joint(enc, pred)
                    loss = w.rnnt_loss(logits, y, ll, tl, blank=0, reduction="mean", inplace_grad=True)
            loss.backward()
            return loss
        loss = step()
        ms = timeit(step, args.steps)
        rec = {"what": "joint+rnnt_loss fwd+bwd (autograd)", "precision": prec, "shape": [B, T, U + 1, J, V],
               "ragged": bool(args.ragged), "buckets": args.buckets if fused else None,
               "ms": round(ms, 2), "utt_per_s": round(B / ms * 1e3, 1), "loss": float(loss)}
        if base is None:
            base = (float(loss), enc.grad.clone(), joint.ffn_out.weight.grad.clone())
        else:
            rec["loss_rel_diff_vs_fp32"] = abs(float(loss) - base[0]) / abs(base[0])
            for name, got, ref in (("grad_enc", enc.grad, base[1]), ("grad_w", joint.ffn_out.weight.grad, base[2])):
                rms = ref.pow(2).mean().sqrt()
                rec[name + "_max_err_over_rms"] = float((got - ref).abs().max() / rms)
                rec[name + "_rms_err_over_rms"] = float((got - ref).pow(2).mean().sqrt() / rms)
        print(json.dumps(rec), flush=True)
        del joint


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["joint", "ctc", "ctcdec", "greedy", "beam", "step", "hotword"])
    ap.add_argument("--n-steps", type=int, default=64)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--U", type=int, default=150)
    ap.add_argument("--V", type=int, default=5000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dw", action="store_true", help="joint: also time the library fp32 GEMM on the three contractions (yardstick)")
    ap.add_argument("--fwd-only", action="store_true", help="joint: stop after the exact forward sweep")
    ap.add_argument("--ragged", action="store_true", help="step: frames in [0.8 T, T] sorted, labels in [U/3, U]")
    ap.add_argument("--buckets", type=int, default=4, help="step: label-length groups of the fused node (1 = off)")
    ap.add_argument("--only", default="", help="step: comma-separated subset of the configurations")
    ap.add_argument("--streams", type=int, default=64, help="greedy: independent streams decoded together")
    ap.add_argument("--tile", type=int, default=0, help="lane-GEMM tile policy of the decoders (wr_tune_set key 6)")
    a = ap.parse_args()
    if a.tile:
        from wenet_celoss_amd import _lib
        _lib.load().wr_tune_set(6, a.tile)
    {"joint": bench_joint, "ctc": bench_ctc, "greedy": bench_greedy, "beam": bench_beam, "step": bench_step, "ctcdec": bench_ctcdec, "hotword": bench_hotword}[a.what](a)
