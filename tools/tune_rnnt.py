#!/usr/bin/env python3
"""Interleaved A/B of the RNN-T streaming kernels' launch knobs in ONE process (rule: never compare
timings across processes).  Prints median ms of fwd (lse+sweep) and bwd (grad) per variant.
SWEEP=grad | r2 | grid pick the variant lists of rounds 1 / 2; STEPS=n (default 1) times n back-to-back fwd + bwd steps per
measurement -- sustained operation runs the parts ~5 % slower than isolated launches, and the grid-size effects of round 2
only show there."""
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wenet_celoss_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B, T, U, V = int(os.environ.get("B", 8)), 1000, 150, 5000
U1 = U + 1
logits = torch.empty(B, T, U1, V, device=dev)
for b in range(B):
    logits[b].normal_()
grads = torch.empty_like(logits)
targets = torch.randint(1, V, (B, U), dtype=torch.int32, device=dev)
ll = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), U, dtype=torch.int32, device=dev)
wsb = lib.wr_rnnt_workspace_bytes(B, T, U1); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
costs = torch.empty(B, device=dev); gc = torch.full((B,), 1.0 / B, device=dev)
st = _lib.current_stream(dev); P = _lib.ptr
fwd = lambda: _lib.check(lib.wr_rnnt_loss_fwd(P(logits), 0, P(targets), P(ll), P(tl), B, T, U1, V, 0, P(costs), P(ws), wsb, st))
bwd = lambda: _lib.check(lib.wr_rnnt_loss_bwd(P(logits), 0, P(targets), P(ll), P(tl), B, T, U1, V, 0, -1.0, P(gc), P(grads), P(ws), wsb, st))

variants = [dict(lse=l, grad=g, nt=6, un=u, lun=lu) for u in (4, 8) for lu in (4, 8) for l, g in ((8, 8), (6, 6), (12, 12))]
if os.environ.get("SWEEP") == "grad":          # gradient pass only: grid size x non-temporal bits (1: NT loads, 2: NT stores)
    variants = [dict(lse=12, grad=g, nt=4 | nt, un=8, lun=8) for g in (6, 8, 12, 16, 24) for nt in (2, 3, 0, 1)]
if os.environ.get("SWEEP") == "r2":            # round 2: non-temporal loads x vectors in flight x grid, full fwd + bwd sequence
    variants = [dict(lse=12, grad=g, nt=nt, un=u, lun=8) for nt in (6, 7) for u in (8, 16) for g in (12, 16)]
if os.environ.get("SWEEP") == "grid":          # round 2: workgroups per CU (0 = automatic: short-lived workgroups, see stream_grid)
    variants = [dict(lse=l, grad=g, nt=7, un=16, lun=16) for l, g in ((12, 16), (0, 0), (128, 128), (1024, 1024), (4800, 1024),
                                                                       (4800, 2048), (2360, 512))]
if os.environ.get("SWEEP") == "grid2":         # gradient-pass grid with the row pass on its automatic grid
    variants = [dict(lse=0, grad=g, nt=7, un=16, lun=16) for g in (16, 64, 128, 256, 512, 768, 1024, 1400, 0)]
if os.environ.get("SWEEP") == "auto":          # automatic grids: non-temporal bits and vectors in flight once more
    variants = [dict(lse=0, grad=0, nt=nt, un=u, lun=lu) for nt in (7, 3, 5, 6) for u, lu in ((16, 16), (8, 8), (16, 8), (8, 16))]
STEPS = int(os.environ.get("STEPS", 1))
res = {i: ([], []) for i in range(len(variants))}
for rnd in range(int(os.environ.get("ROUNDS", 5))):
    for i, v in enumerate(variants):
        lib.wr_tune_set(0, v["lse"]); lib.wr_tune_set(1, v["grad"]); lib.wr_tune_set(2, v["nt"]); lib.wr_tune_set(3, v["un"]); lib.wr_tune_set(4, v["lun"])
        fwd(); bwd(); torch.cuda.synchronize()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(STEPS)]
        for e in ev:
            e[0].record(); fwd(); e[1].record(); bwd(); e[2].record()
        torch.cuda.synchronize()
        res[i][0].append(sum(e[0].elapsed_time(e[1]) for e in ev) / STEPS)
        res[i][1].append(sum(e[1].elapsed_time(e[2]) for e in ev) / STEPS)
med = lambda a: sorted(a)[len(a) // 2]
# same-process, same-device calibration: a plain 1:1 device copy of the same tensors (read + write)
cp = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); grads.copy_(logits); e1.record(); torch.cuda.synchronize()
    cp.append(e0.elapsed_time(e1))
print(json.dumps({"calibration": "torch copy_ (read+write)", "ms": round(med(cp), 3),
                  "GBs": round(8.0 * V * B * T * U1 / med(cp) / 1e6)}))
for i, v in enumerate(variants):
    f, b = med(res[i][0]), med(res[i][1])
    print(json.dumps(dict(v, fwd_ms=round(f, 3), bwd_ms=round(b, 3), fwd_GBs=round(4.0 * V * B * T * U1 / f / 1e6),
                          bwd_GBs=round(8.0 * V * B * T * U1 / b / 1e6))))
