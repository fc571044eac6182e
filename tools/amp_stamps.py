"""Phase attribution of the single-term (AMP) joiner forward from in-kernel stamps.

Rebuilds the library with -DWR_JS_STAMPS (a diagnostic build: the shipped kernel executes no stamp), runs the forward at
the B = 8 BASELINE slice for ~2 s so the chip's clock has settled, then reads the per-wave records of every 16th
workgroup: tile build, k-loop (load issue / MFMA sets / epilogues) in cycles and the in-kernel clock
(s_memtime / s_memrealtime x 100 MHz).  Usage: python tools/amp_stamps.py [cells: 64|128] [knob 12 for 64 cells]"""
import sys; sys.path.insert(0, '.')
import os, ctypes, json, time
LEVEL = int(os.environ.get("WR_JS_LEVEL", "1"))     # 2: also drains the vector-memory counter behind each round's stores
os.environ["WR_EXTRA_HIPCC_FLAGS"] = (os.environ.get("WR_EXTRA_HIPCC_FLAGS", "") + f" -DWR_JS_STAMPS={LEVEL}").strip()
import numpy as np, torch
from wenet_celoss_amd import _lib
_lib.build(force=True)
lib = _lib.load(); dev = torch.device('cuda:0')
B, T, U1, J, V = 8, 1000, 151, 512, 5000
g = torch.Generator(device=dev).manual_seed(1)
ep = torch.randn(B, T, J, device=dev, generator=g); pp = torch.randn(B, U1, J, device=dev, generator=g)
w = torch.randn(V, J, device=dev, generator=g) * 0.05; b = torch.randn(V, device=dev, generator=g)
st = _lib.current_stream(dev); P = _lib.ptr
wss = lib.wr_joint_split_workspace_bytes(J, V); ws = torch.empty(wss, dtype=torch.uint8, device=dev)
out = torch.empty(B, T, U1, V, dtype=torch.bfloat16, device=dev)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 64
form = int(sys.argv[2]) if len(sys.argv) > 2 else 0           # knob 12 for 64 cells: 0 two per CU, 1 one per CU
lib.wr_tune_set(12, (3 if form == 3 else 2) if cells == 128 else form)
f = lambda: _lib.check(lib.wr_joint_fwd_split(P(ep), P(pp), P(w), P(b), None, None, B, T, U1, J, V, 0, 1, P(out), 2, P(ws), wss, st))
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(10): f()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
WAVES = 8 if (cells == 128 and form == 3) else 4
WGS, PTS = 2048 * 4 // WAVES, 12
buf = np.zeros(WGS * WAVES * PTS, dtype=np.uint64)
lib.wr_debug_read_js_stamps.restype = ctypes.c_int
lib.wr_debug_read_js_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.wr_debug_read_js_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
r = buf.reshape(WGS, WAVES, PTS).astype(np.int64)
n_wg = (B * T * U1 + cells - 1) // cells
r = r[: min(WGS, (n_wg + 15) // 16)]
r = r[r[:, 0, 2] > 0]
med = lambda x: float(np.median(x))
clk = (r[..., 2] - r[..., 0]) / np.maximum(r[..., 4] - r[..., 3], 1) * 100.0      # MHz
rounds = r[..., 8]
res = {
    "cells": cells, "knob12": 2 if cells == 128 else form, "ms_this_launch": round(ms, 3), "workgroups_sampled": int(r.shape[0]),
    "clock_MHz_median": round(med(clk), 1), "clock_MHz_p10_p90": [round(float(np.percentile(clk, 10)), 1), round(float(np.percentile(clk, 90)), 1)],
    "cycles_per_workgroup": {
        "whole": med(r[..., 2] - r[..., 0]), "tile_build": med(r[..., 1] - r[..., 0]), "k_loop": med(r[..., 2] - r[..., 1]),
        "k_loop.store_drain(level 2)": med(r[..., 5]), "store_drain_per_round": med(r[..., 5] / np.maximum(rounds, 1)), "k_loop.mfma_sets": med(r[..., 6]), "k_loop.epilogues": med(r[..., 7]),
        "rounds": med(rounds), "sets": med(r[..., 10]),
        "epilogue_per_round": med(r[..., 7] / np.maximum(rounds, 1)),
        "mfma_first_set_of_round": med(r[..., 9] / np.maximum(rounds, 1)),
        "mfma_other_sets_each": med((r[..., 6] - r[..., 9]) / np.maximum(r[..., 10] - rounds, 1)),
    },
    "ideal_mfma_cycles_per_set": (4 if cells == 64 else 2) * (cells // 32) * 2 * 32, "waves_per_workgroup": WAVES,
}
res["us_per_workgroup"] = round(res["cycles_per_workgroup"]["whole"] / res["clock_MHz_median"], 2)
print(json.dumps(res, indent=1))
