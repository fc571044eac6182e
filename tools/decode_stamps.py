#!/usr/bin/env python3
"""Where do the microseconds of a decode micro-step go?  (attribute before cutting.)

Builds the product sources a second time with -DWR_STAMPS into tools/micro/bin/libwr_stamps<level>.so -- the same
kernels, plus per-wave `s_memtime` / `s_memrealtime` stamps written to a device array (csrc/decode.hip, WR_STAMP_*; the
product library contains none of it) -- loads that library in place of the product one, runs the greedy search of
BASELINE config 3 (64 streams) and the prefix beam search of config 5 under their normal hipGraph replay, and prints per
kernel of the LAST micro-step executed:

  wgs / waves      workgroups and waves that reported
  span_us          last wave's end - first wave's start (100 MHz s_memrealtime, 10 ns resolution)
  start_skew_us    last workgroup's start - first workgroup's start
  gap_us           this kernel's first start - previous kernel's last end (the dependent-launch boundary)
  phases           median over waves of the s_memtime differences between stamp points, in us at the measured clock:
                   lane_gemm: issue (entry -> all loads issued), first (-> first chunk's operands arrived, its MFMAs
                   issued; level 2: -> every load drained), mfma (-> accumulators read, partials to LDS), barrier
                   (-> workgroup barrier passed), epilogue (-> stores issued), drain (-> stores acknowledged)

    python tools/decode_stamps.py [--level 1|2] [--what greedy,beam] [--streams 64]
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "micro", "bin")
SLOTS, WGS, WAVES, PTS = 12, 1024, 8, 10
NAMES = {0: "lstm_l0 (lane_gemm LstmCell)", 1: "lstm_l1 (lane_gemm LstmCell)", 2: "pred_ffn (lane_gemm JointAct)",
         3: "ffn_out (lane_gemm RowStats/RowMajor)", 4: "greedy_update", 5: "greedy_update<HW>", 6: "beam_topk",
         7: "beam_update", 8: "projection (lane_gemm KMajor)"}
GEMM_PHASES = [("kernarg", 0, 9), ("issue", 0, 1), ("first", 1, 2), ("mfma", 2, 3), ("barrier", 3, 4), ("epilogue", 4, 5), ("drain", 5, 6)]
UPD_PHASES = [("loads+max", 0, 1), ("decide", 1, 2), ("state+copies", 2, 5), ("drain", 5, 6)]
TOPK_PHASES = [("loads+max", 0, 1), ("sum", 1, 2), ("select", 2, 3), ("barrier", 3, 4), ("merge", 4, 5), ("drain", 5, 6)]
BUPD_PHASES = [("candidates", 0, 1), ("classes", 1, 2), ("fuse", 2, 9), ("rank+prune", 9, 3), ("hyps", 3, 4), ("state (slots + layer 0 | inputs + caches)", 4, 5), ("drain", 5, 6)]


def build(level: int) -> str:
    from wenet_celoss_amd import _lib
    os.makedirs(BIN, exist_ok=True)
    out = os.path.join(BIN, f"libwr_stamps{level}.so")
    h = hashlib.sha1((_lib._source_hash() + str(level)).encode()).hexdigest()
    tag = out + ".srchash"
    if os.path.exists(out) and os.path.exists(tag) and open(tag).read() == h:
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + _lib._INCLUDE, "-I" + _lib._CSRC,
             f"-DWR_STAMPS={level}"]
    objs, procs = [], []
    for src in _lib._sources():
        obj = os.path.join(BIN, f"s{level}_" + os.path.basename(src) + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([hipcc] + flags + ["-c", src, "-o", obj], stdout=subprocess.PIPE,
                                            stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        o, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{o}")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout + r.stderr)
    with open(tag, "w") as f:
        f.write(h)
    return out


def load_stamped(level: int):
    from wenet_celoss_amd import _lib
    lib = ctypes.CDLL(build(level))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.wr_debug_read_stamps.restype = ctypes.c_int
    lib.wr_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    _lib._lib = lib                      # every module of the package now calls into the stamped build
    return lib


def read(lib):
    import numpy as np
    buf = np.zeros(SLOTS * WGS * WAVES * PTS, dtype=np.uint64)
    rc = lib.wr_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    assert rc == 0, rc
    return buf.reshape(SLOTS, WGS, WAVES, PTS).astype(np.int64)


def analyse(st, order, phases_of):
    import numpy as np
    out = []
    prev_end = None
    for slot in order:
        a = st[slot]
        on = a[:, :, 7] > 0                                  # waves that reported (realtime at entry)
        if not on.any():
            continue
        w = a[on]                                            # [n_waves, PTS]
        rt0, rt1 = w[:, 7], w[:, 8]
        cyc = (w[:, 6] - w[:, 0]).astype(float)
        us = (rt1 - rt0).astype(float) / 100.0              # 100 MHz
        clock_mhz = float(np.median(cyc[us > 0.5] / us[us > 0.5])) if (us > 0.5).any() else float("nan")
        wg_start = np.array([a[g, on[g], 7].min() for g in range(WGS) if on[g].any()])
        rec = {"kernel": NAMES.get(slot, str(slot)), "wgs": int(on.any(axis=1).sum()), "waves": int(on.sum()),
               "span_us": round((rt1.max() - rt0.min()) / 100.0, 2),
               "start_skew_us": round((wg_start.max() - wg_start.min()) / 100.0, 2),
               "median_wave_us": round(float(np.median(us)), 2), "clock_MHz": round(clock_mhz)}
        if prev_end is not None:
            rec["gap_us"] = round((rt0.min() - prev_end) / 100.0, 2)
        prev_end = rt1.max()
        ph = {}
        for name, p, q in phases_of(slot):
            dlt = (w[:, q] - w[:, p]).astype(float)
            ok = (w[:, q] > 0) & (w[:, p] > 0)
            if ok.any() and clock_mhz == clock_mhz:
                ph[name] = [round(float(np.median(dlt[ok])) / clock_mhz, 2), round(float(np.max(dlt[ok])) / clock_mhz, 2)]
        rec["phases_us[median,max]"] = ph
        out.append(rec)
    return out


def phases_of(slot):
    return {4: UPD_PHASES, 5: UPD_PHASES, 6: TOPK_PHASES, 7: BUPD_PHASES}.get(slot, GEMM_PHASES)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--what", default="greedy,beam")
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--build-only", action="store_true")
    a = ap.parse_args()
    if a.build_only:
        print(build(1)); print(build(2))
        return
    import types
    import torch
    lib = load_stamped(a.level)
    import wenet_celoss_amd as w
    from tools.secondary import _decode_modules
    dev = torch.device("cuda:0")
    if "greedy" in a.what:
        pred, joint = _decode_modules(dev, 5)
        N, T = a.streams, 64
        model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint)
        enc = torch.randn(N, T, 256, device=dev)
        lens = torch.full((N,), T)
        for _ in range(3):
            hyps = w.basic_greedy_search(model, enc, lens, n_steps=64)
        st = read(lib)
        print(json.dumps({"what": f"greedy config 3, {N} streams, stamps level {a.level}", "tokens": sum(len(h) for h in hyps)}))
        for rec in analyse(st, [0, 1, 2, 3, 4], phases_of):
            print(json.dumps(rec))
    if "beam" in a.what:
        pred, joint = _decode_modules(dev, 6)
        ctc = w.CTC(5000, 256).to(dev).eval()
        bs = w.PrefixBeamSearch(None, pred, joint, ctc, 0)
        B, T = 16, 300
        enc = torch.randn(B, T, 256, device=dev)
        lens = torch.full((B,), T, dtype=torch.int32)
        for _ in range(2):
            out = bs.search_encoded(enc, lens, beam_size=8)
        st = read(lib)
        print(json.dumps({"what": f"prefix beam search config 5 (T cut to {T}), B={B}, beam 8, stamps level {a.level}"}))
        for rec in analyse(st, [0, 1, 2, 3, 6, 7], phases_of):
            print(json.dumps(rec))


if __name__ == "__main__":
    main()
