"""Timing of the single-term (AMP) joiner forward at the B = 8 BASELINE slice for builds with experiment macros
(WR_EXTRA_HIPCC_FLAGS=-DWR_X_...): prints ms / TFLOP/s for the 64- and 128-cell tiles."""
import sys; sys.path.insert(0, '.')
import os, json, torch
from wenet_celoss_amd import _lib
from tools.secondary import _median_ms
lib = _lib.load(); dev = torch.device('cuda:0')
B, T, U1, J, V = 8, 1000, 151, 512, 5000
g = torch.Generator(device=dev).manual_seed(1)
ep = torch.randn(B, T, J, device=dev, generator=g); pp = torch.randn(B, U1, J, device=dev, generator=g)
w = torch.randn(V, J, device=dev, generator=g) * 0.05; b = torch.randn(V, device=dev, generator=g)
st = _lib.current_stream(dev); P = _lib.ptr
wss = lib.wr_joint_split_workspace_bytes(J, V); ws = torch.empty(wss, dtype=torch.uint8, device=dev)
flops = 2.0 * B * T * U1 * J * V
out = torch.empty(B, T, U1, V, dtype=torch.bfloat16, device=dev)
ref = None
PARTS = int(os.environ.get('WR_PARTS', '0'))
lib.wr_tune_set(7, PARTS)
for knob, store in ((0, 0), (0, 1), (1, 0), (2, 0)):
    lib.wr_tune_set(12, knob); lib.wr_tune_set(13, store)
    out.zero_()
    f = lambda: _lib.check(lib.wr_joint_fwd_split(P(ep), P(pp), P(w), P(b), None, None, B, T, U1, J, V, 0, 1, P(out), 2, P(ws), wss, st))
    ms = _median_ms(f, 5)
    same = None
    if ref is None: ref = out[:1].clone()
    else: same = bool(torch.equal(ref, out[:1]))
    print(json.dumps({"flags": os.environ.get("WR_EXTRA_HIPCC_FLAGS", ""), "parts": PARTS, "form": {0: "64 cells x 2 per CU", 1: "64 cells", 2: "128 cells", 3: "128 cells x 8 waves"}[knob],
                      "stores": "transposed" if store else "staged",
                      "ms": round(ms, 3), "TFLOPs": round(flops / ms / 1e9, 1), "identical_to_first": same}), flush=True)
lib.wr_tune_set(12, 0); lib.wr_tune_set(13, 0)
