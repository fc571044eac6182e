import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wenet_celoss_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, T, U, V = 32, 1000, 150, 5000; U1 = U + 1
logits = torch.empty(B, T, U1, V, device=dev)
for b in range(B): logits[b].normal_()
grads = torch.empty_like(logits)
targets = torch.randint(1, V, (B, U), dtype=torch.int32, device=dev)
ll = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), U, dtype=torch.int32, device=dev)
wsb = lib.wr_rnnt_workspace_bytes(B, T, U1); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
costs = torch.empty(B, device=dev); gc = torch.full((B,), 1.0 / B, device=dev)
st = _lib.current_stream(dev); P = _lib.ptr
fwd = lambda: _lib.check(lib.wr_rnnt_loss_fwd(P(logits), 0, P(targets), P(ll), P(tl), B, T, U1, V, 0, P(costs), P(ws), wsb, st))
bwd = lambda: _lib.check(lib.wr_rnnt_loss_bwd(P(logits), 0, P(targets), P(ll), P(tl), B, T, U1, V, 0, -1.0, P(gc), P(grads), P(ws), wsb, st))
cfgs = {"old (nt 6, un 8, 12/CU)": (6, 8, 12), "new (nt 7, un 16, 16/CU)": (7, 16, 16)}
res = {k: [] for k in cfgs}
for rnd in range(4):
    for name, (nt, un, g) in cfgs.items():
        lib.wr_tune_set(2, nt); lib.wr_tune_set(3, un); lib.wr_tune_set(1, g)
        for _ in range(2): fwd(); bwd()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fwd(); bwd()
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 20 * 1e3)
for k, v in res.items():
    print(json.dumps({"config": k, "ms_per_step_sustained": [round(x, 3) for x in v], "utt_per_s": round(B / (sorted(v)[len(v) // 2] * 1e-3), 1)}))
