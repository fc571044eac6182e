"""One AMP training step of the loss block as the reference runs it under --use_amp (executor.py:91 autocast):
pre-join projections -> TransducerJoint(precision="bf16") -> rnnt_loss on 16-bit logits -> backward.
Prints ms per step; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
Usage: python3 tools/amp_step.py [B] [steps] [bf16|fp16]"""
import sys; sys.path.insert(0, '.')
import json, torch
import wenet_celoss_amd as w
from tools.secondary import _median_ms
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
adt = torch.float16 if (len(sys.argv) > 3 and sys.argv[3] == "fp16") else torch.bfloat16
T, U, V, E, Pd, J = 1000, 150, 5000, 256, 256, 512
torch.manual_seed(3)
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
    with torch.autocast("cuda", dtype=adt):
        logits = joint(enc, pred)
        loss = w.rnnt_loss(logits, y, ll, tl, blank=0, reduction="mean")
    loss.backward()
    return loss

ms = _median_ms(step, steps)
print(json.dumps({"what": f"AMP loss-block step ({adt} logits)", "B": B, "T": T, "U": U, "V": V, "J": J, "logits_dtype": str(step().dtype),
                  "ms_per_step": round(ms, 2), "utt_per_s": round(B / ms * 1e3, 2),
                  "max_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}))
