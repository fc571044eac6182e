"""A/B of the single-term (AMP) split joiner forward at the B = 8 BASELINE slice: 64- against 128-cell workgroups (knob 12)
and 1..8 column parts (knob 7); prints ms / TFLOP/s per setting and checks that the tilings give bit-identical logits."""
import sys; sys.path.insert(0,'.')
import torch, json
from wenet_celoss_amd import _lib
from tools.secondary import _median_ms
lib=_lib.load(); dev=torch.device('cuda:0')
B,T,U1,J,V=8,1000,151,512,5000
g=torch.Generator(device=dev).manual_seed(1)
ep=torch.randn(B,T,J,device=dev,generator=g); pp=torch.randn(B,U1,J,device=dev,generator=g)
w=torch.randn(V,J,device=dev,generator=g)*0.05; b=torch.randn(V,device=dev,generator=g)
st=_lib.current_stream(dev); P=_lib.ptr
wss=lib.wr_joint_split_workspace_bytes(J,V); ws=torch.empty(wss,dtype=torch.uint8,device=dev)
flops=2.0*B*T*U1*J*V
outs={}
for knob,parts in ((0,1),(0,2),(0,4),(0,8),(2,1),(2,2),(2,4),(2,8)):
    lib.wr_tune_set(12,knob); lib.wr_tune_set(7,parts)
    for dt,code in ((torch.bfloat16,2),):
        out=torch.empty(B,T,U1,V,dtype=dt,device=dev)
        f=lambda: _lib.check(lib.wr_joint_fwd_split(P(ep),P(pp),P(w),P(b),None,None,B,T,U1,J,V,0,1,P(out),code,P(ws),wss,st))
        ms=_median_ms(f,5)
        print(json.dumps({"parts":parts,"cells":128 if knob==2 else 64,"logits":str(dt),"ms":round(ms,3),"TFLOPs":round(flops/ms/1e9,1),"frac_2500":round(flops/ms/1e9/2500,4)}))
        key=(str(dt))
        if key in outs: print("  identical to the other tiling:", bool(torch.equal(outs[key], out[:1])))
        outs[key]=out[:1].clone()
        del out
lib.wr_tune_set(12,0); lib.wr_tune_set(7,0)
