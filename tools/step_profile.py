#!/usr/bin/env python
"""Per-shape GEMM table of one instrumented step (HIP events around each siu3r_gemm launch)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siu3r_amd import ops
from siu3r_amd.model import SIU3RModel
from siu3r_amd import synthetic_weights as OW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda", 0)
model = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision=prec, device=dev)
images = torch.rand(B, 2, 3, 512, 512).to(dev)
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, 2, 1, 1).to(dev)
for _ in range(3):
    with torch.no_grad():
        model(images, K, enable_query_class_logit_lift=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    with torch.no_grad():
        model(images, K, enable_query_class_logit_lift=True)
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
tm = ops.KernelTimer(); ops.set_kernel_timer(tm)
with torch.no_grad():
    model(images, K, enable_query_class_logit_lift=True)
ops.set_kernel_timer(None)
rows = sorted(tm.by_shape().items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print(f"gemm total {tot:.2f} ms in {sum(v['launches'] for _, v in rows)} launches")
print("     M      N      K   Z am om kh cfg  S sk ln rp rs |  n   ms_tot  us_each  TFLOP/s")
import json
out = []
for k, v in rows[:60]:
    var, m, n, kk, z, am, om, kh, cfg, S, sk, ln, rp, rs = k
    print(f"{m:6d} {n:6d} {kk:6d} {z:3d} {am:2d} {om:2d} {kh:2d} {cfg:3d} {S:2d} {sk:2d} {ln:2d} {rp:2d} {rs:2d} | {v['launches']:3d} {v['ms']:7.3f} {v['ms']/v['launches']*1e3:8.1f} {v['flops']/v['ms']/1e9:8.1f}")
    out.append(dict(m=m, n=n, k=kk, z=z, a_mode=am, out_mode=om, kh=kh, cfg=cfg, splitk=S, skinny=sk, ln=ln, rope=rp, res=rs, launches=v['launches'], ms=v['ms']))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=0)
