"""Per-segment shader-cycle sums of the ping-pong GEMM (a -DSIU3R_PP_DBG=64 build selected with SIU3R_LIB_OVERRIDE):
python tools/pp_trace.py <bf16|bf16x3> M N K [cfg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from siu3r_amd import _lib, ops
mode, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = int(sys.argv[5]) if len(sys.argv) > 5 else 1
split = mode == "bf16x3"
adt = torch.float32 if split else torch.bfloat16
a = (torch.rand(M, K, device="cuda") * 2 - 1).to(adt)
pw = ops.pack_linear((torch.rand(N, K, device="cuda") * 2 - 1) * 0.1, torch.zeros(N, device="cuda"), split)
out = torch.empty(M, N, device="cuda", dtype=adt)
_lib.check(_lib.lib().siu3r_gemm_tune(0, cfg))
for _ in range(3): ops.linear(a, pw, out=out)
buf = torch.zeros(4096, 2, 8, dtype=torch.int64, device="cuda")
ops.set_gemm_trace(buf)
ops.linear(a, pw, out=out)
torch.cuda.synchronize()
ops.set_gemm_trace(None)
b = buf.cpu().numpy()
b = b[b[:, 0, :].sum(1) != 0]
steps = K // (16 if split else 32)
names = ["-", "reads+dma issued, reads back", "(wait)", "split/convert", "vm wait + barrier A", "mfma issue", "vm wait + barrier B", "loop overhead"]
print(f"{mode} {M}x{N}x{K}: {len(b)} workgroups, {steps} steps; cycles per step (mean over workgroups)")
for g in range(2):
    tot = b[:, g, :].sum(1).mean() / steps
    print(f" group {g}: " + "  ".join(f"{n}={b[:, g, i].mean() / steps:6.0f}" for i, n in enumerate(names)) + f"  | total {tot:6.0f}")
