import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from siu3r_amd import ops
from mb_gemm import graph_time
for mode in ("bf16", "bf16x3"):
    split = mode == "bf16x3"; adt = torch.float32 if split else torch.bfloat16
    for (M, N, K) in [(2048, 4096, 1024), (2050, 4096, 1024), (2176, 4096, 1024), (2048, 1024, 4096), (2050, 1024, 4096), (2048, 3072, 1024), (2050, 3072, 1024), (2050, 1024, 1024)]:
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(adt)
        pw = ops.pack_linear(torch.rand(N, K, device="cuda") * 0.1, torch.zeros(N, device="cuda"), split)
        out = torch.empty(M, N, device="cuda", dtype=adt)
        t = graph_time(lambda: ops.linear(a, pw, out=out))
        print(f"{mode} {M}x{N}x{K}: {t*1e6:6.1f} us {2.0*M*N*K/t/1e12:6.1f} TF/s")
