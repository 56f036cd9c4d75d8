"""Batched (blockIdx.z) GEMM timing: python tools/mb_batch.py Z M N K  (bf16x3; x is a non-collapsible [Z, M+1, K][:, :-1] view)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time
Z, M, N, K = (int(v) for v in sys.argv[1:5])
x = (torch.rand(Z, M + 1, K, device="cuda") * 2 - 1)[:, :-1]
pw = ops.pack_linear((torch.rand(N, K, device="cuda") * 2 - 1) * 0.1, torch.zeros(N, device="cuda"), True)
out = torch.empty(Z, M, N, device="cuda")
for cfg in (-1, 1, 2, 3):
    ops.gemm_tune(0, cfg)
    log = []
    ops.set_plan_log(log)
    ops.linear(x, pw, out=out)
    ops.set_plan_log(None)
    t = min(graph_time(lambda: ops.linear(x, pw, out=out), n=10) for _ in range(3))
    pl = log[-1]
    print(f"Z={Z} {M}x{N}x{K} cfg={cfg} (plan {pl.tile_cfg} S={pl.splitk} sk={pl.skinny_rows}): {t*1e6:7.1f} us {2.0*Z*M*N*K/t/1e12:6.1f} TF/s")
