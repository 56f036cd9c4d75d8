"""Remainder-row (skinny) launches: python tools/mb_skinny.py -> per network shape, the tiled launch with / without the skinny remainder launch
(siu3r_gemm_tune key 1 = no skinny, key 4 = force) and the skinny kernels alone (SIU3R_GEMM_NO_GEMV=1 selects the MFMA one in a second process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time

for (M, N, K, name) in ((2050, 4096, 1024, "enc fc1"), (2050, 1024, 4096, "enc fc2"), (2050, 3072, 1024, "enc qkv"), (2050, 1024, 1024, "enc proj"),
                        (1025, 3072, 768, "dec fc1 (one side)"), (1025, 768, 3072, "dec fc2 (one side)")):
    a = torch.rand(M, K, device="cuda") * 2 - 1
    pw = ops.pack_linear((torch.rand(N, K, device="cuda") * 2 - 1) * 0.1, torch.zeros(N, device="cuda"), True)
    out = torch.empty(M, N, device="cuda")
    res = {}
    for label, keys in (("auto", ()), ("no skinny", ((1, 1),)), ("forced skinny", ((4, 1),))):
        for k, v in keys:
            ops.gemm_tune(k, v)
        log = []
        ops.set_plan_log(log)
        ops.linear(a, pw, out=out)
        ops.set_plan_log(None)
        pl = log[-1]
        t = min(graph_time(lambda: ops.linear(a, pw, out=out), n=10) for _ in range(3))
        res[label] = f"{t*1e6:6.1f} us (cfg {pl.tile_cfg} S={pl.splitk} sk={pl.skinny_rows})"
        for k, v in keys:
            ops.gemm_tune(k, 0)
    # the remainder launch alone: a problem that is only remainder rows
    rows = M % 256
    a2 = a[:rows].contiguous()
    out2 = torch.empty(rows, N, device="cuda")
    ops.gemm_tune(4, 1)
    ops.gemm_tune(0, 2)
    t2 = min(graph_time(lambda: ops.linear(a2, pw, out=out2), n=10) for _ in range(3))
    ops.gemm_tune(4, 0)
    ops.gemm_tune(0, 0)
    print(f"{name:20s} {M}x{N}x{K}: " + " | ".join(f"{k}: {v}" for k, v in res.items()) + f" | {rows} rows alone: {t2*1e6:5.1f} us")
