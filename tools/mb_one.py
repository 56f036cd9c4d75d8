"""One graph-timed GEMM: python tools/mb_one.py <bf16|bf16x3> M N K [cfg] (cfg: siu3r_gemm_tune key 0).  A/B builds: SIU3R_LIB_OVERRIDE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import _lib, ops
from mb_gemm import graph_time
mode, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
split = mode == "bf16x3"
adt = torch.float32 if split else torch.bfloat16
a = (torch.rand(M, K, device="cuda") * 2 - 1).to(adt)
pw = ops.pack_linear((torch.rand(N, K, device="cuda") * 2 - 1) * 0.1, torch.zeros(N, device="cuda"), split)
out = torch.empty(M, N, device="cuda", dtype=adt)
_lib.check(_lib.lib().siu3r_gemm_tune(0, cfg))
log = []
ops.set_plan_log(log)
ops.linear(a, pw, out=out)
ops.set_plan_log(None)
pl = log[-1]
print(f"plan: tile_cfg={pl.tile_cfg} {pl.bm}x{pl.bn} splitk={pl.splitk} skinny_rows={pl.skinny_rows} {pl.kernel.decode()}")
if len(sys.argv) > 6:
    ops.gemm_tune(int(sys.argv[6]), int(sys.argv[7]))
ts = [graph_time(lambda: ops.linear(a, pw, out=out), n=10) for _ in range(3)]
t = min(ts)
print(f"{os.environ.get('SIU3R_LIB_OVERRIDE', 'base'):>40} {mode} {M}x{N}x{K} cfg={cfg}: {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TF/s (runs {[round(x*1e6,1) for x in ts]})")
