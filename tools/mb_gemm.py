"""Graph-timed GEMM microbench: python tools/mb_gemm.py M N K [act]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd import ops

def graph_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e-3

if __name__ == "__main__":
  shapes = [(2050, 4096, 1024, 1), (2050, 3072, 1024, 0), (2050, 1024, 4096, 0), (2050, 1024, 1024, 0), (1025, 768, 768, 0), (1025, 3072, 768, 1), (8192, 8192, 1024, 0)]
  for (M, N, K, act) in shapes:
      a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
      pw = ops.pack_linear(torch.rand(N, K, device="cuda") * 0.1, torch.zeros(N, device="cuda"), False)
      out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
      t = graph_time(lambda: ops.linear(a, pw, out=out, act=act))
      print(f"M={M} N={N} K={K} act={act}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:7.1f} TF/s")
