"""Per-workgroup phase stamps of the LDS-DMA GEMM (tuning aid): python tools/gemm_trace.py M N K [act]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from siu3r_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4]); act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
pw = ops.pack_linear(torch.rand(N, K, device="cuda") * 0.1, torch.zeros(N, device="cuda"), False)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.linear(a, pw, out=out, act=act)
buf = torch.zeros(16384, 8, dtype=torch.int64, device="cuda")
ops.set_gemm_trace(buf)
ops.linear(a, pw, out=out, act=act)
torch.cuda.synchronize()
ops.set_gemm_trace(None)
b = buf.cpu().numpy()
b = b[b[:, 0] != 0]
t0 = b[:, 0].min()
print(f"{len(b)} workgroups")
ph = ["setup", "prologue-issue", "first-tile-wait", "k-loop", "epi: stage", "epi: rows+stores issued", "epi: store drain"]
d = np.diff(b[:, :8], axis=1)
for i, n in enumerate(ph):
    print(f"{n:26s} mean {d[:, i].mean():9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
