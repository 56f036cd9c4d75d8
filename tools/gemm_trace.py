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
print(f"{len(b)} workgroups; kernel span {(b[:,5].max()-t0)} cycles")
ph = ["setup", "prologue-issue", "first-tile-wait", "k-loop", "epilogue"]
d = np.diff(b[:, :6], axis=1)
for i, n in enumerate(ph):
    print(f"{n:16s} mean {d[:, i].mean():9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
print("start time (rel) percentiles:", [int(np.percentile(b[:, 0] - t0, q)) for q in (0, 25, 50, 75, 100)])
print("end   time (rel) percentiles:", [int(np.percentile(b[:, 5] - t0, q)) for q in (0, 25, 50, 75, 100)])
hw = b[:, 6] & 0xffffffff; xcc = b[:, 6] >> 32
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)   # cu_id | se_id | xcc
u, c = np.unique(cu, return_counts=True)
print("distinct CUs", len(u), "workgroups per CU min/max", c.min(), c.max())
order = np.argsort(b[:, 0])
for i in order[:6].tolist() + order[-3:].tolist():
    print("wg", i, "cu", hex(int(cu[i])), "start", int(b[i, 0] - t0), "phases", d[i].tolist())
