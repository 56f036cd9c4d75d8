import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siu3r_amd import ops
D, Nk, Nq = 64, 64, 128
q = torch.zeros(1, Nq, 1, D); k = torch.zeros(1, Nk, 1, D); v = torch.zeros(1, Nk, 1, D)
for i in range(Nq): q[0, i, 0, i % 64] = 8.0
for j in range(Nk): k[0, j, 0, j] = 8.0
v[0, :, 0, 0] = torch.arange(Nk).float(); v[0, :, 0, 1] = 1.0
out = ops.attention(q.cuda().bfloat16(), k.cuda().bfloat16(), v.cuda().bfloat16(), heads=1, head_dim=D, scale=1.0).float()[0]
print("selected key per query (expect i%64):", out[:, 0].round().int().tolist()[:70])
print("col1 (expect 1):", [round(x, 3) for x in out[:12, 1].tolist()])
