import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siu3r_amd import ops
torch.set_printoptions(precision=3, linewidth=200)
B, H, Nq, Nk, D = 1, 1, 128, 64, 64
g = torch.Generator().manual_seed(0)
for qs in (1.0, 3.0, 6.0):
    q = ((torch.rand(B, Nq, H, D, generator=g) * 2 - 1) * qs).cuda().bfloat16()
    k = ((torch.rand(B, Nk, H, D, generator=g) * 2 - 1) * qs).cuda().bfloat16()
    v = torch.zeros(B, Nk, H, D); v[0, :, 0, 0] = 1.0; v[0, :, 0, 1] = torch.arange(Nk).float(); v = v.cuda().bfloat16()
    out = ops.attention(q, k, v, heads=H, head_dim=D, scale=D ** -0.5).float()
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * D ** -0.5
    p = s.softmax(-1)
    print("qs", qs, "col0 (should be 1):", out[0, :6, 0].tolist(), " col1 (expected key mean):", out[0, :4, 1].tolist(), "ref", (p[0, 0, :4] @ torch.arange(Nk).float().cuda()).tolist(), "smax", s.max().item())
