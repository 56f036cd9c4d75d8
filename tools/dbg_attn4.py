import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siu3r_amd import ops
for D in (64, 32):
    Nk = 64
    B, H, Nq = Nk, D, 128
    q = torch.zeros(B, Nq, H, D).cuda().bfloat16()
    k = torch.zeros(B, Nk, H, D).cuda().bfloat16()
    v = torch.zeros(B, Nk, H, D)
    for k0 in range(Nk):
        for d0 in range(D):
            v[k0, k0, d0, d0] = float(Nk)
    out = ops.attention(q, k, v.cuda().bfloat16(), heads=H, head_dim=D, scale=1.0).float().view(B, Nq, H, D)
    o = out[:, 0]  # [k0, d0(h), d]
    exp = torch.eye(D).cuda()[None].expand(B, D, D)
    bad = ((o - exp).abs() > 1e-3)
    print("D", D, "mismatching (k0,d0) pairs:", int(bad.any(-1).sum()), "of", B * H)
    idx = bad.any(-1).nonzero()[:12].tolist()
    for k0, d0 in idx:
        print("   k0", k0, "d0", d0, "-> landed at d", o[k0, d0].nonzero().flatten().tolist(), "vals", o[k0, d0][o[k0, d0] != 0].tolist())
    print("   rows all-wave check (query 77):", int(((out[:, 77] - exp).abs() > 1e-3).any(-1).sum()))
