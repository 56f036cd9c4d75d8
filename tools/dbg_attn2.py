import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siu3r_amd import ops
def ref(q, k, v, scale, mask=None):
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    if mask is not None: s = s.masked_fill(mask[:, None].bool(), float("-inf"))
    return torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v)
g = torch.Generator().manual_seed(0)
B, H, Nq, Nk, D = 1, 1, 128, 64, 32
q, k, v = [(torch.rand(B, n, H, D, generator=g) * 2 - 1).cuda().bfloat16() for n in (Nq, Nk, Nk)]
for pat in ("one", "rand"):
    mb = torch.zeros(B, Nq, Nk, dtype=torch.bool)
    if pat == "one": mb[:, :, 5] = True
    else: mb = torch.rand(B, Nq, Nk, generator=g) < 0.5; mb[:, :, 0] = False
    m = mb.to(torch.uint8).cuda()
    out = ops.attention(q, k, v, heads=H, head_dim=D, scale=D ** -0.5, mask=m).float().view(B, Nq, H, D)
    r = ref(q.float(), k.float(), v.float(), D ** -0.5, m)
    e = (out - r).abs().amax(-1)[0, :, 0]
    print(pat, "per-query err:", [round(x, 3) for x in e[:40].tolist()])
    # which single key would explain: recompute ref with each alternative masked key
    if pat == "one":
        for kk in range(0, 64):
            mb2 = torch.zeros(B, Nq, Nk, dtype=torch.bool); mb2[:, :, kk] = True
            r2 = ref(q.float(), k.float(), v.float(), D ** -0.5, mb2.cuda())
            e2 = (out - r2).abs().max().item()
            if e2 < 0.02: print("  output matches masking key", kk, e2)
