import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siu3r_amd import ops
def ref(q, k, v, scale, mask=None):
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    if mask is not None: s = s.masked_fill(mask[:, None].bool(), float("-inf"))
    return torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v).flatten(2)
g = torch.Generator().manual_seed(0)
for (B, H, Nq, Nk, D, use_mask) in [(1, 1, 128, 64, 64, False), (1, 1, 128, 64, 32, False), (2, 8, 100, 520, 32, False), (2, 8, 100, 520, 32, True), (2, 4, 100, 520, 64, True), (1, 2, 128, 64, 32, True)]:
    q, k, v = [((torch.rand(B, n, H, D, generator=g) * 2 - 1) * s).cuda().bfloat16() for n, s in ((Nq, 6.0), (Nk, 6.0), (Nk, 1.0))]
    m = None
    if use_mask:
        ld = (Nk + 63) // 64 * 64
        mb = torch.rand(B, Nq, Nk, generator=g) < 0.7; mb[:, :, 0] = False
        m = torch.ones(B, Nq, ld, dtype=torch.uint8); m[:, :, :Nk] = mb.to(torch.uint8); m = m.cuda()
    out = ops.attention(q, k, v, heads=H, head_dim=D, scale=D ** -0.5, mask=m).float()
    r = ref(q.float(), k.float(), v.float(), D ** -0.5, None if m is None else m[:, :, :Nk])
    print(os.environ.get("SIU3R_ATTN_NO_FAST", "fast"), (B, H, Nq, Nk, D, use_mask), "max err", (out - r).abs().max().item(), "mean", (out - r).abs().mean().item())
