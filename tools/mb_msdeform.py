"""MS-deformable attention sampling (csrc/elementwise.hip) on the network's two shapes at 2 x 512^2: the ViT-Adapter extractor (16 heads x 64,
one 32 x 32 level, 2 x 5376 queries) and the Mask2Former pixel decoder (8 heads x 32, levels 64^2 / 32^2 / 16^2, queries = all level
positions), fp32 values (bf16x3 mode).  python tools/mb_msdeform.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time

g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).cuda()
for name, B, shapes, heads, d, Q in (("adapter extractor", 2, [(32, 32)], 16, 64, 5376), ("pixel decoder", 2, [(64, 64), (32, 32), (16, 16)], 8, 32, 5376)):
    S = sum(h * w for h, w in shapes)
    L = len(shapes)
    value = rnd(B, S, heads * d)
    offs_aw = torch.cat((rnd(B, Q, heads * L * 4 * 2) * 3.0, rnd(B, Q, heads * L * 4)), -1).contiguous()
    ref = torch.rand(Q, L, 2, generator=g).cuda()
    out = ops.msdeform_sample(value, offs_aw, ref, shapes, heads, 4, torch.float32)
    t = min(graph_time(lambda: ops.msdeform_sample(value, offs_aw, ref, shapes, heads, 4, torch.float32), n=10) for _ in range(3))
    gather = B * Q * heads * L * 4 * 4 * d * 4
    print(f"{name:18s} B={B} Q={Q} heads={heads} d={d} L={L}: {t * 1e6:7.1f} us   gathered {gather / 1e6:.0f} MB -> {gather / t / 1e12:.2f} TB/s of L1/L2 traffic; checksum {float(out.double().sum()):.6f}")
