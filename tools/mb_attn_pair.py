"""The pair-shape ViT attention (2 views x 16 heads x 1025 x 1025, head_dim 64), graph-timed: python tools/mb_attn_pair.py [bf16x3|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time
x3 = (sys.argv[1] if len(sys.argv) > 1 else "bf16x3") == "bf16x3"
adt = torch.float32 if x3 else torch.bfloat16
qkv = (torch.rand(2, 1025, 3, 16, 64, device="cuda") * 2 - 1).to(adt)
t = graph_time(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], heads=16, head_dim=64, scale=0.125, split3=x3), n=20)
print(f"{os.environ.get('SIU3R_LIB_OVERRIDE', 'base')[-12:]:>12s} {'bf16x3' if x3 else 'bf16'}: {t*1e6:7.1f} us")
