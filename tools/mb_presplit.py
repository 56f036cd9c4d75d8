"""The four GEMMs of an encoder block (M = 2 x 1025 tokens, bf16x3) graph-timed with fp32 and with pre-split operands / outputs
(ops.Planes): where the planes pay and what writing them costs.  python tools/mb_presplit.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2050
C0, C1 = 1024, 4096
dev = "cuda"
g = torch.Generator().manual_seed(1)
rnd = lambda *s, sc=1.0: ((torch.rand(*s, generator=g) * 2 - 1) * sc).to(dev)
x = rnd(M, C0); st = ops.RowStats(x); xp = ops.Planes(x)
a = rnd(M, C0); r = rnd(M, C0)
w_proj = ops.pack_linear(rnd(C0, C0, sc=0.05), rnd(C0), True)
ops.linear(a, w_proj, residual=r, out=x, stats_out=st, planes_out=xp)
assert xp.valid, "the producer's plan cannot write planes"
gam, bet = 1 + 0.1 * rnd(C0), rnd(C0)
w_qkv = ops.pack_linear_ln(rnd(3 * C0, C0, sc=0.05), rnd(3 * C0), gam, bet, True); w_qkv.meta["ln_eps"] = 1e-6
w_fc1 = ops.pack_linear_ln(rnd(C1, C0, sc=0.05), rnd(C1), gam, bet, True); w_fc1.meta["ln_eps"] = 1e-6
w_fc2 = ops.pack_linear(rnd(C0, C1, sc=0.05), rnd(C0), True)
qkv = torch.empty(M, 3 * C0, device=dev); h = torch.empty(M, C1, device=dev); hp = ops.Planes(h, storage=torch.empty_like(h))
x2 = torch.empty(M, C0, device=dev); s2 = ops.RowStats(x2); xp2 = ops.Planes(x2)
ops.linear(x, w_fc1, act=ops.ACT_GELU, ln=st, out=h, planes_out=hp)

def t(name, fn):
    ts = [graph_time(fn, n=10) for _ in range(3)]
    print(f"{name:64s} {min(ts) * 1e6:7.1f} us   (runs {[round(v * 1e6, 1) for v in ts]})")

t("qkv   LN-folded            A fp32", lambda: ops.linear(x, w_qkv, ln=st, out=qkv))
t("qkv   LN-folded            A planes", lambda: ops.linear(x, w_qkv, ln=st, out=qkv, a_planes=xp))
t("proj  residual + stats     C fp32", lambda: ops.linear(a, w_proj, residual=r, out=x2, stats_out=s2))
t("proj  residual + stats     C fp32 + planes", lambda: ops.linear(a, w_proj, residual=r, out=x2, stats_out=s2, planes_out=xp2))
t("fc1   LN-folded + GELU     A fp32,   C fp32", lambda: ops.linear(x, w_fc1, act=ops.ACT_GELU, ln=st, out=h))
t("fc1   LN-folded + GELU     A planes, C fp32", lambda: ops.linear(x, w_fc1, act=ops.ACT_GELU, ln=st, out=h, a_planes=xp))
t("fc1   LN-folded + GELU     A planes, C planes only", lambda: ops.linear(x, w_fc1, act=ops.ACT_GELU, ln=st, out=h, a_planes=xp, planes_out=hp, planes_only=True))
ops.linear(x, w_fc1, act=ops.ACT_GELU, ln=st, out=h, planes_out=hp)  # (h fp32 and its planes both valid again)
t("fc2   residual + stats     A fp32,   C fp32", lambda: ops.linear(h, w_fc2, residual=x, out=x2, stats_out=s2))
t("fc2   residual + stats     A planes, C fp32", lambda: ops.linear(h, w_fc2, residual=x, out=x2, stats_out=s2, a_planes=hp))
t("fc2   residual + stats     A planes, C fp32 + planes", lambda: ops.linear(h, w_fc2, residual=x, out=x2, stats_out=s2, a_planes=hp, planes_out=xp2))
