"""The Gaussian head's last layer (head.4: 256 -> 83 over the two 512^2 maps of a pair, bf16x3) as the network runs it, against the same
product with the output padded to 96 / 128 columns (aligned rows; 128: the fast row pass) -- what the ragged 83-column row pass costs.
python tools/mb_head4.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time

g = torch.Generator().manual_seed(0)
M, K = 512 * 512, 256
x = (torch.rand(1, 2, M, K, generator=g) * 2 - 1).cuda()
W = [(torch.rand(83, K, generator=g) * 0.1 - 0.05).cuda() for _ in range(2)]
b = [torch.rand(83, generator=g).cuda() for _ in range(2)]
pw83 = ops.stack_packed([ops.pack_linear(w_, b_, True) for w_, b_ in zip(W, b)])
for ld in (83, 84, 96, 128):  # the network's 83 columns in rows of ld floats (a view of a padded buffer)
    buf = torch.empty(1, 2, M, ld, device="cuda")
    out = buf[..., :83]
    for cfg in (0, -1, 2, 3):
        ops.gemm_tune(0, cfg)
        ts = [graph_time(lambda: ops.linear_grouped(x, pw83, out=out), n=5) for _ in range(3)]
        print(f"N =  83 in rows of {ld:3d} floats          tile_cfg {cfg:2d}: {min(ts) * 1e6:7.1f} us   {(x.numel() + out.numel()) * 4 / min(ts) / 1e12:5.2f} TB/s")
    ops.gemm_tune(0, 0)
for npad in (83, 96, 128):
    pws = []
    for w_, b_ in zip(W, b):
        wp = torch.zeros(npad, K, device="cuda"); wp[:83] = w_
        bp = torch.zeros(npad, device="cuda"); bp[:83] = b_
        pws.append(ops.pack_linear(wp, bp, True))
    pw = ops.stack_packed(pws)
    out = torch.empty(1, 2, M, npad, device="cuda")
    log = []
    ops.set_plan_log(log)
    ops.linear_grouped(x, pw, out=out)
    ops.set_plan_log(None)
    for cfg in (0, -1, 2, 3):
        ops.gemm_tune(0, cfg)
        ts = [graph_time(lambda: ops.linear_grouped(x, pw, out=out), n=5) for _ in range(3)]
        print(f"N = {npad:3d} (row stride {npad * 4} B) tile_cfg {cfg:2d}: {min(ts) * 1e6:7.1f} us   {(x.numel() + out.numel()) * 4 / min(ts) / 1e12:5.2f} TB/s   [{log[-1].kernel.decode()[:60] if cfg == 0 else ''}]")
    ops.gemm_tune(0, 0)

# the dedicated row-stream kernel (csrc/proj.hip)
wf, bias, n = ops.pack_proj(W, b)
out = torch.empty(1, 2, M, 83, device="cuda")
ts = [graph_time(lambda: ops.proj_rows_x3(x, wf, bias, n, out), n=5) for _ in range(3)]
print(f"siu3r_proj_rows_x3 N = 83, K = 256: {min(ts) * 1e6:7.1f} us   {(x.numel() + out.numel()) * 4 / min(ts) / 1e12:5.2f} TB/s")
x2 = (torch.rand(1, 2, M, 128, generator=g) * 2 - 1).cuda()
W2 = [(torch.rand(3, 128, generator=g) * 0.1 - 0.05).cuda() for _ in range(2)]
b2 = [torch.rand(3, generator=g).cuda() for _ in range(2)]
wf2, bias2, n2 = ops.pack_proj(W2, b2)
out2 = torch.empty(1, 2, M, 3, device="cuda")
ts = [graph_time(lambda: ops.proj_rows_x3(x2, wf2, bias2, n2, out2), n=5) for _ in range(3)]
print(f"siu3r_proj_rows_x3 N = 3, K = 128: {min(ts) * 1e6:7.1f} us   {(x2.numel() + out2.numel()) * 4 / min(ts) / 1e12:5.2f} TB/s")
pw3 = ops.stack_packed([ops.pack_linear(w_, b_, True) for w_, b_ in zip(W2, b2)])
ops.gemm_tune(0, 0)
ts = [graph_time(lambda: ops.linear_grouped(x2, pw3, out=out2), n=5) for _ in range(3)]
print(f"grouped GEMM        N = 3, K = 128: {min(ts) * 1e6:7.1f} us")
