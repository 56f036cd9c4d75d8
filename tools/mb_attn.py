"""Graph-timed attention microbench."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time
def bench(B, H, Nq, Nk, D, adt=torch.bfloat16, split=False, mask=False):
    q = (torch.rand(B, Nq, H, D, device="cuda") * 2 - 1).to(adt)
    k = (torch.rand(B, Nk, H, D, device="cuda") * 2 - 1).to(adt)
    v = (torch.rand(B, Nk, H, D, device="cuda") * 2 - 1).to(adt)
    m = None
    if mask:
        ld = (Nk + 63) // 64 * 64
        m = (torch.rand(B, Nq, ld, device="cuda") > 0.5).to(torch.uint8)
        t = graph_time(lambda: ops.attention(q, k, v, heads=H, head_dim=D, scale=D ** -0.5, split3=split, mask=m))
    else:
        t = graph_time(lambda: ops.attention(q, k, v, heads=H, head_dim=D, scale=D ** -0.5, split3=split))
    print(f"attn B={B} H={H} Nq={Nq} Nk={Nk} D={D} {str(adt)[6:]} split={split} mask={mask}: {t*1e6:8.1f} us {4.0*B*H*Nq*Nk*D/t/1e12:7.1f} TF/s")
if __name__ == "__main__":
    bench(2, 16, 1025, 1025, 64)
    bench(1, 12, 1025, 1025, 64)
    bench(8, 16, 1025, 1025, 64)
    bench(2, 16, 1025, 1025, 64, torch.float32, True)
    bench(8, 16, 1025, 1025, 64, torch.float32, True)
    bench(2, 16, 1025, 2050, 64, torch.float32, True)
    bench(8, 1, 100, 8192, 32, mask=True)
    bench(8, 1, 100, 2048, 32, mask=True)
    bench(2, 12, 1025, 1025, 64, torch.float32, True)
    bench(1, 8, 100, 8192, 32, torch.float32, True, mask=True)
    bench(1, 8, 100, 2048, 32, torch.float32, True, mask=True)
