"""Kernel micro-benchmarks on the GPU box (not part of the product): TF/s of the GEMM / attention kernels
on SIU3R shapes @512^2.  Usage: python tools/microbench.py > gpurun_out/microbench.txt"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm(M, N, K, adt, split, act=0):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(adt)
    w = torch.rand(N, K, device="cuda") * 0.1
    pw = ops.pack_linear(w, torch.zeros(N, device="cuda"), split)
    t = timeit(lambda: ops.linear(a, pw, out_dtype=adt, act=act))
    fl = 2.0 * M * N * K
    print(f"gemm M={M} N={N} K={K} a={adt} split={split}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s (algorithmic)")


def bench_conv(B, H, W, cin, cout, k, adt, split):
    x = (torch.rand(B, H, W, cin, device="cuda") * 2 - 1).to(adt)
    w = torch.rand(cout, cin, k, k, device="cuda") * 0.1
    pw = ops.pack_conv(w, None, split)
    t = timeit(lambda: ops.conv2d(x, pw, stride=1, pad=k // 2, out_dtype=adt), iters=10)
    fl = 2.0 * B * H * W * cout * cin * k * k
    print(f"conv {B}x{H}x{W} {cin}->{cout} k{k} a={adt} split={split}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")


def bench_attn(B, H, Nq, Nk, D, adt, split):
    q = (torch.rand(B, Nq, H, D, device="cuda") * 2 - 1).to(adt)
    k = (torch.rand(B, Nk, H, D, device="cuda") * 2 - 1).to(adt)
    v = (torch.rand(B, Nk, H, D, device="cuda") * 2 - 1).to(adt)
    t = timeit(lambda: ops.attention(q, k, v, heads=H, head_dim=D, scale=D ** -0.5, split3=split))
    fl = 4.0 * B * H * Nq * Nk * D
    print(f"attn B={B} H={H} Nq={Nq} Nk={Nk} D={D} a={adt} split={split}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    for adt, split in ((torch.bfloat16, False), (torch.float32, False), (torch.float32, True)):
        bench_gemm(2050, 3072, 1024, adt, split)
        bench_gemm(2050, 4096, 1024, adt, split, act=1)
        bench_gemm(2050, 1024, 4096, adt, split)
        bench_gemm(8192, 8192, 8192, adt, split)
        bench_conv(1, 512, 512, 256, 256, 3, adt, split)
        bench_conv(2, 128, 128, 256, 256, 3, adt, split)
        bench_attn(2, 16, 1025, 1025, 64, adt, split)
        bench_attn(2, 12, 1025, 1025, 64, adt, split)
        bench_attn(8, 16, 1025, 1025, 64, adt, split)
