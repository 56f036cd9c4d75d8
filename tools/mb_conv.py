"""Graph-timed convolution micro-benchmark (NHWC implicit GEMM): python tools/mb_conv.py [bf16|bf16x3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
split = mode == "bf16x3"
adt = torch.float32 if split else torch.bfloat16
def conv(B, H, W, cin, cout, k, label=""):
    x = (torch.rand(B, H, W, cin, device="cuda") * 2 - 1).to(adt)
    pw = ops.pack_conv(torch.rand(cout, cin, k, k, device="cuda") * 0.1, torch.zeros(cout, device="cuda"), split)
    t = graph_time(lambda: ops.conv2d(x, pw, pad=k // 2, out_dtype=adt), n=5)
    M, K = B * H * W, k * k * cin
    tiles = ((M + 127) // 128) * ((cout + 63) // 64)
    nkt = (K + 31) // 32 if split else (K + 63) // 64
    print(f"conv{k}x{k} {B}x{H}x{W}x{cin}->{cout} {mode}: {t*1e6:8.1f} us {2.0*M*cout*K/t/1e12:6.1f} TF/s | {tiles} tiles, {t*1e6/(max(1, tiles/512)*nkt):.2f} us per K tile and round {label}")
def dense(M, N, K):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(adt)
    pw = ops.pack_linear(torch.rand(N, K, device="cuda") * 0.1, torch.zeros(N, device="cuda"), split)
    out = torch.empty(M, N, device="cuda", dtype=adt)
    t = graph_time(lambda: ops.linear(a, pw, out=out), n=5)
    tiles = ((M + 127) // 128) * ((N + 63) // 64)
    nkt = (K + 31) // 32 if split else (K + 63) // 64
    print(f"dense {M}x{N}x{K} {mode}: {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:6.1f} TF/s | {tiles} tiles, {t*1e6/(max(1, tiles/512)*nkt):.2f} us per K tile and round")
conv(1, 512, 512, 256, 256, 3)
conv(1, 512, 512, 256, 256, 1)
dense(262144, 256, 256)
dense(262144, 256, 2304)
conv(1, 512, 512, 64, 256, 3)
conv(1, 512, 512, 128, 256, 3)
conv(1, 256, 256, 256, 256, 3)
conv(1, 128, 128, 256, 256, 3)
dense(8192, 4096, 1024)
