"""Cost of a dependent kernel boundary: a chain of trivial launches (siu3r_scale_inplace on 64 floats; the smallest GEMM) as nodes of one HIP
graph vs eagerly on a stream.   python tools/launch_floor.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd import ops, raster

x = torch.ones(64, device="cuda")
a = torch.rand(100, 256, device="cuda")
pw = ops.pack_linear(torch.rand(256, 256, device="cuda") * 0.1, None, True)
out = torch.empty(100, 256, device="cuda")
big = torch.ones(64 << 20, device="cuda")  # 256 MB: a kernel that leaves the L2s full of dirty lines


def chain(fn, n):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3


def eager(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, (t1 - t0) / n * 1e6


for name, fn in (("scale 64 floats", lambda: raster.scale_inplace_(x, 1.0)), ("gemm 100x256x256 bf16x3", lambda: ops.linear(a, pw, out=out)),
                 ("torch add_ 64 floats", lambda: x.add_(0.0)), ("scale 256 MB", lambda: raster.scale_inplace_(big, 1.0))):
    g50 = chain(fn, 50)
    ev, host = eager(fn, 200)
    print(f"{name:26s}: graph chain {g50:6.2f} us / node | eager stream {ev:6.2f} us / launch (host enqueue {host:5.2f} us)")
