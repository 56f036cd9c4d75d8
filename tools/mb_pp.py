"""Ping-pong GEMM kernels (gemm_pp_kernel.h) vs the 128 x 64 kernels: correctness against torch fp32 and graph-timed TF/s per tile
configuration.  python tools/mb_pp.py [check] [bench] [big]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from siu3r_amd import _lib, ops
from mb_gemm import graph_time

CFG = {0: "auto", -1: "128x64", 1: "pp256x256", 2: "pp256x128", 3: "pp128x128"}


def tune(v):
    _lib.check(_lib.lib().siu3r_gemm_tune(0, v))


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-20)).item()


def check():
    torch.manual_seed(0)
    bad = 0
    for mode in ("bf16x3", "bf16"):
        split = mode == "bf16x3"
        adt = torch.float32 if split else torch.bfloat16
        tol = 2e-5 if split else 1.5e-2
        for (M, N, K, act, res) in [(512, 512, 256, 0, False), (300, 520, 1024, 1, True), (2050, 1024, 1024, 0, True), (1025, 768, 3072, 0, False),
                                    (256, 256, 64, 0, False), (777, 264, 136, 2, False), (4096, 4096, 512, 0, False), (2050, 520, 1024, 1, True),
                                    (1025, 768, 768, 0, True), (30, 256, 512, 0, False), (2, 136, 4096, 0, True), (16400, 256, 128, 0, False)]:
            a = (torch.rand(M, K, device="cuda") * 2 - 1)
            w = (torch.rand(N, K, device="cuda") * 2 - 1) * 0.1
            b = torch.rand(N, device="cuda")
            r = torch.rand(M, N, device="cuda") if res else None
            pw = ops.pack_linear(w, b, split)
            ax = a.to(adt)
            ref = ax.float() @ (w if split else w.to(torch.bfloat16).float()).t() + b
            if act == 1:
                ref = torch.nn.functional.gelu(ref)
            elif act == 2:
                ref = torch.relu(ref)
            if res:
                ref = ref + r
            for cfg in (0, 1, 2, 3):
                tune(cfg)
                out = ops.linear(ax, pw, act=act, residual=r, out_dtype=torch.float32)
                e = rel(out, ref)
                ok = e < tol
                bad += (not ok)
                print(f"check {mode} {M}x{N}x{K} act={act} res={res} {CFG[cfg]}: {e:.2e} {'ok' if ok else 'FAIL'}")
        # convolutions: 3x3 with cin 64 (tap cursor) and the small-cin path
        for (B, H, W, cin, cout, k) in [(1, 64, 64, 64, 256, 3), (2, 32, 48, 128, 128, 1), (1, 64, 64, 8 if not split else 4, 64, 7), (1, 40, 40, 48, 96, 3)]:
            x = (torch.rand(B, H, W, cin, device="cuda") * 2 - 1)
            w = (torch.rand(cout, cin, k, k, device="cuda") * 2 - 1) * 0.1
            b = torch.rand(cout, device="cuda")
            pw = ops.pack_conv(w, b, split)
            xx = x.to(adt)
            ref = torch.nn.functional.conv2d(xx.float().permute(0, 3, 1, 2), w if split else w.to(torch.bfloat16).float(), b, padding=k // 2).permute(0, 2, 3, 1)
            for cfg in (1, 2, 3):
                tune(cfg)
                out = ops.conv2d(xx, pw, pad=k // 2, out_dtype=torch.float32)
                e = rel(out, ref)
                ok = e < tol
                bad += (not ok)
                print(f"check {mode} conv{k} {B}x{H}x{W}x{cin}->{cout} {CFG[cfg]}: {e:.2e} {'ok' if ok else 'FAIL'}")
    tune(0)
    print("CHECK", "FAILED" if bad else "passed", bad)
    return bad


def bench(big):
    rows = []
    shapes = [(8192, 8192, 1024), (4096, 4096, 4096), (2048, 4096, 1024), (2050, 4096, 1024), (2050, 3072, 1024), (2050, 1024, 4096), (2050, 1024, 1024),
              (16400, 4096, 1024), (16400, 1024, 1024), (262144, 256, 2304), (262144, 256, 256)]
    if not big:
        shapes = shapes[:7]
    for mode in ("bf16x3", "bf16"):
        split = mode == "bf16x3"
        adt = torch.float32 if split else torch.bfloat16
        for (M, N, K) in shapes:
            a = (torch.rand(M, K, device="cuda") * 2 - 1).to(adt)
            pw = ops.pack_linear((torch.rand(N, K, device="cuda") * 2 - 1) * 0.1, torch.zeros(N, device="cuda"), split)
            out = torch.empty(M, N, device="cuda", dtype=adt)
            for cfg in (0, -1, 1, 2, 3):
                tune(cfg)
                t = graph_time(lambda: ops.linear(a, pw, out=out), n=10)
                tf = 2.0 * M * N * K / t / 1e12
                rows.append(dict(mode=mode, M=M, N=N, K=K, cfg=CFG[cfg], us=t * 1e6, tflops=tf))
                print(f"bench {mode} {M}x{N}x{K} {CFG[cfg]:>10}: {t*1e6:8.1f} us {tf:7.1f} TF/s", flush=True)
        # the full-resolution 3x3 convolution of the Gaussian heads
        x = (torch.rand(1, 512, 512, 256, device="cuda") * 2 - 1).to(adt)
        pw = ops.pack_conv((torch.rand(256, 256, 3, 3, device="cuda") * 2 - 1) * 0.1, torch.zeros(256, device="cuda"), split)
        for cfg in (0, -1, 1, 2, 3):
            tune(cfg)
            t = graph_time(lambda: ops.conv2d(x, pw, pad=1, out_dtype=adt), n=5)
            tf = 2.0 * 262144 * 256 * 2304 / t / 1e12
            rows.append(dict(mode=mode, M=262144, N=256, K=2304, conv=3, cfg=CFG[cfg], us=t * 1e6, tflops=tf))
            print(f"bench {mode} conv3x3 512x512x256->256 {CFG[cfg]:>10}: {t*1e6:8.1f} us {tf:7.1f} TF/s", flush=True)
    tune(0)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/mb_pp.json", "w"), indent=1)


if __name__ == "__main__":
    args = sys.argv[1:] or ["check", "bench"]
    rc = 0
    if "check" in args:
        rc = check()
    if "bench" in args:
        bench("big" in args)
    sys.exit(1 if rc else 0)
