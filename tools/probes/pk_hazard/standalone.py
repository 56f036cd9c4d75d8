"""The panoptic argmax miscomputation WITHOUT the network: the round-3/4 code object of the argmax kernel (`e0`, tools/probes/pk_hazard/gen.py)
and its immune one-edit variants, launched on one stream over a synthetic probability volume, while a second stream runs a synthetic
MFMA + LDS neighbour kernel (burn.hip: 512 threads, one workgroup per CU, 96 KB of LDS -- the footprint of the library's ping-pong GEMM
workgroups).  Every label map is compared with the map the same code object produces on an otherwise idle chip.

    python tools/probes/pk_hazard/standalone.py [B] [iterations]

torch is used for device memory and streams only."""
import ctypes, os, subprocess, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
if not os.path.exists(os.path.join(BUILD, "ppa_fix.hsaco")):
    subprocess.check_call([sys.executable, os.path.join(HERE, "gen.py")])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
T, H, W, MS, Q, NK = 2, 512, 512, 256, 100, 25
hip = ctypes.CDLL("libamdhip64.so")
hip.hipModuleLaunchKernel.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 6 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


def load(path, name):
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipModuleLoad(ctypes.byref(mod), path.encode()) == 0, path
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, name.encode()) == 0, name
    return fn


def launch(fn, grid, block, stream, *vals):
    arr = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.byref(v), ctypes.c_void_p) for v in vals])
    rc = hip.hipModuleLaunchKernel(fn, grid[0], grid[1], 1, block, 1, 1, 0, ctypes.c_void_p(stream), arr, None)
    assert rc == 0, rc


VARS = [0, 1, 6, 4, 5, 8, 10, 11]
NAMES = {0: "e0 compiler's code", 1: "e1 vmcnt(0)", 6: "e6 scalar products, packed tail", 4: "e4 products in fresh registers", 5: "e5 scalar tail",
         8: "e8 packed sum into a fresh destination", 10: "e10 scalar adds", 11: "e11 gathers addressed from other registers"}
fns = {v: load(os.path.join(BUILD, f"ppa_e{v}.hsaco"), f"ppa_e{v}") for v in VARS}
VARS.append("fix"); NAMES["fix"] = "shipped form: no packed arithmetic at all"
fns["fix"] = load(os.path.join(BUILD, "ppa_fix.hsaco"), "ppa_fix")
burn = load(os.path.join(BUILD, "burn.hsaco"), "burn_mfma")
burn_wave = load(os.path.join(BUILD, "burn.hsaco"), "burn_wave")
burn_wave_dep = load(os.path.join(BUILD, "burn.hsaco"), "burn_wave_dep")

g = torch.Generator(device="cuda").manual_seed(5)
# a volume with soft segment borders: smooth per-query fields, so that neighbouring queries compete over wide bands (as the real masks do)
low = torch.randn(B * T, Q, 16, 16, device="cuda", generator=g)
p256 = torch.sigmoid(torch.nn.functional.interpolate(low, size=(MS, MS), mode="bicubic", align_corners=False) * 2.0).permute(0, 2, 3, 1).contiguous().view(B, T, MS, MS, Q)
scores = (0.55 + 0.45 * torch.rand(B, Q, device="cuda", generator=g)).contiguous()
kept = torch.stack([torch.randperm(Q, device="cuda", generator=g)[:NK].sort().values for _ in range(B)]).int().contiguous()
kept_full = torch.zeros(B, Q, dtype=torch.int32, device="cuda")
kept_full[:, :NK] = kept
n_keep = torch.full((B,), NK, dtype=torch.int32, device="cuda")
scr = torch.zeros(8192, dtype=torch.int32, device="cuda")
labs = {v: torch.zeros(B, T, H, W, dtype=torch.int32, device="cuda") for v in VARS}
ref = {}
src = torch.randint(0, 2 ** 31 - 1, (1 << 22,), dtype=torch.int32, device="cuda", generator=g)  # 16 MiB of operands for the neighbour
sink = torch.zeros(256 * 512, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def argmax(v, stream):
    launch(fns[v], ((T * H * W + 255) // 256, B), 256, stream.cuda_stream, ctypes.c_void_p(p256.data_ptr()), ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(kept_full.data_ptr()),
           ctypes.c_void_p(n_keep.data_ptr()), ctypes.c_void_p(labs[v].data_ptr()), ctypes.c_void_p(scr.data_ptr()), ctypes.c_void_p(scr.data_ptr() + 16384),
           ctypes.c_int(T), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(MS), ctypes.c_int(Q), ctypes.c_float(0.5))


torch.cuda.synchronize()
for v in VARS:  # reference maps: an idle chip, one launch at a time
    argmax(v, sa)
    torch.cuda.synchronize()
    ref[v] = labs[v].clone()
same = all(torch.equal(ref[v], ref[VARS[0]]) for v in VARS)
print(f"B = {B}: reference label maps of the {len(VARS)} code objects on an idle chip identical: {same}")
# neighbours from the library itself (the kernels that run beside the panoptic stage in the network's step)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from siu3r_amd import ops
LIBN = {}
if os.environ.get("PKH_LIB", "1") == "1":
    xa = torch.randn(1, 4096, 4096, device="cuda"); wa = ops.pack_linear(torch.randn(4096, 4096, device="cuda") * 0.02, None, True); oa = torch.empty(1, 4096, 4096, device="cuda")
    xm = torch.randn(1, 2050, 1024, device="cuda"); wm = ops.pack_linear(torch.randn(4096, 1024, device="cuda") * 0.02, None, True); om = torch.empty(1, 2050, 4096, device="cuda")
    xc = torch.randn(2, 256, 256, 256, device="cuda"); wc = ops.pack_conv(torch.randn(256, 256, 3, 3, device="cuda") * 0.02, None, True)
    xs = torch.randn(1, 100, 256, device="cuda"); ws_ = ops.pack_linear(torch.randn(256, 256, device="cuda") * 0.05, None, True)
    def n_tile(cfg):
        def f():
            ops.gemm_tune(0, cfg)
            for _ in range(24):
                ops.linear(xa, wa, out=oa)
            ops.gemm_tune(0, 0)
        return f
    def n_enc():
        for _ in range(120):
            ops.linear(xm, wm, out=om, act=ops.ACT_GELU)
    def n_conv():
        for _ in range(40):
            ops.conv2d(xc, wc, pad=1, act=ops.ACT_RELU, out_dtype=torch.float32)
    def n_small():
        for _ in range(600):
            ops.linear(xs, ws_, out_dtype=torch.float32)
    xl = torch.randn(16, 5376, 1024, device="cuda"); gl = torch.ones(1024, device="cuda"); bl_ = torch.zeros(1024, device="cuda")
    def n_ln():
        for _ in range(20):
            ops.layernorm(xl, gl, bl_, 1e-6, torch.float32)
    def n_bf16():
        ops.gemm_tune(0, -1)
        for _ in range(40):
            ops.linear(xab, wab, out=oab)
        ops.gemm_tune(0, 0)
    xab = torch.randn(1, 4096, 4096, device="cuda").bfloat16(); wab = ops.pack_linear(torch.randn(4096, 4096, device="cuda") * 0.02, None, False); oab = torch.empty(1, 4096, 4096, device="cuda", dtype=torch.bfloat16)
    LIBN = {"lib: ping-pong GEMM 4096^3, 256 x 256 tiles": n_tile(1), "lib: ping-pong GEMM 4096^3, 256 x 128 tiles": n_tile(2), "lib: ping-pong GEMM 4096^3, 128 x 128 tiles": n_tile(3),
            "lib: 128 x 64 LDS-DMA GEMM 4096^3": n_tile(-1), "lib: encoder fc1 2050 x 4096 x 1024 + GELU (auto tile)": n_enc, "lib: 3 x 3 convolution 2 x 256^2 x 256 -> 256": n_conv,
            "lib: 100-row GEMMs (the Mask2Former population)": n_small, "lib: LayerNorm 86016 x 1024": n_ln, "lib: 128 x 64 LDS-DMA GEMM 4096^3, bf16 operands": n_bf16}
    only = os.environ.get("PKH_ONLY")
    if only:
        LIBN = {k: f for k, f in LIBN.items() if only in k}
SYN = [(9, "neighbour: one-wave workgroups of bf16 MFMAs, two per SIMD"), (10, "neighbour: one-wave workgroups, ONE dependent MFMA chain each, two per SIMD"), (11, "neighbour: one dependent-chain MFMA wave per SIMD")] if os.environ.get("PKH_ONLY") else [(9, "neighbour: one-wave workgroups of bf16 MFMAs, two per SIMD"), (10, "neighbour: one-wave workgroups, ONE dependent MFMA chain each, two per SIMD"), (0, "neighbour: MFMA + LDS reads"), (1, "neighbour: MFMA + LDS reads + vector work"), (3, "neighbour: MFMA + LDS + vector work + global loads")]
for mode, what in [(None, "no neighbour")] + SYN + [(k, k) for k in LIBN]:
    wrong = {v: [0, 0] for v in VARS}
    for it in range(N):
        if isinstance(mode, str):
            with torch.cuda.stream(sb):
                LIBN[mode]()
        elif mode == 9:
            launch(burn_wave, (2048, 1), 64, sb.cuda_stream, ctypes.c_void_p(sink.data_ptr()), ctypes.c_int(60000))
        elif mode == 10:
            launch(burn_wave_dep, (2048, 1), 64, sb.cuda_stream, ctypes.c_void_p(sink.data_ptr()), ctypes.c_int(120000))
        elif mode == 11:
            launch(burn_wave_dep, (1024, 1), 64, sb.cuda_stream, ctypes.c_void_p(sink.data_ptr()), ctypes.c_int(240000))
        elif mode is not None:
            launch(burn, (256, 1), 512, sb.cuda_stream, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(sink.data_ptr()), ctypes.c_int(20000), ctypes.c_int(mode))
        for v in VARS:
            argmax(v, sa)
        torch.cuda.synchronize()
        for v in VARS:
            d = int((labs[v] != ref[v]).sum())
            wrong[v][0] += d > 0
            wrong[v][1] += d
    print(f"-- {what}: launches with a wrong label map of {N} / wrong pixels in total")
    for v in VARS:
        print(f"     {NAMES[v]:45s} {wrong[v][0]:3d} / {wrong[v][1]}")
