"""Per-kernel parity tests: HIP path (through the C ABI in libsiu3r_hip.so) vs the CPU oracle /
plain fp32 torch on the same seeded inputs.  Tolerances (max |err| / max |ref|):
  bf16 operand mode ......... 2e-2  (bf16 has 8 mantissa bits; documented deviation from 1e-3)
  bf16x3 / fp32 kernels ...... 2e-4  (well inside the north-star's 1e-3)
"""
import math
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_BF16 = 2e-2
TOL_F32 = 2e-4


def _ops():
    from siu3r_amd import ops

    return ops


def rel_err(got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def check(name, got, ref, tol):
    e = rel_err(got, ref)
    print(f"[parity] {name}: rel_err={e:.3e} tol={tol:.1e}")
    assert math.isfinite(e) and e <= tol, f"{name}: rel_err {e:.3e} > {tol:.1e}"


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


MODES = [("bf16", torch.bfloat16, False, TOL_BF16), ("f32", torch.float32, False, TOL_BF16),
         ("bf16x3", torch.float32, True, TOL_F32)]


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
def test_gemm_dense(mode):
    ops = _ops()
    name, adt, split, tol = mode
    M, N, K = 300, 200, 264  # ragged in every dimension; K % 8 == 0 but not % 64
    a, w, b, r = gen(M, K, seed=1), gen(N, K, seed=2, scale=0.2), gen(N, seed=3), gen(M, N, seed=4)
    pw = ops.pack_linear(w.cuda(), b.cuda(), split)
    ref = F.gelu(a.to(adt).float() @ w.t() + b) + r
    out = ops.linear(a.cuda().to(adt), pw, out_dtype=torch.float32, act=ops.ACT_GELU, residual=r.cuda())
    check(f"gemm_dense[{name}] gelu+residual f32 out", out, ref, tol)
    out2 = ops.linear(a.cuda().to(adt), pw, out_dtype=adt, act=ops.ACT_RELU)
    check(f"gemm_dense[{name}] relu act-dtype out", out2, F.relu(a.to(adt).float() @ w.t() + b), max(tol, 5e-3 if adt == torch.bfloat16 else 0))
    # asymmetric A = I check (transpose detecting): out == W^T rows
    eye = torch.eye(K)[:64]
    out3 = ops.linear(eye.cuda().to(adt), ops.pack_linear(w.cuda(), None, split), out_dtype=torch.float32)
    check(f"gemm_dense[{name}] A=I", out3, w.t()[:64], tol)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
def test_gemm_rope_epilogue(mode):
    """QKV GEMM with RoPE2D fused on the q|k columns == linear followed by the oracle's rope2d."""
    from oracle import siu3r_oracle as O

    ops = _ops()
    name, adt, split, tol = mode
    B, N, H, D, K = 2, 37, 3, 64, 128
    x = gen(B, N, K, seed=60)
    w, b = gen(3 * H * D, K, seed=61, scale=0.3), gen(3 * H * D, seed=62)
    pos = torch.randint(0, 20, (B, N, 2), generator=torch.Generator().manual_seed(5))
    cos, sin = O.rope2d_table(20, D)
    qkv = (x.to(adt).float() @ w.t() + b).view(B, N, 3, H, D)
    q = O.rope2d(qkv[:, :, 0].permute(0, 2, 1, 3), pos).permute(0, 2, 1, 3)
    k = O.rope2d(qkv[:, :, 1].permute(0, 2, 1, 3), pos).permute(0, 2, 1, 3)
    ref = torch.stack((q, k, qkv[:, :, 2]), dim=2).reshape(B, N, 3 * H * D)
    pw = ops.pack_linear(w.cuda(), b.cuda(), split)
    out = ops.linear(x.cuda().to(adt), pw, out_dtype=torch.float32, rope=(cos.cuda(), sin.cuda(), pos.cuda(), 2 * H * D))
    check(f"gemm_rope_epilogue[{name}]", out, ref, tol)


def test_proj_rows_x3_dedicated_kernel():
    """The heads' last 1 x 1 convolutions as row streams (csrc/proj.hip, siu3r_proj_rows_x3): two heads x two batch items, a ragged row
    count, 83 columns of 256 (rows of 83 floats) and 3 columns of 128 -- against fp32 torch and against the GEMM route it replaces."""
    ops = _ops()
    for (K, N, M) in ((256, 83, 1000), (128, 3, 777), (128, 4, 4096)):
        B, G = 2, 2
        x = gen(B, G, M, K, seed=71)
        ws = [gen(N, K, seed=72 + g, scale=0.2) for g in range(G)]
        bs = [gen(N, seed=74 + g) for g in range(G)]
        assert ops.proj_rows_ok(K, N)
        wf, bias, n = ops.pack_proj([w.cuda() for w in ws], [b.cuda() for b in bs])
        out = torch.full((B, G, M, N), float("nan"), device="cuda")
        ops.proj_rows_x3(x.cuda(), wf, bias, n, out)
        ref = torch.stack([torch.stack([x[b, g] @ ws[g].t() + bs[g] for g in range(G)]) for b in range(B)])
        check(f"proj_rows_x3 {M} x {N} x {K}", out, ref, 2e-5)
        pw = ops.stack_packed([ops.pack_linear(w.cuda(), b.cuda(), True) for w, b in zip(ws, bs)])
        old = ops.linear_grouped(x.cuda(), pw, out_dtype=torch.float32)
        check(f"proj_rows_x3 vs grouped GEMM {M} x {N} x {K}", out, old, 2e-6)
        wf1, _, _ = ops.pack_proj([ws[0].cuda()], [None])  # one head, no bias, B * G = 1
        o1 = torch.full((1, 1, M, N), float("nan"), device="cuda")
        ops.proj_rows_x3(x[:1, :1].contiguous().cuda(), wf1, None, n, o1)
        check(f"proj_rows_x3 (no bias) {M} x {N} x {K}", o1[0, 0], x[0, 0] @ ws[0].t(), 2e-5)
        buf = torch.full((B, 2, M, N), float("nan"), device="cuda")  # one head for view 1 of every batch item: a strided [B, 1, M, N] destination
        ops.proj_rows_x3(x[:, 1:2].contiguous().cuda(), wf[1:2].contiguous(), bias[1:2].contiguous(), n, buf[:, 1].unsqueeze(1))
        check(f"proj_rows_x3 (batch-strided destination) {M} x {N} x {K}", buf[:, 1], ref[:, 1], 2e-5)
        assert torch.isnan(buf[:, 0]).all()


def test_stem7x7_x3_dedicated_kernel():
    """The Gaussian heads' stem as one dedicated kernel (csrc/stem.hip, siu3r_stem7x7_x3): two heads (weight sets) x two batch items, a
    ragged tile grid (48 x 32: borders on every side of most tiles), with and without the upsample source and the bias, fp32 output and
    pre-split planes -- against fp32 torch (dpt_gs_head.py:71-77, 158-162), and against the implicit-GEMM convolution it replaces."""
    ops = _ops()
    B, G, H, W = 2, 2, 48, 32
    img = gen(B, G, 3, H, W, seed=61)
    ws = [gen(256, 3, 7, 7, seed=62 + g, scale=0.2) for g in range(G)]
    bs = [gen(256, seed=64 + g) for g in range(G)]
    low = gen(B, G, 256, H // 2, W // 2, seed=66)
    imgp = ops.pack_image_nhwc(img.flatten(0, 1).cuda(), torch.float32, 4).view(B, G, H, W, 4)
    lowc = low.permute(0, 1, 3, 4, 2).contiguous().cuda()
    for use_up, use_bias in ((True, True), (False, True), (True, False)):
        wfrag, bias = ops.pack_stem7([w.cuda() for w in ws], [b.cuda() if use_bias else None for b in bs])
        ref = torch.stack([torch.stack([F.relu(F.conv2d(img[b, g][None], ws[g], bs[g] if use_bias else None, padding=3))[0] for g in range(G)]) for b in range(B)])
        if use_up:
            ref = ref + F.interpolate(low.flatten(0, 1), scale_factor=2, mode="bilinear", align_corners=True).view(B, G, 256, H, W)
        ref = ref.permute(0, 1, 3, 4, 2)
        out = torch.full((B, G, H, W, 256), float("nan"), device="cuda")
        ops.stem7x7_x3(imgp, wfrag, bias, lowc if use_up else None, out)
        check(f"stem7x7_x3 (up {use_up}, bias {use_bias})", out, ref, 2e-5)
        # pre-split planes: per pixel and 32 channels [hi 32 | lo 32] bf16; hi = the upper 16 bits, hi + lo = the value to 2^-16
        pl = torch.full((B, G, H, W, 256), float("nan"), device="cuda")
        ops.stem7x7_x3(imgp, wfrag, bias, lowc if use_up else None, pl, planes=True)
        raw = pl.view(torch.bfloat16).view(B, G, H, W, 8, 2, 32)
        hi, lo = raw[..., 0, :].float().reshape(B, G, H, W, 256), raw[..., 1, :].float().reshape(B, G, H, W, 256)
        assert torch.equal(hi.view(torch.int32), out.view(torch.int32) & -65536), "hi plane = the upper 16 bits of the fp32 output"
        assert ((hi + lo - out).abs() <= out.abs() * 2.0 ** -15 + 1e-30).all()
        # the implicit-GEMM route (same bf16x3 products, another summation order)
        for g in range(G):
            pw = ops.pack_conv(ws[g].cuda(), bs[g].cuda() if use_bias else None, True, cin_pad=4)
            old = ops.conv2d(imgp[:, g].contiguous(), pw, stride=1, pad=3, act=ops.ACT_RELU, out_dtype=torch.float32, up_src=lowc[:, g].contiguous() if use_up else None)
            check(f"stem7x7_x3 vs implicit GEMM, head {g}", out[:, g], old, 2e-6)


@pytest.mark.parametrize("tile", [1, 2, 3], ids=["pp256x256", "pp256x128", "pp128x128"])
def test_conv2d_upsample_add_on_ping_pong_tiles(tile):
    """The same fused stem on forced ping-pong tiles with 64 output channels and two images (the fast row pass carries the upsample-add:
    fp32 source map, batch index and pixel coordinates recovered from the row number) and with 40 channels (the general pass)."""
    ops = _ops()
    B, H, W = 2, 24, 40
    img = gen(B, 3, H, W, seed=15).abs()
    ops.gemm_tune(0, tile)
    log = []
    ops.set_plan_log(log)
    try:
        for C_ in (64, 40):
            low = gen(B, C_, H // 2, W // 2, seed=16)
            w, b = gen(C_, 3, 7, 7, seed=17, scale=0.2), gen(C_, seed=18)
            ref = (F.relu(F.conv2d(img, w, b, padding=3)) + F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=True)).permute(0, 2, 3, 1)
            pw = ops.pack_conv(w.cuda(), b.cuda(), True, cin_pad=4)
            out = ops.conv2d(ops.pack_image_nhwc(img.cuda(), torch.float32, 4), pw, stride=1, pad=3, act=ops.ACT_RELU, out_dtype=torch.float32,
                             up_src=low.permute(0, 2, 3, 1).contiguous().cuda())
            assert log[-1].tile_cfg == tile
            check(f"conv7x7+up_add on tile {tile}, {C_} channels", out, ref, TOL_F32)
    finally:
        ops.set_plan_log(None)
        ops.gemm_tune(0, 0)


@pytest.mark.parametrize("tile", [1, 2, 3], ids=["pp256x256", "pp256x128", "pp128x128"])
@pytest.mark.parametrize("split", [True, False], ids=["bf16x3", "bf16"])
def test_gemm_ping_pong_row_passes(tile, split):
    """The ping-pong tiles' two row passes on forced tiles and a ragged M: the fast pass (fp32 / bf16 C, N % 64 == 0) with RoPE on the
    q | k columns, folded LayerNorm + GELU, residual + row statistics, bf16 C with a bf16 residual; and the general pass on the same
    problems made ineligible (N % 64 != 0)."""
    from oracle import siu3r_oracle as O

    ops = _ops()
    adt, tol = (torch.float32, TOL_F32) if split else (torch.bfloat16, TOL_BF16)
    M, K, H, D = 300, 256, 2, 64
    x = gen(M, K, seed=110)
    xa = x.cuda().to(adt)
    log = []
    ops.gemm_tune(0, tile)
    ops.set_plan_log(log)
    try:
        # RoPE on the first 2 H D columns of a 3 H D wide projection (fp32 C)
        w, b = gen(3 * H * D, K, seed=111, scale=0.3), gen(3 * H * D, seed=112)
        pos = torch.randint(0, 20, (1, M, 2), generator=torch.Generator().manual_seed(6))
        cos, sin = O.rope2d_table(20, D)
        qkv = (xa.float().cpu() @ w.t() + b).view(1, M, 3, H, D)
        q = O.rope2d(qkv[:, :, 0].permute(0, 2, 1, 3), pos).permute(0, 2, 1, 3)
        k = O.rope2d(qkv[:, :, 1].permute(0, 2, 1, 3), pos).permute(0, 2, 1, 3)
        ref = torch.stack((q, k, qkv[:, :, 2]), dim=2).reshape(M, 3 * H * D)
        out = ops.linear(xa, ops.pack_linear(w.cuda(), b.cuda(), split), out_dtype=torch.float32, rope=(cos.cuda(), sin.cuda(), pos.cuda(), 2 * H * D))
        assert log[-1].tile_cfg == tile
        check(f"pp row pass rope [{tile}]", out, ref, tol)
        # residual + statistics out (fp32 C), then the folded LayerNorm + GELU that consumes them
        w1, b1 = gen(K, K, seed=113, scale=0.2), gen(K, seed=114)
        r = gen(M, K, seed=115)
        y = torch.empty(M, K, device="cuda")
        st = ops.RowStats(y)
        yb = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
        ops.linear(xa, ops.pack_linear(w1.cuda(), b1.cuda(), split), residual=r.cuda(), out=y, stats_out=st, aux_out=None if split else yb)
        yref = xa.float().cpu() @ w1.t() + b1 + r
        check(f"pp row pass residual [{tile}]", y, yref, tol)
        part = st.buf.cpu()
        assert float((part[:, :, 0].mean(1) - y.cpu().mean(1)).abs().max()) <= 1e-5
        gm, bt = 1 + 0.2 * gen(K, seed=116), gen(K, seed=117)
        w2, b2 = gen(320, K, seed=118, scale=0.2), gen(320, seed=119)
        pw2 = ops.pack_linear_ln(w2.cuda(), b2.cuda(), gm.cuda(), bt.cuda(), split)
        o2 = ops.linear(y if split else yb, pw2, out_dtype=torch.float32, act=ops.ACT_GELU, ln=st)
        ref2 = F.gelu(F.layer_norm(y.cpu(), (K,), gm, bt, 1e-6) @ w2.t() + b2)
        check(f"pp row pass ln+gelu [{tile}]", o2, ref2, max(tol, 2e-2 if not split else 0))
        # bf16 C with a bf16 residual (fast pass, BF layout) -- and ReLU
        if not split:
            rb = gen(M, K, seed=120).to(torch.bfloat16)
            o3 = ops.linear(xa, ops.pack_linear(w1.cuda(), b1.cuda(), False), out_dtype=torch.bfloat16, act=ops.ACT_RELU, residual=rb.cuda())
            check(f"pp row pass bf16 C [{tile}]", o3, torch.relu(xa.float().cpu() @ w1.t() + b1) + rb.float(), TOL_BF16)
        # the general pass: N = 200 is not a multiple of 64
        w4, b4 = gen(200, K, seed=121, scale=0.2), gen(200, seed=122)
        r4 = gen(M, 200, seed=123)
        o4 = ops.linear(xa, ops.pack_linear(w4.cuda(), b4.cuda(), split), out_dtype=torch.float32, act=ops.ACT_GELU, residual=r4.cuda())
        assert log[-1].tile_cfg == tile
        check(f"pp general pass [{tile}]", o4, F.gelu(xa.float().cpu() @ w4.t() + b4) + r4, tol)
    finally:
        ops.set_plan_log(None)
        ops.gemm_tune(0, 0)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
def test_gemm_batched_strided(mode):
    """tokens[:, :-1] view (strip the intrinsics token) feeding a 1x1 conv / linear."""
    ops = _ops()
    name, adt, split, tol = mode
    B, Nt, K, N = 2, 131, 128, 96
    x = gen(B, Nt, K, seed=5)
    w, b = gen(N, K, seed=6, scale=0.3), gen(N, seed=7)
    pw = ops.pack_linear(w.cuda(), b.cuda(), split)
    xg = x.cuda().to(adt)
    out = ops.linear(xg[:, :-1], pw, out_dtype=torch.float32)
    check(f"gemm_batched[{name}]", out, x[:, :-1].to(adt).float() @ w.t() + b, tol)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("cfg", [(3, 1, 1, 16, 24), (3, 2, 1, 24, 40), (1, 1, 0, 64, 16), (7, 1, 3, 3, 32), (3, 1, 1, 64, 40), (3, 2, 1, 96, 72)],
                         ids=["k3s1", "k3s2", "k1", "k7c3", "k3c64_ring", "k3s2c96_ring"])
def test_conv2d(mode, cfg):
    ops = _ops()
    name, adt, split, tol = mode
    k, s, pd, cin, cout = cfg
    B, H, W = 2, 20, 28
    x = gen(B, cin, H, W, seed=8)
    w, b = gen(cout, cin, k, k, seed=9, scale=0.3), gen(cout, seed=10)
    cin_pad = 8 if cin == 3 else None
    pw = ops.pack_conv(w.cuda(), b.cuda(), split, cin_pad=cin_pad)
    if cin == 3:
        xg = ops.pack_image_nhwc8(x.cuda(), adt)
    else:
        xg = x.permute(0, 2, 3, 1).contiguous().cuda().to(adt)
    ref = F.conv2d(F.relu(x.to(adt).float()), w, b, stride=s, padding=pd).permute(0, 2, 3, 1)
    out = ops.conv2d(xg, pw, stride=s, pad=pd, out_dtype=torch.float32, relu_in=True)
    check(f"conv2d[{name}] k{k}s{s} relu_in", out, ref, tol)
    # without the fused input ReLU the small-cin shapes take the per-lane tap-state gather of the LDS-DMA kernels (bf16 and bf16x3),
    # and cin = 96 (K = 864, kpad = 896) runs the tap-cursor kernel over a zero-padded last K tile
    ref = F.conv2d(x.to(adt).float(), w, b, stride=s, padding=pd).permute(0, 2, 3, 1)
    out = ops.conv2d(xg, pw, stride=s, pad=pd, out_dtype=torch.float32)
    check(f"conv2d[{name}] k{k}s{s}", out, ref, tol)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
def test_conv2d_upsample_add_epilogue(mode):
    """GS head: feat_up(path_1) + ReLU(conv7x7(img)) fused (dpt_gs_head.py:160-162)."""
    ops = _ops()
    name, adt, split, tol = mode
    B, H, W, C = 1, 16, 24, 32
    img = gen(B, 3, H, W, seed=11).abs()
    low = gen(B, C, H // 2, W // 2, seed=12)
    w, b = gen(C, 3, 7, 7, seed=13, scale=0.2), gen(C, seed=14)
    ref = (F.relu(F.conv2d(img.to(adt).float(), w, b, padding=3)) + F.interpolate(low.to(adt).float(), scale_factor=2, mode="bilinear", align_corners=True)).permute(0, 2, 3, 1)
    for cp in ((8, 4) if split else (8,)):  # the bf16x3 mode packs RGB + one zero channel (ops.image_channels)
        pw = ops.pack_conv(w.cuda(), b.cuda(), split, cin_pad=cp)
        out = ops.conv2d(ops.pack_image_nhwc(img.cuda(), adt, cp), pw, stride=1, pad=3, act=ops.ACT_RELU, out_dtype=torch.float32,
                         up_src=low.permute(0, 2, 3, 1).contiguous().cuda().to(adt))
        check(f"conv7x7+up_add[{name}] cin_pad={cp}", out, ref, tol)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("k", [2, 4])
def test_conv_transpose(mode, k):
    ops = _ops()
    name, adt, split, tol = mode
    B, H, W, cin, cout = 2, 6, 5, 48, 24
    x = gen(B, cin, H, W, seed=15)
    w, b = gen(cin, cout, k, k, seed=16, scale=0.3), gen(cout, seed=17)
    pw = ops.pack_conv_transpose(w.cuda(), b.cuda(), split)
    ref = F.conv_transpose2d(x.to(adt).float(), w, b, stride=k).permute(0, 2, 3, 1)
    out = ops.conv_transpose2d(x.permute(0, 2, 3, 1).contiguous().cuda().to(adt), pw, out_dtype=torch.float32)
    check(f"conv_transpose[{name}] k{k}", out, ref, tol)


@pytest.mark.parametrize("split", [False, True], ids=["bf16", "bf16x3"])
def test_patch_embed(split):
    ops = _ops()
    B, H, W, Cc = 2, 64, 96, 80
    img = gen(B, 3, H, W, seed=18).abs()
    w, b = gen(Cc, 3, 16, 16, seed=19, scale=0.1), gen(Cc, seed=20)
    pw = ops.pack_matrix(w.reshape(Cc, -1).cuda(), b.cuda(), split)
    n = (H // 16) * (W // 16)
    out = torch.zeros(B, n + 1, Cc, device="cuda")
    ops.patch_embed(img.cuda(), pw, out)
    ref = F.conv2d(img, w, b, stride=16).flatten(2).transpose(1, 2)
    check(f"patch_embed[{'bf16x3' if split else 'bf16'}]", out[:, :n], ref, TOL_F32 if split else TOL_BF16)
    assert float(out[:, n].abs().max()) == 0.0  # extra token untouched


@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_layernorm(odt):
    ops = _ops()
    x = gen(37, 1024, seed=21, scale=3.0) + 0.5
    g, b = gen(1024, seed=22) + 1.0, gen(1024, seed=23)
    out = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6, out_dtype=odt)
    check(f"layernorm[{odt}]", out, F.layer_norm(x, (1024,), g, b, 1e-6), 1e-5 if odt == torch.float32 else 8e-3)
    x3 = gen(2, 9, 768, seed=24)
    out3 = ops.layernorm(x3.cuda()[:, :-1], g[:768].cuda().contiguous(), b[:768].cuda().contiguous(), 1e-5, out_dtype=odt)
    check("layernorm strided", out3, F.layer_norm(x3[:, :-1], (768,), g[:768], b[:768], 1e-5), 1e-5 if odt == torch.float32 else 8e-3)


def test_rope2d_seam():
    """seam 1: curope.rope_2d(tokens, positions, base, fwd) in place on a [B,H,N,D]->transpose(1,2) view."""
    from oracle import siu3r_oracle as O

    ops = _ops()
    B, H, N, D = 2, 4, 33, 64
    tok = gen(B, H, N, D, seed=25)
    pos = torch.stack([torch.randint(0, 33, (B, N), generator=torch.Generator().manual_seed(1)),
                       torch.randint(0, 33, (B, N), generator=torch.Generator().manual_seed(2))], -1)
    ref = O.rope2d(tok, pos, 100.0, 1.0)
    t = tok.cuda()  # [B,H,N,D]
    view = t.transpose(1, 2)  # [B,N,H,D] view accepted by the reference (kernels.cu:91) only if stride(2)==D
    cont = view.contiguous()
    assert ops.rope_2d(cont, pos.cuda(), 100.0, 1.0) is None
    check("rope2d fwd", cont.transpose(1, 2), ref, 2e-6)
    ops.rope_2d(cont, pos.cuda(), 100.0, -1.0)  # backward = inverse rotation
    check("rope2d fwd+bwd = identity", cont.transpose(1, 2), tok, 2e-6)
    # qkv-packed view [B,N,3,H,D][:, :, 0] has stride(2)==D and is accepted in place
    qkv = gen(B, N, 3, H, D, seed=26).cuda()
    ref_q = O.rope2d(qkv[:, :, 0].permute(0, 2, 1, 3).cpu(), pos)
    ops.rope_2d(qkv[:, :, 0], pos.cuda(), 100.0, 1.0)
    check("rope2d strided view", qkv[:, :, 0].permute(0, 2, 1, 3), ref_q, 2e-6)
    bf = tok.transpose(1, 2).contiguous().cuda().bfloat16()
    ops.rope_2d(bf, pos.cuda(), 100.0, 1.0)
    check("rope2d bf16", bf.transpose(1, 2), O.rope2d(tok.bfloat16().float(), pos), 8e-3)
    # the reference's kernel dispatches float, double and half (kernels.cu:101): the seam takes those too
    hf = tok.transpose(1, 2).contiguous().cuda().half()
    ops.rope_2d(hf, pos.cuda(), 100.0, 1.0)
    check("rope2d fp16", hf.transpose(1, 2), O.rope2d(tok.half().float(), pos), 1e-3)
    db = tok.transpose(1, 2).contiguous().cuda().double()
    ops.rope_2d(db, pos.cuda(), 100.0, 1.0)
    check("rope2d fp64", db.transpose(1, 2), ref, 2e-6)
    # error behaviour mirrors TORCH_CHECK -> RuntimeError (curope.cpp:54-59, kernels.cu:91-94)
    with pytest.raises(RuntimeError):
        ops.rope_2d(t[0], pos.cuda(), 100.0, 1.0)
    with pytest.raises(RuntimeError):
        ops.rope_2d(view, pos.cuda(), 100.0, 1.0)  # non-contiguous head stride
    with pytest.raises(RuntimeError):
        ops.rope_2d(cont, pos.cuda()[:, :5], 100.0, 1.0)
    with pytest.raises(RuntimeError):
        ops.rope_2d(gen(1, 2, 2, 6).cuda(), torch.zeros(1, 2, 2, dtype=torch.int64).cuda(), 100.0, 1.0)
    e = torch.zeros(0, 4, 2, 64, device="cuda")
    ops.rope_2d(e, torch.zeros(0, 4, 2, dtype=torch.int64, device="cuda"), 100.0, 1.0)  # empty is a no-op


def _attn_ref(q, k, v, scale, qpos=None, kpos=None, mask=None):
    from oracle import siu3r_oracle as O

    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))  # [B,H,N,D]
    if qpos is not None:
        q, k = O.rope2d(q, qpos), O.rope2d(k, kpos)
    a = (q @ k.transpose(-1, -2)) * scale
    if mask is not None:
        a = a.masked_fill(mask[:, None].bool(), float("-inf"))
    o = a.softmax(-1) @ v
    return o.permute(0, 2, 1, 3).flatten(2)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("shape", [(2, 3, 257, 257), (1, 2, 130, 321), (1, 4, 64, 1025)], ids=["self257", "cross", "long"])
def test_attention_rope_d64(mode, shape):
    from oracle import siu3r_oracle as O

    ops = _ops()
    name, adt, split, tol = mode
    B, H, Nq, Nk = shape
    D = 64
    qkv = gen(B, Nq, 3, H, D, seed=27)
    if Nq == Nk:
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q, k, v = qkv[:, :, 0], gen(B, Nk, H, D, seed=28), gen(B, Nk, H, D, seed=29)
    g = torch.Generator().manual_seed(3)
    qpos = torch.randint(0, 34, (B, Nq, 2), generator=g)
    kpos = qpos if Nq == Nk else torch.randint(0, 34, (B, Nk, 2), generator=g)
    cos, sin = O.rope2d_table(34, D)
    ref = _attn_ref(q.to(adt).float(), k.to(adt).float(), v.to(adt).float(), D ** -0.5, qpos, kpos)
    dev = lambda t: t.cuda().to(adt)
    if Nq == Nk:
        packed = dev(qkv)
        qg, kg, vg = packed[:, :, 0], packed[:, :, 1], packed[:, :, 2]
    else:
        qg, kg, vg = dev(q.contiguous()), dev(k), dev(v)
    out = ops.attention(qg, kg, vg, heads=H, head_dim=D, scale=D ** -0.5, rope=(cos.cuda(), sin.cuda()),
                        qpos=qpos.cuda(), kpos=kpos.cuda(), split3=split)
    check(f"attention_rope_d64[{name}] {shape}", out, ref, tol)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
def test_attention_masked_d32(mode):
    ops = _ops()
    name, adt, split, tol = mode
    B, H, Nq, Nk, D = 2, 8, 100, 520, 32
    q, k, v = gen(B, Nq, H, D, seed=30), gen(B, Nk, H, D, seed=31), gen(B, Nk, H, D, seed=32)
    mask = (torch.rand(B, Nq, Nk, generator=torch.Generator().manual_seed(4)) < 0.7)
    mask[:, :, 0] = False  # every row keeps at least one key (the reference re-opens fully blocked rows)
    mask[0, 5, :] = True
    mask[0, 5, 300] = False  # single surviving key far from the first tile
    ref = _attn_ref(q.to(adt).float(), k.to(adt).float(), v.to(adt).float(), D ** -0.5, mask=mask)
    dev = lambda t: t.cuda().to(adt)
    mpad = torch.ones(B, Nq, 576, dtype=torch.uint8)  # row stride padded to a multiple of 64; padding is never read
    mpad[:, :, :Nk] = mask.to(torch.uint8)
    out = ops.attention(dev(q), dev(k), dev(v), heads=H, head_dim=D, scale=D ** -0.5, mask=mpad.cuda(), split3=split)
    check(f"attention_masked_d32[{name}]", out, ref, tol)
    out2 = ops.attention(dev(q), dev(q), dev(q), heads=H, head_dim=D, scale=D ** -0.5, split3=split)
    check(f"attention_self_d32[{name}]", out2, _attn_ref(*(q.to(adt).float(),) * 3, D ** -0.5), tol)


@pytest.mark.parametrize("adt", [torch.float32, torch.bfloat16])
def test_resize_and_pointwise(adt):
    ops = _ops()
    tol = 1e-5 if adt == torch.float32 else 8e-3
    x = gen(2, 32, 9, 13, seed=33)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().to(adt)
    xr = x.to(adt).float()
    for (size, ac) in [((18, 26), True), ((18, 26), False), ((36, 52), False), ((5, 7), False), ((4, 6), False)]:
        out = ops.resize_bilinear(xg, size, ac, out_dtype=torch.float32)
        ref = F.interpolate(xr, size=size, mode="bilinear", align_corners=ac).permute(0, 2, 3, 1)
        check(f"resize {size} align={ac} [{adt}]", out, ref, tol)
    x2 = gen(2, 32, 8, 12, seed=34)
    half = F.interpolate(x2, scale_factor=0.5, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    out = ops.resize_bilinear(x2.permute(0, 2, 3, 1).contiguous().cuda(), (4, 6), False)
    check("resize scale 0.5", out, half, 1e-5)
    add = gen(2, 18, 26, 32, seed=35)
    sc, sh = gen(32, seed=36) + 1.5, gen(32, seed=37)
    out = ops.resize_bilinear(xg, (18, 26), False, addend=add.cuda(), ch_scale=sc.cuda(), ch_shift=sh.cuda(), out_dtype=torch.float32)
    ref = (F.interpolate(xr, size=(18, 26), mode="bilinear", align_corners=False).permute(0, 2, 3, 1) + add) * sc + sh
    check(f"resize+add+affine [{adt}]", out, ref, tol)
    out = ops.affine_add(xg, xg, sc.cuda(), sh.cuda(), out_dtype=torch.float32)
    check(f"affine_add [{adt}]", out, (2 * xr.permute(0, 2, 3, 1)) * sc + sh, tol)
    out = ops.maxpool3x3s2(xg)
    check(f"maxpool [{adt}]", out, F.max_pool2d(xr, 3, 2, 1).permute(0, 2, 3, 1), tol)
    a, b = gen(6, 5, 64, seed=38), gen(5, 64, seed=39)
    check("add broadcast", ops.add(a.cuda(), b.cuda()), a + b, 1e-6)
    gamma, beta = gen(32, seed=40) + 1.0, gen(32, seed=41)
    gn_in = gen(2, 64, 9, 13, seed=42) * 2 + 0.3
    out = ops.groupnorm(gn_in.permute(0, 2, 3, 1).contiguous().cuda().to(adt), F.pad(gamma, (0, 32), value=1.0).cuda(),
                        F.pad(beta, (0, 32)).cuda(), groups=8, relu=True, out_dtype=torch.float32)
    ref = F.relu(F.group_norm(gn_in.to(adt).float(), 8, F.pad(gamma, (0, 32), value=1.0), F.pad(beta, (0, 32)), 1e-5)).permute(0, 2, 3, 1)
    check(f"groupnorm+relu [{adt}]", out, ref, 2e-5 if adt == torch.float32 else 8e-3)


@pytest.mark.parametrize("adt", [torch.float32, torch.bfloat16])
def test_dwconv_gelu(adt):
    ops = _ops()
    B, H, W, Cc = 2, 4, 6, 32
    n = H * W // 4
    x = gen(B, 21 * n, Cc, seed=43)
    w, b = gen(Cc, 1, 3, 3, seed=44), gen(Cc, seed=45)
    xr = x.to(adt).float()
    outs = []
    for (a, e, hh, ww) in ((0, 16 * n, 2 * H, 2 * W), (16 * n, 20 * n, H, W), (20 * n, 21 * n, H // 2, W // 2)):
        t = xr[:, a:e].transpose(1, 2).reshape(B, Cc, hh, ww)
        outs.append(F.conv2d(t, w, b, padding=1, groups=Cc).flatten(2).transpose(1, 2))
    ref = F.gelu(torch.cat(outs, 1))
    w9c = w.reshape(Cc, 9).t().contiguous()
    out = ops.dwconv3x3_gelu(x.cuda().to(adt), w9c.cuda(), b.cuda(), H, W)
    check(f"dwconv3x3_gelu [{adt}]", out, ref, 1e-5 if adt == torch.float32 else 8e-3)


@pytest.mark.parametrize("adt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(16, 64, 1, 4), (8, 32, 3, 4)], ids=["adapter", "pixdec"])
def test_msdeform_sample(adt, cfg):
    from oracle import siu3r_oracle as O

    ops = _ops()
    heads, d, L, P = cfg
    B = 2
    shapes = [(6, 8)] if L == 1 else [(3, 4), (6, 8), (12, 16)]
    S = sum(a * b for a, b in shapes)
    qshapes = [(12, 16), (6, 8), (3, 4)] if L == 1 else shapes
    ref_pts = O.reference_points(qshapes)
    Q = ref_pts.shape[0]
    ref_l = ref_pts[:, None, :].expand(-1, L, -1).contiguous()
    value = gen(B, S, heads * d, seed=46)
    offs = gen(B, Q, heads, L, P, 2, seed=47, scale=3.0)  # several pixels, some samples fall outside
    logits = gen(B, Q, heads, L * P, seed=48, scale=2.0)
    norm = torch.tensor([[s[1], s[0]] for s in shapes], dtype=torch.float32)
    loc = ref_l[None, :, None, :, None, :] + offs / norm[None, None, None, :, None, :]
    aw = logits.softmax(-1).view(B, Q, heads, L, P)
    ref = O.msdeform_core(value.to(adt).float().view(B, S, heads, d), shapes, loc, aw)
    offs_aw = torch.cat([offs.reshape(B, Q, -1), logits.reshape(B, Q, -1)], -1).contiguous()
    out = ops.msdeform_sample(value.cuda().to(adt), offs_aw.cuda(), ref_l.cuda(), shapes, heads, P, torch.float32)
    check(f"msdeform_sample [{adt}] L={L}", out, ref, 2e-5 if adt == torch.float32 else 8e-3)


def test_pts3d_and_gaussian_adapter():
    from oracle import siu3r_oracle as O

    ops = _ops()
    xyz = gen(1000, 3, seed=49, scale=2.0)
    xyz[0] = 0.0  # |xyz| = 0 -> clip(min=1e-8) branch
    d = xyz.norm(dim=-1, keepdim=True)
    ref = xyz / d.clip(min=1e-8) * torch.expm1(d)
    check("pts3d_exp", ops.pts3d_exp_(xyz.cuda().clone()), ref, 2e-6)
    raw = gen(3, 333, 83, seed=50, scale=4.0)
    raw[0, 0, 1:4] = 30.0  # softplus threshold + scale clamp 0.3
    raw[0, 1, 4:8] = 0.0   # zero quaternion -> eps path
    g = O.gaussian_adapter(torch.zeros(3, 333, 3), raw)
    for adt in (torch.float32,):
        out = ops.gaussian_adapter(raw.cuda().to(adt))
        for kk in ("opacities", "scales", "rotations", "harmonics"):
            check(f"gaussian_adapter.{kk}", out[kk], g[kk], 3e-6)
        check("gaussian_adapter.covariances", out["covariances"], g["covariances"], 3e-5)


def test_m2f_attn_mask():
    ops = _ops()
    B, T, IH, IW, Q = 2, 2, 8, 12, 100
    ml = gen(B, Q, T, IH, IW, seed=51, scale=3.0)
    ml[0, 3] = -5.0  # fully blocked row -> must be re-opened (video_seg_decoder.py:1306-1308)
    for size in [(4, 6), (8, 12), (2, 3)]:
        am = F.interpolate(ml.flatten(0, 1), size=size, mode="bilinear", align_corners=False)
        am = am.view(B, Q, T, *size).sigmoid().flatten(2) < 0.5
        am[torch.where(am.sum(-1) == am.shape[-1])] = False
        out = ops.m2f_attn_mask(ml.permute(0, 2, 3, 4, 1).contiguous().cuda(), size)
        nk = am.shape[-1]
        assert out.shape[-1] % 64 == 0 and out.shape[-1] >= nk
        out = out[:, :, :nk]
        mism = (out.cpu().bool() != am).float().mean().item()
        print(f"[parity] m2f_attn_mask {size}: mismatch fraction {mism:.2e}")
        assert mism <= 1e-4  # boolean threshold of an fp32 bilinear sample: ties at |x|<1e-7 only
        assert not out[0, 3].any()


def test_bmm_nt_runtime_operand():
    """the Mask2Former mask product: both operands are activations; the bf16x3 form takes the planes interleaved by split_bf16"""
    ops = _ops()
    a, b = gen(2, 700, 256, seed=60), gen(2, 100, 256, seed=61)
    ref = torch.einsum("zmk,znk->zmn", a, b)
    hi, lo, kpad, x3 = ops.split_bf16(b.view(200, 256).cuda(), True, want_x3=True)
    assert torch.equal(x3[:, :, 0].reshape(200, kpad), hi) and torch.equal(x3[:, :, 1].reshape(200, kpad), lo)
    out3 = ops.bmm_nt(a.cuda(), hi.view(2, 100, kpad), lo.view(2, 100, kpad), 100, 256, b_x3=x3)
    check("bmm_nt[bf16x3, LDS-DMA]", out3, ref, TOL_F32)
    out3r = ops.bmm_nt(a.cuda(), hi.view(2, 100, kpad), lo.view(2, 100, kpad), 100, 256)
    check("bmm_nt[bf16x3, register-staged]", out3r, ref, TOL_F32)
    out1 = ops.bmm_nt(a.cuda().bfloat16(), hi.view(2, 100, kpad), None, 100, 256)
    check("bmm_nt[bf16]", out1, torch.einsum("zmk,znk->zmn", a.bfloat16().float(), hi.view(2, 100, kpad).float().cpu()), TOL_BF16)


def test_split_bf16_exact():
    ops = _ops()
    x = gen(7, 100, seed=52)
    hi, lo, kpad = ops.split_bf16(x.cuda(), True)
    assert kpad == 128 and float(hi[:, 100:].float().abs().max()) == 0.0
    h = x.bfloat16()
    assert torch.equal(hi[:, :100].cpu(), h)
    assert torch.equal(lo[:, :100].cpu(), (x - h.float()).bfloat16())


def test_gemm_wide_tiles_subprocess():
    """The 128x128-tile variants (A/B option SIU3R_GEMM_NARROW_MAX=0, read at library load) against torch:
    dense with a K tail, 3x3 conv (uniform-tap mode) with fused input ReLU, 7x7 stem conv (per-lane tap mode)."""
    import os
    import subprocess
    import sys

    code = r'''
import torch, torch.nn.functional as F
from siu3r_amd import ops
g = torch.Generator().manual_seed(3)
r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1)
a, w, b = r(300, 1096).cuda(), r(200, 1096).cuda() * 0.05, r(200).cuda()
for adt in (torch.bfloat16, torch.float32):
    y = ops.linear(a.to(adt), ops.pack_linear(w, b, False), out_dtype=torch.float32)
    ref = a.to(adt).float() @ w.to(torch.bfloat16).float().t() + b
    assert (y - ref).abs().max() / ref.abs().max() < 1e-2, "dense"
x, cw = r(2, 24, 20, 128).cuda(), r(96, 128, 3, 3).cuda() * 0.05
y = ops.conv2d(x.to(torch.bfloat16), ops.pack_conv(cw, None, False), pad=1, out_dtype=torch.float32, relu_in=True)
ref = F.conv2d(F.relu(x.to(torch.bfloat16).float()).permute(0, 3, 1, 2), cw.to(torch.bfloat16).float(), padding=1).permute(0, 2, 3, 1)
assert (y - ref).abs().max() / ref.abs().max() < 1e-2, "conv3x3"
x, cw = r(1, 40, 36, 8).cuda(), r(80, 8, 7, 7).cuda() * 0.05
y = ops.conv2d(x.to(torch.bfloat16), ops.pack_conv(cw, None, False), pad=3, out_dtype=torch.float32)
ref = F.conv2d(x.to(torch.bfloat16).float().permute(0, 3, 1, 2), cw.to(torch.bfloat16).float(), padding=3).permute(0, 2, 3, 1)
assert (y - ref).abs().max() / ref.abs().max() < 1e-2, "conv7x7"
print("WIDE_OK")
'''
    env = dict(os.environ, SIU3R_GEMM_NARROW_MAX="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert "WIDE_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("split", [False, True], ids=["bf16", "bf16x3"])
def test_batch_strided_views_are_read_in_place(split):
    """The ViT-Adapter's maps are slices of per-item token sequences (vit_adapter.py:393-433): resize / affine / conv-transpose read the
    batch-strided views directly and must give what they give on dense copies."""
    ops = _ops()
    Z, h, w, C = 2, 8, 6, 64
    n2, n3, n4 = 4 * h * w, h * w, h * w // 4
    seq = gen(Z, n2 + n3 + n4, C, seed=101).cuda()
    tok = gen(Z, h * w + 1, C, seed=102).cuda()          # patch tokens + one trailing token per item
    c2, c3, c4 = seq[:, :n2].view(Z, 2 * h, 2 * w, C), seq[:, n2:n2 + n3].view(Z, h, w, C), seq[:, n2 + n3:].view(Z, h // 2, w // 2, C)
    x = tok[:, :-1].view(Z, h, w, C)
    assert not c3.is_contiguous() and not x.is_contiguous()
    sc, sh = gen(C, seed=103).cuda() + 1.5, gen(C, seed=104).cuda()
    for (src, size, add) in [(x, (2 * h, 2 * w), c2), (x, (h // 2, w // 2), c4), (x, (4 * h, 4 * w), None)]:
        got = ops.resize_bilinear(src, size, False, addend=add, ch_scale=sc, ch_shift=sh)
        ref = ops.resize_bilinear(src.contiguous(), size, False, addend=None if add is None else add.contiguous(), ch_scale=sc, ch_shift=sh)
        assert torch.equal(got, ref)
    assert torch.equal(ops.affine_add(x, c3, sc, sh), ops.affine_add(x.contiguous(), c3.contiguous(), sc, sh))
    wt, bt = gen(C, 32, 2, 2, seed=105, scale=0.1).cuda(), gen(32, seed=106).cuda()
    pw = ops.pack_conv_transpose(wt, bt, split)
    a = c2 if split else c2.to(torch.bfloat16)
    res = gen(Z, 4 * h, 4 * w, 32, seed=107).cuda()
    got = ops.conv_transpose2d(a, pw, residual=res)
    ref = ops.conv_transpose2d(a.contiguous(), pw, residual=res)
    assert torch.equal(got, ref)
    check("conv_transpose on a strided view", got, F.conv_transpose2d(a.float().permute(0, 3, 1, 2), wt if split else wt.to(torch.bfloat16).float(), bt, stride=2).permute(0, 2, 3, 1) + res,
          TOL_F32 if split else TOL_BF16)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("case", [(2, 3, 257, 300, 64, False), (2, 8, 100, 520, 32, True), (1, 2, 128, 64, 64, True), (1, 8, 100, 2100, 32, True), (2, 4, 37, 1100, 64, False)],
                         ids=["d64", "d32mask", "d64mask1tile", "d32mask_splitkv", "d64_splitkv"])
def test_attention_peaky(mode, case):
    """Concentrated softmax (|logit| up to ~40): with near-uniform weights a wrong key<->value pairing, a wrong row
    maximum or a wrong normaliser hides inside the bf16 tolerance; here every one of them is an O(1) error.
    Also covers a structured one-hot case whose answer is exact (weights one-hot, values = key index)."""
    ops = _ops()
    name, adt, split, tol = mode
    B, H, Nq, Nk, D, use_mask = case
    q, k, v = gen(B, Nq, H, D, seed=40, scale=6.0), gen(B, Nk, H, D, seed=41, scale=6.0), gen(B, Nk, H, D, seed=42)
    mask = mpad = None
    if use_mask:
        mask = torch.rand(B, Nq, Nk, generator=torch.Generator().manual_seed(5)) < 0.7
        mask[:, :, 0] = False
        ld = (Nk + 63) // 64 * 64
        mpad = torch.ones(B, Nq, ld, dtype=torch.uint8)
        mpad[:, :, :Nk] = mask.to(torch.uint8)
        mpad = mpad.cuda()
    dev = lambda t: t.cuda().to(adt)
    out = ops.attention(dev(q), dev(k), dev(v), heads=H, head_dim=D, scale=D ** -0.5, mask=mpad, split3=split)
    # without the bf16x3 split the kernel rounds its operands to bf16: the reference sees the same rounded inputs
    rdt = adt if split else torch.bfloat16
    ref = _attn_ref(q.to(rdt).float(), k.to(rdt).float(), v.to(rdt).float(), D ** -0.5, mask=mask)
    check(f"attention_peaky[{name},{case}]", out, ref, tol)
    # one-hot weights: query i selects key i % Nk exactly; value column 0 = key index, column 1 = 1
    q1, k1, v1 = torch.zeros(1, Nq, 1, D), torch.zeros(1, min(Nk, D), 1, D), torch.zeros(1, min(Nk, D), 1, D)
    n1 = k1.shape[1]
    for i in range(Nq):
        q1[0, i, 0, i % n1] = 8.0
    for j in range(n1):
        k1[0, j, 0, j] = 8.0
    v1[0, :, 0, 0] = torch.arange(n1).float()
    v1[0, :, 0, 1] = 1.0
    o1 = ops.attention(dev(q1), dev(k1), dev(v1), heads=1, head_dim=D, scale=1.0, split3=split).float().cpu()[0]
    assert torch.equal(o1[:, 0].round().long(), torch.arange(Nq) % n1), "one-hot attention selected the wrong keys"
    assert float((o1[:, 1] - 1).abs().max()) < 1e-2


@pytest.mark.parametrize("tile", [-1, 2, 3], ids=["t128x64", "pp256x128", "pp128x128"])
def test_gemm_split_k_stress(tile):
    """Forced 8-way split-K with MANY tiles (every XCD holds slices of several tiles, tickets and slabs of 64+ tiles live at once),
    launched back to back on one stream with alternating inputs: every launch must be bit-identical to the first launch of its input
    (a ticket that was not reset, a slab read before its writer's data left the XCD's L2, or a slab shared by two tiles would show
    as a difference here), and equal to the unsplit kernel up to fp32 summation order."""
    ops = _ops()
    M, N, K = 1024, 1024, 4096
    w = gen(N, K, seed=81, scale=0.05)
    pw = ops.pack_linear(w.cuda(), gen(N, seed=82).cuda(), True)
    xs = [gen(M, K, seed=83 + i).cuda() for i in range(2)]
    log = []
    ops.gemm_tune(0, tile)
    ops.set_plan_log(log)
    try:
        first = [ops.linear(x, pw, out_dtype=torch.float32, splitk=8) for x in xs]
        assert log[-1].tile_cfg == tile and log[-1].splitk == 8, (log[-1].tile_cfg, log[-1].splitk)
        outs = [ops.linear(xs[i % 2], pw, out_dtype=torch.float32, splitk=8) for i in range(40)]
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, first[i % 2]), f"launch {i} differs from the first launch of the same input"
        ops.gemm_tune(2, 1)
        try:
            whole = ops.linear(xs[0], pw, out_dtype=torch.float32)
        finally:
            ops.gemm_tune(2, 0)
        assert float((first[0] - whole).abs().max()) <= 1e-4 * float(whole.abs().max())
    finally:
        ops.set_plan_log(None)
        ops.gemm_tune(0, 0)


@pytest.mark.parametrize("mode", [MODES[0], MODES[2]], ids=["bf16", "bf16x3"])
def test_gemm_layernorm_fold_outliers(mode):
    """The folded LayerNorm computes rstd * (x W'^T - mean * c1) + c2: with a large row mean the two products cancel.  Rows with a
    mean of 50 standard deviations and rows with single 1000-sigma channels (the massive activations ViT blocks develop) must stay
    inside half of the north-star bar in bf16x3: the split product's error is ~2^-17 of |x| |W|, i.e. it grows with |mean| / sigma
    (measured 2.1e-4 at 50 sigma, against 1e-5 for centred rows); a checkpoint whose rows sit further out than that should take
    the unfolded LayerNorm (SIU3R_NO_LNFOLD=1)."""
    ops = _ops()
    name, adt, split, tol = mode
    M, C, N = 300, 1024, 512
    x = gen(M, C, seed=91)
    x[:100] += 30.0                      # mean ~ 50 sigma
    x[100:200, 7] = 600.0                # one massive channel
    x[200:, 511] = -400.0
    x[200:, 512] = 400.0
    gamma, beta = gen(C, seed=92) * 0.5 + 1.0, gen(C, seed=93)
    w, b = gen(N, C, seed=94, scale=0.05), gen(N, seed=95)
    # the rows and their statistics come out of a producing GEMM's epilogue, as in the network: 0 @ W + 0 + residual = x exactly
    xg = torch.empty(M, C, device="cuda")
    st = ops.RowStats(xg)
    xb = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    ops.linear(torch.zeros(M, 64, device="cuda").to(adt), ops.pack_linear(gen(C, 64, seed=96).cuda(), torch.zeros(C).cuda(), split), residual=x.cuda(), out=xg,
               stats_out=st, aux_out=xb)
    assert torch.equal(xg.cpu(), x)
    pw = ops.pack_linear_ln(w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), split)
    pw.meta["ln_eps"] = 1e-6
    out = ops.linear(xg if split else xb, pw, out_dtype=torch.float32, ln=st)
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-6) @ w.t() + b
    # bf16 mode multiplies the bf16-rounded UN-normalised rows: its error scales with |x| / sigma, not with the output (documented)
    check(f"layernorm_fold_outliers[{name}]", out, ref, 5e-4 if split else 0.35)


PIPE_CASES = [(2, 16, 1025, 1025, 1.0), (1, 4, 1025, 2050, 4.0), (1, 2, 300, 200, 6.0), (1, 3, 129, 70, 6.0), (2, 4, 128, 64, 1.0), (1, 5, 1000, 1025, 2.0)]


def _attention_pipe_cases(split):
    """The 8-wave pipelined ViT attention (attention_pipe.hip): the pair shape with its intrinsics-token side path (Nq = 128 n + 1),
    a multi-view cross-attention (keys of two other views), ragged query / key tiles, a single key tile, and Nq with no side path
    next to a ragged key tail.  Strided q / k / v views of one packed projection output, as the model passes them."""
    ops = _ops()
    adt, tol = (torch.float32, TOL_F32) if split else (torch.bfloat16, TOL_BF16)
    D = 64
    for (B, H, Nq, Nk, scale) in PIPE_CASES:
        if Nq == Nk:
            qkv = gen(B, Nq, 3, H, D, seed=50, scale=scale).cuda().to(adt)
            q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        else:
            q = gen(B, Nq, H, D, seed=51, scale=scale).cuda().to(adt)
            kv = gen(B, Nk, 2, H, D, seed=52, scale=scale).cuda().to(adt)
            k, v = kv[:, :, 0], kv[:, :, 1]
        out = ops.attention(q, k, v, heads=H, head_dim=D, scale=D ** -0.5, split3=split)
        ref = _attn_ref(q.float().cpu(), k.float().cpu(), v.float().cpu(), D ** -0.5)
        check(f"attention_pipe[{'bf16x3' if split else 'bf16'}] {(B, H, Nq, Nk, scale)}", out, ref, tol)
        # the side-path row on its own: it is 1 of Nq rows, a wrong merge would hide in a max-norm over the whole tensor
        if Nq % 128 == 1:
            check("  intrinsics-token row", out[:, -1], ref[:, -1], tol)


def test_attention_pipe_bf16x3():
    _attention_pipe_cases(True)


def test_attention_pipe_bf16():
    """the same kernel on bf16 tensors (opt-in: SIU3R_ATTN_PIPE_BF16=1, read when the library is first used)"""
    code = "import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); import test_kernels_gpu as T; T._attention_pipe_cases(False); print('PIPE_BF16_OK')"
    env = dict(os.environ, SIU3R_ATTN_PIPE_BF16="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert "PIPE_BF16_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.parametrize("mode", [MODES[0], MODES[2]], ids=["bf16", "bf16x3"])
@pytest.mark.parametrize("C", [192, 200, 1024], ids=["C192", "C200_ragged", "C1024"])
def test_gemm_layernorm_fold(mode, C):
    """LayerNorm folded into the consuming GEMM (reference croco/blocks.py:127-130: x + f(LN(x))): the producing GEMM's epilogue emits
    per-row (mean, M2) partials of its 64-column tiles (+ a bf16 copy of the row), the consumer multiplies the UN-normalised rows by
    W diag(gamma) and applies rstd (acc - mean c1) + c2 in its epilogue.  Rows carry a large common offset (mean >> std) to exercise
    the Welford merge."""
    ops = _ops()
    name, adt, split, tol = mode
    Z, Nt, K0, N2 = 2, 37, 96, 136
    a = gen(Z, Nt, K0, seed=11)
    wp, bp = gen(C, K0, seed=12, scale=0.5), gen(C, seed=13) + 3.0          # producer: x = a Wp^T + bp + res   (row mean ~ 3)
    res = gen(Z, Nt, C, seed=14)
    gamma, beta = 1.0 + 0.3 * gen(C, seed=15), gen(C, seed=16)
    w2, b2 = gen(N2, C, seed=17, scale=0.2), gen(N2, seed=18)
    x_ref = a.to(adt).float() @ wp.t() + bp + res
    x = torch.empty(Z, Nt, C, device="cuda")
    xb = torch.empty(Z, Nt, C, device="cuda", dtype=torch.bfloat16)
    st = ops.RowStats(x)
    ops.linear(a.cuda().to(adt), ops.pack_linear(wp.cuda(), bp.cuda(), split), residual=res.cuda(), out=x, stats_out=st, aux_out=xb)
    check(f"ln_fold[{name}] producer", x, x_ref, tol)
    assert torch.equal(xb.float().cpu(), x.cpu().to(torch.bfloat16).float())
    # merged statistics == the row statistics
    T = st.tiles
    cnt = torch.tensor([64] * (T - 1) + [C - 64 * (T - 1)], dtype=torch.float32)
    part = st.buf.cpu().view(Z * Nt, T, 2)
    mean = (part[:, :, 0] * cnt).sum(1) / C
    m2 = part[:, :, 1].sum(1) + (cnt * (part[:, :, 0] - mean[:, None]) ** 2).sum(1)
    xr = x.cpu().view(-1, C)
    assert float((mean - xr.mean(1)).abs().max()) <= 1e-5 and float((m2 / C - xr.var(1, unbiased=False)).abs().max()) <= 1e-4 * float(xr.var(1, unbiased=False).max())
    pw = ops.pack_linear_ln(w2.cuda(), b2.cuda(), gamma.cuda(), beta.cuda(), split)
    pw.meta["ln_eps"] = 1e-6
    A = x if split else xb
    ref = F.layer_norm(x.cpu(), (C,), gamma, beta, 1e-6).to(adt).float() @ w2.t() + b2
    out = ops.linear(A, pw, out_dtype=torch.float32, ln=st)
    check(f"ln_fold[{name}] consumer", out, ref, max(tol, 2e-2 if not split else 0))   # bf16: x is rounded BEFORE the normalisation
    out_v = ops.linear(A[:, :-1], pw, out_dtype=torch.float32, ln=st)                    # strided view (intrinsics token stripped)
    check(f"ln_fold[{name}] consumer on a view", out_v, ref.view(Z, Nt, N2)[:, :-1], max(tol, 2e-2 if not split else 0))


@pytest.mark.parametrize("mode", [MODES[0], MODES[2]], ids=["bf16", "bf16x3"])
@pytest.mark.parametrize("rows", [1, 2, 4, 5], ids=["stub1", "stub2", "stub4", "stub5"])
def test_gemm_stub_row_tiles(mode, rows):
    """M = k * 128 + r with a few valid rows in the last row tile (the 2 x 1025 token layout: r = 2, one decoder side: r = 1; rows beyond M are
    out-of-range fetches) through every epilogue feature such a tile can meet: bias + GELU + residual, folded LayerNorm with row statistics
    out, batched launch, split-K (few tiles, long K)."""
    ops = _ops()
    name, adt, split, tol = mode
    M, N, K = 256 + rows, 200, 1024
    a, w, b, r = gen(M, K, seed=71), gen(N, K, seed=72, scale=0.1), gen(N, seed=73), gen(M, N, seed=74)
    pw = ops.pack_linear(w.cuda(), b.cuda(), split)
    out = ops.linear(a.cuda().to(adt), pw, out_dtype=torch.float32, act=ops.ACT_GELU, residual=r.cuda())
    check(f"stub[{name}] r={rows} gelu+residual", out, F.gelu(a.to(adt).float() @ w.t() + b) + r, tol)
    # long K and few tiles -> split-K slices, each with its own FMA-path partial
    M2, N2, K2 = 128 + rows, 128, 4096
    a2, w2 = gen(M2, K2, seed=75), gen(N2, K2, seed=76, scale=0.05)
    out2 = ops.linear(a2.cuda().to(adt), ops.pack_linear(w2.cuda(), None, split), out_dtype=torch.float32)
    check(f"stub[{name}] r={rows} split-K", out2, a2.to(adt).float() @ w2.t(), tol)
    # batched [Z, 128 + rows, C] tokens: producer with statistics out, consumer with the folded LayerNorm
    Z, Nt, C = 2, 128 + rows, 192
    x0 = gen(Z, Nt, 96, seed=77)
    wp, bp = gen(C, 96, seed=78, scale=0.5), gen(C, seed=79) + 2.0
    x = torch.empty(Z, Nt, C, device="cuda")
    xb = torch.empty(Z, Nt, C, device="cuda", dtype=torch.bfloat16)
    st = ops.RowStats(x)
    ops.linear(x0.cuda().to(adt), ops.pack_linear(wp.cuda(), bp.cuda(), split), out=x, stats_out=st, aux_out=xb)
    check(f"stub[{name}] r={rows} producer", x, x0.to(adt).float() @ wp.t() + bp, tol)
    gamma, beta = 1.0 + 0.3 * gen(C, seed=80), gen(C, seed=81)
    w3, b3 = gen(136, C, seed=82, scale=0.2), gen(136, seed=83)
    pw3 = ops.pack_linear_ln(w3.cuda(), b3.cuda(), gamma.cuda(), beta.cuda(), split)
    pw3.meta["ln_eps"] = 1e-6
    ref = F.layer_norm(x.cpu(), (C,), gamma, beta, 1e-6).to(adt).float() @ w3.t() + b3
    out3 = ops.linear(x if split else xb, pw3, out_dtype=torch.float32, ln=st)
    check(f"stub[{name}] r={rows} folded LayerNorm", out3, ref, max(tol, 2e-2 if not split else 0))


@pytest.mark.parametrize("mode", [MODES[0], MODES[2]], ids=["bf16", "bf16x3"])
@pytest.mark.parametrize("rows", [1, 2, 4, 7], ids=["r1", "r2", "r4", "r7"])
def test_gemm_skinny_remainder_rows(mode, rows):
    """The last rows of a dense problem on the skinny launch (forced: siu3r_gemm_tune key 4), <= 4 rows on the matrix-vector kernel and
    more on the MFMA one, through every epilogue feature the network sends there: bias + GELU + residual, K that is not a multiple of
    the chunk (K = 1000 -> kpad 1024, K = 4096 = 16 chunks), N that is not a multiple of 64, a batched launch with folded LayerNorm +
    statistics out, RoPE on the output.  The tiled rows of the same launch must be untouched by the remainder launch."""
    from oracle import siu3r_oracle as O

    ops = _ops()
    name, adt, split, tol = mode
    ops.gemm_tune(4, 1)
    ops.gemm_tune(0, 2)  # the 256 x 128 ping-pong tile (the bf16 mode would otherwise stay on the 128 x 64 family, which has no skinny launch)
    try:
        log = []
        ops.set_plan_log(log)
        for (M0, N, K) in ((256, 200, 1000), (256, 128, 4096), (512, 1024, 768)):
            M = M0 + rows
            a, w, b, r = gen(M, K, seed=91), gen(N, K, seed=92, scale=0.1), gen(N, seed=93), gen(M, N, seed=94)
            pw = ops.pack_linear(w.cuda(), b.cuda(), split)
            out = ops.linear(a.cuda().to(adt), pw, out_dtype=torch.float32, act=ops.ACT_GELU, residual=r.cuda())
            check(f"skinny[{name}] r={rows} {M}x{N}x{K} gelu+residual", out, F.gelu(a.to(adt).float() @ w.t() + b) + r, tol)
        assert all(pl.skinny_rows == rows for pl in log), [(pl.tile_cfg, pl.skinny_rows) for pl in log]
        ops.set_plan_log(None)
        # batched tokens [Z, 256 + rows, C]: producer with statistics out, consumer with the folded LayerNorm and RoPE
        Z, Nt, C, H, D = 2, 256 + rows, 192, 2, 64
        x0 = gen(Z, Nt, 96, seed=95)
        wp, bp = gen(C, 96, seed=96, scale=0.5), gen(C, seed=97) + 2.0
        x = torch.empty(Z, Nt, C, device="cuda")
        xb = torch.empty(Z, Nt, C, device="cuda", dtype=torch.bfloat16)
        st = ops.RowStats(x)
        ops.linear(x0.cuda().to(adt), ops.pack_linear(wp.cuda(), bp.cuda(), split), out=x, stats_out=st, aux_out=xb)
        check(f"skinny[{name}] r={rows} producer", x, x0.to(adt).float() @ wp.t() + bp, tol)
        stats_ref = x.view(Z * Nt, C // 64, 64)
        check(f"skinny[{name}] r={rows} row statistics (means)", st.buf[..., 0], stats_ref.mean(-1), 1e-5)
        gamma, beta = 1.0 + 0.3 * gen(C, seed=98), gen(C, seed=99)
        N3 = H * D
        w3, b3 = gen(N3, C, seed=100, scale=0.2), gen(N3, seed=101)
        pw3 = ops.pack_linear_ln(w3.cuda(), b3.cuda(), gamma.cuda(), beta.cuda(), split)
        pw3.meta["ln_eps"] = 1e-6
        pos = torch.randint(0, 9, (Z, Nt, 2), generator=torch.Generator().manual_seed(5))
        cos, sin = O.rope2d_table(9, D)
        ref = F.layer_norm(x.cpu(), (C,), gamma, beta, 1e-6).to(adt).float() @ w3.t() + b3
        ref = O.rope2d(ref.view(Z, Nt, H, D).transpose(1, 2), pos).transpose(1, 2).reshape(Z, Nt, N3)
        out3 = ops.linear(x if split else xb, pw3, out_dtype=torch.float32, ln=st, rope=(cos.cuda(), sin.cuda(), pos.cuda(), N3))
        check(f"skinny[{name}] r={rows} folded LayerNorm + RoPE", out3, ref, max(tol, 2e-2 if not split else 0))
    finally:
        ops.set_plan_log(None)
        ops.gemm_tune(4, 0)
        ops.gemm_tune(0, 0)


def test_explicit_splitk_is_honoured_or_refused():
    """siu3r_gemm_params.splitk > 1 means exactly that many slices: a kernel that cannot split (the register-staged 128 x 128 family
    is reached by no public path any more; the 128 x 64 LDS-DMA kernel splits) either runs them or the launch fails loudly."""
    ops = _ops()
    a, w = gen(256, 2048, seed=5), gen(128, 2048, seed=6, scale=0.05)
    pw = ops.pack_linear(w.cuda(), None, True)
    log = []
    ops.set_plan_log(log)
    try:
        out = ops.linear(a.cuda(), pw, out_dtype=torch.float32, splitk=4)
    finally:
        ops.set_plan_log(None)
    assert log[-1].splitk == 4
    check("explicit split-K 4", out, a @ w.t(), 2e-4)
    with pytest.raises(RuntimeError, match="splitk"):
        ops.linear(a.cuda(), pw, out_dtype=torch.float32, splitk=128)  # more slices than kpad / 64 allows


@pytest.mark.parametrize("mode", [MODES[0], MODES[2]], ids=["bf16", "bf16x3"])
def test_gemm_grouped_two_sides(mode):
    """Two weight sets in one launch (blockIdx.z = b * 2 + side), incl. the flipped read (side s multiplies the OTHER side's rows:
    the cross-attention memory of a decoder block), folded LayerNorm statistics following the flip, RoPE on the output."""
    from oracle import siu3r_oracle as O

    ops = _ops()
    name, adt, split, tol = mode
    B, G, M, C, H, D = 2, 2, 21, 128, 2, 64
    N = H * D
    x = gen(B, G, M, C, seed=31) + 1.5
    ws = [gen(N, C, seed=32 + g, scale=0.3) for g in range(G)]
    bs = [gen(N, seed=36 + g) for g in range(G)]
    gm, bt = [1 + 0.2 * gen(C, seed=40 + g) for g in range(G)], [gen(C, seed=44 + g) for g in range(G)]
    xg = x.cuda()
    # statistics of x via an identity-free route: produce x with a GEMM so that stats / bf16 copy exist
    eye = torch.eye(C)
    xx = torch.empty(B, G, M, C, device="cuda")
    xb = torch.empty(B, G, M, C, device="cuda", dtype=torch.bfloat16)
    st = ops.RowStats(xx)
    ops.linear(torch.zeros(B * G * M, C, device="cuda").to(adt), ops.pack_linear(eye.cuda(), None, split), residual=xg.view(-1, C), out=xx.view(-1, C), stats_out=st, aux_out=xb.view(-1, C))
    assert torch.equal(xx.cpu(), x)
    pos = torch.randint(0, 9, (B * G, M, 2), generator=torch.Generator().manual_seed(3))
    cos, sin = O.rope2d_table(9, D)
    pw = ops.stack_packed([ops.pack_linear_ln(ws[g].cuda(), bs[g].cuda(), gm[g].cuda(), bt[g].cuda(), split) for g in range(G)])
    A = xx if split else xb
    for flip in (False, True):
        out = ops.linear_grouped(A, pw, out_dtype=torch.float32, ln=st, flip=flip, rope=(cos.cuda(), sin.cuda(), pos.cuda(), N))
        ref = torch.empty(B, G, M, N)
        for b in range(B):
            for g in range(G):
                src = x[b, G - 1 - g] if flip else x[b, g]
                y = F.layer_norm(src, (C,), gm[g], bt[g], 1e-6).to(adt).float() @ ws[g].t() + bs[g]
                ref[b, g] = O.rope2d(y.view(1, M, H, D).permute(0, 2, 1, 3), pos[b * G + g][None]).permute(0, 2, 1, 3).reshape(M, N)
        check(f"grouped[{name}] flip={flip}", out, ref, max(tol, 2e-2 if not split else 0))
    # plain grouped product with bias, residual, statistics out
    pw2 = ops.stack_packed([ops.pack_linear(ws[g].cuda(), bs[g].cuda(), split) for g in range(G)])
    r = gen(B, G, M, N, seed=50)
    o2 = torch.empty(B, G, M, N, device="cuda")
    st2 = ops.RowStats(o2)
    ops.linear_grouped(xx.to(adt), pw2, residual=r.cuda(), out=o2, stats_out=st2)
    ref2 = torch.stack([torch.stack([x[b, g].to(adt).float() @ ws[g].t() + bs[g] for g in range(G)]) for b in range(B)]) + r
    check(f"grouped[{name}] bias+residual", o2, ref2, tol)
    part = st2.buf.cpu().view(B * G * M, 2, 2)
    assert float((part[:, :, 0].mean(1) - o2.cpu().view(-1, N).mean(1)).abs().max()) <= 1e-5


@pytest.mark.parametrize("tile", [-1, 2, 3], ids=["t128x64", "pp256x128", "pp128x128"])
@pytest.mark.parametrize("mode", [MODES[0], MODES[2]], ids=["bf16", "bf16x3"])
def test_gemm_split_k(mode, tile):
    """Few tiles + long K -> K is cut over several workgroups (slabs + ticket, slices summed in order): same values as the unsplit
    kernel up to fp32 summation order, bit-identical run to run (the tickets reset themselves: the second launch reuses the stream's
    workspace), every epilogue feature still applied once (bias, GELU, residual, row statistics), dense and 3x3-conv addressing;
    for the 128 x 64 kernels and the ping-pong tiles."""
    ops = _ops()
    name, adt, split, tol = mode
    M, N, K = 200, 136, 4608
    a, w, b, r = gen(M, K, seed=71), gen(N, K, seed=72, scale=0.05), gen(N, seed=73), gen(M, N, seed=74)
    pw = ops.pack_linear(w.cuda(), b.cuda(), split)
    ref = F.gelu(a.to(adt).float() @ w.t() + b) + r
    log = []
    ops.gemm_tune(0, tile)
    ops.set_plan_log(log)
    try:
        out = torch.empty(M, N, device="cuda")
        st = ops.RowStats(out)
        ops.linear(a.cuda().to(adt), pw, act=ops.ACT_GELU, residual=r.cuda(), out=out, stats_out=st)
        assert log[-1].tile_cfg == tile and log[-1].splitk >= 2, f"this shape is meant to take the split-K path ({log[-1].tile_cfg}, {log[-1].splitk})"
        check(f"split_k[{name}] dense gelu+residual", out, ref, tol)
        out2 = ops.linear(a.cuda().to(adt), pw, out_dtype=torch.float32, act=ops.ACT_GELU, residual=r.cuda())
        assert torch.equal(out, out2), "split-K must be deterministic"
        part = st.buf.cpu()
        cnt = torch.tensor([64, 64, 8.0])
        assert float(((part[:, :, 0] * cnt).sum(1) / N - out.cpu().mean(1)).abs().max()) <= 1e-5
        ops.gemm_tune(2, 1)
        try:
            whole = ops.linear(a.cuda().to(adt), pw, out_dtype=torch.float32, act=ops.ACT_GELU, residual=r.cuda())
            assert log[-1].splitk == 1
        finally:
            ops.gemm_tune(2, 0)
        assert float((out - whole).abs().max()) <= 1e-4 * float(whole.abs().max())
        # 3x3 convolution, 256 input channels, small image: 8 tiles, K = 2304
        x = gen(2, 256, 16, 16, seed=75)
        wc, bc = gen(96, 256, 3, 3, seed=76, scale=0.05), gen(96, seed=77)
        pc = ops.pack_conv(wc.cuda(), bc.cuda(), split)
        xg = x.permute(0, 2, 3, 1).contiguous().cuda().to(adt)
        refc = F.conv2d(F.relu(x.to(adt).float()), wc, bc, padding=1).permute(0, 2, 3, 1)
        outc = ops.conv2d(xg, pc, pad=1, out_dtype=torch.float32, relu_in=True)
        check(f"split_k[{name}] conv3x3 relu_in", outc, refc, tol)
        assert log[-1].splitk >= 2
    finally:
        ops.set_plan_log(None)
        ops.gemm_tune(0, 0)



def _planes_reference(x):
    """the two bf16 planes a bf16x3 product multiplies, in the c_x3 layout [rows, C / 32, (hi 32 | lo 32)] as int16 bit patterns"""
    x = x.detach().float().cpu().contiguous()
    bits = x.view(torch.int32)
    hi_f = (bits & -65536).view(torch.float32)
    hi = (bits >> 16).to(torch.int16)
    lo = (x - hi_f).to(torch.bfloat16).view(torch.int16)
    R, C = x.shape
    return torch.stack((hi.view(R, C // 32, 32), lo.view(R, C // 32, 32)), dim=2).reshape(R, C * 2)


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_gemm_presplit_planes(cfg):
    """Pre-split bf16x3 activations (siu3r_gemm_params.c_x3 / a_x3, ops.Planes) on every ping-pong tile: the producer's planes are the
    exact hi / lo bit patterns of its fp32 output (tile rows, remainder rows, split-K launches alike); a consumer that reads them --
    plain, with a folded LayerNorm, on the long-K matrix-vector remainder rows, from a planes-only GELU output -- returns the SAME BITS
    as with the fp32 operand; and asking for planes where the plan cannot serve them fails loudly at the C ABI."""
    import ctypes as C

    from siu3r_amd import _lib

    ops = _ops()
    ops.gemm_tune(0, cfg)
    try:
        M, C0, C1 = 2050, 1024, 4096
        a = gen(M, C0, seed=201).cuda()
        r = gen(M, C0, seed=202).cuda()
        pw = ops.pack_linear(gen(C0, C0, seed=203, scale=0.05).cuda(), gen(C0, seed=204).cuda(), True)
        x1 = torch.empty(M, C0, device="cuda")
        st = ops.RowStats(x1)
        xp = ops.Planes(x1)
        log = []
        ops.set_plan_log(log)
        ops.linear(a, pw, residual=r, out=x1, stats_out=st, planes_out=xp)
        ops.set_plan_log(None)
        assert xp.valid and not xp.only and log[-1].tile_cfg == cfg and log[-1].c_x3_ok == 1
        check(f"presplit[{cfg}] producer (fp32 output unchanged)", x1, a.cpu() @ gen(C0, C0, seed=203, scale=0.05).t() + gen(C0, seed=204) + r.cpu(), TOL_F32)
        got = xp.t.cpu().contiguous().view(torch.int16)
        assert torch.equal(got, _planes_reference(x1)), "planes are not the hi / lo bit patterns of the fp32 output"
        # consumers: folded LayerNorm + GELU (planes-only output) -> long-K GEMM with residual; each against the same launch on fp32 operands
        gamma, beta = 1.0 + 0.3 * gen(C0, seed=205), gen(C0, seed=206)
        w1 = ops.pack_linear_ln(gen(C1, C0, seed=207, scale=0.05).cuda(), gen(C1, seed=208).cuda(), gamma.cuda(), beta.cuda(), True)
        w1.meta["ln_eps"] = 1e-6
        w2 = ops.pack_linear(gen(C0, C1, seed=209, scale=0.05).cuda(), gen(C0, seed=210).cuda(), True)
        h_ref = ops.linear(x1, w1, act=ops.ACT_GELU, ln=st)
        h_a = ops.linear(x1, w1, act=ops.ACT_GELU, ln=st, a_planes=xp)
        assert torch.equal(h_ref, h_a), "a consumer of the planes must return the bits of the fp32-operand launch"
        h = torch.empty(M, C1, device="cuda")
        hp = ops.Planes(h, storage=h)
        pl2 = ops.linear(h, w2, residual=x1, dry_run=True)
        assert pl2.a_x3_ok == 1 and pl2.tile_cfg == cfg
        ops.linear(x1, w1, act=ops.ACT_GELU, ln=st, a_planes=xp, out=h, planes_out=hp, planes_only=True)
        assert hp.valid and hp.only
        assert torch.equal(h.cpu().contiguous().view(torch.int16), _planes_reference(h_ref)), "planes-only output"
        y_ref = ops.linear(h_ref, w2, residual=x1)
        log = []
        ops.set_plan_log(log)
        y = ops.linear(h, w2, residual=x1, a_planes=hp)
        ops.set_plan_log(None)
        assert log[-1].kernel.count(b",") == 6 and log[-1].kernel.endswith(b", true>"), log[-1].kernel  # the pre-split instantiation (7th template argument) ran
        assert torch.equal(y, y_ref), f"long-K consumer (skinny_rows {log[-1].skinny_rows}, splitk {log[-1].splitk}): max diff {(y - y_ref).abs().max().item():.3e}"
        # a forced split-K producer and consumer
        x2 = torch.empty(M, C0, device="cuda")
        xp2 = ops.Planes(x2)
        ops.linear(h, w2, residual=x1, a_planes=hp, out=x2, planes_out=xp2, splitk=2)
        y_ref2 = ops.linear(h_ref, w2, residual=x1, splitk=2)
        assert xp2.valid and torch.equal(x2, y_ref2) and torch.equal(xp2.t.cpu().contiguous().view(torch.int16), _planes_reference(y_ref2))
        check(f"presplit[{cfg}] chain vs fp32 torch", y_ref2, F.gelu(F.layer_norm(x1.cpu(), (C0,), gamma, beta, 1e-6) @ gen(C1, C0, seed=207, scale=0.05).t() + gen(C1, seed=208))
              @ gen(C0, C1, seed=209, scale=0.05).t() + gen(C0, seed=210) + x1.cpu(), TOL_F32)
        # convolutions: a 3 x 3 convolution writes its ReLU'd map as planes only, the next one gathers its taps from the planes
        # (padding taps, a batch of two images, cin = 64 = two segments per pixel)
        xin = gen(2, 24, 40, 64, seed=220).cuda()
        c1 = ops.pack_conv(gen(64, 64, 3, 3, seed=221, scale=0.1).cuda(), gen(64, seed=222).cuda(), True)
        c2 = ops.pack_conv(gen(128, 64, 3, 3, seed=223, scale=0.1).cuda(), gen(128, seed=224).cuda(), True)
        o_ref = ops.conv2d(xin, c1, pad=1, act=ops.ACT_RELU, relu_in=True)
        y_ref = ops.conv2d(o_ref, c2, pad=1)
        o = torch.empty_like(o_ref)
        op = ops.Planes(o, storage=o)
        assert ops.conv2d(o, c2, pad=1, dry_run=True).a_x3_ok == 1
        ops.conv2d(xin, c1, pad=1, act=ops.ACT_RELU, relu_in=True, out=o, planes_out=op, planes_only=True)
        assert op.valid and op.only and torch.equal(o.view(-1, 64).cpu().contiguous().view(torch.int16), _planes_reference(o_ref.view(-1, 64)))
        log = []
        ops.set_plan_log(log)
        yc = ops.conv2d(o, c2, pad=1, a_planes=op)
        ops.set_plan_log(None)
        assert log[-1].kernel.count(b",") == 6 and b", 1, false, false, true>" in log[-1].kernel, log[-1].kernel
        assert torch.equal(yc, y_ref), f"convolution on planes: max diff {(yc - y_ref).abs().max().item():.3e}"
        # loud failures: planes for a plan that cannot read / write them
        ops.gemm_tune(0, -1)
        p = _lib.GemmParams()
        ops._fill_common(p, x1, pw, x1, 0, None, False)
        p.m, p.lda, p.ldc = M, C0, C0
        p.a_x3 = 1
        assert _lib.lib().siu3r_gemm(C.byref(p), 0) != 0 and b"a_x3" in _lib.lib().siu3r_last_error()
        p.a_x3, p.c_x3 = 0, xp.t.data_ptr()
        assert _lib.lib().siu3r_gemm(C.byref(p), 0) != 0 and b"c_x3" in _lib.lib().siu3r_last_error()
        pl = ops.linear(x1, pw, dry_run=True)
        assert pl.a_x3_ok == 0 and pl.c_x3_ok == 0
        xq = ops.Planes(x1)
        ops.linear(a, pw, residual=r, out=x1, planes_out=xq)  # the 128 x 64 family: no planes, and the caller is told so
        assert not xq.valid
    finally:
        ops.set_plan_log(None)
        ops.gemm_tune(0, 0)


@pytest.mark.parametrize("mode", [MODES[0], MODES[2]], ids=["bf16", "bf16x3"])
def test_attention_kv_of_the_partner_batch_item(mode):
    """siu3r_attn_params.kv_bxor: batch item b attends to the keys / values of item b ^ 1 (the decoder's merged projection leaves a
    side's cross-attention memory in the OTHER side's row block) -- on the pair shape of the pipelined kernel, on a short one of the
    generic kernels, with strided q / k / v views -- bit for bit what the launch on explicitly swapped K / V tensors returns; and the
    concatenated packed weights (ops.cat_packed) of two LayerNorm-folded Linears with DIFFERENT LayerNorms equal the two GEMMs."""
    ops = _ops()
    name, adt, split, tol = mode
    for (B, Nq, Nk, H, D) in ((4, 1025, 1025, 12, 64), (2, 100, 77, 8, 32), (6, 40, 260, 4, 64)):
        buf = gen(B, max(Nq, Nk), 5, H, D, seed=300 + B).cuda().to(adt)   # [q | k | xk | v | xv]-style interleaved storage: strided views
        q, k, v = buf[:, :Nq, 0], buf[:, :Nk, 2], buf[:, :Nk, 4]
        swap = torch.arange(B).view(-1, 2).flip(1).reshape(-1).cuda()
        ref = ops.attention(q, k[swap].contiguous(), v[swap].contiguous(), heads=H, head_dim=D, scale=D ** -0.5, split3=split)
        got = ops.attention(q, k, v, heads=H, head_dim=D, scale=D ** -0.5, split3=split, kv_bxor=1)
        assert torch.equal(got, ref), (name, B, Nq, Nk, (got.float() - ref.float()).abs().max().item())
    with pytest.raises(RuntimeError, match="kv_bxor"):
        ops.attention(buf[:3, :, 0], buf[:3, :, 2], buf[:3, :, 4], heads=H, head_dim=D, scale=1.0, split3=split, kv_bxor=1)  # odd batch
    # two folded Linears behind different LayerNorms, as one GEMM
    M, C, N1, N2 = 300, 192, 128, 64
    x = gen(M, C, seed=310).cuda() + 1.5
    st = ops.RowStats(x)
    xb = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    ops.linear(gen(M, 64, seed=311).cuda().to(adt), ops.pack_linear(gen(C, 64, seed=312, scale=0.5).cuda(), gen(C, seed=313).cuda() + 1.5, split), out=x, stats_out=st, aux_out=xb)
    parts = []
    for i, n in enumerate((N1, N2)):
        pw = ops.pack_linear_ln(gen(n, C, seed=320 + i, scale=0.2).cuda(), gen(n, seed=330 + i).cuda(), (1.0 + 0.3 * gen(C, seed=340 + i)).cuda(), gen(C, seed=350 + i).cuda(), split)
        pw.meta["ln_eps"] = 1e-6
        parts.append(pw)
    A = x if split else xb
    both = ops.linear(A, ops.cat_packed(parts), ln=st, out_dtype=torch.float32)
    sep = torch.cat([ops.linear(A, pw, ln=st, out_dtype=torch.float32) for pw in parts], 1)
    check(f"cat_packed[{name}] vs separate launches", both, sep, 1e-6)


def test_attention_on_presplit_keys_and_values():
    """The q | k | v projection writes a MIXED buffer -- q as fp32, k and v pre-split (siu3r_gemm_params.c_x3 == c, c_x3_col0 = C) -- and
    the pipelined attention kernel reads those planes (siu3r_attn_params.kv_x3): the same bits as the fp32 route (both split K / V as
    hi = upper 16 bits, lo = bf16(x - hi)), on the pair shape with its odd last query (side path) and with the partner batch item's
    keys; RoPE is applied before the split; asking a kernel that cannot read planes fails loudly."""
    from oracle import siu3r_oracle as O

    ops = _ops()
    ops.gemm_tune(0, 2)
    try:
        Z, N, C, H, D = 2, 1025, 768, 12, 64
        x = gen(Z, N, C, seed=400).cuda()
        st = ops.RowStats(x)
        ops.linear(gen(Z, N, 64, seed=401).cuda(), ops.pack_linear(gen(C, 64, seed=402, scale=0.5).cuda(), gen(C, seed=403).cuda(), True), out=x, stats_out=st)
        pw = ops.pack_linear_ln(gen(3 * C, C, seed=404, scale=0.05).cuda(), gen(3 * C, seed=405).cuda(), (1 + 0.2 * gen(C, seed=406)).cuda(), gen(C, seed=407).cuda(), True)
        pw.meta["ln_eps"] = 1e-6
        pos = torch.randint(0, 33, (Z, N, 2), generator=torch.Generator().manual_seed(9)).cuda()
        cos, sin = O.rope2d_table(33, D)
        rope = (cos.cuda(), sin.cuda(), pos, 2 * C)
        att = dict(heads=H, head_dim=D, scale=D ** -0.5, split3=True)
        ref_buf = ops.linear(x, pw, ln=st, rope=rope)
        r5 = ref_buf.view(Z, N, 3, H, D)
        ref = ops.attention(r5[:, :, 0], r5[:, :, 1], r5[:, :, 2], **att)
        ref_x = ops.attention(r5[:, :, 0], r5[:, :, 1], r5[:, :, 2], kv_bxor=1, **att)
        buf = torch.empty_like(ref_buf)
        b5 = buf.view(Z, N, 3, H, D)
        assert ops.attention(b5[:, :, 0], b5[:, :, 1], b5[:, :, 2], dry_run=True, **att)
        P = ops.Planes(buf, storage=buf)
        ops.linear(x, pw, ln=st, rope=rope, out=buf, planes_out=P, planes_from_col=C)
        assert P.valid and not P.only
        assert torch.equal(buf[..., :C], ref_buf[..., :C]), "q columns stay fp32"
        got_kv = buf[..., C:].reshape(Z * N, 2 * C).cpu().contiguous().view(torch.int16)
        assert torch.equal(got_kv, _planes_reference(ref_buf[..., C:].reshape(Z * N, 2 * C))), "k | v columns: planes of the (RoPE'd) fp32 values"
        got = ops.attention(b5[:, :, 0], b5[:, :, 1], b5[:, :, 2], kv_planes=True, **att)
        assert torch.equal(got, ref), f"attention on planes: max diff {(got - ref).abs().max().item():.3e}"
        got_x = ops.attention(b5[:, :, 0], b5[:, :, 1], b5[:, :, 2], kv_planes=True, kv_bxor=1, **att)
        assert torch.equal(got_x, ref_x)
        qf, kf, vf = (r5[:, :, i].permute(0, 2, 1, 3).cpu() for i in range(3))
        check("attention on planes vs fp32 torch", got, F.scaled_dot_product_attention(qf, kf, vf).permute(0, 2, 1, 3).reshape(Z, N, C), TOL_F32)
        with pytest.raises(RuntimeError, match="kv_x3"):  # 40 keys: not the pipelined kernel
            ops.attention(b5[:, :40, 0], b5[:, :40, 1], b5[:, :40, 2], kv_planes=True, **att)
    finally:
        ops.gemm_tune(0, 0)
