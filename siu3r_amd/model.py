"""Host-side mirror of the reference's ``src/models`` interface for the MI355X path.

Module / attribute names, ``forward`` signatures and state-dict keys follow the reference
(src/models/model.py:31-389, backbone_croco.py:24-347, vit_adapter/vit_adapter.py:305-441,
heads/dpt_head.py, heads/dpt_gs_head.py, gaussian_adapter.py:50-110,
mask2former/video_seg_decoder.py:2257-2477) so that a reference checkpoint loads unmodified.
All arithmetic runs in libsiu3r_hip.so through ``siu3r_amd.ops``; PyTorch only allocates tensors and
provides the stream.  Activations are channel-last and (batch, view)-major: row b*V + v.

precision = "bf16"   : bf16 MFMA operands, bf16 inter-kernel activations, fp32 residual streams.
precision = "bf16x3" : fp32 activations, every product as three bf16 MFMAs (hi*hi + hi*lo + lo*hi);
                       this is the mode that meets the 1e-3 parity bar against the fp32 oracle.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import os

import torch
import torch.nn.functional as F

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU
from .gaussians_types import Gaussians
from . import postprocess as pp

ENC_HEADS, DEC_HEADS = 16, 12
ADAPTER_IDX = (5, 11, 17, 23)  # vit_adapter.py:317
DPT_HOOKS = (0, 6, 9, 12)      # dpt_head.py:141


class _Weights:
    """Packed parameters on the GPU, keyed by reference state-dict names."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, split: bool):
        self.sd, self.dev, self.split = sd, device, split
        self.lin: Dict[str, ops.PackedWeight] = {}
        self.vec: Dict[str, torch.Tensor] = {}

    def t(self, name):  # raw fp32 tensor on the GPU
        return self.sd[name].to(self.dev, torch.float32)

    def v(self, name):  # cached fp32 vector
        if name not in self.vec:
            self.vec[name] = self.t(name).contiguous()
        return self.vec[name]

    def linear(self, name, key=None, extra_bias=None):
        key = key or name
        if key not in self.lin:
            w = self.t(name + ".weight")
            w = w.reshape(w.shape[0], -1)
            b = self.t(name + ".bias") if (name + ".bias") in self.sd else None
            if extra_bias is not None:
                b = extra_bias if b is None else b + extra_bias
            self.lin[key] = ops.pack_matrix(w, b, self.split)
        return self.lin[key]

    def linear_ln(self, names, ln_name, eps=1e-6):
        """Linear(s) (several are concatenated along the output axis) behind the LayerNorm `ln_name`, folded (ops.pack_linear_ln)."""
        names = [names] if isinstance(names, str) else list(names)
        key = "+".join(names) + "#ln:" + ln_name
        if key not in self.lin:
            w = torch.cat([self.t(n + ".weight").reshape(self.sd[n + ".weight"].shape[0], -1) for n in names], 0)
            b = torch.cat([self.t(n + ".bias") for n in names], 0) if (names[0] + ".bias") in self.sd else None
            pw = ops.pack_linear_ln(w, b, self.t(ln_name + ".weight"), self.t(ln_name + ".bias"), self.split)
            pw.meta["ln_eps"] = eps
            self.lin[key] = pw
        return self.lin[key]

    def linear_ln_rows(self, name, rows, ln_name, eps=1e-6):
        """rows [lo, hi) of Linear `name` behind the LayerNorm `ln_name`, folded (a slice of a fused qkv weight)"""
        key = f"{name}[{rows[0]}:{rows[1]}]#ln:{ln_name}"
        if key not in self.lin:
            w = self.t(name + ".weight")[rows[0]:rows[1]]
            b = self.t(name + ".bias")[rows[0]:rows[1]] if (name + ".bias") in self.sd else None
            pw = ops.pack_linear_ln(w, b, self.t(ln_name + ".weight"), self.t(ln_name + ".bias"), self.split)
            pw.meta["ln_eps"] = eps
            self.lin[key] = pw
        return self.lin[key]

    def group(self, key, build, prefixes):
        """the same layer of several blocks (the two decoder sides) as one grouped weight: build(prefix) -> PackedWeight"""
        if key not in self.lin:
            self.lin[key] = ops.stack_packed([build(q) for q in prefixes])
        return self.lin[key]

    def stem7(self, names):
        """the 7x7 image stems `names` (Conv2d(3, 256, 7, 1, 3)) as the B fragments of siu3r_stem7x7_x3 -> (fragments, bias [G, 256] or None)"""
        key = "stem7:" + "|".join(names)
        if key not in self.lin:
            self.lin[key] = ops.pack_stem7([self.t(n + ".weight") for n in names], [self.t(n + ".bias") if (n + ".bias") in self.sd else None for n in names])
        return self.lin[key]

    def proj(self, names):
        """the 1 x 1 convolutions / Linears `names` as the B fragments of siu3r_proj_rows_x3 -> (fragments, padded bias or None, N)"""
        key = "proj:" + "|".join(names)
        if key not in self.lin:
            self.lin[key] = ops.pack_proj([self.t(n + ".weight") for n in names], [self.t(n + ".bias") if (n + ".bias") in self.sd else None for n in names])
        return self.lin[key]

    def merged(self, key, names):
        if key not in self.lin:
            w = torch.cat([self.t(n + ".weight") for n in names], 0)
            b = torch.cat([self.t(n + ".bias") for n in names], 0)
            self.lin[key] = ops.pack_matrix(w, b, self.split)
        return self.lin[key]

    def conv(self, name, cin_pad=None, bn=None):
        if name not in self.lin:
            w = self.t(name + ".weight")
            b = self.t(name + ".bias") if (name + ".bias") in self.sd else None
            if bn is not None:  # fold eval-mode BatchNorm (eps 1e-5) into the convolution
                s = self.t(bn + ".weight") / torch.sqrt(self.t(bn + ".running_var") + 1e-5)
                w = w * s[:, None, None, None]
                b0 = b if b is not None else torch.zeros_like(s)
                b = (b0 - self.t(bn + ".running_mean")) * s + self.t(bn + ".bias")
            self.lin[name] = ops.pack_conv(w, b, self.split, cin_pad=cin_pad)
        return self.lin[name]

    def convT(self, name):
        if name not in self.lin:
            self.lin[name] = ops.pack_conv_transpose(self.t(name + ".weight"), self.t(name + ".bias"), self.split)
        return self.lin[name]

    def bn_affine(self, name):
        key = name + "#affine"
        if key not in self.vec:
            s = self.t(name + ".weight") / torch.sqrt(self.t(name + ".running_var") + 1e-5)
            self.vec[key] = s.contiguous()
            self.vec[key + "b"] = (self.t(name + ".bias") - self.t(name + ".running_mean") * s).contiguous()
        return self.vec[key], self.vec[key + "b"]


class _Ctx:
    def __init__(self, w: _Weights, precision: str):
        assert precision in ("bf16", "bf16x3")
        self.w = w
        self.precision = precision
        self.split = precision == "bf16x3"
        self.act = torch.float32 if self.split else torch.bfloat16
        self.dev = w.dev
        self.cache: Dict = {}
        self._streams: Dict = {}
        self.concurrent = os.environ.get("SIU3R_NO_STREAMS", "0") != "1"
        # LayerNorms of the CroCo blocks folded into the GEMM that consumes them (statistics from the producing GEMM's epilogue), and
        # the two decoder sides of a pair merged into grouped launches.  SIU3R_NO_LNFOLD=1 restores the kernel-per-op path (A/B runs)
        self.fold = os.environ.get("SIU3R_NO_LNFOLD", "0") != "1"

    def side_stream(self, i):
        """i-th auxiliary HIP stream: the independent chains of the network (two decoder sides, the four DPT heads,
        the ViT-Adapter/Mask2Former branch) are enqueued on separate streams so that at batch 1, where most launches
        are far smaller than the 256 CUs, they fill the chip together.  Forks and joins are wait_stream() edges."""
        k = i
        if k not in self._streams:
            # default priority on purpose: a high-priority stream for the (critical) segmentation chain was measured to
            # nearly double the step on this runtime (14.5 -> 27.3 ms)
            self._streams[k] = torch.cuda.Stream(device=self.dev)
        return self._streams[k]

    def ln(self, name, x, eps, out_dtype=None):
        return ops.layernorm(x, self.w.v(name + ".weight"), self.w.v(name + ".bias"), eps, out_dtype or self.act)

    def ln2(self, name, x, eps):
        """(fp32 result for the residual stream, the same values in the GEMM-input dtype).  bf16 mode: one kernel writes both;
        bf16x3 mode: the GEMMs take fp32, so both are the one fp32 tensor."""
        if self.split:
            y = ops.layernorm(x, self.w.v(name + ".weight"), self.w.v(name + ".bias"), eps, torch.float32)
            return y, y
        return ops.layernorm2(x, self.w.v(name + ".weight"), self.w.v(name + ".bias"), eps)


# ==================================================================================================
# backbone  (reference backbone_croco.py:24-347; croco/blocks.py:81-191)
# ==================================================================================================
class AsymmetricCroCo:
    def __init__(self, ctx: _Ctx, enc_depth=24, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768):
        self.ctx = ctx
        self.enc_depth, self.dec_depth = enc_depth, dec_depth
        self.enc_embed_dim, self.dec_embed_dim = enc_embed_dim, dec_embed_dim
        self.depth_mode = ("exp", -float("inf"), float("inf"))
        self.conf_mode = None

    # ---- constant tables
    def _positions(self, B2, h, w):
        key = ("pos", B2, h, w)
        c = self.ctx.cache
        if key not in c:
            y, x = torch.arange(h), torch.arange(w)
            pos = torch.cartesian_prod(y, x).view(1, h * w, 2)
            extra = torch.tensor([[[h, 0]]])  # intrinsics token at (y = h, x = 0)  (backbone_croco.py:147-150)
            pos = torch.cat((pos, extra), 1).expand(B2, -1, 2).contiguous().to(self.ctx.dev)
            c[key] = pos
        return c[key]

    def _rope(self, max_pos, D=64, base=100.0):
        key = ("rope", max_pos, D)
        c = self.ctx.cache
        if key not in c:
            Q = D // 4
            q = torch.arange(Q, dtype=torch.float32)
            inv = 1.0 / torch.pow(torch.tensor(base, dtype=torch.float32), q / Q)
            ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None]
            c[key] = (torch.cos(ang).contiguous().to(self.ctx.dev), torch.sin(ang).contiguous().to(self.ctx.dev))
        return c[key]

    def _attn(self, p, xn, pos, rope, heads):
        """Attention.forward (blocks.py:94-112) on normalised tokens xn [Z, N, C] -> pre-projection context."""
        ctx = self.ctx
        Z, N, Cc = xn.shape
        # RoPE2D on q and k is fused into the QKV GEMM epilogue (columns [0, 2C) = q | k heads)
        qkv = ops.linear(xn, ctx.w.linear(p + ".qkv"), out_dtype=ctx.act, rope=(rope[0], rope[1], pos, 2 * Cc)).view(Z, N, 3, heads, Cc // heads)
        return ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], heads=heads, head_dim=Cc // heads,
                             scale=(Cc // heads) ** -0.5, split3=ctx.split)

    def _mlp(self, p, x, ln_name, out=None):
        ctx = self.ctx
        h = ops.linear(ctx.ln(ln_name, x, 1e-6), ctx.w.linear(p + ".fc1"), out_dtype=ctx.act, act=ACT_GELU)
        return ops.linear(h, ctx.w.linear(p + ".fc2"), out_dtype=torch.float32, residual=x, out=out)

    def _enc_block(self, p, x, pos, rope):
        ctx = self.ctx
        a = self._attn(p + ".attn", ctx.ln(p + ".norm1", x, 1e-6), pos, rope, ENC_HEADS)
        x = ops.linear(a, ctx.w.linear(p + ".attn.proj"), out_dtype=torch.float32, residual=x)
        return self._mlp(p + ".mlp", x, p + ".norm2")

    def _new_stream(self, like):
        """a residual-stream buffer with its row statistics and (bf16 mode) the bf16 copy the next GEMM multiplies"""
        x = torch.empty_like(like)
        return x, ops.RowStats(x), (None if self.ctx.split else torch.empty(like.shape, dtype=torch.bfloat16, device=like.device))

    def _enc_block_folded(self, p, S, pos, rope):
        """Block.forward (blocks.py:127-130) in 5 launches: norm1 / norm2 ride in the QKV / fc1 GEMMs (ops.pack_linear_ln), the
        proj / fc2 GEMMs write the new residual stream, its row statistics and its bf16 copy.  S = (x fp32, RowStats, bf16 copy)."""
        ctx = self.ctx
        x, xs, xb = S[:3]
        Z, N, Cc = x.shape
        if ctx.split:
            return self._enc_block_presplit(p, S, pos, rope)
        qkv = ops.linear(xb, ctx.w.linear_ln(p + ".attn.qkv", p + ".norm1"), out_dtype=ctx.act, ln=xs,
                         rope=(rope[0], rope[1], pos, 2 * Cc)).view(Z, N, 3, ENC_HEADS, Cc // ENC_HEADS)
        a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], heads=ENC_HEADS, head_dim=Cc // ENC_HEADS, scale=(Cc // ENC_HEADS) ** -0.5, split3=ctx.split)
        x1, s1, b1 = self._new_stream(x)
        ops.linear(a, ctx.w.linear(p + ".attn.proj"), residual=x, out=x1, stats_out=s1, aux_out=b1)
        h = ops.linear(b1, ctx.w.linear_ln(p + ".mlp.fc1", p + ".norm2"), out_dtype=ctx.act, act=ACT_GELU, ln=s1)
        x2, s2, b2 = self._new_stream(x)
        ops.linear(h, ctx.w.linear(p + ".mlp.fc2"), residual=x1, out=x2, stats_out=s2, aux_out=b2)
        return x2, s2, b2

    def _enc_block_presplit(self, p, S, pos, rope):
        """the same block in the bf16x3 mode: the GEMMs that write the residual stream (proj, fc2) also write it PRE-SPLIT (ops.Planes: the
        hi | lo bf16 planes a bf16x3 product multiplies), fc1 writes its GELU output as planes only, and QKV / fc1 / fc2 read planes -- the
        ping-pong kernel's in-loop split of the fp32 A operand is 12-18 % of those launches.  Every use is conditional on the launch's own
        plan (ops.linear: a_planes / planes_out), results are bit-identical either way.  S = (x fp32, RowStats, None, Planes | None)."""
        ctx = self.ctx
        x, xs, _, xp = S if len(S) == 4 else (*S, None)
        Z, N, Cc = x.shape
        # q stays fp32, k and v leave the projection pre-split (the attention kernel stages them without conversion) when it can read them
        qkv_buf = torch.empty((Z, N, 3 * Cc), dtype=torch.float32, device=x.device)
        qkv = qkv_buf.view(Z, N, 3, ENC_HEADS, Cc // ENC_HEADS)
        att = dict(heads=ENC_HEADS, head_dim=Cc // ENC_HEADS, scale=(Cc // ENC_HEADS) ** -0.5, split3=True)
        kvp = ops.Planes(qkv_buf, storage=qkv_buf) if (_KV_PLANES and ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dry_run=True, **att)) else None
        ops.linear(x, ctx.w.linear_ln(p + ".attn.qkv", p + ".norm1"), ln=xs, a_planes=xp, rope=(rope[0], rope[1], pos, 2 * Cc), out=qkv_buf,
                   planes_out=kvp, planes_from_col=Cc)
        a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], kv_planes=kvp is not None and kvp.valid, **att)
        x1, s1, _ = self._new_stream(x)
        xp1 = ops.Planes(x1)
        ops.linear(a, ctx.w.linear(p + ".attn.proj"), residual=x, out=x1, stats_out=s1, planes_out=xp1)
        x2, s2, _ = self._new_stream(x)
        xp2 = ops.Planes(x2)
        w1, w2 = ctx.w.linear_ln(p + ".mlp.fc1", p + ".norm2"), ctx.w.linear(p + ".mlp.fc2")
        h = torch.empty((Z, N, w1.n), dtype=torch.float32, device=x.device)
        hp = ops.Planes(h, storage=h)  # planes in place of the fp32 values when fc2 can read them (nothing else reads h)
        fc2_reads_planes = bool(ops.linear(h, w2, residual=x1, out=x2, stats_out=s2, planes_out=xp2, dry_run=True).a_x3_ok)
        ops.linear(x1, w1, act=ACT_GELU, ln=s1, a_planes=xp1, out=h, planes_out=hp if fc2_reads_planes else None, planes_only=True)
        ops.linear(h, w2, residual=x1, out=x2, stats_out=s2, a_planes=hp if fc2_reads_planes else None, planes_out=xp2)
        return x2, s2, None, xp2

    def _dec_block(self, p, x, y, xpos, ypos, rope, out=None):
        """DecoderBlock.forward (blocks.py:186-191); x, y are [B, N, C] fp32 strided views."""
        ctx = self.ctx
        B, N, Cc = x.shape
        d = Cc // DEC_HEADS
        a = self._attn(p + ".attn", ctx.ln(p + ".norm1", x, 1e-6), xpos, rope, DEC_HEADS)
        x = ops.linear(a, ctx.w.linear(p + ".attn.proj"), out_dtype=torch.float32, residual=x)
        y_ = ctx.ln(p + ".norm_y", y, 1e-6)
        q = ops.linear(ctx.ln(p + ".norm2", x, 1e-6), ctx.w.linear(p + ".cross_attn.projq"), out_dtype=ctx.act,
                       rope=(rope[0], rope[1], xpos, Cc)).view(B, N, DEC_HEADS, d)
        kv = ops.linear(y_, ctx.w.merged(p + ".cross_attn.projkv", [p + ".cross_attn.projk", p + ".cross_attn.projv"]),
                        out_dtype=ctx.act, rope=(rope[0], rope[1], ypos, Cc)).view(B, y.shape[1], 2, DEC_HEADS, d)
        a = ops.attention(q, kv[:, :, 0], kv[:, :, 1], heads=DEC_HEADS, head_dim=d, scale=d ** -0.5, split3=ctx.split)
        x = ops.linear(a, ctx.w.linear(p + ".cross_attn.proj"), out_dtype=torch.float32, residual=x)
        return self._mlp(p + ".mlp", x, p + ".norm3", out=out)

    def encode_begin(self, images, K):
        """patch embedding + intrinsics token (backbone_croco.py:270-281)."""
        ctx = self.ctx
        B, V, _, H, W = images.shape
        assert V >= 2
        assert H % 16 == 0, f"Input image height ({H}) is not a multiple of patch size (16)."
        assert W % 16 == 0, f"Input image width ({W}) is not a multiple of patch size (16)."
        h, w = H // 16, W // 16
        N = h * w
        Z = B * V
        img = images.reshape(Z, 3, H, W).contiguous().float()
        x = torch.empty((Z, N + 1, self.enc_embed_dim), dtype=torch.float32, device=ctx.dev)
        xs = xb = None
        if ctx.fold:
            xs = ops.RowStats(x)
            xb = None if ctx.split else torch.empty(x.shape, dtype=torch.bfloat16, device=ctx.dev)
        pe = ctx.w.lin.get("backbone.patch_embed.proj") or ctx.w.linear("backbone.patch_embed.proj")
        ops.patch_embed(img, pe, x, stats_out=xs, aux_out=xb)
        # intrinsics token: Linear(9 -> 1024) on the flattened K (backbone_croco.py:278), K padded to 16 columns
        if "backbone.intrinsic_encoder" not in ctx.w.lin:
            wi = F.pad(ctx.w.t("backbone.intrinsic_encoder.weight"), (0, 7))
            ctx.w.lin["backbone.intrinsic_encoder"] = ops.pack_matrix(wi, ctx.w.t("backbone.intrinsic_encoder.bias"), ctx.split)
        kin = torch.zeros((Z, 16), dtype=torch.float32, device=ctx.dev)
        kin[:, :9] = K.reshape(Z, 9).float()
        ops.linear(kin, ctx.w.lin["backbone.intrinsic_encoder"], out=x[:, N], stats_out=xs, aux_out=None if xb is None else xb[:, N])
        return dict(x=x, S=(x, xs, xb), all_feat=[], pos=self._positions(Z, h, w), rope=self._rope(max(h, w) + 1), dims=(B, V, H, W, N))

    def encode_blocks(self, e, lo, hi):
        """encoder blocks lo..hi-1; every block's output is kept (the ViT-Adapter taps blocks 5/11/17/23)."""
        for i in range(lo, hi):
            if self.ctx.fold:
                e["S"] = self._enc_block_folded(f"backbone.enc_blocks.{i}", e["S"], e["pos"], e["rope"])
                e["x"] = e["S"][0]
            else:
                e["x"] = self._enc_block(f"backbone.enc_blocks.{i}", e["x"], e["pos"], e["rope"])
            e["all_feat"].append(e["x"])

    def encode_end(self, e):
        B, V, H, W, N = e["dims"]
        e["av"] = [t.view(B, V, N + 1, -1) for t in e["all_feat"]]
        self._all_feat_bv = [t[..., :-1, :] for t in e["av"]]  # [B, V, N, C] views for the (b,v)-batched adapter
        return e

    # ---- encoder: all B*V views as one batch (backbone_croco.py:270-300)
    def encode(self, images, K):
        e = self.encode_begin(images, K)
        self.encode_blocks(e, 0, self.enc_depth)
        return self.encode_end(e)

    def _ctx_positions(self, B, V, h, w):
        """positions of the memory tokens of views 1..V-1: the other views in ascending order (generate_ctx_views,
        backbone_croco.py:528-539).  Constant per shape."""
        key = ("ctxpos", B, V, h, w)
        c = self.ctx.cache
        if key not in c:
            pv = self._positions(B * V, h, w).view(B, V, h * w + 1, 2)
            c[key] = torch.stack([torch.cat([pv[:, j] for j in range(V) if j != i], dim=1) for i in range(1, V)], dim=1).flatten(0, 1).contiguous()
        return c[key]

    # ---- decoder (backbone_croco.py:302-347 for the pair, :541-584 for V views): view 0 runs dec_blocks with the
    # tokens of all other views as memory; views 1..V-1 run dec_blocks2 (batched), each with the other views' tokens
    # (ascending view order) as memory.  Every layer reads the PREVIOUS layer's tokens of all views.
    def decode_begin(self, enc):
        """enc_norm + decoder_embed, positions, and ALL layer buffers (so that the per-layer, per-side steps -- which the
        model captures as separate HIP graphs -- only write into memory that outlives them)."""
        ctx = self.ctx
        B, V, H, W, N = enc["dims"]
        h, w = H // 16, W // 16
        x, pos = enc["x"], enc["pos"]
        f = ctx.ln("backbone.enc_norm", x, 1e-6, out_dtype=torch.float32)
        merged = ctx.fold and V == 2
        g = torch.empty((B, V, N + 1, self.dec_embed_dim), dtype=torch.float32, device=ctx.dev)
        gs = [g] + [torch.empty_like(g) for _ in range(self.dec_depth)]
        # merged sides (a pair): every layer buffer carries its row statistics and, in bf16 mode, the bf16 copy the next GEMMs read
        gstat = [ops.RowStats(t) for t in gs] if merged else None
        gb16 = [None if (ctx.split or not merged) else torch.empty(t.shape, dtype=torch.bfloat16, device=ctx.dev) for t in gs]
        ops.linear(f, ctx.w.linear("backbone.decoder_embed"), out=g.view(B * V * (N + 1), -1), stats_out=gstat[0] if merged else None,
                   aux_out=None if gb16[0] is None else gb16[0].view(B * V * (N + 1), -1))
        fv = f.view(B, V, N + 1, -1)
        pv = pos.view(B, V, N + 1, 2)
        pos0 = pv[:, 0].contiguous()
        if V == 2:
            pos_rest, mem0_pos, memr_pos = pv[:, 1].contiguous(), pv[:, 1].contiguous(), pos0
        else:
            pos_rest = pv[:, 1:].reshape(B * (V - 1), N + 1, 2).contiguous()
            mem0_pos = pv[:, 1:].reshape(B, (V - 1) * (N + 1), 2).contiguous()
            memr_pos = self._ctx_positions(B, V, h, w)
        return dict(enc=enc, fv=fv, g=gs, gstat=gstat, gb16=gb16, pos0=pos0, pos_rest=pos_rest, mem0_pos=mem0_pos, memr_pos=memr_pos, rope=enc["rope"],
                    merged=merged)

    def decode_layer_merged(self, d, i):
        """Layer i of BOTH decoder sides of a pair (DecoderBlock.forward, blocks.py:186-191; side 0 = dec_blocks on view 0, side 1 =
        dec_blocks2 on view 1, backbone_croco.py:231-255) in 9 launches: every Linear is one grouped GEMM over the two weight sets,
        norm1 / norm2 / norm_y / norm3 are folded into the GEMMs they feed, and the cross-attention memory of a side -- the OTHER
        view's tokens of the previous layer -- is read by the K/V projection through a flipped group order."""
        ctx = self.ctx
        B, V, H, W, N = d["enc"]["dims"]
        x, xs, xb = d["g"][i], d["gstat"][i], d["gb16"][i]
        Cc = x.shape[-1]
        hd = Cc // DEC_HEADS
        rope, pos = d["rope"], d["enc"]["pos"]  # [B*V, N+1, 2], (b, v)-major like blockIdx.z; a side's memory has the same grid positions
        sides = [f"backbone.dec_blocks.{i}", f"backbone.dec_blocks2.{i}"]
        W_ = ctx.w
        grp = lambda tag, build: W_.group(f"dec{i}.{tag}", build, sides)
        A = lambda t, tb: t if ctx.split else tb
        if _DEC_QKVX:
            # ONE projection launch per layer input: a side's rows make their own q | k | v (norm1 folded) AND the cross-attention memory
            # k | v of the OTHER side (that side's norm_y and projk / projv): both read the same un-normalised rows with the same row
            # statistics, the folded LayerNorms differ per output column only.  Columns [q | k | xk | v | xv]: RoPE covers the first 3 C.
            # (two launches of 144 + 96 workgroups become one of 240 that fills the chip: 70.5 -> 55.8 us per layer)
            other = {sides[0]: sides[1], sides[1]: sides[0]}
            build = lambda q: ops.cat_packed([W_.linear_ln_rows(q + ".attn.qkv", (0, 2 * Cc), q + ".norm1"),
                                              W_.linear_ln(other[q] + ".cross_attn.projk", other[q] + ".norm_y"),
                                              W_.linear_ln_rows(q + ".attn.qkv", (2 * Cc, 3 * Cc), q + ".norm1"),
                                              W_.linear_ln(other[q] + ".cross_attn.projv", other[q] + ".norm_y")])
            pj_buf = torch.empty((B, V, N + 1, 5 * Cc), dtype=ctx.act, device=x.device)
            pj = pj_buf.view(B * V, N + 1, 5, DEC_HEADS, hd)
            att = dict(heads=DEC_HEADS, head_dim=hd, scale=hd ** -0.5, split3=ctx.split)
            # bf16x3: k | xk | v | xv leave the projection pre-split (ops.Planes) when the attention kernel can read them; q stays fp32
            kvp = ops.Planes(pj_buf, storage=pj_buf) if (ctx.split and _KV_PLANES and ops.attention(pj[:, :, 0], pj[:, :, 1], pj[:, :, 3], dry_run=True, **att)) else None
            ops.linear_grouped(A(x, xb), grp("qkvx", build), ln=xs, rope=(rope[0], rope[1], pos, 3 * Cc), out=pj_buf, planes_out=kvp, planes_from_col=Cc)
            kvp = kvp is not None and kvp.valid
            a = ops.attention(pj[:, :, 0], pj[:, :, 1], pj[:, :, 3], kv_planes=kvp, **att)
        else:
            pj = None
            qkv = ops.linear_grouped(A(x, xb), grp("qkv", lambda q: W_.linear_ln(q + ".attn.qkv", q + ".norm1")), out_dtype=ctx.act, ln=xs,
                                     rope=(rope[0], rope[1], pos, 2 * Cc)).view(B * V, N + 1, 3, DEC_HEADS, hd)
            a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], heads=DEC_HEADS, head_dim=hd, scale=hd ** -0.5, split3=ctx.split)
        x1, s1, b1 = self._new_stream(x)
        ops.linear_grouped(a.view(B, V, N + 1, Cc), grp("proj", lambda q: W_.linear(q + ".attn.proj")), residual=x, out=x1, stats_out=s1, aux_out=b1)
        qq = ops.linear_grouped(A(x1, b1), grp("projq", lambda q: W_.linear_ln(q + ".cross_attn.projq", q + ".norm2")), out_dtype=ctx.act, ln=s1,
                                rope=(rope[0], rope[1], pos, Cc)).view(B * V, N + 1, DEC_HEADS, hd)
        if pj is not None:
            # side g's memory was projected from side (1 - g)'s rows: batch item b ^ 1 of the merged projection
            a = ops.attention(qq, pj[:, :, 2], pj[:, :, 4], kv_bxor=1, kv_planes=kvp, **att)
        else:
            kv = ops.linear_grouped(A(x, xb), grp("projkv", lambda q: W_.linear_ln([q + ".cross_attn.projk", q + ".cross_attn.projv"], q + ".norm_y")),
                                    out_dtype=ctx.act, ln=xs, flip=True, rope=(rope[0], rope[1], pos, Cc)).view(B * V, N + 1, 2, DEC_HEADS, hd)
            a = ops.attention(qq, kv[:, :, 0], kv[:, :, 1], heads=DEC_HEADS, head_dim=hd, scale=hd ** -0.5, split3=ctx.split)
        x2, s2, b2 = self._new_stream(x)
        ops.linear_grouped(a.view(B, V, N + 1, Cc), grp("xproj", lambda q: W_.linear(q + ".cross_attn.proj")), residual=x1, out=x2, stats_out=s2, aux_out=b2)
        h = ops.linear_grouped(A(x2, b2), grp("fc1", lambda q: W_.linear_ln(q + ".mlp.fc1", q + ".norm3")), out_dtype=ctx.act, act=ACT_GELU, ln=s2)
        ops.linear_grouped(h, grp("fc2", lambda q: W_.linear(q + ".mlp.fc2")), residual=x2, out=d["g"][i + 1], stats_out=d["gstat"][i + 1],
                           aux_out=d["gb16"][i + 1])

    def decode_side(self, d, i, side):
        """Layer i, side 0 = view 0 (dec_blocks), side 1 = views 1..V-1 batched (dec_blocks2); reads layer i's input
        buffer g[i] (all views), writes its own views of g[i+1]."""
        B, V, H, W, N = d["enc"]["dims"]
        g, g_next, rope = d["g"][i], d["g"][i + 1], d["rope"]
        if side == 0:
            mem0 = g[:, 1] if V == 2 else g[:, 1:].reshape(B, (V - 1) * (N + 1), -1)
            self._dec_block(f"backbone.dec_blocks.{i}", g[:, 0], mem0, d["pos0"], d["mem0_pos"], rope, out=g_next[:, 0])
            return
        if V == 2:
            xr, memr, outr = g[:, 1], g[:, 0], g_next[:, 1]
        else:
            xr = g[:, 1:].reshape(B * (V - 1), N + 1, -1)
            memr = torch.stack([torch.cat([g[:, j] for j in range(V) if j != v], dim=1) for v in range(1, V)], dim=1).flatten(0, 1)
            outr = g_next[0, 1:] if B == 1 else None  # views 1.. of g_next form one strided batch only then
        r = self._dec_block(f"backbone.dec_blocks2.{i}", xr, memr, d["pos_rest"], d["memr_pos"], rope, out=outr)
        if outr is None:
            g_next[:, 1:].copy_(r.view(B, V - 1, N + 1, -1))

    def decode_end(self, d):
        B, V, H, W, N = d["enc"]["dims"]
        layers = [d["fv"]] + d["g"][1:]
        last = self.ctx.ln("backbone.dec_norm", layers[-1].reshape(B * V, N + 1, -1), 1e-6, out_dtype=torch.float32)
        layers[-1] = last.view(B, V, N + 1, -1)
        return dict(layers=layers, fv=d["fv"])

    def decode(self, enc):
        """The two sides of a layer are independent (each reads the other's PREVIOUS layer output): view 0 on the
        current stream, the other views on a side stream, joined after every layer."""
        ctx = self.ctx
        d = self.decode_begin(enc)
        if d["merged"]:
            for i in range(self.dec_depth):
                self.decode_layer_merged(d, i)
            return self.decode_end(d)
        main = torch.cuda.current_stream()
        side = ctx.side_stream(0) if ctx.concurrent else main
        for i in range(self.dec_depth):
            if side is not main:
                side.wait_stream(main)
            self.decode_side(d, i, 0)
            with torch.cuda.stream(side):
                self.decode_side(d, i, 1)
            if side is not main:
                main.wait_stream(side)
        return self.decode_end(d)

    def _assemble(self, enc, dec):
        strip = lambda t: t[..., :-1, :]
        B, V, H, W, N = enc["dims"]
        feats = [strip(dec["fv"][:, v]) for v in range(V)]
        all_feats = [[strip(t[:, v]) for t in enc["av"]] for v in range(V)]
        decs = [[strip(t[:, v]) for t in dec["layers"]] for v in range(V)]
        return feats, all_feats, decs

    def forward(self, context: dict, symmetrize_batch=False, return_views=False, after_encoder=None):
        """reference signature backbone_croco.py:263-268 (+ after_encoder hook); context = {"image": [B,2,3,H,W], "intrinsics": [B,2,3,3]}."""
        assert not symmetrize_batch, "symmetrize_batch is a training-time option"
        images, K = context["image"], context["intrinsics"]
        B, V, _, H, W = images.shape
        assert V == 2, "AsymmetricCroCo is the two-view backbone; use AsymmetricCroCoMulti for more views"
        enc = self.encode(images, K)
        if after_encoder is not None:
            after_encoder()  # lets the caller fork the encoder-only consumers (ViT-Adapter branch) before the decoder
        dec = self.decode(enc)
        feats, all_feats, decs = self._assemble(enc, dec)
        shape = torch.tensor([[H, W]] * B)
        res = (feats[0], feats[1], all_feats[0], all_feats[1], decs[0], decs[1], shape, shape.clone())
        if return_views:
            res = res + ({"img": images[:, 0]}, {"img": images[:, 1]})
        return res

    __call__ = forward

    @property
    def patch_size(self):
        return 16

    @property
    def d_out(self):
        return 1024


class AsymmetricCroCoMulti(AsymmetricCroCo):
    """V >= 2 context views (reference backbone_croco.py:350-590): same weights and blocks as AsymmetricCroCo; view 0
    decodes with dec_blocks, views 1..V-1 with dec_blocks2."""

    def forward(self, context: dict, symmetrize_batch=False, return_views=False):
        assert not symmetrize_batch, "symmetrize_batch is a training-time option"
        images, K = context["image"], context["intrinsics"]
        B, V, _, H, W = images.shape
        enc = self.encode(images, K)
        dec = self.decode(enc)
        feats, all_feats, decs = self._assemble(enc, dec)
        shapes = [torch.tensor([[H, W]] * B) for _ in range(V)]
        res = (feats, all_feats, decs, shapes)
        if return_views:
            res = res + ([{"img": images[:, v]} for v in range(V)],)
        return res

    __call__ = forward


# ==================================================================================================
# DPT heads (heads/dpt_block.py, dpt_head.py:36-79, dpt_gs_head.py:121-171)
# ==================================================================================================
class _DPTHead:
    def __init__(self, ctx: _Ctx, prefix: str, gs: bool):
        self.ctx, self.p, self.gs = ctx, prefix, gs

    def _rcu(self, q, x, extra=None):
        """ResidualConvUnit_custom (dpt_block.py:126-147): conv2(relu(conv1(relu(x)))) + x (+ extra)."""
        ctx = self.ctx
        o = ops.conv2d(x, ctx.w.conv(q + ".conv1"), pad=1, out_dtype=ctx.act, act=ACT_RELU, relu_in=True)
        res = x if extra is None else ops.affine_add(x, extra, None, None)
        return ops.conv2d(o, ctx.w.conv(q + ".conv2"), pad=1, out_dtype=ctx.act, residual=res)

    def _fusion(self, q, x0, x1):
        """FeatureFusionBlock_custom.forward (dpt_block.py:198-237).  out_conv (1x1) commutes with the
        align_corners=True bilinear x2 (both linear, weights sum to 1), so it runs at the low resolution."""
        ctx = self.ctx
        out = x0 if x1 is None else self._rcu(q + ".resConfUnit1", x1, extra=x0)
        out = self._rcu(q + ".resConfUnit2", out)
        out = ops.linear(out, ctx.w.linear(q + ".out_conv"), out_dtype=ctx.act)
        B, hh, ww, Cc = out.shape
        return ops.resize_bilinear(out, (2 * hh, 2 * ww), True)

    def trunk(self, tokens: Sequence[torch.Tensor], H, W):
        ctx, p = self.ctx, self.p
        nh, nw = H // 16, W // 16
        layers = []
        for i, hk in enumerate(DPT_HOOKS):
            t = tokens[hk]  # [B, N, C] fp32 strided view
            B = t.shape[0]
            a = f"{p}.dpt.act_postprocess.{i}"
            l = ops.linear(t, ctx.w.linear(a + ".0"), out_dtype=ctx.act).view(B, nh, nw, -1)
            if i in (0, 1):
                l = ops.conv_transpose2d(l, ctx.w.convT(a + ".1"), out_dtype=ctx.act)
            elif i == 3:
                l = ops.conv2d(l, ctx.w.conv(a + ".1"), stride=2, pad=1, out_dtype=ctx.act)
            l = ops.conv2d(l, ctx.w.conv(f"{p}.dpt.scratch.layer_rn.{i}"), pad=1, out_dtype=ctx.act)
            layers.append(l)
        s = f"{p}.dpt.scratch"
        path4 = self._fusion(s + ".refinenet4", layers[3], None)
        path3 = self._fusion(s + ".refinenet3", path4, layers[2])
        path2 = self._fusion(s + ".refinenet2", path3, layers[1])
        return self._fusion(s + ".refinenet1", path2, layers[0])

    def forward_pts3d(self, tokens, H, W):
        """PixelwiseTaskWithDPT.forward + postprocess 'exp' (dpt_head.py:113-120; postprocess.py:45-61)."""
        ctx, p = self.ctx, self.p
        x = self.trunk(tokens, H, W)
        x = ops.conv2d(x, ctx.w.conv(f"{p}.dpt.head.0"), pad=1, out_dtype=ctx.act)
        x = ops.resize_bilinear(x, (H, W), True)
        x = ops.conv2d(x, ctx.w.conv(f"{p}.dpt.head.2"), pad=1, out_dtype=ctx.act, act=ACT_RELU)
        xyz = _head4(ctx, (f"{p}.dpt.head.4",), x.view(x.shape[0], 1, H * W, x.shape[-1]))
        xyz = ops.linear(x, ctx.w.linear(f"{p}.dpt.head.4"), out_dtype=torch.float32) if xyz is None else xyz.view(x.shape[0], H, W, -1)
        return {"pts3d": ops.pts3d_exp_(xyz)}

    def forward_gs(self, tokens, img_nhwc8, H, W, out=None):
        """dpt_gs_head.py:121-171: feat_up(path_1) + ReLU(conv7x7(img)) fused into the 7x7 conv's epilogue.
        out: optional [B, H*W, 83] fp32 destination (a view of the model's [B, V, H*W, 83] raw-Gaussian buffer)."""
        ctx, p = self.ctx, self.p
        path1 = self.trunk(tokens, H, W)
        if _stem_kernel_ok(ctx, img_nhwc8, path1, H, W):  # the dedicated stem kernel (csrc/stem.hip), planes when head.0's plan reads them
            Bn = img_nhwc8.shape[0]
            x = torch.empty((Bn, H, W, 256), dtype=torch.float32, device=ctx.dev)
            w0 = ctx.w.conv(f"{p}.dpt.head.0")
            xp = ops.Planes(x, storage=x) if _CONV_PLANES else None
            reads = xp is not None and not ops._NO_PRESPLIT and bool(ops.conv2d(x, w0, pad=1, act=ACT_RELU, dry_run=True).a_x3_ok)
            wf, wb = ctx.w.stem7((f"{p}.dpt.input_merger.0",))
            ops.stem7x7_x3(img_nhwc8.view(Bn, 1, H, W, 4), wf, wb, path1.view(Bn, 1, H // 2, W // 2, 256), x.view(Bn, 1, H, W, 256), planes=reads)
            if reads:
                xp.valid = xp.only = True
            x = ops.conv2d(x, w0, pad=1, out_dtype=ctx.act, act=ACT_RELU, a_planes=xp if reads else None)
        else:
            x = ops.conv2d(img_nhwc8, ctx.w.conv(f"{p}.dpt.input_merger.0", cin_pad=ops.image_channels(ctx.split)), pad=3, out_dtype=ctx.act,
                           act=ACT_RELU, up_src=path1)
            x = ops.conv2d(x, ctx.w.conv(f"{p}.dpt.head.0"), pad=1, out_dtype=ctx.act, act=ACT_RELU)
        Bn = x.shape[0]
        if out is not None and out.stride(2) == 1:  # [B, H*W, 83]: one view of every batch item of the raw-Gaussian buffer
            if _head4(ctx, (f"{p}.dpt.head.4",), x.view(Bn, 1, H * W, x.shape[-1]), out=out.unsqueeze(1)) is not None:
                return out
        if out is not None:
            return ops.linear(x.view(x.shape[0], H * W, -1), ctx.w.linear(f"{p}.dpt.head.4"), out=out)
        r = _head4(ctx, (f"{p}.dpt.head.4",), x.view(Bn, 1, H * W, x.shape[-1]))
        if r is not None:
            return r.view(Bn, H, W, -1)
        return ops.linear(x, ctx.w.linear(f"{p}.dpt.head.4"), out_dtype=torch.float32)  # [B,H,W,83]


_PROJ_KERNEL = os.environ.get("SIU3R_NO_PROJ_KERNEL", "0") != "1"  # A/B switch: the heads' last 1 x 1 convolutions on the GEMM again


def _head4(ctx, names, x, out=None):
    """the last 1 x 1 convolution of G DPT heads (names): x [B, G, M, K] fp32 -> [B, G, M, N] fp32 (into `out` when given).  bf16x3: the
    row-stream kernel (csrc/proj.hip) where it has an instantiation; returns None when the caller should take the GEMM."""
    B, G, M, K = x.shape
    n = ctx.w.sd[names[0] + ".weight"].shape[0]
    if not (_PROJ_KERNEL and ctx.split and x.dtype == torch.float32 and x.is_contiguous() and ops.proj_rows_ok(K, n)):
        return None
    wf, wb, n = ctx.w.proj(tuple(names))
    if out is None:
        out = torch.empty((B, G, M, n), dtype=torch.float32, device=x.device)
    return ops.proj_rows_x3(x, wf, wb, n, out)


_STEM_KERNEL = os.environ.get("SIU3R_NO_STEM_KERNEL", "0") != "1"  # A/B switch: the Gaussian stems on the implicit-GEMM convolution again


def _stem_kernel_ok(ctx, img, path1, H, W) -> bool:
    """siu3r_stem7x7_x3 serves the bf16x3 mode: fp32 RGB0 image, fp32 256-channel upsample source, H and W multiples of 16"""
    return (_STEM_KERNEL and ctx.split and img.dtype == torch.float32 and img.shape[-1] == 4 and img.is_contiguous() and path1.dtype == torch.float32
            and path1.shape[-1] == 256 and path1.is_contiguous() and H % 16 == 0 and W % 16 == 0)


class _DPTHeadPair:
    """The two DPT heads of a kind (head1 on view 0, head2 on view 1: reference model.py:352-375 runs them one after the other) as ONE
    sequence of grouped launches: activations [B, 2, h, w, C], every convolution / linear multiplies image (b, g) by the weights of head
    g (blockIdx.z = b * 2 + g).  Same arithmetic per head as _DPTHead; half the launches, twice the tiles per launch -- the trunk's
    convolutions at 16^2 .. 128^2 are a fraction of the 256 CUs each on their own.  Used for V == 2, B == 1 (the stem of the Gaussian
    head reads its upsample source per head and stays two launches)."""

    def __init__(self, ctx: _Ctx, prefixes: Sequence[str], gs: bool):
        self.ctx, self.ps, self.gs = ctx, tuple(prefixes), gs

    def _w(self, kind: str, name: str, **kw):
        """layer `name` (relative to the head prefix) of both heads as one grouped weight"""
        w = self.ctx.w
        build = {"conv": lambda q: w.conv(q + name, **kw), "convT": lambda q: w.convT(q + name), "linear": lambda q: w.linear(q + name)}[kind]
        return w.group("pair:" + "|".join(self.ps) + ":" + name, build, self.ps)

    def _rcu(self, q, x, extra=None):
        ctx = self.ctx
        w1, w2 = self._w("conv", q + ".conv1"), self._w("conv", q + ".conv2")
        res = x if extra is None else ops.affine_add(x.flatten(0, 1), extra.flatten(0, 1), None, None).view(x.shape)
        if not (ctx.split and _CONV_PLANES):
            o = ops.conv2d_grouped(x, w1, pad=1, out_dtype=ctx.act, act=ACT_RELU, relu_in=True)
            return ops.conv2d_grouped(o, w2, pad=1, out_dtype=ctx.act, residual=res)
        # bf16x3: conv1's ReLU'd output feeds conv2 only -- written as pre-split planes (ops.Planes) when conv2's plan reads them
        o = torch.empty(x.shape[:-1] + (w1.n,), dtype=torch.float32, device=x.device)
        op = ops.Planes(o, storage=o)
        reads = bool(ops.conv2d_grouped(o, w2, pad=1, residual=res, dry_run=True).a_x3_ok)
        ops.conv2d_grouped(x, w1, pad=1, act=ACT_RELU, relu_in=True, out=o, planes_out=op if reads else None, planes_only=True)
        return ops.conv2d_grouped(o, w2, pad=1, residual=res, a_planes=op if reads else None)

    def _fusion(self, q, x0, x1):
        ctx = self.ctx
        out = x0 if x1 is None else self._rcu(q + ".resConfUnit1", x1, extra=x0)
        out = self._rcu(q + ".resConfUnit2", out)
        B, G, hh, ww, Cc = out.shape
        out = ops.linear_grouped(out.view(B, G, hh * ww, Cc), self._w("linear", q + ".out_conv"), out_dtype=ctx.act)
        return ops.resize_bilinear(out.view(B * G, hh, ww, -1), (2 * hh, 2 * ww), True).view(B, G, 2 * hh, 2 * ww, -1)

    def trunk(self, tokens: Sequence[torch.Tensor], H, W):
        """tokens: per hooked layer [B, 2, N, C] (strided view: both views' patch tokens)"""
        ctx = self.ctx
        nh, nw = H // 16, W // 16
        layers = []
        for i, hk in enumerate(DPT_HOOKS):
            t = tokens[hk]
            B, G = t.shape[:2]
            a = f".dpt.act_postprocess.{i}"
            l = ops.linear_grouped(t, self._w("linear", a + ".0"), out_dtype=ctx.act).view(B, G, nh, nw, -1)
            if i in (0, 1):
                l = ops.conv_transpose2d_grouped(l, self._w("convT", a + ".1"), out_dtype=ctx.act)
            elif i == 3:
                l = ops.conv2d_grouped(l, self._w("conv", a + ".1"), stride=2, pad=1, out_dtype=ctx.act)
            layers.append(ops.conv2d_grouped(l, self._w("conv", f".dpt.scratch.layer_rn.{i}"), pad=1, out_dtype=ctx.act))
        s = ".dpt.scratch"
        path4 = self._fusion(s + ".refinenet4", layers[3], None)
        path3 = self._fusion(s + ".refinenet3", path4, layers[2])
        path2 = self._fusion(s + ".refinenet2", path3, layers[1])
        return self._fusion(s + ".refinenet1", path2, layers[0])

    def forward_pts3d(self, tokens, H, W):
        """-> pts3d [B, 2, H, W, 3]"""
        ctx = self.ctx
        x = self.trunk(tokens, H, W)
        x = ops.conv2d_grouped(x, self._w("conv", ".dpt.head.0"), pad=1, out_dtype=ctx.act)
        B, G = x.shape[:2]
        x = ops.resize_bilinear(x.flatten(0, 1), (H, W), True).view(B, G, H, W, -1)
        x = ops.conv2d_grouped(x, self._w("conv", ".dpt.head.2"), pad=1, out_dtype=ctx.act, act=ACT_RELU)
        xyz = _head4(ctx, tuple(q + ".dpt.head.4" for q in self.ps), x.view(B, G, H * W, -1))
        if xyz is None:
            xyz = ops.linear_grouped(x.view(B, G, H * W, -1), self._w("linear", ".dpt.head.4"), out_dtype=torch.float32)
        return ops.pts3d_exp_(xyz).view(B, G, H, W, 3)

    def forward_gs(self, tokens, img_nhwc8, H, W, out):
        """img_nhwc8 [B, 2, H, W, c]; out: the model's [B, 2, H*W, 83] raw-Gaussian buffer (written in place)"""
        ctx = self.ctx
        path1 = self.trunk(tokens, H, W)
        B, G = path1.shape[:2]
        assert B == 1
        x = torch.empty((B, G, H, W, path1.shape[-1]), dtype=ctx.act, device=ctx.dev)
        w0 = self._w("conv", ".dpt.head.0")
        # bf16x3: the stem's full-resolution map (268 MB per view) is read by head.0 only: pre-split planes when head.0's plan reads them
        xp = ops.Planes(x, storage=x) if (ctx.split and _CONV_PLANES) else None
        reads = xp is not None and bool(ops.conv2d_grouped(x, w0, pad=1, act=ACT_RELU, dry_run=True).a_x3_ok)
        if _stem_kernel_ok(ctx, img_nhwc8, path1, H, W):  # both stems in ONE launch of the dedicated kernel (csrc/stem.hip)
            wf, wb = ctx.w.stem7(tuple(q + ".dpt.input_merger.0" for q in self.ps))
            reads = reads and not ops._NO_PRESPLIT
            ops.stem7x7_x3(img_nhwc8, wf, wb, path1, x, planes=reads)
            if reads:
                xp.valid = xp.only = True
            x = ops.conv2d_grouped(x, w0, pad=1, out_dtype=ctx.act, act=ACT_RELU, a_planes=xp if reads else None)
            return self._gs_head4(x.view(B, G, H * W, -1), out)
        wrote = []
        for g, q in enumerate(self.ps):  # the stem: 7x7 image convolution + ReLU + x2 upsample-add of this head's path_1
            xg = ops.Planes(x[:, g], storage=x[:, g]) if reads else None
            ops.conv2d(img_nhwc8[:, g], ctx.w.conv(q + ".dpt.input_merger.0", cin_pad=ops.image_channels(ctx.split)), pad=3, act=ACT_RELU,
                       up_src=path1[:, g], out=x[:, g], planes_out=xg, planes_only=True)  # (B == 1: the slices of a group are contiguous)
            wrote.append(xg is not None and xg.valid)
        if reads and any(wrote):
            assert all(wrote), "the two stems of a pair run the same plan"
            xp.valid = xp.only = True
        x = ops.conv2d_grouped(x, w0, pad=1, out_dtype=ctx.act, act=ACT_RELU, a_planes=xp if (reads and all(wrote)) else None)
        return self._gs_head4(x.view(B, G, H * W, -1), out)

    def _gs_head4(self, x, out):
        """head.4 of both Gaussian heads into the model's raw-Gaussian buffer: the row-stream kernel (bf16x3, contiguous destination), else the grouped GEMM"""
        r = _head4(self.ctx, tuple(q + ".dpt.head.4" for q in self.ps), x, out=out)
        return r if r is not None else ops.linear_grouped(x, self._w("linear", ".dpt.head.4"), out=out)


class UnifiedGaussianAdapter:
    """gaussian_adapter.py:50-110."""

    def __init__(self, gaussian_scale_min=0.5, gaussian_scale_max=15.0, sh_degree=4):
        assert sh_degree == 4, "the HIP adapter kernel is specialised for sh_degree 4 (83 raw channels)"
        self.sh_degree = sh_degree

    @property
    def d_sh(self):
        return (self.sh_degree + 1) ** 2

    @property
    def d_in(self):
        return 7 + 3 * self.d_sh

    def forward(self, means, raw_gaussians, eps: float = 1e-8) -> Gaussians:
        g = ops.gaussian_adapter(raw_gaussians.contiguous())
        return Gaussians(means=means, **g)


# ==================================================================================================
# ViT-Adapter (vit_adapter/vit_adapter.py:305-441)
# ==================================================================================================
class CroCoViTAdapter:
    def __init__(self, ctx: _Ctx, size):
        self.ctx = ctx
        self.H, self.W = size[0] // 16, size[1] // 16

    def _ref(self, Hi, Wi):
        key = ("adapter_ref", Hi, Wi)
        c = self.ctx.cache
        if key not in c:
            pts = []
            for (hh, ww) in ((Hi // 8, Wi // 8), (Hi // 16, Wi // 16), (Hi // 32, Wi // 32)):
                ry, rx = torch.meshgrid(torch.linspace(0.5, hh - 0.5, hh), torch.linspace(0.5, ww - 0.5, ww), indexing="ij")
                pts.append(torch.stack((rx.reshape(-1) / ww, ry.reshape(-1) / hh), -1))
            c[key] = torch.cat(pts, 0)[:, None, :].contiguous().to(self.ctx.dev)
        return c[key]

    def _extractor(self, p, c, ref, feat, h, w):
        """Extractor.forward (vit_adapter.py:96-121); MSDeformAttn.forward (blocks.py:147-213)."""
        ctx = self.ctx
        qn = ctx.ln(p + ".query_norm", c, 1e-6)
        fn = ctx.ln(p + ".feat_norm", feat, 1e-6)
        offs_aw = ops.linear(qn, ctx.w.merged(p + ".attn.offs_aw", [p + ".attn.sampling_offsets", p + ".attn.attention_weights"]), out_dtype=torch.float32)
        value = ops.linear(fn, ctx.w.linear(p + ".attn.value_proj"), out_dtype=ctx.act)
        samp = ops.msdeform_sample(value, offs_aw, ref, [(h, w)], 16, 4, ctx.act)
        c = ops.linear(samp, ctx.w.linear(p + ".attn.output_proj"), out_dtype=torch.float32, residual=c)
        f1 = ops.linear(ctx.ln(p + ".ffn_norm", c, 1e-6), ctx.w.linear(p + ".ffn.fc1"), out_dtype=ctx.act)
        wk = p + ".ffn.dwconv.dwconv"
        if wk + "#w9c" not in ctx.w.vec:
            ctx.w.vec[wk + "#w9c"] = ctx.w.t(wk + ".weight").reshape(-1, 9).t().contiguous()
        f2 = ops.dwconv3x3_gelu(f1, ctx.w.vec[wk + "#w9c"], ctx.w.v(wk + ".bias"), h, w)
        return ops.linear(f2, ctx.w.linear(p + ".ffn.fc2"), out_dtype=torch.float32, residual=c)

    def spm(self, img8: torch.Tensor):
        """SpatialPriorModule + level embeddings (vit_adapter.py:380-391): needs only the image."""
        ctx = self.ctx
        Z, Hi, Wi, _ = img8.shape
        h, w = Hi // 16, Wi // 16
        sp = "adapter.spm"
        c = ops.conv2d(img8, ctx.w.conv(sp + ".stem.0", cin_pad=ops.image_channels(ctx.split), bn=sp + ".stem.1"), stride=2, pad=1, out_dtype=ctx.act, act=ACT_RELU)
        c = ops.conv2d(c, ctx.w.conv(sp + ".stem.3", bn=sp + ".stem.4"), pad=1, out_dtype=ctx.act, act=ACT_RELU)
        c = ops.conv2d(c, ctx.w.conv(sp + ".stem.6", bn=sp + ".stem.7"), pad=1, out_dtype=ctx.act, act=ACT_RELU)
        c1 = ops.maxpool3x3s2(c)
        c2 = ops.conv2d(c1, ctx.w.conv(sp + ".conv2.0", bn=sp + ".conv2.1"), stride=2, pad=1, out_dtype=ctx.act, act=ACT_RELU)
        c3 = ops.conv2d(c2, ctx.w.conv(sp + ".conv3.0", bn=sp + ".conv3.1"), stride=2, pad=1, out_dtype=ctx.act, act=ACT_RELU)
        c4 = ops.conv2d(c3, ctx.w.conv(sp + ".conv4.0", bn=sp + ".conv4.1"), stride=2, pad=1, out_dtype=ctx.act, act=ACT_RELU)
        le = ctx.w.v("adapter.level_embed")
        n2, n3, n4 = 4 * h * w, h * w, h * w // 4
        cc = torch.empty((Z, n2 + n3 + n4, 1024), dtype=torch.float32, device=ctx.dev)
        # (c1 = fc1(c1) [Z,4h,4w,1024] is only ever read through the pixel decoder's lateral 1 x 1 convolution: finish() applies the
        # composed 64 -> 256 map instead and materialises the 1024-channel map on request only)
        # level_embed is folded into the fc biases (vit_adapter.py:387-391)
        ops.linear(c2.view(Z, n2, -1), ctx.w.linear(sp + ".fc2", key=sp + ".fc2+le", extra_bias=le[0]), out=cc[:, :n2])
        ops.linear(c3.view(Z, n3, -1), ctx.w.linear(sp + ".fc3", key=sp + ".fc3+le", extra_bias=le[1]), out=cc[:, n2:n2 + n3])
        ops.linear(c4.view(Z, n4, -1), ctx.w.linear(sp + ".fc4", key=sp + ".fc4+le", extra_bias=le[2]), out=cc[:, n2 + n3:])
        return dict(c1_feat=c1, cc=cc, outs=[], dims=(Z, Hi, Wi, h, w, n2, n3, n4), ref=self._ref(Hi, Wi))

    def interact(self, a, i, x):
        """interaction i (vit_adapter.py:393-418): the extractor(s) that read encoder block ADAPTER_IDX[i]'s tokens x."""
        Z, Hi, Wi, h, w, n2, n3, n4 = a["dims"]
        a["cc"] = self._extractor(f"adapter.interactions.{i}.extractor", a["cc"], a["ref"], x, h, w)
        if i == 3:
            for j in range(2):
                a["cc"] = self._extractor(f"adapter.interactions.3.extra_extractors.{j}", a["cc"], a["ref"], x, h, w)
        a["outs"].append(x.view(Z, h, w, -1))  # the block's patch tokens as an NHWC map, read in place (batch-strided: the intrinsics token follows each item)

    LATERAL = "mask2former.model.pixel_decoder.adapter_1.0"  # the only consumer of f1 (video_seg_decoder.py: the FPN lateral of the 1/4 level)

    def _lateral_weights(self):
        """f1 = BN1(resize_x4(x1) + ConvT_up(c2) + fc1(c1)) (vit_adapter.py:420-436) is read by ONE layer, the pixel decoder's lateral
        1 x 1 convolution 1024 -> 256 -- and everything in between is linear per pixel, so the composition is applied instead:
            lateral(f1) = resize_x4(Wa x1) + ConvT'(c2) + Wc c1_feat + const,   Wa = W_lat diag(s1),  ConvT'[phase] = Wa W_up[phase],
            Wc = Wa W_fc1 (256 x 64),  const = W_lat b1 + Wa (b_up + b_fc1) + b_lat
        (a bilinear resize commutes with a per-pixel linear map).  86 GFLOP at 2 x 512^2 become 19 -- the 8192 x 4096 x 1024
        conv-transpose was the largest single launch of the segmentation chain -- and three 134 MB maps (fc1(c1), the conv-transpose
        output, f1) are never written.  Composed in fp64, rounded once."""
        W_ = self.ctx.w
        key = "adapter.lateral_fused"
        if key + ".y1" not in W_.lin:
            d = torch.float64
            w_lat = W_.t(self.LATERAL + ".weight").reshape(256, -1).to(d)
            b_lat = W_.t(self.LATERAL + ".bias").to(d) if (self.LATERAL + ".bias") in W_.sd else torch.zeros(256, dtype=d, device=W_.dev)
            s1, b1 = (t.to(d) for t in W_.bn_affine("adapter.norm1"))
            w_up, b_up = W_.t("adapter.up.weight").to(d), W_.t("adapter.up.bias").to(d)          # [ci, co, 2, 2]
            w_fc1 = W_.t("adapter.spm.fc1.weight").reshape(w_up.shape[1], -1).to(d)
            b_fc1 = W_.t("adapter.spm.fc1.bias").to(d)
            wa = w_lat * s1[None, :]
            W_.lin[key + ".y1"] = ops.pack_matrix(wa.float(), None, W_.split)
            W_.lin[key + ".up"] = ops.pack_conv_transpose(torch.einsum("pc,icyx->ipyx", wa, w_up).float(), None, W_.split)
            W_.lin[key + ".c1"] = ops.pack_matrix((wa @ w_fc1).float(), (w_lat @ b1 + wa @ (b_up + b_fc1) + b_lat).float(), W_.split)
        return W_.lin[key + ".y1"], W_.lin[key + ".up"], W_.lin[key + ".c1"]

    def finish(self, a, with_f1: bool = False):
        """-> [f1 | None, f2, f3, f4]; a["lat1"] = the pixel decoder's lateral convolution of f1 (fp32 [Z,4h,4w,256], before its
        GroupNorm), computed without f1 (_lateral_weights).  with_f1: also materialise f1 itself (the reference's return value:
        forward(), intermediates for the parity tests) -- nothing on the model's path reads it."""
        ctx = self.ctx
        Z, Hi, Wi, h, w, n2, n3, n4 = a["dims"]
        cc = a["cc"]
        # bf16 mode: the level-2 slice is copied out as bf16, so that the conv-transpose below runs on the LDS-DMA GEMM instead of the
        # fp32-A kernel; bf16x3 keeps fp32.  The three levels are slices of the token sequence, read in place (batch-strided views)
        c2 = (cc[:, :n2] if ctx.split else cc[:, :n2].to(ctx.act)).view(Z, 2 * h, 2 * w, -1)
        c3 = cc[:, n2:n2 + n3].view(Z, h, w, -1)
        c4 = cc[:, n2 + n3:].view(Z, h // 2, w // 2, -1)
        x1, x2, x3, x4 = a["outs"]
        pw_y1, pw_up, pw_c1 = self._lateral_weights()
        y1 = ops.linear(x1.view(Z, h * w, -1), pw_y1, out_dtype=torch.float32).view(Z, h, w, 256)
        t1 = ops.linear(a["c1_feat"].view(Z, 16 * h * w, -1), pw_c1, out_dtype=torch.float32).view(Z, 4 * h, 4 * w, 256)
        u1 = ops.resize_bilinear(y1, (4 * h, 4 * w), False, addend=t1, out_dtype=torch.float32)
        a["lat1"] = ops.conv_transpose2d(c2, pw_up, out_dtype=torch.float32, residual=u1)
        s2, b2 = ctx.w.bn_affine("adapter.norm2")
        s3, b3 = ctx.w.bn_affine("adapter.norm3")
        s4, b4 = ctx.w.bn_affine("adapter.norm4")
        f1 = None
        if with_f1:
            s1, b1 = ctx.w.bn_affine("adapter.norm1")
            c1 = ops.linear(a["c1_feat"], ctx.w.linear("adapter.spm.fc1"), out_dtype=torch.float32)  # [Z,4h,4w,1024]
            c1 = ops.conv_transpose2d(c2, ctx.w.convT("adapter.up"), out_dtype=torch.float32, residual=c1)
            f1 = ops.resize_bilinear(x1, (4 * h, 4 * w), False, addend=c1, ch_scale=s1, ch_shift=b1, out_dtype=ctx.act)
        f2 = ops.resize_bilinear(x2, (2 * h, 2 * w), False, addend=c2, ch_scale=s2, ch_shift=b2, out_dtype=ctx.act)
        f3 = ops.affine_add(x3, c3, s3, b3, out_dtype=ctx.act)
        f4 = ops.resize_bilinear(x4, (h // 2, w // 2), False, addend=c4, ch_scale=s4, ch_shift=b4, out_dtype=ctx.act)
        return [f1, f2, f3, f4]

    def forward_nhwc(self, img: torch.Tensor, img8: torch.Tensor, all_feat: Sequence[torch.Tensor]):
        """img [Z,3,Hi,Wi]; all_feat: 24 x [Z, N, 1024] fp32 (strided views).  Returns 4 NHWC maps [Z,h_l,w_l,1024]."""
        a = self.spm(img8)
        for i, idx in enumerate(ADAPTER_IDX):
            self.interact(a, i, all_feat[idx])
        return self.finish(a, with_f1=True)

    def forward(self, x, all_feat):
        """reference signature (vit_adapter.py:393): returns [f1..f4] as NCHW-shaped (channels-last) tensors."""
        img8 = ops.pack_image_nhwc(x.contiguous().float(), self.ctx.act, ops.image_channels(self.ctx.split))
        return [f.permute(0, 3, 1, 2) for f in self.forward_nhwc(x, img8, all_feat)]

    __call__ = forward


# ==================================================================================================
# Mask2Former (mask2former/video_seg_decoder.py)
# ==================================================================================================
def _sine_pos_2d(h, w, npf=128):
    """VideoMask2FormerSinePositionEmbedding (video_seg_decoder.py:704-735), normalize=True -> [h*w, 2*npf]."""
    y = torch.arange(1, h + 1, dtype=torch.float32)[:, None].expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32)[None, :].expand(h, w)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).reshape(h * w, 2 * npf)


def _sine_pos_3d(t, h, w, npf=128):
    """VideoMask2Former3DSinePositionEmbedding (video_seg_decoder.py:628-679) -> [t*h*w, 2*npf]."""
    z = torch.arange(1, t + 1, dtype=torch.float32)[:, None, None].expand(t, h, w)
    y = torch.arange(1, h + 1, dtype=torch.float32)[None, :, None].expand(t, h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32)[None, None, :].expand(t, h, w)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[:, -1:, :] + eps) * scale
    x = x / (x[:, :, -1:] + eps) * scale
    z = z / (z[-1:, :, :] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    dim_tz = torch.arange(npf * 2, dtype=torch.float32)
    dim_tz = 10000 ** (2 * torch.div(dim_tz, 2, rounding_mode="floor") / (npf * 2))
    px, py, pz = x[..., None] / dim_t, y[..., None] / dim_t, z[..., None] / dim_tz
    f = lambda p_: torch.stack((p_[..., 0::2].sin(), p_[..., 1::2].cos()), dim=4).flatten(3)
    return (torch.cat((f(py), f(px)), dim=3) + f(pz)).reshape(t * h * w, 2 * npf)


class VideoMask2FormerForVideoSegmentationOutput(dict):
    """Attribute-style output like the reference's ModelOutput (video_seg_decoder.py:2464-2471)."""

    __getattr__ = dict.get


class VideoMask2FormerForVideoSegmentation:
    def __init__(self, ctx: _Ctx, num_queries=100):
        self.ctx, self.num_queries = ctx, num_queries
        # None, or nine uint8 [B, Q, T*h*w] masks (1 = blocked) that the masked cross-attention layers use INSTEAD of thresholding their
        # own mask logits (sigmoid < 0.5, video_seg_decoder.py:1461-1478): tests feed the oracle's masks to show that a logit difference
        # above 1e-3 is a flipped borderline mask pixel and nothing else.  Eager mode only.
        self.forced_attn_masks = None
        self.record_attn_masks = None  # parity diagnostics (tests): a list that receives the nine boolean masks the layers attend through
        self.heads = 8

    # ---- pixel decoder (video_seg_decoder.py:2072-2196), feats NHWC [N2, h_l, w_l, 1024], strides 4,8,16,32
    def _pixel_decoder(self, feats, lat1=None):
        """lat1: adapter_1.0(feats[0]) when the caller already holds it (CroCoViTAdapter.finish composes it without materialising the
        1024-channel 1/4-resolution map; feats[0] may then be None)"""
        ctx = self.ctx
        pd = "mask2former.model.pixel_decoder"
        N2 = feats[1].shape[0]
        lv = list(feats)[::-1][:3]
        shapes = [(f.shape[1], f.shape[2]) for f in lv]
        S = sum(a * b for a, b in shapes)
        key = ("pd_const", tuple(shapes))
        if key not in ctx.cache:
            le = ctx.w.t(pd + ".level_embed").cpu()
            pos = torch.cat([_sine_pos_2d(a, b) + le[i] for i, (a, b) in enumerate(shapes)], 0)
            pts = []
            for (hh, ww) in shapes:
                ry, rx = torch.meshgrid(torch.linspace(0.5, hh - 0.5, hh), torch.linspace(0.5, ww - 0.5, ww), indexing="ij")
                pts.append(torch.stack((rx.reshape(-1) / ww, ry.reshape(-1) / hh), -1))
            ref = torch.cat(pts, 0)[:, None, :].expand(-1, 3, -1).contiguous()
            ctx.cache[key] = (pos.contiguous().to(ctx.dev), ref.to(ctx.dev))
        pos, ref = ctx.cache[key]
        hs = torch.empty((N2, S, 256), dtype=torch.float32, device=ctx.dev)
        o = 0
        for lvl, x in enumerate(lv):
            e = ops.linear(x, ctx.w.linear(f"{pd}.input_projections.{lvl}.0"), out_dtype=torch.float32)
            e = ops.groupnorm(e, ctx.w.v(f"{pd}.input_projections.{lvl}.1.weight"), ctx.w.v(f"{pd}.input_projections.{lvl}.1.bias"))
            n = shapes[lvl][0] * shapes[lvl][1]
            hs[:, o:o + n] = e.view(N2, n, 256)  # layout plumbing: concatenate levels
            o += n
        # hs: fp32 residual stream; hsb: the same tokens in the GEMM-input dtype (a bf16 copy written by the producing LayerNorm in
        # bf16 mode, so that every projection runs on the LDS-DMA GEMM instead of the fp32-A kernel).  The query-position term of
        # the offset / weight projection, W (hs + pos) = W hs + W pos, is a constant [S, 288] residual.
        hsb = hs if ctx.split else hs.to(ctx.act)
        for i in range(6):
            p = f"{pd}.encoder.layers.{i}"
            wkey = p + ".self_attn.offs_aw0"
            if wkey not in ctx.w.lin:
                w_oa = torch.cat((ctx.w.t(p + ".self_attn.sampling_offsets.weight"), ctx.w.t(p + ".self_attn.attention_weights.weight")), 0)
                b_oa = torch.cat((ctx.w.t(p + ".self_attn.sampling_offsets.bias"), ctx.w.t(p + ".self_attn.attention_weights.bias")), 0)
                ctx.w.lin[wkey] = ops.pack_matrix(w_oa, None, ctx.split)
                ctx.cache[wkey] = (w_oa, b_oa)
            rkey = (wkey, tuple(shapes))
            if rkey not in ctx.cache:
                w_oa, b_oa = ctx.cache[wkey]
                ctx.cache[rkey] = (pos @ w_oa.t() + b_oa).contiguous()
            pres = ctx.cache[rkey]
            offs_aw = ops.linear(hsb, ctx.w.lin[wkey], out_dtype=torch.float32, residual=pres if N2 == 1 else pres[None].expand(N2, -1, -1))
            value = ops.linear(hsb, ctx.w.linear(p + ".self_attn.value_proj"), out_dtype=ctx.act)
            samp = ops.msdeform_sample(value, offs_aw, ref, shapes, 8, 4, ctx.act)
            a = ops.linear(samp, ctx.w.linear(p + ".self_attn.output_proj"), out_dtype=torch.float32, residual=hs)
            hs, hsb = ctx.ln2(p + ".self_attn_layer_norm", a, 1e-5)
            f_ = ops.linear(hsb, ctx.w.linear(p + ".fc1"), out_dtype=ctx.act, act=ACT_RELU)
            f_ = ops.linear(f_, ctx.w.linear(p + ".fc2"), out_dtype=torch.float32, residual=hs)
            hs, hsb = ctx.ln2(p + ".final_layer_norm", f_, 1e-5)
        outs, o = [], 0
        for (hh, ww) in shapes:
            outs.append(hs[:, o:o + hh * ww].contiguous().view(N2, hh, ww, 256))
            o += hh * ww
        lat = lat1 if lat1 is not None else ops.linear(feats[0], ctx.w.linear(pd + ".adapter_1.0"), out_dtype=torch.float32)
        up = ops.resize_bilinear(outs[-1], (lat.shape[1], lat.shape[2]), False)
        out = ops.groupnorm(lat, ctx.w.v(pd + ".adapter_1.1.weight"), ctx.w.v(pd + ".adapter_1.1.bias"), addend=up, out_dtype=ctx.act)
        out = ops.conv2d(out, ctx.w.conv(pd + ".layer_1.0"), pad=1, out_dtype=torch.float32)
        out = ops.groupnorm(out, ctx.w.v(pd + ".layer_1.1.weight"), ctx.w.v(pd + ".layer_1.1.bias"), relu=True, out_dtype=ctx.act)
        mask_features = ops.linear(out, ctx.w.linear(pd + ".mask_projection"), out_dtype=torch.float32)
        return mask_features, outs

    def _mask_predictor(self, inter, mask_features_bt, size):
        """VideoMask2FormerMaskPredictor.forward (video_seg_decoder.py:1448-1480).  inter [B,Q,256] fp32 (layer-normed);
        mask_features_bt [B, T*H4*W4, 256].  Returns (mask logits [B,T,H4,W4,Q] fp32, attention mask uint8 [B,Q,T*h*w])."""
        ctx = self.ctx
        p = "mask2former.model.transformer_module.decoder.mask_predictor.mask_embedder"
        e = ops.linear(inter, ctx.w.linear(p + ".0.0"), out_dtype=ctx.act, act=ACT_RELU)
        e = ops.linear(e, ctx.w.linear(p + ".1.0"), out_dtype=ctx.act, act=ACT_RELU)
        if ctx.split:
            e = ops.linear(e, ctx.w.linear(p + ".2.0"), out_dtype=torch.float32)
            B, Q, Cc = e.shape
            hi, lo, kpad, x3 = ops.split_bf16(e.view(B * Q, Cc), True, want_x3=True)
        else:  # bf16 mode: the embedding leaves the GEMM as the bf16 "weight" plane of the mask product (256 = 4 x 64: no K padding)
            hi, lo, x3 = ops.linear(e, ctx.w.linear(p + ".2.0"), out_dtype=torch.bfloat16), None, None
            B, Q, Cc = hi.shape
            kpad = Cc
        m = ops.bmm_nt(mask_features_bt, hi.view(B, Q, kpad), None if lo is None else lo.view(B, Q, kpad), Q, Cc, b_x3=x3)
        m = m.view(B, self._T, self._H4, self._W4, Q)
        am = ops.m2f_attn_mask(m, size) if size is not None else None
        return m, am

    def _decoder(self, ms, mask_features, B, T):
        """TransformerModule.forward + MaskedAttentionDecoder.forward (video_seg_decoder.py:1506-1575,1204-1360)."""
        ctx = self.ctx
        tm = "mask2former.model.transformer_module"
        dp = tm + ".decoder"
        H4, W4 = mask_features.shape[1], mask_features.shape[2]
        self._T, self._H4, self._W4 = T, H4, W4
        mf_bt = mask_features.view(B, T * H4 * W4, 256)
        if not ctx.split:
            mf_bt = mf_bt.to(ctx.act)  # one cast per forward: the ten mask products then run on the bf16 LDS-DMA GEMM
        sizes = [(f.shape[1], f.shape[2]) for f in ms]
        key = ("m2f_pos3d", T, tuple(sizes))
        if key not in ctx.cache:
            ctx.cache[key] = [_sine_pos_3d(T, a, b).contiguous().to(ctx.dev) for (a, b) in sizes]
        pos3 = ctx.cache[key]
        le = ctx.w.v(tm + ".level_embed.weight")
        if "ones256" not in ctx.cache:
            ctx.cache["ones256"] = torch.ones(256, device=ctx.dev)
        ones = ctx.cache["ones256"]
        # Keys / values of the three memory levels for all nine layers up front: layer idx reads level idx % 3, and its
        # key projection Wk (feats + pos) + bk = Wk feats + (Wk pos + bk) has an input-independent second term, so one GEMM per
        # level yields K and V of its three layers (N = 6 x 256) with that constant riding in as the residual -- 3 launches on
        # the bf16 LDS-DMA path instead of 18 fp32-A ones plus the position adds
        feats = []
        for i in range(3):
            f_ = ops.affine_add(ms[i], None, ones, le[i].contiguous(), out_dtype=ctx.act)
            feats.append(f_.view(B, T * sizes[i][0] * sizes[i][1], 256))
        kvs = []
        for lvl in range(3):
            wkey, key = ("m2f_kvw", lvl), ("m2f_kv", T, sizes[lvl])
            if wkey not in ctx.cache:   # shape-independent part, kept after release_source_weights()
                wk, wv, bk, bv = [], [], [], []
                for idx in (lvl, lvl + 3, lvl + 6):
                    wi, bi = ctx.w.t(f"{dp}.layers.{idx}.cross_attn.in_proj_weight"), ctx.w.t(f"{dp}.layers.{idx}.cross_attn.in_proj_bias")
                    wk.append(wi[256:512]); wv.append(wi[512:]); bk.append(bi[256:512]); bv.append(bi[512:])
                wkk = torch.cat(wk, 0)
                ctx.cache[wkey] = (ops.pack_matrix(torch.cat((wkk, torch.cat(wv, 0)), 0), None, ctx.split), wkk, torch.cat(bk, 0), torch.cat(bv, 0))
            pw, wkk, bkk, bvv = ctx.cache[wkey]
            if key not in ctx.cache:
                posk = pos3[lvl] @ wkk.t() + bkk                                      # [S, 768]: constant folding, fp32
                ctx.cache[key] = torch.cat((posk, bvv[None].expand(posk.shape[0], -1)), 1).contiguous()
            res = ctx.cache[key]
            kv = ops.linear(feats[lvl], pw, out_dtype=ctx.act, residual=res if B == 1 else res[None].expand(B, -1, -1))
            kvs.append(kv.view(B, -1, 6, 8, 32))
        Q = self.num_queries
        qf = ctx.w.v(tm + ".queries_features.weight")
        qe = ctx.w.v(tm + ".queries_embedder.weight")
        hs = qf.unsqueeze(0).expand(B, -1, -1).contiguous()
        hsb = hs if ctx.split else hs.to(ctx.act)   # GEMM-input copy of the residual stream (see _pixel_decoder)
        d = 32
        inter, interb = ctx.ln2(dp + ".layernorm", hs, 1e-5)
        masks, inters = [], [inter]
        m, am = self._mask_predictor(interb, mf_bt, sizes[0])
        masks.append(m)
        for idx in range(9):
            p = f"{dp}.layers.{idx}"
            lvl = idx % 3
            if p + ".cross_attn.q" not in ctx.w.lin:
                wi, bi = ctx.w.t(p + ".cross_attn.in_proj_weight"), ctx.w.t(p + ".cross_attn.in_proj_bias")
                ctx.w.lin[p + ".cross_attn.q"] = ops.pack_matrix(wi[:256], None, ctx.split)
                # W (hs + query_pos) + b = W hs + (W query_pos + b): the query-position term is a constant [Q, N] residual
                ctx.cache[p + ".cross_attn.qres"] = (qe @ wi[:256].t() + bi[:256]).contiguous()
                wq, wk_ = ctx.w.t(p + ".self_attn.q_proj.weight"), ctx.w.t(p + ".self_attn.k_proj.weight")
                bq, bk_ = ctx.w.t(p + ".self_attn.q_proj.bias"), ctx.w.t(p + ".self_attn.k_proj.bias")
                wqk = torch.cat((wq, wk_), 0)
                ctx.w.lin[p + ".self_attn.qk0"] = ops.pack_matrix(wqk, None, ctx.split)
                ctx.cache[p + ".self_attn.qkres"] = (qe @ wqk.t() + torch.cat((bq, bk_), 0)).contiguous()
            bres = (lambda r: r if B == 1 else r[None].expand(B, -1, -1))
            # masked cross-attention (nn.MultiheadAttention, :975-983): q from hs + query pos, k from feats + pos3d, v from feats
            q = ops.linear(hsb, ctx.w.lin[p + ".cross_attn.q"], out_dtype=ctx.act, residual=bres(ctx.cache[p + ".cross_attn.qres"])).view(B, Q, 8, d)
            j = idx // 3
            k, v = kvs[lvl][:, :, j], kvs[lvl][:, :, 3 + j]
            if self.forced_attn_masks is not None:  # parity diagnostics (tests): attend through given masks instead of the thresholded ones
                fm = self.forced_attn_masks[idx].to(ctx.dev, torch.uint8)
                am = torch.zeros((B, Q, ((fm.shape[-1] + 63) // 64) * 64), dtype=torch.uint8, device=ctx.dev)
                am[..., :fm.shape[-1]] = fm
            if self.record_attn_masks is not None:
                self.record_attn_masks.append(am.clone())
            a = ops.attention(q, k, v, heads=8, head_dim=d, scale=d ** -0.5, mask=am, split3=ctx.split)
            a = ops.linear(a, ctx.w.linear(p + ".cross_attn.out_proj"), out_dtype=torch.float32, residual=hs)
            hs, hsb = ctx.ln2(p + ".cross_attn_layer_norm", a, 1e-5)
            # self-attention, DETR style (:782-912): pos added to q and k, v from hs
            qk = ops.linear(hsb, ctx.w.lin[p + ".self_attn.qk0"], out_dtype=ctx.act, residual=bres(ctx.cache[p + ".self_attn.qkres"])).view(B, Q, 2, 8, d)
            v = ops.linear(hsb, ctx.w.linear(p + ".self_attn.v_proj"), out_dtype=ctx.act).view(B, Q, 8, d)
            a = ops.attention(qk[:, :, 0], qk[:, :, 1], v, heads=8, head_dim=d, scale=d ** -0.5, split3=ctx.split)
            a = ops.linear(a, ctx.w.linear(p + ".self_attn.out_proj"), out_dtype=torch.float32, residual=hs)
            hs, hsb = ctx.ln2(p + ".self_attn_layer_norm", a, 1e-5)
            f_ = ops.linear(hsb, ctx.w.linear(p + ".fc1"), out_dtype=ctx.act, act=ACT_RELU)
            f_ = ops.linear(f_, ctx.w.linear(p + ".fc2"), out_dtype=torch.float32, residual=hs)
            hs, hsb = ctx.ln2(p + ".final_layer_norm", f_, 1e-5)
            inter, interb = ctx.ln2(dp + ".layernorm", hs, 1e-5)
            m, am = self._mask_predictor(interb, mf_bt, sizes[(idx + 1) % 3])
            masks.append(m)
            inters.append(inter)
        class_logits = ops.linear(interb, ctx.w.linear("mask2former.class_predictor"), out_dtype=torch.float32)
        return class_logits, masks[-1], masks, inters

    def forward_nhwc(self, feats_nhwc: Sequence[torch.Tensor], B: int, T: int, lat1=None):
        mask_features, ms = self._pixel_decoder(feats_nhwc, lat1)
        class_logits, mask_cl, all_masks, inters = self._decoder(ms, mask_features, B, T)
        out = VideoMask2FormerForVideoSegmentationOutput(
            loss=None, class_queries_logits=class_logits,
            masks_queries_logits=mask_cl.permute(0, 4, 1, 2, 3),  # [B,Q,T,h,w] view of the channel-last buffer
            auxiliary_logits=None, attentions=None, word_embeddings=None,
        )
        out["_masks_channel_last"] = mask_cl
        out["_mask_features"], out["_ms"] = mask_features, ms
        return out

    def forward(self, multi_scale_feat: List[torch.Tensor], word_embeddings=None, mask_labels=None, class_labels=None, **kw):
        """reference signature (video_seg_decoder.py:2351-2361); multi_scale_feat: 4 x [B,T,C,h,w]."""
        assert word_embeddings is None and mask_labels is None and class_labels is None, "inference path only"
        B, T = multi_scale_feat[0].shape[:2]
        feats = [f.flatten(0, 1).permute(0, 2, 3, 1).contiguous() for f in multi_scale_feat]
        return self.forward_nhwc(feats, B, T)

    __call__ = forward


# ==================================================================================================
# whole model (model.py:31-389)
# ==================================================================================================
_PTS0_MAIN = os.environ.get("SIU3R_PTS0_MAIN", "0") == "1"
_HEAD_MAP = os.environ.get("SIU3R_HEAD_MAP", "")
_KV_PLANES = not os.environ.get("SIU3R_NO_KV_PLANES")  # A/B: k / v written pre-split by the q | k | v projections (bf16x3)
_CONV_PLANES = not os.environ.get("SIU3R_NO_CONV_PLANES")  # A/B: pre-split planes between the convolutions of the paired DPT heads
_DEC_QKVX = not os.environ.get("SIU3R_NO_DEC_QKVX")  # A/B: the decoder's self-attention q | k | v and the other side's cross-attention k | v as one launch
_ADAPTER_LATE = int(os.environ.get("SIU3R_ADAPTER_LATE", "0"))  # A/B: run the ViT-Adapter interactions behind the encoder instead of beside it
_HEAD_PAIRS = os.environ.get("SIU3R_NO_HEAD_PAIRS", "0") != "1"  # the two heads of a kind as grouped launches (V == 2, B == 1)
_PAIR_MAP = os.environ.get("SIU3R_PAIR_MAP", "")  # A/B: streams of the (Gaussian pair, pts3d pair) chains, e.g. "0,m"
_PTS0_OWN = os.environ.get("SIU3R_PTS0_OWN", "0") == "1"  # A/B: a fifth stream for the pts3d head of view 0 (needs GPU_MAX_HW_QUEUES >= 5 to overlap)
_DEC_PER_LAYER = os.environ.get("SIU3R_DEC_PER_LAYER", "0") == "1"
_PIPE_PTSR = int(os.environ.get("SIU3R_PIPE_PTSR", "0"))  # forward_async: the head stream (0 / 1) the pts3d head of views 1.. queues on


class _PendingForward:
    """Handle of SIU3RModel.forward_async(): the device side of the step is enqueued; result() does the host half (once)."""

    def __init__(self, finish):
        self._finish, self._out, self.done = finish, None, False

    def result(self):
        if not self.done:
            self._out, self.done, self._finish = self._finish(), True, None
        return self._out


class _Run:
    """State of one pass through the network body (the static buffers of a captured shape, in graph mode)."""

    def __init__(self, images, K):
        self.images, self.K = images, K
        self.img_bv = self.img8 = self.enc = self.adapter = self.dstate = self.dec = self.ms = self.seg = self.gaussians = None
        self.gs, self.pts, self.raw = [None, None], [None, None], None
        self.want_f1 = False  # the adapter's 1024-channel 1/4-resolution map (an inspection output: nothing on the path reads it)


class SIU3RModel:
    def __init__(self, state_dict: Dict[str, torch.Tensor], image_size=(512, 512), precision="bf16", device="cuda",
                 num_queries=100, seg_threshold=0.5, label_ids_to_fuse=(0, 1), sh_degree=4):
        if not torch.cuda.is_available():
            raise RuntimeError("SIU3RModel (siu3r_amd) needs an MI355X GPU: there is no CPU fallback")
        self.image_size = tuple(image_size)
        self.precision = precision
        self.seg_threshold, self.label_ids_to_fuse = seg_threshold, set(label_ids_to_fuse)
        sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in state_dict.items()}
        self._w = _Weights(sd, torch.device(device), precision == "bf16x3")
        self._ctx = _Ctx(self._w, precision)
        self.backbone = AsymmetricCroCo(self._ctx)
        self.adapter = CroCoViTAdapter(self._ctx, image_size)
        self.mask2former = VideoMask2FormerForVideoSegmentation(self._ctx, num_queries)
        self.downstream_head1 = _DPTHead(self._ctx, "downstream_head1", gs=False)
        self.downstream_head2 = _DPTHead(self._ctx, "downstream_head2", gs=False)
        self.gaussian_param_head1 = _DPTHead(self._ctx, "gaussian_param_head1", gs=True)
        self.gaussian_param_head2 = _DPTHead(self._ctx, "gaussian_param_head2", gs=True)
        # the two heads of a kind as one sequence of grouped launches (one pair, one item: the benchmarked shape)
        self.gs_pair = _DPTHeadPair(self._ctx, ("gaussian_param_head1", "gaussian_param_head2"), gs=True)
        self.pts_pair = _DPTHeadPair(self._ctx, ("downstream_head1", "downstream_head2"), gs=False)
        self.gaussian_adapter = UnifiedGaussianAdapter(sh_degree=sh_degree)
        self.processor = pp.VideoMask2FormerImageProcessor()
        self.raw_gs_dim = (sh_degree + 1) ** 2 * 3 + 3 + 4 + 1
        self.use_graph = os.environ.get("SIU3R_NO_GRAPH", "0") != "1"
        self._graphs: Dict = {}
        self.pipeline_depth = max(1, int(os.environ.get("SIU3R_PIPELINE_DEPTH", "2")))  # forward_async: steps that may be pending at once
        self._next_slot = 0
        self._inflight: Dict = {}
        self._timeline = None  # a list collects (stage, start event, end event) of the next graph-replayed forward

    def eval(self):
        return self

    def release_source_weights(self):
        """Drop the fp32 source state dict once every layer has been packed (after the first forward)."""
        self._w.sd = {k: v for k, v in self._w.sd.items() if v.numel() <= 65536}

    def forward(self, context_views_images, context_views_intrinsics, mask_labels=None, class_labels=None,
                enable_query_class_logit_lift=False, return_intermediates=False):
        """reference signature model.py:314-321 (V = 2) / model_multi.py (V >= 2 context views).

        The network body is a fixed launch sequence per input shape, made of independent chains: encoder ->
        {ViT-Adapter + Mask2Former | decoder -> four DPT heads} -> Gaussian adapter.  Eager mode enqueues the chains
        on separate HIP streams.  From the third call of a shape on, every chain is replayed from its own HIP graph
        (captured at the second call; the first packs the weights), the graphs launched on those same streams: at
        batch 1 the ~1100 launches of a pass otherwise cost more host time than GPU time, and ONE graph for the
        whole body serialises its branches on this runtime.  The panoptic post-process (host-visible segment table)
        always runs eagerly afterwards.  SIU3R_NO_GRAPH=1 / SIU3R_NO_STREAMS=1 disable the two mechanisms.

        forward() = forward_async(...).result() on the caller's stream."""
        if self._inflight.get(0) is not None and not self._inflight[0].done:
            raise RuntimeError("SIU3RModel.forward: a forward_async() of slot 0 is still pending; call its result() first")
        return self._submit(context_views_images, context_views_intrinsics, mask_labels, class_labels, enable_query_class_logit_lift,
                            return_intermediates, slot=None).result()

    def forward_async(self, context_views_images, context_views_intrinsics, mask_labels=None, class_labels=None,
                      enable_query_class_logit_lift=False):
        """forward() split at its one host synchronisation: enqueues the whole device side of the step (network body + the device
        stage of the panoptic post-process + the asynchronous read-back of the segment table) and returns a handle; handle.result()
        waits for the table (an event), builds the reference's `segments_info` lists and returns what forward() returns.

        Up to `pipeline_depth` (default 2) steps may be pending.  Each pending step owns a SLOT: its own static activation buffers,
        chain graphs, split-K workspaces and HIP streams, so that step n+1's encoder / decoder (latency-bound: 136-288 workgroups per
        launch on 256 CUs) runs beside step n's heads and Mask2Former branch (throughput-bound) instead of behind them.  Steps are
        independent -- the same arithmetic, the same launch sequence per step, bit-identical results to forward() (tests/
        test_model_gpu.py::test_forward_async_interleaved) -- only their position in time changes.  The inputs are copied into the
        slot's buffers on entry: the caller may overwrite them as soon as this returns (stream-ordered)."""
        slot = self._next_slot
        prev = self._inflight.get(slot)
        if prev is not None and not prev.done:
            raise RuntimeError(f"SIU3RModel.forward_async: {self.pipeline_depth} steps are already pending; call result() on the oldest first")
        self._next_slot = (slot + 1) % self.pipeline_depth
        h = self._submit(context_views_images, context_views_intrinsics, mask_labels, class_labels, enable_query_class_logit_lift, False, slot=slot)
        self._inflight[slot] = h
        return h

    def _submit(self, context_views_images, context_views_intrinsics, mask_labels, class_labels, enable_query_class_logit_lift,
                return_intermediates, slot):
        assert mask_labels is None and class_labels is None, "inference path only (training losses are out of scope)"
        ctx = self._ctx
        images = context_views_images.to(ctx.dev)
        K = context_views_intrinsics.to(ctx.dev)
        B, V, _, H, W = images.shape
        key = (B, V, H, W, images.dtype, K.dtype)
        eager = (not self.use_graph) or return_intermediates or ops.kernel_timer_active() or torch.cuda.is_current_stream_capturing()
        ent = None if eager else self._graphs.get(key)
        caller = torch.cuda.current_stream()
        # slot None (forward()): state slot 0, and the chains are joined back into the caller's stream before the tail.  slot s
        # (forward_async): state slot s, and the caller's stream carries ONLY encoder -> decoder: the heads, the joins and the tail stay
        # on the side streams, so that the next step's encoder (enqueued on the caller's stream right behind this step's decoder) starts
        # while this step's heads run; result() joins.  (One set of streams for all slots on purpose: this runtime multiplexes streams
        # onto 4 hardware queues in order of first use, and a second set of streams ended up sharing queues with the first -- slot 1's
        # encoder then simply queued behind slot 0's tail; with 8 queues everything overlapped and the step got 10-35 % SLOWER.)
        sidx = 0 if slot is None else slot
        pipelined = slot is not None and ctx.concurrent
        pp_state = {"tail_stream": caller}

        def make_after_seg(st_, copy_out):
            # runs on the segmentation stream right behind Mask2Former: the panoptic device stage and the asynchronous read-back of its
            # tables overlap the heads; the host picks the tables up (an event) after everything is enqueued
            def after_seg():
                seg_o = VideoMask2FormerForVideoSegmentationOutput(st_.seg) if copy_out else st_.seg
                if copy_out:  # the logits leave the graphs' private memory, which the next replay overwrites
                    for k_ in ("class_queries_logits", "masks_queries_logits"):
                        seg_o[k_] = seg_o[k_].clone()
                        seg_o[k_].record_stream(caller)
                pend = self.processor.begin_panoptic(seg_o, threshold=self.seg_threshold, target_sizes=[(H, W)] * B,
                                                     label_ids_to_fuse=self.label_ids_to_fuse, host_slot=sidx)
                for k_ in ("tab", "probs", "kept_idx", "acc_list", "p256", "seg", "sem", "ins"):
                    pend[k_].record_stream(caller)  # allocated on the segmentation stream, read on the caller's
                pp_state["seg_out"], pp_state["pend"] = seg_o, pend
            return after_seg

        if eager or ent is None:
            need = {}
            if not eager:
                # first call of this shape: eager (packs weights, fills caches) -- and measures the split-K workspace every chain asks for
                ent = self._graphs[key] = {"graphs": None, "st": None, "slots": {}, "need": need}
            st = _Run(images, K)
            st.want_f1 = bool(return_intermediates)

            def run_eager(name, fn):
                if eager:
                    return fn()
                with ops.splitk_meter() as m:
                    fn()
                need[name] = m

            self._run_stages(st, run_eager, make_after_seg(st, False))
        else:
            sl = ent["slots"].get(sidx)
            if sl is None:
                sl = ent["slots"][sidx] = self._capture_slot(ent, key, images, K)
                if sidx == 0:
                    ent["graphs"], ent["st"] = sl["graphs"], sl["st"]  # (tools/ read the first slot through these)
            st = sl["st"]
            if sl.get("done") is not None:
                caller.wait_event(sl["done"])  # the slot's previous step (a no-op when its result() ran on this stream, as it normally has)
            st.images.copy_(images, non_blocking=True)
            st.K.copy_(K, non_blocking=True)
            tl = self._timeline

            def replay(name, fn):
                if tl is not None:  # tools/timeline.py: stage start / end events on the stage's own stream
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                fn() if name == "tail" else sl["graphs"][name].replay()
                if tl is not None:
                    e1.record()
                    tl.append((name, e0, e1))

            pp_state["tail_stream"] = self._run_stages(st, replay, make_after_seg(st, True), pipelined=pipelined)
            if pipelined:
                sl["done"] = torch.cuda.Event()
                sl["done"].record(pp_state["tail_stream"])
        # the Gaussians come from the eager tail (fresh memory)
        gaussians = st.gaussians
        tail_stream = pp_state["tail_stream"]

        def finish():
            cur = torch.cuda.current_stream()
            if tail_stream is not cur:
                cur.wait_stream(tail_stream)  # the Gaussian fields (pipelined: the tail ran on a head stream, every chain joined into it)
                for _, v_ in gaussians.items():
                    if isinstance(v_, torch.Tensor):
                        v_.record_stream(cur)
            seg_out = pp_state["seg_out"]
            if cur is not caller:
                for k_ in ("tab", "probs", "kept_idx", "acc_list", "p256", "seg", "sem", "ins"):
                    pp_state["pend"][k_].record_stream(cur)
            results = self.processor.finish_panoptic(pp_state["pend"])
            if not eager:
                # the channel-last logits / mask features / memory levels live in the graphs' private memory, which the next replay
                # overwrites: they served the post-process above and do not leave with the result
                for k_ in ("_masks_channel_last", "_mask_features", "_ms"):
                    seg_out.pop(k_, None)
            g_, masks, infos, qcl, qscores = pp.post_process_gaussians(gaussians, results, B, V, H, W, enable_query_class_logit_lift)
            if return_intermediates:
                _, _, decs = self.backbone._assemble(st.enc, st.dec)
                all1 = [t[:, 0, :-1] for t in st.enc["av"]]
                all2 = [t[:, 1, :-1] for t in st.enc["av"]]
                self._last = dict(dec1=decs[0], dec2=decs[1], decs=decs, all_feat1=all1, all_feat2=all2, ms=st.ms, lat1=st.adapter["lat1"], pts1=st.pts[0],
                                  pts2=st.pts[1], gs_raw1=st.gs[0].reshape(B, H, W, -1), gs_raw2=st.gs[1].reshape(-1, H, W, st.gs[1].shape[-1]), seg_out=st.seg)
            if enable_query_class_logit_lift:
                return g_, seg_out, masks, infos, qscores
            return g_, seg_out, masks, infos

        return _PendingForward(finish)

    def _capture_slot(self, ent, key, images, K):
        """Static buffers + one HIP graph per chain for one slot of this shape.  The chain's split-K workspace is its own (the chain
        graphs replay concurrently), has exactly the size the chain's GEMMs asked for in the eager pass (none if none splits), exists
        before the capture starts (no fill node in the graph) and is kept next to the graph."""
        ctx = self._ctx
        sl = {"st": _Run(images.clone(), K.clone()), "graphs": {}, "ws": {}}
        st = sl["st"]
        torch.cuda.synchronize()
        conc, ctx.concurrent = ctx.concurrent, False  # a captured chain is serial by construction
        try:
            for name, fn in self._stages(st):
                if name == "tail":  # the tail (2 launches) stays eager: its outputs are then fresh tensors, not graph memory
                    continue
                m = ent["need"].get(name)
                ws = sl["ws"][name] = m.workspace(ctx.dev) if m is not None else None
                g = torch.cuda.CUDAGraph()
                # thread_local: only this thread's calls belong to the capture (a RCCL watchdog thread of a multi-GPU job must not
                # be able to invalidate it)
                with ops.splitk_scope(ws):
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        fn()
                sl["graphs"][name] = g
        finally:
            ctx.concurrent = conc
        torch.cuda.synchronize()
        return sl

    # ---- the chains of the network body.  Each stage only reads what earlier stages left in the _Run
    def _stages(self, st):
        bb = self.backbone

        def dec_pre():
            st.dstate = bb.decode_begin(st.enc)
            B, V, _, H, W = st.images.shape
            st.raw = torch.empty((B, V, H * W, self.raw_gs_dim), dtype=torch.float32, device=self._ctx.dev)

        def dec_post():
            st.dec = bb.decode_end(st.dstate)

        dec = [("dec_pre", dec_pre)]
        if self._merged_decoder(st) and not _DEC_PER_LAYER:
            # all merged layers as ONE chain graph (a graph boundary costs ~10 us on the critical path; SIU3R_DEC_PER_LAYER=1 keeps one
            # graph per layer for tools/timeline.py)
            dec += [("dec_all", lambda: [bb.decode_layer_merged(st.dstate, i) for i in range(bb.dec_depth)])]
        for i in range(bb.dec_depth):
            if self._merged_decoder(st):
                if _DEC_PER_LAYER:
                    dec += [(f"dec{i}", lambda i=i: bb.decode_layer_merged(st.dstate, i))]
            else:
                dec += [(f"decA{i}", lambda i=i: bb.decode_side(st.dstate, i, 0)), (f"decB{i}", lambda i=i: bb.decode_side(st.dstate, i, 1))]
        dec += [("dec_post", dec_post)]
        ad = self.adapter
        bounds = [0] + [i + 1 for i in ADAPTER_IDX]  # encoder segments end at the blocks the adapter taps

        def enc_seg(k):
            bb.encode_blocks(st.enc, bounds[k], bounds[k + 1])
            if k == len(ADAPTER_IDX) - 1:
                if bounds[-1] < bb.enc_depth:
                    bb.encode_blocks(st.enc, bounds[-1], bb.enc_depth)
                bb.encode_end(st.enc)

        def interact(k):
            B, V, _, H, W = st.images.shape
            t = st.enc["all_feat"][ADAPTER_IDX[k]].view(B * V, -1, bb.enc_embed_dim)[:, :-1]
            ad.interact(st.adapter, k, t)

        enc = [("enc_begin", lambda: self._s_encode_begin(st)), ("spm", lambda: setattr(st, "adapter", ad.spm(st.img8)))]
        for k in range(len(ADAPTER_IDX)):
            enc += [(f"enc{k}", lambda k=k: enc_seg(k)), (f"int{k}", lambda k=k: interact(k))]
        if self._paired_heads(st):
            heads = [("gs", lambda: self._s_head_pair(st, True)), ("pts", lambda: self._s_head_pair(st, False))]
        else:
            heads = [("gs0", lambda: self._s_head(st, 0)), ("gsr", lambda: self._s_head(st, 1)), ("pts0", lambda: self._s_head(st, 2)),
                     ("ptsr", lambda: self._s_head(st, 3))]
        return enc + [("seg", lambda: self._s_seg(st))] + dec + heads + [("tail", lambda: self._s_tail(st))]

    def _paired_heads(self, st) -> bool:
        B, V = st.images.shape[:2]
        return _HEAD_PAIRS and V == 2 and B == 1

    def _merged_decoder(self, st) -> bool:
        return self._ctx.fold and st.images.shape[1] == 2

    def _run_stages(self, st, run, after_seg=None, pipelined=False):
        """Enqueue the stages with their fork/join edges.  run(name, fn) either calls fn (eager) or replays its graph.  after_seg(): enqueued
        on the segmentation stream right behind Mask2Former (the device half of the panoptic post-process).  Returns the stream the
        tail ran on (every chain is joined into it).  pipelined (forward_async): the current stream carries only encoder -> decoder;
        the pts3d head of the other views, the joins and the tail go to the head streams, so that whatever the caller enqueues next on
        the current stream (the next step's encoder) is not ordered behind this step's heads."""
        ctx = self._ctx
        stages = dict(self._stages(st))
        main = torch.cuda.current_stream()
        par = ctx.concurrent
        # encoder on the current stream; the ViT-Adapter rides beside it on its own stream: the spatial prior module needs
        # only the image, interaction k only the tokens of encoder block ADAPTER_IDX[k].  The rest of the segmentation
        # branch (adapter output maps, Mask2Former) then overlaps the decoder and the heads
        run("enc_begin", stages["enc_begin"])
        seg_stream = ctx.side_stream(1) if par else main
        if par:
            seg_stream.wait_stream(main)
        late = _ADAPTER_LATE if par else 0  # A/B: 1 = every interaction behind the whole encoder, 2 = the spatial prior module too
        if late < 2:
            with torch.cuda.stream(seg_stream):
                run("spm", stages["spm"])
        for k in range(len(ADAPTER_IDX)):
            run(f"enc{k}", stages[f"enc{k}"])
            if late:
                continue
            if par:
                seg_stream.wait_stream(main)
            with torch.cuda.stream(seg_stream):
                run(f"int{k}", stages[f"int{k}"])
        if late:
            seg_stream.wait_stream(main)
            with torch.cuda.stream(seg_stream):
                if late >= 2:
                    run("spm", stages["spm"])
                for k in range(len(ADAPTER_IDX)):
                    run(f"int{k}", stages[f"int{k}"])
        with torch.cuda.stream(seg_stream):
            run("seg", stages["seg"])
            if after_seg is not None:
                after_seg()
        # decoder: per layer, view 0 and the other views are independent chains (joined after every layer)
        run("dec_pre", stages["dec_pre"])
        dside = ctx.side_stream(0) if par else main
        if "dec_all" in stages:
            run("dec_all", stages["dec_all"])
        for i in range(self.backbone.dec_depth):
            if self._merged_decoder(st):  # both sides of the pair in grouped launches on the main stream
                if "dec_all" not in stages:
                    run(f"dec{i}", stages[f"dec{i}"])
                continue
            if par:
                dside.wait_stream(main)
            run(f"decA{i}", stages[f"decA{i}"])
            with torch.cuda.stream(dside):
                run(f"decB{i}", stages[f"decB{i}"])
            if par:
                main.wait_stream(dside)
        run("dec_post", stages["dec_post"])
        # the two Gaussian heads (5.8 ms each in bf16x3) on their own streams; the two pts3d heads (3.3 + 1.7 ms) back to back on the main
        # stream: three chains of about equal length.  (A fourth stream for pts0 shares a hardware queue with one of the others on this
        # runtime -- it then ran alone AFTER the Gaussian heads; with GPU_MAX_HW_QUEUES=8 it overlaps, but the extra concurrency slows
        # the encoder / decoder critical path by more than it saves: 25.3 -> 31.3 ms.)
        hs = [ctx.side_stream(2 + i) if par else main for i in range(2)]
        for s_ in hs:
            if s_ is not main:
                s_.wait_stream(main)
        # pts0 rides on the segmentation stream, behind Mask2Former: that chain ends ~2.5 ms after the heads start, and the head of view 0
        # then overlaps the other three instead of running alone after them (21.8 -> 20.9 ms body at B = 1; SIU3R_PTS0_MAIN=1 restores
        # the two pts3d heads back to back on the main stream)
        pts0_stream = (ctx.side_stream(4) if _PTS0_OWN else seg_stream) if (par and not _PTS0_MAIN) else main
        if pts0_stream is not main:
            pts0_stream.wait_stream(main)  # the decoder's outputs
        pipelined = pipelined and par
        ptsr_stream = hs[_PIPE_PTSR] if pipelined else main  # pipelined: behind one of the Gaussian heads instead of on the encoder's stream
        placement = list(zip(("gs0", "gsr", "ptsr", "pts0"), hs + [ptsr_stream, pts0_stream]))
        if "gs" in stages:  # paired heads: two chains (both Gaussian heads | both pts3d heads)
            pick = {"0": hs[0], "1": hs[1], "m": main, "s": seg_stream}
            k_gs, k_pts = (_PAIR_MAP.split(",") if (_PAIR_MAP and par) else ("0", "1" if pipelined else "m"))
            placement = [("gs", pick[k_gs] if par else main), ("pts", pick[k_pts] if par else main)]
            for _, s_ in placement:
                if s_ is seg_stream and par:
                    seg_stream.wait_stream(main)
        elif _HEAD_MAP and par:  # A/B: SIU3R_HEAD_MAP="0,1,m,s" = stream of gs0, gsr, ptsr, pts0 (0 / 1 head streams, m main, s segmentation), in launch order
            pick = {"0": hs[0], "1": hs[1], "m": main, "s": seg_stream}
            placement = [(n_, pick[k_]) for n_, k_ in zip(("gs0", "gsr", "ptsr", "pts0"), _HEAD_MAP.split(","))]
            for _, s_ in placement:
                if s_ is seg_stream:
                    seg_stream.wait_stream(main)
        for name, s_ in placement:
            with torch.cuda.stream(s_):
                run(name, stages[name])
        tail_stream = hs[1] if pipelined else main
        for s_ in hs + [seg_stream, pts0_stream]:
            if s_ is not tail_stream:
                tail_stream.wait_stream(s_)
        with torch.cuda.stream(tail_stream):
            run("tail", stages["tail"])
        return tail_stream

    def _s_encode_begin(self, st):
        ctx = self._ctx
        B, V, _, H, W = st.images.shape
        st.img_bv = st.images.reshape(B * V, 3, H, W).contiguous().float()
        st.img8 = ops.pack_image_nhwc(st.img_bv, ctx.act, ops.image_channels(ctx.split))
        st.enc = self.backbone.encode_begin(st.images, st.K)

    def _s_seg(self, st):
        # the adapter is shared by all views (model.py:342-345): one (b,v)-batched pass; Mask2Former sees T = V frames
        B, V, _, H, W = st.images.shape
        st.ms = self.adapter.finish(st.adapter, with_f1=st.want_f1)
        st.seg = self.mask2former.forward_nhwc(st.ms, B, V, lat1=st.adapter["lat1"])

    @staticmethod
    def _rest_views(t):
        """[B, V, ...] -> views 1..V-1 as one batch [B*(V-1), ...] (a view when possible, else a copy)."""
        r = t[:, 1:]
        return r[0] if t.shape[0] == 1 else (r[:, 0] if t.shape[1] == 2 else r.reshape(-1, *t.shape[2:]))

    def _s_head(self, st, i):
        """i = 0: GS head of view 0 (gaussian_param_head1), 1: GS head of views 1.. (head2), 2 / 3: the pts3d heads."""
        B, V, _, H, W = st.images.shape
        first = i in (0, 2)
        toks = [(t[:, 0] if first else self._rest_views(t))[..., :-1, :] for t in st.dec["layers"]]
        if i < 2:
            img8 = st.img8.view(B, V, H, W, st.img8.shape[-1])
            img = (img8[:, 0] if first else self._rest_views(img8)).contiguous()
            head = self.gaussian_param_head1 if first else self.gaussian_param_head2
            # the last GEMM writes straight into the [B, V, H*W, 83] buffer the Gaussian adapter reads (no 350 MB concat),
            # whenever this head's views form one strided batch of it
            dst = st.raw[:, 0] if first else (st.raw[:, 1] if V == 2 else (st.raw[0, 1:] if B == 1 else None))
            st.gs[0 if first else 1] = head.forward_gs(toks, img, H, W, out=dst)
        else:
            head = self.downstream_head1 if first else self.downstream_head2
            st.pts[0 if first else 1] = head.forward_pts3d(toks, H, W)["pts3d"]

    def _s_head_pair(self, st, gs: bool):
        """both heads of a kind (view 0 -> head1, view 1 -> head2) as grouped launches"""
        B, V, _, H, W = st.images.shape
        toks = [t[..., :-1, :] for t in st.dec["layers"]]  # [B, 2, N, C]: the patch tokens of both views
        if gs:
            self.gs_pair.forward_gs(toks, st.img8.view(B, V, H, W, st.img8.shape[-1]), H, W, out=st.raw)
            st.gs = [st.raw[:, 0], st.raw[:, 1]]
        else:
            pts = self.pts_pair.forward_pts3d(toks, H, W)
            st.pts = [pts[:, 0], pts[:, 1]]

    def _s_tail(self, st):
        B, V, _, H, W = st.images.shape
        cat = lambda a, b_: torch.cat((a.reshape(B, 1, H * W, -1), b_.reshape(B, V - 1, H * W, -1)), dim=1)
        raw = st.raw if (V == 2 or B == 1) else cat(st.gs[0], st.gs[1])
        st.gaussians = self.gaussian_adapter.forward(cat(st.pts[0], st.pts[1]), raw)

    __call__ = forward


class SIU3RMultiViewModel(SIU3RModel):
    """V >= 2 context views (reference src/models/model_multi.py): view 0 is decoded by dec_blocks / *_head1, every
    other view by dec_blocks2 / *_head2 (batched here); ViT-Adapter per view, Mask2Former over T = V frames.
    The forward signature and the returned tuple are SIU3RModel's; with V = 2 the two classes compute the same thing."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.backbone = AsymmetricCroCoMulti(self._ctx)
