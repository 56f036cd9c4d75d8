"""CPU oracle: plain-PyTorch fp32 restatement of the SIU3R inference forward.

THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  It restates, function by function,
the reference algorithm of /root/reference (file:line cited on every function) with no
reference imports, so that it can travel to the GPU box.  Parity pin: every function here is
checked against the reference's own Python modules (imported in the build container through
``tests/golden/_ref_import.py``) by ``tests/golden/make_golden.py`` /
``tests/test_oracle_pins.py``; the outputs of those runs are committed as fixtures under
``tests/golden/``.  The third-party rasterizers are *not* covered here (see oracle/raster_ref.c:
"parity unpinned").

Weights are a flat dict keyed by the reference state-dict names.
Activations are torch CPU fp32 tensors.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

T = torch.Tensor
W = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------
def _lin(w: W, name: str, x: T) -> T:
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias"))


def _ln(w: W, name: str, x: T, eps: float) -> T:
    return F.layer_norm(x, (x.shape[-1],), w[name + ".weight"], w[name + ".bias"], eps)


def _conv(w: W, name: str, x: T, stride=1, padding=0) -> T:
    return F.conv2d(x, w[name + ".weight"], w.get(name + ".bias"), stride=stride, padding=padding)


def _bn(w: W, name: str, x: T) -> T:
    """eval-mode (Sync)BatchNorm, eps 1e-5 (vit_adapter.py:208-261,357-360)."""
    return F.batch_norm(
        x, w[name + ".running_mean"], w[name + ".running_var"], w[name + ".weight"], w[name + ".bias"],
        training=False, eps=1e-5,
    )


# ----------------------------------------------------------------------------------------
# RoPE2D  (croco/curope/curope.cpp:11-47 rope_2d_cpu; kernels.cu:17-82; pos_embed.py:126-179)
# ----------------------------------------------------------------------------------------
def rope2d_table(max_pos: int, head_dim: int, base: float = 100.0, fwd: float = 1.0):
    """cos/sin[p, q] with inv_freq_q = fwd / base**(q/Q), Q = head_dim/4 (kernels.cu:46-55)."""
    Q = head_dim // 4
    q = torch.arange(Q, dtype=torch.float32)
    inv_freq = fwd / torch.pow(torch.tensor(base, dtype=torch.float32), q / Q)
    p = torch.arange(max_pos, dtype=torch.float32)[:, None]
    ang = p * inv_freq[None, :]
    return torch.cos(ang), torch.sin(ang)


def rope2d(tokens: T, positions: T, base: float = 100.0, fwd: float = 1.0) -> T:
    """tokens [B, H, N, D] (the layout Attention hands to self.rope, blocks.py:98-103),
    positions [B, N, 2] int64 (y, x).  Head vector = [u_Y v_Y u_X v_X], quarters of D;
    (u, v) -> (u c - v s, v c + u s) with angle pos[axis] * inv_freq (curope.cpp:27-43)."""
    B, H, N, D = tokens.shape
    Q = D // 4
    cos, sin = rope2d_table(int(positions.max()) + 1, D, base, fwd)
    out = torch.empty_like(tokens)
    for axis in range(2):
        c = cos[positions[:, :, axis]][:, None]  # [B,1,N,Q]
        s = sin[positions[:, :, axis]][:, None]
        u = tokens[..., axis * 2 * Q : axis * 2 * Q + Q]
        v = tokens[..., axis * 2 * Q + Q : axis * 2 * Q + 2 * Q]
        out[..., axis * 2 * Q : axis * 2 * Q + Q] = u * c - v * s
        out[..., axis * 2 * Q + Q : axis * 2 * Q + 2 * Q] = v * c + u * s
    return out


# ----------------------------------------------------------------------------------------
# ViT blocks (croco/blocks.py:58-191)
# ----------------------------------------------------------------------------------------
def attention(w: W, p: str, x: T, pos: T, heads: int) -> T:
    """blocks.py:94-112: fused qkv, RoPE on q,k, softmax(q k^T * d^-0.5) v, proj."""
    B, N, C = x.shape
    d = C // heads
    qkv = _lin(w, p + ".qkv", x).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q, k = rope2d(q, pos), rope2d(k, pos)
    a = (q @ k.transpose(-2, -1)) * d ** -0.5
    a = a.softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return _lin(w, p + ".proj", o)


def cross_attention(w: W, p: str, xq: T, mem: T, qpos: T, kpos: T, heads: int) -> T:
    """blocks.py:149-169."""
    B, Nq, C = xq.shape
    Nk = mem.shape[1]
    d = C // heads
    q = _lin(w, p + ".projq", xq).reshape(B, Nq, heads, d).permute(0, 2, 1, 3)
    k = _lin(w, p + ".projk", mem).reshape(B, Nk, heads, d).permute(0, 2, 1, 3)
    v = _lin(w, p + ".projv", mem).reshape(B, Nk, heads, d).permute(0, 2, 1, 3)
    q, k = rope2d(q, qpos), rope2d(k, kpos)
    a = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, Nq, C)
    return _lin(w, p + ".proj", o)


def mlp(w: W, p: str, x: T) -> T:
    """blocks.py:73-79, exact-erf GELU."""
    return _lin(w, p + ".fc2", F.gelu(_lin(w, p + ".fc1", x)))


def enc_block(w: W, p: str, x: T, pos: T, heads: int) -> T:
    """blocks.py:127-130, LayerNorm eps 1e-6 (croco.py:35)."""
    x = x + attention(w, p + ".attn", _ln(w, p + ".norm1", x, 1e-6), pos, heads)
    return x + mlp(w, p + ".mlp", _ln(w, p + ".norm2", x, 1e-6))


def dec_block(w: W, p: str, x: T, y: T, xpos: T, ypos: T, heads: int) -> T:
    """blocks.py:186-191."""
    x = x + attention(w, p + ".attn", _ln(w, p + ".norm1", x, 1e-6), xpos, heads)
    y_ = _ln(w, p + ".norm_y", y, 1e-6)
    x = x + cross_attention(w, p + ".cross_attn", _ln(w, p + ".norm2", x, 1e-6), y_, xpos, ypos, heads)
    return x + mlp(w, p + ".mlp", _ln(w, p + ".norm3", x, 1e-6))


def patch_positions(B: int, h: int, w_: int) -> T:
    """blocks.py:195-207 PositionGetter: cartesian_prod(y, x), int64."""
    y = torch.arange(h)
    x = torch.arange(w_)
    return torch.cartesian_prod(y, x).view(1, h * w_, 2).expand(B, -1, 2).clone()


def backbone(w: W, images: T, intrinsics: T, enc_heads=16, dec_heads=12):
    """AsymmetricCroCo.forward (backbone_croco.py:263-339) for a V=2 context.

    images [B,2,3,H,W], intrinsics [B,2,3,3].  Returns dict with the reference's tuple members:
    feat1/2, all_feat1/2 (24 each), dec1/2 (13 each), all with the intrinsics token stripped."""
    B, V, _, H, Wd = images.shape
    assert V == 2
    assert H % 16 == 0 and Wd % 16 == 0  # patch_embed.py:21-22
    emb = _lin(w, "backbone.intrinsic_encoder", intrinsics.flatten(2))  # [B,2,1024]  (:278)
    img = torch.cat((images[:, 0], images[:, 1]), dim=0)  # both views on batch (:173-183)
    itok = torch.cat((emb[:, 0:1], emb[:, 1:2]), dim=0)  # [2B,1,1024]
    x = _conv(w, "backbone.patch_embed.proj", img, stride=16)  # patch_embed.py:19-29
    h, w_ = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    pos = patch_positions(2 * B, h, w_)
    x = torch.cat((x, itok), dim=1)  # intrinsics token appended (:147)
    add_pos = pos[:, 0:1, :].clone()
    add_pos[:, :, 0] += pos[:, -1, 0].unsqueeze(-1) + 1  # (y = h, x = 0)  (:148-150)
    pos = torch.cat((pos, add_pos), dim=1)
    n_enc = len({k.split(".")[2] for k in w if k.startswith("backbone.enc_blocks.")})
    all_feat = []
    for i in range(n_enc):
        x = enc_block(w, f"backbone.enc_blocks.{i}", x, pos, enc_heads)
        all_feat.append(x)
    x = _ln(w, "backbone.enc_norm", x, 1e-6)
    f1, f2 = x[:B], x[B:]
    pos1, pos2 = pos[:B], pos[B:]
    # decoder (:231-255)
    out1, out2 = [f1], [f2]
    g1 = _lin(w, "backbone.decoder_embed", f1)
    g2 = _lin(w, "backbone.decoder_embed", f2)
    n_dec = len({k.split(".")[2] for k in w if k.startswith("backbone.dec_blocks.")})
    for i in range(n_dec):
        n1 = dec_block(w, f"backbone.dec_blocks.{i}", g1, g2, pos1, pos2, dec_heads)
        n2 = dec_block(w, f"backbone.dec_blocks2.{i}", g2, g1, pos2, pos1, dec_heads)
        g1, g2 = n1, n2
        out1.append(g1)
        out2.append(g2)
    out1[-1] = _ln(w, "backbone.dec_norm", out1[-1], 1e-6)
    out2[-1] = _ln(w, "backbone.dec_norm", out2[-1], 1e-6)
    strip = lambda t: t[:, :-1]  # (:306-315)
    return dict(
        feat1=strip(f1), feat2=strip(f2),
        all_feat1=[strip(t[:B]) for t in all_feat], all_feat2=[strip(t[B:]) for t in all_feat],
        dec1=[strip(t) for t in out1], dec2=[strip(t) for t in out2],
        hw=(h, w_),
    )


def backbone_multi(w: W, images: T, intrinsics: T, enc_heads=16, dec_heads=12):
    """AsymmetricCroCoMulti.forward (backbone_croco.py:541-590) for V >= 2 context views.

    Same weights as the pair model.  Decoder (_decoder, :486-539): view 0 runs dec_blocks with the tokens of all
    other views as memory; views 1..V-1 run dec_blocks2, each with the other views' tokens in ascending view order
    (generate_ctx_views, :500-506).  Returns per-view lists: feats[v], all_feats[v][24], decs[v][13], token stripped."""
    B, V, _, H, Wd = images.shape
    assert V >= 2 and H % 16 == 0 and Wd % 16 == 0
    emb = _lin(w, "backbone.intrinsic_encoder", intrinsics.flatten(2))  # [B,V,1024]
    img = images.flatten(0, 1)                                           # (b v) major (:552)
    itok = emb.flatten(0, 1).unsqueeze(1)
    x = _conv(w, "backbone.patch_embed.proj", img, stride=16)
    h, w_ = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    pos = patch_positions(B * V, h, w_)
    x = torch.cat((x, itok), dim=1)
    add_pos = pos[:, 0:1, :].clone()
    add_pos[:, :, 0] += pos[:, -1, 0].unsqueeze(-1) + 1
    pos = torch.cat((pos, add_pos), dim=1)
    n_enc = len({k.split(".")[2] for k in w if k.startswith("backbone.enc_blocks.")})
    all_feat = []
    for i in range(n_enc):
        x = enc_block(w, f"backbone.enc_blocks.{i}", x, pos, enc_heads)
        all_feat.append(x)
    x = _ln(w, "backbone.enc_norm", x, 1e-6)
    L = x.shape[1]
    feat = x.view(B, V, L, -1)
    pose = pos.view(B, V, L, 2)

    def ctx_views(t):  # [B,V,L,C] -> [B,V,(V-1)L,C]: for view i the other views, ascending
        return torch.stack([torch.cat([t[:, j] for j in range(V) if j != i], dim=1) for i in range(V)], dim=1)

    pos_ctx = ctx_views(pose)
    outs = [feat]
    g = _lin(w, "backbone.decoder_embed", x).view(B, V, L, -1)
    n_dec = len({k.split(".")[2] for k in w if k.startswith("backbone.dec_blocks.")})
    for i in range(n_dec):
        gc = ctx_views(g)
        f1 = dec_block(w, f"backbone.dec_blocks.{i}", g[:, 0], gc[:, 0], pose[:, 0], pos_ctx[:, 0], dec_heads)
        f2 = dec_block(w, f"backbone.dec_blocks2.{i}", g[:, 1:].flatten(0, 1), gc[:, 1:].flatten(0, 1),
                       pose[:, 1:].flatten(0, 1), pos_ctx[:, 1:].flatten(0, 1), dec_heads)
        g = torch.cat((f1.unsqueeze(1), f2.view(B, V - 1, L, -1)), dim=1)
        outs.append(g)
    outs[-1] = _ln(w, "backbone.dec_norm", outs[-1].flatten(0, 1), 1e-6).view(B, V, L, -1)
    strip = lambda t: t[..., :-1, :]
    af = [t.view(B, V, L, -1) for t in all_feat]
    return dict(
        feats=[strip(feat[:, v]) for v in range(V)],
        all_feats=[[strip(t[:, v]) for t in af] for v in range(V)],
        decs=[[strip(t[:, v]) for t in outs] for v in range(V)],
        hw=(h, w_),
    )


# ----------------------------------------------------------------------------------------
# DPT heads (heads/dpt_block.py, dpt_head.py:36-79, dpt_gs_head.py:121-171, postprocess.py:22-63)
# ----------------------------------------------------------------------------------------
def _rcu(w: W, p: str, x: T) -> T:
    """ResidualConvUnit_custom (dpt_block.py:126-147), bn=False, ReLU non-inplace."""
    o = _conv(w, p + ".conv1", F.relu(x), padding=1)
    o = _conv(w, p + ".conv2", F.relu(o), padding=1)
    return o + x


def _fusion(w: W, p: str, x0: T, x1: T | None) -> T:
    """FeatureFusionBlock_custom.forward (dpt_block.py:198-237): bilinear x2 align_corners=True."""
    out = x0
    if x1 is not None:
        out = out + _rcu(w, p + ".resConfUnit1", x1)
    out = _rcu(w, p + ".resConfUnit2", out)
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return _conv(w, p + ".out_conv", out)


def dpt_trunk(w: W, p: str, tokens: Sequence[T], H: int, Wd: int) -> T:
    """Shared DPT trunk up to path_1 (dpt_head.py:45-72): hooks [0,6,9,12]."""
    hooks = [0, 6, 9, 12]
    nh, nw = H // 16, Wd // 16
    layers = []
    for i, hk in enumerate(hooks):
        t = tokens[hk]
        B, N, C = t.shape
        l = t.transpose(1, 2).reshape(B, C, nh, nw)
        a = f"{p}.dpt.act_postprocess.{i}"
        l = _conv(w, a + ".0", l)
        if i == 0:
            l = F.conv_transpose2d(l, w[a + ".1.weight"], w[a + ".1.bias"], stride=4)
        elif i == 1:
            l = F.conv_transpose2d(l, w[a + ".1.weight"], w[a + ".1.bias"], stride=2)
        elif i == 3:
            l = _conv(w, a + ".1", l, stride=2, padding=1)
        l = F.conv2d(l, w[f"{p}.dpt.scratch.layer_rn.{i}.weight"], None, padding=1)
        layers.append(l)
    s = f"{p}.dpt.scratch"
    path4 = _fusion(w, s + ".refinenet4", layers[3], None)[:, :, : layers[2].shape[2], : layers[2].shape[3]]
    path3 = _fusion(w, s + ".refinenet3", path4, layers[2])
    path2 = _fusion(w, s + ".refinenet2", path3, layers[1])
    path1 = _fusion(w, s + ".refinenet1", path2, layers[0])
    return path1


def pts3d_head(w: W, p: str, tokens: Sequence[T], H: int, Wd: int) -> T:
    """regression head (dpt_block.py:357-371) + reg_dense_depth('exp') (postprocess.py:45-61).
    Returns pts3d [B,H,W,3]."""
    x = dpt_trunk(w, p, tokens, H, Wd)
    x = _conv(w, f"{p}.dpt.head.0", x, padding=1)
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.relu(_conv(w, f"{p}.dpt.head.2", x, padding=1))
    x = _conv(w, f"{p}.dpt.head.4", x)
    xyz = x.permute(0, 2, 3, 1)
    d = xyz.norm(dim=-1, keepdim=True)
    return xyz / d.clip(min=1e-8) * torch.expm1(d)


def gs_head(w: W, p: str, tokens: Sequence[T], img: T, H: int, Wd: int) -> T:
    """dpt_gs_head.py:121-171 + head 'gs_params' (dpt_block.py:382-392).  Returns [B, H*W, 83]."""
    x = dpt_trunk(w, p, tokens, H, Wd)
    direct = F.relu(_conv(w, f"{p}.dpt.input_merger.0", img, padding=3))
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) + direct
    x = F.relu(F.conv2d(x, w[f"{p}.dpt.head.0.weight"], None, padding=1))
    x = _conv(w, f"{p}.dpt.head.4", x)
    B, D = x.shape[0], x.shape[1]
    return x.reshape(B, D, H * Wd).transpose(1, 2)  # 'b d h w -> b (h w) d' (model.py:202)


def sh_mask(sh_degree: int = 4) -> T:
    """gaussian_adapter.py:64-71."""
    m = torch.ones((sh_degree + 1) ** 2, dtype=torch.float32)
    for deg in range(1, sh_degree + 1):
        m[deg ** 2 : (deg + 1) ** 2] = 0.1 * 0.25 ** deg
    return m


def gaussian_adapter(means: T, raw: T, sh_degree: int = 4, eps: float = 1e-8) -> Dict[str, T]:
    """UnifiedGaussianAdapter.forward (gaussian_adapter.py:81-110), quaternion_to_matrix (:11-33,
    xyzw), build_covariance (:36-47)."""
    d_sh = (sh_degree + 1) ** 2
    op, sc, rot, sh = raw.split((1, 3, 4, 3 * d_sh), dim=-1)
    op = op.sigmoid().squeeze(-1)
    sc = (0.001 * F.softplus(sc)).clamp_max(0.3)
    rn = rot / (rot.norm(dim=-1, keepdim=True) + eps)
    sh = sh.reshape(*sh.shape[:-1], 3, d_sh) * sh_mask(sh_degree)
    i, j, k, r = torch.unbind(rn, dim=-1)
    two_s = 2 / ((rn * rn).sum(dim=-1) + eps)
    R = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    ).reshape(*rn.shape[:-1], 3, 3)
    S = sc.diag_embed()
    cov = R @ S @ S.transpose(-1, -2) @ R.transpose(-1, -2)
    return dict(means=means, covariances=cov, harmonics=sh, opacities=op, scales=sc, rotations=rot)


# ----------------------------------------------------------------------------------------
# multi-scale deformable attention (vit_adapter/blocks.py:147-267 == mask2former/utils.py:8-59)
# ----------------------------------------------------------------------------------------
def reference_points(shapes: Sequence[Sequence[int]]) -> T:
    """blocks.py:10-24: pixel-centre points normalised by the level size, order (x, y)."""
    pts = []
    for (h, w_) in shapes:
        ry, rx = torch.meshgrid(
            torch.linspace(0.5, h - 0.5, h, dtype=torch.float32),
            torch.linspace(0.5, w_ - 0.5, w_, dtype=torch.float32), indexing="ij",
        )
        pts.append(torch.stack((rx.reshape(-1) / w_, ry.reshape(-1) / h), -1))
    return torch.cat(pts, 0)  # [Lq, 2]


def msdeform_core(value: T, shapes: Sequence[Sequence[int]], loc: T, aw: T) -> T:
    """value [B, S, h, d]; loc [B, Q, h, L, P, 2] in [0,1]; aw [B,Q,h,L,P] -> [B,Q,h*d].
    grid_sample bilinear / zeros / align_corners=False per level (blocks.py:217-267)."""
    B, S, h, d = value.shape
    _, Q, _, L, P, _ = loc.shape
    vals = value.split([a * b for a, b in shapes], dim=1)
    grids = 2 * loc - 1
    samp = []
    for lvl, (hh, ww) in enumerate(shapes):
        v = vals[lvl].flatten(2).transpose(1, 2).reshape(B * h, d, hh, ww)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        samp.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = aw.transpose(1, 2).reshape(B * h, 1, Q, L * P)
    out = (torch.stack(samp, dim=-2).flatten(-2) * aw).sum(-1).view(B, h * d, Q)
    return out.transpose(1, 2).contiguous()


def msdeform_attn(w: W, p: str, query: T, ref: T, feat: T, shapes, heads: int, levels: int, points: int) -> T:
    """MSDeformAttn.forward (blocks.py:147-213) / PixelDecoder variant (video_seg_decoder.py:1656-1722).
    ref [Lq, levels_ref, 2] broadcast over batch; 2-d reference points branch."""
    B, Q, C = query.shape
    S = feat.shape[1]
    value = _lin(w, p + ".value_proj", feat).view(B, S, heads, C // heads)
    off = _lin(w, p + ".sampling_offsets", query).view(B, Q, heads, levels, points, 2)
    aw = _lin(w, p + ".attention_weights", query).view(B, Q, heads, levels * points)
    aw = aw.softmax(-1).view(B, Q, heads, levels, points)
    norm = torch.tensor([[s[1], s[0]] for s in shapes], dtype=torch.float32)  # (w, h)
    loc = ref[None, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = msdeform_core(value, shapes, loc, aw)
    return _lin(w, p + ".output_proj", out)


# ----------------------------------------------------------------------------------------
# ViT-Adapter (vit_adapter/vit_adapter.py)
# ----------------------------------------------------------------------------------------
def _spm(w: W, x: T):
    """SpatialPriorModule.forward (vit_adapter.py:276-302)."""
    p = "adapter.spm"
    c = F.relu(_bn(w, p + ".stem.1", _conv(w, p + ".stem.0", x, stride=2, padding=1)))
    c = F.relu(_bn(w, p + ".stem.4", _conv(w, p + ".stem.3", c, padding=1)))
    c = F.relu(_bn(w, p + ".stem.7", _conv(w, p + ".stem.6", c, padding=1)))
    c1 = F.max_pool2d(c, kernel_size=3, stride=2, padding=1)
    c2 = F.relu(_bn(w, p + ".conv2.1", _conv(w, p + ".conv2.0", c1, stride=2, padding=1)))
    c3 = F.relu(_bn(w, p + ".conv3.1", _conv(w, p + ".conv3.0", c2, stride=2, padding=1)))
    c4 = F.relu(_bn(w, p + ".conv4.1", _conv(w, p + ".conv4.0", c3, stride=2, padding=1)))
    c1 = _conv(w, p + ".fc1", c1)
    c2 = _conv(w, p + ".fc2", c2).flatten(2).transpose(1, 2)
    c3 = _conv(w, p + ".fc3", c3).flatten(2).transpose(1, 2)
    c4 = _conv(w, p + ".fc4", c4).flatten(2).transpose(1, 2)
    return c1, c2, c3, c4


def _conv_ffn(w: W, p: str, x: T, H: int, Wd: int) -> T:
    """ConvFFN / DWConv (vit_adapter.py:16-59): depth-wise 3x3 on the three scales, then GELU."""
    x = _lin(w, p + ".fc1", x)
    B, N, C = x.shape
    n = N // 21
    outs = []
    for (a, b, hh, ww) in ((0, 16 * n, H * 2, Wd * 2), (16 * n, 20 * n, H, Wd), (20 * n, 21 * n, H // 2, Wd // 2)):
        t = x[:, a:b].transpose(1, 2).reshape(B, C, hh, ww)
        t = F.conv2d(t, w[p + ".dwconv.dwconv.weight"], w[p + ".dwconv.dwconv.bias"], padding=1, groups=C)
        outs.append(t.flatten(2).transpose(1, 2))
    x = F.gelu(torch.cat(outs, dim=1))
    return _lin(w, p + ".fc2", x)


def _extractor(w: W, p: str, c: T, ref: T, feat: T, H: int, Wd: int) -> T:
    """Extractor.forward (vit_adapter.py:96-121); LayerNorm eps 1e-6; 16 heads, 1 level, 4 points."""
    a = msdeform_attn(w, p + ".attn", _ln(w, p + ".query_norm", c, 1e-6), ref,
                      _ln(w, p + ".feat_norm", feat, 1e-6), [(H, Wd)], 16, 1, 4)
    c = c + a
    return c + _conv_ffn(w, p + ".ffn", _ln(w, p + ".ffn_norm", c, 1e-6), H, Wd)


def adapter(w: W, img: T, all_feat: Sequence[T]) -> List[T]:
    """CroCoViTAdapter.forward (vit_adapter.py:393-441) for one view.  Returns [f1..f4] NCHW."""
    B, _, Hi, Wi = img.shape
    H, Wd = Hi // 16, Wi // 16
    ref = reference_points([(Hi // 8, Wi // 8), (Hi // 16, Wi // 16), (Hi // 32, Wi // 32)])[:, None, :]
    c1, c2, c3, c4 = _spm(w, img)
    le = w["adapter.level_embed"]
    c2, c3, c4 = c2 + le[0], c3 + le[1], c4 + le[2]
    n2, n3 = c2.shape[1], c3.shape[1]
    c = torch.cat([c2, c3, c4], dim=1)
    outs = []
    dim = all_feat[0].shape[2]
    for i, idx in enumerate((5, 11, 17, 23)):
        x = all_feat[idx]
        c = _extractor(w, f"adapter.interactions.{i}.extractor", c, ref, x, H, Wd)
        if i == 3:
            for j in range(2):
                c = _extractor(w, f"adapter.interactions.3.extra_extractors.{j}", c, ref, x, H, Wd)
        outs.append(x.transpose(1, 2).reshape(B, dim, H, Wd))
    c2 = c[:, :n2].transpose(1, 2).reshape(B, dim, H * 2, Wd * 2)
    c3 = c[:, n2 : n2 + n3].transpose(1, 2).reshape(B, dim, H, Wd)
    c4 = c[:, n2 + n3 :].transpose(1, 2).reshape(B, dim, H // 2, Wd // 2)
    c1 = F.conv_transpose2d(c2, w["adapter.up.weight"], w["adapter.up.bias"], stride=2) + c1
    x1, x2, x3, x4 = outs
    x1 = F.interpolate(x1, scale_factor=4, mode="bilinear", align_corners=False)
    x2 = F.interpolate(x2, scale_factor=2, mode="bilinear", align_corners=False)
    x4 = F.interpolate(x4, scale_factor=0.5, mode="bilinear", align_corners=False)
    return [_bn(w, "adapter.norm1", c1 + x1), _bn(w, "adapter.norm2", c2 + x2),
            _bn(w, "adapter.norm3", c3 + x3), _bn(w, "adapter.norm4", c4 + x4)]


# ----------------------------------------------------------------------------------------
# Mask2Former (mask2former/video_seg_decoder.py)
# ----------------------------------------------------------------------------------------
def sine_pos_2d(h: int, w_: int, num_pos_feats: int = 128) -> T:
    """VideoMask2FormerSinePositionEmbedding (video_seg_decoder.py:704-735), normalize=True,
    scale 2*pi, temperature 10000.  Returns [2*npf, h, w]."""
    y = torch.arange(1, h + 1, dtype=torch.float32)[:, None].expand(h, w_)
    x = torch.arange(1, w_ + 1, dtype=torch.float32)[None, :].expand(h, w_)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)


def sine_pos_3d(t: int, h: int, w_: int, num_pos_feats: int = 128) -> T:
    """VideoMask2Former3DSinePositionEmbedding (video_seg_decoder.py:628-679).  [t, 2*npf, h, w]."""
    z = torch.arange(1, t + 1, dtype=torch.float32)[:, None, None].expand(t, h, w_)
    y = torch.arange(1, h + 1, dtype=torch.float32)[None, :, None].expand(t, h, w_)
    x = torch.arange(1, w_ + 1, dtype=torch.float32)[None, None, :].expand(t, h, w_)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[:, -1:, :] + eps) * scale
    x = x / (x[:, :, -1:] + eps) * scale
    z = z / (z[-1:, :, :] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    dim_tz = torch.arange(num_pos_feats * 2, dtype=torch.float32)
    dim_tz = 10000 ** (2 * torch.div(dim_tz, 2, rounding_mode="floor") / (num_pos_feats * 2))
    px, py, pz = x[..., None] / dim_t, y[..., None] / dim_t, z[..., None] / dim_tz
    f = lambda p_: torch.stack((p_[..., 0::2].sin(), p_[..., 1::2].cos()), dim=4).flatten(3)
    pos = torch.cat((f(py), f(px)), dim=3) + f(pz)
    return pos.permute(0, 3, 1, 2)


def pixel_decoder(w: W, feats: Sequence[T]):
    """VideoMask2FormerPixelDecoder.forward (video_seg_decoder.py:2072-2196).
    feats: 4 maps [N, 1024, h_l, w_l] (strides 4,8,16,32), N = B*T already folded.
    Returns (mask_features [N,256,H/4,W/4], [ms0 (/32), ms1 (/16), ms2 (/8)])."""
    pd = "mask2former.model.pixel_decoder"
    embeds, poss, shapes = [], [], []
    for lvl, x in enumerate(list(feats)[::-1][:3]):
        e = _conv(w, f"{pd}.input_projections.{lvl}.0", x)
        e = F.group_norm(e, 32, w[f"{pd}.input_projections.{lvl}.1.weight"], w[f"{pd}.input_projections.{lvl}.1.bias"], 1e-5)
        embeds.append(e)
        poss.append(sine_pos_2d(x.shape[2], x.shape[3]))
        shapes.append((x.shape[2], x.shape[3]))
    hs = torch.cat([e.flatten(2).transpose(1, 2) for e in embeds], 1)
    pos = torch.cat([p_.flatten(1).transpose(0, 1) + w[pd + ".level_embed"][i] for i, p_ in enumerate(poss)], 0)[None]
    ref = reference_points(shapes)[:, None, :].expand(-1, 3, -1)  # valid ratios are 1 (:1862-1881)
    for i in range(6):
        p = f"{pd}.encoder.layers.{i}"
        a = msdeform_attn(w, p + ".self_attn", hs + pos, ref, hs, shapes, 8, 3, 4)
        hs = _ln(w, p + ".self_attn_layer_norm", hs + a, 1e-5)
        f_ = _lin(w, p + ".fc2", F.relu(_lin(w, p + ".fc1", hs)))
        hs = _ln(w, p + ".final_layer_norm", hs + f_, 1e-5)
    outs, o = [], 0
    N = hs.shape[0]
    for (hh, ww) in shapes:
        outs.append(hs[:, o : o + hh * ww].transpose(1, 2).reshape(N, -1, hh, ww))
        o += hh * ww
    lat = F.conv2d(feats[0], w[pd + ".adapter_1.0.weight"], None)
    lat = F.group_norm(lat, 32, w[pd + ".adapter_1.1.weight"], w[pd + ".adapter_1.1.bias"], 1e-5)
    out = lat + F.interpolate(outs[-1], size=lat.shape[-2:], mode="bilinear", align_corners=False)
    out = F.conv2d(out, w[pd + ".layer_1.0.weight"], None, padding=1)
    out = F.relu(F.group_norm(out, 32, w[pd + ".layer_1.1.weight"], w[pd + ".layer_1.1.bias"], 1e-5))
    mask_features = _conv(w, pd + ".mask_projection", out)
    return mask_features, outs


def _mha(q: T, k: T, v: T, w_in: T, b_in: T, w_out: T, b_out: T, heads: int, mask: T | None) -> T:
    """nn.MultiheadAttention forward restated (seq-first inputs [L,B,C]); bool mask [B*h, Lq, Lk], True = blocked."""
    Lq, B, C = q.shape
    Lk = k.shape[0]
    d = C // heads
    wq, wk, wv = w_in.chunk(3, 0)
    bq, bk, bv = b_in.chunk(3, 0)
    Q = F.linear(q, wq, bq).reshape(Lq, B * heads, d).transpose(0, 1)
    K = F.linear(k, wk, bk).reshape(Lk, B * heads, d).transpose(0, 1)
    V = F.linear(v, wv, bv).reshape(Lk, B * heads, d).transpose(0, 1)
    a = (Q * d ** -0.5) @ K.transpose(1, 2)
    if mask is not None:
        a = a.masked_fill(mask, float("-inf"))
    a = a.softmax(-1)
    o = (a @ V).transpose(0, 1).reshape(Lq, B, C)
    return F.linear(o, w_out, b_out)


def _mask_predictor(w: W, p: str, hs: T, pix: T, size, heads: int):
    """VideoMask2FormerMaskPredictor.forward (video_seg_decoder.py:1448-1480).
    hs [Q,B,C] (already layer-normed), pix [B,T,C,H,W]."""
    e = hs.transpose(0, 1)
    for i in range(3):
        e = _lin(w, f"{p}.mask_embedder.{i}.0", e)
        if i < 2:
            e = F.relu(e)
    m = torch.einsum("bqc,btchw->bqthw", e, pix)
    b, q, t = m.shape[:3]
    am = F.interpolate(m.flatten(0, 1), size=size, mode="bilinear", align_corners=False)
    am = am.view(b, q, t, size[0], size[1]).sigmoid().flatten(2).unsqueeze(1).repeat(1, heads, 1, 1)
    am = (am.flatten(0, 1) < 0.5)
    return m, am


def m2f_decoder(w: W, ms: Sequence[T], mask_features: T, B: int, Tn: int, heads: int = 8):
    """VideoMask2FormerTransformerModule.forward (video_seg_decoder.py:1506-1575) +
    MaskedAttentionDecoder.forward (:1204-1360, post-norm layers :957-1025) + class head (:2386-2399).
    ms: 3 maps [B*T,256,h,w] (/32,/16,/8); mask_features [B*T,256,H/4,W/4]."""
    tm = "mask2former.model.transformer_module"
    C = mask_features.shape[1]
    pix = mask_features.view(B, Tn, C, *mask_features.shape[-2:])
    feats, poss, sizes = [], [], []
    for i in range(3):
        f_ = ms[i]
        hh, ww = f_.shape[-2:]
        sizes.append((hh, ww))
        pos3 = sine_pos_3d(Tn, hh, ww)[None].expand(B, -1, -1, -1, -1).flatten(3)  # [B,T,C,hw]
        poss.append(pos3.permute(1, 3, 0, 2).flatten(0, 1))  # [T*hw, B, C]
        f_ = f_.flatten(2) + w[tm + ".level_embed.weight"][i][None, :, None]
        feats.append(f_.view(B, Tn, C, hh * ww).permute(1, 3, 0, 2).flatten(0, 1))
    qf = w[tm + ".queries_features.weight"].unsqueeze(1).repeat(1, B, 1)
    qe = w[tm + ".queries_embedder.weight"].unsqueeze(1).repeat(1, B, 1)
    dp = tm + ".decoder"
    hs = qf
    inter = _ln(w, dp + ".layernorm", hs, 1e-5)
    inters, masks, used_masks = [inter], [], []
    m, am = _mask_predictor(w, dp + ".mask_predictor", inter, pix, sizes[0], heads)
    masks.append(m)
    n_layers = len({k.split(".")[5] for k in w if k.startswith(dp + ".layers.")})
    for idx in range(n_layers):
        p = f"{dp}.layers.{idx}"
        lvl = idx % 3
        am = am.clone()
        am[torch.where(am.sum(-1) == am.shape[-1])] = False  # fully blocked rows re-opened (:1306-1308)
        used_masks.append(am[::heads].to(torch.uint8))  # [B, Q, T*h*w] (identical over the heads): what layer idx attends through
        a = _mha(hs + qe, feats[lvl] + poss[lvl], feats[lvl], w[p + ".cross_attn.in_proj_weight"],
                 w[p + ".cross_attn.in_proj_bias"], w[p + ".cross_attn.out_proj.weight"],
                 w[p + ".cross_attn.out_proj.bias"], heads, am)
        hs = _ln(w, p + ".cross_attn_layer_norm", hs + a, 1e-5)
        # self-attention, DETR style (:782-912): pos added to q,k; q pre-scaled
        x = hs.permute(1, 0, 2)
        xp = x + qe.permute(1, 0, 2)
        Bq, L, _ = x.shape
        d = C // heads
        sh = lambda t_: t_.view(Bq, L, heads, d).transpose(1, 2)
        q_ = sh(_lin(w, p + ".self_attn.q_proj", xp) * d ** -0.5)
        k_ = sh(_lin(w, p + ".self_attn.k_proj", xp))
        v_ = sh(_lin(w, p + ".self_attn.v_proj", x))
        o = ((q_ @ k_.transpose(-1, -2)).softmax(-1) @ v_).transpose(1, 2).reshape(Bq, L, C)
        o = _lin(w, p + ".self_attn.out_proj", o).permute(1, 0, 2)
        hs = _ln(w, p + ".self_attn_layer_norm", hs + o, 1e-5)
        f_ = _lin(w, p + ".fc2", F.relu(_lin(w, p + ".fc1", hs)))
        hs = _ln(w, p + ".final_layer_norm", hs + f_, 1e-5)
        inter = _ln(w, dp + ".layernorm", hs, 1e-5)
        m, am = _mask_predictor(w, dp + ".mask_predictor", inter, pix, sizes[(idx + 1) % 3], heads)
        inters.append(inter)
        masks.append(m)
    class_logits = _lin(w, "mask2former.class_predictor", inters[-1].transpose(0, 1))
    return class_logits, masks[-1], dict(all_masks=masks, all_inter=inters, attn_masks=used_masks)


# ----------------------------------------------------------------------------------------
# panoptic post-process (image_processing_video_mask2former.py:1238-1481)
# ----------------------------------------------------------------------------------------
def panoptic_postprocess(class_logits: T, mask_logits: T, target_hw, threshold=0.5, mask_threshold=0.5,
                         overlap=0.8, fuse=(0, 1)):
    """Returns list (per batch item) of dict(segmentation, segments_info, query_class_logits, query_scores).
    Reproduces the hard-coded (256,256) intermediate size (:1298), the empty branch returning a float
    map of -1 (:1351-1375) and the 'height,width' rebinding quirk of the no-accepted-query branch (:1468-1472)."""
    Bn, Q, Tn, height, width = mask_logits.shape
    num_labels = class_logits.shape[-1] - 1
    ml = mask_logits.permute(0, 2, 1, 3, 4).reshape(Bn * Tn, Q, height, width)
    ml = F.interpolate(ml, size=(256, 256), mode="bilinear", align_corners=False).view(Bn, Tn, Q, 256, 256)
    mask_probs = ml.sigmoid()
    class_probs = class_logits.softmax(-1)
    scores, labels = class_probs.max(-1)
    results = []
    for i in range(Bn):
        keep = labels[i].ne(num_labels) & (scores[i] > threshold)
        mp, sc, lb, cp = mask_probs[i][:, keep], scores[i][keep], labels[i][keep], class_probs[i][keep]
        if int(keep.sum()) == 0:
            height, width = target_hw
            seg = torch.zeros((Tn, height, width)) - 1
            qcl = torch.zeros((Tn, 1, num_labels + 1, height, width))
            qcl[:, 0, -1] = 1
            results.append(dict(segmentation=seg, segments_info=[], query_class_logits=qcl, query_scores=[0.0]))
            continue
        seg = torch.zeros((Tn, target_hw[0], target_hw[1]), dtype=torch.int32)
        mp = F.interpolate(mp, size=tuple(target_hw), mode="bilinear", align_corners=False)
        weighted = mp * sc[None, :, None, None]
        lab_map = weighted.argmax(1)
        cur, stuff_mem, segs, kept, kept_scores = 0, {}, [], [], []
        for k in range(lb.shape[0]):
            cls = int(lb[k])
            should_fuse = cls in fuse
            mk = lab_map == k
            area = int(mk.sum())
            orig = int((weighted[:, k] >= mask_threshold).sum())
            exists = area > 0 and orig > 0
            if exists:
                ratio = (mk.sum() / (weighted[:, k] >= mask_threshold).sum()).item()
                if not ratio > overlap:
                    exists = False
            if exists:
                if cls in stuff_mem:
                    sid_f = stuff_mem[cls]
                else:
                    cur += 1
                    sid_f = cur
                sid = cur if not should_fuse else sid_f
                seg[mk] = sid
                s_ = round(sc[k].item(), 6)
                segs.append(dict(id=sid, label_id=cls, was_fused=should_fuse, score=s_))
                kept.append(k)
                kept_scores.append(s_)
                if should_fuse and cls not in stuff_mem:
                    stuff_mem[cls] = cur
        qcl = (cp[None, :, :, None, None] * mp[:, :, None])[:, kept]
        if qcl.shape[1] <= 0:
            qcl = torch.zeros((Tn, 1, num_labels + 1, height, width))
            qcl[:, 0, -1] = 1
        results.append(dict(segmentation=seg, segments_info=segs, query_class_logits=qcl, query_scores=kept_scores))
    return results


def scatter_labels(results, B: int, V: int, H: int, Wd: int):
    """SIU3RModel.post_process_gaussians label scatter (model.py:267-294): semantic = label_id+1, instance = id."""
    sem = torch.zeros(B, V, H, Wd, dtype=torch.int32)
    ins = torch.zeros(B, V, H, Wd, dtype=torch.int32)
    for b, r in enumerate(results):
        for seg in r["segments_info"]:
            m = r["segmentation"] == seg["id"]
            sem[b][m] = seg["label_id"] + 1
            ins[b][m] = seg["id"]
    return sem.reshape(B, -1), ins.reshape(B, -1)


def lift_ids(render_qc_logit: T, q_score, num_queries: int = 100, fuse=(0, 1), thr: float = 0.3):
    """Query-class-logit lifting of one batch item (pipeline.py:137-193): render_qc_logit [v, q, c+1, h, w]."""
    v, q, c, h, w_ = render_qc_logit.shape
    c_logit, q_index = render_qc_logit.max(dim=1)
    c_logit = torch.cat([c_logit[:, -1:], c_logit[:, :-1]], dim=1)
    q_index = torch.cat([q_index[:, -1:], q_index[:, :-1]], dim=1)
    sem_logits, sem_id = c_logit.max(dim=1)
    q_index = q_index.gather(1, sem_id[:, None]).squeeze(1) + 1
    sem_id = sem_id.clone()
    sem_id[sem_logits < thr] = 0
    q_index[sem_id == 0] = 0
    info = []
    for q_idx, sc in enumerate(q_score):
        ids = sem_id[q_index == q_idx + 1]
        if ids.numel() == 0:
            continue
        info.append({"id": q_idx + 1, "label_id": int(ids[0]), "was_fused": False, "score": sc})
    for stuff in fuse:
        m = sem_id == (stuff + 1)
        q_index[m] = num_queries + stuff + 1
        for i in info:
            if i["label_id"] == stuff + 1:
                i["was_fused"] = True
                i["id"] = int(q_index[m][0])
    return sem_id, q_index, info


# ----------------------------------------------------------------------------------------
# whole model (model.py:314-389)
# ----------------------------------------------------------------------------------------
def model_forward(w: W, images: T, intrinsics: T, keep_intermediates: bool = True) -> Dict[str, object]:
    B, V, _, H, Wd = images.shape
    bb = backbone(w, images, intrinsics)
    ms1 = adapter(w, images[:, 0], bb["all_feat1"])
    ms2 = adapter(w, images[:, 1], bb["all_feat2"])
    pts1 = pts3d_head(w, "downstream_head1", bb["dec1"], H, Wd)
    pts2 = pts3d_head(w, "downstream_head2", bb["dec2"], H, Wd)
    gs1 = gs_head(w, "gaussian_param_head1", bb["dec1"], images[:, 0], H, Wd)
    gs2 = gs_head(w, "gaussian_param_head2", bb["dec2"], images[:, 1], H, Wd)
    means = torch.stack((pts1.reshape(B, H * Wd, 3), pts2.reshape(B, H * Wd, 3)), dim=1)
    raw = torch.stack((gs1, gs2), dim=1)
    g = gaussian_adapter(means, raw)
    feats = [torch.stack([a, b], dim=1).flatten(0, 1) for a, b in zip(ms1, ms2)]  # fold (B,T) (:2090-2092)
    mask_features, ms = pixel_decoder(w, feats)
    class_logits, mask_logits, extra = m2f_decoder(w, ms, mask_features, B, V)
    results = panoptic_postprocess(class_logits, mask_logits, (H, Wd))
    sem, ins = scatter_labels(results, B, V, H, Wd)
    out = dict(
        means=g["means"].flatten(1, 2), covariances=g["covariances"].flatten(1, 2),
        harmonics=g["harmonics"].flatten(1, 2), opacities=g["opacities"].flatten(1, 2),
        scales=g["scales"].flatten(1, 2), rotations=g["rotations"].flatten(1, 2),
        semantic_labels=sem, instance_labels=ins,
        class_queries_logits=class_logits, masks_queries_logits=mask_logits,
        seg_masks=[r["segmentation"] for r in results], seg_infos=[r["segments_info"] for r in results],
        query_class_logits=[r["query_class_logits"] for r in results],
        query_scores=[r["query_scores"] for r in results],
        attn_masks=extra["attn_masks"],
    )
    if keep_intermediates:
        out.update(bb=bb, ms1=ms1, ms2=ms2, pts1=pts1, pts2=pts2, gs_raw1=gs1, gs_raw2=gs2,
                   mask_features=mask_features, ms=ms, m2f_extra=extra)
    return out


def model_forward_multi(w: W, images: T, intrinsics: T, keep_intermediates: bool = True) -> Dict[str, object]:
    """SIU3RMultiViewModel.forward (model_multi.py:314-389): view 0 -> *_head1, every other view -> *_head2;
    the adapter runs per view, Mask2Former over T = V frames."""
    B, V, _, H, Wd = images.shape
    bb = backbone_multi(w, images, intrinsics)
    ms_v = [adapter(w, images[:, v], bb["all_feats"][v]) for v in range(V)]
    pts = [pts3d_head(w, "downstream_head1" if v == 0 else "downstream_head2", bb["decs"][v], H, Wd) for v in range(V)]
    gs = [gs_head(w, "gaussian_param_head1" if v == 0 else "gaussian_param_head2", bb["decs"][v], images[:, v], H, Wd) for v in range(V)]
    means = torch.stack([p.reshape(B, H * Wd, 3) for p in pts], dim=1)
    raw = torch.stack(gs, dim=1)
    g = gaussian_adapter(means, raw)
    feats = [torch.stack([ms_v[v][l] for v in range(V)], dim=1).flatten(0, 1) for l in range(len(ms_v[0]))]
    mask_features, ms = pixel_decoder(w, feats)
    class_logits, mask_logits, extra = m2f_decoder(w, ms, mask_features, B, V)
    results = panoptic_postprocess(class_logits, mask_logits, (H, Wd))
    sem, ins = scatter_labels(results, B, V, H, Wd)
    out = dict(
        means=g["means"].flatten(1, 2), covariances=g["covariances"].flatten(1, 2),
        harmonics=g["harmonics"].flatten(1, 2), opacities=g["opacities"].flatten(1, 2),
        scales=g["scales"].flatten(1, 2), rotations=g["rotations"].flatten(1, 2),
        semantic_labels=sem, instance_labels=ins,
        class_queries_logits=class_logits, masks_queries_logits=mask_logits,
        seg_masks=[r["segmentation"] for r in results], seg_infos=[r["segments_info"] for r in results],
        query_class_logits=[r["query_class_logits"] for r in results],
        query_scores=[r["query_scores"] for r in results],
        attn_masks=extra["attn_masks"],
    )
    if keep_intermediates:
        out.update(bb=bb, ms_v=ms_v, pts=pts, gs_raw=gs, mask_features=mask_features, ms=ms)
    return out
