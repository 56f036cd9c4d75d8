"""Build-owned deterministic weight generator for SIU3R-shaped models.

Data synthesis shared by the oracle (oracle/weights.py re-exports it), the golden-vector generator, bench.py and the
CLI's plumbing mode: it contains no part of the algorithm.  No checkpoint is available offline
(SURVEY.md section 8(c)), so both sides of every parity check regenerate the same
655.5 M parameters from this counter-based generator: pure uint32 integer hashing
followed by exactly-representable float conversions, hence bit-identical on any host.

Parameter names and shapes follow the reference state-dict schema (SURVEY.md Appendix E;
reference: src/models/model.py:31-114 and the sub-module constructors cited per block
below).  ``tests/test_oracle_pins.py`` checks this spec against the reference's own
``state_dict()`` when /root/reference is present.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

ENC_DEPTH, ENC_DIM, ENC_HEADS = 24, 1024, 16
DEC_DEPTH, DEC_DIM, DEC_HEADS = 12, 768, 12
NUM_QUERIES, NUM_CLASSES_P1 = 100, 21
M2F_DIM, M2F_HEADS, M2F_ENC_FFN, M2F_DEC_FFN = 256, 8, 1024, 2048
M2F_ENC_LAYERS, M2F_DEC_LAYERS = 6, 9
RAW_GS_DIM = 83  # 1 opacity + 3 scale + 4 quat + 3*25 SH   (model.py:93)


def _lin(spec, name, out_f, in_f, bias=True):
    spec[name + ".weight"] = (out_f, in_f)
    if bias:
        spec[name + ".bias"] = (out_f,)


def _norm(spec, name, dim):
    spec[name + ".weight"] = (dim,)
    spec[name + ".bias"] = (dim,)


def _bn(spec, name, dim):
    _norm(spec, name, dim)
    spec[name + ".running_mean"] = (dim,)
    spec[name + ".running_var"] = (dim,)
    spec[name + ".num_batches_tracked"] = ()


def _conv(spec, name, out_c, in_c, k, bias=True):
    spec[name + ".weight"] = (out_c, in_c, k, k)
    if bias:
        spec[name + ".bias"] = (out_c,)


def _dpt(spec, p, gs: bool):
    """DPT head parameters (heads/dpt_block.py:289-532, dpt_head.py:123-148,
    dpt_gs_head.py:99-119,215-240)."""
    dims_in = [ENC_DIM, DEC_DIM, DEC_DIM, DEC_DIM]
    layer_dims = [96, 192, 384, 768]
    for i, ld in enumerate(layer_dims):
        # layerN_rn and layer_rn.N are the same tensor under two keys (Appendix E)
        spec[f"{p}.dpt.scratch.layer{i+1}_rn.weight"] = (256, ld, 3, 3)
    for i, ld in enumerate(layer_dims):
        spec[f"{p}.dpt.scratch.layer_rn.{i}.weight"] = (256, ld, 3, 3)
    for r in (1, 2, 3, 4):
        q = f"{p}.dpt.scratch.refinenet{r}"
        _conv(spec, q + ".out_conv", 256, 256, 1)
        for u in (1, 2):
            _conv(spec, f"{q}.resConfUnit{u}.conv1", 256, 256, 3)
            _conv(spec, f"{q}.resConfUnit{u}.conv2", 256, 256, 3)
    if gs:
        _conv(spec, f"{p}.dpt.head.0", 256, 256, 3, bias=False)
        _conv(spec, f"{p}.dpt.head.4", RAW_GS_DIM, 256, 1)
    else:
        _conv(spec, f"{p}.dpt.head.0", 128, 256, 3)
        _conv(spec, f"{p}.dpt.head.2", 128, 128, 3)
        _conv(spec, f"{p}.dpt.head.4", 3, 128, 1)
    a = f"{p}.dpt.act_postprocess"
    _conv(spec, a + ".0.0", 96, dims_in[0], 1)
    spec[a + ".0.1.weight"] = (96, 96, 4, 4)  # ConvTranspose2d (in,out,k,k)
    spec[a + ".0.1.bias"] = (96,)
    _conv(spec, a + ".1.0", 192, dims_in[1], 1)
    spec[a + ".1.1.weight"] = (192, 192, 2, 2)
    spec[a + ".1.1.bias"] = (192,)
    _conv(spec, a + ".2.0", 384, dims_in[2], 1)
    _conv(spec, a + ".3.0", 768, dims_in[3], 1)
    _conv(spec, a + ".3.1", 768, 768, 3)
    if gs:
        _conv(spec, f"{p}.dpt.input_merger.0", 256, 3, 7)


def _extractor(spec, p):
    """vit_adapter.py:62-121 (Extractor) + blocks.py:87-146 (MSDeformAttn, 16 heads, 1 level, 4 pts)."""
    for n in ("query_norm", "feat_norm"):
        _norm(spec, f"{p}.{n}", ENC_DIM)
    _lin(spec, f"{p}.attn.sampling_offsets", 16 * 1 * 4 * 2, ENC_DIM)
    _lin(spec, f"{p}.attn.attention_weights", 16 * 1 * 4, ENC_DIM)
    _lin(spec, f"{p}.attn.value_proj", ENC_DIM, ENC_DIM)
    _lin(spec, f"{p}.attn.output_proj", ENC_DIM, ENC_DIM)
    _lin(spec, f"{p}.ffn.fc1", 256, ENC_DIM)
    spec[f"{p}.ffn.dwconv.dwconv.weight"] = (256, 1, 3, 3)
    spec[f"{p}.ffn.dwconv.dwconv.bias"] = (256,)
    _lin(spec, f"{p}.ffn.fc2", ENC_DIM, 256)
    _norm(spec, f"{p}.ffn_norm", ENC_DIM)


def param_spec() -> "OrderedDict[str, tuple]":
    """name -> shape, in the reference's state_dict order."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    # --- backbone (backbone_croco.py:30-113; croco/croco.py:23-114; croco/blocks.py:58-191)
    _conv(s, "backbone.patch_embed.proj", ENC_DIM, 3, 16)
    for i in range(ENC_DEPTH):
        p = f"backbone.enc_blocks.{i}"
        _norm(s, p + ".norm1", ENC_DIM)
        _lin(s, p + ".attn.qkv", 3 * ENC_DIM, ENC_DIM)
        _lin(s, p + ".attn.proj", ENC_DIM, ENC_DIM)
        _norm(s, p + ".norm2", ENC_DIM)
        _lin(s, p + ".mlp.fc1", 4 * ENC_DIM, ENC_DIM)
        _lin(s, p + ".mlp.fc2", ENC_DIM, 4 * ENC_DIM)
    _norm(s, "backbone.enc_norm", ENC_DIM)
    _lin(s, "backbone.decoder_embed", DEC_DIM, ENC_DIM)

    def dec(prefix):
        for i in range(DEC_DEPTH):
            p = f"{prefix}.{i}"
            _norm(s, p + ".norm1", DEC_DIM)
            _lin(s, p + ".attn.qkv", 3 * DEC_DIM, DEC_DIM)
            _lin(s, p + ".attn.proj", DEC_DIM, DEC_DIM)
            for n in ("projq", "projk", "projv", "proj"):
                _lin(s, f"{p}.cross_attn.{n}", DEC_DIM, DEC_DIM)
            _norm(s, p + ".norm2", DEC_DIM)
            _norm(s, p + ".norm3", DEC_DIM)
            _lin(s, p + ".mlp.fc1", 4 * DEC_DIM, DEC_DIM)
            _lin(s, p + ".mlp.fc2", DEC_DIM, 4 * DEC_DIM)
            _norm(s, p + ".norm_y", DEC_DIM)

    dec("backbone.dec_blocks")
    _norm(s, "backbone.dec_norm", DEC_DIM)
    dec("backbone.dec_blocks2")
    _lin(s, "backbone.intrinsic_encoder", ENC_DIM, 9)
    # --- adapter (vit_adapter/vit_adapter.py:200-392)
    s["adapter.level_embed"] = (3, ENC_DIM)
    _conv(s, "adapter.spm.stem.0", 64, 3, 3, bias=False)
    _bn(s, "adapter.spm.stem.1", 64)
    _conv(s, "adapter.spm.stem.3", 64, 64, 3, bias=False)
    _bn(s, "adapter.spm.stem.4", 64)
    _conv(s, "adapter.spm.stem.6", 64, 64, 3, bias=False)
    _bn(s, "adapter.spm.stem.7", 64)
    _conv(s, "adapter.spm.conv2.0", 128, 64, 3, bias=False)
    _bn(s, "adapter.spm.conv2.1", 128)
    _conv(s, "adapter.spm.conv3.0", 256, 128, 3, bias=False)
    _bn(s, "adapter.spm.conv3.1", 256)
    _conv(s, "adapter.spm.conv4.0", 256, 256, 3, bias=False)
    _bn(s, "adapter.spm.conv4.1", 256)
    for i, c in enumerate((64, 128, 256, 256)):
        _conv(s, f"adapter.spm.fc{i+1}", ENC_DIM, c, 1)
    for i in range(4):
        _extractor(s, f"adapter.interactions.{i}.extractor")
        if i == 3:
            for j in range(2):
                _extractor(s, f"adapter.interactions.3.extra_extractors.{j}")
    s["adapter.up.weight"] = (ENC_DIM, ENC_DIM, 2, 2)  # ConvTranspose2d
    s["adapter.up.bias"] = (ENC_DIM,)
    for i in range(1, 5):
        _bn(s, f"adapter.norm{i}", ENC_DIM)
    # --- mask2former (mask2former/video_seg_decoder.py:1973-2060, 1483-1504, 1178-1200, 2257-2299)
    pd = "mask2former.model.pixel_decoder"
    s[pd + ".level_embed"] = (3, M2F_DIM)
    for i in range(3):
        _conv(s, f"{pd}.input_projections.{i}.0", M2F_DIM, ENC_DIM, 1)
        _norm(s, f"{pd}.input_projections.{i}.1", M2F_DIM)
    for i in range(M2F_ENC_LAYERS):
        p = f"{pd}.encoder.layers.{i}"
        _lin(s, p + ".self_attn.sampling_offsets", M2F_HEADS * 3 * 4 * 2, M2F_DIM)
        _lin(s, p + ".self_attn.attention_weights", M2F_HEADS * 3 * 4, M2F_DIM)
        _lin(s, p + ".self_attn.value_proj", M2F_DIM, M2F_DIM)
        _lin(s, p + ".self_attn.output_proj", M2F_DIM, M2F_DIM)
        _norm(s, p + ".self_attn_layer_norm", M2F_DIM)
        _lin(s, p + ".fc1", M2F_ENC_FFN, M2F_DIM)
        _lin(s, p + ".fc2", M2F_DIM, M2F_ENC_FFN)
        _norm(s, p + ".final_layer_norm", M2F_DIM)
    _conv(s, pd + ".mask_projection", M2F_DIM, M2F_DIM, 1)
    _conv(s, pd + ".adapter_1.0", M2F_DIM, ENC_DIM, 1, bias=False)
    _norm(s, pd + ".adapter_1.1", M2F_DIM)
    _conv(s, pd + ".layer_1.0", M2F_DIM, M2F_DIM, 3, bias=False)
    _norm(s, pd + ".layer_1.1", M2F_DIM)
    tm = "mask2former.model.transformer_module"
    s[tm + ".queries_embedder.weight"] = (NUM_QUERIES, M2F_DIM)
    s[tm + ".queries_features.weight"] = (NUM_QUERIES, M2F_DIM)
    for i in range(M2F_DEC_LAYERS):
        p = f"{tm}.decoder.layers.{i}"
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            _lin(s, f"{p}.self_attn.{n}", M2F_DIM, M2F_DIM)
        _norm(s, p + ".self_attn_layer_norm", M2F_DIM)
        s[p + ".cross_attn.in_proj_weight"] = (3 * M2F_DIM, M2F_DIM)
        s[p + ".cross_attn.in_proj_bias"] = (3 * M2F_DIM,)
        _lin(s, p + ".cross_attn.out_proj", M2F_DIM, M2F_DIM)
        _norm(s, p + ".cross_attn_layer_norm", M2F_DIM)
        _lin(s, p + ".fc1", M2F_DEC_FFN, M2F_DIM)
        _lin(s, p + ".fc2", M2F_DIM, M2F_DEC_FFN)
        _norm(s, p + ".final_layer_norm", M2F_DIM)
    _norm(s, tm + ".decoder.layernorm", M2F_DIM)
    for i in range(3):
        _lin(s, f"{tm}.decoder.mask_predictor.mask_embedder.{i}.0", M2F_DIM, M2F_DIM)
    s[tm + ".level_embed.weight"] = (3, M2F_DIM)
    _lin(s, "mask2former.class_predictor", NUM_CLASSES_P1, M2F_DIM)
    s["mask2former.criterion.empty_weight"] = (NUM_CLASSES_P1,)
    # --- heads
    _dpt(s, "downstream_head1", gs=False)
    _dpt(s, "downstream_head2", gs=False)
    _dpt(s, "gaussian_param_head1", gs=True)
    _dpt(s, "gaussian_param_head2", gs=True)
    return s


def _hash_uniform(key: int, n: int) -> np.ndarray:
    """n floats in [-1, 1): murmur-style uint32 finaliser over the element counter."""
    out = np.empty(n, dtype=np.float32)
    CH = 1 << 22
    k = np.uint32(key & 0xFFFFFFFF)
    for s0 in range(0, n, CH):
        m = min(CH, n - s0)
        x = np.arange(s0, s0 + m, dtype=np.uint32)
        with np.errstate(over="ignore"):
            x ^= k
            x *= np.uint32(0x9E3779B1)
            x ^= x >> np.uint32(15)
            x *= np.uint32(0x85EBCA77)
            x ^= x >> np.uint32(13)
            x *= np.uint32(0xC2B2AE3D)
            x ^= x >> np.uint32(16)
        # 24 mantissa bits -> exact in fp32
        out[s0 : s0 + m] = (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 / (1 << 24)) - np.float32(1.0)
    return out


# ---- panoptic shaping ---------------------------------------------------------------------------------------------------
# With plainly random weights the 100 object queries of the masked-attention decoder collapse onto one vector after a few layers
# (every sub-layer output swamps the query's own features), every query then predicts the same class with a near-uniform
# distribution, nothing passes `score > 0.5` and the panoptic post-process only ever runs its empty branch -- no segment ids, no
# query x class logit maps to lift, an all-zero integer output to "compare".  These few seeded rules keep the generator a pure
# function of (name, shape, seed) but give the segmentation path something to do, like a trained checkpoint would:
#   * queries_features: unit-scale per-query features (the residual stream the decoder starts from);
#   * the three write-back projections of each decoder layer (cross_attn.out_proj, self_attn.out_proj, fc2) are damped, so a
#     query keeps its identity through the nine post-norm layers;
#   * class_predictor: a sharp classifier (weights x CLASS_GAIN) with a void prior (bias[20] += VOID_BIAS) that lets roughly one
#     query in ten beat "no object" with probability > 0.5;
#   * constant offsets on the mask embedding and on the per-pixel mask features, whose product shifts all mask logits negative:
#     masks become small and mostly disjoint, so that several kept queries pass the `area / original_area > 0.8` test.
PANOPTIC_DAMP = 0.15
CLASS_GAIN = 12.0
VOID_BIAS = 34.0
STUFF_BIAS = 10.0           # a prior for the two stuff classes (wall, floor): exercises the `label_ids_to_fuse` id sharing
MASK_EMBED_BIAS = 0.105     # mask logit = <mask_embed(query), mask_feature(pixel)>: the two constant offsets add
MASK_FEATURE_BIAS = 0.105   # -256 * 0.105 * 0.105 = -2.8 to every logit, so a query is "on" for a small, mostly exclusive set of pixels
_TM = "mask2former.model.transformer_module"


def make_tensor(name: str, shape: tuple, seed: int = 0) -> torch.Tensor:
    if name.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64)
    n = int(np.prod(shape)) if len(shape) else 1
    key = zlib.crc32(name.encode()) ^ (seed * 0x9E3779B9 & 0xFFFFFFFF)
    u = _hash_uniform(key, n)
    if name == _TM + ".queries_features.weight":
        return torch.from_numpy(u.reshape(shape))
    if name == "mask2former.class_predictor.bias":
        v = np.float32(0.1) * u
        v[-1] += np.float32(VOID_BIAS)
        v[:2] += np.float32(STUFF_BIAS)
        return torch.from_numpy(v.reshape(shape))
    if name == _TM + ".decoder.mask_predictor.mask_embedder.2.0.bias":
        return torch.from_numpy((np.float32(0.1) * u + np.float32(MASK_EMBED_BIAS)).reshape(shape))
    if name == "mask2former.model.pixel_decoder.mask_projection.bias":
        return torch.from_numpy((np.float32(0.1) * u - np.float32(MASK_FEATURE_BIAS)).reshape(shape))
    if name.endswith("empty_weight"):
        v = np.ones(n, dtype=np.float32)
    elif name.endswith("running_var"):
        v = np.float32(1.0) + np.float32(0.3) * u
    elif name.endswith("running_mean"):
        v = np.float32(0.1) * u
    elif name.endswith("sampling_offsets.bias"):
        v = np.float32(2.0) * u
    elif name.endswith("attention_weights.bias"):
        v = np.float32(0.5) * u
    elif len(shape) >= 2:
        fan_in = n // shape[0]
        a = np.float32(0.8 * np.sqrt(3.0 / fan_in))
        if name.startswith(_TM + ".decoder.layers.") and name.endswith(("cross_attn.out_proj.weight", "self_attn.out_proj.weight", "fc2.weight")):
            a = a * np.float32(PANOPTIC_DAMP)
        if name == "mask2former.class_predictor.weight":
            a = a * np.float32(CLASS_GAIN)
        v = a * u
    elif name.endswith(".weight"):  # 1-d: norm scales
        v = np.float32(1.0) + np.float32(0.1) * u
    else:  # biases
        v = np.float32(0.1) * u
    return torch.from_numpy(v.reshape(shape) if len(shape) else v.reshape(()))


def make_weights(seed: int = 0, only_prefix: str | None = None) -> "OrderedDict[str, torch.Tensor]":
    """Full synthetic state dict (fp32).  ``layerN_rn`` and ``layer_rn.N`` alias the same storage."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_spec().items():
        if only_prefix is not None and not name.startswith(only_prefix):
            continue
        if ".scratch.layer_rn." in name:
            idx = int(name.split(".scratch.layer_rn.")[1].split(".")[0])
            alias = name.split(".scratch.layer_rn.")[0] + f".scratch.layer{idx+1}_rn.weight"
            sd[name] = sd[alias]
            continue
        sd[name] = make_tensor(name, shape, seed)
    return sd
