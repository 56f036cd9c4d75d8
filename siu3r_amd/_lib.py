"""ctypes binding of libsiu3r_hip.so (the C ABI declared in include/siu3r_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol is absent the
import raises, and every op raises RuntimeError when handed a non-GPU tensor.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIU3R_LIB_OVERRIDE") or os.path.join(_HERE, "libsiu3r_hip.so")  # override: kernel A/B builds (tools/)

BF16, F32 = 0, 1


class GemmParams(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("c", C.c_void_p),
        ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("kpad", C.c_int32),
        ("lda", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64),
        ("a_dtype", C.c_int32), ("c_dtype", C.c_int32), ("r_dtype", C.c_int32),
        ("act", C.c_int32), ("relu_in", C.c_int32),
        ("batch", C.c_int32),
        ("sa", C.c_int64), ("sw", C.c_int64), ("sc", C.c_int64), ("sr", C.c_int64),
        ("a_mode", C.c_int32),
        ("ih", C.c_int32), ("iw", C.c_int32), ("cin", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("oh", C.c_int32), ("ow", C.c_int32),
        ("out_mode", C.c_int32), ("up", C.c_int32), ("cout", C.c_int32),
        ("up_src", C.c_void_p), ("up_dtype", C.c_int32),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("rope_pos", C.c_void_p), ("rope_ncols", C.c_int32),
        ("map_gx", C.c_int32), ("map_rm", C.c_int32), ("map_rn", C.c_int32),
        ("trace", C.c_void_p), ("w_x3", C.c_void_p),
        ("bmod", C.c_int32), ("sa_i", C.c_int64), ("sc_i", C.c_int64), ("sr_i", C.c_int64), ("sbias", C.c_int64),
        ("ln_stats", C.c_void_p), ("ln_c1", C.c_void_p), ("ln_c2", C.c_void_p), ("ln_tiles", C.c_int32), ("ln_eps", C.c_float),
        ("ln_ldm", C.c_int64), ("ln_sz", C.c_int64), ("ln_sz_i", C.c_int64),
        ("stats_out", C.c_void_p), ("st_ldm", C.c_int64), ("st_sz", C.c_int64), ("st_sz_i", C.c_int64),
        ("c_aux", C.c_void_p),
        ("splitk", C.c_int32), ("sk_ws", C.c_void_p), ("sk_cnt", C.c_void_p),
        ("sk_ws_floats", C.c_int64), ("sk_cnt_n", C.c_int32), ("tile_cfg", C.c_int32), ("m_main", C.c_int32), ("sk_gx", C.c_int32),
        ("c_x3", C.c_void_p), ("a_x3", C.c_int32), ("c_x3_col0", C.c_int32),
    ]


class GemmPlan(C.Structure):
    """siu3r_gemm_plan_t: what siu3r_gemm launches for a parameter block."""
    _fields_ = [
        ("tile_cfg", C.c_int32), ("bm", C.c_int32), ("bn", C.c_int32), ("splitk", C.c_int32), ("skinny_rows", C.c_int32),
        ("counters", C.c_int32), ("ws_floats", C.c_int64), ("kernel", C.c_char * 160), ("a_x3_ok", C.c_int32), ("c_x3_ok", C.c_int32),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p),
        ("dtype", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("D", C.c_int32),
        ("q_sb", C.c_int64), ("q_sn", C.c_int64), ("q_sh", C.c_int64),
        ("k_sb", C.c_int64), ("k_sn", C.c_int64), ("k_sh", C.c_int64),
        ("v_sb", C.c_int64), ("v_sn", C.c_int64), ("v_sh", C.c_int64),
        ("scale", C.c_float),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("rope_max_pos", C.c_int32),
        ("qpos", C.c_void_p), ("kpos", C.c_void_p), ("mask", C.c_void_p),
        ("split3", C.c_int32), ("mask_ld", C.c_int64),
        ("ws", C.c_void_p), ("splits", C.c_int32), ("kv_bxor", C.c_int32), ("kv_x3", C.c_int32),
    ]


class RasterCam(C.Structure):
    """siu3r_raster_cam (include/siu3r_hip.h); the CPU oracle's raster_cam has the same layout."""
    _fields_ = [
        ("mode", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("w2c", C.c_float * 16), ("proj", C.c_float * 16),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("campos", C.c_float * 3), ("bg", C.c_float * 3),
        ("sh_degree", C.c_int32), ("sh_band4", C.c_int32), ("k2_znear_cull", C.c_float),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("near_plane", C.c_float), ("far_plane", C.c_float), ("eps2d", C.c_float), ("radius_clip", C.c_float),
        ("extent_sigma", C.c_float), ("opacity_aware_extent", C.c_int32),
        ("alpha_min", C.c_float), ("alpha_max", C.c_float), ("t_min", C.c_float), ("dilation", C.c_float),
        ("nt_post_blend", C.c_int32), ("k2_near", C.c_float), ("k2_far", C.c_float),
    ]


# name -> argtypes; restype is c_int unless listed in _RESTYPES.  Mirrors include/siu3r_hip.h.
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
ABI_VERSION = 10  # SIU3R_ABI_VERSION of include/siu3r_hip.h these ctypes declarations mirror

SIGNATURES = {
    "siu3r_last_error": [],
    "siu3r_abi_version": [],
    "siu3r_rope2d": [_P, _I, _I, _I, _I, _I, _L, _L, _L, _P, _F, _F, _P],
    "siu3r_gemm": [C.POINTER(GemmParams), _P],
    "siu3r_gemm_plan": [C.POINTER(GemmParams), C.POINTER(GemmPlan)],
    "siu3r_gemm_tune": [_I, _I],
    "siu3r_layernorm": [_P, _P, _I, _P, _P, _L, _I, _L, _L, _F, _P],
    "siu3r_layernorm2": [_P, _P, _I, _P, _P, _P, _L, _I, _L, _L, _L, _F, _P],
    "siu3r_attention": [C.POINTER(AttnParams), _P],
    "siu3r_attention_kv_x3_ok": [C.POINTER(AttnParams)],
    "siu3r_add": [_P, _P, _P, _L, _L, _I, _P],
    "siu3r_pack_image_nhwc": [_P, _P, _I, _I, _I, _I, _I, _P],
    "siu3r_resize_bilinear": [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "siu3r_affine_add": [_P, _I, _P, _I, _P, _I, _P, _P, _L, _I, _P],
    "siu3r_resize_bilinear_strided": [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _L, _P],
    "siu3r_affine_add_strided": [_P, _I, _P, _I, _P, _I, _P, _P, _L, _I, _L, _L, _L, _P],
    "siu3r_maxpool3x3s2": [_P, _P, _I, _I, _I, _I, _I, _P],
    "siu3r_maxpool2x2s2": [_P, _P, _I, _I, _I, _I, _I, _P],
    "siu3r_lpips_layer": [_P, _P, _P, _P, _L, _I, _F, _P],
    "siu3r_stem7x7_x3": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "siu3r_proj_rows_x3": [_P, _P, _P, _P, _I, _I, _L, _I, _I, _I, _L, _P],
    "siu3r_dwconv3x3_gelu": [_P, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "siu3r_msdeform_sample": [_P, _I, _P, _P, C.POINTER(C.c_int32), _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "siu3r_groupnorm": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "siu3r_pts3d_exp": [_P, _L, _P],
    "siu3r_gaussian_adapter": [_P, _I, _P, _P, _P, _P, _P, _L, _P],
    "siu3r_m2f_attn_mask": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "siu3r_split_bf16": [_P, _P, _P, _P, _L, _I, _I, _L, _P],
    "siu3r_raster_geometry": [_I, _I, _L, C.POINTER(C.c_int32)],
    "siu3r_raster_project": [C.POINTER(RasterCam), _I, _P, _L, _P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "siu3r_raster_sort": [_I, _L, _P, _P, _P, _P, _P, _P, _P, _P],
    "siu3r_raster_bin": [C.POINTER(RasterCam), _I, _L, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "siu3r_raster_composite_rgb": [C.POINTER(RasterCam), _I, _P, _L, _P, _P, _L, _P, _P, _P, _P, _P, _P],
    "siu3r_raster_tile_lists": [C.POINTER(RasterCam), _I, _P, _P, _L, _P, _P, _P, _L, _P, _P],
    "siu3r_raster_composite_feat": [C.POINTER(RasterCam), _I, _P, _L, _P, _P, _L, _P, _P, _I, _P, _P, _P],
    "siu3r_raster_composite_feat_ws_bytes": [_I, _I, _I, _L],
    "siu3r_raster_tune": [_I, _I],
    "siu3r_raster_composite_feat_ws": [C.POINTER(RasterCam), _I, _P, _L, _P, _P, _L, _P, _P, _I, _P, _P, _P, _L, _P],
    "siu3r_scale_inplace": [_P, _L, _F, _P],
    "siu3r_quat_scale_to_cov6": [_P, _P, _P, _L, _P],
    "siu3r_sh_eval": [_P, _P, _P, _I, _I, _P, _L, _P],
    "siu3r_blend_background": [_P, _P, _P, _I, _L, _P],
    "siu3r_sh_eval_dp": [_P, _P, _P, _I, _I, _P, _L, _P],
    "siu3r_blend_background_dp": [_P, _P, _P, _I, _L, _P],
    "siu3r_raster_project_dp": [C.POINTER(RasterCam), _I, _P, _P, _P, _L, _P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "siu3r_raster_project_c2w": [C.POINTER(RasterCam), _I, _P, _P, _P, _F, _L, _P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "siu3r_lift_ids": [_P, _I, _I, _I, _I, _I, _F, _I, C.c_uint32, _P, _P, _P, _P, _P],
    "siu3r_panoptic_stage1": [_P] * 20 + [_I] * 9 + [_F, _F, _F, C.c_uint32, _P],
    "siu3r_panoptic_qcl": [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
}
_RESTYPES = {"siu3r_last_error": C.c_char_p, "siu3r_raster_composite_feat_ws_bytes": C.c_int64}

_lib = None


def lib():
    """Load the shared library once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime: import it first so that this library binds to the SAME runtime
    # (two runtimes in one process -> "no ROCm-capable device is detected" on the second one)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m siu3r_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the SIU3R hot path."
        )
    l = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(l, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    if l.siu3r_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {l.siu3r_abi_version()}, these bindings expect {ABI_VERSION}: rebuild it "
                           f"(python -m siu3r_amd.build)")
    _lib = l
    return l


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().siu3r_last_error()
        raise RuntimeError((msg.decode() if msg else f"{what} failed") + f" [rc={rc}]")
