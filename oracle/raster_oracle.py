"""ctypes front-end of oracle/libraster_ref.so (CPU oracle of the splat rasterizers; TEST INFRASTRUCTURE ONLY).
Build with `make -C oracle`.  The camera struct has the same layout as siu3r_raster_cam, so tests build one
parameter block and hand it to both sides."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libraster_ref.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        _LIB = C.CDLL(path)
        _LIB.raster_ref_forward.restype = C.c_int64
        _LIB.raster_ref_struct_size.restype = C.c_int
    return _LIB


def forward(cam, means, cov6, opacities, colors, want_lists=True):
    """cam: any ctypes struct with the raster_cam layout.  numpy fp32 inputs.  Returns a dict of numpy arrays."""
    l = lib()
    assert C.sizeof(cam) == l.raster_ref_struct_size(), (C.sizeof(cam), l.raster_ref_struct_size())
    means = np.ascontiguousarray(means, np.float32)
    cov6 = np.ascontiguousarray(cov6, np.float32)
    opacities = np.ascontiguousarray(opacities, np.float32)
    colors = np.ascontiguousarray(colors, np.float32)
    G = means.shape[0]
    H, W = cam.height, cam.width
    T = ((W + 15) // 16) * ((H + 15) // 16)
    channels = colors.shape[1] if colors.ndim >= 2 else 0
    radii = np.zeros((G, 2), np.int32)
    tt = np.zeros((G,), np.int32)
    nt = np.zeros((G,), np.int32)
    if cam.mode == 0:
        image = np.zeros((3, H, W), np.float32)
    else:
        image = np.zeros((H, W, channels), np.float32)
    depth = np.zeros((H, W), np.float32)
    alpha = np.zeros((H, W), np.float32)
    tile_start = np.zeros((T + 1,), np.int32)
    cap = max(1, 64 * G + 1024)
    ids = np.zeros((cap,), np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    D = l.raster_ref_forward(C.byref(cam), C.c_int64(G), p(means), p(cov6), p(opacities), p(colors), C.c_int32(channels), p(radii), p(tt),
                             p(nt), p(image), p(depth), p(alpha), p(tile_start), p(ids) if want_lists else None, C.c_int64(cap))
    assert D >= 0, "oracle id buffer too small"
    return dict(radii=radii, tiles_touched=tt, n_touched=nt, image=image, depth=depth, alpha=alpha, tile_start=tile_start, ids=ids[:D], D=int(D))


def quat_scale_to_cov6(quats, scales):
    quats, scales = np.ascontiguousarray(quats, np.float32), np.ascontiguousarray(scales, np.float32)
    out = np.zeros((quats.shape[0], 6), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().raster_ref_quat_scale_to_cov6(C.c_int64(quats.shape[0]), p(quats), p(scales), p(out))
    return out


def sh_eval(degree, means, campos, sh):
    means, campos, sh = (np.ascontiguousarray(a, np.float32) for a in (means, campos, sh))
    out = np.zeros((means.shape[0], 3), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().raster_ref_sh_eval(C.c_int64(means.shape[0]), C.c_int(degree), C.c_int(sh.shape[1]), p(means), p(campos), p(sh), p(out))
    return out


def blend_background(colors, alpha, bg):
    colors = np.ascontiguousarray(colors, np.float32).copy()
    alpha, bg = np.ascontiguousarray(alpha, np.float32), np.ascontiguousarray(bg, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().raster_ref_blend_background(C.c_int64(alpha.size), C.c_int(colors.shape[-1]), p(colors), p(alpha), p(bg))
    return colors
