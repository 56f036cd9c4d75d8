"""Gaussian splat rasterizer front-end: torch tensors -> the C ABI of csrc/raster.hip.

Two conventions behind one tile-binned HIP renderer (include/siu3r_hip.h, siu3r_raster_cam):
  K2 = diff-gaussian-rasterization-w-pose semantics (reference call site src/models/cuda_splatting.py:90-118)
  K3 = gsplat.rasterization semantics (reference call site src/models/gaussian_renderer.py:92-106)
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import RasterCam, check
from .ops import _gpu, _p, _stream

TILE = 16


def _set(arr, values):
    for i, v in enumerate(values):
        arr[i] = float(v)


def make_cam_k2(w2c, full_proj, tanfovx, tanfovy, campos, bg, width, height, sh_degree=4, sh_band4=False) -> RasterCam:
    c = RasterCam()
    c.mode, c.width, c.height = 0, int(width), int(height)
    _set(c.w2c, w2c.reshape(-1).tolist())
    _set(c.proj, full_proj.reshape(-1).tolist())
    c.tanfovx, c.tanfovy = float(tanfovx), float(tanfovy)
    _set(c.campos, campos)
    _set(c.bg, bg)
    c.sh_degree, c.sh_band4, c.k2_znear_cull = int(sh_degree), int(bool(sh_band4)), 0.2
    c.alpha_min, c.alpha_max, c.t_min, c.dilation = 1.0 / 255.0, 0.99, 1e-4, 0.3
    c.eps2d, c.extent_sigma = 0.3, 3.33
    return c


def make_cam_k3(w2c, fx, fy, cx, cy, width, height, near_plane=0.01, far_plane=1e10, eps2d=0.3, radius_clip=0.0,
                opacity_aware_extent=False) -> RasterCam:
    c = RasterCam()
    c.mode, c.width, c.height = 1, int(width), int(height)
    _set(c.w2c, w2c.reshape(-1).tolist())
    c.fx, c.fy, c.cx, c.cy = float(fx), float(fy), float(cx), float(cy)
    c.near_plane, c.far_plane, c.eps2d, c.radius_clip = float(near_plane), float(far_plane), float(eps2d), float(radius_clip)
    c.extent_sigma, c.opacity_aware_extent = 3.33, int(bool(opacity_aware_extent))
    c.alpha_min, c.alpha_max, c.t_min, c.dilation = 1.0 / 255.0, 0.999, 1e-4, 0.3
    c.k2_znear_cull = 0.2
    return c


def cov6_from_cov3x3(cov: torch.Tensor) -> torch.Tensor:
    """[G,3,3] -> [G,6] upper-triangular (xx,xy,xz,yy,yz,zz), the order of torch.triu_indices(3,3)
    (reference cuda_splatting.py:107,115).  Pure indexing (layout plumbing)."""
    return torch.stack((cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]), dim=-1).contiguous()


class _State(dict):
    """Per-view binning state.  "D" (the number of (tile, Gaussian) pairs) lives on the device until someone asks for it:
    reading it is the only host synchronisation of a rendered view, and it is deferred to the first access."""

    def __getitem__(self, k):
        if k == "D" and not dict.__contains__(self, "D"):
            d = int(dict.__getitem__(self, "D_dev").item())
            if d > dict.__getitem__(self, "cap"):
                raise RuntimeError(f"rasterizer pair buffers overflowed: D = {d} > capacity {dict.__getitem__(self, 'cap')} "
                                   f"(pass a larger pair_capacity)")
            dict.__setitem__(self, "D", d)
        return dict.__getitem__(self, k)


def default_pair_capacity(G: int) -> int:
    """Upper bound used to size the key / id buffers without reading the pair count back: 8 pairs per Gaussian (the 2 M-Gaussian
    1080p stress scene needs 6.7), at least 1 M.  96 B per Gaussian of HBM in the worst case."""
    return int(min(max(8 * G, 1 << 20), (1 << 31) - 1024))


def _bin_and_sort(cam: RasterCam, means, cov6, opac, colors, channels, pair_capacity=None):
    G = means.shape[0]
    dev = means.device
    gw, gh = (cam.width + TILE - 1) // TILE, (cam.height + TILE - 1) // TILE
    T = gw * gh
    cap = int(pair_capacity) if pair_capacity else default_pair_capacity(G)
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
    st = _State(mean2d=f(G, 2), conic_op=f(G, 4), depth=f(G), radii=i32(G, 2), rect=i32(G, 4), tiles_touched=i32(G),
                rgb=f(G, 3) if cam.mode == 0 else None, tile_count=i32(8 * T), tile_start=i32(T + 2), cursor=i32(8 * T), cap=cap)
    check(_lib.lib().siu3r_raster_bin(C.byref(cam), G, _p(means), _p(cov6), _p(opac), _p(colors), channels, _p(st["mean2d"]),
                                      _p(st["conic_op"]), _p(st["depth"]), _p(st["radii"]), _p(st["rect"]), _p(st["tiles_touched"]),
                                      _p(st["rgb"]), _p(st["tile_count"]), _p(st["tile_start"]), _p(st["cursor"]), cap, _stream()))
    # no read-back of D here (the CUDA originals resize their buffers behind one): the buffers are sized by the bound above, the
    # kernels clamp to it, and D is checked when it is first asked for (SplattingCUDA.forward does so once per call, after the
    # last view has been enqueued)
    st["D_dev"] = st["tile_start"][T + 1]
    st["keys"] = torch.empty((cap,), dtype=torch.int64, device=dev)
    st["ids"] = i32(cap)
    check(_lib.lib().siu3r_raster_sort(C.byref(cam), G, _p(st["rect"]), _p(st["depth"]), _p(st["tile_start"]), _p(st["cursor"]),
                                       _p(st["keys"]), _p(st["ids"]), cap, _stream()))
    return st


def rasterize_k2(cam: RasterCam, means, cov6, shs, opacities) -> Dict[str, torch.Tensor]:
    """means [G,3], cov6 [G,6], shs [G,ncoef,3], opacities [G] (fp32, GPU) -> image [3,H,W], radii [G,2] i32,
    depth [H,W], opacity [H,W], n_touched [G] i32 (+ binning state for parity tests / HBM accounting)."""
    _gpu(means, cov6, shs, opacities)
    means, cov6, shs, opacities = (t.contiguous().float() for t in (means, cov6, shs, opacities))
    G, dev = means.shape[0], means.device
    st = _bin_and_sort(cam, means, cov6, opacities, shs, shs.shape[1])
    H, W = cam.height, cam.width
    image = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    depth = torch.empty((H, W), dtype=torch.float32, device=dev)
    alpha = torch.empty((H, W), dtype=torch.float32, device=dev)
    n_touched = torch.empty((G,), dtype=torch.int32, device=dev)
    check(_lib.lib().siu3r_raster_composite_rgb(C.byref(cam), _p(st["tile_start"]), _p(st["ids"]), _p(st["mean2d"]), _p(st["conic_op"]),
                                                _p(st["depth"]), _p(st["rgb"]), _p(image), _p(depth), _p(alpha), _p(n_touched), G, _stream()))
    return dict(image=image, radii=st["radii"], depth=depth, opacity=alpha, n_touched=n_touched, state=st)


def rasterize_k3(cam: RasterCam, means, cov6, opacities, feats) -> Dict[str, torch.Tensor]:
    """feats [G,C] -> colors [H,W,C], alphas [H,W] (+ state)."""
    _gpu(means, cov6, opacities, feats)
    means, cov6, opacities, feats = (t.contiguous().float() for t in (means, cov6, opacities, feats))
    dev = means.device
    st = _bin_and_sort(cam, means, cov6, opacities, None, 0)
    H, W, Cc = cam.height, cam.width, feats.shape[1]
    out = torch.empty((H, W, Cc), dtype=torch.float32, device=dev)
    alpha = torch.empty((H, W), dtype=torch.float32, device=dev)
    check(_lib.lib().siu3r_raster_composite_feat(C.byref(cam), _p(st["tile_start"]), _p(st["ids"]), _p(st["mean2d"]), _p(st["conic_op"]),
                                                 _p(feats), Cc, _p(out), _p(alpha), _stream()))
    return dict(colors=out, alphas=alpha, radii=st["radii"], state=st)


def scale_inplace_(x: torch.Tensor, s: float):
    _gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    check(_lib.lib().siu3r_scale_inplace(_p(x), x.numel(), float(s), _stream()))
    return x


def quat_scale_to_cov6(quats_wxyz: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """[G,4] (w,x,y,z) + [G,3] -> [G,6] (gsplat's quat_scale_to_covar, upper triangle)."""
    _gpu(quats_wxyz, scales)
    q, sc = quats_wxyz.contiguous().float(), scales.contiguous().float()
    out = torch.empty((q.shape[0], 6), dtype=torch.float32, device=q.device)
    check(_lib.lib().siu3r_quat_scale_to_cov6(_p(q), _p(sc), _p(out), q.shape[0], _stream()))
    return out


def sh_eval(means: torch.Tensor, campos, sh: torch.Tensor, degree: int) -> torch.Tensor:
    """means [G,3], campos (3 floats, host), sh [G,ncoef,3] -> rgb [G,3] = max(SH . coeffs + 0.5, 0)."""
    _gpu(means, sh)
    means, sh = means.contiguous().float(), sh.contiguous().float()
    cam = (C.c_float * 3)(*[float(v) for v in campos])
    out = torch.empty((means.shape[0], 3), dtype=torch.float32, device=means.device)
    check(_lib.lib().siu3r_sh_eval(_p(means), C.cast(cam, C.c_void_p), _p(sh), sh.shape[1], int(degree), _p(out), means.shape[0], _stream()))
    return out


def blend_background_(colors: torch.Tensor, alphas: torch.Tensor, bg) -> torch.Tensor:
    """colors [H,W,C<=3] += (1 - alphas[H,W]) * bg, in place."""
    _gpu(colors, alphas)
    assert colors.is_contiguous() and alphas.is_contiguous() and colors.dtype == torch.float32
    b = (C.c_float * 3)(*([float(v) for v in bg] + [0.0] * (3 - len(bg))))
    check(_lib.lib().siu3r_blend_background(_p(colors), _p(alphas), C.cast(b, C.c_void_p), colors.shape[-1], alphas.numel(), _stream()))
    return colors


def algorithmic_bytes(G: int, G_v: int, D: int, P: int, channels: Optional[int] = None) -> int:
    """Algorithmic HBM bytes of one rendered view: RGB path 348 G + 48 G_v + 88 D + 20 P;
    C-channel path (44+4C) G + 36 G_v + (76+4C) D + (4C+4) P."""
    if channels is None:
        return 348 * G + 48 * G_v + 88 * D + 20 * P
    return (44 + 4 * channels) * G + 36 * G_v + (76 + 4 * channels) * D + (4 * channels + 4) * P
