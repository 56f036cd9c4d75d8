"""Gaussian splat rasterizer front-end: torch tensors -> the C ABI of csrc/raster.hip.

Two conventions behind one tile-binned HIP renderer (include/siu3r_hip.h, siu3r_raster_cam):
  K2 = diff-gaussian-rasterization-w-pose semantics (reference call site src/models/cuda_splatting.py:90-118)
  K3 = gsplat.rasterization semantics (reference call site src/models/gaussian_renderer.py:92-106)
All V views of a call travel through every launch together (the reference loops over views in Python,
cuda_splatting.py:82-121): project -> one stable depth radix sort per view -> depth-ordered coarse bins -> composite.
"""
from __future__ import annotations

import collections
import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from ._lib import RasterCam, check
from .ops import _gpu, _p, _stream

TILE = 16


def _set(arr, values):
    for i, v in enumerate(values):
        arr[i] = float(v)


def make_cam_k2(w2c, full_proj, tanfovx, tanfovy, campos, bg, width, height, sh_degree=4, sh_band4=False, nt_post_blend=True, near=None,
                far=None) -> RasterCam:
    """sh_degree -1: the colours handed to the rasterizer are precomputed [G,1,3] values, blended as given (no SH, no clamp).
    w2c / full_proj / tanfov / campos None + (near, far): the pose comes from device tensors (`pose_c2w` of the rasterize_* calls,
    siu3r_raster_project_c2w) and only the projection planes travel in the block."""
    c = RasterCam()
    c.mode, c.width, c.height = 0, int(width), int(height)
    if w2c is not None:
        _set(c.w2c, w2c.reshape(-1).tolist())
        _set(c.proj, full_proj.reshape(-1).tolist())
        c.tanfovx, c.tanfovy = float(tanfovx), float(tanfovy)
        _set(c.campos, campos)
    if near is not None:
        c.k2_near, c.k2_far = float(near), float(far)
    _set(c.bg, bg)
    c.sh_degree, c.sh_band4, c.k2_znear_cull = int(sh_degree), int(bool(sh_band4)), 0.2
    c.alpha_min, c.alpha_max, c.t_min, c.dilation = 1.0 / 255.0, 0.99, 1e-4, 0.3
    c.eps2d, c.extent_sigma = 0.3, 3.33
    c.near_plane = 0.2
    c.nt_post_blend = int(bool(nt_post_blend))
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
    c.nt_post_blend = 1
    return c


def cov6_from_cov3x3(cov: torch.Tensor) -> torch.Tensor:
    """[G,3,3] -> [G,6] upper-triangular (xx,xy,xz,yy,yz,zz), the order of torch.triu_indices(3,3)
    (reference cuda_splatting.py:107,115).  Pure indexing (layout plumbing)."""
    return torch.stack((cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]), dim=-1).contiguous()


def geometry(width: int, height: int, G: int) -> Dict[str, int]:
    out = (C.c_int32 * 8)()
    check(_lib.lib().siu3r_raster_geometry(int(width), int(height), int(G), out))
    return dict(gw=out[0], gh=out[1], T=out[2], cb=out[3], NB=out[4], nchunks_sort=out[5], nchunks_bin=out[6], cam_bytes=out[7])


def default_entry_capacity(G: int) -> int:
    """Coarse-bin entries per view (8 B each): a Gaussian lands in one 64 x 64 px bin unless it straddles a bin border;
    3 per Gaussian covers the 2 M-Gaussian 1080p stress scene (large splats) with margin.  An overflow is detected after
    the frame (stats) and the call is repeated with the exact count."""
    return int(min(max(3 * G + 65536, 1 << 18), (1 << 31) - 1024))


def default_pair_capacity(G: int) -> int:
    """(tile, Gaussian) pairs per view for the materialised tile lists (4 B each): 8 per Gaussian, at least 1 M."""
    return int(min(max(8 * G, 1 << 20), (1 << 31) - 1024))


# Capacities that a scene of a given size needed before: a caller that renders many frames of similar scenes (an evaluation loop) pays
# the detect-and-repeat of an overflowing bound once, not on every call.  key: (kind, G, V, width, height) -> capacity
_CAPACITY_HINTS: Dict[tuple, int] = {}


def _hinted(kind: str, G: int, V: int, W: int, H: int, default: int) -> int:
    return max(default, _CAPACITY_HINTS.get((kind, G, V, W, H), 0))


def _remember(kind: str, G: int, V: int, W: int, H: int, needed: int):
    _CAPACITY_HINTS[(kind, G, V, W, H)] = int(min(needed + needed // 4 + 1024, (1 << 31) - 1024))


class RasterOverflow(RuntimeError):
    """call_id: sequence number of the deferred call whose views overflowed (None for the synchronous paths)"""

    def __init__(self, msg, call_id=None):
        super().__init__(msg)
        self.call_id = call_id


class _State(dict):
    """Buffers of one rasterizer call (V views).  The per-tile Gaussian lists are not needed to render an RGB view; they are
    produced on first access of "tile_start" / "ids" (tests, exports).  "D" / "Gv" / "E" read the device-side totals."""

    def stats(self) -> torch.Tensor:
        if not dict.__contains__(self, "stats_host"):
            dict.__setitem__(self, "stats_host", dict.__getitem__(self, "stats").cpu())
        return dict.__getitem__(self, "stats_host")

    def totals(self, k: int) -> List[int]:
        return [int(x) for x in self.stats()[:, k].tolist()]

    def verify(self):
        """Raise when a capacity-bounded buffer overflowed (only needed by callers that passed check=False)."""
        st = self.stats()
        e_max, cap_e = int(st[:, 2].max()), dict.__getitem__(self, "cap_e")
        if e_max > cap_e:
            raise RasterOverflow(f"rasterizer coarse-bin entries overflowed: E = {e_max} > entry_capacity {cap_e} (pass entry_capacity >= {e_max})")
        # the materialised (tile, Gaussian) pair lists of the N-channel path (tl_scan sets flag bit 1 when a view's pairs exceed
        # pair_capacity: far splats would silently be missing from tiles).  Deferred callers (check_overflow=False) must call verify().
        if dict.__contains__(self, "tile_start_all"):
            T, cap_d = dict.__getitem__(self, "T"), dict.__getitem__(self, "cap_d")
            d_max = int(dict.__getitem__(self, "tile_start_all")[:, T + 1].max())
            if d_max > cap_d:  # (the sticky flag bit of an earlier, repeated attempt is not consulted: the scan's own total is)
                raise RasterOverflow(f"rasterizer tile-pair lists overflowed: D = {d_max} > pair_capacity {cap_d} (pass pair_capacity >= {d_max})")

    def _lists(self):
        if not dict.__contains__(self, "ids"):
            V, T = dict.__getitem__(self, "V"), dict.__getitem__(self, "T")
            dev = dict.__getitem__(self, "stats").device
            hint_key = ("pairs", dict.__getitem__(self, "G"), V, dict.__getitem__(self, "W"), dict.__getitem__(self, "H"))
            cap_d = dict.__getitem__(self, "cap_d_hint") or max(1, max(self.totals(1)))
            cap_d = max(cap_d, _CAPACITY_HINTS.get(hint_key, 0))
            while True:
                tcount = torch.empty((V, T), dtype=torch.int32, device=dev)
                tstart = torch.empty((V, T + 2), dtype=torch.int32, device=dev)
                ids = torch.empty((V, cap_d), dtype=torch.int32, device=dev)
                check(_lib.lib().siu3r_raster_tile_lists(dict.__getitem__(self, "cams"), V, _p(dict.__getitem__(self, "bin_start")),
                                                         _p(dict.__getitem__(self, "entries")), dict.__getitem__(self, "cap_e"), _p(tcount),
                                                         _p(tstart), _p(ids), cap_d, _p(dict.__getitem__(self, "stats")), _stream()))
                if dict.__getitem__(self, "defer"):
                    break
                d_max = int(tstart[:, T + 1].max())
                if d_max <= cap_d:
                    break
                cap_d = d_max  # the bound was too small: repeat with the exact pair count
                _remember(*hint_key, d_max)
            dict.__setitem__(self, "cap_d", cap_d)
            dict.__setitem__(self, "tile_start_all", tstart)
            dict.__setitem__(self, "ids_all", ids)
            dict.__setitem__(self, "ids", ids[0])
            dict.__setitem__(self, "tile_start", tstart[0])

    def __getitem__(self, k):
        if k in ("ids", "tile_start", "ids_all", "tile_start_all", "cap_d"):
            self._lists()
        elif k == "D":
            return self.totals(1)[0]
        elif k == "Gv":
            return self.totals(0)[0]
        elif k == "E":
            return self.totals(2)[0]
        return dict.__getitem__(self, k)


def _cam_array(cams: Sequence[RasterCam]):
    arr = (RasterCam * len(cams))()
    for i, c in enumerate(cams):
        C.memmove(C.byref(arr, i * C.sizeof(RasterCam)), C.byref(c), C.sizeof(RasterCam))
    return arr


def _cov_stride(cov: torch.Tensor) -> int:
    """[G,6] upper-triangular or [G,3,3] full covariances (read in place by the projection kernel)."""
    if cov.dim() == 2 and cov.shape[1] == 6:
        return 6
    if cov.dim() == 3 and cov.shape[1:] == (3, 3):
        return 9
    raise RuntimeError(f"covariances must be [G,6] or [G,3,3], got {tuple(cov.shape)}")


def _pose_dev(pose_dev, V, dev):
    """(viewmats [V,4,4], Ks [V,3,3]) device tensors -> contiguous fp32 on `dev` (layout plumbing only; nothing is read back)"""
    vm, Ks = pose_dev
    vm = vm.detach().to(device=dev, dtype=torch.float32).contiguous()
    Ks = Ks.detach().to(device=dev, dtype=torch.float32).contiguous()
    if tuple(vm.shape) != (V, 4, 4) or tuple(Ks.shape) != (V, 3, 3):
        raise RuntimeError(f"device-side poses must be viewmats [V,4,4] and Ks [V,3,3] with V = {V}, got {tuple(vm.shape)} / {tuple(Ks.shape)}")
    return vm, Ks


def _project_sort_bin(cams: Sequence[RasterCam], means, cov, opac, colors, channels, entry_capacity=None, check_overflow=True,
                      sh_planar=False, pose_dev=None, pose_c2w=None) -> _State:
    """pose_dev: None, or (viewmats [V,4,4] world->camera, Ks [V,3,3] pixel units) as DEVICE tensors (gsplat family): the kernels take the
    pose from them (siu3r_raster_project_dp) and `cams` only carries frame size, planes and thresholds.
    pose_c2w: None, or (extrinsics [V,4,4] camera-to-world, normalised intrinsics [V,3,3], t_scale) as DEVICE tensors + a float, either
    family: what SplattingCUDA.forward receives; inverse / fov / projection are derived on the device (siu3r_raster_project_c2w)."""
    V, G, dev = len(cams), means.shape[0], means.device
    arr = _cam_array(cams)
    geo = geometry(cams[0].width, cams[0].height, G)
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
    cap_e = int(entry_capacity) if entry_capacity else _hinted("entries", G, V, cams[0].width, cams[0].height, default_entry_capacity(G))
    st = _State(V=V, G=G, W=cams[0].width, H=cams[0].height, T=geo["T"], geo=geo, cams=arr, cams_dev=torch.empty((V * geo["cam_bytes"],), dtype=torch.uint8, device=dev),
                rec=f(V, G, 12), radii=i32(V, G, 2), rect=i32(V, G, 4), tiles_touched_all=i32(V, G), keys=i32(V, G), keys_b=i32(V, G), sorted_ids=i32(V, G), ids_b=i32(V, G),
                stats=torch.empty((V, 4), dtype=torch.int64, device=dev), defer=not check_overflow, cap_d_hint=None)
    st["tiles_touched"] = st["tiles_touched_all"][0]
    lib = _lib.lib()
    if pose_c2w is not None:
        e_, k_, t_scale = pose_c2w
        e_, k_ = st["pose_dev"] = _pose_dev((e_, k_), V, dev)  # (kept with the state: the launches read them asynchronously)
        check(lib.siu3r_raster_project_c2w(arr, V, _p(st["cams_dev"]), _p(e_), _p(k_), float(t_scale), G, _p(means), _p(cov), _cov_stride(cov), _p(opac),
                                           _p(colors), channels, int(bool(sh_planar)), _p(st["rec"]), _p(st["radii"]), _p(st["rect"]),
                                           _p(st["tiles_touched_all"]), _p(st["keys"]), _p(st["stats"]), _stream()))
    elif pose_dev is None:
        check(lib.siu3r_raster_project(arr, V, _p(st["cams_dev"]), G, _p(means), _p(cov), _cov_stride(cov), _p(opac), _p(colors), channels,
                                       int(bool(sh_planar)), _p(st["rec"]), _p(st["radii"]), _p(st["rect"]), _p(st["tiles_touched_all"]),
                                       _p(st["keys"]), _p(st["stats"]), _stream()))
    else:
        vm, Ks = st["pose_dev"] = _pose_dev(pose_dev, V, dev)  # (kept with the state: the launches below read them asynchronously)
        check(lib.siu3r_raster_project_dp(arr, V, _p(st["cams_dev"]), _p(vm), _p(Ks), G, _p(means), _p(cov), _cov_stride(cov), _p(opac), _p(colors),
                                          channels, int(bool(sh_planar)), _p(st["rec"]), _p(st["radii"]), _p(st["rect"]),
                                          _p(st["tiles_touched_all"]), _p(st["keys"]), _p(st["stats"]), _stream()))
    rs_hist, rs_tot = i32(V, 256, geo["nchunks_sort"]), i32(V, 256)
    check(lib.siu3r_raster_sort(V, G, _p(st["keys"]), _p(st["keys_b"]), _p(st["sorted_ids"]), _p(st["ids_b"]), _p(rs_hist), _p(rs_tot), _p(st["stats"]), _stream()))
    bin_hist, bin_tot = i32(V, geo["NB"], geo["nchunks_bin"]), i32(V, geo["NB"])
    st["bin_start"] = i32(V, geo["NB"] + 1)
    st["entries"] = torch.empty((V, cap_e, 2), dtype=torch.int32, device=dev)
    st["cap_e"] = cap_e
    check(lib.siu3r_raster_bin(arr, V, G, _p(st["keys"]), _p(st["sorted_ids"]), _p(st["rect"]), _p(bin_hist), _p(bin_tot), _p(st["bin_start"]),
                               _p(st["entries"]), cap_e, _p(st["stats"]), _stream()))
    return st


# ---- deferred overflow checks.  check_overflow=True reads the counters right after the call: a device synchronisation per call, behind
# which the host prepares the next call while the GPU idles (25 % of a 6-view 512^2 frame).  check_overflow="deferred" enqueues an
# asynchronous copy of the counters into pinned memory instead and looks at it when the NEXT deferred call starts (by then the copy has
# long landed) or when check_pending() is called; an overflowing view cannot pass unnoticed meanwhile: the composite kernel fills it
# with NaN.  On detection the needed capacity is remembered (later calls of that scene size are sized right) and RasterOverflow names
# the call that overflowed -- its sequence number (`RasterOverflow.call_id`, returned as out["call_id"] by the deferred call) and its
# scene size -- so that the caller knows WHICH result to discard and repeat; the exception may surface at the start of a LATER deferred
# call (which has not run yet) or at check_pending().  Callers in this repository: bench.py's render legs (check_pending() after the
# timed calls); evaluate.py renders with the synchronous check (check_overflow=True: transparent repeat).
_PENDING: "collections.deque" = collections.deque()
_PINNED_FREE: list = []
_CALL_SEQ = [0]


def _defer_check(st: "_State"):
    stats = dict.__getitem__(st, "stats")
    i = next((i for i, h in enumerate(_PINNED_FREE) if h.shape == stats.shape), None)
    host = _PINNED_FREE.pop(i) if i is not None else torch.empty(stats.shape, dtype=stats.dtype, pin_memory=True)
    host.copy_(stats, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _CALL_SEQ[0] += 1
    _PENDING.append((ev, host, dict.__getitem__(st, "cap_e"), (dict.__getitem__(st, "G"), dict.__getitem__(st, "V"), dict.__getitem__(st, "W"), dict.__getitem__(st, "H")), _CALL_SEQ[0]))
    return _CALL_SEQ[0]


def check_pending(block: bool = True):
    """Verify the deferred calls (all of them, waiting for their counters, or with block=False only those whose counters have
    arrived).  Raises RasterOverflow for the first call that overflowed its entry buffers (its views were rendered as NaN)."""
    bad = None
    while _PENDING:
        ev, host, cap_e, key, seq = _PENDING[0]
        if not block and not ev.query():
            break
        ev.synchronize()
        _PENDING.popleft()
        e_max = int(host[:, 2].max())
        if len(_PINNED_FREE) < 8:
            _PINNED_FREE.append(host)
        if e_max > cap_e:
            _remember("entries", *key, e_max)
            bad = bad or (e_max, cap_e, key, seq)
    if bad:
        raise RasterOverflow(f"deferred rasterizer call #{bad[3]} (G, V, W, H = {bad[2]}) overflowed its coarse-bin entries: E = {bad[0]} > entry_capacity "
                             f"{bad[1]}; its views were rendered as NaN.  The needed capacity is remembered: repeat that call", call_id=bad[3])


def _with_retry(run, entry_capacity, check_overflow):
    """run(entry_capacity) -> result dict with "state".  The coarse-bin entries are sized by a bound; when a frame overflows it (the
    kernels drop the excess, flag it and poison the view) the whole call is repeated once with the exact count: the CUDA originals
    resize their buffers behind a device-to-host copy instead.  check_overflow: True = look now (synchronises), "deferred" = look when
    the next deferred call starts / at check_pending(), False = the caller calls state.verify()."""
    if check_overflow == "deferred":
        check_pending(block=False)
        out = run(entry_capacity)
        out["call_id"] = _defer_check(out["state"])
        return out
    out = run(entry_capacity)
    if not check_overflow:
        return out
    st = out["state"]
    e_max = max(st.totals(2))
    if e_max > st["cap_e"]:
        _remember("entries", st["G"], st["V"], st["W"], st["H"], e_max)
        out = run(e_max)
        out["state"].verify()
    return out


def rasterize_views_k2(cams: Sequence[RasterCam], means, cov6, shs, opacities, want_n_touched=True, entry_capacity=None,
                       check_overflow=True, sh_planar=False, pose_c2w=None) -> Dict[str, torch.Tensor]:
    """V views of one Gaussian set.  means [G,3]; cov6 [G,6] (upper triangle) or [G,3,3]; shs [G,ncoef,3], or with sh_planar
    [G,3,25] (Gaussians.harmonics as stored); opacities [G] (fp32, GPU) -> image [V,3,H,W], radii [V,G,2] i32, depth [V,H,W],
    opacity [V,H,W], n_touched [V,G] i32 (None when not wanted) + the call's state."""
    _gpu(means, cov6, shs, opacities)
    means, cov6, shs, opacities = (t.contiguous().float() for t in (means, cov6, shs, opacities))
    V, G, dev = len(cams), means.shape[0], means.device
    H, W = cams[0].height, cams[0].width
    ncoef = shs.shape[2] if sh_planar else shs.shape[1]

    def run(cap):
        st = _project_sort_bin(cams, means, cov6, opacities, shs, ncoef, cap, check_overflow, sh_planar, pose_c2w=pose_c2w)
        image = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((V, H, W), dtype=torch.float32, device=dev)
        alpha = torch.empty((V, H, W), dtype=torch.float32, device=dev)
        n_touched = torch.empty((V, G), dtype=torch.int32, device=dev) if want_n_touched else None
        check(_lib.lib().siu3r_raster_composite_rgb(st["cams"], V, _p(st["cams_dev"]), G, _p(st["bin_start"]), _p(st["entries"]), st["cap_e"],
                                                    _p(st["rec"]), _p(image), _p(depth), _p(alpha), _p(n_touched), _stream()))
        return dict(image=image, radii=st["radii"], depth=depth, opacity=alpha, n_touched=n_touched, state=st)

    return _with_retry(run, entry_capacity, check_overflow)


def rasterize_k2(cam: RasterCam, means, cov6, shs, opacities, **kw) -> Dict[str, torch.Tensor]:
    """single view: image [3,H,W], radii [G,2], depth [H,W], opacity [H,W], n_touched [G] (+ state)."""
    o = rasterize_views_k2([cam], means, cov6, shs, opacities, **kw)
    return dict(image=o["image"][0], radii=o["radii"][0], depth=o["depth"][0], opacity=o["opacity"][0],
                n_touched=None if o["n_touched"] is None else o["n_touched"][0], state=o["state"])


def tune(key: int, value: int) -> None:
    """siu3r_raster_tune: key 0 -- 1 = the 32-channel kernel for every N-channel composite (features that may be non-finite; A/B), 0 = default;
    key 1 -- accumulator blocks per chunk of the matrix-core composite (1 .. 6).  Process-wide, not for concurrent use."""
    check(_lib.lib().siu3r_raster_tune(int(key), int(value)))


def rasterize_views_k3(cams: Sequence[RasterCam], means, cov6, opacities, feats, entry_capacity=None, pair_capacity=None,
                       check_overflow=True, pose_dev=None, matrix_form=True, pose_c2w=None) -> Dict[str, torch.Tensor]:
    """feats [G,C] -> colors [V,H,W,C], alphas [V,H,W] (+ state).  gsplat semantics: the per-tile lists are materialised once; for
    C >= 32 they are cut per 8 x 8 quadrant and blended on the matrix cores, all channels at once (identical bits for FINITE features).
    matrix_form=False: the 32-channel kernel, whose pixels only ever see the rows that blend into them -- for features that may hold
    inf / NaN (include/siu3r_hip.h, siu3r_raster_composite_feat_ws)."""
    _gpu(means, cov6, opacities, feats)
    if check_overflow == "deferred":
        raise ValueError('check_overflow="deferred" is only safe on the RGB composites (an overflowing view is NaN-poisoned there); the '
                         "N-channel composite has no such marker: pass True (check now) or False (and call state.verify())")
    means, cov6, opacities, feats = (t.contiguous().float() for t in (means, cov6, opacities, feats))
    V, G, dev = len(cams), means.shape[0], means.device
    H, W, Cc = cams[0].height, cams[0].width, feats.shape[1]

    def run(cap):
        st = _project_sort_bin(cams, means, cov6, opacities, None, 0, cap, check_overflow, pose_dev=pose_dev, pose_c2w=pose_c2w)
        st["cap_d_hint"] = int(pair_capacity) if pair_capacity else default_pair_capacity(G)
        out = torch.empty((V, H, W, Cc), dtype=torch.float32, device=dev)
        alpha = torch.empty((V, H, W), dtype=torch.float32, device=dev)
        lib = _lib.lib()
        tstart, ids_all, cap_d = st["tile_start_all"], st["ids_all"], st["cap_d"]
        # workspace of the per-quadrant lists (C >= 32: every wave of the composite walks its own 8 x 8 quadrant's list)
        ws = torch.empty((int(lib.siu3r_raster_composite_feat_ws_bytes(W, H, V, cap_d)) // 4,), dtype=torch.int32, device=dev) if (Cc >= 32 and matrix_form) else None
        st["feat_ws"] = ws
        check(lib.siu3r_raster_composite_feat_ws(st["cams"], V, _p(st["cams_dev"]), G, _p(tstart), _p(ids_all), cap_d, _p(st["rec"]), _p(feats), Cc,
                                                 _p(out), _p(alpha), _p(ws), 0 if ws is None else ws.numel() * 4, _stream()))
        return dict(colors=out, alphas=alpha, radii=st["radii"], state=st)

    return _with_retry(run, entry_capacity, check_overflow)


def rasterize_views_k3_rgb(cams: Sequence[RasterCam], means, cov6, opacities, rgb, entry_capacity=None, check_overflow=True,
                           pose_dev=None) -> Dict[str, torch.Tensor]:
    """gsplat semantics with THREE precomputed colour channels (rgb [G,3]) -> colors [V,H,W,3], alphas [V,H,W] (+ state), through the
    fused sort-free composite: the colour travels in the per-Gaussian record and no per-tile list is written to HBM (the N-channel
    path materialises the lists because every 32-channel chunk re-walks them)."""
    _gpu(means, cov6, opacities, rgb)
    means, cov6, opacities, rgb = (t.contiguous().float() for t in (means, cov6, opacities, rgb))
    assert rgb.shape[1] == 3
    V, G, dev = len(cams), means.shape[0], means.device
    H, W = cams[0].height, cams[0].width

    def run(cap):
        st = _project_sort_bin(cams, means, cov6, opacities, rgb, 3, cap, check_overflow, pose_dev=pose_dev)
        out = torch.empty((V, H, W, 3), dtype=torch.float32, device=dev)
        alpha = torch.empty((V, H, W), dtype=torch.float32, device=dev)
        check(_lib.lib().siu3r_raster_composite_rgb(st["cams"], V, _p(st["cams_dev"]), G, _p(st["bin_start"]), _p(st["entries"]), st["cap_e"],
                                                    _p(st["rec"]), _p(out), None, _p(alpha), None, _stream()))
        return dict(colors=out, alphas=alpha, radii=st["radii"], state=st)

    return _with_retry(run, entry_capacity, check_overflow)


def rasterize_k3(cam: RasterCam, means, cov6, opacities, feats, **kw) -> Dict[str, torch.Tensor]:
    o = rasterize_views_k3([cam], means, cov6, opacities, feats, **kw)
    return dict(colors=o["colors"][0], alphas=o["alphas"][0], radii=o["radii"][0], state=o["state"])


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
    """means [G,3], campos (3 floats on the host, or a DEVICE tensor of 3 floats: no read-back), sh [G,ncoef,3] -> rgb [G,3] =
    max(SH . coeffs + 0.5, 0)."""
    _gpu(means, sh)
    means, sh = means.contiguous().float(), sh.contiguous().float()
    out = torch.empty((means.shape[0], 3), dtype=torch.float32, device=means.device)
    if isinstance(campos, torch.Tensor) and campos.is_cuda:
        cp = campos.detach().float().contiguous()
        assert cp.numel() == 3
        check(_lib.lib().siu3r_sh_eval_dp(_p(means), _p(cp), _p(sh), sh.shape[1], int(degree), _p(out), means.shape[0], _stream()))
        return out
    cam = (C.c_float * 3)(*[float(v) for v in campos])
    check(_lib.lib().siu3r_sh_eval(_p(means), C.cast(cam, C.c_void_p), _p(sh), sh.shape[1], int(degree), _p(out), means.shape[0], _stream()))
    return out


def blend_background_(colors: torch.Tensor, alphas: torch.Tensor, bg) -> torch.Tensor:
    """colors [H,W,C] += (1 - alphas[H,W]) * bg, in place.  bg: C <= 3 floats on the host, or a DEVICE tensor of C floats (any C, no read-back)."""
    _gpu(colors, alphas)
    assert colors.is_contiguous() and alphas.is_contiguous() and colors.dtype == torch.float32
    if isinstance(bg, torch.Tensor) and bg.is_cuda:
        b = bg.detach().float().contiguous()
        assert b.numel() == colors.shape[-1], (tuple(b.shape), tuple(colors.shape))
        check(_lib.lib().siu3r_blend_background_dp(_p(colors), _p(alphas), _p(b), colors.shape[-1], alphas.numel(), _stream()))
        return colors
    b = (C.c_float * 3)(*([float(v) for v in bg] + [0.0] * (3 - len(bg))))
    check(_lib.lib().siu3r_blend_background(_p(colors), _p(alphas), C.cast(b, C.c_void_p), colors.shape[-1], alphas.numel(), _stream()))
    return colors


def algorithmic_bytes(G: int, G_v: int, D: int, P: int, channels: Optional[int] = None) -> int:
    """Algorithmic HBM bytes of one rendered view: RGB path 348 G + 48 G_v + 88 D + 20 P;
    C-channel path (44+4C) G + 36 G_v + (76+4C) D + (4C+4) P."""
    if channels is None:
        return 348 * G + 48 * G_v + 88 * D + 20 * P
    return (44 + 4 * channels) * G + 36 * G_v + (76 + 4 * channels) * D + (4 * channels + 4) * P
