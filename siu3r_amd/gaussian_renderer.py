"""Mirror of the reference's src/models/gaussian_renderer.py (SplattingCUDA.forward, :29-116) and of the
query-class-logit lifting in src/pipeline.py:132-193, on top of the HIP rasterizer."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib, raster
from ._lib import check
from .cuda_splatting import render_cuda
from .gaussians_types import Gaussians
from .ops import _gpu, _p, _stream


class SplattingCUDA:
    def __init__(self, deferred_overflow_check: bool = False) -> None:
        """deferred_overflow_check False (default): the rasterizer's overflow counters are read right after every call (one device
        synchronisation per call, transparent repeat with the exact buffer size), as the CUDA original's buffer resize does.  True: the
        colour render does not synchronise; the counters are looked at when the next render starts or at check_pending() (an
        overflowing view is NaN meanwhile, never subtly wrong; the needed buffer size is remembered and RasterOverflow asks for the call
        to be repeated): a render loop then keeps the GPU busy while the host prepares the next call (bench.py's render legs)."""
        self.near = 0.1
        self.far = 100.0
        self.scale_factor = 1 / self.near
        self.background_color = torch.tensor([0.0, 0.0, 0.0], dtype=torch.float32)
        self.deferred_overflow_check = deferred_overflow_check

    @staticmethod
    def check_pending():
        """barrier of the deferred overflow checks: raises raster.RasterOverflow if a render since the last check overflowed"""
        raster.check_pending(block=True)

    def forward(self, gaussians: Gaussians, extrinsics, intrinsics, image_shape, render_color: bool = True,
                render_feature: bool = False, render_id: bool = False, render_qc_logits: bool = False,
                cam_rot_delta=None, cam_trans_delta=None):
        """reference signature gaussian_renderer.py:29-41.  extrinsics [b,v,4,4] camera-to-world (OpenCV),
        intrinsics [b,v,3,3] normalised.  NOTE (quirk 1, reproduced): means / covariances are rescaled x10 / x100
        IN PLACE on the Gaussians (:43-46).  Camera tensors on the GPU are consumed there: the x10 translation scale (:44), the inverse,
        the field of view and the projection matrix are derived inside the projection call (siu3r_raster_project_c2w) -- with
        deferred_overflow_check the whole forward enqueues without a host synchronisation
        (tests/test_raster_gpu.py::test_splatting_forward_does_not_synchronise)."""
        b, v, _, _ = extrinsics.shape
        on_dev = extrinsics.is_cuda and intrinsics.is_cuda
        if on_dev:
            extrinsics, intr, t_scale = extrinsics.detach().float(), intrinsics.detach().float(), self.scale_factor
        else:
            extrinsics = extrinsics.detach().float().cpu().clone()
            extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * self.scale_factor
            intr, t_scale = intrinsics.detach().float().cpu(), 1.0
        raster.scale_inplace_(gaussians.covariances, self.scale_factor ** 2)
        raster.scale_inplace_(gaussians.means, self.scale_factor)
        near, far = 1.0, self.far * self.scale_factor
        color = depth = None
        all_qc: Optional[List[torch.Tensor]] = None
        if render_color:
            colors, depths = [], []
            for i in range(b):  # each batch item has its own Gaussians; its v views are rendered back to back
                c_i, d_i = render_cuda(
                    extrinsics[i], intr[i], torch.full((v,), near), torch.full((v,), far), image_shape,
                    self.background_color[None].repeat(v, 1), gaussians.means[i][None].expand(v, -1, -1),
                    gaussians.covariances[i][None].expand(v, -1, -1, -1), gaussians.harmonics[i][None].expand(v, -1, -1, -1),
                    gaussians.opacities[i][None].expand(v, -1), check_overflow="deferred" if self.deferred_overflow_check else True,
                    translation_scale=t_scale)
                colors.append(c_i)
                depths.append(d_i)
            color = torch.stack(colors).clamp_(0.0, 1.0)  # (:73) clamp is pure data conditioning on the output buffer
            depth = torch.stack(depths)
        if render_qc_logits:
            height, width = image_shape
            all_qc = []
            for i in range(b):
                means, opac = gaussians.means[i], gaussians.opacities[i]
                cov6 = gaussians.covariances[i]  # [G,3,3], read in place
                qcl = gaussians.seg_query_class_logits[i]  # [n, q, c]
                n, q, c = qcl.shape
                feats = qcl.reshape(n, q * c)
                cams, pose = [], None
                if on_dev:  # frame size and planes only: the pose is taken from the device tensors
                    cams = [raster.make_cam_k3(torch.eye(4), 1.0, 1.0, 0.0, 0.0, width, height, near_plane=near, far_plane=far) for _ in range(v)]
                    pose = (extrinsics[i], intr[i], t_scale)
                else:
                    for j in range(v):
                        K = intr[i, j].clone()
                        K[0, :] *= width
                        K[1, :] *= height
                        w2c = torch.linalg.inv(extrinsics[i, j])
                        cams.append(raster.make_cam_k3(w2c, K[0, 0], K[1, 1], K[0, 2], K[1, 2], width, height, near_plane=near, far_plane=far))
                out = raster.rasterize_views_k3(cams, means, cov6, opac, feats, pose_c2w=pose)  # all v views in one call: [v, h, w, q*c]
                # reference layout 'n h w (q c) -> n q c h w' as a view of the channel-last buffer
                all_qc.append(out["colors"].view(v, height, width, q, c).permute(0, 3, 4, 1, 2))
        return {"render_color": color, "render_depth": depth, "render_qc_logits": all_qc}

    __call__ = forward


def lift_query_class_logits(render_qc_logits: List[torch.Tensor], query_scores: List[List[float]], num_queries: int = 100,
                            label_ids_to_fuse=(0, 1), sem_threshold: float = 0.3):
    """Lifting of rendered query x class logit maps to per-pixel semantic / instance ids
    (reference src/pipeline.py:132-193).  render_qc_logits: list(b) of [v, q, c+1, h, w] (views of channel-last buffers).
    Returns (sem_ids [b,v,h,w] int64, ins_ids [b,v,h,w] int64, seg_infos)."""
    all_sem, all_ins, seg_infos = [], [], []
    stuff_mask = 0
    for s in label_ids_to_fuse:
        stuff_mask |= 1 << s
    for qc, q_score in zip(render_qc_logits, query_scores):
        _gpu(qc)
        v, q, c, h, w = qc.shape
        cl = qc.permute(0, 3, 4, 1, 2)  # channel-last [v,h,w,q,c]
        if not cl.is_contiguous():
            cl = cl.contiguous()
        dev = qc.device
        sem = torch.empty((v, h, w), dtype=torch.int64, device=dev)
        ins = torch.empty((v, h, w), dtype=torch.int64, device=dev)
        first = torch.empty((q,), dtype=torch.int32, device=dev)
        qlab = torch.empty((q,), dtype=torch.int32, device=dev)
        check(_lib.lib().siu3r_lift_ids(_p(cl), v, h, w, q, c, sem_threshold, num_queries, stuff_mask, _p(sem), _p(ins),
                                        _p(first), _p(qlab), _stream()))
        qlab_h = qlab.cpu().tolist()
        info = []
        for q_idx, sc in enumerate(q_score):
            if q_idx >= q or qlab_h[q_idx] < 0:
                continue
            info.append({"id": q_idx + 1, "label_id": qlab_h[q_idx], "was_fused": False, "score": sc})
        for stuff in label_ids_to_fuse:
            for i_ in info:
                if i_["label_id"] == stuff + 1:
                    i_["was_fused"] = True
                    i_["id"] = num_queries + stuff + 1
        all_sem.append(sem)
        all_ins.append(ins)
        seg_infos.append(info)
    return torch.stack(all_sem), torch.stack(all_ins), seg_infos


_PREPARED: dict = {}  # rasterize_splats' one-entry cache of the view-independent preparation (release_prepared_splats() drops it)


def _clear_prepared():
    for f in _PREPARED.get("finalizers", ()):  # one finalizer per live source tensor: detached when the entry goes, so that a long-lived
        f.detach()                             # tensor edited in place between frames does not collect one per rebuild
    _PREPARED.clear()


def release_prepared_splats():
    _clear_prepared()


def _drop_prepared(token):
    if _PREPARED.get("token") is token:
        _clear_prepared()


def _tensor_sig(t):
    return (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype)


def rasterize_splats(splats: dict, camtoworlds: torch.Tensor, Ks: torch.Tensor, width: int, height: int, sh_degree: int = 4,
                     radius_clip: float = 0.1, near_plane: float = 0.01, far_plane: float = 1e10, backgrounds=(1.0, 1.0, 1.0)):
    """The reference viewer's render call (viewer.py:301-336 + 376-401): gsplat.rasterization semantics with
    quats (w,x,y,z) / log-scales / logit-opacities / SH coefficients `sh0` [G,1,3] + `shN` [G,K-1,3] as loaded from the
    exported PLY, pixel-unit intrinsics, white background, radius_clip = 0.1 px.  camtoworlds [C,4,4], Ks [C,3,3]; when they are
    device tensors (the viewer uploads its camera state, viewer.py:391-392) the pose never comes back to the host.
    Returns (render_colors [C,H,W,3], render_alphas [C,H,W,1], info)."""
    import weakref

    from . import raster

    # view-independent preparation (covariances from quats / scales, activations, the concatenated coefficient block): computed once per
    # splat set -- a viewer renders many frames of one scene (the block alone is 600 MB for 2 M Gaussians).  The one-entry cache holds
    # WEAK references to the source tensors it was built from and is valid while every one of them is alive and the tensor passed now has
    # the same storage address, version counter (an in-place edit bumps `_version`, which views and `.detach()` share with their base),
    # shape, strides and dtype -- so another view of the same live storage still hits.  It is dropped as soon as one of the sources dies
    # (a new scene whose tensors reuse the old addresses is then a rebuild): it neither serves a stale scene nor pins the prepared tensors.
    names = ("means", "quats", "scales", "opacities", "sh0", "shN")
    src = [splats.get(k) for k in names]
    prep = _PREPARED.get("entry")
    valid = prep is not None and all((r is None and t is None) or (r is not None and t is not None and r() is not None and sig == _tensor_sig(t))
                                     for (r, sig), t in zip(prep[0], src))
    if not valid:
        means = splats["means"].float()
        cov6 = raster.quat_scale_to_cov6(splats["quats"], torch.exp(splats["scales"].float()))
        opac = torch.sigmoid(splats["opacities"].float())
        coeffs = torch.cat([splats["sh0"], splats["shN"]], 1).float() if splats.get("shN") is not None else splats["sh0"].float()
        token = object()
        refs, fins = [], []
        for t in src:
            if t is None:
                refs.append((None, None))
            else:
                refs.append((weakref.ref(t), _tensor_sig(t)))
                fins.append(weakref.finalize(t, _drop_prepared, token))
        _clear_prepared()
        prep = _PREPARED["entry"] = (refs, means, cov6, opac, coeffs)
        _PREPARED["token"] = token
        _PREPARED["finalizers"] = fins
    _, means, cov6, opac, coeffs = prep
    assert coeffs.shape[1] >= (sh_degree + 1) ** 2
    cols, alphas, visible, pairs = [], [], [], []
    on_dev = camtoworlds.is_cuda and Ks.is_cuda
    if on_dev:
        c2w_d = camtoworlds.detach().float()
        vm_d, Ks_d = torch.linalg.inv(c2w_d), Ks.detach().float()
        eye = torch.eye(4)
    else:
        c2w_h, Ks_h = camtoworlds.detach().cpu().float(), Ks.detach().cpu().float()
    for i in range(camtoworlds.shape[0]):  # colours are view-dependent (SH): one call per camera
        if on_dev:
            cam = raster.make_cam_k3(eye, 1.0, 1.0, 0.0, 0.0, width, height, near_plane, far_plane, radius_clip=radius_clip)
            rgb = raster.sh_eval(means, c2w_d[i, :3, 3], coeffs, sh_degree)
            pose = (vm_d[i:i + 1], Ks_d[i:i + 1])
        else:
            K = Ks_h[i]
            cam = raster.make_cam_k3(torch.linalg.inv(c2w_h[i]), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), width, height, near_plane,
                                     far_plane, radius_clip=radius_clip)
            rgb = raster.sh_eval(means, c2w_h[i, :3, 3].tolist(), coeffs, sh_degree)
            pose = None
        o = raster.rasterize_views_k3_rgb([cam], means, cov6, opac, rgb, pose_dev=pose)  # three channels: the fused composite, no tile lists in HBM
        o = dict(colors=o["colors"][0], alphas=o["alphas"][0], state=o["state"])
        raster.blend_background_(o["colors"], o["alphas"], backgrounds)
        cols.append(o["colors"])
        alphas.append(o["alphas"][..., None])
        visible.append(o["state"]["tiles_touched"])
        pairs.append(o["state"]["D"])
    return torch.stack(cols), torch.stack(alphas), dict(tiles_touched=visible, tile_pairs=pairs)
