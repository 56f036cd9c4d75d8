"""Mirror of the reference's src/models/cuda_splatting.py (render_cuda, get_projection_matrix) and
src/utils/projection.py:247-261 (get_fov) on top of the HIP rasterizer (K2 semantics).  Camera tensors that live on the GPU stay
there: inverse, field of view and projection matrix are derived by a one-thread-per-view kernel inside the projection call
(siu3r_raster_project_c2w) and the host never reads a pose value (no synchronisation per render); camera tensors on the CPU are
prepared on the CPU in fp32 as before (the oracle's tests take that route).  All per-Gaussian / per-pixel arithmetic runs in
csrc/raster.hip."""
from __future__ import annotations

from math import isqrt

import torch

from . import raster


def get_fov(intrinsics: torch.Tensor) -> torch.Tensor:
    """reference utils/projection.py:247-261: fov from K^-1 rays (acos of the dot product of the edge rays)."""
    K = intrinsics.detach().float().cpu()
    inv = torch.linalg.inv(K)

    def ray(v):
        r = inv @ torch.tensor(v, dtype=torch.float32)
        return r / r.norm(dim=-1, keepdim=True)

    left, right, top, bottom = ray([0, 0.5, 1]), ray([1, 0.5, 1]), ray([0.5, 0, 1]), ray([0.5, 1, 1])
    return torch.stack(((left * right).sum(-1).acos(), (top * bottom).sum(-1).acos()), dim=-1)


def get_projection_matrix(near, far, fov_x, fov_y) -> torch.Tensor:
    """reference cuda_splatting.py:16-43 (z in [0,1], w = z_view)."""
    tan_x, tan_y = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
    top, right = tan_y * near, tan_x * near
    bottom, left = -top, -right
    (b,) = near.shape
    r = torch.zeros((b, 4, 4), dtype=torch.float32)
    r[:, 0, 0] = 2 * near / (right - left)
    r[:, 1, 1] = 2 * near / (top - bottom)
    r[:, 0, 2] = (right + left) / (right - left)
    r[:, 1, 2] = (top + bottom) / (top - bottom)
    r[:, 3, 2] = 1
    r[:, 2, 2] = far / (far - near)
    r[:, 2, 3] = -(far * near) / (far - near)
    return r


def render_cuda(extrinsics, intrinsics, near, far, image_shape, background_color, gaussian_means, gaussian_covariances,
                gaussian_sh_coefficients, gaussian_opacities, use_sh: bool = True, cam_rot_delta=None, cam_trans_delta=None,
                sh_band4: bool = False, return_aux: bool = False, entry_capacity=None, check_overflow=True, translation_scale: float = 1.0):
    """reference signature cuda_splatting.py:46-60 (batch = views).  Returns (images [b,3,h,w], depths [b,h,w]); return_aux adds the
    per-call outputs (radii, n_touched, opacity, binning state).  entry_capacity: optional size of the coarse-bin entry buffers (an
    overflow of the default bound is detected and the call repeated with the exact size).  check_overflow: True (synchronous, as the
    CUDA original's buffer resize is) / "deferred" / False: see raster._with_retry.  translation_scale: the extrinsics' translation is
    multiplied by it inside the pose preparation (SplattingCUDA.forward's x10 scene rescale, gaussian_renderer.py:43-44, without a copy).
    extrinsics / intrinsics on the GPU: consumed there (no read-back; near / far / background_color are small host-side parameters and
    should be CPU tensors then -- GPU ones are fetched, which synchronises)."""
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    assert cam_rot_delta is None and cam_trans_delta is None, "pose gradients are training-only (out of scope)"
    b = extrinsics.shape[0]
    h, w = image_shape
    n = gaussian_sh_coefficients.shape[-1]
    degree = isqrt(n) - 1
    on_dev = extrinsics.is_cuda and intrinsics.is_cuda
    near_h, far_h = near.detach().float().cpu(), far.detach().float().cpu()
    if on_dev:
        ext_d, intr_d = extrinsics.detach().float(), intrinsics.detach().float()
    else:
        ext = extrinsics.detach().float().cpu()
        if translation_scale != 1.0:
            ext = ext.clone()
            ext[..., :3, 3] = ext[..., :3, 3] * translation_scale
        fov = get_fov(intrinsics)
        fov_x, fov_y = fov.unbind(-1)
        tan_x, tan_y = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
        proj = get_projection_matrix(near_h, far_h, fov_x, fov_y)
        w2c = torch.linalg.inv(ext)
        full = proj @ w2c  # column-vector form of the reference's row-vector view @ proj (cuda_splatting.py:74-77)
    # the reference's callers pass the same Gaussians expanded over the views (gaussian_renderer.py:50-67): consecutive views that
    # share their Gaussian storage go through the rasterizer as ONE call (blockIdx.y = view)
    groups, i = [], 0
    while i < b:
        key = (gaussian_means[i].data_ptr(), gaussian_covariances[i].data_ptr(), gaussian_sh_coefficients[i].data_ptr(), gaussian_opacities[i].data_ptr())
        j = i + 1
        while j < b and key == (gaussian_means[j].data_ptr(), gaussian_covariances[j].data_ptr(), gaussian_sh_coefficients[j].data_ptr(),
                                gaussian_opacities[j].data_ptr()):
            j += 1
        groups.append((i, j))
        i = j
    images, depths, aux = [], [], []
    bg_h = background_color.detach().float().cpu()
    for (i0, i1) in groups:
        if on_dev:
            cams = [raster.make_cam_k2(None, None, None, None, None, bg_h[i].tolist(), w, h, sh_degree=degree, sh_band4=sh_band4, near=float(near_h[i]),
                                       far=float(far_h[i])) for i in range(i0, i1)]
            pose = (ext_d[i0:i1], intr_d[i0:i1], float(translation_scale))
        else:
            cams = [raster.make_cam_k2(w2c[i], full[i], float(tan_x[i]), float(tan_y[i]), ext[i, :3, 3].tolist(), bg_h[i].tolist(), w, h, sh_degree=degree,
                                       sh_band4=sh_band4) for i in range(i0, i1)]
            pose = None
        # the projection kernel reads the 3x3 covariances and the [g, xyz, n] coefficients as stored: neither the 6-entry
        # repacking (:107,115) nor the 'g xyz n -> g n xyz' rearrangement (:65) is materialised
        sh_i = gaussian_sh_coefficients[i0]
        planar = sh_i.shape[-1] == 25
        out = raster.rasterize_views_k2(cams, gaussian_means[i0], gaussian_covariances[i0], sh_i if planar else sh_i.permute(0, 2, 1).contiguous(),
                                        gaussian_opacities[i0], want_n_touched=return_aux, entry_capacity=entry_capacity, sh_planar=planar,
                                        check_overflow=check_overflow, pose_c2w=pose)
        images.append(out["image"])
        depths.append(out["depth"])
        aux.append(out)
    res = (torch.cat(images), torch.cat(depths))
    return res + (aux,) if return_aux else res
