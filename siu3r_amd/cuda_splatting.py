"""Mirror of the reference's src/models/cuda_splatting.py (render_cuda, get_projection_matrix) and
src/utils/projection.py:247-261 (get_fov) on top of the HIP rasterizer (K2 semantics).  Camera matrices are 4x4
host-side parameters (computed in fp32 on the CPU, as plain parameter preparation); all per-Gaussian / per-pixel
arithmetic runs in csrc/raster.hip."""
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
                sh_band4: bool = False, return_aux: bool = False):
    """reference signature cuda_splatting.py:46-60 (batch = views).  Returns (images [b,3,h,w], depths [b,h,w])."""
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    assert cam_rot_delta is None and cam_trans_delta is None, "pose gradients are training-only (out of scope)"
    b = extrinsics.shape[0]
    h, w = image_shape
    n = gaussian_sh_coefficients.shape[-1]
    degree = isqrt(n) - 1
    ext = extrinsics.detach().float().cpu()
    fov = get_fov(intrinsics)
    fov_x, fov_y = fov.unbind(-1)
    tan_x, tan_y = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
    proj = get_projection_matrix(near.detach().float().cpu(), far.detach().float().cpu(), fov_x, fov_y)
    w2c = torch.linalg.inv(ext)
    full = proj @ w2c  # column-vector form of the reference's row-vector view @ proj (cuda_splatting.py:74-77)
    images, depths, aux = [], [], []
    prev = None
    for i in range(b):
        means = gaussian_means[i]
        # the reference's callers pass the same Gaussians expanded over the views: repack them (6-entry covariances, 'g xyz n ->
        # g n xyz' coefficients (:65), a 157 MB copy at 524 288 Gaussians) once, not per view
        key = (gaussian_covariances[i].data_ptr(), gaussian_sh_coefficients[i].data_ptr())
        if prev is None or prev[0] != key:
            prev = (key, raster.cov6_from_cov3x3(gaussian_covariances[i]), gaussian_sh_coefficients[i].permute(0, 2, 1).contiguous())
        cov6, shs = prev[1], prev[2]
        cam = raster.make_cam_k2(w2c[i], full[i], float(tan_x[i]), float(tan_y[i]), ext[i, :3, 3].tolist(),
                                 background_color[i].detach().float().cpu().tolist(), w, h, sh_degree=degree, sh_band4=sh_band4)
        out = raster.rasterize_k2(cam, means, cov6, shs, gaussian_opacities[i])
        images.append(out["image"])
        depths.append(out["depth"])
        aux.append(out)
    res = (torch.stack(images), torch.stack(depths))
    for out in aux:  # one synchronisation per call, after every view has been enqueued: the deferred pair-count check
        out["state"]["D"]
    return res + (aux,) if return_aux else res
