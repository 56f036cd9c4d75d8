"""`diff_gaussian_rasterization` (the MonoGS "w-pose" fork, @43e21bff) as the reference calls it -- src/models/cuda_splatting.py:82-118:

    settings = GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
                                             projmatrix_raw, sh_degree, campos, prefiltered, debug)
    image, radii, depth, opacity, n_touched = GaussianRasterizer(settings)(means3D=..., means2D=..., shs=... | colors_precomp=...,
                                                                           opacities=[G,1], cov3D_precomp=[G,6], theta=None, rho=None)

Conventions of that API, kept here: `viewmatrix` / `projmatrix` are ROW-vector matrices (the transposes of world->camera and of
proj @ world->camera); `cov3D_precomp` holds the upper triangle (xx, xy, xz, yy, yz, zz); `shs` is [G, (deg+1)^2, 3]; `opacities` [G, 1];
outputs: image [3,H,W], radii [G] int32, depth [1,H,W], opacity [1,H,W], n_touched [G] int32.  `scales` / `rotations` instead of
`cov3D_precomp`, and the pose gradients `theta` / `rho`, are not used by the reference's inference path and are rejected."""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch

from .. import raster


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: Optional[torch.Tensor] = None
    sh_degree: int = 0
    campos: Optional[torch.Tensor] = None
    prefiltered: bool = False
    debug: bool = False


def make_cam(s: GaussianRasterizationSettings, sh_band4: bool = False) -> "raster.RasterCam":
    """the settings of one view as the rasterizer's camera block (column-vector matrices)"""
    assert s.scale_modifier == 1.0, "scale_modifier != 1 is not used by the reference"
    view = s.viewmatrix.detach().float().cpu()
    full = s.projmatrix.detach().float().cpu()
    campos = s.campos if s.campos is not None else torch.linalg.inv(view.T)[:3, 3]
    return raster.make_cam_k2(view.T.contiguous(), full.T.contiguous(), float(s.tanfovx), float(s.tanfovy), [float(v) for v in campos.detach().float().cpu()],
                              [float(v) for v in s.bg.detach().float().cpu()], s.image_width, s.image_height, sh_degree=s.sh_degree, sh_band4=sh_band4)


class GaussianRasterizer:
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        self.raster_settings = raster_settings

    def markVisible(self, positions):  # the fork's frustum test; the reference never calls it
        raise NotImplementedError("markVisible is not on the SIU3R inference path")

    def forward(self, means3D, means2D=None, opacities=None, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide exactly one of either SHs or precomputed colors!")
        if cov3D_precomp is None or scales is not None or rotations is not None:
            raise Exception("the SIU3R path passes cov3D_precomp (cuda_splatting.py:115); scale/rotation pairs are not supported")
        if theta is not None or rho is not None:
            raise Exception("pose gradients (theta, rho) are training-only")
        s = self.raster_settings
        cam = make_cam(s)
        if shs is None:
            # precomputed colours: blended as given, like the upstream package (no SH evaluation, no clamp: feature-valued or negative
            # colours keep their sign) -- siu3r_raster_cam.sh_degree = -1, colours [G, 1, 3]
            shs = colors_precomp.reshape(-1, 1, 3)
            cam.sh_degree = -1
        o = raster.rasterize_k2(cam, means3D, cov3D_precomp, shs, opacities.reshape(-1))
        return o["image"], o["radii"][:, 0].contiguous(), o["depth"][None], o["opacity"][None], o["n_touched"]

    __call__ = forward
