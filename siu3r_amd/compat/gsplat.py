"""`gsplat.rasterization` (@961678f4) as the reference calls it -- src/models/gaussian_renderer.py:92-106:

    colors, alphas, meta = rasterization(means=[G,3], quats=None, scales=None, covars=[G,3,3], opacities=[G], colors=[G,C],
                                         viewmats=[V,4,4] world->camera, Ks=[V,3,3] pixel units, width, height, sh_degree=None,
                                         near_plane, far_plane)

-> colors [V,H,W,C], alphas [V,H,W,1], meta.  All V views go through the rasterizer in ONE call (camera array in device memory,
blockIdx.y = view); the 3x3 covariances are read as stored.  The viewer's form (quats + scales + SH colours, viewer.py:301-336) is
`siu3r_amd.gaussian_renderer.rasterize_splats`."""
from __future__ import annotations

import torch

from .. import raster


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, covars=None, sh_degree=None, near_plane=0.01,
                  far_plane=1e10, radius_clip=0.0, eps2d=0.3, backgrounds=None, **unsupported):
    if unsupported:
        raise TypeError(f"gsplat.rasterization arguments outside the SIU3R call sites: {sorted(unsupported)}")
    if covars is None:
        if quats is None or scales is None:
            raise ValueError("either covars or (quats, scales) is needed")
        covars = raster.quat_scale_to_cov6(quats, scales)
    if sh_degree is not None:
        raise NotImplementedError("view-dependent colours: use siu3r_amd.gaussian_renderer.rasterize_splats (viewer.py:301-336)")
    vm, K = viewmats.detach().float().cpu(), Ks.detach().float().cpu()
    cams = [raster.make_cam_k3(vm[i], float(K[i, 0, 0]), float(K[i, 1, 1]), float(K[i, 0, 2]), float(K[i, 1, 2]), width, height,
                               near_plane=near_plane, far_plane=far_plane, eps2d=eps2d, radius_clip=radius_clip) for i in range(vm.shape[0])]
    o = raster.rasterize_views_k3(cams, means, covars, opacities, colors)
    out, alphas = o["colors"], o["alphas"][..., None]
    if backgrounds is not None:
        for i in range(out.shape[0]):
            raster.blend_background_(out[i], o["alphas"][i], [float(v) for v in backgrounds[i].detach().float().cpu()])
    meta = {"radii": o["radii"], "width": width, "height": height, "tile_size": 16, "n_cameras": len(cams)}
    return out, alphas, meta
