"""`gsplat.rasterization` (@961678f4) as the reference calls it.

Pipeline call -- src/models/gaussian_renderer.py:92-106 (N-channel features, covariances given):

    colors, alphas, meta = rasterization(means=[G,3], quats=None, scales=None, covars=[G,3,3], opacities=[G], colors=[G,C],
                                         viewmats=[V,4,4] world->camera, Ks=[V,3,3] pixel units, width, height, sh_degree=None,
                                         near_plane, far_plane)

Viewer calls -- viewer.py:319-335 (`rasterize_splats`: quats + scales, SH colours [G,K,3], `sh_degree`, `radius_clip`, white `backgrounds`,
packed=True, absgrad / sparse_grad False, rasterize_mode="classic") and viewer.py:354-369 (`rasterize_qc_logits`: the same geometry with
[G, q * 21] feature colours, packed=False).

-> colors [V,H,W,C], alphas [V,H,W,1], meta.  All V views of a feature render go through the rasterizer in ONE call (camera array in device
memory, blockIdx.y = view); covariances are read as stored.  `viewmats` / `Ks` / `backgrounds` stay on the device: the kernels take the
pose from the tensors (siu3r_raster_project_dp), nothing is copied to the host per call.  Arguments that only steer gsplat's own
memory layout or its backward pass (`packed`, `sparse_grad`, `absgrad`, `channel_chunk`) do not change the forward result and are
accepted; arguments that select an algorithm this renderer does not implement are refused by name.

One extension, `finite_features` (default True).  For C >= 32 feature channels the blend runs on the matrix cores, where every pixel of an
8 x 8 quadrant takes part in every entry of the quadrant's list (weight 0 where the entry does not reach it; list tails padded with
Gaussian 0's row): bit-identical to gsplat's per-pixel walk for finite features, but an inf / NaN feature value turns into NaN in pixels
the Gaussian does not touch (0 * inf).  Callers whose features may be non-finite pass `finite_features=False` and get the 32-channel
kernel, whose pixels only see the rows that blend into them (tests/test_raster_gpu.py::test_k3_composite_with_a_non_finite_feature_row)."""
from __future__ import annotations

import torch

from .. import raster

_RESULT_NEUTRAL = ("packed", "sparse_grad", "absgrad", "channel_chunk", "segmented")


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, near_plane=0.01, far_plane=1e10, radius_clip=0.0,
                  eps2d=0.3, sh_degree=None, packed=True, tile_size=16, backgrounds=None, render_mode="RGB", sparse_grad=False, absgrad=False,
                  rasterize_mode="classic", channel_chunk=32, distributed=False, camera_model="pinhole", covars=None, finite_features=True,
                  **other):
    other = {k: v for k, v in other.items() if k not in _RESULT_NEUTRAL}
    if other:
        raise TypeError(f"gsplat.rasterization arguments outside the SIU3R call sites: {sorted(other)}")
    if tile_size != 16:
        raise NotImplementedError("tile_size must be 16 (the renderer's tile; gsplat's default and what the reference uses)")
    if render_mode != "RGB":
        raise NotImplementedError(f"render_mode={render_mode!r}: only 'RGB' (colours / features + alphas) is on the SIU3R paths")
    if rasterize_mode != "classic":
        raise NotImplementedError(f"rasterize_mode={rasterize_mode!r}: the reference renders 'classic' (no anti-aliasing compensation), viewer.py:317")
    if camera_model != "pinhole" or distributed:
        raise NotImplementedError("pinhole cameras on one device only")
    if covars is None:
        if quats is None or scales is None:
            raise ValueError("either covars or (quats, scales) is needed")
        covars = raster.quat_scale_to_cov6(quats, scales)  # (normalises the quaternions, like gsplat)
    V = viewmats.shape[0]
    dev = means.device
    pose = (viewmats, Ks)  # device tensors: never read on the host
    eye = torch.eye(4)
    cams = [raster.make_cam_k3(eye, 1.0, 1.0, 0.0, 0.0, width, height, near_plane=near_plane, far_plane=far_plane, eps2d=eps2d, radius_clip=radius_clip)
            for _ in range(V)]  # frame size / planes / thresholds; the pose fields are overwritten on the device
    bg = None
    if backgrounds is not None:
        bg = backgrounds.detach().to(device=dev, dtype=torch.float32)
        bg = bg[None].expand(V, -1) if bg.dim() == 1 else bg  # viewer.py:333 passes one [3] colour for its single camera
    meta = {"width": width, "height": height, "tile_size": 16, "n_cameras": V}
    if sh_degree is None:
        if colors.dim() != 2:
            raise NotImplementedError(f"colors must be [G, C] shared by all cameras (got {tuple(colors.shape)})")
        if colors.shape[1] == 3:  # three channels travel in the per-Gaussian record: the fused composite, no per-tile lists in HBM
            o = raster.rasterize_views_k3_rgb(cams, means, covars, opacities, colors, pose_dev=pose)
        else:
            o = raster.rasterize_views_k3(cams, means, covars, opacities, colors, pose_dev=pose, matrix_form=bool(finite_features))
        out, alphas = o["colors"], o["alphas"]
        meta["radii"] = o["radii"]
    else:
        # view-dependent colours: colors = SH coefficients [G, K, 3]; directions from the camera centre = inverse(viewmat)[:3, 3]
        if colors.dim() != 3 or colors.shape[2] != 3 or colors.shape[1] < (sh_degree + 1) ** 2:
            raise ValueError(f"sh_degree={sh_degree} needs colors [G, K >= {(sh_degree + 1) ** 2}, 3], got {tuple(colors.shape)}")
        vm = viewmats.detach().to(device=dev, dtype=torch.float32)
        campos = torch.linalg.inv(vm)[:, :3, 3].contiguous()  # [V, 3] on the device
        outs, als, radii = [], [], []
        for v in range(V):
            rgb = raster.sh_eval(means, campos[v], colors, sh_degree)
            o = raster.rasterize_views_k3_rgb([cams[v]], means, covars, opacities, rgb, pose_dev=(vm[v:v + 1], Ks[v:v + 1]))
            outs.append(o["colors"][0])
            als.append(o["alphas"][0])
            radii.append(o["radii"][0])
        out, alphas = torch.stack(outs), torch.stack(als)
        meta["radii"] = torch.stack(radii)
    if bg is not None:
        for v in range(V):
            raster.blend_background_(out[v], alphas[v], bg[v])
    return out, alphas[..., None], meta
