"""PLY -> splats -> novel view: the non-interactive part of the reference's viewer.py.

`load_ply` mirrors GaussianRenderer.load_ply (viewer.py:134-296): properties are read by NAME (f_rest_* / scale_* / rot_* sorted by
their numeric suffix), the SH rest coefficients come back as [P, K-1, 3] ('(P, 3, K-1)' transposed, :165-170, 215-224), and -- the
reference's default -- a 5-pixel border of every H x W view is cropped away because inaccurate intrinsics leave bad border Gaussians
(:232-279).  The tensors stay in FILE units: scale = log, opacity = the post-sigmoid value the exporter wrote (ply_export.py:81), and
`rasterize_splats` applies exp / sigmoid on top exactly like viewer.py:313-314 (the viewer's double squash of opacity is a quirk of
the reference that is kept).  Rendering runs on the HIP rasterizer (gaussian_renderer.rasterize_splats, gsplat
semantics); the viser / nerfview event loop is out of scope."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .ply_export import read_ply_vertices

_SH_REST_TO_DEGREE = {0: 0, 9: 1, 24: 2, 45: 3, 72: 4}


def load_ply(path, H: int = 256, W: int = 256, crop: bool = True, num_classes: int = 21, device="cpu") -> Dict[str, torch.Tensor]:
    v = read_ply_vertices(path)
    names = v.dtype.names
    col = lambda n: np.asarray(v[n])
    by_suffix = lambda pre: sorted((n for n in names if n.startswith(pre)), key=lambda x: int(x.split("_")[-1]))
    xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
    P = xyz.shape[0]
    rest = by_suffix("f_rest_")
    if len(rest) not in _SH_REST_TO_DEGREE:
        raise ValueError(f"{len(rest)} f_rest_* properties do not form a spherical-harmonics basis")
    deg = _SH_REST_TO_DEGREE[len(rest)]
    f_dc = np.stack((col("f_dc_0"), col("f_dc_1"), col("f_dc_2")), axis=1)[:, :, None]                      # [P, 3, 1]
    f_rest = (np.stack([col(n) for n in rest], axis=1) if rest else np.zeros((P, 0))).reshape(P, 3, (deg + 1) ** 2 - 1)
    scales = np.stack([col(n) for n in by_suffix("scale_")], axis=1)
    rots = np.stack([col(n) for n in sorted((n for n in names if n.startswith("rot")), key=lambda x: int(x.split("_")[-1]))], axis=1)
    qc_names = [n for n in names if n.startswith("seg_query_class_logits_")]
    nq = len(qc_names) // num_classes
    qc = np.stack([col(n) for n in qc_names], axis=1) if qc_names else np.zeros((P, 0))
    t = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=device)
    out = dict(means=t(xyz), quats=t(rots), scales=t(scales), opacities=t(col("opacity")),
               sh0=t(f_dc).transpose(1, 2).contiguous(), shN=t(f_rest).transpose(1, 2).contiguous(),
               semantic_label=t(col("semantic_label"), torch.long), instance_label=t(col("instance_label"), torch.long),
               qc_logits=t(qc).view(-1, nq, num_classes) if nq else t(qc).view(P, 0, num_classes))
    if crop:
        if P % (H * W) != 0:
            raise ValueError(f"{P} vertices are not a whole number of {H} x {W} views: pass the model's image size or crop=False")
        c = 5
        for k, x in out.items():
            y = x.view(-1, H, W, *x.shape[1:])[:, c:H - c, c:W - c]
            out[k] = y.reshape(-1, *x.shape[1:]).contiguous()
    out["max_sh_degree"] = deg
    return out


def render_view(splats: Dict[str, torch.Tensor], camtoworld: torch.Tensor, K: torch.Tensor, width: int, height: int,
                sh_degree: Optional[int] = None, radius_clip: float = 0.1):
    """GaussianRenderer._viewer_render_fn (viewer.py:376-401): one camera (c2w 4x4, pixel-unit K 3x3) -> RGB [H, W, 3] in [0, 1]
    (white background, radius_clip 0.1 px) on the HIP rasterizer."""
    from .gaussian_renderer import rasterize_splats

    dev = "cuda"
    if all(v.is_cuda for v in splats.values() if isinstance(v, torch.Tensor)):
        s = splats  # (rasterize_splats keeps its view-independent preparation with the dict: frame after frame of one scene)
    else:
        s = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in splats.items()}
    deg = s.get("max_sh_degree", 4) if sh_degree is None else sh_degree
    colors, alphas, info = rasterize_splats(s, camtoworld[None].float(), K[None].float(), width, height, sh_degree=deg, radius_clip=radius_clip)
    return colors[0].clamp(0.0, 1.0), alphas[0], info
