"""Seeded synthetic cameras for benchmarks / smoke tests (context view 0 = identity, the others small
SE(3) perturbations, rotation <= 10 deg, translation <= jitter)."""
import math

import torch


def perturbed_camera(seed: int, jitter: float = 0.03) -> torch.Tensor:
    g = torch.Generator().manual_seed(1000 + seed)
    ang = (torch.rand(3, generator=g) * 2 - 1) * math.radians(10.0)
    cx, sx, cy, sy, cz, sz = math.cos(ang[0]), math.sin(ang[0]), math.cos(ang[1]), math.sin(ang[1]), math.cos(ang[2]), math.sin(ang[2])
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    c2w = torch.eye(4)
    c2w[:3, :3] = (Rz @ Ry @ Rx).float()
    c2w[:3, 3] = (torch.rand(3, generator=g) * 2 - 1) * jitter
    return c2w


def target_views(n: int, seed: int = 0) -> torch.Tensor:
    """[n,4,4] camera-to-world; view 0 is the identity (the first context view's frame)."""
    return torch.stack([torch.eye(4)] + [perturbed_camera(seed + i) for i in range(1, n)])


def default_intrinsics() -> torch.Tensor:
    return torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]], dtype=torch.float32)


def random_scene(G, seed=0, spread=1.5, depth=(1.5, 8.0), scale=(0.01, 0.12), n_sh=25):
    """Seeded random Gaussian scene in front of a camera at the origin looking down +z:
    means [G,3], covariances [G,3,3] (random rotation x diag(scale^2)), opacities [G], SH [G,3,n_sh]."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    means = torch.stack(((r(G) * 2 - 1) * spread, (r(G) * 2 - 1) * spread, depth[0] + r(G) * (depth[1] - depth[0])), -1)
    q = torch.randn(G, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    x, y, z, w = q.unbind(-1)
    R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                     2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)), -1).view(G, 3, 3)
    s = scale[0] + r(G, 3) * (scale[1] - scale[0])
    cov = R @ torch.diag_embed(s * s) @ R.transpose(1, 2)
    opac = 0.05 + 0.9 * r(G)
    sh = (r(G, 3, n_sh) * 2 - 1) * 0.5
    return means.float(), cov.float(), opac.float(), sh.float()


def pixel_aligned_scene(H: int, W: int, views: int = 2, seed: int = 0, n_sh: int = 25):
    """The kind of Gaussian set the network emits for an indoor pair (model.py:289-304: view-major, then row-major pixels): ONE
    Gaussian per pixel of each context view, on that pixel's ray at a smooth seeded depth (1.5 .. 3.5 units), about one pixel
    footprint in size.  Used where the renderer is timed: with synthetic (random) network weights the predicted means are
    noise and leave ~6 % of the Gaussians in view, which says nothing about a renderer.
    Returns means [G,3], covariances [G,3,3], opacities [G], harmonics [G,3,n_sh] with G = views * H * W (CPU fp32)."""
    g = torch.Generator().manual_seed(seed)
    K = default_intrinsics()
    fx, fy, cx, cy = K[0, 0] * W, K[1, 1] * H, K[0, 2] * W, K[1, 2] * H
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32) + 0.5, torch.arange(W, dtype=torch.float32) + 0.5, indexing="ij")
    means, covs = [], []
    for v in range(views):
        c2w = torch.eye(4) if v == 0 else perturbed_camera(seed + 100 + v, jitter=0.15)
        ph = torch.rand(4, generator=g) * 6.28
        z = 2.5 + 0.6 * torch.sin(xs / W * 5.0 + ph[0]) * torch.cos(ys / H * 4.0 + ph[1]) + 0.35 * torch.sin(ys / H * 9.0 + ph[2]) \
            + 0.05 * torch.rand(H, W, generator=g)
        pc = torch.stack(((xs - cx) / fx * z, (ys - cy) / fy * z, z), -1).reshape(-1, 3)
        means.append(pc @ c2w[:3, :3].T + c2w[:3, 3])
        s = (z.reshape(-1, 1) / fx) * (0.6 + 0.9 * torch.rand(H * W, 3, generator=g))
        q = torch.randn(H * W, 4, generator=g)
        q = q / q.norm(dim=-1, keepdim=True)
        x, y, zq, w = q.unbind(-1)
        R = torch.stack((1 - 2 * (y * y + zq * zq), 2 * (x * y - zq * w), 2 * (x * zq + y * w), 2 * (x * y + zq * w), 1 - 2 * (x * x + zq * zq),
                         2 * (y * zq - x * w), 2 * (x * zq - y * w), 2 * (y * zq + x * w), 1 - 2 * (x * x + y * y)), -1).view(-1, 3, 3)
        covs.append(R @ torch.diag_embed(s * s) @ R.transpose(1, 2))
    G = views * H * W
    opac = 0.3 + 0.65 * torch.rand(G, generator=g)
    sh = (torch.rand(G, 3, n_sh, generator=g) * 2 - 1) * 0.1
    sh[:, :, 0] = (torch.rand(G, 3, generator=g) - 0.5) / 0.28209479177387814
    return torch.cat(means).float(), torch.cat(covs).float(), opac.float(), sh.float()
