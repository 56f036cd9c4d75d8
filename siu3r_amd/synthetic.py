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
