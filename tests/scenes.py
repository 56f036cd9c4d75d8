"""Seeded synthetic Gaussian scenes / cameras shared by the rasterizer tests and bench legs."""
import math

import torch


def random_scene(G, seed=0, spread=1.5, depth=(1.5, 8.0), scale=(0.01, 0.12), n_sh=25):
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


def look_at_camera(seed=0, jitter=0.3):
    """camera-to-world (OpenCV) near the origin looking down +z, small seeded SE(3) perturbation."""
    g = torch.Generator().manual_seed(1000 + seed)
    ang = (torch.rand(3, generator=g) * 2 - 1) * math.radians(10.0)
    cx, sx = math.cos(ang[0]), math.sin(ang[0])
    cy, sy = math.cos(ang[1]), math.sin(ang[1])
    cz, sz = math.cos(ang[2]), math.sin(ang[2])
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    c2w = torch.eye(4)
    c2w[:3, :3] = (Rz @ Ry @ Rx).float()
    c2w[:3, 3] = (torch.rand(3, generator=g) * 2 - 1) * jitter
    return c2w


def default_K():
    return torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]], dtype=torch.float32)
