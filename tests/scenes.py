"""Seeded synthetic Gaussian scenes / cameras shared by the rasterizer tests and bench legs."""
import math

import torch


from siu3r_amd.synthetic import random_scene  # noqa: E402,F401  (one generator for tests, tools and the bench)


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
