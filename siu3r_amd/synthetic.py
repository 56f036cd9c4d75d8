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
