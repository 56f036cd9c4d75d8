"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this).

CPU restatement (torch fp32) of LPIPS with the VGG16 backbone as the reference evaluator uses it (src/evaluator.py:55-57, 263:
torchmetrics LearnedPerceptualImagePatchSimilarity("vgg", normalize=True)).  torchmetrics (pinned 1.7.3: uv.lock:3373-3374, pyproject.toml:32) is a third-party
dependency that is NOT in /root/reference and not installed here: this follows its published algorithm
(torchmetrics/functional/image/lpips.py: class _LPIPS -- ScalingLayer, Vgg16 slices, _normalize_tensor with eps inside the square root,
NetLinLayer, _spatial_average) and is **parity unpinned**: there is no reference-held vector for it and no way to generate one offline.
Known-answer properties it is checked against instead: d(x, x) = 0, symmetry, non-negativity with non-negative lin weights, invariance of a
tap's distance to a positive rescaling of that tap's features."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

VGG_SLICES = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))
VGG_CHANNELS = (64, 128, 256, 512, 512)
SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)


def random_weights(seed: int = 0) -> Dict[str, torch.Tensor]:
    """a seeded stand-in with the schema of the real network (torchvision key names + lpips lin layers): He-scaled convolutions so that the
    activations neither die nor blow up over 13 layers, non-negative lin weights like the trained ones"""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    cin = 3
    for sl, c in zip(VGG_SLICES, VGG_CHANNELS):
        for i in sl:
            sd[f"features.{i}.weight"] = torch.randn(c, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
            sd[f"features.{i}.bias"] = torch.randn(c, generator=g) * 0.05
            cin = c
    for k, c in enumerate(VGG_CHANNELS):
        sd[f"lin{k}.model.1.weight"] = torch.rand(1, c, 1, 1, generator=g) * (2.0 / c)
    return sd


def features(sd: Dict[str, torch.Tensor], x01: torch.Tensor, prefix: str = "features."):
    shift = torch.tensor(SHIFT).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE).view(1, 3, 1, 1)
    h = (2.0 * x01.float() - 1.0 - shift) / scale          # normalize=True, then the ScalingLayer
    taps = []
    for k, sl in enumerate(VGG_SLICES):
        if k:
            h = F.max_pool2d(h, kernel_size=2, stride=2)
        for i in sl:
            h = F.relu(F.conv2d(h, sd[f"{prefix}{i}.weight"].float(), sd[f"{prefix}{i}.bias"].float(), padding=1))
        taps.append(h)
    return taps


def lpips(sd: Dict[str, torch.Tensor], img0: torch.Tensor, img1: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """img0, img1 [N,3,H,W] in [0,1] -> [N]"""
    f0, f1 = features(sd, img0), features(sd, img1)
    total = torch.zeros(img0.shape[0])
    for k, (a, b) in enumerate(zip(f0, f1)):
        na = a / torch.sqrt(eps + (a * a).sum(1, keepdim=True))
        nb = b / torch.sqrt(eps + (b * b).sum(1, keepdim=True))
        w = sd[f"lin{k}.model.1.weight"].float().view(1, -1, 1, 1)
        total += (w * (na - nb) ** 2).sum(1).mean((1, 2))
    return total
