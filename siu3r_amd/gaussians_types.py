"""Attribute bag crossing model -> renderer -> export (mirror of reference src/utils/gaussians_types.py:4-38)."""
from __future__ import annotations

from torch import Tensor


class Gaussians:
    FIELDS = ("means", "covariances", "harmonics", "opacities", "scales", "rotations")

    def __init__(self, means=None, covariances=None, harmonics=None, opacities=None, scales=None, rotations=None, **kwargs):
        self.means: Tensor = means
        self.covariances: Tensor = covariances
        self.harmonics: Tensor = harmonics
        self.opacities: Tensor = opacities
        self.scales: Tensor = scales
        self.rotations: Tensor = rotations
        for key, value in kwargs.items():
            setattr(self, key, value)

    def detach_cpu_copy(self) -> "Gaussians":
        out = Gaussians()
        for name, value in vars(self).items():
            setattr(out, name, value.detach().cpu() if isinstance(value, Tensor) else value)
        return out
