"""The Gaussian set handed from the model to the renderer, the lifting step and the PLY writer.

Same contract as the reference's container (src/utils/gaussians_types.py): six tensor fields that always exist (None until
set), arbitrary extra attributes (semantic_labels, instance_labels, seg_query_class_logits, ...) that travel along, and
``detach_cpu_copy()``.  Implemented as a thin namespace over one dict so that field iteration, copying and validation share
a single code path."""
from __future__ import annotations

from typing import Any, Dict, Iterator, Tuple

import torch


class Gaussians:
    FIELDS: Tuple[str, ...] = ("means", "covariances", "harmonics", "opacities", "scales", "rotations")
    __slots__ = ("_attrs",)

    def __init__(self, **fields: Any):
        object.__setattr__(self, "_attrs", {name: None for name in self.FIELDS})
        self._attrs.update(fields)

    # attribute access goes through the dict: core fields and late additions are indistinguishable to callers
    def __getattr__(self, name: str) -> Any:
        try:
            return object.__getattribute__(self, "_attrs")[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name: str, value: Any) -> None:
        self._attrs[name] = value

    def items(self) -> Iterator[Tuple[str, Any]]:
        return iter(self._attrs.items())

    def as_dict(self) -> Dict[str, Any]:
        return dict(self._attrs)

    def map_tensors(self, fn) -> "Gaussians":
        """New container with fn applied to every tensor attribute (lists of tensors included); other values are shared."""
        def conv(v):
            if isinstance(v, torch.Tensor):
                return fn(v)
            if isinstance(v, (list, tuple)) and v and all(isinstance(t, torch.Tensor) for t in v):
                return type(v)(fn(t) for t in v)
            return v

        return Gaussians(**{k: conv(v) for k, v in self._attrs.items()})

    def detach_cpu_copy(self) -> "Gaussians":
        return self.map_tensors(lambda t: t.detach().cpu())

    def __repr__(self) -> str:
        shape = lambda v: tuple(v.shape) if isinstance(v, torch.Tensor) else type(v).__name__
        return "Gaussians(" + ", ".join(f"{k}={shape(v)}" for k, v in self._attrs.items() if v is not None) + ")"
