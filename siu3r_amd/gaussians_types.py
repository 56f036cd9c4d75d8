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

    def __init__(self, *core: Any, **fields: Any):
        """Gaussians(means, covariances, harmonics, opacities, scales, rotations) positionally, like the reference's constructor
        (utils/gaussians_types.py:5-13), and / or by keyword (any extra attribute travels along)."""
        if len(core) > len(self.FIELDS):
            raise TypeError(f"Gaussians takes at most {len(self.FIELDS)} positional fields ({len(core)} given)")
        object.__setattr__(self, "_attrs", {name: None for name in self.FIELDS})
        for name, value in zip(self.FIELDS, core):
            if name in fields:
                raise TypeError(f"Gaussians got multiple values for '{name}'")
            self._attrs[name] = value
        self._attrs.update(fields)

    # attribute access goes through the dict: core fields and late additions are indistinguishable to callers
    def __getattr__(self, name: str) -> Any:
        if name == "__dict__":  # vars(g): the attribute bag itself (the reference's container is a plain object)
            return object.__getattribute__(self, "_attrs")
        try:
            return object.__getattribute__(self, "_attrs")[name]
        except (KeyError, AttributeError):  # AttributeError: `_attrs` not restored yet (copy / pickle probe dunder names first)
            raise AttributeError(name) from None

    def __setattr__(self, name: str, value: Any) -> None:
        self._attrs[name] = value

    # copy.copy / copy.deepcopy / pickle: the state is the dict
    def __getstate__(self) -> Dict[str, Any]:
        return dict(self._attrs)

    def __setstate__(self, state: Dict[str, Any]) -> None:
        object.__setattr__(self, "_attrs", dict(state))

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
