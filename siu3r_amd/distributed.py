"""Multi-GPU data parallelism for the inference path: independent image pairs are sharded across ranks
(one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).
There is no data-path collective: the only exchange is ONE all-gather of a fixed-length fp64 vector of
additive metric statistics per rank; the reference's equivalent is two barriers around a
rank-0 file-based evaluation (reference src/pipeline.py:315-326).
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise the process group from RANK/WORLD_SIZE/MASTER_* (torch.distributed.run sets them)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SIU3R_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Pair i goes to rank i mod world (what DistributedSampler does for val/test), without tail padding:
    ranks may differ by one item and nothing is evaluated twice."""
    return list(range(rank, n_items, world))


STAT_KEYS = ("n_pairs", "n_images", "sum_psnr", "sum_sq_err", "n_pixels", "n_gaussians", "n_segments", "label_checksum")


def pack_stats(stats: Dict[str, float]) -> torch.Tensor:
    return torch.tensor([float(stats.get(k, 0.0)) for k in STAT_KEYS], dtype=torch.float64)


def _collective_device(device):
    """tensors of a collective live on the GPU for RCCL ("nccl") and on the host for gloo (CPU tests, single-GPU dry runs)"""
    return device if (device is not None and dist.get_backend() == "nccl") else None


def all_gather_stats(vec: torch.Tensor, device=None) -> torch.Tensor:
    """The single collective of the path: [world, len(STAT_KEYS)] fp64 on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return vec[None].clone()
    world = dist.get_world_size()  # (a group of one still runs the collective: the single-GPU RCCL test relies on it)
    device = _collective_device(device)
    v = vec.to(device) if device is not None else vec.cpu()
    out = [torch.empty_like(v) for _ in range(world)]
    dist.all_gather(out, v)
    return torch.stack(out).cpu()


def reduce_stats(gathered: torch.Tensor) -> Dict[str, float]:
    tot = gathered.sum(0)
    res = {k: float(tot[i]) for i, k in enumerate(STAT_KEYS)}
    res["psnr"] = res["sum_psnr"] / res["n_images"] if res["n_images"] else float("nan")
    return res


def max_over_ranks(x: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return x
    t = torch.tensor([x], dtype=torch.float64, device=_collective_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def all_gather_objects(obj) -> list:
    """Variable-length companion of the statistics gather, for the one metric that is not additive (mean average precision: per-scene
    match records, a few KB per scene): every rank's object, in rank order, on every rank.  Pickled by torch.distributed
    (`all_gather_object`); a single process returns [obj]."""
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out

