"""world_size-2 gloo test (CPU) of the multi-GPU path: pair sharding + the single all-gather of metric statistics."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from siu3r_amd import distributed as D


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, _, w = D.init_from_env(backend="gloo")
    mine = D.shard_indices(n_items, r, w)
    stats = dict(n_pairs=len(mine), n_images=2 * len(mine), sum_psnr=float(sum(20.0 + i for i in mine)) * 2, label_checksum=float(sum(mine)))
    gathered = D.all_gather_stats(D.pack_stats(stats))
    tot = D.reduce_stats(gathered)
    t = D.max_over_ranks(float(rank + 1))
    D.barrier()
    q.put((rank, mine, tot, gathered.shape, t))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    n_items, world, port = 7, 2, 29617
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]          # i -> rank i mod world, no padding, nothing twice
    for _, _, tot, shape, t in res:
        assert tuple(shape) == (2, len(D.STAT_KEYS))
        assert tot["n_pairs"] == n_items and tot["n_images"] == 2 * n_items
        assert abs(tot["psnr"] - (20.0 + sum(range(n_items)) / n_items)) < 1e-12   # additive statistics == single-process result
        assert tot["label_checksum"] == sum(range(n_items)) and t == 2.0


def test_single_process_paths():
    assert D.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    g = D.all_gather_stats(D.pack_stats(dict(n_pairs=3, n_images=6, sum_psnr=60.0)))
    assert g.shape == (1, len(D.STAT_KEYS)) and D.reduce_stats(g)["psnr"] == 10.0
    assert D.max_over_ranks(1.5) == 1.5


def _metric_worker(rank, world, port, q):
    import numpy as np

    from siu3r_amd import metrics as M

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, _, w = D.init_from_env(backend="gloo")
    acc = M.MetricAccumulator()
    for i in D.shard_indices(5, r, w):       # scene i: a deterministic little prediction / ground-truth pair
        rng = np.random.default_rng(100 + i)
        gs = rng.integers(0, 8, (6, 10))
        gi = rng.integers(0, 3, (6, 10))
        ps = np.where(rng.random((6, 10)) < 0.3, rng.integers(0, 8, (6, 10)), gs)
        acc.add_segmentation("context", ps, gi, gs, gi)
        acc.add_render(rng.random((4, 4, 3)).astype(np.float32), rng.random((4, 4, 3)).astype(np.float32))
    gathered = D.all_gather_stats(torch.from_numpy(acc.to_vector()))      # the additive statistics: one fixed-length all-gather
    res = M.MetricAccumulator.from_vectors(gathered.numpy()).compute()
    # mean average precision is not additive: the scenes' match records travel in a second, variable-length gather
    recs = []
    for i in range(rank, 5, world):
        rng = np.random.default_rng(100 + i)
        gs = rng.integers(0, 8, (6, 10))
        gi = rng.integers(0, 3, (6, 10))
        ps = np.where(rng.random((6, 10)) < 0.3, rng.integers(0, 8, (6, 10)), gs)
        recs.append((i, M.map_scene_records(M.map_scene_inputs(ps, gi, gs, gi, None))))
    every = sorted((t for per_rank in D.all_gather_objects(recs) for t in per_rank), key=lambda t: t[0])
    res["context_map"] = M.mean_average_precision([r for _, r in every])
    q.put((rank, res))
    dist.destroy_process_group()


def test_metric_vector_gather_world2():
    """Sharded evaluation == single-process evaluation (the statistics are additive; reference: rank-0 file-based evaluator)."""
    import numpy as np

    from siu3r_amd import metrics as M

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_metric_worker, args=(r, 2, 29631, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    whole = M.MetricAccumulator()
    for i in range(5):
        rng = np.random.default_rng(100 + i)
        gs = rng.integers(0, 8, (6, 10))
        gi = rng.integers(0, 3, (6, 10))
        pr = np.where(rng.random((6, 10)) < 0.3, rng.integers(0, 8, (6, 10)), gs)
        whole.add_segmentation("context", pr, gi, gs, gi)
        whole.add_render(rng.random((4, 4, 3)).astype(np.float32), rng.random((4, 4, 3)).astype(np.float32))
        map_recs = locals().setdefault("map_recs", [])
        map_recs.append(M.map_scene_records(M.map_scene_inputs(pr, gi, gs, gi, None)))
    want = whole.compute()
    want["context_map"] = M.mean_average_precision(map_recs)
    for _, got in res:
        assert got["context_map"] == want["context_map"] and got["context_map"]["map"] >= 0
        assert set(got) == set(want)
        assert abs(got["psnr"] - want["psnr"]) < 1e-12 and abs(got["context_pq"] - want["context_pq"]) < 1e-12
        assert np.allclose(got["context_ious_per_class"], want["context_ious_per_class"], atol=1e-12)
