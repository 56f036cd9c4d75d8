#!/usr/bin/env python
"""Evaluation driver of the MI355X path: one process per GPU over ScanNet validation pairs (BASELINE.json configs[2] / [3]).

Counterpart of the reference's validation loop (src/pipeline.py:286-326: `validation_step` -> `Visualizer` files -> rank-0
`Evaluator.evaluate`): per batch of pairs
    SIU3RModel.forward(context views, lift)  ->  SplattingCUDA.forward(target poses: colour, depth, query x class logit maps)
    ->  lifting (pipeline.py:132-193)  ->  files in the reference's on-disk layout (siu3r_amd/eval_io.py)
and at the end ONE all-gather of every rank's additive metric vector (siu3r_amd/metrics.py) instead of the reference's
two barriers around a rank-0 pass over a shared filesystem; rank 0 writes results.json.  Pair i goes to rank i mod world.

    python evaluate.py --data_root /data/scannet --model_path siu3r_epoch100.ckpt --output_path outputs/val
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 evaluate.py --data_root ... --batch 8
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def context_positions(context_ids, target_ids):
    """positions of the context views inside each target list (pipeline.py:101-108)"""
    return [[pos for pos, idx in enumerate(t) if idx in c] for c, t in zip(context_ids, target_ids)]


def run_batch(model, renderer, batch, image_size, dev):
    """pipeline.py:76-215 `step_w_query_class_logit_lift`"""
    from siu3r_amd.gaussian_renderer import lift_query_class_logits

    imgs = batch["context_views_images"].to(dev)
    Kc = batch["context_views_intrinsics"][:, :, :3, :3].to(dev)
    Kt = batch["target_views_intrinsics"][:, :, :3, :3]
    ext_t = batch["target_views_extrinsics"]
    with torch.no_grad():
        gaussians, seg_out, seg_masks, seg_infos, q_scores = model(imgs, Kc, enable_query_class_logit_lift=True)
        rend = renderer.forward(gaussians, ext_t, Kt, image_size, render_color=True, render_qc_logits=True)
        sem, ins, infos = lift_query_class_logits(rend["render_qc_logits"], q_scores, num_queries=model.mask2former.num_queries,
                                                  label_ids_to_fuse=sorted(model.label_ids_to_fuse))
    pos = context_positions(batch["context_views_id"], batch["target_views_id"])
    csem = torch.stack([sem[i, p] for i, p in enumerate(pos)])
    cins = torch.stack([ins[i, p] for i, p in enumerate(pos)])
    return dict(render=rend, target_sem=sem, target_ins=ins, context_sem=csem, context_ins=cins, seg_infos=infos, gaussians=gaussians)


def write_batch(out_dir, batch, res):
    """what Visualizer.write_file stores for the evaluator (visualizer.py:136-270)"""
    from siu3r_amd import eval_io as E

    names, cids, tids = batch["scene_names"], batch["context_views_id"], batch["target_views_id"]
    E.save_recon_images(res["render"]["render_color"], res["render"]["render_depth"], batch["target_views_images"], batch["target_views_depths"],
                        out_dir, names, cids, tids)
    E.save_seg_ids("context", res["context_sem"], res["context_ins"], out_dir, names, cids, tids, res["seg_infos"])
    E.save_seg_ids("target", res["target_sem"], res["target_ins"], out_dir, names, cids, tids, res["seg_infos"])
    E.save_gt_seg_masks("context", batch["context_mask_labels"], batch["context_class_labels"], out_dir, names, cids, tids)
    E.save_gt_seg_masks("target", batch["target_mask_labels"], batch["target_class_labels"], out_dir, names, cids, tids)
    return [E.scene_dir(out_dir, n, c).name for n, c in zip(names, cids)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_root", required=True, help="ScanNet root holding val_pair.json and val/<scan>/{color,depth,panoptic,extrinsic}")
    ap.add_argument("--val_pair_json", default="val_pair.json")
    ap.add_argument("--model_path", default=None, help="Lightning .ckpt / state dict; default: seeded synthetic weights (plumbing only)")
    ap.add_argument("--output_path", default="outputs/val")
    ap.add_argument("--batch", type=int, default=8, help="pairs per forward (configs[2]: 8)")
    ap.add_argument("--limit", type=int, default=0, help="evaluate only the first N pairs")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16", "bf16x3"])
    ap.add_argument("--lpips_weights", default=None, help="file holding the LPIPS (VGG16 + lin) tensors; default: the `lpips.*` keys of --model_path when it "
                    "is a Pipeline checkpoint (src/pipeline.py:35).  Without either, results.json has no `lpips` key")
    a = ap.parse_args()

    from siu3r_amd import distributed as D, eval_io as E, metrics as M, scannet
    from siu3r_amd.cli_common import load_weights
    from siu3r_amd.gaussian_renderer import SplattingCUDA
    from siu3r_amd.model import SIU3RModel

    rank, local, world = D.init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("evaluate.py needs an MI355X: the HIP path has no CPU fallback")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    data = scannet.ScanNetValPairs(a.data_root, a.val_pair_json)
    n = min(len(data), a.limit) if a.limit else len(data)
    mine = scannet.shard(n, rank, world)
    size = (data.image_size, data.image_size)
    raw = None
    if a.model_path and Path(a.model_path).exists():
        from siu3r_amd.checkpoint import read_checkpoint_file

        raw = read_checkpoint_file(a.model_path)  # read ONCE: the model's tensors and (below) the `lpips.*` keys come from the same object
    model = SIU3RModel(load_weights(a.model_path, ckpt=raw), image_size=size, precision=a.precision, device=dev)
    renderer = SplattingCUDA()
    lp = None
    lp_file = a.lpips_weights or a.model_path
    if lp_file:
        from siu3r_amd.checkpoint import load_lpips_weights
        from siu3r_amd.lpips import LPIPS

        try:
            lw = load_lpips_weights(lp_file, ckpt=raw if lp_file == a.model_path else None)
        except RuntimeError as e:   # an incomplete network: an error when the file was named for it, a note when it is just the model file
            if a.lpips_weights:
                raise
            print(f"evaluate.py: {e}; results.json will have no `lpips` key", file=sys.stderr)
            lw = None
        if lw is None and a.lpips_weights:
            raise RuntimeError(f"{a.lpips_weights}: no LPIPS network found")
        lp = LPIPS(lw, device=dev) if lw is not None else None
    del raw
    out_dir = Path(a.output_path)
    out_dir.mkdir(parents=True, exist_ok=True)
    acc, my_scenes, t0 = M.MetricAccumulator(), [], time.perf_counter()
    seen = set()
    for s in range(0, len(mine), a.batch):
        idx = mine[s:s + a.batch]
        items = scannet.own_items(idx, [data[i] for i in idx], n, seen)  # substitutes of unlabeled pairs belong to their owner rank
        if not items:
            continue
        batch = scannet.collate(items)
        res = run_batch(model, renderer, batch, size, dev)
        my_scenes += write_batch(out_dir, batch, res)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    map_recs = {}
    E.accumulate_dir(out_dir, acc, scenes=my_scenes, map_records=map_recs, lpips=lp)  # this rank's shard, read back from the files it wrote (PNG truncation included)
    gathered = D.all_gather_stats(torch.from_numpy(acc.to_vector()), device=dev)  # the additive statistics: ONE fixed-length all-gather
    all_recs = D.all_gather_objects(map_recs)  # mean average precision is not additive: per-scene match records (a few KB each)
    if rank == 0:
        result = M.MetricAccumulator.from_vectors(gathered.numpy()).compute()
        for mode in ("context", "target"):
            recs = E.ordered_map_records([t for per_rank in all_recs for t in per_rank.get(mode, [])])  # scene order, not rank order: ties
            if recs:
                result[f"{mode}_map"] = M.mean_average_precision(recs)  # (evaluator.py:388-399: the whole torchmetrics result dict)
        with open(out_dir / "results.json", "w") as fh:
            json.dump(result, fh, indent=4)
        print(json.dumps({"pairs": n, "world": world, "pairs_per_s_rank0": len(mine) / max(elapsed, 1e-9),
                          **{k: (v["map"] if k.endswith("_map") else v) for k, v in result.items() if not k.endswith("per_class")}}))
    D.barrier()


if __name__ == "__main__":
    main()
