"""The reference's evaluation hand-off on disk: what `Visualizer` writes during validation
(src/visualizer.py:276-392 `visualize_recon_image`, :461-508 `save_seg_ids`, :510-556 `save_gt_seg_masks`) and what `Evaluator.evaluate`
reads back (src/evaluator.py:108-150, 240-404):

    <save_dir>/<scene>_context<a>_<b>/
        rgb/<scene>_<tid>.png, rgb_gt/...       uint8 RGB, `(img * 255).astype(uint8)` (truncation, visualizer.py:291,358)
        depth/<scene>_<tid>.png, depth_gt/...   int32 millimetres (`(depth * 1000).astype(int32)`, :309-310, mode "I")
        {context,target}_seg_pred/<scene>_pred<vid>.png + pred.json    segment id = 1000 * semantic + instance as R + 256 G + 65536 B
        {context,target}_seg_gt/<scene>_gt<vid>.png                    same encoding, instance = 1 + index in the sorted instance list
    <save_dir>/results.json                     keys psnr, ssim, lpips (when LPIPS weights are given), absrel, rmse, {context,target}_{pq,pqs_per_class,miou,ious_per_class}
                                                (evaluator.py:368-399); per scene render_scores.json / depth_scores.json

so that the reference's tooling can score this build's outputs and vice versa.  The numbers come from siu3r_amd.metrics (additive
statistics).  Pinned to the reference: depth errors and mIoU -- tests/golden/evaluator_tree.npz holds what the reference's own
Evaluator.evaluate returns for a synthetic tree (tests/golden/make_golden_eval.py).  PQ / PSNR / SSIM are torchmetrics classes in the
reference: restated, parity unpinned (see that module).  LPIPS / mAP are not computed.  Host-side IO (numpy + PIL); nothing here
runs on the GPU."""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import metrics as M


def _np(x) -> np.ndarray:
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def scene_dir(save_dir, scene_name: str, context_view_id: Sequence[int]) -> Path:
    return Path(save_dir) / f"{scene_name}_context{'_'.join(map(str, context_view_id))}"


def encode_segment_ids(sem: np.ndarray, ins: np.ndarray) -> np.ndarray:
    """[H,W] int -> [H,W,3] uint8 with R = id % 256, G = id // 256, B = id // 65536 (each truncated to 8 bits by the uint8 store,
    visualizer.py:495-500)."""
    sid = 1000 * np.asarray(sem, np.int64) + np.asarray(ins, np.int64)
    return np.stack((sid % 256, sid // 256, sid // 256 // 256), -1).astype(np.uint8)


def save_recon_images(rendered_images, rendered_depths, gt_images, gt_depths, save_dir, scene_names, context_views_id, target_views_id):
    """rendered / gt images [B,N,3,H,W] in [0,1], depths [B,N,H,W] (metres) -> rgb / rgb_gt / depth / depth_gt PNGs.  Like the
    reference, a scene whose `rgb` directory already exists is skipped (visualizer.py:340-341: its de-duplication of the padded tail
    of a distributed sampler)."""
    from PIL import Image

    R, Dp, Gi, Gd = _np(rendered_images), _np(rendered_depths), _np(gt_images), _np(gt_depths)
    for i, scene in enumerate(scene_names):
        sub = scene_dir(save_dir, scene, context_views_id[i])
        os.makedirs(sub, exist_ok=True)
        if (sub / "rgb").exists():
            continue
        for d in ("rgb", "rgb_gt", "depth", "depth_gt"):
            os.makedirs(sub / d, exist_ok=True)
        for j, tid in enumerate(target_views_id[i]):
            name = f"{scene}_{tid}.png"
            Image.fromarray((np.transpose(R[i, j], (1, 2, 0)) * 255).astype(np.uint8)).save(sub / "rgb" / name)
            Image.fromarray((np.transpose(Gi[i, j], (1, 2, 0)) * 255).astype(np.uint8)).save(sub / "rgb_gt" / name)
            Image.fromarray((Dp[i, j] * 1000).astype(np.int32)).save(sub / "depth" / name)
            Image.fromarray((Gd[i, j] * 1000).astype(np.int32)).save(sub / "depth_gt" / name)


def save_seg_ids(mode: str, semantic_ids, instance_ids, save_dir, scene_names, context_views_id, target_views_id=None, seg_infos=None):
    """semantic / instance ids [B,N,H,W] -> <mode>_seg_pred/<scene>_pred<vid>.png + pred.json (visualizer.py:461-508)."""
    from PIL import Image

    if mode not in ("context", "target"):
        raise ValueError(f"Unknown mode: {mode}")
    S, I = _np(semantic_ids), _np(instance_ids)
    for idx, scene in enumerate(scene_names):
        base = scene_dir(save_dir, scene, context_views_id[idx])
        os.makedirs(base / f"{mode}_seg_pred", exist_ok=True)
        with open(base / f"{mode}_seg_pred" / "pred.json", "w") as fh:
            json.dump(list(seg_infos[idx]) if seg_infos is not None else [], fh, indent=4)
        views = context_views_id[idx] if mode == "context" else target_views_id[idx]
        for sem, ins, vid in zip(S[idx], I[idx], views):
            Image.fromarray(encode_segment_ids(sem, ins)).save(base / f"{mode}_seg_pred" / f"{scene}_pred{vid}.png")


def gt_ids_from_mask_labels(mask_label, class_label):
    """mask_label [n_inst, N, H, W] (0/1), class_label [n_inst] (0-based) -> (semantic [N,H,W] = class + 1, instance = index + 1),
    later instances overwriting earlier ones (visualizer.py:533-537)."""
    m, c = _np(mask_label), _np(class_label)
    sem = np.zeros(m.shape[1:], np.int64)
    ins = np.zeros(m.shape[1:], np.int64)
    for k in range(m.shape[0]):
        on = m[k] == 1
        ins[on] = k + 1
        sem[on] = int(c[k]) + 1
    return sem, ins


def save_gt_seg_masks(mode: str, mask_labels, class_labels, save_dir, scene_names, context_views_id, target_views_id=None):
    """ground-truth ids -> <mode>_seg_gt/<scene>_gt<vid>.png (visualizer.py:510-556)."""
    from PIL import Image

    if mode not in ("context", "target"):
        raise ValueError(f"Unknown mode: {mode}")
    for idx, scene in enumerate(scene_names):
        base = scene_dir(save_dir, scene, context_views_id[idx])
        os.makedirs(base / f"{mode}_seg_gt", exist_ok=True)
        sem, ins = gt_ids_from_mask_labels(mask_labels[idx], class_labels[idx])
        views = context_views_id[idx] if mode == "context" else target_views_id[idx]
        for s_, i_, vid in zip(sem, ins, views):
            Image.fromarray(encode_segment_ids(s_, i_)).save(base / f"{mode}_seg_gt" / f"{scene}_gt{vid}.png")


# ---- reading back (evaluator.py:108-150) ------------------------------------------------------------------------------
def load_image01(path) -> np.ndarray:
    from PIL import Image

    return np.array(Image.open(path)).astype(np.float32) / 255.0


def load_segmentation_dir(pred_dir: Path, gt_dir: Path):
    """all views of a scene concatenated along H (evaluator.py:146-150): (pred_sem, pred_ins, gt_sem, gt_ins) int64 [N*H, W]"""
    from PIL import Image

    ps, pi, gs, gi = [], [], [], []
    for item in sorted(Path(pred_dir).glob("*.png")):
        s, i = M.decode_segment_ids(np.array(Image.open(item)))
        ps.append(s)
        pi.append(i)
        s, i = M.decode_segment_ids(np.array(Image.open(Path(gt_dir) / item.name.replace("pred", "gt"))))
        gs.append(s)
        gi.append(i)
    cat = lambda l: np.concatenate(l, 0)
    return cat(ps), cat(pi), cat(gs), cat(gi)


def accumulate_dir(path, acc: Optional[M.MetricAccumulator] = None, scenes: Optional[Sequence[str]] = None, write_scene_scores: bool = True,
                   map_records: Optional[Dict[str, list]] = None, lpips=None) -> M.MetricAccumulator:
    """Fold the scene directories under `path` (all, or the named subset: a rank's shard) into additive statistics.  map_records: a dict
    that receives, per mode ("context" / "target"), one (scene directory name, COCO match record) pair per scene for the mean average precision
    (metrics.map_scene_records; not additive: the caller gathers the lists and hands them to `ordered_map_records` ->
    metrics.mean_average_precision on rank 0 -- the accumulation breaks score ties by input order, so the order must not depend on the sharding).  lpips: a siu3r_amd.lpips.LPIPS (GPU);
    when given, every rendered image also gets its `lpips` score (evaluator.py:263)."""
    from PIL import Image

    acc = acc or M.MetricAccumulator()
    root = Path(path)
    dirs = sorted(d for d in root.iterdir() if d.is_dir() and "_context" in d.name)
    if scenes is not None:
        keep = set(scenes)
        dirs = [d for d in dirs if d.name in keep]
    for d in dirs:
        if (d / "rgb").is_dir():
            scores = []
            for item in sorted((d / "rgb").glob("*.png")):
                pred, gt = load_image01(item), load_image01(d / "rgb_gt" / item.name)
                sc = {"item": item.name, **acc.add_render_u8(pred, gt)}  # the files already hold the truncated uint8 values
                if lpips is not None and min(pred.shape[:2]) >= 16:
                    sc.update(acc.add_lpips(float(lpips(pred, gt)[0])))
                scores.append(sc)
            if write_scene_scores:
                with open(d / "render_scores.json", "w") as fh:
                    json.dump(scores, fh, indent=4)
        if (d / "depth").is_dir() and (d / "depth_gt").is_dir():
            scores = []
            for item in sorted((d / "depth").glob("*.png")):  # int32 millimetres -> metres (evaluator.py:340-345)
                dm, gm = (np.array(Image.open(f)).astype(np.float32) / 1000.0 for f in (item, d / "depth_gt" / item.name))
                scores.append({"item": item.name, **acc.add_depth(dm, gm)})
            if write_scene_scores:
                with open(d / "depth_scores.json", "w") as fh:
                    json.dump(scores, fh, indent=4)
        for mode in ("context", "target"):
            if (d / f"{mode}_seg_pred").is_dir() and (d / f"{mode}_seg_gt").is_dir() and any((d / f"{mode}_seg_pred").glob("*.png")):
                maps = load_segmentation_dir(d / f"{mode}_seg_pred", d / f"{mode}_seg_gt")
                acc.add_segmentation(mode, *maps)
                if map_records is not None:
                    pj = d / f"{mode}_seg_pred" / "pred.json"  # (evaluator.py:175-180: label and score of a predicted id come from here)
                    preds = json.load(open(pj)) if pj.exists() else None
                    map_records.setdefault(mode, []).append((d.name, M.map_scene_records(M.map_scene_inputs(*maps, preds))))
    return acc


def ordered_map_records(named: Sequence) -> list:
    """(scene name, record) pairs from any number of ranks -> the records in the order a single process walks the result tree (sorted
    scene directories, evaluator.py:346-399).  mean_average_precision sorts detection scores stably, so ties -- every score is 1.0 when
    pred.json is absent -- resolve by this order: without it the value would change with the world size."""
    return [r for _, r in sorted(named, key=lambda t: t[0])]


def evaluate_dir(path, write: bool = True, lpips=None) -> Dict[str, object]:
    """single-process counterpart of Evaluator.evaluate (evaluator.py:240-404): results.json with the BASELINE metric's keys."""
    recs: Dict[str, list] = {}
    res = accumulate_dir(path, map_records=recs, lpips=lpips).compute()
    for mode, r in recs.items():
        res[f"{mode}_map"] = M.mean_average_precision(ordered_map_records(r))
    if write:
        with open(Path(path) / "results.json", "w") as fh:
            json.dump(res, fh, indent=4)
    return res
