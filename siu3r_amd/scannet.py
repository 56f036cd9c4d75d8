"""ScanNet validation-pair reader -- the evaluation branch of the reference's
`ScanNetDataset.__getitem__` (src/data/components/scannet_dataset.py:165-366, val branch 165-170) and of `collate_fn`
(src/data/datamodules/scannet_datamodule.py:13-86), without Lightning / the HF image processor:

    <root>/val_pair.json                       [{"scan": ..., "context_ids": [a, b], "target_ids": [...]}, ...]
    <root>/val/<scan>/color/<id>.jpg           already 256 x 256 (the network's fixed input size, scannet_dataset.py:66-73)
    <root>/val/<scan>/depth/<id>.png           uint16 millimetres
    <root>/val/<scan>/panoptic/<id>.png        RGB, segment id = R + 256 G + 65536 B = 1000 * semantic + instance
    <root>/val/<scan>/extrinsic/<id>.txt       camera-to-world 4x4
    <root>/val/<scan>/intrinsic.txt            4x4 (or 3x3) pixel intrinsics

What the items carry: images /255 (no other normalisation), intrinsics with fx, cx, fy, cy divided by 256
(`intrinsics_normalize`, :76-88), extrinsics relative to the FIRST context view (`relative_pose`, :90-115), and ground-truth panoptic
segments in the processor's (mask_labels, class_labels) form (image_processing_video_mask2former.py:270-310, 977-1052: `reduce_labels`
with ignore_index 255, instances unified over the views of a group and sorted, class = semantic - 1).  Host-side IO only."""
from __future__ import annotations

import json
import os
import os.path as osp
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch


def intrinsics_normalize(K: np.ndarray, size: float = 256.0) -> np.ndarray:
    """scannet_dataset.py:76-88"""
    return np.array([[K[0][0] / size, 0, K[0][2] / size], [0, K[1][1] / size, K[1][2] / size], [0, 0, 1]])


def relative_pose(context_ext: Sequence[np.ndarray], target_ext: Sequence[np.ndarray]) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """scannet_dataset.py:90-115: the first context view becomes the canonical frame."""
    inv = np.linalg.inv(context_ext[0])
    return [inv @ e for e in context_ext], [inv @ e for e in target_ext]


def decode_panoptic_png(rgb: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """[H,W,3] uint8 -> (semantic, instance) with segment id = R + 256 G + 65536 B = 1000 * semantic + instance (:248-262).
    (Like the reference, the arithmetic runs in the PNG's dtype promoted by numpy: the ids of ScanNet fit.)"""
    seg = rgb[:, :, 0].astype(np.int64) + rgb[:, :, 1].astype(np.int64) * 256 + rgb[:, :, 2].astype(np.int64) * 256 * 256
    return seg // 1000, seg % 1000


def instance_to_semantic(sem: np.ndarray, ins: np.ndarray) -> Dict[int, int]:
    """per view {instance id -> semantic id}, semantic labels visited in ascending order (:270-281): an instance id shared by two
    semantic labels keeps the larger label."""
    labels = np.unique(sem)
    if len(labels) == 1 and labels[0] == 0:
        raise ValueError("No semantic label in the view")
    out: Dict[int, int] = {}
    for s in labels:
        for i in np.unique(ins[sem == s]):
            out[int(i)] = int(s)
    return out


def panoptic_ground_truth(instance_maps: Sequence[np.ndarray], inst2sem: Sequence[Dict[int, int]]):
    """-> (mask_labels int64 [n_inst, N, H, W], class_labels int64 [n_inst]) as the reference's processor builds them with
    reduce_labels=True, ignore_index=255 (image_processing_video_mask2former.py:270-310, 977-1052): instance 0 is background, the
    per-view dictionaries are merged (later views win), the instances of the whole group are sorted, class = semantic - 1."""
    merged: Dict[int, int] = {}
    for d in inst2sem:
        merged.update(d)
    maps = [np.where(m == 0, 255, m - 1) for m in instance_maps]
    per_frame = [np.unique(m)[np.unique(m) != 255] for m in maps]
    video = np.unique(np.concatenate(per_frame)) if per_frame else np.zeros((0,), np.int64)
    H, W = maps[0].shape
    masks = torch.zeros((len(video), len(maps), H, W), dtype=torch.int64)
    for f, m in enumerate(maps):
        for k, inst in enumerate(video):
            masks[k, f] = torch.from_numpy((m == inst).astype(np.int64))
    classes = torch.tensor([merged[int(i) + 1] - 1 for i in video], dtype=torch.int64)
    return masks, classes


class ScanNetValPairs:
    """`ScanNetDataset(train=False)`: one item per entry of val_pair.json."""

    def __init__(self, root: str, val_pair_json: str = "val_pair.json", image_size: int = 256):
        self.root, self.image_size = root, image_size
        self.scans_dir = osp.join(root, "train" if "demo" in val_pair_json else "val")  # (:42-45)
        with open(osp.join(root, val_pair_json)) as fh:
            self.val_pairs = json.load(fh)

    def __len__(self) -> int:
        return len(self.val_pairs)

    def _views(self, scan_path: str, ids: Sequence[int]):
        from PIL import Image

        imgs = [np.array(Image.open(osp.join(scan_path, "color", f"{i}.jpg"))) for i in ids]
        depths = [np.array(Image.open(osp.join(scan_path, "depth", f"{i}.png"))) / 1000.0 for i in ids]
        ext = [np.loadtxt(osp.join(scan_path, "extrinsic", f"{i}.txt")) for i in ids]
        pan = [np.array(Image.open(osp.join(scan_path, "panoptic", f"{i}.png"))) for i in ids]
        return imgs, depths, ext, pan

    def __getitem__(self, idx: int) -> dict:
        """Views without any semantic label raise ValueError in the reference, which then moves on to the next pair (:360-366).  The
        item carries "pair_index": the pair actually read (!= idx for such a substitute).  A sharded driver must not count a
        substitute: the rank that owns that pair evaluates it anyway (`own_items`)."""
        for hop in range(len(self)):
            try:
                j = (idx + hop) % len(self)
                item = self._item(j)
                item["pair_index"] = j
                return item
            except ValueError:
                continue
        raise ValueError("no valid validation pair")

    def _item(self, idx: int) -> dict:
        pair = self.val_pairs[idx]
        scan = pair["scan"]
        scan_path = osp.join(self.scans_dir, scan)
        cids, tids = list(pair["context_ids"]), list(pair["target_ids"])
        c_img, c_dep, c_ext, c_pan = self._views(scan_path, cids)
        t_img, t_dep, t_ext, t_pan = self._views(scan_path, tids)
        K = np.loadtxt(osp.join(scan_path, "intrinsic.txt"))
        c_ext, t_ext = relative_pose(c_ext, t_ext)
        Kn = intrinsics_normalize(K, 256.0)
        out = {"scene_names": scan, "context_views_id": cids, "target_views_id": tids,
               "context_views_images": [np.transpose(i, (2, 0, 1)) for i in c_img], "target_views_images": [np.transpose(i, (2, 0, 1)) for i in t_img],
               "context_views_depths": c_dep, "target_views_depths": t_dep,
               "context_views_intrinsics": [Kn for _ in cids], "target_views_intrinsics": [Kn for _ in tids],
               "context_views_extrinsics": c_ext, "target_views_extrinsics": t_ext}
        for name, pans in (("context", c_pan), ("target", t_pan)):
            sem_ins = [decode_panoptic_png(p) for p in pans]
            maps = [np.asarray(i) for _, i in sem_ins]
            i2s = [instance_to_semantic(s, i) for s, i in sem_ins]
            out[f"{name}_mask_labels"], out[f"{name}_class_labels"] = panoptic_ground_truth(maps, i2s)
        return out


def collate(examples: Sequence[dict]) -> dict:
    """scannet_datamodule.py:13-86: images /255 as float32 [B,V,3,H,W]; depths, intrinsics, extrinsics float32; label lists as is."""
    ex = [e for e in examples if e is not None]
    if not ex:
        raise ValueError("No valid examples found in the batch")
    arr = lambda k: np.array([e[k] for e in ex])
    out = {"scene_names": [e["scene_names"] for e in ex], "context_views_id": [e["context_views_id"] for e in ex], "target_views_id": [e["target_views_id"] for e in ex]}
    for side in ("context", "target"):
        out[f"{side}_views_images"] = torch.tensor(arr(f"{side}_views_images")) / 255.0
        for k in ("depths", "intrinsics", "extrinsics"):
            out[f"{side}_views_{k}"] = torch.tensor(arr(f"{side}_views_{k}"), dtype=torch.float32)
        out[f"{side}_mask_labels"] = [e[f"{side}_mask_labels"] for e in ex]
        out[f"{side}_class_labels"] = [e[f"{side}_class_labels"] for e in ex]
    return out


def own_items(indices: Sequence[int], items: Sequence[dict], n_items: int, seen: set) -> List[dict]:
    """The items of `indices` a rank evaluates and counts: a substitute (pair_index != requested index) inside [0, n_items) is dropped --
    its owner rank reads the same pair natively, and the reference's evaluator counts a scene directory once however often it was
    written; one beyond n_items (a --limit run) is kept the first time this rank meets it."""
    out = []
    for i, it in zip(indices, items):
        j = it["pair_index"]
        if j != i and (j < n_items or j in seen):
            continue
        seen.add(j)
        out.append(it)
    return out


def shard(n_items: int, rank: int, world: int) -> List[int]:
    """pair i -> rank i mod world, no padding (Lightning's DistributedSampler pads the tail by repetition and the
    reference de-duplicates on disk, visualizer.py:340-341)."""
    return list(range(rank, n_items, world))
