"""Evaluator-compatible metrics as ADDITIVE statistics (reference src/evaluator.py:28-404, src/utils/miou.py:34-77).

The reference evaluates on rank 0 from PNG files: PSNR per image (torchmetrics PeakSignalNoiseRatio on PNG-round-tripped
renders, evaluator.py:257-268, visualizer.py:291/358: `(img * 255).astype(uint8)` = truncation), panoptic quality from
`(semantic, instance)` maps decoded from `1000 * sem + ins` (evaluator.py:126-150, 282-329) with things = [3..20],
stuffs = [1, 2] (1-based, :32-37), mean IoU from an intersection/union histogram.  Everything the BASELINE metric needs is
additive over images, so here each rank accumulates a fixed-length fp64 vector and ONE all-gather (siu3r_amd/distributed.py)
reproduces the single-process result.  torchmetrics is not in this image: PSNR / PQ follow the published definitions
(PQ: Kirillov et al. 2019, as implemented by torchmetrics.detection.PanopticQuality with allow_unknown_preds_category=True)
and are PARITY-UNPINNED against the library; the tests check them against hand-computed cases.  Host-side code (numpy)."""
from __future__ import annotations

from typing import Dict, Iterable, Sequence

import numpy as np

NUM_CLASSES = 21                      # 0 = void/unlabelled, 1..20 = ScanNet-20 (1-based evaluator categories)
THINGS = tuple(range(3, 21))          # evaluator.py:32-34 (cfg.things + 1)
STUFFS = (1, 2)                       # evaluator.py:35-37 (wall, floor)


def png_roundtrip(img01: np.ndarray) -> np.ndarray:
    """float image in [0,1] -> what the evaluator reads back from the PNG the visualizer wrote (truncation, not rounding)."""
    return (np.clip(img01, 0.0, 1.0) * 255).astype(np.uint8).astype(np.float32) / 255.0


def psnr(pred: np.ndarray, target: np.ndarray, data_range: float | None = None) -> float:
    """10 log10(data_range^2 / MSE); data_range defaults to target.max() - target.min() (torchmetrics' data_range=None)."""
    pred, target = np.asarray(pred, np.float64), np.asarray(target, np.float64)
    dr = float(target.max() - target.min()) if data_range is None else float(data_range)
    mse = float(np.mean((pred - target) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(dr * dr / mse)


def decode_segment_ids(rgb_u8: np.ndarray):
    """RGB PNG -> (semantic, instance) via segment_id = R + 256 G + 65536 B = 1000 * sem + ins (evaluator.py:126-144)."""
    sid = rgb_u8[..., 0].astype(np.int64) + rgb_u8[..., 1].astype(np.int64) * 256 + rgb_u8[..., 2].astype(np.int64) * 65536
    return sid // 1000, sid % 1000


def _segments(sem: np.ndarray, ins: np.ndarray, things, stuffs, void_unknown: bool):
    """(category, instance) colour per pixel: stuff instances are merged (instance 0); categories outside things+stuffs are
    void when void_unknown (allow_unknown_preds_category) else an error."""
    cat = sem.astype(np.int64).copy()
    known = np.isin(cat, np.asarray(tuple(things) + tuple(stuffs)))
    if not void_unknown and not known.all():
        raise ValueError("unknown categories in the target")
    cat[~known] = 0
    inst = np.where(np.isin(cat, np.asarray(tuple(stuffs))) | (cat == 0), 0, ins.astype(np.int64))
    return cat * 100000 + inst, cat


def panoptic_stats(pred_sem, pred_ins, gt_sem, gt_ins, things: Sequence[int] = THINGS, stuffs: Sequence[int] = STUFFS,
                   num_classes: int = NUM_CLASSES) -> np.ndarray:
    """One update of PanopticQuality: returns [num_classes, 4] = (sum IoU of TPs, TP, FP, FN) per category.
    A predicted and a ground-truth segment of the same category match iff IoU > 0.5 (unique by construction), where the union
    excludes the part of the prediction that lies on void ground truth; unmatched predictions mostly (> 50 %) on void are not FPs."""
    p_col, p_cat = _segments(np.asarray(pred_sem), np.asarray(pred_ins), things, stuffs, True)
    g_col, g_cat = _segments(np.asarray(gt_sem), np.asarray(gt_ins), things, stuffs, True)
    p_col, g_col = p_col.reshape(-1), g_col.reshape(-1)
    p_ids, p_area = np.unique(p_col, return_counts=True)
    g_ids, g_area = np.unique(g_col, return_counts=True)
    p_area, g_area = dict(zip(p_ids.tolist(), p_area.tolist())), dict(zip(g_ids.tolist(), g_area.tolist()))
    pairs, inter = np.unique(np.stack((p_col, g_col)), axis=1, return_counts=True)
    inter = {(int(a), int(b)): int(n) for (a, b), n in zip(pairs.T, inter)}
    out = np.zeros((num_classes, 4), np.float64)
    matched_p, matched_g = set(), set()
    for (pc, gc), n in inter.items():
        if pc // 100000 == 0 or gc // 100000 == 0 or pc // 100000 != gc // 100000:
            continue
        union = p_area[pc] + g_area[gc] - n - inter.get((pc, 0), 0)
        iou = n / union
        if iou > 0.5:
            c = pc // 100000
            out[c, 0] += iou
            out[c, 1] += 1
            matched_p.add(pc)
            matched_g.add(gc)
    for gc in g_area:
        if gc // 100000 != 0 and gc not in matched_g:
            out[gc // 100000, 3] += 1
    for pc, a in p_area.items():
        if pc // 100000 != 0 and pc not in matched_p and inter.get((pc, 0), 0) / a <= 0.5:
            out[pc // 100000, 2] += 1
    return out


def pq_from_stats(stats: np.ndarray, classes: Iterable[int] = THINGS + STUFFS) -> Dict[str, object]:
    """per-class PQ = sum IoU / (TP + FP/2 + FN/2) (0 where the class never occurs), mean over `classes` (evaluator.py:384-387)."""
    per = []
    for c in sorted(classes):
        s, tp, fp, fn = stats[c]
        d = tp + 0.5 * fp + 0.5 * fn
        per.append(float(s / d) if d > 0 else 0.0)
    return dict(per_class=per, pq=float(np.mean(per)))


def miou_stats(pred_sem, gt_sem, num_classes: int = NUM_CLASSES) -> np.ndarray:
    """[num_classes, 2] = (intersection, union) pixel counts per class (utils/miou.py:12-30, input_format='index')."""
    p, g = np.asarray(pred_sem).reshape(-1), np.asarray(gt_sem).reshape(-1)
    out = np.zeros((num_classes, 2), np.float64)
    for c in range(num_classes):
        pc, gc = p == c, g == c
        out[c, 0] = np.count_nonzero(pc & gc)
        out[c, 1] = np.count_nonzero(pc | gc)
    return out


class MetricAccumulator:
    """Additive statistics of one rank.  Vector layout (fp64): [n_images, sum_psnr] + context PQ [C,4] + target PQ [C,4] +
    context mIoU [C,2] + target mIoU [C,2]  ->  2 + 12 C doubles (254 for C = 21: 2 KB per rank in the all-gather)."""

    def __init__(self, num_classes: int = NUM_CLASSES, things=THINGS, stuffs=STUFFS):
        self.C, self.things, self.stuffs = num_classes, tuple(things), tuple(stuffs)
        self.n_images, self.sum_psnr = 0.0, 0.0
        self.pq = {k: np.zeros((num_classes, 4)) for k in ("context", "target")}
        self.iou = {k: np.zeros((num_classes, 2)) for k in ("context", "target")}

    def add_render(self, pred01: np.ndarray, gt01: np.ndarray):
        self.sum_psnr += psnr(png_roundtrip(pred01), png_roundtrip(gt01))
        self.n_images += 1

    def add_segmentation(self, which: str, pred_sem, pred_ins, gt_sem, gt_ins):
        """one scene: all its views concatenated along H, as the evaluator does (evaluator.py:146-150)."""
        self.pq[which] += panoptic_stats(pred_sem, pred_ins, gt_sem, gt_ins, self.things, self.stuffs, self.C)
        self.iou[which] += miou_stats(pred_sem, gt_sem, self.C)

    def to_vector(self) -> np.ndarray:
        return np.concatenate(([self.n_images, self.sum_psnr], self.pq["context"].ravel(), self.pq["target"].ravel(),
                               self.iou["context"].ravel(), self.iou["target"].ravel())).astype(np.float64)

    @classmethod
    def from_vectors(cls, vecs: np.ndarray, num_classes: int = NUM_CLASSES, things=THINGS, stuffs=STUFFS) -> "MetricAccumulator":
        v = np.asarray(vecs, np.float64).reshape(-1, 2 + 12 * num_classes).sum(0)
        m = cls(num_classes, things, stuffs)
        m.n_images, m.sum_psnr = v[0], v[1]
        o, C = 2, num_classes
        m.pq["context"] = v[o:o + 4 * C].reshape(C, 4); o += 4 * C
        m.pq["target"] = v[o:o + 4 * C].reshape(C, 4); o += 4 * C
        m.iou["context"] = v[o:o + 2 * C].reshape(C, 2); o += 2 * C
        m.iou["target"] = v[o:o + 2 * C].reshape(C, 2)
        return m

    def compute(self) -> Dict[str, object]:
        """the keys of the reference's results.json (evaluator.py:370-399) that the BASELINE metric uses."""
        res: Dict[str, object] = {}
        if self.n_images:
            res["psnr"] = self.sum_psnr / self.n_images
        for k in ("context", "target"):
            if self.pq[k].sum() > 0:
                r = pq_from_stats(self.pq[k], self.things + self.stuffs)
                res[f"{k}_pqs_per_class"], res[f"{k}_pq"] = r["per_class"], r["pq"]
            if self.iou[k][:, 1].sum() > 0:
                inter, union = self.iou[k][:, 0], self.iou[k][:, 1]
                iou = np.where(union > 0, inter / np.maximum(union, 1), 0.0)
                res[f"{k}_ious_per_class"], res[f"{k}_miou"] = iou.tolist(), float(np.mean(iou))
        return res
