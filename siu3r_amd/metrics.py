"""Evaluator-compatible metrics as ADDITIVE statistics (reference src/evaluator.py:28-404, src/utils/miou.py:34-77).

The reference evaluates on rank 0 from PNG files: PSNR per image (torchmetrics PeakSignalNoiseRatio on PNG-round-tripped
renders, evaluator.py:257-268, visualizer.py:291/358: `(img * 255).astype(uint8)` = truncation), panoptic quality from
`(semantic, instance)` maps decoded from `1000 * sem + ins` (evaluator.py:126-150, 282-329) with things = [3..20],
stuffs = [1, 2] (1-based, :32-37), mean IoU from an intersection/union histogram.  Everything the BASELINE metric needs is
additive over images, so here each rank accumulates a fixed-length fp64 vector and ONE all-gather (siu3r_amd/distributed.py)
reproduces the single-process result.  Image quality adds SSIM (torchmetrics StructuralSimilarityIndexMeasure defaults, :231-233)
and depth quality adds AbsRel / RMSE after a least-squares scale + shift fit over the valid ground-truth pixels (:229-236,
:346-366); all three are per-image values the evaluator averages, i.e. additive too.  `context_map` / `target_map` (torchmetrics
MeanAveragePrecision(iou_type="segm", class_metrics=True), evaluator.py:93-106, 152-226, 388-399) are NOT additive -- COCO matching sorts
the detections of the whole set by score -- so they travel as small per-scene match records (`map_scene_records`: what COCOeval.evaluateImg
keeps: scores, matched / ignored flags per IoU threshold, ground-truth ignore flags) in a second, variable-length gather and are
accumulated on rank 0 (`mean_average_precision`).  LPIPS: siu3r_amd/lpips.py (GPU; the weights come from the checkpoint, none are shipped), accumulated here as n_lpips / sum_lpips.
torchmetrics (pinned 1.7.3, uv.lock:3373) is not in this image: PSNR / SSIM / PQ restate its algorithms -- PQ: Kirillov et al. 2019
as torchmetrics.detection.PanopticQuality(allow_unknown_preds_category=True, return_per_class=True) implements it, per-class order
= sorted things then sorted stuffs -- and are PARITY-UNPINNED against the library; the tests check them against hand-computed
cases (void / unknown categories, absent classes, mostly-void segments); mAP restates pycocotools' COCOeval (evaluateImg / accumulate /
summarize, the backend torchmetrics drives) for masks without crowds, equally unpinned, with hand-computed cases.  Host-side code (numpy)."""
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


def _gauss1d(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    d = np.arange((1 - size) / 2.0, (1 + size) / 2.0, 1.0)
    g = np.exp(-((d / sigma) ** 2) / 2.0)
    return g / g.sum()


def ssim(pred: np.ndarray, target: np.ndarray, data_range: float | None = None, kernel_size: int = 11, sigma: float = 1.5,
         k1: float = 0.01, k2: float = 0.03) -> float:
    """StructuralSimilarityIndexMeasure() of torchmetrics on ONE image [H, W, C] (or [H, W]): separable 11-tap Gaussian window
    (sigma 1.5) per channel over the reflect-padded image, the local variances clamped at 0, the SSIM map cropped by the pad again
    and averaged; data_range None = max(pred.max - pred.min, target.max - target.min)."""
    p, t = np.asarray(pred, np.float64), np.asarray(target, np.float64)
    if p.ndim == 2:
        p, t = p[..., None], t[..., None]
    dr = max(float(p.max() - p.min()), float(t.max() - t.min())) if data_range is None else float(data_range)
    c1, c2 = (k1 * dr) ** 2, (k2 * dr) ** 2
    pad = (kernel_size - 1) // 2
    if min(p.shape[0], p.shape[1]) <= 2 * pad:  # reflect padding needs pad < size, the crop a non-empty interior (torchmetrics raises)
        raise ValueError(f"ssim: image {p.shape[:2]} too small for an {kernel_size}-tap window")
    g = _gauss1d(kernel_size, sigma)

    def blur(x):  # [H + 2 pad, W + 2 pad, C] -> valid separable convolution [H, W, C]
        H = x.shape[0] - 2 * pad
        y = sum(g[i] * x[i:i + H] for i in range(kernel_size))
        W = x.shape[1] - 2 * pad
        return sum(g[i] * y[:, i:i + W] for i in range(kernel_size))

    rp = lambda x: np.pad(x, ((pad, pad), (pad, pad), (0, 0)), mode="reflect")
    pp, tp = rp(p), rp(t)
    mu_p, mu_t = blur(pp), blur(tp)
    s_pp = np.maximum(blur(pp * pp) - mu_p * mu_p, 0.0)
    s_tt = np.maximum(blur(tp * tp) - mu_t * mu_t, 0.0)
    s_pt = blur(pp * tp) - mu_p * mu_t
    full = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    return float(full[pad:-pad, pad:-pad].mean())


def fit_scale_and_shift(pred: np.ndarray, gt: np.ndarray):
    """least-squares (scale, shift) with scale * pred + shift ~ gt over gt > 0 (evaluator.py:229-236)"""
    m = gt > 0
    x, y = np.asarray(pred, np.float64)[m], np.asarray(gt, np.float64)[m]
    A = np.stack((x, np.ones_like(x)), 1)
    sol, *_ = np.linalg.lstsq(A, y, rcond=None)
    return float(sol[0]), float(sol[1])


def depth_errors(depth: np.ndarray, depth_gt: np.ndarray):
    """(absrel, rmse) of a depth map in metres against ground truth after the scale + shift alignment, over the pixels with
    ground truth (evaluator.py:346-366); (nan, nan) when no pixel has ground truth."""
    gt = np.asarray(depth_gt, np.float64)
    m = gt > 0
    if not m.any():
        return float("nan"), float("nan")
    sc, sh = fit_scale_and_shift(depth, gt)
    d = np.asarray(depth, np.float64)[m] * sc + sh
    return float(np.mean(np.abs(d - gt[m]) / gt[m])), float(np.sqrt(np.mean((d - gt[m]) ** 2)))


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
    excludes the part of the prediction that lies on void ground truth AND the part of the ground-truth segment that lies under
    void prediction (torchmetrics _calculate_iou: both void overlaps leave the union); unmatched predictions mostly (> 50 %) on void ground truth
    are not FPs and, symmetrically, unmatched ground-truth segments mostly covered by void PREDICTION (unknown categories become
    void under allow_unknown_preds_category) are not FNs (torchmetrics _panoptic_quality_update_sample, after COCO panopticapi)."""
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
        # torchmetrics _calculate_iou: union = pred - pred_on_void_target + target - target_under_void_prediction - intersection
        union = p_area[pc] + g_area[gc] - n - inter.get((pc, 0), 0) - inter.get((0, gc), 0)
        iou = n / union
        if iou > 0.5:
            c = pc // 100000
            out[c, 0] += iou
            out[c, 1] += 1
            matched_p.add(pc)
            matched_g.add(gc)
    for gc, a in g_area.items():
        if gc // 100000 != 0 and gc not in matched_g and inter.get((0, gc), 0) / a <= 0.5:
            out[gc // 100000, 3] += 1
    for pc, a in p_area.items():
        if pc // 100000 != 0 and pc not in matched_p and inter.get((pc, 0), 0) / a <= 0.5:
            out[pc // 100000, 2] += 1
    return out


def pq_from_stats(stats: np.ndarray, classes: Iterable[int] = THINGS + STUFFS) -> Dict[str, object]:
    """per-class PQ = sum IoU / (TP + FP/2 + FN/2) (0 where the class never occurs) in the order `classes` is given (the evaluator's
    `*_pqs_per_class`: torchmetrics' continuous ids = sorted things, then sorted stuffs), and its mean over ALL of them, absent
    classes included (np.mean of the per-class list, evaluator.py:384-387)."""
    per = []
    for c in classes:
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
    """Additive statistics of one rank.  Vector layout (fp64): [n_images, sum_psnr, n_ssim, sum_ssim, n_depth, sum_absrel, sum_rmse,
    n_lpips, sum_lpips] + context PQ [C,4] + target PQ [C,4] + context mIoU [C,2] + target mIoU [C,2]  ->  9 + 12 C doubles (261 for
    C = 21: 2 KB per rank in the all-gather)."""
    HEAD = 9

    def __init__(self, num_classes: int = NUM_CLASSES, things=THINGS, stuffs=STUFFS):
        self.C, self.things, self.stuffs = num_classes, tuple(things), tuple(stuffs)
        self.n_images, self.sum_psnr, self.n_ssim, self.sum_ssim = 0.0, 0.0, 0.0, 0.0
        self.n_depth, self.sum_absrel, self.sum_rmse = 0.0, 0.0, 0.0
        self.n_lpips, self.sum_lpips = 0.0, 0.0
        self.pq = {k: np.zeros((num_classes, 4)) for k in ("context", "target")}
        self.iou = {k: np.zeros((num_classes, 2)) for k in ("context", "target")}

    def add_render(self, pred01: np.ndarray, gt01: np.ndarray):
        return self.add_render_u8(png_roundtrip(pred01), png_roundtrip(gt01))

    def add_render_u8(self, pred: np.ndarray, gt: np.ndarray):
        """images as read back from the PNGs (uint8 / 255); returns the item's scores (render_scores.json)"""
        sc = {"psnr": psnr(pred, gt)}
        self.sum_psnr += sc["psnr"]
        self.n_images += 1
        if min(np.shape(pred)[:2]) > 10:  # (images smaller than the 11-tap window have no SSIM)
            sc["ssim"] = ssim(pred, gt)
            self.sum_ssim += sc["ssim"]
            self.n_ssim += 1
        return sc

    def add_lpips(self, value: float):
        """one rendered image's LPIPS distance (siu3r_amd.lpips.LPIPS; evaluator.py:263, 268)"""
        self.sum_lpips += float(value)
        self.n_lpips += 1
        return {"lpips": float(value)}

    def add_depth(self, depth_m: np.ndarray, depth_gt_m: np.ndarray):
        """one target view's rendered and ground-truth depth in metres (as read back from the millimetre PNGs); views without any
        ground-truth pixel are skipped (the reference would average a NaN)"""
        a, r = depth_errors(depth_m, depth_gt_m)
        if a == a:
            self.sum_absrel += a
            self.sum_rmse += r
            self.n_depth += 1
        return {"absrel": a, "rmse": r}

    def add_segmentation(self, which: str, pred_sem, pred_ins, gt_sem, gt_ins):
        """one scene: all its views concatenated along H, as the evaluator does (evaluator.py:146-150)."""
        self.pq[which] += panoptic_stats(pred_sem, pred_ins, gt_sem, gt_ins, self.things, self.stuffs, self.C)
        self.iou[which] += miou_stats(pred_sem, gt_sem, self.C)

    def to_vector(self) -> np.ndarray:
        return np.concatenate(([self.n_images, self.sum_psnr, self.n_ssim, self.sum_ssim, self.n_depth, self.sum_absrel, self.sum_rmse, self.n_lpips, self.sum_lpips], self.pq["context"].ravel(), self.pq["target"].ravel(),
                               self.iou["context"].ravel(), self.iou["target"].ravel())).astype(np.float64)

    @classmethod
    def from_vectors(cls, vecs: np.ndarray, num_classes: int = NUM_CLASSES, things=THINGS, stuffs=STUFFS) -> "MetricAccumulator":
        v = np.asarray(vecs, np.float64).reshape(-1, cls.HEAD + 12 * num_classes).sum(0)
        m = cls(num_classes, things, stuffs)
        m.n_images, m.sum_psnr, m.n_ssim, m.sum_ssim, m.n_depth, m.sum_absrel, m.sum_rmse, m.n_lpips, m.sum_lpips = v[:cls.HEAD]
        o, C = cls.HEAD, num_classes
        m.pq["context"] = v[o:o + 4 * C].reshape(C, 4); o += 4 * C
        m.pq["target"] = v[o:o + 4 * C].reshape(C, 4); o += 4 * C
        m.iou["context"] = v[o:o + 2 * C].reshape(C, 2); o += 2 * C
        m.iou["target"] = v[o:o + 2 * C].reshape(C, 2)
        return m

    def compute(self) -> Dict[str, object]:
        """the additive keys of the reference's results.json (evaluator.py:368-399); `context_map` / `target_map` are added by the caller from
        the gathered per-scene records (mean_average_precision); `lpips` appears when the caller scored images with siu3r_amd.lpips.LPIPS."""
        res: Dict[str, object] = {}
        if self.n_images:
            res["psnr"] = self.sum_psnr / self.n_images
        if self.n_ssim:
            res["ssim"] = self.sum_ssim / self.n_ssim
        if self.n_lpips:
            res["lpips"] = self.sum_lpips / self.n_lpips
        if self.n_depth:
            res["absrel"] = self.sum_absrel / self.n_depth
            res["rmse"] = self.sum_rmse / self.n_depth
        for k in ("context", "target"):
            if self.pq[k].sum() > 0:
                r = pq_from_stats(self.pq[k], self.things + self.stuffs)
                res[f"{k}_pqs_per_class"], res[f"{k}_pq"] = r["per_class"], r["pq"]
            if self.iou[k][1:, 1].sum() > 0:
                # include_background=False (evaluator.py:60-66): class 0 is dropped, the list has C - 1 entries (utils/miou.py:64-77)
                inter, union = self.iou[k][1:, 0], self.iou[k][1:, 1]
                iou = np.where(union > 0, inter / np.maximum(union, 1), 0.0)
                res[f"{k}_ious_per_class"], res[f"{k}_miou"] = iou.tolist(), float(np.mean(iou))
        return res


# ---- mean average precision (COCO protocol over instance masks) ---------------------------------------------------------------------
MAP_IOU_THRS = np.linspace(0.5, 0.95, 10)
MAP_REC_THRS = np.linspace(0.0, 1.0, 101)
MAP_MAX_DETS = (1, 10, 100)
MAP_AREAS = ((0.0, 1e10), (0.0, 32.0 ** 2), (32.0 ** 2, 96.0 ** 2), (96.0 ** 2, 1e10))  # all, small, medium, large (pixels)


def map_scene_inputs(pred_sem, pred_ins, gt_sem, gt_ins, pred_json=None, stuffs: Sequence[int] = STUFFS) -> Dict[str, np.ndarray]:
    """The evaluator's preparation of one scene (all views concatenated along H) for MeanAveragePrecision, evaluator.py:152-226:
    ground truth = every instance id != 0 whose category (first pixel's semantic id) is not a stuff class, label 0-based; detections =
    every predicted instance id != 0, label / score from pred.json (`label_id - 1`, the MEAN score of the infos sharing the id: fused
    stuff segments) or, without pred.json, the first pixel's semantic id - 1 and score 1.  (An id without an entry in pred.json makes
    the reference append a mask without label or score, which torchmetrics then rejects; here such an id is skipped.)
    The masks of an id map are disjoint, so areas and pairwise intersections come from one joint histogram of (predicted id, ground-truth
    id) instead of D x G mask products.  Returns det_labels / det_scores / det_area [D], gt_labels / gt_area [G], inter [D, G]."""
    ps, pi, gs, gi = (np.asarray(a).reshape(-1) for a in (pred_sem, pred_ins, gt_sem, gt_ins))
    g_ids, g_first, g_cnt = np.unique(gi, return_index=True, return_counts=True)
    d_ids, d_first, d_cnt = np.unique(pi, return_index=True, return_counts=True)
    g_keep = [k for k, g in enumerate(g_ids) if g != 0 and int(gs[g_first[k]]) not in stuffs]
    # (the first pixel in scan order is what `gt_semantics[gt_mask][0]` reads; return_index gives exactly that pixel)
    d_keep, dl, dsc = [], [], []
    for k, d in enumerate(d_ids):
        if d == 0:
            continue
        if pred_json is None:
            d_keep.append(k); dl.append(int(ps[d_first[k]]) - 1); dsc.append(1.0)
        else:
            info = [i for i in pred_json if i["id"] == int(d)]
            if info:
                d_keep.append(k); dl.append(int(info[0]["label_id"]) - 1); dsc.append(float(np.mean([i["score"] for i in info])))
    joint = np.zeros((len(d_ids), len(g_ids)), np.int64)
    np.add.at(joint, (np.searchsorted(d_ids, pi), np.searchsorted(g_ids, gi)), 1)
    return dict(det_labels=np.asarray(dl, np.int64), det_scores=np.asarray(dsc, np.float64), det_area=d_cnt[d_keep].astype(np.float64),
                gt_labels=np.asarray([int(gs[g_first[k]]) - 1 for k in g_keep], np.int64), gt_area=g_cnt[g_keep].astype(np.float64),
                inter=joint[np.ix_(d_keep, g_keep)].astype(np.float64))


def map_scene_records(inp: Dict[str, np.ndarray]) -> dict:
    """COCOeval.evaluateImg for one image (scene) and every category in it, masks without crowds: per category the detections sorted
    by score (stable, at most 100), and per area range the [T, D] "matched" and "ignored" flags and the ground truth's ignore flags.
    Small, picklable: this is what travels to rank 0."""
    det_labels, det_scores, gt_labels = inp["det_labels"], inp["det_scores"], inp["gt_labels"]
    recs = {}
    for c in sorted(set(det_labels.tolist()) | set(gt_labels.tolist())):
        d_idx = np.flatnonzero(det_labels == c)
        g_idx = np.flatnonzero(gt_labels == c)
        d_idx = d_idx[np.argsort(-det_scores[d_idx], kind="mergesort")][:MAP_MAX_DETS[-1]]
        d_area, g_area = inp["det_area"][d_idx], inp["gt_area"][g_idx]
        inter = inp["inter"][np.ix_(d_idx, g_idx)]
        iou = inter / np.maximum(d_area[:, None] + g_area[None, :] - inter, 1.0)
        per_area = []
        for lo, hi in MAP_AREAS:
            g_ig = (g_area < lo) | (g_area > hi)
            order = np.argsort(g_ig, kind="mergesort")  # ignored ground truth last
            g_ig_s, iou_s = g_ig[order], iou[:, order]
            T, D, G = len(MAP_IOU_THRS), len(d_idx), len(g_idx)
            gtm, dtm, dt_ig = np.zeros((T, G), bool), np.zeros((T, D), bool), np.zeros((T, D), bool)
            for ti, t in enumerate(MAP_IOU_THRS):
                for di in range(D):
                    best, m = min(t, 1 - 1e-10), -1
                    for gi_ in range(G):
                        if gtm[ti, gi_]:
                            continue
                        if m > -1 and not g_ig_s[m] and g_ig_s[gi_]:
                            break  # (a regular match is kept rather than traded for an ignored one)
                        if iou_s[di, gi_] < best:
                            continue
                        best, m = iou_s[di, gi_], gi_
                    if m == -1:
                        continue
                    dt_ig[ti, di], dtm[ti, di], gtm[ti, m] = g_ig_s[m], True, True
            d_out = (d_area < lo) | (d_area > hi)
            dt_ig = dt_ig | (~dtm & d_out[None, :])
            per_area.append(dict(dtm=dtm, dt_ig=dt_ig, g_ig=g_ig_s))
        recs[int(c)] = dict(scores=det_scores[d_idx], areas=per_area)
    return recs


def mean_average_precision(scene_records: Sequence[dict]) -> Dict[str, object]:
    """COCOeval.accumulate + summarize over the scenes' records, with torchmetrics MeanAveragePrecision's result keys (class_metrics=True):
    map, map_50, map_75, map_{small,medium,large}, mar_{1,10,100}, mar_{small,medium,large}, map_per_class, mar_100_per_class, classes.
    -1 where nothing could be evaluated (no ground truth), as the library reports it."""
    classes = sorted({c for r in scene_records for c in r})
    T, R, K, A, Mx = len(MAP_IOU_THRS), len(MAP_REC_THRS), len(classes), len(MAP_AREAS), len(MAP_MAX_DETS)
    precision, recall = -np.ones((T, R, K, A, Mx)), -np.ones((T, K, A, Mx))
    for ki, c in enumerate(classes):
        E = [r[c] for r in scene_records if c in r]
        for ai in range(A):
            for mi, md in enumerate(MAP_MAX_DETS):
                sc = np.concatenate([e["scores"][:md] for e in E]) if E else np.zeros(0)
                inds = np.argsort(-sc, kind="mergesort")
                dtm = np.concatenate([e["areas"][ai]["dtm"][:, :md] for e in E], axis=1)[:, inds]
                dtig = np.concatenate([e["areas"][ai]["dt_ig"][:, :md] for e in E], axis=1)[:, inds]
                gig = np.concatenate([e["areas"][ai]["g_ig"] for e in E])
                npig = int(np.count_nonzero(~gig))
                if npig == 0:
                    continue
                tp = np.cumsum(dtm & ~dtig, axis=1).astype(np.float64)
                fp = np.cumsum(~dtm & ~dtig, axis=1).astype(np.float64)
                for ti in range(T):
                    nd = tp.shape[1]
                    rc = tp[ti] / npig
                    pr = tp[ti] / (fp[ti] + tp[ti] + np.spacing(1))
                    recall[ti, ki, ai, mi] = rc[-1] if nd else 0.0
                    pr = pr.tolist()
                    for i in range(nd - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    q = np.zeros(R)
                    pos = np.searchsorted(rc, MAP_REC_THRS, side="left")
                    for ri, pi_ in enumerate(pos):
                        if pi_ < nd:
                            q[ri] = pr[pi_]
                    precision[ti, :, ki, ai, mi] = q

    def ap(iou=None, area=0, k=slice(None)):
        p = precision[:, :, k, area, 2] if iou is None else precision[np.isclose(MAP_IOU_THRS, iou), :, k, area, 2]
        p = p[p > -1]
        return float(p.mean()) if p.size else -1.0

    def ar(area=0, md=2, k=slice(None)):
        r = recall[:, k, area, md]
        r = r[r > -1]
        return float(r.mean()) if r.size else -1.0

    return dict(map=ap(), map_50=ap(0.5), map_75=ap(0.75), map_small=ap(area=1), map_medium=ap(area=2), map_large=ap(area=3),
                mar_1=ar(md=0), mar_10=ar(md=1), mar_100=ar(md=2), mar_small=ar(1), mar_medium=ar(2), mar_large=ar(3),
                map_per_class=[ap(k=ki) for ki in range(K)], mar_100_per_class=[ar(k=ki) for ki in range(K)], classes=[int(c) for c in classes])

