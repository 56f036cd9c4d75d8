"""CPU tests of the additive evaluator metrics (siu3r_amd/metrics.py) on hand-computed cases."""
import numpy as np

from siu3r_amd import metrics as M


def test_psnr_and_png_truncation():
    gt = np.zeros((4, 4, 3), np.float32)
    gt[..., 0] = 1.0
    pred = gt.copy()
    pred[0, 0, 0] = 0.5
    # MSE = 0.25 / 48, data_range = 1
    assert abs(M.psnr(pred, gt) - 10 * np.log10(48 / 0.25)) < 1e-9
    assert np.allclose(M.png_roundtrip(np.array([0.999, 0.5, 1.0], np.float32)), np.array([254, 127, 255], np.float32) / 255.0, atol=0)  # truncation, not rounding
    acc = M.MetricAccumulator()
    acc.add_render(pred, gt)
    assert abs(acc.compute()["psnr"] - M.psnr(M.png_roundtrip(pred), M.png_roundtrip(gt))) < 1e-12


def _scene():
    H, W = 8, 12
    gs, gi = np.zeros((H, W), int), np.zeros((H, W), int)
    gs[:, :4], gs[:, 4:8], gi[:, 4:8] = 1, 5, 1      # wall (stuff) | chair #1
    gs[:4, 8:], gi[:4, 8:] = 5, 2                     # chair #2 (upper right); lower right stays void
    ps, pi = gs.copy(), gi.copy()
    ps[:, 4:6], pi[:, 4:6] = 1, 7                     # left half of chair #1 predicted as wall (instance ids of stuff are ignored)
    ps[4:, 8:], pi[4:, 8:] = 7, 3                     # a table predicted on void ground truth: not a false positive
    return ps, pi, gs, gi


def test_panoptic_quality_hand_case():
    ps, pi, gs, gi = _scene()
    st = M.panoptic_stats(ps, pi, gs, gi)
    # wall: pred 48 px, gt 32 px, inter 32 -> IoU 2/3 > 0.5: TP
    assert st[1].tolist() == [32 / 48, 1, 0, 0]
    # chair #1: pred 16 px (cols 6-7), gt 32, inter 16 -> IoU 0.5, not > 0.5: FP + FN; chair #2 exact: TP with IoU 1
    assert st[5].tolist() == [1.0, 1, 1, 1]
    assert st[7].tolist() == [0, 0, 0, 0]            # table lies entirely on void
    r = M.pq_from_stats(st, classes=(5, 7, 1))  # the evaluator's order: things, then stuffs
    assert np.allclose(r["per_class"], [1.0 / 2.0, 0.0, 2 / 3]) and abs(r["pq"] - (0.5 + 2 / 3) / 3) < 1e-12
    assert M.miou_stats(ps, gs)[1].tolist() == [32, 48] and M.miou_stats(ps, gs)[5].tolist() == [32, 48]


def test_accumulator_is_additive():
    ps, pi, gs, gi = _scene()
    a, b, whole = M.MetricAccumulator(), M.MetricAccumulator(), M.MetricAccumulator()
    rng = np.random.default_rng(0)
    for acc in (a, whole):
        acc.add_segmentation("context", ps, pi, gs, gi)
        acc.add_render(np.full((2, 2, 3), 0.3, np.float32), np.array([[[0, 1, 0.5]] * 2] * 2, np.float32))
    ps2 = np.where(rng.random(ps.shape) < 0.2, 5, ps)
    for acc in (b, whole):
        acc.add_segmentation("context", ps2, pi, gs, gi)
        acc.add_segmentation("target", ps, pi, gs, gi)
    merged = M.MetricAccumulator.from_vectors(np.stack((a.to_vector(), b.to_vector())))
    assert a.to_vector().shape == (9 + 12 * 21,)
    assert merged.compute() == whole.compute()
    assert set(whole.compute()) >= {"psnr", "context_pq", "context_miou", "target_pq", "target_miou"}


def test_panoptic_quality_void_and_unknown_categories():
    """torchmetrics PanopticQuality(allow_unknown_preds_category=True) semantics on the cases the hand case above does not reach:
    unknown predicted categories become void; the IoU's union leaves out BOTH void overlaps (the part of the prediction on void
    ground truth and the part of the ground-truth segment under void prediction: _calculate_iou); an unmatched ground-truth segment
    mostly covered by void prediction is NOT a false negative, one covered by less is; instance 0 of a thing is an instance like
    any other; absent classes count as PQ 0 in the evaluator's mean."""
    H, W = 10, 10
    gs, gi = np.zeros((H, W), int), np.zeros((H, W), int)
    gs[:5, :], gi[:5, :] = 5, 1                       # chair A: 50 px
    gs[5:, :5], gi[5:, :5] = 7, 0                     # table, instance id 0: 25 px; the remaining 25 px are void
    ps, pi = gs.copy(), gi.copy()
    ps[:3, :] = 99                                    # 30 of chair A's 50 px predicted as an unknown category -> void
    pi[:3, :] = 4
    ps[5:, 5:8], pi[5:, 5:8] = 7, 0                   # the table prediction also covers 15 px of void ground truth
    st = M.panoptic_stats(ps, pi, gs, gi)
    # chair: prediction 20 px, gt 50 px of which 30 lie under void prediction: union = 20 + 50 - 20 - 0 - 30 = 20 -> IoU 1.0, matched
    # (without the void-prediction term the IoU would read 0.4 and the pair would count as FP: the round-4 advisor's finding)
    assert st[5].tolist() == [1.0, 1, 0, 0]
    # table: pred 40 px (25 on gt, 15 on void) -> union = 40 + 25 - 25 - 15 = 25 -> IoU 1.0
    assert st[7].tolist() == [1.0, 1, 0, 0]
    # a void hole INSIDE a matched ground-truth segment: 8 x 8 chair, the prediction has the right shape but a 4 x 4 hole of void in
    # it and spills 8 px over the border: inter 48, pred 56, gt 64, void-pred on gt 16 -> union = 56 + 64 - 48 - 16 = 56 -> 6/7
    gh, gih = np.zeros((12, 12), int), np.zeros((12, 12), int)
    gh[:8, :8], gih[:8, :8] = 5, 1
    gh[:8, 8:], gih[:8, 8:] = 1, 0                    # wall beside it (so that the spill does not land on void ground truth)
    ph, pih = gh.copy(), gih.copy()
    ph[2:6, 2:6], pih[2:6, 2:6] = 0, 0                # void hole
    ph[:8, 8], pih[:8, 8] = 5, 1                      # spill of 8 px onto the wall
    sth = M.panoptic_stats(ph, pih, gh, gih)
    assert np.allclose(sth[5], [48 / 56, 1, 0, 0])
    assert np.allclose(sth[1], [24 / 32, 1, 0, 0])    # wall: pred 24, gt 32, inter 24
    # the false-negative rule proper: the 20 px of chair A that are not void-predicted go to ANOTHER category (table #3), so the chair
    # has no same-category candidate: unmatched, 60 % of it under void prediction -> not a false negative; the table piece is a FP
    ps1, pi1 = ps.copy(), pi.copy()
    ps1[3:5, :], pi1[3:5, :] = 7, 3
    st1 = M.panoptic_stats(ps1, pi1, gs, gi)
    assert st1[5].tolist() == [0, 0, 0, 0] and st1[7].tolist() == [1.0, 1, 1, 0]
    ps2, pi2 = ps1.copy(), pi1.copy()
    ps2[1:3, :], pi2[1:3, :] = 7, 3                   # now only 10 of the chair's 50 px are void-predicted (20 %): a false negative again
    st2 = M.panoptic_stats(ps2, pi2, gs, gi)
    assert st2[5].tolist() == [0, 0, 0, 1] and st2[7].tolist() == [1.0, 1, 1, 0]
    ps3 = ps.copy()
    ps3[:3, :] = 5
    pi3 = pi.copy()
    pi3[:3, :] = 9                                    # a second chair instance takes the upper 30 px instead: 0.6 IoU -> it is the match
    st3 = M.panoptic_stats(ps3, pi3, gs, gi)
    assert st3[5].tolist() == [0.6, 1, 1, 0]
    r = M.pq_from_stats(st1, classes=M.THINGS + M.STUFFS)
    assert len(r["per_class"]) == 20 and abs(r["per_class"][M.THINGS.index(7)] - 1.0 / 1.5) < 1e-12 and abs(r["pq"] - (1.0 / 1.5) / 20) < 1e-12
    # unknown categories in the GROUND TRUTH are void as well (torchmetrics always allows them there)
    gs4 = gs.copy()
    gs4[:5, :] = 42
    st4 = M.panoptic_stats(gs, gi, gs4, gi)
    assert st4[5].tolist() == [0, 0, 0, 0] and st4[7].tolist() == [1.0, 1, 0, 0]  # the chair prediction lies on void entirely: no FP


def test_ssim_definition():
    """SSIM restated from torchmetrics' defaults (11-tap sigma-1.5 Gaussian, reflect pad 5, crop 5, k1 = 0.01, k2 = 0.03):
    identical images -> 1; a constant offset on a flat image -> the closed form of the luminance term; symmetric; additive keys."""
    rng = np.random.default_rng(0)
    a = rng.random((32, 40, 3))
    assert abs(M.ssim(a, a) - 1.0) < 1e-12
    b = np.clip(a + rng.normal(0, 0.1, a.shape), 0, 1)
    s_ab = M.ssim(a, b, data_range=1.0)
    assert 0.0 < s_ab < 1.0 and abs(s_ab - M.ssim(b, a, data_range=1.0)) < 1e-12
    # flat images x and y: variances 0 -> SSIM = (2xy + c1) / (x^2 + y^2 + c1) (c2 cancels) with data_range fixed
    x, y = np.full((24, 24, 1), 0.4), np.full((24, 24, 1), 0.6)
    c1 = (0.01 * 1.0) ** 2
    assert abs(M.ssim(x, y, data_range=1.0) - (2 * 0.24 + c1) / (0.16 + 0.36 + c1)) < 1e-12
    # window weights: normalised, symmetric, the published 11-tap values
    g = M._gauss1d()
    assert abs(g.sum() - 1) < 1e-15 and np.allclose(g, g[::-1]) and abs(g[5] / g[4] - np.exp(0.5 / 2.25)) < 1e-12
    # a single bright pixel far from the border: the SSIM map differs from 1 only inside the 11 x 11 window around it
    z = np.zeros((40, 40, 1))
    z2 = z.copy()
    z2[20, 20, 0] = 1.0
    full = 1.0 - M.ssim(z, z2, data_range=1.0)
    assert 0 < full < 121 / (30 * 30)
    acc = M.MetricAccumulator()
    sc = acc.add_render(a.astype(np.float32), b.astype(np.float32))
    res = acc.compute()
    assert set(sc) == {"psnr", "ssim"} and abs(res["ssim"] - sc["ssim"]) < 1e-15 and abs(res["psnr"] - sc["psnr"]) < 1e-12


def test_depth_errors_scale_shift():
    rng = np.random.default_rng(1)
    gt = rng.uniform(0.5, 4.0, (20, 30))
    gt[rng.random(gt.shape) < 0.2] = 0.0
    pred = gt * 0.5 + 0.25                             # an exact affine image of the ground truth: both errors vanish after the fit
    pred[gt == 0] = 7.0                                # (pixels without ground truth do not enter)
    a, r = M.depth_errors(pred, gt)
    assert a < 1e-12 and r < 1e-12
    sc, sh = M.fit_scale_and_shift(pred, gt)
    assert abs(sc - 2.0) < 1e-9 and abs(sh + 0.5) < 1e-9
    noisy = pred + rng.normal(0, 0.02, pred.shape)
    a2, r2 = M.depth_errors(noisy, gt)
    assert 0 < a2 < 0.1 and 0 < r2 < 0.1
    acc = M.MetricAccumulator()
    acc.add_depth(noisy, gt)
    acc.add_depth(pred, np.zeros_like(gt))             # no ground truth at all: skipped, not a NaN in the mean
    res = acc.compute()
    assert abs(res["absrel"] - a2) < 1e-15 and abs(res["rmse"] - r2) < 1e-15
    merged = M.MetricAccumulator.from_vectors(np.stack((acc.to_vector(), acc.to_vector())))
    assert abs(merged.compute()["rmse"] - r2) < 1e-12 and merged.n_depth == 2


def _boxes(H, W, boxes):
    """id / category maps from a list of (id, category, y0, y1, x0, x1)"""
    sem, ins = np.zeros((H, W), int), np.zeros((H, W), int)
    for i, c, y0, y1, x0, x1 in boxes:
        sem[y0:y1, x0:x1], ins[y0:y1, x0:x1] = c, i
    return sem, ins


def test_mean_average_precision_hand_cases():
    """COCO protocol over instance masks (torchmetrics MeanAveragePrecision(iou_type="segm", class_metrics=True), evaluator.py:93-106,
    152-226, 388-399) on cases worked out by hand: one class, two ground-truth chairs of 50 px each; three detections -- score 0.9 exact on
    chair 1 (IoU 1), score 0.8 covering 31 of chair 2's 50 px (IoU 0.62), score 0.7 somewhere else (false positive).
      IoU thresholds 0.5, 0.55, 0.6: TP, TP, FP -> recall [0.5, 1, 1], precision [1, 1, 2/3] -> AP 1
      thresholds 0.65 .. 0.95:       TP, FP, FP -> recall 0.5 throughout, precision [1, 1/2, 1/3]: 1 at the 51 recall points <= 0.5, 0 at
                                     the other 50 -> AP 51 / 101
      map = (3 * 1 + 7 * 51/101) / 10, map_50 = 1, map_75 = 51/101; mar_100 = (3 * 1 + 7 * 0.5) / 10; mar_1 (best detection only) = 0.5"""
    H, W = 20, 40
    gs, gi = _boxes(H, W, [(1, 5, 0, 5, 0, 10), (2, 5, 10, 15, 0, 10)])
    ps, pi = _boxes(H, W, [(1, 5, 0, 5, 0, 10), (3, 5, 15, 20, 30, 40)])
    ps[10:13, 0:10], pi[10:13, 0:10] = 5, 2   # 30 px of chair 2 ...
    ps[13, 0], pi[13, 0] = 5, 2               # ... and one more: 31 of 50, nothing outside -> IoU 0.62
    pred_json = [dict(id=1, label_id=5, score=0.9), dict(id=2, label_id=5, score=0.8), dict(id=3, label_id=5, score=0.7)]
    inp = M.map_scene_inputs(ps, pi, gs, gi, pred_json)
    assert inp["det_labels"].tolist() == [4, 4, 4] and inp["gt_labels"].tolist() == [4, 4] and inp["det_area"].tolist() == [50, 31, 50]
    assert inp["inter"].tolist() == [[50, 0], [0, 31], [0, 0]]
    r = M.mean_average_precision([M.map_scene_records(inp)])
    ap_hi = 51 / 101
    assert abs(r["map"] - (3 + 7 * ap_hi) / 10) < 1e-9 and abs(r["map_50"] - 1) < 1e-9 and abs(r["map_75"] - ap_hi) < 1e-9
    assert abs(r["mar_100"] - 0.65) < 1e-12 and abs(r["mar_10"] - 0.65) < 1e-12 and abs(r["mar_1"] - 0.5) < 1e-12
    assert r["classes"] == [4] and abs(r["map_per_class"][0] - r["map"]) < 1e-12
    # all four instances have 50 px: "small" (< 32^2); no medium / large ground truth -> -1 there, as the library reports it
    assert abs(r["map_small"] - r["map"]) < 1e-12 and r["map_medium"] == -1.0 and r["map_large"] == -1.0 and r["mar_large"] == -1.0


def test_mean_average_precision_is_not_additive_and_handles_stuff_and_absent_classes():
    """Two scenes: a confident false positive in a scene WITHOUT ground truth of that class ranks above the other scene's true positive:
    detections sorted over the whole set -> recall [0, 1], precision [0, 1/2] -> AP 0.5 (per-scene APs would be "undefined" and 1: the
    metric cannot be accumulated as a sum -- hence the per-scene records and the second gather).  Ground-truth stuff segments are not
    instances (evaluator.py:160-161) while predicted stuff ids stay detections (class without ground truth: -1, left out of the mean);
    fused stuff ids take the MEAN score of their infos; without pred.json every detection scores 1."""
    H, W = 48, 48
    g1s, g1i = _boxes(H, W, [(1, 1, 0, 48, 0, 8)])                              # scene 1: only wall (stuff): no chair ground truth
    p1s, p1i = _boxes(H, W, [(101, 1, 0, 48, 0, 8), (2, 5, 10, 20, 10, 20)])    # a wall id (fused) and a confident chair that is not there
    j1 = [dict(id=101, label_id=1, score=0.6), dict(id=101, label_id=1, score=0.8), dict(id=2, label_id=5, score=0.9)]
    g2s, g2i = _boxes(H, W, [(1, 5, 0, 40, 0, 40)])                              # scene 2: one chair of 1600 px (medium)
    p2s, p2i = g2s.copy(), g2i.copy()
    j2 = [dict(id=1, label_id=5, score=0.8)]
    i1, i2 = M.map_scene_inputs(p1s, p1i, g1s, g1i, j1), M.map_scene_inputs(p2s, p2i, g2s, g2i, j2)
    assert i1["gt_labels"].shape[0] == 0 and sorted(i1["det_labels"].tolist()) == [0, 4] and abs(float(i1["det_scores"][i1["det_labels"] == 0][0]) - 0.7) < 1e-12
    recs = [M.map_scene_records(i1), M.map_scene_records(i2)]
    r = M.mean_average_precision(recs)
    assert r["classes"] == [0, 4] and r["map_per_class"][0] == -1.0 and abs(r["map_per_class"][1] - 0.5) < 1e-9
    # area ranges: the false positive is 100 px -- outside "medium" and unmatched, hence ignored there: the medium range sees only the true positive
    assert abs(r["map"] - 0.5) < 1e-9 and abs(r["map_medium"] - 1.0) < 1e-9 and r["map_small"] == -1.0 and abs(r["mar_100"] - 1.0) < 1e-12
    assert abs(M.mean_average_precision(recs[1:])["map"] - 1.0) < 1e-9        # the second scene alone: AP 1
    no_json = M.map_scene_inputs(p2s, p2i, g2s, g2i, None)
    assert no_json["det_scores"].tolist() == [1.0] and no_json["det_labels"].tolist() == [4]

