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
    r = M.pq_from_stats(st, classes=(1, 5, 7))
    assert np.allclose(r["per_class"], [2 / 3, 1.0 / 2.0, 0.0])
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
    assert a.to_vector().shape == (2 + 12 * 21,)
    assert merged.compute() == whole.compute()
    assert set(whole.compute()) >= {"psnr", "context_pq", "context_miou", "target_pq", "target_miou"}
