"""Golden vectors for the evaluator's in-repo arithmetic (build container only; needs /root/reference):
  evaluator_tree.npz   a seeded, build-owned result tree (two scenes: rendered / ground-truth depth in millimetres, predicted /
                       ground-truth (semantic, instance) maps of the context and target views) and what the REFERENCE's
                       Evaluator.evaluate (src/evaluator.py:240-404) writes into results.json for it with the depth-quality and
                       mIoU switches on: absrel, rmse (scale + shift fit :229-236, errors :346-366), context / target
                       ious_per_class and miou (the reference's own MeanIoU, src/utils/miou.py, through process_segmentation
                       :126-150).  torchmetrics is not in this image: `torchmetrics.Metric` is stubbed with the minimal state holder
                       MeanIoU needs; PSNR / SSIM / LPIPS / PQ / mAP are torchmetrics classes themselves and stay switched off.
Run:  python tests/golden/make_golden_eval.py
The test (tests/test_data_io.py::test_evaluator_tree_against_reference) rebuilds the tree from the stored arrays with the
product's own writers and compares siu3r_amd.eval_io.evaluate_dir with the stored results.
"""
import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _ref_import as R  # noqa: E402


def synth_tree(seed=0, H=48, W=64):
    """two scenes x (2 context + 3 target views): depth maps with holes in the ground truth, id maps with void, stuff, several things,
    a class only the prediction has, a class only the ground truth has"""
    rng = np.random.default_rng(seed)
    scenes = {}
    for s, name in enumerate(("scene0000_00", "scene0001_00")):
        d = {}
        for mode, n in (("context", 2), ("target", 3)):
            gs, gi = np.zeros((n, H, W), np.int64), np.zeros((n, H, W), np.int64)
            gs[:, : H // 4], gi[:, : H // 4] = 1, 1
            gs[:, 3 * H // 4:], gi[:, 3 * H // 4:] = 2, 2
            for v in range(n):
                for k, cls in enumerate((5, 7, 12, 5)):
                    y0, x0 = rng.integers(H // 4, H // 2), rng.integers(0, W - 20)
                    gs[v, y0:y0 + 10, x0:x0 + 16], gi[v, y0:y0 + 10, x0:x0 + 16] = cls, 3 + k
            ps, pi = gs.copy(), gi.copy()
            flip = rng.random(ps.shape) < 0.08
            ps[flip], pi[flip] = rng.integers(0, 21, int(flip.sum())), rng.integers(0, 9, int(flip.sum()))
            ps[:, H // 2:H // 2 + 4, :8], pi[:, H // 2:H // 2 + 4, :8] = 15 + s, 8   # a class the ground truth never has
            d[mode] = (ps, pi, gs, gi)
        gt = rng.uniform(0.6, 4.0, (3, H, W))
        gt[rng.random(gt.shape) < 0.15] = 0.0                       # holes
        pred = (gt * 0.8 + 0.3 + rng.normal(0, 0.05, gt.shape)).clip(0.05, None)
        pred[gt == 0] = rng.uniform(0.5, 4.0, int((gt == 0).sum()))
        d["depth"] = ((pred * 1000).astype(np.int32), (gt * 1000).astype(np.int32))
        scenes[name] = d
    return scenes


def write_tree(root, scenes, cids=(10, 30), tids=(12, 20, 28)):
    """the evaluator's on-disk layout, with PIL only (visualizer.py:270-392)"""
    from PIL import Image

    enc = lambda sem, ins: np.stack((lambda sid: (sid % 256, sid // 256 % 256, sid // 65536))(sem * 1000 + ins), -1).astype(np.uint8)
    for name, d in scenes.items():
        base = Path(root) / f"{name}_context{'_'.join(map(str, cids))}"
        for sub in ("depth", "depth_gt", "context_seg_pred", "context_seg_gt", "target_seg_pred", "target_seg_gt"):
            (base / sub).mkdir(parents=True, exist_ok=True)
        for j, t in enumerate(tids):
            Image.fromarray(d["depth"][0][j]).save(base / "depth" / f"{name}_{t}.png")
            Image.fromarray(d["depth"][1][j]).save(base / "depth_gt" / f"{name}_{t}.png")
        for mode, ids in (("context", cids), ("target", tids)):
            ps, pi, gs, gi = d[mode]
            for j, v in enumerate(ids):
                Image.fromarray(enc(ps[j], pi[j])).save(base / f"{mode}_seg_pred" / f"{name}_pred{v}.png")
                Image.fromarray(enc(gs[j], gi[j])).save(base / f"{mode}_seg_gt" / f"{name}_gt{v}.png")


def main():
    assert R.reference_available()
    R.install_stubs()
    tm = types.ModuleType("torchmetrics")

    class Metric(torch.nn.Module):  # the state holder MeanIoU needs (add_state / += / compute); nothing of torchmetrics' logic
        def __init__(self, **kw):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default.clone() if isinstance(default, torch.Tensor) else list(default))

    tm.Metric = Metric
    det, img = types.ModuleType("torchmetrics.detection"), types.ModuleType("torchmetrics.image")
    for m_, names in ((det, ("MeanAveragePrecision", "PanopticQuality")), (img, ("PeakSignalNoiseRatio", "StructuralSimilarityIndexMeasure", "LearnedPerceptualImagePatchSimilarity"))):
        for n_ in names:
            setattr(m_, n_, type(n_, (), {}))
    sys.modules.update({"torchmetrics": tm, "torchmetrics.detection": det, "torchmetrics.image": img})
    ru = types.ModuleType("rootutils")
    ru.setup_root = lambda *a, **k: None
    sys.modules["rootutils"] = ru
    sys.modules["hydra"].main = lambda **k: (lambda f: f)
    sys.modules.setdefault("tqdm", types.ModuleType("tqdm")).tqdm = lambda it, **k: it
    from src.config import EvaluatorCfg
    from src.evaluator import Evaluator
    from src.utils.scannet_constant import PANOPTIC_SEMANTIC2NAME

    cfg = EvaluatorCfg(dataset_name="scannet", eval_context_miou=True, eval_context_pq=False, eval_context_map=False, eval_target_miou=True,
                       eval_target_pq=False, eval_target_map=False, eval_image_quality=False, eval_depth_quality=True,
                       id2label=PANOPTIC_SEMANTIC2NAME, stuffs=[0, 1], things=list(range(2, 20)), device="cpu")
    scenes = synth_tree()
    with tempfile.TemporaryDirectory() as tmp:
        write_tree(tmp, scenes)
        ev = Evaluator(cfg)
        ev.setup()
        res = ev.evaluate(tmp)
        per_item = {d.name: json.load(open(d / "depth_scores.json")) for d in sorted(Path(tmp).iterdir()) if d.is_dir()}
    out = {}
    for name, d in scenes.items():
        out[f"{name}.depth"], out[f"{name}.depth_gt"] = d["depth"]
        for mode in ("context", "target"):
            for k, a in zip(("pred_sem", "pred_ins", "gt_sem", "gt_ins"), d[mode]):
                out[f"{name}.{mode}.{k}"] = a.astype(np.int16)
    for k in ("absrel", "rmse", "context_miou", "target_miou"):
        out[f"result.{k}"] = np.asarray(res[k], np.float64)
    for k in ("context_ious_per_class", "target_ious_per_class"):
        out[f"result.{k}"] = np.asarray(res[k], np.float64)
    out["result.depth_items"] = np.asarray([[it["absrel"], it["rmse"]] for s_ in sorted(per_item) for it in per_item[s_]], np.float64)
    np.savez_compressed(os.path.join(HERE, "evaluator_tree.npz"), **out)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items() if k.startswith("result")})


if __name__ == "__main__":
    main()
