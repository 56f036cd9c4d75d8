"""Host-side data path (CPU): ScanNet validation-pair reader, the evaluator's on-disk format, the viewer's PLY loader and the CLI
preprocessing.  Pins: tests/golden/scannet_gt.npz (ground-truth conversion, relative poses and intrinsics produced by the REFERENCE's
own dataset / processor code, tests/golden/make_golden_io.py)."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fake_scannet(root, n_scans=2, n_frames=4, seed=0):
    """a tiny tree in ScanNet's processed layout (256 x 256 frames) + val_pair.json"""
    from PIL import Image

    z = np.load(os.path.join(GOLDEN, "scannet_gt.npz"))
    rng = np.random.default_rng(seed)
    pairs = []
    for s in range(n_scans):
        scan = f"scene{s:04d}_00"
        d = os.path.join(root, "val", scan)
        for sub in ("color", "depth", "panoptic", "extrinsic"):
            os.makedirs(os.path.join(d, sub), exist_ok=True)
        np.savetxt(os.path.join(d, "intrinsic.txt"), z["K"])
        for f in range(n_frames):
            fid = 10 * f
            Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).save(os.path.join(d, "color", f"{fid}.jpg"), quality=95)
            Image.fromarray((1000 + rng.integers(0, 3000, (256, 256))).astype(np.uint16)).save(os.path.join(d, "depth", f"{fid}.png"))
            sem, ins = z["sem"][f % 3].astype(np.int64), z["ins"][f % 3].astype(np.int64)
            sid = 1000 * sem + ins
            Image.fromarray(np.stack((sid % 256, sid // 256, sid // 65536), -1).astype(np.uint8)).save(os.path.join(d, "panoptic", f"{fid}.png"))
            ext = np.eye(4)
            ext[:3, 3] = rng.normal(size=3)
            c, s_ = np.cos(0.1 * f), np.sin(0.1 * f)
            ext[:3, :3] = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
            np.savetxt(os.path.join(d, "extrinsic", f"{fid}.txt"), ext)
        pairs.append({"scan": scan, "context_ids": [0, 20], "target_ids": [0, 10, 20]})
    with open(os.path.join(root, "val_pair.json"), "w") as fh:
        json.dump(pairs, fh)
    return pairs


def test_ground_truth_conversion_matches_reference_processor():
    from siu3r_amd import scannet

    z = np.load(os.path.join(GOLDEN, "scannet_gt.npz"))
    sems, inss = list(z["sem"].astype(np.int64)), list(z["ins"].astype(np.int64))
    i2s = [scannet.instance_to_semantic(s, i) for s, i in zip(sems, inss)]
    masks, classes = scannet.panoptic_ground_truth(inss, i2s)
    assert masks.dtype == torch.int64 and tuple(masks.shape) == tuple(z["mask_labels"].shape)
    assert np.array_equal(masks.numpy(), z["mask_labels"].astype(np.int64)) and classes.tolist() == z["class_labels"].tolist()
    rc, rt = scannet.relative_pose(list(z["cext"]), list(z["text"]))
    assert np.allclose(np.stack(rc), z["rel_c"], atol=0) and np.allclose(np.stack(rt), z["rel_t"], atol=0)
    assert np.allclose(rc[0], np.eye(4), atol=1e-12)
    assert np.array_equal(scannet.intrinsics_normalize(z["K"]), z["Kn"])
    with pytest.raises(ValueError):
        scannet.instance_to_semantic(np.zeros((4, 4), np.int64), np.zeros((4, 4), np.int64))


def test_val_pair_reader_and_collate(tmp_path):
    from siu3r_amd import scannet

    pairs = _fake_scannet(str(tmp_path))
    ds = scannet.ScanNetValPairs(str(tmp_path))
    assert len(ds) == len(pairs)
    it = ds[1]
    assert it["scene_names"] == "scene0001_00" and it["context_views_id"] == [0, 20] and it["target_views_id"] == [0, 10, 20]
    assert it["context_views_images"][0].shape == (3, 256, 256) and it["context_views_images"][0].dtype == np.uint8
    assert np.allclose(it["context_views_extrinsics"][0], np.eye(4), atol=1e-12)          # canonical frame = first context view
    assert np.allclose(it["target_views_extrinsics"][0], np.eye(4), atol=1e-12)           # target 0 IS context 0
    assert np.allclose(it["target_views_extrinsics"][2], it["context_views_extrinsics"][1])
    assert it["context_views_depths"][0].max() <= 4.0 and it["context_views_depths"][0].min() >= 1.0   # millimetres -> metres
    assert tuple(it["target_mask_labels"].shape[1:]) == (3, 256, 256) and it["target_class_labels"].dtype == torch.int64
    b = scannet.collate([ds[0], ds[1]])
    assert b["context_views_images"].shape == (2, 2, 3, 256, 256) and b["context_views_images"].dtype == torch.float32 and float(b["context_views_images"].max()) <= 1.0
    assert b["target_views_extrinsics"].shape == (2, 3, 4, 4) and b["target_views_intrinsics"].shape == (2, 3, 3, 3)
    assert abs(float(b["context_views_intrinsics"][0, 0, 0, 0]) - 577.87 / 256) < 1e-6
    assert scannet.shard(7, 1, 2) == [1, 3, 5]


def test_eval_files_round_trip(tmp_path):
    """writers in the reference's layout -> readers -> results.json keys; PSNR on the truncated uint8 files"""
    from siu3r_amd import eval_io as E, metrics as M

    rng = np.random.default_rng(0)
    B, N, H, W = 2, 3, 16, 20
    names, cids, tids = ["scene0000_00", "scene0001_00"], [[0, 20], [5, 15]], [[0, 10, 20], [5, 10, 15]]
    rend, gt = rng.random((B, N, 3, H, W)).astype(np.float32), rng.random((B, N, 3, H, W)).astype(np.float32)
    dep, dgt = 1 + rng.random((B, N, H, W)).astype(np.float32), 1 + rng.random((B, N, H, W)).astype(np.float32)
    E.save_recon_images(torch.from_numpy(rend), torch.from_numpy(dep), torch.from_numpy(gt), torch.from_numpy(dgt), tmp_path, names, cids, tids)
    sem = rng.integers(0, 21, (B, N, H, W))
    ins = np.where(sem == 0, 0, rng.integers(1, 103, (B, N, H, W)))
    infos = [[{"id": 3, "label_id": 5, "was_fused": False, "score": 0.9}], []]
    E.save_seg_ids("target", torch.from_numpy(sem), torch.from_numpy(ins), tmp_path, names, cids, tids, infos)
    E.save_seg_ids("context", torch.from_numpy(sem[:, [0, 2]]), torch.from_numpy(ins[:, [0, 2]]), tmp_path, names, cids, tids, infos)
    # ground truth as (mask_labels, class_labels): instance k+1 carries class c -> semantic c+1
    ml = [torch.zeros(2, N, H, W, dtype=torch.int64) for _ in range(B)]
    for m in ml:
        m[0, :, : H // 2] = 1
        m[1, :, H // 2:] = 1
    cl = [torch.tensor([0, 4]), torch.tensor([1, 7])]
    E.save_gt_seg_masks("target", ml, cl, tmp_path, names, cids, tids)
    E.save_gt_seg_masks("context", [m[:, [0, 2]] for m in ml], cl, tmp_path, names, cids, tids)
    d0 = tmp_path / "scene0000_00_context0_20"
    assert sorted(p.name for p in d0.iterdir()) == ["context_seg_gt", "context_seg_pred", "depth", "depth_gt", "rgb", "rgb_gt", "target_seg_gt", "target_seg_pred"]
    assert sorted(p.name for p in (d0 / "target_seg_pred").iterdir()) == ["pred.json", "scene0000_00_pred0.png", "scene0000_00_pred10.png", "scene0000_00_pred20.png"]
    assert (d0 / "context_seg_gt" / "scene0000_00_gt20.png").exists() and json.load(open(d0 / "target_seg_pred" / "pred.json")) == infos[0]
    from PIL import Image

    im = np.array(Image.open(d0 / "rgb" / "scene0000_00_10.png"))
    assert np.array_equal(im, (np.transpose(rend[0, 1], (1, 2, 0)) * 255).astype(np.uint8))          # truncation, not rounding
    dm = np.array(Image.open(d0 / "depth" / "scene0000_00_10.png"))
    assert np.array_equal(dm.astype(np.int64), (dep[0, 1] * 1000).astype(np.int32))   # mode 'I' is stored as a 16-bit PNG by PIL (same in the reference)
    ps, pi, gs, gi = E.load_segmentation_dir(d0 / "target_seg_pred", d0 / "target_seg_gt")
    assert ps.shape == (N * H, W) and np.array_equal(ps, np.concatenate(list(sem[0]), 0)) and np.array_equal(pi, np.concatenate(list(ins[0]), 0))
    assert set(np.unique(gs)) == {1, 5} and set(np.unique(gi)) == {1, 2}
    res = E.evaluate_dir(tmp_path)
    assert json.load(open(tmp_path / "results.json")) == res
    assert {"psnr", "context_pq", "target_pq", "context_pqs_per_class", "target_miou", "target_ious_per_class"} <= set(res)
    want = np.mean([M.psnr(M.png_roundtrip(np.transpose(rend[b, n], (1, 2, 0))), M.png_roundtrip(np.transpose(gt[b, n], (1, 2, 0)))) for b in range(B) for n in range(N)])
    assert abs(res["psnr"] - want) < 1e-9
    # a second writer pass over an existing scene is skipped (the reference's tail de-duplication)
    E.save_recon_images(torch.zeros(B, N, 3, H, W), torch.from_numpy(dep), torch.from_numpy(gt), torch.from_numpy(dgt), tmp_path, names, cids, tids)
    assert np.array_equal(np.array(Image.open(d0 / "rgb" / "scene0000_00_10.png")), im)
    # sharded accumulation == whole-directory accumulation
    a = E.accumulate_dir(tmp_path, scenes=["scene0000_00_context0_20"], write_scene_scores=False)
    b = E.accumulate_dir(tmp_path, scenes=["scene0001_00_context5_15"], write_scene_scores=False)
    both = M.MetricAccumulator.from_vectors(np.stack((a.to_vector(), b.to_vector()))).compute()
    assert abs(both["psnr"] - res["psnr"]) < 1e-12 and abs(both["target_pq"] - res["target_pq"]) < 1e-12
    # mean average precision (evaluator.py:388-399: the torchmetrics result dict under context_map / target_map): not additive -- the
    # shards' per-scene match records, concatenated, give the whole directory's value
    assert {"map", "map_50", "mar_100", "map_per_class", "classes"} <= set(res["target_map"]) and "context_map" in res
    ra, rb = {}, {}
    E.accumulate_dir(tmp_path, scenes=["scene0000_00_context0_20"], write_scene_scores=False, map_records=ra)
    E.accumulate_dir(tmp_path, scenes=["scene0001_00_context5_15"], write_scene_scores=False, map_records=rb)
    assert M.mean_average_precision(E.ordered_map_records(rb["target"] + ra["target"])) == res["target_map"]  # (rank order != scene order)
    assert [n for n, _ in ra["target"]] == ["scene0000_00_context0_20"]


def test_segment_id_encoding_is_the_references():
    from siu3r_amd import eval_io as E, metrics as M

    sem = np.array([[0, 1, 20, 20]])
    ins = np.array([[0, 101, 999, 7]])
    rgb = E.encode_segment_ids(sem, ins)
    assert rgb.dtype == np.uint8 and rgb.tolist() == [[[0, 0, 0], [1101 % 256, 1101 // 256, 0], [20999 % 256, 20999 // 256, 0], [20007 % 256, 20007 // 256, 0]]]
    s, i = M.decode_segment_ids(rgb)
    assert np.array_equal(s, sem) and np.array_equal(i, ins)


def test_viewer_ply_loader(tmp_path):
    """export_ply -> load_ply: name-sorted properties, [P, K, 3] coefficient layout, 5-px border crop per view, file units kept"""
    from siu3r_amd.ply_export import export_ply
    from siu3r_amd.viewer import load_ply

    H, W, V, q = 16, 20, 2, 2
    G = V * H * W
    g = torch.Generator().manual_seed(0)
    means, scales, rot = torch.randn(G, 3, generator=g), 0.01 + torch.rand(G, 3, generator=g), torch.randn(G, 4, generator=g)
    sh, op = torch.randn(G, 3, 25, generator=g), torch.rand(G, generator=g)
    sem, ins = torch.randint(0, 21, (G,), generator=g, dtype=torch.int32), torch.randint(0, 100, (G,), generator=g, dtype=torch.int32)
    qcl = torch.rand(G, q, 21, generator=g)
    p = export_ply(means, scales, rot, sh, op, sem, ins, qcl, tmp_path / "o.ply", save_sh_dc_only=False)
    full = load_ply(p, H, W, crop=False)
    assert full["max_sh_degree"] == 4 and full["means"].shape == (G, 3) and full["sh0"].shape == (G, 1, 3) and full["shN"].shape == (G, 24, 3)
    assert torch.equal(full["means"], means) and torch.allclose(full["scales"], scales.log()) and torch.equal(full["opacities"], op)
    assert torch.equal(full["quats"], rot[:, [3, 0, 1, 2]])                                  # file order is wxyz of the raw xyzw quaternion
    assert torch.equal(full["sh0"][:, 0], sh[:, :, 0]) and torch.equal(full["shN"], sh[:, :, 1:].transpose(1, 2))
    assert torch.equal(full["semantic_label"], sem.long()) and torch.equal(full["qc_logits"], qcl)
    c = load_ply(p, H, W, crop=True)
    keep = torch.zeros(V, H, W, dtype=torch.bool)
    keep[:, 5:H - 5, 5:W - 5] = True
    assert c["means"].shape[0] == V * (H - 10) * (W - 10) and torch.equal(c["means"], means[keep.reshape(-1)])
    assert torch.equal(c["qc_logits"], qcl[keep.reshape(-1)]) and torch.equal(c["instance_label"], ins.long()[keep.reshape(-1)])
    dc = load_ply(export_ply(means, scales, rot, sh, op, sem, ins, None, tmp_path / "dc.ply", save_sh_dc_only=True), H, W, crop=False)
    assert dc["max_sh_degree"] == 0 and dc["shN"].shape == (G, 0, 3) and dc["qc_logits"].shape[1] == 0
    with pytest.raises(ValueError):
        load_ply(p, 7, 9, crop=True)


def test_cli_preprocessing(tmp_path):
    """reference inference.py:13-38: int() truncation of the long side, centre crop; intrinsics over the crop size"""
    from PIL import Image

    from siu3r_amd.cli_common import normalised_intrinsics, preprocess_image

    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, (667, 1000, 3), dtype=np.uint8)).save(tmp_path / "wide.png")
    Image.fromarray(rng.integers(0, 256, (1000, 667, 3), dtype=np.uint8)).save(tmp_path / "tall.png")
    for name in ("wide.png", "tall.png"):
        t = preprocess_image(tmp_path / name, 256)
        assert t.shape == (3, 256, 256) and t.dtype == torch.float32 and 0.0 <= float(t.min()) and float(t.max()) <= 1.0
    # 1000 x 667 -> long side int(1000 * 256 / 667) = 383 (rounding would give 384): crop offset (383 - 256) // 2 = 63
    img = Image.open(tmp_path / "wide.png").convert("RGB").resize((383, 256), Image.Resampling.LANCZOS).crop((63, 0, 63 + 256, 256))
    assert torch.equal(preprocess_image(tmp_path / "wide.png", 256), torch.from_numpy(np.array(img).astype(np.float32)).permute(2, 0, 1) / 255.0)
    K = normalised_intrinsics(318.0, 318.0, 128.0, 128.0, 3, 256)
    assert K.shape == (1, 3, 3, 3) and abs(float(K[0, 0, 0, 0]) - 318 / 256) < 1e-7 and float(K[0, 2, 0, 2]) == 0.5
    assert float(normalised_intrinsics(636.0, 636.0, 256.0, 256.0, 2, 512)[0, 0, 1, 1]) == float(K[0, 0, 1, 1])


def test_multiview_cli_lists_images_like_the_reference(tmp_path):
    import importlib.util

    from PIL import Image

    spec = importlib.util.spec_from_file_location("inference_multiview", os.path.join(os.path.dirname(GOLDEN), "..", "inference_multiview.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for n in ("b.png", "a.png", "z.jpg", "c.jpeg", "note.txt"):
        if n.endswith("txt"):
            open(tmp_path / n, "w").write("x")
        else:
            Image.new("RGB", (8, 8)).save(tmp_path / n)
    assert [p.name for p in mod.list_images(tmp_path)] == ["z.jpg", "a.png", "b.png", "c.jpeg"]   # *.jpg, then *.png, then *.jpeg


def test_substituted_pairs_are_counted_once_across_ranks():
    """a pair without semantic labels is replaced by the next pair (reference scannet_dataset.py:360-366); under the
    i mod world sharding that next pair has an owner of its own, so the substitute must not be evaluated / counted a second time."""
    from siu3r_amd import scannet

    n, world = 7, 2
    bad = {2, 5}  # unlabeled pairs: reading them yields the next valid pair

    def read(i):
        j = i
        while j % n in bad:
            j += 1
        return {"pair_index": j % n}

    counted = []
    for rank in range(world):
        seen = set()
        mine = scannet.shard(n, rank, world)
        items = scannet.own_items(mine, [read(i) for i in mine], n, seen)
        counted += [it["pair_index"] for it in items]
    assert sorted(counted) == [0, 1, 3, 4, 6], counted  # every valid pair exactly once, on its owner
    # a --limit run: the substitute lies beyond the evaluated range and has no owner -> kept once
    seen = set()
    kept = scannet.own_items([2, 2], [{"pair_index": 3}, {"pair_index": 3}], 3, seen)
    assert [k["pair_index"] for k in kept] == [3]


def test_evaluator_tree_against_reference(tmp_path):
    """tests/golden/evaluator_tree.npz: a synthetic result tree and what the REFERENCE's Evaluator.evaluate returns for it (depth
    quality + mIoU; generated by tests/golden/make_golden_eval.py).  The tree is rebuilt with the product's writers and scored with the
    product's reader / metrics: absrel, rmse (overall and per item), per-class IoUs and mIoU must agree."""
    from siu3r_amd import eval_io as E

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "evaluator_tree.npz"))
    scenes = sorted({k.split(".")[0] for k in z.files if k.startswith("scene")})
    cids, tids = (10, 30), (12, 20, 28)
    for name in scenes:
        dp, dg = z[f"{name}.depth"].astype(np.float64) / 1000.0, z[f"{name}.depth_gt"].astype(np.float64) / 1000.0
        # +0.5 mm: the writer truncates metres * 1000 to int32 (visualizer.py:309-310); the stored values are whole millimetres
        blank = np.zeros((1, len(tids), 3) + dp.shape[1:], np.float32)
        E.save_recon_images(blank, (dp + 0.0005)[None], blank, (dg + 0.0005 * (dg > 0))[None], tmp_path, [name], [cids], [tids])
        for mode, ids in (("context", cids), ("target", tids)):
            ps, pi, gs, gi = (z[f"{name}.{mode}.{k}"].astype(np.int64) for k in ("pred_sem", "pred_ins", "gt_sem", "gt_ins"))
            E.save_seg_ids(mode, ps[None], pi[None], tmp_path, [name], [cids], [tids])
            base = E.scene_dir(tmp_path, name, cids)
            os.makedirs(base / f"{mode}_seg_gt", exist_ok=True)
            from PIL import Image
            for s_, i_, v in zip(gs, gi, ids):
                Image.fromarray(E.encode_segment_ids(s_, i_)).save(base / f"{mode}_seg_gt" / f"{name}_gt{v}.png")
    res = E.evaluate_dir(tmp_path)
    # the reference computes in fp32 (torch.linalg.lstsq, torch.mean); this build in fp64
    assert abs(res["absrel"] - float(z["result.absrel"])) <= 2e-6 * float(z["result.absrel"]) + 1e-8, (res["absrel"], z["result.absrel"])
    assert abs(res["rmse"] - float(z["result.rmse"])) <= 2e-6 * float(z["result.rmse"]) + 1e-8, (res["rmse"], z["result.rmse"])
    for mode in ("context", "target"):
        np.testing.assert_allclose(res[f"{mode}_ious_per_class"], z[f"result.{mode}_ious_per_class"], rtol=1e-6, atol=1e-7)
        assert len(res[f"{mode}_ious_per_class"]) == 20
        assert abs(res[f"{mode}_miou"] - float(z[f"result.{mode}_miou"])) < 1e-6
    items = []
    for d in sorted(p for p in tmp_path.iterdir() if p.is_dir()):
        items += [[it["absrel"], it["rmse"]] for it in json.load(open(d / "depth_scores.json"))]
    np.testing.assert_allclose(np.asarray(items), z["result.depth_items"], rtol=5e-5, atol=1e-7)
    assert os.path.exists(tmp_path / "results.json") and "psnr" in res and "ssim" in res  # (blank renders: psnr inf is a legal value)
