"""CPU tests (-m "not gpu"): the oracle (oracle/) against the golden vectors produced by the REFERENCE's own code
(tests/golden/make_golden.py, run in the build container).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from crafted import crafted_panoptic_inputs, permuted
from golden_utils import (FIELDS, GOLDEN, compare_integer_outputs, compare_summary, default_K, fixture_images, fixture_images_multi,
                          load_model_fixture, load_multi_fixture, segments_match)
from oracle import siu3r_oracle as O
from oracle import weights as OW

_SD = {}


def _weights():
    if "sd" not in _SD:
        _SD["sd"] = OW.make_weights(0)
    return _SD["sd"]


def test_small_ops_against_reference_vectors():
    z = np.load(os.path.join(GOLDEN, "small_ops.npz"))
    out = O.rope2d(torch.from_numpy(z["rope_tok"]), torch.from_numpy(z["rope_pos"]))
    assert float((out - torch.from_numpy(z["rope_out"])).abs().max()) <= 2e-6  # reference RoPE2D (pos_embed.py:126-179)
    raw = torch.from_numpy(z["ga_raw"])
    g = O.gaussian_adapter(torch.zeros(3, 50, 3), raw)
    for k, name in (("ga_cov", "covariances"), ("ga_sh", "harmonics"), ("ga_op", "opacities"), ("ga_scale", "scales"), ("ga_rot", "rotations")):
        ref = torch.from_numpy(z[k])
        assert float((g[name] - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max())), name


def test_projection_helpers_against_reference_vectors():
    """get_fov / get_projection_matrix are host-side parameter code of the PRODUCT's renderer front-end; they contain no
    GPU work, so they are pinned here against the reference's values (utils/projection.py:247-261, cuda_splatting.py:16-43)."""
    from siu3r_amd import cuda_splatting as cs

    z = np.load(os.path.join(GOLDEN, "small_ops.npz"))
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None]
    fov = cs.get_fov(K)
    assert np.allclose(fov.numpy(), z["fov"], atol=1e-6)
    proj = cs.get_projection_matrix(torch.tensor([1.0]), torch.tensor([1000.0]), fov[:, 0], fov[:, 1])
    assert np.allclose(proj.numpy(), z["proj"], atol=1e-6)


@pytest.mark.parametrize("oname,order", [("natural", [0, 1, 2, 3]), ("quirk_first", [2, 0, 3, 1])])
def test_panoptic_postprocess_against_reference_vectors(oname, order):
    z = np.load(os.path.join(GOLDEN, "panoptic_crafted.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "panoptic_crafted.json")))[oname]
    cls, msk = permuted(*crafted_panoptic_inputs(), order)
    res = O.panoptic_postprocess(cls, msk, (64, 64))
    for b, r in enumerate(res):
        seg = z[f"{oname}.{b}.segmentation"]
        assert str(r["segmentation"].dtype) == meta[b]["dtype"]
        assert np.array_equal(r["segmentation"].numpy(), seg)
        assert r["segments_info"] == meta[b]["segments_info"] and r["query_scores"] == meta[b]["query_scores"]
        assert list(r["query_class_logits"].shape) == list(z[f"{oname}.{b}.qcl_shape"])
        assert np.array_equal(r["query_class_logits"].reshape(-1)[::97].numpy(), z[f"{oname}.{b}.qcl_sample"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_lifting_against_reference_vectors(case):
    meta = json.load(open(os.path.join(GOLDEN, "lifting.json")))[case]
    z = np.load(os.path.join(GOLDEN, f"lifting_{case}.npz"))
    sem, ins, info = O.lift_ids(torch.from_numpy(z["x"]), meta["scores"])
    assert np.array_equal(sem.numpy(), z["sem_id"]) and np.array_equal(ins.numpy(), z["ins_id"])
    assert info == meta["info"]


@pytest.mark.parametrize("size", [256, 512])
def test_model_forward_against_reference_vectors(size):
    """Full SIU3RModel.forward of the reference (synthetic weights, asset pair) vs the oracle: strided samples + norms."""
    z, meta = load_model_fixture(size)
    with torch.no_grad():
        out = O.model_forward(_weights(), fixture_images(size), default_K(), keep_intermediates=False)
    for f in FIELDS:
        compare_summary(f, out[f], z, 2e-4)
    compare_summary("class_queries_logits", out["class_queries_logits"], z, 2e-4)
    compare_summary("masks_queries_logits", out["masks_queries_logits"], z, 2e-4)
    segments_match(out["seg_infos"], meta["seg_infos"], 3e-6)  # scores: fp32 softmax values the reference rounds to 6 decimals
    assert np.allclose(out["query_scores"][0], meta["query_scores"][0], atol=3e-6)
    assert str(out["seg_masks"][0].dtype) == str(z["seg_mask.dtype"])
    assert np.array_equal(out["seg_masks"][0].unique().numpy(), z["seg_mask.unique"])
    # the synthetic weights are shaped so that the panoptic branch is NOT empty (siu3r_amd/synthetic_weights.py): several accepted
    # segments incl. stuff queries sharing one fused id, and the integer maps / lifted logit volume are pinned too
    infos = meta["seg_infos"][0]
    assert len(infos) >= 4 and any(i["was_fused"] for i in infos) and len({i["id"] for i in infos}) < len(infos)
    qcl = out["query_class_logits"][0]
    qcl_gm = qcl.permute(0, 3, 4, 1, 2).reshape(-1, qcl.shape[1], qcl.shape[2])  # 'n q c h w -> (n h w) q c' (model.py:261-263)
    # (the reference and this restatement evaluate the mask logits in different fp32 orders: at 512^2 two border pixels change owner)
    compare_integer_outputs(out["semantic_labels"], out["instance_labels"], out["seg_masks"][0], qcl_gm, z, 2e-5, min_agree=0.9999)


@pytest.mark.parametrize("V,size", [(3, 128), (8, 256)], ids=["v3_128", "v8_256"])
def test_multiview_forward_against_reference_vectors(V, size):
    """SIU3RMultiViewModel.forward of the reference vs oracle.model_forward_multi: V = 3 at 128^2, and V = 8 at 256^2 -- the network half
    of BASELINE configs[4] (8 views), pinned to the reference's own model_multi.py (20 s of its CPU time at 256^2; the 512^2 run takes
    74 s and 19.5 TFLOP).  Fields and logits (samples, dense windows, norms), the segment table, and the integer outputs of the panoptic
    branch: strided samples + whole-map histograms of the id maps, samples of the query x class logit volume."""
    z, meta = load_multi_fixture(V, size)
    with torch.no_grad():
        out = O.model_forward_multi(_weights(), fixture_images_multi(size, V), default_K(1, V), keep_intermediates=False)
    for f in FIELDS + ("class_queries_logits", "masks_queries_logits"):
        compare_summary(f, out[f], z, 2e-4)
    segments_match(out["seg_infos"], meta["seg_infos"], 3e-6)
    assert np.allclose(out["query_scores"][0], meta["query_scores"][0], atol=3e-6) and len(meta["seg_infos"][0]) >= 3
    qcl = out["query_class_logits"][0]
    qcl_gm = qcl.permute(0, 3, 4, 1, 2).reshape(-1, qcl.shape[1], qcl.shape[2])  # 'n q c h w -> (n h w) q c' (model.py:261-263)
    # (two fp32 evaluation orders of the mask logits: a handful of border pixels may change owner, as in the two-view fixtures)
    compare_integer_outputs(out["semantic_labels"], out["instance_labels"], out["seg_masks"][0], qcl_gm, z, 2e-5, min_agree=0.9999)


def test_sh_basis_forms_agree():
    """The two SH evaluations of the raster oracle -- explicit 3DGS-style polynomials (K2 path, raster_ref.c sh_to_rgb) and
    Sloan's recurrences (gsplat path, raster_ref_sh_eval) -- are the same real basis with the same sign convention
    (odd |m| negative) up to degree 4: one-hot coefficient probes through both give the same colours."""
    import ctypes as C

    from oracle import raster_oracle as RO

    rng = np.random.default_rng(0)
    G = 64
    means = rng.normal(size=(G, 3)).astype(np.float32) + np.array([0, 0, 4], np.float32)
    campos = np.array([0.1, -0.2, 0.3], np.float32)
    cov6 = np.tile(np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32), (G, 1))
    from siu3r_amd._lib import RasterCam  # plain ctypes struct (no GPU needed)

    cam = RasterCam()
    cam.mode, cam.width, cam.height, cam.sh_degree, cam.sh_band4 = 0, 64, 64, 4, 1
    for i, v in enumerate([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]):
        cam.w2c[i] = v
        cam.proj[i] = v
    cam.tanfovx = cam.tanfovy = 0.5
    cam.k2_znear_cull, cam.dilation = 0.2, 0.3
    for i in range(3):
        cam.campos[i] = float(campos[i])
    for k in range(25):
        sh = np.zeros((G, 25, 3), np.float32)
        sh[:, k, :] = 1.0
        a = RO.sh_eval(4, means, campos, sh)
        out = np.zeros((G, 3), np.float32)
        # the K2 evaluation is reachable through the full forward only: per-Gaussian colours land in the image for isolated splats;
        # compare the bases directly instead through the library's exported helper when present
        fn = getattr(RO.lib(), "raster_ref_sh_to_rgb", None)
        if fn is None:
            pytest.skip("oracle library without the exported K2 SH helper")
        for g_ in range(G):
            fn(C.byref(cam), means[g_].ctypes.data_as(C.c_void_p), sh[g_].ctypes.data_as(C.c_void_p), out[g_].ctypes.data_as(C.c_void_p))
        assert np.abs(a - out).max() <= 2e-6, (k, np.abs(a - out).max())
