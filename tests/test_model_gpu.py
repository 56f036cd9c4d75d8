"""End-to-end parity of the HIP SIU3R forward (through libsiu3r_hip.so) against the CPU oracle on the same
seeded synthetic weights and inputs.

Tolerances (documented in DESIGN.md):
  precision="bf16x3": 1e-3 (north-star bar: "within 1e-3 rel on fp32"), metric = max|err| / max|ref| per tensor
  precision="bf16"  : 6e-2 max-normalised / 3e-2 rel-L2 -- bf16 operands cannot meet 1e-3 (the reference itself
                      moves by ~1e-2 under bf16 autocast); reported, not hidden.
Integer outputs (segmentation / semantic / instance ids) must be bit-exact in both modes.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

_STATE = {}


def _setup(H, W, B=1):
    key = (H, W, B)
    if key in _STATE:
        return _STATE[key]
    from oracle import siu3r_oracle as O
    from oracle import weights as OW

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    sd = _STATE["sd"]
    g = torch.Generator().manual_seed(7)
    img = torch.rand(B, 2, 3, H, W, generator=g)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, 2, 1, 1)
    with torch.no_grad():
        ref = O.model_forward(sd, img, K)
    _STATE[key] = (sd, img, K, ref)
    return _STATE[key]


ELEMENTWISE_FRAC_TOL, ELEMENTWISE_WORST_TOL = 0.10, 0.25  # see _report


def _errs(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    mx = ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()
    l2 = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    return mx, l2


def _report(name, got, ref, tol_max, tol_l2, fails):
    """max-normalised error and relative L2 of the difference (the asserted pair), and beside them the ELEMENT-WISE relative error
    |err| / max(|ref|, 1e-3 * absmax): its worst value and the fraction of elements above 1e-3 (golden_utils.elementwise_rel)"""
    from golden_utils import elementwise_rel

    mx, l2 = _errs(got, ref)
    r = ref.detach().float().cpu()
    rel, frac = elementwise_rel(got.detach().float().cpu(), r, float(r.abs().max()))
    ok = math.isfinite(mx) and mx <= tol_max and l2 <= tol_l2
    # the tail, asserted at what the data supports (round 6; 147 bf16x3 comparisons of one GPU run: at most 6.6 % of a tensor's elements
    # above 1e-3 -- covariances and rotations, whose near-zero entries sit a thousand times below the tensor's range -- worst element
    # 9.5e-2): in the 1e-3 mode at least 90 % of the elements are within 1e-3 element-wise and none is off by more than 0.25
    if tol_max <= 1e-3:
        ok = ok and frac <= ELEMENTWISE_FRAC_TOL and rel <= ELEMENTWISE_WORST_TOL
    print(f"[model-parity] {name:34s} max_norm_err={mx:.3e} rel_l2={l2:.3e} elementwise_rel worst={rel:.3e} (>1e-3: {frac:.3%}) {'ok' if ok else 'FAIL'}")
    if not ok:
        fails.append((name, mx, l2))


def test_lateral_of_f1_without_f1():
    """CroCoViTAdapter.finish composes pixel_decoder.adapter_1.0(f1) from f1's linear ingredients (resize of the projected tokens +
    a 1024 -> 256 conv-transpose + a 64 -> 256 map of the spatial prior's stem) and never writes the 1024-channel maps; with
    return_intermediates f1 is materialised the reference's way (vit_adapter.py:420-436) as well: the lateral convolution of THAT f1
    must be what the composed route hands to the pixel decoder."""
    from siu3r_amd import ops
    from siu3r_amd.model import CroCoViTAdapter

    model, _, ref = _run("bf16x3", 128, 128)
    last = model._last
    f1, lat1 = last["ms"][0], last["lat1"]
    assert f1 is not None and f1.shape[-1] == 1024 and lat1.shape == (*f1.shape[:3], 256)
    want = ops.linear(f1, model._ctx.w.linear(CroCoViTAdapter.LATERAL), out_dtype=torch.float32)
    err = float((lat1 - want).abs().max() / want.abs().max())
    print(f"[lateral] composed vs lateral(f1): max-normalised {err:.3e}")
    assert err <= 1e-4, err
    # and the production path (graphs, no intermediates) carries no f1 at all
    sd, img, K, _ = _setup(128, 128, 1)
    for _ in range(3):
        model(img.cuda(), K.cuda())
    ent = next(iter(model._graphs.values()))
    assert ent["st"].ms[0] is None and ent["st"].adapter["lat1"] is not None


def _run(precision, H, W, B=1):
    from siu3r_amd.model import SIU3RModel

    sd, img, K, ref = _setup(H, W, B)
    model = SIU3RModel(sd, image_size=(H, W), precision=precision)
    with torch.no_grad():
        g, seg, masks, infos, qs = model(img.cuda(), K.cuda(), enable_query_class_logit_lift=True, return_intermediates=True)
    torch.cuda.synchronize()
    return model, (g, seg, masks, infos, qs), ref


@pytest.mark.parametrize("precision,tol_max,tol_l2", [("bf16x3", 1e-3, 1e-3), ("bf16", 6e-2, 3e-2)])
@pytest.mark.parametrize("size", [(128, 128), (256, 192)], ids=["128", "256x192"])
def test_forward_parity(precision, tol_max, tol_l2, size):
    H, W = size
    model, (g, seg, masks, infos, qs), ref = _run(precision, H, W)
    fails = []
    last = model._last
    bb = ref["bb"]
    for i in (0, 5, 11, 23):
        _report(f"enc_block{i}.view1", last["all_feat1"][i], bb["all_feat1"][i], tol_max, tol_l2, fails)
    for i in (0, 1, 6, 12):
        _report(f"dec1[{i}]", last["dec1"][i], bb["dec1"][i], tol_max, tol_l2, fails)
        _report(f"dec2[{i}]", last["dec2"][i], bb["dec2"][i], tol_max, tol_l2, fails)
    for lvl in range(4):
        got = last["ms"][lvl].view(1, 2, *last["ms"][lvl].shape[1:])
        _report(f"adapter.f{lvl+1}.view1", got[:, 0].permute(0, 3, 1, 2), ref["ms1"][lvl], tol_max, tol_l2, fails)
        _report(f"adapter.f{lvl+1}.view2", got[:, 1].permute(0, 3, 1, 2), ref["ms2"][lvl], tol_max, tol_l2, fails)
    _report("pts3d.view1", last["pts1"], ref["pts1"], tol_max, tol_l2, fails)
    _report("pts3d.view2", last["pts2"], ref["pts2"], tol_max, tol_l2, fails)
    _report("gs_raw.view1", last["gs_raw1"].flatten(1, 2), ref["gs_raw1"], tol_max, tol_l2, fails)
    so = last["seg_out"]
    _report("mask_features", so["_mask_features"].permute(0, 3, 1, 2), ref["mask_features"], tol_max, tol_l2, fails)
    for i in range(3):
        _report(f"pixel_decoder.ms{i}", so["_ms"][i].permute(0, 3, 1, 2), ref["ms"][i], tol_max, tol_l2, fails)
    for f in ("means", "covariances", "harmonics", "opacities", "scales", "rotations"):
        # covariances are quadratic in the scales (exp of a bf16 logit): their max-normalised error is an outlier
        # statistic that moves between 0.08 and 0.16 with any rounding-order change; the bf16-mode max bound is
        # 0.2 for them (rel-L2 keeps the common 3e-2 bound).  bf16x3 mode keeps 1e-3.
        tm = 0.2 if (f == "covariances" and precision == "bf16") else tol_max
        _report(f"gaussians.{f}", getattr(g, f), ref[f], tm, tol_l2, fails)
    _report("class_queries_logits", seg.class_queries_logits, ref["class_queries_logits"], tol_max, tol_l2, fails)
    _report("masks_queries_logits", seg.masks_queries_logits, ref["masks_queries_logits"], tol_max, tol_l2, fails)
    # integer / structural outputs.  The shaped synthetic weights give a NON-EMPTY panoptic result (several segments, fused stuff
    # ids): the ids are an argmax over fp32 scores, so a border pixel may change owner when the logits move by 1e-4 (bf16x3) or
    # 1e-2 (bf16); the segment table must match (bf16x3) and the maps must agree almost everywhere.  Bit-exactness of the integer
    # kernels on identical inputs: tests/test_postprocess_gpu.py
    from golden_utils import labels_agree, segments_match

    assert len(ref["seg_infos"][0]) >= 3, "the synthetic weights should keep some queries"
    x3 = precision == "bf16x3"
    strip = lambda segs: [(s_["id"], s_["label_id"], s_["was_fused"]) for s_ in segs]
    same_table = [strip(i) for i in infos] == [strip(i) for i in ref["seg_infos"]]
    print("[model-parity] segments:", [strip(i) for i in infos], "oracle:", [strip(i) for i in ref["seg_infos"]])
    if x3:
        segments_match(infos, ref["seg_infos"], 1e-3)
        assert [len(q) for q in qs] == [len(q) for q in ref["query_scores"]]
    frac = (0.999 if x3 else 0.85) if same_table else 0.5  # bf16: measured 0.91-0.96 (noise-like synthetic masks: long borders)
    labels_agree("semantic_labels", g.semantic_labels, ref["semantic_labels"], frac)
    # (bf16 only: a borderline query accepted or dropped renumbers every later segment id -- the id maps are then not comparable)
    labels_agree("instance_labels", g.instance_labels, ref["instance_labels"], frac if same_table else 0.0)
    for a, b in zip(masks, ref["seg_masks"]):
        assert a.dtype == b.dtype
        labels_agree("segmentation", a, b, frac if same_table else 0.0)
    if same_table:
        for a, b in zip(g.seg_query_class_logits, ref["query_class_logits"]):
            bb_ = b.permute(0, 3, 4, 1, 2).reshape(-1, b.shape[1], b.shape[2])
            assert a.shape == bb_.shape
            # (bf16 only: the lifted logits are mask-probability x class-score products at full resolution -- 6.7e-2 / 6.2e-2 measured)
            _report("query_class_logits", a, bb_, tol_max if x3 else 1e-1, tol_l2 if x3 else 1e-1, fails)
    assert not fails, f"parity failures: {fails}"


@pytest.mark.parametrize("precision,tol", [("bf16x3", 1e-3), ("bf16", 6e-2)])
@pytest.mark.parametrize("size", [256, 512])
def test_forward_against_reference_golden(precision, tol, size):
    """HIP forward at the BASELINE sizes (256^2 shipped scripts, 512^2 benchmark) against the golden vectors produced by
    the REFERENCE's own forward (tests/golden/make_golden.py): strided samples + L2 norms of every Gaussian field and of
    the Mask2Former logits; integer label checksums and segment lists exact."""
    from golden_utils import FIELDS, compare_integer_outputs, compare_summary, default_K, fixture_images, load_model_fixture, segments_match
    from oracle import weights as OW
    from siu3r_amd.model import SIU3RModel

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    z, meta = load_model_fixture(size)
    model = SIU3RModel(_STATE["sd"], image_size=(size, size), precision=precision)
    with torch.no_grad():
        g, seg, masks, infos, qs = model(fixture_images(size).cuda(), default_K().cuda(), enable_query_class_logit_lift=True)
    torch.cuda.synchronize()
    for f in FIELDS:
        compare_summary(f, getattr(g, f), z, tol * (2 if (f == "covariances" and precision == "bf16") else 1))
    compare_summary("class_queries_logits", seg.class_queries_logits, z, tol)
    compare_summary("masks_queries_logits", seg.masks_queries_logits, z, tol)
    assert str(masks[0].dtype) == str(z["seg_mask.dtype"])
    # the panoptic branch is non-empty with the shaped synthetic weights (>= 4 segments, fused stuff ids): the segment table, the id maps
    # and the lifted query x class logit volume are pinned to the reference's output
    assert len(meta["seg_infos"][0]) >= 4 and any(i["was_fused"] for i in meta["seg_infos"][0])
    if precision == "bf16x3":
        segments_match(infos, meta["seg_infos"], 2e-6 + tol * 0.05)
        # (border pixels may change owner: the mask logits differ from the reference's by ~1e-5 relative)
        compare_integer_outputs(g.semantic_labels, g.instance_labels, masks[0], g.seg_query_class_logits[0], z, 1e-3, min_agree=0.999)
    else:
        # bf16 operands move the mask logits by ~1e-2: pixels on a segment border may change owner and a borderline query may be
        # accepted or dropped; the labelled area must still agree almost everywhere
        print("[golden] bf16 segments:", [(i["id"], i["label_id"]) for i in infos[0]], "reference:", [(i["id"], i["label_id"]) for i in meta["seg_infos"][0]])
        same_table = [(i["id"], i["label_id"], i["was_fused"]) for i in infos[0]] == [(i["id"], i["label_id"], i["was_fused"]) for i in meta["seg_infos"][0]]
        compare_integer_outputs(g.semantic_labels, g.instance_labels, masks[0], g.seg_query_class_logits[0] if same_table else None, z, 0.2,
                                min_agree=0.93 if same_table else 0.5)  # measured 0.956 (the synthetic masks are noise-like: long borders)
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision,tol", [("bf16x3", 1e-3), ("bf16", 6e-2)])
def test_multiview_forward(precision, tol):
    """SIU3RMultiViewModel (V = 3, 128^2): against the golden vectors of the reference's model_multi.py forward, and
    per-view against oracle.model_forward_multi; then the same model replayed through its HIP graphs, and B = 2."""
    from golden_utils import FIELDS, compare_summary, default_K, fixture_images_multi, load_multi_fixture
    from oracle import siu3r_oracle as O
    from oracle import weights as OW
    from siu3r_amd.model import SIU3RMultiViewModel

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    sd = _STATE["sd"]
    z, meta = load_multi_fixture()
    img, K = fixture_images_multi(128), default_K(1, 3)
    model = SIU3RMultiViewModel(sd, image_size=(128, 128), precision=precision)
    ctol = lambda f: tol * (3.4 if (f == "covariances" and precision == "bf16") else 1)
    with torch.no_grad():
        g, seg, masks, infos, qs = model(img.cuda(), K.cuda(), enable_query_class_logit_lift=True, return_intermediates=True)
        ref = O.model_forward_multi(sd, img, K)
    for f in FIELDS:
        compare_summary(f, getattr(g, f), z, ctol(f))
    compare_summary("class_queries_logits", seg.class_queries_logits, z, tol)
    compare_summary("masks_queries_logits", seg.masks_queries_logits, z, tol)
    from golden_utils import labels_agree, segments_match

    x3 = precision == "bf16x3"
    table = lambda segs: [[(s_["id"], s_["label_id"], s_["was_fused"]) for s_ in i] for i in segs]
    same_table = table(infos) == table(meta["seg_infos"])
    assert len(meta["seg_infos"][0]) >= 3
    if x3:
        segments_match(infos, meta["seg_infos"], 1e-3)
        # the integer outputs against the reference's own (round 5: strided samples + whole-map histograms, like the two-view fixtures)
        from golden_utils import compare_integer_outputs
        compare_integer_outputs(g.semantic_labels, g.instance_labels, masks[0], g.seg_query_class_logits[0], z, 1e-3, min_agree=0.999)
    frac = (0.999 if x3 else 0.85) if same_table else 0.5  # bf16: measured 0.91-0.96 (noise-like synthetic masks: long borders)
    fails = []
    for v in range(3):
        for i in (1, 6, 12):
            _report(f"multi.dec[v{v}][{i}]", model._last["decs"][v][i], ref["bb"]["decs"][v][i], tol, tol, fails)
    _report("multi.means", g.means, ref["means"], tol, tol, fails)
    _report("multi.harmonics", g.harmonics, ref["harmonics"], tol, tol, fails)
    labels_agree("multi.semantic_labels", g.semantic_labels, ref["semantic_labels"], frac)
    labels_agree("multi.instance_labels", g.instance_labels, ref["instance_labels"], frac)
    assert not fails, fails
    # graph replay (third call of the shape) reproduces the eager result bit for bit
    with torch.no_grad():
        outs = [model(img.cuda(), K.cuda(), enable_query_class_logit_lift=True) for _ in range(3)]
    for o in outs:
        assert torch.equal(o[0].means, g.means) and torch.equal(o[0].harmonics, g.harmonics) and torch.equal(o[1].class_queries_logits, seg.class_queries_logits)
    # B = 2 (views 1.. do not form one strided batch: the copy path): item 0 = the fixture, item 1 = its views permuted
    img2 = torch.cat((img, img[:, [1, 2, 0]]), 0)
    with torch.no_grad():
        g2 = model(img2.cuda(), default_K(2, 3).cuda())[0]
    # (the launch geometry depends on the row count -- split-K slices, tile walk -- so fp32 sums differ in their last bits between batch
    # sizes; bf16 activations turn such a bit into a rounding flip, i.e. into bf16-level differences)
    assert float((g2.means[0] - g.means[0]).abs().max()) <= (5e-4 if x3 else 3e-2) * float(g.means.abs().max())
    del model
    torch.cuda.empty_cache()


def test_multiview_eight_views_and_batched_views_against_oracle():
    """configs[4]'s view count at a size the CPU oracle finishes in seconds: V = 8 @64^2 (every field of all eight views), and the
    B = 2, V = 3 copy path (views 1.. of several items are not one strided batch) with ALL fields -- against oracle.model_forward_multi,
    which is pinned to the reference's SIU3RMultiViewModel at V = 3."""
    from golden_utils import default_K
    from oracle import siu3r_oracle as O
    from oracle import weights as OW
    from siu3r_amd.model import SIU3RMultiViewModel

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    sd = _STATE["sd"]
    for (B, V, S, seed) in ((1, 8, 64, 31), (2, 3, 96, 32)):
        g_ = torch.Generator().manual_seed(seed)
        img = torch.rand(B, V, 3, S, S, generator=g_)
        K = default_K(B, V)
        model = SIU3RMultiViewModel(sd, image_size=(S, S), precision="bf16x3")
        with torch.no_grad():
            ref = O.model_forward_multi(sd, img, K, keep_intermediates=False)
            outs = [model(img.cuda(), K.cuda(), enable_query_class_logit_lift=True) for _ in range(3)]  # eager, capture, replay
        g, seg = outs[2][0], outs[2][1]
        assert g.means.shape == (B, V * S * S, 3)
        fails = []
        for f in ("means", "covariances", "harmonics", "opacities", "scales", "rotations"):
            _report(f"multi B={B} V={V} {f}", getattr(g, f), ref[f], 1e-3, 1e-3, fails)
        # (the logits pass through nine thresholded attention masks: a borderline pixel may flip between fp32 evaluation orders)
        # (the smallest bound that holds: 5e-3, the thresholded-mask caveat of test_parity_sweep; the measured value is printed)
        _report(f"multi B={B} V={V} class logits", seg.class_queries_logits, ref["class_queries_logits"], 5e-3, 5e-3, fails)
        _report(f"multi B={B} V={V} mask logits", seg.masks_queries_logits, ref["masks_queries_logits"], 5e-3, 5e-3, fails)
        assert not fails, fails
        if max(_errs(seg.class_queries_logits, ref["class_queries_logits"])[0], _errs(seg.masks_queries_logits, ref["masks_queries_logits"])[0]) > 1e-3:
            # above 1e-3 only through flipped attention-mask pixels: with the oracle's masks forced in, within 1e-3 (see test_parity_sweep)
            model.use_graph = False
            model.mask2former.forced_attn_masks = ref["attn_masks"]
            with torch.no_grad():
                seg_f = model(img.cuda(), K.cuda())[1]
            model.mask2former.forced_attn_masks = None
            model.use_graph = True
            _report(f"multi B={B} V={V} class logits (oracle masks forced)", seg_f.class_queries_logits, ref["class_queries_logits"], 1e-3, 1e-3, fails)
            _report(f"multi B={B} V={V} mask logits (oracle masks forced)", seg_f.masks_queries_logits, ref["masks_queries_logits"], 1e-3, 1e-3, fails)
            assert not fails, fails
        assert torch.equal(outs[0][0].means, g.means) and torch.equal(outs[0][1].masks_queries_logits, seg.masks_queries_logits)
        agree = float((g.semantic_labels.cpu() == ref["semantic_labels"]).float().mean())
        print(f"[multi] B={B} V={V}: semantic label agreement {agree:.5f}, segments {[len(i) for i in outs[2][3]]}")
        assert agree >= 0.99
        del model
        torch.cuda.empty_cache()


def test_multiview_eight_views_256_against_reference_golden():
    """BASELINE configs[4]'s network half -- SIU3RMultiViewModel on EIGHT views -- against golden vectors of the reference's own
    model_multi.py forward at 256^2 (tests/golden/model_multi_v8_256.npz; 524 288 Gaussians, 9 segments): every Gaussian field and both
    logit tensors (samples, dense windows, norms) within 1e-3 max-normalised (the logits 5e-3: the thresholded-mask caveat, and when above
    1e-3 the flipped attention-mask pixels are counted and shown to be the whole difference), the segment table, and the id maps /
    lifted logit volume by samples + histograms.  (The 512^2 shape of configs[4] is exercised for finiteness / sizes in
    tests/test_configs_gpu.py; its network half differs from this test only in the token count.)"""
    from golden_utils import FIELDS, compare_integer_outputs, compare_summary, default_K, fixture_images_multi, load_multi_fixture, segments_match
    from oracle import weights as OW
    from siu3r_amd.model import SIU3RMultiViewModel

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    V, S = 8, 256
    z, meta = load_multi_fixture(V, S)
    img, K = fixture_images_multi(S, V), default_K(1, V)
    model = SIU3RMultiViewModel(_STATE["sd"], image_size=(S, S), precision="bf16x3")
    with torch.no_grad():
        outs = [model(img.cuda(), K.cuda(), enable_query_class_logit_lift=True) for _ in range(3)]  # eager, capture, replay
    torch.cuda.synchronize()
    g, seg, masks, infos, qs = outs[2]
    assert g.means.shape == (1, V * S * S, 3) and len(meta["seg_infos"][0]) >= 4
    for f in FIELDS:
        compare_summary(f, getattr(g, f), z, 1e-3)
    e_c = compare_summary("class_queries_logits", seg.class_queries_logits, z, 5e-3)
    e_m = compare_summary("masks_queries_logits", seg.masks_queries_logits, z, 5e-3)
    print(f"[golden] V=8 @256^2 logits: class {e_c:.3e}, mask {e_m:.3e} (1e-3 unless a thresholded attention-mask pixel flipped)")
    segments_match(infos, meta["seg_infos"], 1e-3)
    compare_integer_outputs(g.semantic_labels, g.instance_labels, masks[0], g.seg_query_class_logits[0], z, 2e-3, min_agree=0.998)  # (9 segments over 8 views of noise-like masks: measured 0.9989)
    assert torch.equal(outs[0][0].means, g.means) and torch.equal(outs[0][1].masks_queries_logits, seg.masks_queries_logits)  # replay == eager
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_inference_cli_writes_ply(tmp_path, precision):
    """inference.py (reference inference.py:41-150 counterpart) end to end: two image files -> output.ply with the
    reference's vertex schema and one vertex per pixel of both views."""
    import subprocess
    import sys

    import numpy as np
    from PIL import Image

    from golden_utils import fixture_images
    from siu3r_amd.ply_export import read_ply_vertices

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pair = (fixture_images(256)[0] * 255).round().byte().permute(0, 2, 3, 1).numpy()
    for i in (0, 1):
        Image.fromarray(np.ascontiguousarray(pair[i])).resize((320, 288)).save(tmp_path / f"v{i}.png")
    out = subprocess.run([sys.executable, os.path.join(root, "inference.py"), "--image_path1", str(tmp_path / "v0.png"), "--image_path2", str(tmp_path / "v1.png"),
                          "--output_path", str(tmp_path / "out"), "--precision", precision], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout + out.stderr
    v = read_ply_vertices(tmp_path / "out" / "output.ply")
    assert len(v) == 2 * 256 * 256 == 131072  # the default --size 256 (reference inference.py:13-38: center crop + resize to 256): configs[0]
    names = v.dtype.names
    assert names[:9] == ("x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2") and "semantic_label" in names and "rot_3" in names
    assert np.isfinite(v["x"]).all() and np.isfinite(v["opacity"]).all() and (v["opacity"] >= 0).all() and (v["opacity"] <= 1).all()


def test_graph_replay_with_new_inputs_equals_eager():
    """The captured chains must not keep anything from the input they were captured on: replaying with a second and a third input gives
    bit for bit what a graph-free model computes on those inputs (every logit, every label)."""
    from oracle import weights as OW
    from siu3r_amd.model import SIU3RModel

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    g = torch.Generator().manual_seed(21)
    imgs = torch.rand(4, 1, 2, 3, 256, 256, generator=g).cuda()
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).cuda()
    model = SIU3RModel(_STATE["sd"], image_size=(256, 256), precision="bf16x3")
    plain = SIU3RModel(_STATE["sd"], image_size=(256, 256), precision="bf16x3")
    plain.use_graph = False
    with torch.no_grad():
        for n in range(4):  # eager, capture on input 1, replays on inputs 2 and 3
            a = model(imgs[n], K, enable_query_class_logit_lift=True)
            b = plain(imgs[n], K, enable_query_class_logit_lift=True)
            torch.cuda.synchronize()
            for f in ("means", "covariances", "harmonics", "opacities", "semantic_labels", "instance_labels"):
                assert torch.equal(getattr(a[0], f), getattr(b[0], f)), (n, f)
            assert torch.equal(a[1].class_queries_logits, b[1].class_queries_logits), n
            assert torch.equal(a[1].masks_queries_logits, b[1].masks_queries_logits), n
            assert a[3] == b[3], n
    del model, plain
    torch.cuda.empty_cache()


def test_presplit_activations_do_not_change_a_bit():
    """The encoder's pre-split bf16x3 activations (ops.Planes: proj / fc2 write the residual stream also as hi | lo planes, fc1 writes its
    GELU output as planes only; QKV / fc1 / fc2 read planes) are an operand FORMAT, not an approximation: the forward with them gives the
    bits of the forward without them, and they are really in use (the plan log shows the pre-split kernel instantiation)."""
    from oracle import weights as OW
    from siu3r_amd import ops
    from siu3r_amd.model import SIU3RModel

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    g = torch.Generator().manual_seed(23)
    img = torch.rand(1, 2, 3, 512, 512, generator=g).cuda()
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).cuda()
    outs, kernels = [], []
    saved = ops._NO_PRESPLIT
    try:
        for off in (False, True):
            ops._NO_PRESPLIT = off
            m = SIU3RModel(_STATE["sd"], image_size=(512, 512), precision="bf16x3")
            m.use_graph = False
            log = []
            ops.set_plan_log(log)
            with torch.no_grad():
                outs.append(m(img, K, enable_query_class_logit_lift=True))
            torch.cuda.synchronize()
            ops.set_plan_log(None)
            kernels.append(sum(1 for pl in log if b"gemm_pp_kernel<true" in pl.kernel and pl.kernel.count(b",") == 6 and pl.kernel.endswith(b", true>")))  # 7th template argument = pre-split A
            del m
    finally:
        ops._NO_PRESPLIT = saved
        ops.set_plan_log(None)
    a, b = outs
    for f in ("means", "covariances", "harmonics", "opacities", "semantic_labels", "instance_labels"):
        assert torch.equal(getattr(a[0], f), getattr(b[0], f)), f
    assert torch.equal(a[1].class_queries_logits, b[1].class_queries_logits) and torch.equal(a[1].masks_queries_logits, b[1].masks_queries_logits)
    print(f"[presplit] launches on the pre-split instantiation: {kernels[0]} with planes, {kernels[1]} without")
    assert kernels[0] >= 24 and kernels[1] == 0, kernels
    torch.cuda.empty_cache()


def test_forward_async_interleaved():
    """forward_async(): three steps pending at once over three slots, results picked up out of phase with the submissions, every slot
    used for its eager / capture / replay rounds -- each step's result must be bit for bit what a graph-free synchronous forward()
    computes on that step's input (the slots share nothing but the weights)."""
    from oracle import weights as OW
    from siu3r_amd.model import SIU3RModel

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    g = torch.Generator().manual_seed(22)
    n_in = 9
    imgs = torch.rand(n_in, 1, 2, 3, 256, 256, generator=g).cuda()
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).cuda()
    model = SIU3RModel(_STATE["sd"], image_size=(256, 256), precision="bf16x3")
    model.pipeline_depth = 3
    plain = SIU3RModel(_STATE["sd"], image_size=(256, 256), precision="bf16x3")
    plain.use_graph = False
    import collections
    pend, got = collections.deque(), []
    with torch.no_grad():
        for n in range(n_in):
            buf = imgs[n].clone()
            pend.append(model.forward_async(buf, K, enable_query_class_logit_lift=True))
            buf.fill_(0.5)  # the input was copied into the slot on entry (stream-ordered): the caller may overwrite it
            if len(pend) == 3:
                got.append(pend.popleft().result())
        while pend:
            got.append(pend.popleft().result())
        with pytest.raises(RuntimeError):  # a fourth pending step has no slot
            hs = [model.forward_async(imgs[0], K) for _ in range(4)]
        for h in model._inflight.values():
            h.result()
        torch.cuda.synchronize()
        for n in range(n_in):
            a, b = got[n], plain(imgs[n], K, enable_query_class_logit_lift=True)
            torch.cuda.synchronize()
            for f in ("means", "covariances", "harmonics", "opacities", "semantic_labels", "instance_labels"):
                assert torch.equal(getattr(a[0], f), getattr(b[0], f)), (n, f)
            assert torch.equal(a[1].class_queries_logits, b[1].class_queries_logits), n
            assert torch.equal(a[1].masks_queries_logits, b[1].masks_queries_logits), n
            assert a[3] == b[3] and a[4] == b[4], n
            assert all(torch.equal(x, y) for x, y in zip(a[0].seg_query_class_logits, b[0].seg_query_class_logits)), n
        # a synchronous forward() after the pipeline drained uses slot 0 again
        a, b = model(imgs[3], K), got[3]
        assert torch.equal(a[0].means, b[0].means) and torch.equal(a[1].masks_queries_logits, b[1].masks_queries_logits)
    del model, plain
    torch.cuda.empty_cache()


def test_batch_of_pairs_matches_single_pairs():
    """B = 2 pairs in one forward (graph-captured on the third call) == each pair alone, up to the fp32 rounding of sums whose slicing
    depends on the launch geometry (split-K at few tiles), and the id maps up to border pixels."""
    from oracle import weights as OW
    from siu3r_amd.model import SIU3RModel

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    g = torch.Generator().manual_seed(3)
    img = torch.rand(2, 2, 3, 128, 128, generator=g).cuda()
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(2, 2, 1, 1).cuda()
    model = SIU3RModel(_STATE["sd"], image_size=(128, 128), precision="bf16x3")
    with torch.no_grad():
        outs = [model(img, K) for _ in range(3)]            # eager, capture, replay
        singles = [model(img[i:i + 1], K[i:i + 1]) for i in range(2)]
    for o in outs[1:]:
        assert torch.equal(o[0].means, outs[0][0].means) and torch.equal(o[1].class_queries_logits, outs[0][1].class_queries_logits)
    for i in range(2):
        gb, gs_ = outs[0][0], singles[i][0]
        for f in ("means", "covariances", "harmonics", "opacities"):
            a, b = getattr(gb, f)[i], getattr(gs_, f)[0]
            assert float((a - b).abs().max()) <= 5e-4 * float(b.abs().max()), f  # (measured 2.3e-4 on the covariances: B = 1 and B = 2 also pick different attention grids)
        assert float((gb.semantic_labels[i] == gs_.semantic_labels[0]).float().mean()) >= 0.995
        assert float((gb.instance_labels[i] == gs_.instance_labels[0]).float().mean()) >= 0.995
        assert [(s_["id"], s_["label_id"], s_["was_fused"]) for s_ in outs[0][3][i]] == [(s_["id"], s_["label_id"], s_["was_fused"]) for s_ in singles[i][3][0]]
    del model
    torch.cuda.empty_cache()


def test_forward_through_the_checkpoint_loader(tmp_path):
    """SURVEY 8(f)1 on the GPU (reference inference.py:119-136: Pipeline.load_from_checkpoint -> pipeline.model(images, intrinsics)): the
    655 M-parameter synthetic state dict is written as a Lightning-shaped .ckpt (`model.` prefix, metric modules, a pickled config class
    that cannot be imported), read back by load_siu3r_state_dict, and the loaded weights drive SIU3RModel at 256^2 in the benchmarked
    bf16x3 mode to the REFERENCE's golden outputs at 1e-3."""
    from golden_utils import FIELDS, compare_summary, default_K, fixture_images, load_model_fixture, segments_match
    from oracle import weights as OW
    from siu3r_amd import checkpoint as ck
    from siu3r_amd.model import SIU3RModel
    from test_checkpoint import _write_lightning_ckpt

    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    path = tmp_path / "siu3r_epoch100.ckpt"
    _write_lightning_ckpt(path, _STATE["sd"])
    sd = ck.load_siu3r_state_dict(path, verbose=False)
    assert set(sd) == set(_STATE["sd"]) and not any(k.startswith(("lpips.", "psnr.", "model.")) for k in sd)
    z, meta = load_model_fixture(256)
    model = SIU3RModel(sd, image_size=(256, 256), precision="bf16x3")
    with torch.no_grad():
        g, seg, masks, infos, qs = model(fixture_images(256).cuda(), default_K().cuda(), enable_query_class_logit_lift=True)
    for f in FIELDS:
        compare_summary(f, getattr(g, f), z, 1e-3)
    compare_summary("class_queries_logits", seg.class_queries_logits, z, 1e-3)
    compare_summary("masks_queries_logits", seg.masks_queries_logits, z, 1e-3)
    segments_match(infos, meta["seg_infos"], 2e-6 + 1e-3 * 0.05)


SWEEP = [(1, 2, 64, 64, 1), (1, 2, 96, 160, 2), (1, 2, 160, 96, 3), (2, 2, 64, 96, 4), (1, 3, 96, 96, 5), (1, 4, 64, 64, 6), (1, 2, 224, 224, 7)]


@pytest.mark.parametrize("case", SWEEP, ids=[f"B{b}V{v}_{h}x{w}" for (b, v, h, w, _) in SWEEP])
def test_parity_sweep(case):
    """The parity statement of bench.py's `config.parity`, driver-verified on seven shapes / seeds (odd aspect ratios, B = 2, V = 3 and 4)
    against the pinned CPU oracle in the benchmarked bf16x3 mode:
      * every Gaussian field <= 1e-3 max-normalised (north_star's bar; measured <= 8e-5);
      * Mask2Former class / mask logits <= 1e-3 on most inputs, and a few 1e-3 (seen: up to 5.7e-3) when one of the nine THRESHOLDED attention masks
        (sigmoid(mask) < 0.5, reference video_seg_decoder.py:1306-1308, 1461-1478) flips a borderline pixel between two fp32 evaluation
        orders -- both logit tensors then move together (measured 1-3e-3 in 3 of 7 cases); in exactly those cases the test re-runs
        the forward with the ORACLE's nine boolean masks forced in and demands <= 1e-3: the flipped pixels are the whole difference;
      * graph replay bit-identical to eager; label maps agree >= 0.995 (argmax over fp32 scores at segment borders)."""
    from oracle import siu3r_oracle as O
    from oracle import weights as OW
    from siu3r_amd.model import SIU3RModel, SIU3RMultiViewModel

    B, V, H, W, seed = case
    if "sd" not in _STATE:
        _STATE["sd"] = OW.make_weights(0)
    sd = _STATE["sd"]
    gen = torch.Generator().manual_seed(seed)
    img = torch.rand(B, V, 3, H, W, generator=gen)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, V, 1, 1)
    K[:, :, 0, 0] *= 1.0 + 0.1 * torch.rand(B, V, generator=gen)
    model = (SIU3RModel if V == 2 else SIU3RMultiViewModel)(sd, image_size=(H, W), precision="bf16x3")
    with torch.no_grad():
        ref = (O.model_forward if V == 2 else O.model_forward_multi)(sd, img, K, keep_intermediates=False)
        outs = [model(img.cuda(), K.cuda()) for _ in range(3)]  # eager, capture, replay
    gs, seg = outs[2][0], outs[2][1]
    err = lambda a, b: float((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    fields = {f: err(getattr(gs, f), ref[f]) for f in ("means", "covariances", "harmonics", "opacities", "scales", "rotations")}
    logits = {"class": err(seg.class_queries_logits, ref["class_queries_logits"]), "mask": err(seg.masks_queries_logits, ref["masks_queries_logits"])}
    agree = min(float((gs.semantic_labels.cpu() == ref["semantic_labels"]).float().mean()), float((gs.instance_labels.cpu() == ref["instance_labels"]).float().mean()))
    print(f"[parity-sweep] B={B} V={V} {H}x{W}: fields max {max(fields.values()):.2e} ({max(fields, key=fields.get)}), class {logits['class']:.2e}, "
          f"mask {logits['mask']:.2e}, labels agree {agree:.5f}")
    assert max(fields.values()) <= 1e-3, fields
    # raw logits: <= 1e-3 unless a thresholded attention-mask pixel flipped (proven below, case by case, with the oracle's masks forced in:
    # then <= 1e-3 again).  How far ONE flipped pixel moves the logits depends on how many keys the query's mask leaves open, not on the
    # arithmetic: 1-3e-3 in rounds 4 / 5, 5.7e-3 on B2V2_64x96 in round 6 (the composed lateral convolution rounds differently and
    # flips a different borderline pixel); the cap only catches a broken run, the proof is the forced-mask assertion
    assert max(logits.values()) <= 2e-2, logits
    assert agree >= 0.995
    if max(logits.values()) > 2e-4:
        # well above the 3e-5 the logits otherwise show (and in some rounds / on some shapes above the 1e-3 bar: 1-3e-3 in 3 of 7 cases in
        # round 4, 9.5e-4 at 224^2 in round 5): then it must be a flipped pixel of a thresholded attention mask and nothing else -- with
        # the oracle's nine boolean masks forced into the HIP run (eager) the same logits are within 1e-3 (in fact back at the 3e-5 level)
        model.use_graph = False
        model.mask2former.forced_attn_masks = ref["attn_masks"]
        with torch.no_grad():
            _, seg_f = model(img.cuda(), K.cuda())[:2]
        model.mask2former.forced_attn_masks = None
        forced = {"class": err(seg_f.class_queries_logits, ref["class_queries_logits"]), "mask": err(seg_f.masks_queries_logits, ref["masks_queries_logits"])}
        print(f"[parity-sweep]   with the oracle's attention masks forced: class {forced['class']:.2e}, mask {forced['mask']:.2e}")
        assert max(forced.values()) <= 1e-3, forced
        # ... and WHICH pixels flipped, and on which side: the nine masks the HIP run attended through against the fp32 oracle's, with an
        # fp64 run of the oracle as the referee (the thresholded quantity sigmoid(mask) - 0.5 is within fp32 rounding of zero there)
        model.mask2former.record_attn_masks = rec = []
        with torch.no_grad():
            model(img.cuda(), K.cuda())
        model.mask2former.record_attn_masks = None
        assert len(rec) == len(ref["attn_masks"]) == 9
        sd64 = {k_: (v_.double() if v_.is_floating_point() else v_) for k_, v_ in sd.items()}
        with torch.no_grad():
            ref64 = (O.model_forward if V == 2 else O.model_forward_multi)(sd64, img.double(), K.double(), keep_intermediates=False)
        n_flip = n_tot = hip_right = orc_right = 0
        for hm, om, dm in zip(rec, ref["attn_masks"], ref64["attn_masks"]):
            hm = hm[..., :om.shape[-1]].cpu().bool()
            om, dm = om.bool(), dm.bool()
            fl = hm != om
            n_flip, n_tot = n_flip + int(fl.sum()), n_tot + om.numel()
            hip_right, orc_right = hip_right + int((fl & (hm == dm)).sum()), orc_right + int((fl & (om == dm)).sum())
        print(f"[parity-sweep]   attention-mask pixels on which the HIP run and the fp32 oracle differ: {n_flip} of {n_tot} over the nine layers; "
              f"the fp64 oracle sides with the HIP run on {hip_right} of them and with the fp32 oracle on {orc_right}")
        assert 0 < n_flip <= 1e-4 * n_tot + 64, (n_flip, n_tot)
        model.use_graph = True
    assert torch.equal(outs[0][0].means, gs.means) and torch.equal(outs[0][1].masks_queries_logits, seg.masks_queries_logits)
    del model
    torch.cuda.empty_cache()
