"""Helpers shared by the golden-fixture tests (oracle on CPU, HIP path on the GPU box)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRIDE = 4093
WINDOW = 65536


def window_of(numel):
    """the contiguous window tests/golden/make_golden.py stores of every tensor (dense comparison)"""
    if numel <= WINDOW:
        return 0, numel
    return (numel // 3) // 64 * 64, WINDOW


def load_model_fixture(size):
    z = np.load(os.path.join(GOLDEN, f"model_{size}.npz"), allow_pickle=False)
    meta = json.load(open(os.path.join(GOLDEN, f"model_{size}.json")))
    return z, meta


def fixture_images(size):
    """The fixture input: the reference's assets/living_room pair (decoded pixels committed as data in
    input_pair_u8.npz), /255, bilinearly resized exactly as tests/golden/make_golden.py does."""
    a = torch.from_numpy(np.load(os.path.join(GOLDEN, "input_pair_u8.npz"))["pair"]).permute(0, 3, 1, 2).float() / 255.0
    if a.shape[-1] != size:
        a = torch.stack([torch.nn.functional.interpolate(x[None], size=(size, size), mode="bilinear", align_corners=False)[0] for x in a])
    return a[None]


def default_K(B=1, V=2):
    return torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, V, 1, 1)


def multi_views(pair, V):
    """V <= 8 views made of the asset pair only (no new image data): the pair, then its horizontal mirrors, vertical mirrors, transposes"""
    views = [pair[0], pair[1], pair[0].flip(-1), pair[1].flip(-1), pair[0].flip(-2), pair[1].flip(-2), pair[0].transpose(-1, -2), pair[1].transpose(-1, -2)]
    assert 2 <= V <= len(views)
    return torch.stack(views[:V])[None]


def fixture_images_multi(size=128, V=3):
    """multi-view fixture input (tests/golden/make_golden.py::multi_fixture): V = 3 is the asset pair and the first image mirrored"""
    return multi_views(fixture_images(size)[0], V)


def load_multi_fixture(V=3, size=128):
    z = np.load(os.path.join(GOLDEN, f"model_multi_v{V}_{size}.npz"), allow_pickle=False)
    return z, json.load(open(os.path.join(GOLDEN, f"model_multi_v{V}_{size}.json")))


REL_FLOOR = 1e-3  # element-wise relative error: |err| / max(|ref|, REL_FLOOR * absmax(ref))


def elementwise_rel(got: torch.Tensor, want: torch.Tensor, absmax: float):
    """(worst, fraction above 1e-3) of the ELEMENT-WISE relative error |got - want| / max(|want|, 1e-3 * absmax): the north-star's "1e-3 rel"
    read per element, with a floor at a thousandth of the tensor's range so that entries that are zero up to rounding do not divide by
    nothing (the max-normalised figure next to it divides every element by absmax)."""
    g, w = got.double().reshape(-1), want.double().reshape(-1)
    rel = (g - w).abs() / torch.clamp(w.abs(), min=REL_FLOOR * absmax + 1e-300)
    return float(rel.max()) if rel.numel() else 0.0, float((rel > 1e-3).double().mean()) if rel.numel() else 0.0


def compare_summary(name, t: torch.Tensor, z, tol, rel_tol=None):
    """rel_tol: bound on the element-wise relative error (elementwise_rel) of the samples and of the dense window; None = reported only"""
    f = t.detach().float().cpu().reshape(-1)
    want = torch.from_numpy(z[f"{name}.sample"])
    got = f[::STRIDE]
    assert list(t.shape) == list(z[f"{name}.shape"]), (name, t.shape, z[f"{name}.shape"])
    scale = float(z[f"{name}.absmax"]) + 1e-30
    err = float((got - want).abs().max()) / scale
    l2 = abs(float(f.double().norm()) - float(z[f"{name}.l2"])) / (float(z[f"{name}.l2"]) + 1e-30)
    rel, frac = elementwise_rel(got, want, scale)
    print(f"[golden] {name:24s} sample max-normalised err {err:.3e}  |l2 rel diff| {l2:.3e}  element-wise rel worst {rel:.3e} (> 1e-3 on {frac:.2%} of the elements)")
    assert err <= tol and l2 <= tol, (name, err, l2)
    assert rel_tol is None or rel <= rel_tol, (name, "element-wise relative error", rel, rel_tol)
    if tol <= 1e-3:  # the 1e-3 mode: the element-wise tail at what the data supports (tests/test_model_gpu.py::_report)
        assert frac <= 0.10 and rel <= 0.25, (name, "element-wise tail", rel, frac)
    if f"{name}.window" in z.files:
        # one contiguous window (64 Ki elements) compared densely: max-normalised error and the relative L2 norm OF THE DIFFERENCE
        o, n = window_of(f.numel())
        w_want, w_got = torch.from_numpy(z[f"{name}.window"]).double(), f[o:o + n].double()
        assert w_want.numel() == n, (name, w_want.numel(), n)
        werr = float((w_got - w_want).abs().max()) / scale
        wl2 = float((w_got - w_want).norm() / (w_want.norm() + 1e-30))
        wrel, wfrac = elementwise_rel(w_got, w_want, scale)
        print(f"[golden] {name:24s} dense window [{o}:{o + n}] max-normalised err {werr:.3e}  rel l2 of the difference {wl2:.3e}  "
              f"element-wise rel worst {wrel:.3e} (> 1e-3 on {wfrac:.2%})")
        assert werr <= tol and wl2 <= tol, (name, werr, wl2)
        assert rel_tol is None or wrel <= rel_tol, (name, "element-wise relative error (window)", wrel, rel_tol)
        if tol <= 1e-3:
            assert wfrac <= 0.10 and wrel <= 0.25, (name, "element-wise tail (window)", wrel, wfrac)
        err = max(err, werr)
    return err


FIELDS = ("means", "covariances", "harmonics", "opacities", "scales", "rotations")


ISTRIDE = 97


def compare_integer_outputs(semantic_labels, instance_labels, seg_mask, qcl, z, qcl_tol, min_agree=1.0):
    """Integer outputs of the panoptic branch against the reference-generated fixture: strided samples and histograms of the id maps
    (bit-exact when min_agree == 1, else the agreeing fraction of sampled pixels is reported and bounded), samples of the
    query x class logit volume."""
    def frac(name, t, key):
        got = t.detach().cpu().reshape(-1)[::ISTRIDE].numpy()
        want = z[key]
        assert got.shape == want.shape, (name, got.shape, want.shape)
        a = float((got == want).mean())
        print(f"[golden] {name:24s} sampled pixels agreeing {a:.5f}")
        assert a >= min_agree, (name, a)

    frac("semantic_labels", semantic_labels, "semantic_labels.sample")
    frac("instance_labels", instance_labels, "instance_labels.sample")
    frac("segmentation", seg_mask, "seg_mask.sample")
    # whole-map histograms: every id's pixel count within (1 - min_agree) of the map (exact when min_agree == 1)
    for name, got, want in (("segmentation", torch.bincount(seg_mask.detach().cpu().reshape(-1).long().clamp_min(0), minlength=8).numpy(), z["seg_mask.hist"]),
                            ("semantic_labels", torch.bincount(semantic_labels.detach().cpu().reshape(-1).long(), minlength=22).numpy(), z["semantic_labels.hist"])):
        assert got.shape == want.shape, (name, got, want)
        moved = int(np.abs(got - want).sum()) // 2
        print(f"[golden] {name:24s} pixels that changed id (histogram distance) {moved} of {int(want.sum())}")
        assert moved <= (1.0 - min_agree) * float(want.sum()) + 1e-9, (name, got.tolist(), want.tolist())
    if qcl is not None:
        assert list(qcl.shape) == list(z["qcl.shape"]), (qcl.shape, z["qcl.shape"])
        got = qcl.detach().float().cpu().reshape(-1)[::STRIDE]
        err = float((got - torch.from_numpy(z["qcl.sample"])).abs().max())
        print(f"[golden] query_class_logits       sample max abs err {err:.3e}")
        assert err <= qcl_tol, err


def segments_match(got, want, score_tol):
    """segments_info lists: ids / labels / fused flags exact, scores (rounded to 6 decimals by the reference) within score_tol"""
    assert len(got) == len(want), (got, want)
    for a, b in zip(got, want):
        assert len(a) == len(b), (a, b)
        for x, y in zip(a, b):
            assert (x["id"], x["label_id"], x["was_fused"]) == (y["id"], y["label_id"], y["was_fused"]), (x, y)
            assert abs(x["score"] - y["score"]) <= score_tol, (x, y)


def labels_agree(name, got, want, min_frac):
    """Fraction of identical entries of two integer id maps.  Ids come out of an argmax over fp32 scores: with a non-empty panoptic
    result, a pixel on a segment border can change owner between two fp32 evaluation orders (even between the reference and its
    CPU restatement at 512^2: 2 of 524 288 pixels), so end-to-end comparisons bound the disagreeing fraction; bit-exactness of the
    integer kernels themselves is tested on identical inputs (tests/test_postprocess_gpu.py)."""
    a, b = got.detach().cpu().reshape(-1), want.detach().cpu().reshape(-1)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    f = float((a == b).float().mean())
    print(f"[labels] {name:24s} agreement {f:.6f}")
    assert f >= min_frac, (name, f)
    return f
