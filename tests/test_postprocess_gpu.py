"""Integer-exact parity of the on-device panoptic post-process against the CPU oracle on crafted inputs
(keep threshold, overlap-ratio rejection, stuff fusing, empty item, kept-but-none-accepted item, and the
reference's stale (height, width) quirk)."""
import pytest
import torch

from crafted import crafted_panoptic_inputs, permuted

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("order", [[0, 1, 2, 3], [2, 0, 3, 1]], ids=["natural", "quirk-first"])
@pytest.mark.parametrize("target", [(64, 64), (96, 80)], ids=["64", "96x80"])
def test_panoptic_postprocess_exact(order, target):
    from oracle import siu3r_oracle as O
    from siu3r_amd.postprocess import VideoMask2FormerImageProcessor, post_process_gaussians
    from siu3r_amd.gaussians_types import Gaussians

    cls, msk = permuted(*crafted_panoptic_inputs(), order)
    ref = O.panoptic_postprocess(cls, msk, target)
    out = dict(class_queries_logits=cls.cuda(), masks_queries_logits=msk.cuda())
    res = VideoMask2FormerImageProcessor().post_process_panoptic_segmentation(
        out, threshold=0.5, target_sizes=[target] * len(order), label_ids_to_fuse={0, 1})
    assert len(res) == len(ref)
    for b, (r, m) in enumerate(zip(ref, res)):
        assert m["segmentation"].dtype == r["segmentation"].dtype, (b, m["segmentation"].dtype)
        assert torch.equal(m["segmentation"].cpu(), r["segmentation"]), f"segmentation differs for item {b}"
        # ids / labels / fused flags are integer outputs: exact.  Scores are fp32 softmax values rounded to 6 decimals
        # by the reference (:1444): compare within 2e-6 (the last printed digit may round differently).
        strip = lambda segs: [{k: v for k, v in s_.items() if k != "score"} for s_ in segs]
        assert strip(m["segments_info"]) == strip(r["segments_info"]), (b, m["segments_info"], r["segments_info"])
        assert all(abs(x["score"] - y["score"]) <= 2e-6 for x, y in zip(m["segments_info"], r["segments_info"]))
        assert len(m["query_scores"]) == len(r["query_scores"]) and all(abs(x - y) <= 2e-6 for x, y in zip(m["query_scores"], r["query_scores"]))
        assert tuple(m["query_class_logits"].shape) == tuple(r["query_class_logits"].shape), (b, m["query_class_logits"].shape, r["query_class_logits"].shape)
        err = float((m["query_class_logits"].cpu() - r["query_class_logits"]).abs().max())
        print(f"[parity] query_class_logits item {b}: max abs err {err:.2e}")
        assert err <= 2e-6
    B, (H, W) = len(order), target
    sem_ref, ins_ref = O.scatter_labels(ref, B, 2, H, W)
    g = Gaussians(means=torch.zeros(B, 2, H * W, 3), covariances=torch.zeros(B, 2, H * W, 3, 3), harmonics=torch.zeros(B, 2, H * W, 3, 25),
                  opacities=torch.zeros(B, 2, H * W), scales=torch.zeros(B, 2, H * W, 3), rotations=torch.zeros(B, 2, H * W, 4))
    g, masks, infos, qcl, qs = post_process_gaussians(g, res, B, 2, H, W, True)
    assert torch.equal(g.semantic_labels.cpu(), sem_ref) and torch.equal(g.instance_labels.cpu(), ins_ref)
    assert g.semantic_labels.dtype == torch.int32 and g.means.shape == (B, 2 * H * W, 3)


def _network_like_logits(Q=100, T=2, h=128, w=128, C=21, kept=14, seed=0):
    """class / mask logits with the statistics of the network's own output at 2 x 512^2 (mask size 128^2): `kept` confident non-void
    queries (two stuff classes among them), smooth mask logits around -2.8 with unit-scale variation, so that segments border each other."""
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(1, Q, C, generator=g)
    cls[:, :, C - 1] += 9.0
    lab = [1, 1, 0, 5, 12, 19, 13, 1, 10, 3, 19, 7, 4, 15][:kept]
    for i, c in enumerate(lab):
        cls[0, 7 * i + 2, c] += 14.0
    coarse = torch.randn(1 * Q * T, 1, h // 8, w // 8, generator=g)
    msk = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False).view(1, Q, T, h, w) * 2.2 - 2.8
    return cls, msk + 0.05 * torch.randn(1, Q, T, h, w, generator=g)


@pytest.mark.parametrize("target", [(512, 512), (256, 256)], ids=["512", "256"])
def test_panoptic_postprocess_full_size(target):
    """The kept-query kernels (argmax over kept queries at the target size, area / original-area counts, stuff fusing, the
    [T*H*W, q, 21] query x class logit volume) at the benchmark's size on network-like logits: integer outputs bit-exact against the
    oracle ON IDENTICAL INPUT LOGITS."""
    from oracle import siu3r_oracle as O
    from siu3r_amd.postprocess import VideoMask2FormerImageProcessor

    cls, msk = _network_like_logits()
    ref = O.panoptic_postprocess(cls, msk, target)[0]
    assert len(ref["segments_info"]) >= 5 and any(s_["was_fused"] for s_ in ref["segments_info"])
    out = dict(class_queries_logits=cls.cuda(), masks_queries_logits=msk.cuda())
    res = VideoMask2FormerImageProcessor().post_process_panoptic_segmentation(out, threshold=0.5, target_sizes=[target], label_ids_to_fuse={0, 1})[0]
    strip = lambda segs: [(s_["id"], s_["label_id"], s_["was_fused"]) for s_ in segs]
    assert strip(res["segments_info"]) == strip(ref["segments_info"])
    assert all(abs(x["score"] - y["score"]) <= 2e-6 for x, y in zip(res["segments_info"], ref["segments_info"]))
    same = float((res["segmentation"].cpu() == ref["segmentation"]).float().mean())
    print(f"[parity] full-size segmentation {target}: {len(ref['segments_info'])} segments, identical pixels {same:.7f}")
    assert torch.equal(res["segmentation"].cpu(), ref["segmentation"]), f"segmentation differs on {1 - same:.2e} of the pixels"
    err = float((res["query_class_logits"].cpu() - ref["query_class_logits"]).abs().max())
    assert tuple(res["query_class_logits"].shape) == tuple(ref["query_class_logits"].shape) and err <= 2e-6, err
