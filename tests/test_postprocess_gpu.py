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
