"""Container-only pins (skipped where /root/reference is absent): the weight spec against the reference's own
state_dict, and the oracle against a live forward of the imported reference model."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import _ref_import as R  # noqa: E402

pytestmark = [pytest.mark.refpin, pytest.mark.skipif(not R.reference_available(), reason="/root/reference not present")]


def test_weight_spec_and_forward_match_reference():
    from oracle import siu3r_oracle as O
    from oracle import weights as OW

    sd = OW.make_weights(0)
    model = R.build_reference_model((128, 128))
    ref_sd = model.state_dict()
    assert set(ref_sd) == set(sd)
    assert all(tuple(ref_sd[k].shape) == tuple(sd[k].shape) and ref_sd[k].dtype == sd[k].dtype for k in ref_sd)
    model.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 2, 3, 128, 128, generator=g)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1)
    with torch.no_grad():
        ref = model(img, K, enable_query_class_logit_lift=True)
        out = O.model_forward(sd, img, K, keep_intermediates=False)
    for f in ("means", "covariances", "harmonics", "opacities", "scales", "rotations"):
        a, b = out[f], getattr(ref[0], f)
        assert float((a - b).abs().max()) <= 5e-5 * float(b.abs().max()), f
    assert torch.equal(out["class_queries_logits"], ref[1].class_queries_logits)
    assert torch.equal(out["masks_queries_logits"], ref[1].masks_queries_logits)
    strip = lambda segs: [[(s_["id"], s_["label_id"], s_["was_fused"]) for s_ in i] for i in segs]
    assert strip(out["seg_infos"]) == strip(ref[3]) and len(ref[3][0]) >= 3
