"""CPU: the LPIPS restatement (oracle/lpips_oracle.py; parity unpinned, see its header) against the properties the published algorithm has,
the weight extraction from checkpoint-shaped key names, and the evaluator plumbing of the `lpips` score."""
import numpy as np
import pytest
import torch


def test_oracle_properties():
    from oracle import lpips_oracle as LO

    sd = LO.random_weights(3)
    g = torch.Generator().manual_seed(1)
    a, b = torch.rand(2, 3, 48, 64, generator=g), torch.rand(2, 3, 48, 64, generator=g)
    d_ab, d_ba, d_aa = LO.lpips(sd, a, b), LO.lpips(sd, b, a), LO.lpips(sd, a, a)
    assert d_ab.shape == (2,) and bool((d_ab > 0).all())
    assert torch.equal(d_ab, d_ba) and float(d_aa.abs().max()) == 0.0
    # a small perturbation is closer than an unrelated image, and the distance grows with it
    near, nearer = (a + 0.05 * (b - a)).clamp(0, 1), (a + 0.01 * (b - a)).clamp(0, 1)
    assert bool((LO.lpips(sd, a, nearer) < LO.lpips(sd, a, near)).all()) and bool((LO.lpips(sd, a, near) < d_ab).all())
    # per-item independence: a batch is the concatenation of its single pairs
    assert torch.allclose(LO.lpips(sd, a[1:], b[1:]), d_ab[1:], rtol=1e-6, atol=0)
    # the unit normalisation makes a tap blind to a positive rescaling of the LAST convolution of the stack (only tap 5 sees it)
    sd2 = dict(sd)
    sd2["features.28.weight"], sd2["features.28.bias"] = sd["features.28.weight"] * 3.0, sd["features.28.bias"] * 3.0
    assert torch.allclose(LO.lpips(sd2, a, b), d_ab, rtol=2e-5, atol=0)


def test_weights_from_checkpoint_shaped_keys():
    """the key tails of a Pipeline checkpoint (lpips.net.net.slice<k>.<i>.*, lpips.net.lin<k>.model.1.weight, the lins.<k> aliases and
    the scaling buffers), of the metric alone, and of a torchvision + lin dict all resolve to the same tensors; an incomplete set is refused"""
    from oracle import lpips_oracle as LO
    from siu3r_amd.lpips import VGG_SLICES, weights_from_state_dict

    sd = LO.random_weights(0)
    base = weights_from_state_dict(sd)
    assert base is not None and base["conv28.weight"].shape == (512, 512, 3, 3) and base["lin0"].shape == (64,)
    assert base["shift"].tolist() == pytest.approx([-0.030, -0.088, -0.188]) and base["scale"].tolist() == pytest.approx([0.458, 0.448, 0.450])
    pl = {"model.backbone.enc_norm.weight": torch.ones(4)}
    for k, sl in enumerate(VGG_SLICES):
        for i in sl:
            for p in ("weight", "bias"):
                pl[f"lpips.net.net.slice{k + 1}.{i}.{p}"] = sd[f"features.{i}.{p}"]
    for k in range(5):
        pl[f"lpips.net.lin{k}.model.1.weight"] = sd[f"lin{k}.model.1.weight"]
        pl[f"lpips.net.lins.{k}.model.1.weight"] = sd[f"lin{k}.model.1.weight"] * 0 + 7  # (an alias entry must not win over lin<k>)
    pl["lpips.net.scaling_layer.shift"] = torch.tensor([-0.03, -0.088, -0.188]).view(1, 3, 1, 1)
    pl["lpips.net.scaling_layer.scale"] = torch.tensor([0.458, 0.448, 0.45]).view(1, 3, 1, 1)
    got = weights_from_state_dict(pl)
    assert set(got) == set(base)
    for k in base:
        assert torch.equal(got[k], base[k]), k
    assert weights_from_state_dict({"model.backbone.enc_norm.weight": torch.ones(4)}) is None
    del pl["lpips.net.net.slice3.12.bias"]
    with pytest.raises(RuntimeError, match="incomplete"):
        weights_from_state_dict(pl)


def test_accumulator_carries_lpips_through_the_gather_vector():
    from siu3r_amd import metrics as M

    a, b = M.MetricAccumulator(), M.MetricAccumulator()
    a.add_lpips(0.25)
    a.add_lpips(0.35)
    b.add_lpips(0.6)
    assert "lpips" not in M.MetricAccumulator().compute()
    whole = M.MetricAccumulator.from_vectors(np.stack([a.to_vector(), b.to_vector()]))
    assert whole.compute()["lpips"] == pytest.approx(0.4)
