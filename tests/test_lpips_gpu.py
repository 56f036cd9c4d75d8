"""GPU: siu3r_amd.lpips.LPIPS (bf16x3 implicit-GEMM convolutions, siu3r_maxpool2x2s2, siu3r_lpips_layer) against the CPU restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_maxpool2x2_and_layer_distance_kernels():
    from siu3r_amd import ops

    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 13, 18, 64, generator=g)
    got = ops.maxpool2x2s2(x.cuda()).cpu()
    ref = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert got.shape == (2, 6, 9, 64) and torch.equal(got, ref)
    for C in (64, 512):
        f0, f1 = torch.randn(3, 7, 9, C, generator=g).relu(), torch.randn(3, 7, 9, C, generator=g).relu()
        f0[0, 0, 0] = 0  # an all-zero feature vector: eps keeps the quotient finite (0 / sqrt(eps))
        w = torch.rand(C, generator=g)
        d = ops.lpips_layer(f0.cuda(), f1.cuda(), w.cuda(), 1e-8).cpu()
        n0, n1 = f0 / torch.sqrt(1e-8 + (f0 * f0).sum(-1, keepdim=True)), f1 / torch.sqrt(1e-8 + (f1 * f1).sum(-1, keepdim=True))
        ref = (w * (n0 - n1) ** 2).sum(-1)
        assert d.shape == (3, 7, 9) and torch.allclose(d, ref, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("shape", [(64, 80), (128, 128), (50, 70)], ids=["64x80", "128x128", "odd50x70"])
def test_lpips_against_the_cpu_restatement(shape):
    """whole metric: five taps of the VGG16 stack on random He-scaled weights; 1e-3 relative (measured ~1e-5: fp32 activations, bf16x3 products)"""
    from oracle import lpips_oracle as LO
    from siu3r_amd.lpips import LPIPS

    H, W = shape
    sd = LO.random_weights(5)
    g = torch.Generator().manual_seed(H)
    a = torch.rand(2, 3, H, W, generator=g)
    b = (a + 0.2 * torch.randn(2, 3, H, W, generator=g)).clamp(0, 1)
    ref = LO.lpips(sd, a, b)
    m = LPIPS(sd)
    got = m(a, b).cpu()
    rel = float(((got - ref).abs() / ref.abs()).max())
    print(f"[lpips] {H}x{W}: ref {ref.tolist()} hip {got.tolist()} rel {rel:.2e}")
    assert rel <= 1e-3
    assert float(m(a, a).abs().max()) == 0.0
    # the evaluator's call: one HWC image pair as read back from the PNGs
    one = m(a[0].permute(1, 2, 0).numpy(), b[0].permute(1, 2, 0).numpy())
    assert one.shape == (1,) and abs(float(one[0]) - float(ref[0])) <= 1e-3 * float(ref[0])


def test_evaluate_dir_scores_lpips(tmp_path):
    """the render scores of a result tree get an `lpips` entry per image and results.json its mean"""
    from PIL import Image

    from oracle import lpips_oracle as LO
    from siu3r_amd import eval_io as E
    from siu3r_amd.lpips import LPIPS

    sd = LO.random_weights(2)
    d = tmp_path / "scene0000_00_context_0_1"
    (d / "rgb").mkdir(parents=True)
    (d / "rgb_gt").mkdir()
    g = np.random.default_rng(0)
    refs = []
    for i in range(2):
        gt = g.random((48, 64, 3))
        pr = np.clip(gt + 0.1 * g.standard_normal(gt.shape), 0, 1)
        for name, im in (("rgb", pr), ("rgb_gt", gt)):
            Image.fromarray((im * 255).astype(np.uint8)).save(d / name / f"{i:06d}.png")
        a, b = (torch.from_numpy(E.load_image01(d / n / f"{i:06d}.png")).permute(2, 0, 1)[None] for n in ("rgb", "rgb_gt"))
        refs.append(float(LO.lpips(sd, a, b)[0]))
    res = E.evaluate_dir(tmp_path, write=True, lpips=LPIPS(sd))
    assert res["lpips"] == pytest.approx(np.mean(refs), rel=1e-3)
    import json

    scores = json.load(open(d / "render_scores.json"))
    assert [s["lpips"] for s in scores] == pytest.approx(refs, rel=1e-3) and "psnr" in scores[0]
