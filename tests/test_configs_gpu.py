"""BASELINE.json configurations beyond the single benchmark pair, at their full sizes:
  configs[2]  batch of 8 pairs @512^2 through the full forward (all heads, panoptic post-process, lifting inputs)
  configs[4]  8 context views @512^2 -> 2 097 152 Gaussians -> one 1920x1080 frame through BOTH render semantics (viewer / gsplat-style and
              SplattingCUDA / diff-gaussian-rasterization-style), from a camera that sees more than half of the Gaussians, against the C
              oracle (oracle/raster_ref.c) on the whole frame.
(configs[3] = configs[2] on 8 GPUs: the sharding is covered by tests/test_distributed_cpu.py; no 8-GPU node is available to the tests.)"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_SD = {}


def _weights():
    from oracle import weights as OW

    if "sd" not in _SD:
        _SD["sd"] = OW.make_weights(0)
    return _SD["sd"]


def test_config3_batch_of_eight_pairs_512():
    """item 0 of the batch is the golden input of the 512^2 reference run: its outputs must match the reference's golden vectors at 1e-3
    inside the batch of 8, and items must equal the same pair run alone (fields to 5e-4 -- measured 4e-5 .. 2e-4 --, id maps up to border pixels)."""
    from golden_utils import FIELDS, compare_integer_outputs, compare_summary, default_K, fixture_images, labels_agree, load_model_fixture, segments_match
    from siu3r_amd.model import SIU3RModel

    B, S = 8, 512
    z, meta = load_model_fixture(S)
    g = torch.Generator().manual_seed(11)
    fx_ = fixture_images(S)
    img = torch.cat([fx_, torch.rand(B - 2, 2, 3, S, S, generator=g), fx_.flip(1)]).cuda()  # item 7: the asset pair with its views swapped
    K = default_K().repeat(B, 1, 1, 1).cuda()
    model = SIU3RModel(_weights(), image_size=(S, S), precision="bf16x3")
    with torch.no_grad():
        outs = [model(img, K, enable_query_class_logit_lift=True) for _ in range(3)]  # eager, capture, replay
        gs, seg, masks, infos, qs = outs[2]
        assert gs.means.shape == (B, 2 * S * S, 3) and len(infos) == B
        assert torch.equal(outs[0][0].means, gs.means) and torch.equal(outs[0][0].instance_labels, gs.instance_labels)
        for f in FIELDS:
            compare_summary(f, getattr(gs, f)[0:1], z, 1e-3)
        compare_summary("class_queries_logits", seg.class_queries_logits[0:1], z, 1e-3)
        compare_summary("masks_queries_logits", seg.masks_queries_logits[0:1], z, 1e-3)
        segments_match(infos[0:1], meta["seg_infos"], 2e-6 + 1e-3 * 0.05)
        compare_integer_outputs(gs.semantic_labels[0:1], gs.instance_labels[0:1], masks[0], gs.seg_query_class_logits[0], z, 1e-3, min_agree=0.999)
        assert all(len(i) >= 1 for i in infos), [len(i) for i in infos]
        for i in (0, 5, 7):
            one = model(img[i:i + 1], K[i:i + 1], enable_query_class_logit_lift=True)
            for f in ("means", "covariances", "harmonics", "opacities"):
                a, b = getattr(gs, f)[i], getattr(one[0], f)[0]
                assert float((a - b).abs().max()) <= 5e-4 * float(b.abs().max()), (i, f)
            # (the launch geometry depends on the row count -- split-K at B = 1, none at B = 8 -- so sums differ in the last fp32 bits and a
            # border pixel may change owner).  The three single calls are: eager, graph capture, graph REPLAY WITH A NEW INPUT -- the last one
            # caught a hipMemsetAsync node that was not re-executed in order on replay (m2f attention-mask row counts)
            natural = True
            rel = lambda a, b: float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))
            print(f"[config3] item {i}: class {rel(seg.class_queries_logits[i], one[1].class_queries_logits[0]):.2e} mask {rel(seg.masks_queries_logits[i], one[1].masks_queries_logits[0]):.2e} "
                  f"segments {len(infos[i])} vs {len(one[3][0])}; eager-batch vs single sem agreement {float((outs[0][0].semantic_labels[i] == one[0].semantic_labels[0]).float().mean()):.5f}")
            labels_agree(f"semantic item {i}", gs.semantic_labels[i], one[0].semantic_labels[0], 0.9995 if natural else 0.95)
            labels_agree(f"instance item {i}", gs.instance_labels[i], one[0].instance_labels[0], 0.9995 if natural else 0.95)
            if natural:
                segments_match(infos[i:i + 1], one[3], 1e-4)
    del model
    torch.cuda.empty_cache()


def _pick_camera(means, tan_x, tan_y, near):
    """A camera looking down +z at the Gaussian cloud from far enough back that most of it is inside the frustum: median centre, distance
    from the 80 % quantile of the lateral spread (chosen from the means alone; the test then asserts what the rasterizer reports)."""
    c = means.median(0).values
    d = (means - c).abs()
    need = torch.maximum(torch.quantile(d[:, 0], 0.8) / tan_x, torch.quantile(d[:, 1], 0.8) / tan_y)
    back = float(need + torch.quantile(d[:, 2], 0.8)) + near
    c2w = torch.eye(4)
    c2w[:3, 3] = torch.tensor([float(c[0]), float(c[1]), float(c[2]) - back])
    return c2w


def test_config5_eight_views_two_million_gaussians_1080p():
    from oracle import raster_oracle as RO
    from siu3r_amd import cuda_splatting as cs, raster
    from siu3r_amd.gaussian_renderer import SplattingCUDA, rasterize_splats
    from siu3r_amd.gaussians_types import Gaussians
    from siu3r_amd.model import SIU3RMultiViewModel

    V, S, W, H = 8, 512, 1920, 1080
    g = torch.Generator().manual_seed(5)
    img = torch.rand(1, V, 3, S, S, generator=g).cuda()
    Kc = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, V, 1, 1).cuda()
    model = SIU3RMultiViewModel(_weights(), image_size=(S, S), precision="bf16x3")
    with torch.no_grad():
        out = model(img, Kc)
    G_ = out[0]
    G = G_.means.shape[1]
    assert G == V * S * S == 2_097_152
    for f in ("means", "covariances", "harmonics", "opacities", "scales", "rotations"):
        assert torch.isfinite(getattr(G_, f)).all(), f
    means_c = G_.means[0].float().cpu()

    # ---- viewer semantics (gsplat-style; reference viewer.py:301-336): pixel intrinsics, radius_clip 0.1 px, white background
    fx = 0.5 * W  # 90 degrees horizontally
    c2w = _pick_camera(means_c, (W / 2) / fx, (H / 2) / fx, 0.01)
    Kp = torch.tensor([[fx, 0, W / 2], [0, fx, H / 2], [0, 0, 1]])
    x, y, z_, w = G_.rotations[0].unbind(-1)
    quats = torch.stack((w, x, y, z_), -1).contiguous()
    lscale = G_.scales[0].log()
    logit = torch.logit(G_.opacities[0].clamp(1e-6, 1 - 1e-6))
    coeffs = G_.harmonics[0].permute(0, 2, 1).contiguous()  # [G, 25, 3]
    splats = dict(means=G_.means[0], quats=quats, scales=lscale, opacities=logit, sh0=coeffs[:, :1].contiguous(), shN=coeffs[:, 1:].contiguous())
    colors, alphas, info = rasterize_splats(splats, c2w[None], Kp[None], W, H, sh_degree=4, radius_clip=0.1)
    vis = int((info["tiles_touched"][0] > 0).sum())
    print(f"[config5] viewer semantics: {vis} of {G} Gaussians visible ({vis / G:.3f}), {info['tile_pairs'][0]} tile pairs")
    assert vis >= 0.5 * G, f"only {vis / G:.3f} of the Gaussians are in view"
    cov6_ref = RO.quat_scale_to_cov6(quats.cpu().numpy(), torch.exp(lscale).cpu().numpy())
    rgb_ref = RO.sh_eval(4, means_c.numpy(), c2w[:3, 3].numpy(), coeffs.cpu().numpy())
    cam = raster.make_cam_k3(torch.linalg.inv(c2w), fx, fx, W / 2, H / 2, W, H, radius_clip=0.1)
    ref = RO.forward(cam, means_c.numpy(), cov6_ref, torch.sigmoid(logit).cpu().numpy(), rgb_ref, want_lists=False)
    want = RO.blend_background(ref["image"], ref["alpha"], np.ones(3, np.float32))
    assert info["tile_pairs"][0] == ref["D"]
    assert np.array_equal(info["tiles_touched"][0].cpu().numpy(), ref["tiles_touched"])
    assert np.abs(alphas[0, ..., 0].cpu().numpy() - ref["alpha"]).max() <= 1e-5
    assert np.abs(colors[0].cpu().numpy() - want).max() <= 2e-5
    del colors, alphas, ref, want, rgb_ref

    # ---- SplattingCUDA semantics (diff-gaussian-rasterization-style; reference gaussian_renderer.py:29-116): normalised intrinsics,
    # x10 scene rescale in place, near 1, black background, colour + depth
    Kn = torch.tensor([[0.5, 0, 0.5], [0, 0.5 * W / H, 0.5], [0, 0, 1]])
    fov = cs.get_fov(Kn[None])
    tan = (0.5 * fov).tan()[0]
    c2w2 = _pick_camera(means_c, float(tan[0]), float(tan[1]), 0.2)
    gs = Gaussians(means=G_.means.clone(), covariances=G_.covariances.clone(), harmonics=G_.harmonics, opacities=G_.opacities)
    out2 = SplattingCUDA().forward(gs, c2w2[None, None], Kn[None, None], (H, W), render_color=True)
    col, dep = out2["render_color"][0, 0].cpu().numpy(), out2["render_depth"][0, 0].cpu().numpy()
    e = c2w2.clone()
    e[:3, 3] *= 10.0
    proj = cs.get_projection_matrix(torch.tensor([1.0]), torch.tensor([1000.0]), fov[:, 0], fov[:, 1])[0]
    w2c = torch.linalg.inv(e)
    cam2 = raster.make_cam_k2(w2c, proj @ w2c, float(tan[0]), float(tan[1]), e[:3, 3].tolist(), [0, 0, 0], W, H, sh_degree=4)
    cov6 = raster.cov6_from_cov3x3(G_.covariances[0].float().cpu() * 100.0).numpy()
    ref2 = RO.forward(cam2, (means_c * 10.0).numpy(), cov6, G_.opacities[0].float().cpu().numpy(), coeffs.cpu().numpy(), want_lists=False)
    vis2 = int((ref2["tiles_touched"] > 0).sum())
    print(f"[config5] SplattingCUDA semantics: {vis2} of {G} Gaussians visible ({vis2 / G:.3f}), {ref2['D']} tile pairs")
    assert vis2 >= 0.5 * G, f"only {vis2 / G:.3f} of the Gaussians are in view"
    assert float(np.abs(col - np.clip(ref2["image"], 0.0, 1.0)).max()) <= 2e-5
    assert float(np.abs(dep - ref2["depth"]).max()) <= 5e-5 * max(1.0, float(ref2["depth"].max()))
    del model
    torch.cuda.empty_cache()


def test_batch_of_eight_label_maps_are_run_to_run_identical():
    """Round 4 regression: at B = 8 (six streams on four hardware queues, 419 MB mask-probability volume) one forward in four moved ~100
    border pixels of item 0 to a neighbouring segment, with bit-identical logits -- the round-3/4 code object of the argmax kernel still
    does on this tree (tools/pk_hazard_probe.py: 11-21 of 40 forwards): its crossed packed add (v_pk_add_f32 … op_sel:[0,1]) returns wrong low
    halves while the other streams' bf16 MFMAs share the SIMD -- a gfx950 hazard between waves (csrc/postprocess.hip, sample256;
    tools/probes/pk_hazard/xwave2.hip), nothing is stale.  Forty consecutive
    forwards must give identical segmentation / label maps (and logits, and Gaussians)."""
    from golden_utils import default_K, fixture_images
    from siu3r_amd.model import SIU3RModel

    B, S = 8, 512
    g = torch.Generator().manual_seed(11)
    fx_ = fixture_images(S)
    img = torch.cat([fx_, torch.rand(B - 2, 2, 3, S, S, generator=g), fx_.flip(1)]).cuda()
    K = default_K().repeat(B, 1, 1, 1).cuda()
    model = SIU3RModel(_weights(), image_size=(S, S), precision="bf16x3")
    ref, bad = None, []
    with torch.no_grad():
        for it in range(40):
            o = model(img, K, enable_query_class_logit_lift=True)
            torch.cuda.synchronize()
            cur = (o[0].instance_labels.clone(), o[0].semantic_labels.clone(), torch.stack(list(o[2])).clone(), o[1].masks_queries_logits.clone(),
                   o[1].class_queries_logits.clone(), *(getattr(o[0], f).clone() for f in GAUSSIAN_FIELDS))  # (round 5: all six Gaussian fields too)
            if ref is None:
                ref = cur
            elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
                bad.append((it, [int((a != b).sum()) for a, b in zip(cur, ref)]))
    assert not bad, bad
    del model
    torch.cuda.empty_cache()


GAUSSIAN_FIELDS = ("means", "covariances", "harmonics", "opacities", "scales", "rotations")


def test_two_hundred_single_pair_forwards_are_identical():
    """Round 5 companion of the batch-of-eight test: every cross-stream hand-off of the forward (six streams, per-chain HIP graphs, the
    panoptic device stage behind Mask2Former, the eager tail) exercised 200 times at B = 1 @256^2: label maps, segmentation, both logit
    tensors and all six Gaussian fields bit-identical to the first forward.  (The round-4 label flake showed up at B = 8 only; round 6
    traced it to a packed add of the compiled argmax loop that miscomputes beside other waves' bf16 MFMAs, not to any hand-off:
    csrc/postprocess.hip, sample256.)"""
    from golden_utils import default_K, fixture_images
    from siu3r_amd.model import SIU3RModel

    S = 256
    img, K = fixture_images(S).cuda(), default_K().cuda()
    model = SIU3RModel(_weights(), image_size=(S, S), precision="bf16x3")
    ref, bad = None, []
    with torch.no_grad():
        for it in range(200):
            o = model(img, K, enable_query_class_logit_lift=True)
            if it % 8 == 0:
                torch.cuda.synchronize()  # (mostly back to back: the next forward is enqueued while this one's tail runs)
            cur = (o[0].instance_labels, o[0].semantic_labels, torch.stack(list(o[2])), o[1].masks_queries_logits, o[1].class_queries_logits,
                   *(getattr(o[0], f) for f in GAUSSIAN_FIELDS))
            if ref is None:
                ref = tuple(t.clone() for t in cur)
            elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
                bad.append((it, [int((a != b).sum()) for a, b in zip(cur, ref)]))
    assert not bad, bad[:8]
    del model
    torch.cuda.empty_cache()
