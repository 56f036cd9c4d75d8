"""HIP tile-binned rasterizer vs the C oracle (oracle/raster_ref.c) on seeded synthetic scenes.
Integer outputs (radii, tiles_touched, per-tile sorted Gaussian lists, n_touched) must be BIT-EXACT; rendered maps
agree to fp32 rounding (both sides use the same operation order and the same polynomial exp)."""
import numpy as np
import pytest
import torch

from scenes import default_K, look_at_camera, random_scene

pytestmark = pytest.mark.gpu


def _k2_cam(H, W, seed, band4=False):
    from siu3r_amd import cuda_splatting as cs, raster

    c2w = look_at_camera(seed)
    K = default_K()[None]
    fov = cs.get_fov(K)
    tan = (0.5 * fov).tan()[0]
    proj = cs.get_projection_matrix(torch.tensor([1.0]), torch.tensor([1000.0]), fov[:, 0], fov[:, 1])[0]
    w2c = torch.linalg.inv(c2w)
    return raster.make_cam_k2(w2c, proj @ w2c, float(tan[0]), float(tan[1]), c2w[:3, 3].tolist(), [0.1, 0.2, 0.3], W, H, sh_degree=4, sh_band4=band4)


def _k3_cam(H, W, seed, near=1.0, far=1000.0):
    from siu3r_amd import raster

    c2w = look_at_camera(seed)
    K = default_K()
    return raster.make_cam_k3(torch.linalg.inv(c2w), K[0, 0] * W, K[1, 1] * H, K[0, 2] * W, K[1, 2] * H, W, H, near_plane=near, far_plane=far)


def _check_lists(st, ref):
    """per-tile ranges and front-to-back Gaussian lists (materialised by siu3r_raster_tile_lists: wave ballot + prefix popcount over the
    depth-ordered coarse bins) against the oracle's (tile, depth, id) sort"""
    T1 = ref["tile_start"].shape[0]
    assert st["D"] == ref["D"]
    assert np.array_equal(st["tile_start"].cpu().numpy()[:T1], ref["tile_start"]), "tile ranges differ"
    assert np.array_equal(st["ids"].cpu().numpy()[: ref["D"]], ref["ids"]), "per-tile sorted Gaussian lists differ"


@pytest.mark.parametrize("shape", [(192, 256), (190, 250)], ids=["192x256", "ragged190x250"])
@pytest.mark.parametrize("band4", [False, True], ids=["sh3", "sh4"])
def test_k2_rgb_depth_ntouched(shape, band4):
    from oracle import raster_oracle as RO
    from siu3r_amd import raster

    H, W = shape
    means, cov, opac, sh = random_scene(20000, seed=3)
    means[:50, 2] = -1.0   # behind the camera -> culled
    means[50:60, 2] = 0.15  # inside the 0.2 near cull of the CUDA kernel
    cam = _k2_cam(H, W, seed=1, band4=band4)
    cov6 = raster.cov6_from_cov3x3(cov)
    shs = sh.permute(0, 2, 1).contiguous()
    ref = RO.forward(cam, means.numpy(), cov6.numpy(), opac.numpy(), shs.numpy())
    out = raster.rasterize_k2(cam, means.cuda(), cov6.cuda(), shs.cuda(), opac.cuda())
    assert np.array_equal(out["radii"].cpu().numpy(), ref["radii"]), "radii differ"
    assert np.array_equal(out["state"]["tiles_touched"].cpu().numpy(), ref["tiles_touched"]), "tiles_touched differ"
    _check_lists(out["state"], ref)
    assert np.array_equal(out["n_touched"].cpu().numpy(), ref["n_touched"]), "n_touched differs"
    for name, got, want in (("image", out["image"], ref["image"]), ("depth", out["depth"], ref["depth"]), ("opacity", out["opacity"], ref["alpha"])):
        err = float(np.abs(got.cpu().numpy() - want).max())
        print(f"[parity] k2 {name}: max abs err {err:.2e} (max |ref| {np.abs(want).max():.2e})")
        assert err <= 2e-6 * max(1.0, float(np.abs(want).max()))
    # determinism: unique 64-bit keys -> identical lists run to run
    out2 = raster.rasterize_k2(cam, means.cuda(), cov6.cuda(), shs.cuda(), opac.cuda())
    assert torch.equal(out["state"]["ids"][: ref["D"]], out2["state"]["ids"][: ref["D"]])
    assert torch.equal(out["image"], out2["image"])


@pytest.mark.parametrize("channels", [37, 64, 5])
def test_k3_channels(channels):
    from oracle import raster_oracle as RO
    from siu3r_amd import raster

    H, W = 160, 208
    means, cov, opac, _ = random_scene(12000, seed=5, depth=(0.5, 12.0))
    feats = torch.rand(12000, channels, generator=torch.Generator().manual_seed(9))
    cam = _k3_cam(H, W, seed=2, near=1.0, far=10.0)  # near/far planes cull part of the scene
    cov6 = raster.cov6_from_cov3x3(cov)
    ref = RO.forward(cam, means.numpy(), cov6.numpy(), opac.numpy(), feats.numpy())
    out = raster.rasterize_k3(cam, means.cuda(), cov6.cuda(), opac.cuda(), feats.cuda())
    assert np.array_equal(out["radii"].cpu().numpy(), ref["radii"])
    _check_lists(out["state"], ref)
    err = float(np.abs(out["colors"].cpu().numpy() - ref["image"]).max())
    erra = float(np.abs(out["alphas"].cpu().numpy() - ref["alpha"]).max())
    print(f"[parity] k3 C={channels}: colours max abs err {err:.2e}, alphas {erra:.2e}")
    assert err <= 2e-6 and erra <= 2e-6


@pytest.fixture
def feat_form():
    """raster.tune(0, ...) for one test: 1 = the 32-channel kernel everywhere, 0 = the shipped default (matrix-core form where it applies)"""
    from siu3r_amd import raster

    yield lambda v: raster.tune(0, v)
    raster.tune(0, 0)


@pytest.mark.parametrize("channels", [168, 3 * 64 + 8, 130, 105, 63, 32, 21])
def test_k3_all_channel_composite_equals_the_chunked_one(channels, feat_form):
    """The matrix-core list composite (composite_feat5_kernel: alpha / transmittance once per (pixel, entry) for all channels, lists cut to
    the wave's 8 x 8 quadrant, the blend as rank-2 v_mfma_f32_32x32x2_f32 updates -- exact f32, accumulating like the oracle's fmaf chain --
    with records and feature rows staged in a wave-private LDS ring by LDS-DMA) against the 32-channel-chunk kernel it replaces (raster.tune(0, 1)):
    bit-identical maps; and against the C oracle, on a ragged frame with several views in one call: q x 21 = 168 logit channels (6 blocks,
    the last one a shifted window), 200 (two chunks, the second shifted back), 130, 105 and 63 (q x 21 with odd q: feature rows that are only
    4-byte aligned), 32 and 21 (below one block: the 32-channel kernel serves it)."""
    from oracle import raster_oracle as RO
    from siu3r_amd import raster

    H, W, G = 152, 200, 9000
    means, cov, opac, _ = random_scene(G, seed=15, depth=(0.5, 9.0), scale=(0.01, 0.12))
    opac[:300] = 0.002                      # below alpha_min: never blend (the extent test drops them at load time)
    opac[300:600] = 0.999                   # clamp at alpha_max, early termination behind them
    feats = torch.randn(G, channels, generator=torch.Generator().manual_seed(19))
    cams = [_k3_cam(H, W, seed=s_, near=1.0, far=9.0) for s_ in (2, 4, 6)]
    cov6 = raster.cov6_from_cov3x3(cov)
    args = (cams, means.cuda(), cov6.cuda(), opac.cuda(), feats.cuda())
    feat_form(1)
    old = raster.rasterize_views_k3(*args)
    feat_form(0)
    new = raster.rasterize_views_k3(*args)
    assert torch.equal(new["colors"], old["colors"]) and torch.equal(new["alphas"], old["alphas"]), float((new["colors"] - old["colors"]).abs().max())
    for v in (0, 2):
        ref = RO.forward(cams[v], means.numpy(), cov6.numpy(), opac.numpy(), feats.numpy(), want_lists=False)
        assert ref["D"] > 3000
        assert float(np.abs(new["colors"][v].cpu().numpy() - ref["image"]).max()) <= 5e-6 * max(1.0, float(np.abs(ref["image"]).max()))
        assert float(np.abs(new["alphas"][v].cpu().numpy() - ref["alpha"]).max()) <= 2e-6


def test_k3_all_channel_composite_on_needles(feat_form):
    """Long thin splats seen diagonally (1 : 3000 axes, hundreds of pixels long): the conic's a c and b^2 agree to many digits, and the
    per-quadrant footprint test divides by their difference.  The quadrant lists must still keep every entry that blends: maps
    bit-identical to the kernel that walks the uncut tile lists."""
    from siu3r_amd import raster

    H, W, G, channels = 152, 200, 1500, 64
    g = torch.Generator().manual_seed(23)
    means, cov, opac, _ = random_scene(G, seed=29, depth=(1.0, 6.0))
    ang = torch.rand(G, generator=g) * 3.14159265
    d = torch.stack((torch.cos(ang), torch.sin(ang), 0.2 * torch.randn(G, generator=g)), -1)
    d = d / d.norm(dim=-1, keepdim=True)
    long = 0.3 + 2.7 * torch.rand(G, generator=g)
    cov = (long * long)[:, None, None] * d[:, :, None] * d[:, None, :] + 1e-6 * torch.eye(3)
    opac = 0.3 + 0.69 * torch.rand(G, generator=g)
    feats = torch.randn(G, channels, generator=g)
    cams = [_k3_cam(H, W, seed=s_, near=0.5, far=9.0) for s_ in (2, 4)]
    args = (cams, means.cuda(), raster.cov6_from_cov3x3(cov).cuda(), opac.cuda(), feats.cuda())
    feat_form(1)
    old = raster.rasterize_views_k3(*args)
    assert float(old["alphas"].mean()) > 0.5 and int(old["state"]["D"]) > 20 * G
    feat_form(0)
    new = raster.rasterize_views_k3(*args)
    assert torch.equal(new["colors"], old["colors"]) and torch.equal(new["alphas"], old["alphas"]), float((new["colors"] - old["colors"]).abs().max())


def test_k3_composite_with_a_non_finite_feature_row(feat_form):
    """include/siu3r_hip.h, siu3r_raster_composite_feat_ws: the matrix-core form multiplies every listed row into every pixel of the
    quadrant (weight 0 where the entry does not reach), so an inf / NaN feature poisons pixels the 32-channel kernel leaves alone.  Pinned:
    the 32-channel kernel (raster.tune(0, 1), what compat.gsplat.rasterization(finite_features=False) selects) keeps the damage to the
    pixels the Gaussian blends into; the matrix-core form's NaN set contains that set, and everything outside ITS NaN set is bit-identical."""
    from siu3r_amd import raster

    H, W, G, channels = 96, 128, 2000, 64
    means, cov, opac, _ = random_scene(G, seed=31, depth=(1.0, 6.0), scale=(0.02, 0.1))
    feats = torch.randn(G, channels, generator=torch.Generator().manual_seed(3))
    bad = 777
    feats[bad, 5] = float("inf")
    cams = [_k3_cam(H, W, seed=2, near=0.5, far=9.0)]
    args = (cams, means.cuda(), raster.cov6_from_cov3x3(cov).cuda(), opac.cuda())
    feat_form(1)
    clean = raster.rasterize_views_k3(*args, torch.nan_to_num(feats, posinf=0.0).cuda())["colors"]
    old = raster.rasterize_views_k3(*args, feats.cuda())["colors"]
    feat_form(0)
    new = raster.rasterize_views_k3(*args, feats.cuda())["colors"]
    bad_old, bad_new = ~torch.isfinite(old).all(-1), ~torch.isfinite(new).all(-1)
    assert int(bad_old.sum()) > 0, "the scene must show the bad Gaussian"
    assert bool((bad_old & ~bad_new).sum() == 0) and int(bad_new.sum()) >= int(bad_old.sum())
    assert int(bad_new.sum()) < H * W // 2                       # (a neighbourhood, not the frame)
    assert torch.equal(new[~bad_new], old[~bad_new])
    ch = [c for c in range(channels) if c != 5]
    assert torch.equal(old[..., ch], clean[..., ch])             # the 32-channel kernel: only channel 5, only where the Gaussian blends


def test_k3_all_channel_composite_empty_and_culled():
    """The N-channel path with nothing to blend: no Gaussians at all, and all of them behind the camera -- zero maps, zero alphas, no
    list entries, from the quadrant-list kernels like from the 32-channel one."""
    from siu3r_amd import raster

    H, W, C = 72, 104, 168
    cams = [_k3_cam(H, W, seed=s_, near=0.5, far=9.0) for s_ in (1, 3)]
    z = lambda *s: torch.zeros(*s, device="cuda")
    out = raster.rasterize_views_k3(cams, z(0, 3), z(0, 6), z(0), z(0, C))
    assert out["colors"].shape == (2, H, W, C) and float(out["colors"].abs().max()) == 0.0 and float(out["alphas"].abs().max()) == 0.0
    means, cov, opac, _ = random_scene(300, seed=5)
    far_away = means.clone()
    far_away[:, 2] = 1e6   # beyond every camera's far plane, whichever way it looks
    feats = torch.randn(300, C, generator=torch.Generator().manual_seed(2))
    out = raster.rasterize_views_k3(cams, far_away.cuda(), raster.cov6_from_cov3x3(cov).cuda(), opac.cuda(), feats.cuda())
    assert int(out["state"]["D"]) == 0 and float(out["colors"].abs().max()) == 0.0 and float(out["alphas"].abs().max()) == 0.0
    assert int(out["radii"].abs().sum()) == 0


def test_empty_and_all_culled():
    from siu3r_amd import raster

    cam = _k2_cam(64, 64, seed=0)
    z = lambda *s: torch.zeros(*s, device="cuda")
    out = raster.rasterize_k2(cam, z(0, 3), z(0, 6), z(0, 25, 3), z(0))
    assert out["image"].shape == (3, 64, 64) and torch.allclose(out["image"][0], torch.full((64, 64), 0.1, device="cuda"))
    means, cov, opac, sh = random_scene(100, seed=1)
    means[:, 2] = -5.0
    out = raster.rasterize_k2(cam, means.cuda(), raster.cov6_from_cov3x3(cov).cuda(), sh.permute(0, 2, 1).contiguous().cuda(), opac.cuda())
    assert int(out["state"]["D"]) == 0 and int(out["radii"].abs().sum()) == 0 and float(out["opacity"].abs().max()) == 0.0


def test_splatting_cuda_mirror_and_lifting():
    """SplattingCUDA.forward signature / in-place x10 rescale quirk / output layout, then lifting vs the oracle."""
    from oracle import siu3r_oracle as O
    from siu3r_amd.gaussian_renderer import SplattingCUDA, lift_query_class_logits
    from siu3r_amd.gaussians_types import Gaussians

    H, W, G, q, c = 64, 80, 5000, 3, 21
    means, cov, opac, sh = random_scene(G, seed=7, spread=0.15, depth=(0.15, 0.8), scale=(0.001, 0.01))
    gen = torch.Generator().manual_seed(11)
    qcl = torch.rand(G, q, c, generator=gen) * torch.rand(G, q, 1, generator=gen)
    g = Gaussians(means=means[None].cuda(), covariances=cov[None].cuda(), harmonics=sh[None].cuda(), opacities=opac[None].cuda(),
                  scales=None, rotations=None)
    g.seg_query_class_logits = [qcl.cuda()]
    ext = torch.stack([look_at_camera(0, 0.02), look_at_camera(1, 0.02)])[None]
    K = default_K()[None, None].repeat(1, 2, 1, 1)
    m0 = g.means.clone()
    out = SplattingCUDA().forward(g, ext, K, (H, W), render_color=True, render_qc_logits=True)
    assert torch.allclose(g.means, m0 * 10.0) and out["render_color"].shape == (1, 2, 3, H, W) and out["render_depth"].shape == (1, 2, H, W)
    assert float(out["render_color"].min()) >= 0.0 and float(out["render_color"].max()) <= 1.0
    rq = out["render_qc_logits"][0]
    assert rq.shape == (2, q, c, H, W)
    scores = [[0.9, 0.8, 0.7]]
    sem, ins, infos = lift_query_class_logits(out["render_qc_logits"], scores, num_queries=100, label_ids_to_fuse=(0, 1))
    rs, ri, rinfo = O.lift_ids(rq.cpu(), scores[0])
    assert sem.dtype == torch.int64 and torch.equal(sem[0].cpu(), rs) and torch.equal(ins[0].cpu(), ri)
    assert infos[0] == rinfo


@pytest.mark.parametrize("degree", [0, 3, 4])
def test_viewer_render_semantics(degree):
    """rasterize_splats (reference viewer.py:301-336: quats / log-scales / logit-opacities / SH, white background,
    radius_clip 0.1) vs the C oracle restatement: covariances and view-dependent colours bit-exact, tile lists exact."""
    from oracle import raster_oracle as RO
    from siu3r_amd import raster
    from siu3r_amd.gaussian_renderer import rasterize_splats

    G, H, W = 6000, 144, 208
    means, cov, opac, sh = random_scene(G, seed=7)
    g = torch.Generator().manual_seed(8)
    quats = torch.randn(G, 4, generator=g) * 2.0          # un-normalised, as the viewer feeds them
    lscale = torch.log(0.01 + 0.1 * torch.rand(G, 3, generator=g))
    logit = torch.logit(opac.clamp(0.02, 0.98))
    coeffs = sh.permute(0, 2, 1).contiguous()               # [G, 25, 3]
    c2w = look_at_camera(3)
    K = default_K().clone()
    K[0] *= W
    K[1] *= H
    splats = dict(means=means.cuda(), quats=quats.cuda(), scales=lscale.cuda(), opacities=logit.cuda(), sh0=coeffs[:, :1].cuda(), shN=coeffs[:, 1:].cuda())
    colors, alphas, info = rasterize_splats(splats, c2w[None], K[None], W, H, sh_degree=degree, radius_clip=0.1)
    # oracle chain
    cov6_ref = RO.quat_scale_to_cov6(quats.numpy(), torch.exp(lscale).numpy())
    cov6 = raster.quat_scale_to_cov6(quats.cuda(), torch.exp(lscale).cuda()).cpu().numpy()
    assert np.array_equal(cov6, cov6_ref), f"cov6 differs: {np.abs(cov6 - cov6_ref).max()}"
    rgb_ref = RO.sh_eval(degree, means.numpy(), c2w[:3, 3].numpy(), coeffs.numpy())
    rgb = raster.sh_eval(means.cuda(), c2w[:3, 3].tolist(), coeffs.cuda(), degree).cpu().numpy()
    assert np.abs(rgb - rgb_ref).max() <= 1e-6, np.abs(rgb - rgb_ref).max()
    cam = raster.make_cam_k3(torch.linalg.inv(c2w), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H, radius_clip=0.1)
    ref = RO.forward(cam, means.numpy(), cov6_ref, torch.sigmoid(logit).numpy(), rgb_ref)
    want = RO.blend_background(ref["image"], ref["alpha"], np.ones(3, np.float32))
    assert info["tile_pairs"][0] == ref["D"]
    assert np.array_equal(info["tiles_touched"][0].cpu().numpy(), ref["tiles_touched"])
    assert np.abs(alphas[0, ..., 0].cpu().numpy() - ref["alpha"]).max() <= 2e-6
    assert np.abs(colors[0].cpu().numpy() - want).max() <= 5e-6
    assert float(colors.min()) >= 0.0 and ref["D"] > 1000


def test_prepared_splat_cache_hits_views_and_sheds_finalizers():
    """rasterize_splats' one-entry cache of the view-independent preparation (round-5 advisor): another view of the same live storage
    hits, an in-place edit rebuilds WITHOUT piling finalizers onto the long-lived tensors, and the entry dies with its sources."""
    import gc

    from siu3r_amd import gaussian_renderer as GR

    G, H, W = 800, 64, 80
    means, cov, opac, sh = random_scene(G, seed=9)
    g = torch.Generator().manual_seed(10)
    coeffs = sh.permute(0, 2, 1).contiguous()
    splats = dict(means=means.cuda(), quats=torch.randn(G, 4, generator=g).cuda(), scales=torch.log(0.02 + 0.1 * torch.rand(G, 3, generator=g)).cuda(),
                  opacities=torch.logit(opac.clamp(0.02, 0.98)).cuda(), sh0=coeffs[:, :1].cuda(), shN=coeffs[:, 1:].cuda())
    c2w, K = look_at_camera(3), default_K().clone()
    K[0] *= W
    K[1] *= H
    GR.release_prepared_splats()
    a, _, _ = GR.rasterize_splats(splats, c2w[None], K[None], W, H, sh_degree=2)
    entry = GR._PREPARED["entry"]
    GR.rasterize_splats(splats, c2w[None], K[None], W, H, sh_degree=2)
    assert GR._PREPARED["entry"] is entry
    views = {k: v.detach() for k, v in splats.items()}        # fresh tensor objects over the same storage: still a hit
    b, _, _ = GR.rasterize_splats(views, c2w[None], K[None], W, H, sh_degree=2)
    assert GR._PREPARED["entry"] is entry and torch.equal(a, b)
    old_fins = list(GR._PREPARED["finalizers"])
    for _ in range(5):                                            # in-place edits between frames: rebuilds, one finalizer set at a time
        splats["means"].add_(0.01)
        GR.rasterize_splats(splats, c2w[None], K[None], W, H, sh_degree=2)
        assert GR._PREPARED["entry"] is not entry and len(GR._PREPARED["finalizers"]) == 6
        entry = GR._PREPARED["entry"]
    assert not any(f.alive for f in old_fins)
    del splats, views
    gc.collect()
    assert not GR._PREPARED


def test_capacity_overflow_is_detected_and_retried():
    """The coarse-bin entry buffers and the tile-list buffers are sized by bounds and checked after the frame: an undersized bound
    never writes out of range, is reported (check_overflow=False + verify()), and by default the call is repeated with the exact size."""
    from siu3r_amd import raster

    G, H, W = 4000, 96, 128
    means, cov, opac, sh = random_scene(G, seed=11, scale=(0.05, 0.2))
    cam = _k3_cam(H, W, 2)
    args = (cam, means.cuda(), raster.cov6_from_cov3x3(cov.cuda()), opac.cuda(), torch.rand(G, 5).cuda())
    full = raster.rasterize_k3(*args)
    D, E = full["state"]["D"], full["state"]["E"]
    assert D > 2000 and E >= full["state"]["Gv"] > 0
    # undersized entry buffer, no check: flagged, nothing written beyond the buffer (the guard element stays intact)
    small = raster.rasterize_k3(*args, entry_capacity=E // 2, check_overflow=False)
    torch.cuda.synchronize()
    with pytest.raises(raster.RasterOverflow, match="overflowed"):
        small["state"].verify()
    # default: the call is repeated with the exact count and gives the same answer, lists included
    retry = raster.rasterize_k3(*args, entry_capacity=E // 2, pair_capacity=D // 3)
    assert retry["state"]["cap_e"] == E and retry["state"]["cap_d"] == D
    assert torch.equal(retry["colors"], full["colors"]) and torch.equal(retry["alphas"], full["alphas"])
    assert torch.equal(retry["state"]["ids"][:D], full["state"]["ids"][:D])
    k2 = _k2_cam(H, W, seed=2)
    shs = sh.permute(0, 2, 1).contiguous().cuda()
    a = raster.rasterize_k2(k2, args[1], args[2], shs, args[3])
    b = raster.rasterize_k2(k2, args[1], args[2], shs, args[3], entry_capacity=a["state"]["E"] // 3)
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["n_touched"], b["n_touched"])


def test_deferred_overflow_check_poisons_and_reports():
    """check_overflow="deferred": no synchronisation in the call; an overflowing view comes back as NaN (never subtly wrong), the next
    deferred call / check_pending() raises RasterOverflow and remembers the needed capacity, and the repeated call equals the
    synchronous result bit for bit.  SplattingCUDA(deferred_overflow_check=True) is the same mechanism behind the reference API."""
    from siu3r_amd import raster

    G, H, W = 4000, 96, 128
    means, cov, opac, sh = random_scene(G, seed=11, scale=(0.05, 0.2))
    k2 = [_k2_cam(H, W, seed=2), _k2_cam(H, W, seed=3)]
    shs = sh.permute(0, 2, 1).contiguous().cuda()
    a = (means.cuda(), raster.cov6_from_cov3x3(cov.cuda()), shs, opac.cuda())
    raster.check_pending()
    full = raster.rasterize_views_k2(k2, *a)
    E = max(full["state"].totals(2))
    ok = raster.rasterize_views_k2(k2, *a, check_overflow="deferred")
    raster.check_pending()  # nothing overflowed: silent
    assert torch.equal(ok["image"], full["image"]) and torch.equal(ok["n_touched"], full["n_touched"])
    bad = raster.rasterize_views_k2(k2, *a, entry_capacity=E // 3, check_overflow="deferred")
    assert len(raster._PENDING) == 1
    torch.cuda.synchronize()
    assert torch.isnan(bad["image"]).all() and torch.isnan(bad["depth"]).all() and torch.isnan(bad["opacity"]).all()
    with pytest.raises(raster.RasterOverflow, match="deferred") as exc:
        raster.check_pending()
    assert exc.value.call_id == bad["call_id"] != ok["call_id"]  # the exception names the call whose result is to be discarded
    assert not raster._PENDING
    again = raster.rasterize_views_k2(k2, *a, check_overflow="deferred")  # the remembered capacity covers the scene now
    raster.check_pending()
    assert again["state"]["cap_e"] >= E and torch.equal(again["image"], full["image"]) and torch.equal(again["depth"], full["depth"])
    # the next deferred call is the implicit check point of the previous one (the exception carries the PREVIOUS call's id)
    prev = raster.rasterize_views_k2(k2, *a, entry_capacity=E // 3, check_overflow="deferred")
    torch.cuda.synchronize()
    with pytest.raises(raster.RasterOverflow) as exc:
        raster.rasterize_views_k2(k2, *a, check_overflow="deferred")
    assert exc.value.call_id == prev["call_id"]
    raster.check_pending()
    # the N-channel composite has no NaN marker: a deferred check is refused there
    with pytest.raises(ValueError, match="deferred"):
        raster.rasterize_views_k3([_k3_cam(H, W, 2)], a[0], a[1], a[3], torch.rand(G, 40).cuda(), check_overflow="deferred")


def test_views_batched_equals_view_by_view():
    """V cameras in one call (blockIdx.y = view) == V single-view calls, bit for bit (K2 and K3)."""
    from siu3r_amd import raster

    G, H, W = 9000, 112, 144
    means, cov, opac, sh = random_scene(G, seed=21)
    cov6 = raster.cov6_from_cov3x3(cov).cuda()
    shs = sh.permute(0, 2, 1).contiguous().cuda()
    cams = [_k2_cam(H, W, seed=s_) for s_ in range(3)]
    allv = raster.rasterize_views_k2(cams, means.cuda(), cov6, shs, opac.cuda())
    for i, cam in enumerate(cams):
        one = raster.rasterize_k2(cam, means.cuda(), cov6, shs, opac.cuda())
        for k in ("image", "depth", "opacity", "radii", "n_touched"):
            assert torch.equal(allv[k][i], one[k]), (k, i)
    feats = torch.rand(G, 40, generator=torch.Generator().manual_seed(4)).cuda()
    cams3 = [_k3_cam(H, W, seed=s_, near=0.5, far=50.0) for s_ in range(3)]
    allv = raster.rasterize_views_k3(cams3, means.cuda(), cov6, opac.cuda(), feats)
    for i, cam in enumerate(cams3):
        one = raster.rasterize_k3(cam, means.cuda(), cov6, opac.cuda(), feats)
        assert torch.equal(allv["colors"][i], one["colors"]) and torch.equal(allv["alphas"][i], one["alphas"])
        D = one["state"]["D"]
        assert torch.equal(allv["state"]["ids_all"][i, :D], one["state"]["ids"][:D])


def test_crowded_tile_and_equal_depths():
    """More than 8192 Gaussians on one tile (the old per-tile LDS sort's limit) and many exactly equal depths (ties are broken by
    the Gaussian index): lists, n_touched and maps against the oracle."""
    from oracle import raster_oracle as RO
    from siu3r_amd import raster

    G, H, W = 12000, 64, 80
    g = torch.Generator().manual_seed(5)
    means = torch.stack(((torch.rand(G, generator=g) - 0.5) * 0.05, (torch.rand(G, generator=g) - 0.5) * 0.05,
                         2.0 + torch.randint(0, 40, (G,), generator=g).float() * 0.25), -1)  # 40 distinct depths -> ~300-way ties
    s = 0.002 + 0.004 * torch.rand(G, 3, generator=g)
    cov = torch.diag_embed(s * s)
    opac = 0.01 + 0.05 * torch.rand(G, generator=g)  # faint: pixels do not saturate, the whole list is blended
    sh = (torch.rand(G, 3, 25, generator=g) - 0.5)
    c2w = torch.eye(4)
    from siu3r_amd import cuda_splatting as cs

    K = default_K()[None]
    fov = cs.get_fov(K)
    tan = (0.5 * fov).tan()[0]
    proj = cs.get_projection_matrix(torch.tensor([1.0]), torch.tensor([1000.0]), fov[:, 0], fov[:, 1])[0]
    cam = raster.make_cam_k2(torch.eye(4), proj, float(tan[0]), float(tan[1]), [0, 0, 0], [0, 0, 0], W, H, sh_degree=4)
    cov6 = raster.cov6_from_cov3x3(cov)
    shs = sh.permute(0, 2, 1).contiguous()
    ref = RO.forward(cam, means.numpy(), cov6.numpy(), opac.numpy(), shs.numpy())
    assert int(np.diff(ref["tile_start"]).max()) > 8192, "scene is not crowded enough"
    out = raster.rasterize_k2(cam, means.cuda(), cov6.cuda(), shs.cuda(), opac.cuda())
    _check_lists(out["state"], ref)
    assert np.array_equal(out["n_touched"].cpu().numpy(), ref["n_touched"])
    assert float(np.abs(out["image"].cpu().numpy() - ref["image"]).max()) <= 5e-6
    assert float(np.abs(out["opacity"].cpu().numpy() - ref["alpha"]).max()) <= 5e-6


def test_k2_non_finite_colour_stays_inside_its_footprint():
    """The RGB composite evaluates alpha branch-free for every lane of a quadrant (round 6) but blends inside ONE predicated block: a
    Gaussian whose precomputed colour is not finite may only reach the pixels it blends into -- every other pixel of the frame,
    including those of the same 8 x 8 quadrants, keeps the bits of the render without that Gaussian's colour (a zero-weight
    multiply-add over all lanes would turn 0 * inf into NaN there)."""
    from siu3r_amd import raster

    H, W = 96, 128
    means, cov, opac, sh = random_scene(3000, seed=21)
    cov6 = raster.cov6_from_cov3x3(cov)
    g = torch.Generator().manual_seed(22)
    col = torch.rand(3000, 1, 3, generator=g)
    cam = _k2_cam(H, W, seed=2)
    cam.sh_degree = -1  # precomputed colours, blended as given
    base = raster.rasterize_k2(cam, means.cuda(), cov6.cuda(), col.cuda(), opac.cuda())
    nt = base["n_touched"].cpu().numpy()
    cand = np.nonzero((nt > 4) & (nt < 200))[0]
    assert cand.size > 10
    bad = int(cand[cand.size // 2])
    col2 = col.clone()
    col2[bad, 0, 1] = float("inf")
    out = raster.rasterize_k2(cam, means.cuda(), cov6.cuda(), col2.cuda(), opac.cuda())
    img, ref = out["image"].cpu().numpy(), base["image"].cpu().numpy()
    hit = ~np.isfinite(img).all(0)
    rad = int(np.asarray(base["radii"].cpu().numpy()).reshape(-1, 2)[bad].max())
    ys, xs = np.nonzero(hit)
    assert hit.sum() > 0 and ys.max() - ys.min() <= 2.3 * rad + 3 and xs.max() - xs.min() <= 2.3 * rad + 3, (int(hit.sum()), rad)  # (radii are 3 sigma; alpha >= 1/255 reaches 3.33 sigma)
    blocks = hit[: H // 8 * 8, : W // 8 * 8].reshape(H // 8, 8, W // 8, 8).sum((1, 3))
    assert ((blocks > 0) & (blocks < 64)).any(), "the poisoned pixels are whole 8 x 8 quadrants"
    assert np.array_equal(img[:, ~hit], ref[:, ~hit]), "a pixel outside the footprint changed"
    assert np.array_equal(out["n_touched"].cpu().numpy(), nt) and torch.equal(out["depth"], base["depth"])


@pytest.mark.parametrize("post", [True, False], ids=["post_blend", "pre_blend"])
def test_n_touched_gate_is_a_parameter(post):
    """n_touched counts a pixel while the transmittance after (MonoGS fork: `test_T > 0.5f`) or before the blend exceeds 0.5:
    an unpinned constant of the fork, exposed as raster_cam.nt_post_blend on both sides."""
    from oracle import raster_oracle as RO
    from siu3r_amd import raster

    means, cov, opac, sh = random_scene(8000, seed=13)
    cam = _k2_cam(96, 128, seed=4)
    cam.nt_post_blend = int(post)
    cov6 = raster.cov6_from_cov3x3(cov)
    shs = sh.permute(0, 2, 1).contiguous()
    ref = RO.forward(cam, means.numpy(), cov6.numpy(), opac.numpy(), shs.numpy())
    out = raster.rasterize_k2(cam, means.cuda(), cov6.cuda(), shs.cuda(), opac.cuda())
    assert np.array_equal(out["n_touched"].cpu().numpy(), ref["n_touched"])
    assert int(ref["n_touched"].sum()) > 0


def test_splatting_cuda_colour_against_oracle():
    """SplattingCUDA.forward (reference gaussian_renderer.py:29-116) end to end: x10 in-place rescale, near = 1, fov / projection from the
    normalised intrinsics, all views of a batch item in one rasterizer call, clamp -- rendered colour and depth against the C oracle fed
    with independently prepared cameras."""
    from oracle import raster_oracle as RO
    from siu3r_amd import cuda_splatting as cs, raster
    from siu3r_amd.gaussian_renderer import SplattingCUDA
    from siu3r_amd.gaussians_types import Gaussians

    H, W, G, V = 96, 128, 6000, 3
    means, cov, opac, sh = random_scene(G, seed=17, spread=0.15, depth=(0.15, 0.8), scale=(0.001, 0.01))
    g = Gaussians(means=means[None].cuda(), covariances=cov[None].cuda(), harmonics=sh[None].cuda(), opacities=opac[None].cuda(), scales=None, rotations=None)
    ext = torch.stack([look_at_camera(s_, 0.02) for s_ in range(V)])[None]
    K = default_K()[None, None].repeat(1, V, 1, 1)
    out = SplattingCUDA().forward(g, ext, K, (H, W), render_color=True)
    col, dep = out["render_color"][0].cpu().numpy(), out["render_depth"][0].cpu().numpy()
    cov6 = raster.cov6_from_cov3x3(cov * 100.0).numpy()
    shs = sh.permute(0, 2, 1).contiguous().numpy()
    for v in range(V):
        e = ext[0, v].clone()
        e[:3, 3] *= 10.0
        fov = cs.get_fov(K[0, v][None])
        tan = (0.5 * fov).tan()[0]
        proj = cs.get_projection_matrix(torch.tensor([1.0]), torch.tensor([1000.0]), fov[:, 0], fov[:, 1])[0]
        w2c = torch.linalg.inv(e)
        cam = raster.make_cam_k2(w2c, proj @ w2c, float(tan[0]), float(tan[1]), e[:3, 3].tolist(), [0, 0, 0], W, H, sh_degree=4)
        ref = RO.forward(cam, (means * 10.0).numpy(), cov6, opac.numpy(), shs, want_lists=False)
        assert ref["D"] > 1000
        assert float(np.abs(col[v] - np.clip(ref["image"], 0.0, 1.0)).max()) <= 5e-6
        assert float(np.abs(dep[v] - ref["depth"]).max()) <= 5e-5 * max(1.0, float(ref["depth"].max()))


def _splat_inputs(seed=17, H=96, W=128, G=6000, V=3, q=3):
    from siu3r_amd.gaussians_types import Gaussians

    means, cov, opac, sh = random_scene(G, seed=seed, spread=0.15, depth=(0.15, 0.8), scale=(0.001, 0.01))
    qcl = torch.randn(G, q, 21, generator=torch.Generator().manual_seed(seed + 1))
    mk = lambda: Gaussians(means=means[None].cuda(), covariances=cov[None].cuda(), harmonics=sh[None].cuda(), opacities=opac[None].cuda(), scales=None,
                           rotations=None, seg_query_class_logits=[qcl.cuda()])
    ext = torch.stack([look_at_camera(s_, 0.02) for s_ in range(V)])[None]
    K = default_K()[None, None].repeat(1, V, 1, 1)
    return mk, ext, K


def test_splatting_forward_with_the_camera_tensors_on_the_gpu():
    """SplattingCUDA.forward with extrinsics / intrinsics as GPU tensors (what the pipeline holds, gaussian_renderer.py:29-41): inverse, x10
    translation scale, field of view and projection matrix are derived on the device (siu3r_raster_project_c2w, fp64 rounded once).  The
    finished camera blocks are read back: they agree with the host preparation to fp32 rounding, and the host path fed with exactly those
    blocks renders the identical bits -- i.e. the device route changes the camera's last bits, nothing else.  Against the host route
    itself the images agree to the sensitivity of a frame to such a pose change."""
    import ctypes as C

    from siu3r_amd import _lib, cuda_splatting as cs, raster
    from siu3r_amd.gaussian_renderer import SplattingCUDA

    H, W, V = 96, 128, 3
    mk, ext, K = _splat_inputs(H=H, W=W, V=V)
    host = SplattingCUDA().forward(mk(), ext, K, (H, W), render_color=True, render_qc_logits=True)
    devo = SplattingCUDA().forward(mk(), ext.cuda(), K.cuda(), (H, W), render_color=True, render_qc_logits=True)
    for k in ("render_color", "render_depth"):
        d = (host[k] - devo[k]).abs()
        assert float(d.mean()) <= 2e-6 * max(1.0, float(host[k].abs().max())) and float((d > 1e-3 * max(1.0, float(host[k].abs().max()))).float().mean()) < 1e-3, k
    d = (host["render_qc_logits"][0] - devo["render_qc_logits"][0]).abs()
    assert float(d.mean()) <= 1e-5 and float((d > 1e-3).float().mean()) < 1e-3
    # the camera blocks the device derived, K2 family
    g = mk()
    raster.scale_inplace_(g.covariances, 100.0)
    raster.scale_inplace_(g.means, 10.0)
    near, far = torch.full((V,), 1.0), torch.full((V,), 1000.0)
    bg = torch.zeros(V, 3)
    args = (near, far, (H, W), bg, g.means[0][None].expand(V, -1, -1), g.covariances[0][None].expand(V, -1, -1, -1), g.harmonics[0][None].expand(V, -1, -1, -1),
            g.opacities[0][None].expand(V, -1))
    img_d, dep_d, aux = cs.render_cuda(ext[0].cuda(), K[0].cuda(), *args, return_aux=True, translation_scale=10.0)
    st = aux[0]["state"]
    blocks = st["cams_dev"].cpu().numpy().tobytes()
    cams = []
    for v in range(V):
        c = _lib.RasterCam.from_buffer_copy(blocks[v * C.sizeof(_lib.RasterCam):(v + 1) * C.sizeof(_lib.RasterCam)])
        e = ext[0, v].clone()
        e[:3, 3] *= 10.0
        fov = cs.get_fov(K[0, v][None])
        tan = (0.5 * fov).tan()[0]
        proj = cs.get_projection_matrix(torch.tensor([1.0]), torch.tensor([1000.0]), fov[:, 0], fov[:, 1])[0]
        w2c = torch.linalg.inv(e)
        full = proj @ w2c
        assert np.allclose(np.array(c.w2c), w2c.reshape(-1).numpy(), rtol=2e-6, atol=2e-6)
        assert np.allclose(np.array(c.proj), full.reshape(-1).numpy(), rtol=5e-6, atol=5e-6)
        assert abs(c.tanfovx - float(tan[0])) <= 1e-6 and abs(c.tanfovy - float(tan[1])) <= 1e-6
        assert np.allclose(np.array(c.campos), e[:3, 3].numpy(), rtol=1e-7, atol=0)
        cams.append(c)
    same = raster.rasterize_views_k2(cams, g.means[0], g.covariances[0], g.harmonics[0], g.opacities[0], want_n_touched=True, sh_planar=True)
    assert torch.equal(same["image"], img_d) and torch.equal(same["depth"], dep_d)
    assert torch.equal(same["radii"], aux[0]["radii"]) and torch.equal(same["n_touched"], aux[0]["n_touched"])


def test_splatting_forward_does_not_synchronise():
    """SURVEY section 8(b) "no hidden syncs": with the camera tensors on the GPU and the deferred overflow check, SplattingCUDA.forward
    (colour + depth + the q x 21 logit maps need the synchronous check, so: colour + depth) enqueues its work without a single
    synchronising torch call -- torch.cuda.set_sync_debug_mode("error") raises on .cpu() / .item() / blocking copies."""
    from siu3r_amd.gaussian_renderer import SplattingCUDA

    H, W, V = 96, 128, 3
    mk, ext, K = _splat_inputs(H=H, W=W, V=V)
    ext_d, K_d = ext.cuda(), K.cuda()
    r = SplattingCUDA(deferred_overflow_check=True)
    warm = r.forward(mk(), ext_d, K_d, (H, W), render_color=True)   # (first call: buffer-size hints, pinned staging)
    r.check_pending()
    g = mk()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = r.forward(g, ext_d, K_d, (H, W), render_color=True)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    r.check_pending()
    assert torch.equal(out["render_color"], warm["render_color"]) and torch.equal(out["render_depth"], warm["render_depth"])
    # the host-tensor route does synchronise when handed GPU tensors' values (.cpu()): the mode would catch it
    torch.cuda.set_sync_debug_mode("error")
    try:
        with pytest.raises(RuntimeError):
            ext_d.cpu()
    finally:
        torch.cuda.set_sync_debug_mode("default")


@pytest.mark.parametrize("case", ["a", "b"])
def test_lifting_on_reference_generated_fixtures(case):
    """siu3r_lift_ids (wave-per-pixel reduction, guarded atomicMin) on tests/golden/lifting_{a,b}.npz, the vectors produced by executing the
    reference's own statements (src/pipeline.py:137-193): semantic / instance id maps and the segment table must be exact, including
    the stuff classes that fuse several queries into one id (case b: queries 2 and 4 -> 101)."""
    import json
    import os

    from siu3r_amd.gaussian_renderer import lift_query_class_logits

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(root, "lifting.json")))[case]
    z = np.load(os.path.join(root, f"lifting_{case}.npz"))
    x = torch.from_numpy(z["x"]).cuda()                     # [v, q, c, h, w]
    sem, ins, infos = lift_query_class_logits([x], [meta["scores"]], num_queries=100, label_ids_to_fuse=(0, 1))
    assert np.array_equal(sem[0].cpu().numpy(), z["sem_id"]), "semantic ids differ from the reference's"
    assert np.array_equal(ins[0].cpu().numpy(), z["ins_id"]), "instance ids differ from the reference's"
    assert infos[0] == meta["info"]
    # the channel-last view the renderer hands over (no copy inside lift_query_class_logits) gives the same answer
    xcl = x.permute(0, 3, 4, 1, 2).contiguous().permute(0, 3, 4, 1, 2)
    sem2, ins2, infos2 = lift_query_class_logits([xcl], [meta["scores"]])
    assert torch.equal(sem2, sem) and torch.equal(ins2, ins) and infos2 == infos


def test_lifting_ties_and_stuff_fusion_crafted():
    """Exact ties between queries and between classes (argmax takes the first maximum, torch semantics: pipeline.py:150-158), pixels below
    the semantic threshold (background 0 / instance 0), two queries of the same stuff class fused into id 100 + label, a query that wins
    no pixel (absent from the table) -- against the pinned CPU restatement."""
    from oracle import siu3r_oracle as O
    from siu3r_amd.gaussian_renderer import lift_query_class_logits

    v, q, c, h, w = 2, 6, 21, 12, 20
    x = torch.zeros(v, q, c, h, w)
    x[:, 0, 1, :, :5] = 0.9            # query 0: stuff channel 1 (label 2 -> fused id 102) on the left band
    x[:, 1, 1, :, 5:9] = 0.9           # query 1: the same stuff class next to it -> fused with query 0
    x[:, 2, 7, :, 9:13] = 0.8          # query 2: a thing class
    x[:, 3, 7, :, 9:13] = 0.8          # query 3: exact tie with query 2 on the same pixels -> the lower query index wins
    x[:, 4, 3, :, 13:16] = 0.6
    x[:, 4, 9, :, 13:16] = 0.6         # query 4: exact tie between two classes -> the lower class index wins
    x[:, 5, 11, :, 16:] = 0.2          # query 5: below the 0.3 semantic threshold everywhere -> wins nothing
    x[1, 2, 7, 3, 10] = 0.80000001     # (same fp32 value: still a tie)
    scores = [0.9, 0.8, 0.7, 0.6, 0.5, 0.4]
    sem, ins, infos = lift_query_class_logits([x.cuda()], [scores], num_queries=100, label_ids_to_fuse=(0, 1), sem_threshold=0.3)
    rs, ri, rinfo = O.lift_ids(x, scores)
    assert torch.equal(sem[0].cpu(), rs) and torch.equal(ins[0].cpu(), ri)
    assert infos[0] == rinfo
    ids = {i["id"] for i in infos[0]}
    assert 102 in ids and sum(i["was_fused"] for i in infos[0]) == 2 and all(i["id"] != 4 and i["id"] != 6 for i in infos[0])
    assert int((ins[0] == 0).sum()) == v * h * (w - 16)
