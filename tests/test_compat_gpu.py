"""Seam 2 / seam 3 drop-ins (siu3r_amd/compat) driven with the reference's exact calling convention -- src/models/cuda_splatting.py:62-121
(row-vector view / projection matrices, opacities [G,1], cov3D_precomp [G,6], shs [G,25,3]) and src/models/gaussian_renderer.py:80-106
(gsplat.rasterization with covars, pixel-unit Ks, all views at once) -- against the C oracle and against render_cuda / SplattingCUDA,
which must be bit-identical (same kernels, same parameter blocks)."""
import numpy as np
import pytest
import torch

from scenes import default_K, look_at_camera, random_scene

pytestmark = pytest.mark.gpu


def _reference_style_settings(ext, K, near, far, H, W, bg, degree):
    """the tensor algebra of cuda_splatting.py:69-102, verbatim in spirit: row-vector matrices"""
    from siu3r_amd import cuda_splatting as cs
    from siu3r_amd.compat.diff_gaussian_rasterization import GaussianRasterizationSettings

    fov_x, fov_y = cs.get_fov(K).unbind(-1)
    tan_x, tan_y = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
    proj = cs.get_projection_matrix(near, far, fov_x, fov_y).transpose(1, 2)       # 'b i j -> b j i'
    view = torch.linalg.inv(ext).transpose(1, 2)
    full = view @ proj
    return [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tan_x[i].item(), tanfovy=tan_y[i].item(), bg=bg[i], scale_modifier=1.0,
                                          viewmatrix=view[i], projmatrix=full[i], projmatrix_raw=proj[i], sh_degree=degree, campos=ext[i, :3, 3],
                                          prefiltered=False, debug=False) for i in range(ext.shape[0])]


def test_diff_gaussian_rasterization_dropin():
    from oracle import raster_oracle as RO
    from siu3r_amd import cuda_splatting as cs
    from siu3r_amd.compat.diff_gaussian_rasterization import GaussianRasterizer, make_cam

    H, W, G, V = 96, 128, 5000, 3
    means, cov, opac, sh = random_scene(G, seed=31)
    ext = torch.stack([look_at_camera(s_, 0.05) for s_ in range(V)])
    K = default_K()[None].repeat(V, 1, 1)
    near, far, bg = torch.full((V,), 1.0), torch.full((V,), 1000.0), torch.tensor([[0.1, 0.2, 0.3]]).repeat(V, 1)
    settings = _reference_style_settings(ext, K, near, far, H, W, bg, degree=4)
    shs = sh.permute(0, 2, 1).contiguous().cuda()                                   # 'g xyz n -> g n xyz'
    row, col = torch.triu_indices(3, 3)
    cov6 = cov[:, row, col].cuda()
    # the module's own render_cuda on the same inputs (expanded over the views, as SplattingCUDA passes them)
    imgs, deps, aux = cs.render_cuda(ext, K, near, far, (H, W), bg, means.cuda()[None].expand(V, -1, -1), cov.cuda()[None].expand(V, -1, -1, -1),
                                     sh.cuda()[None].expand(V, -1, -1, -1), opac.cuda()[None].expand(V, -1), return_aux=True)
    for i in range(V):
        image, radii, depth, opacity, n_touched = GaussianRasterizer(settings[i])(
            means3D=means.cuda(), means2D=torch.zeros_like(means).cuda(), shs=shs, colors_precomp=None, opacities=opac.cuda()[..., None],
            cov3D_precomp=cov6, theta=None, rho=None)
        assert image.shape == (3, H, W) and depth.shape == (1, H, W) and opacity.shape == (1, H, W) and radii.shape == (G,) and n_touched.shape == (G,)
        assert radii.dtype == torch.int32 and n_touched.dtype == torch.int32
        ref = RO.forward(make_cam(settings[i]), means.numpy(), cov6.cpu().numpy(), opac.numpy(), shs.cpu().numpy(), want_lists=False)
        assert ref["D"] > 500
        assert np.array_equal(radii.cpu().numpy(), ref["radii"][:, 0]) and np.array_equal(n_touched.cpu().numpy(), ref["n_touched"])
        for got, want in ((image, ref["image"]), (depth[0], ref["depth"]), (opacity[0], ref["alpha"])):
            assert float(np.abs(got.cpu().numpy() - want).max()) <= 2e-6 * max(1.0, float(np.abs(want).max()))
        # the seam and the mirrored render_cuda are the same kernels on the same parameter block
        assert torch.equal(image, imgs[i]) and torch.equal(depth[0], deps[i])
        assert torch.equal(n_touched, aux[0]["n_touched"][i]) and torch.equal(radii, aux[0]["radii"][i][:, 0])
    # precomputed colours (use_sh=False branch, cuda_splatting.py:112): blended as given -- negative / feature-valued colours included
    # (the upstream package does not clamp them; round 4 refused them)
    rgb = torch.rand(G, 3) * 2.0 - 0.7
    image_c, radii_c, depth_c, _, nt_c = GaussianRasterizer(settings[0])(means3D=means.cuda(), means2D=None, shs=None, colors_precomp=rgb.cuda(),
                                                                           opacities=opac.cuda()[..., None], cov3D_precomp=cov6)
    cam0 = make_cam(settings[0]._replace(sh_degree=-1))
    ref = RO.forward(cam0, means.numpy(), cov6.cpu().numpy(), opac.numpy(), rgb[:, None, :].numpy(), want_lists=False)
    assert float(ref["image"].min()) < -0.05, "the case must exercise negative colours"
    assert float(np.abs(image_c.cpu().numpy() - ref["image"]).max()) <= 5e-6
    assert np.array_equal(radii_c.cpu().numpy(), ref["radii"][:, 0]) and np.array_equal(nt_c.cpu().numpy(), ref["n_touched"])
    with pytest.raises(Exception):
        GaussianRasterizer(settings[0])(means3D=means.cuda(), means2D=None, shs=shs, colors_precomp=rgb.cuda(), opacities=opac.cuda()[..., None], cov3D_precomp=cov6)


def test_gsplat_rasterization_dropin():
    from oracle import raster_oracle as RO
    from siu3r_amd import raster
    from siu3r_amd.compat.gsplat import rasterization
    from siu3r_amd.gaussian_renderer import SplattingCUDA
    from siu3r_amd.gaussians_types import Gaussians

    H, W, G, V, q, c = 64, 96, 4000, 3, 5, 7
    means, cov, opac, _ = random_scene(G, seed=33, spread=0.15, depth=(0.15, 0.8), scale=(0.001, 0.01))
    ext = torch.stack([look_at_camera(s_, 0.02) for s_ in range(V)])
    Kn = default_K()[None].repeat(V, 1, 1)
    feats = torch.randn(G, q, c)
    # --- the call of gaussian_renderer.py:80-106 (scaled scene: x10 / x100, near = 1)
    e10 = ext.clone()
    e10[:, :3, 3] *= 10.0
    Ks = Kn.clone()
    Ks[:, 0, :] *= W
    Ks[:, 1, :] *= H
    viewmats = torch.linalg.inv(e10)
    colors, alphas, meta = rasterization(means=(means * 10).cuda(), quats=None, scales=None, covars=(cov * 100).cuda(), opacities=opac.cuda(),
                                         colors=feats.reshape(G, q * c).cuda(), viewmats=viewmats.cuda(), Ks=Ks.cuda(), width=W, height=H, sh_degree=None,
                                         near_plane=1.0, far_plane=1000.0)
    assert colors.shape == (V, H, W, q * c) and alphas.shape == (V, H, W, 1) and meta["n_cameras"] == V
    cov6 = raster.cov6_from_cov3x3(cov * 100).numpy()
    for v in range(V):
        cam = raster.make_cam_k3(viewmats[v], Ks[v, 0, 0], Ks[v, 1, 1], Ks[v, 0, 2], Ks[v, 1, 2], W, H, near_plane=1.0, far_plane=1000.0)
        ref = RO.forward(cam, (means * 10).numpy(), cov6, opac.numpy(), feats.reshape(G, q * c).numpy(), want_lists=False)
        assert ref["D"] > 500
        assert float(np.abs(colors[v].cpu().numpy() - ref["image"]).max()) <= 5e-6 * max(1.0, float(np.abs(ref["image"]).max()))
        assert float(np.abs(alphas[v, ..., 0].cpu().numpy() - ref["alpha"]).max()) <= 5e-6
    # --- SplattingCUDA.forward(render_qc_logits=True) goes through the same call: bit-identical
    g = Gaussians(means=means[None].cuda(), covariances=cov[None].cuda(), harmonics=torch.zeros(1, G, 3, 25).cuda(), opacities=opac[None].cuda(),
                  scales=None, rotations=None)
    g.seg_query_class_logits = [feats.cuda()]
    out = SplattingCUDA().forward(g, ext[None], Kn[None], (H, W), render_color=False, render_qc_logits=True)
    qc = out["render_qc_logits"][0]                                                  # [v, q, c, h, w]
    assert torch.equal(qc.permute(0, 3, 4, 1, 2).reshape(V, H, W, q * c), colors)
    # layout / backward-only switches of gsplat are accepted (no effect on the forward result); unknown arguments and algorithms this
    # renderer does not implement are refused by name
    c2, a2, _ = rasterization(means=(means * 10).cuda(), quats=None, scales=None, covars=(cov * 100).cuda(), opacities=opac.cuda(),
                              colors=feats.reshape(G, q * c).cuda(), viewmats=viewmats.cuda(), Ks=Ks.cuda(), width=W, height=H, near_plane=1.0,
                              far_plane=1000.0, packed=False, absgrad=False, sparse_grad=False, rasterize_mode="classic")
    assert torch.equal(c2, colors) and torch.equal(a2, alphas)
    with pytest.raises(TypeError):
        rasterization(means.cuda(), None, None, opac.cuda(), feats.reshape(G, -1).cuda(), viewmats.cuda(), Ks.cuda(), W, H, covars=cov.cuda(), with_ut=True)
    with pytest.raises(NotImplementedError):
        rasterization(means.cuda(), None, None, opac.cuda(), feats.reshape(G, -1).cuda(), viewmats.cuda(), Ks.cuda(), W, H, covars=cov.cuda(), rasterize_mode="antialiased")


def test_gsplat_rasterization_with_the_viewers_argument_lists():
    """viewer.py:301-336 (`rasterize_splats`: quats, exp(scales), sigmoid(opacities), SH colours [G,K,3], sh_degree, radius_clip = 0.1,
    packed=True, absgrad / sparse_grad False, rasterize_mode="classic", backgrounds = ones(3) ON THE DEVICE) and viewer.py:354-369
    (`rasterize_qc_logits`: the same geometry, [G, q*21] feature colours, packed=False), argument for argument, with device-side camera
    tensors as the viewer builds them (:391-392).  Checked against the C oracle chain (quat -> cov6, SH -> rgb, K3 composite, white
    background) and against siu3r_amd.gaussian_renderer.rasterize_splats (the same kernels)."""
    from oracle import raster_oracle as RO
    from siu3r_amd import raster
    from siu3r_amd.compat.gsplat import rasterization
    from siu3r_amd.gaussian_renderer import rasterize_splats

    G, H, W, degree = 6000, 144, 208, 4
    means, cov, opac, sh = random_scene(G, seed=17)
    g = torch.Generator().manual_seed(18)
    quats = torch.randn(G, 4, generator=g) * 2.0
    lscale = torch.log(0.01 + 0.1 * torch.rand(G, 3, generator=g))
    logit = torch.logit(opac.clamp(0.02, 0.98))
    coeffs = sh.permute(0, 2, 1).contiguous()               # [G, 25, 3]
    c2w = look_at_camera(5)
    K = default_K().clone()
    K[0] *= W
    K[1] *= H
    splats = dict(means=means.cuda(), quats=quats.cuda(), scales=lscale.cuda(), opacities=logit.cuda(), sh0=coeffs[:, :1].cuda(), shN=coeffs[:, 1:].cuda())
    camtoworlds, Ks = c2w[None].cuda(), K[None].cuda()
    # --- viewer.py:304-336 verbatim in spirit
    colors = torch.cat([splats["sh0"], splats["shN"]], 1)
    render_colors, render_alphas, info = rasterization(
        means=splats["means"], quats=splats["quats"], scales=torch.exp(splats["scales"]), opacities=torch.sigmoid(splats["opacities"]), colors=colors,
        viewmats=torch.linalg.inv(camtoworlds), Ks=Ks, width=W, height=H, packed=True, absgrad=False, sparse_grad=False, rasterize_mode="classic",
        backgrounds=torch.ones(3, dtype=torch.float32).to(camtoworlds.device), sh_degree=degree, radius_clip=0.1)
    assert render_colors.shape == (1, H, W, 3) and render_alphas.shape == (1, H, W, 1)
    # oracle chain on the pose bits the device computed (the inverse is torch's, on either side)
    w2c = torch.linalg.inv(camtoworlds)[0].cpu()
    campos = torch.linalg.inv(torch.linalg.inv(camtoworlds))[0, :3, 3].cpu()
    cov6_ref = RO.quat_scale_to_cov6(quats.numpy(), torch.exp(lscale.cuda()).cpu().numpy())
    rgb_ref = RO.sh_eval(degree, means.numpy(), campos.numpy(), coeffs.numpy())
    cam = raster.make_cam_k3(w2c, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H, radius_clip=0.1)
    ref = RO.forward(cam, means.numpy(), cov6_ref, torch.sigmoid(logit.cuda()).cpu().numpy(), rgb_ref)
    want = RO.blend_background(ref["image"], ref["alpha"], np.ones(3, np.float32))
    assert ref["D"] > 1000
    assert np.abs(render_alphas[0, ..., 0].cpu().numpy() - ref["alpha"]).max() <= 2e-6
    assert np.abs(render_colors[0].cpu().numpy() - want).max() <= 5e-6
    assert np.array_equal(info["radii"][0].cpu().numpy(), ref["radii"])
    # the repo's own viewer entry point with device-side cameras: the same launches
    c3, a3, _ = rasterize_splats(splats, camtoworlds, Ks, W, H, sh_degree=degree, radius_clip=0.1)
    assert float((c3 - render_colors).abs().max()) <= 1e-6 and torch.equal(a3, render_alphas)  # (camera centre: c2w[:3, 3] here, inverse(viewmat)[:3, 3] there)
    # --- viewer.py:338-373 `rasterize_qc_logits`
    nq, ncls = 3, 21
    qc = torch.randn(G, nq, ncls, generator=g).cuda()
    render_qc, _, _ = rasterization(means=splats["means"], quats=splats["quats"], scales=torch.exp(splats["scales"]), opacities=torch.sigmoid(splats["opacities"]),
                                    colors=qc.flatten(start_dim=1), viewmats=torch.linalg.inv(camtoworlds), Ks=Ks, width=W, height=H, packed=False,
                                    absgrad=False, sparse_grad=False, rasterize_mode="classic")
    render_qc = render_qc.view(-1, H, W, nq, ncls).permute(0, 3, 4, 1, 2)
    cam0 = raster.make_cam_k3(w2c, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H)
    ref_q = RO.forward(cam0, means.numpy(), cov6_ref, torch.sigmoid(logit.cuda()).cpu().numpy(), qc.flatten(start_dim=1).cpu().numpy(), want_lists=False)
    got = render_qc[0].permute(2, 3, 0, 1).reshape(H, W, nq * ncls).cpu().numpy()
    assert float(np.abs(got - ref_q["image"]).max()) <= 5e-6 * max(1.0, float(np.abs(ref_q["image"]).max()))
