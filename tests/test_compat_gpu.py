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
    # precomputed colours (use_sh=False branch, cuda_splatting.py:112): degree-0 evaluation of the same colours
    rgb = torch.rand(G, 3)
    image_c, *_ = GaussianRasterizer(settings[0])(means3D=means.cuda(), means2D=None, shs=None, colors_precomp=rgb.cuda(), opacities=opac.cuda()[..., None],
                                                  cov3D_precomp=cov6)
    sh0 = ((rgb - 0.5) / 0.28209479177387814)[:, None, :]
    cam0 = make_cam(settings[0]._replace(sh_degree=0))
    ref = RO.forward(cam0, means.numpy(), cov6.cpu().numpy(), opac.numpy(), sh0.numpy(), want_lists=False)
    assert float(np.abs(image_c.cpu().numpy() - ref["image"]).max()) <= 5e-6
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
    with pytest.raises(TypeError):
        rasterization(means.cuda(), None, None, opac.cuda(), feats.reshape(G, -1).cuda(), viewmats, Ks, W, H, covars=cov.cuda(), packed=False)
