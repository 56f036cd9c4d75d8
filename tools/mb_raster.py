"""Rasterizer micro-benchmarks.
  python tools/mb_raster.py stress [G] [W] [H]   BASELINE.json configs[4] shape: ~2M random Gaussians, one 1080p frame, SH degree 4 -> RGB + depth
  python tools/mb_raster.py pair [views]         a pixel-aligned 2 x 512^2 Gaussian set (what the network emits) -> `views` 512^2 target views,
                                                 one rasterizer call (K2 semantics, 3x3 covariances and planar SH read in place)
"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd import cuda_splatting as cs, raster, synthetic

mode = sys.argv[1] if len(sys.argv) > 1 else "stress"
nt = os.environ.get("MB_NT", "1") == "1"


def timed(fn, n=10):
    for _ in range(2):
        o = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        o = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, o


if mode == "stress":
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 2_097_152
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
    H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
    means, cov, opac, sh = (t.cuda() for t in synthetic.random_scene(G, seed=1, spread=3.0, depth=(2.0, 9.0), scale=(0.004, 0.03)))
    c2w = synthetic.perturbed_camera(0, jitter=0.1)
    w2c = torch.linalg.inv(c2w)
    fx = fy = 0.9 * W
    fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
    proj = cs.get_projection_matrix(torch.tensor([0.1]), torch.tensor([100.0]), torch.tensor([fovx]), torch.tensor([fovy]))[0]
    cam = raster.make_cam_k2(w2c=w2c, full_proj=proj @ w2c, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2), campos=c2w[:3, 3], bg=torch.zeros(3),
                             width=W, height=H, sh_degree=4)
    ms, o = timed(lambda: raster.rasterize_views_k2([cam], means, cov, sh, opac, want_n_touched=nt, check_overflow=False, sh_planar=True))
    st = o["state"]
    st.verify()
    Gv, D = st["Gv"], st["D"]
    b = raster.algorithmic_bytes(G, Gv, D, H * W)
    print("entries E =", st["E"], "cap", st["cap_e"], "geometry", st["geo"])
    print(f"stress G={G} visible={Gv} pairs={D} px={H*W}: {ms:.3f} ms/frame, algorithmic {b/1e6:.1f} MB -> {b/ms/1e6:.1f} GB/s, alpha mean {o['opacity'].mean().item():.3f}")
else:
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    H = W = 512
    means, cov, opac, sh = (t.cuda() for t in synthetic.pixel_aligned_scene(H, W, 2, seed=0))
    G = means.shape[0]
    ext = synthetic.target_views(V)
    K = synthetic.default_intrinsics()[None].repeat(V, 1, 1)
    means10, cov100 = means * 10.0, cov * 100.0  # SplattingCUDA's scene rescale (gaussian_renderer.py:43-46)
    ext10 = ext.clone()
    ext10[:, :3, 3] *= 10.0
    args = (ext10, K, torch.full((V,), 1.0), torch.full((V,), 1000.0), (H, W), torch.zeros(V, 3), means10[None].expand(V, -1, -1),
            cov100[None].expand(V, -1, -1, -1), sh[None].expand(V, -1, -1, -1), opac[None].expand(V, -1))
    ms, o = timed(lambda: cs.render_cuda(*args, return_aux=True))
    st = o[2][0]["state"]
    Gv, D, E = st.totals(0), st.totals(1), st.totals(2)
    b = sum(raster.algorithmic_bytes(G, gv, d, H * W) for gv, d in zip(Gv, D))
    print(f"pair G={G} views={V}: visible {Gv} pairs {D} entries {E}")
    print(f"pair: {ms:.3f} ms per call = {ms/V:.3f} ms/frame, algorithmic {b/1e6:.1f} MB -> {b/ms/1e6:.1f} GB/s, alpha mean {o[2][0]['opacity'].mean().item():.3f}")
