"""N-channel (gsplat-semantics) list composite on the pixel-aligned pair scene, 6 views @512^2 in one call: ms per frame of the whole call and of the
composite kernel alone (graph-free: HIP events around the composite entry point on prepared lists), per channel count and kernel form
(form 1: 32-channel chunks, raster.tune(0, 1); form 5 = default: matrix-core rank-2 updates over per-quadrant lists).  python tools/mb_feat.py [channels ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd import _lib, raster, synthetic
from siu3r_amd.ops import _p, _stream

H = W = 512
nv = 6
chans = [int(a) for a in sys.argv[1:]] or [168, 84, 21]
dev = torch.device("cuda", 0)
means, cov, opac, sh = (t.to(dev) for t in synthetic.pixel_aligned_scene(H, W, 2, seed=0))
ext = synthetic.target_views(nv)
ext[:, :3, 3] *= 10.0
Kt = synthetic.default_intrinsics()
m10, c100 = (means * 10.0).contiguous(), (cov * 100.0).contiguous()
cams = []
for j in range(nv):
    Kp = Kt.clone(); Kp[0, :] *= W; Kp[1, :] *= H
    cams.append(raster.make_cam_k3(torch.linalg.inv(ext[j]), Kp[0, 0], Kp[1, 1], Kp[0, 2], Kp[1, 2], W, H, near_plane=1.0, far_plane=1000.0))


def ev_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for C in chans:
    feats = torch.randn(means.shape[0], C, generator=torch.Generator().manual_seed(5)).to(dev)
    outs = {}
    for form in ("1", "5"):
        raster.tune(0, 1 if form == "1" else 0)
        run = lambda: raster.rasterize_views_k3(cams, m10, c100, opac, feats)
        o = run()
        st = o["state"]
        ms_call = ev_ms(run) / nv
        out = torch.empty_like(o["colors"]); al = torch.empty_like(o["alphas"])
        ws = st["feat_ws"]
        comp = lambda: _lib.check(_lib.lib().siu3r_raster_composite_feat_ws(st["cams"], nv, _p(st["cams_dev"]), means.shape[0], _p(st["tile_start_all"]), _p(st["ids_all"]),
                                                                            st["cap_d"], _p(st["rec"]), _p(feats), C, _p(out), _p(al), _p(ws), 0 if ws is None else ws.numel() * 4, _stream()))
        ms_comp = ev_ms(comp, 10) / nv
        Gv, Dp = st.totals(0), st.totals(1)
        b = sum(raster.algorithmic_bytes(means.shape[0], gv, d, H * W, channels=C) for gv, d in zip(Gv, Dp)) / nv
        outs[form] = o["colors"]
        print(f"C={C:4d} form {form}: call {ms_call:.3f} ms/frame ({b / ms_call / 1e6:.0f} GB/s alg. = {b / ms_call / 1e6 / 8000:.3f} of 8 TB/s), composite kernel {ms_comp:.3f} ms/frame; "
              f"D/view {sum(Dp) / nv:.0f}", flush=True)
    raster.tune(0, 0)
    print(f"        forms bit-identical: {torch.equal(outs['1'], outs['5'])}")
