"""Rasterizer stress (BASELINE.json configs[4] shape): ~2M Gaussians, 1080p, SH degree 4 -> RGB + depth.
python tools/mb_raster.py [G] [W] [H]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from siu3r_amd import raster
from scenes import random_scene, look_at_camera
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2_097_152
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
means, cov, opac, sh = random_scene(G, seed=1, spread=3.0, depth=(2.0, 9.0), scale=(0.004, 0.03))
means, cov, opac, sh = (t.cuda() for t in (means, cov, opac, sh))
cov6 = raster.cov6_from_cov3x3(cov)
shs = sh.permute(0, 2, 1).contiguous()  # [G, 25, 3]
c2w = look_at_camera(0, jitter=0.1)
w2c = torch.linalg.inv(c2w)
fx = fy = 0.9 * W
cam = raster.make_cam_k3(w2c, fx, fy, W / 2, H / 2, W, H)
from siu3r_amd import cuda_splatting as cs
import math
fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
proj = cs.get_projection_matrix(torch.tensor([0.1]), torch.tensor([100.0]), torch.tensor([fovx]), torch.tensor([fovy]))[0]
full = proj @ w2c
cam2 = raster.make_cam_k2(w2c=w2c, full_proj=full, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2), campos=c2w[:3, 3], bg=torch.zeros(3), width=W, height=H, sh_degree=4)
for _ in range(2):
    o = raster.rasterize_k2(cam2, means, cov6, shs, opac)
torch.cuda.synchronize()
st = o["state"]
Gv = int((st["tiles_touched"] > 0).sum()); D = st["D"]
n = 10
t0 = time.perf_counter()
for _ in range(n):
    o = raster.rasterize_k2(cam2, means, cov6, shs, opac, check_overflow=False)  # capacity check deferred: one verify() below
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
o["state"].verify()
print("entries E =", o["state"]["E"], "cap", o["state"]["cap_e"], "geometry", o["state"]["geo"])
b = raster.algorithmic_bytes(G, Gv, D, H * W)
print(f"G={G} visible={Gv} pairs={D} px={H*W}: {ms:.3f} ms/frame, algorithmic {b/1e6:.1f} MB -> {b/ms/1e6:.1f} GB/s, alpha mean {o['opacity'].mean().item():.3f}")
