"""The Gaussian heads' fused stem -- ReLU(conv7x7(image)) + x2 upsample of the 256-channel map (dpt_gs_head.py:160-162) -- at 512^2, per tile
family: python tools/mb_stem.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
img = ops.pack_image_nhwc(torch.rand(B, 3, 512, 512, device="cuda"), torch.float32, 4)
low = torch.rand(B, 256, 256, 256, device="cuda")
pw = ops.pack_conv(torch.rand(256, 3, 7, 7, device="cuda") * 0.1, torch.zeros(256, device="cuda"), True, cin_pad=4)
for cfg, name in ((0, "auto"), (-1, "128x64"), (1, "pp256x256"), (2, "pp256x128"), (3, "pp128x128")):
    ops.gemm_tune(0, cfg)
    t = graph_time(lambda: ops.conv2d(img, pw, stride=1, pad=3, act=ops.ACT_RELU, out_dtype=torch.float32, up_src=low), n=5)
    print(f"stem B={B} {name:>10}: {t*1e6:8.1f} us  ({B * 262144 * 256 * 4 / t / 1e12:.2f} TB/s of output)")
ops.gemm_tune(0, 0)
