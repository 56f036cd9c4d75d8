"""Where the Gaussian stem's 220 us go: the fused launch, the same convolution without the upsample-add, a dense product of the same
shape (M = 512^2, N = 256, K = 196 / 224), and the convolution writing planes.  python tools/mb_stem2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import ops
from mb_gemm import graph_time
B = 1
img = ops.pack_image_nhwc(torch.rand(B, 3, 512, 512, device="cuda"), torch.float32, 4)
low = torch.rand(B, 256, 256, 256, device="cuda")
pw = ops.pack_conv(torch.rand(256, 3, 7, 7, device="cuda") * 0.1, torch.zeros(256, device="cuda"), True, cin_pad=4)
out = torch.empty(B, 512, 512, 256, device="cuda")
t = graph_time(lambda: ops.conv2d(img, pw, stride=1, pad=3, act=ops.ACT_RELU, out=out, up_src=low), n=5)
print(f"conv7x7 + ReLU + upsample-add: {t*1e6:8.1f} us")
t = graph_time(lambda: ops.conv2d(img, pw, stride=1, pad=3, act=ops.ACT_RELU, out=out), n=5)
print(f"conv7x7 + ReLU               : {t*1e6:8.1f} us")
for K in (200, 224, 256):
    x = torch.rand(1, 512 * 512, K, device="cuda")
    w = ops.pack_linear(torch.rand(256, K, device="cuda") * 0.1, torch.zeros(256, device="cuda"), True)
    o2 = out.view(1, 512 * 512, 256)
    t = graph_time(lambda: ops.linear(x, w, out=o2, act=ops.ACT_RELU), n=5)
    print(f"dense 262144 x 256 x {K} + ReLU: {t*1e6:8.1f} us")
t = graph_time(lambda: ops.resize_bilinear(low, (512, 512), True), n=5)
print(f"x2 upsample alone (resize kernel, writes 268 MB): {t*1e6:8.1f} us")
for Bn in (1, 8):
    imgs = ops.pack_image_nhwc(torch.rand(Bn * 2, 3, 512, 512, device="cuda"), torch.float32, 4).view(Bn, 2, 512, 512, 4)
    lows = torch.rand(Bn, 2, 256, 256, 256, device="cuda")
    wfrag, bias = ops.pack_stem7([torch.rand(256, 3, 7, 7, device="cuda") * 0.1 for _ in range(2)], [torch.rand(256, device="cuda") for _ in range(2)])
    o = torch.empty(Bn, 2, 512, 512, 256, device="cuda")
    for planes in (False, True):
        t = graph_time(lambda: ops.stem7x7_x3(imgs, wfrag, bias, lows, o, planes=planes), n=5)
        print(f"dedicated stem kernel, B = {Bn}, BOTH heads in one launch, planes {planes}: {t*1e6:8.1f} us ({t*1e6/2/Bn:.1f} per view)   {o.numel() * 4 / t / 1e12:.2f} TB/s of output")
    t = graph_time(lambda: ops.stem7x7_x3(imgs, wfrag, bias, None, o), n=5)
    print(f"dedicated stem kernel, B = {Bn}, no upsample source: {t*1e6:8.1f} us")
