"""Host vs device time of pipelined steps (SIU3RModel.forward_async): how long the host needs to enqueue a step, how long result()
blocks, and when each step's body starts / ends on the GPU.   python tools/pipeline_probe.py [depth] [precision] [steps]"""
import collections
import sys
import time

import torch

sys.path.insert(0, ".")
from siu3r_amd import synthetic_weights as OW
from siu3r_amd.model import SIU3RModel

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
H = W = 512
m = SIU3RModel(OW.make_weights(0), image_size=(H, W), precision=prec)
m.pipeline_depth = depth
img = torch.rand(1, 2, 3, H, W, generator=torch.Generator().manual_seed(1234)).cuda()
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).cuda()


def run(n, log=None):
    pend = collections.deque()
    with torch.no_grad():
        for i in range(n):
            t0 = time.perf_counter()
            pend.append(m.forward_async(img, K, enable_query_class_logit_lift=True))
            t1 = time.perf_counter()
            if len(pend) >= depth:
                pend.popleft().result()
            t2 = time.perf_counter()
            if log is not None:
                log.append((t0, t1, t2))
        while pend:
            pend.popleft().result()


run(4 + 2 * depth)
torch.cuda.synchronize()
log = []
t0 = time.perf_counter()
run(steps, log)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{prec} depth {depth}: {steps / dt:.2f} pairs/s, {dt / steps * 1e3:.2f} ms/step")
for i, (a, b, c) in enumerate(log):
    print(f"  step {i:2d}: submit at {1e3 * (a - t0):7.2f} ms, enqueue took {1e3 * (b - a):6.2f} ms, result() of the oldest took {1e3 * (c - b):6.2f} ms")

# stage timeline of a few pipelined steps (events on each stage's own stream)
m._timeline = []
ev0 = torch.cuda.Event(enable_timing=True)
ev0.record()
run(4)
torch.cuda.synchronize()
tl, m._timeline = m._timeline, None
step = -1
for n, a, b in tl:
    if n == "enc_begin":
        step += 1
    if n in ("enc_begin", "enc0", "enc3", "int3", "seg", "dec_pre", "dec_all", "dec_post", "gs0", "gsr", "ptsr", "pts0", "tail"):
        print(f"  step {step} {n:10s} {ev0.elapsed_time(a):7.2f} -> {ev0.elapsed_time(b):7.2f}  ({a.elapsed_time(b):5.2f})")
