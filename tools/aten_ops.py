"""Which torch (aten) operators still launch device work inside one forward, and from which source line: python tools/aten_ops.py [precision]
(the product kernels go through ctypes and do not appear here)."""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from siu3r_amd import synthetic_weights as OW
from siu3r_amd.model import SIU3RModel

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
os.environ["SIU3R_NO_GRAPH"] = "1"
dev = torch.device("cuda", 0)
model = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision=prec, device=dev)
images = torch.rand(1, 2, 3, 512, 512).to(dev)
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).to(dev)
with torch.no_grad():
    for _ in range(3):
        model(images, K, enable_query_class_logit_lift=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    with torch.no_grad():
        model(images, K, enable_query_class_logit_lift=True)
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEWS = {"view", "reshape", "as_strided", "select", "slice", "permute", "expand", "transpose", "unsqueeze", "squeeze", "t", "empty", "empty_like",
         "empty_strided", "detach", "alias", "narrow", "unbind", "split", "_unsafe_view", "flatten", "unflatten", "chunk", "lift_fresh", "resolve_conj",
         "resolve_neg", "_local_scalar_dense", "item", "is_nonzero", "result_type", "stride", "size", "numel", "_reshape_alias", "view_as", "expand_as"}
cnt, dur = Counter(), Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_children or ev.name[6:] in VIEWS:
        continue
    where = next((s for s in ev.stack if root in s and "tools/" not in s), ev.stack[0] if ev.stack else "?")
    key = (ev.name, where.replace(root + "/", "")[:110])
    cnt[key] += 1
    dur[key] += ev.device_time_total
for key, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{n:4d} x {dur[key]:8.1f} us  {key[0]:22s} {key[1]}")
print(f"total: {sum(cnt.values())} launching aten ops, {sum(dur.values()):.0f} us of device time")
