"""Split of a step: graph body vs eager panoptic post-process (host sync included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd.model import SIU3RModel
from siu3r_amd import postprocess as pp
from siu3r_amd import synthetic_weights as OW
dev = torch.device("cuda", 0)
m = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision="bf16", device=dev)
img = torch.rand(1, 2, 3, 512, 512).to(dev)
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).to(dev)
for _ in range(4):
    out = m(img, K, enable_query_class_logit_lift=True)
ent = next(iter(m._graphs.values()))
st = ent["st"]
def body():
    m._run_stages(st, lambda name, fn: fn() if name == "tail" else ent["graphs"][name].replay())
for f, name in ((body, "graph body"),):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms")
def post():
    res = m.processor.post_process_panoptic_segmentation(st.seg, threshold=0.5, target_sizes=[(512, 512)], label_ids_to_fuse={0, 1})
    return res
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): post()
torch.cuda.synchronize(); print("panoptic post-process", (time.perf_counter() - t0) / 20 * 1e3, "ms")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m(img, K, enable_query_class_logit_lift=True)
torch.cuda.synchronize(); print("full forward (lift)", (time.perf_counter() - t0) / 20 * 1e3, "ms")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m(img, K)
torch.cuda.synchronize(); print("full forward (no lift)", (time.perf_counter() - t0) / 20 * 1e3, "ms")
