"""Stage timeline of one graph-replayed forward as it really overlaps on the streams: start / end of every stage graph relative to the
step's first event.  python tools/timeline.py [bf16|bf16x3] [B]"""
import os, sys, time
os.environ.setdefault("SIU3R_DEC_PER_LAYER", "1")  # one graph per decoder layer, so that the layers show up separately
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd.model import SIU3RModel
from siu3r_amd import synthetic_weights as OW
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
m = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision=prec, device=dev)
img = torch.rand(B, 2, 3, 512, 512).to(dev)
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, 2, 1, 1).to(dev)
for _ in range(4):
    m(img, K, enable_query_class_logit_lift=True)
torch.cuda.synchronize()
m._timeline = []
t0 = time.perf_counter()
ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
out = m(img, K, enable_query_class_logit_lift=True)
ev1 = torch.cuda.Event(enable_timing=True); ev1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
tl, m._timeline = m._timeline, None
print(f"{prec} B={B}: forward wall {wall:.2f} ms, GPU span {ev0.elapsed_time(ev1):.2f} ms; segments {[len(i) for i in out[3]]}")
rows = [(n, ev0.elapsed_time(a), ev0.elapsed_time(b)) for n, a, b in tl]
for n, a, b in rows:
    print(f"  {n:10s} {a:7.2f} -> {b:7.2f}  ({b - a:5.2f})")
print(f"  network body ends {max(b for _, _, b in rows):.2f} ms; post-process + outputs until {ev0.elapsed_time(ev1):.2f} ms")
