"""Isolated duration of every stage graph of the bf16 forward (replayed alone, 20x): where the critical path is."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd.model import SIU3RModel
from siu3r_amd import synthetic_weights as OW
dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision=prec, device=dev)
img = torch.rand(1, 2, 3, 512, 512).to(dev)
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).to(dev)
for _ in range(4):
    m(img, K)
ent = next(iter(m._graphs.values()))
tot = {}
for name, g in ent["graphs"].items():  # the tail stage is eager (not in this dict)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    tot[name] = e0.elapsed_time(e1) / 20
grp = lambda pre: sum(v for k, v in tot.items() if k.startswith(pre))
print({k: round(v, 3) for k, v in tot.items() if not (k.startswith("dec") and (k[3] in "AB" or k[3].isdigit()))})
print("encoder", round(grp("enc"), 2), "| spm+int", round(tot["spm"] + grp("int"), 2), "| seg", round(tot["seg"], 2), "| dec pre/post", round(tot["dec_pre"] + tot["dec_post"], 2),
      "| decA", round(grp("decA"), 2), "| decB", round(grp("decB"), 2), "| dec (merged sides)", round(sum(v for k, v in tot.items() if k.startswith("dec") and k[3].isdigit()), 2), "| heads", {k: round(tot[k], 2) for k in ("gs0", "gsr", "pts0", "ptsr", "gs", "pts") if k in tot}, "| tail: eager")
print("sum", round(sum(tot.values()), 2))
