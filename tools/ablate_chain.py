"""What would the step gain if a chain were FREE?  The captured stage graphs of one forward are replayed on their real streams with
some of them left out: the upper bound of anything that can be won by shortening that chain (the network body only: no panoptic
stage, no host pick-up).    python tools/ablate_chain.py [precision] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd.model import SIU3RModel
from siu3r_amd import synthetic_weights as OW

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
m = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision=prec, device=dev)
img = torch.rand(B, 2, 3, 512, 512).to(dev)
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, 2, 1, 1).to(dev)
for _ in range(4):
    m(img, K)
torch.cuda.synchronize()
ent = next(iter(m._graphs.values()))
sl = ent["slots"][0]
st = sl["st"]
names = list(sl["graphs"])
seg_chain = [n for n in names if n == "spm" or n.startswith("int") or n == "seg"]
heads = [n for n in names if n in ("gs", "pts", "gs0", "gsr", "pts0", "ptsr")]
dec = [n for n in names if n.startswith("dec")]
enc = [n for n in names if n.startswith("enc")]


def step_ms(skip, reps=20):
    def run(name, fn):
        if name in skip:
            return
        fn() if name == "tail" else sl["graphs"][name].replay()
    for _ in range(3):
        m._run_stages(st, run, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        m._run_stages(st, run, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"{prec} B={B}: network body replayed from its stage graphs (no panoptic stage / host pick-up), ms per step")
full = step_ms(set())
print(f"  everything                                   {full:7.2f}")
for label, skip in (("without ViT-Adapter + Mask2Former (spm, int*, seg)", seg_chain), ("without Mask2Former only (seg)", ["seg"]),
                    ("without the DPT heads", heads), ("without the Gaussian heads only", [n for n in heads if n.startswith("gs")]),
                    ("without the pts3d heads only", [n for n in heads if n.startswith("pts")]),
                    ("without decoder + heads", dec + heads), ("encoder only (+ enc_begin)", seg_chain + dec + heads),
                    ("encoder + adapter + Mask2Former (no decoder, no heads)", dec + heads)):
    t = step_ms(set(skip) | ({"tail"} if any(n in skip for n in heads) else set()))
    print(f"  {label:52s} {t:7.2f}   ({full - t:+.2f})")
