"""Replay one stage graph (argv[1]) 5x: run under rocprofv3 --kernel-trace --stats to see what it is made of."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd.model import SIU3RModel
from siu3r_amd import synthetic_weights as OW
dev = torch.device("cuda", 0)
m = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision=(sys.argv[2] if len(sys.argv) > 2 else "bf16"), device=dev)
img = torch.rand(1, 2, 3, 512, 512).to(dev)
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).to(dev)
for _ in range(3):
    m(img, K)
torch.cuda.synchronize()
g = next(iter(m._graphs.values()))["graphs"][sys.argv[1]]
print("MARK")
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
