"""Round-4 probe of the B = 8 panoptic label flake: 40 forwards of one batch (DBG_B pairs, default 8),
the segmentation map of item 0 of every forward against the first one; prints the forwards that differ and where.  python tools/label_flake_probe.py"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from golden_utils import fixture_images, default_K
from siu3r_amd import synthetic_weights as OW
from siu3r_amd.model import SIU3RModel
import os
B, S = int(os.environ.get("DBG_B", "8")), 512
g = torch.Generator().manual_seed(11)
fx_ = fixture_images(S)
img = torch.cat([fx_, torch.rand(B - 2, 2, 3, S, S, generator=g), fx_.flip(1)]).cuda()
K = default_K().repeat(B, 1, 1, 1).cuda()
m = SIU3RModel(OW.make_weights(0), image_size=(S, S), precision="bf16x3")
ref = None
print(f"# B={B}", flush=True)
with torch.no_grad():
    for it in range(int(os.environ.get("DBG_N", "40"))):
        o = m(img, K, enable_query_class_logit_lift=True)
        torch.cuda.synchronize()
        seg0 = o[2][0].clone()
        qcl0 = o[0].seg_query_class_logits[0].clone()
        if ref is None:
            ref, qref = seg0, qcl0
            continue
        if qcl0.shape == qref.shape:
            dq = (qcl0 != qref)
            if dq.any():
                rows = dq.any(-1).any(-1).nonzero().flatten()
                print(f'iter {it}: qcl differs in {int(dq.sum())} values, {len(rows)} pixels, first pixel lin {int(rows[0])} last {int(rows[-1])}, max abs {float((qcl0 - qref).abs().max()):.3e}')
        d = (seg0 != ref).nonzero()
        if len(d):
            v, y, x = d[:, 0], d[:, 1], d[:, 2]
            lin = (v * S * S + y * S + x)
            runs = (lin[1:] != lin[:-1] + 1).sum().item() + 1
            print(f"iter {it}: {len(d)} px differ in {runs} runs; views {sorted(set(v.tolist()))} y range {int(y.min())}-{int(y.max())} x range {int(x.min())}-{int(x.max())}; "
                  f"first lin {int(lin[0])} (mod 256 = {int(lin[0]) % 256}); values ref {ref[v[0], y[0], x[0]].item()} got {seg0[v[0], y[0], x[0]].item()}; run lengths {[int(t) for t in torch.diff(torch.cat([torch.tensor([-1], device=lin.device), (lin[1:] != lin[:-1] + 1).nonzero().flatten(), torch.tensor([len(lin) - 1], device=lin.device)]))][:12]}")
