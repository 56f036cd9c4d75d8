"""old-tree probe 4: device timestamps (100 MHz) -- does pp_argmax start before pp_mask256 has ended?"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from golden_utils import fixture_images, default_K
from siu3r_amd import synthetic_weights as OW, _lib
from siu3r_amd.model import SIU3RModel
B, S = int(os.environ.get("DBG_B", "8")), 512
N = int(os.environ.get("DBG_N", "40"))
g = torch.Generator().manual_seed(11)
fx_ = fixture_images(S)
img = torch.cat([fx_, torch.rand(B - 2, 2, 3, S, S, generator=g), fx_.flip(1)]).cuda()
K = default_K().repeat(B, 1, 1, 1).cuda()
m = SIU3RModel(OW.make_weights(0), image_size=(S, S), precision="bf16x3")
L = _lib.lib()
recs = np.zeros(8192 * 8, dtype=np.uint32); ts = np.zeros(8, dtype=np.uint64); n = ctypes.c_uint(0)
def fetch():
    rc = L.siu3r_pp_dbg_fetch(recs.ctypes.data_as(ctypes.c_void_p), ts.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    assert rc == 0, rc
    return ts.astype(np.int64).copy()
fetch()
ref = None; nflaky = 0
with torch.no_grad():
    for it in range(N):
        o = m(img, K, enable_query_class_logit_lift=True)
        torch.cuda.synchronize()
        t = fetch()
        seg0 = o[2][0].clone()
        if ref is None:
            ref = seg0
        nd = int((seg0 != ref).sum())
        nflaky += nd > 0
        # times in us relative to the first mask256 block
        u = lambda x: (x - t[3]) / 100.0
        print(f"iter {it}: differing px {nd:4d} | mask256: head blocks end {u(t[2]):8.1f} us, all end {u(t[1]):8.1f} us | argmax first-blocks start {u(t[0]):8.1f} us  (gap to mask256 end {u(t[0]) - max(u(t[1]), u(t[2])):7.1f} us); NaN weighted values {t[5]} pixel range {t[6] if t[5] else 0}..{t[7]}", flush=True)
print("DONE", it + 1, "flaky", nflaky)
