"""old-tree probe 9: seven variants of the argmax kernel launched right behind the real one on the same buffers (same stream), every
forward; each label map against a host recomputation (item 0) and against the sc0-sc1 variant (all items)."""
import sys, os, ctypes, numpy as np, torch
import torch.nn.functional as F
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
NV = 7
labs = torch.zeros(NV, B, 2, S, S, dtype=torch.int32, device="cuda"); scr = torch.zeros(8192, dtype=torch.int32, device="cuda")
m = SIU3RModel(OW.make_weights(0), image_size=(S, S), precision="bf16x3")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
def rd(ptr, shape, dtype):
    a = np.empty(shape, dtype=dtype)
    assert hip.hipMemcpy(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), a.nbytes, 2) == 0
    return a
info = {}
orig_bp = m.processor.begin_panoptic
def wrapped(*a, **k):
    p = orig_bp(*a, **k)
    class_logits, mcl, scores, labels, lab_map, area, orig = p["keep"]
    info.update(dims=p["dims"], p256=p["p256"].data_ptr(), scores=scores.data_ptr(), kept=p["kept_idx"].data_ptr(), lab=lab_map.data_ptr(), tab=p["tab"].data_ptr())
    return p
m.processor.begin_panoptic = wrapped
L = _lib.lib()
L.siu3r_pp_dbg_set_variants.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
L.siu3r_pp_dbg_set_variants(labs.data_ptr(), labs[0].numel(), scr.data_ptr())
ASM = os.environ.get("SIU3R_PP_ASM") is not None
names = ["0 real (plain, compiler's code)", "10 asm replica: 4 gathers + uniform load, vmcnt(2) / vmcnt(1), packed multiplies", "11 = 10 + 16 idle cycles behind every wait", "12 full wait vmcnt(0), packed multiplies",
         "13 = 10 with scalar multiplies", "14 uniform load first, then gathers, vmcnt(1) / vmcnt(0), packed", "15 = 10 with the uniform load sc0 sc1", "16 = 10 with the first product written into the address registers of the 4th gather (in flight)"] if ASM else ["0 real (plain)", "1 plain again", "2 sc0 sc1 loads", "3 plain + vmcnt(0) + 16 nops before the arithmetic", "4 plain, no packed arithmetic", "8 plain, loop not unrolled (#pragma unroll 1)",
         "6 plain, item 0 only (grid.y = 1)", "9 plain, unrolled, without the LDS atomic on s_orig in the loop"]
wrong0 = np.zeros(NV + 1, dtype=np.int64); wrong_fw = np.zeros(NV + 1, dtype=np.int64); wrong_other = np.zeros(NV + 1, dtype=np.int64)
am = None
with torch.no_grad():
    for it in range(N):
        o = m(img, K, enable_query_class_logit_lift=True)
        torch.cuda.synchronize()
        Bq, T, Q, Cc, IH, IW, H, W = info["dims"]
        lab_real = torch.from_numpy(rd(info["lab"], (B, 2, S, S), np.int32)).cuda()
        if am is None:  # the inputs are the same every forward: one host recomputation of item 0
            tab = rd(info["tab"], (5 * Bq * Q + 2 * Bq,), np.int32); nk = tab[5 * Bq * Q:5 * Bq * Q + Bq]
            scores = rd(info["scores"], (Bq, Q), np.float32); kept = rd(info["kept"], (Bq, Q), np.int32); n0 = int(nk[0])
            vol = torch.from_numpy(rd(info["p256"], (T, 256, 256, Q), np.float32))[..., torch.from_numpy(kept[0, :n0].astype(np.int64))]
            up = F.interpolate(vol.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False)
            am = (up * torch.from_numpy(scores[0, kept[0, :n0]]).view(1, n0, 1, 1)).argmax(1).int().cuda()
        allv = [lab_real] + [labs[v] for v in range(NV)]
        line = []
        for v, lv in enumerate(allv):
            w0 = int((lv[0] != am).sum())
            wo = 0 if (v == 6 and not ASM) else int((lv[1:] != labs[2 if ASM else 1][1:]).sum())
            wrong0[v] += w0; wrong_fw[v] += w0 > 0; wrong_other[v] += wo
            line.append(f"{w0}" + (f"+{wo}" if wo else ""))
        print(f"iter {it}: wrong label px of item 0 (+ other items vs the sc0-sc1 variant) per variant: " + " | ".join(line), flush=True)
print("\nvariant: forwards with a wrong item-0 map / wrong px in total / other items' px differing from variant 2")
for v in range(NV + 1):
    print(f"  {names[v]:58s} {wrong_fw[v]:3d} / {wrong0[v]:6d} / {wrong_other[v]}")
print("DONE", it + 1)
