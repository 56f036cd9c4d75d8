"""The round-4 "stale read" of the panoptic stage, reproduced where it lives: in the argmax kernel's own arithmetic.

Every forward of a B = 8 batch (the model's chains on their streams), right behind the panoptic device stage and on its stream, the
round-3/4 argmax kernel is launched again on the stage's buffers as stand-alone code objects: `e0` = the compiler's code of that kernel
(hipcc -O3, gfx950), `eN` = the SAME assembly with one edit (tools/probes/pk_hazard/gen.py).  Every label map of item 0 is compared with a
host recomputation from the buffers (which are bit-identical every forward).  Findings (profiles/r06_pk_hazard.txt):
the unedited code is wrong in 35-50 % of the forwards -- always in the FIRST workgroups of a launch, any launch, however long after the
volume was written; so is the code with full waits, with idle cycles behind the waits or between the dependent packed operations, with
scalar products or reordered loads.  It is never wrong when `v_pk_add_f32 v[18:19], v[18:19], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]` -- an
in-place packed add on the two register pairs that were the address operands of this iteration's first two gathers -- gets a fresh
destination (e8), other operand registers (e4, e9, e11), or is replaced by two scalar adds (e5, e10).  The loads are innocent: nothing
is stale.  Follow-ups without the network: tools/probes/pk_hazard/standalone.py (beside the library's bf16x3 128 x 64 GEMM every variant
that keeps the crossed packed add fails; register placement only moves the rate) and tools/probes/pk_hazard/xwave2.hip (the instruction
alone: a packed fp32 operation with op_sel:[x,1] returns wrong low halves beside another wave's bf16 MFMAs -- the root cause).
    python tools/pk_hazard_probe.py            (DBG_B pairs per step, DBG_N forwards, DBG_VLIST = variants, 8 of 0..11)"""
import sys, os, ctypes, numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_utils import fixture_images, default_K
from siu3r_amd import synthetic_weights as OW, _lib
from siu3r_amd.model import SIU3RModel
B, S = int(os.environ.get("DBG_B", "8")), 512
N = int(os.environ.get("DBG_N", "40"))
g = torch.Generator().manual_seed(11)
fx_ = fixture_images(S)
img = torch.cat([fx_, torch.rand(B - 2, 2, 3, S, S, generator=g), fx_.flip(1)]).cuda()
K = default_K().repeat(B, 1, 1, 1).cuda()
NV = 8
labs = torch.zeros(NV, B, 2, S, S, dtype=torch.int32, device="cuda"); scr = torch.zeros(8192, dtype=torch.int32, device="cuda")
m = SIU3RModel(OW.make_weights(0), image_size=(S, S), precision="bf16x3")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
def rd(ptr, shape, dtype):
    a = np.empty(shape, dtype=dtype)
    assert hip.hipMemcpy(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), a.nbytes, 2) == 0
    return a
BUILD = os.path.join(ROOT, "tools", "probes", "pk_hazard", "_build")
if not os.path.exists(os.path.join(BUILD, "ppa_e11.hsaco")):
    import subprocess
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "probes", "pk_hazard", "gen.py")])
funcs = []
VLIST = [int(x) for x in os.environ.get("DBG_VLIST", "0,1,2,3,4,5,6,7").split(",")]
for v in VLIST:
    mod = ctypes.c_void_p(); fn = ctypes.c_void_p()
    assert hip.hipModuleLoad(ctypes.byref(mod), os.path.join(BUILD, f"ppa_e{v}.hsaco").encode()) == 0
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, f"ppa_e{v}".encode()) == 0
    funcs.append(fn)
hip.hipModuleLaunchKernel.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 6 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
def launch(fn, stream, p256, scores, kept, nkeep, lab, T, H, W, MS, Q, thr, gx, gy):
    vals = [ctypes.c_void_p(p256), ctypes.c_void_p(scores), ctypes.c_void_p(kept), ctypes.c_void_p(nkeep), ctypes.c_void_p(lab), ctypes.c_void_p(scr.data_ptr()), ctypes.c_void_p(scr.data_ptr() + 16384),
            ctypes.c_int(T), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(MS), ctypes.c_int(Q), ctypes.c_float(thr)]
    arr = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.byref(v), ctypes.c_void_p) for v in vals])
    rc = hip.hipModuleLaunchKernel(fn, gx, gy, 1, 256, 1, 1, 0, ctypes.c_void_p(stream), arr, None)
    assert rc == 0, rc

def _old_layout_volume(pend, B, T, Q, MS=256):
    """the stage's probability planes [B, T, Q, MS, MS] (kept queries only, query-major: round 6) as the channel-last volume
    [B, T, MS, MS, Q] of rounds 2-5, which the round-3/4 code objects index"""
    import torch
    planes, kept, tab = pend["p256"], pend["kept_idx"], pend["tab"]
    nk = tab[5 * B * Q:5 * B * Q + B]
    vol = torch.zeros((B, T, MS, MS, Q), dtype=torch.float32, device=planes.device)
    for b in range(B):
        n = int(nk[b])
        vol[b][..., kept[b, :n].long()] = planes[b, :, :n].permute(0, 2, 3, 1)
    return vol

info = {}
orig_bp = m.processor.begin_panoptic
def wrapped(*a, **k):
    p = orig_bp(*a, **k)
    class_logits, mcl, scores, labels, lab_map, area, orig = p["keep"]
    Bq, T, Q, Cc, IH, IW, H, W = p["dims"]
    nkeep_ptr = p["tab"].data_ptr() + 4 * 5 * Bq * Q
    st = torch.cuda.current_stream().cuda_stream
    vol = _old_layout_volume(p, Bq, T, Q)  # (round 6: the stage keeps query-major planes of the kept queries; the code objects index the old volume)
    info["vol_keepalive"] = vol
    for v in range(NV):
        launch(funcs[v], st, vol.data_ptr(), scores.data_ptr(), p["kept_idx"].data_ptr(), nkeep_ptr, labs[v].data_ptr(), T, H, W, 256, Q, 0.5, (T * H * W + 255) // 256, Bq)
    info.update(dims=p["dims"], p256=vol.data_ptr(), scores=scores.data_ptr(), kept=p["kept_idx"].data_ptr(), lab=lab_map.data_ptr(), tab=p["tab"].data_ptr())
    return p
m.processor.begin_panoptic = wrapped
ALLN = {8: "e8 packed sum of v[18:21] into a FRESH destination (not in place)", 9: "e9 products in fresh registers, packed sum IN PLACE on them", 10: "e10 only the packed add replaced by two scalar adds", 11: "e11 gathers addressed from v[30:37]: v[18:21] never address operands, products still land there"}
names = ["real (library, plain)", "e0 compiler's code, unedited", "e1 first wait vmcnt(1) -> vmcnt(0)", "e2 s_nop 0 -> s_nop 7 between the dependent packed operations", "e3 s_nop 7 behind both waits",
         "e4 products into fresh registers instead of the finished loads' address registers", "e5 packed products, scalar tail", "e6 scalar products, packed tail", "e7 loads reordered so that a pair is filled by consecutive loads"]
names = [names[0]] + [ALLN[v] if v in ALLN else names[v + 1] for v in VLIST]
wrong0 = np.zeros(NV + 1, dtype=np.int64); wrong_fw = np.zeros(NV + 1, dtype=np.int64)
am = None
with torch.no_grad():
    for it in range(N):
        o = m(img, K, enable_query_class_logit_lift=True)
        torch.cuda.synchronize()
        Bq, T, Q, Cc, IH, IW, H, W = info["dims"]
        lab_real = torch.from_numpy(rd(info["lab"], (B, 2, S, S), np.int32)).cuda()
        if am is None:
            tab = rd(info["tab"], (5 * Bq * Q + 2 * Bq,), np.int32); nk = tab[5 * Bq * Q:5 * Bq * Q + Bq]
            scores = rd(info["scores"], (Bq, Q), np.float32); kept = rd(info["kept"], (Bq, Q), np.int32); n0 = int(nk[0])
            vol = torch.from_numpy(rd(info["p256"], (T, 256, 256, Q), np.float32))[..., torch.from_numpy(kept[0, :n0].astype(np.int64))]
            up = F.interpolate(vol.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False)
            am = (up * torch.from_numpy(scores[0, kept[0, :n0]]).view(1, n0, 1, 1)).argmax(1).int().cuda()
        line = []
        for v, lv in enumerate([lab_real] + [labs[v] for v in range(NV)]):
            w0 = int((lv[0] != am).sum())
            wrong0[v] += w0; wrong_fw[v] += w0 > 0
            line.append(str(w0))
        print(f"iter {it}: wrong label px of item 0 per variant: " + " | ".join(line), flush=True)
print("\nvariant: forwards with a wrong item-0 map / wrong px in total")
for v in range(NV + 1):
    print(f"  {names[v]:90s} {wrong_fw[v]:3d} / {wrong0[v]:6d}")
print("DONE", it + 1)
