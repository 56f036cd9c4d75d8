"""Replay tuner: record every siu3r_gemm parameter block of one real model step (all step buffers kept alive), then graph-time each
unique launch under every tile configuration (siu3r_gemm_params.tile_cfg) and print what the library's cost model picks next to the
measured best.  python tools/gemm_replay.py [B] [precision] [out.json] [--sweep-s]   (--sweep-s: also every split-K count per tile)"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from siu3r_amd import _lib, model as M, ops
from siu3r_amd import synthetic_weights as OW
from siu3r_amd.model import SIU3RModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
os.environ["SIU3R_NO_STREAMS"] = "1"
os.environ["SIU3R_NO_GRAPH"] = "1"
os.environ["SIU3R_GEMM_NO_TUNED"] = "1"  # "auto" below is the cost model
dev = torch.device("cuda", 0)

keep = []
_orig_init = M._Run.__init__


def _init(self, *a, **k):
    _orig_init(self, *a, **k)
    keep.append(self)


M._Run.__init__ = _init
model = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision=prec, device=dev)
images = torch.rand(B, 2, 3, 512, 512).to(dev)
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, 2, 1, 1).to(dev)
with torch.no_grad():
    model(images, K)
    model(images, K)
torch.cuda.synchronize()

records = []
_orig_launch = ops._gemm_launch


def _rec(p, dev=None):
    q = _lib.GemmParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(q))
    records.append(q)
    _orig_launch(p, dev)


ops._gemm_launch = _rec
# every temporary of the recorded step must outlive the replay: hold the allocator's blocks by disabling frees via references
import gc

gc.disable()
_empties = []
_orig_empty = torch.empty


def _empty(*a, **k):
    t = _orig_empty(*a, **k)
    _empties.append(t)
    return t


torch.empty = _empty
with torch.no_grad():
    model(images, K)
torch.empty = _orig_empty
ops._gemm_launch = _orig_launch
torch.cuda.synchronize()
print(f"{len(records)} launches recorded, {len(_empties)} buffers held")

ws, cnt = ops._splitk_workspace(dev)


def sig(p):
    return (p.m, p.n, p.k, max(1, p.batch), p.a_mode, p.out_mode, p.kh, int(bool(p.ln_stats)), int(bool(p.rope_cos)), int(bool(p.residual)), p.act, p.stride)


uniq = {}
for p in records:
    uniq.setdefault(sig(p), []).append(p)


def time_launch(p, cfg, S=0, n=10):
    """cfg: tile configuration (0 = the library's choice); S: split-K slices (0 = the library's choice for that tile)"""
    q = _lib.GemmParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(q))
    q.tile_cfg = cfg
    q.splitk = S
    q.sk_ws, q.sk_cnt, q.sk_ws_floats, q.sk_cnt_n = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
    pl = ops.gemm_plan(q)
    if (cfg != 0 and pl.tile_cfg != cfg) or (S != 0 and pl.splitk != S):
        return None, pl
    if (q.a_x3 and not pl.a_x3_ok) or (q.c_x3 and not pl.c_x3_ok):
        return None, pl  # the recorded launch reads / writes pre-split planes: only plans that can are candidates
    st = torch.cuda.current_stream().cuda_stream
    lib = _lib.lib()
    lib.siu3r_gemm(C.byref(q), st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s2 = torch.cuda.current_stream().cuda_stream
        for _ in range(n):
            lib.siu3r_gemm(C.byref(q), s2)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best, pl


rows = []
SWEEP_S = "--sweep-s" in sys.argv
print("     M      N      K   Z am om kh ln rp rs |  n | auto: cfg S sk   us | 128x64 us | pp256^2 | pp256x128 | pp128^2 | best (cfg, S)")
tot_auto = tot_best = tot_old = 0.0
for s, ps in sorted(uniq.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[0][3] * len(kv[1])):
    p = ps[0]
    x3 = bool(p.w_x3) and p.a_dtype == _lib.F32
    ksteps = p.kpad // (16 if x3 else 32)
    res = {}
    ta, pla = time_launch(p, 0)
    for cfg in (-1, 1, 2, 3):
        t, pl = time_launch(p, cfg)
        if t is None:
            continue
        res[(cfg, pl.splitk)] = t
        if SWEEP_S:
            for S in (1, 2, 3, 4, 6, 8):
                if S == pl.splitk or (S > 1 and ksteps // S < (8 if cfg > 0 else 16)):
                    continue
                t2, pl2 = time_launch(p, cfg, S)
                if t2 is not None:
                    res[(cfg, S)] = t2
    ca, sa, ska = pla.tile_cfg, pla.splitk, pla.skinny_rows
    bk = min(res, key=res.get)
    n = len(ps)
    tot_auto += ta * n
    tot_best += res[bk] * n
    olds = [v for (c_, _), v in res.items() if c_ == -1]
    tot_old += (min(olds) if olds else ta) * n

    def f(c_):
        cand = {k_: v for k_, v in res.items() if k_[0] == c_}
        if not cand:
            return "       -   "
        k_ = min(cand, key=cand.get)
        return f"{cand[k_]:8.1f}({k_[1]})"

    flag = "" if ta <= res[bk] * 1.06 else "  <-- model picks worse"
    print(f"{s[0]:6d} {s[1]:6d} {s[2]:6d} {s[3]:3d} {s[4]:2d} {s[5]:2d} {s[6]:2d} {s[7]:2d} {s[8]:2d} {s[9]:2d} | {n:3d} | {ca:3d} {sa} {ska:2d} {ta:7.1f} | {f(-1)} | {f(1)} | {f(2)} | {f(3)} | {bk}{flag}")
    rows.append(dict(sig=list(s), launches=n, auto=dict(cfg=ca, splitk=sa, skinny=ska, us=ta), us={f"{c_},{S_}": v for (c_, S_), v in res.items()}))
print(f"sum over the step: auto {tot_auto/1e3:.2f} ms, best-per-shape {tot_best/1e3:.2f} ms, 128x64 family {tot_old/1e3:.2f} ms")
out = [a for a in sys.argv[3:] if not a.startswith("--")]
if out:
    json.dump(dict(batch=B, precision=prec, rows=rows), open(out[0], "w"), indent=0)
