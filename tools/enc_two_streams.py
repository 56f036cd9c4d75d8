"""Probe: the encoder of a pair as ONE chain over both views (M = 2050 rows per GEMM, as shipped) against TWO chains, one view each
(M = 1025), on two streams.  The views of a pair are independent in the encoder (backbone_croco.py:270-300: a batch of 2 images); a
launch of one chain is a single round of tiles (prologue -> K loop -> store burst, nothing overlapping it), two chains side by side let
one chain's K loops run under the other's prologues, epilogues and launch gaps.
  python tools/enc_two_streams.py [bf16x3|bf16] [blocks]
Measured (round 6, one board): bf16x3 6.305 ms | one view alone 4.543 | two chains 6.299; bf16 3.895 | 2.974 | 3.660 -- no gain, the batched chain stays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd.model import SIU3RModel
from siu3r_amd import synthetic_weights as OW

dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 24
m = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision=prec, device=dev)
bb = m.backbone
img = torch.rand(1, 2, 3, 512, 512).to(dev)
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).to(dev)


def chain(e):
    bb.encode_blocks(e, 0, nblk)
    return e["x"]


def capture(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()  # warm-up (weights packed, plans made)
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=stream):
        out = fn()
    return g, out


def timed(run, n=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
# (a) one chain, both views
eb = bb.encode_begin(img, K)
g_both, x_both = capture(lambda: chain(dict(eb, all_feat=[])), s0)
t_both = timed(lambda: g_both.replay())

# (b) two chains: view v alone
def view_state(v):
    """view v of the two-view state as a state of its own (rows, row statistics and positions copied)"""
    from siu3r_amd import ops
    x, xs, xb = eb["S"][:3]
    n = x.shape[1]
    x1 = x[v:v + 1].clone()
    st = None
    if xs is not None:
        st = ops.RowStats(x1)
        st.buf.copy_(xs.buf[v * n:(v + 1) * n])
    return dict(eb, x=x1, S=(x1, st, None if xb is None else xb[v:v + 1].clone()), pos=eb["pos"][v:v + 1].contiguous(), all_feat=[])


ev = [view_state(0), view_state(1)]

gA, xA = capture(lambda: chain(dict(ev[0], all_feat=[])), s0)
gB, xB = capture(lambda: chain(dict(ev[1], all_feat=[])), s1)
t_one = timed(lambda: gA.replay())


def run_two():
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur); s1.wait_stream(cur)
    with torch.cuda.stream(s0):
        gA.replay()
    with torch.cuda.stream(s1):
        gB.replay()
    cur.wait_stream(s0); cur.wait_stream(s1)


t_two = timed(run_two)
print(f"{prec}, {nblk} encoder blocks: both views in one chain {t_both:.3f} ms | one view alone {t_one:.3f} ms | two one-view chains on two streams {t_two:.3f} ms")
