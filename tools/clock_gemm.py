"""Shader clock and board power while ONE GEMM shape runs back to back (sysfs at 50 Hz), next to its rate: is the ping-pong GEMM's
"52 % of peak" a pipeline problem or the board's power limit?  python tools/clock_gemm.py [bf16x3|bf16]
Reads: achieved TF/s, MFMA-issue fraction of the peak AT THE MEASURED CLOCK (peak scales with sclk / 2400 MHz)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from siu3r_amd import _lib, ops
from clock_probe import find_sensors, Sampler, summarize

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
split = mode == "bf16x3"
adt = torch.float32 if split else torch.bfloat16
sensors = find_sensors()
print("sensors:", sensors)
s = Sampler(sensors)
s.start()
time.sleep(1.0)
for (M, N, K, cfg, zero) in [(4096, 4096, 4096, 1, False), (4096, 4096, 4096, 1, True), (16400, 4096, 1024, 1, False), (2048, 4096, 1024, 2, False), (2050, 4096, 1024, 0, False),
                             (2050, 1024, 1024, 0, False)]:
    a = ((torch.rand(M, K, device="cuda") * 2 - 1) * (0 if zero else 1)).to(adt)
    pw = ops.pack_linear((torch.rand(N, K, device="cuda") * 2 - 1) * (0 if zero else 0.1), torch.zeros(N, device="cuda"), split)
    out = torch.empty(M, N, device="cuda", dtype=adt)
    _lib.check(_lib.lib().siu3r_gemm_tune(0, cfg))
    for _ in range(3):
        ops.linear(a, pw, out=out)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            ops.linear(a, pw, out=out)
    torch.cuda.synchronize()
    tag = f"{M}x{N}x{K} cfg{cfg}{' zeros' if zero else ''}"
    s.tag = tag
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.5:
        g.replay()
        n += 50
    torch.cuda.synchronize()
    dt = time.time() - t0
    s.tag = "idle"
    tf = 2.0 * M * N * K * n / dt / 1e12
    rs = [r["sclk_hz"] * 1e-6 for r in s.rows if r["tag"] == tag and r.get("sclk_hz", -1) > 0]
    rs.sort()
    clk = rs[len(rs) // 2] if rs else 0
    passes = 3 if split else 1
    frac_nom = tf * passes / 2500.0
    frac_clk = frac_nom * 2400.0 / clk if clk else 0
    print(summarize(s.rows, tag))
    print(f"   {tag}: {dt / n * 1e6:8.1f} us/launch  {tf:7.1f} TF/s alg.  MFMA issue {frac_nom * 100:5.1f} % of the 2.4 GHz peak, {frac_clk * 100:5.1f} % of the peak at {clk:.0f} MHz")
    time.sleep(0.5)
_lib.check(_lib.lib().siu3r_gemm_tune(0, 0))
s.stop_flag = True
