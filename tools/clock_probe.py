"""Shader clock and board power while the network runs (sysfs, sampled from a thread at ~50 Hz): is the step's 'work-bound' behaviour --
kernels 20 % slower inside the multi-stream step than alone, neutral pipelining -- contention for CUs or a clock that drops under load?
Phases: idle; ONE stage graph replayed alone (enc0: under-filled encoder launches; gs: full-chip convolutions); whole steps back to back.
    python tools/clock_probe.py [precision]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd.model import SIU3RModel
from siu3r_amd import synthetic_weights as OW


def find_sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name, key in (("freq1_input", "sclk_hz"), ("freq2_input", "mclk_hz"), ("power1_average", "power_uw"), ("power1_input", "power_uw")):
            p = os.path.join(hw, name)
            if os.path.exists(p) and key not in out:
                try:
                    int(open(p).read())
                    out[key] = p
                except Exception:
                    pass
        if out:
            break
    return out


class Sampler(threading.Thread):
    def __init__(self, sensors):
        super().__init__(daemon=True)
        self.sensors, self.rows, self.stop_flag, self.tag = sensors, [], False, "idle"

    def run(self):
        while not self.stop_flag:
            r = {"tag": self.tag}
            for k, p in self.sensors.items():
                try:
                    r[k] = int(open(p).read())
                except Exception:
                    r[k] = -1
            self.rows.append(r)
            time.sleep(0.02)


def summarize(rows, tag):
    rs = [r for r in rows if r["tag"] == tag]
    if not rs:
        return f"{tag}: no samples"
    out = [f"{tag:>22}: {len(rs):4d} samples"]
    for k, scale, unit in (("sclk_hz", 1e-6, "MHz"), ("power_uw", 1e-6, "W"), ("mclk_hz", 1e-6, "MHz")):
        v = [r[k] * scale for r in rs if r.get(k, -1) > 0]
        if v:
            v.sort()
            out.append(f"{k.split('_')[0]} min {v[0]:7.0f} median {v[len(v) // 2]:7.0f} max {v[-1]:7.0f} {unit}")
    return "   ".join(out)


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
    sensors = find_sensors()
    print("sensors:", sensors)
    dev = torch.device("cuda", 0)
    m = SIU3RModel(OW.make_weights(0), image_size=(512, 512), precision=prec, device=dev)
    img = torch.rand(1, 2, 3, 512, 512).to(dev)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).to(dev)
    for _ in range(4):
        m(img, K)
    torch.cuda.synchronize()
    ent = next(iter(m._graphs.values()))
    s = Sampler(sensors)
    s.start()
    time.sleep(1.5)
    res = {}
    for name in ("enc0", "gs", "seg"):
        g = ent["graphs"].get(name)
        if g is None:
            continue
        g.replay()
        torch.cuda.synchronize()
        s.tag = "alone:" + name
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(200, int(2500 / {"enc0": 1.6, "gs": 3.5, "seg": 4.0}[name]))
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / n
        s.tag = "idle"
        time.sleep(0.7)
    s.tag = "steps"
    t0 = time.perf_counter()
    n = 150
    for _ in range(n):
        m(img, K)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / n * 1e3
    s.tag = "idle"
    time.sleep(0.5)
    s.stop_flag = True
    s.join()
    print(f"{prec}: stage graphs alone (ms): {({k: round(v, 3) for k, v in res.items()})}; step {step_ms:.2f} ms")
    for tag in ("idle", "alone:enc0", "alone:gs", "alone:seg", "steps"):
        print(summarize(s.rows, tag))


if __name__ == "__main__":
    main()
