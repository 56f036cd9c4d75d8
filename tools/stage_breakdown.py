"""Per-kernel composition of ONE replay of a stage graph from a rocprofv3 kernel trace of tools/stage_profile.py (which replays the stage 5x at the
end of the process): python tools/stage_breakdown.py gpurun_out/<tag>_kernel_trace.csv  -> kernels of the last replay, grouped by name"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
# the five replays are identical sequences at the end: find the period by matching the tail
names = [r["Kernel_Name"] for r in rows]
per = None
for L in range(n // 5, 4, -1):  # the LONGEST period that repeats five times at the tail (a stage may itself contain repeating layers)
    if all(names[n - (r + 1) * L:n - r * L] == names[n - L:] for r in range(1, 5)):
        per = L
        break
assert per, "no repeating tail found"
last = rows[n - per:]
t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
agg = collections.OrderedDict()
busy = 0
for r in last:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    a = agg.setdefault(r["Kernel_Name"], [0, 0])
    a[0] += 1
    a[1] += d
print(f"one replay: {per} launches, span {(t1 - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us")
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d / 1e3:9.1f} us {c:4d} x {d / c / 1e3:7.1f} us  {k.replace('(anonymous namespace)::', '')[:120]}")
