#!/bin/bash
# usage: tools/stage_profile.sh <stage>: kernels of one stage graph (one replay), aggregated
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sp -- python $GRAFT_REPO_ROOT/tools/stage_profile.py $1 $2 > /dev/null 2>&1
f=$(find /tmp/sp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last_pp = max(i for i, n in enumerate(names) if "pp_" in n or "lift" in n or "gaussian_adapter" in n)
seg = rows[last_pp + 1:]
n = len(seg) // 5
one = seg[-n:]
agg = {}
for r in one:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]
    d = agg.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernels per replay", n, "sum us", round(sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{v[1]:8.1f} us {v[0]:4d} x {v[1]/v[0]:7.1f}  {k}")
PY
