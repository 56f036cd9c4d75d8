#!/bin/bash
# usage: tools/pmc_pass.sh <tag>  -> gpurun_out/<tag>_pmc_summary.json
# Two separate counter passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950), kernel-trace only.
tag=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${tag}_$c
  SIU3R_NO_GRAPH=1 SIU3R_NO_STREAMS=1 timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$c -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-render > /tmp/pmc_${tag}_$c.out 2> /tmp/pmc_${tag}_$c.err < /dev/null
  tail -2 /tmp/pmc_${tag}_$c.err
done
python - $tag <<'PY'
import csv, glob, json, sys, os
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"/tmp/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counter csv for", c, glob.glob(f"/tmp/pmc_{tag}_{c}/**/*", recursive=True)[:8]); continue
    rows = list(csv.DictReader(open(fs[0])))
    print(c, len(rows), "rows; columns", list(rows[0].keys()))
    for r in rows:
        if r.get("Counter_Name") != c: continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d = out.setdefault(n, {"launches": {}, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
        d[c] += float(r["Counter_Value"])
        d["launches"][c] = d["launches"].get(c, 0) + 1
summ = {}
for n, d in out.items():
    lf, lw = d["launches"].get("FETCH_SIZE", 0), d["launches"].get("WRITE_SIZE", 0)
    summ[n] = dict(launches=max(lf, lw), fetch_size_per_launch=d["FETCH_SIZE"] / max(lf, 1), write_size_per_launch=d["WRITE_SIZE"] / max(lw, 1))
json.dump(summ, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"{tag}_pmc_summary.json"), "w"), indent=1)
for n, v in sorted(summ.items(), key=lambda kv: -kv[1]["fetch_size_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{n[:60]:60s} n={v['launches']:5d} fetch/launch={v['fetch_size_per_launch']:12.1f} write/launch={v['write_size_per_launch']:12.1f}")
PY
