#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <command...>  -> gpurun_out/<tag>_pmc_summary.json
# Separate counter passes (FETCH_SIZE, WRITE_SIZE do not fit one pass on gfx950; a third pass collects MFMA / VALU busy cycles),
# kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).  FETCH_SIZE is doubled when compared with byte counts (same guide).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE ${PMC_EXTRA}; do
  cs=$(echo $c | tr ',' ' ')
  d=/tmp/pmc_${tag}_$(echo $c | tr ',' '_')
  rm -rf $d
  ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $d -- "$@" > $d.out 2> $d.err < /dev/null )
  tail -2 $d.err
done
python - $tag "FETCH_SIZE WRITE_SIZE ${PMC_EXTRA}" <<'PY'
import csv, glob, json, sys, os
tag, groups = sys.argv[1], sys.argv[2].split()
out = {}
for grp in groups:
    d = f"/tmp/pmc_{tag}_{grp.replace(',', '_')}"
    fs = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counter csv for", grp, glob.glob(f"{d}/**/*", recursive=True)[:8]); continue
    rows = list(csv.DictReader(open(fs[0])))
    print(grp, len(rows), "rows")
    for r in rows:
        c = r.get("Counter_Name")
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        e = out.setdefault(n, {})
        e.setdefault(c, [0.0, 0])
        e[c][0] += float(r["Counter_Value"]); e[c][1] += 1
summ = {n: {c: dict(total=v[0], launches=v[1], per_launch=v[0] / max(v[1], 1)) for c, v in e.items()} for n, e in out.items()}
json.dump(summ, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"{tag}_pmc_summary.json"), "w"), indent=1)
key = lambda kv: -kv[1].get("FETCH_SIZE", {}).get("total", 0)
for n, e in sorted(summ.items(), key=key)[:16]:
    print(f"{n[:56]:56s} " + " ".join(f"{c}={v['per_launch']:.4g}x{v['launches']}" for c, v in e.items()))
PY
