#!/bin/bash
# usage: tools/trace_step.sh <tag> -> gpurun_out/<tag>_trace.csv (kernel trace of a few steps, graph replay)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-render "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_err.txt < /dev/null
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python - "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as fh:
    fh.write("start_us,dur_us,queue,grid,wg,name\n")
    for r in rows:
        n = r["Kernel_Name"]
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        fh.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:.1f},{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.1f},{r.get('Queue_Id','')},{r.get('Grid_Size_X', r.get('Grid_Size',''))},{r.get('Workgroup_Size_X', r.get('Workgroup_Size',''))},{n}\n")
print(len(rows), "kernels")
PY
else echo "no trace"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/${tag}_err.txt; fi
tail -c 400 $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json
