#!/bin/bash
# usage: tools/rocprof_bench.sh <tag> [bench args...]   -> gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-render "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_err.txt < /dev/null
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv; head -40 "$f"; else echo "no stats file"; find /tmp/prof_$tag | head; tail -5 $GRAFT_REPO_ROOT/gpurun_out/${tag}_err.txt; fi
tail -1 $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json
