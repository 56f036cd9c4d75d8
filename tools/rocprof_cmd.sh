#!/bin/bash
# usage: tools/rocprof_cmd.sh <tag> <command...>   -> gpurun_out/<tag>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats of any command)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_out.txt 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_err.txt < /dev/null )
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv; head -${ROCPROF_HEAD:-30} "$f" | cut -c1-220; else echo "no stats file"; find /tmp/prof_$tag | head; tail -5 $GRAFT_REPO_ROOT/gpurun_out/${tag}_err.txt; fi
if [ -n "$ROCPROF_TRACE" ]; then t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && cp "$t" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_trace.csv; fi
tail -3 $GRAFT_REPO_ROOT/gpurun_out/${tag}_out.txt
