#!/bin/bash
# usage (GPU box): bash tools/ab_depth.sh <tag> -> gpurun_out/<tag>_depth.txt: bench.py pairs/s with 1 / 2 / 3 steps in flight (forward_async), default and 8 hardware queues
tag=${1:-r04}
O=gpurun_out/${tag}_depth.txt
: > $O
Bq="--no-second-mode --no-roofline --no-render --no-cpu-baseline --warmup 5 --steps 30"
for d in 1 2 3; do
  echo "depth $d: $(python bench.py $Bq --depth $d 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")')" >> $O
done
for d in 2 3; do
  echo "depth $d GPU_MAX_HW_QUEUES=8: $(GPU_MAX_HW_QUEUES=8 python bench.py $Bq --depth $d 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")')" >> $O
done
cat $O
