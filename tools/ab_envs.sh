# usage (GPU box): bash tools/ab_envs.sh -> launch-floor (a 100 x 256 x 256 GEMM in a 10-node graph chain) and bench.py pairs/s under runtime switches
Bq="--no-second-mode --no-roofline --no-render --no-cpu-baseline --warmup 5 --steps 30"
r() { python bench.py $Bq 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'; }
f() { python tools/mb_one.py bf16x3 100 256 256 2>/dev/null | tail -1 | sed 's/.*cfg=0://'; }
echo "base: floor $(f) | $(r)"
for e in "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_NO_SCRATCH_RECLAIM=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=2" "HIP_FORCE_DEV_KERNARG=1 HSA_NO_SCRATCH_RECLAIM=1 HSA_ENABLE_INTERRUPT=0"; do
echo "$e: floor $(env $e python tools/mb_one.py bf16x3 100 256 256 2>/dev/null | tail -1 | sed 's/.*cfg=0://') | $(env $e python bench.py $Bq 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")')"
done
