Bq="--no-second-mode --no-roofline --no-render --no-cpu-baseline --warmup 5 --steps 30"
r() { python bench.py $Bq 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'; }
echo "base: $(r)"
echo "own stream, 4 queues: $(SIU3R_PTS0_OWN=1 r)"
echo "own stream, 5 queues: $(SIU3R_PTS0_OWN=1 GPU_MAX_HW_QUEUES=5 r)"
echo "seg stream, 5 queues: $(GPU_MAX_HW_QUEUES=5 r)"
echo "own stream, 6 queues: $(SIU3R_PTS0_OWN=1 GPU_MAX_HW_QUEUES=6 r)"
echo "pts0 main: $(SIU3R_PTS0_MAIN=1 r)"
echo "base again: $(r)"
