Bq="--no-second-mode --no-roofline --no-render --no-cpu-baseline --warmup 5 --steps 30"
r() { python bench.py $Bq 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'; }
echo "base: $(r)"
for m in "0,1,m,s" "0,1,m,m" "0,0,m,s" "0,0,m,m" "m,m,m,m" "0,1,s,m" "0,m,1,s" "0,m,m,s" "0,1,1,0" "m,0,m,0"; do
echo "map $m: $(SIU3R_HEAD_MAP=$m r)"
done
