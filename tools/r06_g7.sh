cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for e in "" "SIU3R_NO_STREAMS=1"; do
  env $e python bench.py --no-cpu-baseline --no-render --no-roofline --no-second-mode --batch 8 --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s] B=8 %s: %.2f pairs/s %.2f ms' % ('$e' or 'default', d['config']['precision'], d['value'], d['ms_per_step']))"
done
python tools/ablate_chain.py bf16x3 8 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_b8.txt 2>&1
cat gpurun_out/r06_b8.txt
