#!/bin/bash
# usage: tools/ab_env.sh "<ENV=1 ...>" <bench args...>: bench.py without and with the environment setting, same box, value / ms per step of both modes
envs=$1; shift
for e in "" "$envs"; do
  for i in 1 2; do
    env $e python bench.py --no-cpu-baseline --no-render --no-roofline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('second_mode') or {}
print('[%s] %s: %.2f pairs/s %.2f ms | %s %.2f pairs/s %.2f ms' % ('$e' or 'default', d['config']['precision'], d['value'], d['ms_per_step'], s.get('precision'), s.get('value',0), s.get('ms_per_step',0)))"
  done
done
