cd $GRAFT_REPO_ROOT
bash tools/evidence_round.sh r06 > gpurun_out/r06_evidence.log 2>&1
tail -3 gpurun_out/r06_evidence.log
python -c "
import json
d=json.load(open('gpurun_out/r06_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('second_mode',{}).get('value'))
"
