cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for st in spm int0 int3 seg; do
  ROCPROF_TRACE=1 ROCPROF_HEAD=1 bash tools/rocprof_cmd.sh r06_stage_$st python tools/stage_profile.py $st bf16x3 > /dev/null 2>&1
  python tools/stage_sequence.py gpurun_out/r06_stage_${st}_kernel_trace.csv > gpurun_out/r06_stage_${st}_sequence.txt 2>&1
  python tools/stage_breakdown.py gpurun_out/r06_stage_${st}_kernel_trace.csv > gpurun_out/r06_stage_${st}_breakdown.txt 2>&1
  rm -f gpurun_out/r06_stage_${st}_kernel_trace.csv gpurun_out/r06_stage_${st}_kernel_stats.csv
  head -3 gpurun_out/r06_stage_${st}_breakdown.txt
done
