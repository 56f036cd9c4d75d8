cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "1 bf16x3" "8 bf16x3" "1 bf16" "8 bf16"; do
  set -- $cfg
  timeout 1500 python tools/gemm_replay.py $1 $2 gpurun_out/replay9_b$1_$2.json --sweep-s > gpurun_out/r06_gemm_b$1_$2_replay.txt 2>&1
  tail -2 gpurun_out/r06_gemm_b$1_$2_replay.txt | cut -c1-200
done
ls -la gpurun_out | grep replay
