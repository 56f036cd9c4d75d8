cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python tools/clock_gemm.py bf16x3 2>&1 | grep -v amdgpu.ids
python tools/clock_gemm.py bf16 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_clock_gemm.txt 2>&1
cat gpurun_out/r06_clock_gemm.txt
