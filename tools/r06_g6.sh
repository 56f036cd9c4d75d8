cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== base"; python tools/mb_msdeform.py 2>&1 | grep -v amdgpu.ids
echo "== new"; SIU3R_LIB_OVERRIDE=$PWD/siu3r_amd/libsiu3r_hip_msd.so python tools/mb_msdeform.py 2>&1 | grep -v amdgpu.ids
SIU3R_LIB_OVERRIDE=$PWD/siu3r_amd/libsiu3r_hip_msd.so python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "msdeform" 2>&1 | tail -2
} > gpurun_out/r06_msdeform.txt 2>&1
cat gpurun_out/r06_msdeform.txt
