cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_raster_gpu.py tests/test_compat_gpu.py tests/test_postprocess_gpu.py tests/test_abi.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_t3.txt
cat gpurun_out/r06_t3.txt
