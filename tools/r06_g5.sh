cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -q -m gpu -s 2>&1 | grep -E "model-parity|\[golden\]|passed|failed|Error" > gpurun_out/r06_parity_lines.txt
tail -3 gpurun_out/r06_parity_lines.txt
