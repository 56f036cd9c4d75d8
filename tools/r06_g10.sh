cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "parity_sweep or multi" 2>&1 | grep -E "parity-sweep|passed|failed|Error|assert" > gpurun_out/r06_t10.txt
cat gpurun_out/r06_t10.txt
