cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -6
python tools/stage_times.py bf16x3 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-roofline 2>&1 | tail -1 | cut -c1-260
} > gpurun_out/r06_t9.txt 2>&1
cat gpurun_out/r06_t9.txt
