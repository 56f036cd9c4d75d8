#!/bin/bash
# usage (on the GPU box): bash tools/pmc_round.sh r02   -> gpurun_out/r02_pmc_summary.json, gpurun_out/r02_mfma_busy.json
# Counter passes of the four measured commands (bench in both precisions, rasterizer stress frame, rasterizer pair scene), each
# counter group in its own rocprofv3 run with --kernel-trace only (tools/pmc_cmd.sh).
tag=${1:-r02}
export PMC_EXTRA="SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_BUSY_CYCLES"
B="--steps 5 --warmup 3 --no-second-mode --no-cpu-baseline --no-roofline --no-render"
bash tools/pmc_cmd.sh ${tag}_bench_bf16x3 python bench.py $B --precision bf16x3 > gpurun_out/${tag}_pmc_bench_bf16x3.log 2>&1
bash tools/pmc_cmd.sh ${tag}_bench_bf16 python bench.py $B --precision bf16 > gpurun_out/${tag}_pmc_bench_bf16.log 2>&1
PMC_EXTRA="" bash tools/pmc_cmd.sh ${tag}_raster_stress python tools/mb_raster.py stress > gpurun_out/${tag}_pmc_raster_stress.log 2>&1
PMC_EXTRA="" bash tools/pmc_cmd.sh ${tag}_raster_pair python tools/mb_raster.py pair 6 > gpurun_out/${tag}_pmc_raster_pair.log 2>&1
python tools/pmc_merge.py $tag
