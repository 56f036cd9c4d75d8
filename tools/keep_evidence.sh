#!/bin/bash
# tools/keep_evidence.sh r03: copy the files of tools/evidence_round.sh that are kept under version control from gpurun_out/ to profiles/
tag=${1:-r03}
for f in bench.json bench_b8.json bench_kernel_stats.csv bench_under_rocprof.json step_kernels.txt eager_step_kernels.txt mb_feat.txt pmc_feat.txt pmc_summary.json mfma_busy.json raster_stress_kernel_stats.csv \
         raster_pair_kernel_stats.csv config3.json config5.json timeline_bf16x3.txt timeline_bf16.txt stage_times.txt mb_presplit.txt switches.txt profile_board.txt mb_pp.txt mb_attn.txt dma_probe.txt store_probe.txt; do
  [ -s gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f || echo "not kept (absent): gpurun_out/${tag}_$f"
done
# (the replay tables the tile table was generated from are copied by hand when the table is regenerated: profiles/<tag>_gemm_*_replay.txt)
ls profiles | grep ${tag}_ | wc -l
