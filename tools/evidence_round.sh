#!/bin/bash
# usage (on the GPU box): bash tools/evidence_round.sh r03 -> gpurun_out/<tag>_*: the round's measured evidence (copy what is kept into profiles/)
tag=${1:-r03}
O=gpurun_out
mkdir -p $O
# launches / device time per step and kernel: difference of a 25-step and a 5-step profile
Bq="--no-second-mode --no-roofline --no-render --no-cpu-baseline --warmup 3"
ROCPROF_HEAD=1 bash tools/rocprof_cmd.sh ${tag}_s5 python bench.py $Bq --steps 5 > /dev/null
ROCPROF_HEAD=1 bash tools/rocprof_cmd.sh ${tag}_s25 python bench.py $Bq --steps 25 > /dev/null
python - $tag <<'PY' > $O/${tag}_step_kernels.txt
import csv, sys
tag = sys.argv[1]
ld = lambda f: {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = ld(f"gpurun_out/{tag}_s5_kernel_stats.csv"), ld(f"gpurun_out/{tag}_s25_kernel_stats.csv")
rows = sorted(((t - a.get(n, (0, 0))[1]) / 20e3, (c - a.get(n, (0, 0))[0]) / 20, n) for n, (c, t) in b.items() if c - a.get(n, (0, 0))[0] > 0)[::-1]
foreign = [r for r in rows if r[2].startswith("void at::") or r[2].startswith("__amd_rocclr") or "at::native" in r[2]]
print(f"one bf16x3 step (profiled: streams serialised): {sum(r[1] for r in rows):.0f} launches, {sum(r[0] for r in rows) / 1e3:.2f} ms of kernel time;"
      f" not ours (torch / runtime copies, fills, cats): {sum(r[1] for r in foreign):.0f} launches, {sum(r[0] for r in foreign):.0f} us")
for us, c, n in rows:
    print(f"{us:9.1f} us/step {c:7.1f} launches/step  {n[:170]}")
PY
# the same table for EAGER SINGLE-STREAM steps (no graphs, no concurrent chains: every kernel alone on the chip, back to back) -- the regime
# bench.py's roofline leg times with HIP events; its per-kernel averages are what that leg's numbers must agree with
ROCPROF_HEAD=1 SIU3R_NO_GRAPH=1 SIU3R_NO_STREAMS=1 bash tools/rocprof_cmd.sh ${tag}_e5 python bench.py $Bq --steps 5 > /dev/null
ROCPROF_HEAD=1 SIU3R_NO_GRAPH=1 SIU3R_NO_STREAMS=1 bash tools/rocprof_cmd.sh ${tag}_e25 python bench.py $Bq --steps 25 > /dev/null
python - $tag <<'PY' > $O/${tag}_eager_step_kernels.txt
import csv, sys
tag = sys.argv[1]
ld = lambda f: {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = ld(f"gpurun_out/{tag}_e5_kernel_stats.csv"), ld(f"gpurun_out/{tag}_e25_kernel_stats.csv")
rows = sorted(((t - a.get(n, (0, 0))[1]) / 20e3, (c - a.get(n, (0, 0))[0]) / 20, n) for n, (c, t) in b.items() if c - a.get(n, (0, 0))[0] > 0)[::-1]
print(f"one bf16x3 step, EAGER on ONE stream (SIU3R_NO_GRAPH=1 SIU3R_NO_STREAMS=1; rocprofv3 --kernel-trace --stats, 25-step minus 5-step profile): "
      f"{sum(r[1] for r in rows):.0f} launches, {sum(r[0] for r in rows) / 1e3:.2f} ms of kernel time")
for us, c, n in rows:
    print(f"{us:9.1f} us/step {c:7.1f} launches/step  {n[:170]}")
PY
bash tools/pmc_round.sh $tag > $O/${tag}_pmc_round.log 2>&1
# The profiles bench.py READS (per-kernel averages of the eager and the replayed step, counter summaries) are installed into profiles/ of THIS
# copy of the tree before the bench lines run: the line's `consistency` / `traffic` / `in_step` then refer to this call on this board
for f in eager_step_kernels.txt step_kernels.txt pmc_summary.json mfma_busy.json; do [ -s $O/${tag}_$f ] && cp $O/${tag}_$f profiles/${tag}_$f; done
python -c "import torch; print(torch.cuda.get_device_properties(0).uuid)" 2>/dev/null | tail -1 > $O/${tag}_profile_board.txt; cp $O/${tag}_profile_board.txt profiles/${tag}_profile_board.txt  # the board these profiles are from
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench.json 2> $O/${tag}_bench.err  # (the driver's arguments)
timeout 400 python bench.py --batch 8 --no-cpu-baseline --no-render > $O/${tag}_bench_b8.json 2>> $O/${tag}_bench.err
ROCPROF_HEAD=3 bash tools/rocprof_cmd.sh ${tag}_bench python bench.py --no-cpu-baseline > /dev/null
tail -1 $O/${tag}_bench_out.txt > $O/${tag}_bench_under_rocprof.json
ROCPROF_HEAD=12 bash tools/rocprof_cmd.sh ${tag}_raster_stress python tools/mb_raster.py stress > $O/${tag}_raster_stress.log 2>&1
ROCPROF_HEAD=12 bash tools/rocprof_cmd.sh ${tag}_raster_pair python tools/mb_raster.py pair 6 > $O/${tag}_raster_pair.log 2>&1
timeout 600 python tools/config3.py > $O/${tag}_config3.json 2> $O/${tag}_config3.err
timeout 300 python tools/config5.py bf16x3 > $O/${tag}_config5.json 2> $O/${tag}_config5.err
timeout 200 python tools/timeline.py bf16x3 1 > $O/${tag}_timeline_bf16x3.txt 2>&1
timeout 200 python tools/timeline.py bf16 1 > $O/${tag}_timeline_bf16.txt 2>&1
timeout 200 python tools/stage_times.py bf16x3 > $O/${tag}_stage_times.txt 2>&1
timeout 200 python tools/mb_presplit.py > $O/${tag}_mb_presplit.txt 2>&1   # the encoder block's GEMMs with fp32 and with pre-split operands
timeout 200 python tools/mb_feat.py 168 84 40 > $O/${tag}_mb_feat.txt 2>&1   # N-channel list composite: 32-channel-chunk kernel vs matrix-core form
bash tools/pmc_feat.sh > $O/${tag}_pmc_feat.txt 2>&1
( for e in SIU3R_NO_PRESPLIT SIU3R_NO_DEC_QKVX SIU3R_NO_CONV_PLANES SIU3R_NO_KV_PLANES NONE; do   # same-box A/B of the round's switches
    env $e=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-second-mode --no-roofline --no-render --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$e=1', round(d['value'], 2), 'pairs/s', round(d['ms_per_step'], 3), 'ms')"
  done ) > $O/${tag}_switches.txt 2>&1
if [ -n "$EVIDENCE_MICRO" ]; then  # micro-benchmarks and probes of kernels that did not change since round 3: on request
timeout 400 python tools/mb_pp.py bench big > $O/${tag}_mb_pp.txt 2>&1
timeout 200 python tools/mb_attn.py > $O/${tag}_mb_attn.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/probes/dma_probe.hip -o /tmp/dma_probe 2>/dev/null && timeout 120 /tmp/dma_probe > $O/${tag}_dma_probe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o /tmp/store_probe 2>/dev/null && timeout 120 /tmp/store_probe > $O/${tag}_store_probe.txt 2>&1
fi
ls -la $O | grep ${tag}_ | wc -l
