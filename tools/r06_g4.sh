cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python -m pytest tests/test_raster_gpu.py tests/test_compat_gpu.py tests/test_abi.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -15
python tools/ablate_chain.py bf16x3 1 2>&1 | grep -v amdgpu.ids
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r06_bench_a.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_a.json"))
print("pairs/s", d["value"], "ms", d["ms_per_step"])
for k in ("render", "render_pair_scene", "render_stress", "render_qc_logits", "network_plus_render"):
    v = d.get(k) or d.get("config", {}).get(k)
    if v: print(k, {kk: v[kk] for kk in ("ms_per_frame", "value", "ms_per_step") if kk in v})
PY
} > gpurun_out/r06_t4.txt 2>&1
cat gpurun_out/r06_t4.txt
