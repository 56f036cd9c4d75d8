"""The driver contract of bench.py: one JSON line on stdout with the agreed keys (run short, without the CPU baselines)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "image-pairs/s" and d["value"] > 1.0 and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - d["config"]["pairs_per_step_per_gpu"]) < 1e-6  # value x step time = pairs per step
    assert d["config"]["precision"] == "bf16x3" and d["config"]["n_segments_per_step"] > 0, "the timed step must carry the non-empty panoptic workload"
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in ro, k
    assert ro["bound"] == "mfma" and ro["unit"] == "TFLOP/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and 0 < ro["frac"] < 1
    assert ro["traffic"] is None or ro["traffic"] > 0
    sm = d["second_mode"]
    assert sm["precision"] == "bf16" and sm["value"] > d["value"] and "roofline" in sm
    for leg in ("render", "render_pair_scene", "render_stress"):
        assert d[leg]["ms_per_frame"] > 0 and d[leg]["roofline"]["bound"] == "hbm" and 0 < d[leg]["roofline"]["frac"] < 1
    assert d["render_pair_scene"]["visible_frac"] >= 0.5 and d["render_stress"]["visible_frac"] >= 0.5
