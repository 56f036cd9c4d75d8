"""Merge the per-command counter summaries of tools/pmc_round.sh into <tag>_pmc_summary.json (sections bench_bf16x3, bench_bf16,
raster_stress, raster_pair: per kernel FETCH_SIZE / WRITE_SIZE in KiB, total / launches / per launch) and <tag>_mfma_busy.json
(per GEMM / attention kernel: SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE x 1024 SIMDs, and the MFMA count it implies)."""
import json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
merged, busy = {}, {}
XCDS = 8.0
for sec in ("bench_bf16x3", "bench_bf16", "raster_stress", "raster_pair"):
    f = os.path.join(out_dir, f"{tag}_{sec}_pmc_summary.json")
    if not os.path.exists(f):
        print("missing", f)
        continue
    s = json.load(open(f))
    merged[sec] = {n: {c: e[c] for c in ("FETCH_SIZE", "WRITE_SIZE") if c in e} for n, e in s.items()}
    if sec.startswith("bench"):
        b = {}
        for n, e in s.items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in e or e["SQ_VALU_MFMA_BUSY_CYCLES"]["total"] <= 0:
                continue
            mf, act = e["SQ_VALU_MFMA_BUSY_CYCLES"]["total"], e.get("GRBM_GUI_ACTIVE", {}).get("total", 0.0)
            b[n] = {"launches": e["SQ_VALU_MFMA_BUSY_CYCLES"]["launches"], "SQ_VALU_MFMA_BUSY_CYCLES": mf, "GRBM_GUI_ACTIVE": act,
                    "SQ_BUSY_CYCLES": e.get("SQ_BUSY_CYCLES", {}).get("total"),
                    "mfma_busy_frac": mf / (act / XCDS * 1024.0) if act > 0 else None,  # 256 CUs x 4 SIMDs, one MFMA pipe each
                    "mfma_32x32x16_count": mf / 32.0}
        tot_mf = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in b.values())
        tot_act = sum(e.get("GRBM_GUI_ACTIVE", {}).get("total", 0.0) for e in s.values())
        busy[sec] = {"kernels": dict(sorted(b.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"])),
                     "all_kernels_mfma_busy_frac": tot_mf / (tot_act / XCDS * 1024.0) if tot_act > 0 else None,
                     "note": "SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md); denominator = "
                             "GRBM_GUI_ACTIVE / 8 (the counter comes back summed over the 8 XCDs: per launch, value / 8 = launch duration x ~2.5 GHz; "
                             "kernels are serialised by the profiler) x 1024 SIMDs"}
json.dump(merged, open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w"), indent=1)
json.dump(busy, open(os.path.join(out_dir, f"{tag}_mfma_busy.json"), "w"), indent=1)
for sec, b in busy.items():
    print(sec, "all kernels MFMA busy:", b["all_kernels_mfma_busy_frac"])
    for n, v in list(b["kernels"].items())[:8]:
        print(f"   {n[:70]:70s} busy {v['mfma_busy_frac']}")
