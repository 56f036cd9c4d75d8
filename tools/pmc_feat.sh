#!/bin/bash
# tools/pmc_feat.sh: SQ / MFMA counters of the N-channel list composite (tools/mb_feat4.py: 168 channels, pair scene, 6 views), one --pmc pass per group
cd /tmp && export TMPDIR=/tmp
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  d=/tmp/pmcf_$(echo $grp | tr ' ' '_' | cut -c1-40); rm -rf $d
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python tools/mb_feat.py 168 > $d.out 2> $d.err < /dev/null )
  python - $d <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no csv", open(sys.argv[1] + ".err").read()[-400:]); sys.exit()
acc = {}
for r in csv.DictReader(open(fs[0])):
    if "composite_feat" not in r["Kernel_Name"] and "ql_build" not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
    a = acc.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for (n, c), (v, k) in sorted(acc.items()):
    print(f"{n:40s} {c:28s} per launch {v / k:.4g}  ({k} launches)")
ks = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if ks:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(ks[0])) if "composite_feat" in r["Kernel_Name"]]
    if d: print(f"   composite_feat launches {len(d)}, mean duration {sum(d) / len(d) / 1e3:.1f} us")
PY
done
