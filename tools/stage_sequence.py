"""Kernel-by-kernel sequence of ONE replay of a stage graph (start offset, duration, workgroups, name) from the kernel trace of
tools/stage_profile.py: python tools/stage_sequence.py gpurun_out/<tag>_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
n = len(rows)
per = next(L for L in range(n // 5, 4, -1) if all(names[n - (r + 1) * L:n - r * L] == names[n - L:] for r in range(1, 5)))
last = rows[n - per:]
t0 = int(last[0]["Start_Timestamp"])
gi = lambda r, k: int(r.get(k, "1") or 1)
for r in last:
    wgs = (gi(r, "Grid_Size_X") // max(gi(r, "Workgroup_Size_X"), 1)) * (gi(r, "Grid_Size_Y") // max(gi(r, "Workgroup_Size_Y"), 1)) * (gi(r, "Grid_Size_Z") // max(gi(r, "Workgroup_Size_Z"), 1))
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  wgs {wgs:6d}  {r['Kernel_Name'].replace('(anonymous namespace)::', '')[:100]}")
