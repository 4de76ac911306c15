#!/usr/bin/env python3
"""Kernel-by-kernel timeline of the last hipGraph replays of config 1's step in a rocprofv3 --kernel-trace directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
big = [i for i, r in enumerate(rows) if "k_channelize_mfma" in r["Kernel_Name"] and "short" not in r["Kernel_Name"]
       and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 12_000]
i0, i1 = big[-4], big[-2]
t0, prev_end = int(rows[i0]["Start_Timestamp"]), None
for r in rows[i0 : i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = 0 if prev_end is None else s - prev_end
    prev_end = max(e, prev_end or e)
    print(f"{s/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {gap/1e3:7.1f}  {r['Kernel_Name'].replace('void ', '')[:70]}")
