#!/usr/bin/env python3
"""Scratch.  `python profiles/c3_timeline.py run` = bench.sub_bench_c3 with a few steps (the command to put under
rocprofv3 --kernel-trace); `python profiles/c3_timeline.py <trace dir>` = the kernels between the last two channelizer
passes: start, duration and stream/queue, to see what overlaps with what."""
import csv, glob, json, sys
from pathlib import Path
if sys.argv[1] == "run":
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    r = bench.sub_bench_c3(steps=6, warm=3, cpu_seconds_of_signal=0.05)
    print(json.dumps({k: r[k] for k in ("ms_per_step",)}), r["roofline"]["kernel_ms"])
    sys.exit(0)
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
big = [i for i, r in enumerate(rows) if "ring_pairs" in r["Kernel_Name"] or "ring_multi" in r["Kernel_Name"]]
big = [i for i in big if int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) > 2_000_000]
i0, i1 = big[-3], big[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0 : i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:10.1f} us  dur {(e-s)/1e3:8.1f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'].replace('void ', '').replace('iqa::', '')[:60]}")
