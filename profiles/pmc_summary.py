#!/usr/bin/env python3
"""Summarise the rocprofv3 passes made by profiles/profile_gpu.sh into small files fit for profiles/.

HBM traffic per launch of the dominant kernel, as MI355X_MICROARCH.md prescribes for gfx950:
FETCH_SIZE and WRITE_SIZE come from separate --pmc passes, are reported in KiB, and FETCH_SIZE
under-reports wide (16 B/lane) coalesced streaming reads by exactly 2x, so
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import json
import sys
from pathlib import Path

out_dir, tag = Path(sys.argv[1]), sys.argv[2]
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "k_channelize_mfma"


def find(sub, pattern):
    hits = glob.glob(str(out_dir / sub / "**" / pattern), recursive=True)
    return hits[0] if hits else None


def biggest_counter(sub, counter):
    """Counter value of the largest dispatch of the dominant kernel (the full-capture launch)."""
    path = find(sub, "*counter_collection.csv")
    best = None
    if not path:
        return None
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if KERNEL in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                v = float(row["Counter_Value"])
                grid = int(row.get("Grid_Size", 0) or 0)
                if best is None or grid > best[0]:
                    best = (grid, v)
    return None if best is None else best[1]


summary = {"tag": tag, "kernel": None}
stats = find("stats", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    (Path(out_dir) / f"{tag}_kernel_stats.csv").write_text(open(stats).read())
    for r in rows:
        if KERNEL in r["Name"]:
            summary["kernel"] = r["Name"].split("(")[0].replace("void ", "")
            summary["stats_calls"] = int(r["Calls"])
            summary["stats_max_ns"] = float(r["MaxNs"])
            break
trace = find("stats", "*kernel_trace.csv")
if trace:
    durs = []
    with open(trace) as fh:
        for row in csv.DictReader(fh):
            if KERNEL in row["Kernel_Name"]:
                grid = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0)
                durs.append((grid, int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    if durs:
        gmax = max(g for g, _ in durs)
        big = [d for g, d in durs if g == gmax]
        summary["full_launch_avg_ms"] = sum(big) / len(big) / 1e6
        summary["full_launches"] = len(big)
fetch = biggest_counter("fetch", "FETCH_SIZE")
write = biggest_counter("write", "WRITE_SIZE")
summary["FETCH_SIZE_KiB"] = fetch
summary["WRITE_SIZE_KiB"] = write
if fetch is not None and write is not None:
    summary["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
try:
    bench = json.loads(open(out_dir / "stats_bench.json").read().strip().splitlines()[-1])
    summary["workload_frames"] = bench["config"]["frames_per_gpu"]
    summary["bench_kernel_ms_hip_events"] = bench["roofline"]["kernel_ms"]
    summary["algorithmic_bytes_per_launch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
except Exception as exc:  # noqa: BLE001
    summary["bench_parse_error"] = str(exc)
(Path(out_dir) / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary))
