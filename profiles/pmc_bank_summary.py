#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of profiles/pmc_bank.sh (multi-lane channelizer, BASELINE config 3's shape) into
profiles/<tag>_bank_*.  Usage: python profiles/pmc_bank_summary.py gpurun_out/pmc_bank_<tag> <tag>

HBM traffic as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and WRITE_SIZE from separate --pmc passes, in KiB;
FETCH_SIZE tallies the wide (16 B per lane) streaming reads of the ring kernel at half their bytes -> doubled for that
kernel; the combine kernel reads 8 B per lane and is counted as is (its figure equals its algorithmic bytes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict
from pathlib import Path

out, tag = Path(sys.argv[1]), sys.argv[2]
dest = Path(__file__).resolve().parent


def newest(sub, pattern):
    hits = glob.glob(str(out / sub / "**" / pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def counters(sub):
    path = newest(sub, "*counter_collection.csv")
    agg = defaultdict(lambda: defaultdict(list))
    if path:
        for r in csv.DictReader(open(path)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def med(v):
    return sorted(v)[len(v) // 2]


summary = {"tag": tag, "command": f"PREC={os.environ.get('PREC', 'fast')} python profiles/bench_bank.py c3 bank (K=6, WARM=2)",
           "workload": "BASELINE config 3: 60 s @ 20 MS/s = 1.2e9 frames, five targets nfm/am/usb/lsb/nfm = 10 lanes, D = 208"}
stats = newest("stats", "*kernel_stats.csv")
if stats:
    rows = [r for r in csv.DictReader(open(stats)) if "iqa::" in r["Name"]]
    (dest / f"{tag}_bank_kernel_stats.csv").write_text(
        "Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n" +
        "".join(f'"{r["Name"]}",{r["Calls"]},{r["TotalDurationNs"]},{r["AverageNs"]},{r["MinNs"]},{r["MaxNs"]},{r["Percentage"]}\n' for r in rows))
    summary["kernel_stats"] = {r["Name"].split("(")[0].replace("void ", ""): {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
                               for r in rows}
fetch, write = counters("fetch"), counters("write")
traffic = {}
for k in set(fetch) | set(write):
    if "iqa::" not in k:
        continue
    f = med(fetch[k]["FETCH_SIZE"]) if fetch[k]["FETCH_SIZE"] else 0.0
    w = med(write[k]["WRITE_SIZE"]) if write[k]["WRITE_SIZE"] else 0.0
    wide = "ring" in k  # 16-byte-per-lane LDS-DMA reads: tallied at half their bytes on gfx950
    traffic[k] = {"launches_profiled": len(fetch[k]["FETCH_SIZE"]), "FETCH_SIZE_KiB_median": f, "WRITE_SIZE_KiB_median": w,
                  "read_bytes_per_launch": (2.0 if wide else 1.0) * f * 1024.0, "written_bytes_per_launch": w * 1024.0}
summary["hbm_traffic"] = traffic
n, targets = 1.2e9, 5
algo = 4.0 * n + targets * 4.0 * 48000.0 / 20e6 * n
# bytes per CAPTURE: every launch of every channelizer kernel (the shared launches with int32 and with 64-bit sums, the
# combine launches) over all profiled captures, divided by their number (K + WARM of profiles/pmc_bank.sh)
captures = float(os.environ.get("CAPTURES", "8"))
prec = os.environ.get("PREC", "fast")
summary["workload_key"] = "c3_product_precisions" if prec == "product" else "c3_all_fast"
summary["precisions"] = prec
ring_bytes = comb_bytes = 0.0
for k in traffic:
    tot = ((2.0 if "ring" in k else 1.0) * sum(fetch[k]["FETCH_SIZE"]) + sum(write[k]["WRITE_SIZE"])) * 1024.0 / captures
    traffic[k]["bytes_per_capture"] = tot
    if "combine" in k:
        comb_bytes += tot
    elif "channelize" in k:
        ring_bytes += tot
if ring_bytes:
    summary["algorithmic_bytes_per_capture"] = algo
    summary["multi_lane_kernel_bytes_per_capture"] = ring_bytes
    summary["multi_lane_kernel_traffic_over_algorithmic"] = ring_bytes / algo
    summary["combine_kernels_bytes_per_capture"] = comb_bytes
    summary["channelizer_stage_traffic_over_algorithmic"] = (ring_bytes + comb_bytes) / algo
by_kernel = defaultdict(dict)
for p in ("p1", "p2", "p3"):
    c = counters(p)
    for k, v in c.items():
        if "ring_multi" in k or "ring_pairs" in k:
            by_kernel[k].update({name: med(vals) for name, vals in v.items()})
summary["ring_counters_median_by_kernel"] = by_kernel
summary["derived"] = {}
for k, cm in by_kernel.items():
    if not cm.get("GRBM_GUI_ACTIVE"):
        continue
    cyc = cm["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    summary["derived"][k] = {"kernel_cycles": cyc, "note": "GRBM_GUI_ACTIVE / 8",
                             "mfma_busy_of_all_simd_cycles": cm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024.0),
                             "mfma_busy_of_the_240_working_cus": cm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 960.0),
                             "lds_array_busy_per_working_cu": cm.get("SQ_LDS_IDX_ACTIVE", 0) / 240.0 / cyc,
                             "lds_bank_conflict_cycles": cm.get("SQ_LDS_BANK_CONFLICT"),
                             "l2_hit_rate": cm.get("TCC_HIT_sum", 0) / max(1.0, cm.get("TCC_HIT_sum", 0) + cm.get("TCC_MISS_sum", 0))}
(dest / f"{tag}_bank_pmc_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
print(json.dumps(summary, indent=1))
