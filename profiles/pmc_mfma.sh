#!/bin/bash
# PMC passes for the MFMA channelizer (wave-cycle anatomy).  Usage: profiles/pmc_mfma.sh <tag>
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DBG=${DBG:-0}
CMD="python3 $R/profiles/quick_mfma_check.py"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/p1" -- $CMD > "$OUT/p1.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$OUT/p2" -- $CMD > "$OUT/p2.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/p3" -- $CMD > "$OUT/p3.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
for p in ("p1", "p2", "p3"):
    fs = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counters; tail of log:"); print(open(f"{out}/{p}.log").read()[-600:]); continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_channelize_mfma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{p} {k:32s} n={len(v)} median={sorted(v)[len(v)//2]:.4g}")
PY
