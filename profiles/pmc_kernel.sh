#!/bin/bash
# PMC anatomy of one kernel under a micro script.  Usage: profiles/pmc_kernel.sh <tag> <script.py> <kernel-name-substring>
# (counter passes only -- no tracing domains beside them, as the pool requires)
set -u
TAG=${1:-x}; SCRIPT=${2:-profiles/micro_resample.py}; KERN=${3:-k_resample}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmck_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/$SCRIPT"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$OUT/p1" -- $CMD > "$OUT/p1.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/p2" -- $CMD > "$OUT/p2.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/p3" -- $CMD > "$OUT/p3.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d "$OUT/p4" -- $CMD > "$OUT/p4.log" 2>&1 || exit 1
python3 - "$OUT" "$KERN" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys
from collections import defaultdict
out, kern = sys.argv[1], sys.argv[2]
for p in ("p1", "p2", "p3", "p4"):
    fs = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counters"); print(open(f"{out}/{p}.log").read()[-400:]); continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if kern in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{p} {k:48s} {c:26s} n={len(v)} median={sorted(v)[len(v)//2]:.5g}")
PY
