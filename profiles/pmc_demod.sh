#!/bin/bash
# PMC anatomy of the fused demodulator kernels (profiles/micro_demod.py).  Usage: profiles/pmc_demod.sh <tag>
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcd_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/profiles/micro_demod.py"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$OUT/p1" -- $CMD > "$OUT/p1.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/p2" -- $CMD > "$OUT/p2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
for p in ("p1", "p2"):
    fs = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counters"); print(open(f"{out}/{p}.log").read()[-400:]); continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        for k in ("k_fused_apply<0", "k_fused_reduce<0", "k_resample"):
            if k in r["Kernel_Name"]:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{p} {k:18s} {c:24s} n={len(v)} median={sorted(v)[len(v)//2]:.4g}")
PY
