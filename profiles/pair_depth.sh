#!/bin/bash
# Scratch: time and HBM read traffic of the lane-pair launch (config 3's shape) for experiment builds of the library.
# Usage: profiles/pair_depth.sh <lib.so>...   (default build first)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pair_depth
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in default "$@"; do
  tag=$(basename "$lib" .so)
  if [ "$lib" = default ]; then unset IQA_LIB; else export IQA_LIB="$R/$lib"; fi
  K=10 WARM=4 timeout -k 10 200 python3 $R/profiles/bench_bank.py ${WHICH:-c3} ${MODE:-bank} 2>/dev/null | cut -c1-60 | sed "s/^/$tag: /"
  K=3 WARM=1 timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$tag" -- python3 $R/profiles/bench_bank.py ${WHICH:-c3} ${MODE:-bank} > "$OUT/$tag.log" 2>&1
  python3 - "$OUT/$tag" "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "ring_pairs" in r["Kernel_Name"] or "ring_multi" in r["Kernel_Name"]] if f else []
if v:
    print(f"{sys.argv[2]}: FETCH_SIZE median {sorted(v)[len(v)//2]:.0f} KiB -> x2 = {2 * 1024 * sorted(v)[len(v)//2] / 1e9:.2f} GB per launch (capture 4.80 GB)")
PY
done
