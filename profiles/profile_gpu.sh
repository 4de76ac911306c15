#!/bin/bash
# Run on the GPU box (inside gpurun): rocprofv3 kernel stats + two separate PMC passes of bench.py.
# Usage: profiles/profile_gpu.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write}/...
# PMC passes are collected on their own (no sys/hip/hsa tracing), as the pool requires.
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-extras"   # the defaults (100 warm-up + 1000 timed captures): the sustained, power-limited state of the judged run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch_bench.json" 2> "$OUT/fetch.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write_bench.json" 2> "$OUT/write.err" || exit 1
python3 "$R/profiles/pmc_summary.py" "$OUT" "$TAG"
