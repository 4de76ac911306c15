#!/bin/bash
# Power / clock samples (rocm-smi, read-only) while bench.py runs.  Usage: profiles/power_trace.sh <tag> [bench args]
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/power_$TAG.txt
( for i in $(seq 1 400); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|junction|memory" | tr '\n' '|' ; echo; sleep 0.05; done ) > "$OUT" &
SAMPLER=$!
python3 "$R/bench.py" --no-cpu-baseline "$@" > "$R/gpurun_out/power_${TAG}_bench.json"
kill $SAMPLER 2>/dev/null
wait $SAMPLER 2>/dev/null
cat "$R/gpurun_out/power_${TAG}_bench.json"
