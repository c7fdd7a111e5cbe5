#!/bin/bash
# Kernel trace + stats only (no counter passes) of a bench variant: tools/rocprof_stats_only.sh <tag> <bench args...>
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unshared --no-end-to-end "$@" > "$OUT/stats.log" 2>&1 || { tail -20 "$OUT/stats.log"; exit 1; }
find "$OUT" -name '*kernel_stats.csv' | head -3
