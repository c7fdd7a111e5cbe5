#!/bin/bash
# kernel-level timing of one small batch size: tools/rocprof_small.sh <preset> <ds> <C>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_small_$3_$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/tools/bench_small.py $1 $2 $3 > "$OUT/run.log" 2>&1 || { tail -20 "$OUT/run.log"; exit 1; }
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
