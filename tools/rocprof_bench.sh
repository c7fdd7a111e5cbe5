#!/bin/bash
# Profiles the default bench workload on the GPU box: kernel trace + stats, then PMC passes
# (counters in their own runs, as the pool requires).  Output under gpurun_out/prof_<tag>/.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unshared --no-end-to-end $BENCH_ARGS"      # BENCH_ARGS: e.g. "--loss full" or "--workload cfg5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/stats.log" 2>&1 || { tail -20 "$OUT/stats.log"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 $ARGS > "$OUT/pmc_sq.log" 2>&1 || { tail -20 "$OUT/pmc_sq.log"; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- python3 $ARGS > "$OUT/pmc_sq2.log" 2>&1 || { tail -20 "$OUT/pmc_sq2.log"; exit 1; }
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_lanes" -- python3 $ARGS > "$OUT/pmc_lanes.log" 2>&1 || { tail -20 "$OUT/pmc_lanes.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { tail -20 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1 || { tail -20 "$OUT/pmc_write.log"; exit 1; }
find "$OUT" -name '*.csv' | head -30
