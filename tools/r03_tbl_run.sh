#!/bin/bash
# table_score_frames_kernel: frames per lane group (ROPE_TABLE_FRAMES = 2 / 4 / 8) and lanes per group (ROPE_TABLE_LANES = 64 / 16), kernel time from rocprofv3 --stats on the lockstep tool
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -x > gpurun_out/r03/tbl_tests.log 2>&1; tail -2 gpurun_out/r03/tbl_tests.log
cd /tmp && export TMPDIR=/tmp
for cfg in "2 64" "2 16" "2 8" "4 16"; do
  set -- $cfg
  f=$1
  export ROPE_TABLE_FRAMES=$1 ROPE_TABLE_LANES=$2
  rm -rf $R/gpurun_out/r03/prof_tbl
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/prof_tbl -o t -- python3 $R/tools/prof_batch.py 512 256 > /dev/null 2>&1
  find $R/gpurun_out/r03/prof_tbl -name "*trace.csv" -delete
  echo "F=$f lanes=$ROPE_TABLE_LANES $(grep "table_score_frames" $(find $R/gpurun_out/r03/prof_tbl -name "*kernel_stats.csv") | cut -c200-330)"
  python $R/tools/prof_batch.py 1024 512 2>&1 | tail -1
done
