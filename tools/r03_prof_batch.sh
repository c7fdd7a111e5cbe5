#!/bin/bash
# host/device split of the lockstep path + rocprofv3 kernel stats of the same; gpurun_out/r03/prof_batch*.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/prof_batch.txt
for cfg in "256 64" "1024 512" "4096 1024" "256 64 1 640_480_color" "1024 512 1 640_480_color" "2328 582 1 640_480_color" ; do python $ROOT/tools/prof_batch.py $cfg 2>&1 | tail -n 1 >> $OUT/prof_batch.txt; done
cat $OUT/prof_batch.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b_def -o def -- python3 $ROOT/tools/prof_batch.py 2048 1024 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b_640 -o r640 -- python3 $ROOT/tools/prof_batch.py 1164 582 1 640_480_color > /dev/null 2>&1
for f in $(find $OUT/prof_b_def $OUT/prof_b_640 -name "*kernel_stats.csv"); do echo "== $f"; head -n 18 $f | cut -c1-220; done > $OUT/prof_batch_kernels.txt
find $OUT/prof_b_def $OUT/prof_b_640 -name "*trace.csv" -delete
cat $OUT/prof_batch_kernels.txt
