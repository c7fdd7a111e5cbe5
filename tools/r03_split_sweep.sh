#!/bin/bash
# Lockstep batches: are the small-batch split thresholds (tuned on single frames in round 2) right for rows of many frames?
# frames/s through run_many by launch-structure setting; gpurun_out/r03/split_sweep.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03/split_sweep.txt
mkdir -p $ROOT/gpurun_out/r03
: > $OUT
row() { echo "$1: $(env $2 python $ROOT/tools/bench_frames.py $3 2>&1 | tail -n 2 | head -n 1 | sed 's/, [0-9]* candidate.*//')" >> $OUT; }
for b in 64 256; do
  for s in "ROPE_X=0" "ROPE_STRATEGY=2" "ROPE_SPLIT_MIN_MANY=2" "ROPE_SPLIT_MIN_MANY=4" "ROPE_SPLIT_MIN=4" "ROPE_SPLIT_TARGET=1024" "ROPE_SPLIT_TARGET=256" "ROPE_GEO_ROWS=256"; do
    row "defaults batch $b [$s]" "ROPE_PREFETCH=1 ROPE_BATCH=$b $s" "1024"
    row "640x480 batch $b [$s]" "ROPE_PREFETCH=1 ROPE_BATCH=$b $s" "512 1 640_480_color"
  done
done
cat $OUT
