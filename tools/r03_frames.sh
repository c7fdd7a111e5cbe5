#!/bin/bash
# frames/s of ONE Predictor (one engine context) through run_many, by lockstep batch size; gpurun_out/r03/frames_batch.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03/frames_batch.txt
mkdir -p $ROOT/gpurun_out/r03
: > $OUT
row() { echo "$1: $(env $2 python $ROOT/tools/bench_frames.py $3 2>&1 | tail -n 2 | head -n 1)" >> $OUT; }
for b in 1 16 64 256; do row "defaults (1280x720 / 8, 25^3 grid), run_many, batch $b" "ROPE_PREFETCH=1 ROPE_BATCH=$b" "1024"; done
for b in 1 16 64 256; do row "640x480 / 1, 9^3 grid, run_many, batch $b" "ROPE_PREFETCH=1 ROPE_BATCH=$b" "512 1 640_480_color"; done
cat $OUT
