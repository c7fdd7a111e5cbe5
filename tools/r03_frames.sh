#!/bin/bash
# frames/s of ONE Predictor (one engine context) through run_many, by lockstep batch size; gpurun_out/r03/frames_batch.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03/frames_batch.txt
mkdir -p $ROOT/gpurun_out/r03
: > $OUT
row() { echo "$1: $(env $2 python $ROOT/tools/bench_frames.py $3 2>&1 | tail -n 2 | head -n 1)" >> $OUT; }
# enough frames for several batches of the largest size: a batch's preparation and upload hide behind the batch before it
for b in 1 16 64 256 512 1024; do row "defaults (1280x720 / 8, 25^3 grid), run_many, batch $b" "ROPE_PREFETCH=1 ROPE_BATCH=$b" "$((b == 1 ? 1024 : 8192))"; done
for b in 1 16 64 256 512 582; do row "640x480 / 1, 9^3 grid, run_many, batch $b" "ROPE_PREFETCH=1 ROPE_BATCH=$b" "$((b == 1 ? 512 : 4096)) 1 640_480_color"; done
cat $OUT
