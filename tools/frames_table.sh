#!/bin/bash
# End-to-end frames/s of Predictor.run / run_many / PredictorPool at the reference's default settings and at 640x480,
# one line per configuration, into gpurun_out/frames_table.txt (copied to profiles/ by hand).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/frames_table.txt
: > $OUT
row() { echo "$1: $(env $2 python $ROOT/tools/bench_frames.py $3 2>&1 | tail -n 2 | head -n 1)" >> $OUT; }
row "defaults (1280x720 / 8, 25^3 grid), one Predictor, Python stage loop" "ROPE_NATIVE=0" "300"
row "defaults, one Predictor (rope_predict)" "ROPE_X=0" "300"
row "defaults, one Predictor, run_many (next frame prepared meanwhile)" "ROPE_PREFETCH=1" "300"
for k in 2 4 8 12; do row "defaults, PredictorPool of $k" "ROPE_POOL=$k" "600"; done
row "640x480 / 1, 9^3 grid, one Predictor" "ROPE_X=0" "200 1 640_480_color"
row "640x480 / 1, run_many" "ROPE_PREFETCH=1" "200 1 640_480_color"
for k in 4 8; do row "640x480 / 1, PredictorPool of $k" "ROPE_POOL=$k" "320 1 640_480_color"; done
cat $OUT
