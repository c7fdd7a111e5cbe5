#!/bin/bash
# small-batch split tuning on one box: busy workgroups aimed at per launch / fewest / most per (tile, candidate); ARGS = bench_small.py arguments
ARGS=${ARGS:-"1280_720_color 8 2,6,18,26,64,256"}
for cfg in ${CFGS:-"512,8,64 512,4,64 512,12,64 512,8,64"}; do
  IFS=, read t m c <<< "$cfg"
  echo "== target $t min $m cap $c"
  ROPE_SPLIT_TARGET=$t ROPE_SPLIT_MIN=$m ROPE_SPLIT_CAP=$c timeout -k 10 120 python tools/bench_small.py $ARGS | cut -c1-100 || exit 1
done
