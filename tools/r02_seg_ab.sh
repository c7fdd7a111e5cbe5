#!/bin/bash
# segmentation stage on one box: box steps as tensor operations (ROPE_SEG_HIP=0) against the HIP kernels of rope_seg.hip
for rep in 1 2; do for f in 0 1; do
  echo "== ROPE_SEG_HIP=$f"
  ROPE_SEG_HIP=$f timeout -k 10 300 python tools/time_seg_batch.py 8 2>&1 | grep -E "batch of|nms|roi_align|other" || exit 1
done; done
