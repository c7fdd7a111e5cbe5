#!/bin/bash
# round-2 baseline: phase ablation, bench, and meshlet-size variants of the unchanged kernel
set -o pipefail
OUT=gpurun_out/r02_base; mkdir -p $OUT
python tools/profile_phases.py > $OUT/phases.log 2>&1 && \
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err && \
for v in "128 64" "64 64" "96 64" "64 48" "48 40"; do set -- $v; echo "== tris $1 verts $2"; ROPE_MESHLET_TRIS=$1 ROPE_MESHLET_VERTS=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    r = d['roofline']; print('poses/s %.0f  score %.3f layer %.3f ms' % (d['value'], r['score_launch_ms'], r['layer_launch_ms']))
"; done > $OUT/meshlet_variants.log 2>&1
cat $OUT/phases.log $OUT/meshlet_variants.log
