#!/bin/bash
# A/B on ONE box (devices differ by several per cent): bench line of every librope_hip_var_*.so and of the product library, twice
for rep in 1 2; do
for f in rope_s3d_amd/csrc/librope_hip_var_*.so rope_s3d_amd/csrc/librope_hip.so; do
  ROPE_HIP_LIB=$PWD/$f timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    r = d['roofline']; print('%-28s poses/s %.0f  unshared %.0f  score %.3f layer %.3f ms' % ('$f'.split('/')[-1], d['value'], d.get('unshared_value', 0), r['score_launch_ms'], r['layer_launch_ms']))
"; done; done
