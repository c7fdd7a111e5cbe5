#!/bin/bash
# one kernel iteration on the GPU box: parity tests of the raster path, then the bench line (and optionally the phase ablation)
set -o pipefail
TAG=${1:-iter}; OUT=gpurun_out/r02_$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden.py tests/test_golden_camera.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?
tail -4 $OUT/tests.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $OUT/tests.log | head -20; exit $rc; }
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python - <<PY
import json
d = json.load(open('$OUT/bench.json')); r = d['roofline']
print('poses/s %.0f  unshared %.0f  score %.3f layer %.3f ms' % (d['value'], d['unshared_value'], r['score_launch_ms'], r['layer_launch_ms']))
PY
if [ -f rope_s3d_amd/csrc/librope_hip_profile.so ] && [ "$2" = "phases" ]; then
  ROPE_HIP_LIB=$PWD/rope_s3d_amd/csrc/librope_hip_profile.so timeout -k 10 300 python tools/profile_phases.py > $OUT/phases.log 2>&1; cat $OUT/phases.log
fi
