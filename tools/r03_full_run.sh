#!/bin/bash
# the full `_error` loss after a change of its loss pass: parity, then the scoring launch of the grid and the lockstep path
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden.py tests/test_gpu_batch.py tests/test_gpu_predictor.py -q -m gpu -x > gpurun_out/r03/full_tests.log 2>&1; tail -2 gpurun_out/r03/full_tests.log
python bench.py --steps 10 --warmup 2 --loss full --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full loss', round(d['value']), round(d['unshared_value']), d['roofline']['score_launch_ms'], d['roofline']['layer_launch_ms'])"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth loss', round(d['value']), round(d['unshared_value']), d['roofline']['score_launch_ms'], d['roofline']['layer_launch_ms'])"
cd /tmp
python $R/tools/prof_batch.py 1024 512 1 640_480_color 2>&1 | tail -1
python $R/tools/prof_batch.py 1024 512 2>&1 | tail -1
