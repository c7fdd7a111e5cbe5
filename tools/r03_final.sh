#!/bin/bash
# Everything profiles/ needs of one build, on one box: the bench line, rocprofv3 stats + PMC passes, the micro-benchmark,
# the in-kernel clock (profiling build), the frames table and the lockstep batch's kernel stats.  Run through gpurun; then
#   python tools/summarize_prof.py gpurun_out/prof_r03 r03 && cp gpurun_out/r03/{...} profiles/   (tools/r03_collect.sh)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd $ROOT
if [ "$1" == "frames" ]; then      # second call (the two together exceed one gpurun call): the frame pipeline's tables
    timeout -k 10 650 bash tools/r03_frames.sh > /dev/null 2>&1
    timeout -k 10 450 bash tools/r03_prof_batch.sh > /dev/null 2>&1
    cat $OUT/frames_batch.txt $OUT/prof_batch.txt
    exit 0
fi
python bench.py --steps 20 --warmup 3 > $OUT/BENCH_r03.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python bench.py --steps 10 --warmup 2 --loss full --no-cpu-baseline --no-end-to-end > $OUT/BENCH_r03_fullloss.json 2>> $OUT/bench.err
python bench.py --steps 5 --warmup 1 --workload cfg5 --no-cpu-baseline > $OUT/BENCH_r03_cfg5.json 2>> $OUT/bench.err
timeout -k 10 300 tools/valu_issue_bench.bin > $OUT/r03_valu_issue.json 2> $OUT/valu.err
ROPE_HIP_LIB=$ROOT/rope_s3d_amd/csrc/librope_hip_profile.so timeout -k 10 200 python tools/kernel_clock.py > $OUT/r03_kernel_clock.json 2> $OUT/clock.err
timeout -k 10 600 bash tools/rocprof_bench.sh r03 > $OUT/rocprof_r03.log 2>&1 || { tail -5 $OUT/rocprof_r03.log; exit 1; }
python -c "
import json
d=json.load(open('$OUT/BENCH_r03.json')); r=d['roofline']
print('value %.0f unshared %.0f score %.3f layer %.3f frac %.3f stale %s' % (d['value'], d['unshared_value'], r['score_launch_ms'], r['layer_launch_ms'], r['frac'], r['pmc_stale']))
print('end_to_end', d.get('end_to_end'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_errors_vs_port'])
"
