#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_lanes
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/prof_lanes -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unshared > /tmp/prof_lanes.log 2>&1 || { tail -5 /tmp/prof_lanes.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob('/tmp/prof_lanes/*/*_counter_collection.csv')[0])):
    k = 'SCORE' if ('raster_queue_kernel<0, 0' in r['Kernel_Name'] or 'raster_score_kernel<0, 0' in r['Kernel_Name']) else ('LAYER' if ('raster_score_kernel<0, 3' in r['Kernel_Name'] or 'raster_queue_kernel<0, 3' in r['Kernel_Name']) else None)
    if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, {c: f'{v:.4g}' for c, v in m.items()})
    if 'SQ_THREAD_CYCLES_VALU' in m and 'SQ_ACTIVE_INST_VALU' in m:
        print('   average active lanes per VALU cycle: %.1f of 64' % (m['SQ_THREAD_CYCLES_VALU'] / m['SQ_ACTIVE_INST_VALU'] / 4 * 4 / 1 if False else m['SQ_THREAD_CYCLES_VALU'] / m['SQ_ACTIVE_INST_VALU']))
PY
