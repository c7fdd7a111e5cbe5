#!/bin/bash
# MFMA utilisation of the segmentation stage (counters in their own run), summarised on the box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
SCRIPT=${1:-bench_seg.py}          # bench_seg.py: frame by frame; bench_seg_batch.py: batches of 8 (run with ROPE_SEG_GRAPH=0 for per-kernel counters)
OUT=/tmp/prof_seg
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 $ROOT/tools/$SCRIPT 5 > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/$SCRIPT 5 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
python3 - <<'PY' > $ROOT/gpurun_out/seg_profile.txt
import csv, glob, collections
agg = collections.defaultdict(float)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(glob.glob('/tmp/prof_seg/pmc/*/*_counter_collection.csv')[0])):
    agg[r['Counter_Name']] += float(r['Counter_Value'])
    per[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
print('totals over the run:', {k: f'{v:.4g}' for k, v in agg.items()})
print('MFMA busy cycles / SQ busy cycles (whole run): %.4f' % (agg['SQ_VALU_MFMA_BUSY_CYCLES'] / agg['SQ_BUSY_CYCLES']))
print('top kernels by MFMA busy cycles (MFMA busy / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) = per-SIMD matrix-pipe utilisation):')
for k, v in sorted(per.items(), key=lambda kv: -kv[1]['SQ_VALU_MFMA_BUSY_CYCLES'])[:8]:
    g = v['GRBM_GUI_ACTIVE'] / 8 * 1024
    print('  %-70s mfma %.3g  util %.3f' % (k, v['SQ_VALU_MFMA_BUSY_CYCLES'], v['SQ_VALU_MFMA_BUSY_CYCLES'] / g if g else 0))
print('kernel time, top 10:')
for r in list(csv.reader(open(glob.glob('/tmp/prof_seg/stats/*/*_kernel_stats.csv')[0])))[:11]:
    print('  ', r[0][:80].ljust(80), r[1:5])
PY
cat $ROOT/gpurun_out/seg_profile.txt
