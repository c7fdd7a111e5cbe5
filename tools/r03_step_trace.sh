#!/bin/bash
# The steps of a lockstep batch, from a kernel trace: one finalize_frames_kernel launch closes a step; its grid tells the step's
# rows (in 256s), the time since the step before tells what the step cost.  gpurun_out/r03/step_trace.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 512" "1024 512 1 640_480_color"; do
  rm -rf /tmp/step_trace
  rocprofv3 --kernel-trace --output-format csv -d /tmp/step_trace -o t -- python3 $ROOT/tools/prof_batch.py $cfg > /tmp/step_trace.log 2>&1
  python3 - "$cfg" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob('/tmp/step_trace/**/*kernel_trace.csv', recursive=True)[0])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size']) if 'Grid_Size' in r else int(r.get('Grid_Size_X', 0))) for r in rows)
fin = [e for e in ev if 'finalize_frames_kernel' in e[2]]
# the last batch of the run: steps after the last table_score_frames_kernel launch
last_tbl = max(e[0] for e in ev if 'table_score_frames' in e[2])
steps = [e for e in fin if e[0] > last_tbl]
t_prev = max(e[1] for e in ev if 'argmin_sets' in e[2] and e[0] > last_tbl - 1) if any('argmin_sets' in e[2] for e in ev) else steps[0][0]
print(f"config {sys.argv[1]}: {len(steps)} steps after the Lookup stage of the last batch")
tot = 0
buckets = {}
for s in steps:
    rows_ = s[3] // 256 * 256 if s[3] >= 256 else s[3]
    dt = (s[1] - t_prev) / 1e3
    t_prev = s[1]
    tot += dt
    k = 'rows <= 256' if s[3] <= 256 else ('rows <= 512' if s[3] <= 512 else ('rows <= 1024' if s[3] <= 1024 else 'rows > 1024'))
    b = buckets.setdefault(k, [0, 0.0]); b[0] += 1; b[1] += dt
print(f"  total {tot / 1e3:.2f} ms")
for k, (n, t) in sorted(buckets.items()):
    print(f"  {k:14s} {n:4d} steps {t / 1e3:8.2f} ms  ({t / max(n, 1):7.1f} us per step)")
PY
done > $OUT/step_trace.txt 2>&1
cat $OUT/step_trace.txt
