#!/bin/bash
# Kernel trace of the segmentation stage in batches of 8: per batch, device-busy time, kernel count and the largest idle gaps.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/prof_seg_trace
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/bench_seg_batch.py 6 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
grep -E "batch of|same through" $OUT/run.log
python3 - <<'PY' > $ROOT/gpurun_out/seg_trace.txt
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/prof_seg_trace/*/*_kernel_trace.csv')[0])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# one steady-state batch = from the proposal NMS of one batch to the proposal NMS of the next (nms_mask_kernel runs twice per batch)
marks = [i for i, e in enumerate(ev) if 'nms_mask_kernel' in e[2]][0::2]
print('proposal-NMS to proposal-NMS intervals (ms):', [round((ev[b][0] - ev[a][0]) / 1e6, 1) for a, b in zip(marks, marks[1:])])
iv = [(ev[b][0] - ev[a][0], a, b) for a, b in zip(marks, marks[1:])][1:]
for which, (a, b) in (('batch() loop, a late period', (marks[7], marks[8])), ('the longest period', max(iv)[1:]), ('batches() loop, a late period', (marks[-2], marks[-1]))):
    w = ev[a:b]
    span = w[-1][1] - w[0][0]
    busy, cur_s, cur_e, gaps = 0, w[0][0], w[0][1], []
    for s_, e_, n in w[1:]:
        if s_ > cur_e:
            busy += cur_e - cur_s; gaps.append((s_ - cur_e, n)); cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    busy += cur_e - cur_s
    print(f"{which}: {len(w)} kernels over {span / 1e6:.2f} ms, device busy {busy / 1e6:.2f} ms ({busy / span:.2f}); {len(gaps)} idle gaps, {sum(g for g, _ in gaps) / 1e6:.2f} ms")
    by = collections.defaultdict(lambda: [0, 0.0])
    for s_, e_, n in w:
        by[n[:70]][0] += 1; by[n[:70]][1] += (e_ - s_) / 1e6
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    {t:7.3f} ms {c:4d} x  {n}")
    gaps.sort(reverse=True)
    longest = sorted(((e_ - s_) / 1e6, n[:60]) for s_, e_, n in w)[-3:]
    print("  longest kernels (ms):", longest)
    print("  largest idle gaps (ms, kernel that ended the gap): " + "; ".join(f"{g / 1e6:.3f} {n[:40]}" for g, n in gaps[:10]))
    hist = collections.Counter(min(int(g / 1e3) // 20 * 20, 200) for g, _ in gaps)
    print("  gaps by length (us: count): " + ", ".join(f"{k}+: {hist[k]}" for k in sorted(hist)))
PY
cat $ROOT/gpurun_out/seg_trace.txt
