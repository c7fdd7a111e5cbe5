#!/bin/bash
# gpurun_out/ (scratch) -> profiles/ (tracked): run HERE after tools/r03_final.sh came back, on the tree that was profiled
cd /root/repo
python tools/summarize_prof.py gpurun_out/prof_r03 r03 > /dev/null
O=gpurun_out/r03
cp $O/BENCH_r03.json $O/BENCH_r03_fullloss.json $O/BENCH_r03_cfg5.json $O/r03_valu_issue.json $O/r03_kernel_clock.json profiles/
cp $O/frames_batch.txt profiles/r03_frames_batch.txt
cp $O/prof_batch.txt profiles/r03_batch_split.txt
cp $O/prof_batch_kernels.txt profiles/r03_batch_kernels.txt
ls -la profiles | grep r03
