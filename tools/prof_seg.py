#!/usr/bin/env python3
"""torch.profiler over MaskRCNNSegmenter.batch (8 frames of 160x90): where host and device time go."""
import os, sys
import numpy as np, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import maskrcnn as M
seg = M.MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
frames = [np.random.default_rng(i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(8)]
for _ in range(3): seg.batch(frames)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): seg.batch(frames)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=18, max_name_column_width=60))
