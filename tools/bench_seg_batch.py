#!/usr/bin/env python3
"""The segmentation stage as the pipeline runs it: batches of 8 frames of 160x90 through MaskRCNNSegmenter.batch."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd.maskrcnn import MaskRCNNSegmenter

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if os.environ.get('ROPE_SEG_FIND') == '1':                  # MIOpen's find step (timed trials per convolution shape) instead of its immediate-mode pick
    torch.backends.cudnn.benchmark = True
t_start = time.perf_counter()
seg = MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
frames = [np.random.default_rng(i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(8)]
for _ in range(3):
    seg.batch(frames)
torch.cuda.synchronize()
print(f"construction + three warm-up batches: {time.perf_counter() - t_start:.1f} s")
t0 = time.perf_counter()
for _ in range(n):
    seg.batch(frames)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"batch of 8 frames 160x90: {dt * 1e3:.1f} ms = {dt / 8 * 1e3:.2f} ms/frame = {8 / dt:.1f} frames/s")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in seg.batches([frames] * n):                      # the next group's trunk on a second stream beside this group's box steps
    pass
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"the same through batches(): {dt * 1e3:.1f} ms = {dt / 8 * 1e3:.2f} ms/frame = {8 / dt:.1f} frames/s")
