#!/usr/bin/env python3
"""Mask R-CNN stage, batch of 8 frames of 160x90 (the pipeline's shape): wall time per stage with syncs between stages."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import maskrcnn as M
seg = M.MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = [np.random.default_rng(i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(B)]
for _ in range(3): seg.batch(frames)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): seg.batch(frames)
torch.cuda.synchronize()
print(f"batch of {B}, no extra syncs: {(time.perf_counter() - t0) / 10 * 1e3:.1f} ms = {(time.perf_counter() - t0) / 10 / B * 1e3:.2f} ms/frame")
acc = {}
def timed(f, name):
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
        acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
    return g
for n in ('_nms_batched', '_roi_align', '_apply_deltas'):
    setattr(M, n, timed(getattr(M, n), n))
for n in ('backbone', 'fpn', 'head', 'mask', 'rpn'):
    m = getattr(seg.net, n); m.forward = timed(m.forward, n)
M.F.grid_sample = timed(M.F.grid_sample, 'grid_sample')
M._pack_levels = timed(M._pack_levels, '_pack_levels')
for _ in range(3): seg.batch(frames)
acc.clear()
N = 10
t0 = time.perf_counter()
for _ in range(N): seg.batch(frames)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N
print(f"batch of {B}: {tot*1e3:.1f} ms = {tot/B*1e3:.2f} ms/frame (with per-stage syncs)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:18s} {v/N*1e3:7.2f} ms/batch")
print(f"  {'other':18s} {(tot - sum(acc.values())/N)*1e3:7.2f} ms/batch")
