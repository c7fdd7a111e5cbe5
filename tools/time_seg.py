#!/usr/bin/env python3
"""Where a Mask R-CNN frame goes: wall time of the stages of MaskRCNN.detect (random weights, worst case 100 detections)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import maskrcnn as M
seg = M.MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
acc = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
        acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
    setattr(mod, name, g)
for n in ('_nms', '_roi_align', '_apply_deltas', '_pyramid_anchors'):
    wrap(M, n)
for n in ('backbone', 'fpn', 'head', 'mask'):
    m = getattr(seg.net, n); f = m.forward
    def mk(f, n):
        def g(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
            acc[n] = acc.get(n, 0) + time.perf_counter() - t; return r
        return g
    m.forward = mk(f, n)
img = np.random.default_rng(0).integers(0, 255, (480, 640, 3), dtype=np.uint8)
for _ in range(3): seg(img)
acc.clear()
N = 10
t0 = time.perf_counter()
for _ in range(N): seg(img)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N
print(f"total {tot*1e3:.1f} ms/frame (with per-stage syncs)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:18s} {v/N*1e3:7.2f} ms")
print(f"  {'other':18s} {(tot - sum(acc.values())/N)*1e3:7.2f} ms")
