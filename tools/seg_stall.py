#!/usr/bin/env python3
"""Per-batch wall time of the segmentation stage with the allocator's and the garbage collector's counters beside it:
how the every-other-batch 80 ms stall was told apart from allocation, collection, graph replay and pinned memory (NO_PIN=1,
ROPE_SEG_GRAPH=0, OMP_NUM_THREADS=4 are the switches that were tried; profiles/r02_seg_boxes.txt section 2).
"""
import os, sys, time, gc
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd.maskrcnn import MaskRCNNSegmenter
if os.environ.get('NO_PIN'):
    _empty = torch.empty
    def empty_nopin(*a, **k):
        k.pop('pin_memory', None); return _empty(*a, **k)
    torch.empty = empty_nopin
seg = MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
frames = [np.random.default_rng(i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(8)]
for _ in range(4): seg.batch(frames)
gct = [0.0]
def cb(phase, info):
    if phase == 'start': cb.t = time.perf_counter()
    else: gct[0] += time.perf_counter() - cb.t
gc.callbacks.append(cb)
def stats():
    s = torch.cuda.memory_stats()
    return s['num_device_alloc'], s['num_device_free'], s['num_alloc_retries'], s['reserved_bytes.all.current'] >> 20
import cProfile, pstats
for i in range(10):
    a = stats(); g0 = gct[0]
    pr = cProfile.Profile(); pr.enable()
    t = time.perf_counter(); seg.batch(frames); dt = time.perf_counter() - t
    pr.disable()
    b = stats()
    top = sorted(pstats.Stats(pr).stats.items(), key=lambda kv: -kv[1][2])[:3]
    print(f"batch {i}: {dt*1e3:6.1f} ms  device allocs +{b[0]-a[0]} frees +{b[1]-a[1]} retries +{b[2]-a[2]} reserved {b[3]} MiB  gc {1e3*(gct[0]-g0):.1f} ms  top: " +
          "; ".join(f"{k[2][:28]} {v[2]*1e3:.1f}" for k, v in top))
