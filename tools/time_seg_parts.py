#!/usr/bin/env python3
"""Segmentation stage of a batch of eight in parts: upload, moulding, trunk replay, everything after the trunk, results.
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd.maskrcnn import MaskRCNNSegmenter
seg = MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
net = seg.net
frames = [np.random.default_rng(i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(8)]
for _ in range(3): seg.batch(frames)
def T(f, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    imgs = seg._upload(frames)
    print('upload       %.2f ms' % T(lambda: seg._upload(frames)))
    x, geo = net._mould(imgs)
    print('mould        %.2f ms' % T(lambda: net._mould(imgs)))
    print('trunk replay %.2f ms' % T(lambda: net._trunk_replayed(x)))
    feats, probs, deltas = net._trunk_replayed(x)
    feats = [f.clone() for f in feats]; probs = probs.clone(); deltas = deltas.clone()
    print('detect       %.2f ms' % T(lambda: net._detect(feats, probs, deltas, *geo)))
    print('results      %.2f ms' % T(lambda: seg._results(net._detect(feats, probs, deltas, *geo))))
    print('batch        %.2f ms' % T(lambda: seg.batch(frames)))
    # device time of detect: events
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): net._detect(feats, probs, deltas, *geo)
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
