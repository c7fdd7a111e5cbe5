#!/usr/bin/env python3
"""Per-frame latency of one Predictor (median, p90, p99, max, frames over 10 ms): shows scheduler freezes if there are any.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import SyntheticPredictor
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '1280_720_color', 8, 'SLU', noise=False, seed=1)
p = sp.predictor
lim = sp.urdf_reader.joint_limits
poses = [np.random.default_rng(7919 + f).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0]) for f in range(300)]
sp.run(poses[0])
frames = []
for q in poses:
    sp.renderer.setJointAngles(q); frames.append(sp.renderer.render())
lat = []
t00 = time.perf_counter()
for f in frames:
    t = time.perf_counter(); p.run(*f); lat.append(time.perf_counter() - t)
tot = time.perf_counter() - t00
lat = np.array(lat) * 1e3
print(f"{len(lat)} frames in {tot:.3f} s = {len(lat)/tot:.1f} fps; latency ms: median {np.median(lat):.2f} p90 {np.percentile(lat, 90):.2f} p99 {np.percentile(lat, 99):.2f} max {lat.max():.2f}; frames over 10 ms: {(lat > 10).sum()}, their sum {lat[lat > 10].sum():.0f} ms")
print('threads:', len(os.listdir('/proc/self/task')), 'torch imported:', 'torch' in sys.modules)
