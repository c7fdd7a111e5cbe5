#!/usr/bin/env python3
"""Concurrency soak of PredictorPool: k Predictors / threads over n frames, every frame's angles compared with one Predictor's.

    python tools/stress_pool.py [k] [n_frames] [rounds]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import SyntheticPredictor
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
from rope_s3d_amd.prediction.pool import PredictorPool

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '1280_720_color', 8, 'SLU', noise=False, seed=1)
lim = sp.urdf_reader.joint_limits
frames = []
for f in range(n):
    sp.renderer.setJointAngles(np.random.default_rng(40000 + f).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0]))
    frames.append(sp.renderer.render())
colors, depths = [c for c, _ in frames], [d for _, d in frames]
want = np.array([sp.predictor.run(c, d) for c, d in frames])
pool = PredictorPool(k, DEFAULT_CAMERA_POSE, 8, base_intrin='1280_720_color', color_dict=sp.predictor.color_dict)
for r in range(rounds):
    t0 = time.perf_counter()
    got = pool.run_many(colors, depths)
    dt = time.perf_counter() - t0
    bad = int((got != want).any(1).sum())
    print(f"round {r}: {k} predictors, {n} frames in {dt:.2f} s = {n / dt:.0f} frames/s, frames differing from the single Predictor: {bad}", flush=True)
    assert bad == 0
print("ok")
