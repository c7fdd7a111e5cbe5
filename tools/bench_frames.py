#!/usr/bin/env python3
"""End-to-end frames/s of the whole prediction pipeline on synthetic frames (BASELINE configs[2]-like, without the
Mask R-CNN stage: link masks come from the colour-coded render, as the reference's SyntheticPredictor does).

    python tools/bench_frames.py [n_frames] [ds_factor] [base_intrin] [lookup_divisions]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import SyntheticPredictor
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
from rope_s3d_amd.prediction.analysis import joint_error_stats

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
intr = sys.argv[3] if len(sys.argv) > 3 else '1280_720_color'
div = int(sys.argv[4]) if len(sys.argv) > 4 else None
sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, intr, ds, 'SLU', noise=False, seed=1, lookup_divisions=div)
p = sp.predictor
p.NATIVE = os.environ.get('ROPE_NATIVE', '1') != '0'       # 0: the Python stage loop instead of rope_predict
if 'ROPE_SPECULATE' in os.environ:
    p.SPECULATE = int(os.environ['ROPE_SPECULATE'])        # Descent joints evaluated as one batch (1 = the reference's two renders at a time)
    p.SPECULATE_BATCH = p.SPECULATE
if 'ROPE_BATCH' in os.environ:
    p.BATCH = int(os.environ['ROPE_BATCH'])                # run_many: frames in lockstep per device batch (1: frame after frame; default: by plane size)
print(f"stage loop: {'rope_predict (C++)' if p.NATIVE else 'Python'}")
print(f"render {p.intrinsics.width}x{p.intrinsics.height}, lookup grid {len(p.lookup_angles)} poses, crop {list(p.lookup_crop)}")
lim = sp.urdf_reader.joint_limits
poses = [np.random.default_rng(7919 + f).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0]) for f in range(n)]
sp.run(poses[0])                                   # warm-up
# frames first (the camera's job in the reference), then the timed prediction of each: Predictor.run covers
# down-sampling, target preparation and upload, and every stage with its device batches
frames = []
for q in poses:
    sp.renderer.setJointAngles(q)
    frames.append(sp.renderer.render())
p.evaluations = 0
res = np.zeros((2, n, 6))
t0 = time.perf_counter()
if int(os.environ.get('ROPE_POOL', '0')) > 1:                # PredictorPool: k Predictors / threads on this GPU
    from rope_s3d_amd.prediction.pool import PredictorPool
    pool = PredictorPool(int(os.environ['ROPE_POOL']), DEFAULT_CAMERA_POSE, ds, base_intrin=intr, color_dict=p.color_dict, lookup_divisions=div)
    pool.run_many([frames[0][0]] * len(pool), [frames[0][1]] * len(pool))
    for q_ in pool.predictors:
        q_.evaluations = 0
    t0 = time.perf_counter()
    res[0] = poses
    res[1] = pool.run_many([c for c, _ in frames], [d for _, d in frames])
    p = pool
elif os.environ.get('ROPE_PREFETCH', '0') != '0':             # Predictor.run_many: BATCH frames in lockstep (BATCH 1: frame i+1 prepared while frame i is on the GPU)
    res[0] = poses
    p.run_many([c for c, _ in frames[:2 * (p.BATCH or 256)]], [d for _, d in frames[:2 * (p.BATCH or 256)]])      # warm-up: buffers of a batch's size
    p.evaluations = 0
    t0 = time.perf_counter()
    res[1] = p.run_many([c for c, _ in frames], [d for _, d in frames])
else:
    for f in range(n):
        res[0, f], res[1, f] = poses[f], p.run(*frames[f])
dt = time.perf_counter() - t0
st = joint_error_stats(res[1], res[0])
print(f"{n} frames in {dt:.2f} s = {n / dt:.1f} frames/s, {p.evaluations / n:.0f} candidate evaluations/frame, "
      f"{p.evaluations / dt:.0f} poses/s end to end (Predictor.run on frames already in host memory)")
print("joint-angle |error| mean (rad) S,L,U:", np.round(st['mean'][:3], 5), " p95:", np.round(st['p95'][:3], 5), " max:", np.round(st['max'][:3], 5))
