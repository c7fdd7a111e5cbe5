#!/usr/bin/env python3
"""Where a lockstep batch of frames spends its time: host preparation, target upload, the stage machine (rope_predict_batch).

    python tools/prof_batch.py [n_frames] [batch] [ds_factor] [base_intrin] [lookup_divisions]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import SyntheticPredictor
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
intr = sys.argv[4] if len(sys.argv) > 4 else '1280_720_color'
div = int(sys.argv[5]) if len(sys.argv) > 5 else None
sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, intr, ds, 'SLU', noise=False, seed=1, lookup_divisions=div)
p = sp.predictor
lim = sp.urdf_reader.joint_limits
frames = []
for f in range(n):
    sp.renderer.setJointAngles(np.random.default_rng(7919 + f).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0]))
    frames.append(sp.renderer.render())
t = time.perf_counter()
preps = [p.prepare(c, d) for c, d in frames]
t_prep = time.perf_counter() - t
p.run_batch(preps[:B])                                  # warm-up
p._setStages()
native = p._native_stages()
t_up = t_run = 0.0
p.evaluations = 0
for lo in range(0, n, B):
    grp = preps[lo:lo + B]
    t = time.perf_counter()
    tq, t32, fl = np.stack([g.tq for g in grp]), np.stack([g.lookup_f32 for g in grp]), np.stack([g.flags for g in grp])
    t_stack = time.perf_counter() - t
    p.engine.set_targets(tq, t32, fl)
    t_up += time.perf_counter() - t
    t = time.perf_counter()
    _, _, ne = p.engine.predict_batch(native, lim, p.camera_pose, p.min_ang_inc, p.lookup_angles, p.lookup_crop, p._lookup_table, p.SPECULATE_BATCH if B >= p.SPECULATE_BATCH_FROM else p.SPECULATE)
    t_run += time.perf_counter() - t
    p.evaluations += ne
print(f"{p.intrinsics.width}x{p.intrinsics.height}, grid {len(p.lookup_angles)}, {n} frames, batch {B}: per frame — prepare {1e3 * t_prep / n:.3f} ms (serial), "
      f"stack+upload {1e3 * t_up / n:.3f} ms (stack {1e3 * t_stack / len(grp):.3f}), stage machine {1e3 * t_run / n:.3f} ms "
      f"= {n / t_run:.0f} frames/s, {p.evaluations / t_run / 1e6:.2f} M poses/s in the stage machine")
