#!/usr/bin/env python3
"""Phase ablation of raster_score_kernel on the bench workload (profiling aid; needs a GPU)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from bench import slu_grid
from rope_s3d_amd import engine as eng  # noqa: E402
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.projection import Intrinsics, camera_matrix
from rope_s3d_amd.robot import RobotModel

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LOSS = {'depth': eng.LOSS_DEPTH, 'full': eng.LOSS_FULL}[sys.argv[2] if len(sys.argv) > 2 else 'depth']
robot = RobotModel.from_urdf()
intr = Intrinsics('640_480_color')
PV = camera_matrix(DEFAULT_CAMERA_POSE, intr, ZNEAR, ZFAR)
e = eng.Engine(0)
e.set_robot(robot)
e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
rng = np.random.default_rng(7919)
q = rng.uniform(robot.joint_limits[:, 0], robot.joint_limits[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
depth, ids = e.render(q, 6)
bits = np.zeros(ids.shape, np.uint64)
for l in range(6):
    bits |= (ids == l).astype(np.uint64) << np.uint64(l)
e.set_target(eng.pack_target(depth.astype(np.float64), bits), None, np.array([3] * 6 + [0, 0], np.uint8))
cand = slu_grid(robot.joint_limits, grid)
e.upload_candidates(cand)
e.eval_resident(6, LOSS)
e.sync()
names = {0: 'full', 128: 'partial last row chunk dropped (bound)', 16: 'no loss pass', 32 | 16: 'no small-tri loops, no loss', 64 | 16: 'no row pass, no loss', 32 | 64 | 16: 'setup only (no S, no rows), no loss', 8: 'no pixel loop', 8 | 16: 'no pixel loop, no loss', 4 | 16: 'no triangle phase, no loss',
         2 | 16: 'no vertex/triangle, no loss', 1: 'link cull + exit only'}
if not hasattr(e._lib, 'rope_debug_skip'):
    raise SystemExit("needs the profiling build: python tools/build_variants.py profile && ROPE_HIP_LIB=$PWD/rope_s3d_amd/csrc/librope_hip_profile.so")
e.set_strategy(e.NO_LAYERS)
t = e.profile_eval(6, LOSS, None, reps=3)
print(f"{'full, no shared layers':32s} raster {t['raster']:8.3f} ms   ({len(cand) / t['raster'] * 1e3:10.0f} poses/s)")
e.set_strategy(0)
for mask, name in names.items():
    e.debug_skip(mask)
    t = e.profile_eval(6, LOSS, None, reps=3)
    print(f"{name:32s} raster {t['raster']:8.3f} ms   ({len(cand) / t['raster'] * 1e3:10.0f} poses/s)")
e.debug_skip(0)
