#!/usr/bin/env python3
"""Phase ablation of the small-batch chain (MODE_SPLIT raster + scoring) at the Predictor's 160x90; needs the profiling build."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.projection import Intrinsics, camera_matrix
from rope_s3d_amd.robot import RobotModel

robot = RobotModel.from_urdf()
intr = Intrinsics('1280_720_color')
intr.downscale(8)
PV = camera_matrix(DEFAULT_CAMERA_POSE, intr, ZNEAR, ZFAR)
e = eng.Engine(0)
e.set_robot(robot)
e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
if not hasattr(e._lib, 'rope_debug_skip'):
    raise SystemExit("needs the profiling build: python tools/build_variants.py profile && ROPE_HIP_LIB=$PWD/rope_s3d_amd/csrc/librope_hip_profile.so")
rng = np.random.default_rng(1)
lim = robot.joint_limits
q = rng.uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
depth, ids = e.render(q, 6)
bits = np.zeros(ids.shape, np.uint64)
for l in range(6):
    bits |= (ids == l).astype(np.uint64) << np.uint64(l)
e.set_target(eng.pack_target(depth.astype(np.float64), bits), depth, np.array([3] * 6 + [0, 0], np.uint8))
names = {0: 'full', 16: 'no loss pass', 32 | 64 | 16: 'set-up only, no loss', 8 | 16: 'no pixel work', 4 | 16: 'no cull', 2 | 16: 'list only', 1: 'mask check + exit'}
for C in (2, 26):
    cand = q + rng.uniform(-.05, .05, (C, 6)) * np.array([1, 1, 1, 0, 0, 0])
    e.upload_candidates(cand)
    for mask, name in names.items():
        e.debug_skip(mask)
        k = e.profile_eval(6, eng.LOSS_FULL, None, reps=50)
        print(f"C={C:3d} {name:24s} fk+bounds {k['fk'] * 1e3:6.1f}  raster {k['raster'] * 1e3:6.1f}  finalize {k['finalize'] * 1e3:6.1f}  total {k['total'] * 1e3:6.1f} us")
e.debug_skip(0)
