#!/usr/bin/env python3
"""Latency of small candidate batches (descent pairs, speculative descent rows, sweeps) at the default 160x90."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.projection import Intrinsics, camera_matrix
from rope_s3d_amd.robot import RobotModel

robot = RobotModel.from_urdf()
intr = Intrinsics(sys.argv[1] if len(sys.argv) > 1 else '1280_720_color')
intr.downscale(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
PV = camera_matrix(DEFAULT_CAMERA_POSE, intr, ZNEAR, ZFAR)
e = eng.Engine(0)
e.set_robot(robot)
e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
rng = np.random.default_rng(1)
lim = robot.joint_limits
q = rng.uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
depth, ids = e.render(q, 6)
bits = np.zeros(ids.shape, np.uint64)
for l in range(6):
    bits |= (ids == l).astype(np.uint64) << np.uint64(l)
e.set_target(eng.pack_target(depth.astype(np.float64), bits), depth, np.array([3] * 6 + [0, 0], np.uint8))
for C in ([int(c) for c in sys.argv[3].split(',')] if len(sys.argv) > 3 else (2, 8, 26, 64, 256)):
    cand = q + rng.uniform(-.05, .05, (C, 6)) * np.array([1, 1, 1, 0, 0, 0])
    e.eval(cand, 6, eng.LOSS_FULL)
    t0 = time.perf_counter()
    for _ in range(200):
        e.eval(cand, 6, eng.LOSS_FULL)
    wall = (time.perf_counter() - t0) / 200
    e.upload_candidates(cand)
    k = e.profile_eval(6, eng.LOSS_FULL, None, reps=50)
    print(f"C={C:4d}  wall {wall * 1e6:7.1f} us/eval   device: fk+bounds {k['fk'] * 1e3:6.1f}  raster {k['raster'] * 1e3:6.1f}  finalize {k['finalize'] * 1e3:6.1f}  total {k['total'] * 1e3:6.1f} us")
