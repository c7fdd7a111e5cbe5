#!/usr/bin/env python3
"""Two frames in flight on one GPU: two engine contexts (two HIP streams), each with its own frame and candidate
grid, evaluated alternately without waiting.  The short kernels of one pass (FK, boxes, shared layers, finalize) fill
the tail of the other's scoring launch.  Same workload as bench.py (cfg1); prints poses/s for 1 and 2 contexts."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from bench import slu_grid
from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.projection import Intrinsics, camera_matrix
from rope_s3d_amd.robot import RobotModel

robot = RobotModel.from_urdf()
intr = Intrinsics('640_480_color')
PV = camera_matrix(DEFAULT_CAMERA_POSE, intr, ZNEAR, ZFAR)
cand = slu_grid(robot.joint_limits, 16)


def make(seed):
    e = eng.Engine(0)
    e.set_robot(robot)
    e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    rng = np.random.default_rng(seed)
    q = rng.uniform(robot.joint_limits[:, 0], robot.joint_limits[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
    depth, _ = e.render(q, 6)
    e.set_target(eng.pack_target(depth.astype(np.float64)), None, np.zeros(8, np.uint8))
    e.upload_candidates(cand)
    e.eval_resident(6, eng.LOSS_DEPTH)
    e.sync()
    return e


K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for n_ctx in (1, 2, 3):
    es = [make(7919 + i) for i in range(n_ctx)]
    t0 = time.perf_counter()
    for s in range(K):
        es[s % n_ctx].eval_resident(6, eng.LOSS_DEPTH)
    for e in es:
        e.sync()
    dt = time.perf_counter() - t0
    best = [e.download(want_err=False)[2] for e in es]
    print(f"{n_ctx} context(s): {K} passes of {len(cand)} candidates in {dt * 1e3:.1f} ms = {K * len(cand) / dt:,.0f} poses/s  (argmin {best})")
