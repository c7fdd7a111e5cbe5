#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench workload: the frame's target planes and the candidate grid are handed over from host
memory inside the timed region (rope_set_target + rope_eval: upload, evaluate, results back), 4096 candidates per frame."""
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
e = eng.Engine(0)
e.set_robot(robot)
e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
cand = slu_grid(robot.joint_limits, 16)
q = np.random.default_rng(7919).uniform(robot.joint_limits[:, 0], robot.joint_limits[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
depth, _ = e.render(q, 6)
tq, t32, flags = eng.pack_target(depth.astype(np.float64)), np.ascontiguousarray(depth), np.zeros(8, np.uint8)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for resident in (True, False):
    e.set_target(tq, t32, flags); e.upload_candidates(cand); e.eval_resident(6, eng.LOSS_DEPTH); e.sync()
    t0 = time.perf_counter()
    for _ in range(K):
        if resident:
            e.eval_resident(6, eng.LOSS_DEPTH)
        else:
            e.set_target(tq, t32, flags)
            e.eval(cand, 6, eng.LOSS_DEPTH)
    e.sync()
    dt = time.perf_counter() - t0
    print(f"{'inputs resident in HBM' if resident else 'target planes + candidates from host memory, errors back, every pass'}: "
          f"{dt / K * 1e3:.3f} ms per pass = {K * len(cand) / dt:,.0f} poses/s")
