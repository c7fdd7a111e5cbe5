#!/usr/bin/env python3
"""Streaming rate of the stored-lookup-table score kernel (HBM-bound): bytes of table per second."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.projection import Intrinsics, camera_matrix
from rope_s3d_amd.robot import RobotModel
from rope_s3d_amd.simulation.lookup import lookup_grid

robot = RobotModel.from_urdf()
for preset, ds, d in (('1280_720_color', 8, 25), ('640_480_color', 1, 16)):
    intr = Intrinsics(preset); intr.downscale(ds)
    PV = camera_matrix(DEFAULT_CAMERA_POSE, intr, ZNEAR, ZFAR)
    e = eng.Engine(0); e.set_robot(robot); e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    depth, ids = e.render([0.3, 0.4, 0.9, 0, 0, 0], 6)
    e.set_target(eng.pack_target(depth.astype(np.float64)), depth, np.zeros(8, np.uint8))
    crop = [0, intr.height - 1, 0, intr.width - 1]
    cand = lookup_grid(robot.joint_limits, 'SLU', [d] * 6)
    t0 = time.perf_counter(); e.lookup_build(cand, 6, crop); tb = time.perf_counter() - t0
    e.lookup_score()
    t0 = time.perf_counter()
    for _ in range(20): e.lookup_score()
    dt = (time.perf_counter() - t0) / 20
    nbytes = len(cand) * intr.width * intr.height * 4
    print(f"{intr.width}x{intr.height}, {len(cand)} rows, table {nbytes / 1e9:.2f} GB: build {tb * 1e3:.1f} ms, score {dt * 1e3:.3f} ms/frame = {nbytes / dt / 1e12:.2f} TB/s (host-timed, includes launch + copy of the argmin)")
