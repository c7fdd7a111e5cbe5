#!/usr/bin/env python3
"""Generates tests/golden/camera_160x120.npz from the CPU oracle (oracle/camera_ref.py).

Camera-pose path: three frames of the robot at known joint angles seen from one camera, scored under five trial
camera poses — the integer sums of both losses per (pose, frame), the three error values of each pose, and the
final pose of a short stage list for both predictors.  No reference fixture exists for this path;
the file freezes the arithmetic contract as tests/golden/hotpath_160x120.npz does for the joint path.
    python tests/golden/make_golden_camera.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, os.pardir, os.pardir)))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, os.pardir)))

import helpers  # noqa: E402
from oracle import camera_ref  # noqa: E402
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR  # noqa: E402

TRUE_POSE = np.array(DEFAULT_CAMERA_POSE, float) + np.array([.06, -.05, .04, .01, -.015, .02])
SHORT_STAGES = [['tensorsweep', 5, .05, [True, False, True, False, False, False]],
                ['smartsweep', 4, .04, [False, True, False, False, False, True]],
                ['zp_sweep', 5, 0.05],
                ['descent', 3, 0.5, .001, [True] * 6, [0.01] * 6]]


def scene():
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=4)
    o = helpers.make_oracle(rb, intr, PV)
    P = intr.gl_projection(ZNEAR, ZFAR)
    rng = np.random.default_rng(2024)
    lim = rb.joint_limits
    qs = rng.uniform(lim[:, 0], lim[:, 1], (3, 6)) * np.array([1, 1, 1, 0, 0, 0])
    o.PV = np.ascontiguousarray(P @ camera_ref.view_of_pose(TRUE_POSE))
    frames = [o.render(q, 6) for q in qs]
    tgt = np.stack([d for d, _ in frames]).astype(np.float64)
    ids = np.stack([i for _, i in frames])
    names = rb.link_names[:6]
    seg = [{n: {'mask': ids[i] == l} for l, n in enumerate(names) if (ids[i] == l).any()} for i in range(len(qs))]
    poses = np.array(DEFAULT_CAMERA_POSE, float) + rng.uniform(-.12, .12, (5, 6))
    poses[0] = TRUE_POSE
    return rb, intr, o, P, qs, tgt, seg, names, poses


def main():
    rb, intr, o, P, qs, tgt, seg, names, poses = scene()
    seg_ref = camera_ref.CameraReference(o, P, 'segmented', qs, tgt, seg, names, stages=SHORT_STAGES)
    ml_ref = camera_ref.CameraReference(o, P, 'modelless', qs, tgt, stages=SHORT_STAGES)
    out = {'joint_poses': qs, 'camera_poses': poses, 'targets': tgt.astype(np.float32), 'planes': seg_ref.planes, 'flags': seg_ref.flags,
           'sums_full': np.stack([seg_ref.frame_sums(p, 'full') for p in poses]),
           'sums_sweep': np.stack([seg_ref.frame_sums(p, 'sweep') for p in poses]),
           'err_segmented': np.array([seg_ref.error(p) for p in poses]),
           'err_pooled': np.array([seg_ref.sweep_error(p) for p in poses]),
           'err_modelless': np.array([ml_ref.error(p) for p in poses])}
    out['final_segmented'], tr = seg_ref.run(DEFAULT_CAMERA_POSE)
    out['trace_segmented'] = np.stack([a for _, a in tr])
    out['final_modelless'], tr = ml_ref.run(DEFAULT_CAMERA_POSE)
    out['trace_modelless'] = np.stack([a for _, a in tr])
    path = os.path.join(HERE, 'camera_160x120.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), 'bytes; argmin of the segmented error:', int(np.argmin(out['err_segmented'])))


if __name__ == '__main__':
    main()
