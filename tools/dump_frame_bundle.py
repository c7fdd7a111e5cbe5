#!/usr/bin/env python3
"""Write the raw arrays examples/predict_frame.c reads — INPUTS only: robot meshes + joint chain and limits, the camera pose, the
base intrinsics and the down-sampling factor, the colour-coded frame with its depth, the links' colours, the lookup divisions.
Nothing the Python host computes (camera matrix, crop, pose grids, packed target) is in it: the C host derives those through the
library, and tests/test_c_host.py compares its angles with Predictor.run's."""
import os

import numpy as np


def dump_frame_bundle(directory: str, predictor, color: np.ndarray, depth: np.ndarray, base_intrin, lookup_divisions: int) -> str:
    """`predictor`: a constructed synthetic-path Predictor (robot, camera pose, ds_factor, colour dictionary); `color` / `depth`: the
    full-size frame as the camera delivers it; `base_intrin`: the camera's full-size Intrinsics (preset name or object)."""
    from rope_s3d_amd.projection import Intrinsics
    os.makedirs(directory, exist_ok=True)
    rb, base = predictor.renderer.robot, Intrinsics(base_intrin)
    arrays = {
        'verts.f32': np.asarray(rb.verts, np.float32), 'faces.i32': np.asarray(rb.faces, np.int32),
        'vtx_off.i32': np.asarray(rb.vtx_off, np.int32), 'tri_off.i32': np.asarray(rb.tri_off, np.int32),
        'joint_fixed.f64': np.asarray(rb.joint_fixed, np.float64), 'joint_axes.f64': np.asarray(rb.joint_axes, np.float64),
        'limits.f64': np.asarray(predictor.u_reader.joint_limits, np.float64), 'camera_pose.f64': np.asarray(predictor.camera_pose, np.float64),
        'intrinsics.f64': np.array([base.width, base.height, base.cx, base.cy, base.fx, base.fy], np.float64),
        'setup.i32': np.array([predictor.ds_factor, lookup_divisions], np.int32),
        'link_blue.i32': np.array([predictor.color_dict[k][0] for k in predictor.link_names], np.int32),
        'color.u8': np.asarray(color, np.uint8), 'depth.f32': np.asarray(depth, np.float32),
    }
    for name, a in arrays.items():
        np.ascontiguousarray(a).tofile(os.path.join(directory, name))
    return directory
