#!/usr/bin/env python3
"""Write the raw arrays examples/predict_frame.c reads: robot meshes + joint chain, camera, one prepared target frame,
lookup grid and crop — everything a host needs to drive librope_hip.so through the C ABI alone."""
import os

import numpy as np


def dump_frame_bundle(directory: str, predictor, prepared) -> str:
    """`predictor`: a constructed Predictor (robot, camera, lookup grid, crop); `prepared`: Predictor.prepare(colour, depth)."""
    from rope_s3d_amd.constants import ZFAR, ZNEAR
    from rope_s3d_amd.projection import view_matrix
    os.makedirs(directory, exist_ok=True)
    rb, intr = predictor.renderer.robot, predictor.intrinsics
    PV = intr.gl_projection(ZNEAR, ZFAR) @ view_matrix(predictor.camera_pose)
    arrays = {
        'verts.f32': np.asarray(rb.verts, np.float32), 'faces.i32': np.asarray(rb.faces, np.int32),
        'vtx_off.i32': np.asarray(rb.vtx_off, np.int32), 'tri_off.i32': np.asarray(rb.tri_off, np.int32),
        'joint_fixed.f64': np.asarray(rb.joint_fixed, np.float64), 'joint_axes.f64': np.asarray(rb.joint_axes, np.float64),
        'PV.f64': np.asarray(PV, np.float64), 'clip.f64': np.array([ZNEAR, ZFAR], np.float64),
        'dims.i32': np.array([intr.width, intr.height], np.int32),
        'limits.f64': np.asarray(predictor.u_reader.joint_limits, np.float64), 'camera_pose.f64': np.asarray(predictor.camera_pose, np.float64),
        'tq.u64': np.asarray(prepared.tq, np.uint64), 't32.f32': np.asarray(prepared.lookup_f32, np.float32),
        'flags.u8': np.asarray(prepared.flags, np.uint8),
        'grid.f64': np.asarray(predictor.lookup_angles, np.float64), 'crop.i32': np.asarray(predictor.lookup_crop, np.int32),
    }
    for name, a in arrays.items():
        np.ascontiguousarray(a).tofile(os.path.join(directory, name))
    return directory
