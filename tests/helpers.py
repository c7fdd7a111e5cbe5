"""Shared scene construction for the tests: robot, camera, oracle, synthetic targets."""
import functools

import numpy as np

from oracle import oracle as orc
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.projection import Intrinsics, view_matrix
from rope_s3d_amd.robot import RobotModel
from rope_s3d_amd.urdf import URDFReader


@functools.lru_cache(maxsize=4)
def robot(urdf: str = None) -> RobotModel:
    return RobotModel.from_urdf(URDFReader(urdf) if urdf else URDFReader())


def camera(preset='640_480_color', ds=1, pose=DEFAULT_CAMERA_POSE, as_predictor=False):
    """as_predictor: Predictor hands its down-scaled Intrinsics OBJECT to Renderer, whose constructor — like the
    reference's (projection.py:20-46, render.py:41) — rebuilds it from its string form: six significant digits."""
    intr = Intrinsics(preset)
    if ds != 1:
        intr.downscale(ds)
    if as_predictor:
        intr = Intrinsics(intr)
    PV = intr.gl_projection(ZNEAR, ZFAR) @ view_matrix(pose)
    return intr, PV


def make_oracle(rb: RobotModel, intr, PV) -> orc.Oracle:
    return orc.Oracle(rb.verts, rb.faces, rb.vtx_off, rb.tri_off, rb.joint_fixed, rb.joint_axes, PV,
                      intr.width, intr.height, ZNEAR, ZFAR)


def synthetic_target(depth_f32: np.ndarray, ids: np.ndarray):
    """What Predictor._loadSynthetic caches (predict.py:445-469), as engine/oracle inputs.

    -> tq (uint64 plane), t32 lookup plane, link_flags (8,), plus the float64 pieces for
    the numpy-literal error."""
    tgt = depth_f32.astype(np.float64)
    bits = np.zeros(ids.shape, np.uint64)
    flags = np.zeros(8, np.uint8)
    masks, masked = {}, {}
    for l in range(6):
        m = (ids == l) | ((ids == 255) if l == 0 else False)       # base_link's colour 0 == background 0
        if m.sum() > 0:
            bits |= m.astype(np.uint64) << np.uint64(l)
            flags[l] |= 1
            tm = m * tgt
            if np.sum(tm != 0) > 0.05 * np.sum(m):
                flags[l] |= 2
            masks[l], masked[l] = m, tm
    tq = orc.pack_target(tgt, bits)
    lookup = tgt * (ids != 255)        # predict.py:449-454: depth where any of the six link colours matches
    return tq, np.ascontiguousarray(lookup.astype(np.float32)), flags, tgt, masks, masked


def slu_grid(limits: np.ndarray, d: int) -> np.ndarray:
    """SLU lookup grid, joint 0 fastest (robotpose/simulation/lookup.py:56-66)."""
    divs = np.array([d, d, d, 1, 1, 1])
    num = int(np.prod(divs))
    ang = np.zeros((num, 6))
    for idx in range(3):
        rng = np.linspace(limits[idx, 0], limits[idx, 1], divs[idx])
        repeat = int(np.prod(divs[:idx]))
        tile = num // (repeat * divs[idx])
        ang[:, idx] = np.tile(np.repeat(rng, repeat), tile)
    return ang
