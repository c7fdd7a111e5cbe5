"""Shared scene construction for the tests: robot, camera, oracle, synthetic targets."""
import functools

import numpy as np

from oracle import oracle as orc
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.projection import Intrinsics, camera_matrix
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
    PV = camera_matrix(pose, intr, ZNEAR, ZFAR)          # the one implementation every caller shares (rope_camera_matrix)
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


def keras_arrays(state_dict, num_classes: int = 7) -> dict:
    """A MaskRCNN state_dict as the variables of a Keras weight file: <layer>/<scope>/<var>:0 below model_weights (as
    model.save() nests them), the RPN inside rpn_model, kernels in Keras layouts."""
    from rope_s3d_amd.maskrcnn import matterport_layer_map
    lm = matterport_layer_map(num_classes)
    names = {'weight': {'conv': 'kernel', 'deconv': 'kernel', 'dense': 'kernel', 'bn': 'gamma'}, 'bias': {'bn': 'beta'},
             'running_mean': {'bn': 'moving_mean'}, 'running_var': {'bn': 'moving_variance'}}
    arrays = {}
    for key, t in state_dict.items():
        prefix, var = key.rsplit('.', 1)
        if var == 'num_batches_tracked':
            continue
        layer, kind = lm[prefix]
        v = t.numpy()
        if var == 'weight' and kind in ('conv', 'deconv'):
            v = v.transpose(2, 3, 1, 0)                                   # OIHW -> HWIO; (in, out, kh, kw) -> (kh, kw, out, in)
        elif var == 'weight' and kind == 'dense':
            v = v.T
        outer = 'rpn_model' if layer.startswith('rpn_') else layer
        arrays[f"model_weights/{outer}/{layer}/{names[var].get(kind, var)}:0"] = v
    return arrays


def write_keras_weights(state_dict, path: str, num_classes: int = 7) -> str:
    from rope_s3d_amd.data import hdf5
    return hdf5.write_arrays(path, keras_arrays(state_dict, num_classes), {'backend': 'tensorflow', 'keras_version': '2.4.0'})
