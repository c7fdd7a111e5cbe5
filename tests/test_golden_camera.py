"""Golden vectors of the camera-pose path (tests/golden/camera_160x120.npz, made by tests/golden/make_golden_camera.py
from the oracle): the oracle keeps reproducing them on CPU, the HIP engine and both predictors reproduce them on the GPU."""
import os
import sys

import numpy as np
import pytest

from oracle import camera_ref
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.prediction import camera_pose_prediction as cpp

import helpers

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
import make_golden_camera as mk  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'camera_160x120.npz'))


@pytest.fixture(scope='module')
def cpu():
    return mk.scene()


def test_oracle_reproduces_camera_golden(cpu):
    rb, intr, o, P, qs, tgt, seg, names, poses = cpu
    assert np.array_equal(qs, G['joint_poses']) and np.array_equal(poses, G['camera_poses'])
    assert np.array_equal(tgt.astype(np.float32), G['targets'])
    ref = camera_ref.CameraReference(o, P, 'segmented', qs, tgt, seg, names)
    assert np.array_equal(ref.planes, G['planes']) and np.array_equal(ref.flags, G['flags'])
    for k in (0, 3):                                     # two of the five poses keep the CPU suite short
        assert np.array_equal(ref.frame_sums(poses[k], 'full'), G['sums_full'][k])
        assert np.array_equal(ref.frame_sums(poses[k], 'sweep'), G['sums_sweep'][k])
    # the float epilogues of the product on the golden sums give the golden errors, bit for bit
    n_pix = float(intr.width * intr.height)
    flags = np.tile(G['flags'], (len(qs), 1))
    assert np.array_equal(cpp.camfull_error(G['sums_full'], n_pix, flags), G['err_segmented'], equal_nan=True)
    assert np.array_equal(cpp.pooled_sweep_error(G['sums_sweep'], n_pix), G['err_pooled'])
    assert np.array_equal(cpp.modelless_error(G['sums_sweep'], n_pix), G['err_modelless'])
    # pose 0 is the one the frames were rendered from, at the same resolution: no non-zero difference is left and the
    # reference's mean over an empty selection is NaN (camera_pose_prediction.py:966) — kept
    assert np.isnan(G['err_segmented'][0]) and np.isfinite(G['err_segmented'][1:]).all()


@pytest.mark.gpu
def test_engine_reproduces_camera_golden_sums(cpu):
    from rope_s3d_amd import engine as eng
    rb, intr, o, P, qs, tgt, seg, names, poses = cpu
    e = eng.Engine(0)
    e.set_robot(rb)
    e.set_camera(P @ camera_ref.view_of_pose(DEFAULT_CAMERA_POSE), intr.width, intr.height, ZNEAR, ZFAR)
    e.set_frames(qs, np.stack([eng.pack_target(d) for d in tgt]), tgt.astype(np.float32), np.tile(G['planes'][None], (len(qs), 1, 1, 1)))
    PV = np.stack([P @ camera_ref.view_of_pose(p) for p in poses])
    assert np.array_equal(e.eval_views(PV, 6, eng.LOSS_CAMFULL), G['sums_full'])
    assert np.array_equal(e.eval_views(PV, 6, eng.LOSS_TSWEEP)[..., :5], G['sums_sweep'][..., :5])


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['modelless', 'segmented'])
def test_predictors_reproduce_camera_golden_traces(cpu, mode):
    from rope_s3d_amd import CameraPredictor, ModellessCameraPredictor
    rb, intr, o, P, qs, tgt, seg, names, poses = cpu
    ids = [np.full(tgt.shape[1:], 255, np.uint8) for _ in qs]
    for i in range(len(qs)):
        for l, n in enumerate(names):
            if n in seg[i]:
                ids[i][seg[i][n]['mask']] = l

    def segmenter(color):                                 # the frames' own link ids, looked up by identity of the colour image
        k = int(color[0, 0, 0])
        return {'class_ids': np.array([l + 1 for l, n in enumerate(names) if n in seg[k]]),
                'scores': np.ones(len(seg[k])), 'masks': np.stack([seg[k][n]['mask'] for n in names if n in seg[k]], -1)}
    colors = np.stack([np.full(tgt.shape[1:] + (3,), i, np.uint8) for i in range(len(qs))])
    if mode == 'modelless':
        p = ModellessCameraPredictor(DEFAULT_CAMERA_POSE, 1, base_intrinsics='640_480_color_4')   # exact: no string round trip
    else:
        p = CameraPredictor(DEFAULT_CAMERA_POSE, 1, base_intrinsics='640_480_color_4', segmenter=segmenter)
    p.stages = [tuple(s) for s in mk.SHORT_STAGES]
    got = p.run(colors, tgt, qs)
    assert np.array_equal(np.stack([a for _, a in p.trace]), G[f'trace_{mode}'])
    assert np.array_equal(got, G[f'final_{mode}'])
