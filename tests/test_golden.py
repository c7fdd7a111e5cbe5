"""Golden vectors (tests/golden/hotpath_160x120.npz, made by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them on CPU, the HIP engine must reproduce them on the GPU."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR

import helpers

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'hotpath_160x120.npz'))
CASES = (('full6', 1, 6, None), ('full4', 1, 4, None), ('depth6', 0, 6, None), ('lookup6', 2, 6, G['lookup_crop']))


@pytest.fixture(scope='module')
def cpu():
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=4)
    o = helpers.make_oracle(rb, intr, PV)
    d, ids = o.render(G['target_pose'])
    return rb, intr, PV, o, helpers.synthetic_target(d, ids)


def test_oracle_reproduces_golden_keys(cpu):
    rb, intr, PV, o, _ = cpu
    for n in (4, 6):
        for q, want in zip(G['poses'], G[f'keys_n{n}']):
            assert np.array_equal(o.raster_key(q, n), want)


@pytest.mark.parametrize('name,loss,n,crop', CASES)
def test_oracle_reproduces_golden_sums_and_errors(cpu, name, loss, n, crop):
    rb, intr, PV, o, (tq, t32, flags, *_) = cpu
    err, sums = o.eval(G['candidates'], loss, n, tq, t32, crop, flags, threads=4, want_sums=True)
    assert np.array_equal(sums, G[f'sums_{name}']) and np.array_equal(err.view(np.uint64), G[f'err_{name}'].view(np.uint64))


@pytest.fixture(scope='module')
def gpu(cpu):
    from rope_s3d_amd import engine as eng
    rb, intr, PV, o, tgt = cpu
    e = eng.Engine(0)
    e.set_robot(rb)
    e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    return e


@pytest.mark.gpu
def test_engine_reproduces_golden_keys(cpu, gpu):
    for n in (4, 6):
        for q, want in zip(G['poses'], G[f'keys_n{n}']):
            d, ids = gpu.render(q, n)
            o = cpu[3]
            d_ref, id_ref = o.resolve(want)
            assert np.array_equal(ids, id_ref) and np.array_equal(d.view(np.uint32), d_ref.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize('name,loss,n,crop', CASES)
def test_engine_reproduces_golden_sums_and_errors(cpu, gpu, name, loss, n, crop):
    tq, t32, flags, *_ = cpu[4]
    gpu.set_target(tq, t32, flags)
    err, sums, bi, be = gpu.eval(G['candidates'], n, loss, crop, want_sums=True)
    want = G[f'err_{name}']
    assert np.array_equal(sums, G[f'sums_{name}']) and np.array_equal(err.view(np.uint64), want.view(np.uint64))
    assert bi == int(np.argmin(want))


@pytest.mark.gpu
def test_predictor_reproduces_golden_trace():
    from rope_s3d_amd import SyntheticPredictor
    sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '640_480_color', 4, 'SLU', noise=False, lookup_divisions=4)
    assert list(sp.predictor.lookup_crop) == list(G['frame_crop6'])
    pose, predicted = sp.run(G['frame_q_true'])
    got = np.stack([a for _, a in sp.predictor.trace])
    assert np.array_equal(got, G['frame_trace']) and np.array_equal(predicted, G['frame_final'])
    assert np.abs(predicted - G['frame_q_true'])[:3].max() < 0.05


@pytest.mark.gpu
def test_engine_renders_equal_the_independent_exact_rasteriser():
    """The HIP engine against tests/golden/pins_raster_160x120.npz (an independent rasteriser in Python integers over the real mesh,
    tests/golden/make_pins.py): same covered pixels, same link on every pixel."""
    from rope_s3d_amd import engine as eng
    pins = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pins_raster_160x120.npz'))
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=4)
    e = eng.Engine(0)
    e.set_robot(rb)
    e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    for k, q in enumerate(pins['poses']):
        depth, ids = e.render(q, 6)
        near = pins[f'near{k}']
        assert np.array_equal(ids != 255, pins[f'ids{k}'] != 255) and np.array_equal(ids[~near], pins[f'ids{k}'][~near]), k
        assert ((depth != 0) == (ids != 255)).all()

