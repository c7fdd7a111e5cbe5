"""GPU: properties at BASELINE.json's full sizes (640x480, 4096 candidates; 1280x720 mh50), where the CPU oracle
would take minutes, plus edge cases of the C ABI."""
import os

import numpy as np
import pytest

from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.projection import Intrinsics, camera_matrix, view_matrix

import helpers
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def make_engine(rb, preset, pose=DEFAULT_CAMERA_POSE, ds=1):
    intr, PV = helpers.camera(preset, ds=ds, pose=pose)
    e = eng.Engine(0)
    e.set_robot(rb)
    e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    return e, intr, PV


@pytest.fixture(scope='module')
def full():
    rb = helpers.robot()
    e, intr, PV = make_engine(rb, '640_480_color')
    q_true = np.random.default_rng(7919).uniform(rb.joint_limits[:, 0], rb.joint_limits[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
    depth, ids = e.render(q_true, 6)
    tq, t32, flags, *_ = helpers.synthetic_target(depth, ids)
    e.set_target(tq, t32, flags)
    return rb, e, q_true, depth, ids, (tq, t32, flags)


def test_render_against_itself_has_zero_difference(full):
    """Scoring the target's own pose: every sample matches bit for bit, so no D != 0 and no mask mismatch."""
    rb, e, q_true, depth, ids, _ = full
    err, sums, _, _ = e.eval(q_true[None], 6, eng.LOSS_FULL, want_sums=True)
    assert sums[0, 0] == 0 and sums[0, 1] == 0            # count and sum of non-zero depth differences
    assert all(sums[0, 5 + 3 * l] == 0 for l in range(1, 6))   # per-link mask mismatches
    assert np.isnan(err[0])                                # mean of an empty selection, as numpy (predict.py:507)


def test_full_grid_is_permutation_equivariant_and_repeatable(full):
    rb, e, *_ = full
    cand = helpers.slu_grid(rb.joint_limits, 16)           # 4096 candidates (BASELINE configs[1])
    err_a, sums_a, bi_a, be_a = e.eval(cand, 6, eng.LOSS_DEPTH, want_sums=True)
    err_b, sums_b, bi_b, _ = e.eval(cand, 6, eng.LOSS_DEPTH, want_sums=True)
    assert np.array_equal(sums_a, sums_b) and bi_a == bi_b             # atomics in any order, same bits
    perm = np.random.default_rng(3).permutation(len(cand))
    err_p, sums_p, bi_p, be_p = e.eval(cand[perm], 6, eng.LOSS_DEPTH, want_sums=True)
    assert np.array_equal(sums_p, sums_a[perm])                        # grouping into shared layers changes nothing
    assert np.array_equal(err_p.view(np.uint64), err_a[perm].view(np.uint64))
    assert be_p == be_a and perm[bi_p] == bi_a or err_a[perm[bi_p]] == be_a
    assert bi_a == int(np.nanargmin(err_a))


def test_full_grid_layers_equal_direct_rendering(full):
    rb, e, *_ = full
    cand = helpers.slu_grid(rb.joint_limits, 16)
    for loss in (eng.LOSS_DEPTH, eng.LOSS_FULL):
        _, sums_a, bi_a, _ = e.eval(cand, 6, loss, want_sums=True)
        # no shared layers at all; layers without the second level (per q0); one workgroup per (tile, candidate) instead of the queue
        for flag in (e.NO_LAYERS, e.NO_PARENTS, e.NO_QUEUE, e.NO_LAYERS | e.NO_QUEUE, e.CLIP_KERNELS):
            e.set_strategy(flag)
            try:
                _, sums_b, bi_b, _ = e.eval(cand, 6, loss, want_sums=True)
            finally:
                e.set_strategy(0)
            assert np.array_equal(sums_a, sums_b) and bi_a == bi_b, (loss, flag)
    # a grid whose first joint takes few values and the second many: parents with many layers each, and n_render = 4
    grid = np.zeros((3 * 40 * 4, 6))
    q0, q1, q2 = np.meshgrid(np.linspace(-.5, 1.2, 3), np.linspace(-.9, 1.4, 40), np.linspace(-.6, 2.0, 4), indexing='ij')
    grid[:, 0], grid[:, 1], grid[:, 2] = q0.ravel(), q1.ravel(), q2.ravel()
    grid = grid[np.random.default_rng(5).permutation(len(grid))]
    for n in (4, 6):
        _, sums_a, bi_a, _ = e.eval(grid, n, eng.LOSS_FULL, want_sums=True)
        e.set_strategy(e.NO_LAYERS)
        try:
            _, sums_b, bi_b, _ = e.eval(grid, n, eng.LOSS_FULL, want_sums=True)
        finally:
            e.set_strategy(0)
        assert np.array_equal(sums_a, sums_b) and bi_a == bi_b, n


def test_mismatch_counts_match_rendered_images(full):
    """Integer link-mismatch sums of the engine == the same counts taken from its own rendered id images."""
    rb, e, q_true, depth, ids, _ = full
    for q in ([0.2, 0.1, 0.5, 0, 0, 0], [1.2, -0.5, 2.0, 0, 0, 0]):
        _, sums, _, _ = e.eval(np.array([q]), 6, eng.LOSS_FULL, want_sums=True)
        d_r, id_r = e.render(q, 6)
        for l in range(1, 6):
            assert sums[0, 5 + 3 * l] == np.count_nonzero((ids == l) != (id_r == l))
        D = np.abs(depth.astype(np.float64) - d_r)
        assert sums[0, 0] == np.count_nonzero(D)
        assert abs(sums[0, 1] / 2.0 ** 32 - D.sum()) <= 1e-9 * D.sum() + 1e-6


def test_mh50_1280x720_against_oracle():
    """BASELINE configs[4] geometry: other robot, other resolution (150 tiles, partial tile columns/rows)."""
    rb = helpers.robot('urdfs/motoman_mh50_support/urdf/mh50.urdf')
    pose = [0, -4.0, 1.5, 0, 0, 0]
    e, intr, PV = make_engine(rb, '1280_720_color', pose)
    o = helpers.make_oracle(rb, intr, PV)
    q = [0.4, 0.3, 0.2, 0.5, -0.4, 1.0]
    d_ref, id_ref = o.render(q, 6)
    d, ids = e.render(q, 6)
    assert (id_ref != 255).sum() > 20000
    assert np.array_equal(ids, id_ref) and np.array_equal(d.view(np.uint32), d_ref.view(np.uint32))
    tq, t32, flags, *_ = helpers.synthetic_target(d_ref, id_ref)
    e.set_target(tq, t32, flags)
    cand = helpers.slu_grid(rb.joint_limits, 3)
    err_ref, sums_ref = o.eval(cand, 1, 6, tq, t32, None, flags, threads=8, want_sums=True)
    err, sums, bi, _ = e.eval(cand, 6, eng.LOSS_FULL, want_sums=True)
    assert np.array_equal(sums, sums_ref) and np.array_equal(err.view(np.uint64), err_ref.view(np.uint64))


@pytest.mark.parametrize('ds', [8, 5])
def test_ragged_resolutions_against_oracle(ds):
    """160x90 and 256x144: neither dimension is a multiple of the 128x48 tile."""
    rb = helpers.robot()
    e, intr, PV = make_engine(rb, '1280_720_color', ds=ds)
    o = helpers.make_oracle(rb, intr, PV)
    for q in ([0.3, 0.4, 0.5, 0, 0, 0], [-0.7, 1.5, -0.8, 1, 1, 1]):
        d_ref, id_ref = o.render(q, 6)
        d, ids = e.render(q, 6)
        assert np.array_equal(ids, id_ref) and np.array_equal(d.view(np.uint32), d_ref.view(np.uint32))


def test_robot_partly_and_wholly_out_of_view():
    """Close camera: triangles cross the image border and the near plane; far-away look direction: nothing drawn."""
    rb = helpers.robot()
    e, intr, PV = make_engine(rb, '640_480_color', pose=[0.1, -0.45, 0.5, 0, 0.3, 0.2], ds=2)
    o = helpers.make_oracle(rb, intr, PV)
    d_ref, id_ref = o.render([0.5, 0.5, 0.5, 0, 0, 0], 6)
    d, ids = e.render([0.5, 0.5, 0.5, 0, 0, 0], 6)
    assert np.array_equal(ids, id_ref) and np.array_equal(d.view(np.uint32), d_ref.view(np.uint32))
    # a camera 0.2 m from the upper arm, and one inside the robot's envelope: triangles reach and cross the near plane (5 cm)
    # and are cut there, as OpenGL cuts them (render.py:92-98, projection.py:161-169) — images, sums and errors bit for bit
    # (0.2 m in front of the forearm: nothing crosses yet; 0.12 m, tilted: 1 800 triangles do; inside the envelope: 700 - 1 300)
    for pose, q in (([0.3, -0.2, 0.77, 0, 0, 0], [0, 0, 0, 0, 0, 0]), ([0.3, -0.12, 0.77, 0, 0.2, 0.3], [0, 0, 0, 0, 0, 0]),
                    ([0.2, -0.1, 0.6, 0.3, 0.1, -0.4], [0.5, 0.4, 0.6, 0.2, 0.3, 0.1]), ([0.05, -0.12, 0.25, 0, -0.4, 0.1], [0, 0, 0, 0, 0, 0]),
                    ([0.4, -0.06, 0.8, 0, 0, 0.5], [0, 0.3, 0.2, 0, 0, 0])):
        ec, intrc, PVc = make_engine(rb, '640_480_color', pose=pose, ds=2)
        oc = helpers.make_oracle(rb, intrc, PVc)
        d_ref, id_ref = oc.render(q, 6)
        d, ids = ec.render(q, 6)
        assert (id_ref != 255).mean() > 0.2, pose                      # the robot fills a good part of the view
        assert np.array_equal(ids, id_ref) and np.array_equal(d.view(np.uint32), d_ref.view(np.uint32)), pose
        tq, t32, flags, *_ = helpers.synthetic_target(d_ref, id_ref)
        ec.set_target(tq, t32, flags)
        cand = np.array(q) + np.random.default_rng(2).uniform(-.3, .3, (40, 6))
        for c_ in (cand, cand[:3]):                                    # the big-batch and the small-batch (split) launch
            err_ref, sums_ref = oc.eval(c_, eng.LOSS_FULL, 6, tq, t32, None, flags, threads=8, want_sums=True)
            err, sums, _, _ = ec.eval(c_, 6, eng.LOSS_FULL, want_sums=True)
            assert np.array_equal(sums, sums_ref) and np.array_equal(err.view(np.uint64), err_ref.view(np.uint64)), pose
    e2, intr2, PV2 = make_engine(rb, '640_480_color', pose=[0, -1.5, 0.75, 0, 0, 3.1], ds=2)     # looking away
    d, ids = e2.render([0, 0, 0, 0, 0, 0], 6)
    assert (ids == 255).all() and (d == 0).all()
    tq = eng.pack_target(np.zeros((intr2.height, intr2.width)))
    e2.set_target(tq, None, np.zeros(8, np.uint8))
    err, sums, bi, _ = e2.eval(np.zeros((3, 6)), 6, eng.LOSS_DEPTH, want_sums=True)
    assert (sums == 0).all() and np.isnan(err).all() and bi == 0


def test_single_and_odd_candidate_counts(full):
    rb, e, *_ = full
    cand = helpers.slu_grid(rb.joint_limits, 16)
    err_all, *_ = e.eval(cand, 4, eng.LOSS_FULL)
    for sl in (slice(0, 1), slice(5, 6), slice(100, 1331)):
        err, _, bi, be = e.eval(cand[sl], 4, eng.LOSS_FULL)
        assert np.array_equal(err.view(np.uint64), err_all[sl].view(np.uint64))
        assert be == err[bi] and bi == int(np.argmin(err))


def test_abi_rejects_bad_calls():
    rb = helpers.robot()
    e = eng.Engine(0)
    with pytest.raises(eng.EngineError):
        e.upload_candidates(np.zeros((4, 6)))                      # robot/camera not set
    e.set_robot(rb)
    intr, PV = helpers.camera('640_480_color', ds=4)
    with pytest.raises(eng.EngineError):
        e.set_camera(PV, 0, 10, ZNEAR, ZFAR)
    e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    e.upload_candidates(np.zeros((4, 6)))
    with pytest.raises(eng.EngineError):
        e.eval_resident(6, eng.LOSS_DEPTH)                         # no target yet
    e.set_target(eng.pack_target(np.zeros((intr.height, intr.width))), None, np.zeros(8, np.uint8))
    with pytest.raises(eng.EngineError):
        e.eval_resident(7, eng.LOSS_DEPTH)                         # more links than the robot has
    with pytest.raises(eng.EngineError):
        e.eval_resident(6, eng.LOSS_LOOKUP, [0, 10, 0, 10])        # lookup loss needs the float32 plane
    with pytest.raises(eng.EngineError):
        e.upload_candidates(np.full((2, 6), np.nan))
    with pytest.raises(ValueError):
        e.set_target(np.zeros((3, 3), np.uint64))
    e.eval_resident(6, eng.LOSS_DEPTH)                              # still usable after the refused calls
    e.sync()
    # rope_eval_views reuses the per-candidate buffers for (view, frame) rows: the resident candidates and the last results
    # are gone after it, and saying so beats reading a 256-row mapped block (or NULL) with C = K*N rows
    err0 = e.download()[0]
    H, W = intr.height, intr.width
    e.set_frames(np.zeros((300, 6)), np.zeros((300, H, W), np.uint64))
    e.eval_views(np.stack([PV]), 6, eng.LOSS_DEPTH)
    with pytest.raises(eng.EngineError):
        e.eval_resident(6, eng.LOSS_DEPTH)
    with pytest.raises(eng.EngineError):
        e.profile_eval(6, eng.LOSS_DEPTH, None, reps=1)
    with pytest.raises(eng.EngineError):
        e.download()
    e.upload_candidates(np.zeros((4, 6)))
    with pytest.raises(eng.EngineError):
        e.download()                                                # new candidates, nothing evaluated yet
    e.eval_resident(6, eng.LOSS_DEPTH)
    assert np.array_equal(e.download()[0].view(np.uint64), err0.view(np.uint64))


def test_more_candidates_than_one_launch_holds():
    """70 000 rows (> 65 535 per launch) at a small resolution: chunked evaluation, merged argmin."""
    rb = helpers.robot()
    e, intr, PV = make_engine(rb, '640_480_color', ds=8)
    d, ids = e.render([0.3, 0.4, 0.9, 0, 0, 0], 6)
    e.set_target(eng.pack_target(d.astype(np.float64)), None, np.zeros(8, np.uint8))
    base = helpers.slu_grid(rb.joint_limits, 10)                      # 1000 distinct poses
    cand = np.tile(base, (70, 1))                                      # 70 000 rows
    err, _, bi, be = e.eval(cand, 6, eng.LOSS_DEPTH)
    ref, _, bi0, be0 = e.eval(base, 6, eng.LOSS_DEPTH)
    assert err.shape == (70000,) and np.array_equal(err.reshape(70, 1000).view(np.uint64), np.tile(ref.view(np.uint64), (70, 1)))
    assert bi == bi0 and be == be0                                     # first occurrence wins across launches too


def test_lookup_grid_beyond_one_launch():
    """41^3 = 68 921 grid poses (the reference allows 200 divisions per joint, lookup.py:50, constants.py:30): the stored
    table is built batch after batch, and both the table scores and the on-the-fly lookup loss through one rope_eval call
    equal the batched Python evaluation row for row."""
    import ctypes as C
    rb = helpers.robot()
    e, intr, PV = make_engine(rb, '640_480_color', ds=8)
    d, ids = e.render([0.3, 0.4, 0.9, 0, 0, 0], 6)
    tq, t32, flags = helpers.synthetic_target(d, ids)[:3]
    e.set_target(tq, t32, flags)
    grid = helpers.slu_grid(rb.joint_limits, 41)
    assert len(grid) == 68921 > e.MAX_BATCH
    crop = [5, intr.height - 4, 7, intr.width - 6]
    want, _, bi0, be0 = e.eval(grid, 6, eng.LOSS_LOOKUP, crop)          # Python-side batches of <= 65 535 rows
    e.lookup_build(grid, 6, crop)
    scores, bi, be = e.lookup_score(want_scores=True)
    assert np.array_equal(scores.view(np.uint64), want.view(np.uint64)) and (bi, be) == (bi0, be0)
    # the same rows through ONE call across the C boundary (what rope_predict's Lookup stage does without a table)
    err = np.empty(len(grid))
    bi1, be1 = C.c_int32(), C.c_double()
    crop_a = np.ascontiguousarray(crop, np.int32)
    rc = e._lib.rope_eval(e._ctx, grid.ctypes.data_as(C.c_void_p), len(grid), 6, eng.LOSS_LOOKUP, crop_a.ctypes.data_as(C.c_void_p),
                          err.ctypes.data_as(C.c_void_p), None, C.byref(bi1), C.byref(be1))
    assert rc == 0 and np.array_equal(err.view(np.uint64), want.view(np.uint64)) and (bi1.value, be1.value) == (bi0, be0)


def test_frames_of_more_than_256_tiles():
    """2560x1440 is 20 x 15 = 300 tiles: more than the raster queue keeps per-tile weights for, so its pairs go in candidate
    order in one segment.  600 candidates through the queue equal the one-workgroup-per-pair launch, the unshared form, and the
    oracle on a sample."""
    rb = helpers.robot()
    from rope_s3d_amd.projection import Intrinsics, camera_matrix
    intr = Intrinsics('[ 2560x1440  p[1276.78 722.986]  f[1810.46 1809.72]  Inverse Brown Conrady [0 0 0 0 0] ]')
    assert -(-intr.width // 128) * -(-intr.height // 96) > 256
    PV = camera_matrix(DEFAULT_CAMERA_POSE, intr, ZNEAR, ZFAR)
    e = eng.Engine(0)
    e.set_robot(rb)
    e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    o = helpers.make_oracle(rb, intr, PV)
    d, ids = e.render([0.3, 0.4, 0.9, 0, 0, 0], 6)
    tq, t32, flags = helpers.synthetic_target(d, ids)[:3]
    e.set_target(tq, t32, flags)
    rng = np.random.default_rng(23)
    lim = rb.joint_limits
    cand = np.array([0.3, 0.4, 0.9, 0, 0, 0]) + rng.uniform(-0.4, 0.4, (600, 6)) * np.array([1, 1, 1, 0, 0, 0])
    cand[::3, :2] = cand[0, :2]                                        # some rows share their first two joints: layers
    err_a, sums_a, bi_a, _ = e.eval(cand, 6, eng.LOSS_DEPTH, want_sums=True)
    for flag in (e.NO_QUEUE, e.NO_LAYERS):
        e.set_strategy(flag)
        try:
            err_b, sums_b, bi_b, _ = e.eval(cand, 6, eng.LOSS_DEPTH, want_sums=True)
        finally:
            e.set_strategy(0)
        assert np.array_equal(sums_a, sums_b) and bi_a == bi_b, flag
    pick = rng.choice(len(cand), 6, replace=False)
    ref = o.eval(cand[pick], orc.LOSS_DEPTH, 6, tq, t32, None, flags, threads=6)
    assert np.array_equal(err_a[pick].view(np.uint64), ref.view(np.uint64))


def test_stored_lookup_table_on_a_large_crop():
    """The stored table keeps of every row the rectangle that holds its samples and adds the sums of |T| outside it from a
    per-frame total: at 1280x720 with the whole frame as the crop (rectangles up to ~10^5 samples, rows that hold nothing because
    the pose leaves the frame, a second target against the same table) the scores equal the loss rendered on the fly, bit for bit."""
    rb = helpers.robot()
    e, intr, PV = make_engine(rb, '1280_720_color', ds=1)
    lim = rb.joint_limits
    rng = np.random.default_rng(17)
    grid = rng.uniform(lim[:, 0], lim[:, 1], (300, 6)) * np.array([1, 1, 1, 0, 0, 0])
    crop = [0, intr.height - 1, 0, intr.width - 1]
    e.lookup_build(grid, 6, crop)
    for q in ([0.3, 0.4, 0.9, 0, 0, 0], [-1.2, 0.9, -0.4, 0, 0, 0]):
        d, ids = e.render(q, 6)
        tq, t32, flags = helpers.synthetic_target(d, ids)[:3]
        e.set_target(tq, t32, flags)
        scores, bi, be = e.lookup_score(want_scores=True)
        want, _, bi0, be0 = e.eval(grid, 6, eng.LOSS_LOOKUP, crop)
        assert np.array_equal(scores.view(np.uint64), want.view(np.uint64)) and (bi, be) == (bi0, be0)
    # a camera that sees nothing of the robot: every row's rectangle is empty, every score the frame's own total
    e.set_camera(PV @ np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 500.0], [0, 0, 0, 1.0]]), intr.width, intr.height, 0.05, 100.0)
    e.set_target(tq, t32, flags)
    e.lookup_build(grid[:40], 6, crop)
    scores, _, _ = e.lookup_score(want_scores=True)
    want, _, _, _ = e.eval(grid[:40], 6, eng.LOSS_LOOKUP, crop)
    assert np.array_equal(scores.view(np.uint64), want.view(np.uint64)) and len(set(scores.tolist())) == 1


@pytest.mark.parametrize('seed', list(range(1, 1 + int(__import__('os').environ.get('ROPE_FUZZ_SEEDS', '12')))))
def test_random_scenes_against_oracle(seed):
    """Random cameras (near, oblique, partly off-screen), all six joints anywhere in their limits, any number of rendered
    links, odd image sizes, every loss, batch sizes on both sides of the small-batch split: sums and errors bit for bit."""
    from rope_s3d_amd.urdf import URDFReader
    rng = np.random.default_rng(seed)
    rb = helpers.robot()
    full_lim = URDFReader().joint_limits
    ds = int(rng.choice([3, 4, 5, 7]))
    pose = np.array([0, -1.5, 0.75, 0, 0, 0], float) + rng.uniform(-1, 1, 6) * np.array([.6, .6, .4, .25, .25, .4])
    e, intr, PV = make_engine(rb, '640_480_color', pose=pose, ds=1)
    # a size no preset produces: crop the projection to W x H by rebuilding the camera at the odd resolution
    W, H = 640 // ds + int(rng.integers(0, 3)), 480 // ds + int(rng.integers(0, 3))
    from rope_s3d_amd.constants import ZFAR, ZNEAR
    from rope_s3d_amd.projection import Intrinsics, view_matrix
    it = Intrinsics('640_480_color')
    P = np.zeros((4, 4))
    P[0, 0], P[1, 1] = 2 * it.fx / ds / W, 2 * it.fy / ds / H
    P[0, 2], P[1, 2] = 1 - 2 * it.cx / ds / W, 2 * it.cy / ds / H - 1
    P[2, 2], P[2, 3], P[3, 2] = (ZFAR + ZNEAR) / (ZNEAR - ZFAR), 2 * ZFAR * ZNEAR / (ZNEAR - ZFAR), -1
    PV = P @ view_matrix(pose)
    e.set_camera(PV, W, H, ZNEAR, ZFAR)
    o = orc.Oracle(rb.verts, rb.faces, rb.vtx_off, rb.tri_off, rb.joint_fixed, rb.joint_axes, PV, W, H, ZNEAR, ZFAR)
    q_t = rng.uniform(full_lim[:, 0], full_lim[:, 1])
    d, ids = o.render(q_t, 6)
    tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
    e.set_target(tq, t32, flags)
    for C in (3, 40, 300):
        cand = rng.uniform(full_lim[:, 0], full_lim[:, 1], (C, 6))
        cand[: C // 2, :2] = cand[0, :2]                   # some rows share their first two joints: layers may engage
        for loss in (eng.LOSS_DEPTH, eng.LOSS_FULL, eng.LOSS_TSWEEP):
            n = int(rng.integers(1, 7))
            err, sums, bi, be = e.eval(cand, n, loss, want_sums=True)
            err_ref, sums_ref = o.eval(cand[:40], loss, n, tq, t32, None, flags, threads=8, want_sums=True)
            assert np.array_equal(sums[:40], sums_ref), (seed, C, loss, n, W, H)
            assert np.array_equal(err[:40].view(np.uint64), err_ref.view(np.uint64))
        crop = [int(H * .2), int(H * .8), int(W * .1), int(W * .9)]
        err, sums, *_ = e.eval(cand[:40], 6, eng.LOSS_LOOKUP, crop=crop, want_sums=True)
        err_ref, sums_ref = o.eval(cand[:40], eng.LOSS_LOOKUP, 6, tq, t32, crop, flags, threads=8, want_sums=True)
        assert np.array_equal(sums, sums_ref) and np.array_equal(err.view(np.uint64), err_ref.view(np.uint64))


@pytest.mark.parametrize('seed', list(range(1, 1 + int(__import__('os').environ.get('ROPE_RENDER_SEEDS', '4')))))
def test_random_full_resolution_renders_against_oracle(seed):
    """Both robots at their full resolutions under random cameras and joint vectors: depth and link-id images bit for bit
    (the close cameras put triangles across the image border, behind the eye and hundreds of pixels wide)."""
    from rope_s3d_amd.urdf import URDFReader
    rng = np.random.default_rng(500 + seed)
    big = seed % 2 == 0
    urdf = 'urdfs/motoman_mh50_support/urdf/mh50.urdf' if big else None
    rb = helpers.robot(urdf)
    lim = URDFReader(urdf).joint_limits if urdf else URDFReader().joint_limits
    base = np.array([0, -4.0, 1.5, 0, 0, 0] if big else [0, -1.5, 0.75, 0, 0, 0], float)
    pose = base + rng.uniform(-1, 1, 6) * np.array([.8, .8, .5, .3, .3, .5]) * (2.5 if big else 1.0) * np.array([1, 1, 1, .4, .4, .4])
    e, intr, PV = make_engine(rb, '1280_720_color' if big else '640_480_color', pose)
    o = helpers.make_oracle(rb, intr, PV)
    for _ in range(2):
        q = rng.uniform(lim[:, 0], lim[:, 1])
        n = int(rng.integers(1, 7))
        d_ref, id_ref = o.render(q, n)
        d, ids = e.render(q, n)
        assert np.array_equal(ids, id_ref), (seed, n, int((ids != id_ref).sum()))
        assert np.array_equal(d.view(np.uint32), d_ref.view(np.uint32)), (seed, n)


def test_full_grid_sample_against_oracle(full):
    """The bench workload itself (4096 candidates, 640x480, shared layers, big-batch path): a random sample of its rows
    against the oracle, sums and errors bit for bit, for the depth-only and the full loss; and the 1280x720 grid of cfg5."""
    rb, e, q_true, depth, ids, (tq, t32, flags) = full
    intr, PV = helpers.camera('640_480_color')
    o = helpers.make_oracle(rb, intr, PV)
    cand = helpers.slu_grid(rb.joint_limits, 16)
    pick = np.sort(np.random.default_rng(17).choice(len(cand), 48, replace=False))
    for loss in (eng.LOSS_DEPTH, eng.LOSS_FULL):
        err, sums, *_ = e.eval(cand, 6, loss, want_sums=True)
        err_ref, sums_ref = o.eval(cand[pick], loss, 6, tq, t32, None, flags, threads=8, want_sums=True)
        assert np.array_equal(sums[pick], sums_ref), loss
        assert np.array_equal(err[pick].view(np.uint64), err_ref.view(np.uint64)), loss
    rb5 = helpers.robot('urdfs/motoman_mh50_support/urdf/mh50.urdf')
    e5, intr5, PV5 = make_engine(rb5, '1280_720_color', [0, -4.0, 1.5, 0, 0, 0])
    o5 = helpers.make_oracle(rb5, intr5, PV5)
    d5, i5 = e5.render([0.4, 0.3, 0.2, 0, 0, 0], 6)
    tq5, t325, fl5, *_ = helpers.synthetic_target(d5, i5)
    e5.set_target(tq5, t325, fl5)
    cand5 = helpers.slu_grid(rb5.joint_limits, 12)
    pick5 = np.sort(np.random.default_rng(18).choice(len(cand5), 24, replace=False))
    err, sums, *_ = e5.eval(cand5, 6, eng.LOSS_DEPTH, want_sums=True)
    err_ref, sums_ref = o5.eval(cand5[pick5], eng.LOSS_DEPTH, 6, tq5, t325, None, fl5, threads=8, want_sums=True)
    assert np.array_equal(sums[pick5], sums_ref) and np.array_equal(err[pick5].view(np.uint64), err_ref.view(np.uint64))


def test_cfg5_full_size_grid():
    """BASELINE configs[4] at its stated size on one GPU: mh50, 1280x720, 32^3 = 32 768 candidates per frame.  All rows:
    repeatable, equivariant under a permutation of the rows, and the same with nothing shared between candidates and with
    one workgroup per (tile, candidate); 32 sampled rows against the oracle bit for bit.  (The config's "fp16 depth buffers"
    have no counterpart: no depth buffer reaches HBM, and the 24-bit window depth in LDS is wider than fp16.)"""
    rb = helpers.robot('urdfs/motoman_mh50_support/urdf/mh50.urdf')
    e, intr, PV = make_engine(rb, '1280_720_color', [0, -4.0, 1.5, 0, 0, 0])
    o = helpers.make_oracle(rb, intr, PV)
    q_true = np.random.default_rng(7919).uniform(rb.joint_limits[:, 0], rb.joint_limits[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
    d, ids = e.render(q_true, 6)
    tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
    e.set_target(tq, t32, flags)
    cand = helpers.slu_grid(rb.joint_limits, 32)
    assert len(cand) == 32768
    err, sums, bi, be = e.eval(cand, 6, eng.LOSS_DEPTH, want_sums=True)
    assert bi == int(np.nanargmin(err)) and be == err[bi]
    pick = np.sort(np.random.default_rng(19).choice(len(cand), 32, replace=False))
    err_ref, sums_ref = o.eval(cand[pick], eng.LOSS_DEPTH, 6, tq, t32, None, flags, threads=16, want_sums=True)
    assert np.array_equal(sums[pick], sums_ref) and np.array_equal(err[pick].view(np.uint64), err_ref.view(np.uint64))
    perm = np.random.default_rng(4).permutation(len(cand))
    err_p, sums_p, bi_p, be_p = e.eval(cand[perm], 6, eng.LOSS_DEPTH, want_sums=True)
    assert np.array_equal(sums_p, sums[perm]) and np.array_equal(err_p.view(np.uint64), err[perm].view(np.uint64)) and be_p == be
    for flag in (e.NO_LAYERS, e.NO_QUEUE):
        e.set_strategy(flag)
        try:
            err_b, sums_b, bi_b, _ = e.eval(cand, 6, eng.LOSS_DEPTH, want_sums=True)
        finally:
            e.set_strategy(0)
        assert np.array_equal(sums_b, sums) and bi_b == bi, flag
    # the full _error on the same grid: every link term, sampled rows against the oracle
    err_f, sums_f, *_ = e.eval(cand, 6, eng.LOSS_FULL, want_sums=True)
    err_ref, sums_ref = o.eval(cand[pick[:12]], eng.LOSS_FULL, 6, tq, t32, None, flags, threads=16, want_sums=True)
    assert np.array_equal(sums_f[pick[:12]], sums_ref) and np.array_equal(err_f[pick[:12]].view(np.uint64), err_ref.view(np.uint64))


@pytest.mark.parametrize('pose,q0', [([0.3, -0.12, 0.77, 0, 0.2, 0.3], [0, 0, 0, 0, 0, 0]), ([0.2, -0.1, 0.6, 0.3, 0.1, -0.4], [0.5, 0.4, 0.6, 0.2, 0.3, 0.1])])
def test_clipping_kernels_layers_queue_and_every_loss(pose, q0):
    """The instantiations that clip at the near plane, where they broke before: a camera inside the near plane's reach of the
    robot (triangles are cut), a batch large enough for the queue with shared layers (a grid over U under 48 (S, L) pairs), the
    lookup loss with a crop, the depth, full and TensorSweep losses — every launch structure (layers / none / queue / none)
    gives the same sums, the same sums again on a second and third pass (round 3: the spilled layer-queue clipping kernel did
    not), and sampled rows equal the oracle's bit for bit (round 2: a memory fault while these kernels were being written)."""
    rb = helpers.robot()
    e, intr, PV = make_engine(rb, '640_480_color', pose=pose, ds=2)
    o = helpers.make_oracle(rb, intr, PV)
    d_ref, id_ref = o.render(q0, 6)
    tq, t32, flags, tgt, _, _ = helpers.synthetic_target(d_ref, id_ref)
    full32 = np.ascontiguousarray(tgt, np.float32)
    e.set_target(tq, t32, flags)
    e.set_target_tsweep(full32)
    rng = np.random.default_rng(41)
    sl = np.array(q0)[:2] + rng.uniform(-.25, .25, (48, 2))
    cand = np.zeros((48 * 12, 6))
    cand[:, :2] = np.repeat(sl, 12, axis=0)
    cand[:, 2] = np.tile(np.array(q0)[2] + np.linspace(-.4, .4, 12), 48)
    cand[:, 3:] = np.array(q0)[3:]
    H, W = intr.height, intr.width
    crop = [int(H * .15), int(H * .9), int(W * .1), int(W * .95)]
    pick = np.sort(rng.choice(len(cand), 10, replace=False))
    for loss in (eng.LOSS_LOOKUP, eng.LOSS_DEPTH, eng.LOSS_FULL, eng.LOSS_TSWEEP):
        cr = crop if loss == eng.LOSS_LOOKUP else None
        _, base, bi, _ = e.eval(cand, 6, loss, crop=cr, want_sums=True)
        for flag in (0, 0, e.NO_LAYERS, e.NO_QUEUE, e.NO_LAYERS | e.NO_QUEUE, e.NO_PARENTS):
            e.set_strategy(flag)
            try:
                _, again, bi2, _ = e.eval(cand, 6, loss, crop=cr, want_sums=True)
            finally:
                e.set_strategy(0)
            assert np.array_equal(again, base) and bi2 == bi, (loss, flag)
        t_plane = full32 if loss == eng.LOSS_TSWEEP else t32
        err_ref, sums_ref = o.eval(cand[pick], loss, 6, tq, t_plane, cr, flags, threads=8, want_sums=True)
        assert np.array_equal(base[pick], sums_ref), loss
        # small batches (the split path) through the clipping instantiations as well
        _, few, *_ = e.eval(cand[pick[:3]], 6, loss, crop=cr, want_sums=True)
        assert np.array_equal(few, sums_ref[:3]), loss


def test_profiling_build_finds_no_index_out_of_range(tmp_path):
    """The profiling build of the library (-DROPE_PROFILE) checks every index into the raster kernels' shared arrays and queue
    segments at run time: the clipping test above, the bench grid and a small batch leave the violation count at zero."""
    import subprocess
    import sys
    import textwrap
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir))
    lib = os.path.join(root, 'rope_s3d_amd', 'csrc', 'librope_hip_profile.so')
    if not os.path.exists(lib):
        pytest.skip("librope_hip_profile.so not built (python tools/build_variants.py profile)")
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import helpers
        from rope_s3d_amd import engine as eng
        from rope_s3d_amd.constants import ZFAR, ZNEAR
        rb = helpers.robot()
        for pose, ds in (([0.3, -0.12, 0.77, 0, 0.2, 0.3], 2), ([0, -1.5, 0.75, 0, 0, 0], 1)):
            intr, PV = helpers.camera('640_480_color', ds=ds, pose=pose)
            e = eng.Engine(0)
            e.set_robot(rb)
            e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
            d, ids = e.render([0, 0, 0, 0, 0, 0], 6)
            tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
            e.set_target(tq, t32, flags)
            cand = helpers.slu_grid(rb.joint_limits, 12 if ds == 2 else 16) * 0.3
            for loss, crop in ((eng.LOSS_FULL, None), (eng.LOSS_LOOKUP, [10, intr.height - 10, 10, intr.width - 10])):
                for flag in (0, e.CLIP_KERNELS, e.NO_LAYERS, e.NO_QUEUE):
                    e.set_strategy(flag)
                    e.eval(cand, 6, loss, crop=crop)
                    e.eval(cand[:5], 6, loss, crop=crop)
            assert e.debug_bounds() == 0, e.debug_bounds()
        print('bounds ok')
    ''') % (root, os.path.join(root, 'tests'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=dict(os.environ, ROPE_HIP_LIB=lib))
    assert r.returncode == 0 and 'bounds ok' in r.stdout, r.stdout + r.stderr
