"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as orc
from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import ZFAR, ZNEAR

import helpers

pytestmark = pytest.mark.gpu

POSES = [
    [0, 0, 0, 0, 0, 0],
    [0.3, 0.4, 0.5, 0, 0, 0],
    [-0.7, -0.9, 2.2, 0, 0, 0],
    [1.5, 1.2, -0.8, 0.5, -0.7, 0.3],
    [0.8, 0.1, 1.0, -2.0, 1.5, 3.0],
]


@pytest.fixture(scope='module')
def scene():
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color')
    o = helpers.make_oracle(rb, intr, PV)
    e = eng.Engine(0)
    e.set_robot(rb)
    e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    return rb, intr, PV, o, e


def test_link_matrices_bit_exact(scene):
    rb, intr, PV, o, e = scene
    cand = np.array(POSES, np.float64)
    d, ids = o.render(POSES[1])
    tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
    e.set_target(tq, t32, flags)
    e.eval(cand, 6, eng.LOSS_DEPTH)
    got = e.debug_mvp(len(cand), 6)
    for i, q in enumerate(POSES):
        want = o.mvp(q, 6)
        assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), f"pose {i}"


@pytest.mark.parametrize('n_render', [4, 6])
def test_render_bit_exact(scene, n_render):
    rb, intr, PV, o, e = scene
    for q in POSES:
        d_ref, id_ref = o.render(q, n_render)
        d, ids = e.render(q, n_render)
        assert np.array_equal(ids, id_ref), f"segment ids differ at {q}: {(ids != id_ref).sum()} px"
        assert np.array_equal(d.view(np.uint32), d_ref.view(np.uint32)), f"depth differs at {q}"


@pytest.mark.parametrize('loss,n_render', [(eng.LOSS_DEPTH, 6), (eng.LOSS_FULL, 6), (eng.LOSS_FULL, 4),
                                           (eng.LOSS_LOOKUP, 6), (eng.LOSS_TSWEEP, 6)])
def test_eval_sums_and_errors_bit_exact(scene, loss, n_render):
    rb, intr, PV, o, e = scene
    d, ids = o.render([0.35, 0.45, 0.9, 0, 0, 0])
    tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
    if loss == eng.LOSS_TSWEEP:
        t32 = d.copy()
    e.set_target(tq, t32, flags)
    cand = helpers.slu_grid(rb.joint_limits, 4)
    crop = [150, 479, 200, 600] if loss == eng.LOSS_LOOKUP else None
    err_ref, sums_ref = o.eval(cand, loss, n_render, tq, t32, crop, flags, threads=8, want_sums=True)
    err, sums, bi, be = e.eval(cand, n_render, loss, crop, want_sums=True)
    used = 23 if loss == eng.LOSS_FULL else 5
    assert np.array_equal(sums[:, :used], sums_ref[:, :used])
    assert np.array_equal(err.view(np.uint64), err_ref.view(np.uint64))
    assert bi == int(np.argmin(err_ref)) and be == err_ref[bi]


def test_shared_layers_do_not_change_results(scene):
    """Links 0-2 rendered once per distinct (S, L) and composited == every candidate rendering all its links."""
    rb, intr, PV, o, e = scene
    d, ids = o.render([0.35, 0.45, 0.9, 0, 0, 0])
    tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
    e.set_target(tq, t32, flags)
    cand = helpers.slu_grid(rb.joint_limits, 5)            # 125 candidates, 25 distinct (S, L) prefixes
    for loss, n in ((eng.LOSS_FULL, 6), (eng.LOSS_FULL, 2), (eng.LOSS_DEPTH, 4)):
        err_a, sums_a, bi_a, _ = e.eval(cand, n, loss, want_sums=True)
        e.set_strategy(e.NO_LAYERS)
        try:
            err_b, sums_b, bi_b, _ = e.eval(cand, n, loss, want_sums=True)
        finally:
            e.set_strategy(0)
        assert np.array_equal(sums_a, sums_b) and np.array_equal(err_a.view(np.uint64), err_b.view(np.uint64)) and bi_a == bi_b
        err_ref = o.eval(cand[::7], loss, n, tq, t32, None, flags, threads=8)
        assert np.array_equal(err_a[::7].view(np.uint64), err_ref.view(np.uint64))


def test_stored_lookup_table_scores_equal_on_the_fly(scene):
    """rope_lookup_build + rope_lookup_score == rope_eval(LOSS_LOOKUP) == oracle, bit for bit, also after the
    candidate buffers were reused by other evaluations in between."""
    rb, intr, PV, o, e = scene
    cand = helpers.slu_grid(rb.joint_limits, 5)
    crop = [150, 479, 200, 600]
    e.lookup_build(cand, 6, crop)
    for q in ([0.35, 0.45, 0.9, 0, 0, 0], [1.0, -0.3, 1.7, 0, 0, 0]):
        d, ids = o.render(q)
        tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
        e.set_target(tq, t32, flags)
        scores, bi, be = e.lookup_score(want_scores=True)
        err, _, bi2, be2 = e.eval(cand[:7], 6, eng.LOSS_FULL)         # something else uses the context in between
        err_fly, _, bi_fly, be_fly = e.eval(cand, 6, eng.LOSS_LOOKUP, crop)
        assert np.array_equal(scores.view(np.uint64), err_fly.view(np.uint64)) and bi == bi_fly and be == be_fly
        ref = o.eval(cand[::5], orc.LOSS_LOOKUP, 6, tq, t32, crop, flags, threads=8)
        assert np.array_equal(scores[::5].view(np.uint64), ref.view(np.uint64))
        scores2, bi3, _ = e.lookup_score(want_scores=True)             # table survives the other launches
        assert np.array_equal(scores2.view(np.uint64), scores.view(np.uint64)) and bi3 == bi


def test_small_batch_split_does_not_change_results(scene):
    """1-6 candidates: meshlets of a tile spread over several workgroups merged by atomicMin == one workgroup per tile."""
    rb, intr, PV, o, e = scene
    d, ids = o.render([0.35, 0.45, 0.9, 0, 0, 0])
    tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
    e.set_target(tq, t32, flags)
    cand = np.array(POSES + [[0.36, 0.44, 0.91, 0, 0, 0]], np.float64)
    for n_c in (1, 2, 6):
        for loss, n in ((eng.LOSS_FULL, 6), (eng.LOSS_FULL, 4), (eng.LOSS_DEPTH, 6)):
            err_a, sums_a, bi_a, _ = e.eval(cand[:n_c], n, loss, want_sums=True)
            # one workgroup per tile; and forward kinematics + screen boxes as a launch of their own instead of inside the split
            # raster's workgroups (the default for small batches)
            for flag in (e.NO_SPLIT, e.SEPARATE_GEOMETRY):
                e.set_strategy(flag)
                try:
                    err_b, sums_b, bi_b, _ = e.eval(cand[:n_c], n, loss, want_sums=True)
                finally:
                    e.set_strategy(0)
                assert np.array_equal(sums_a, sums_b) and np.array_equal(err_a.view(np.uint64), err_b.view(np.uint64)) and bi_a == bi_b, flag
            err_c, sums_c, bi_c, _ = e.eval(cand[:n_c], n, loss, want_sums=True)       # buffers and stamps came back usable
            assert np.array_equal(sums_a, sums_c) and bi_a == bi_c
            ref = o.eval(cand[:n_c], loss, n, tq, t32, None, flags, threads=4)
            assert np.array_equal(err_a.view(np.uint64), ref.view(np.uint64))


def test_robot_from_plain_meshes_gives_the_same_results(scene):
    """rope_set_robot_mesh (meshlets built inside the library from vertex / index arrays, as a C host would call it)
    against the default path and the Python partitioner: the image does not depend on the partition."""
    import os
    from rope_s3d_amd import engine as eng
    from rope_s3d_amd.robot import RobotModel
    rb, intr, PV, o, e = scene[:5]
    lim = rb.joint_limits
    cand = np.random.default_rng(3).uniform(lim[:, 0], lim[:, 1], (24, 6))
    q = cand[0]
    depth, ids = e.render(q, 6)
    tq = eng.pack_target(depth.astype(np.float64))
    e.set_target(tq, None, np.zeros(8, np.uint8))
    want_err, want_sums, _, _ = e.eval(cand, 6, eng.LOSS_DEPTH, want_sums=True)
    os.environ['ROPE_MESHLET_BUILDER'] = 'grow'
    try:
        python_built = RobotModel.from_urdf()
    finally:
        del os.environ['ROPE_MESHLET_BUILDER']
    for setter, model in ((e.set_robot_mesh, rb), (e.set_robot, python_built)):
        setter(model)
        try:
            d2, i2 = e.render(q, 6)
            assert np.array_equal(i2, ids) and np.array_equal(d2.view(np.uint32), depth.view(np.uint32))
            e.set_target(tq, None, np.zeros(8, np.uint8))
            err, sums, _, _ = e.eval(cand, 6, eng.LOSS_DEPTH, want_sums=True)
            assert np.array_equal(sums, want_sums) and np.array_equal(err.view(np.uint64), want_err.view(np.uint64))
        finally:
            e.set_robot(rb)


def test_robot_mesh_entry_point_rejects_broken_meshes(scene):
    import copy
    rb, e = scene[0], scene[4]
    bad = copy.copy(rb)
    bad.faces = rb.faces.copy()
    bad.faces[7, 2] = 10 ** 6                                   # an index outside its link's vertices
    with pytest.raises(eng.EngineError, match='indexes outside'):
        e.set_robot_mesh(bad)
    bad.faces = rb.faces
    bad.vtx_off = rb.vtx_off.copy()
    bad.vtx_off[3] = bad.vtx_off[2]                             # an empty link
    with pytest.raises(eng.EngineError, match='empty'):
        e.set_robot_mesh(bad)
    e.set_robot(rb)                                             # the context is still usable
    depth, ids = e.render(np.zeros(6), 6)
    assert (ids != 255).any()


def test_first_small_batch_on_fresh_contexts():
    """The very first evaluation of a context is a two-row batch (a Predictor's first SFlip): forward kinematics and boxes inside the
    split raster's workgroups, the per-(candidate, tile) stamps of a freshly allocated array, the merged tiles of a freshly
    allocated buffer.  Forty fresh contexts give the oracle's bits every time (round 3: once in a few suite runs a Predictor's
    first SFlip took the other branch — every fill and copy now goes through the context's own stream)."""
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=4)
    o = helpers.make_oracle(rb, intr, PV)
    d_ref, id_ref = o.render([0.5, 0.2, 0.9, 0, 0, 0], 6)
    tq, t32, flags, *_ = helpers.synthetic_target(d_ref, id_ref)
    rows = np.array([[-0.7853981633974483, -0.9948, 0.2327, 0, 0, 0], [0.7853981633974483, -0.9948, 0.2327, 0, 0, 0]])
    want, want_sums = o.eval(rows, eng.LOSS_FULL, 4, tq, t32, None, flags, threads=2, want_sums=True)
    for k in range(40):
        e = eng.Engine(0)
        e.set_robot(rb)
        e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
        e.set_target(tq, t32, flags)
        if k % 2:                                       # as a Predictor does it: a lookup table first
            e.lookup_build(helpers.slu_grid(rb.joint_limits, 3), 6, [5, intr.height - 5, 5, intr.width - 5])
            e.lookup_score()
        err, sums, _, _ = e.eval(rows, 4, eng.LOSS_FULL, want_sums=True)
        assert np.array_equal(sums, want_sums) and np.array_equal(err.view(np.uint64), want.view(np.uint64)), k
        err2, sums2, _, _ = e.eval(rows[::-1], 4, eng.LOSS_FULL, want_sums=True)
        assert np.array_equal(sums2, want_sums[::-1]), k
        e.close()
