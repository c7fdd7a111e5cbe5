"""GPU: many frames at once (rope_set_targets / rope_eval_targets / rope_lookup_score_targets / rope_predict_batch).

Every row of a batch over B frames' targets must have the bits the single-target calls give with that frame as the target, and
B frames walking the stage list in lockstep must end — stage by stage — where rope_predict takes each of them alone, and where
the sequential restatement of the reference (oracle/predictor_ref.py) does."""
import os

import numpy as np
import pytest

from oracle import predictor_ref
from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, LINK_BLUE, ZFAR, ZNEAR
from rope_s3d_amd.imgproc import resize_linear

import helpers

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.fixture(scope='module')
def scene():
    """Engine at 320x240 (six tiles), five frames of different poses with their targets."""
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=2)
    e = eng.Engine(0)
    e.set_robot(rb)
    e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    lim = rb.joint_limits
    rng = np.random.default_rng(4242)
    frames = []
    for f in range(5):
        q = rng.uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0.3, 0.3, 0])
        depth, ids = e.render(q, 6)
        if f == 3:                                      # one frame with a link missing from the target and holes in the depth
            ids = np.where(ids == 4, 255, ids).astype(np.uint8)
            depth = np.where(ids == 255, 0, depth).astype(np.float32)
            depth[60:90, 100:140] = 0
        tq, t32, flags, tgt, _, _ = helpers.synthetic_target(depth, ids)
        frames.append(dict(q=q, tq=tq, t32=t32, flags=flags, full=np.ascontiguousarray(tgt, np.float32)))
    return e, rb, intr, frames


def _single(e, fr, cand, n_render, loss, crop=None, tsweep=False):
    e.set_target(fr['tq'], fr['t32'], fr['flags'])
    if tsweep:
        e.set_target_tsweep(fr['full'])
    return e.eval(cand, n_render, loss, crop=crop)[0]


@pytest.mark.parametrize('rows_per_frame,shared', [(3, False), (26, False), (120, False), (200, True)])
@pytest.mark.parametrize('loss,n_render', [(eng.LOSS_FULL, 6), (eng.LOSS_FULL, 4), (eng.LOSS_DEPTH, 6), (eng.LOSS_TSWEEP, 6), (eng.LOSS_LOOKUP, 6)])
def test_eval_targets_rows_have_the_single_target_bits(scene, rows_per_frame, shared, loss, n_render):
    e, rb, intr, frames = scene
    lim = rb.joint_limits
    rng = np.random.default_rng(rows_per_frame * 31 + loss)
    B = len(frames)
    cand = rng.uniform(lim[:, 0], lim[:, 1], size=(B * rows_per_frame, 6)) * np.array([1, 1, 1, 0.3, 0.3, 0])
    if shared:                                          # sweeps of the third joint: rows share (q0, q1) inside a frame -> layers
        cand[:, 0] = np.repeat(rng.uniform(lim[0, 0], lim[0, 1], size=B * 4), rows_per_frame // 4)
        cand[:, 1] = np.repeat(rng.uniform(lim[1, 0], lim[1, 1], size=B * 4), rows_per_frame // 4)
    frame_of = rng.permutation(np.repeat(np.arange(B), rows_per_frame)).astype(np.int32)      # frames interleaved
    crop = np.array([20, 200, 30, 290], np.int32) if loss == eng.LOSS_LOOKUP else None
    ts = loss == eng.LOSS_TSWEEP
    e.set_targets(np.stack([f['tq'] for f in frames]), np.stack([f['t32'] for f in frames]), np.stack([f['flags'] for f in frames]),
                  np.stack([f['full'] for f in frames]) if ts else None)
    got = e.eval_targets(cand, frame_of, n_render, loss, crop)
    for f in range(B):
        sel = frame_of == f
        want = _single(e, frames[f], cand[sel], n_render, loss, crop, ts)
        assert np.array_equal(_bits(got[sel]), _bits(want)), f"frame {f}"
    # and again on resident targets (totals cached), rows in another order
    e.set_targets(np.stack([f['tq'] for f in frames]), np.stack([f['t32'] for f in frames]), np.stack([f['flags'] for f in frames]),
                  np.stack([f['full'] for f in frames]) if ts else None)
    perm = rng.permutation(len(cand))
    again = e.eval_targets(cand[perm], frame_of[perm], n_render, loss, crop)
    assert np.array_equal(_bits(again), _bits(got[perm]))


def test_eval_targets_refuses_bad_input(scene):
    e, rb, intr, frames = scene
    e.set_targets(np.stack([f['tq'] for f in frames]), None, np.stack([f['flags'] for f in frames]))
    with pytest.raises(eng.EngineError):
        e.eval_targets(np.zeros((2, 6)), [0, len(frames)], 6, eng.LOSS_FULL)          # frame index outside the targets
    with pytest.raises(eng.EngineError):
        e.eval_targets(np.zeros((2, 6)), [0, 1], 6, eng.LOSS_LOOKUP, [0, 10, 0, 10])   # no float32 planes
    # the camera-pose path's frames and the targets share their planes: one or the other
    e.set_frames(np.zeros((1, 6)), frames[0]['tq'][None])
    with pytest.raises(eng.EngineError):
        e.eval_targets(np.zeros((2, 6)), [0, 0], 6, eng.LOSS_FULL)


def test_staged_targets_go_up_beside_the_resident_ones(scene):
    """rope_stage_targets / rope_commit_targets: the second set of planes.  While a set is staged the resident one still answers;
    after the commit the staged one does, with the bits rope_set_targets gives; sets of different sizes alternate."""
    e, rb, intr, frames = scene
    lim = rb.joint_limits
    rng = np.random.default_rng(99)
    cand = rng.uniform(lim[:, 0], lim[:, 1], (12, 6))
    sets = [[0, 1, 2], [3, 4], [4, 3, 2, 1, 0], [2]]

    def planes(idx, pinned):
        arrs = (np.stack([frames[i]['tq'] for i in idx]), np.stack([frames[i]['t32'] for i in idx]), np.stack([frames[i]['flags'] for i in idx]),
                np.stack([frames[i]['full'] for i in idx]))
        if pinned:
            out = []
            for a in arrs:
                b = eng.pinned_empty(a.shape, a.dtype)
                b[...] = a
                out.append(b)
            arrs = tuple(out)
        return arrs

    def expect(idx, loss):
        e.set_targets(*planes(idx, False))
        fo = np.arange(len(cand)) % len(idx)
        return fo, e.eval_targets(cand, fo, 6, loss)

    want = {(k, loss): expect(idx, loss) for k, idx in enumerate(sets) for loss in (eng.LOSS_FULL, eng.LOSS_TSWEEP)}
    with pytest.raises(eng.EngineError):
        e.commit_targets()                                   # nothing staged
    e.set_targets(*planes(sets[0], False))
    for k in range(1, len(sets)):
        e.stage_targets(*planes(sets[k], pinned=(k % 2 == 1)))
        for loss in (eng.LOSS_FULL, eng.LOSS_TSWEEP):        # the resident set is untouched by the upload
            fo, err = want[(k - 1, loss)]
            assert np.array_equal(_bits(e.eval_targets(cand, fo, 6, loss)), _bits(err)), (k, loss)
        e.commit_targets()
        assert e.n_targets == len(sets[k])
        for loss in (eng.LOSS_FULL, eng.LOSS_TSWEEP):
            fo, err = want[(k, loss)]
            assert np.array_equal(_bits(e.eval_targets(cand, fo, 6, loss)), _bits(err)), (k, loss)
    # a change of image size between staging and committing is refused, and the context keeps working
    e.stage_targets(*planes(sets[1], True))
    intr2, PV2 = helpers.camera('640_480_color', ds=4)
    e.set_camera(PV2, intr2.width, intr2.height, ZNEAR, ZFAR)
    with pytest.raises(eng.EngineError):
        e.commit_targets()
    _, PV = helpers.camera('640_480_color', ds=2)
    e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
    fo, err = expect(sets[0], eng.LOSS_FULL)
    assert np.array_equal(_bits(err), _bits(want[(0, eng.LOSS_FULL)][1]))


def test_lookup_score_targets_equals_per_frame(scene):
    e, rb, intr, frames = scene
    grid = helpers.slu_grid(rb.joint_limits, 6)
    crop = np.array([10, 230, 20, 300], np.int32)
    e.lookup_build(grid, 6, crop)
    e.set_targets(np.stack([f['tq'] for f in frames]), np.stack([f['t32'] for f in frames]), np.stack([f['flags'] for f in frames]))
    scores, best, best_score = e.lookup_score_targets(want_scores=True)
    for f, fr in enumerate(frames):
        e.set_target(fr['tq'], fr['t32'], fr['flags'])
        s1, b1, bs1 = e.lookup_score(want_scores=True)
        assert np.array_equal(_bits(scores[f]), _bits(s1))
        assert best[f] == b1 and _bits(best_score[f]) == _bits(bs1)
        # and the table's scores are the rendered ones
        on_the_fly = e.eval(grid, 6, eng.LOSS_LOOKUP, crop=crop)[0]
        assert np.array_equal(_bits(on_the_fly), _bits(s1))


N_FRAMES = int(os.environ.get('ROPE_BATCH_FRAMES', '64'))


def _frames(renderer, lim, n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        q = rng.uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
        renderer.setJointAngles(q)
        color, depth = renderer.render()
        out.append((q, color, depth))
    return out


@pytest.mark.parametrize('do_angles,table', [('SLU', True), ('SL', True), ('SLU', False)])
def test_predict_batch_equals_frame_by_frame_and_the_reference(do_angles, table):
    """64 frames in lockstep: angles and per-stage traces equal rope_predict's on every frame, and the sequential restatement's
    on a sample of them."""
    from rope_s3d_amd import Predictor, SyntheticPredictor
    sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '640_480_color', 4, do_angles, noise=False, seed=5, lookup_divisions=4)
    p = sp.predictor
    if not table:                                       # the grid rendered and scored per frame instead of the stored table
        p.lookup_table_budget = 0
        p._loadLookup()
        assert not p._lookup_table
    rb = helpers.robot()
    lim = rb.joint_limits
    n = N_FRAMES if table else 6
    frames = _frames(sp.renderer, lim, n, 99 + len(do_angles))
    colors, depths = [f[1] for f in frames], [f[2] for f in frames]
    preps = [p.prepare(c, d) for c, d in zip(colors, depths)]
    got = p.run_batch(preps)
    traces = p.traces
    assert got.shape == (n, 6) and len(traces) == n
    evals_batch = p.evaluations
    p.evaluations = 0
    for i in range(n):
        one = p.run(colors[i], depths[i])
        assert np.array_equal(_bits(one), _bits(got[i])), f"frame {i}"
        for (k1, a1), (k2, a2) in zip(p.trace, traces[i]):
            assert k1 == k2 and np.array_equal(_bits(a1), _bits(a2)), f"frame {i} stage {k1}"
    # a lockstep batch asks for a Descent joint's under/over pair at a time (the reference's own order), the single frame for the
    # pairs of up to three joints at once: fewer poses rendered, the same decisions
    assert evals_batch <= p.evaluations if n >= p.SPECULATE_BATCH_FROM else evals_batch == p.evaluations
    # run_many takes the batched path by default, in groups smaller than the sequence
    many = p.run_many(colors, depths, batch=max(2, n // 3))
    assert np.array_equal(_bits(many), _bits(got))
    # the sequential restatement of the reference on a sample
    intr, PV = helpers.camera('640_480_color', ds=4, as_predictor=True)
    o = helpers.make_oracle(rb, intr, PV)
    names = rb.link_names
    link_blue = {nm: int(LINK_BLUE[i]) for i, nm in enumerate(names)}
    for i in range(0, n, max(1, n // 3))[:3]:
        tgt_depth = resize_linear(depths[i], intr.width, intr.height).astype(np.float64)
        tgt_blue = resize_linear(colors[i], intr.width, intr.height)[..., 0]
        want, trace, _ = predictor_ref.predict_reference(o, tgt_depth, tgt_blue, names, link_blue, lim, DEFAULT_CAMERA_POSE,
                                                         helpers.slu_grid(lim, 4), p.lookup_crop, do_angles)
        assert np.array_equal(want, got[i])
        for (k_ref, a_ref), (k_got, a_got) in zip(trace, traces[i]):
            assert np.array_equal(a_ref, a_got), f"frame {i} stage {k_got}"


def test_run_many_with_camera_poses_that_change_between_groups():
    """run_many groups consecutive frames under one camera pose; a change of pose falls between two groups — after the commit of
    one group's staged targets and before the upload of the next — and every frame gets the angles of run() with its own pose."""
    from rope_s3d_amd import SyntheticPredictor
    pose_a = np.array(DEFAULT_CAMERA_POSE, float)
    pose_b = pose_a + np.array([0.05, -0.1, 0.03, 0.0, 0.02, -0.04])
    sp = SyntheticPredictor(pose_a, '640_480_color', 4, 'SL', noise=False, seed=8, lookup_divisions=4)
    p = sp.predictor
    lim = helpers.robot().joint_limits
    colors, depths, poses = [], [], []
    for k, (pose, n) in enumerate([(pose_a, 5), (pose_b, 6), (pose_a, 3)]):
        sp.renderer.setCameraPose(pose)
        for _, c, d in _frames(sp.renderer, lim, n, 500 + k):
            colors.append(c); depths.append(d); poses.append(pose.copy())
    many = p.run_many(colors, depths, camera_poses=np.array(poses), batch=4)
    for i in range(len(colors)):
        one = p.run(colors[i], depths[i], poses[i])
        assert np.array_equal(_bits(one), _bits(many[i])), f"frame {i}"


def test_predict_batch_segmentation_path_and_tensor_sweep():
    """The segmentation path's targets (instance merge, dilate 8 / erode 7 body mask) through the batch, and a stage list with
    TensorSweep stages natively (ROPE_STAGE_TSWEEP): both equal the Python stage loop frame by frame."""
    from rope_s3d_amd import Predictor, SyntheticPredictor
    from rope_s3d_amd.prediction import predict as predict_mod
    from rope_s3d_amd.prediction.stages import Descent, InterpolativeSweep, Lookup, SFlip, TensorSweep
    from rope_s3d_amd.segmentation import ColorSegmenter
    rb = helpers.robot()
    lim = rb.joint_limits
    sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '640_480_color', 4, 'SLU', noise=False, seed=6, lookup_divisions=4)
    frames = _frames(sp.renderer, lim, 12, 1234)
    colors, depths = [f[1] for f in frames], [f[2].astype(np.float64) for f in frames]
    p = Predictor(DEFAULT_CAMERA_POSE, 4, base_intrin='640_480_color', segmenter=ColorSegmenter(['BG'] + rb.link_names, split_instances=True),
                  lookup_divisions=4)
    got = p.run_many(colors, [d.copy() for d in depths])
    p.NATIVE = False
    for i in range(len(frames)):
        assert np.array_equal(_bits(p.run(colors[i], depths[i].copy())), _bits(got[i])), f"segmentation path, frame {i}"
    # a custom stage list with TensorSweep stages
    custom = [Lookup(), SFlip(4), TensorSweep(6, 12, 'U'), Descent(4, 6, 'SL', [0.05, 0.05, 0.1, 0.5, 0.5, 0.5], early_stop=0.1),
              TensorSweep(4, 9, 'SL', range=0.2), InterpolativeSweep(6, 10, 'U', 0.1)]
    orig = predict_mod.getStages
    predict_mod.getStages = lambda angs: custom
    try:
        q = sp.predictor
        q.NATIVE = True
        assert q._setStages() is None and q._native_stages() is not None
        got = q.run_many(colors, depths)
        native_single = [q.run(colors[i], depths[i]) for i in range(len(frames))]
        q.NATIVE = False
        for i in range(len(frames)):
            want = q.run(colors[i], depths[i])
            assert np.array_equal(_bits(want), _bits(got[i])), f"tensor sweep list, frame {i} (batch)"
            assert np.array_equal(_bits(want), _bits(native_single[i])), f"tensor sweep list, frame {i} (rope_predict)"
    finally:
        predict_mod.getStages = orig
        sp.predictor.NATIVE = True
