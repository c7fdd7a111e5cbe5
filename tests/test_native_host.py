"""rope_predict.cpp without a GPU: the stage loop is host code over the C ABI, so here its rope_eval calls are answered
by the CPU oracle (tests/native_shim.cpp) and its decisions are compared, stage by stage, with the sequential
restatement of the reference (oracle/predictor_ref.py).  The -m gpu suite repeats this with the real engine."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import predictor_ref
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, LINK_BLUE
from rope_s3d_amd.crop import crop_pose_grid
from rope_s3d_amd.engine import STAGE_DESCENT, STAGE_ISWEEP, STAGE_LOOKUP, STAGE_SFLIP, PredictArgs, StageDesc

import helpers

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir))
THREADS = min(os.cpu_count() or 1, 8)
NAN = float('nan')


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('shim') / 'libpredict_shim.so')
    # ROPE_SHIM_SANITIZE: the product's host C++ under AddressSanitizer + UBSan (tests/test_host_sanitized.py runs this
    # module once more in a child process with the sanitiser runtime preloaded)
    san = ['-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-g'] if os.environ.get('ROPE_SHIM_SANITIZE') else []
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-ffp-contract=off', '-fPIC', '-shared'] + san +
                          [os.path.join(ROOT, 'rope_s3d_amd', 'csrc', 'rope_predict.cpp'), os.path.join(ROOT, 'rope_s3d_amd', 'csrc', 'rope_meshlets.cpp'),
                           os.path.join(ROOT, 'tests', 'native_shim.cpp'), '-o', out])
    lib = C.CDLL(out)
    lib.rope_predict.argtypes = [C.c_void_p, C.POINTER(PredictArgs), C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.shim_last_error.restype = C.c_char_p
    return lib


def _stages(do_angles):
    def desc(kind, n, count=0, joints='', init=None, redux=0.5, stop=0.01, rng=NAN):
        d = StageDesc(kind, n, count, sum(1 << 'SLURBT'.index(c) for c in joints))
        d.init_rate[:] = [NAN if r is None else r for r in (init or [None] * 6)]
        d.rate_reduction, d.early_stop, d.range = redux, stop, rng
        return d
    if do_angles == 'SL':        # stages.py:138-150
        return [desc(STAGE_LOOKUP, 6), desc(STAGE_SFLIP, 4), desc(STAGE_ISWEEP, 4, 10, 'L', rng=0.1), desc(STAGE_ISWEEP, 4, 10, 'S', rng=0.1),
                desc(STAGE_SFLIP, 4)]
    return [desc(STAGE_LOOKUP, 6), desc(STAGE_SFLIP, 4), desc(STAGE_DESCENT, 4, 10, 'SL', [0.05, 0.05, 0.1, 0.5, 0.5, 0.5], stop=0.1),
            desc(STAGE_SFLIP, 4), desc(STAGE_ISWEEP, 6, 25, 'U'), desc(STAGE_SFLIP, 4), desc(STAGE_SFLIP, 6), desc(STAGE_ISWEEP, 6, 10, 'U', rng=0.1),
            desc(STAGE_DESCENT, 6, 40, 'SLU', stop=0.0075)]          # stages.py:152-168


TILTED = [0.1, -1.6, 0.8, 0.05, 0.08, -0.1]          # non-zero camera angles: SFlip's axis then mixes cam[4] and cam[5] (predict.py:245)


@pytest.mark.parametrize('do_angles,seed,speculate,pose', [('SLU', 7919, 3, DEFAULT_CAMERA_POSE), ('SLU', 7920, 1, DEFAULT_CAMERA_POSE),
                                                           ('SL', 7921, 3, DEFAULT_CAMERA_POSE), ('SLU', 7922, 3, TILTED)]
                         + [('SLU' if k % 3 else 'SL', 9000 + k, 1 + k % 3, TILTED if k % 2 else DEFAULT_CAMERA_POSE)
                            for k in range(int(os.environ.get('ROPE_SHIM_SEEDS', '0')))])
def test_stage_loop_against_sequential_reference(shim, do_angles, seed, speculate, pose):
    rb = helpers.robot()
    lim = rb.joint_limits
    intr, PV = helpers.camera('640_480_color', ds=8, pose=pose, as_predictor=True)          # 80x60: the oracle renders ~500 poses per run
    o = helpers.make_oracle(rb, intr, PV)
    q_true = np.random.default_rng(seed).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
    depth, ids = o.render(q_true, 6)
    tq, t32, flags, tgt, _, _ = helpers.synthetic_target(depth, ids)
    blue = np.where(ids == 255, 0, np.asarray(LINK_BLUE)[np.minimum(ids, 5)]).astype(np.uint8)
    grid = helpers.slu_grid(lim, 4)
    cover = o.coverage(crop_pose_grid(lim, intr.size, 6)[0], 6, threads=THREADS) != 0
    r, c = np.where(cover)
    crop = np.array([max(r.min() - 10, 0), min(r.max() + 10, intr.height - 1), max(c.min() - 10, 0), min(c.max() + 10, intr.width - 1)], np.int32)
    names = rb.link_names
    want, trace, n_eval = predictor_ref.predict_reference(o, tgt, blue, names, {n: int(LINK_BLUE[i]) for i, n in enumerate(names)}, lim,
                                                          pose, grid, crop, do_angles)
    calls = []

    @C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32))
    def answer(cand, n, n_render, loss, crop_p, err_out, best_idx):
        rows = np.ctypeslib.as_array(cand, (n, 6)).copy()
        calls.append(n)
        if loss == orc.LOSS_LOOKUP:
            err = o.eval(rows, loss, n_render, tq, t32, np.ctypeslib.as_array(crop_p, (4,)), flags, threads=THREADS)
        else:
            err = o.eval(rows, loss, n_render, tq, link_flags=flags, threads=THREADS)
        if err_out:
            np.ctypeslib.as_array(err_out, (n,))[:] = err
        if best_idx:
            best_idx[0] = int(np.argmin(np.where(np.isnan(err), np.inf, err)))
        return 0
    shim.shim_set_callback(answer)
    stages = _stages(do_angles)
    arr = (StageDesc * len(stages))(*stages)
    limits, cam, inc = np.ascontiguousarray(lim, np.float64), np.asarray(pose, np.float64), np.array([.005] * 6)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    args = PredictArgs(arr, len(arr), speculate, p(limits), p(cam), p(inc), p(grid), len(grid), 0, p(crop))
    out, got_trace, n = np.empty(6), np.empty((len(arr), 6)), C.c_int64()
    rc = shim.rope_predict(C.c_void_p(1), C.byref(args), p(out), p(got_trace), C.byref(n))
    assert rc == 0, shim.shim_last_error()
    assert len(trace) == len(stages)
    for k, (kind, ang) in enumerate(trace):
        assert np.array_equal(got_trace[k], ang), (k, kind, got_trace[k], ang)
    assert np.array_equal(out, want)
    if speculate == 1:
        # the serial order asks for exactly the reference's renders, minus the lower-limit render SFlip throws away
        assert n_eval - 4 <= n.value <= n_eval
    assert calls[0] == len(grid) and max(calls[1:]) <= 26          # the lookup grid, then batches of at most 1 + 25 poses
    assert shim.shim_ranges_opened() >= len(stages) and shim.shim_range_depth() == 0      # one named range per stage, all closed


def test_frames_in_lockstep_equal_frame_by_frame(shim):
    """rope_predict_batch (B frames through the stage list in lockstep, every step one batch over all frames) against the
    sequential restatement per frame, and against rope_predict per frame: same angles after every stage.  Includes a custom list
    with a TensorSweep stage (ROPE_STAGE_TSWEEP) and a frame without depth (all errors NaN) among the others."""
    rb = helpers.robot()
    lim = rb.joint_limits
    pose = TILTED
    intr, PV = helpers.camera('640_480_color', ds=8, pose=pose, as_predictor=True)
    o = helpers.make_oracle(rb, intr, PV)
    grid = helpers.slu_grid(lim, 3)
    cover = o.coverage(crop_pose_grid(lim, intr.size, 6)[0], 6, threads=THREADS) != 0
    r, c = np.where(cover)
    crop = np.array([max(r.min() - 10, 0), min(r.max() + 10, intr.height - 1), max(c.min() - 10, 0), min(c.max() + 10, intr.width - 1)], np.int32)
    names = rb.link_names
    blue_of = {n: int(LINK_BLUE[i]) for i, n in enumerate(names)}
    rng = np.random.default_rng(77)
    targets = []
    for f in range(4):
        q_true = rng.uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
        depth, ids = o.render(q_true, 6)
        if f == 2:                                           # a frame with nothing in it: NaN errors all the way
            depth, ids = np.zeros_like(depth), np.full(ids.shape, 255, np.uint8)
        tq, t32, flags, tgt, _, _ = helpers.synthetic_target(depth, ids)
        blue = np.where(ids == 255, 0, np.asarray(LINK_BLUE)[np.minimum(ids, 5)]).astype(np.uint8)
        targets.append(dict(tq=tq, t32=t32, flags=flags, tgt=tgt, blue=blue, full=np.ascontiguousarray(tgt, np.float32)))

    def score(rows, fr, n_render, loss, crop_p):
        if loss == orc.LOSS_LOOKUP:
            return o.eval(rows, loss, n_render, fr['tq'], fr['t32'], np.ctypeslib.as_array(crop_p, (4,)), fr['flags'], threads=THREADS)
        if loss == orc.LOSS_TSWEEP:
            return o.eval(rows, loss, n_render, fr['tq'], fr['full'], None, fr['flags'], threads=THREADS)
        return o.eval(rows, loss, n_render, fr['tq'], link_flags=fr['flags'], threads=THREADS)

    batches = []

    @C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double))
    def answer_targets(cand, frame_of, n, n_render, loss, crop_p, err_out):
        rows = np.ctypeslib.as_array(cand, (n, 6)).copy()
        fo = np.ctypeslib.as_array(frame_of, (n,)).copy()
        batches.append((n, len(set(fo.tolist()))))
        out = np.ctypeslib.as_array(err_out, (n,))
        for f in sorted(set(fo.tolist())):
            with np.errstate(all='ignore'):
                out[fo == f] = score(rows[fo == f], targets[f], n_render, loss, crop_p)
        return 0

    current = [0]

    @C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32))
    def answer_single(cand, n, n_render, loss, crop_p, err_out, best_idx):
        rows = np.ctypeslib.as_array(cand, (n, 6)).copy()
        with np.errstate(all='ignore'):
            err = score(rows, targets[current[0]], n_render, loss, crop_p)
        if err_out:
            np.ctypeslib.as_array(err_out, (n,))[:] = err
        if best_idx:
            ok = ~np.isnan(err)
            best_idx[0] = int(np.flatnonzero(ok)[np.argmin(err[ok])]) if ok.any() else 0
        return 0
    shim.shim_set_targets_callback.argtypes = [C.c_void_p]
    shim.shim_set_targets_callback(C.cast(answer_targets, C.c_void_p))
    shim.shim_set_callback(answer_single)
    shim.rope_predict_batch.argtypes = [C.c_void_p, C.POINTER(PredictArgs), C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    limits, cam, inc = np.ascontiguousarray(lim, np.float64), np.asarray(pose, np.float64), np.array([.005] * 6)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    tsweep = StageDesc(4, 6, 7, 1 << 2)                      # ROPE_STAGE_TSWEEP over U, 7 divisions, whole range
    tsweep.range = NAN
    tsweep.init_rate[:] = [NAN] * 6
    for label, stages in (('SLU', _stages('SLU')), ('SL', _stages('SL')), ('custom', _stages('SL')[:2] + [tsweep] + _stages('SLU')[2:4])):
        arr = (StageDesc * len(stages))(*stages)
        B = len(targets)
        for speculate in (3, 1):
            args = PredictArgs(arr, len(arr), speculate, p(limits), p(cam), p(inc), p(grid), len(grid), 0, p(crop))
            out, trace, n = np.empty((B, 6)), np.empty((B, len(arr), 6)), C.c_int64()
            batches.clear()
            assert shim.rope_predict_batch(C.c_void_p(1), C.byref(args), B, p(out), p(trace), C.byref(n)) == 0, shim.shim_last_error()
            assert any(frames == B for _, frames in batches)              # steps really carry the rows of all frames
            total = 0
            for f in range(B):
                current[0] = f
                one, one_trace, n1 = np.empty(6), np.empty((len(arr), 6)), C.c_int64()
                assert shim.rope_predict(C.c_void_p(1), C.byref(args), p(one), p(one_trace), C.byref(n1)) == 0, shim.shim_last_error()
                assert np.array_equal(one_trace.view(np.uint64), trace[f].view(np.uint64)), (label, speculate, f)
                assert np.array_equal(one.view(np.uint64), out[f].view(np.uint64))
                total += n1.value
                if label != 'custom' and speculate == 3:
                    with np.errstate(all='ignore'):
                        want, ref_trace, _ = predictor_ref.predict_reference(o, targets[f]['tgt'], targets[f]['blue'], names, blue_of, lim, pose,
                                                                             grid, crop, label)
                    for k, (kind, ang) in enumerate(ref_trace):
                        assert np.array_equal(trace[f, k], ang, equal_nan=True), (label, f, k, kind)
            assert n.value == total
    # the table aliasing makes frames depend on their order: refused for a batch
    live = grid.copy()
    args = PredictArgs(arr, len(arr), 3, p(limits), p(cam), p(inc), p(grid), len(grid), 0, p(crop), p(live))
    assert shim.rope_predict_batch(C.c_void_p(1), C.byref(args), B, p(out), p(trace), None) == -1
    assert b'order' in shim.shim_last_error()


def test_argument_checks_without_an_engine(shim):
    lim = np.ascontiguousarray(helpers.robot().joint_limits, np.float64)
    cam, inc = np.asarray(DEFAULT_CAMERA_POSE, np.float64), np.array([.005] * 6)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    out = np.empty(6)
    for stage, text in ((StageDesc(STAGE_ISWEEP, 6, 3, 4), b'at least 4 divisions'), (StageDesc(STAGE_LOOKUP, 6), b'needs the pose grid'),
                        (StageDesc(7, 6), b'unknown stage kind'), (StageDesc(STAGE_SFLIP, 9), b'to_render must be 1..6')):
        arr = (StageDesc * 1)(stage)
        args = PredictArgs(arr, 1, 3, p(lim), p(cam), p(inc), None, 0, 0, None)
        assert shim.rope_predict(C.c_void_p(1), C.byref(args), p(out), None, None) == -1
        assert text in shim.shim_last_error()


def test_stage_loop_with_nan_errors_follows_the_reference(shim):
    """A frame without any depth makes every E(a) NaN (mean of an empty set, predict.py:503-507).  The reference's loop then
    runs on Python's NaN semantics — comparisons false, min() keeping a leading NaN, interp1d's NaN spline — and the
    library's loop has to land on the same angles."""
    rb = helpers.robot()
    lim = rb.joint_limits
    away = [0, 5.0, 0.75, 0, 0, 0]                       # the default camera moved past the robot: it now looks away from it
    intr, PV = helpers.camera('640_480_color', ds=8, pose=away, as_predictor=True)
    o = helpers.make_oracle(rb, intr, PV)
    depth = np.zeros((intr.height, intr.width), np.float32)
    ids = np.full(depth.shape, 255, np.uint8)
    tq, t32, flags, tgt, _, _ = helpers.synthetic_target(depth, ids)
    blue = np.zeros(depth.shape, np.uint8)
    grid = helpers.slu_grid(lim, 3)
    crop = np.array([0, intr.height - 1, 0, intr.width - 1], np.int32)
    names = rb.link_names
    with np.errstate(all='ignore'):
        want, trace, _ = predictor_ref.predict_reference(o, tgt, blue, names, {n: int(LINK_BLUE[i]) for i, n in enumerate(names)}, lim,
                                                         away, grid, crop, 'SLU')
    seen_nan = []

    @C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32))
    def answer(cand, n, n_render, loss, crop_p, err_out, best_idx):
        rows = np.ctypeslib.as_array(cand, (n, 6)).copy()
        if loss == orc.LOSS_LOOKUP:
            err = o.eval(rows, loss, n_render, tq, t32, np.ctypeslib.as_array(crop_p, (4,)), flags, threads=THREADS)
        else:
            err = o.eval(rows, loss, n_render, tq, link_flags=flags, threads=THREADS)
            seen_nan.append(bool(np.isnan(err).all()))
        if err_out:
            np.ctypeslib.as_array(err_out, (n,))[:] = err
        if best_idx:                       # the engine's rule: first smallest, NaN never wins, all NaN -> row 0
            ok = ~np.isnan(err)
            best_idx[0] = int(np.flatnonzero(ok)[np.argmin(err[ok])]) if ok.any() else 0
        return 0
    shim.shim_set_callback(answer)
    stages = _stages('SLU')
    arr = (StageDesc * len(stages))(*stages)
    limits, cam, inc = np.ascontiguousarray(lim, np.float64), np.asarray(away, np.float64), np.array([.005] * 6)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    args = PredictArgs(arr, len(arr), 3, p(limits), p(cam), p(inc), p(grid), len(grid), 0, p(crop))
    out, got_trace = np.empty(6), np.empty((len(arr), 6))
    assert shim.rope_predict(C.c_void_p(1), C.byref(args), p(out), p(got_trace), None) == 0, shim.shim_last_error()
    assert seen_nan and all(seen_nan)
    for k, (kind, ang) in enumerate(trace):
        assert np.array_equal(got_trace[k], ang), (k, kind, got_trace[k], ang)
    assert np.array_equal(out, want)


def test_partitioner_and_robot_builder_in_the_host_build(shim):
    """rope_partition_mesh and rope_set_robot_mesh (csrc/rope_meshlets.cpp) from the same host build as the stage loop: every
    triangle lands in exactly one patch within the limits, and the arrays handed to rope_set_robot are consistent
    (the shim's stand-in reads them end to end).  Under ROPE_SHIM_SANITIZE this is the sanitised run of that file."""
    rb = helpers.robot()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    shim.rope_partition_mesh.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    for l in range(rb.n_links):
        V = np.ascontiguousarray(rb.verts[rb.vtx_off[l]:rb.vtx_off[l + 1]], np.float32)
        F = np.ascontiguousarray(rb.faces[rb.tri_off[l]:rb.tri_off[l + 1]], np.int32)
        for max_t, max_v in ((128, 64), (32, 24)):
            order, first = np.empty(len(F), np.int32), np.empty(len(F) + 1, np.int32)
            m = shim.rope_partition_mesh(p(V), len(V), p(F), len(F), max_t, max_v, p(order), p(first))
            assert m > 0 and first[0] == 0 and first[m] == len(F) and np.array_equal(np.sort(order), np.arange(len(F)))
            sizes = np.diff(first[:m + 1])
            assert sizes.min() >= 1 and sizes.max() <= max_t
            assert max(len(np.unique(F[order[first[i]:first[i + 1]]])) for i in range(m)) <= max_v
    shim.rope_set_robot_mesh.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 2
    shim.shim_last_robot.restype = C.POINTER(C.c_int64)
    verts, faces = np.ascontiguousarray(rb.verts, np.float32), np.ascontiguousarray(rb.faces, np.int32)
    vo, to = np.ascontiguousarray(rb.vtx_off, np.int32), np.ascontiguousarray(rb.tri_off, np.int32)
    jf, ja = np.ascontiguousarray(rb.joint_fixed), np.ascontiguousarray(rb.joint_axes)
    ctx = C.create_string_buffer(8)                          # any non-null context: the shim's stand-ins never look inside
    assert shim.rope_set_robot_mesh(ctx, p(verts), p(faces), p(vo), p(to), rb.n_links, p(jf), p(ja)) == 0
    got = shim.shim_last_robot()
    assert got[0] == len(rb.meshlets.header) and got[1] == len(rb.meshlets.verts) and got[2] == len(rb.faces)
    bad = faces.copy()
    bad[5, 1] = 10 ** 6                                     # an index outside its link: refused, nothing read out of range
    assert shim.rope_set_robot_mesh(ctx, p(verts), p(bad), p(vo), p(to), rb.n_links, p(jf), p(ja)) == -1


def test_reference_table_aliasing_over_a_sequence_of_frames(shim):
    """The reference's Lookup stage hands out a numpy VIEW of its angle table's row (predict.py:171) and Descent steps it in
    place (predict.py:212-215): over a sequence of frames the table drifts and later frames start from the drifted rows.
    rope_predict reproduces that when given the live table (lookup_angles_live), decision by decision against the restatement
    that lets numpy do the aliasing itself — and the default (no live table) keeps frames independent."""
    rb = helpers.robot()
    lim = rb.joint_limits
    pose = DEFAULT_CAMERA_POSE
    intr, PV = helpers.camera('640_480_color', ds=8, pose=pose, as_predictor=True)
    o = helpers.make_oracle(rb, intr, PV)
    grid = helpers.slu_grid(lim, 4)
    cover = o.coverage(crop_pose_grid(lim, intr.size, 6)[0], 6, threads=THREADS) != 0
    r, c = np.where(cover)
    crop = np.array([max(r.min() - 10, 0), min(r.max() + 10, intr.height - 1), max(c.min() - 10, 0), min(c.max() + 10, intr.width - 1)], np.int32)
    names = rb.link_names
    blue_of = {n: int(LINK_BLUE[i]) for i, n in enumerate(names)}
    stages = _stages('SLU')
    arr = (StageDesc * len(stages))(*stages)
    limits, cam, inc = np.ascontiguousarray(lim, np.float64), np.asarray(pose, np.float64), np.array([.005] * 6)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    # five frames of an arm that barely moves: every frame's Lookup lands on the same grid row
    rng = np.random.default_rng(31)
    q0 = np.array([0.30, 0.25, 0.95, 0, 0, 0])
    frames = [q0 + rng.uniform(-.02, .02, 6) * np.array([1, 1, 1, 0, 0, 0]) for _ in range(5)]
    live_ref, live_nat = grid.copy(), grid.copy()
    differs_from_independent = 0
    for f, q_true in enumerate(frames):
        depth, ids = o.render(q_true, 6)
        tq, t32, flags, tgt, _, _ = helpers.synthetic_target(depth, ids)
        blue = np.where(ids == 255, 0, np.asarray(LINK_BLUE)[np.minimum(ids, 5)]).astype(np.uint8)

        @C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32))
        def answer(cand, n, n_render, loss, crop_p, err_out, best_idx):
            rows = np.ctypeslib.as_array(cand, (n, 6)).copy()
            if loss == orc.LOSS_LOOKUP:
                err = o.eval(rows, loss, n_render, tq, t32, np.ctypeslib.as_array(crop_p, (4,)), flags, threads=THREADS)
            else:
                err = o.eval(rows, loss, n_render, tq, link_flags=flags, threads=THREADS)
            if err_out:
                np.ctypeslib.as_array(err_out, (n,))[:] = err
            if best_idx:
                best_idx[0] = int(np.argmin(np.where(np.isnan(err), np.inf, err)))
            return 0
        shim.shim_set_callback(answer)
        want, trace, _ = predictor_ref.predict_reference(o, tgt, blue, names, blue_of, lim, pose, grid, crop, 'SLU', lookup_live=live_ref)
        want = np.array(want)                                        # the reference returns the view itself; keep this frame's values
        args = PredictArgs(arr, len(arr), 1 + f % 3, p(limits), p(cam), p(inc), p(grid), len(grid), 0, p(crop), p(live_nat))
        out, got_trace = np.empty(6), np.empty((len(arr), 6))
        assert shim.rope_predict(C.c_void_p(1), C.byref(args), p(out), p(got_trace), None) == 0, shim.shim_last_error()
        for k, (kind, ang) in enumerate(trace):
            assert np.array_equal(got_trace[k], ang), (f, k, kind, got_trace[k], ang)
        assert np.array_equal(out, want) and np.array_equal(live_nat, live_ref), f
        # the same frame on the default path: the grid row itself, whatever came before
        args = PredictArgs(arr, len(arr), 3, p(limits), p(cam), p(inc), p(grid), len(grid), 0, p(crop), None)
        ind, ind_trace = np.empty(6), np.empty((len(arr), 6))
        assert shim.rope_predict(C.c_void_p(1), C.byref(args), p(ind), p(ind_trace), None) == 0
        assert any(np.array_equal(ind_trace[0], row) for row in grid)
        if f == 0:
            assert np.array_equal(ind_trace, got_trace)              # a fresh table: both modes agree on the first frame
        elif not np.array_equal(ind_trace[0], got_trace[0]):
            differs_from_independent += 1
    drifted = np.nonzero((live_ref != grid).any(1))[0]
    assert len(drifted) >= 1 and differs_from_independent >= 1, (drifted, differs_from_independent)
    assert np.array_equal(live_ref[:, 3:], grid[:, 3:])               # the joints no Descent stage of 'SLU' touches stay put
