"""Closed-form pins of the CPU oracle (the reference ships no tests or fixtures for this path, so
these hand-derived expectations are what anchors the oracle: SURVEY.md §8c)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from rope_s3d_amd.constants import ZFAR, ZNEAR
from rope_s3d_amd.simulation.kinematics import ForwardKinematics

import helpers


# ------------------------------------------------------------------ kinematics
def test_sincos_within_one_ulp_of_libm():
    xs = np.concatenate([np.random.default_rng(1).uniform(-7, 7, 5000), np.linspace(-6.3, 6.3, 127), [0.0, 1e-9, -1e-9]])
    for x in xs:
        s, c = orc.sincos(float(x))
        assert abs(s - np.sin(x)) <= np.spacing(abs(np.sin(x))) and abs(c - np.cos(x)) <= np.spacing(abs(np.cos(x)))


def test_fk_home_pose_closed_form():
    """q = 0: link origins are the running sums of the URDF joint origins
    (mh5l_limited.urdf:119-160): link_5_b at (0.088+0.405, 0, 0.330+0.400+0.040)."""
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=8)
    fk = helpers.make_oracle(rb, intr, PV).fk(np.zeros(6)).reshape(7, 3, 4)
    want = np.array([[0, 0, 0], [0, 0, .33], [.088, 0, .33], [.088, 0, .73], [.088, 0, .77], [.493, 0, .77], [.573, 0, .77]])
    assert np.allclose(fk[:, :, 3], want, atol=1e-15)
    for l in range(7):
        assert np.array_equal(fk[l, :, :3], np.eye(3))


def test_fk_quarter_turns_closed_form():
    """S = +90 deg about +z carries link x-offsets into +y; U axis is -y so U = +90 deg lifts the forearm up."""
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=8)
    o = helpers.make_oracle(rb, intr, PV)
    fk = o.fk([np.pi / 2, 0, 0, 0, 0, 0]).reshape(7, 3, 4)
    assert np.allclose(fk[5][:, 3], [0, 0.493, 0.77], atol=1e-15)
    fk = o.fk([0, 0, np.pi / 2, 0, 0, 0]).reshape(7, 3, 4)
    # forearm (0.405 along x of link_4) now points along +z: link_5_b = link_3_u origin + R*(0,0,.04) + R*(.405,0,0)
    assert np.allclose(fk[5][:, 3], [0.088 - 0.04, 0, 0.73 + 0.405], atol=1e-15)


def test_fk_matches_independent_numpy_chain():
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=8)
    o = helpers.make_oracle(rb, intr, PV)
    host = ForwardKinematics()
    for q in np.random.default_rng(3).uniform(-2, 2, (20, 6)):
        ref = host.calc(q)[:, :3, :]
        assert np.allclose(o.fk(q).reshape(7, 3, 4), ref, atol=1e-14)


# ------------------------------------------------------------------ rasteriser on analytic geometry
W, H = 64, 48


def _flat_scene(tris_by_link):
    """Camera at the origin looking down -Z with fx=fy=64, cx=32, cy=24: a point (x,y,-2) lands at
    window (32x+32, 32y+24).  Every 'link' is a list of triangles in world space."""
    P = np.zeros((4, 4))
    P[0, 0], P[1, 1] = 2 * 64 / W, 2 * 64 / H
    P[0, 2], P[1, 2] = 1 - 2 * 32 / W, 2 * 24 / H - 1
    P[2, 2], P[2, 3], P[3, 2] = (ZFAR + ZNEAR) / (ZNEAR - ZFAR), 2 * ZFAR * ZNEAR / (ZNEAR - ZFAR), -1
    verts, faces, voff, toff = [], [], [0], [0]
    for tris in tris_by_link:
        v = np.array(tris, np.float32).reshape(-1, 3)
        verts.append(v)
        faces.append(np.arange(len(v), dtype=np.int32).reshape(-1, 3))
        voff.append(voff[-1] + len(v))
        toff.append(toff[-1] + len(v) // 3)
    fixed = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], float), (6, 1))
    axes = np.tile(np.array([0, 0, 1.0]), (6, 1))
    return orc.Oracle(np.concatenate(verts), np.concatenate(faces), voff, toff, fixed, axes, P, W, H, ZNEAR, ZFAR)


def _win(sx, sy, z=-2.0):
    """world point that projects to window coordinates (sx, sy) at depth z"""
    return [(sx - 32) / 64 * -z, (sy - 24) / 64 * -z, z]


def _quad(x0, y0, x1, y1, z=-2.0):
    a, b, c, d = _win(x0, y0, z), _win(x1, y0, z), _win(x1, y1, z), _win(x0, y1, z)
    return [a, b, c], [a, c, d]          # CCW seen from the camera


def test_pixel_centre_sampling_and_image_flip():
    """A quad with edges on pixel boundaries covers exactly the pixels inside; window y is up, image row 0 is the top."""
    t1, t2 = _quad(24, 16, 40, 40)
    o = _flat_scene([[t1, t2]])
    d, ids = o.render(np.zeros(6), 1)
    cov = ids == 0
    assert cov.sum() == 16 * 24
    rows, cols = np.where(cov)
    assert (cols.min(), cols.max()) == (24, 39)
    assert (rows.min(), rows.max()) == (H - 1 - 39, H - 1 - 16)


def test_top_left_rule_no_double_hits_no_holes():
    """Edges through pixel centres: the two triangles of a quad partition it, and the quad owns its
    left/top boundary samples but not the right/bottom ones (y up: 'top' = larger window y)."""
    t1, t2 = _quad(24.5, 16.5, 40.5, 40.5)
    a = _flat_scene([[t1]]).render(np.zeros(6), 1)[1] == 0
    b = _flat_scene([[t2]]).render(np.zeros(6), 1)[1] == 0
    assert not (a & b).any()
    both = a | b
    assert both.sum() == 16 * 24
    rows, cols = np.where(both)
    assert (cols.min(), cols.max()) == (24, 39)                    # left edge sx=24.5 owns px 24; right edge sx=40.5 does not own px 40
    assert (H - 1 - rows.max(), H - 1 - rows.min()) == (17, 40)    # top edge sy=40.5 owns py 40; bottom edge sy=16.5 does not own py 16


def test_back_faces_are_culled():
    t1, t2 = _quad(24, 16, 40, 40)
    flipped = [[t1[0], t1[2], t1[1]], [t2[0], t2[2], t2[1]]]
    assert (_flat_scene([flipped]).render(np.zeros(6), 1)[1] == 255).all()


def test_nearer_surface_wins_and_depth_is_metric():
    near = _quad(20, 10, 36, 30, z=-1.5)
    far = _quad(28, 20, 50, 44, z=-3.0)
    o = _flat_scene([list(far), list(near)])             # far quad is link 0, near quad is link 1
    d, ids = o.render(np.zeros(6), 2)
    overlap = np.zeros((H, W), bool)
    overlap[H - 30:H - 20, 28:36] = True
    assert (ids[overlap] == 1).all()
    assert np.abs(d[ids == 1] - 1.5).max() < 5e-6 and np.abs(d[ids == 0] - 3.0).max() < 2e-5
    assert (d[ids == 255] == 0).all()


def test_near_plane_clipping_against_an_analytic_floor():
    """OpenGL clips primitives against the view volume before the viewport transform (pyrender draws through GL; znear 0.05 m,
    projection.py:161-169): a floor quad that runs from half a metre BEHIND the camera to four metres in front of it must
    still cover every pixel whose ray meets it beyond the near plane, at the ray's metric depth.  Without clipping both of
    its triangles would vanish (a vertex behind the eye has no window position)."""
    y0 = -0.3
    a, b, c, d = [-1, y0, 0.5], [1, y0, 0.5], [1, y0, -4.0], [-1, y0, -4.0]          # normal +y: seen from above, CCW
    o = _flat_scene([[[a, b, c], [a, c, d]]])
    depth, ids = o.render(np.zeros(6), 1)
    px, py = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)                     # window coordinates, y up
    dx, dy = (px - 32) / 64, (py - 24) / 64                                          # ray (dx, dy, -1)
    with np.errstate(divide='ignore', invalid='ignore'):
        t = np.where(dy < 0, y0 / dy, np.inf)                                        # distance along -z where the ray meets the floor
    inside = (t > ZNEAR) & (t < 4.0) & (np.abs(dx * t) < 1.0)
    # pixels whose ray passes within a hair of a boundary of the visible region are left out of the comparison
    edge = (np.abs(t - ZNEAR) < 1e-3) | (np.abs(t - 4.0) < 2e-2) | (np.abs(np.abs(dx * t) - 1.0) < 2e-2)
    got = (ids == 0)[::-1]                                                            # image row 0 is the top: flip to y up
    z = depth[::-1]
    assert inside.sum() > 500
    assert np.array_equal(got[~edge], inside[~edge])
    m = inside & ~edge
    assert np.abs(z[m] - t[m]).max() < 2e-4 * 4.0
    # both triangles have a vertex behind the eye: the whole visible floor, bottom row of the image included, exists only
    # because they are cut at the plane rather than dropped
    assert got[0].all() and got.sum() > 1000
    # the same floor seen from below is back-facing after clipping as well
    assert (_flat_scene([[[a, c, b], [a, d, c]]]).render(np.zeros(6), 1)[1] == 255).all()


def test_equal_depth_lower_link_id_wins():
    q = _quad(24, 16, 40, 40)
    d, ids = _flat_scene([list(q), list(q)]).render(np.zeros(6), 2)
    assert set(np.unique(ids)) == {0, 255}


def test_depth_readback_is_pyrender_float32_formula():
    """z = 2nf / (f+n - (2d-1)(f-n)) evaluated in float32 as numpy does in pyrender's _read_main_framebuffer."""
    d24 = np.array([0, 1, 12345, 8388608, 16220000, 16777214], np.uint32)
    key = (d24 << 8) | 3
    o = _flat_scene([[_quad(24, 16, 40, 40)[0]]])
    got, ids = o.resolve(key.reshape(1, -1))
    di = d24.astype(np.float32) / np.float32(16777215.0)
    di = np.float32(2.0) * di - np.float32(1.0)
    want = np.float32(2.0 * ZNEAR * ZFAR) / (np.float32(ZFAR + ZNEAR) - di * np.float32(ZFAR - ZNEAR))
    assert np.array_equal(got.reshape(-1).view(np.uint32), want.astype(np.float32).view(np.uint32))
    assert (ids == 3).all()


# ------------------------------------------------------------------ loss terms against numpy
@pytest.fixture(scope='module')
def frame():
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=4)
    o = helpers.make_oracle(rb, intr, PV)
    d_t, id_t = o.render([0.35, 0.45, 0.9, 0, 0, 0])
    d_r, id_r = o.render([0.30, 0.50, 1.0, 0, 0, 0])
    return rb, o, d_t, id_t, d_r, id_r


@pytest.mark.parametrize('n', [4, 6])
def test_error_matches_numpy_restatement_of_reference(frame, n):
    """Fixed-point sums + float64 finalise vs Predictor._error written out in numpy (predict.py:475-509)."""
    rb, o, d_t, id_t, d_r, id_r = frame
    tq, t32, flags, tgt, masks, masked = helpers.synthetic_target(d_t, id_t)
    key = o.raster_key([0.30, 0.50, 1.0, 0, 0, 0], n)
    d_r, id_r = o.resolve(key)
    got = o.finalize(o.sums(key, orc.LOSS_FULL, n, tq), orc.LOSS_FULL, n, d_t.size, flags)
    names = rb.link_names
    blue = {nme: i for i, nme in enumerate(names)}
    want = orc.error_numpy(n, id_r, d_r, tgt, {names[l]: masked[l] for l in masked}, {names[l]: masks[l] for l in masks}, names, blue)
    assert abs(got - want) <= 1e-9 * abs(want)


def test_depth_only_loss_is_last_term(frame):
    rb, o, d_t, id_t, d_r, id_r = frame
    tq, *_ = helpers.synthetic_target(d_t, id_t)
    key = o.raster_key([0.30, 0.50, 1.0, 0, 0, 0], 6)
    got = o.finalize(o.sums(key, orc.LOSS_DEPTH, 6, tq), orc.LOSS_DEPTH, 6, d_t.size, np.zeros(8, np.uint8))
    diff = np.abs(d_t.astype(np.float64) - d_r)
    want = np.mean(diff[diff != 0]) * np.std(diff)
    assert abs(got - want) <= 1e-9 * want


def test_identical_render_gives_nan_like_numpy(frame):
    """D == 0 everywhere: mean of an empty selection is NaN in the reference too (predict.py:507)."""
    rb, o, d_t, id_t, *_ = frame
    tq, *_ = helpers.synthetic_target(d_t, id_t)
    key = o.raster_key([0.35, 0.45, 0.9, 0, 0, 0], 6)
    assert np.isnan(o.finalize(o.sums(key, orc.LOSS_DEPTH, 6, tq), orc.LOSS_DEPTH, 6, d_t.size, np.zeros(8, np.uint8)))


def test_lookup_score_matches_float32_numpy(frame):
    """mean|T - sqrt(D)| * std|T - sqrt(D)| over the crop with T not sqrt-ed (predict.py:117,167-169)."""
    rb, o, d_t, id_t, d_r, id_r = frame
    tq, t32, flags, *_ = helpers.synthetic_target(d_t, id_t)
    crop = np.array([20, 119, 30, 150], np.int32)
    key = o.raster_key([0.30, 0.50, 1.0, 0, 0, 0], 6)
    n = (crop[1] - crop[0] + 1) * (crop[3] - crop[2] + 1)
    got = o.finalize(o.sums(key, orc.LOSS_LOOKUP, 6, None, t32, crop), orc.LOSS_LOOKUP, 6, n, flags)
    sl = (slice(crop[0], crop[1] + 1), slice(crop[2], crop[3] + 1))
    want = orc.lookup_score_numpy(t32[sl], d_r[sl][None])[0]
    assert abs(got - want) <= 1e-6 * want


def test_tensor_sweep_score_sign_quirk(frame):
    """predict.py:367: `mean *- std` -> the score is NEGATIVE mean*std."""
    rb, o, d_t, id_t, d_r, id_r = frame
    key = o.raster_key([0.30, 0.50, 1.0, 0, 0, 0], 6)
    got = o.finalize(o.sums(key, orc.LOSS_TSWEEP, 6, None, d_t), orc.LOSS_TSWEEP, 6, d_t.size, np.zeros(8, np.uint8))
    diff = np.abs(np.sqrt(d_t) - np.sqrt(d_r)).astype(np.float64)
    assert got < 0 and abs(got + diff.mean() * diff.std()) <= 1e-6 * abs(got)


def test_eval_batch_threads_agree(frame):
    rb, o, d_t, id_t, *_ = frame
    tq, t32, flags, *_ = helpers.synthetic_target(d_t, id_t)
    cand = helpers.slu_grid(rb.joint_limits, 2)
    a = o.eval(cand, orc.LOSS_FULL, 6, tq, t32, None, flags, threads=1)
    b = o.eval(cand, orc.LOSS_FULL, 6, tq, t32, None, flags, threads=4)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_oracle_rasteriser_against_the_independent_exact_one():
    """tests/golden/pins_raster_160x120.npz: the REAL mh5l mesh at three poses through an independent rasteriser in Python integers
    and exact rationals (tests/golden/make_pins.py: hand-rounded float32 vertex shading, brute-force edge functions per pixel with
    the top-left rule, exact barycentric depth).  The C oracle must cover exactly the same pixels, give every pixel the same link,
    and its 24-bit depth (a float32 plane, quantised) must sit within two units of the exact value."""
    pins = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pins_raster_160x120.npz'))
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=4)
    o = helpers.make_oracle(rb, intr, PV)
    for k, q in enumerate(pins['poses']):
        key = o.raster_key(q, 6)
        ids = np.where(key == 0xFFFFFFFF, 255, key & 0xFF).astype(np.uint8)
        want_ids, want_d24, near = pins[f'ids{k}'], pins[f'd24_{k}'], pins[f'near{k}']
        assert (want_ids != 255).sum() > 1000
        assert np.array_equal(ids != 255, want_ids != 255), f"pose {k}: coverage differs"
        assert np.array_equal(ids[~near], want_ids[~near]), f"pose {k}: link ids differ"
        drawn = want_ids != 255
        assert np.abs((key >> 8).astype(np.int64) - want_d24.astype(np.int64))[drawn & ~near].max() <= 2
        # and the metric depth read-back of those keys is pyrender's formula on the quantised value
        depth, _ = o.resolve(key)
        d = want_d24[drawn].astype(np.float64) / (2 ** 24 - 1)
        z = 2 * 0.05 * 100.0 / (100.0 + 0.05 - (2 * d - 1) * (100.0 - 0.05))
        assert np.abs(depth[drawn] - z).max() < 2e-6 * z.max() + 4 * 2.0 ** -24 * z.max() ** 2 / 0.05


def test_box_morphology_against_scipy_ndimage():
    """imgproc.dilate / imgproc.erode (cv2.dilate / cv2.erode with a ones(k, k) kernel, default anchor (k // 2, k // 2) — for an even
    k the window reaches one sample further up/left than down/right — and a border that never wins) against scipy.ndimage's
    maximum_filter / minimum_filter as a second implementation: the body mask's dilate 8 / erode 7 (predict.py:428,437) and every
    kernel size NoiseMaker.holes uses (3..24, noise.py:20,26).  cv2 itself is not installed anywhere this runs: what this pins is
    that the window geometry written down in imgproc.py is computed correctly, twice."""
    from scipy import ndimage
    from rope_s3d_amd import imgproc
    rng = np.random.default_rng(9)
    for k in list(range(3, 25)) + [7, 8]:
        for shape in ((45, 80), (31, 33)):
            img = (rng.uniform(size=shape) > 0.93).astype(np.float64) * rng.uniform(0.5, 2.0, shape)
            # scipy's window for size k and origin o covers [x - k // 2 - o, x + (k - 1) // 2 - o]; OpenCV's anchor k // 2 covers
            # [x - k // 2, x + k - 1 - k // 2]: the same window for odd AND even k with origin 0
            want_d = ndimage.maximum_filter(img, size=k, mode='constant', cval=-np.inf, origin=0)
            want_e = ndimage.minimum_filter(img, size=k, mode='constant', cval=np.inf, origin=0)
            assert np.array_equal(imgproc.dilate(img, k), want_d), k
            assert np.array_equal(imgproc.erode(img, k), want_e), k
            # hand check of the geometry on one sample, away from the borders
            y, x, a = shape[0] // 2, shape[1] // 2, k // 2
            if y - a >= 0 and x - a >= 0 and y - a + k <= shape[0] and x - a + k <= shape[1]:
                assert want_d[y, x] == img[y - a:y - a + k, x - a:x - a + k].max()
    # the body mask of the segmentation path: erode7(dilate8(.)) of a blob keeps the blob and closes holes up to 7 wide
    blob = np.zeros((40, 60))
    blob[10:30, 15:45] = 1
    blob[18:21, 25:31] = 0
    body = imgproc.erode(imgproc.dilate(blob, 8), 7)
    assert body[18:21, 25:31].all() and np.array_equal(body, ndimage.minimum_filter(ndimage.maximum_filter(blob, 8, mode='constant', cval=-np.inf), 7,
                                                                                    mode='constant', cval=np.inf))

