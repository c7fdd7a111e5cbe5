"""examples/predict_frame.c: a plain-C host on the C ABI alone.  CPU: it builds against include/rope_s3d.h and
librope_hip.so and fails loudly without a GPU.  GPU: its joint angles equal Predictor.run's, digit for digit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE

import helpers

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def exe(tmp_path_factory):
    from rope_s3d_amd import build
    build.build()
    out = str(tmp_path_factory.mktemp('chost') / 'predict_frame')
    csrc = os.path.join(ROOT, 'rope_s3d_amd', 'csrc')
    subprocess.check_call(['gcc', '-O2', '-Wall', '-Werror', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'predict_frame.c'),
                           '-L' + csrc, '-lrope_hip', '-Wl,-rpath,' + csrc, '-Wl,-rpath-link,/opt/rocm/lib', '-lm', '-o', out])
    return out


def test_c_host_builds_and_refuses_to_run_without_a_gpu(exe, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    rb = helpers.robot()
    files = {'verts.f32': rb.verts.astype(np.float32), 'faces.i32': rb.faces.astype(np.int32), 'vtx_off.i32': rb.vtx_off.astype(np.int32),
             'tri_off.i32': rb.tri_off.astype(np.int32), 'joint_fixed.f64': rb.joint_fixed, 'joint_axes.f64': rb.joint_axes,
             'limits.f64': rb.joint_limits, 'camera_pose.f64': np.asarray(DEFAULT_CAMERA_POSE, float),
             'intrinsics.f64': np.array([64, 48, 32.05, 23.7, 61.15, 61.15]), 'setup.i32': np.array([4, 3], np.int32),
             'link_blue.i32': np.array([0, 42, 85, 127, 170, 212], np.int32), 'color.u8': np.zeros((48, 64, 3), np.uint8),
             'depth.f32': np.zeros((48, 64), np.float32)}
    for name, a in files.items():
        np.ascontiguousarray(a).tofile(tmp_path / name)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and 'rope_create' in r.stderr and r.stdout == ''          # no device: says so, computes nothing
    (tmp_path / 'depth.f32').write_bytes(b'\0' * 8)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and 'does not match' in r.stderr


@pytest.mark.gpu
def test_c_host_predicts_the_same_angles_as_the_python_host(exe, tmp_path):
    from dump_frame_bundle import dump_frame_bundle
    from rope_s3d_amd import SyntheticPredictor
    sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '640_480_color', 4, 'SLU', noise=False, seed=5, lookup_divisions=5)
    p = sp.predictor
    lim = helpers.robot().joint_limits
    for seed in range(3):
        q = np.random.default_rng(300 + seed).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
        sp.renderer.setJointAngles(q)
        color, depth = sp.renderer.render()
        want = p.run(color, depth)
        # the bundle holds inputs only — meshes, chain, limits, camera pose, base intrinsics, the frame: no matrix, crop or grid
        bundle = dump_frame_bundle(str(tmp_path / f'bundle{seed}'), p, color, depth, '640_480_color', 5)
        assert not any(os.path.exists(os.path.join(bundle, f)) for f in ('PV.f64', 'crop.i32', 'grid.f64', 'tq.u64', 't32.f32'))
        r = subprocess.run([exe, bundle], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = np.array([float(x) for x in r.stdout.split()])
        assert np.array_equal(got, want), (got, want)
        c = p.lookup_crop
        assert f'crop {c[0]} {c[1]} {c[2]} {c[3]}, {len(p.lookup_angles)} lookup poses' in r.stderr
        if seed == 0:
            # the same frame 2 x 20 times through the lockstep path from C: page-locked planes, the second group staged while the
            # first is predicted (rope_stage_targets / rope_commit_targets / rope_predict_batch)
            r = subprocess.run([exe, bundle, '20'], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr
            assert '40 frames in two lockstep batches: every one equal to the single frame' in r.stderr
            assert np.array_equal(np.array([float(x) for x in r.stdout.split()]), want)
