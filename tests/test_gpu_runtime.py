"""GPU: one HIP runtime per process, whatever the import order, and RCCL actually executing (world size 1)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir))


def _run(code, timeout=600, **env):
    e = dict(os.environ, PYTHONPATH=ROOT, **env)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        if k not in env:
            e.pop(k, None)
    return subprocess.run([sys.executable, '-c', textwrap.dedent(code)], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_engine_first_then_torch_share_one_runtime():
    """The product's own order: Predictor builds its engine (librope_hip.so) and only then imports the Mask R-CNN module (torch).
    In a fresh process, without any earlier torch import, torch must still see the GPU, run on it, and hand its stream to the
    library's segmentation kernels."""
    r = _run('''
        import sys
        from rope_s3d_amd import engine as eng
        e = eng.Engine(0)                               # librope_hip.so first
        assert 'torch' not in sys.modules
        import torch
        assert torch.cuda.is_available(), "torch lost the GPU: two HIP runtimes in the process"
        x = torch.arange(8, device='cuda', dtype=torch.float32)
        assert float((x * 2).sum().item()) == 56.0
        maps = open('/proc/self/maps').read()
        copies = sorted({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l})
        assert len(copies) == 1, copies
        # the default segmenter path of Predictor.__init__: a Mask R-CNN on cuda, after the engine
        from rope_s3d_amd.maskrcnn import MaskRCNNSegmenter
        seg = MaskRCNNSegmenter(7, device='cuda:0', min_confidence=0.0)
        import numpy as np
        out = seg(np.zeros((90, 160, 3), np.uint8))
        assert out['masks'].shape[:2] == (90, 160)
        print('ok', copies[0])
    ''')
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout + r.stderr


def test_rccl_runs_with_one_rank(tmp_path):
    """backend 'nccl' (= RCCL) with a world of one: init_process_group on the device, the all-gather of a CUDA tensor in
    parallel.gather_rows, barrier and teardown all execute — in bench.py and in predict_dataset.py, as the driver's launcher
    starts them (RANK / WORLD_SIZE / MASTER_* in the environment)."""
    import json
    import socket
    from rope_s3d_amd.data.dataset import make_synthetic_dataset
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               ROPE_DIST_ALWAYS='1')
    b = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
                        '--no-unshared'], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stdout + b.stderr
    line = json.loads(b.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['collective'] == 'nccl' and line['value'] > 1e5
    d = make_synthetic_dataset(str(tmp_path / 'synth6'), 6, base_intrin='640_480_color', seed=4100)
    env['MASTER_PORT'] = str(port + 1)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'predict_dataset.py'), d, '-ds_factor', '4'], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert 'gathered over nccl' in p.stdout
    with_rccl = np.load(tmp_path / 'predictions_synth6.npy')
    env.pop('ROPE_DIST_ALWAYS')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k)
    os.remove(tmp_path / 'predictions_synth6.npy')
    q = subprocess.run([sys.executable, os.path.join(ROOT, 'predict_dataset.py'), d, '-ds_factor', '4'], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stdout + q.stderr
    assert np.array_equal(with_rccl, np.load(tmp_path / 'predictions_synth6.npy')) and with_rccl.shape == (6, 6)
