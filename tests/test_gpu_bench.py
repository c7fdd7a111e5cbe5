"""GPU: bench.py as the driver runs it — the JSON contract, and N > 1 started by bench.py itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir))


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    d = _run(['--steps', '3', '--warmup', '1', '--cpu-sample', '256'])
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['steps'] == 3 and d['unit'] == 'poses/s' and d['scaling'] == 'weak'
    assert d['config']['candidates_per_step'] == 4096 and 'configs[1]' in d['config']['workload']
    r = d['roofline']
    assert r['bound'] == 'hbm' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert abs(r['achieved'] - r['bytes_per_candidate'] * 4096 / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-6 * r['achieved']
    assert d['unshared_value'] < d['value'] * 1.05                     # drawing six links per candidate is never the faster layout
    assert d['cpu_baseline']['gpu_errors_vs_port']['identical_bits'] is True and d['cpu_baseline']['kind'] == 'port'
    # the committed counters belong to this build (tools/summarize_prof.py stamps them with the hash of the sources): a kernel
    # edit without a new profile would make bench.py drop them and say so
    from rope_s3d_amd import build, engine
    assert r['build_id'] == build.source_hash() == engine.build_id()
    assert r['pmc_stale'] is False, "profiles/ were taken on another build: run tools/rocprof_bench.sh + tools/summarize_prof.py again"
    assert r['traffic'] and r['valu_issue'] and 0.2 < r['valu_issue']['frac'] < 1.2
    assert 1.0 < r['valu_issue']['clock_ghz_kernel_grbm'] < 2.6


def test_bench_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it: two ranks (both on this box's one GPU, collectives over gloo),
    the line says so, and the aggregate is the two ranks' work over the slower rank's time."""
    d = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--backend', 'gloo', '--no-cpu-baseline'], {'ROPE_FORCE_DEVICE': '0'})
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['config']['parallelism'] == 'frames x2'
    assert abs(d['value'] - 2 * 4096 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']


def test_cfg5_candidate_split_over_two_ranks_finds_the_single_rank_argmin():
    """BASELINE configs[4] geometry (mh50, 1280x720, 32^3 candidates of ONE frame) with the optional intra-frame split of SURVEY
    §8e: two ranks take half the candidate grid each, all-gather (joint vector, best error, best index), global argmin — the
    same row and the same error bits as the single-rank run."""
    one = _run(['--workload', 'cfg5', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-unshared'])
    two = _run(['--workload', 'cfg5', '--gpus', '2', '--split-candidates', '--steps', '2', '--warmup', '1', '--backend', 'gloo', '--no-cpu-baseline',
                '--no-unshared'], {'ROPE_FORCE_DEVICE': '0'})
    assert one['config']['candidates_per_step'] == 32768 and two['config']['candidates_per_step'] == 16384
    assert two['n_gpus'] == 2 and two['scaling'] == 'strong' and two['config']['parallelism'] == 'candidates of one frame /2'
    assert two['config']['argmin_index'] == one['config']['argmin_index']
    assert two['config']['argmin_error'] == one['config']['argmin_error']          # JSON round-trips a double exactly
