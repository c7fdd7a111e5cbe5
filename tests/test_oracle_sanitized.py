"""AddressSanitizer + UBSan run of the oracle's C code (CPU build only: GPU sanitizers are not available on the pool)."""
import os
import subprocess

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir, 'oracle')


def test_oracle_selftest_under_asan_ubsan():
    subprocess.check_call(['make', '-C', HERE, 'selftest'], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1')
    out = subprocess.run([os.path.join(HERE, 'selftest')], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'oracle selftest ok: 384 samples' in out.stdout
