"""AddressSanitizer + UBSan over the PRODUCT's host C++ (csrc/rope_predict.cpp: the stage loop; csrc/rope_meshlets.cpp: the
meshlet builder) — CPU build only, GPU sanitizers are not available on the pool.  tests/test_native_host.py is run once more
in a child process whose shim library is built with -fsanitize=address,undefined and which has the sanitiser runtime preloaded
(the interpreter itself is not instrumented)."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir))


def test_host_cpp_under_asan_ubsan():
    libasan = subprocess.run(['g++', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "no libasan next to g++"
    env = dict(os.environ, ROPE_SHIM_SANITIZE='1', LD_PRELOAD=libasan, OMP_NUM_THREADS='1',
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    # two stage-list runs (speculative and serial order) and the partitioner / robot builder: the code paths, not the seeds
    sel = 'test_partitioner_and_robot_builder_in_the_host_build or (test_stage_loop_against_sequential_reference and (7919 or 7921))'
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_native_host.py'), '-x', '-q', '-k', sel,
                          '-p', 'no:cacheprovider'], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert 'AddressSanitizer' not in out.stderr and 'runtime error' not in out.stderr, out.stderr[-4000:]
    assert ' passed' in out.stdout and 'failed' not in out.stdout
