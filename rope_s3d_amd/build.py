"""Builds librope_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import shutil
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB_PATH = os.path.join(_CSRC, 'librope_hip.so')
_SOURCES = ['rope_kernels.hip', 'rope_abi.hip', 'rope_seg.hip', 'rope_predict.cpp', 'rope_meshlets.cpp']
_DEPS = _SOURCES + ['rope_kernels.h', os.path.join('..', '..', 'include', 'rope_s3d.h')]

# -ffp-contract=off: the arithmetic contract with the CPU oracle is "one IEEE operation per
# written step"; fused operations appear only where the source says fmaf.
HIPCC_FLAGS = ['-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-shared', '-std=c++17']


def source_hash() -> str:
    """Identity of what the library is built from: SHA-256 over the sources and headers (first 16 hex digits).  Compiled into the
    library (rope_build_id) and written into the profiles (tools/summarize_prof.py), so that bench.py can tell whether the
    committed counters belong to the build it is timing."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(_DEPS):
        h.update(d.encode())
        with open(os.path.join(_CSRC, d), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _built_id(path: str) -> str:
    try:
        with open(path + '.id') as f:
            return f.read().strip()
    except OSError:
        return ''


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = _built_id(LIB_PATH)
    if built:                                           # the library says what it was built from: time stamps do not survive a copy
        return built != source_hash()
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(_CSRC, d)) > t for d in _DEPS)


def build(force: bool = False, verbose: bool = False, out_name: str = 'librope_hip.so') -> str:
    if not force and not needs_build() and out_name == 'librope_hip.so':
        return LIB_PATH
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    extra = os.environ.get('ROPE_HIPCC_EXTRA', '').split()      # experiments only, e.g. -DROPE_SMALL_TRI_PIXELS=8
    build_id = source_hash()
    cmd = [hipcc] + HIPCC_FLAGS + [f'-DROPE_BUILD_ID="{build_id}"'] + extra + _SOURCES + ['-o', out_name]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
    with open(os.path.join(_CSRC, out_name) + '.id', 'w') as f:         # beside the library (git-ignored like it): needs_build reads it
        f.write(build_id + '\n')
    return os.path.join(_CSRC, out_name)


if __name__ == '__main__':
    print(build(force=True, verbose=True))
