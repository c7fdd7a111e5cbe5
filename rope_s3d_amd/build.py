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


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(_CSRC, d)) > t for d in _DEPS)


def build(force: bool = False, verbose: bool = False, out_name: str = 'librope_hip.so') -> str:
    if not force and not needs_build() and out_name == 'librope_hip.so':
        return LIB_PATH
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    extra = os.environ.get('ROPE_HIPCC_EXTRA', '').split()      # experiments only, e.g. -DROPE_SMALL_TRI_PIXELS=8
    cmd = [hipcc] + HIPCC_FLAGS + extra + _SOURCES + ['-o', out_name]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
    return os.path.join(_CSRC, out_name)


if __name__ == '__main__':
    print(build(force=True, verbose=True))
