import os
import sys

import pytest

# (engine.load_library maps PyTorch's copy of the HIP runtime before librope_hip.so, so the order of imports does not matter any
# more: tests/test_gpu_runtime.py checks that in a fresh process.)

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no librope_hip.so (built artefacts are git-ignored): build it before the first test needs it.
    Building is not using: the product still refuses to run when the library is absent (test_missing_library_fails_loudly)."""
    try:
        from rope_s3d_amd import build
        if not os.path.exists(build.LIB_PATH):
            build.build()
    except Exception as e:                      # no hipcc here: the tests that need the library will say so themselves
        print(f"conftest: could not build librope_hip.so ({e})")
