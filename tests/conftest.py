import os
import sys

import pytest

try:
    # PyTorch ships its own copy of the HIP runtime.  When it is imported AFTER librope_hip.so has brought in the system copy, the
    # process holds two runtimes and torch finds no GPU; imported first, both use the one copy.  Tests that run the
    # segmentation stage and the engine in one process depend on this order whatever subset of the files is selected.
    import torch  # noqa: F401
except ImportError:
    pass

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
