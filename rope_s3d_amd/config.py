"""Explicit path/config object.

The reference resolves everything relative to the current directory through
data/paths.json and `exec` (robotpose/paths.py:18-28).  Here the same keys are
plain attributes anchored at the repository root, overridable by environment
variables or `Paths.set`, with no dependence on cwd.
"""
import os

_REPO_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir))

_DEFAULTS = {
    'DATASETS': 'data/',
    'ROBOT_LOOKUPS': 'lookup/',
    'MODELS': 'models/',
    'OUTPUT': 'output/',
    'URDFS': 'urdfs/',
    # active robot of the reference: data/paths.json:13
    'URDF': 'urdfs/motoman_mh5_support_limited/urdf/mh5l_limited.urdf',
}

_overrides = {}


class Paths:
    """Attribute view of the path table (reference: robotpose/paths.py:18-41)."""

    def __init__(self):
        for key, rel in _DEFAULTS.items():
            val = _overrides.get(key, os.environ.get('ROPE_' + key, rel))
            if not os.path.isabs(val):
                val = os.path.join(_REPO_ROOT, val)
            setattr(self, key, val)

    def create(self):
        for key in ('DATASETS', 'ROBOT_LOOKUPS', 'OUTPUT'):
            os.makedirs(getattr(self, key), exist_ok=True)

    def set(self, key: str, value: str):
        if key not in _DEFAULTS:
            raise KeyError(key)
        _overrides[key] = value.replace('\\', '/')
        self.__init__()


def repo_root() -> str:
    return _REPO_ROOT
