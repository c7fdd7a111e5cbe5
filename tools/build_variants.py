#!/usr/bin/env python3
"""Builds tuning variants of librope_hip.so (experiments only): name -> extra hipcc defines."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import build
VARIANTS = {
    'profile': '-DROPE_PROFILE',       # librope_hip_profile.so: the phase-skipping switches of tools/profile_phases.py
    'base': '',
    'c2r2': '-DROPE_SMALL_TRI_COLS=2 -DROPE_SMALL_TRI_ROWS=2',
    'c4r2': '-DROPE_SMALL_TRI_COLS=4 -DROPE_SMALL_TRI_ROWS=2',
    'c2r4': '-DROPE_SMALL_TRI_COLS=2 -DROPE_SMALL_TRI_ROWS=4',
    'c3r3': '-DROPE_SMALL_TRI_COLS=3 -DROPE_SMALL_TRI_ROWS=3',
    'c6r6': '-DROPE_SMALL_TRI_COLS=6 -DROPE_SMALL_TRI_ROWS=6',
}
for name, flags in VARIANTS.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    os.environ['ROPE_HIPCC_EXTRA'] = flags
    print(name, build.build(force=True, out_name='librope_hip_profile.so' if name == 'profile' else f'librope_hip_var_{name}.so'))
