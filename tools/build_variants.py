#!/usr/bin/env python3
"""Builds tuning variants of librope_hip.so (experiments only): name -> extra hipcc defines."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import build
VARIANTS = {
    'c4r4': '',
    'c8r4': '-DROPE_SMALL_TRI_COLS=8',
    'c8r8': '-DROPE_SMALL_TRI_COLS=8 -DROPE_SMALL_TRI_ROWS=8',
    'c4r8': '-DROPE_SMALL_TRI_ROWS=8',
    'c2r2': '-DROPE_SMALL_TRI_COLS=2 -DROPE_SMALL_TRI_ROWS=2',
}
for name, flags in VARIANTS.items():
    os.environ['ROPE_HIPCC_EXTRA'] = flags
    print(name, build.build(force=True, out_name=f'librope_hip_var_{name}.so'))
