#!/usr/bin/env python3
"""Builds tuning variants of librope_hip.so (experiments only): name -> extra hipcc defines."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import build
VARIANTS = {
    'full4': '',
    'full6': '-DROPE_MIN_WAVES_FULL=6',
}
for name, flags in VARIANTS.items():
    os.environ['ROPE_HIPCC_EXTRA'] = flags
    print(name, build.build(force=True, out_name=f'librope_hip_var_{name}.so'))
