#!/usr/bin/env python3
"""Builds tuning variants of librope_hip.so (experiments only): name -> extra hipcc defines."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import build
VARIANTS = {
    'base': '',
    'h48_w6': '-DROPE_TILE_H=48 -DROPE_MIN_WAVES_PER_SIMD=6',
    'h32_w6': '-DROPE_TILE_H=32 -DROPE_MIN_WAVES_PER_SIMD=6',
    'h48': '-DROPE_TILE_H=48',
}
for name, flags in VARIANTS.items():
    os.environ['ROPE_HIPCC_EXTRA'] = flags
    print(name, build.build(force=True, out_name=f'librope_hip_var_{name}.so'))
