#!/usr/bin/env python3
"""Builds tuning variants of librope_hip.so (experiments only): name -> extra hipcc defines."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import build
VARIANTS = {
    'base': '',
    'v64_h96_w12': '-DROPE_MESHLET_MAX_VERTS=64 -DROPE_TILE_H=96 -DROPE_NWAVES=12',
    'v64_w256_w12': '-DROPE_MESHLET_MAX_VERTS=64 -DROPE_TILE_W=256 -DROPE_NWAVES=12',
    'v64_w192_h64_w12': '-DROPE_MESHLET_MAX_VERTS=64 -DROPE_TILE_W=192 -DROPE_TILE_H=64 -DROPE_NWAVES=12',
    'v64_w160_h72_w12': '-DROPE_MESHLET_MAX_VERTS=64 -DROPE_TILE_W=160 -DROPE_TILE_H=72 -DROPE_NWAVES=12',
}
for name, flags in VARIANTS.items():
    os.environ['ROPE_HIPCC_EXTRA'] = flags
    print(name, build.build(force=True, out_name=f'librope_hip_var_{name}.so'))
