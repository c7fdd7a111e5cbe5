#!/usr/bin/env python3
"""Convert a dataset between the reference's HDF5 file and the .npy directory form.

    python tools/convert_dataset.py <dataset> h5     # <dir>/<name>.h5 next to (or from) the arrays
    python tools/convert_dataset.py <dataset> npy    # og_img.npy, depthmaps.npy, ... + attrs.json from the .h5
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd.data.dataset import Dataset, write_dataset, write_h5_dataset

name, to = sys.argv[1], sys.argv[2]
ds = Dataset(name)
known = {'name', 'length', 'resolution', 'color_intrinsics'}
extra = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in ds.attrs.items() if k not in known}
args = (ds.dataset_dir, np.asarray(ds.og_img), np.asarray(ds.depthmaps), np.asarray(ds.angles), np.asarray(ds.camera_pose), ds.intrinsics)
pos = np.asarray(ds.positions) if ds.positions is not None else None
if to == 'h5':
    print(write_h5_dataset(*args, positions=pos, extra_attrs=extra))
elif to == 'npy':
    print(write_dataset(*args, positions=pos, extra_attrs=extra))
else:
    sys.exit("target must be 'h5' or 'npy'")
