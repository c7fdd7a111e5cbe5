"""The Dataset reader on a file it did not write: tests/golden/refset_by_hand/refset_by_hand.h5 is assembled byte by byte
from the HDF5 file-format specification (tests/golden/make_h5_by_hand.py — no libhdf5, no h5py) in the reference's layout
(robotpose/data/building.py:195-242) and with the structures h5py's defaults produce for it: superblock 0, symbol-table
groups, version-1 object headers, gzip-chunked and contiguous datasets, variable-length UTF-8 strings (attributes and
`paths/*`) in a global heap.  h5py itself is not installed here, so this is the independent writer that can exist offline."""
import os
import sys

import numpy as np
import pytest

from rope_s3d_amd.data import hdf5

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_h5_by_hand as byhand  # noqa: E402

FIXTURE = os.path.join(HERE, 'golden', 'refset_by_hand')

pytestmark = pytest.mark.skipif(not hdf5.available(), reason="no libhdf5 on this machine")


def _check(ds):
    og, depth, angles, positions, poses, preview = byhand.frames()
    assert ds.length == 3 and len(ds) == 3
    assert ds.intrinsics == byhand.ATTRS['color_intrinsics']
    for k, v in byhand.ATTRS.items():
        got = ds.attrs[k]
        assert (list(got) == list(v)) if k == 'resolution' else (got == v), k
    assert list(ds.og_resolution) == [24, 32]
    for name, want in (('og_img', og), ('depthmaps', depth), ('angles', angles), ('positions', positions), ('camera_pose', poses),
                       ('preview_img', preview)):
        arr = getattr(ds, name)
        assert tuple(arr.shape) == want.shape
        got = np.asarray(arr[:])
        assert got.dtype == want.dtype and np.array_equal(got, want), name
    # the access pattern of predict_dataset.py:39-41
    assert np.array_equal(np.copy(ds.og_img[1:3]), og[1:3]) and np.array_equal(np.copy(ds.depthmaps[0:2]), depth[0:2])
    assert np.array_equal(np.copy(ds.camera_pose[2:3]), poses[2:3]) and np.array_equal(ds.og_img[2], og[2])
    assert [str(s) for s in ds.file['paths/images'][:]] == [f'raw/{i:04d}_color.png' for i in range(3)]


def test_reader_on_the_hand_assembled_reference_layout_file():
    from rope_s3d_amd.data.dataset import Dataset
    from rope_s3d_amd.projection import Intrinsics
    ds = Dataset(FIXTURE)
    try:
        _check(ds)
        intr = Intrinsics(ds.intrinsics)                     # the RealSense string the reference stores (projection.py:47-78)
        assert (intr.width, intr.height) == (32, 24)
    finally:
        ds.close()


def test_the_committed_fixture_is_what_the_script_assembles(tmp_path):
    """Same content from a fresh assembly (the deflate streams may differ between zlib builds, so the comparison is made
    through the reader, not on the bytes), and the fixed-layout part of the file — superblock, signature — byte for byte."""
    from rope_s3d_amd.data.dataset import Dataset
    fresh = byhand.build(str(tmp_path / 'refset_by_hand' / 'refset_by_hand.h5'))
    ds = Dataset(os.path.dirname(fresh))
    try:
        _check(ds)
    finally:
        ds.close()
    a, b = open(fresh, 'rb').read(), open(os.path.join(FIXTURE, 'refset_by_hand.h5'), 'rb').read()
    assert a[:8] == b[:8] == b'\x89HDF\r\n\x1a\n' and a[8:40] == b[8:40]
