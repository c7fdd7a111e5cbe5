"""The reference's on-disk dataset format: one gzip-chunked HDF5 file per set (robotpose/data/building.py:195-242)
with `angles`, `positions`, `coordinates/depthmaps`, `images/{original,preview,camera_poses}`, `paths/*` and the
file attributes `name, length, resolution, color_intrinsics, depth_intrinsics, depth_scale, ...`, read by
robotpose/data/dataset.py:176-192.

h5py is the reference's binding and is used when importable.  Without it the HDF5 C library itself is driven through
ctypes (any libhdf5 1.10+: `ROPE_LIBHDF5`, the loader path, or the usual install prefixes), which is what this module
implements: lazy datasets that read hyperslabs of the leading axis — `np.copy(ds.og_img[a:b])` of
predict_dataset.py:39-41 decompresses only those frames — plus a writer of the same layout (export / tests)."""
import ctypes as C
import ctypes.util
import glob
import os
import threading

import numpy as np

_HID = C.c_int64
_HSZ = C.c_uint64
_LIB = None
_LOCK = threading.RLock()          # libhdf5 is only thread-safe when built so: one reader at a time

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_VARIABLE = C.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5S_SCALAR = 0


class HDF5Unavailable(RuntimeError):
    pass


def _candidates():
    if os.environ.get('ROPE_LIBHDF5'):
        yield os.environ['ROPE_LIBHDF5']
    found = ctypes.util.find_library('hdf5')
    if found:
        yield found
    for pat in ('/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*', '/usr/lib/x86_64-linux-gnu/libhdf5.so*', '/usr/lib64/libhdf5.so*',
                '/usr/local/lib/libhdf5.so*', '/opt/conda/lib/libhdf5.so*'):
        for p in sorted(glob.glob(pat)):
            yield p


def lib():
    """The HDF5 C library, loaded once.  Raises HDF5Unavailable naming what was tried."""
    global _LIB
    if _LIB is not None:
        return _LIB
    tried = []
    for path in _candidates():
        try:
            h = C.CDLL(path)
            h.H5open()
        except (OSError, AttributeError):
            tried.append(path)
            continue
        _declare(h)
        _LIB = h
        return h
    raise HDF5Unavailable("reading the reference's .h5 datasets needs h5py or the HDF5 C library (libhdf5); set ROPE_LIBHDF5 to "
                          f"its path.  Tried: {tried or 'nothing found'}")


def available() -> bool:
    try:
        lib()
        return True
    except HDF5Unavailable:
        return False


def _declare(h):
    sig = {
        'H5Fopen': (_HID, [C.c_char_p, C.c_uint, _HID]), 'H5Fcreate': (_HID, [C.c_char_p, C.c_uint, _HID, _HID]),
        'H5Fclose': (C.c_int, [_HID]),
        'H5Gcreate2': (_HID, [_HID, C.c_char_p, _HID, _HID, _HID]), 'H5Gclose': (C.c_int, [_HID]),
        'H5Gopen2': (_HID, [_HID, C.c_char_p, _HID]),
        'H5Lexists': (C.c_int, [_HID, C.c_char_p, _HID]),
        'H5Dopen2': (_HID, [_HID, C.c_char_p, _HID]), 'H5Dclose': (C.c_int, [_HID]),
        'H5Dcreate2': (_HID, [_HID, C.c_char_p, _HID, _HID, _HID, _HID, _HID]),
        'H5Dget_space': (_HID, [_HID]), 'H5Dget_type': (_HID, [_HID]),
        'H5Dread': (C.c_int, [_HID, _HID, _HID, _HID, _HID, C.c_void_p]),
        'H5Dwrite': (C.c_int, [_HID, _HID, _HID, _HID, _HID, C.c_void_p]),
        'H5Dvlen_reclaim': (C.c_int, [_HID, _HID, _HID, C.c_void_p]),
        'H5Sget_simple_extent_ndims': (C.c_int, [_HID]),
        'H5Sget_simple_extent_dims': (C.c_int, [_HID, C.POINTER(_HSZ), C.POINTER(_HSZ)]),
        'H5Sselect_hyperslab': (C.c_int, [_HID, C.c_int, C.POINTER(_HSZ), C.POINTER(_HSZ), C.POINTER(_HSZ), C.POINTER(_HSZ)]),
        'H5Screate_simple': (_HID, [C.c_int, C.POINTER(_HSZ), C.POINTER(_HSZ)]), 'H5Screate': (_HID, [C.c_int]),
        'H5Sclose': (C.c_int, [_HID]),
        'H5Tget_class': (C.c_int, [_HID]), 'H5Tget_size': (C.c_size_t, [_HID]), 'H5Tget_sign': (C.c_int, [_HID]),
        'H5Tis_variable_str': (C.c_int, [_HID]), 'H5Tcopy': (_HID, [_HID]), 'H5Tset_size': (C.c_int, [_HID, C.c_size_t]),
        'H5Tset_cset': (C.c_int, [_HID, C.c_int]), 'H5Tclose': (C.c_int, [_HID]),
        'H5Aopen': (_HID, [_HID, C.c_char_p, _HID]), 'H5Aclose': (C.c_int, [_HID]),
        'H5Aget_type': (_HID, [_HID]), 'H5Aget_space': (_HID, [_HID]), 'H5Aread': (C.c_int, [_HID, _HID, C.c_void_p]),
        'H5Acreate2': (_HID, [_HID, C.c_char_p, _HID, _HID, _HID, _HID]), 'H5Awrite': (C.c_int, [_HID, _HID, C.c_void_p]),
        'H5Aiterate2': (C.c_int, [_HID, C.c_int, C.c_int, C.POINTER(_HSZ), C.c_void_p, C.c_void_p]),
        'H5Pcreate': (_HID, [_HID]), 'H5Pset_chunk': (C.c_int, [_HID, C.c_int, C.POINTER(_HSZ)]),
        'H5Pset_deflate': (C.c_int, [_HID, C.c_uint]), 'H5Pclose': (C.c_int, [_HID]),
        'H5Pset_create_intermediate_group': (C.c_int, [_HID, C.c_uint]),
        'H5Eset_auto2': (C.c_int, [_HID, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(h, name)
        f.restype, f.argtypes = res, args
    h.H5Eset_auto2(0, None, None)             # no error stack on stderr: failures become Python exceptions here
    it = getattr(h, 'H5Literate', None) or getattr(h, 'H5Literate1')       # 1.12+ keeps the 1.10 form under this name
    it.restype, it.argtypes = C.c_int, [_HID, C.c_int, C.c_int, C.POINTER(_HSZ), C.c_void_p, C.c_void_p]
    h._iterate_links = it


def _glob(name: str) -> int:
    return _HID.in_dll(lib(), name).value


_NATIVE = {np.dtype(np.uint8): 'H5T_NATIVE_UINT8_g', np.dtype(np.int8): 'H5T_NATIVE_INT8_g',
           np.dtype(np.uint16): 'H5T_NATIVE_UINT16_g', np.dtype(np.int16): 'H5T_NATIVE_INT16_g',
           np.dtype(np.uint32): 'H5T_NATIVE_UINT32_g', np.dtype(np.int32): 'H5T_NATIVE_INT32_g',
           np.dtype(np.uint64): 'H5T_NATIVE_UINT64_g', np.dtype(np.int64): 'H5T_NATIVE_INT64_g',
           np.dtype(np.float32): 'H5T_NATIVE_FLOAT_g', np.dtype(np.float64): 'H5T_NATIVE_DOUBLE_g',
           np.dtype(np.bool_): 'H5T_NATIVE_UINT8_g'}


def _native(dtype) -> int:
    return _glob(_NATIVE[np.dtype(dtype)])


def _np_dtype(tid: int):
    h = lib()
    cls, size = h.H5Tget_class(tid), h.H5Tget_size(tid)
    if cls == H5T_FLOAT:
        return np.dtype({4: np.float32, 8: np.float64}.get(size, np.float64))      # half floats widen on read
    if cls == H5T_INTEGER:
        return np.dtype(('i' if h.H5Tget_sign(tid) else 'u') + str(size))
    if cls == H5T_STRING:
        return np.dtype(object)
    raise TypeError(f"HDF5 type class {cls} is not used by the dataset format")


def _dims(sid: int) -> tuple:
    h = lib()
    n = h.H5Sget_simple_extent_ndims(sid)
    if n <= 0:
        return ()
    d = (_HSZ * n)()
    h.H5Sget_simple_extent_dims(sid, d, None)
    return tuple(int(v) for v in d)


def _read_strings(read, tid: int, sid: int, n: int):
    """n strings (variable- or fixed-length) through `read(memtype, buffer)`."""
    h = lib()
    if h.H5Tis_variable_str(tid) > 0:
        buf = (C.c_char_p * n)()
        if read(tid, buf) < 0:
            raise IOError("HDF5 string read failed")
        out = [(b or b'').decode('utf-8', 'replace') for b in buf]
        h.H5Dvlen_reclaim(tid, sid, 0, buf)
        return out
    size = h.H5Tget_size(tid)
    raw = C.create_string_buffer(size * n)
    if read(tid, raw) < 0:
        raise IOError("HDF5 string read failed")
    return [raw.raw[i * size:(i + 1) * size].split(b'\0')[0].decode('utf-8', 'replace') for i in range(n)]


class H5Array:
    """A dataset of an open file: shape, dtype, len() and numpy-style indexing; the leading-axis part of an index
    (an integer or a unit-step slice) selects the hyperslab that is read, the rest is applied to the result."""

    def __init__(self, file, path: str):
        h = lib()
        self._file, self.path = file, path
        self._id = h.H5Dopen2(file._id, path.encode(), 0)
        if self._id < 0:
            raise KeyError(f"no dataset '{path}' in {file.filename}")
        sid, tid = h.H5Dget_space(self._id), h.H5Dget_type(self._id)
        self.shape, self.dtype = _dims(sid), _np_dtype(tid)
        h.H5Sclose(sid)
        h.H5Tclose(tid)

    def __len__(self) -> int:
        if not self.shape:
            raise TypeError("len() of a scalar dataset")
        return self.shape[0]

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def _read(self, start: int, count: int) -> np.ndarray:
        """Rows [start, start+count) of the leading axis (the whole value of a scalar dataset)."""
        with _LOCK:
            return self._read_locked(start, count)

    def _read_locked(self, start: int, count: int) -> np.ndarray:
        h = lib()
        shape = (count,) + self.shape[1:] if self.shape else ()
        if self.shape and (count == 0 or 0 in self.shape):
            return np.empty(shape, self.dtype)
        fspace = mspace = 0                                     # H5S_ALL
        if self.shape:
            rank = len(self.shape)
            st, ct = (_HSZ * rank)(start, *([0] * (rank - 1))), (_HSZ * rank)(*shape)
            fspace = h.H5Dget_space(self._id)
            if h.H5Sselect_hyperslab(fspace, 0, st, None, ct, None) < 0:
                h.H5Sclose(fspace)
                raise IOError(f"{self.path}: hyperslab selection failed")
            mspace = h.H5Screate_simple(rank, ct, None)
        try:
            if self.dtype == object:
                tid = h.H5Dget_type(self._id)
                n = int(np.prod(shape)) if shape else 1
                sid = mspace if mspace else h.H5Dget_space(self._id)
                try:
                    vals = _read_strings(lambda t, b: h.H5Dread(self._id, t, mspace, fspace, 0, b), tid, sid, n)
                finally:
                    h.H5Tclose(tid)
                    if not mspace:
                        h.H5Sclose(sid)
                out = np.empty(n, object)
                out[:] = vals
                return out.reshape(shape) if shape else out[0]
            out = np.empty(shape, self.dtype)
            if h.H5Dread(self._id, _native(self.dtype), mspace, fspace, 0, out.ctypes.data_as(C.c_void_p)) < 0:
                raise IOError(f"{self.path}: read failed (a filter this libhdf5 lacks?)")
            return out if shape else out[()]
        finally:
            if mspace:
                h.H5Sclose(mspace)
            if fspace:
                h.H5Sclose(fspace)

    def __getitem__(self, key):
        if not self.shape:
            return self._read(0, 1)
        lead, rest = (key[0], key[1:]) if isinstance(key, tuple) and key else (key, ())
        if lead is Ellipsis or (isinstance(key, tuple) and not key):
            lead, rest = slice(None), ()
        n = self.shape[0]
        if isinstance(lead, (int, np.integer)):
            i = int(lead) + (n if lead < 0 else 0)
            if not 0 <= i < n:
                raise IndexError(f"index {lead} out of range for axis 0 with size {n}")
            out = self._read(i, 1)[0]
            return out[rest] if rest else out
        if isinstance(lead, slice):
            a, b, step = lead.indices(n)
            if step == 1:
                out = self._read(a, max(b - a, 0))
                return out[(slice(None),) + rest] if rest else out
        return self._read(0, n)[key]                # fancy or strided index: read, then let numpy do it

    def __array__(self, dtype=None, copy=None):
        out = self._read(0, self.shape[0] if self.shape else 1)
        return out.astype(dtype) if dtype is not None else out

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __repr__(self):
        return f'<HDF5 dataset "{self.path}": shape {self.shape}, type {self.dtype}>'

    def close(self):
        if self._id >= 0:
            lib().H5Dclose(self._id)
            self._id = -1


_ATTR_CB = C.CFUNCTYPE(C.c_int, _HID, C.c_char_p, C.c_void_p, C.c_void_p)


class H5File:

    def __init__(self, filename: str, mode: str = 'r'):
        h = lib()
        self.filename = filename
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        self._id = h.H5Fopen(filename.encode(), H5F_ACC_RDONLY if mode == 'r' else H5F_ACC_RDWR, 0)
        if self._id < 0:
            raise IOError(f"{filename} is not an HDF5 file (or cannot be opened '{mode}')")
        self._open = []

    def __contains__(self, path: str) -> bool:
        h, cur = lib(), ''
        for part in path.strip('/').split('/'):
            cur = f'{cur}/{part}'
            if h.H5Lexists(self._id, cur.encode(), 0) <= 0:
                return False
        return True

    def __getitem__(self, path: str) -> H5Array:
        a = H5Array(self, path)
        self._open.append(a)
        return a

    def walk(self, group: str = '/'):
        """Paths of every dataset below `group`, depth first in name order."""
        h, names = lib(), []

        def visit(loc, name, info, data):
            names.append(name.decode())
            return 0
        gid = h.H5Gopen2(self._id, group.encode(), 0)
        if gid < 0:
            raise KeyError(f"no group '{group}' in {self.filename}")
        cb = _ATTR_CB(visit)
        h._iterate_links(gid, 0, 0, None, C.cast(cb, C.c_void_p), None)
        h.H5Gclose(gid)
        base = group.rstrip('/')
        for n in names:
            path = f'{base}/{n}'
            did = h.H5Dopen2(self._id, path.encode(), 0)
            if did >= 0:
                h.H5Dclose(did)
                yield path.lstrip('/')
            else:
                yield from self.walk(path)

    @property
    def attrs(self) -> dict:
        """File (root group) attributes as Python values: str, int, float or numpy arrays."""
        h, names = lib(), []

        def visit(loc, name, info, data):
            names.append(name)
            return 0
        cb = _ATTR_CB(visit)
        h.H5Aiterate2(self._id, 0, 0, None, C.cast(cb, C.c_void_p), None)      # H5_INDEX_NAME, H5_ITER_INC
        return {n.decode(): self._attr(n) for n in names}

    def _attr(self, name: bytes):
        h = lib()
        aid = h.H5Aopen(self._id, name, 0)
        tid, sid = h.H5Aget_type(aid), h.H5Aget_space(aid)
        try:
            shape = _dims(sid)
            n = int(np.prod(shape)) if shape else 1
            dt = _np_dtype(tid)
            if dt == object:
                vals = _read_strings(lambda t, b: h.H5Aread(aid, t, b), tid, sid, n)
                return vals[0] if not shape else np.array(vals, object).reshape(shape)
            wide = np.float64 if dt.kind == 'f' else (np.int64 if dt.kind == 'i' else np.uint64)
            out = np.empty(n, wide)
            if h.H5Aread(aid, _native(wide), out.ctypes.data_as(C.c_void_p)) < 0:
                raise IOError(f"attribute {name!r}: read failed")
            if shape:
                return out.reshape(shape)
            return float(out[0]) if dt.kind == 'f' else int(out[0])
        finally:
            h.H5Sclose(sid)
            h.H5Tclose(tid)
            h.H5Aclose(aid)

    def close(self):
        for a in self._open:
            a.close()
        self._open = []
        if self._id >= 0:
            lib().H5Fclose(self._id)
            self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------ writer
def _write_attr(loc: int, name: str, value):
    h = lib()
    if isinstance(value, str):
        tid = h.H5Tcopy(_glob('H5T_C_S1_g'))
        h.H5Tset_size(tid, H5T_VARIABLE)
        h.H5Tset_cset(tid, H5T_CSET_UTF8)
        sid = h.H5Screate(H5S_SCALAR)
        aid = h.H5Acreate2(loc, name.encode(), tid, sid, 0, 0)
        buf = (C.c_char_p * 1)(value.encode())
        rc = h.H5Awrite(aid, tid, buf)
        h.H5Tclose(tid)
    else:
        arr = np.asarray(value)
        arr = np.array(arr, dtype=np.float64 if arr.dtype.kind == 'f' else np.int64, order='C')     # keeps 0-d: a scalar dataspace, as h5py writes
        if arr.ndim:
            dims = (_HSZ * arr.ndim)(*arr.shape)
            sid = h.H5Screate_simple(arr.ndim, dims, None)
        else:
            sid = h.H5Screate(H5S_SCALAR)
        aid = h.H5Acreate2(loc, name.encode(), _native(arr.dtype), sid, 0, 0)
        rc = h.H5Awrite(aid, _native(arr.dtype), arr.ctypes.data_as(C.c_void_p))
    h.H5Sclose(sid)
    h.H5Aclose(aid)
    if rc < 0:
        raise IOError(f"attribute {name}: write failed")


def _write_array(fid: int, path: str, data: np.ndarray, gzip: int = None):
    h = lib()
    data = np.ascontiguousarray(data)
    rank = data.ndim
    dims = (_HSZ * rank)(*data.shape)
    sid = h.H5Screate_simple(rank, dims, None)
    dcpl = 0
    if gzip is not None and data.size:
        dcpl = h.H5Pcreate(_glob('H5P_CLS_DATASET_CREATE_ID_g'))
        chunk = (_HSZ * rank)(1, *data.shape[1:])            # one frame per chunk: a slice of frames inflates only those
        h.H5Pset_chunk(dcpl, rank, chunk)
        h.H5Pset_deflate(dcpl, int(gzip))
    lcpl = h.H5Pcreate(_glob('H5P_CLS_LINK_CREATE_ID_g'))          # groups on the way are made as needed
    h.H5Pset_create_intermediate_group(lcpl, 1)
    did = h.H5Dcreate2(fid, path.encode(), _native(data.dtype), sid, lcpl, dcpl, 0)
    h.H5Pclose(lcpl)
    rc = h.H5Dwrite(did, _native(data.dtype), 0, 0, 0, data.ctypes.data_as(C.c_void_p)) if (did >= 0 and data.size) else (0 if did >= 0 else -1)
    if dcpl:
        h.H5Pclose(dcpl)
    if did >= 0:
        h.H5Dclose(did)
    h.H5Sclose(sid)
    if rc < 0:
        raise IOError(f"{path}: write failed")


def write_h5_dataset(path: str, og_img, depthmaps, angles, camera_pose, color_intrinsics: str, positions=None, preview=None,
                     attrs: dict = None, compression_level: int = 4) -> str:
    """One dataset file in the reference's layout (building.py:195-242; `paths/*` are ingest bookkeeping and left out)."""
    h = lib()
    og_img, depthmaps = np.asarray(og_img, np.uint8), np.asarray(depthmaps, np.float64)
    n = len(og_img)
    if os.path.exists(path):
        os.remove(path)
    fid = h.H5Fcreate(path.encode(), H5F_ACC_TRUNC, 0, 0)
    if fid < 0:
        raise IOError(f"cannot create {path}")
    try:
        meta = {'name': os.path.splitext(os.path.basename(path))[0], 'length': n, 'resolution': np.array(og_img.shape[1:3]),
                'color_intrinsics': str(color_intrinsics), 'depth_intrinsics': str(color_intrinsics), 'depth_scale': 1.0}
        meta.update(attrs or {})
        for k, v in meta.items():
            _write_attr(fid, k, v)
        for g in ('coordinates', 'images'):
            h.H5Gclose(h.H5Gcreate2(fid, g.encode(), 0, 0, 0))
        _write_array(fid, 'angles', np.asarray(angles, np.float64), compression_level)
        _write_array(fid, 'positions', np.zeros((n, 6, 3)) if positions is None else np.asarray(positions, np.float64), compression_level)
        _write_array(fid, 'coordinates/depthmaps', depthmaps, compression_level)
        _write_array(fid, 'images/original', og_img, compression_level)
        _write_array(fid, 'images/preview', og_img[:, ::10, ::10] if preview is None else np.asarray(preview, np.uint8))
        _write_array(fid, 'images/camera_poses', np.asarray(camera_pose, np.float64))
    finally:
        h.H5Fclose(fid)
    return path


def write_arrays(path: str, arrays: dict, attrs: dict = None, gzip: int = None) -> str:
    """A plain HDF5 file of named arrays ({'group/sub/name': ndarray}); groups are created as needed."""
    h = lib()
    if os.path.exists(path):
        os.remove(path)
    fid = h.H5Fcreate(path.encode(), H5F_ACC_TRUNC, 0, 0)
    if fid < 0:
        raise IOError(f"cannot create {path}")
    try:
        for k, v in (attrs or {}).items():
            _write_attr(fid, k, v)
        for name, data in arrays.items():
            _write_array(fid, name, np.asarray(data), gzip)
    finally:
        h.H5Fclose(fid)
    return path
