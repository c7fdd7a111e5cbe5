#!/usr/bin/env python3
"""Assembles tests/golden/refset_by_hand/refset_by_hand.h5 BYTE BY BYTE from the HDF5 file-format specification, without
libhdf5 or h5py: a dataset file in the reference's layout (robotpose/data/building.py:195-242) that the reader under test
(rope_s3d_amd/data/hdf5.py over libhdf5, or h5py) did not write.

It uses the structures h5py's default settings produce for that code: superblock version 0, "old-style" groups (symbol
table message -> version-1 B-tree + local heap + symbol table node), version-1 object headers, gzip-compressed chunked
datasets indexed by a version-1 B-tree (`compression="gzip"`), contiguous datasets (`preview`, `camera_poses`),
variable-length UTF-8 strings in a global heap collection both as attributes (`file.attrs['name'] = str`) and as the
elements of the gzip-chunked `paths/*` datasets (`h5py.string_dtype()`), scalar int64 / float64 attributes and the
`resolution` int64 pair, and an attribute on `coordinates/depthmaps`.

    python tests/golden/make_h5_by_hand.py          # rewrites the fixture; the frames come from default_rng(20211)

Everything is little-endian; offsets and lengths are 8 bytes.  Section numbers: HDF5 File Format Specification 2.0.
"""
import os
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K, CHUNK_K = 4, 16, 32          # superblock v0 defaults (II.A); chunk B-trees use 32


def frames(n=3, h=24, w=32):
    """The arrays the fixture holds (and the test regenerates to compare)."""
    rng = np.random.default_rng(20211)
    og = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    og[:, :, : w // 2] //= 4                                        # something for gzip to find
    depth = np.round(rng.uniform(0.4, 2.5, (n, h, w)), 3)           # float64 metres (building.py:172-179)
    angles = rng.uniform(-1.5, 1.5, (n, 6))
    positions = rng.uniform(-1, 1, (n, 6, 3))
    poses = np.tile(np.array([0, -1.5, 0.75, 0, 0, 0.0]), (n, 1))
    preview = og[:, ::4, ::4].copy()
    return og, depth, angles, positions, poses, preview


ATTRS = {'name': 'refset_by_hand', 'length': 3, 'build_date': '2021-03-01 10:00:00.000000', 'compile_date': '2021-03-01 10:00:03.500000',
         'compile_time': 3.5, 'resolution': (24, 32),
         'depth_intrinsics': '[ 32x24  p[16.1 11.9]  f[30.5 30.4]  Brown Conrady [0 0 0 0 0] ]',
         'color_intrinsics': '[ 32x24  p[16.0252 11.8644]  f[30.5764 30.5764]  Inverse Brown Conrady [0 0 0 0 0] ]',
         'depth_scale': 0.001}


class File:
    def __init__(self):
        self.b = bytearray(96)                       # the superblock goes in last
        self.gheap = []                              # (index, bytes) of the one global heap collection
        self.gheap_addr = None

    def alloc(self, data: bytes) -> int:
        while len(self.b) % 8:
            self.b.append(0)
        addr = len(self.b)
        self.b += data
        return addr

    # ---- global heap (III.E): every variable-length string lives here
    def reserve_gheap(self, size=4096):
        self.gheap_addr = self.alloc(bytes(size))
        self.gheap_size = size

    def vlen(self, text: str) -> bytes:
        """the 16-byte descriptor stored where the string 'is': length, collection address, object index"""
        raw = text.encode('utf-8')
        self.gheap.append(raw)
        return struct.pack('<IQI', len(raw), self.gheap_addr, len(self.gheap))

    def finish_gheap(self):
        out = bytearray(b'GCOL' + bytes([1, 0, 0, 0]) + struct.pack('<Q', self.gheap_size))
        for i, raw in enumerate(self.gheap, 1):
            out += struct.pack('<HHIQ', i, 1, 0, len(raw)) + raw + bytes(-len(raw) % 8)
        free = self.gheap_size - len(out)
        assert free >= 16, "global heap collection too small"
        out += struct.pack('<HHIQ', 0, 0, 0, free)               # object 0: the free space, its size includes this header
        out += bytes(self.gheap_size - len(out))
        self.b[self.gheap_addr:self.gheap_addr + self.gheap_size] = out


def pad8(b: bytes) -> bytes:
    return b + bytes(-len(b) % 8)


# ---- datatype messages (IV.A.2.d)
def dt_int(size, signed):
    return struct.pack('<BBBBI', 0x10, 0x08 if signed else 0x00, 0, 0, size) + struct.pack('<HH', 0, 8 * size)


def dt_f64():
    return struct.pack('<BBBBI', 0x11, 0x20, 63, 0, 8) + struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)


def dt_vlen_str():
    base = struct.pack('<BBBBI', 0x13, 0x10, 0, 0, 1)           # string, null-terminated, UTF-8, one byte
    return struct.pack('<BBBBI', 0x19, 0x01, 0x01, 0, 16) + base   # variable-length, of strings, UTF-8


def dataspace(shape):
    return struct.pack('<BBBBI', 1, len(shape), 0, 0, 0) + b''.join(struct.pack('<Q', d) for d in shape)   # IV.A.2.b, version 1


def message(mtype, data, flags=0):
    data = pad8(data)
    return struct.pack('<HHBBBB', mtype, len(data), flags, 0, 0, 0) + data


def attribute(f: File, name: str, value) -> bytes:
    """IV.A.2.m, version 1: name, datatype and dataspace each padded to eight bytes, then the value"""
    if isinstance(value, str):
        dt, sp, raw = dt_vlen_str(), dataspace(()), f.vlen(value)
    elif isinstance(value, float):
        dt, sp, raw = dt_f64(), dataspace(()), struct.pack('<d', value)
    elif isinstance(value, int):
        dt, sp, raw = dt_int(8, True), dataspace(()), struct.pack('<q', value)
    else:
        arr = np.asarray(value, np.int64)
        dt, sp, raw = dt_int(8, True), dataspace(arr.shape), arr.tobytes()
    nm = name.encode() + b'\0'
    body = struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(sp)) + pad8(nm) + pad8(dt) + pad8(sp) + raw
    return message(0x000C, body)


def object_header(f: File, messages) -> int:
    body = b''.join(messages)
    return f.alloc(struct.pack('<BBHII', 1, 0, len(messages), 1, len(body)) + bytes(4) + body)      # IV.A.1.a, version 1


def chunk_btree(f: File, shape, chunk, elem, chunks) -> int:
    """III.A.1, node type 1, one leaf: keys (bytes in the chunk, filter mask, offsets + 0) around the chunks' addresses"""
    rank = len(shape)
    assert len(chunks) <= 2 * CHUNK_K
    key_size = 8 + 8 * (rank + 1)
    node = bytearray(b'TREE' + struct.pack('<BBHQQ', 1, 0, len(chunks), UNDEF, UNDEF))
    for off, addr, nbytes in chunks:
        node += struct.pack('<II', nbytes, 0) + b''.join(struct.pack('<Q', o) for o in off) + struct.pack('<Q', 0)
        node += struct.pack('<Q', addr)
    last = [shape[0]] + [0] * (rank - 1)                       # the key after the last child: one chunk row past the end
    node += struct.pack('<II', 0, 0) + b''.join(struct.pack('<Q', o) for o in last) + struct.pack('<Q', 0)
    node += bytes(24 + (2 * CHUNK_K + 1) * key_size + 2 * CHUNK_K * 8 - len(node))
    return f.alloc(bytes(node))


def dataset(f: File, arr, gzip=None, attrs=None, strings=None) -> int:
    """-> object header address.  gzip: chunks of one leading index each, deflate level `gzip`; None: contiguous."""
    if strings is not None:
        shape, dt, elem = (len(strings),), dt_vlen_str(), 16
        raw_of = lambda lo, hi: b''.join(f.vlen(s) for s in strings[lo:hi])
    else:
        shape, elem = arr.shape, arr.dtype.itemsize
        dt = dt_f64() if arr.dtype == np.float64 else dt_int(arr.dtype.itemsize, arr.dtype.kind == 'i')
        raw_of = lambda lo, hi: np.ascontiguousarray(arr[lo:hi]).tobytes()
    msgs = [message(0x0001, dataspace(shape)), message(0x0003, dt, flags=1)]
    if gzip is None:
        data = raw_of(0, shape[0])
        addr = f.alloc(data)
        msgs.append(message(0x0005, bytes([2, 2, 2, 0])))                                   # fill value v2: late allocation, written if set, undefined
        msgs.append(message(0x0008, struct.pack('<BBQQ', 3, 1, addr, len(data))))           # layout v3, contiguous
    else:
        chunk = (1,) + tuple(shape[1:])
        chunks = []
        for i in range(shape[0]):
            z = zlib.compress(raw_of(i, i + 1), gzip)
            chunks.append(((i,) + (0,) * (len(shape) - 1), f.alloc(z), len(z)))
        bt = chunk_btree(f, shape, chunk, elem, chunks)
        msgs.append(message(0x0005, bytes([2, 3, 2, 0])))                                   # incremental allocation
        layout = struct.pack('<BBBQ', 3, 2, len(shape) + 1, bt) + b''.join(struct.pack('<I', c) for c in chunk) + struct.pack('<I', elem)
        msgs.append(message(0x0008, layout))
        msgs.append(message(0x000B, struct.pack('<BBHI', 1, 1, 0, 0) + struct.pack('<HHHH', 1, 0, 1, 1) + struct.pack('<II', gzip, 0)))   # one filter: deflate
    for k, v in (attrs or {}).items():
        msgs.append(attribute(f, k, v))
    return object_header(f, msgs)


def group(f: File, members: dict, attrs=None):
    """members: name -> (object header address, None | (btree, heap)).  -> (header address, btree, heap) (III.A-D)"""
    names = sorted(members)                                      # a symbol table node is ordered by name
    assert 1 <= len(names) <= 2 * LEAF_K
    heap_data, offs = bytearray(8), {}                            # offset 0: the empty string
    for n in names:
        offs[n] = len(heap_data)
        heap_data += pad8(n.encode() + b'\0')
    data_addr = f.alloc(bytes(heap_data))
    heap = f.alloc(b'HEAP' + bytes([0, 0, 0, 0]) + struct.pack('<QQQ', len(heap_data), 1, data_addr))     # free list head 1 = none
    snod = bytearray(b'SNOD' + struct.pack('<BBH', 1, 0, len(names)))
    for n in names:
        addr, sub = members[n]
        snod += struct.pack('<QQII', offs[n], addr, 1 if sub else 0, 0) + (struct.pack('<QQ', *sub) if sub else bytes(16))
    snod += bytes(8 + 2 * LEAF_K * 40 - len(snod))
    snod_addr = f.alloc(bytes(snod))
    node = bytearray(b'TREE' + struct.pack('<BBHQQ', 0, 0, 1, UNDEF, UNDEF))
    node += struct.pack('<QQQ', 0, snod_addr, offs[names[-1]])   # key 0 = "", the child, key 1 = its largest name
    node += bytes(24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(node))
    bt = f.alloc(bytes(node))
    msgs = [message(0x0011, struct.pack('<QQ', bt, heap))] + [attribute(f, k, v) for k, v in (attrs or {}).items()]
    return object_header(f, msgs), bt, heap


def build(path: str) -> str:
    og, depth, angles, positions, poses, preview = frames()
    f = File()
    f.reserve_gheap()
    n = len(og)
    jsons = [f'raw/{i:04d}.json' for i in range(n)]
    maps = [f'raw/{i:04d}_depth.npy' for i in range(n)]
    imgs = [f'raw/{i:04d}_color.png' for i in range(n)]
    coord = group(f, {'depthmaps': (dataset(f, depth, gzip=4, attrs={'depth_scale': ATTRS['depth_scale']}), None)})
    images = group(f, {'original': (dataset(f, og, gzip=4), None), 'preview': (dataset(f, preview), None),
                       'camera_poses': (dataset(f, poses), None)})
    paths = group(f, {'jsons': (dataset(f, None, gzip=4, strings=jsons), None), 'depthmaps': (dataset(f, None, gzip=4, strings=maps), None),
                      'images': (dataset(f, None, gzip=4, strings=imgs), None)})
    root, bt, heap = group(f, {'angles': (dataset(f, angles, gzip=4), None), 'positions': (dataset(f, positions, gzip=4), None),
                               'coordinates': (coord[0], coord[1:]), 'images': (images[0], images[1:]), 'paths': (paths[0], paths[1:])},
                           attrs=ATTRS)
    f.finish_gheap()
    while len(f.b) % 8:
        f.b.append(0)
    sb = b'\x89HDF\r\n\x1a\n' + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack('<HHI', LEAF_K, INTERNAL_K, 0)      # II.A, version 0
    sb += struct.pack('<QQQQ', 0, UNDEF, len(f.b), UNDEF)                                                        # base, free space, end of file, driver
    sb += struct.pack('<QQII', 0, root, 1, 0) + struct.pack('<QQ', bt, heap)                                     # the root group's symbol table entry
    assert len(sb) == 96
    f.b[:96] = sb
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'wb') as out:
        out.write(f.b)
    return path


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    p = build(os.path.join(here, 'refset_by_hand', 'refset_by_hand.h5'))
    print(p, os.path.getsize(p), 'bytes')
