"""Robot geometry as the engine consumes it.

Host-side preparation only (runs once per URDF): welded link meshes concatenated
into one shared vertex/face buffer (what the reference holds as six pyrender
meshes, robotpose/simulation/render_utils.py:22-41 — the seventh, link_6_t, is
dropped there at :31-32), the fixed part of every joint transform (what Klampt
reads from <origin>/<axis>, kinematics.py:25-27), and the meshlet partition the
HIP rasteriser walks.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np

from .constants import NUM_RENDER_LINKS
from .stl import load_mesh
from .urdf import URDFReader

import os as _os

# Measured on the bench workload: region-grown patches of <=128 triangles over <=64 vertices
# (one vertex-shading iteration per wave, 1.8 cm median radius) beat Morton runs of 128/128 by 8 %.
MESHLET_MAX_TRIS = int(_os.environ.get('ROPE_MESHLET_TRIS', 128))      # engine limit: 64 (rope_kernels.h)
MESHLET_MAX_VERTS = int(_os.environ.get('ROPE_MESHLET_VERTS', 64))     # engine limit: 64 (rope_kernels.h)


def _rpy_matrix(rpy) -> np.ndarray:
    """URDF fixed-axis roll/pitch/yaw -> R = Rz(yaw)·Ry(pitch)·Rx(roll)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _morton3(q: np.ndarray) -> np.ndarray:
    """Interleave three 10-bit integer columns into a 30-bit Morton code."""
    def spread(v):
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


@dataclass
class Meshlets:
    """Spatially compact groups of <=128 triangles over <=64 private vertices (engine limits: 128 / 128).

    header  (M,8) uint32 words: centre x,y,z and radius (float32 bit patterns),
            first vertex, first triangle, (n_verts | n_tris<<16), link id
    verts   (Vm,3) float32   meshlet-local vertex copies
    tris    (Tm,)  uint32    three 8-bit local indices packed i0 | i1<<8 | i2<<16
    link_first (n_links+1,)  meshlet range of every link
    """
    header: np.ndarray
    verts: np.ndarray
    tris: np.ndarray
    link_first: np.ndarray


def _grow_partition(V: np.ndarray, F: np.ndarray, hard_t: int, hard_v: int, soft_margin: int = 4, absorb_below: int = 24):
    """Region growing over the triangle adjacency graph: compact patches (about half the radius of Morton runs)."""
    import heapq
    max_t, max_v = hard_t, max(hard_v - soft_margin, 3)
    T = len(F)
    cent = V[F].astype(np.float64).mean(1)
    e = np.sort(np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]]), axis=1)
    fid = np.tile(np.arange(T), 3)
    key = e[:, 0].astype(np.int64) * (V.shape[0] + 1) + e[:, 1]
    order = np.argsort(key, kind='stable')
    ks, fs = key[order], fid[order]
    same = ks[1:] == ks[:-1]
    nbr = [[] for _ in range(T)]
    for x, y in zip(fs[:-1][same].tolist(), fs[1:][same].tolist()):
        nbr[x].append(y)
        nbr[y].append(x)
    visited = np.zeros(T, bool)
    lo = cent.min(0)
    span = np.maximum(cent.max(0) - lo, 1e-9)
    seeds = np.argsort(_morton3(np.minimum(((cent - lo) / span * 1023).astype(np.int64), 1023)), kind='stable').tolist()
    sp, clusters, carry = 0, [], []
    while True:
        seed = None
        while carry:
            _, t = heapq.heappop(carry)
            if not visited[t]:
                seed = t
                break
        if seed is None:
            while sp < T and visited[seeds[sp]]:
                sp += 1
            if sp >= T:
                break
            seed = seeds[sp]
        c0 = cent[seed]
        tris, verts, heap, inheap = [], set(), [(0.0, seed)], {seed}
        while heap and len(tris) < max_t:
            _, t = heapq.heappop(heap)
            if visited[t]:
                continue
            nv = verts | set(F[t].tolist())
            if len(nv) > max_v:
                continue
            verts = nv
            tris.append(t)
            visited[t] = True
            for n in nbr[t]:
                if not visited[n] and n not in inheap:
                    inheap.add(n)
                    dd = cent[n] - c0
                    heapq.heappush(heap, (float(dd @ dd), n))
        carry = [(d, t) for d, t in heap if not visited[t]]
        heapq.heapify(carry)
        clusters.append(tris)
    # Growth stops a little short of the vertex limit (soft cap) and strands small islands between finished
    # patches; fold every island into the neighbouring patch it shares most vertices with, up to the hard caps.
    owner = np.empty(T, np.int64)
    for ci, tris in enumerate(clusters):
        owner[tris] = ci
    vsets = [set(F[tris].ravel().tolist()) for tris in clusters]
    alive = [True] * len(clusters)
    for ci in sorted(range(len(clusters)), key=lambda i: len(clusters[i])):
        if len(clusters[ci]) >= absorb_below:
            break
        best, best_shared = -1, 0
        for cj in {int(owner[n]) for t in clusters[ci] for n in nbr[t]} - {ci}:
            if not alive[cj] or len(clusters[cj]) + len(clusters[ci]) > hard_t:
                continue
            shared = len(vsets[ci] & vsets[cj])
            if len(vsets[ci]) + len(vsets[cj]) - shared <= hard_v and shared > best_shared:
                best, best_shared = cj, shared
        if best >= 0:
            clusters[best] = clusters[best] + clusters[ci]
            vsets[best] |= vsets[ci]
            owner[clusters[ci]] = best
            alive[ci] = False
    return [np.array(c) for c, a in zip(clusters, alive) if a]


def _native_partition(V: np.ndarray, F: np.ndarray, hard_t: int, hard_v: int):
    """The same region growing in the library (csrc/rope_meshlets.cpp, rope_partition_mesh): host code, no GPU."""
    import ctypes as C
    from .engine import load_library
    lib = load_library()
    V = np.ascontiguousarray(V, np.float32)
    F = np.ascontiguousarray(F, np.int32)
    order, first = np.empty(len(F), np.int32), np.empty(len(F) + 1, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    m = lib.rope_partition_mesh(p(V), len(V), p(F), len(F), hard_t, hard_v, p(order), p(first))
    if m < 0:
        raise ValueError("rope_partition_mesh rejected the mesh (indices outside the vertices, or limits beyond 128 / 64)")
    return [order[first[i]:first[i + 1]].astype(np.int64) for i in range(m)]


def build_meshlets(link_verts: List[np.ndarray], link_faces: List[np.ndarray]) -> Meshlets:
    """ROPE_MESHLET_BUILDER: 'native' (default: region growing in the library, ~50 ms), 'grow' (the same in Python,
    ~2 s), 'morton' (runs of the Morton order).  The image does not depend on the choice."""
    builder = _os.environ.get('ROPE_MESHLET_BUILDER', 'native')
    if builder in ('native', 'grow'):
        part = _native_partition if builder == 'native' else _grow_partition
        return _build_from_partitions(link_verts, link_faces,
                                      [part(V, F, MESHLET_MAX_TRIS, MESHLET_MAX_VERTS) for V, F in zip(link_verts, link_faces)])
    headers, vpool, tpool, link_first = [], [], [], [0]
    v_base = t_base = 0
    for link, (V, F) in enumerate(zip(link_verts, link_faces)):
        cent = V[F].astype(np.float64).mean(axis=1)
        lo, hi = cent.min(0), cent.max(0)
        span = np.where(hi > lo, hi - lo, 1.0)
        code = _morton3(np.minimum(((cent - lo) / span * 1023.0).astype(np.int64), 1023))
        order = np.argsort(code, kind='stable')
        Fs = F[order]
        start = 0
        nF = len(Fs)
        while start < nF:
            # largest prefix of the remaining faces with <=128 tris and <=128 distinct vertices
            stop = min(start + MESHLET_MAX_TRIS, nF)
            uniq = np.unique(Fs[start:stop])
            while len(uniq) > MESHLET_MAX_VERTS:
                stop -= max(1, (len(uniq) - MESHLET_MAX_VERTS + 1) // 2)
                uniq = np.unique(Fs[start:stop])
            local = np.searchsorted(uniq, Fs[start:stop]).astype(np.uint32)
            pts = V[uniq]
            p64 = pts.astype(np.float64)
            c = (p64.min(0) + p64.max(0)) * 0.5
            rad = float(np.sqrt(((p64 - c) ** 2).sum(1).max())) * (1.0 + 1e-5) + 1e-7
            hdr = np.zeros(8, np.uint32)
            hdr[0:4] = np.array([c[0], c[1], c[2], rad], np.float32).view(np.uint32)
            hdr[4], hdr[5] = v_base, t_base
            hdr[6] = len(uniq) | ((stop - start) << 16)
            hdr[7] = link
            headers.append(hdr)
            vpool.append(pts)
            tpool.append(local[:, 0] | (local[:, 1] << 8) | (local[:, 2] << 16))
            v_base += len(uniq)
            t_base += stop - start
            start = stop
        link_first.append(len(headers))
    return Meshlets(np.ascontiguousarray(np.stack(headers)),
                    np.ascontiguousarray(np.concatenate(vpool).astype(np.float32)),
                    np.ascontiguousarray(np.concatenate(tpool).astype(np.uint32)),
                    np.array(link_first, np.int32))


def _build_from_partitions(link_verts, link_faces, partitions) -> Meshlets:
    headers, vpool, tpool, link_first = [], [], [], [0]
    v_base = t_base = 0
    for link, (V, F, parts) in enumerate(zip(link_verts, link_faces, partitions)):
        for tri_ids in parts:
            Fs = F[tri_ids]
            uniq = np.unique(Fs)
            local = np.searchsorted(uniq, Fs).astype(np.uint32)
            pts = V[uniq]
            p64 = pts.astype(np.float64)
            c = (p64.min(0) + p64.max(0)) * 0.5
            rad = float(np.sqrt(((p64 - c) ** 2).sum(1).max())) * (1.0 + 1e-5) + 1e-7
            hdr = np.zeros(8, np.uint32)
            hdr[0:4] = np.array([c[0], c[1], c[2], rad], np.float32).view(np.uint32)
            hdr[4], hdr[5] = v_base, t_base
            hdr[6] = len(uniq) | (len(Fs) << 16)
            hdr[7] = link
            headers.append(hdr)
            vpool.append(pts)
            tpool.append(local[:, 0] | (local[:, 1] << 8) | (local[:, 2] << 16))
            v_base += len(uniq)
            t_base += len(Fs)
        link_first.append(len(headers))
    return Meshlets(np.ascontiguousarray(np.stack(headers)), np.ascontiguousarray(np.concatenate(vpool).astype(np.float32)),
                    np.ascontiguousarray(np.concatenate(tpool).astype(np.uint32)), np.array(link_first, np.int32))


@dataclass
class RobotModel:
    name: str
    link_names: List[str]
    joint_limits: np.ndarray            # (6,2)
    joint_fixed: np.ndarray             # (6,12) row-major 3x4: parent link frame -> joint frame
    joint_axes: np.ndarray              # (6,3) unit
    verts: np.ndarray                   # (V,3) float32, links concatenated
    faces: np.ndarray                   # (T,3) int32, indices local to the link
    vtx_off: np.ndarray                 # (n_links+1,) int32
    tri_off: np.ndarray                 # (n_links+1,) int32
    meshlets: Meshlets = field(repr=False, default=None)

    @property
    def n_links(self) -> int:
        return len(self.vtx_off) - 1

    @staticmethod
    def from_urdf(reader: URDFReader = None, n_links: int = NUM_RENDER_LINKS) -> 'RobotModel':
        reader = reader or URDFReader()
        lv, lf = [], []
        for path in reader.mesh_paths[:n_links]:
            v, f = load_mesh(path)
            lv.append(v)
            lf.append(f)
        fixed = np.zeros((6, 12))
        axes = np.zeros((6, 3))
        for i in range(6):
            A = np.zeros((3, 4))
            A[:, :3] = _rpy_matrix(reader.joint_rpy[i]) if np.any(reader.joint_rpy[i]) else np.eye(3)
            A[:, 3] = reader.joint_origins[i]
            fixed[i] = A.reshape(-1)
            a = reader.joint_axes[i]
            axes[i] = a / np.linalg.norm(a)
        return RobotModel(
            name=reader.name, link_names=list(reader.mesh_names[:n_links]),
            joint_limits=reader.joint_limits.copy(), joint_fixed=fixed, joint_axes=axes,
            verts=np.ascontiguousarray(np.concatenate(lv)), faces=np.ascontiguousarray(np.concatenate(lf)),
            vtx_off=np.cumsum([0] + [len(v) for v in lv]).astype(np.int32),
            tri_off=np.cumsum([0] + [len(f) for f in lf]).astype(np.int32),
            meshlets=build_meshlets(lv, lf))
