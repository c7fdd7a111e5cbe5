"""Binary STL reader with vertex weld.

Restates what the reference gets from `trimesh.load(file)` (trimesh 3.9.8, called at
robotpose/simulation/render_utils.py:40): triangle soup -> `merge_vertices` with the
default tolerance (positions rounded to 1e-8, first occurrence kept, face order
unchanged).  Only positions and faces are kept: the SEG render mode the
prediction path uses ignores normals (render.py:94-98).
"""
import numpy as np

_MERGE_DIGITS = 8   # trimesh.constants.tol.merge = 1e-8


def read_binary_stl(path: str):
    """Return (triangles (T,3,3) float32, attr (T,) uint16) of a binary STL file."""
    with open(path, 'rb') as f:
        raw = f.read()
    if len(raw) < 84:
        raise ValueError(f"{path}: too short for a binary STL")
    n = int(np.frombuffer(raw, dtype='<u4', count=1, offset=80)[0])
    if len(raw) != 84 + 50 * n:
        raise ValueError(f"{path}: size {len(raw)} does not match {n} facets (ASCII STL is not supported)")
    rec = np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')])
    data = np.frombuffer(raw, dtype=rec, count=n, offset=84)
    return np.ascontiguousarray(data['v']), np.ascontiguousarray(data['a'])


def weld(triangles: np.ndarray):
    """(T,3,3) float32 soup -> (vertices (V,3) float32, faces (T,3) int32).

    Merge key = round(position * 1e8) as int64; unique vertices are kept in order
    of first appearance, exactly one output face per input facet.
    """
    soup = triangles.reshape(-1, 3).astype(np.float64)
    key = np.round(soup * (10 ** _MERGE_DIGITS)).astype(np.int64)
    _, first, inverse = np.unique(key, axis=0, return_index=True, return_inverse=True)
    inverse = np.asarray(inverse).reshape(-1)
    order = np.argsort(first, kind='stable')          # unique ids sorted by first appearance
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    verts = soup[first[order]].astype(np.float32)
    faces = rank[inverse].reshape(-1, 3).astype(np.int32)
    return verts, faces


def load_mesh(path: str):
    tris, _ = read_binary_stl(path)
    return weld(tris)
