"""ctypes front-end of the CPU oracle (oracle/rope_oracle.c) + numpy restatements.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by rope_s3d_amd/.

PARITY UNPINNED — the reference has no tests/fixtures for this path and its
dependencies (pyrender, klampt, tensorflow, cv2) are absent, so the oracle is pinned
only by closed-form checks (tests/test_oracle_*.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LOSS_DEPTH, LOSS_FULL, LOSS_LOOKUP, LOSS_TSWEEP = 0, 1, 2, 3
KEY_EMPTY = 0xFFFFFFFF
Q32 = 4294967296.0
TQ_MAX = (1 << 39) - 1


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, 'librope_oracle.so')
    src = os.path.join(_HERE, 'rope_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'librope_oracle.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_finalize.restype = C.c_double
        _LIB.orc_finalize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        _LIB.orc_sincos.argtypes = [C.c_double, C.c_void_p, C.c_void_p]
        _LIB.orc_resolve.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        _LIB.orc_sums.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.orc_raster.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _LIB.orc_eval_batch.argtypes = ([C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
                                        + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int])
        _LIB.orc_sum_words.restype = C.c_int
        _LIB.orc_coverage_batch.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def sincos(x: float):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def pack_target(depth: np.ndarray, mask_bits: np.ndarray = None) -> np.ndarray:
    """Target depth (metres, any float dtype) + per-link mask bits -> uint64 plane.

    bits 0..38: depth in Q32 metres, round-half-even, clipped to [0, 2^39-1]
    (NaN / negative -> 0); bits 40..47: bit l set where link l's target mask is true.
    """
    d = np.asarray(depth, np.float64)
    q = np.rint(np.where(np.isfinite(d) & (d > 0), d, 0.0) * Q32)
    q = np.minimum(q, float(TQ_MAX)).astype(np.uint64)
    if mask_bits is not None:
        q |= np.asarray(mask_bits, np.uint64) << np.uint64(40)
    return np.ascontiguousarray(q)


class Oracle:
    """Full-frame CPU renderer/scorer over plain arrays (no dependence on the product code)."""

    def __init__(self, verts, faces, vtx_off, tri_off, joint_fixed, joint_axes, PV, W, H, znear=0.05, zfar=100.0):
        self.verts = np.ascontiguousarray(verts, np.float32)
        self.faces = np.ascontiguousarray(faces, np.int32)
        self.vtx_off = np.ascontiguousarray(vtx_off, np.int32)
        self.tri_off = np.ascontiguousarray(tri_off, np.int32)
        self.joint_fixed = np.ascontiguousarray(joint_fixed, np.float64)
        self.joint_axes = np.ascontiguousarray(joint_axes, np.float64)
        self.PV = np.ascontiguousarray(PV, np.float64)
        self.W, self.H, self.znear, self.zfar = int(W), int(H), float(znear), float(zfar)
        self.sum_words = lib().orc_sum_words()

    def fk(self, q) -> np.ndarray:
        q = np.ascontiguousarray(q, np.float64)
        out = np.zeros((7, 12))
        lib().orc_fk(_p(self.joint_fixed), _p(self.joint_axes), _p(q), _p(out))
        return out

    def mvp(self, q, n) -> np.ndarray:
        fk = self.fk(q)
        out = np.zeros((n, 16), np.float32)
        lib().orc_mvp(_p(self.PV), _p(fk), int(n), _p(out))
        return out

    def raster_key(self, q, n) -> np.ndarray:
        mvp = self.mvp(q, n)
        key = np.empty((self.H, self.W), np.uint32)
        lib().orc_raster(_p(self.verts), _p(self.faces), _p(self.vtx_off), _p(self.tri_off), int(n),
                         _p(mvp), self.W, self.H, _p(key))
        return key

    def resolve(self, key):
        depth = np.empty(key.shape, np.float32)
        ids = np.empty(key.shape, np.uint8)
        lib().orc_resolve(_p(np.ascontiguousarray(key)), key.size, self.znear, self.zfar, _p(depth), _p(ids))
        return depth, ids

    def render(self, q, n=6):
        """-> (depth float32 HxW metres, link id uint8 HxW, 255 = background)."""
        return self.resolve(self.raster_key(q, n))

    def sums(self, key, loss, n, tq=None, t32=None, crop=None) -> np.ndarray:
        s = np.zeros(self.sum_words, np.uint64)
        crop_a = np.ascontiguousarray(crop, np.int32) if crop is not None else None
        lib().orc_sums(_p(np.ascontiguousarray(key)), self.W, self.H, self.znear, self.zfar, int(loss), int(n),
                       _p(tq), _p(t32), _p(crop_a), _p(s))
        return s

    def finalize(self, sums, loss, n, n_pix, link_flags) -> float:
        lf = np.ascontiguousarray(link_flags, np.uint8)
        return lib().orc_finalize(_p(np.ascontiguousarray(sums, np.uint64)), int(loss), int(n), float(n_pix), _p(lf))

    def eval(self, cand, loss, n, tq=None, t32=None, crop=None, link_flags=None, threads=1, want_sums=False):
        cand = np.ascontiguousarray(cand, np.float64).reshape(-1, 6)
        Cn = len(cand)
        err = np.zeros(Cn)
        sums = np.zeros((Cn, self.sum_words), np.uint64) if want_sums else None
        lf = np.ascontiguousarray(link_flags if link_flags is not None else np.zeros(8), np.uint8)
        crop_a = np.ascontiguousarray(crop, np.int32) if crop is not None else None
        lib().orc_eval_batch(_p(self.verts), _p(self.faces), _p(self.vtx_off), _p(self.tri_off),
                             _p(self.joint_fixed), _p(self.joint_axes), _p(self.PV), self.W, self.H,
                             self.znear, self.zfar, int(loss), int(n), _p(tq), _p(t32), _p(crop_a), _p(lf),
                             _p(cand), Cn, _p(err), _p(sums), int(threads))
        return (err, sums) if want_sums else err


    def coverage(self, cand, n, threads=1) -> np.ndarray:
        cand = np.ascontiguousarray(cand, np.float64).reshape(-1, 6)
        cover = np.zeros((self.H, self.W), np.uint8)
        lib().orc_coverage_batch(_p(self.verts), _p(self.faces), _p(self.vtx_off), _p(self.tri_off),
                                 _p(self.joint_fixed), _p(self.joint_axes), _p(self.PV), self.W, self.H, int(n),
                                 _p(cand), len(cand), _p(cover), int(threads))
        return cover


# ---------------------------------------------------------------- numpy literal restatements
def error_numpy(n, render_blue, render_depth, tgt_depth, masked_targets, target_masks, link_names, link_blue):
    """Predictor._error in numpy, term by term (robotpose/prediction/predict.py:475-509).

    render_blue: HxW channel-0 of the render; link_blue[name]: that link's channel-0 value.
    Used to bound the distance between the fixed-point contract and the reference's
    float64 arithmetic (tests assert <= 1e-9 relative).
    """
    err = 0
    for link in link_names[1:n]:
        if link in masked_targets:
            target_masked, joint_mask = masked_targets[link], target_masks[link]
            render_mask = render_blue == link_blue[link]
            render_masked = render_depth * render_mask
            err += np.mean(joint_mask != render_mask) * 5
            if np.sum(target_masked != 0) > (.05 * np.sum(joint_mask)):
                diff = np.abs(target_masked - render_masked)
                if diff[diff != 0].size > 0:
                    err += np.mean(diff[diff != 0]) * 10
    diff = np.abs(tgt_depth - render_depth)
    with np.errstate(all='ignore'):
        err += np.mean(diff[diff != 0]) * np.std(diff)
    return err


def lookup_score_numpy(target_crop_f32, depth_crops_f32):
    """Lookup stage score, float32 like the TF ops (predict.py:117,167-169): mean|T - sqrt(D)| * std(...)."""
    diff = np.abs(target_crop_f32[None].astype(np.float32) - np.sqrt(depth_crops_f32.astype(np.float32)))
    return diff.mean((1, 2), dtype=np.float64) * diff.std((1, 2), dtype=np.float64)
