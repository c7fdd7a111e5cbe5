"""ctypes binding of the HIP engine (librope_hip.so, ABI in include/rope_s3d.h).

This module is the only way the Python host code reaches the GPU.  There is no CPU
fallback: if the shared library is missing or no HIP device is usable, construction
raises `EngineUnavailable`.
"""
import ctypes as C
import os

import numpy as np

from .build import LIB_PATH
from .robot import RobotModel

LOSS_DEPTH, LOSS_FULL, LOSS_LOOKUP, LOSS_TSWEEP, LOSS_CAMFULL = 0, 1, 2, 3, 4
SUM_WORDS = 23
Q32 = 4294967296.0
TQ_MAX = (1 << 39) - 1

ABI_SYMBOLS = (
    'rope_create', 'rope_destroy', 'rope_last_error', 'rope_set_robot', 'rope_set_camera', 'rope_set_target',
    'rope_candidates_upload', 'rope_eval_resident', 'rope_sync', 'rope_results_download', 'rope_eval',
    'rope_lookup_build', 'rope_lookup_score', 'rope_render', 'rope_coverage', 'rope_debug_mvp', 'rope_profile_eval', 'rope_set_strategy',
    'rope_set_frames', 'rope_eval_views', 'rope_predict', 'rope_set_robot_mesh', 'rope_partition_mesh', 'rope_pack_target', 'rope_downsample_even',
    'rope_seg_nms', 'rope_seg_roi_align', 'rope_seg_bias_act',
    'rope_set_target_tsweep', 'rope_set_targets', 'rope_stage_targets', 'rope_commit_targets', 'rope_eval_targets', 'rope_lookup_score_targets', 'rope_predict_batch',
    'rope_prepare_synthetic', 'rope_host_alloc', 'rope_host_free', 'rope_build_id', 'rope_camera_matrix', 'rope_lookup_grid', 'rope_crop_divisions',
    'rope_prepare_segmented')


STAGE_LOOKUP, STAGE_DESCENT, STAGE_SFLIP, STAGE_ISWEEP, STAGE_TSWEEP = 0, 1, 2, 3, 4


class StageDesc(C.Structure):
    """rope_stage (include/rope_s3d.h)."""
    _fields_ = [('kind', C.c_int32), ('to_render', C.c_int32), ('count', C.c_int32), ('joints', C.c_uint32),
                ('init_rate', C.c_double * 6), ('rate_reduction', C.c_double), ('early_stop', C.c_double), ('range', C.c_double)]


class PredictArgs(C.Structure):
    """rope_predict_args (include/rope_s3d.h)."""
    _fields_ = [('stages', C.POINTER(StageDesc)), ('n_stages', C.c_int32), ('speculate', C.c_int32),
                ('limits', C.c_void_p), ('camera_pose', C.c_void_p), ('min_ang_inc', C.c_void_p),
                ('lookup_angles', C.c_void_p), ('n_lookup', C.c_int32), ('use_table', C.c_int32), ('lookup_crop', C.c_void_p),
                ('lookup_angles_live', C.c_void_p)]


class EngineUnavailable(RuntimeError):
    pass


class EngineError(RuntimeError):
    pass


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch ships its own copy of libamdhip64 (SONAME libamdhip64.so.7, found through its RPATH);
    librope_hip.so asks the loader for libamdhip64.so.7 and, loaded first, gets the system copy — a later `import torch` (the
    segmentation stage: Predictor._load_segmenter imports maskrcnn after the engine exists) then brings a SECOND runtime into the
    process, torch finds no GPU, and the rope_seg_* kernels would be handed streams and pointers of a runtime that is not theirs.
    So when torch is installed and not yet loaded, its copy is mapped first: the loader then satisfies librope_hip.so — and
    torch's own libraries later — with that one copy.  Nothing of torch is imported here."""
    import sys
    if 'torch' in sys.modules or os.environ.get('ROPE_SYSTEM_HIP_RUNTIME'):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.submodule_search_locations:
            return
        lib = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
        if os.path.exists(lib):
            C.CDLL(lib, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass                                             # no torch, or a torch without its own runtime: the system copy serves everyone


def load_library(path: str = None):
    """dlopen librope_hip.so and declare the prototypes.  Does not touch the GPU."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get('ROPE_HIP_LIB', LIB_PATH)
    if not os.path.exists(path):
        raise EngineUnavailable(f"{path} not found: build it with `python -m rope_s3d_amd.build` "
                                "(hipcc, gfx950); the engine has no CPU fallback")
    _share_torch_hip_runtime()
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise EngineUnavailable(f"cannot load {path}: {e}") from e
    vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
    lib.rope_create.argtypes = [C.POINTER(vp), i32]
    lib.rope_destroy.argtypes = [vp]
    lib.rope_destroy.restype = None
    lib.rope_last_error.argtypes = [vp]
    lib.rope_last_error.restype = C.c_char_p
    lib.rope_set_robot.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp]
    lib.rope_set_camera.argtypes = [vp, vp, i32, i32, dbl, dbl]
    lib.rope_set_target.argtypes = [vp, vp, vp, vp]
    lib.rope_candidates_upload.argtypes = [vp, vp, i32]
    lib.rope_eval_resident.argtypes = [vp, i32, i32, vp]
    lib.rope_sync.argtypes = [vp]
    lib.rope_results_download.argtypes = [vp, vp, vp, vp, vp]
    lib.rope_eval.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.rope_lookup_build.argtypes = [vp, vp, i32, i32, vp]
    lib.rope_lookup_score.argtypes = [vp, vp, vp, vp]
    lib.rope_render.argtypes = [vp, vp, i32, vp, vp]
    lib.rope_coverage.argtypes = [vp, vp, i32, i32, vp]
    lib.rope_debug_mvp.argtypes = [vp, vp, i32, i32]
    lib.rope_profile_eval.argtypes = [vp, i32, i32, vp, i32, vp]
    lib.rope_set_strategy.argtypes = [vp, i32]
    if hasattr(lib, 'rope_debug_skip'):                 # librope_hip_profile.so only (ROPE_HIP_LIB=...)
        lib.rope_debug_skip.argtypes = [vp, i32]
        lib.rope_debug_clock.argtypes = [vp, vp]
        lib.rope_debug_bounds.argtypes = [vp, vp]
    lib.rope_predict.argtypes = [vp, C.POINTER(PredictArgs), vp, vp, C.POINTER(C.c_int64)]
    lib.rope_set_robot_mesh.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
    lib.rope_partition_mesh.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp]
    lib.rope_pack_target.argtypes = [vp, vp, C.c_int64, vp]
    lib.rope_downsample_even.argtypes = [vp, i32, i32, i32, C.c_int64, i32, i32, vp]
    lib.rope_seg_nms.argtypes = [vp, vp, vp, i32, i32, C.c_float, i32, vp, vp, vp]
    lib.rope_seg_roi_align.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, C.c_float, vp, vp, vp]
    lib.rope_seg_bias_act.argtypes = [vp, vp, vp, C.c_int64, i32, C.c_int64, i32, vp]
    lib.rope_set_frames.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.rope_eval_views.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.rope_set_target_tsweep.argtypes = [vp, vp]
    lib.rope_set_targets.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.rope_stage_targets.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.rope_commit_targets.argtypes = [vp]
    lib.rope_eval_targets.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    lib.rope_lookup_score_targets.argtypes = [vp, vp, vp, vp]
    lib.rope_predict_batch.argtypes = [vp, C.POINTER(PredictArgs), i32, vp, vp, C.POINTER(C.c_int64)]
    lib.rope_camera_matrix.argtypes = [vp, dbl, dbl, dbl, dbl, i32, i32, dbl, dbl, vp]
    lib.rope_crop_divisions.argtypes = [C.c_int64, i32, vp]
    lib.rope_prepare_segmented.argtypes = [vp, i32, C.c_int64, i32, i32, i32, vp, i32, vp, i32, i32, vp, vp, vp, vp]
    lib.rope_lookup_grid.argtypes = [vp, vp, vp, C.c_int64]
    lib.rope_lookup_grid.restype = C.c_int64
    lib.rope_build_id.argtypes = []
    lib.rope_build_id.restype = C.c_char_p
    lib.rope_host_alloc.argtypes = [C.c_size_t]
    lib.rope_host_alloc.restype = C.c_void_p
    lib.rope_host_free.argtypes = [vp]
    lib.rope_host_free.restype = None
    lib.rope_prepare_synthetic.argtypes = [vp, C.c_int64, vp, i32, C.c_int64, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp]
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def pack_target(depth: np.ndarray, mask_bits: np.ndarray = None) -> np.ndarray:
    """Target depth (metres) + per-link mask bits -> the uint64 plane rope_set_target takes.

    bits 0..38: depth in Q32 metres (round half even, clipped to [0, 2^39-1]; NaN and
    negatives count as "no depth" = 0); bits 40..47: bit l set where link l's mask is true.
    """
    d = np.asarray(depth, np.float64)
    if _lib is not None or os.path.exists(os.environ.get('ROPE_HIP_LIB', LIB_PATH)):       # one pass in the library (host code, no GPU)
        d = np.ascontiguousarray(d)
        bits = None if mask_bits is None else np.ascontiguousarray(mask_bits, np.uint8)
        out = np.empty(d.shape, np.uint64)
        if load_library().rope_pack_target(_p(d), _p(bits), d.size, _p(out)) == 0:
            return out
    q = np.rint(np.where(np.isfinite(d) & (d > 0), d, 0.0) * Q32)
    q = np.minimum(q, float(TQ_MAX)).astype(np.uint64)
    if mask_bits is not None:
        q |= np.asarray(mask_bits, np.uint64) << np.uint64(40)
    return np.ascontiguousarray(q)


def build_id() -> str:
    """rope_build_id of the loaded library: the hash of the sources it was built from."""
    return load_library().rope_build_id().decode()


def pinned_empty(shape, dtype) -> np.ndarray:
    """np.empty in page-locked host memory (rope_host_alloc): planes handed to set_target(s) from it reach the device in one
    transfer.  Falls back to ordinary memory when the runtime refuses.  The memory lives as long as the array (or any view of it)."""
    import weakref
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    lib = load_library()
    ptr = lib.rope_host_alloc(n) if n else None
    if not ptr:
        return np.empty(shape, dtype)
    buf = (C.c_ubyte * n).from_address(ptr)
    weakref.finalize(buf, lib.rope_host_free, C.c_void_p(ptr))
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def prepare_synthetic(color: np.ndarray, depth: np.ndarray, f: int, link_blue, n_lookup_links: int, tq: np.ndarray, lookup_f32: np.ndarray,
                      flags: np.ndarray, tgt_depth: np.ndarray = None) -> bool:
    """rope_prepare_synthetic: one frame of the synthetic path into the given output arrays (host code, interpreter lock released).
    False when the arrays' layout is not one the library takes (the caller then goes the numpy way)."""
    if color.dtype != np.uint8 or color.ndim != 3 or color.shape[2] != 3 or color.strides[2] != 1 or color.strides[1] != 3 or color.strides[0] < 0:
        return False
    kind = {np.dtype(np.float32): 1, np.dtype(np.float64): 2}.get(depth.dtype)
    if kind is None or depth.ndim != 2 or depth.shape != color.shape[:2] or depth.strides[1] != depth.itemsize or depth.strides[0] < 0:
        return False
    H0, W0 = depth.shape
    if f < 1 or (f > 1 and f % 2) or H0 % f or W0 % f:
        return False
    for a, dt in ((tq, np.uint64), (lookup_f32, np.float32), (flags, np.uint8)) + (((tgt_depth, np.float64),) if tgt_depth is not None else ()):
        assert a.dtype == dt and a.flags.c_contiguous
    assert tq.shape == (H0 // f, W0 // f) == lookup_f32.shape and flags.size >= 8
    lb = np.ascontiguousarray(link_blue, np.int32)
    rc = load_library().rope_prepare_synthetic(C.c_void_p(color.ctypes.data), color.strides[0], C.c_void_p(depth.ctypes.data), kind, depth.strides[0],
                                               H0, W0, int(f), _p(lb), len(lb), int(n_lookup_links), _p(tq), _p(lookup_f32), _p(tgt_depth), _p(flags))
    return rc == 0


def prepare_segmented(depth: np.ndarray, f: int, masks: np.ndarray, link_of, n_links: int, n_lookup_links: int, tq: np.ndarray,
                      lookup_f32: np.ndarray, flags: np.ndarray, tgt_depth: np.ndarray = None) -> bool:
    """rope_prepare_segmented: one frame of the segmentation path (instance masks (H, W, K) bool + the link of every instance) into
    the given output arrays.  False when the layout is not one the library takes."""
    kind = {np.dtype(np.float32): 1, np.dtype(np.float64): 2}.get(depth.dtype)
    if kind is None or depth.ndim != 2 or depth.strides[1] != depth.itemsize or depth.strides[0] < 0:
        return False
    H0, W0 = depth.shape
    if f < 1 or (f > 1 and f % 2) or H0 % f or W0 % f:
        return False
    m = np.ascontiguousarray(masks).view(np.uint8) if masks.dtype == bool else np.ascontiguousarray(masks, np.uint8)
    if m.ndim != 3 or m.shape[:2] != (H0 // f, W0 // f):
        return False
    lo = np.ascontiguousarray(link_of, np.int32)
    if len(lo) != m.shape[2]:
        return False
    assert tq.shape == (H0 // f, W0 // f) == lookup_f32.shape and tq.flags.c_contiguous and lookup_f32.flags.c_contiguous and flags.size >= 8
    rc = load_library().rope_prepare_segmented(C.c_void_p(depth.ctypes.data), kind, depth.strides[0], H0, W0, int(f), _p(m), m.shape[2], _p(lo),
                                               int(n_links), int(n_lookup_links), _p(tq), _p(lookup_f32), _p(tgt_depth), _p(flags))
    return rc == 0


class Engine:
    """One engine context on one GPU: robot + camera + current target, batched evaluation."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.rope_create(C.byref(self._ctx), int(device))
        if rc != 0:
            msg = self._lib.rope_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise EngineUnavailable(f"rope_create(device={device}) failed ({rc}): {msg}")
        self.device = device
        self.W = self.H = 0
        self.n_links = 0
        self.n_candidates = 0
        self._strategy = 0            # last value given to rope_set_strategy (the ROPE_STRATEGY environment default is the library's own)

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value:
            self._lib.rope_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise EngineError(f"{what} failed ({rc}): {self._lib.rope_last_error(self._ctx).decode()}")

    # -- static state -------------------------------------------------------------------------
    def set_robot(self, robot: RobotModel):
        m = robot.meshlets
        hdr = np.ascontiguousarray(m.header, np.uint32)
        verts = np.ascontiguousarray(m.verts, np.float32)
        tris = np.ascontiguousarray(m.tris, np.uint32)
        first = np.ascontiguousarray(m.link_first, np.int32)
        jf = np.ascontiguousarray(robot.joint_fixed, np.float64)
        ja = np.ascontiguousarray(robot.joint_axes, np.float64)
        self._check(self._lib.rope_set_robot(self._ctx, _p(hdr), len(hdr), _p(verts), len(verts), _p(tris), len(tris),
                                             _p(first), len(first) - 1, _p(jf), _p(ja)), 'rope_set_robot')
        self.n_links = len(first) - 1

    def set_robot_mesh(self, robot: RobotModel):
        """rope_set_robot_mesh: plain vertex / index arrays in, the library builds the meshlets (what a C host calls)."""
        verts, faces = np.ascontiguousarray(robot.verts, np.float32), np.ascontiguousarray(robot.faces, np.int32)
        vo, to = np.ascontiguousarray(robot.vtx_off, np.int32), np.ascontiguousarray(robot.tri_off, np.int32)
        jf, ja = np.ascontiguousarray(robot.joint_fixed, np.float64), np.ascontiguousarray(robot.joint_axes, np.float64)
        self._check(self._lib.rope_set_robot_mesh(self._ctx, _p(verts), _p(faces), _p(vo), _p(to), len(vo) - 1, _p(jf), _p(ja)),
                    'rope_set_robot_mesh')
        self.n_links = len(vo) - 1

    def set_camera(self, PV: np.ndarray, W: int, H: int, znear: float, zfar: float):
        PV = np.ascontiguousarray(PV, np.float64)
        assert PV.shape == (4, 4)
        self._check(self._lib.rope_set_camera(self._ctx, _p(PV), int(W), int(H), float(znear), float(zfar)), 'rope_set_camera')
        self.W, self.H = int(W), int(H)

    def set_target(self, tq: np.ndarray, t32: np.ndarray = None, link_flags=None):
        tq = np.ascontiguousarray(tq, np.uint64)
        if tq.shape != (self.H, self.W):
            raise ValueError(f"target plane must be {(self.H, self.W)}, got {tq.shape}")
        if t32 is not None:
            t32 = np.ascontiguousarray(t32, np.float32)
            if t32.shape != (self.H, self.W):
                raise ValueError("float32 target plane has the wrong shape")
        lf = np.zeros(8, np.uint8)
        if link_flags is not None:
            lf[:len(link_flags)] = link_flags
        self._check(self._lib.rope_set_target(self._ctx, _p(tq), _p(t32), _p(lf)), 'rope_set_target')

    def set_target_tsweep(self, t32_full: np.ndarray = None):
        """The float32 plane ROPE_LOSS_TSWEEP reads (the whole target depth) when it is not the lookup plane; None: the one plane."""
        if t32_full is not None:
            t32_full = np.ascontiguousarray(t32_full, np.float32)
            if t32_full.shape != (self.H, self.W):
                raise ValueError("float32 target plane has the wrong shape")
        self._check(self._lib.rope_set_target_tsweep(self._ctx, _p(t32_full)), 'rope_set_target_tsweep')

    # -- many frames at once --------------------------------------------------------------------
    def _target_planes(self, tq, t32, link_flags, t32_tsweep):
        tq = np.ascontiguousarray(tq, np.uint64)
        n = len(tq)
        if tq.shape != (n, self.H, self.W):
            raise ValueError(f"target planes must be {(n, self.H, self.W)}, got {tq.shape}")
        planes = []
        for a in (t32, t32_tsweep):
            if a is not None:
                a = np.ascontiguousarray(a, np.float32)
                if a.shape != tq.shape:
                    raise ValueError("float32 target planes have the wrong shape")
            planes.append(a)
        lf = np.zeros((n, 8), np.uint8)
        if link_flags is not None:
            lf[:] = np.asarray(link_flags, np.uint8).reshape(n, 8)
        return n, tq, planes[0], planes[1], lf

    def set_targets(self, tq: np.ndarray, t32: np.ndarray = None, link_flags: np.ndarray = None, t32_tsweep: np.ndarray = None):
        """The targets of N frames resident at once (rope_set_targets): tq (N,H,W) uint64, t32 / t32_tsweep (N,H,W) float32 or None,
        link_flags (N,8) uint8."""
        n, tq, t32, ts, lf = self._target_planes(tq, t32, link_flags, t32_tsweep)
        self._check(self._lib.rope_set_targets(self._ctx, n, _p(tq), _p(t32), _p(ts), _p(lf)), 'rope_set_targets')
        self.n_targets = n

    def stage_targets(self, tq: np.ndarray, t32: np.ndarray = None, link_flags: np.ndarray = None, t32_tsweep: np.ndarray = None):
        """set_targets into the context's second set of planes, on a stream of its own (rope_stage_targets): returns with the copies
        on their way while the resident targets stay in use — from this thread or another.  The arrays must stay alive and
        untouched until commit_targets() (they are kept here); page-locked ones (pinned_empty) make the copy asynchronous."""
        n, tq, t32, ts, lf = self._target_planes(tq, t32, link_flags, t32_tsweep)
        self._staged = (n, tq, t32, ts, lf)
        rc = self._lib.rope_stage_targets(self._ctx, n, _p(tq), _p(t32), _p(ts), _p(lf))
        if rc:
            self._staged = None
            raise RuntimeError(f"rope_stage_targets failed ({rc})")

    def commit_targets(self):
        """The staged targets become the resident ones (rope_commit_targets)."""
        self._check(self._lib.rope_commit_targets(self._ctx), 'rope_commit_targets')
        self.n_targets = self._staged[0] if getattr(self, '_staged', None) else 0
        self._staged = None

    def eval_targets(self, cand, frame_of, n_render: int, loss: int, crop=None) -> np.ndarray:
        """Row i of `cand` scored against the resident target frame_of[i] -> errors (R,)."""
        cand = np.ascontiguousarray(cand, np.float64).reshape(-1, 6)
        fo = np.ascontiguousarray(frame_of, np.int32).reshape(-1)
        if len(fo) != len(cand):
            raise ValueError("one frame index per row")
        crop_a = np.ascontiguousarray(crop, np.int32) if crop is not None else None
        err = np.empty(len(cand), np.float64)
        self._check(self._lib.rope_eval_targets(self._ctx, _p(cand), _p(fo), len(cand), int(n_render), int(loss), _p(crop_a), _p(err)), 'rope_eval_targets')
        return err

    def lookup_score_targets(self, want_scores: bool = False):
        """The stored table against every resident target -> (scores (N, rows) or None, first-argmin row per frame, its score per frame)."""
        n = self.n_targets
        scores = np.empty((n, self._table_rows), np.float64) if want_scores else None
        bi, be = np.empty(n, np.int32), np.empty(n, np.float64)
        self._check(self._lib.rope_lookup_score_targets(self._ctx, _p(bi), _p(be), _p(scores)), 'rope_lookup_score_targets')
        return scores, bi, be

    def _predict_args(self, stages, limits, camera_pose, min_ang_inc, lookup_angles, lookup_crop, use_table, speculate, lookup_live):
        arr = stages if isinstance(stages, C.Array) else (StageDesc * len(stages))(*stages)
        limits = np.ascontiguousarray(limits, np.float64).reshape(6, 2)
        cam = np.ascontiguousarray(camera_pose, np.float64).reshape(6)
        inc = np.ascontiguousarray(min_ang_inc, np.float64).reshape(6)
        grid = np.ascontiguousarray(lookup_angles, np.float64).reshape(-1, 6) if lookup_angles is not None else None
        crop = np.ascontiguousarray(lookup_crop, np.int32).reshape(4) if lookup_crop is not None else None
        if lookup_live is not None:                  # the reference's table aliasing: edited in place by the library
            if grid is None or lookup_live.shape != grid.shape or lookup_live.dtype != np.float64 or not lookup_live.flags.c_contiguous:
                raise ValueError("lookup_live must be a C-contiguous float64 array of the grid's shape")
        a = PredictArgs(arr, len(arr), int(speculate), _p(limits), _p(cam), _p(inc), _p(grid), 0 if grid is None else len(grid),
                        1 if use_table else 0, _p(crop), _p(lookup_live))
        return a, (arr, limits, cam, inc, grid, crop, lookup_live)          # the arrays the struct points into stay alive with the tuple

    def predict_batch(self, stages, limits, camera_pose, min_ang_inc, lookup_angles=None, lookup_crop=None, use_table=False,
                      speculate: int = 3):
        """rope_predict_batch: the stage machine over all resident targets in lockstep.
        -> (angles (N, 6), trace (N, n_stages, 6), candidate poses evaluated)."""
        a, keep = self._predict_args(stages, limits, camera_pose, min_ang_inc, lookup_angles, lookup_crop, use_table, speculate, None)
        n = self.n_targets
        out, trace, cnt = np.empty((n, 6)), np.empty((n, a.n_stages, 6)), C.c_int64()
        self._check(self._lib.rope_predict_batch(self._ctx, C.byref(a), n, _p(out), _p(trace), C.byref(cnt)), 'rope_predict_batch')
        return out, trace, int(cnt.value)

    # -- evaluation ---------------------------------------------------------------------------
    def upload_candidates(self, cand: np.ndarray):
        cand = np.ascontiguousarray(cand, np.float64).reshape(-1, 6)
        self._check(self._lib.rope_candidates_upload(self._ctx, _p(cand), len(cand)), 'rope_candidates_upload')
        self.n_candidates = len(cand)

    def eval_resident(self, n_render: int, loss: int, crop=None):
        crop_a = np.ascontiguousarray(crop, np.int32) if crop is not None else None
        self._check(self._lib.rope_eval_resident(self._ctx, int(n_render), int(loss), _p(crop_a)), 'rope_eval_resident')

    def sync(self):
        self._check(self._lib.rope_sync(self._ctx), 'rope_sync')

    def download(self, want_err=True, want_sums=False):
        n = self.n_candidates
        err = np.empty(n, np.float64) if want_err else None
        sums = np.empty((n, SUM_WORDS), np.uint64) if want_sums else None
        bi, be = C.c_int32(), C.c_double()
        self._check(self._lib.rope_results_download(self._ctx, _p(err), _p(sums), C.byref(bi), C.byref(be)), 'rope_results_download')
        return err, sums, int(bi.value), float(be.value)

    MAX_BATCH = 65535          # candidates per launch (grid y dimension)

    def eval(self, cand, n_render: int, loss: int, crop=None, want_sums=False):
        """-> (err (C,), sums or None, first-argmin index, its error).  Any number of candidates: batches beyond
        65 535 rows are evaluated in several launches and merged (first index of the smallest error, NaN never wins)."""
        cand = np.ascontiguousarray(cand, np.float64).reshape(-1, 6)
        if len(cand) <= self.MAX_BATCH:
            # one call across the boundary: upload + enqueue + sync + download (rope_eval)
            n = len(cand)
            err = np.empty(n, np.float64)
            sums = np.empty((n, SUM_WORDS), np.uint64) if want_sums else None
            crop_a = np.ascontiguousarray(crop, np.int32) if crop is not None else None
            bi, be = C.c_int32(), C.c_double()
            self._check(self._lib.rope_eval(self._ctx, _p(cand), n, int(n_render), int(loss), _p(crop_a), _p(err), _p(sums),
                                            C.byref(bi), C.byref(be)), 'rope_eval')
            self.n_candidates = n
            return err, sums, int(bi.value), float(be.value)
        errs, sums, best, best_err = [], [], -1, np.nan
        for lo in range(0, len(cand), self.MAX_BATCH):
            e, s, bi, be = self.eval(cand[lo:lo + self.MAX_BATCH], n_render, loss, crop, want_sums)
            errs.append(e)
            sums.append(s)
            if best < 0 or be < best_err or (np.isnan(best_err) and not np.isnan(be)):
                best, best_err = lo + bi, be
        self.n_candidates = min(len(cand) - (len(cand) - 1) // self.MAX_BATCH * self.MAX_BATCH, self.MAX_BATCH)
        return np.concatenate(errs), (np.concatenate(sums) if want_sums else None), best, best_err

    def lookup_build(self, cand, n_render: int, crop):
        """Render the pose grid once into an HBM-resident table of cropped sqrt-depth images."""
        cand = np.ascontiguousarray(cand, np.float64).reshape(-1, 6)
        crop_a = np.ascontiguousarray(crop, np.int32)
        self._check(self._lib.rope_lookup_build(self._ctx, _p(cand), len(cand), int(n_render), _p(crop_a)), 'rope_lookup_build')
        self.n_candidates = len(cand)
        self._table_rows = len(cand)

    def lookup_score(self, want_scores: bool = False):
        """-> (scores or None, first-argmin row, its score) of the stored table against the current target."""
        scores = np.empty(self._table_rows, np.float64) if want_scores else None
        bi, be = C.c_int32(), C.c_double()
        self._check(self._lib.rope_lookup_score(self._ctx, _p(scores), C.byref(bi), C.byref(be)), 'rope_lookup_score')
        return scores, int(bi.value), float(be.value)

    def predict(self, stages, limits, camera_pose, min_ang_inc, lookup_angles=None, lookup_crop=None, use_table=False,
                speculate: int = 3, lookup_live: np.ndarray = None):
        """rope_predict: the whole stage machine of one frame in one call.  `stages` = StageDesc array (or list).
        -> (angles (6,), trace (n_stages, 6), candidate poses evaluated)."""
        a, keep = self._predict_args(stages, limits, camera_pose, min_ang_inc, lookup_angles, lookup_crop, use_table, speculate, lookup_live)
        out, trace, n = np.empty(6), np.empty((a.n_stages, 6)), C.c_int64()
        self._check(self._lib.rope_predict(self._ctx, C.byref(a), _p(out), _p(trace), C.byref(n)), 'rope_predict')
        return out, trace, int(n.value)

    def render(self, q, n_render: int = 6):
        """-> (depth float32 HxW metres, link id uint8 HxW with 255 = background)."""
        q = np.ascontiguousarray(q, np.float64).reshape(6)
        depth = np.empty((self.H, self.W), np.float32)
        ids = np.empty((self.H, self.W), np.uint8)
        self._check(self._lib.rope_render(self._ctx, _p(q), int(n_render), _p(depth), _p(ids)), 'rope_render')
        return depth, ids

    def coverage(self, cand, n_render: int) -> np.ndarray:
        cand = np.ascontiguousarray(cand, np.float64).reshape(-1, 6)
        cover = np.empty((self.H, self.W), np.uint8)
        self._check(self._lib.rope_coverage(self._ctx, _p(cand), len(cand), int(n_render), _p(cover)), 'rope_coverage')
        self.n_candidates = len(cand)
        return cover

    # -- camera-pose path ------------------------------------------------------------------------
    def set_frames(self, q, tq, t32=None, link_planes=None):
        """N frames: joint vectors (N,6), uint64 depth planes (N,H,W), optional float32 planes (N,H,W) and per-link
        planes (N,6,H,W) — the targets every candidate camera is scored against (rope_set_frames)."""
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 6)
        n = len(q)
        tq = np.ascontiguousarray(tq, np.uint64)
        if tq.shape != (n, self.H, self.W):
            raise ValueError(f"frame planes must be {(n, self.H, self.W)}, got {tq.shape}")
        if t32 is not None:
            t32 = np.ascontiguousarray(t32, np.float32)
            if t32.shape != tq.shape:
                raise ValueError("float32 frame planes have the wrong shape")
        if link_planes is not None:
            link_planes = np.ascontiguousarray(link_planes, np.uint64)
            if link_planes.shape != (n, 6, self.H, self.W):
                raise ValueError("link planes must be (N, 6, H, W)")
        self._check(self._lib.rope_set_frames(self._ctx, n, _p(q), _p(tq), _p(t32), _p(link_planes)), 'rope_set_frames')
        self.n_frames = n

    def eval_views(self, PV, n_render: int, loss: int) -> np.ndarray:
        """K candidate cameras (K,4,4 P·V) x the resident frames -> (K, N, 23) exact integer sums."""
        PV = np.ascontiguousarray(PV, np.float64).reshape(-1, 4, 4)
        sums = np.empty((len(PV), self.n_frames, SUM_WORDS), np.uint64)
        self._check(self._lib.rope_eval_views(self._ctx, _p(PV), len(PV), int(n_render), int(loss), _p(sums)), 'rope_eval_views')
        self.n_candidates = len(PV) * self.n_frames
        return sums

    def debug_mvp(self, C_: int, n_render: int) -> np.ndarray:
        out = np.empty((C_, n_render, 16), np.float32)
        self._check(self._lib.rope_debug_mvp(self._ctx, _p(out), int(C_), int(n_render)), 'rope_debug_mvp')
        return out

    NO_LAYERS, NO_SPLIT, NO_PARENTS, NO_QUEUE, CLIP_KERNELS, SEPARATE_GEOMETRY = 1, 2, 4, 8, 16, 32

    def set_strategy(self, flags: int):
        """rope_set_strategy: launch structure only (shared layers / small-batch split / second sharing level off);
        results are bit-identical for every value."""
        self._check(self._lib.rope_set_strategy(self._ctx, int(flags)), 'rope_set_strategy')
        self._strategy = int(flags)

    def debug_skip(self, mask: int):
        """Kernel-phase ablation; only the profiling build of the library has it (tools/build_variants.py profile)."""
        if not hasattr(self._lib, 'rope_debug_skip'):
            raise EngineError("rope_debug_skip: not in this library; build librope_hip_profile.so and point ROPE_HIP_LIB at it")
        self._check(self._lib.rope_debug_skip(self._ctx, int(mask)), 'rope_debug_skip')

    def debug_clock(self):
        """Profiling build only: (shader clock in GHz during the last scoring launch of a large batch, that launch's length in ms),
        from s_memtime / s_memrealtime stamps of its workgroups."""
        if not hasattr(self._lib, 'rope_debug_clock'):
            raise EngineError("rope_debug_clock: not in this library; build librope_hip_profile.so and point ROPE_HIP_LIB at it")
        out = np.zeros(2)
        self._check(self._lib.rope_debug_clock(self._ctx, _p(out)), 'rope_debug_clock')
        return float(out[0]), float(out[1])

    def debug_bounds(self) -> int:
        """Profiling build only: indices into the raster kernels' shared arrays / queue segments found out of range so far."""
        if not hasattr(self._lib, 'rope_debug_bounds'):
            raise EngineError("rope_debug_bounds: not in this library; build librope_hip_profile.so and point ROPE_HIP_LIB at it")
        v = C.c_int()
        self._check(self._lib.rope_debug_bounds(self._ctx, C.byref(v)), 'rope_debug_bounds')
        return int(v.value)

    def profile_eval(self, n_render: int, loss: int, crop=None, reps: int = 10):
        """-> dict of average milliseconds per pass (HIP events on the engine stream): fk (+bounds), layer (shared
        upstream links), score (per-candidate raster + loss), finalize (+argmin), total; raster = layer + score."""
        crop_a = np.ascontiguousarray(crop, np.int32) if crop is not None else None
        ms = np.zeros(5, np.float32)
        self._check(self._lib.rope_profile_eval(self._ctx, int(n_render), int(loss), _p(crop_a), int(reps), _p(ms)), 'rope_profile_eval')
        return {'fk': float(ms[0]), 'layer': float(ms[1]), 'score': float(ms[2]), 'raster': float(ms[1] + ms[2]),
                'finalize': float(ms[3]), 'total': float(ms[4])}
