"""Camera intrinsics, camera pose and the GL projection the renderer uses.

`Intrinsics` keeps the reference surface (robotpose/projection.py:20-193): presets,
RealSense-style string parsing, integer downscale, width/height/size and a string
form usable as a cache key.  pyrealsense2 / pyrender are not needed: the string
form is produced directly and the projection is returned as a 4x4 matrix.
"""
import re
from typing import Union

import numpy as np

from .constants import ZNEAR, ZFAR

_PRESETS = {                                   # projection.py:91-108
    '1280_720_color': ((1280, 720), (638.391, 361.493), (905.23, 904.858)),
    '1280_720_depth': ((1280, 720), (639.459, 359.856), (635.956, 635.956)),
    '640_480_color': ((640, 480), (320.503, 237.288), (611.528, 611.528)),
    '640_480_depth': ((640, 480), (321.635, 241.618), (385.134, 385.134)),
}

_MODELS = ('Brown Conrady', 'Inverse Brown Conrady', 'Ftheta', 'Kannala Brandt4',
           'Modified Brown Conrady', 'None')


def _g(x: float) -> str:
    """C++ ostream default float formatting (6 significant digits), as librealsense prints."""
    return '%g' % x


class Intrinsics:

    bases = list(_PRESETS)

    def __init__(self, input: Union[str, 'Intrinsics'] = None):
        self.resolution, self.pp, self.f = None, None, None
        self.model = 'Brown Conrady'
        self.coeffs = [0.0] * 5
        if input is not None:
            text = str(input)
            if any(text == b or (b + '_') in text for b in self.bases):
                self.fromPreset(text)
            else:
                self.fromString(text)

    def fromString(self, text: str):
        """Parse "[ WxH  p[ppx ppy]  f[fx fy]  Model [c0 c1 c2 c3 c4] ]" (projection.py:47-78)."""
        num = r'[-+]?[0-9]*\.?[0-9]+(?:[eE][-+]?[0-9]+)?'
        m = re.search(r'([1-9][0-9]*) *x *([1-9][0-9]*)', text)
        pp = re.search(rf'p\[ *({num}) +({num}) *\]', text)
        ff = re.search(rf'f\[ *({num}) +({num}) *\]', text)
        model = re.search(r'\] +([A-Za-z0-9 ]*?) +\[', text)
        co = re.search(rf'\[ *({num}) +({num}) +({num}) +({num}) +({num}) *\] *\]', text)
        if not (m and pp and ff and model and co):
            raise ValueError(f"Cannot parse intrinsics string: {text!r}")
        self.resolution = (int(m.group(1)), int(m.group(2)))
        self.pp = (float(pp.group(1)), float(pp.group(2)))
        self.f = (float(ff.group(1)), float(ff.group(2)))
        self.model = model.group(1).strip()
        self.coeffs = [float(co.group(i)) for i in range(1, 6)]

    def fromPreset(self, preset: str = '1280_720_color'):
        """Preset name, optionally suffixed '_k' for an integer downscale (projection.py:81-124)."""
        self.model = 'Brown Conrady'
        self.coeffs = [0.0] * 5
        for base in self.bases:
            if preset == base:
                self.resolution, self.pp, self.f = _PRESETS[base]
                return
            if (base + '_') in preset:
                self.resolution, self.pp, self.f = _PRESETS[base]
                self.downscale(int(preset.replace(base + '_', '')))
                return
        raise ValueError(f"Input {preset} not valid.\nPreset must be one of: {self.bases}")

    def downscale(self, ds_factor: int):
        """Integer downscale; both dimensions must divide exactly (projection.py:127-136)."""
        assert ds_factor >= 1, "Downscaling by a factor of less than 1 (upscaling) is not supported."
        scaled = [x / ds_factor for x in self.resolution]
        if not all(int(x) == round(x) for x in scaled):
            raise ValueError(f"Downscaling by a factor of {ds_factor} is not valid for this resolution. "
                             f"This yields {scaled} as a resolution, which cannot be interpreted.")
        self.resolution = tuple(x // ds_factor for x in self.resolution)
        self.pp = tuple(x / ds_factor for x in self.pp)
        self.f = tuple(x / ds_factor for x in self.f)

    @property
    def width(self) -> int:
        return max(self.resolution)

    @property
    def height(self) -> int:
        return min(self.resolution)

    @property
    def size(self) -> int:
        return int(np.prod(np.array(self.resolution)))

    @property
    def fx(self): return self.f[0]
    @property
    def fy(self): return self.f[1]
    @property
    def cx(self): return self.pp[0]
    @property
    def cy(self): return self.pp[1]

    def __str__(self) -> str:
        # layout of pyrealsense2's intrinsics repr, the reference's cache key (projection.py:183-184)
        c = ' '.join(_g(x) for x in self.coeffs)
        return (f"[ {self.width}x{self.height}  p[{_g(self.pp[0])} {_g(self.pp[1])}]  "
                f"f[{_g(self.f[0])} {_g(self.f[1])}]  {self.model} [{c}] ]")

    def __eq__(self, other) -> bool:
        return isinstance(other, Intrinsics) and self.__dict__ == other.__dict__

    def __ne__(self, other) -> bool:
        return not self.__eq__(other)

    def gl_projection(self, znear: float = ZNEAR, zfar: float = ZFAR) -> np.ndarray:
        """4x4 OpenGL projection of pyrender 0.1.45's IntrinsicsCamera (called through
        projection.py:169 with default znear/zfar)."""
        W, H = float(self.width), float(self.height)
        P = np.zeros((4, 4))
        P[0, 0] = 2.0 * self.fx / W
        P[1, 1] = 2.0 * self.fy / H
        P[0, 2] = 1.0 - 2.0 * self.cx / W
        P[1, 2] = 2.0 * self.cy / H - 1.0
        P[3, 2] = -1.0
        P[2, 2] = (zfar + znear) / (znear - zfar)
        P[2, 3] = (2.0 * zfar * znear) / (znear - zfar)
        return P


def camera_pose_matrix(pose6) -> np.ndarray:
    """[x,y,z,a3,a4,a5] -> 4x4 camera-to-world pose.

    Follows Renderer.setCameraPose (render.py:107-111): a4 += pi/2, then
    makePose(x,y,z,pitch=a3,roll=a4,yaw=a5) = Rz(yaw)·Ry(pitch)·Rx(roll) with the
    element formulas of angToPoseArr (render_utils.py:56-85).
    """
    x, y, z, a3, a4, a5 = [float(v) for v in pose6]
    yaw, pitch, roll = a5, a3, a4 + np.pi / 2
    c = np.cos(np.array([yaw, pitch, roll]))
    s = np.sin(np.array([yaw, pitch, roll]))
    M = np.zeros((4, 4))
    M[0, 0] = c[0] * c[1]
    M[1, 0] = c[1] * s[0]
    M[2, 0] = -1 * s[1]
    M[0, 1] = c[0] * s[1] * s[2] - c[2] * s[0]
    M[1, 1] = c[0] * c[2] + np.prod(s)
    M[2, 1] = c[1] * s[2]
    M[0, 2] = s[0] * s[2] + c[0] * c[2] * s[1]
    M[1, 2] = c[2] * s[0] * s[1] - c[0] * s[2]
    M[2, 2] = c[1] * c[2]
    M[3, 3] = 1.0
    M[0, 3], M[1, 3], M[2, 3] = x, y, z
    return M


def view_matrix(pose6) -> np.ndarray:
    """World-to-camera matrix: rigid inverse of the camera pose (camera looks along -Z, +Y up)."""
    M = camera_pose_matrix(pose6)
    R, t = M[:3, :3], M[:3, 3]
    V = np.eye(4)
    V[:3, :3] = R.T
    V[:3, 3] = -R.T @ t
    return V


def camera_matrix(pose6, intrinsics: Intrinsics, znear: float = ZNEAR, zfar: float = ZFAR) -> np.ndarray:
    """P·V of a camera pose and its intrinsics — the matrix the engine, the oracle and a C host all start from.

    One implementation for every caller: rope_camera_matrix in the library (host code, no GPU: the written double operations
    in the written order, libm's sin / cos), so that the Python Predictor and a plain C host hand the engine the same bits.
    The numpy expression `intrinsics.gl_projection(znear, zfar) @ view_matrix(pose6)` is the same matrix up to the last digit
    (its matrix products go through BLAS, whose summation order and fused operations are not ours to fix); it is what runs when
    the library has not been built."""
    pose = np.ascontiguousarray(pose6, np.float64).reshape(6)
    try:
        import ctypes as C
        from .engine import EngineUnavailable, load_library
        try:
            lib = load_library()
        except EngineUnavailable:
            lib = None
        if lib is not None:
            PV = np.empty((4, 4))
            rc = lib.rope_camera_matrix(pose.ctypes.data_as(C.c_void_p), float(intrinsics.fx), float(intrinsics.fy), float(intrinsics.cx),
                                        float(intrinsics.cy), int(intrinsics.width), int(intrinsics.height), float(znear), float(zfar),
                                        PV.ctypes.data_as(C.c_void_p))
            if rc == 0:
                return PV
    except ImportError:
        pass
    return intrinsics.gl_projection(znear, zfar) @ view_matrix(pose)

