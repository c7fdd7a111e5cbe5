"""Renderer with the reference's surface, drawn by the HIP engine.

Reference: robotpose/simulation/render.py:25-163 (pyrender scene of the six link meshes, an
intrinsics camera, SEG-flag offscreen render returning a flat-colour image and metric
depth).  Here `render()` is one rope_render call; colours come from the link-id image
through the same DEFAULT_RENDER_COLORS table.
"""
from typing import List, Tuple, Union

import numpy as np

from ..constants import (BACKGROUND_ID, DEFAULT_RENDER_COLORS, NUM_RENDER_LINKS,
                         RENDERER_FALLBACK_CAMERA_POSE, ZFAR, ZNEAR)
from ..engine import Engine
from ..projection import Intrinsics, camera_matrix, camera_pose_matrix, view_matrix
from ..robot import RobotModel
from ..urdf import URDFReader

_robot_cache = {}


def active_robot() -> RobotModel:
    """RobotModel of the active URDF, parsed and partitioned once per process."""
    reader = URDFReader()
    key = reader.path
    if key not in _robot_cache:
        _robot_cache[key] = RobotModel.from_urdf(reader)
    return _robot_cache[key]


def shade_depth(depth: np.ndarray, ids: np.ndarray, intr: Intrinsics, ambient: float = 0.12) -> np.ndarray:
    """Mode 'real' (render.py:92-98 without the SEG flag: the untextured meshes lit by one directional light that sits at the
    camera and shines along its view axis, render.py:57-59).  The reference's picture comes out of pyrender's physically based
    shader; this is the same scene with the simplest model of it — grey Lambert surface, head-on light, normals from the
    rendered depth (differences taken inside one link only, so silhouettes stay sharp) — for the viewers that show it to a
    person.  Nothing on the prediction path reads this image.  -> (H, W, 3) uint8."""
    H, W = depth.shape
    z = depth.astype(np.float64)
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    X, Y = (u - intr.cx) / intr.fx * z, (v - intr.cy) / intr.fy * z
    P = np.stack([X, Y, z], -1)

    def diff(axis):
        fwd = np.roll(P, -1, axis) - P
        bwd = P - np.roll(P, 1, axis)
        ok_f = (np.roll(ids, -1, axis) == ids)
        ok_b = (np.roll(ids, 1, axis) == ids)
        edge = np.zeros(ids.shape, bool)
        if axis == 0:
            edge[-1, :] = True
            ok_f &= ~edge
            edge[:] = False
            edge[0, :] = True
            ok_b &= ~edge
        else:
            edge[:, -1] = True
            ok_f &= ~edge
            edge[:] = False
            edge[:, 0] = True
            ok_b &= ~edge
        return np.where(ok_f[..., None], fwd, np.where(ok_b[..., None], bwd, 0.0))
    n = np.cross(diff(1), diff(0))
    norm = np.linalg.norm(n, axis=-1)
    facing = np.where(norm > 0, np.abs(n[..., 2]) / np.where(norm > 0, norm, 1.0), 1.0)      # light along the view axis: |n . z|
    shade = np.clip(ambient + (1.0 - ambient) * facing, 0.0, 1.0)
    grey = np.where(ids != BACKGROUND_ID, np.round(200.0 * shade), 0.0).astype(np.uint8)
    return np.repeat(grey[..., None], 3, -1)


class Renderer:

    def __init__(self, mode: str = 'seg', camera_pose: np.ndarray = None,
                 camera_intrin: Union[str, Intrinsics] = '1280_720_color', suppress_warnings: bool = False,
                 intrinsic_ds_factor: int = None, device: int = 0):
        self.suppress_warnings = suppress_warnings
        self.intrinsics = Intrinsics(camera_intrin)
        if intrinsic_ds_factor is not None:
            self.intrinsics.downscale(intrinsic_ds_factor)
        self.robot = active_robot()
        self.engine = Engine(device)
        self.engine.set_robot(self.robot)
        self.limit_parts, self.limit_number = False, None
        self._angles = np.zeros(6)
        self.joint_names = list(self.robot.link_names)
        self.setCameraPose(camera_pose if camera_pose is not None else RENDERER_FALLBACK_CAMERA_POSE)
        self.setMode(mode)

    # -- scene state ----------------------------------------------------------------------------
    def setJointAngles(self, angles: List[float]):
        self._angles = np.asarray(angles, dtype=np.float64).reshape(6).copy()

    def setCameraPose(self, pose_in: np.ndarray):
        """Camera pose with the reference's pitch convention (render.py:107-111)."""
        self._camera_pose6 = np.asarray(pose_in, dtype=np.float64).copy()
        PV = camera_matrix(self._camera_pose6, self.intrinsics, ZNEAR, ZFAR)
        self.engine.set_camera(PV, self.intrinsics.width, self.intrinsics.height, ZNEAR, ZFAR)

    def setMode(self, mode: str):
        valid_modes = ['seg', 'seg_full', 'real']
        assert mode in valid_modes, f"Mode invalid; must be one of: {valid_modes}"
        self.mode = mode
        self._updateMode()

    def setMaxParts(self, number_of_parts: int):
        """Limit how many links are drawn (render.py:121-129)."""
        if number_of_parts is not None:
            self.limit_parts, self.limit_number = True, int(number_of_parts)
        else:
            self.limit_parts = False
        self._updateMode()

    def _updateMode(self):
        n = min(self.limit_number, NUM_RENDER_LINKS) if self.limit_parts else NUM_RENDER_LINKS
        self._n_render = n
        if self.mode == 'seg':
            self._colors = [DEFAULT_RENDER_COLORS[i] for i in range(n)]
        else:                                       # 'seg_full'; 'real' shades the surface instead of looking colours up
            self._colors = [DEFAULT_RENDER_COLORS[0]] * n
        lut = np.zeros((256, 3), np.uint8)
        for i, c in enumerate(self._colors):
            lut[i] = c
        lut[BACKGROUND_ID] = 0
        self._lut = lut

    # -- output -----------------------------------------------------------------------------------
    @property
    def n_render(self) -> int:
        return self._n_render

    def render_ids(self):
        """-> (depth float32 HxW, link ids uint8 HxW with 255 = background)."""
        return self.engine.render(self._angles, self._n_render)

    def render(self):
        """-> (colour uint8 HxWx3, depth float32 HxW), as pyrender's SEG pass (render.py:92-98)."""
        depth, ids = self.render_ids()
        if self.mode == 'real':
            return shade_depth(depth, ids, self.intrinsics), depth
        return self._lut[ids], depth

    @property
    def resolution(self) -> Tuple[int]:
        return (self.intrinsics.height, self.intrinsics.width)

    @property
    def camera_pose(self) -> np.ndarray:
        return camera_pose_matrix(self._camera_pose6)

    @property
    def color_dict(self) -> dict:
        if self.mode == 'seg':
            return {name: color for name, color in zip(self.joint_names[:self._n_render], self._colors)}
        return {'robot': DEFAULT_RENDER_COLORS[0]}
