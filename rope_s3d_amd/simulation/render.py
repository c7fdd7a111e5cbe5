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
from ..projection import Intrinsics, camera_pose_matrix, view_matrix
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
        PV = self.intrinsics.gl_projection(ZNEAR, ZFAR) @ view_matrix(self._camera_pose6)
        self.engine.set_camera(PV, self.intrinsics.width, self.intrinsics.height, ZNEAR, ZFAR)

    def setMode(self, mode: str):
        valid_modes = ['seg', 'seg_full', 'real']
        assert mode in valid_modes, f"Mode invalid; must be one of: {valid_modes}"
        if mode == 'real':
            raise NotImplementedError("'real' (lit, textured) rendering is not part of the prediction path")
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
        else:
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
