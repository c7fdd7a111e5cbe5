"""Host-side forward kinematics with the reference's surface (robotpose/simulation/kinematics.py:17-55).

The engine computes the same chain on the device for every candidate; this numpy mirror
exists for callers that want link poses on the host (`ForwardKinematics.calc`).
"""
from typing import Union

import numpy as np

from ..robot import RobotModel
from ..urdf import URDFReader


class ForwardKinematics:

    def __init__(self, robot: RobotModel = None):
        self.load(robot)

    def load(self, robot: RobotModel = None):
        self.reader = URDFReader()
        # full chain (7 links) regardless of how many are rendered
        self.fixed = np.zeros((6, 3, 4))
        from ..robot import _rpy_matrix
        for i in range(6):
            self.fixed[i, :, :3] = _rpy_matrix(self.reader.joint_rpy[i]) if np.any(self.reader.joint_rpy[i]) else np.eye(3)
            self.fixed[i, :, 3] = self.reader.joint_origins[i]
        ax = self.reader.joint_axes
        self.axes = ax / np.linalg.norm(ax, axis=1, keepdims=True)

    def calc(self, p_in: Union[list, np.ndarray]) -> np.ndarray:
        """Joint angles (6,) -> (7,4,4) link poses, base_link first (identity)."""
        q = np.asarray(p_in, dtype=np.float64).reshape(6)
        poses = np.zeros((7, 4, 4))
        T = np.eye(4)
        poses[0] = T
        for i in range(6):
            a, s, c = self.axes[i], np.sin(q[i]), np.cos(q[i])
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            R = c * np.eye(3) + s * K + (1 - c) * np.outer(a, a)
            A = np.eye(4)
            A[:3, :3] = self.fixed[i, :, :3] @ R
            A[:3, 3] = self.fixed[i, :, 3]
            T = T @ A
            poses[i + 1] = T
        return poses
