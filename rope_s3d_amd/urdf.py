"""URDF reader for the active robot.

Keeps the reference's `URDFReader` surface (robotpose/urdf.py:25-100): the first
seven <link> visual mesh paths and names, and the first six <joint> limits.
It additionally reads each joint's origin xyz/rpy and axis, which the reference
leaves to Klampt's URDF importer (robotpose/simulation/kinematics.py:25-27).
"""
import os
import xml.etree.ElementTree as ET
from typing import List

import numpy as np

from .config import Paths


def _floats(text: str) -> List[float]:
    return [float(x) for x in text.split()]


class URDFReader:

    def __init__(self, path: str = None):
        p = Paths()
        self._urdfs_dir = p.URDFS
        self.internal_path = path if path is not None else p.URDF
        if not os.path.isabs(self.internal_path):
            self.internal_path = os.path.join(os.path.dirname(self._urdfs_dir.rstrip('/')), self.internal_path)
        self.load()

    def load(self):
        root = ET.parse(self.internal_path).getroot()
        links = root.findall('link')[:7]
        self.mesh_paths = []
        self.mesh_names = []
        for link in links:
            fn = link.find('visual').find('geometry').find('mesh').get('filename')
            # urdf.py:54 - strip the ROS package scheme and anchor under the urdfs folder
            self.mesh_paths.append(os.path.join(self._urdfs_dir, fn.replace('package://', '')))
            self.mesh_names.append(link.get('name'))

        joints = root.findall('joint')[:6]
        self.joint_limits = np.array(
            [[float(j.find('limit').get('lower')), float(j.find('limit').get('upper'))] for j in joints])

        # What Klampt reads from the same elements (kinematics.py:25-52)
        self.joint_names = [j.get('name') for j in joints]
        self.joint_parents = [j.find('parent').get('link') for j in joints]
        self.joint_children = [j.find('child').get('link') for j in joints]
        self.joint_origins = np.zeros((6, 3))
        self.joint_rpy = np.zeros((6, 3))
        self.joint_axes = np.zeros((6, 3))
        for i, j in enumerate(joints):
            o = j.find('origin')
            if o is not None:
                self.joint_origins[i] = _floats(o.get('xyz', '0 0 0'))
                self.joint_rpy[i] = _floats(o.get('rpy', '0 0 0'))
            a = j.find('axis')
            self.joint_axes[i] = _floats(a.get('xyz')) if a is not None else [1, 0, 0]
            if j.get('type') not in ('revolute', 'continuous'):
                raise ValueError(f"joint {j.get('name')}: only revolute joints are supported")
        # the serial chain base_link -> link_1_s -> ... is assumed by the engine
        for i in range(6):
            if self.joint_parents[i] != self.mesh_names[i] or self.joint_children[i] != self.mesh_names[i + 1]:
                raise ValueError("URDF joints 1..6 must chain the first seven links in order")
        for link in links:
            o = link.find('visual').find('origin')
            if o is not None and (any(_floats(o.get('xyz', '0 0 0'))) or any(_floats(o.get('rpy', '0 0 0')))):
                raise ValueError("non-zero <visual><origin> is not supported")

    @property
    def path(self) -> str:
        return self.internal_path

    @property
    def name(self) -> str:
        return os.path.basename(os.path.normpath(self.internal_path)).replace('.urdf', '')

    @property
    def available_paths(self) -> List[str]:
        return sorted(os.path.join(r, x) for r, _, files in os.walk(self._urdfs_dir) for x in files if x.endswith('.urdf'))

    @property
    def available_names(self) -> List[str]:
        return [os.path.basename(x).replace('.urdf', '') for x in self.available_paths]
