"""Image bounds of the robot for n visible links (reference: robotpose/crop.py:27-178).

The reference renders a pose grid per link count with pyrender, sums the depth images and
takes the padded bounding box, caching the result in lookup/crop_data.h5.  Here the grid
is one rope_coverage call per link count (OR of "pixel covered" on the GPU) and the cache
is an in-process dict plus an optional .npy file.
"""
import os
from typing import List, Union

import numpy as np

from .config import Paths
from .constants import (CROP_MAX_PER_JOINT, CROP_PADDING, CROP_RENDER_WEIGHTING, CROP_SEC_ALLOTTED_APPROX,
                        CROP_VARYING, MAX_LINKS)
from .projection import Intrinsics
from .utils import get_extremes, str_to_arr

_cache = {}


def crop_pose_grid(joint_limits: np.ndarray, intrinsics_size: int, num_links: int):
    """Poses rendered to find the crop for `num_links` visible links (crop.py:114-146)."""
    w = np.array(CROP_RENDER_WEIGHTING[:num_links - 1], dtype=float)
    w = w / np.sum(w)
    num_poses = CROP_SEC_ALLOTTED_APPROX / (intrinsics_size * 1.2 * (10 ** -8) + .002)
    nz = w[w != 0]
    base_div = w * ((num_poses / np.prod(nz)) ** (1 / len(nz)))
    base_div[base_div < 1] = 1
    base_div[base_div > CROP_MAX_PER_JOINT] = CROP_MAX_PER_JOINT
    base_div = base_div.astype(int)
    divisions = np.ones((6,), dtype=int)
    divisions[:num_links - 1] = base_div
    num = int(np.prod(divisions))
    angles = np.zeros((num, 6))
    for idx in np.where(str_to_arr(CROP_VARYING))[0]:
        rng = np.linspace(joint_limits[idx, 0], joint_limits[idx, 1], divisions[idx])
        repeat = int(np.prod(divisions[:idx]))
        tile = num // (repeat * divisions[idx])
        angles[:, idx] = np.tile(np.repeat(rng, repeat), tile)
    return angles, divisions


class Crop:

    def __init__(self, camera_pose: np.ndarray, intrinsics: Union[str, Intrinsics], renderer=None, use_disk_cache: bool = True):
        from .simulation.render import Renderer
        self.intrinsics = Intrinsics(intrinsics)
        self._pose = np.asarray(camera_pose, dtype=np.float64)
        self._renderer = renderer
        urdf_name = (renderer.robot.name if renderer is not None else None)
        if urdf_name is None:
            from .urdf import URDFReader
            urdf_name = URDFReader().name
        self.name = f'{urdf_name}/{self._list_to_str(self._pose)}/{self.intrinsics}'
        if self.name in _cache:
            self.data = _cache[self.name].copy()
            return
        path = self._disk_path() if use_disk_cache else None
        if path and os.path.exists(path):
            self.data = np.load(path)
        else:
            if self._renderer is None:
                self._renderer = Renderer('seg', self._pose, self.intrinsics)
            self.data = self._create()
            if path:
                try:
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    np.save(path, self.data)
                except OSError:
                    pass
        _cache[self.name] = self.data.copy()

    def _disk_path(self) -> str:
        import hashlib
        h = hashlib.sha1(self.name.encode()).hexdigest()[:16]
        return os.path.join(Paths().ROBOT_LOOKUPS, f'crop_{h}.npy')

    @staticmethod
    def _list_to_str(lst) -> str:
        return "[" + "".join(f" {item:.4f}" for item in lst) + " ]"

    def _create(self) -> np.ndarray:
        r = self._renderer
        limits = r.robot.joint_limits
        data = np.zeros((MAX_LINKS, 4), int)
        cover = r.engine.coverage(np.zeros((1, 6)), 1)          # base only, home pose (crop.py:55-58)
        data[1] = self._calculate_crop(cover != 0)
        for num_links in range(2, MAX_LINKS):
            angles, _ = crop_pose_grid(limits, self.intrinsics.size, num_links)
            cover = np.zeros((self.intrinsics.height, self.intrinsics.width), bool)
            for s in range(0, len(angles), 32768):
                cover |= r.engine.coverage(angles[s:s + 32768], num_links) != 0
            data[num_links] = self._calculate_crop(cover)
        data[0] = data[-1]                                        # crop.py:84
        return data

    def _calculate_crop(self, covered: np.ndarray) -> List[int]:
        e = get_extremes(covered)
        return [max(e[0] - CROP_PADDING, 0), min(e[1] + CROP_PADDING, self.intrinsics.height - 1),
                max(e[2] - CROP_PADDING, 0), min(e[3] + CROP_PADDING, self.intrinsics.width - 1)]

    def __getitem__(self, key: int) -> np.ndarray:
        return self.data[0 if key is None else key]

    def size(self, n: int) -> int:
        c = self.data[n]
        return int((c[1] - c[0]) * (c[3] - c[2]))


def applyCrop(mat: np.ndarray, crop) -> np.ndarray:
    return mat[crop[0]:crop[1] + 1, crop[2]:crop[3] + 1]


def applyBatchCrop(mat: np.ndarray, crop) -> np.ndarray:
    return mat[:, crop[0]:crop[1] + 1, crop[2]:crop[3] + 1]
