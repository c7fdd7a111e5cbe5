"""Dataset with the attribute surface the prediction path reads (reference:
robotpose/data/dataset.py:176-192): length, angles, positions, depthmaps, og_img, camera_pose,
preview_img, intrinsics, attrs — sliceable, `np.copy(ds.x[a:b])` works (predict_dataset.py:39-41).

Storage is a directory of .npy arrays (memory-mapped on load) + attrs.json instead of the
reference's gzip HDF5 (h5py is not available in this image); the zip -> HDF5 builder is
ingest tooling and out of scope.
"""
import json
import os

import numpy as np

from ..config import Paths

_ARRAYS = {'angles': 'angles.npy', 'positions': 'positions.npy', 'depthmaps': 'depthmaps.npy',
           'og_img': 'og_img.npy', 'camera_pose': 'camera_pose.npy', 'preview_img': 'preview_img.npy'}


def dataset_dir(name: str) -> str:
    return name if os.path.isabs(name) or os.path.isdir(name) else os.path.join(Paths().DATASETS, name)


class Dataset:

    def __init__(self, name: str, rebuild: bool = False, permissions: str = 'r'):
        if rebuild:
            raise NotImplementedError("rebuilding from the raw zip is ingest tooling (out of scope)")
        self.name, self.permissions = name, permissions
        self.dataset_dir = dataset_dir(name)
        if not os.path.isfile(os.path.join(self.dataset_dir, 'attrs.json')):
            raise ValueError(f"The requested dataset is not available: {self.dataset_dir}")
        self.load()

    def load(self):
        with open(os.path.join(self.dataset_dir, 'attrs.json')) as f:
            self.attrs = json.load(f)
        mode = 'r' if self.permissions == 'r' else 'r+'
        for attr, fn in _ARRAYS.items():
            path = os.path.join(self.dataset_dir, fn)
            setattr(self, attr, np.load(path, mmap_mode=mode) if os.path.exists(path) else None)
        self.length = int(self.attrs['length'])
        self.og_resolution = self.attrs.get('resolution')
        self.intrinsics = self.attrs['color_intrinsics']

    def __len__(self) -> int:
        return self.length

    def __repr__(self) -> str:
        return f"RobotPose dataset located at {self.dataset_dir}."


def write_dataset(name: str, og_img: np.ndarray, depthmaps: np.ndarray, angles: np.ndarray, camera_pose: np.ndarray,
                  color_intrinsics: str, positions: np.ndarray = None, extra_attrs: dict = None) -> str:
    """Write arrays in the layout `Dataset` reads.  depthmaps are float64 metres (building.py:172-179)."""
    d = dataset_dir(name)
    os.makedirs(d, exist_ok=True)
    n = len(og_img)
    np.save(os.path.join(d, 'og_img.npy'), np.asarray(og_img, np.uint8))
    np.save(os.path.join(d, 'depthmaps.npy'), np.asarray(depthmaps, np.float64))
    np.save(os.path.join(d, 'angles.npy'), np.asarray(angles, np.float64))
    np.save(os.path.join(d, 'camera_pose.npy'), np.asarray(camera_pose, np.float64))
    np.save(os.path.join(d, 'positions.npy'), np.zeros((n, 6, 3)) if positions is None else np.asarray(positions, np.float64))
    attrs = {'name': os.path.basename(os.path.normpath(d)), 'length': int(n),
             'resolution': [int(og_img.shape[1]), int(og_img.shape[2])], 'color_intrinsics': str(color_intrinsics)}
    attrs.update(extra_attrs or {})
    with open(os.path.join(d, 'attrs.json'), 'w') as f:
        json.dump(attrs, f, indent=1)
    return d


def make_synthetic_dataset(name: str, n_frames: int, base_intrin: str = '640_480_color', camera_pose=None,
                           do_angles: str = 'SLU', seed: int = 7919, device: int = 0) -> str:
    """Synthetic RGB-D frames rendered by the engine: frame f uses default_rng(seed+f), pose uniform in the
    joint limits of `do_angles`, colour = flat link colours, depth = metric float64."""
    from ..constants import DEFAULT_CAMERA_POSE
    from ..simulation.render import Renderer
    from ..utils import str_to_arr
    pose = np.asarray(DEFAULT_CAMERA_POSE if camera_pose is None else camera_pose, float)
    r = Renderer('seg', pose, base_intrin, device=device)
    lim = r.robot.joint_limits
    H, W = r.resolution
    og, dm, ang = np.zeros((n_frames, H, W, 3), np.uint8), np.zeros((n_frames, H, W)), np.zeros((n_frames, 6))
    for f in range(n_frames):
        q = np.random.default_rng(seed + f).uniform(lim[:, 0], lim[:, 1]) * str_to_arr(do_angles)
        r.setJointAngles(q)
        og[f], dm[f] = r.render()
        ang[f] = q
    return write_dataset(name, og, dm, ang, np.tile(pose, (n_frames, 1)), str(r.intrinsics),
                         extra_attrs={'synthetic': True, 'color_dict': r.color_dict})
