"""Dataset with the attribute surface the prediction path reads (reference:
robotpose/data/dataset.py:176-192): length, angles, positions, depthmaps, og_img, camera_pose,
preview_img, intrinsics, attrs — sliceable, `np.copy(ds.x[a:b])` works (predict_dataset.py:39-41).

Two storage forms are read.  The reference's own: `<datasets>/<name>/<name>.h5`, gzip-chunked HDF5 with the groups
of building.py:195-242 — through h5py when importable, else through the HDF5 C library (data/hdf5.py); frames are
decompressed as they are sliced.  And a directory of .npy arrays (memory-mapped) + attrs.json, which needs neither
and is what the synthetic-set generator writes.  The zip -> HDF5 builder is ingest tooling and out of scope
(SURVEY §2 row 13); `write_h5_dataset` / `tools/convert_dataset.py` convert between the two forms.
"""
import json
import os

import numpy as np

from ..config import Paths

_ARRAYS = {'angles': 'angles.npy', 'positions': 'positions.npy', 'depthmaps': 'depthmaps.npy',
           'og_img': 'og_img.npy', 'camera_pose': 'camera_pose.npy', 'preview_img': 'preview_img.npy'}


# dataset attribute -> path inside the HDF5 file (dataset.py:176-192)
_H5_PATHS = {'angles': 'angles', 'positions': 'positions', 'depthmaps': 'coordinates/depthmaps', 'og_img': 'images/original',
             'camera_pose': 'images/camera_poses', 'preview_img': 'images/preview'}


def dataset_dir(name: str) -> str:
    if name.endswith('.h5') and os.path.isfile(name):
        return os.path.dirname(os.path.abspath(name))
    return name if os.path.isabs(name) or os.path.isdir(name) else os.path.join(Paths().DATASETS, name)


def _h5_path(name: str, directory: str):
    if name.endswith('.h5') and os.path.isfile(name):
        return os.path.abspath(name)
    cand = os.path.join(directory, os.path.basename(os.path.normpath(directory)) + '.h5')
    return cand if os.path.isfile(cand) else None


def _open_h5(path: str, mode: str):
    """-> (file object with [] and .attrs, closer).  h5py first: it is what wrote the file."""
    try:
        import h5py
        f = h5py.File(path, mode)
        return f, dict(f.attrs)
    except ImportError:
        from .hdf5 import H5File
        f = H5File(path, 'r' if mode == 'r' else 'r+')
        return f, f.attrs


class Dataset:

    def __init__(self, name: str, rebuild: bool = False, permissions: str = 'r'):
        if rebuild:
            raise NotImplementedError("rebuilding from the raw zip is ingest tooling (out of scope)")
        self.name, self.permissions = name, permissions
        self.dataset_dir = dataset_dir(name)
        self.file = None
        self.dataset_path = _h5_path(name, self.dataset_dir)
        if self.dataset_path is None and not os.path.isfile(os.path.join(self.dataset_dir, 'attrs.json')):
            raise ValueError(f"The requested dataset is not available: {self.dataset_dir} holds neither "
                             f"{os.path.basename(os.path.normpath(self.dataset_dir))}.h5 nor attrs.json")
        self.load()

    def load(self):
        mode = 'r' if self.permissions == 'r' else 'r+'
        if os.path.isfile(os.path.join(self.dataset_dir, 'attrs.json')) and not (self.name.endswith('.h5') and self.dataset_path):
            with open(os.path.join(self.dataset_dir, 'attrs.json')) as f:
                self.attrs = json.load(f)
            for attr, fn in _ARRAYS.items():
                path = os.path.join(self.dataset_dir, fn)
                setattr(self, attr, np.load(path, mmap_mode=mode) if os.path.exists(path) else None)
        else:
            self.file, self.attrs = _open_h5(self.dataset_path, mode)
            for attr, path in _H5_PATHS.items():
                setattr(self, attr, self.file[path] if path in self.file else None)
            if isinstance(self.attrs.get('color_dict'), str):          # a dict has no HDF5 form: stored as JSON text
                self.attrs['color_dict'] = json.loads(self.attrs['color_dict'])
        self.length = int(self.attrs['length'])
        self.og_resolution = self.attrs.get('resolution')
        self.intrinsics = str(self.attrs['color_intrinsics'])

    def close(self):
        if self.file is not None:
            self.file.close()
            self.file = None

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


class _LazyFrames:
    """Sequence of frames rendered when they are read: ds.og_img[i], ds.og_img[a:b] (a fresh array each time, like an HDF5 slice)."""

    def __init__(self, owner, what: int, shape, dtype):
        self._owner, self._what, self.shape, self.dtype = owner, what, (owner.length,) + tuple(shape), np.dtype(dtype)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        if isinstance(key, slice):
            idx = range(*key.indices(self.shape[0]))
            out = np.empty((len(idx),) + self.shape[1:], self.dtype)
            for k, i in enumerate(idx):
                out[k] = self._owner.frame(i)[self._what]
            return out
        i = int(key)
        if i < 0:
            i += self.shape[0]
        if not 0 <= i < self.shape[0]:
            raise IndexError(i)
        return self._owner.frame(i)[self._what].astype(self.dtype, copy=True)


class SyntheticDataset:
    """The frames make_synthetic_dataset(seed=...) would write, rendered on the GPU as they are read instead of stored: frame f uses
    default_rng(seed + f), pose uniform in the joint limits of `do_angles`.  Same attribute surface as Dataset; a
    10 000-frame 640x480 set (BASELINE configs[3]) is 34 GB on disk and nothing here.
    Name form for the callers that take a dataset name: 'synthetic:<frames>[:<seed>[:<intrinsics preset>]]'."""

    PREFIX = 'synthetic:'

    def __init__(self, n_frames: int, base_intrin: str = '640_480_color', camera_pose=None, do_angles: str = 'SLU', seed: int = 7919,
                 device: int = 0):
        import threading
        from ..constants import DEFAULT_CAMERA_POSE
        from ..simulation.render import Renderer
        from ..utils import str_to_arr
        pose = np.asarray(DEFAULT_CAMERA_POSE if camera_pose is None else camera_pose, float)
        self._r = Renderer('seg', pose, base_intrin, device=device)
        self._lock = threading.Lock()                    # one engine context: frames are rendered one at a time
        self._last = (None, None)
        lim = self._r.robot.joint_limits
        self.length, self.name = int(n_frames), f'{self.PREFIX}{n_frames}:{seed}:{base_intrin}'
        self.angles = np.stack([np.random.default_rng(seed + f).uniform(lim[:, 0], lim[:, 1]) * str_to_arr(do_angles) for f in range(n_frames)])
        self.camera_pose = np.tile(pose, (n_frames, 1))
        self.positions = np.zeros((n_frames, 6, 3))
        H, W = self._r.resolution
        self.og_img, self.depthmaps = _LazyFrames(self, 0, (H, W, 3), np.uint8), _LazyFrames(self, 1, (H, W), np.float64)
        self.preview_img = None
        self.intrinsics = str(self._r.intrinsics)
        self.og_resolution = [H, W]
        self.attrs = {'name': self.name, 'length': self.length, 'resolution': [H, W], 'color_intrinsics': self.intrinsics,
                      'synthetic': True, 'color_dict': self._r.color_dict}
        self.dataset_dir = self.name

    @classmethod
    def from_name(cls, name: str, device: int = 0):
        parts = name[len(cls.PREFIX):].split(':')
        return cls(int(parts[0]), parts[2] if len(parts) > 2 else '640_480_color', seed=int(parts[1]) if len(parts) > 1 else 7919, device=device)

    def frame(self, i: int):
        with self._lock:
            if self._last[0] != i:                       # colour and depth of a frame are asked for one after the other
                self._r.setJointAngles(self.angles[i])
                self._last = (i, self._r.render())
            return self._last[1]

    def close(self):
        pass

    def __len__(self) -> int:
        return self.length


def open_dataset(name: str, device: int = 0):
    """Dataset(name), or the frames of a 'synthetic:<frames>[:<seed>[:<preset>]]' name rendered on the fly."""
    if isinstance(name, str) and name.startswith(SyntheticDataset.PREFIX):
        return SyntheticDataset.from_name(name, device)
    return Dataset(name)


def write_h5_dataset(name: str, og_img, depthmaps, angles, camera_pose, color_intrinsics: str, positions=None,
                     extra_attrs: dict = None, compression_level: int = 4) -> str:
    """The same arrays as one `<name>/<name>.h5` in the reference's layout (building.py:195-242): what the reference's
    own Dataset class opens.  Needs the HDF5 C library (data/hdf5.py)."""
    from . import hdf5
    d = dataset_dir(name)
    os.makedirs(d, exist_ok=True)
    attrs = {}
    for k, v in (extra_attrs or {}).items():
        attrs[k] = json.dumps(v) if isinstance(v, dict) else (int(v) if isinstance(v, bool) else v)
    path = os.path.join(d, os.path.basename(os.path.normpath(d)) + '.h5')
    return hdf5.write_h5_dataset(path, og_img, depthmaps, angles, camera_pose, color_intrinsics, positions, attrs=attrs,
                                 compression_level=compression_level)
