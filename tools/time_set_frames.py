import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
from rope_s3d_amd.projection import Intrinsics, camera_matrix
from rope_s3d_amd.robot import RobotModel
robot = RobotModel.from_urdf()
intr = Intrinsics('1280_720_color'); intr.downscale(8)
PV = camera_matrix(DEFAULT_CAMERA_POSE, intr, ZNEAR, ZFAR)
e = eng.Engine(0); e.set_robot(robot); e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
N = 10
q = np.zeros((N, 6)); tq = np.zeros((N, 90, 160), np.uint64); t32 = np.zeros((N, 90, 160), np.float32); tl = np.zeros((N, 6, 90, 160), np.uint64)
for name, args in (('tq only', (q, tq)), ('tq+t32', (q, tq, t32)), ('all', (q, tq, t32, tl))):
    e.set_frames(*args)
    t0 = time.perf_counter()
    for _ in range(5): e.set_frames(*args)
    print(name, (time.perf_counter() - t0) / 5 * 1e3, 'ms')
rng = np.random.default_rng(0)
for rep in range(3):
    planes = rng.integers(0, 2**40, (6, 90, 160), dtype=np.uint64)
    tl2 = np.tile(planes[None], (N, 1, 1, 1))
    tq2 = rng.integers(0, 2**39, (N, 90, 160), dtype=np.uint64)
    t0 = time.perf_counter(); e.set_frames(q, tq2, tq2.astype(np.float32), tl2); print('fresh arrays', (time.perf_counter() - t0) * 1e3, 'ms')
    PVs = np.stack([PV] * 6)
    t0 = time.perf_counter(); e.eval_views(PVs, 6, eng.LOSS_CAMFULL); print('  first eval_views after it', (time.perf_counter() - t0) * 1e3, 'ms')
    t0 = time.perf_counter(); e.eval_views(PVs, 6, eng.LOSS_CAMFULL); print('  second', (time.perf_counter() - t0) * 1e3, 'ms')
