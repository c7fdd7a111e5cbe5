import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rope_s3d_amd import SyntheticPredictor
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
from rope_s3d_amd.prediction import predict as P
intr = sys.argv[1] if len(sys.argv) > 1 else '1280_720_color'
ds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
div = int(sys.argv[3]) if len(sys.argv) > 3 else 25
sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, intr, ds, 'SLU', noise=False, seed=1, lookup_divisions=div)
p = sp.predictor
lim = sp.urdf_reader.joint_limits
poses = [np.random.default_rng(7919 + f).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0]) for f in range(50)]
frames = []
for q in poses:
    sp.renderer.setJointAngles(q); frames.append(sp.renderer.render())
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] = acc.get(name, 0) + time.perf_counter() - t; acc[name+'_n'] = acc.get(name+'_n', 0) + 1; return r
    setattr(obj, name, g)
p.NATIVE = os.environ.get('ROPE_NATIVE', '1') != '0'
for n in ('_loadSynthetic', '_stage_lookup', '_stage_descent', '_stage_sflip', '_stage_isweep', '_errors', '_downsample', '_upload_target', '_native_stages'):
    wrap(p, n)
wrap(p.engine, 'predict')
wrap(p.engine, 'lookup_score')
wrap(p.engine, 'set_target')
p.run(*frames[0])
acc.clear()
t0 = time.perf_counter()
for c, d in frames: p.run(c, d)
tot = time.perf_counter() - t0
print(f"total {tot/50*1e3:.2f} ms/frame")
for k in sorted(acc):
    if not k.endswith('_n'): print(f"{k:18s} {acc[k]/50*1e3:7.3f} ms/frame  calls/frame {acc[k+'_n']/50:.1f}")
