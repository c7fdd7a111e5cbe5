import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import numpy as np
from rope_s3d_amd import SyntheticPredictor
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '1280_720_color', 8, 'SLU', noise=False, seed=1)
lim = sp.urdf_reader.joint_limits
poses = [np.random.default_rng(7919 + f).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0]) for f in range(20)]
sp.run(poses[0])
frames = []
for q in poses:
    sp.renderer.setJointAngles(q); frames.append(sp.renderer.render())
pr = cProfile.Profile(); pr.enable()
for c, d in frames: sp.predictor.run(c, d)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(s.getvalue()[:4500])
