import cProfile, pstats, io, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from rope_s3d_amd import CameraPredictor, Renderer
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
from rope_s3d_amd.segmentation import ColorSegmenter
true_pose = np.array(DEFAULT_CAMERA_POSE, float) + np.array([.06, -.05, .04, .01, -.015, .02])
rb = Renderer('seg', DEFAULT_CAMERA_POSE, '1280_720_color')
rng = np.random.default_rng(4242); lim = rb.robot.joint_limits
qs = rng.uniform(lim[:, 0], lim[:, 1], (10, 6)) * np.array([1, 1, 1, 0, 0, 0])
rb.setCameraPose(true_pose); cs, ds = [], []
for q in qs:
    rb.setJointAngles(q); d, i = rb.render_ids(); cs.append(rb._lut[i]); ds.append(d.astype(np.float64))
cs, ds = np.stack(cs), np.stack(ds)
p = CameraPredictor(np.array(DEFAULT_CAMERA_POSE, float), 8, segmenter=ColorSegmenter(['BG'] + rb.robot.link_names[:6]))
p.run(cs, ds, qs)
pr = cProfile.Profile(); pr.enable(); p.run(cs, ds, qs); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3800])
