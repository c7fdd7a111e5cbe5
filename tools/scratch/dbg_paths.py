import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import helpers
from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import ZFAR, ZNEAR
rb = helpers.robot()
intr, PV = helpers.camera('640_480_color')
o = helpers.make_oracle(rb, intr, PV)
e = eng.Engine(0); e.set_robot(rb); e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
d, ids = o.render([0.35, 0.45, 0.9, 0, 0, 0])
tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
e.set_target(tq, t32, flags)
cand = helpers.slu_grid(rb.joint_limits, 4)
_, ref = o.eval(cand, eng.LOSS_DEPTH, 6, tq, t32, None, flags, threads=16, want_sums=True)
for name, st, c in [('layers', 0, cand), ('split', e.NO_LAYERS, cand), ('plain', e.NO_LAYERS | e.NO_SPLIT, cand), ('one split', 0, cand[:1]), ('one plain', e.NO_SPLIT, cand[:1])]:
    e.set_strategy(st)
    _, s, _, _ = e.eval(c, 6, eng.LOSS_DEPTH, None, want_sums=True)
    bad = np.nonzero((s[:, :5] != ref[:len(c), :5]).any(1))[0]
    print(name, 'bad rows', len(bad), 'of', len(c), 'first', bad[:5], 'cnt diff', (s[bad[:5], 0].astype(np.int64) - ref[bad[:5], 0].astype(np.int64)))
