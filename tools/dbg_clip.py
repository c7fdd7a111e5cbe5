"""Which rows / words differ between the plain and the clipping instantiations on the bench grid (debugging aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir, 'tests'))
import helpers
from rope_s3d_amd import engine as eng
from rope_s3d_amd.constants import ZFAR, ZNEAR

rb = helpers.robot()
intr, PV = helpers.camera('640_480_color')
e = eng.Engine(0)
e.set_robot(rb)
e.set_camera(PV, intr.width, intr.height, ZNEAR, ZFAR)
q = np.array([0.4, 0.3, 0.8, 0, 0, 0])
depth, ids = e.render(q, 6)
tq, t32, flags, *_ = helpers.synthetic_target(depth, ids)
e.set_target(tq, t32, flags)
cand = helpers.slu_grid(rb.joint_limits, 16)
for loss in (eng.LOSS_DEPTH, eng.LOSS_FULL):
    _, a, _, _ = e.eval(cand, 6, loss, want_sums=True)
    for flag in (16, 16 | 8, 16 | 1, 16 | 1 | 8, 16, 0):
        e.set_strategy(flag)
        _, b, _, _ = e.eval(cand, 6, loss, want_sums=True)
        _, b2, _, _ = e.eval(cand, 6, loss, want_sums=True)
        e.set_strategy(0)
        bad = np.where((a != b).any(1))[0]
        print(f"loss {loss} flag {flag}: {len(bad)} rows differ; repeat equal: {np.array_equal(b, b2)}; rows {bad[:12]}; "
              f"words {sorted(set(np.where(a != b)[1]))[:8]}")
        if len(bad):
            r = bad[0]
            print("   ", a[r][:6].astype(np.int64) - b[r][:6].astype(np.int64), "q0 groups:", sorted(set((bad % 16).tolist()))[:16], sorted(set(((bad // 16) % 16).tolist()))[:16])
