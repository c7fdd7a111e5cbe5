#!/usr/bin/env python3
"""Camera-pose path on the GPU: batch throughput of rope_eval_views and end-to-end time of both predictors.

    python tools/bench_camera.py            # prints one JSON object

(a) throughput: 640x480, 16 frames x 256 trial cameras = 4096 (view, frame) candidates per call, TSWEEP and CAMFULL
    sums, planes resident in HBM; (b) end to end: the default stage lists at the reference's default 1280x720 / 8 on
    10 synthetic frames (the reference renders views x frames GL passes one at a time)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from rope_s3d_amd import CameraPredictor, ModellessCameraPredictor, Renderer, engine as eng
    from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
    from rope_s3d_amd.prediction import camera_pose_prediction as cpp
    from rope_s3d_amd.projection import view_matrix
    from rope_s3d_amd.segmentation import ColorSegmenter

    out = {}
    true_pose = np.array(DEFAULT_CAMERA_POSE, float) + np.array([.06, -.05, .04, .01, -.015, .02])

    def frames(r, n, seed):
        rng = np.random.default_rng(seed)
        lim = r.robot.joint_limits
        qs = rng.uniform(lim[:, 0], lim[:, 1], (n, 6)) * np.array([1, 1, 1, 0, 0, 0])
        r.setCameraPose(true_pose)
        cs, ds, ids = [], [], []
        for q in qs:
            r.setJointAngles(q)
            d, i = r.render_ids()
            cs.append(r._lut[i]); ds.append(d.astype(np.float64)); ids.append(i)
        return qs, np.stack(cs), np.stack(ds), np.stack(ids)

    # ---- (a) batch throughput at 640x480
    r = Renderer('seg', DEFAULT_CAMERA_POSE, '640_480_color')
    N, K = 16, 256
    qs, cs, ds, ids = frames(r, N, 7919)
    e = r.engine
    planes = np.zeros((N, 6) + ds.shape[1:], np.uint64)
    for i in range(N):
        for l in range(6):
            m = ids[i] == l
            planes[i, l] = eng.pack_target(m * ds[i], m.astype(np.uint64))
    e.set_frames(qs, np.stack([eng.pack_target(d) for d in ds]), ds.astype(np.float32), planes)
    P = r.intrinsics.gl_projection(ZNEAR, ZFAR)
    rng = np.random.default_rng(1)
    poses = true_pose + rng.uniform(-.1, .1, (K, 6)) * np.array([1, 1, 1, .3, .3, .3])
    PV = np.stack([P @ view_matrix(p) for p in poses])
    for name, loss in (('tsweep', eng.LOSS_TSWEEP), ('camfull', eng.LOSS_CAMFULL)):
        e.eval_views(PV, 6, loss)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            e.eval_views(PV, 6, loss)
        dt = (time.perf_counter() - t0) / reps
        out[f'views_{name}_640x480'] = {'candidates_per_call': K * N, 'ms_per_call': dt * 1e3, 'renders_per_s': K * N / dt}

    # ---- (b) end to end at 1280x720 / 8
    rb = Renderer('seg', DEFAULT_CAMERA_POSE, '1280_720_color')
    qs, cs, ds, ids = frames(rb, 10, 4242)
    start = np.array(DEFAULT_CAMERA_POSE, float)
    for name, mk in (('modelless', lambda: ModellessCameraPredictor(start, 8)),
                     ('segmented', lambda: CameraPredictor(start, 8, segmenter=ColorSegmenter(['BG'] + rb.robot.link_names[:6])))):
        p = mk()
        p.run(cs, ds, qs)                       # warm-up (buffers, first launches)
        p.evaluations = 0
        t0 = time.perf_counter()
        got = p.run(cs, ds, qs)
        dt = time.perf_counter() - t0
        out[f'{name}_run_160x90_10frames'] = {'seconds': dt, 'renders': p.evaluations, 'renders_per_s': p.evaluations / dt,
                                              'final_pose': [float(x) for x in got]}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
