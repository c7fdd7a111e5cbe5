#!/usr/bin/env python3
"""Generates tests/golden/hotpath_160x120.npz from the CPU oracle.

The reference ships no fixtures for this path and cannot run here, so the
golden vectors are outputs of the oracle (oracle/rope_oracle.c + oracle/predictor_ref.py),
whose rules are themselves pinned by the closed-form tests in tests/test_oracle_pins.py.
They freeze the arithmetic contract: any change to either implementation that moves a bit
shows up against this file.      python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, os.pardir, os.pardir)))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, os.pardir)))

import helpers  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import predictor_ref  # noqa: E402
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, LINK_BLUE  # noqa: E402
from rope_s3d_amd.imgproc import resize_linear  # noqa: E402

POSES = np.array([[0, 0, 0, 0, 0, 0], [0.3, 0.4, 0.5, 0, 0, 0], [-0.7, -0.9, 2.2, 0, 0, 0], [1.5, 1.2, -0.8, 0.5, -0.7, 0.3]], float)
TARGET_POSE = np.array([0.35, 0.45, 0.9, 0, 0, 0])
LOOKUP_CROP = np.array([20, 119, 30, 150], np.int32)
FRAME_SEED = 7919


def main():
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=4)
    o = helpers.make_oracle(rb, intr, PV)
    out = {'poses': POSES, 'target_pose': TARGET_POSE, 'lookup_crop': LOOKUP_CROP}
    for n in (4, 6):
        out[f'keys_n{n}'] = np.stack([o.raster_key(q, n) for q in POSES])
    d, ids = o.render(TARGET_POSE)
    tq, t32, flags, *_ = helpers.synthetic_target(d, ids)
    cand = helpers.slu_grid(rb.joint_limits, 4)
    out['candidates'] = cand
    for name, loss, n, crop in (('full6', orc.LOSS_FULL, 6, None), ('full4', orc.LOSS_FULL, 4, None),
                                ('depth6', orc.LOSS_DEPTH, 6, None), ('lookup6', orc.LOSS_LOOKUP, 6, LOOKUP_CROP)):
        err, sums = o.eval(cand, loss, n, tq, t32, crop, flags, threads=4, want_sums=True)
        out[f'err_{name}'], out[f'sums_{name}'] = err, sums

    # one full prediction: frame rendered at 640x480, predicted at 160x120, 4^3 lookup grid
    intr_full, PV_full = helpers.camera('640_480_color')
    o_full = helpers.make_oracle(rb, intr_full, PV_full)
    lim = rb.joint_limits
    q_true = np.random.default_rng(FRAME_SEED).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
    depth, ids_full = o_full.render(q_true)
    blue = np.where(ids_full == 255, 0, LINK_BLUE[np.minimum(ids_full, 6)]).astype(np.uint8)
    tgt_depth = resize_linear(depth, intr.width, intr.height).astype(np.float64)
    tgt_blue = resize_linear(blue, intr.width, intr.height)
    names = rb.link_names
    link_blue = {nme: int(LINK_BLUE[i]) for i, nme in enumerate(names)}
    # crop of six visible links at 160x120 as the product's Crop computes it (checked against the oracle in
    # tests/test_gpu_predictor.py::test_crop_matches_oracle)
    from rope_s3d_amd.crop import crop_pose_grid
    cover = helpers.make_oracle(rb, *helpers.camera('640_480_color', ds=4, as_predictor=True)).coverage(crop_pose_grid(lim, intr.size, 6)[0], 6, threads=8) != 0
    r, c = np.where(cover)
    crop6 = np.array([max(r.min() - 10, 0), min(r.max() + 10, intr.height - 1), max(c.min() - 10, 0), min(c.max() + 10, intr.width - 1)], np.int32)
    # Predictor's renderer works with the intrinsics as rebuilt from their six-digit string form (helpers.camera)
    intr_p, PV_p = helpers.camera('640_480_color', ds=4, as_predictor=True)
    o_p = helpers.make_oracle(rb, intr_p, PV_p)
    final, trace, n_eval = predictor_ref.predict_reference(o_p, tgt_depth, tgt_blue, names, link_blue, lim, DEFAULT_CAMERA_POSE,
                                                           cand, crop6, 'SLU')
    out.update(frame_q_true=q_true, frame_crop6=crop6, frame_final=final, frame_trace=np.stack([a for _, a in trace]),
               frame_evaluations=np.int64(n_eval))
    path = os.path.join(HERE, 'hotpath_160x120.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), 'bytes;', 'final', final, 'true', q_true)


if __name__ == '__main__':
    main()
