#!/usr/bin/env python3
"""CPU-side estimate (no GPU): how often a meshlet of links 3-5 is shaded and culled per candidate on the bench
grid with (a) the fixed 128x96 screen tiles and (b) windows of the same size anchored at the links' own bounding box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR          # noqa: E402
from rope_s3d_amd.projection import Intrinsics, camera_matrix                  # noqa: E402
from rope_s3d_amd.robot import RobotModel                                    # noqa: E402
from rope_s3d_amd.simulation.kinematics import ForwardKinematics            # noqa: E402

TW, TH = 128, 96


def main():
    rb = RobotModel.from_urdf()
    ml = rb.meshlets
    intr = Intrinsics('640_480_color')
    W, H = intr.width, intr.height
    PV = camera_matrix(DEFAULT_CAMERA_POSE, intr, ZNEAR, ZFAR)
    fk = ForwardKinematics(rb)
    M = len(ml.header)
    link = ml.header[:, 7].astype(int)
    v0 = ml.header[:, 4].astype(int)
    nv = (ml.header[:, 6] & 0xFFFF).astype(int)
    nt = (ml.header[:, 6] >> 16).astype(int)
    vm = np.repeat(np.arange(M), nv)
    lim = rb.joint_limits
    rng = np.random.default_rng(1)
    fixed_inc = anch_inc = fixed_tiles = anch_tiles = n_ml = 0
    sizes = []
    lo_l, hi_l = (int(x) for x in os.environ.get('LINKS', '3,6').split(','))
    for _ in range(256):
        q = rng.uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
        T = fk.calc(q)
        boxes = []
        for l in range(lo_l, hi_l):
            sel = np.nonzero(link == l)[0]
            for m in sel:
                P = ml.verts[v0[m]:v0[m] + nv[m]].astype(np.float64)
                c = (PV @ T[l] @ np.c_[P, np.ones(len(P))].T).T
                sx = (c[:, 0] / c[:, 3] * 0.5 + 0.5) * W
                sy = (1 - (c[:, 1] / c[:, 3] * 0.5 + 0.5)) * H          # image rows
                x0, x1 = max(int(np.floor(sx.min() - 1)), 0), min(int(np.ceil(sx.max() + 1)), W - 1)
                y0, y1 = max(int(np.floor(sy.min() - 1)), 0), min(int(np.ceil(sy.max() + 1)), H - 1)
                if x0 <= x1 and y0 <= y1:
                    boxes.append((x0, x1, y0, y1, nt[m]))
        b = np.array(boxes)
        n_ml += len(b)
        # (a) fixed grid
        tiles = set()
        for x0, x1, y0, y1, _ in b:
            nx, ny = x1 // TW - x0 // TW + 1, y1 // TH - y0 // TH + 1
            fixed_inc += nx * ny
            for ty in range(y0 // TH, y1 // TH + 1):
                for tx in range(x0 // TW, x1 // TW + 1):
                    tiles.add((tx, ty))
        fixed_tiles += len(tiles)
        # (b) windows anchored at the bounding box (x anchor rounded down to a multiple of 4)
        ax, ay = (b[:, 0].min() // 4) * 4, b[:, 2].min()
        bw, bh = b[:, 1].max() - ax + 1, b[:, 3].max() - ay + 1
        sizes.append((bw, bh))
        wins = set()
        for x0, x1, y0, y1, _ in b:
            nx, ny = (x1 - ax) // TW - (x0 - ax) // TW + 1, (y1 - ay) // TH - (y0 - ay) // TH + 1
            anch_inc += nx * ny
            for ty in range((y0 - ay) // TH, (y1 - ay) // TH + 1):
                for tx in range((x0 - ax) // TW, (x1 - ax) // TW + 1):
                    wins.add((tx, ty))
        anch_tiles += len(wins)
    s = np.array(sizes)
    print(f"links {lo_l}..{hi_l - 1}: meshlets on screen per candidate {n_ml / 256:.0f}")
    print(f"fixed tiles : {fixed_tiles / 256:.2f} tiles per candidate, meshlet-tile incidences x{fixed_inc / n_ml:.3f}")
    print(f"anchored    : {anch_tiles / 256:.2f} windows per candidate, meshlet-window incidences x{anch_inc / n_ml:.3f}")
    print(f"bbox width  median {np.median(s[:, 0]):.0f} p90 {np.percentile(s[:, 0], 90):.0f} max {s[:, 0].max()};  "
          f"height median {np.median(s[:, 1]):.0f} p90 {np.percentile(s[:, 1], 90):.0f} max {s[:, 1].max()}")
    print(f"fits one 128x96 window: {np.mean((s[:, 0] <= TW) & (s[:, 1] <= TH)):.2f}; one 96x128: {np.mean((s[:, 0] <= TH) & (s[:, 1] <= TW)):.2f}; "
          f"either: {np.mean(((s[:, 0] <= TW) & (s[:, 1] <= TH)) | ((s[:, 0] <= TH) & (s[:, 1] <= TW))):.2f}")


if __name__ == '__main__':
    main()
