#!/usr/bin/env python3
"""CPU-side estimate (no GPU) of what meshlet-level culls would remove on the bench workload: for a sample of the
16^3 S/L/U grid, per link the share of meshlets whose triangles are ALL back-facing (exact per-triangle test and the
conservative normal-cone test the bounds kernel can afford), and the number of fixed-grid tiles the links meet."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR          # noqa: E402
from rope_s3d_amd.projection import Intrinsics, view_matrix                  # noqa: E402
from rope_s3d_amd.robot import RobotModel                                    # noqa: E402
from rope_s3d_amd.simulation.kinematics import ForwardKinematics            # noqa: E402


def main():
    rb = RobotModel.from_urdf()
    ml = rb.meshlets
    intr = Intrinsics('640_480_color')
    V = view_matrix(DEFAULT_CAMERA_POSE)
    eye_w = np.linalg.inv(V)[:3, 3]
    fk = ForwardKinematics(rb)
    M = len(ml.header)
    link = ml.header[:, 7].astype(int)
    v0, t0 = ml.header[:, 4].astype(int), ml.header[:, 5].astype(int)
    nv, nt = (ml.header[:, 6] & 0xFFFF).astype(int), (ml.header[:, 6] >> 16).astype(int)
    # per meshlet: triangle normals (unnormalised) and one vertex of each
    tri_n, tri_a, tri_m = [], [], []
    cones = np.zeros((M, 8))            # apex-free cone: axis xyz, min dot(axis, n_hat); centre xyz, radius
    for m in range(M):
        P = ml.verts[v0[m]:v0[m] + nv[m]].astype(np.float64)
        pk = ml.tris[t0[m]:t0[m] + nt[m]]
        a, b, c = P[pk & 0xFF], P[(pk >> 8) & 0xFF], P[(pk >> 16) & 0xFF]
        n = np.cross(b - a, c - a)
        ln = np.linalg.norm(n, axis=1)
        ok = ln > 1e-14
        nh = n[ok] / ln[ok, None]
        ax = nh.mean(0)
        ax /= max(np.linalg.norm(ax), 1e-30)
        cones[m, :3] = ax
        cones[m, 3] = (nh @ ax).min() if len(nh) else -1.0
        if not ok.all():
            cones[m, 3] = -1.0           # a degenerate triangle: never cull the meshlet
        ctr = (P.min(0) + P.max(0)) / 2
        cones[m, 4:7] = ctr
        cones[m, 7] = np.sqrt(((P - ctr) ** 2).sum(1).max())
        tri_n.append(n), tri_a.append(a), tri_m.append(np.full(len(n), m))
    tri_n, tri_a, tri_m = np.concatenate(tri_n), np.concatenate(tri_a), np.concatenate(tri_m)
    lim = rb.joint_limits
    rng = np.random.default_rng(1)
    tot = np.zeros((6, 4))
    for _ in range(64):
        q = rng.uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
        T = fk.calc(q)
        for l in range(6):
            sel = link == l
            e_l = np.linalg.inv(T[l])[:3, :] @ np.append(eye_w, 1.0)      # eye in the link frame
            tsel = sel[tri_m]
            back = ((tri_a[tsel] - e_l) * tri_n[tsel]).sum(1) >= 0           # n . (a - eye) >= 0: faces away
            mm = tri_m[tsel]
            all_back = np.ones(M, bool)
            np.logical_and.at(all_back, mm, back)
            exact = all_back[sel]
            # cone: every normal within angle acos(c) of axis; the view direction to any point of the sphere deviates
            # from d = centre - eye by at most asin(r/|d|).  all back-facing if angle(axis, d) + acos(c) + asin(r/|d|) <= 90 deg - margin
            d = cones[sel, 4:7] - e_l
            dist = np.linalg.norm(d, axis=1)
            cosad = (d * cones[sel, :3]).sum(1) / dist
            ang = np.arccos(np.clip(cosad, -1, 1)) + np.arccos(np.clip(cones[sel, 3], -1, 1)) + np.arcsin(np.clip(cones[sel, 7] / dist, 0, 1))
            cone = ang <= np.pi / 2 - 0.12
            assert not (cone & ~exact).any()
            w = nt[sel]
            tot[l] += [w.sum(), w[exact].sum(), w[cone].sum(), back.sum()]
    for l in range(6):
        print(f"link {l}: triangles back-facing {tot[l, 3] / tot[l, 0]:.2f}; in wholly back-facing meshlets {tot[l, 1] / tot[l, 0]:.2f}; "
              f"cone test (0.12 rad margin) finds {tot[l, 2] / tot[l, 0]:.2f}")
    s = tot[3:].sum(0)
    print(f"links 3-5: back {s[3] / s[0]:.2f}, wholly-back meshlets {s[1] / s[0]:.2f}, cone {s[2] / s[0]:.2f}")
    s = tot.sum(0)
    print(f"all links: back {s[3] / s[0]:.2f}, wholly-back meshlets {s[1] / s[0]:.2f}, cone {s[2] / s[0]:.2f}")


if __name__ == '__main__':
    main()
