#!/usr/bin/env python3
"""An INDEPENDENT, exact rasteriser for the real mh5l mesh, in Python integers — a second implementation of DESIGN.md §3's
rendering rules that shares no code and no arithmetic style with oracle/rope_oracle.c or the HIP kernels:

  * vertex shading: the float32 operations of the specification (four fused multiply-add chains, one reciprocal, the viewport
    fused multiply-adds, the snap to 1/256 pixel) evaluated EXACTLY as rationals over big integers and rounded once per operation
    to float32 by hand (round to nearest, ties to even) — no hardware float32, no libm, no double rounding;
  * coverage: per triangle, brute force over every pixel of its bounding box with the three edge functions in Python's unbounded
    integers, OpenGL's top-left rule (y up), back faces culled, pixel centres at 256 p + 128;
  * visibility: the window depth at a sample centre by exact barycentric interpolation (a Fraction), nearest wins, ties to the lower
    link id.  The oracle evaluates a float32 plane and quantises to 24 bits; the two may disagree on WHICH link wins only where two
    links' surfaces are within a few 2^-24 of each other, and those pixels are listed instead of compared.

Inputs taken from the oracle: the six float32 link matrices P·V·T of a pose (forward kinematics and the matrix product are pinned
separately by closed forms, tests/test_oracle_pins.py).  Output: tests/golden/pins_raster_160x120.npz — per pose the coverage mask,
the link-id image, the exact depth quantised to 24 bits, and the mask of near-tie pixels.  tests/test_oracle_pins.py holds the
oracle (CPU suite) and tests/test_golden.py the engine (GPU suite) to these.

    python tests/golden/make_pins.py        # ~2-3 minutes: a million exact float32 operations per pose in pure Python
"""
import os
import sys
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, os.pardir, os.pardir))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

S = 149                      # every finite float32 is an integer multiple of 2^-149
POSES = [[0.4, 0.3, 0.8, 0, 0, 0], [-0.6, 1.2, -0.3, 0.5, -0.7, 1.0], [1.3, -0.8, 2.1, -1.0, 0.9, 0.2]]
D24 = (1 << 24) - 1


def f32_to_int(x) -> int:
    """float32 -> the integer I with x = I * 2^-149 (exact)."""
    b = int(np.float32(x).view(np.uint32))
    sign, e, m = b >> 31, (b >> 23) & 0xFF, b & 0x7FFFFF
    assert e != 0xFF, "non-finite"
    v = (m | 0x800000) << (e - 1) if e else m
    return -v if sign else v


def round_to_f32(num: int, den: int = 1) -> int:
    """The float32 nearest to num / (den * 2^149) (ties to even), returned as its own integer (multiple of 2^-149 units).
    num, den: Python ints, den > 0."""
    if num == 0:
        return 0
    sign = -1 if num < 0 else 1
    n = abs(num)
    # value v = n / den (in units of 2^-149).  Want m = round(v / 2^k) with k >= 0 chosen so that m has at most 24 bits.
    q = n // den
    L = q.bit_length()                                   # v in [2^(L-1), 2^L) when q > 0
    k = max(L - 24, 0)                                   # subnormal / small values: k = 0, ulp = one unit
    scale = den << k
    m, r = divmod(n, scale)
    twice = 2 * r
    if twice > scale or (twice == scale and (m & 1)):
        m += 1
    if m.bit_length() > 24:                              # the rounding carried into the next binade
        m >>= 1
        k += 1
    assert (m << k).bit_length() <= 128 + S + 24, "overflow"
    return sign * (m << k)


def fma(a: int, b: int, c: int) -> int:                  # fmaf(a, b, c): a*b is in units of 2^-298
    return round_to_f32(a * b + (c << S), 1 << S)


def mul(a: int, b: int) -> int:
    return round_to_f32(a * b, 1 << S)


def rcp(a: int) -> int:                                  # 1.0f / a
    sign = -1 if a < 0 else 1
    return sign * round_to_f32(1 << (2 * S), abs(a))


def to_float(i: int) -> float:
    return float(Fraction(i, 1 << S))


def shade(m, x, y, z, hw, hh):
    """DESIGN.md §3 step 2 -> (X, Y, d as a Fraction) or None when the vertex has no window position."""
    cx = fma(m[0], x, fma(m[1], y, fma(m[2], z, m[3])))
    cy = fma(m[4], x, fma(m[5], y, fma(m[6], z, m[7])))
    cz = fma(m[8], x, fma(m[9], y, fma(m[10], z, m[11])))
    cw = fma(m[12], x, fma(m[13], y, fma(m[14], z, m[15])))
    if not (cw > 0 and -cw <= cz <= cw):
        return None
    rw = rcp(cw)
    sx = fma(mul(cx, rw), hw, hw)
    sy = fma(mul(cy, rw), hh, hh)
    d = fma(mul(cz, rw), HALF, HALF)
    lim = 10 ** 6 << S
    if not (abs(sx) < lim and abs(sy) < lim):
        return None

    def snap(s):                                         # rint(256 * s): 256 * s is exact in float32 here (|s| < 1e6), ties to even
        n, den = s * 256, 1 << S
        q, r = divmod(n, den)
        if 2 * r > den or (2 * r == den and (q & 1)):
            q += 1
        return q
    return snap(sx), snap(sy), Fraction(d, 1 << S)


def owns(ax, ay, bx, by):
    dy, dx = by - ay, bx - ax
    return dy < 0 or (dy == 0 and dx < 0)


def edge(ax, ay, bx, by, fx, fy):
    return (bx - ax) * (fy - ay) - (by - ay) * (fx - ax)


def render(o, rb, q, W, H):
    mvp = o.mvp(q, 6)
    hw, hh = f32_to_int(np.float32(0.5) * np.float32(W)), f32_to_int(np.float32(0.5) * np.float32(H))
    best = {}                                            # pixel -> (depth Fraction, link)
    second = {}                                          # pixel -> nearest depth of ANOTHER link than the winner's (for the near-tie list)
    for l in range(6):
        m = [f32_to_int(v) for v in mvp[l]]
        V = rb.verts[rb.vtx_off[l]:rb.vtx_off[l + 1]]
        sv = [shade(m, f32_to_int(v[0]), f32_to_int(v[1]), f32_to_int(v[2]), hw, hh) for v in V]
        for tri in rb.faces[rb.tri_off[l]:rb.tri_off[l + 1]]:
            a, b, c = sv[tri[0]], sv[tri[1]], sv[tri[2]]
            if a is None or b is None or c is None:
                continue
            (ax, ay, da), (bx, by, db), (cx_, cy_, dc) = a, b, c
            area2 = (bx - ax) * (cy_ - ay) - (cx_ - ax) * (by - ay)
            if area2 <= 0:                               # GL_BACK culled: counter-clockwise (y up) is the front
                continue
            x0 = max(-((-(min(ax, bx, cx_) - 128)) >> 8), 0)
            x1 = min((max(ax, bx, cx_) - 128) >> 8, W - 1)
            y0 = max(-((-(min(ay, by, cy_) - 128)) >> 8), 0)
            y1 = min((max(ay, by, cy_) - 128) >> 8, H - 1)
            b01 = 0 if owns(ax, ay, bx, by) else -1
            b12 = 0 if owns(bx, by, cx_, cy_) else -1
            b20 = 0 if owns(cx_, cy_, ax, ay) else -1
            for py in range(y0, y1 + 1):
                fy = 256 * py + 128
                for px in range(x0, x1 + 1):
                    fx = 256 * px + 128
                    e01, e12, e20 = edge(ax, ay, bx, by, fx, fy), edge(bx, by, cx_, cy_, fx, fy), edge(cx_, cy_, ax, ay, fx, fy)
                    if e01 + b01 < 0 or e12 + b12 < 0 or e20 + b20 < 0:
                        continue
                    d = (e12 * da + e20 * db + e01 * dc) / area2          # exact barycentric depth at the sample centre
                    if not (d < 1):                      # GL_LESS against the cleared depth of 1.0 (quantisation aside)
                        continue
                    pix = ((H - 1 - py), px)             # image rows top-down
                    cur = best.get(pix)
                    if cur is None or (d, l) < cur:
                        if cur is not None and cur[1] != l:
                            second[pix] = min(second.get(pix, cur[0]), cur[0])
                        best[pix] = (d, l)
                    elif cur[1] != l:
                        second[pix] = min(second.get(pix, d), d)
    ids = np.full((H, W), 255, np.uint8)
    d24 = np.zeros((H, W), np.uint32)
    near = np.zeros((H, W), bool)
    for (r, c), (d, l) in best.items():
        ids[r, c] = l
        d24[r, c] = int(round(d * D24))                  # round half even on an exact Fraction
        if (r, c) in second and second[(r, c)] - d < Fraction(8, 1 << 24):
            near[r, c] = True
    return ids, d24, near


if __name__ == '__main__':
    import helpers
    HALF = f32_to_int(np.float32(0.5))
    rb = helpers.robot()
    intr, PV = helpers.camera('640_480_color', ds=4)
    o = helpers.make_oracle(rb, intr, PV)
    out = {'poses': np.array(POSES)}
    for k, q in enumerate(POSES):
        ids, d24, near = render(o, rb, q, intr.width, intr.height)
        out[f'ids{k}'], out[f'd24_{k}'], out[f'near{k}'] = ids, d24, near
        key = o.raster_key(q, 6)
        o_ids = np.where(key == 0xFFFFFFFF, 255, key & 0xFF).astype(np.uint8)
        print(f"pose {k}: {int((ids != 255).sum())} covered pixels; coverage equal to the oracle's: {np.array_equal(ids != 255, o_ids != 255)}; "
              f"ids differ on {int((ids != o_ids).sum())} pixels ({int(((ids != o_ids) & ~near).sum())} outside the near-tie list of {int(near.sum())}); "
              f"max |d24 - oracle| = {int(np.abs(d24.astype(np.int64) - (key >> 8).astype(np.int64))[ids != 255].max())}")
    np.savez_compressed(os.path.join(HERE, 'pins_raster_160x120.npz'), **out)
    print('written')
else:
    HALF = f32_to_int(np.float32(0.5))
