/*
 * rope_oracle.c — CPU restatement of RoPE-S3D's render-and-compare hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rope_s3d_amd/ may import, link or call
 * this file; it is the checker used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED: the reference (/root/reference, 100 % Python) ships no tests,
 * golden vectors or fixtures for this path, and its arithmetic lives in wheels that
 * are absent here (pyrender 0.1.45 / OpenGL, klampt 0.8.7, tensorflow 2.4.1,
 * opencv 4.5.1).  This file restates those libraries' published algorithms at the
 * reference's call sites; it is pinned only by closed-form checks in tests/
 * (URDF chain values, analytic triangles, numpy/scipy identities).
 *
 * What is restated, and where the reference does it:
 *   fk_chain        klampt link transforms      robotpose/simulation/kinematics.py:36-55
 *   mvp             pyrender P·V·M              robotpose/simulation/render.py:52-60,88-90
 *   raster          pyrender SEG offscreen pass robotpose/simulation/render.py:92-98
 *                   (GL rules: pixel-centre sampling, top-left fill, GL_LESS on a
 *                   24-bit window depth interpolated as a float32 plane in screen
 *                   space, back-face culling, znear .05 / zfar 100)
 *   resolve         pyrender depth read-back    z = 2nf/(f+n-(2d-1)(f-n)), d==1 -> 0, f32 ops
 *   sums/finalize   Predictor._error            robotpose/prediction/predict.py:475-509
 *                   Lookup score                robotpose/prediction/predict.py:165-171
 *                   TensorSweep score           robotpose/prediction/predict.py:363-369
 *
 * Arithmetic contract shared with the HIP engine: every floating
 * point step below is a single IEEE-754 operation in the written order (build with
 * -ffp-contract=off); reductions over pixels are exact integer sums of Q32
 * fixed-point metres, so any traversal order gives the same bits.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SUBPIX 256            /* 8 sub-pixel bits */
#define HALFPIX 128
#define KEY_EMPTY 0xFFFFFFFFu
#define D24_MAX 16777215u
#define MAX_LINKS 6

/* ---------------------------------------------------------------- sincos --- */
/* Cody-Waite reduction by pi/2 plus the msun/fdlibm kernel polynomials.  The
 * reference gets sin/cos from Klampt's C++ (libm); this routine agrees with libm
 * to <= 1 ulp (checked in tests) and is written out so that the HIP engine can
 * compute the very same bits on the device. */
static const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
static const double INV_PIO2 = 6.36619772367581382433e-01;
static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

static double k_sin(double x, double y)
{
    double z = x * x, w = z * z;
    double r = (S2 + z * (S3 + z * S4)) + (z * w) * (S5 + z * S6);
    double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

static double k_cos(double x, double y)
{
    double z = x * x, w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    double ww = 1.0 - hz;
    return ww + (((1.0 - ww) - hz) + (z * r - x * y));
}

void orc_sincos(double x, double *s, double *c)
{
    double fn = nearbyint(x * INV_PIO2);
    double r = x - fn * PIO2_1;
    double w = fn * PIO2_1T;
    double y0 = r - w;
    double y1 = (r - y0) - w;
    int n = (int)((long long)fn & 3);
    double sn = k_sin(y0, y1), cs = k_cos(y0, y1);
    switch (n) {
    case 0: *s = sn;  *c = cs;  break;
    case 1: *s = cs;  *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
    }
}

/* ------------------------------------------------------------------- FK ---- */
/* 3x4 affine stored row-major as 12 doubles: [R00 R01 R02 tx | R10 .. ty | R20 .. tz]. */
static void aff_mul(const double *A, const double *B, double *O)
{
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            O[4 * r + c] = (A[4 * r + 0] * B[0 + c] + A[4 * r + 1] * B[4 + c]) + A[4 * r + 2] * B[8 + c];
        O[4 * r + 3] = ((A[4 * r + 0] * B[3] + A[4 * r + 1] * B[7]) + A[4 * r + 2] * B[11]) + A[4 * r + 3];
    }
}

static void axis_rot(const double *a, double q, double *R /* 12 */)
{
    double s, c;
    orc_sincos(q, &s, &c);
    double t = 1.0 - c, ax = a[0], ay = a[1], az = a[2];
    R[0] = (t * ax) * ax + c;       R[1] = (t * ax) * ay - s * az;  R[2] = (t * ax) * az + s * ay;  R[3] = 0.0;
    R[4] = (t * ax) * ay + s * az;  R[5] = (t * ay) * ay + c;       R[6] = (t * ay) * az - s * ax;  R[7] = 0.0;
    R[8] = (t * ax) * az - s * ay;  R[9] = (t * ay) * az + s * ax;  R[10] = (t * az) * az + c;      R[11] = 0.0;
}

/* joint_fixed: 6 x 12 (parent->joint frame: origin xyz + rpy), axes: 6 x 3 (unit),
 * q: 6 joint angles.  out: 7 x 12 link world transforms, out[0] = identity (base_link).
 * T_i = T_{i-1} · F_i · Rot(axis_i, q_i)   (kinematics.py:43-52 via Klampt setConfig). */
void orc_fk(const double *joint_fixed, const double *axes, const double *q, double *out)
{
    static const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memcpy(out, I12, sizeof I12);
    for (int i = 0; i < 6; i++) {
        double R[12], A[12];
        axis_rot(axes + 3 * i, q[i], R);
        aff_mul(joint_fixed + 12 * i, R, A);
        aff_mul(out + 12 * i, A, out + 12 * (i + 1));
    }
}

/* MVP_l = PV (4x4 row-major, double) · M_l (3x4 affine), rounded to float32.  out: n x 16. */
void orc_mvp(const double *PV, const double *fk, int n, float *out)
{
    for (int l = 0; l < n; l++) {
        const double *M = fk + 12 * l;
        for (int r = 0; r < 4; r++) {
            const double *p = PV + 4 * r;
            for (int c = 0; c < 3; c++)
                out[16 * l + 4 * r + c] = (float)((p[0] * M[0 + c] + p[1] * M[4 + c]) + p[2] * M[8 + c]);
            out[16 * l + 4 * r + 3] = (float)(((p[0] * M[3] + p[1] * M[7]) + p[2] * M[11]) + p[3]);
        }
    }
}

/* --------------------------------------------------------------- raster ---- */
/* A shaded vertex.  cls: 0 usable; 1 behind the near plane (z < -w in clip space: its triangle is CLIPPED there, as OpenGL
 * clips every primitive against the view volume before the viewport transform — pyrender draws through GL, render.py:92-98,
 * with znear 0.05 m, projection.py:161-169); 2 unusable (beyond the far plane, w <= 0 in front of the near plane, or window
 * coordinates beyond 1e6 px): its triangles are dropped whole (documented deviation: no far-plane clipping). */
typedef struct { int32_t X, Y; float d; int cls; float cx, cy, cz, cw; } svert;

/* viewport transform + sub-pixel snap of the clip coordinates already in *o; cls 0 or 2 */
static inline void to_window(svert *o, float hw, float hh)
{
    int ok = (o->cw > 0.0f) && (o->cz <= o->cw);
    float rw = 1.0f / o->cw;
    float sx = fmaf(o->cx * rw, hw, hw);
    float sy = fmaf(o->cy * rw, hh, hh);
    o->d = fmaf(o->cz * rw, 0.5f, 0.5f);
    ok = ok && (fabsf(sx) < 1.0e6f) && (fabsf(sy) < 1.0e6f);
    if (ok) {
        o->X = (int32_t)rintf(sx * (float)SUBPIX);
        o->Y = (int32_t)rintf(sy * (float)SUBPIX);
        o->cls = 0;
    } else {
        o->X = o->Y = 0;
        o->cls = 2;
    }
}

static inline svert shade_vertex(const float *m, const float *v, float hw, float hh)
{
    svert o;
    float x = v[0], y = v[1], z = v[2];
    o.cx = fmaf(m[0], x, fmaf(m[1], y, fmaf(m[2], z, m[3])));
    o.cy = fmaf(m[4], x, fmaf(m[5], y, fmaf(m[6], z, m[7])));
    o.cz = fmaf(m[8], x, fmaf(m[9], y, fmaf(m[10], z, m[11])));
    o.cw = fmaf(m[12], x, fmaf(m[13], y, fmaf(m[14], z, m[15])));
    if (o.cz < -o.cw) {                              /* behind the near plane (a point behind the eye is, too) */
        o.X = o.Y = 0;
        o.d = 0.0f;
        o.cls = 1;
        return o;
    }
    to_window(&o, hw, hh);
    return o;
}

/* Where the edge from `in` (inside: z >= -w) to `out` (behind the near plane) meets that plane, in clip space, then to the
 * window.  Always evaluated from the inside vertex towards the outside one, so both triangles that share the edge get the
 * same point bit for bit (no cracks).  One IEEE operation per written step. */
static inline svert clip_near(const svert *in, const svert *out, float hw, float hh)
{
    svert n;
    float bi = in->cz + in->cw, bo = out->cz + out->cw;
    float t = bi / (bi - bo);
    n.cx = fmaf(t, out->cx - in->cx, in->cx);
    n.cy = fmaf(t, out->cy - in->cy, in->cy);
    n.cz = fmaf(t, out->cz - in->cz, in->cz);
    n.cw = fmaf(t, out->cw - in->cw, in->cw);
    to_window(&n, hw, hh);
    return n;
}

static inline int64_t edge_fn(int32_t ax, int32_t ay, int32_t bx, int32_t by, int64_t px, int64_t py)
{
    return (int64_t)(bx - ax) * (py - ay) - (int64_t)(by - ay) * (px - ax);
}

/* edge a->b of a CCW (y-up) triangle owns its boundary samples when it is a left edge
 * (going down) or a top edge (horizontal, going left) */
static inline int owns(int32_t ax, int32_t ay, int32_t bx, int32_t by)
{
    int32_t dy = by - ay, dx = bx - ax;
    return (dy < 0) || (dy == 0 && dx < 0);
}

static inline int floor_div256(int32_t v) { return v >> 8; }               /* arithmetic shift = floor */
static inline int ceil_div256(int32_t v) { return -((-v) >> 8); }

/* one triangle of usable window vertices into the key image (back-face cull, top-left rule, GL_LESS on 24-bit depth) */
static void raster_tri(const svert a, const svert b, const svert c, int l, int W, int H, uint32_t *key)
{
    int64_t area2 = (int64_t)(b.X - a.X) * (c.Y - a.Y) - (int64_t)(c.X - a.X) * (b.Y - a.Y);
    if (area2 <= 0) return;                       /* GL_BACK culled, CCW = front */
    int32_t minX = a.X < b.X ? a.X : b.X; if (c.X < minX) minX = c.X;
    int32_t maxX = a.X > b.X ? a.X : b.X; if (c.X > maxX) maxX = c.X;
    int32_t minY = a.Y < b.Y ? a.Y : b.Y; if (c.Y < minY) minY = c.Y;
    int32_t maxY = a.Y > b.Y ? a.Y : b.Y; if (c.Y > maxY) maxY = c.Y;
    int x0 = ceil_div256(minX - HALFPIX), x1 = floor_div256(maxX - HALFPIX);
    int y0 = ceil_div256(minY - HALFPIX), y1 = floor_div256(maxY - HALFPIX);
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > W - 1) x1 = W - 1;
    if (y1 > H - 1) y1 = H - 1;
    if (x0 > x1 || y0 > y1) return;
    int64_t b01 = owns(a.X, a.Y, b.X, b.Y) ? 0 : -1;
    int64_t b12 = owns(b.X, b.Y, c.X, c.Y) ? 0 : -1;
    int64_t b20 = owns(c.X, c.Y, a.X, a.Y) ? 0 : -1;
    /* window-depth plane in float32, anchored at the pixel that holds vertex a */
    int32_t pxa = a.X >> 8, pya = a.Y >> 8;
    int64_t fxa = (int64_t)pxa * SUBPIX + HALFPIX, fya = (int64_t)pya * SUBPIX + HALFPIX;
    int64_t E20a = edge_fn(c.X, c.Y, a.X, a.Y, fxa, fya), E01a = edge_fn(a.X, a.Y, b.X, b.Y, fxa, fya);
    float inv = 1.0f / (float)(double)area2;
    float e1 = b.d - a.d, e2 = c.d - a.d;
    float fA20 = (float)(-(a.Y - c.Y)), fB20 = (float)(a.X - c.X);
    float fA01 = (float)(-(b.Y - a.Y)), fB01 = (float)(b.X - a.X);
    float gx = (((e1 * fA20) + (e2 * fA01)) * inv) * 256.0f;
    float gy = (((e1 * fB20) + (e2 * fB01)) * inv) * 256.0f;
    float dc = a.d + (((e1 * (float)(double)E20a) + (e2 * (float)(double)E01a)) * inv);
    for (int py = y0; py <= y1; py++) {
        int64_t fy = (int64_t)py * SUBPIX + HALFPIX;
        for (int px = x0; px <= x1; px++) {
            int64_t fx = (int64_t)px * SUBPIX + HALFPIX;
            int64_t E01 = edge_fn(a.X, a.Y, b.X, b.Y, fx, fy);
            int64_t E12 = edge_fn(b.X, b.Y, c.X, c.Y, fx, fy);
            int64_t E20 = edge_fn(c.X, c.Y, a.X, a.Y, fx, fy);
            if ((E01 + b01) < 0 || (E12 + b12) < 0 || (E20 + b20) < 0) continue;
            float d = fmaf(gx, (float)(px - pxa), fmaf(gy, (float)(py - pya), dc));
            float qf = rintf(d * 16777215.0f);
            uint32_t d24 = !(qf >= 0.0f) ? 0u : (qf >= 16777215.0f ? D24_MAX : (uint32_t)qf);
            if (d24 >= D24_MAX) continue;            /* GL_LESS against the cleared 1.0 */
            uint32_t k = (d24 << 8) | (uint32_t)l;
            uint32_t *dst = key + (size_t)(H - 1 - py) * W + px;
            if (k < *dst) *dst = k;
        }
    }
}

/* Rasterise `n_links` links into key image `key` (H x W, row 0 = top of the image).
 * key = (d24 << 8) | link_id, KEY_EMPTY where nothing was drawn. */
void orc_raster(const float *verts, const int32_t *faces, const int32_t *vtx_off, const int32_t *tri_off,
                int n_links, const float *mvp, int W, int H, uint32_t *key)
{
    for (int i = 0; i < W * H; i++) key[i] = KEY_EMPTY;
    float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    for (int l = 0; l < n_links; l++) {
        int nv = vtx_off[l + 1] - vtx_off[l];
        svert *sv = (svert *)malloc(sizeof(svert) * (size_t)(nv > 0 ? nv : 1));
        const float *m = mvp + 16 * l;
        for (int i = 0; i < nv; i++) sv[i] = shade_vertex(m, verts + 3 * (size_t)(vtx_off[l] + i), hw, hh);
        for (int t = tri_off[l]; t < tri_off[l + 1]; t++) {
            svert v[3] = {sv[faces[3 * t]], sv[faces[3 * t + 1]], sv[faces[3 * t + 2]]};
            if (v[0].cls == 2 || v[1].cls == 2 || v[2].cls == 2) continue;
            const int n_near = (v[0].cls == 1) + (v[1].cls == 1) + (v[2].cls == 1);
            if (n_near == 0) { raster_tri(v[0], v[1], v[2], l, W, H, key); continue; }
            if (n_near == 3) continue;
            /* near-plane clipping: rotate the vertex order (the winding stays) so that the odd one out comes first */
            const int odd = n_near == 1 ? (v[0].cls == 1 ? 0 : (v[1].cls == 1 ? 1 : 2)) : (v[0].cls == 0 ? 0 : (v[1].cls == 0 ? 1 : 2));
            const svert p = v[odd], q = v[(odd + 1) % 3], r = v[(odd + 2) % 3];
            if (n_near == 1) {
                /* p is cut off: the quad A q r B with A on p-q and B on r-p, as two triangles */
                const svert A = clip_near(&q, &p, hw, hh), B = clip_near(&r, &p, hw, hh);
                if (A.cls || B.cls) continue;
                raster_tri(A, q, r, l, W, H, key);
                raster_tri(A, r, B, l, W, H, key);
            } else {
                /* only p is in front: the triangle p A B with A on p-q and B on p-r */
                const svert A = clip_near(&p, &q, hw, hh), B = clip_near(&p, &r, hw, hh);
                if (A.cls || B.cls) continue;
                raster_tri(p, A, B, l, W, H, key);
            }
        }
        free(sv);
    }
}

static inline float linear_depth(uint32_t d24, float c_num, float c_sum, float c_dif)
{
    float d = (float)d24 / 16777215.0f;
    float t = 2.0f * d - 1.0f;
    float u = t * c_dif;
    float den = c_sum - u;
    return c_num / den;
}

/* key image -> metric depth (float32) + link id (uint8, 255 = background). */
void orc_resolve(const uint32_t *key, int n, double znear, double zfar, float *depth, uint8_t *id)
{
    float c_num = (float)(2.0 * znear * zfar), c_sum = (float)(zfar + znear), c_dif = (float)(zfar - znear);
    for (int i = 0; i < n; i++) {
        if (key[i] == KEY_EMPTY) { depth[i] = 0.0f; id[i] = 255; }
        else { depth[i] = linear_depth(key[i] >> 8, c_num, c_sum, c_dif); id[i] = (uint8_t)(key[i] & 0xFF); }
    }
}

/* ----------------------------------------------------------------- sums ---- */
/* Layout of one candidate's integer sums (uint64 words).  Shared by every loss kind. */
enum { SUM_CNT = 0, SUM_S1 = 1, SUM_AA = 2, SUM_AB = 3, SUM_BB = 4, SUM_LINK0 = 5, SUM_WORDS = 5 + 3 * MAX_LINKS };
/* per link l: [SUM_LINK0+3l] mismatch count, [+1] non-zero count, [+2] sum of |T_l - z_l| in Q32 */

enum { LOSS_DEPTH = 0, LOSS_FULL = 1, LOSS_LOOKUP = 2, LOSS_TSWEEP = 3 };

static inline uint64_t q32_of_f32(float z) { return (uint64_t)((double)z * 4294967296.0); }

static inline void acc_sq(uint64_t *s, uint64_t dq)
{
    uint64_t a = dq >> 20, b = dq & 0xFFFFFu;
    s[SUM_S1] += dq;
    s[SUM_AA] += a * a;
    s[SUM_AB] += a * b;
    s[SUM_BB] += b * b;
}

/* tq: H x W uint64, bits 0..38 target depth in Q32 metres, bits 40..47 per-link mask bits.
 * t32: H x W float32 plane used by LOSS_LOOKUP (lookup target, NOT sqrt-ed: predict.py:167)
 *      and LOSS_TSWEEP (full target, sqrt-ed here: predict.py:364).
 * crop: r0,r1,c0,c1 inclusive (LOSS_LOOKUP only). */
void orc_sums(const uint32_t *key, int W, int H, double znear, double zfar, int loss, int n_render,
              const uint64_t *tq, const float *t32, const int32_t *crop, uint64_t *sums)
{
    float c_num = (float)(2.0 * znear * zfar), c_sum = (float)(zfar + znear), c_dif = (float)(zfar - znear);
    memset(sums, 0, sizeof(uint64_t) * SUM_WORDS);
    int r0 = 0, r1 = H - 1, c0 = 0, c1 = W - 1;
    if (loss == LOSS_LOOKUP) { r0 = crop[0]; r1 = crop[1]; c0 = crop[2]; c1 = crop[3]; }
    for (int r = r0; r <= r1; r++)
        for (int c = c0; c <= c1; c++) {
            size_t i = (size_t)r * W + c;
            uint32_t k = key[i];
            float z = (k == KEY_EMPTY) ? 0.0f : linear_depth(k >> 8, c_num, c_sum, c_dif);
            int id = (k == KEY_EMPTY) ? 255 : (int)(k & 0xFF);
            if (loss == LOSS_LOOKUP || loss == LOSS_TSWEEP) {
                float a = (loss == LOSS_TSWEEP) ? sqrtf(t32[i]) : t32[i];
                float diff = fabsf(a - sqrtf(z));
                acc_sq(sums, q32_of_f32(diff));
                continue;
            }
            uint64_t T = tq[i] & 0x7FFFFFFFFFull, zq = q32_of_f32(z);
            uint64_t dq = T > zq ? T - zq : zq - T;
            if (dq) { sums[SUM_CNT]++; acc_sq(sums, dq); }
            if (loss == LOSS_FULL) {
                unsigned mask = (unsigned)((tq[i] >> 40) & 0xFF);
                for (int l = 1; l < n_render; l++) {
                    int M = (mask >> l) & 1, R = (id == l);
                    uint64_t a = M ? T : 0, b = R ? zq : 0;
                    uint64_t dl = a > b ? a - b : b - a;
                    sums[SUM_LINK0 + 3 * l] += (uint64_t)(M != R);
                    if (dl) { sums[SUM_LINK0 + 3 * l + 1]++; sums[SUM_LINK0 + 3 * l + 2] += dl; }
                }
            }
        }
}

static double mean_std_parts(const uint64_t *s, double N, double *m1_out)
{
    double m1 = ((double)s[SUM_S1] * 0x1p-32) / N;
    double S2 = ((double)s[SUM_AA] * 0x1p40 + (double)s[SUM_AB] * 0x1p21) + (double)s[SUM_BB];
    double m2 = (S2 * 0x1p-64) / N;
    double var = m2 - m1 * m1;
    if (var < 0.0) var = 0.0;
    *m1_out = m1;
    return sqrt(var);
}

/* sums -> scalar error.  link_flags[l]: bit0 = link l present in the target, bit1 = the
 * ">5 % of the mask has depth" test of predict.py:495 passed (target-only facts). */
double orc_finalize(const uint64_t *s, int loss, int n_render, double n_pix, const uint8_t *link_flags)
{
    double m1, sd = mean_std_parts(s, n_pix, &m1);
    if (loss == LOSS_LOOKUP) return m1 * sd;
    if (loss == LOSS_TSWEEP) return m1 * -sd;
    double err = 0.0;
    if (loss == LOSS_FULL)
        for (int l = 1; l < n_render; l++) {
            if (!(link_flags[l] & 1)) continue;
            err += ((double)s[SUM_LINK0 + 3 * l] / n_pix) * 5.0;
            if ((link_flags[l] & 2) && s[SUM_LINK0 + 3 * l + 1] > 0)
                err += (((double)s[SUM_LINK0 + 3 * l + 2] * 0x1p-32) / (double)s[SUM_LINK0 + 3 * l + 1]) * 10.0;
        }
    double meanD = ((double)s[SUM_S1] * 0x1p-32) / (double)s[SUM_CNT];   /* 0/0 -> NaN as np.mean([]) */
    err += meanD * sd;
    return err;
}

/* ------------------------------------------------------- batch evaluation -- */
typedef struct {
    const float *verts; const int32_t *faces; const int32_t *vtx_off; const int32_t *tri_off;
    const double *joint_fixed; const double *axes; const double *PV;
    int W, H; double znear, zfar;
    int loss, n_render;
    const uint64_t *tq; const float *t32; const int32_t *crop; const uint8_t *link_flags;
    const double *cand; int C; double *err_out; uint64_t *sums_out;
    int tid, nthreads;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)j->W * j->H);
    double n_pix = (double)j->W * j->H;
    if (j->loss == LOSS_LOOKUP)
        n_pix = (double)(j->crop[1] - j->crop[0] + 1) * (double)(j->crop[3] - j->crop[2] + 1);
    for (int c = j->tid; c < j->C; c += j->nthreads) {
        double fk[7 * 12]; float mvp[MAX_LINKS * 16]; uint64_t sums[SUM_WORDS];
        orc_fk(j->joint_fixed, j->axes, j->cand + 6 * c, fk);
        orc_mvp(j->PV, fk, j->n_render, mvp);
        orc_raster(j->verts, j->faces, j->vtx_off, j->tri_off, j->n_render, mvp, j->W, j->H, key);
        orc_sums(key, j->W, j->H, j->znear, j->zfar, j->loss, j->n_render, j->tq, j->t32, j->crop, sums);
        if (j->sums_out) memcpy(j->sums_out + (size_t)SUM_WORDS * c, sums, sizeof sums);
        j->err_out[c] = orc_finalize(sums, j->loss, j->n_render, n_pix, j->link_flags);
    }
    free(key);
    return NULL;
}

/* FK + raster + score for C candidates on `nthreads` host threads (candidates interleaved). */
void orc_eval_batch(const float *verts, const int32_t *faces, const int32_t *vtx_off, const int32_t *tri_off,
                    const double *joint_fixed, const double *axes, const double *PV, int W, int H,
                    double znear, double zfar, int loss, int n_render, const uint64_t *tq, const float *t32,
                    const int32_t *crop, const uint8_t *link_flags, const double *cand, int C,
                    double *err_out, uint64_t *sums_out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t jobs[256]; pthread_t th[256];
    for (int t = 0; t < nthreads; t++) {
        job_t j = {verts, faces, vtx_off, tri_off, joint_fixed, axes, PV, W, H, znear, zfar, loss, n_render,
                   tq, t32, crop, link_flags, cand, C, err_out, sums_out, t, nthreads};
        jobs[t] = j;
    }
    if (nthreads == 1) { worker(&jobs[0]); return; }
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

/* OR over candidates of "pixel covered" (Crop._create's depth-sum loop, robotpose/crop.py:60-81). */
typedef struct { const job_t *j; uint8_t *cover; } cov_job;

static void *cov_worker(void *arg)
{
    cov_job *cj = (cov_job *)arg;
    const job_t *j = cj->j;
    size_t n = (size_t)j->W * j->H;
    uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * n);
    for (int c = j->tid; c < j->C; c += j->nthreads) {
        double fk[7 * 12]; float mvp[MAX_LINKS * 16];
        orc_fk(j->joint_fixed, j->axes, j->cand + 6 * c, fk);
        orc_mvp(j->PV, fk, j->n_render, mvp);
        orc_raster(j->verts, j->faces, j->vtx_off, j->tri_off, j->n_render, mvp, j->W, j->H, key);
        for (size_t i = 0; i < n; i++)
            if (key[i] != KEY_EMPTY) cj->cover[i] = 1;      /* racing writers all store 1 */
    }
    free(key);
    return NULL;
}

void orc_coverage_batch(const float *verts, const int32_t *faces, const int32_t *vtx_off, const int32_t *tri_off,
                        const double *joint_fixed, const double *axes, const double *PV, int W, int H,
                        int n_render, const double *cand, int C, uint8_t *cover, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t jobs[256]; cov_job cj[256]; pthread_t th[256];
    memset(cover, 0, (size_t)W * H);
    for (int t = 0; t < nthreads; t++) {
        job_t j = {verts, faces, vtx_off, tri_off, joint_fixed, axes, PV, W, H, 0.0, 0.0, 0, n_render,
                   NULL, NULL, NULL, NULL, cand, C, NULL, NULL, t, nthreads};
        jobs[t] = j;
        cj[t].j = &jobs[t];
        cj[t].cover = cover;
    }
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, cov_worker, &cj[t]);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

int orc_sum_words(void) { return SUM_WORDS; }
