// rope_kernels.hip — CDNA4 (gfx950) kernels of the render-and-compare pose engine.
//
// One pass over a batch of candidate joint vectors:
//   fk_mvp_kernel           per candidate: joint angles -> link world transforms -> P·V·M; clears the
//                           candidate's accumulators (stands in for klampt FK,
//                           robotpose/simulation/kinematics.py:36-55, and pyrender's node poses, render.py:88-90)
//   bounds_kernel           per (meshlet, candidate): screen box + the candidate's masks of touched tiles
//   fk_bounds_kernel        batches of <= 256 rows: the two above in one launch, one workgroup per candidate
//   raster_score_kernel     per (128x96 screen tile, row): meshlet list, vertex shading, triangle cull and
//     <LOSS, MODE>          set-up, z-test into an LDS depth/id tile, then the loss terms of the covered samples
//                           as exact integer sums (stands in for pyrender's SEG pass, render.py:92-98, and
//                           Predictor._error / the lookup reduction, predict.py:475-509,165-171).  Modes:
//                             LAYER  links 0-2 once per distinct (q0,q1) -> key tiles + their loss sums in HBM
//                             SCORE  a candidate's remaining links, merged with its layer by min; loss delta
//                             SPLIT  few candidates: a tile's meshlets over several workgroups, merged by atomicMin
//                             TABLE  rows of the stored lookup table (cropped sqrt-depth)
//                             DUMP / COVER  single-pose render, crop search
//   score_gtile_kernel      after SPLIT: scores the merged images in row bands and hands the buffer back empty
//   finalize_argmin_kernel  integer sums -> float64 errors, wave-shuffle argmin
//   table_score_kernel      Lookup stage against the stored table: pure streaming
//   camera-pose path        the same kernels with a view matrix per candidate (fk) and target planes per frame
//                           (RasterArgs::frame_of); ROPE_LOSS_CAMFULL = CameraPredictor._error's sums
//
// Arithmetic contract: all floating point steps are single IEEE-754
// operations in the written order (built with -ffp-contract=off; fmaf where fused),
// pixel reductions are exact integer sums, so results do not depend on the tiling or
// on the order in which workgroups, waves or atomics complete.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "rope_kernels.h"

namespace rope {

// ------------------------------------------------------------------ sincos -----
// Cody-Waite reduction by pi/2 + msun kernel polynomials, <= 1 ulp of libm.
__device__ static inline double k_sin(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x, w = z * z;
    double r = (S2 + z * (S3 + z * S4)) + (z * w) * (S5 + z * S6);
    double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

__device__ static inline double k_cos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x, w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    double ww = 1.0 - hz;
    return ww + (((1.0 - ww) - hz) + (z * r - x * y));
}

__device__ static inline void det_sincos(double x, double &s, double &c)
{
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
    const double INV_PIO2 = 6.36619772367581382433e-01;
    double fn = rint(x * INV_PIO2);
    double r = x - fn * PIO2_1;
    double w = fn * PIO2_1T;
    double y0 = r - w;
    double y1 = (r - y0) - w;
    int n = (int)((long long)fn & 3);
    double sn = k_sin(y0, y1), cs = k_cos(y0, y1);
    s = (n == 0) ? sn : (n == 1) ? cs : (n == 2) ? -sn : -cs;
    c = (n == 0) ? cs : (n == 1) ? -sn : (n == 2) ? -cs : sn;
}

// ------------------------------------------------------------------- FK --------
__device__ static inline void aff_mul(const double *A, const double *B, double *O)
{
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++)
            O[4 * r + c] = (A[4 * r + 0] * B[0 + c] + A[4 * r + 1] * B[4 + c]) + A[4 * r + 2] * B[8 + c];
        O[4 * r + 3] = ((A[4 * r + 0] * B[3] + A[4 * r + 1] * B[7]) + A[4 * r + 2] * B[11]) + A[4 * r + 3];
    }
}

// A_j = F_j · Rot(axis_j, q_j): parent-link frame -> link j+1 frame (12 doubles, 3x4 row-major)
__device__ static inline void joint_matrix(double q, int j, const double *__restrict__ joint_fixed, const double *__restrict__ joint_axes,
                                           double *A)
{
    double s, co;
    det_sincos(q, s, co);
    const double t = 1.0 - co, ax = joint_axes[3 * j], ay = joint_axes[3 * j + 1], az = joint_axes[3 * j + 2];
    double R[12], F[12];
    R[0] = (t * ax) * ax + co;      R[1] = (t * ax) * ay - s * az;  R[2] = (t * ax) * az + s * ay;  R[3] = 0.0;
    R[4] = (t * ax) * ay + s * az;  R[5] = (t * ay) * ay + co;      R[6] = (t * ay) * az - s * ax;  R[7] = 0.0;
    R[8] = (t * ax) * az - s * ay;  R[9] = (t * ay) * az + s * ax;  R[10] = (t * az) * az + co;     R[11] = 0.0;
#pragma unroll
    for (int k = 0; k < 12; k++) F[k] = joint_fixed[12 * j + k];
    aff_mul(F, R, A);
}

// element (r, k) of P·V·T rounded to float32 (T: 3x4 affine, last row 0 0 0 1)
__device__ static inline float mvp_element(const double *__restrict__ PV, const double *T, int r, int k)
{
    const double p0 = PV[4 * r], p1 = PV[4 * r + 1], p2 = PV[4 * r + 2], p3 = PV[4 * r + 3];
    return k < 3 ? (float)((p0 * T[0 + k] + p1 * T[4 + k]) + p2 * T[8 + k]) : (float)(((p0 * T[3] + p1 * T[7]) + p2 * T[11]) + p3);
}

// One thread per candidate.  Writes n_render 4x4 float matrices.
__global__ void __launch_bounds__(256)
fk_mvp_kernel(const double *__restrict__ cand, int C, int n_render, const double *__restrict__ joint_fixed,
              const double *__restrict__ joint_axes, const double *__restrict__ PV_all, const int32_t *__restrict__ view_of,
              float *__restrict__ mvp, uint64_t *__restrict__ sums, uint32_t *__restrict__ mask_lo, uint32_t *__restrict__ mask_hi,
              int mask_words, int *__restrict__ queue_counters, uint32_t *__restrict__ tile_tris, uint32_t *__restrict__ tile_tris_lo, int n_tiles)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c == 0 && queue_counters)                                                 // raster_queue_kernel's, filled later in the pass
        for (int k = 0; k < 2 * QUEUE_COUNTERS; k++) queue_counters[k] = 0;      // the scoring queue's, then the layer queue's
    if (tile_tris)
        for (int k = 0; k < n_tiles; k++) tile_tris[(size_t)c * n_tiles + k] = tile_tris_lo[(size_t)c * n_tiles + k] = 0;
    // camera-pose path: every candidate names its own view matrix (camera_pose_prediction.py:116-124)
    const double *__restrict__ PV = PV_all + (view_of ? 16 * (size_t)view_of[c] : 0);
    // first kernel of a pass: clear what the later kernels accumulate into (saves three memset launches)
    for (int k = 0; k < ROPE_SUM_WORDS; k++) sums[(size_t)c * ROPE_SUM_WORDS + k] = 0;
    for (int k = 0; k < mask_words; k++) { mask_lo[(size_t)c * mask_words + k] = 0; mask_hi[(size_t)c * mask_words + k] = 0; }
    double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int l = 0; l < n_render; l++) {
        if (l > 0) {
            double A[12], N[12];
            joint_matrix(cand[6 * c + l - 1], l - 1, joint_fixed, joint_axes, A);
            aff_mul(T, A, N);
#pragma unroll
            for (int k = 0; k < 12; k++) T[k] = N[k];
        }
        float *o = mvp + ((size_t)c * ROPE_MAX_LINKS + l) * 16;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) o[4 * r + k] = mvp_element(PV, T, r, k);
    }
}

// --------------------------------------------------------------- pixel maths ----
__device__ static inline float linear_depth(uint32_t d24, float c_num, float c_sum, float c_dif)
{
    float d = (float)d24 / 16777215.0f;
    float t = 2.0f * d - 1.0f;
    float u = t * c_dif;
    float den = c_sum - u;
    return c_num / den;
}

// floor(z * 2^32) for a finite z >= 0 (same value as (uint64_t)((double)z * 2^32)).
// Any size: by shifting the significand.
__device__ static inline uint64_t q32_of_f32_wide(float z)
{
    const uint32_t b = __float_as_uint(z);
    const int ex = (int)(b >> 23) & 0xFF;
    // zero and denormals are below one unit of 2^-32; no branches: both shifts are always executed, one of them by 0
    const uint64_t m = ex ? (uint64_t)((b & 0x7FFFFFu) | 0x800000u) : 0ull;
    const int sh = ex - 127 - 23 + 32;                        // z = m * 2^(ex-150)
    return (m << min(max(sh, 0), 63)) >> min(max(-sh, 0), 63);
}

// Below 2^32 (every depth and depth difference there is: the far plane is at 100 m) in five full-rate operations instead of a
// dozen with 64-bit shifts: z = trunc(z) + fraction, both parts exact in float32; the integer part is the high word, and
// fraction * 2^32 — a power-of-two scaling, exact, below 2^32 — truncated is the low word.
__device__ static inline uint64_t q32_of_f32(float z)
{
    if (__builtin_expect(!(z < 4294967296.0f), 0)) return q32_of_f32_wide(z);
    const float whole = truncf(z);
    const uint32_t hi = (uint32_t)whole, lo = (uint32_t)((z - whole) * 4294967296.0f);
    return ((uint64_t)hi << 32) | lo;
}

// floor(sqrt(x) * 2^32) for x given in Q32: sqrt(dq * 2^-32) * 2^32 = sqrt(dq) * 2^16; one correctly rounded
// float64 square root (dq < 2^39 converts exactly), then truncation
__device__ static inline uint64_t sqrt_q32(uint64_t dq) { return (uint64_t)(sqrt((double)dq) * 65536.0); }

// NEG subtracts instead of adds (modulo 2^64): lets "sums(tile) - sums(base)" live in one set of registers
template <bool NEG>
__device__ static inline void acc(uint64_t &w, uint64_t x) { if (NEG) w -= x; else w += x; }

template <bool NEG>
__device__ static inline void acc_sq(uint64_t *s, uint64_t dq)
{
    uint32_t a = (uint32_t)(dq >> 20), b = (uint32_t)(dq & 0xFFFFFu);
    acc<NEG>(s[SUM_S1], dq);
    acc<NEG>(s[SUM_AA], (uint64_t)a * a);
    acc<NEG>(s[SUM_AB], (uint64_t)a * b);
    acc<NEG>(s[SUM_BB], (uint64_t)b * b);
}

// Loss terms of one pixel given its z-buffer key.  `pix` indexes the H x W target planes.
template <int LOSS, bool NEG = false>
__device__ static inline void score_pixel(uint32_t key, size_t pix, int n_render, const uint64_t t /* tq[pix] */,
                                          const float ta /* t32[pix] */, const uint64_t *__restrict__ tl, size_t plane,
                                          float c_num, float c_sum, float c_dif, uint64_t *s, uint64_t *pk = nullptr,
                                          unsigned link_mask = 0xFFu /* ROPE_LOSS_FULL: the links whose terms are wanted (uniform) */)
{
    const bool empty = (key == KEY_EMPTY);
    float z = empty ? 0.0f : linear_depth(key >> 8, c_num, c_sum, c_dif);
    if (LOSS == ROPE_LOSS_LOOKUP || LOSS == ROPE_LOSS_TSWEEP) {
        float a = ta;
        if (LOSS == ROPE_LOSS_TSWEEP) a = sqrtf(a);
        float diff = fabsf(a - sqrtf(z));
        uint64_t dq = q32_of_f32(diff);
        if (dq) acc_sq<NEG>(s, dq);
        return;
    }
    if (LOSS == ROPE_LOSS_CAMFULL) {
        // CameraPredictor._error (camera_pose_prediction.py:933-970): every difference enters as its square root;
        // per link (base_link included) mask mismatches and the mean of sqrt|T_l - D*R_l| over its non-zero entries
        const uint64_t T = t & 0x7FFFFFFFFFull, zq = q32_of_f32(z);
        const uint64_t dq = T > zq ? T - zq : zq - T;
        if (dq) { acc<NEG>(s[SUM_CNT], 1); acc<NEG>(s[SUM_S1], dq); acc<NEG>(s[SUM_AA], sqrt_q32(dq)); }
        const int id = empty ? 255 : (int)(key & 0xFF);
#pragma unroll
        for (int l = 0; l < ROPE_MAX_LINKS; l++) {
            if (l < n_render) {
                const uint64_t tll = tl[(size_t)l * plane + pix];
                // the reference compares blue values (:943-946), and base_link's blue is 0 — the black background's too
                // (constants.py:82-89): its render mask also holds every pixel nothing was drawn on
                const bool M = (tll >> 40) & 1, R = (id == l) || (l == 0 && empty);
                if (!M && !R && !(tll & 0x7FFFFFFFFFull)) continue;
                const uint64_t a = tll & 0x7FFFFFFFFFull, b = R ? zq : 0;
                const uint64_t dl = a > b ? a - b : b - a;
                pk[NEG ? 1 : 0] += (uint64_t)(M != R) << (10 * l);          // six 10-bit fields (see ROPE_LOSS_FULL below)
                pk[NEG ? 3 : 2] += (uint64_t)(dl != 0) << (10 * l);
                if (dl) acc<NEG>(s[SUM_LINK0 + 3 * l + 2], sqrt_q32(dl));
            }
        }
        return;
    }
    if (empty && t == 0) return;                  // nothing rendered, no target: every term is zero
    const uint64_t T = t & 0x7FFFFFFFFFull, zq = q32_of_f32(z);
    const uint64_t dq = T > zq ? T - zq : zq - T;
    acc<NEG>(s[SUM_CNT], (uint64_t)(dq != 0));    // unconditional: a zero difference adds zeros
    acc_sq<NEG>(s, dq);
    if (LOSS == ROPE_LOSS_FULL) {
        const unsigned mask = (unsigned)(t >> 40) & 0xFFu;
        const int id = empty ? 255 : (int)(key & 0xFF);
#pragma unroll
        for (int l = 1; l < ROPE_MAX_LINKS; l++) {
            if (l < n_render && ((link_mask >> l) & 1u)) {
                const bool M = (mask >> l) & 1, R = (id == l);
                const uint64_t a = M ? T : 0, b = R ? zq : 0;
                const uint64_t dl = a > b ? a - b : b - a;
                // the two counts of every link live as 12-bit fields of packed words (a thread meets at most 16 samples):
                // pk[0]/pk[1] mask mismatches added / subtracted, pk[2]/pk[3] non-zero link differences added / subtracted
                pk[NEG ? 1 : 0] += (uint64_t)(M != R) << (12 * (l - 1));
                pk[NEG ? 3 : 2] += (uint64_t)(dl != 0) << (12 * (l - 1));
                acc<NEG>(s[SUM_LINK0 + 3 * l + 2], dl);
            }
        }
    }
}

// Pixels of a tile that take part in the loss: inside the image and, for the lookup loss, the crop.
__device__ static inline bool pixel_active(int row, int col, int W, int H, int r0, int r1, int c0, int c1)
{
    return row < H && col < W && row >= r0 && row <= r1 && col >= c0 && col <= c1;
}

// Loss sums of one tile.  DELTA = false: plain sums of a tile with nothing rendered.
// DELTA = true: sums(tile) - sums(base tile), which only the samples whose key differs contribute to —
// all others are skipped without touching the target planes.  Arithmetic is
// modulo 2^64, the frame total of the "nothing rendered" sums is added back by finalize_argmin_kernel.
// Rows [r_lo, r_hi] and 4-sample column groups [g_lo, g_hi] of a tile that a launch may have drawn into.
struct TileRect { int r_lo, r_hi, g_lo, g_hi; int block_g; };   // block_g: log2 of a wave's block width in groups (3, 4 or 5)

template <int LOSS, bool DELTA>
__device__ static inline void score_tile(const uint32_t *tile, const uint32_t *__restrict__ base /* global, or nullptr = nothing */,
                                         int row0, int col0, const FrameParams &fp, int n_render,
                                         const uint64_t *__restrict__ tq, const float *__restrict__ t32,
                                         const uint64_t *__restrict__ tl, uint64_t *lds_sums, const TileRect rc)
{
    const size_t plane = (size_t)fp.W * fp.H;
    uint64_t s[ROPE_SUM_WORDS];
#pragma unroll
    for (int k = 0; k < ROPE_SUM_WORDS; k++) s[k] = 0;
    uint64_t pk[4] = {0, 0, 0, 0};                      // ROPE_LOSS_FULL: packed per-link counts (score_pixel)
    if (DELTA) {
        // Only samples whose key differs from the base tile (the shared layer, or nothing) change the sums, and only
        // the rectangle this launch could draw into can hold any.  The z-test is a minimum over keys, so the tile
        // holds this launch's own samples and min(tile, base) is the finished image: the base is never copied to LDS.
        // A wave takes a compact block of the tile — 2^BLOCK_G groups wide, 64 >> BLOCK_G rows high — instead of two
        // whole rows: the robot's image is a blob, and the blocks inside it keep all their lanes busy.
        static_assert((TILE_W / 4) == 32, "block mapping below assumes 32 four-sample groups per tile row");
        const int bg = rc.block_g, brows = 64 >> bg, bpr_sh = 5 - bg;            // blocks per band of rows: 32 >> bg
        const int n_items = (rc.r_hi - rc.r_lo + 1) * (TILE_W / 4);
        for (int it = threadIdx.x; it < n_items; it += blockDim.x) {
            const int b = it >> 6, l = it & 63;
            const int g = ((b & ((1 << bpr_sh) - 1)) << bg) | (l & ((1 << bg) - 1));
            const int i4 = (rc.r_lo + (b >> bpr_sh) * brows + (l >> bg)) * (TILE_W / 4) + g;
            uint4 k4 = reinterpret_cast<const uint4 *>(tile)[i4];
            const uint4 b4 = base ? reinterpret_cast<const uint4 *>(base)[i4] : make_uint4(KEY_EMPTY, KEY_EMPTY, KEY_EMPTY, KEY_EMPTY);
            k4.x = min(k4.x, b4.x); k4.y = min(k4.y, b4.y); k4.z = min(k4.z, b4.z); k4.w = min(k4.w, b4.w);
            if (k4.x == b4.x && k4.y == b4.y && k4.z == b4.z && k4.w == b4.w) continue;
            const uint32_t keys[4] = {k4.x, k4.y, k4.z, k4.w}, bas[4] = {b4.x, b4.y, b4.z, b4.w};
            const int row = row0 + (4 * i4) / TILE_W, colg = col0 + (4 * i4) % TILE_W;
            if (row >= fp.H || colg >= fp.W) continue;
            const size_t pixg = (size_t)row * fp.W + colg;
            // the four samples sit side by side: fetch their target words together, one wait instead of four
            constexpr bool USE_F = (LOSS == ROPE_LOSS_LOOKUP || LOSS == ROPE_LOSS_TSWEEP);
            uint64_t tw[4] = {0, 0, 0, 0};
            float tf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (colg + 3 < fp.W && !(fp.W & 3)) {
                if (USE_F) {
                    const float4 f = *reinterpret_cast<const float4 *>(t32 + pixg);
                    tf[0] = f.x; tf[1] = f.y; tf[2] = f.z; tf[3] = f.w;
                } else {
                    const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(tq + pixg), b = *reinterpret_cast<const ulonglong2 *>(tq + pixg + 2);
                    tw[0] = a.x; tw[1] = a.y; tw[2] = b.x; tw[3] = b.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (colg + j < fp.W) { if (USE_F) tf[j] = t32[pixg + j]; else tw[j] = tq[pixg + j]; }
            }
            // ROPE_LOSS_FULL: a link's terms at a sample depend on the image only through "is this the link drawn here", so in
            // sums(tile) - sums(base) every link that is drawn in neither cancels, sample by sample and exactly.  The links that
            // are drawn somewhere in this wave's block (usually one or two of the five) are the only ones whose terms are worked
            // out: a uniform branch per link, taken from a vote.
            unsigned links = 0xFFu;
            if (LOSS == ROPE_LOSS_FULL) {
                unsigned mine = 0;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (keys[j] != bas[j]) mine |= (keys[j] == KEY_EMPTY ? 0u : 1u << (keys[j] & 7u)) | (bas[j] == KEY_EMPTY ? 0u : 1u << (bas[j] & 7u));
                links = 0;
#pragma unroll
                for (int l = 1; l < ROPE_MAX_LINKS; l++) links |= __ballot((mine >> l) & 1u) ? 1u << l : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (keys[j] == bas[j]) continue;
                if (!pixel_active(row, colg + j, fp.W, fp.H, fp.r0, fp.r1, fp.c0, fp.c1)) continue;
                score_pixel<LOSS, false>(keys[j], pixg + j, n_render, tw[j], tf[j], tl, plane, fp.c_num, fp.c_sum, fp.c_dif, s, pk, links);
                score_pixel<LOSS, true>(bas[j], pixg + j, n_render, tw[j], tf[j], tl, plane, fp.c_num, fp.c_sum, fp.c_dif, s, pk, links);
            }
        }
    } else {
        for (int i = threadIdx.x; i < TILE_W * TILE_H; i += blockDim.x) {
            int row = row0 + i / TILE_W, col = col0 + i % TILE_W;
            if (!pixel_active(row, col, fp.W, fp.H, fp.r0, fp.r1, fp.c0, fp.c1)) continue;
            const size_t pix = (size_t)row * fp.W + col;
            constexpr bool USE_F = (LOSS == ROPE_LOSS_LOOKUP || LOSS == ROPE_LOSS_TSWEEP);
            score_pixel<LOSS>(KEY_EMPTY, pix, n_render, USE_F ? 0ull : tq[pix], USE_F ? t32[pix] : 0.0f, tl, plane, fp.c_num, fp.c_sum, fp.c_dif, s, pk);
        }
    }
    if (LOSS == ROPE_LOSS_CAMFULL) {
#pragma unroll
        for (int l = 0; l < ROPE_MAX_LINKS; l++) {
            s[SUM_LINK0 + 3 * l] = ((pk[0] >> (10 * l)) & 0x3FFull) - ((pk[1] >> (10 * l)) & 0x3FFull);
            s[SUM_LINK0 + 3 * l + 1] = ((pk[2] >> (10 * l)) & 0x3FFull) - ((pk[3] >> (10 * l)) & 0x3FFull);
        }
    }
    if (LOSS == ROPE_LOSS_FULL) {
#pragma unroll
        for (int l = 1; l < ROPE_MAX_LINKS; l++) {
            s[SUM_LINK0 + 3 * l] = ((pk[0] >> (12 * (l - 1))) & 0xFFFull) - ((pk[1] >> (12 * (l - 1))) & 0xFFFull);
            s[SUM_LINK0 + 3 * l + 1] = ((pk[2] >> (12 * (l - 1))) & 0xFFFull) - ((pk[3] >> (12 * (l - 1))) & 0xFFFull);
        }
    }
    // one LDS atomic per word and wave: the partial sums (modulo 2^64) are added up across the lanes first, and
    // waves that met no sample skip the whole step
    bool any = false;
#pragma unroll
    for (int k = 0; k < ROPE_SUM_WORDS; k++) any = any || s[k] != 0;
    if (__ballot(any) == 0) return;
#pragma unroll
    for (int k = 0; k < ROPE_SUM_WORDS; k++) {
        const bool used = (LOSS == ROPE_LOSS_FULL || LOSS == ROPE_LOSS_CAMFULL) ? true : (k < SUM_LINK0);
        if (!used) continue;
        uint64_t v = s[k];
        // a word nobody in the wave added to (the terms of a link that is drawn nowhere in the wave's block: most of the 20 words
        // of the full loss) needs no exchange — the twelve steps of one cost more than a thread's share of the samples
        if ((LOSS == ROPE_LOSS_FULL || LOSS == ROPE_LOSS_CAMFULL) && __ballot(v != 0) == 0) continue;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd((unsigned long long *)&lds_sums[k], (unsigned long long)v);
    }
}

// Sums of every tile when nothing is rendered into it; the raster kernel adds
// (actual - empty) for the tiles it touches, finalize adds the frame total back.
// grid = (tiles, frames): the planes of frame f start f planes (tl: 6 f planes) into the arrays
template <int LOSS>
__global__ void __launch_bounds__(NTHREADS)
empty_tile_kernel(FrameParams fp, const uint64_t *__restrict__ tq, const float *__restrict__ t32, const uint64_t *__restrict__ tl,
                  uint64_t *__restrict__ empty_sums /* frames x n_tiles x SUM_WORDS */)
{
    __shared__ uint64_t lds_sums[ROPE_SUM_WORDS];
    if (threadIdx.x < ROPE_SUM_WORDS) lds_sums[threadIdx.x] = 0;
    __syncthreads();
    const size_t plane = (size_t)fp.W * fp.H, f = blockIdx.y;
    int tile = blockIdx.x, tx = tile % fp.tiles_x, ty = tile / fp.tiles_x;
    score_tile<LOSS, false>(nullptr, nullptr, ty * TILE_H, tx * TILE_W, fp, ROPE_MAX_LINKS, tq ? tq + f * plane : nullptr, t32 ? t32 + f * plane : nullptr,
                            tl ? tl + f * plane * ROPE_MAX_LINKS : nullptr, lds_sums, TileRect{0, TILE_H - 1, 0, TILE_W / 4 - 1, 5});
    __syncthreads();
    if (threadIdx.x < ROPE_SUM_WORDS) empty_sums[((size_t)f * gridDim.x + tile) * ROPE_SUM_WORDS + threadIdx.x] = lds_sums[threadIdx.x];
}

// one workgroup per frame
__global__ void total_tiles_kernel(const uint64_t *__restrict__ empty_sums, int n_tiles, uint64_t *__restrict__ total)
{
    int k = threadIdx.x;
    if (k >= ROPE_SUM_WORDS) return;
    const uint64_t *e = empty_sums + (size_t)blockIdx.x * n_tiles * ROPE_SUM_WORDS;
    uint64_t t = 0;
    for (int i = 0; i < n_tiles; i++) t += e[(size_t)i * ROPE_SUM_WORDS + k];
    total[(size_t)blockIdx.x * ROPE_SUM_WORDS + k] = t;
}

// ------------------------------------------------------------------ raster -----
// Profiling build only (-DROPE_PROFILE): run-time checks of the indices into the raster kernel's shared arrays and queue segments;
// violations are counted (rope_debug_bounds) instead of trusted.  The shipped library has none of this code.
#ifdef ROPE_PROFILE
__device__ unsigned int g_bounds_violations;
#define ROPE_CHECK_INDEX(idx, limit) do { if ((long long)(idx) < 0 || (long long)(idx) >= (long long)(limit)) atomicAdd(&g_bounds_violations, 1u); } while (0)
hipError_t read_bounds_violations(unsigned int *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds_violations), sizeof(unsigned int)); }
#else
#define ROPE_CHECK_INDEX(idx, limit) do { } while (0)
#endif

// number of set bits of `m` below this lane (v_mbcnt: no lane-mask registers to keep alive)
__device__ static inline int bits_below_lane(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

struct SVert { int32_t X, Y; float d; };
#define SV_BAD INT32_MIN

__device__ static inline SVert shade_vertex(const float *m, float x, float y, float z, float hw, float hh)
{
    float cx = fmaf(m[0], x, fmaf(m[1], y, fmaf(m[2], z, m[3])));
    float cy = fmaf(m[4], x, fmaf(m[5], y, fmaf(m[6], z, m[7])));
    float cz = fmaf(m[8], x, fmaf(m[9], y, fmaf(m[10], z, m[11])));
    float cw = fmaf(m[12], x, fmaf(m[13], y, fmaf(m[14], z, m[15])));
    bool ok = (cw > 0.0f) && (cz >= -cw) && (cz <= cw);
    float rw = 1.0f / cw;
    float sx = fmaf(cx * rw, hw, hw);
    float sy = fmaf(cy * rw, hh, hh);
    SVert o;
    o.d = fmaf(cz * rw, 0.5f, 0.5f);
    ok = ok && (fabsf(sx) < 1.0e6f) && (fabsf(sy) < 1.0e6f);
    o.X = ok ? (int32_t)rintf(sx * 256.0f) : SV_BAD;
    o.Y = ok ? (int32_t)rintf(sy * 256.0f) : 0;
    return o;
}

// ---- near-plane clipping (the rare path: a meshlet that reaches the near plane).  OpenGL clips every primitive against the
// view volume before the viewport transform; pyrender draws through GL (render.py:92-98) with znear 0.05 m
// (projection.py:161-169).  Same operations in the same order as oracle/rope_oracle.c (to_window, clip_near).
struct ClipVert { float cx, cy, cz, cw; };

__device__ static inline ClipVert clip_coords(const float *m, float x, float y, float z)
{
    ClipVert v;
    v.cx = fmaf(m[0], x, fmaf(m[1], y, fmaf(m[2], z, m[3])));
    v.cy = fmaf(m[4], x, fmaf(m[5], y, fmaf(m[6], z, m[7])));
    v.cz = fmaf(m[8], x, fmaf(m[9], y, fmaf(m[10], z, m[11])));
    v.cw = fmaf(m[12], x, fmaf(m[13], y, fmaf(m[14], z, m[15])));
    return v;
}

// viewport transform + sub-pixel snap; false: unusable (beyond the far plane, w <= 0, or off by more than 1e6 px)
__device__ static inline bool to_window(const ClipVert &v, float hw, float hh, SVert &o)
{
    bool ok = (v.cw > 0.0f) && (v.cz <= v.cw);
    const float rw = 1.0f / v.cw;
    const float sx = fmaf(v.cx * rw, hw, hw);
    const float sy = fmaf(v.cy * rw, hh, hh);
    o.d = fmaf(v.cz * rw, 0.5f, 0.5f);
    ok = ok && (fabsf(sx) < 1.0e6f) && (fabsf(sy) < 1.0e6f);
    o.X = ok ? (int32_t)rintf(sx * 256.0f) : SV_BAD;
    o.Y = ok ? (int32_t)rintf(sy * 256.0f) : 0;
    return ok;
}

// where the edge from `in` (z >= -w) to `out` (behind the near plane) meets the plane; always from the inside vertex, so
// the two triangles that share the edge get the same point
__device__ static inline ClipVert clip_near(const ClipVert &in, const ClipVert &out)
{
    const float bi = in.cz + in.cw, bo = out.cz + out.cw;
    const float t = bi / (bi - bo);
    ClipVert n;
    n.cx = fmaf(t, out.cx - in.cx, in.cx);
    n.cy = fmaf(t, out.cy - in.cy, in.cy);
    n.cz = fmaf(t, out.cz - in.cz, in.cz);
    n.cw = fmaf(t, out.cw - in.cw, in.cw);
    return n;
}

__device__ static inline bool owns(int32_t ax, int32_t ay, int32_t bx, int32_t by)
{
    int32_t dy = by - ay, dx = bx - ax;
    return (dy < 0) || (dy == 0 && dx < 0);
}

__device__ static inline int64_t edge_fn(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t fx, int32_t fy)
{
    return (int64_t)(bx - ax) * (int64_t)(fy - ay) - (int64_t)(by - ay) * (int64_t)(fx - ax);
}

// Screen bounding box (GL window pixels, y up, inclusive) of every meshlet of every candidate, from the
// eight corners of its link-frame box, plus per candidate a bit mask of the tiles any meshlet may touch.
// Conservative by construction (1 px margin over rounding and sub-pixel snapping): it only ever removes
// work that cannot produce a sample, never a sample.  Empty boxes are stored as x0 > x1.
// box of meshlet m under the link matrices `mvp6` (6 x 16 floats); marks the tiles it may touch in s_mask (LDS)
// s_tris (LDS, one counter per tile, or nullptr): triangles of the candidate's own links (>= n_shared) whose meshlet may touch
// the tile — the weight of the (candidate, tile) pair in the raster queue
__device__ static inline short4 meshlet_box(const FrameParams &fp, const RobotParams &rp, int m, int n_render, int n_shared,
                                            const float *mvp6, uint32_t (*s_mask)[MAX_MASK_WORDS], uint32_t *s_tris = nullptr,
                                            uint32_t *s_tris_lo = nullptr /* the same for the shared links lo_first .. n_shared - 1 */, int lo_first = 0)
{
    short4 bb = make_short4(1, 0, 1, 0);
    const int l = (int)rp.ml_header[8 * m + 7];
    if (l >= n_render) return bb;
    const float4 ctr = reinterpret_cast<const float4 *>(rp.ml_aabb)[2 * m];
    const float4 ext = reinterpret_cast<const float4 *>(rp.ml_aabb)[2 * m + 1];
    const float *mm = mvp6 + l * 16;
    const float hw = 0.5f * (float)fp.W, hh = 0.5f * (float)fp.H;
    float sxlo = 3.0e38f, sxhi = -3.0e38f, sylo = 3.0e38f, syhi = -3.0e38f;
    bool behind = false, front = false, near = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float x = ctr.x + ((k & 1) ? ext.x : -ext.x), y = ctr.y + ((k & 2) ? ext.y : -ext.y),
                    z = ctr.z + ((k & 4) ? ext.z : -ext.z);
        const float cx = fmaf(mm[0], x, fmaf(mm[1], y, fmaf(mm[2], z, mm[3])));
        const float cy = fmaf(mm[4], x, fmaf(mm[5], y, fmaf(mm[6], z, mm[7])));
        const float cz = fmaf(mm[8], x, fmaf(mm[9], y, fmaf(mm[10], z, mm[11])));
        const float cw = fmaf(mm[12], x, fmaf(mm[13], y, fmaf(mm[14], z, mm[15])));
        // z + w is affine in the position, so its smallest value over the box is at a corner: a meshlet none of whose
        // corners comes within a millimetre of the near plane has no vertex behind it (margin over the rounding of both)
        if (cz + cw < fmaf(1.0e-5f, fabsf(cw), 1.0e-3f)) near = true;
        if (cw <= 1e-4f) { behind = true; continue; }
        front = true;
        const float rw = 1.0f / cw;
        const float sx = fmaf(cx * rw, hw, hw), sy = fmaf(cy * rw, hh, hh);
        sxlo = fminf(sxlo, sx); sxhi = fmaxf(sxhi, sx);
        sylo = fminf(sylo, sy); syhi = fmaxf(syhi, sy);
    }
    if (!front) return bb;
    int x0, x1, y0, y1;
    // straddles the eye plane: cannot bound; reaches the near plane: its triangles are clipped there and the pieces may lie
    // anywhere inside the projection of what is in front — not bounded either (rare: a camera within centimetres)
    if (behind || near) { x0 = 0; x1 = fp.W - 1; y0 = 0; y1 = fp.H - 1; }
    else {
        // sample centre p+0.5 inside [lo,hi] (+ margin)  <=>  p in [lo-1.5, hi+0.5]
        x0 = (int)fmaxf(floorf(sxlo - 1.5f), 0.0f); x1 = (int)fminf(ceilf(sxhi + 0.5f), (float)(fp.W - 1));
        y0 = (int)fmaxf(floorf(sylo - 1.5f), 0.0f); y1 = (int)fminf(ceilf(syhi + 0.5f), (float)(fp.H - 1));
    }
    if (x0 > x1 || y0 > y1) return bb;
    // bit 14 of x0: the meshlet's unclamped screen extent is at most COMPACT_PX both ways
    // bit 13 of x0 (images are at most 8192 wide): the meshlet reaches the near plane and takes the clipping path
    const bool compact = !behind && !near && (sxhi - sxlo) <= (float)COMPACT_PX && (syhi - sylo) <= (float)COMPACT_PX;
    bb = make_short4((short)(x0 | (compact ? 0x4000 : 0) | (near ? 0x2000 : 0)), (short)x1, (short)y0, (short)y1);
    const int tx0 = x0 / TILE_W, tx1 = x1 / TILE_W;
    const int ty0 = (fp.H - 1 - y1) / TILE_H, ty1 = (fp.H - 1 - y0) / TILE_H;
    for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) {
            const int t = ty * fp.tiles_x + tx;
            if (s_mask) atomicOr(&s_mask[l < n_shared ? 0 : 1][t >> 5], 1u << (t & 31));
            if (s_tris && l >= n_shared) atomicAdd(&s_tris[t], rp.ml_header[8 * m + 6] >> 16);
            if (s_tris_lo && l < n_shared && l >= lo_first) atomicAdd(&s_tris_lo[t], rp.ml_header[8 * m + 6] >> 16);
        }
    return bb;
}

__global__ void __launch_bounds__(256)
bounds_kernel(FrameParams fp, RobotParams rp, int n_render, int n_shared, const float *__restrict__ mvp_all,
              short4 *__restrict__ bounds, uint32_t *__restrict__ mask_lo, uint32_t *__restrict__ mask_hi, int mask_words,
              const int32_t *__restrict__ layer_of, const int32_t *__restrict__ layer_rep,
              uint32_t *__restrict__ tile_tris /* C x n_tiles (n_tiles <= QUEUE_WEIGHT_TILES), cleared by fk_mvp_kernel; or nullptr */,
              uint32_t *__restrict__ tile_tris_lo /* with tile_tris: the same for the shared links lo_first .. n_shared - 1 */, int lo_first)
{
    __shared__ uint32_t s_mask[2][MAX_MASK_WORDS];      // [0]: links < n_shared, [1]: the others
    __shared__ uint32_t s_tris[QUEUE_WEIGHT_TILES], s_tris_lo[QUEUE_WEIGHT_TILES];
    const int cand = blockIdx.y, m = blockIdx.x * blockDim.x + threadIdx.x, n_tiles = fp.tiles_x * fp.tiles_y;
    for (int i = threadIdx.x; i < mask_words; i += blockDim.x) s_mask[0][i] = s_mask[1][i] = 0;
    if (tile_tris)
        for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) s_tris[i] = s_tris_lo[i] = 0;
    __syncthreads();
    // the shared links are only ever drawn for a layer's representative candidate: the others need neither their
    // boxes nor their tile mask (raster_score_kernel reads the representative's mask_lo)
    const bool shared_elsewhere = n_shared > 0 && layer_of && layer_rep[layer_of[cand]] != cand;
    if (m < rp.n_meshlets) {
        const bool skip = shared_elsewhere && (int)rp.ml_header[8 * m + 7] < n_shared;
        bounds[(size_t)cand * rp.n_meshlets + m] =
            skip ? make_short4(1, 0, 1, 0) : meshlet_box(fp, rp, m, n_render, n_shared, mvp_all + (size_t)cand * ROPE_MAX_LINKS * 16, s_mask,
                                                         tile_tris ? s_tris : nullptr, tile_tris ? s_tris_lo : nullptr, lo_first);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < mask_words; i += blockDim.x) {
        if (s_mask[0][i]) atomicOr(&mask_lo[(size_t)cand * mask_words + i], s_mask[0][i]);
        if (s_mask[1][i]) atomicOr(&mask_hi[(size_t)cand * mask_words + i], s_mask[1][i]);
    }
    if (tile_tris)
        for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) {
            if (s_tris[i]) atomicAdd(&tile_tris[(size_t)cand * n_tiles + i], s_tris[i]);
            if (s_tris_lo[i]) atomicAdd(&tile_tris_lo[(size_t)cand * n_tiles + i], s_tris_lo[i]);
        }
}

// Small batches: forward kinematics and screen boxes of one candidate in ONE workgroup and one launch.  The five
// joint matrices are built side by side, one thread chains them, 96 threads form the float32 link matrices, then
// all 1024 walk the meshlets.  Same helper code and operation order as fk_mvp_kernel + bounds_kernel: same bits.
__global__ void __launch_bounds__(1024)
fk_bounds_kernel(FrameParams fp, RobotParams rp, const double *__restrict__ cand, int n_render, int n_shared,
                 const double *__restrict__ joint_fixed, const double *__restrict__ joint_axes, const double *__restrict__ PV_all,
                 const int32_t *__restrict__ view_of, float *__restrict__ mvp_all, short4 *__restrict__ bounds,
                 uint64_t *__restrict__ sums, uint32_t *__restrict__ mask_lo, uint32_t *__restrict__ mask_hi, int mask_words)
{
    __shared__ double s_A[ROPE_MAX_LINKS - 1][12], s_T[ROPE_MAX_LINKS][12];
    __shared__ float s_m[ROPE_MAX_LINKS * 16];
    __shared__ uint32_t s_mask[2][MAX_MASK_WORDS];
    const int c = blockIdx.x, tid = threadIdx.x;
    const double *__restrict__ PV = PV_all + (view_of ? 16 * (size_t)view_of[c] : 0);
    if (tid < ROPE_SUM_WORDS) sums[(size_t)c * ROPE_SUM_WORDS + tid] = 0;
    for (int i = tid; i < mask_words; i += blockDim.x) s_mask[0][i] = s_mask[1][i] = 0;
    if (tid < ROPE_MAX_LINKS - 1 && tid + 1 < n_render) joint_matrix(cand[6 * c + tid], tid, joint_fixed, joint_axes, s_A[tid]);
    __syncthreads();
    if (tid == 0) {
        double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, N[12];
        for (int l = 0; l < n_render; l++) {
            if (l > 0) {
                aff_mul(T, s_A[l - 1], N);
#pragma unroll
                for (int k = 0; k < 12; k++) T[k] = N[k];
            }
#pragma unroll
            for (int k = 0; k < 12; k++) s_T[l][k] = T[k];
        }
    }
    __syncthreads();
    if (tid < n_render * 16) {
        const float v = mvp_element(PV, s_T[tid >> 4], (tid >> 2) & 3, tid & 3);
        s_m[tid] = v;
        mvp_all[(size_t)c * ROPE_MAX_LINKS * 16 + tid] = v;
    }
    __syncthreads();
    for (int m = tid; m < rp.n_meshlets; m += blockDim.x)
        bounds[(size_t)c * rp.n_meshlets + m] = meshlet_box(fp, rp, m, n_render, n_shared, s_m, s_mask);
    __syncthreads();
    for (int i = tid; i < mask_words; i += blockDim.x) {
        mask_lo[(size_t)c * mask_words + i] = s_mask[0][i];
        mask_hi[(size_t)c * mask_words + i] = s_mask[1][i];
    }
}

// Tile frame: u = px - col0 in [0,TILE_W), v = py - vy0 in [0,TILE_H) with py the GL window
// row (y up); LDS slot = (TILE_H-1-v)*TILE_W + u, i.e. image rows top-down.
struct TileFrame {
    int col0, vy0;            // window coordinates of (u,v) = (0,0); vy0 may be negative on the last tile row
    int u1, v0;               // valid samples: u in [0,u1], v in [v0,TILE_H-1]
};

// Half-space of edge a->b for tile-relative integer sample coordinates:
//   covered side  <=>  A*u + B*v >= K        (exactly edge_fn(...)+bias >= 0 at the sample centre)
struct Edge { int32_t A, B, K; };
constexpr int32_t EDGE_COEF_LIMIT = 1 << 22;     // |A|,|B| below this keep A*u+B*v inside int32
constexpr int64_t EDGE_K_LIMIT = 1 << 30;

__device__ static inline Edge make_edge(int32_t ax, int32_t ay, int32_t bx, int32_t by, const TileFrame &tf)
{
    Edge e;
    e.A = -(by - ay);
    e.B = bx - ax;
    const int64_t bias = owns(ax, ay, bx, by) ? 0 : -1;
    const int64_t Cc = (int64_t)128 * ((int64_t)e.A + e.B) - (int64_t)e.A * ax - (int64_t)e.B * ay + bias;
    int64_t K = -(Cc >> 8) - (int64_t)e.A * tf.col0 - (int64_t)e.B * tf.vy0;
    K = K > EDGE_K_LIMIT ? EDGE_K_LIMIT : (K < -EDGE_K_LIMIT ? -EDGE_K_LIMIT : K);
    e.K = (int32_t)K;
    return e;
}

// Same half-space for a triangle of a "compact" meshlet (screen extent <= COMPACT_PX in x and y, meeting this
// tile): every coordinate difference is below 2^14 and every tile-relative vertex coordinate below 2^16, so the
// products fit 32 bits and the operands 24 bits.  Algebraically identical to make_edge: with tile-relative
// vertices the -A*col0 - B*vy0 term of K is a multiple of 256 inside Cc.
__device__ static inline Edge make_edge_compact(int32_t ax, int32_t ay, int32_t bx, int32_t by, const TileFrame &tf)
{
    Edge e;
    e.A = -(by - ay);
    e.B = bx - ax;
    const int32_t rx = ax - 256 * tf.col0, ry = ay - 256 * tf.vy0;
    const int32_t Cc = 128 * (e.A + e.B) - __mul24(e.A, rx) - __mul24(e.B, ry) + (owns(ax, ay, bx, by) ? 0 : -1);
    e.K = -(Cc >> 8);
    return e;
}

__device__ static inline int32_t edge_fn_compact(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t fx, int32_t fy)
{
    return __mul24(bx - ax, fy - ay) - __mul24(by - ay, fx - ax);
}

// Window-depth plane of a front-facing triangle, anchored at the pixel that holds vertex a.
struct Plane { float gx, gy, dc; };

// depth plane from the float32 images of area2 and of the two edge functions at the anchor
__device__ static inline Plane plane_from(const SVert &a, const SVert &b, const SVert &c, float area_f, float E20a_f, float E01a_f)
{
    const float inv = 1.0f / area_f;
    const float e1 = b.d - a.d, e2 = c.d - a.d;
    const float fA20 = (float)(-(a.Y - c.Y)), fB20 = (float)(a.X - c.X);
    const float fA01 = (float)(-(b.Y - a.Y)), fB01 = (float)(b.X - a.X);
    Plane p;
    p.gx = (((e1 * fA20) + (e2 * fA01)) * inv) * 256.0f;
    p.gy = (((e1 * fB20) + (e2 * fB01)) * inv) * 256.0f;
    p.dc = a.d + (((e1 * E20a_f) + (e2 * E01a_f)) * inv);
    return p;
}

// compact meshlets: the integers fit 32 bits, and (float)(double)x == (float)x for a 32-bit integer x
__device__ static inline Plane make_plane_compact(const SVert &a, const SVert &b, const SVert &c, int32_t area2)
{
    const int32_t fxa = (a.X >> 8) * 256 + 128, fya = (a.Y >> 8) * 256 + 128;
    return plane_from(a, b, c, (float)area2, (float)edge_fn_compact(c.X, c.Y, a.X, a.Y, fxa, fya),
                      (float)edge_fn_compact(a.X, a.Y, b.X, b.Y, fxa, fya));
}

__device__ static inline Plane make_plane(const SVert &a, const SVert &b, const SVert &c, int64_t area2)
{
    const int32_t fxa = (a.X >> 8) * 256 + 128, fya = (a.Y >> 8) * 256 + 128;
    const int64_t E20a = edge_fn(c.X, c.Y, a.X, a.Y, fxa, fya), E01a = edge_fn(a.X, a.Y, b.X, b.Y, fxa, fya);
    return plane_from(a, b, c, (float)(double)area2, (float)(double)E20a, (float)(double)E01a);
}

// rint(d * (2^24 - 1)) clamped to [0, 2^24 - 1], NaN -> 0: `!(qf >= 0) ? 0 : (qf >= 16777215 ? D24_MAX : (uint32_t)qf)` of
// the specification.  v_cvt_u32_f32 saturates by itself — NaN and negatives give 0, large values 2^32 - 1 — so one conversion
// and one minimum say the same for every input (the C cast alone would be undefined out of range, hence the instruction).
__device__ static inline uint32_t depth24(float d)
{
    const float qf = rintf(d * 16777215.0f);
    uint32_t q;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(qf));
    return min(q, D24_MAX);
}

__device__ static inline void depth_test_write(uint32_t *tile, int u, int v, const Plane &pl, float dx, float dy, uint32_t link)
{
    const uint32_t d24 = depth24(fmaf(pl.gx, dx, fmaf(pl.gy, dy, pl.dc)));
    if (d24 < D24_MAX)                              // GL_LESS against the cleared depth of 1.0
        atomicMin(&tile[(TILE_H - 1 - v) * TILE_W + u], (d24 << 8) | link);
}

// One triangle of window vertices into the tile, sample by sample in 64-bit arithmetic by ONE lane: the exact form the
// fast paths are algebraically equal to.  Used where speed does not matter: triangles with edges beyond 16384 px, and
// the pieces of triangles clipped at the near plane.  [wx0, wx1] x [wy0, wy1]: the tile in window pixels.
__device__ __forceinline__ void raster_exact(uint32_t *tile, const SVert &a, const SVert &b, const SVert &c, uint32_t l, const TileFrame &tf,
                                           int wx0, int wx1, int wy0, int wy1)
{
    const int64_t area2 = (int64_t)(b.X - a.X) * (int64_t)(c.Y - a.Y) - (int64_t)(c.X - a.X) * (int64_t)(b.Y - a.Y);
    if (area2 <= 0) return;                                   // GL_BACK culled, CCW = front
    const int32_t minX = min(a.X, min(b.X, c.X)), maxX = max(a.X, max(b.X, c.X));
    const int32_t minY = min(a.Y, min(b.Y, c.Y)), maxY = max(a.Y, max(b.Y, c.Y));
    const int x0 = max(-((-(minX - 128)) >> 8), wx0), x1 = min((maxX - 128) >> 8, wx1);
    const int y0 = max(-((-(minY - 128)) >> 8), wy0), y1 = min((maxY - 128) >> 8, wy1);
    if (x0 > x1 || y0 > y1) return;
    const Plane pl = make_plane(a, b, c, area2);
    const int32_t pxa = a.X >> 8, pya = a.Y >> 8;
    const int64_t b01 = owns(a.X, a.Y, b.X, b.Y) ? 0 : -1, b12 = owns(b.X, b.Y, c.X, c.Y) ? 0 : -1, b20 = owns(c.X, c.Y, a.X, a.Y) ? 0 : -1;
    for (int py = y0; py <= y1; py++)
        for (int px = x0; px <= x1; px++) {
            const int32_t fx = px * 256 + 128, fy = py * 256 + 128;
            if (((edge_fn(a.X, a.Y, b.X, b.Y, fx, fy) + b01) | (edge_fn(b.X, b.Y, c.X, c.Y, fx, fy) + b12) |
                 (edge_fn(c.X, c.Y, a.X, a.Y, fx, fy) + b20)) < 0) continue;
            depth_test_write(tile, px - tf.col0, py - tf.vy0, pl, (float)(px - pxa), (float)(py - pya), l);
        }
}

// A meshlet that reaches the near plane: one triangle per lane, from the link-frame vertices, cut at the plane where it
// crosses it, and every piece drawn by raster_exact.  Slow and rare — a camera within centimetres of the robot; the
// meshlets that stay in front of the plane never come here.  Called in a pass of its own after the tile's other meshlets
// (inside their loop its register needs cost the hot path 4 %: measured).
__device__ __forceinline__ void clipped_meshlet(uint32_t *tile, const float *mm, const float *verts, const uint32_t *tris, int nt,
                                                                 uint32_t l, float hw, float hh, const TileFrame tf, int wx0, int wx1, int wy0, int wy1)
{
    for (int t = (int)(threadIdx.x & 63); t < nt; t += 64) {
        const uint32_t packed = tris[t];
        const float *p0 = verts + 3 * (size_t)(packed & 0xFF), *p1 = verts + 3 * (size_t)((packed >> 8) & 0xFF), *p2 = verts + 3 * (size_t)((packed >> 16) & 0xFF);
        const ClipVert c0 = clip_coords(mm, p0[0], p0[1], p0[2]), c1 = clip_coords(mm, p1[0], p1[1], p1[2]), c2 = clip_coords(mm, p2[0], p2[1], p2[2]);
        const bool n0 = c0.cz < -c0.cw, n1 = c1.cz < -c1.cw, n2 = c2.cz < -c2.cw;       // behind the near plane
        const int n_near = (int)n0 + (int)n1 + (int)n2;
        if (n_near == 3) continue;
        SVert s0 = {SV_BAD, 0, 0.0f}, s1 = s0, s2 = s0;
        // a vertex in front of the plane that has no window position (beyond the far plane, ...) drops the triangle
        if ((!n0 && !to_window(c0, hw, hh, s0)) || (!n1 && !to_window(c1, hw, hh, s1)) || (!n2 && !to_window(c2, hw, hh, s2))) continue;
        // up to two triangles come out; they are drawn by ONE call below (a single inlined copy of the sample loop)
        SVert ta = s0, tb = s1, tc = s2, ua = s0, ub = s0, uc = s0;
        int n_out = 1;
        if (n_near != 0) {
            // rotate the vertex order (the winding stays) so that the odd one out comes first
            const int odd = n_near == 1 ? (n0 ? 0 : (n1 ? 1 : 2)) : (!n0 ? 0 : (!n1 ? 1 : 2));
            const ClipVert cp = odd == 0 ? c0 : (odd == 1 ? c1 : c2), cq = odd == 0 ? c1 : (odd == 1 ? c2 : c0), cr = odd == 0 ? c2 : (odd == 1 ? c0 : c1);
            const SVert sp = odd == 0 ? s0 : (odd == 1 ? s1 : s2), sq = odd == 0 ? s1 : (odd == 1 ? s2 : s0), sr = odd == 0 ? s2 : (odd == 1 ? s0 : s1);
            SVert A, B;
            if (n_near == 1) {
                // p is cut off: the quad A q r B with A on p-q and B on r-p, as the triangles (A, q, r) and (A, r, B)
                if (!to_window(clip_near(cq, cp), hw, hh, A) || !to_window(clip_near(cr, cp), hw, hh, B)) continue;
                ta = A; tb = sq; tc = sr;
                ua = A; ub = sr; uc = B;
                n_out = 2;
            } else {
                // only p is in front: the triangle p A B with A on p-q and B on p-r
                if (!to_window(clip_near(cp, cq), hw, hh, A) || !to_window(clip_near(cp, cr), hw, hh, B)) continue;
                ta = sp; tb = A; tc = B;
            }
        }
        for (int k = 0; k < n_out; k++) {
            raster_exact(tile, k ? ua : ta, k ? ub : tb, k ? uc : tc, l, tf, wx0, wx1, wy0, wy1);
        }
    }
}

// largest u with A*u <= n (A > 0): float estimate, then exact fix-up by one either way; clamped to [-4, TILE_W+3]
__device__ static inline int floor_div_pos(int32_t n, int32_t A, float rcpA)
{
    float q = floorf((float)n * rcpA);
    q = fminf(fmaxf(q, -3.0f), (float)(TILE_W + 2));
    int u = (int)q;
    u -= (__mul24(A, u) > n);
    u += (__mul24(A, u + 1) <= n);
    return u;
}

static_assert(TILE_W <= 256 && TILE_H <= 256, "row-pass items carry tile coordinates in 8 bits");
static_assert(TILE_H % 8 == 0 && TILE_W % 4 == 0, "score_gtile_kernel slices a tile into up to 8 row bands of 16-byte groups");
static_assert(TILE_H % 16 == 0, "the loss pass of raster_score_kernel walks the tile in blocks of 16 rows");

// Samples of row v covered by the half-space: narrows [lo,hi].  One code path for both signs of A (the lanes of
// a wave hold edges of every orientation): A > 0: u >= ceil(n/A) = floor((n+A-1)/A);  A < 0: u <= floor(-n/-A).
__device__ static inline void clip_span(const Edge &e, int v, int &lo, int &hi)
{
    const int32_t n = e.K - __mul24(e.B, v);       // A*u >= n   (|B| < 2^22, 0 <= v < TILE_H, |K| <= 2^30)
    const bool neg = e.A < 0, zero = e.A == 0;
    const int32_t Aa = zero ? 1 : abs(e.A);
    const int32_t m = neg ? -n : n + Aa - 1;
    const int q = floor_div_pos(m, Aa, __builtin_amdgcn_rcpf((float)Aa));
    lo = (neg || zero) ? lo : max(lo, q);
    hi = neg ? min(hi, q) : hi;
    hi = (zero && n > 0) ? -1 : hi;
}


// One (screen tile, row) of a raster launch; `row` = candidate, or in MODE_LAYER a shared layer.  zme / zsplit: MODE_SPLIT's
// share of the tile's meshlets.  Every early return is taken by the whole workgroup.  CLIP: the instantiation that can cut
// triangles at the near plane — a kernel of its own, launched when the camera is close enough to the robot for that to
// happen at all (the host decides from the robot's reach, launch_raster): with the clipping code present the register
// allocation of everything else suffers (4 - 8 % on the bench workload, wherever the code sits: measured).
template <int LOSS, int MODE, bool CLIP>
__device__ __forceinline__ void raster_tile(const FrameParams &fp, const RobotParams &rp, const RasterArgs &ra, const int row,
                                            const int tile_id, const int zme, const int zsplit)
{
    const int n_render = ra.n_render;
    const int cand = (MODE == MODE_LAYER) ? ra.cand_of_row[row] : row;
    // camera-pose path and batches over several frames' targets: the candidate names the frame whose target planes it is scored
    // against (a shared layer: its representative candidate's frame — layers never span frames, upload_candidates)
    const size_t frame = ((MODE == MODE_SCORE || MODE == MODE_LAYER) && ra.frame_of) ? (size_t)ra.frame_of[cand] : 0;
    constexpr bool GEO = (MODE == MODE_SPLIT_GEO);      // forward kinematics and the share's screen boxes worked out here
    const size_t plane = (size_t)fp.W * fp.H;
    const uint64_t *__restrict__ tq = ra.tq ? ra.tq + frame * plane : nullptr;
    const float *__restrict__ t32 = ra.t32 ? ra.t32 + frame * plane : nullptr;
    const uint64_t *__restrict__ tl = ra.tl ? ra.tl + frame * plane * ROPE_MAX_LINKS : nullptr;
    __shared__ __attribute__((aligned(16))) uint32_t tile[TILE_W * TILE_H];
    __shared__ float s_mvp[ROPE_MAX_LINKS * 16];
    __shared__ uint16_t s_list[MAX_MESHLETS];
    __shared__ int s_count;
    __shared__ SVert s_vert[NWAVES][MESHLET_MAX_VERTS];
    __shared__ int s_qoff[NWAVES][64];
    __shared__ unsigned long long s_qmask[NWAVES][TILE_H > 64 ? TILE_H : 64];   // one word per chunk of 64 row items: at most 64 triangles x TILE_H rows
    __shared__ int s_next;
    __shared__ int s_near;                            // the list holds a meshlet that reaches the near plane
    __shared__ uint32_t s_keep[NWAVES][MESHLET_MAX_TRIS];
    __shared__ uint64_t lds_sums[ROPE_SUM_WORDS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile_id % fp.tiles_x, ty = tile_id / fp.tiles_x;
    const int col0 = tx * TILE_W, row0 = ty * TILE_H;
    // tile rectangle in GL window pixel coordinates (y up), clamped to the image
    const int wx0 = col0, wx1 = min(col0 + TILE_W, fp.W) - 1;
    const int wy1 = fp.H - 1 - row0, wy0 = max(fp.H - row0 - TILE_H, 0);
    const float hw = 0.5f * (float)fp.W, hh = 0.5f * (float)fp.H;
    TileFrame tf;
    tf.col0 = col0;
    tf.vy0 = fp.H - row0 - TILE_H;
    tf.u1 = wx1 - col0;
    tf.v0 = wy0 - tf.vy0;

    // tiles no meshlet of this candidate can touch keep their "empty" sums: nothing to do
    const size_t mw = (size_t)cand * ra.mask_words + (tile_id >> 5);
    // tiles of the shared links: kept for the layer's representative candidate only (bounds_kernel)
    const size_t mw_lo = (MODE != MODE_LAYER && ra.layer_of) ? (size_t)ra.layer_rep[ra.layer_of[cand]] * ra.mask_words + (tile_id >> 5) : mw;
    const bool hit_lo = GEO ? true : (ra.mask_lo[mw_lo] >> (tile_id & 31)) & 1u, hit_hi = GEO ? true : (ra.mask_hi[mw] >> (tile_id & 31)) & 1u;
    if (MODE == MODE_LAYER ? !hit_lo : !(hit_lo || hit_hi)) return;
    if (ROPE_SKIP(fp, 1)) return;
    // shared layer (links below l_begin, rendered once per distinct upstream pose) covering this tile
    const uint32_t *layer_tile = nullptr;
    if (MODE != MODE_LAYER && ra.layer_of && hit_lo)
        layer_tile = ra.layers + ((size_t)ra.layer_of[cand] * (fp.tiles_x * fp.tiles_y) + tile_id) * (TILE_W * TILE_H);

    // A candidate whose own links cannot touch this tile (mask_hi comes from the very boxes the list below is built
    // from) keeps its layer's sums: no list, no tile, no barrier.
    if (MODE == MODE_SCORE && layer_tile && !hit_hi) {
        if (tid < ROPE_SUM_WORDS) {
            const uint64_t d = ra.layer_sums[((size_t)ra.layer_of[cand] * (fp.tiles_x * fp.tiles_y) + tile_id) * ROPE_SUM_WORDS + tid];
            if (d) atomicAdd((unsigned long long *)&ra.sums[(size_t)cand * ROPE_SUM_WORDS + tid], (unsigned long long)d);
        }
        return;
    }
    if (tid == 0) { s_count = 0; s_next = 0; s_near = 0; }
    if (tid < ROPE_SUM_WORDS) lds_sums[tid] = 0;
    __syncthreads();

    // --- tile initialisation (a copy of the layer tile, or "empty"), link matrices, and the list of meshlets whose
    // screen box meets this tile: all in one phase so that the global loads overlap
    if (GEO) {
        // the candidate's link matrices, as fk_bounds_kernel forms them (same helpers, same order: same bits); the joint
        // matrices side by side, one thread chains them; the doubles live in the tile's LDS, which is cleared afterwards
        double (*s_A)[12] = reinterpret_cast<double (*)[12]>(tile);
        double (*s_T)[12] = s_A + (ROPE_MAX_LINKS - 1);
        if (tile_id == 0 && zme == 0 && tid < ROPE_SUM_WORDS) ra.sums[(size_t)cand * ROPE_SUM_WORDS + tid] = 0;      // score_gtile_kernel adds into them
        if (tid < ROPE_MAX_LINKS - 1 && tid + 1 < n_render) joint_matrix(ra.cand_q[6 * cand + tid], tid, ra.joint_fixed, ra.joint_axes, s_A[tid]);
        __syncthreads();
        if (tid == 0) {
            double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, N[12];
            for (int l = 0; l < n_render; l++) {
                if (l > 0) {
                    aff_mul(T, s_A[l - 1], N);
#pragma unroll
                    for (int k = 0; k < 12; k++) T[k] = N[k];
                }
#pragma unroll
                for (int k = 0; k < 12; k++) s_T[l][k] = T[k];
            }
        }
        __syncthreads();
        if (tid < n_render * 16) s_mvp[tid] = mvp_element(ra.PV, s_T[tid >> 4], (tid >> 2) & 3, tid & 3);
        __syncthreads();
    } else if (tid < ROPE_MAX_LINKS * 16) s_mvp[tid] = ra.mvp[((size_t)cand * ROPE_MAX_LINKS) * 16 + tid];
    // the shared layer is not copied in: depth testing is a minimum, so it is merged where the tile is consumed
    for (int i = tid; i < TILE_W * TILE_H / 4; i += NTHREADS)
        reinterpret_cast<uint4 *>(tile)[i] = make_uint4(KEY_EMPTY, KEY_EMPTY, KEY_EMPTY, KEY_EMPTY);
    {
        const int m_begin = rp.link_first[ra.l_begin], m_end = rp.link_first[ra.l_end];
        const short4 *bb = ra.bounds + (size_t)cand * rp.n_meshlets;
        // GEO: a thread per meshlet of this workgroup's share (m = first + k * zsplit) instead of a walk over all of them
        for (int m = GEO ? m_begin + ((zme - m_begin % zsplit + zsplit) % zsplit) + tid * zsplit : m_begin + tid; m < m_end; m += GEO ? NTHREADS * zsplit : NTHREADS) {
            if (!GEO && zsplit > 1 && (m % zsplit) != zme) continue;       // this workgroup's share of the meshlets
            const short4 b = GEO ? meshlet_box(fp, rp, m, n_render, 0, s_mvp, nullptr) : bb[m];
            const int bx0 = b.x & 0x1FFF;
            if (bx0 <= b.y && bx0 <= wx1 && b.y >= wx0 && b.z <= wy1 && b.w >= wy0) {
                int pos = atomicAdd(&s_count, 1);
                ROPE_CHECK_INDEX(pos, MAX_MESHLETS);
                s_list[pos] = (uint16_t)(m | ((b.x & 0x4000) ? 0x8000 : 0) | ((b.x & 0x2000) ? 0x4000 : 0));
                if (b.x & 0x2000) s_near = 1;                       // some meshlet of this tile takes the clipping pass below
            }
        }
    }
    __syncthreads();
    const int n_list = s_count;
    if (GEO && n_list > 0 && tid == 0) ra.touched[(size_t)cand * (fp.tiles_x * fp.tiles_y) + tile_id] = ra.pass_id;      // for score_gtile_kernel
    const TileRect rc = {0, TILE_H - 1, 0, TILE_W / 4 - 1, 2};          // 4 groups x 16 rows per wave: measured best of 32x2, 8x8, 4x16, 2x32
    if (n_list == 0 && MODE != MODE_LAYER && !(MODE == MODE_TABLE && layer_tile)) {
        // nothing of this row lands in the tile: its sums stay those of the shared layer, or "empty"
        if (MODE == MODE_SCORE && layer_tile && tid < ROPE_SUM_WORDS) {
            const uint64_t d = ra.layer_sums[((size_t)ra.layer_of[cand] * (fp.tiles_x * fp.tiles_y) + tile_id) * ROPE_SUM_WORDS + tid];
            if (d) atomicAdd((unsigned long long *)&ra.sums[(size_t)cand * ROPE_SUM_WORDS + tid], (unsigned long long)d);
        }
        return;
    }

    // --- one meshlet per wave at a time
    SVert *const wv = s_vert[wave];
    int *const woff = s_qoff[wave];
    unsigned long long *const wmask = s_qmask[wave];
    uint32_t *const wkeep = s_keep[wave];
    // ---- pass 2 (set-up, classify, rasterise) of one batch of up to 64 surviving triangles held in registers
    int pend_n = 0;                                   // wave-uniform: lanes [0, pend_n) hold a pending survivor
    bool pend_compact = true;                         // wave-uniform: all of them come from compact meshlets
    SVert pa = {0, 0, 0.0f}, pb = pa, pc = pa;
    uint32_t plink = 0;
    auto pass2 = [&](const int n_active, const bool compact) {
        {
            int rows = 0;                          // > 0: queued for the row-parallel pass
            // what a queued triangle's owner lane keeps for the lanes that will take its rows
            Edge q0 = {0, 0, 0}, q1 = {0, 0, 0}, q2 = {0, 0, 0};
            Plane qpl = {0.0f, 0.0f, 0.0f};
            // box and link in one word (u_lo | u_hi << 8 | v0 << 16 | link << 24), anchor offsets in another (each within
            // 16 bits: edges reach less than 16384 px here and the box meets the tile)
            int q_box = 0, q_anchor = 0;
            if (lane < n_active) {
                const SVert a = pa, b = pb, c = pc;
                const uint32_t l = plink;
                {
                    const int32_t minX = min(a.X, min(b.X, c.X)), maxX = max(a.X, max(b.X, c.X));
                    const int32_t minY = min(a.Y, min(b.Y, c.Y)), maxY = max(a.Y, max(b.Y, c.Y));
                    const int x0 = max(-((-(minX - 128)) >> 8), wx0), x1 = min((maxX - 128) >> 8, wx1);
                    const int y0 = max(-((-(minY - 128)) >> 8), wy0), y1 = min((maxY - 128) >> 8, wy1);
                    {
                        Plane pl;
                        Edge e0, e1, e2;
                        int32_t big = 0;
                        if (compact) {
                            pl = make_plane_compact(a, b, c, __mul24(b.X - a.X, c.Y - a.Y) - __mul24(c.X - a.X, b.Y - a.Y));
                            e0 = make_edge_compact(a.X, a.Y, b.X, b.Y, tf);
                            e1 = make_edge_compact(b.X, b.Y, c.X, c.Y, tf);
                            e2 = make_edge_compact(c.X, c.Y, a.X, a.Y, tf);
                        } else {
                            pl = make_plane(a, b, c, (int64_t)(b.X - a.X) * (int64_t)(c.Y - a.Y) - (int64_t)(c.X - a.X) * (int64_t)(b.Y - a.Y));
                            e0 = make_edge(a.X, a.Y, b.X, b.Y, tf);
                            e1 = make_edge(b.X, b.Y, c.X, c.Y, tf);
                            e2 = make_edge(c.X, c.Y, a.X, a.Y, tf);
                            big = max(max(abs(e0.A), abs(e0.B)), max(max(abs(e1.A), abs(e1.B)), max(abs(e2.A), abs(e2.B))));
                        }
                        const int32_t pxa = a.X >> 8, pya = a.Y >> 8;
                        const int w = x1 - x0 + 1, h = y1 - y0 + 1;
                        if (big >= EDGE_COEF_LIMIT) {
                            // enormous triangle (edge extent >= 16384 px): exact 64-bit walk, one lane
                            raster_exact(tile, a, b, c, l, tf, wx0, wx1, wy0, wy1);
                        } else if (w <= SMALL_TRI_COLS && h <= SMALL_TRI_ROWS) {
                            // small box: walk its rows, four samples of a row at a time without branching on coverage tests
                            if (!ROPE_SKIP(fp, 32)) {
                                const int u0 = x0 - col0;
                                const int vv = y0 - tf.vy0;       // coefficients are below 2^22 here, u0 and vv below 2^8
                                int t0 = __mul24(e0.A, u0) + __mul24(e0.B, vv), t1 = __mul24(e1.A, u0) + __mul24(e1.B, vv),
                                    t2 = __mul24(e2.A, u0) + __mul24(e2.B, vv);
                                const float dx0 = (float)(x0 - pxa);
                                for (int py = y0; py <= y1; py++, t0 += e0.B, t1 += e1.B, t2 += e2.B) {
                                    const int v = py - tf.vy0;
                                    const float dyf = (float)(py - pya);
#pragma unroll
                                    for (int j = 0; j < SMALL_TRI_COLS; j++) {
                                        const bool in = (j < w) && (t0 + j * e0.A >= e0.K) && (t1 + j * e1.A >= e1.K) && (t2 + j * e2.A >= e2.K);
                                        if (in) depth_test_write(tile, u0 + j, v, pl, dx0 + (float)j, dyf, l);
                                    }
                                }
                            }
                        } else {
                            // one item per row — or per column when the box is taller than wide (upright slivers of
                            // the arm's cylinders): the half-spaces are symmetric in u and v, so the owner swaps A and B
                            // and the items clip a vertical span instead of a horizontal one
                            const bool tall = h > w;
                            rows = tall ? w : h;
                            q0 = e0; q1 = e1; q2 = e2; qpl = pl;
                            if (tall) {
                                q0.A = e0.B; q0.B = e0.A; q1.A = e1.B; q1.B = e1.A; q2.A = e2.B; q2.B = e2.A;
                                q_box = (y0 - tf.vy0) | ((y1 - tf.vy0) << 8) | ((x0 - col0) << 16) | ((int)l << 24) | (1 << 27);
                            } else {
                                q_box = (x0 - col0) | ((x1 - col0) << 8) | ((y0 - tf.vy0) << 16) | ((int)l << 24);
                            }
                            q_anchor = ((col0 - pxa) & 0xFFFF) | ((tf.vy0 - pya) << 16);
                        }
                    }
                }
            }
            // ---- queue the larger triangles, then spread their rows over the lanes.  Queue order is lane
            // order, so a scan of `rows` over the lanes gives every queued triangle its first item; the
            // owner of an item is found from a per-chunk bit mask of "a triangle starts at this item".
            const unsigned long long qmask = __ballot(rows > 0);
            if (qmask == 0 || ROPE_SKIP(fp, 64)) return;
            int incl = rows;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            const int total = __shfl(incl, 63, 64);
            const int excl = incl - rows;
            const int qpos = bits_below_lane(qmask);
            const int nchunks = (total + 63) >> 6;
            ROPE_CHECK_INDEX(nchunks - 1, TILE_H > 64 ? TILE_H : 64);
            for (int i = lane; i < nchunks; i += 64) wmask[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (rows > 0) {
                ROPE_CHECK_INDEX(qpos, 64);
                ROPE_CHECK_INDEX(excl, 65536);
                woff[qpos] = excl | (lane << 16);          // first item (total <= 4096) and owner lane of slot qpos
                atomicOr(&wmask[excl >> 6], 1ull << (excl & 63));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int qn = __popcll(qmask);
            // (profiling build, bit 128: the last chunk dropped when it is a partly filled one after full ones — timing only, wrong
            // images: the upper bound of what carrying its items over to the next batch could save)
            const int nchunks_run = (ROPE_SKIP(fp, 128) && (total & 63) && total > 64) ? nchunks - 1 : nchunks;
            for (int chunk = 0; chunk < nchunks_run; chunk++) {
                const int base = chunk << 6, item = base + lane;
                const int before = __popcll(__ballot(rows > 0 && excl < base));   // triangles that start in earlier chunks
                const unsigned long long mk = wmask[chunk];
                const int slot = max(min(before + bits_below_lane(mk) + (int)((mk >> lane) & 1ull) - 1, qn - 1), 0);
                const int so = woff[slot];
                const int src = so >> 16;                      // owner lane: its registers hold the set-up
                // every lane takes part in the exchanges (inactive source lanes would read as garbage)
                Edge e0, e1, e2;
                e0.A = __shfl(q0.A, src, 64); e0.B = __shfl(q0.B, src, 64); e0.K = __shfl(q0.K, src, 64);
                e1.A = __shfl(q1.A, src, 64); e1.B = __shfl(q1.B, src, 64); e1.K = __shfl(q1.K, src, 64);
                e2.A = __shfl(q2.A, src, 64); e2.B = __shfl(q2.B, src, 64); e2.K = __shfl(q2.K, src, 64);
                Plane pl;
                pl.gx = __shfl(qpl.gx, src, 64); pl.gy = __shfl(qpl.gy, src, 64); pl.dc = __shfl(qpl.dc, src, 64);
                const int box = __shfl(q_box, src, 64), anchor = __shfl(q_anchor, src, 64);
                int lo = box & 0xFF, hi = (box >> 8) & 0xFF;         // span limits along the item's line
                const int c0q = (box >> 16) & 0xFF, dxa = (int)(short)(anchor & 0xFFFF), dya = anchor >> 16;
                const uint32_t lq = ((uint32_t)box >> 24) & 7u;
                const bool tall = (box >> 27) & 1;
                if (item < total) {
                    const int c = c0q + (item - (so & 0xFFFF));          // the row (or, transposed, the column) of this item
                    clip_span(e0, c, lo, hi);
                    clip_span(e1, c, lo, hi);
                    clip_span(e2, c, lo, hi);
                    // one loop for rows and columns alike: a chunk usually holds both kinds, and two loops one after the
                    // other each run as long as their own longest span.  Sample i of the line: (u, v) = (c, i) for a
                    // column, (i, c) for a row; same operands and operation order as depth_test_write either way.
                    const float fixf = (float)(c + (tall ? dxa : dya));
                    const int var_off = tall ? dya : dxa, step = tall ? -TILE_W : 1;
                    int idx = tall ? (TILE_H - 1 - lo) * TILE_W + c : (TILE_H - 1 - c) * TILE_W + lo;
                    for (int i = lo; i <= hi; i++, idx += step) {
                        const float varf = (float)(i + var_off);
                        const uint32_t d24 = depth24(fmaf(pl.gx, tall ? fixf : varf, fmaf(pl.gy, tall ? varf : fixf, pl.dc)));
                        if (d24 < D24_MAX) atomicMin(&tile[idx], (d24 << 8) | lq);          // GL_LESS against the cleared depth of 1.0
                    }
                }
            }
        }
    };
    for (;;) {
        int li = 0;
        if (lane == 0) li = atomicAdd(&s_next, 1);
        li = __builtin_amdgcn_readfirstlane(li);
        if (li >= n_list) break;
        const int m_entry = __builtin_amdgcn_readfirstlane((int)s_list[li]);
        if (CLIP && (m_entry & 0x4000)) continue;         // reaches the near plane: drawn by the clipping pass after this loop
        const int m = m_entry & 0x3FFF;
        const bool compact = (m_entry & 0x8000) != 0;     // wave-uniform: chooses the 32-bit set-up
        const uint4 h1 = reinterpret_cast<const uint4 *>(rp.ml_header)[2 * m + 1];
        const int v0 = (int)h1.x, t0 = (int)h1.y, nv = (int)(h1.z & 0xFFFF), nt = (int)(h1.z >> 16);
        const uint32_t l = h1.w;
        if (ROPE_SKIP(fp, 2)) continue;
        {
            // the link matrix is the same for the whole wave: keep it in scalar registers
            float mm[16];
#pragma unroll
            for (int k = 0; k < 16; k++)
                mm[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_mvp[16 * l + k])));
            static_assert(MESHLET_MAX_VERTS <= 64, "one vertex per lane");
            if (lane < nv) {
                const float *p = rp.ml_verts + 3 * (size_t)(v0 + lane);
                wv[lane] = shade_vertex(mm, p[0], p[1], p[2], hw, hh);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ROPE_SKIP(fp, 4)) continue;
        // ---- pass 1: cull.  Keep front-facing triangles whose bounding box holds a sample of this tile and
        // compact them, so that the expensive set-up below runs on full lanes (about one triangle in four survives).
        int ns = 0;
        static_assert(MESHLET_MAX_TRIS <= 128, "the cull loop fetches its index words for two steps of 64 up front");
        const uint32_t packed_a = lane < nt ? rp.ml_tris[t0 + lane] : 0u;              // both loads in flight together
        const uint32_t packed_b = 64 + lane < nt ? rp.ml_tris[t0 + 64 + lane] : 0u;
        for (int tb = 0; tb < nt; tb += 64) {
            const int t = tb + lane;
            bool keep = false;
            uint32_t packed = 0;
            if (t < nt) {
                packed = tb == 0 ? packed_a : packed_b;
                const SVert a = wv[packed & 0xFF], b = wv[(packed >> 8) & 0xFF], c = wv[(packed >> 16) & 0xFF];
                if (a.X != SV_BAD && b.X != SV_BAD && c.X != SV_BAD) {
                    bool front;
                    if (compact) front = __mul24(b.X - a.X, c.Y - a.Y) - __mul24(c.X - a.X, b.Y - a.Y) > 0;
                    else front = (int64_t)(b.X - a.X) * (int64_t)(c.Y - a.Y) - (int64_t)(c.X - a.X) * (int64_t)(b.Y - a.Y) > 0;
                    const int32_t minX = min(a.X, min(b.X, c.X)), maxX = max(a.X, max(b.X, c.X));
                    const int32_t minY = min(a.Y, min(b.Y, c.Y)), maxY = max(a.Y, max(b.Y, c.Y));
                    const int x0 = max(-((-(minX - 128)) >> 8), wx0), x1 = min((maxX - 128) >> 8, wx1);
                    const int y0 = max(-((-(minY - 128)) >> 8), wy0), y1 = min((maxY - 128) >> 8, wy1);
                    keep = front && x0 <= x1 && y0 <= y1;
                }
            }
            const unsigned long long km = __ballot(keep);
            if (keep) ROPE_CHECK_INDEX(ns + bits_below_lane(km), MESHLET_MAX_TRIS);
            if (keep) wkeep[ns + bits_below_lane(km)] = packed;
            ns += __popcll(km);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ROPE_SKIP(fp, 8)) continue;
        // ---- hand the survivors to free lanes of the pending batch; pass 2 runs whenever 64 are waiting, so that the
        // expensive set-up and the pixel work execute on full waves (survivors of several meshlets share a batch)
        for (int taken = 0; taken < ns;) {
            const int take = min(64 - pend_n, ns - taken);
            if (lane >= pend_n && lane < pend_n + take) {
                const uint32_t packed = wkeep[taken + lane - pend_n];
                pa = wv[packed & 0xFF]; pb = wv[(packed >> 8) & 0xFF]; pc = wv[(packed >> 16) & 0xFF];
                plink = l;
            }
            pend_n += take;
            taken += take;
            pend_compact = pend_compact && compact;
            if (pend_n == 64) {
                pass2(64, pend_compact);
                pend_n = 0;
                pend_compact = true;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (pend_n > 0) pass2(pend_n, pend_compact);
    if (CLIP && s_near) {
        // the meshlets that reach the near plane (bounds_kernel), one per wave at a time: cut and drawn triangle by triangle
        for (int li = wave; li < n_list; li += NWAVES) {
            const int m_entry = __builtin_amdgcn_readfirstlane((int)s_list[li]);
            if (!(m_entry & 0x4000)) continue;
            const uint4 h1 = reinterpret_cast<const uint4 *>(rp.ml_header)[2 * (m_entry & 0x3FFF) + 1];
            clipped_meshlet(tile, s_mvp + 16 * h1.w, rp.ml_verts + 3 * (size_t)h1.x, rp.ml_tris + h1.y, (int)(h1.z >> 16), h1.w, hw, hh, tf, wx0, wx1, wy0, wy1);
        }
    }
    __syncthreads();

    if (MODE == MODE_LAYER) {
        const size_t slot = (size_t)row * (fp.tiles_x * fp.tiles_y) + tile_id;
        // the parent launch skipped the tiles its own candidate's shared links do not touch: nothing to merge there
        const int prow = ra.base_layers ? ra.base_of_row[row] : 0;
        const bool have_base = ra.base_layers && ((ra.mask_lo[(size_t)ra.base_rep[prow] * ra.mask_words + (tile_id >> 5)] >> (tile_id & 31)) & 1u);
        if (have_base) {
            const uint4 *b4 = reinterpret_cast<const uint4 *>(ra.base_layers + ((size_t)prow * (fp.tiles_x * fp.tiles_y) + tile_id) * (TILE_W * TILE_H));
            for (int i = tid; i < TILE_W * TILE_H / 4; i += NTHREADS) {
                uint4 k = reinterpret_cast<uint4 *>(tile)[i];
                const uint4 b = b4[i];
                k.x = min(k.x, b.x); k.y = min(k.y, b.y); k.z = min(k.z, b.z); k.w = min(k.w, b.w);
                reinterpret_cast<uint4 *>(tile)[i] = k;
            }
            __syncthreads();
        }
        uint4 *dst = reinterpret_cast<uint4 *>(ra.layers + slot * (TILE_W * TILE_H));
        for (int i = tid; i < TILE_W * TILE_H / 4; i += NTHREADS) dst[i] = reinterpret_cast<const uint4 *>(tile)[i];
        // loss sums of the layer alone (relative to "nothing rendered"): every candidate on this layer starts from them
        if (ra.layer_sums) {
            score_tile<LOSS, true>(tile, nullptr, row0, col0, fp, n_render, tq, t32, tl, lds_sums, rc);
            __syncthreads();
            if (tid < ROPE_SUM_WORDS) ra.layer_sums[slot * ROPE_SUM_WORDS + tid] = lds_sums[tid];
        }
        return;
    }
    if (MODE == MODE_SPLIT || GEO) {
        // merge this workgroup's share into the candidate's tile in global memory (min is order-free)
        uint32_t *dst = ra.gtile + ((size_t)cand * (fp.tiles_x * fp.tiles_y) + tile_id) * (TILE_W * TILE_H);
        for (int i = tid; i < TILE_W * TILE_H; i += NTHREADS)
            if (tile[i] != KEY_EMPTY) atomicMin(&dst[i], tile[i]);
        return;
    }
    if (MODE == MODE_TABLE) {
        // one row of the lookup table: sqrt of the metric depth over the crop (predict.py:117 sqrt of the table,
        // lookup.py:92 crop); samples nothing was drawn on stay 0 from the table's memset
        const int cw = fp.c1 - fp.c0 + 1, ch = fp.r1 - fp.r0 + 1;
        float *dst = ra.table + (size_t)cand * cw * ch;
        for (int i = tid; i < TILE_W * TILE_H; i += NTHREADS) {
            const int row = row0 + i / TILE_W, col = col0 + i % TILE_W;
            const uint32_t k = layer_tile ? min(tile[i], layer_tile[i]) : tile[i];
            if (k == KEY_EMPTY || !pixel_active(row, col, fp.W, fp.H, fp.r0, fp.r1, fp.c0, fp.c1)) continue;
            dst[(size_t)(row - fp.r0) * cw + (col - fp.c0)] = sqrtf(linear_depth(k >> 8, fp.c_num, fp.c_sum, fp.c_dif));
        }
        return;
    }
    if (MODE == MODE_DUMP) {
        for (int i = tid; i < TILE_W * TILE_H; i += NTHREADS) {
            int row = row0 + i / TILE_W, col = col0 + i % TILE_W;
            if (row < fp.H && col < fp.W) ra.key_out[(size_t)row * fp.W + col] = tile[i];
        }
        return;
    }
    if (MODE == MODE_COVER) {
        for (int i = tid; i < TILE_W * TILE_H; i += NTHREADS) {
            int row = row0 + i / TILE_W, col = col0 + i % TILE_W;
            if (row < fp.H && col < fp.W && tile[i] != KEY_EMPTY) ra.cover[(size_t)row * fp.W + col] = 1;
        }
        return;
    }
    if (ROPE_SKIP(fp, 16)) return;
    score_tile<LOSS, true>(tile, layer_tile, row0, col0, fp, n_render, tq, t32, tl, lds_sums, rc);
    __syncthreads();
    if (tid < ROPE_SUM_WORDS) {
        uint64_t delta = lds_sums[tid];
        if (layer_tile) delta += ra.layer_sums[((size_t)ra.layer_of[cand] * (fp.tiles_x * fp.tiles_y) + tile_id) * ROPE_SUM_WORDS + tid];
        if (delta) atomicAdd((unsigned long long *)&ra.sums[(size_t)cand * ROPE_SUM_WORDS + tid], (unsigned long long)delta);
    }
}

// MODE_SCORE: reduce the loss and add (actual - empty) into the candidate's sums.
// MODE_DUMP : write the tile's keys to a full-frame key image (single-pose render).
// MODE_COVER: set cover[pixel] = 1 where anything was drawn (crop search).
template <int LOSS, int MODE, bool CLIP>
__global__ void __launch_bounds__(NTHREADS, CLIP ? ROPE_MIN_WAVES_CLIP : ((LOSS == ROPE_LOSS_FULL || LOSS == ROPE_LOSS_CAMFULL) ? ROPE_MIN_WAVES_FULL : ROPE_MIN_WAVES_PER_SIMD))
raster_score_kernel(FrameParams fp, RobotParams rp, RasterArgs ra)
{
    // Workgroups go to the 8 XCDs round-robin by linear id, i.e. by (row * n_tiles + blockIdx.x) mod 8: with an even
    // tile count a given screen tile would only ever meet 4, 2 or 1 of them, and the few tiles that hold the robot
    // would pile up there.  Rotating the tile index by a per-row hash spreads every tile over all XCDs.
    const int n_tiles_all = fp.tiles_x * fp.tiles_y;
    const int tile_id = (int)((blockIdx.x + ((blockIdx.y * 0x9E3779B1u) >> 12)) % (unsigned)n_tiles_all);
    constexpr bool SHARES = (MODE == MODE_SPLIT || MODE == MODE_SPLIT_GEO);
    raster_tile<LOSS, MODE, CLIP>(fp, rp, ra, (int)blockIdx.y, tile_id, SHARES ? (int)blockIdx.z : 0, SHARES ? (int)gridDim.z : 1);
}

// Large batches: the (candidate, tile) pairs that have anything to draw, taken from a queue by a grid that just fills the
// chip (two workgroups per CU) — of the tiles x candidates pairs of a pass about four in five have nothing to do, and a
// launch over all of them spends a fifth of a millisecond starting workgroups that leave at once.  score_queue_kernel
// builds the queue.  A workgroup asks for its next pair while it works on the current one.
#ifdef ROPE_PROFILE
// Profiling build only: per workgroup of the last raster_queue_kernel<.., MODE_SCORE> launch, s_memtime (shader cycles) and
// s_memrealtime (constant 100 MHz) at its start and end — the clock the chip held during THIS kernel (rope_debug_clock;
// MI355X_MICROARCH.md "DVFS give-back" item 6).  A buffer of its own: nothing else reads it.
__device__ unsigned long long g_clock_stamps[4 * 1024];
#endif

template <int LOSS, int MODE, bool CLIP>
__global__ void __launch_bounds__(NTHREADS, CLIP ? ROPE_MIN_WAVES_CLIP : ((LOSS == ROPE_LOSS_FULL || LOSS == ROPE_LOSS_CAMFULL) ? ROPE_MIN_WAVES_FULL : ROPE_MIN_WAVES_PER_SIMD))
raster_queue_kernel(FrameParams fp, RobotParams rp, RasterArgs ra, const uint32_t *__restrict__ items, size_t segment,
                    int *__restrict__ counters /* [1] next ticket, [2 + k] pairs queued in class k */)
{
    __shared__ int s_item;
#ifdef ROPE_PROFILE
    if (MODE == MODE_SCORE && threadIdx.x == 0 && blockIdx.x < 1024) {
        g_clock_stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime();
        g_clock_stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (threadIdx.x == 0) s_item = atomicAdd(&counters[1], 1);
    __syncthreads();
    // the queue is QUEUE_CLASSES segments of `segment` entries, heaviest pairs first (score_queue_kernel): ticket -> (class, place)
    int cls_n[QUEUE_CLASSES], n_items = 0;
#pragma unroll
    for (int k = 0; k < QUEUE_CLASSES; k++) { cls_n[k] = counters[2 + k]; n_items += cls_n[k]; }
    for (;;) {
        const int item = s_item;
        if (item >= n_items) break;                        // every wave of the workgroup reads the same value: all leave together
        __syncthreads();
        int next = 0;
        if (threadIdx.x == 0) next = atomicAdd(&counters[1], 1);      // in flight while this pair is drawn
        int place = item, cls = 0;
#pragma unroll
        for (int k = 0; k < QUEUE_CLASSES - 1; k++)
            if (cls == k && place >= cls_n[k]) { place -= cls_n[k]; cls = k + 1; }
        if (segment) ROPE_CHECK_INDEX(place, segment);
        const uint32_t it = items[(size_t)cls * segment + place];
        raster_tile<LOSS, MODE, CLIP>(fp, rp, ra, (int)(it & 0xFFFFu), (int)(it >> 16), 0, 1);      // row = candidate (MODE_SCORE) or layer (MODE_LAYER)
        __syncthreads();                                   // the tile's LDS is free again
        if (threadIdx.x == 0) s_item = next;
        __syncthreads();
    }
#ifdef ROPE_PROFILE
    if (MODE == MODE_SCORE && threadIdx.x == 0 && blockIdx.x < 1024) {
        g_clock_stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
        g_clock_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef ROPE_PROFILE
hipError_t read_clock_stamps(unsigned long long *out /* 4 x 1024 */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock_stamps), sizeof(g_clock_stamps));
}
#endif

// The queue of raster_queue_kernel: 32 threads per (candidate, word of its tile masks), eight candidates per workgroup.
// Pairs whose tile the candidate's own links reach (or, without shared layers, any of its links) are queued — one atomic
// on the queue's counter per workgroup: thousands of them on one address are served one after the other (40 us for the
// 4096 candidates of the bench with one atomic each); where only the shared layer reaches, the candidate takes the
// layer's stored loss sums here and now.
__global__ void __launch_bounds__(256)
score_queue_kernel(RasterArgs ra, int n_rows, int n_tiles, uint32_t *__restrict__ items, size_t segment, int *__restrict__ counters,
                   const uint32_t *__restrict__ tile_tris)
{
    __shared__ int s_n[QUEUE_CLASSES], s_base[QUEUE_CLASSES];
    const int k = threadIdx.x & 31, slot = threadIdx.x >> 5, cand = 8 * blockIdx.x + slot, w = blockIdx.y;
    // rows of a shared-layer launch (ra.cand_of_row set): the pairs its representative candidate's shared links reach, weighed by
    // those links' triangles (tile_tris is the shared links' array then)
    const bool live = cand < n_rows, for_layers = ra.cand_of_row != nullptr, layers = !for_layers && ra.layer_of != nullptr;
    const int rep = (live && for_layers) ? ra.cand_of_row[cand] : cand;
    const uint32_t hi = (live && !for_layers) ? ra.mask_hi[(size_t)cand * ra.mask_words + w] : 0u;
    const uint32_t lo = live ? ra.mask_lo[(size_t)(layers ? ra.layer_rep[ra.layer_of[cand]] : rep) * ra.mask_words + w] : 0u;
    const uint32_t work = for_layers ? lo : (layers ? hi : (hi | lo));
    if (threadIdx.x < QUEUE_CLASSES) s_n[threadIdx.x] = 0;
    __syncthreads();
    // The heaviest pairs go first: a pair drawing 20 000 triangles takes ten times the average, and one that starts when the
    // queue is nearly empty keeps its workgroup busy long after the others have left (7 % of the launch was that tail).
    // Weight = triangles of the meshlets whose boxes touch the tile (bounds_kernel), in classes of a factor two.
    const bool mine = (work >> k) & 1u;
    int cls = QUEUE_CLASSES - 1, pos = 0;
    if (mine) {
        if (tile_tris) {
            const uint32_t tris = tile_tris[(size_t)rep * n_tiles + 32 * w + k];
            cls = min(max(QUEUE_TOP_LOG2 - (31 - __clz((int)(tris | 1u))), 0), QUEUE_CLASSES - 1);     // >= 2^TOP: class 0, a factor two per class
        }
        pos = atomicAdd(&s_n[cls], 1);
    }
    __syncthreads();
    if (threadIdx.x < QUEUE_CLASSES) s_base[threadIdx.x] = s_n[threadIdx.x] ? atomicAdd(&counters[2 + threadIdx.x], s_n[threadIdx.x]) : 0;
    __syncthreads();
    if (mine && segment) ROPE_CHECK_INDEX(s_base[cls] + pos, segment);
    if (mine) items[(size_t)cls * segment + s_base[cls] + pos] = (uint32_t)cand | ((uint32_t)(32 * w + k) << 16);
    if (layers && live && k < ROPE_SUM_WORDS) {
        uint64_t acc = 0;
        for (uint32_t only = lo & ~hi; only; only &= only - 1) {
            const int tile = 32 * w + __ffs((int)only) - 1;
            acc += ra.layer_sums[((size_t)ra.layer_of[cand] * n_tiles + tile) * ROPE_SUM_WORDS + k];
        }
        if (acc) atomicAdd((unsigned long long *)&ra.sums[(size_t)cand * ROPE_SUM_WORDS + k], (unsigned long long)acc);
    }
}

// Small batches, second half: MODE_SPLIT left every (candidate, tile) image merged in global memory.  Grid =
// (tiles, candidates, row slices): a workgroup scores its slice of rows straight from there (delta against
// "nothing rendered"), adds the sums to the candidate's, and puts the slice back to "empty" — the buffer is clean
// again when the pass ends, so no clearing launch is needed before the next one.
template <int LOSS>
__global__ void __launch_bounds__(256)
score_gtile_kernel(FrameParams fp, RasterArgs ra)
{
    __shared__ uint64_t lds_sums[ROPE_SUM_WORDS];
    const int tid = threadIdx.x, tile_id = blockIdx.x, cand = blockIdx.y;
    const size_t mw = (size_t)cand * ra.mask_words + (tile_id >> 5);
    if (ra.touched) {                                                                  // MODE_SPLIT_GEO stamps the tiles it drew into
        if (ra.touched[(size_t)cand * (fp.tiles_x * fp.tiles_y) + tile_id] != ra.pass_id) return;
    } else if (!(((ra.mask_lo[mw] | ra.mask_hi[mw]) >> (tile_id & 31)) & 1u)) return;      // MODE_SPLIT left this tile alone
    const size_t frame = ra.frame_of ? (size_t)ra.frame_of[cand] : 0, plane = (size_t)fp.W * fp.H;
    const uint64_t *__restrict__ tq = ra.tq ? ra.tq + frame * plane : nullptr;
    const float *__restrict__ t32 = ra.t32 ? ra.t32 + frame * plane : nullptr;
    const uint64_t *__restrict__ tl = ra.tl ? ra.tl + frame * plane * ROPE_MAX_LINKS : nullptr;
    if (tid < ROPE_SUM_WORDS) lds_sums[tid] = 0;
    __syncthreads();
    const int rows = TILE_H / (int)gridDim.z;
    const TileRect rc = {(int)blockIdx.z * rows, (int)blockIdx.z * rows + rows - 1, 0, TILE_W / 4 - 1, 4};   // 16 groups x 4 rows (bands are multiples of 4 rows)
    uint32_t *g = ra.gtile + ((size_t)cand * (fp.tiles_x * fp.tiles_y) + tile_id) * (TILE_W * TILE_H);
    const int tx = tile_id % fp.tiles_x, ty = tile_id / fp.tiles_x;
    if (!ROPE_SKIP(fp, 16)) score_tile<LOSS, true>(g, nullptr, ty * TILE_H, tx * TILE_W, fp, ra.n_render, tq, t32, tl, lds_sums, rc);
    __syncthreads();
    uint4 *g4 = reinterpret_cast<uint4 *>(g) + rc.r_lo * (TILE_W / 4);
    for (int i = tid; i < rows * (TILE_W / 4); i += blockDim.x) {
        const uint4 k = g4[i];
        if ((k.x & k.y & k.z & k.w) != KEY_EMPTY) g4[i] = make_uint4(KEY_EMPTY, KEY_EMPTY, KEY_EMPTY, KEY_EMPTY);
    }
    if (tid < ROPE_SUM_WORDS && lds_sums[tid])
        atomicAdd((unsigned long long *)&ra.sums[(size_t)cand * ROPE_SUM_WORDS + tid], (unsigned long long)lds_sums[tid]);
}

// ------------------------------------------------------------- lookup table -----
// Lookup stage against a stored table (predict.py:165-171): per row k of the table, the exact sums of |T - sqrtD_k| in Q32
// over the crop.  The crop is the box of every pose of the grid; one pose covers a fraction of it, and where a row holds
// nothing the term is |T - 0|, the same for every row.  So a row is stored as the groups of samples that hold something
// (table_count_kernel / table_fill_kernel), the sums of |T| over the whole crop are taken once per frame (crop_total_kernel), and
// a row's sums are total + sum over its groups of (|T - D| - |T|) — integers, so exactly the sums over the whole crop.

// One workgroup: the crop of the target plane as one contiguous array — rows padded with zeros to a multiple of four samples, so
// that a group of the table (columns 4k .. 4k+3 of a crop row) is one aligned float4 at offset 4 x (its number) — and the sums of |T|
// over it.  (The same sums group by group, for a table row to subtract instead of working them out again per (row, frame), made the
// scoring kernel half as slow again, 9.8 against 6.6 ms for 256 frames x 15 625 rows: it waits for its gathers, not for its arithmetic.)
// grid = frames: frame f's plane starts f planes into t32, its crop f crops into t32c, its totals f x ROPE_SUM_WORDS into total
__global__ void __launch_bounds__(1024)
crop_total_kernel(FrameParams fp, const float *__restrict__ t32, float *__restrict__ t32c, uint64_t *__restrict__ total /* ROPE_SUM_WORDS */)
{
    __shared__ uint64_t lds[4];
    const int cw = fp.c1 - fp.c0 + 1, ch = fp.r1 - fp.r0 + 1, gw = (cw + 3) >> 2, n_groups = gw * ch;
    t32 += (size_t)blockIdx.x * fp.W * fp.H;
    t32c += (size_t)blockIdx.x * n_groups * 4;
    total += (size_t)blockIdx.x * ROPE_SUM_WORDS;
    if (threadIdx.x < 4) lds[threadIdx.x] = 0;
    __syncthreads();
    uint64_t s[ROPE_SUM_WORDS];
    s[SUM_S1] = s[SUM_AA] = s[SUM_AB] = s[SUM_BB] = 0;
    for (int g = threadIdx.x; g < n_groups; g += blockDim.x) {
        const int r = g / gw, c = 4 * (g - r * gw);
        const float *src = t32 + (size_t)(fp.r0 + r) * fp.W + fp.c0 + c;
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            t[j] = c + j < cw ? src[j] : 0.0f;
            acc_sq<false>(s, q32_of_f32(fabsf(t[j] - 0.0f)));
        }
        reinterpret_cast<float4 *>(t32c)[g] = make_float4(t[0], t[1], t[2], t[3]);
    }
    const int words[4] = {SUM_S1, SUM_AA, SUM_AB, SUM_BB};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t v = s[words[k]];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd((unsigned long long *)&lds[k], (unsigned long long)v);
    }
    __syncthreads();
    if (threadIdx.x < ROPE_SUM_WORDS) {
        uint64_t v = 0;
        for (int k = 0; k < 4; k++) if ((int)threadIdx.x == words[k]) v = lds[k];
        total[threadIdx.x] = v;
    }
}

// Build time.  A dense row (the cropped sqrt-depth image of one grid pose) is mostly zeros: the crop is the box of every pose of the
// grid together, one pose covers a fraction of it, and the robot a fraction of its own bounding box.  A sample that holds nothing
// contributes |T - 0| - |T| = 0 to the row's sums, exactly — so the stored table keeps only the GROUPS of four consecutive samples
// of a crop row (columns 4k .. 4k+3) in which anything was drawn: per group four times its number — the offset of its first
// sample in a crop whose rows are padded to whole groups (crop_total_kernel's layout) — and its four values (columns past the
// crop's edge read as 0).  table_count_kernel counts a row's groups and reserves their place,
// table_fill_kernel writes them (in the order of their place in the crop: neighbouring lanes then read neighbouring samples).
__global__ void __launch_bounds__(256)
table_count_kernel(int cw, int ch, const float *__restrict__ table, uint32_t *__restrict__ counts, unsigned long long *__restrict__ offs,
                   unsigned long long *__restrict__ used)
{
    __shared__ int s_n;
    const int gw = (cw + 3) >> 2, n_groups = gw * ch;
    const float *row = table + (size_t)blockIdx.x * cw * ch;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    int mine = 0;
    for (int g = threadIdx.x; g < n_groups; g += blockDim.x) {
        const int r = g / gw, c = 4 * (g - r * gw);
        bool any = false;
        for (int j = 0; j < 4; j++) any = any || (c + j < cw && row[(size_t)r * cw + c + j] != 0.0f);
        mine += any;
    }
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[blockIdx.x] = (uint32_t)s_n;
        offs[blockIdx.x] = s_n ? atomicAdd(used, (unsigned long long)s_n) : 0ull;
    }
}

__global__ void __launch_bounds__(256)
table_fill_kernel(int cw, int ch, const float *__restrict__ table, const unsigned long long *__restrict__ offs, uint32_t *__restrict__ goff,
                  float4 *__restrict__ gval)
{
    // The groups of a row are written IN ORDER of their place in the crop: the scoring kernels hand consecutive groups to
    // consecutive lanes, and a run of groups inside the robot's outline is then a run of neighbouring 16-byte pieces of the target —
    // eight lanes to a cache line instead of one line per lane.
    __shared__ int s_wave[4];
    const int gw = (cw + 3) >> 2, n_groups = gw * ch, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *row = table + (size_t)blockIdx.x * cw * ch;
    const unsigned long long base = offs[blockIdx.x];
    int running = 0;
    for (int g0 = 0; g0 < n_groups; g0 += 256) {
        const int g = g0 + (int)threadIdx.x;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        bool any = false;
        if (g < n_groups) {
            const int r = g / gw, c = 4 * (g - r * gw);
            for (int j = 0; j < 4; j++) { v[j] = c + j < cw ? row[(size_t)r * cw + c + j] : 0.0f; any = any || v[j] != 0.0f; }
        }
        const unsigned long long b = __ballot(any);
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int before = 0, all = 0;
        for (int w = 0; w < 4; w++) { before += w < wave ? s_wave[w] : 0; all += s_wave[w]; }
        if (any) {
            const int pos = running + before + __popcll(b & ((1ull << lane) - 1ull));
            goff[base + pos] = (uint32_t)(4 * g);
            gval[base + pos] = make_float4(v[0], v[1], v[2], v[3]);
        }
        running += all;
        __syncthreads();
    }
}

// One group against one cropped target: s += sum of |T - D| minus sum of |T| over its four samples (words S1, AA, AB, BB), modulo 2^64.
__device__ static inline void table_group_delta(const float4 t, const float4 d, uint64_t *s)
{
    acc_sq<false>(s, q32_of_f32(fabsf(t.x - d.x))); acc_sq<true>(s, q32_of_f32(fabsf(t.x - 0.0f)));
    acc_sq<false>(s, q32_of_f32(fabsf(t.y - d.y))); acc_sq<true>(s, q32_of_f32(fabsf(t.y - 0.0f)));
    acc_sq<false>(s, q32_of_f32(fabsf(t.z - d.z))); acc_sq<true>(s, q32_of_f32(fabsf(t.z - 0.0f)));
    acc_sq<false>(s, q32_of_f32(fabsf(t.w - d.w))); acc_sq<true>(s, q32_of_f32(fabsf(t.w - 0.0f)));
}

__global__ void __launch_bounds__(256)
table_score_kernel(const uint32_t *__restrict__ counts, const unsigned long long *__restrict__ offs, const uint32_t *__restrict__ goff,
                   const float4 *__restrict__ gval, const float *__restrict__ t32c, const uint64_t *__restrict__ total, uint64_t *__restrict__ sums)
{
    __shared__ uint64_t lds[4];
    if (threadIdx.x < 4) lds[threadIdx.x] = 0;
    __syncthreads();
    const int n = (int)counts[blockIdx.x];
    const unsigned long long base = offs[blockIdx.x];
    uint64_t s[ROPE_SUM_WORDS];
    s[SUM_S1] = s[SUM_AA] = s[SUM_AB] = s[SUM_BB] = 0;
    for (int g = threadIdx.x; g < n; g += blockDim.x) table_group_delta(*reinterpret_cast<const float4 *>(t32c + goff[base + g]), gval[base + g], s);
    const int words[4] = {SUM_S1, SUM_AA, SUM_AB, SUM_BB};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t v = s[words[k]];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd((unsigned long long *)&lds[k], (unsigned long long)v);
    }
    __syncthreads();
    if (threadIdx.x < ROPE_SUM_WORDS) {
        uint64_t v = 0;
        for (int k = 0; k < 4; k++) if ((int)threadIdx.x == words[k]) v = total[words[k]] + lds[k];
        sums[(size_t)blockIdx.x * ROPE_SUM_WORDS + threadIdx.x] = v;
    }
}

__device__ static inline double mean_std_parts(const uint64_t *s, double N, double &m1);

// The same for the targets of many frames (rope_lookup_score_targets): grid = (table rows, frame chunks), and inside a workgroup
// one group of LANES lanes per (row, TABLE_FRAMES frames) — a row's few hundred groups are a handful per lane, so this needs no
// barrier and no LDS: a lane reads a group of the row once and holds it against TABLE_FRAMES frames' samples at that place (their
// loads in flight together), the lanes' partial sums meet by lane exchange and the group's first lane writes the finished lookup
// scores (finalize_one's steps for ROPE_LOSS_LOOKUP on the same sums: same bits) to scores[frame x rows + row].  The kernel lives on its arithmetic
// (two Q32 conversions and six 64-bit multiply-adds per sample) and on the waves in flight: few frames per wave, see the launch.
template <int TABLE_FRAMES, int LANES /* of a wave that share one frame: 64, or 16 (four frames side by side in a wave) */>
__global__ void __launch_bounds__(256)
table_score_frames_kernel(int crop_px /* padded: 4 x groups */, const uint32_t *__restrict__ counts, const unsigned long long *__restrict__ offs,
                          const uint32_t *__restrict__ goff, const float4 *__restrict__ gval, const float *__restrict__ t32c /* frames x crop_px */,
                          const uint64_t *__restrict__ totals /* frames x ROPE_SUM_WORDS */, int n_frames, double n_pix,
                          double *__restrict__ scores)
{
    // With LANES = 16 a wave holds four frame groups side by side: a row's few hundred groups are then twenty per lane instead of
    // five with the last step a third empty, the lane exchange is four steps instead of six and serves four times the frames, and
    // the float64 epilogue of four frames runs in four lanes at once.
    constexpr int PARTS = 64 / LANES, PER_WAVE = PARTS * TABLE_FRAMES;
    const int n = (int)counts[blockIdx.x], lane = threadIdx.x & 63, wave = threadIdx.x >> 6, part = lane / LANES, sub = lane % LANES;
    const unsigned long long base = offs[blockIdx.x];
    const int per = (n_frames + (int)gridDim.y - 1) / (int)gridDim.y, f_lo = (int)blockIdx.y * per, f_hi = min(f_lo + per, n_frames);
    for (int fw = f_lo + wave * PER_WAVE; fw < f_hi; fw += 4 * PER_WAVE) {
        const int f0 = fw + part * TABLE_FRAMES;                 // this lane group's first frame
        const int nf = max(0, min(TABLE_FRAMES, f_hi - f0));
        uint64_t s[TABLE_FRAMES][SUM_BB + 1];
#pragma unroll
        for (int j = 0; j < TABLE_FRAMES; j++)
#pragma unroll
            for (int k = 0; k <= SUM_BB; k++) s[j][k] = 0;
        const float *__restrict__ t0 = t32c + (size_t)min(f0, n_frames - 1) * crop_px;
        if (nf > 0)
            for (int g = sub; g < n; g += LANES) {
                const uint32_t off = goff[base + g];
                const float4 d = gval[base + g];
                float4 t[TABLE_FRAMES];
#pragma unroll
                for (int j = 0; j < TABLE_FRAMES; j++)
                    t[j] = j < nf ? *reinterpret_cast<const float4 *>(t0 + (size_t)j * crop_px + off) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < TABLE_FRAMES; j++) table_group_delta(t[j], d, s[j]);
            }
#pragma unroll
        for (int j = 0; j < TABLE_FRAMES; j++) {
            uint64_t r[ROPE_SUM_WORDS];
#pragma unroll
            for (int k = SUM_S1; k <= SUM_BB; k++) {
                uint64_t v = s[j][k];
#pragma unroll
                for (int off = LANES / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);      // stays inside the lane group
                r[k] = v;
            }
            if (sub == 0 && j < nf) {
                const uint64_t *total = totals + (size_t)(f0 + j) * ROPE_SUM_WORDS;
#pragma unroll
                for (int k = SUM_S1; k <= SUM_BB; k++) r[k] += total[k];
                double m1;
                const double sd = mean_std_parts(r, n_pix, m1);
                scores[(size_t)(f0 + j) * gridDim.x + blockIdx.x] = m1 * sd;
            }
        }
    }
}

// ---------------------------------------------------------------- finalize -----
__device__ static inline double mean_std_parts(const uint64_t *s, double N, double &m1)
{
    m1 = ((double)s[SUM_S1] * 0x1p-32) / N;
    double S2 = ((double)s[SUM_AA] * 0x1p40 + (double)s[SUM_AB] * 0x1p21) + (double)s[SUM_BB];
    double m2 = (S2 * 0x1p-64) / N;
    double var = m2 - m1 * m1;
    if (var < 0.0) var = 0.0;
    return sqrt(var);
}

// sums -> float64 error of one candidate (the reference's order of operations, predict.py:480-509)
__device__ static inline double finalize_one(uint64_t *__restrict__ sums, const uint64_t *__restrict__ total_empty, int c, int loss,
                                             int n_render, double n_pix, const LinkFlags &lf)
{
    uint64_t s[ROPE_SUM_WORDS];
#pragma unroll
    for (int k = 0; k < ROPE_SUM_WORDS; k++) {
        // words of links that were not rendered (or of a loss without link terms) read as zero
        const bool live = k < SUM_LINK0 || (loss == ROPE_LOSS_FULL && k < SUM_LINK0 + 3 * n_render);
        s[k] = live ? sums[(size_t)c * ROPE_SUM_WORDS + k] + total_empty[k] : 0;
        sums[(size_t)c * ROPE_SUM_WORDS + k] = s[k];
    }
    double m1, sd = mean_std_parts(s, n_pix, m1);
    if (loss == ROPE_LOSS_LOOKUP) return m1 * sd;
    if (loss == ROPE_LOSS_TSWEEP) return m1 * -sd;
    double e = 0.0;
    if (loss == ROPE_LOSS_FULL)
        for (int l = 1; l < n_render; l++) {
            if (!(lf.f[l] & 1)) continue;
            e += ((double)s[SUM_LINK0 + 3 * l] / n_pix) * 5.0;
            if ((lf.f[l] & 2) && s[SUM_LINK0 + 3 * l + 1] > 0)
                e += (((double)s[SUM_LINK0 + 3 * l + 2] * 0x1p-32) / (double)s[SUM_LINK0 + 3 * l + 1]) * 10.0;
        }
    double meanD = ((double)s[SUM_S1] * 0x1p-32) / (double)s[SUM_CNT];
    e += meanD * sd;
    return e;
}

// One block: every candidate's error, then the first index of the smallest one by wave shuffles
// (NaN never wins against a number).  err[C] and err[C+1] receive the best error and its index, so that
// one device-to-host copy returns everything.
// large batches: the errors come from a grid of blocks, the single argmin block then only reduces them
__global__ void __launch_bounds__(256)
finalize_only_kernel(uint64_t *__restrict__ sums, const uint64_t *__restrict__ total_empty, int C, int loss, int n_render,
                     double n_pix, LinkFlags lf, double *__restrict__ err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) err[i] = finalize_one(sums, total_empty, i, loss, n_render, n_pix, lf);
}

// batches over several frames' targets: every row adds its own frame's "nothing rendered" totals and reads its frame's link flags
__global__ void __launch_bounds__(256)
finalize_frames_kernel(uint64_t *__restrict__ sums, const uint64_t *__restrict__ totals /* frames x SUM_WORDS */, const int32_t *__restrict__ frame_of,
                       const LinkFlags *__restrict__ flags /* per frame */, int C, int loss, int n_render, double n_pix, double *__restrict__ err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const int f = frame_of[i];
    err[i] = finalize_one(sums, totals + (size_t)f * ROPE_SUM_WORDS, i, loss, n_render, n_pix, flags[f]);
}

// first index of the smallest of err[0..C) (a NaN never beats a number; all NaN: index 0), one workgroup per set of C values:
// best[2 set] = the error, best[2 set + 1] = the index — finalize_argmin_kernel's rule
__global__ void __launch_bounds__(1024)
argmin_sets_kernel(const double *__restrict__ err_all, int C, double *__restrict__ best)
{
    __shared__ double s_e[16];
    __shared__ int s_i[16];
    const double *err = err_all + (size_t)blockIdx.x * C;
    double be = __builtin_inf();
    int bi = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        const double e = err[i];
        if (e < be || (e == be && i < bi) || (bi == 0x7FFFFFFF && !(e != e))) { be = e; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double oe = __shfl_xor(be, off, 64);
        int oi = __shfl_xor(bi, off, 64);
        if (oe < be || (oe == be && oi < bi)) { be = oe; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_e[wave] = be; s_i[wave] = bi; }
    __syncthreads();
    if (wave == 0) {
        const int nw = blockDim.x >> 6;
        be = lane < nw ? s_e[lane] : __builtin_inf();
        bi = lane < nw ? s_i[lane] : 0x7FFFFFFF;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            double oe = __shfl_xor(be, off, 64);
            int oi = __shfl_xor(bi, off, 64);
            if (oe < be || (oe == be && oi < bi)) { be = oe; bi = oi; }
        }
        if (lane == 0) {
            best[2 * (size_t)blockIdx.x] = (bi == 0x7FFFFFFF) ? err[0] : be;
            best[2 * (size_t)blockIdx.x + 1] = (bi == 0x7FFFFFFF) ? 0.0 : (double)bi;
        }
    }
}

__global__ void __launch_bounds__(1024)
finalize_argmin_kernel(uint64_t *__restrict__ sums, const uint64_t *__restrict__ total_empty, int C, int loss, int n_render,
                       double n_pix, LinkFlags lf, double *__restrict__ err, int have_err)
{
    __shared__ double s_e[16];
    __shared__ int s_i[16];
    double be = __builtin_inf();
    int bi = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        const double e = have_err ? err[i] : finalize_one(sums, total_empty, i, loss, n_render, n_pix, lf);
        if (!have_err) err[i] = e;
        if (e < be || (e == be && i < bi) || (bi == 0x7FFFFFFF && !(e != e))) { be = e; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double oe = __shfl_xor(be, off, 64);
        int oi = __shfl_xor(bi, off, 64);
        if (oe < be || (oe == be && oi < bi)) { be = oe; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_e[wave] = be; s_i[wave] = bi; }
    __syncthreads();
    if (wave == 0) {
        const int nw = blockDim.x >> 6;
        be = lane < nw ? s_e[lane] : __builtin_inf();
        bi = lane < nw ? s_i[lane] : 0x7FFFFFFF;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            double oe = __shfl_xor(be, off, 64);
            int oi = __shfl_xor(bi, off, 64);
            if (oe < be || (oe == be && oi < bi)) { be = oe; bi = oi; }
        }
        if (lane == 0) {
            // all-NaN input: index 0 and its (NaN) error, as the first element
            err[C] = (bi == 0x7FFFFFFF) ? err[0] : be;
            err[C + 1] = (bi == 0x7FFFFFFF) ? 0.0 : (double)bi;
        }
    }
}

// key image -> metric depth + link id (single-pose render path)
__global__ void resolve_kernel(const uint32_t *__restrict__ key, int n, float c_num, float c_sum, float c_dif,
                               float *__restrict__ depth, uint8_t *__restrict__ ids)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k = key[i];
    depth[i] = (k == KEY_EMPTY) ? 0.0f : linear_depth(k >> 8, c_num, c_sum, c_dif);
    ids[i] = (k == KEY_EMPTY) ? 255 : (uint8_t)(k & 0xFF);
}

// ------------------------------------------------------------ launch helpers ---
template <int LOSS, int MODE>
static void launch_one(dim3 grid, hipStream_t st, const FrameParams &fp, const RobotParams &rp, const RasterArgs &a, bool clip)
{
    if (clip) hipLaunchKernelGGL((raster_score_kernel<LOSS, MODE, true>), grid, dim3(NTHREADS), 0, st, fp, rp, a);
    else hipLaunchKernelGGL((raster_score_kernel<LOSS, MODE, false>), grid, dim3(NTHREADS), 0, st, fp, rp, a);
}

template <int LOSS, int MODE>
static void launch_queue_one(dim3 grid, hipStream_t st, const FrameParams &fp, const RobotParams &rp, const RasterArgs &a, uint32_t *items, size_t segment, int *counters, bool clip)
{
    if (clip) hipLaunchKernelGGL((raster_queue_kernel<LOSS, MODE, true>), grid, dim3(NTHREADS), 0, st, fp, rp, a, items, segment, counters);
    else hipLaunchKernelGGL((raster_queue_kernel<LOSS, MODE, false>), grid, dim3(NTHREADS), 0, st, fp, rp, a, items, segment, counters);
}

hipError_t launch_fk(hipStream_t st, const double *cand, int C, int n_render, const double *joint_fixed,
                     const double *joint_axes, const double *PV, const int32_t *view_of, float *mvp, uint64_t *sums,
                     uint32_t *mask_lo, uint32_t *mask_hi, int mask_words, int *queue_counters, uint32_t *tile_tris, uint32_t *tile_tris_lo,
                     int n_tiles)
{
    hipLaunchKernelGGL(fk_mvp_kernel, dim3((C + 255) / 256), dim3(256), 0, st, cand, C, n_render, joint_fixed, joint_axes, PV, view_of,
                       mvp, sums, mask_lo, mask_hi, mask_words, queue_counters, tile_tris, tile_tris_lo, n_tiles);
    return hipGetLastError();
}

hipError_t launch_bounds(hipStream_t st, int C, const FrameParams &fp, const RobotParams &rp, int n_render, int n_shared,
                         const float *mvp, short4 *bounds, uint32_t *mask_lo, uint32_t *mask_hi, int mask_words,
                         const int32_t *layer_of, const int32_t *layer_rep, uint32_t *tile_tris, uint32_t *tile_tris_lo, int lo_first)
{
    // the masks (and the tile weights) were cleared by fk_mvp_kernel earlier in the same pass
    hipLaunchKernelGGL(bounds_kernel, dim3((rp.n_meshlets + 255) / 256, C), dim3(256), 0, st, fp, rp, n_render, n_shared, mvp,
                       bounds, mask_lo, mask_hi, mask_words, layer_of, layer_rep, tile_tris, tile_tris_lo, lo_first);
    return hipGetLastError();
}

hipError_t launch_fk_bounds(hipStream_t st, const double *cand, int C, const FrameParams &fp, const RobotParams &rp, int n_render,
                            int n_shared, const double *joint_fixed, const double *joint_axes, const double *PV,
                            const int32_t *view_of, float *mvp, short4 *bounds, uint64_t *sums, uint32_t *mask_lo,
                            uint32_t *mask_hi, int mask_words)
{
    hipLaunchKernelGGL(fk_bounds_kernel, dim3(C), dim3(1024), 0, st, fp, rp, cand, n_render, n_shared, joint_fixed, joint_axes, PV,
                       view_of, mvp, bounds, sums, mask_lo, mask_hi, mask_words);
    return hipGetLastError();
}

hipError_t launch_raster(int mode, int loss, int rows, hipStream_t st, const FrameParams &fp, const RobotParams &rp,
                         const RasterArgs &a, bool clip)
{
    dim3 grid(fp.tiles_x * fp.tiles_y, rows, (mode == MODE_SPLIT || mode == MODE_SPLIT_GEO) ? a.split : 1);
    if (mode == MODE_DUMP) launch_one<ROPE_LOSS_DEPTH, MODE_DUMP>(grid, st, fp, rp, a, clip);
    else if (mode == MODE_COVER) launch_one<ROPE_LOSS_DEPTH, MODE_COVER>(grid, st, fp, rp, a, clip);
    else if (mode == MODE_TABLE) launch_one<ROPE_LOSS_LOOKUP, MODE_TABLE>(grid, st, fp, rp, a, clip);
    else if (mode == MODE_SPLIT) launch_one<ROPE_LOSS_DEPTH, MODE_SPLIT>(grid, st, fp, rp, a, clip);
    else if (mode == MODE_SPLIT_GEO) launch_one<ROPE_LOSS_DEPTH, MODE_SPLIT_GEO>(grid, st, fp, rp, a, clip);
    else if (mode == MODE_LAYER) {
        if (loss == ROPE_LOSS_DEPTH) launch_one<ROPE_LOSS_DEPTH, MODE_LAYER>(grid, st, fp, rp, a, clip);
        else if (loss == ROPE_LOSS_FULL) launch_one<ROPE_LOSS_FULL, MODE_LAYER>(grid, st, fp, rp, a, clip);
        else if (loss == ROPE_LOSS_LOOKUP) launch_one<ROPE_LOSS_LOOKUP, MODE_LAYER>(grid, st, fp, rp, a, clip);
        else launch_one<ROPE_LOSS_TSWEEP, MODE_LAYER>(grid, st, fp, rp, a, clip);
    }
    else if (loss == ROPE_LOSS_DEPTH) launch_one<ROPE_LOSS_DEPTH, MODE_SCORE>(grid, st, fp, rp, a, clip);
    else if (loss == ROPE_LOSS_FULL) launch_one<ROPE_LOSS_FULL, MODE_SCORE>(grid, st, fp, rp, a, clip);
    else if (loss == ROPE_LOSS_LOOKUP) launch_one<ROPE_LOSS_LOOKUP, MODE_SCORE>(grid, st, fp, rp, a, clip);
    else if (loss == ROPE_LOSS_CAMFULL) launch_one<ROPE_LOSS_CAMFULL, MODE_SCORE>(grid, st, fp, rp, a, clip);
    else launch_one<ROPE_LOSS_TSWEEP, MODE_SCORE>(grid, st, fp, rp, a, clip);
    return hipGetLastError();
}

hipError_t launch_raster_queue(int loss, int rows, int workgroups, hipStream_t st, const FrameParams &fp, const RobotParams &rp,
                               const RasterArgs &a, uint32_t *items, size_t segment, int *counters, const uint32_t *tile_tris, bool clip)
{
    const int n_tiles = fp.tiles_x * fp.tiles_y;
    hipLaunchKernelGGL(score_queue_kernel, dim3((rows + 7) / 8, a.mask_words), dim3(256), 0, st, a, rows, n_tiles, items, segment, counters, tile_tris);
    const dim3 grid(workgroups);
    switch (loss) {
    case ROPE_LOSS_DEPTH: launch_queue_one<ROPE_LOSS_DEPTH, MODE_SCORE>(grid, st, fp, rp, a, items, segment, counters, clip); break;
    case ROPE_LOSS_FULL: launch_queue_one<ROPE_LOSS_FULL, MODE_SCORE>(grid, st, fp, rp, a, items, segment, counters, clip); break;
    case ROPE_LOSS_LOOKUP: launch_queue_one<ROPE_LOSS_LOOKUP, MODE_SCORE>(grid, st, fp, rp, a, items, segment, counters, clip); break;
    case ROPE_LOSS_CAMFULL: launch_queue_one<ROPE_LOSS_CAMFULL, MODE_SCORE>(grid, st, fp, rp, a, items, segment, counters, clip); break;
    default: launch_queue_one<ROPE_LOSS_TSWEEP, MODE_SCORE>(grid, st, fp, rp, a, items, segment, counters, clip); break;
    }
    return hipGetLastError();
}

hipError_t launch_layer_queue(int loss, int rows, int workgroups, hipStream_t st, const FrameParams &fp, const RobotParams &rp,
                              const RasterArgs &a, uint32_t *items, size_t segment, int *counters, const uint32_t *tile_tris_lo, bool clip)
{
    const int n_tiles = fp.tiles_x * fp.tiles_y;
    hipLaunchKernelGGL(score_queue_kernel, dim3((rows + 7) / 8, a.mask_words), dim3(256), 0, st, a, rows, n_tiles, items, segment, counters, tile_tris_lo);
    const dim3 grid(workgroups);
    if (loss == ROPE_LOSS_DEPTH) launch_queue_one<ROPE_LOSS_DEPTH, MODE_LAYER>(grid, st, fp, rp, a, items, segment, counters, clip);
    else if (loss == ROPE_LOSS_FULL) launch_queue_one<ROPE_LOSS_FULL, MODE_LAYER>(grid, st, fp, rp, a, items, segment, counters, clip);
    else if (loss == ROPE_LOSS_LOOKUP) launch_queue_one<ROPE_LOSS_LOOKUP, MODE_LAYER>(grid, st, fp, rp, a, items, segment, counters, clip);
    else launch_queue_one<ROPE_LOSS_TSWEEP, MODE_LAYER>(grid, st, fp, rp, a, items, segment, counters, clip);
    return hipGetLastError();
}

hipError_t launch_score_gtile(int loss, int rows, int slices, hipStream_t st, const FrameParams &fp, const RasterArgs &a)
{
    dim3 grid(fp.tiles_x * fp.tiles_y, rows, slices);
    switch (loss) {
    case ROPE_LOSS_DEPTH: hipLaunchKernelGGL(score_gtile_kernel<ROPE_LOSS_DEPTH>, grid, dim3(256), 0, st, fp, a); break;
    case ROPE_LOSS_FULL: hipLaunchKernelGGL(score_gtile_kernel<ROPE_LOSS_FULL>, grid, dim3(256), 0, st, fp, a); break;
    case ROPE_LOSS_LOOKUP: hipLaunchKernelGGL(score_gtile_kernel<ROPE_LOSS_LOOKUP>, grid, dim3(256), 0, st, fp, a); break;
    case ROPE_LOSS_CAMFULL: hipLaunchKernelGGL(score_gtile_kernel<ROPE_LOSS_CAMFULL>, grid, dim3(256), 0, st, fp, a); break;
    default: hipLaunchKernelGGL(score_gtile_kernel<ROPE_LOSS_TSWEEP>, grid, dim3(256), 0, st, fp, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_empty(int loss, hipStream_t st, const FrameParams &fp, const uint64_t *tq, const float *t32, const uint64_t *tl,
                        uint64_t *empty_sums, uint64_t *total, int n_frames)
{
    dim3 grid(fp.tiles_x * fp.tiles_y, n_frames);
    switch (loss) {
    case ROPE_LOSS_DEPTH: hipLaunchKernelGGL(empty_tile_kernel<ROPE_LOSS_DEPTH>, grid, dim3(NTHREADS), 0, st, fp, tq, t32, tl, empty_sums); break;
    case ROPE_LOSS_FULL: hipLaunchKernelGGL(empty_tile_kernel<ROPE_LOSS_FULL>, grid, dim3(NTHREADS), 0, st, fp, tq, t32, tl, empty_sums); break;
    case ROPE_LOSS_LOOKUP: hipLaunchKernelGGL(empty_tile_kernel<ROPE_LOSS_LOOKUP>, grid, dim3(NTHREADS), 0, st, fp, tq, t32, tl, empty_sums); break;
    case ROPE_LOSS_CAMFULL: hipLaunchKernelGGL(empty_tile_kernel<ROPE_LOSS_CAMFULL>, grid, dim3(NTHREADS), 0, st, fp, tq, t32, tl, empty_sums); break;
    default: hipLaunchKernelGGL(empty_tile_kernel<ROPE_LOSS_TSWEEP>, grid, dim3(NTHREADS), 0, st, fp, tq, t32, tl, empty_sums); break;
    }
    hipLaunchKernelGGL(total_tiles_kernel, dim3(n_frames), dim3(64), 0, st, empty_sums, fp.tiles_x * fp.tiles_y, total);
    return hipGetLastError();
}

hipError_t launch_finalize(hipStream_t st, uint64_t *sums, const uint64_t *total_empty, int C, int loss, int n_render,
                           double n_pix, const LinkFlags &lf, double *err /* C + 2 doubles */)
{
    const int big = C > 2048;
    if (big) hipLaunchKernelGGL(finalize_only_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, total_empty, C, loss, n_render, n_pix, lf, err);
    hipLaunchKernelGGL(finalize_argmin_kernel, dim3(1), dim3(C <= 64 ? 64 : (C <= 256 ? 256 : 1024)), 0, st, sums, total_empty, C, loss,
                       n_render, n_pix, lf, err, big);
    return hipGetLastError();
}

hipError_t launch_finalize_frames(hipStream_t st, uint64_t *sums, const uint64_t *totals, const int32_t *frame_of, const LinkFlags *flags,
                                  int C, int loss, int n_render, double n_pix, double *err)
{
    hipLaunchKernelGGL(finalize_frames_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, totals, frame_of, flags, C, loss, n_render, n_pix, err);
    return hipGetLastError();
}

hipError_t launch_argmin_sets(hipStream_t st, const double *err, int C, int n_sets, double *best)
{
    hipLaunchKernelGGL(argmin_sets_kernel, dim3(n_sets), dim3(C <= 64 ? 64 : (C <= 256 ? 256 : 1024)), 0, st, err, C, best);
    return hipGetLastError();
}

hipError_t launch_table_score_frames(hipStream_t st, const FrameParams &fp, const uint32_t *counts, const unsigned long long *offs,
                                     const uint32_t *goff, const float4 *gval, int C, const float *t32, int n_frames, float *t32c, uint64_t *totals,
                                     double *scores, double *best)
{
    const int cw = fp.c1 - fp.c0 + 1, ch = fp.r1 - fp.r0 + 1;
    hipLaunchKernelGGL(crop_total_kernel, dim3(n_frames), dim3(1024), 0, st, fp, t32, t32c, totals);
    // A workgroup takes 4 x F frames (F per wave), and the grid walks the table once per such chunk of frames (rows fastest): the
    // chunk's cropped targets stay in L2 while every row gathers from them, and the table streams past once per chunk.
    // Measured for 256 frames x 15 625 rows at 160x90 (tools/r03_tbl_run.sh), a whole wave per frame group: F = 8 7.2 ms (221
    // registers, two waves per SIMD), F = 4 5.1 ms, F = 2 4.8-5.0 ms; one frame per wave and every frame in one workgroup (round 3's
    // first form) 6.6 ms.  With F = 2 and the wave split into lane groups of 32 / 16 / 8: 4.0 / 3.4 / 3.2 ms.
    static const int F = [] { const char *e = std::getenv("ROPE_TABLE_FRAMES"); const int v = e ? std::atoi(e) : 2; return v == 4 || v == 8 ? v : 2; }();
    static const int LANES = [] { const char *e = std::getenv("ROPE_TABLE_LANES"); const int v = e ? std::atoi(e) : 16; return v == 64 || v == 8 ? v : 16; }();
    const int per_wg = 4 * F * (64 / LANES);
    const int chunks = std::max(1, (n_frames + per_wg - 1) / per_wg);
    const dim3 grid(C, chunks);
    const int words = (int)table_crop_words(fp);
    const double n_pix = (double)cw * (double)ch;
#define ROPE_TABLE_LAUNCH(F_, L_) hipLaunchKernelGGL((table_score_frames_kernel<F_, L_>), grid, dim3(256), 0, st, words, counts, offs, goff, gval, t32c, totals, n_frames, n_pix, scores)
    if (LANES == 16) { if (F == 2) ROPE_TABLE_LAUNCH(2, 16); else if (F == 4) ROPE_TABLE_LAUNCH(4, 16); else ROPE_TABLE_LAUNCH(8, 16); }
    else if (LANES == 8) { if (F == 2) ROPE_TABLE_LAUNCH(2, 8); else ROPE_TABLE_LAUNCH(4, 8); }
    else { if (F == 2) ROPE_TABLE_LAUNCH(2, 64); else if (F == 4) ROPE_TABLE_LAUNCH(4, 64); else ROPE_TABLE_LAUNCH(8, 64); }
#undef ROPE_TABLE_LAUNCH
    hipLaunchKernelGGL(argmin_sets_kernel, dim3(n_frames), dim3(C <= 64 ? 64 : (C <= 256 ? 256 : 1024)), 0, st, scores, C, best);
    return hipGetLastError();
}

hipError_t launch_table_count(hipStream_t st, int cw, int ch, const float *table, int C, uint32_t *counts, unsigned long long *offs,
                              unsigned long long *used)
{
    hipLaunchKernelGGL(table_count_kernel, dim3(C), dim3(256), 0, st, cw, ch, table, counts, offs, used);
    return hipGetLastError();
}

hipError_t launch_table_fill(hipStream_t st, int cw, int ch, const float *table, int C, const unsigned long long *offs, uint32_t *goff, float4 *gval)
{
    hipLaunchKernelGGL(table_fill_kernel, dim3(C), dim3(256), 0, st, cw, ch, table, offs, goff, gval);
    return hipGetLastError();
}

hipError_t launch_table_score(hipStream_t st, const FrameParams &fp, const uint32_t *counts, const unsigned long long *offs, const uint32_t *goff,
                              const float4 *gval, int C, const float *t32, float *t32c, uint64_t *total, uint64_t *sums)
{
    hipLaunchKernelGGL(crop_total_kernel, dim3(1), dim3(1024), 0, st, fp, t32, t32c, total);
    hipLaunchKernelGGL(table_score_kernel, dim3(C), dim3(256), 0, st, counts, offs, goff, gval, t32c, total, sums);
    return hipGetLastError();
}

hipError_t launch_resolve(hipStream_t st, const uint32_t *key, int n, const FrameParams &fp, float *depth, uint8_t *ids)
{
    hipLaunchKernelGGL(resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, st, key, n, fp.c_num, fp.c_sum, fp.c_dif, depth, ids);
    return hipGetLastError();
}

}  // namespace rope
