// rope_seg.hip — the two box-shaped steps of the segmentation stage that are not library convolutions, for gfx950:
// greedy non-maximum suppression (proposal layer and detection layer) and the pyramid RoIAlign in front of the two heads.
// The stage's host side is rope_s3d_amd/maskrcnn.py; these entry points take plain device pointers and a HIP stream
// (include/rope_s3d.h).  Both reproduce the stage's PyTorch formulation operation by operation (same float32 steps, same
// bfloat16 roundings), so the detections do not depend on which of the two runs.
//
// Reference: the Matterport Mask R-CNN the reference trains and predicts with (robotpose/prediction/predict.py:94-98,416):
//   ProposalLayer / refine_detections_graph -> tf.image.non_max_suppression
//   PyramidROIAlign -> tf.image.crop_and_resize(bilinear)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rope_s3d.h"

namespace {

// ------------------------------------------------------------------ NMS -----
// Boxes arrive sorted by descending score.  Pass 1: bit (i, j) of the suppression matrix = "box i and a later box j of the
// same group overlap by more than thr", 64 x 64 bits per workgroup, upper triangle only.  Pass 2: one wave per set walks
// the boxes in blocks of 64: a block's own 64 x 64 bits settle it in registers, then the rows of the boxes it kept are
// OR-ed into the "removed" words of the later blocks (independent loads, not one dependent load per box).

__device__ __forceinline__ bool iou_over(const float4 a, const float4 b, const float thr)
{
    // the operation order of maskrcnn.py::_iou_over_b (one IEEE operation per step; the file is built with -ffp-contract=off)
    const float ih = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f);
    const float iw = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
    const float inter = ih * iw;
    const float area_a = fmaxf(a.z - a.x, 0.0f) * fmaxf(a.w - a.y, 0.0f);
    const float area_b = fmaxf(b.z - b.x, 0.0f) * fmaxf(b.w - b.y, 0.0f);
    return inter / fmaxf((area_a + area_b) - inter, 1e-12f) > thr;
}

__global__ void __launch_bounds__(64)
nms_mask_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ groups, int n, int words, float thr,
                unsigned long long *__restrict__ mask)
{
    const int cb = blockIdx.x, rb = blockIdx.y, set = blockIdx.z, lane = threadIdx.x;
    if (cb < rb) return;                                   // below the diagonal: an earlier box is never suppressed by a later one
    __shared__ float4 s_box[64];
    __shared__ int32_t s_grp[64];
    const size_t base = (size_t)set * n;
    const int col = cb * 64 + lane;
    s_box[lane] = col < n ? boxes[base + col] : make_float4(0.f, 0.f, 0.f, 0.f);
    s_grp[lane] = (groups && col < n) ? groups[base + col] : 0;
    __syncthreads();
    const int row = rb * 64 + lane;
    if (row >= n) return;
    const float4 a = boxes[base + row];
    const int32_t ga = groups ? groups[base + row] : 0;
    unsigned long long bits = 0;
    const int j_end = min(64, n - cb * 64);
    for (int j = 0; j < j_end; j++) {
        const int c = cb * 64 + j;
        if (c > row && s_grp[j] == ga && iou_over(a, s_box[j], thr)) bits |= 1ull << j;
    }
    mask[(base + row) * words + cb] = bits;
}

// One workgroup of 16 waves per set.  Block after block of 64 boxes: wave 0 settles the block from its own 64 x 64 bits (a chain
// of dependent steps, one per kept box), then every wave takes some of the kept boxes' rows and ORs them into the later blocks'
// "removed" words in LDS — the rows of a block are all in flight at once (a single wave walking them, four loads at a time, spent
// most of the launch waiting: 0.51 ms for 8 x 6000 boxes).
constexpr int NMS_SCAN_THREADS = 1024;

__global__ void __launch_bounds__(NMS_SCAN_THREADS)
nms_scan_kernel(const unsigned long long *__restrict__ mask, const uint8_t *__restrict__ valid, int n, int words, int limit,
                uint8_t *__restrict__ keep)
{
    extern __shared__ unsigned long long s_removed[];       // `words` words + 64 for the current block's own bits
    unsigned long long *const s_sub = s_removed + words;
    __shared__ int s_rows[64];                                // rows (within the block) of the boxes it kept
    __shared__ unsigned long long s_keepbits;
    __shared__ int s_kept;
    const int set = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int N_WAVES = NMS_SCAN_THREADS / 64;
    const size_t base = (size_t)set * n;
    // padding and the bits past n start out "removed": never kept, never suppressing
    for (int i0 = wave * 64; i0 < words * 64; i0 += NMS_SCAN_THREADS) {
        const int i = i0 + lane;
        const unsigned long long r = __ballot(i >= n || (valid && !valid[base + i]));
        if (lane == 0) s_removed[i0 >> 6] = r;
    }
    if (tid == 0) s_kept = 0;
    __syncthreads();
    unsigned long long own = (wave == 0 && lane < n) ? mask[(base + lane) * words] : 0ull;     // wave 0: this lane's row of the block's own 64 x 64 bits
    for (int blk = 0; blk < words; blk++) {
        const int row = blk * 64 + lane;
        if (wave == 0) {
            s_sub[lane] = own;
            // the next block's own bits depend on nothing here: fetched while this block is settled
            own = (blk + 1 < words && row + 64 < n) ? mask[(base + row + 64) * words + blk + 1] : 0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // the block's own bits settle it: the first box still standing is kept and strikes the later ones it overlaps
            int kept = s_kept;
            unsigned long long keepbits = 0, standing = ~s_removed[blk];
            while (standing && kept < limit) {
                const int t = __builtin_ctzll(standing);
                keepbits |= 1ull << t;
                standing &= ~(s_sub[t] | (1ull << t));        // row t only has bits above t
                kept++;
            }
            if (row < n) keep[base + row] = (uint8_t)((keepbits >> lane) & 1ull);
            if ((keepbits >> lane) & 1ull) s_rows[__popcll(keepbits & ((1ull << lane) - 1ull))] = lane;
            if (lane == 0) { s_keepbits = keepbits; s_kept = kept; }
        }
        __syncthreads();
        const int kept = s_kept;
        if (kept >= limit) {                                  // the remaining blocks only hold zeros
            for (size_t i = (size_t)(blk + 1) * 64 + tid; i < (size_t)n; i += NMS_SCAN_THREADS) keep[base + i] = 0;
            return;
        }
        if (blk + 1 == words) return;
        // rows of the kept boxes into the later blocks' words
        const int n_kept = __popcll(s_keepbits);
        for (int i = wave; i < n_kept; i += N_WAVES) {
            const unsigned long long *const m = mask + (base + blk * 64 + s_rows[i]) * words;
            for (int w = blk + 1 + lane; w < words; w += 64) {
                const unsigned long long bits = m[w];
                if (bits) atomicOr(&s_removed[w], bits);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ RoIAlign -----
struct Levels { int H[4], W[4]; long long off[4]; };

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f)
{
    // round to nearest even, as torch's float -> bfloat16 (NaN stays NaN)
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// one bfloat16 tensor operation: float32 arithmetic, result rounded to bfloat16
__device__ __forceinline__ float bmul(float a, float b) { return bf2f(f2bf(a * b)); }
__device__ __forceinline__ float badd(float a, float b) { return bf2f(f2bf(a + b)); }

// One workgroup per (box, sample row); 8 sample columns at a time, 32 lanes x 8 channels (16 bytes) per sample.
// out: K x pool x pool x C bfloat16 (the heads read it as a channels-last K x C x pool x pool).
__global__ void __launch_bounds__(256)
roi_align_kernel(const uint16_t *__restrict__ rows, const float4 *__restrict__ boxes, const int32_t *__restrict__ frame,
                 Levels lv, int channels, int pool, float inv_level_unit, const float *__restrict__ t, uint16_t *__restrict__ out)
{
    const int k = blockIdx.x, py = blockIdx.y, tid = threadIdx.x;
    const int lanes_per_sample = channels / 8, samples_per_round = 256 / lanes_per_sample;
    const int sub = tid / lanes_per_sample, ch = (tid % lanes_per_sample) * 8;
    const float4 b = boxes[k];                                          // y1, x1, y2, x2 (normalised)
    const float h = b.z - b.x, w = b.w - b.y;
    // level by box area (PyramidROIAlign): round(4 + log2(sqrt(h w) / (224 / size))) clamped to 2..5
    const float lf = fminf(fmaxf(rintf(4.0f + log2f(sqrtf(fmaxf(h * w, 1e-12f)) * inv_level_unit)), 2.0f), 5.0f);
    const int li = (int)lf - 2;
    const int Hf = lv.H[li], Wf = lv.W[li];
    const long long base = lv.off[li] + (long long)frame[k] * ((long long)Hf * Wf);
    const float hm = (float)(Hf - 1), wm = (float)(Wf - 1);
    const float ys = (b.x + t[py] * (b.z - b.x)) * hm;
    const float y0 = floorf(ys);
    const float wy = bf2f(f2bf(ys - y0)), omwy = bf2f(f2bf(1.0f - wy));
    const bool in_y = ys >= 0.0f && ys <= hm;
    const long long y0i = (long long)y0;
    const int y0c = (int)min(max(y0i, 0ll), (long long)(Hf - 1)), y1c = (int)min(max(y0i + 1, 0ll), (long long)(Hf - 1));
    for (int px = sub; px < pool; px += samples_per_round) {
        const float xs = (b.y + t[px] * (b.w - b.y)) * wm;
        const float x0 = floorf(xs);
        const float wx = bf2f(f2bf(xs - x0)), omwx = bf2f(f2bf(1.0f - wx));
        const float inside = (in_y && xs >= 0.0f && xs <= wm) ? 1.0f : 0.0f;
        const long long x0i = (long long)x0;
        const int x0c = (int)min(max(x0i, 0ll), (long long)(Wf - 1)), x1c = (int)min(max(x0i + 1, 0ll), (long long)(Wf - 1));
        const uint4 g00 = *reinterpret_cast<const uint4 *>(rows + (base + (long long)y0c * Wf + x0c) * channels + ch);
        const uint4 g10 = *reinterpret_cast<const uint4 *>(rows + (base + (long long)y1c * Wf + x0c) * channels + ch);
        const uint4 g01 = *reinterpret_cast<const uint4 *>(rows + (base + (long long)y0c * Wf + x1c) * channels + ch);
        const uint4 g11 = *reinterpret_cast<const uint4 *>(rows + (base + (long long)y1c * Wf + x1c) * channels + ch);
        const uint32_t a00[4] = {g00.x, g00.y, g00.z, g00.w}, a10[4] = {g10.x, g10.y, g10.z, g10.w};
        const uint32_t a01[4] = {g01.x, g01.y, g01.z, g01.w}, a11[4] = {g11.x, g11.y, g11.z, g11.w};
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t packed = 0;
#pragma unroll
            for (int hlf = 0; hlf < 2; hlf++) {
                const int sh = 16 * hlf;
                const float v00 = bf2f((uint16_t)(a00[q] >> sh)), v10 = bf2f((uint16_t)(a10[q] >> sh));
                const float v01 = bf2f((uint16_t)(a01[q] >> sh)), v11 = bf2f((uint16_t)(a11[q] >> sh));
                // (g00 (1 - wy) + g10 wy) (1 - wx) + (g01 (1 - wy) + g11 wy) wx, every step a bfloat16 tensor operation
                const float left = bmul(badd(bmul(v00, omwy), bmul(v10, wy)), omwx);
                const float right = bmul(badd(bmul(v01, omwy), bmul(v11, wy)), wx);
                const float val = bmul(badd(left, right), inside);
                packed |= (uint32_t)f2bf(val) << sh;
            }
            o[q] = packed;
        }
        *reinterpret_cast<uint4 *>(out + (((size_t)k * pool + py) * pool + px) * channels + ch) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------- bias / residual / ReLU -----
// What follows a library convolution in the trunk, in one pass over its output instead of up to four (bias add, residual add,
// ReLU as separate tensor operations): y <- [relu]( bf16( bf16(y + bias[channel]) [+ residual] ) ), the roundings of the separate
// operations kept.  Eight bfloat16 per lane (16 bytes).  chan_inner = H*W/8 for NCHW (eight neighbours share a channel),
// 0 for channels-last (the eight are channels c .. c+7).
__global__ void __launch_bounds__(256)
bias_act_kernel(uint16_t *__restrict__ y, const uint16_t *__restrict__ bias, const uint16_t *__restrict__ res, long long n8, int channels,
                int chan_inner, int relu)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    uint4 v = reinterpret_cast<uint4 *>(y)[i];
    uint32_t w[4] = {v.x, v.y, v.z, v.w}, r[4] = {0, 0, 0, 0}, b[4];
    if (chan_inner) {
        const uint32_t bb = bias[(i / chan_inner) % channels];
        b[0] = b[1] = b[2] = b[3] = bb | (bb << 16);
    } else {
        const uint4 bv = *reinterpret_cast<const uint4 *>(bias + (i * 8) % channels);
        b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
    }
    if (res) { const uint4 rv = reinterpret_cast<const uint4 *>(res)[i]; r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w; }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t out = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int sh = 16 * h;
            float f = badd(bf2f((uint16_t)(w[q] >> sh)), bf2f((uint16_t)(b[q] >> sh)));
            if (res) f = badd(f, bf2f((uint16_t)(r[q] >> sh)));
            if (relu) f = f > 0.0f ? f : 0.0f;             // relu(NaN) stays out of it: activations are finite
            out |= (uint32_t)f2bf(f) << sh;
        }
        w[q] = out;
    }
    reinterpret_cast<uint4 *>(y)[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

}  // namespace

extern "C" int rope_seg_bias_act(void *y_bf16, const void *bias_bf16, const void *residual_bf16, int64_t n, int channels, int64_t inner,
                                 int relu, void *stream)
{
    if (!y_bf16 || !bias_bf16 || n < 8 || (n & 7) || channels < 1) return ROPE_E_ARG;
    if (inner > 1 ? (inner & 7) != 0 : (channels & 7) != 0) return ROPE_E_ARG;   // eight neighbours share a channel, or are eight channels
    const long long n8 = n / 8;
    hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint16_t *>(y_bf16),
                       reinterpret_cast<const uint16_t *>(bias_bf16), reinterpret_cast<const uint16_t *>(residual_bf16), n8, channels,
                       inner > 1 ? (int)(inner / 8) : 0, relu);
    return hipGetLastError() == hipSuccess ? ROPE_OK : ROPE_E_HIP;
}

extern "C" int rope_seg_nms(const float *boxes, const int32_t *groups, const uint8_t *valid, int n_sets, int n, float iou_thr,
                            int limit, uint64_t *scratch, uint8_t *keep, void *stream)
{
    if (!boxes || !scratch || !keep || n_sets < 1 || n < 1 || limit < 1) return ROPE_E_ARG;
    const int words = (n + 63) / 64;
    if ((size_t)(words + 64) * 8 > 64 * 1024) return ROPE_E_ARG;          // the scan keeps one bit per box in LDS: n <= 520 000
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, n_sets), dim3(64), 0, st, reinterpret_cast<const float4 *>(boxes), groups, n,
                       words, iou_thr, reinterpret_cast<unsigned long long *>(scratch));
    hipLaunchKernelGGL(nms_scan_kernel, dim3(n_sets), dim3(NMS_SCAN_THREADS), (size_t)(words + 64) * 8, st,
                       reinterpret_cast<const unsigned long long *>(scratch), valid, n, words, limit, keep);
    return hipGetLastError() == hipSuccess ? ROPE_OK : ROPE_E_HIP;
}

extern "C" int rope_seg_roi_align(const void *rows_bf16, const float *boxes, const int32_t *frame, const int32_t *level_hw,
                                  const int64_t *level_off, int n_boxes, int channels, int pool, float inv_level_unit,
                                  const float *t, void *out_bf16, void *stream)
{
    if (!rows_bf16 || !boxes || !frame || !level_hw || !level_off || !t || !out_bf16) return ROPE_E_ARG;
    if (n_boxes < 1 || pool < 1 || channels < 8 || channels > 2048 || (channels & (channels - 1))) return ROPE_E_ARG;   // 16-byte lanes, 256 % (channels / 8) == 0
    Levels lv;
    for (int l = 0; l < 4; l++) {
        lv.H[l] = level_hw[2 * l]; lv.W[l] = level_hw[2 * l + 1]; lv.off[l] = level_off[l];
        if (lv.H[l] < 1 || lv.W[l] < 1 || lv.off[l] < 0) return ROPE_E_ARG;
    }
    hipLaunchKernelGGL(roi_align_kernel, dim3(n_boxes, pool), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint16_t *>(rows_bf16),
                       reinterpret_cast<const float4 *>(boxes), frame, lv, channels, pool, inv_level_unit, t, reinterpret_cast<uint16_t *>(out_bf16));
    return hipGetLastError() == hipSuccess ? ROPE_OK : ROPE_E_HIP;
}
