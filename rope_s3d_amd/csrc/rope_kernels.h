// rope_kernels.h — shared between the kernels (rope_kernels.hip) and the C-ABI host side (rope_abi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rope_s3d.h"

namespace rope {

#ifndef ROPE_TILE_W
#define ROPE_TILE_W 128
#endif
#ifndef ROPE_TILE_H
#define ROPE_TILE_H 96
#endif
#ifndef ROPE_NWAVES
#define ROPE_NWAVES 12                      // 2 workgroups x 12 waves per CU = 6 waves per SIMD, 81.5 KB of LDS each
#endif
#ifndef ROPE_MIN_WAVES_PER_SIMD
#define ROPE_MIN_WAVES_PER_SIMD 6
#endif
#ifndef ROPE_MIN_WAVES_FULL
#define ROPE_MIN_WAVES_FULL 6               // the link counts are packed fields (score_pixel): fits 80 VGPRs without spills
#endif
// The instantiations that can clip at the near plane: ONE workgroup per CU (3 waves per SIMD, up to 168 VGPRs), so that they
// compile without scratch.  At 6 waves per SIMD (80 VGPRs) they spill 65-69 VGPRs and 61-63 SGPRs into 200-216 bytes of scratch
// per lane, and the layer-queue kernel of that build, raster_queue_kernel<DEPTH, LAYER, CLIP>, gave sums that differed from run
// to run on a scene without a single triangle near the plane (round 3: tools/dbg_clip.py, profiles/r03_clip_fault.txt; same
// source, no scratch: deterministic and bit-equal to the plain kernels).  They are the rare path — a camera within the robot's
// reach of the near plane — and correctness there is worth more than occupancy.
#ifndef ROPE_MIN_WAVES_CLIP
#define ROPE_MIN_WAVES_CLIP 3
#endif
constexpr int TILE_W = ROPE_TILE_W;
constexpr int TILE_H = ROPE_TILE_H;
constexpr int NWAVES = ROPE_NWAVES;       // waves per workgroup of the raster kernel
constexpr int NTHREADS = NWAVES * 64;
#ifndef ROPE_SMALL_TRI_ROWS
#define ROPE_SMALL_TRI_ROWS 4
#endif
#ifndef ROPE_SMALL_TRI_COLS
#define ROPE_SMALL_TRI_COLS 4
#endif
constexpr int SMALL_TRI_COLS = ROPE_SMALL_TRI_COLS;
constexpr int SMALL_TRI_ROWS = ROPE_SMALL_TRI_ROWS;   // boxes up to 4 samples wide and this many rows are walked by one lane
constexpr int MAX_MESHLETS = 2048;        // capacity of the per-tile meshlet list in LDS
#ifndef ROPE_MESHLET_MAX_VERTS
#define ROPE_MESHLET_MAX_VERTS 64
#endif
constexpr int MESHLET_MAX_VERTS = ROPE_MESHLET_MAX_VERTS;
constexpr int MESHLET_MAX_TRIS = 128;
constexpr int COMPACT_PX = 60;            // meshlets no larger than this on screen take the 32-bit triangle set-up
// raster queue: pairs in classes by weight (heaviest first); weights are kept for frames of at most this many tiles
#ifndef ROPE_QUEUE_CLASSES
#define ROPE_QUEUE_CLASSES 6
#endif
#ifndef ROPE_QUEUE_TOP_LOG2
#define ROPE_QUEUE_TOP_LOG2 14              // pairs of 2^14 triangles and more are the heaviest class
#endif
constexpr int QUEUE_CLASSES = ROPE_QUEUE_CLASSES, QUEUE_TOP_LOG2 = ROPE_QUEUE_TOP_LOG2, QUEUE_COUNTERS = 2 + QUEUE_CLASSES, QUEUE_WEIGHT_TILES = 256;
constexpr int MAX_MASK_WORDS = 256;        // tile-mask words per candidate (8192 tiles)
constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t D24_MAX = 16777215u;

// uint64 words of one candidate's integer sums
enum { SUM_CNT = 0, SUM_S1 = 1, SUM_AA = 2, SUM_AB = 3, SUM_BB = 4, SUM_LINK0 = 5 };
static_assert(SUM_LINK0 + 3 * ROPE_MAX_LINKS == ROPE_SUM_WORDS, "sum layout");

enum { MODE_SCORE = 0, MODE_DUMP = 1, MODE_COVER = 2, MODE_LAYER = 3, MODE_TABLE = 4, MODE_SPLIT = 5, MODE_SPLIT_GEO = 6 };

struct FrameParams {
    int W, H, tiles_x, tiles_y;
    int r0, r1, c0, c1;                   // rows/cols taking part in the loss (whole frame unless lookup crop)
    float c_num, c_sum, c_dif;            // 2nf, f+n, f-n as float32 (pyrender depth read-back)
#ifdef ROPE_PROFILE
    int debug;                            // librope_hip_profile.so only: bit mask of kernel phases to skip (rope_debug_skip)
#endif
};

// Phase-skipping switches exist only in the profiling build (tools/build_variants.py, -DROPE_PROFILE): the shipped
// library has neither the branches nor rope_debug_skip.
#ifdef ROPE_PROFILE
#define ROPE_SKIP(fp, bit) (((fp).debug & (bit)) != 0)
#else
#define ROPE_SKIP(fp, bit) false
#endif

// Execution strategies that change the launch structure but never a result (rope_set_strategy)
enum { STRATEGY_NO_LAYERS = 1, STRATEGY_NO_SPLIT = 2, STRATEGY_NO_PARENTS = 4, STRATEGY_NO_QUEUE = 8, STRATEGY_CLIP_KERNELS = 16, STRATEGY_SEPARATE_GEOMETRY = 32 };

struct RobotParams {
    const uint32_t *ml_header;            // n_meshlets x 8
    const float *ml_verts;                // x3
    const uint32_t *ml_tris;
    const float *ml_aabb;                 // n_meshlets x 8: box centre xyz,0 and half extents xyz,0 (link frame)
    int link_first[ROPE_MAX_LINKS + 1];
    int n_meshlets;
};

struct LinkFlags { uint8_t f[8]; };

// Arguments of one raster launch.  Grid = (tiles, rows); a row is a candidate, or in MODE_LAYER a
// shared layer (links [0, n_shared) of all candidates with the same first two joint angles).
struct RasterArgs {
    int l_begin, l_end;                   // links rasterised by this launch
    int n_render;                         // links that count for the loss terms
    const float *mvp;                     // C x 6 x 16
    const short4 *bounds;                 // C x n_meshlets screen boxes
    const uint32_t *mask_lo, *mask_hi;    // C x mask_words: tiles touched by links < n_shared / >= n_shared
    int mask_words;
    const int32_t *cand_of_row;           // MODE_LAYER: layer -> representative candidate; else nullptr
    const int32_t *layer_of;              // candidate -> layer, nullptr when layers are not in use
    const int32_t *layer_rep;             // layer -> representative candidate (with layer_of)
    // MODE_LAYER, second level: each row's tile is merged (min) with the tile of its parent row of an earlier
    // MODE_LAYER launch (links shared even more widely: links 0-1 per distinct q0) before it is stored and summed
    const uint32_t *base_layers;
    const int32_t *base_of_row;           // row -> parent row
    const int32_t *base_rep;              // parent row -> its candidate: a parent tile exists only where that candidate's mask_lo says so
    uint32_t *layers;                     // n_layers x n_tiles x (TILE_W*TILE_H) keys
    uint64_t *layer_sums;                 // n_layers x n_tiles x ROPE_SUM_WORDS: loss sums of the layer alone
    const uint64_t *tq; const float *t32;
    // camera-pose path and batches over several frames' targets: planes of all frames back to back (frame stride H*W; tl: 6
    // per-link planes per frame, bit 40 = link mask, bits 0..38 = masked target depth) and each candidate's frame index
    const uint64_t *tl;
    const int32_t *frame_of;
    uint64_t *sums; uint32_t *key_out; uint8_t *cover;
    float *table;                         // MODE_TABLE: C x crop_h x crop_w sqrt-depth (crop = fp.r0..c1)
    // few candidates: the meshlets of a (tile, candidate) are split over `split` workgroups (grid z) that merge
    // their LDS tiles into gtile with atomicMin (MODE_SPLIT); score_gtile_kernel then scores and re-clears it
    int split;
    uint32_t *gtile;                      // C x n_tiles x (TILE_W*TILE_H) keys, 0xFFFFFFFF between passes
    // MODE_SPLIT_GEO: MODE_SPLIT without the geometry launch in front — every workgroup works out its candidate's six link
    // matrices itself (the chain is a few hundred double operations) and the screen boxes of its own share of the meshlets
    // only, so the fk + boxes pass over all 1 450 meshlets, a launch of its own, disappears.  No masks then: a workgroup that
    // lists a meshlet for its tile stamps touched[candidate x tiles + tile] with this pass's number, and score_gtile_kernel
    // scores the tiles that carry it.
    const double *cand_q, *joint_fixed, *joint_axes, *PV;
    int *touched;
    int pass_id;
};

hipError_t launch_fk(hipStream_t st, const double *cand, int C, int n_render, const double *joint_fixed,
                     const double *joint_axes, const double *PV, const int32_t *view_of, float *mvp, uint64_t *sums,
                     uint32_t *mask_lo, uint32_t *mask_hi, int mask_words,
                     int *queue_counters /* 2 x QUEUE_COUNTERS ints cleared for launch_raster_queue and launch_layer_queue, or nullptr */,
                     uint32_t *tile_tris /* C x n_tiles weights cleared for launch_bounds, or nullptr */, uint32_t *tile_tris_lo /* with tile_tris */,
                     int n_tiles);
// screen bounding box of every meshlet of every candidate + the candidate's masks of touched tiles
hipError_t launch_bounds(hipStream_t st, int C, const FrameParams &fp, const RobotParams &rp, int n_render, int n_shared,
                         const float *mvp, short4 *bounds, uint32_t *mask_lo, uint32_t *mask_hi, int mask_words,
                         const int32_t *layer_of /* with layer_rep: shared links only for representatives; or nullptr */,
                         const int32_t *layer_rep,
                         uint32_t *tile_tris /* C x n_tiles: triangles of the candidate's own links per tile (n_tiles <= QUEUE_WEIGHT_TILES), or nullptr */,
                         uint32_t *tile_tris_lo /* with tile_tris: the same for the shared links lo_first .. n_shared - 1 */, int lo_first);
// small batches: launch_fk + launch_bounds as one kernel, one workgroup per candidate
hipError_t launch_fk_bounds(hipStream_t st, const double *cand, int C, const FrameParams &fp, const RobotParams &rp, int n_render,
                            int n_shared, const double *joint_fixed, const double *joint_axes, const double *PV,
                            const int32_t *view_of, float *mvp, short4 *bounds, uint64_t *sums, uint32_t *mask_lo,
                            uint32_t *mask_hi, int mask_words);
// clip: the kernels that can cut triangles at the near plane (slower by several per cent: only when the camera is close
// enough to the robot for a triangle to reach the plane)
hipError_t launch_raster(int mode, int loss, int rows, hipStream_t st, const FrameParams &fp, const RobotParams &rp,
                         const RasterArgs &a, bool clip);
// MODE_SCORE over the (candidate, tile) pairs that have something to draw, from a queue (`items`: rows x tiles entries,
// `counters`: the two ints launch_fk cleared) by `workgroups` resident workgroups; layer-only tiles are settled on the way
hipError_t launch_raster_queue(int loss, int rows, int workgroups, hipStream_t st, const FrameParams &fp, const RobotParams &rp,
                               const RasterArgs &a, uint32_t *items /* QUEUE_CLASSES x segment */, size_t segment, int *counters,
                               const uint32_t *tile_tris /* weights from launch_bounds, or nullptr: one class */, bool clip);
// the same for a MODE_LAYER launch (a.cand_of_row set): (layer, tile) pairs of the representatives' shared links, by their weight
hipError_t launch_layer_queue(int loss, int rows, int workgroups, hipStream_t st, const FrameParams &fp, const RobotParams &rp,
                              const RasterArgs &a, uint32_t *items, size_t segment, int *counters, const uint32_t *tile_tris_lo, bool clip);
// scores what a MODE_SPLIT launch merged into a.gtile; `slices` row slices per tile (a divisor of TILE_H)
hipError_t launch_score_gtile(int loss, int rows, int slices, hipStream_t st, const FrameParams &fp, const RasterArgs &a);
// "nothing rendered" sums of n_frames frames (planes back to back): empty_sums n_frames x n_tiles x SUM_WORDS of scratch,
// total n_frames x SUM_WORDS
hipError_t launch_empty(int loss, hipStream_t st, const FrameParams &fp, const uint64_t *tq, const float *t32, const uint64_t *tl,
                        uint64_t *empty_sums, uint64_t *total, int n_frames = 1);
hipError_t launch_finalize(hipStream_t st, uint64_t *sums, const uint64_t *total_empty, int C, int loss, int n_render,
                           double n_pix, const LinkFlags &lf, double *err /* C + 2: errors, best error, best index */);
// The stored lookup table keeps, of every row, the groups of four consecutive samples of a crop row in which anything was drawn:
// counts[k] groups from offs[k] on in goff (offset of the group's first sample inside the crop) / gval (its four values).
// launch_table_count: dense rows (C x ch x cw) -> counts / offs, *used = groups in all (zero it first); launch_table_fill writes them.
hipError_t launch_table_count(hipStream_t st, int cw, int ch, const float *table, int C, uint32_t *counts, unsigned long long *offs,
                              unsigned long long *used);
hipError_t launch_table_fill(hipStream_t st, int cw, int ch, const float *table, int C, const unsigned long long *offs, uint32_t *goff, float4 *gval);
// floats of one frame's cropped target (t32c) as the table kernels lay it out: crop rows padded to whole
// groups of four samples
static inline size_t table_crop_words(const FrameParams &fp)
{
    return (size_t)(((fp.c1 - fp.c0 + 1) + 3) / 4) * 4 * (size_t)(fp.r1 - fp.r0 + 1);
}
// score every row of a stored lookup table against the float32 target plane
// t32c: scratch of table_crop_words floats (the cropped target, rebuilt by every call);
// total: ROPE_SUM_WORDS words of scratch
hipError_t launch_table_score(hipStream_t st, const FrameParams &fp, const uint32_t *counts, const unsigned long long *offs, const uint32_t *goff,
                              const float4 *gval, int C, const float *t32, float *t32c, uint64_t *total, uint64_t *sums);
// batches over several frames' targets (rope_eval_targets / rope_lookup_score_targets)
hipError_t launch_finalize_frames(hipStream_t st, uint64_t *sums, const uint64_t *totals /* frames x SUM_WORDS */, const int32_t *frame_of,
                                  const LinkFlags *flags /* per frame, device */, int C, int loss, int n_render, double n_pix, double *err /* C */);
// first argmin of n_sets sets of C doubles: best[2 k] error, best[2 k + 1] index
hipError_t launch_argmin_sets(hipStream_t st, const double *err, int C, int n_sets, double *best);
// the stored table against the float32 planes of n_frames frames: t32c n_frames x table_crop_words floats of scratch, totals n_frames x
// SUM_WORDS of scratch, scores n_frames x C, best n_frames x 2 (score, row)
hipError_t launch_table_score_frames(hipStream_t st, const FrameParams &fp, const uint32_t *counts, const unsigned long long *offs,
                                     const uint32_t *goff, const float4 *gval, int C, const float *t32, int n_frames, float *t32c, uint64_t *totals,
                                     double *scores, double *best);
hipError_t launch_resolve(hipStream_t st, const uint32_t *key, int n, const FrameParams &fp, float *depth, uint8_t *ids);

}  // namespace rope
