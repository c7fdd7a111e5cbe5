// rope_abi.hip — host side of librope_hip.so: context, HBM buffers, launch sequencing.
// Declarations and the reference call sites each entry point replaces: include/rope_s3d.h.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rope_kernels.h"

using namespace rope;

struct rope_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // robot
    bool have_robot = false;
    RobotParams rp{};
    uint32_t *d_header = nullptr, *d_tris = nullptr;
    float *d_verts = nullptr, *d_aabb = nullptr;
    double *d_joint_fixed = nullptr, *d_joint_axes = nullptr;
    int n_links = 0, n_meshlets = 0;
    double reach = 0.0;                     // no point of any link, at any joint angles, is farther than this from the base frame's origin
    double h_PV[16] = {};                   // host copy of the camera matrix (the near-plane test below)
    bool clip_views = false;                // rope_eval_views: one of the call's cameras is close enough for near-plane clipping

    // camera
    bool have_camera = false;
    FrameParams fp{};
    double *d_PV = nullptr;
    int n_tiles = 0;

    // target
    bool have_target = false;
    uint64_t *d_tq = nullptr;
    float *d_t32 = nullptr;
    bool have_t32 = false;
    LinkFlags lf{};
    uint64_t target_version = 0;
    // per loss: empty-tile sums and their frame total, valid for (target_version, crop)
    uint64_t *d_empty[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t *d_total[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t empty_version[4] = {0, 0, 0, 0};
    int empty_crop[4][4] = {};

    // candidates + results
    int C = 0, cap = 0;
    double *d_cand = nullptr, *d_err = nullptr, *d_best_err = nullptr;
    double *h_stage = nullptr;             // pinned staging: candidates up, errors + best down
    // small batches: the finalize kernel writes errors + best straight into mapped pinned host memory (no copy command)
    static constexpr int HOST_ERR_ROWS = 256;
    // lockstep batches (rope_eval_targets): a step's rows — 2 to 26 per frame, hundreds of frames — go up, and their errors come
    // back, through mapped host memory as well: a pass reads every row once and writes every error once, and the three copy
    // commands it would otherwise queue (rows, frame indices, errors: 20-30 us of stream time each between 100 us of kernels)
    // were a fifth of the device's time in a lockstep run
    static constexpr int TARGET_HOST_ROWS = 16384;
    double *h_err = nullptr, *d_err_host = nullptr;   // host pointer and its device alias
    // small batches: the candidates stay in mapped host memory and the FK kernel reads them from there — no copy command
    // in front of the first launch of a latency-bound chain
    double *h_cand = nullptr, *d_cand_host = nullptr;
    const double *cand_dev = nullptr;                  // where the resident candidates are: d_cand or d_cand_host
    bool err_on_host = false;
    // rope_eval_views / rope_lookup_score reuse C and the per-candidate buffers for rows of their own: after them the
    // resident candidates (and the results of the last eval) are gone until rope_candidates_upload / the next eval
    bool cand_valid = false, results_valid = false;
    float *d_mvp = nullptr;
    short4 *d_bounds = nullptr;
    uint32_t *d_mask_lo = nullptr, *d_mask_hi = nullptr;
    int mask_words = 0;
    // shared upstream layers: candidates with bit-identical (q0, q1) have identical links 0..2
    int n_layers = 0;
    int32_t *d_layer_of = nullptr, *d_layer_rep = nullptr;
    uint32_t *d_layers = nullptr;
    uint64_t *d_layer_sums = nullptr;
    size_t layers_cap = 0;                 // keys allocated in d_layers
    // second level of sharing: layers with bit-identical q0 have identical links 0..1 ("parents")
    int n_parents = 0;
    int32_t *d_parent_of = nullptr, *d_parent_rep = nullptr;   // layer -> parent; parent -> representative candidate
    uint32_t *d_parents = nullptr;
    size_t parents_cap = 0;
    int parent_idx_cap = 0;
    int layer_rep_cap = 0;
    uint64_t *d_sums = nullptr;
    int32_t *d_best_idx = nullptr;
    int last_n_render = 0;

    // small batches: per-candidate tiles in global memory that split workgroups merge into
    int *d_touched = nullptr;              // MODE_SPLIT_GEO: per (candidate, tile) the number of the pass that last drew into it
    size_t touched_cap = 0;
    int pass_id = 0;
    bool mvp_valid = false;                // d_mvp holds the resident candidates' matrices (not after a pass that kept them in LDS)
    uint32_t *d_gtile = nullptr;
    size_t gtile_cap = 0;
    bool gtile_dirty = true;
    // MODE_SPLIT: busy workgroups aimed at per launch (one generation: 2 per CU), fewest / most per (tile, candidate), and the
    // workgroups per launch of the scoring pass that follows.  Measured at 160x90, 640x360 and 640x480 (tools/r02_split_tune.sh).
    int split_target = 512, split_min = 8, split_min_many = 3, split_cap = 64, score_target = 6144;
    int geo_rows = 48;                     // most rows of a small batch whose raster workgroups do their own forward kinematics and boxes
    int layer_min_wg = 64;                 // fewest busy workgroups of a shared-layer launch for it to pay in a small batch (layers_pay)
    int strategy = 0;                          // STRATEGY_* bits (rope_set_strategy): launch structure only, never a result
    // large batches: queue of the (candidate, tile) pairs with something to draw, worked off by a grid that just fills the chip
    uint32_t *d_qitems = nullptr, *d_tile_tris = nullptr, *d_tile_tris_lo = nullptr;
    size_t q_segment = 0;
    bool q_weighted = false;
    int *d_qctr = nullptr;                     // [0] pairs queued, [1] next pair to hand out; cleared by fk_mvp_kernel
    int n_cu = 256;

    // stored lookup table (cropped sqrt-depth of a pose grid)
    float *d_table = nullptr;                  // dense rows while a table is built (freed once it is packed)
    size_t table_cap = 0;
    // the stored table (rope_lookup_build): per row d_tcount groups of four samples from d_toff on — d_tgoff: where in the crop,
    // d_tgval: the four values
    uint32_t *d_tcount = nullptr, *d_tgoff = nullptr;
    unsigned long long *d_toff = nullptr, *d_tused = nullptr;
    float4 *d_tgval = nullptr;
    uint64_t *d_ttotal = nullptr;
    float *d_tc = nullptr;                  // rope_lookup_score: the cropped target, rows padded to whole groups (table_crop_words)
    size_t tc_cap = 0;
    size_t trow_cap = 0;
    int table_C = 0, table_crop[4] = {0, 0, 0, 0};
    size_t table_groups = 0;                   // groups the stored table holds
    uint64_t *d_zero_total = nullptr;
    // scores of the table's rows: buffers of their own (a grid may hold more rows than one candidate batch)
    uint64_t *d_tsums = nullptr;
    double *d_terr = nullptr;
    int tscore_cap = 0;

    // camera-pose path: frames (joint vector + target planes each) scored under candidate views
    int n_frames = 0;
    bool frames_t32 = false, frames_tl = false;
    std::vector<double> h_fq;               // n_frames x 6
    uint64_t *d_ftq = nullptr, *d_ftl = nullptr, *d_ftotal = nullptr, *d_fempty = nullptr;
    float *d_ft32 = nullptr;
    // one pinned host block and its device twin per rope_eval_views call: candidates | view matrices | view index | frame index
    unsigned char *h_vstage = nullptr, *d_vstage = nullptr;
    size_t vstage_cap = 0;
    const double *dv_cand = nullptr, *dv_PV = nullptr;
    const int32_t *dv_view_of = nullptr, *dv_frame_of = nullptr;
    uint64_t *h_vsums = nullptr;            // pinned: K x N x 23 sums, then N x 23 totals
    unsigned char *h_copy = nullptr;        // pinned: staging of pageable host buffers on their way up (copy_h2d_staged)
    size_t vsums_cap = 0;
    size_t frames_cap = 0, frames_tl_cap = 0, frames_t32_cap = 0;
    int ftotal_cap = 0;
    bool ftotal_valid[5] = {false, false, false, false, false};   // per loss kind: d_ftotal[loss] holds the frames' "nothing rendered" totals

    // batched prediction (rope_set_targets / rope_eval_targets / rope_lookup_score_targets): the targets of n_targets frames.  The
    // packed planes and the lookup planes live in d_ftq / d_ft32, which the camera-pose path's frames use as well: setting one
    // kind unsets the other.
    int n_targets = 0;
    bool targets_t32 = false, targets_ts = false;
    float *d_fts32 = nullptr;               // TensorSweep planes (the whole target depth as float32), when given
    size_t fts_cap = 0;
    LinkFlags *d_fflags = nullptr;          // per frame
    int fflags_cap = 0;
    uint64_t *d_tg_total[4] = {nullptr, nullptr, nullptr, nullptr};   // per loss: n_targets x SUM_WORDS "nothing rendered" totals
    uint64_t *d_tg_empty = nullptr;         // scratch: n_targets x n_tiles x SUM_WORDS
    int tg_total_cap = 0;
    size_t tg_empty_cap = 0;
    bool tg_total_valid[4] = {false, false, false, false};
    // rope_stage_targets / rope_commit_targets: the NEXT set of targets, uploaded on a stream of its own while the resident set is
    // in use; committing swaps the two sets of planes
    hipStream_t copy_stream = nullptr;
    unsigned char *h_copy2 = nullptr;       // pinned block for pageable sources on the upload stream
    uint64_t *s_ftq = nullptr;
    float *s_ft32 = nullptr, *s_fts32 = nullptr;
    LinkFlags *s_fflags = nullptr;
    size_t s_frames_cap = 0, s_t32_cap = 0, s_fts_cap = 0;
    int s_fflags_cap = 0;
    int staged_n = 0, staged_W = 0, staged_H = 0;
    std::string stage_err;                  // rope_stage_targets may run beside an evaluation: its own message
    bool staged_t32 = false, staged_ts = false;
    int tg_crop[4][4] = {};
    int32_t *d_frame_of = nullptr, *h_frame_of = nullptr, *d_frame_of_host = nullptr;   // rows' frame indices: device copy / mapped host memory (small batches)
    int frame_of_cap = 0;
    const int32_t *frame_of_dev = nullptr;
    // the stored lookup table against all targets
    float *d_tg_t32c = nullptr;
    uint64_t *d_tg_ltotal = nullptr;
    double *d_tg_scores = nullptr, *d_tg_best = nullptr;
    size_t tg_t32c_cap = 0, tg_scores_cap = 0;
    int tg_best_cap = 0;
    // single target: the float32 plane ROPE_LOSS_TSWEEP reads when it is not the lookup plane (rope_set_target_tsweep)
    float *d_t32ts = nullptr;
    bool have_t32ts = false;

    // single-pose render scratch
    uint32_t *d_key = nullptr;
    float *d_depth = nullptr;
    uint8_t *d_ids = nullptr, *d_cover = nullptr;
};

// ---- roctx ranges (the reference times its stages with utils.Timer / FancyTimer, robotpose/utils.py:122-180).
// Named ranges around the phases of a pass and around every stage of the stage machine show up in `rocprofv3 --marker-trace`;
// the library is looked up at run time (librocprofiler-sdk-roctx, else the older libroctx64) and its absence costs two null checks.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        if (const char *e = std::getenv("ROPE_ROCTX")) if (e[0] == '0') return;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(name, RTLD_LAZY | RTLD_LOCAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
Roctx &roctx() { static Roctx r; return r; }
}  // namespace

void rope_range_push(const char *name) { if (roctx().push) (void)roctx().push(name); }      // also used by rope_predict.cpp (per stage)
void rope_range_pop() { if (roctx().pop) (void)roctx().pop(); }
struct RopeRange {
    explicit RopeRange(const char *name) { rope_range_push(name); }
    ~RopeRange() { rope_range_pop(); }
};

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return ROPE_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define ARG_FAIL(ctx, msg)      \
    do {                        \
        (ctx)->err = (msg);     \
        return ROPE_E_ARG;      \
    } while (0)

template <typename T>
static hipError_t realloc_dev(T **p, size_t n)
{
    if (*p) { hipError_t e = hipFree(*p); *p = nullptr; if (e != hipSuccess) return e; }
    return hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T));
}

static thread_local std::string g_create_err;

// Host buffers handed over the C ABI are pageable; the runtime's own path for those is slow (measured 0.35 GB/s for
// freshly written numpy arrays).  Go through a pinned block instead: memcpy + asynchronous copy, chunk by chunk.
static int copy_h2d_staged(rope_ctx *c, void *dst, const void *src, size_t bytes);
static int copy_d2h_staged(rope_ctx *c, void *dst, const void *src, size_t bytes);   // returns with the data in dst

extern "C" int rope_create(rope_ctx **out, int device)
{
    if (!out) return ROPE_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        g_create_err = "no usable HIP device";
        return ROPE_E_HIP;
    }
    rope_ctx *c = new (std::nothrow) rope_ctx();
    if (!c) return ROPE_E_NOMEM;
    c->device = device;
    if (const char *e = std::getenv("ROPE_SPLIT_TARGET")) c->split_target = std::max(1, std::atoi(e));     // tuning aid
    if (const char *e = std::getenv("ROPE_STRATEGY")) c->strategy = std::atoi(e) & 63;                     // tuning aid: rope_set_strategy's bits
    if (const char *e = std::getenv("ROPE_SPLIT_CAP")) c->split_cap = std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("ROPE_SPLIT_MIN")) c->split_min = std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("ROPE_GEO_ROWS")) c->geo_rows = std::max(0, std::min(256, std::atoi(e)));     // d_touched holds 256 rows
    if (const char *e = std::getenv("ROPE_LAYER_MIN_WG")) c->layer_min_wg = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("ROPE_SPLIT_MIN_MANY")) c->split_min_many = std::max(1, std::min(64, std::atoi(e)));
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        g_create_err = "hipSetDevice/hipStreamCreate failed";
        delete c;
        return ROPE_E_HIP;
    }
    if (hipHostMalloc((void **)&c->h_err, (rope_ctx::TARGET_HOST_ROWS + 2) * sizeof(double), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->d_err_host, c->h_err, 0) != hipSuccess ||
        hipHostMalloc((void **)&c->h_cand, 6 * rope_ctx::TARGET_HOST_ROWS * sizeof(double), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->d_cand_host, c->h_cand, 0) != hipSuccess ||
        hipHostMalloc((void **)&c->h_frame_of, rope_ctx::TARGET_HOST_ROWS * sizeof(int32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->d_frame_of_host, c->h_frame_of, 0) != hipSuccess ||
        hipMalloc((void **)&c->d_best_idx, sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void **)&c->d_qctr, 2 * QUEUE_COUNTERS * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&c->d_best_err, sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&c->d_PV, 16 * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&c->d_joint_fixed, 72 * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&c->d_joint_axes, 18 * sizeof(double)) != hipSuccess) {
        g_create_err = "hipMalloc failed";
        delete c;
        return ROPE_E_NOMEM;
    }
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cu < 1) c->n_cu = 256;
    *out = c;
    return ROPE_OK;
}

// page-locked host memory (rope_host_alloc, or anything else registered with the runtime)?
static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }     // pageable memory: an error by design
    return attr.type == hipMemoryTypeHost;
}

extern "C" void *rope_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

extern "C" void rope_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

// `stream` / `block` / `err`: the context's own stream, pinned block and message by default; rope_stage_targets, which may run beside
// an evaluation on another thread, passes the upload stream and a block and a message of its own
static int copy_h2d_on(hipStream_t stream, unsigned char **block, std::string &err, void *dst, const void *src, size_t bytes)
{
    constexpr size_t CHUNK = 8u << 20;
    auto failed = [&](hipError_t e, const char *what) { if (e != hipSuccess) err = std::string(what) + ": " + hipGetErrorString(e); return e != hipSuccess; };
    if (bytes >= (64u << 10) && is_pinned_host(src))       // already page-locked: one copy at the link's rate, no staging; the caller synchronises
        return failed(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync (host to device)") ? ROPE_E_HIP : ROPE_OK;
    if (!*block && failed(hipHostMalloc((void **)block, 2 * CHUNK, hipHostMallocDefault), "hipHostMalloc (copy block)")) return ROPE_E_HIP;   // two halves, alternating
    int half = 0;
    for (size_t off = 0; off < bytes; off += CHUNK, half ^= 1) {
        const size_t n = std::min(CHUNK, bytes - off);
        // the half about to be overwritten is free again
        if ((off >= 2 * CHUNK || off == 0) && failed(hipStreamSynchronize(stream), "hipStreamSynchronize (copy block)")) return ROPE_E_HIP;
        std::memcpy(*block + half * CHUNK, static_cast<const unsigned char *>(src) + off, n);
        if (failed(hipMemcpyAsync(static_cast<unsigned char *>(dst) + off, *block + half * CHUNK, n, hipMemcpyHostToDevice, stream), "hipMemcpyAsync (host to device)"))
            return ROPE_E_HIP;
    }
    return ROPE_OK;
}

static int copy_h2d_staged(rope_ctx *c, void *dst, const void *src, size_t bytes)
{
    return copy_h2d_on(c->stream, &c->h_copy, c->err, dst, src, bytes);
}

static int copy_d2h_staged(rope_ctx *c, void *dst, const void *src, size_t bytes)
{
    constexpr size_t CHUNK = 8u << 20;
    if (!c->h_copy) HIP_TRY(c, hipHostMalloc((void **)&c->h_copy, 2 * CHUNK, hipHostMallocDefault));
    for (size_t off = 0; off < bytes; off += 2 * CHUNK) {
        const size_t n = std::min(2 * CHUNK, bytes - off);
        HIP_TRY(c, hipMemcpyAsync(c->h_copy, static_cast<const unsigned char *>(src) + off, n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::memcpy(static_cast<unsigned char *>(dst) + off, c->h_copy, n);
    }
    return ROPE_OK;
}

extern "C" void rope_destroy(rope_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    void *ptrs[] = {c->d_header, c->d_tris, c->d_verts, c->d_joint_fixed, c->d_joint_axes, c->d_PV, c->d_tq, c->d_t32,
                    c->d_cand, c->d_err, c->d_best_err, c->d_mvp, c->d_bounds, c->d_mask_lo, c->d_mask_hi, c->d_layer_of, c->d_layer_rep, c->d_layers, c->d_layer_sums, c->d_parent_of, c->d_parent_rep, c->d_parents, c->d_table, c->d_tcount, c->d_toff, c->d_tused, c->d_tgoff, c->d_tgval, c->d_ttotal, c->d_zero_total, c->d_tsums, c->d_terr, c->d_qitems, c->d_tile_tris, c->d_tile_tris_lo, c->d_qctr, c->d_touched, c->d_gtile, c->d_aabb, c->d_sums, c->d_best_idx, c->d_key,
                    c->d_depth, c->d_ids, c->d_cover, c->d_ftq, c->d_ftl, c->d_ftotal, c->d_fempty, c->d_ft32, c->d_vstage, c->d_empty[0], c->d_empty[1], c->d_empty[2], c->d_empty[3],
                    c->d_total[0], c->d_total[1], c->d_total[2], c->d_total[3], c->d_fts32, c->d_fflags, c->d_tg_total[0], c->d_tg_total[1], c->d_tg_total[2],
                    c->d_tg_total[3], c->d_tg_empty, c->d_frame_of, c->d_tg_t32c, c->d_tg_ltotal, c->d_tg_scores, c->d_tg_best, c->d_t32ts, c->d_tc,
                    c->s_ftq, c->s_ft32, c->s_fts32, c->s_fflags};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_err) (void)hipHostFree(c->h_err);
    if (c->h_cand) (void)hipHostFree(c->h_cand);
    if (c->h_frame_of) (void)hipHostFree(c->h_frame_of);
    if (c->h_vstage) (void)hipHostFree(c->h_vstage);
    if (c->h_vsums) (void)hipHostFree(c->h_vsums);
    if (c->h_copy) (void)hipHostFree(c->h_copy);
    if (c->h_copy2) (void)hipHostFree(c->h_copy2);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

#ifndef ROPE_BUILD_ID
#define ROPE_BUILD_ID "unknown"
#endif
extern "C" const char *rope_build_id(void) { return ROPE_BUILD_ID; }

extern "C" const char *rope_last_error(rope_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

// for the host-side stage machine (rope_predict.cpp), which sees the context only through the C ABI
void rope_set_error(rope_ctx *c, const std::string &msg) { if (c) c->err = msg; }

extern "C" int rope_set_robot(rope_ctx *c, const uint32_t *ml_header, int n_meshlets, const float *ml_verts,
                              int n_ml_verts, const uint32_t *ml_tris, int n_ml_tris, const int32_t *link_first,
                              int n_links, const double *joint_fixed, const double *joint_axes)
{
    if (!c) return ROPE_E_ARG;
    if (!ml_header || !ml_verts || !ml_tris || !link_first || !joint_fixed || !joint_axes) ARG_FAIL(c, "rope_set_robot: null pointer");
    if (n_links < 1 || n_links > ROPE_MAX_LINKS) ARG_FAIL(c, "rope_set_robot: n_links must be 1..6");
    if (n_meshlets < 1 || n_meshlets > MAX_MESHLETS) ARG_FAIL(c, "rope_set_robot: meshlet count out of range");
    if (link_first[0] != 0 || link_first[n_links] != n_meshlets) ARG_FAIL(c, "rope_set_robot: link_first does not span the meshlets");
    // validate every meshlet against the pools so that no kernel can index out of bounds
    std::vector<float> aabb(8 * (size_t)n_meshlets, 0.0f);
    for (int l = 0; l < n_links; l++) {
        if (link_first[l + 1] < link_first[l]) ARG_FAIL(c, "rope_set_robot: link_first not monotone");
        for (int m = link_first[l]; m < link_first[l + 1]; m++) {
            const uint32_t *h = ml_header + 8 * (size_t)m;
            uint32_t v0 = h[4], t0 = h[5], nv = h[6] & 0xFFFF, nt = h[6] >> 16;
            if (h[7] != (uint32_t)l) ARG_FAIL(c, "rope_set_robot: meshlet link id mismatch");
            if (nv < 1 || nv > MESHLET_MAX_VERTS || (size_t)v0 + nv > (size_t)n_ml_verts) ARG_FAIL(c, "rope_set_robot: meshlet vertex range");
            if (nt < 1 || nt > MESHLET_MAX_TRIS || (size_t)t0 + nt > (size_t)n_ml_tris) ARG_FAIL(c, "rope_set_robot: meshlet triangle range");
            for (uint32_t t = 0; t < nt; t++) {
                uint32_t p = ml_tris[t0 + t];
                if ((p & 0xFF) >= nv || ((p >> 8) & 0xFF) >= nv || ((p >> 16) & 0xFF) >= nv) ARG_FAIL(c, "rope_set_robot: triangle index outside its meshlet");
            }
            double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
            for (uint32_t v = 0; v < nv; v++)
                for (int k = 0; k < 3; k++) {
                    double x = ml_verts[3 * (size_t)(v0 + v) + k];
                    if (!std::isfinite(x)) ARG_FAIL(c, "rope_set_robot: non-finite vertex");
                    lo[k] = std::min(lo[k], x);
                    hi[k] = std::max(hi[k], x);
                }
            for (int k = 0; k < 3; k++) {
                // box centre and half extent, rounded outwards
                float ctr = (float)(0.5 * (lo[k] + hi[k]));
                float ext = (float)(std::max(hi[k] - (double)ctr, (double)ctr - lo[k]) * (1.0 + 1e-6) + 1e-7);
                aabb[8 * (size_t)m + k] = ctr;
                aabb[8 * (size_t)m + 4 + k] = ext;
            }
        }
    }
    // how far from the base origin a vertex can get: joints rotate, so a link's points stay within the sum of the joint
    // offsets up to it plus the link's own extent about its origin — whatever the joint angles
    double reach = 0.0, chain = 0.0;
    for (int l = 0; l < n_links; l++) {
        if (l > 0) chain += std::sqrt(joint_fixed[12 * (l - 1) + 3] * joint_fixed[12 * (l - 1) + 3] + joint_fixed[12 * (l - 1) + 7] * joint_fixed[12 * (l - 1) + 7] +
                                      joint_fixed[12 * (l - 1) + 11] * joint_fixed[12 * (l - 1) + 11]);
        double r = 0.0;
        for (int m = link_first[l]; m < link_first[l + 1]; m++) {
            double far2 = 0.0;
            for (int k = 0; k < 3; k++) {
                const double a = std::fabs((double)aabb[8 * (size_t)m + k]) + (double)aabb[8 * (size_t)m + 4 + k];
                far2 += a * a;
            }
            r = std::max(r, std::sqrt(far2));
        }
        reach = std::max(reach, chain + r);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, realloc_dev(&c->d_header, 8 * (size_t)n_meshlets));
    HIP_TRY(c, realloc_dev(&c->d_verts, 3 * (size_t)n_ml_verts));
    HIP_TRY(c, realloc_dev(&c->d_tris, (size_t)n_ml_tris));
    HIP_TRY(c, realloc_dev(&c->d_aabb, 8 * (size_t)n_meshlets));
    HIP_TRY(c, hipMemcpyAsync(c->d_aabb, aabb.data(), 32 * (size_t)n_meshlets, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_header, ml_header, 32 * (size_t)n_meshlets, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_verts, ml_verts, 12 * (size_t)n_ml_verts, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_tris, ml_tris, 4 * (size_t)n_ml_tris, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_joint_fixed, joint_fixed, 72 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_joint_axes, joint_axes, 18 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // Every copy and fill of this library goes through the context's OWN stream and is waited for there: the stream is a
    // non-blocking one, which the null stream's operations are not ordered with, and whether a null-stream hipMemset / hipMemcpy
    // has finished when it returns is the runtime's business (round 3: a first small batch on a fresh context sometimes scored
    // against stamps that a late hipMemset of their array had wiped)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->rp.ml_header = c->d_header;
    c->rp.ml_verts = c->d_verts;
    c->rp.ml_tris = c->d_tris;
    for (int l = 0; l <= ROPE_MAX_LINKS; l++) c->rp.link_first[l] = link_first[l < n_links ? l : n_links];
    c->rp.ml_aabb = c->d_aabb;
    c->rp.n_meshlets = n_meshlets;
    c->cap = 0;                                   // per-candidate buffers depend on the meshlet count
    c->C = 0;
    c->table_C = 0;
    c->n_links = n_links;
    c->n_meshlets = n_meshlets;
    c->reach = reach;
    c->have_robot = true;
    return ROPE_OK;
}

extern "C" int rope_set_camera(rope_ctx *c, const double *PV, int W, int H, double znear, double zfar)
{
    if (!c) return ROPE_E_ARG;
    if (!PV) ARG_FAIL(c, "rope_set_camera: null PV");
    if (W < 1 || H < 1 || W > 8192 || H > 8192) ARG_FAIL(c, "rope_set_camera: image size out of range");
    if (!(znear > 0.0) || !(zfar > znear)) ARG_FAIL(c, "rope_set_camera: need 0 < znear < zfar");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const bool resized = !c->have_camera || W != c->fp.W || H != c->fp.H;
    // the new geometry in locals: the context keeps describing the old image until every buffer of the new one exists
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H, n_tiles = tiles_x * tiles_y;
    if ((n_tiles + 31) / 32 > MAX_MASK_WORDS) ARG_FAIL(c, "rope_set_camera: too many tiles");
    if (resized) {
        // a failed allocation below must not leave old-sized (or freed) buffers behind a camera that looks usable:
        // the context has no camera, no target and no frames until this call has gone through
        c->have_camera = false;
        c->have_target = false;
        c->have_t32ts = false;
        c->n_frames = 0;
        c->n_targets = 0;
        c->staged_n = 0;                          // a staged set of the old size is not committed
        c->C = 0;
        c->cand_valid = c->results_valid = false;
        const size_t n = (size_t)W * H;
        HIP_TRY(c, realloc_dev(&c->d_tq, n));
        HIP_TRY(c, realloc_dev(&c->d_t32, n));
        HIP_TRY(c, realloc_dev(&c->d_t32ts, n));
        HIP_TRY(c, realloc_dev(&c->d_key, n));
        HIP_TRY(c, realloc_dev(&c->d_depth, n));
        HIP_TRY(c, realloc_dev(&c->d_ids, n));
        HIP_TRY(c, realloc_dev(&c->d_cover, n));
        for (int k = 0; k < 4; k++) {
            HIP_TRY(c, realloc_dev(&c->d_empty[k], (size_t)n_tiles * ROPE_SUM_WORDS));
            HIP_TRY(c, realloc_dev(&c->d_total[k], (size_t)ROPE_SUM_WORDS));
            c->empty_version[k] = 0;
        }
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_PV, PV, 16 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::memcpy(c->h_PV, PV, sizeof c->h_PV);
    c->fp.W = W; c->fp.H = H;
    c->fp.tiles_x = tiles_x;
    c->fp.tiles_y = tiles_y;
    c->fp.r0 = 0; c->fp.r1 = H - 1; c->fp.c0 = 0; c->fp.c1 = W - 1;
    c->fp.c_num = (float)(2.0 * znear * zfar);
    c->fp.c_sum = (float)(zfar + znear);
    c->fp.c_dif = (float)(zfar - znear);
    // per-candidate buffers are sized by the tile count (mask words; the queue's weights, one per tile)
    if (n_tiles != c->n_tiles) { c->mask_words = (n_tiles + 31) / 32; c->cap = 0; c->C = 0; c->cand_valid = c->results_valid = false; }
    c->n_tiles = n_tiles;
    c->table_C = 0;                               // a stored lookup table belongs to one camera
    c->have_camera = true;
    return ROPE_OK;
}

extern "C" int rope_set_target(rope_ctx *c, const uint64_t *tq, const float *t32, const uint8_t *link_flags)
{
    if (!c) return ROPE_E_ARG;
    if (!c->have_camera) ARG_FAIL(c, "rope_set_target: call rope_set_camera first");
    if (!tq || !link_flags) ARG_FAIL(c, "rope_set_target: null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = (size_t)c->fp.W * c->fp.H;
    int rc = copy_h2d_staged(c, c->d_tq, tq, n * sizeof(uint64_t));
    if (rc) return rc;
    if (t32) { rc = copy_h2d_staged(c, c->d_t32, t32, n * sizeof(float)); if (rc) return rc; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_t32 = (t32 != nullptr);
    c->have_t32ts = false;
    std::memcpy(c->lf.f, link_flags, 8);
    c->have_target = true;
    c->target_version++;
    return ROPE_OK;
}

extern "C" int rope_set_target_tsweep(rope_ctx *c, const float *t32_full)
{
    if (!c) return ROPE_E_ARG;
    if (!c->have_target) ARG_FAIL(c, "rope_set_target_tsweep: call rope_set_target first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (t32_full) {
        const int rc = copy_h2d_staged(c, c->d_t32ts, t32_full, (size_t)c->fp.W * c->fp.H * sizeof(float));
        if (rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->have_t32ts = (t32_full != nullptr);
    c->empty_version[ROPE_LOSS_TSWEEP] = 0;         // that loss's "nothing rendered" sums belong to the other plane
    return ROPE_OK;
}

// numpy's rint (round half to even) of a double in [0, 2^51) without a library call: adding and taking away 2^52 leaves the
// nearest integer, ties to even, under the default rounding mode (every step one IEEE operation: -ffp-contract=off, no fast-math).
// Beyond 2^51 the result is only used to be clipped to 2^39 - 1.
static inline double rint_nonneg(double x)
{
    double t = x + 4503599627370496.0;
    asm volatile("" : "+x"(t));                          // the sum is rounded to a double here: the two operations stay two operations
    return t - 4503599627370496.0;
}

// depth in metres -> Q32, as rope_pack_target documents it
static inline uint64_t q32_of_depth(double d)
{
    const double Q32 = 4294967296.0, top = 549755813887.0;          // 2^32, 2^39 - 1
    double q = (d > 0.0 && d <= 1.7976931348623157e308) ? rint_nonneg(d * Q32) : 0.0;     // NaN, inf, zero and negatives: no depth
    q = q < top ? q : top;
    return (uint64_t)q;
}

// Host only: float64 metres + link-mask bits -> the packed plane of rope_set_target.  Rounding is numpy's rint (round half to even).
extern "C" int rope_pack_target(const double *depth, const uint8_t *mask_bits, int64_t n, uint64_t *out)
{
    if (!depth || !out || n < 0) return ROPE_E_ARG;
    for (int64_t i = 0; i < n; i++) out[i] = q32_of_depth(depth[i]) | (mask_bits ? (uint64_t)mask_bits[i] << 40 : 0);
    return ROPE_OK;
}

// Host only: cv2.resize(img, (W/f, H/f)) with INTER_LINEAR for an EVEN integer factor f (Predictor._downsample,
// predict.py:378-381).  The source coordinate of every output sample falls exactly between the two central taps of its
// f x f block (f/2-1 and f/2), both with weight 1/2, so the general interpolation collapses to four taps per sample:
//   kind 0  uint8 : OpenCV's fixed-point path — horizontal pass with 11-bit weights, >> 4, vertical pass >> 16, + 2 >> 2
//   kind 1  float32, kind 2 float64 : (a*w + b*w) per row, then (top*w + bottom*w), w = 0.5 in the image's own type
// `channels` interleaved values per pixel; rows are `row_stride` BYTES apart (a strided view is fine).
extern "C" int rope_downsample_even(const void *src, int H, int W, int channels, int64_t row_stride, int f, int kind, void *dst)
{
    if (!src || !dst || H < 1 || W < 1 || channels < 1 || f < 2 || (f & 1) || H % f || W % f || kind < 0 || kind > 2) return ROPE_E_ARG;
    const int oh = H / f, ow = W / f, a = f / 2 - 1, b = f / 2;
    const char *base = (const char *)src;
    for (int y = 0; y < oh; y++) {
        const char *r0 = base + (int64_t)(y * f + a) * row_stride, *r1 = base + (int64_t)(y * f + b) * row_stride;
        for (int x = 0; x < ow; x++)
            for (int ch = 0; ch < channels; ch++) {
                const size_t ia = (size_t)(x * f + a) * channels + ch, ib = (size_t)(x * f + b) * channels + ch;
                const size_t o = ((size_t)y * ow + x) * channels + ch;
                if (kind == 0) {
                    const uint8_t *p0 = (const uint8_t *)r0, *p1 = (const uint8_t *)r1;
                    const int64_t t = ((int64_t)p0[ia] * 1024 + (int64_t)p0[ib] * 1024) >> 4, u = ((int64_t)p1[ia] * 1024 + (int64_t)p1[ib] * 1024) >> 4;
                    int64_t v = (((1024 * t) >> 16) + ((1024 * u) >> 16) + 2) >> 2;
                    ((uint8_t *)dst)[o] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                } else if (kind == 1) {
                    const float *p0 = (const float *)r0, *p1 = (const float *)r1;
                    const float top = p0[ia] * 0.5f + p0[ib] * 0.5f, bot = p1[ia] * 0.5f + p1[ib] * 0.5f;
                    ((float *)dst)[o] = top * 0.5f + bot * 0.5f;
                } else {
                    const double *p0 = (const double *)r0, *p1 = (const double *)r1;
                    const double top = p0[ia] * 0.5 + p0[ib] * 0.5, bot = p1[ia] * 0.5 + p1[ib] * 0.5;
                    ((double *)dst)[o] = top * 0.5 + bot * 0.5;
                }
            }
    }
    return ROPE_OK;
}

// Host only: one frame of the synthetic path from the camera's arrays to what rope_set_target(s) takes, in one pass — the
// down-sampling of Predictor._downsample (predict.py:378-381: cv2.resize INTER_LINEAR; the taps and roundings of
// rope_downsample_even), the link masks read off channel 0 of the colour render and the lookup depth of _loadSynthetic
// (predict.py:445-469), the per-link flags and the packing of _load_target (predict.py:397-413, rope_pack_target).
extern "C" int rope_prepare_synthetic(const uint8_t *color, int64_t color_stride, const void *depth, int depth_kind, int64_t depth_stride,
                                      int H0, int W0, int f, const int32_t *link_blue, int n_links, int n_lookup_links,
                                      uint64_t *tq, float *lookup_f32, double *tgt_depth, uint8_t *flags)
{
    if (!color || !depth || !link_blue || !tq || !lookup_f32 || !flags) return ROPE_E_ARG;
    if (H0 < 1 || W0 < 1 || f < 1 || (f > 1 && (f & 1)) || H0 % f || W0 % f || (depth_kind != 1 && depth_kind != 2)) return ROPE_E_ARG;
    if (n_links < 1 || n_links > ROPE_MAX_LINKS || n_lookup_links < 0 || n_lookup_links > n_links) return ROPE_E_ARG;
    const int H = H0 / f, W = W0 / f, a = f / 2 - 1, b = f / 2;
    // what a channel-0 value says about a pixel, looked up instead of compared link by link: the links whose colour it is (two
    // links may share one), and whether one of them is a lookup link; the counts are kept per value and folded per link at the end
    uint8_t bits_of[256] = {}, hit_of[256] = {};
    int64_t n_val[256] = {}, n_val_depth[256] = {};
    for (int l = 0; l < n_links; l++)
        if (link_blue[l] >= 0 && link_blue[l] <= 255) {
            bits_of[link_blue[l]] |= (uint8_t)(1u << l);
            if (l < n_lookup_links) hit_of[link_blue[l]] = 1;
        }
    int cur = -1;                                                  // counts are kept for the run of equal values and flushed when it ends:
    int64_t run = 0, run_depth = 0;                                // a colour-coded frame is long runs, and a counter in memory per pixel is a chain of store-to-load stalls
    for (int y = 0; y < H; y++) {
        const int ya = f > 1 ? y * f + a : y, yb = f > 1 ? y * f + b : y;
        const uint8_t *c0 = color + (int64_t)ya * color_stride, *c1 = color + (int64_t)yb * color_stride;
        const char *d0 = (const char *)depth + (int64_t)ya * depth_stride, *d1 = (const char *)depth + (int64_t)yb * depth_stride;
        for (int x = 0; x < W; x++) {
            const size_t xa = f > 1 ? (size_t)(x * f + a) : (size_t)x, xb = f > 1 ? (size_t)(x * f + b) : (size_t)x;
            int blue;
            double d;
            if (f > 1) {
                // OpenCV's fixed-point path for uint8 (rope_downsample_even, kind 0); channel 0 of three interleaved
                const int64_t t = ((int64_t)c0[3 * xa] * 1024 + (int64_t)c0[3 * xb] * 1024) >> 4, u = ((int64_t)c1[3 * xa] * 1024 + (int64_t)c1[3 * xb] * 1024) >> 4;
                const int64_t v = (((1024 * t) >> 16) + ((1024 * u) >> 16) + 2) >> 2;
                blue = (int)(v < 0 ? 0 : (v > 255 ? 255 : v));
                if (depth_kind == 1) {
                    const float *p0 = (const float *)d0, *p1 = (const float *)d1;
                    const float tp = p0[xa] * 0.5f + p0[xb] * 0.5f, bt = p1[xa] * 0.5f + p1[xb] * 0.5f;
                    d = (double)(tp * 0.5f + bt * 0.5f);
                } else {
                    const double *p0 = (const double *)d0, *p1 = (const double *)d1;
                    const double tp = p0[xa] * 0.5 + p0[xb] * 0.5, bt = p1[xa] * 0.5 + p1[xb] * 0.5;
                    d = tp * 0.5 + bt * 0.5;
                }
            } else {
                blue = c0[3 * xa];
                d = depth_kind == 1 ? (double)((const float *)d0)[xa] : ((const double *)d0)[xa];
            }
            const unsigned bits = bits_of[blue];
            if (blue != cur) {
                if (cur >= 0) { n_val[cur] += run; n_val_depth[cur] += run_depth; }
                cur = blue; run = 0; run_depth = 0;
            }
            run++;
            run_depth += (d != 0.0);                               // NaN counts, as `depth != 0` does
            const size_t o = (size_t)y * W + x;
            if (tgt_depth) tgt_depth[o] = d;
            lookup_f32[o] = (float)(d * (hit_of[blue] ? 1.0 : 0.0));     // target_depth * hit (predict.py:449-454): NaN stays NaN
            tq[o] = q32_of_depth(d) | ((uint64_t)bits << 40);
        }
    }
    if (cur >= 0) { n_val[cur] += run; n_val_depth[cur] += run_depth; }
    std::memset(flags, 0, 8);
    for (int l = 0; l < n_links; l++) {
        if (link_blue[l] < 0 || link_blue[l] > 255) continue;
        const int64_t n_mask = n_val[link_blue[l]], n_depth = n_val_depth[link_blue[l]];
        if (n_mask > 0) {                                          // np.sum(mask) > 0 (predict.py:465)
            flags[l] |= 1;
            if ((double)n_depth > 0.05 * (double)n_mask) flags[l] |= 2;      // predict.py:495, a fact of the target alone
        }
    }
    return ROPE_OK;
}

// Host only: camera pose + pinhole intrinsics -> P·V, the matrix rope_set_camera takes.  Every step is one IEEE double operation in
// the written order (the library is built with -ffp-contract=off), sin / cos from libm.
//   pose   [x, y, z, a3, a4, a5] as Renderer.setCameraPose takes it (render.py:107-111): roll = a4 + pi/2, pitch = a3, yaw = a5,
//          camera-to-world R = Rz(yaw)·Ry(pitch)·Rx(roll) by the element formulas of angToPoseArr (render_utils.py:56-85),
//          t = (x, y, z); the camera looks along its -Z, +Y up (OpenGL); V = the rigid inverse [R^T | -R^T t]
//   P      pyrender 0.1.45's IntrinsicsCamera.get_projection_matrix (projection.py:161-169): P00 = 2fx/W, P11 = 2fy/H,
//          P02 = 1 - 2cx/W, P12 = 2cy/H - 1, P22 = (f+n)/(n-f), P23 = 2fn/(n-f), P32 = -1
extern "C" int rope_camera_matrix(const double *pose, double fx, double fy, double cx, double cy, int W, int H, double znear, double zfar,
                                  double *PV)
{
    if (!pose || !PV || W < 1 || H < 1 || !(znear > 0.0) || !(zfar > znear)) return ROPE_E_ARG;
    const double yaw = pose[5], pitch = pose[3], roll = pose[4] + 3.141592653589793 / 2;
    const double c0 = std::cos(yaw), c1 = std::cos(pitch), c2 = std::cos(roll), s0 = std::sin(yaw), s1 = std::sin(pitch), s2 = std::sin(roll);
    double R[3][3];
    R[0][0] = c0 * c1;
    R[1][0] = c1 * s0;
    R[2][0] = -1 * s1;
    R[0][1] = c0 * s1 * s2 - c2 * s0;
    R[1][1] = c0 * c2 + (s0 * s1) * s2;
    R[2][1] = c1 * s2;
    R[0][2] = s0 * s2 + c0 * c2 * s1;
    R[1][2] = c2 * s0 * s1 - c0 * s2;
    R[2][2] = c1 * c2;
    double V[4][4] = {};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) V[i][j] = R[j][i];
        V[i][3] = -((R[0][i] * pose[0] + R[1][i] * pose[1]) + R[2][i] * pose[2]);
    }
    V[3][3] = 1.0;
    const double w = (double)W, h = (double)H;
    const double P00 = 2.0 * fx / w, P11 = 2.0 * fy / h, P02 = 1.0 - 2.0 * cx / w, P12 = 2.0 * cy / h - 1.0;
    const double P22 = (zfar + znear) / (znear - zfar), P23 = (2.0 * zfar * znear) / (znear - zfar);
    for (int j = 0; j < 4; j++) {
        PV[j] = P00 * V[0][j] + P02 * V[2][j];
        PV[4 + j] = P11 * V[1][j] + P12 * V[2][j];
        PV[8 + j] = P22 * V[2][j] + P23 * V[3][j];
        PV[12 + j] = -V[2][j];
    }
    for (int k = 0; k < 16; k++)
        if (!std::isfinite(PV[k])) return ROPE_E_ARG;
    return ROPE_OK;
}

// Host only: the Lookup stage's pose grid in the reference's order (lookup.py:39-66).  divisions[j] >= 1: joint j takes that many
// samples between its limits (np.linspace: lo + i * step, the last one hi itself; a single sample is lo), clipped to
// LOOKUP_MAX_DIV_PER_LINK = 200 (constants.py:30); divisions[j] <= 0: joint j is not part of the grid and stays 0.  Joint 0 varies
// fastest.  Returns the number of rows (the product of the sample counts); writes them when `out` is given and `capacity` rows fit
// (ROPE_E_ARG otherwise).
extern "C" int64_t rope_lookup_grid(const double *limits, const int32_t *divisions, double *out, int64_t capacity)
{
    if (!limits || !divisions) return ROPE_E_ARG;
    int64_t div[6], n = 1;
    for (int j = 0; j < 6; j++) {
        div[j] = divisions[j] < 1 ? 1 : (divisions[j] > 200 ? 200 : divisions[j]);
        n *= div[j];
    }
    if (!out) return n;
    if (capacity < n) return ROPE_E_ARG;
    int64_t repeat = 1;
    for (int j = 0; j < 6; j++) {
        const double lo = limits[2 * j], hi = limits[2 * j + 1], dv = (double)(div[j] - 1), delta = hi - lo;
        const double step = div[j] > 1 ? delta / dv : 0.0;
        for (int64_t r = 0; r < n; r++) {
            const int64_t i = (r / repeat) % div[j];
            double v = 0.0;                                         // not part of the grid: np.zeros
            if (divisions[j] >= 1) {
                if (div[j] == 1) v = lo;
                else if (i == div[j] - 1) v = hi;
                else v = step == 0.0 ? ((double)i / dv) * delta + lo : (double)i * step + lo;
            }
            out[6 * r + j] = v;
        }
        repeat *= div[j];
    }
    return n;
}

// Host only: the divisions of the pose grid Crop renders to find the image bounds of the first `num_links` links
// (robotpose/crop.py:114-146), in rope_lookup_grid's convention — feed them to rope_lookup_grid for the poses.  Weights 6:3:3:0:1
// over the joints S L U R B (constants.py:19: CROP_RENDER_WEIGHTING), a budget of 20 s at the reference's cost model of
// pixels x 1.2e-8 + 0.002 s per render (crop.py:121-123), at most 50 samples per joint; joints beyond the rendered links sit at
// their lower limit (np.linspace(lo, hi, 1)), R and T are not part of the grid (CROP_VARYING = 'SLUB') and stay 0.
extern "C" int rope_crop_divisions(int64_t n_pixels, int num_links, int32_t *divisions)
{
    if (!divisions || n_pixels < 1 || num_links < 2 || num_links > 6) return ROPE_E_ARG;
    const double weighting[6] = {6, 3, 3, 0, 1, 0};
    const int n = num_links - 1;
    double w[6], sum = 0.0, prod = 1.0;
    int nz = 0;
    for (int j = 0; j < n; j++) sum += weighting[j];
    for (int j = 0; j < n; j++) {
        w[j] = weighting[j] / sum;
        if (w[j] != 0.0) { prod *= w[j]; nz++; }
    }
    const double num_poses = 20.0 / ((double)n_pixels * 1.2 * 1.0e-8 + .002);
    const double scale = std::pow(num_poses / prod, 1.0 / (double)nz);
    const bool varying[6] = {true, true, true, false, true, false};          // S L U B
    for (int j = 0; j < 6; j++) {
        int d = 1;
        if (j < n) {
            double b = w[j] * scale;
            if (b < 1.0) b = 1.0;
            if (b > 50.0) b = 50.0;
            d = (int)b;
        }
        divisions[j] = varying[j] ? d : 0;
    }
    return ROPE_OK;
}

// cv2.dilate / cv2.erode of a binary image with a ones(k, k) kernel, default anchor (k / 2, k / 2), a border that never wins
// (predict.py:428,437): separable, rows then columns; `grow` true: a pixel is set when any pixel of its window is (outside counts
// as clear), false: when all are (outside counts as set)
static void box_morph(const uint8_t *src, uint8_t *dst, uint8_t *tmp, int H, int W, int k, bool grow)
{
    const int a = k / 2, b = k - 1 - a;                  // the window of x reaches from x - a to x + b
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int lo = x - a < 0 ? 0 : x - a, hi = x + b > W - 1 ? W - 1 : x + b;
            uint8_t v = grow ? 0 : 1;
            for (int i = lo; i <= hi; i++) v = grow ? (uint8_t)(v | src[(size_t)y * W + i]) : (uint8_t)(v & src[(size_t)y * W + i]);
            tmp[(size_t)y * W + x] = v;
        }
    for (int y = 0; y < H; y++) {
        const int lo = y - a < 0 ? 0 : y - a, hi = y + b > H - 1 ? H - 1 : y + b;
        for (int x = 0; x < W; x++) {
            uint8_t v = grow ? 0 : 1;
            for (int i = lo; i <= hi; i++) v = grow ? (uint8_t)(v | tmp[(size_t)i * W + x]) : (uint8_t)(v & tmp[(size_t)i * W + x]);
            dst[(size_t)y * W + x] = v;
        }
    }
}

// Host only: one frame of the SEGMENTATION path from the segmenter's instance masks to what rope_set_target(s) takes — the merge of
// a class's instances (_reorganize_by_link, predict.py:383-395), the body mask erode7(dilate8(sum of the masks)) that zeroes the
// depth outside the robot, once over all classes and once over the lookup links (predict.py:419-438), the depth's own
// down-sampling (predict.py:378-381), the per-link flags and the packing of _load_target (predict.py:397-413).
//   masks     H x W x K bytes (non-zero = inside instance k), as the segmenter returns them;  link_of: K link indices (0..n_links-1),
//             or -1 for an instance of a class that is not one of the rendered links (it still counts for the body mask)
extern "C" int rope_prepare_segmented(const void *depth, int depth_kind, int64_t depth_stride, int H0, int W0, int f, const uint8_t *masks,
                                      int K, const int32_t *link_of, int n_links, int n_lookup_links, uint64_t *tq, float *lookup_f32,
                                      double *tgt_depth, uint8_t *flags)
{
    if (!depth || (K > 0 && (!masks || !link_of)) || !tq || !lookup_f32 || !flags || K < 0) return ROPE_E_ARG;
    if (H0 < 1 || W0 < 1 || f < 1 || (f > 1 && (f & 1)) || H0 % f || W0 % f || (depth_kind != 1 && depth_kind != 2)) return ROPE_E_ARG;
    if (n_links < 1 || n_links > ROPE_MAX_LINKS || n_lookup_links < 0 || n_lookup_links > n_links) return ROPE_E_ARG;
    for (int k = 0; k < K; k++)
        if (link_of[k] < -1 || link_of[k] >= n_links) return ROPE_E_ARG;
    const int H = H0 / f, W = W0 / f, a = f / 2 - 1, b = f / 2;
    const size_t n = (size_t)H * W;
    std::vector<uint8_t> bits(n, 0), any(n, 0), look(n, 0), body(n), body_look(n), t1(n), t2(n);
    for (size_t i = 0; i < n; i++) {
        const uint8_t *m = masks + i * (size_t)K;
        for (int k = 0; k < K; k++)
            if (m[k]) {
                any[i] = 1;
                if (link_of[k] >= 0) {
                    bits[i] |= (uint8_t)(1u << link_of[k]);
                    if (link_of[k] < n_lookup_links) look[i] = 1;
                }
            }
    }
    box_morph(any.data(), t1.data(), t2.data(), H, W, 8, true);
    box_morph(t1.data(), body.data(), t2.data(), H, W, 7, false);
    box_morph(look.data(), t1.data(), t2.data(), H, W, 8, true);
    box_morph(t1.data(), body_look.data(), t2.data(), H, W, 7, false);
    int64_t n_mask[ROPE_MAX_LINKS] = {}, n_depth[ROPE_MAX_LINKS] = {};
    bool present[ROPE_MAX_LINKS] = {};
    for (int k = 0; k < K; k++)
        if (link_of[k] >= 0) present[link_of[k]] = true;             // a class that was detected has a mask, however empty (predict.py:408-413)
    for (int y = 0; y < H; y++) {
        const int ya = f > 1 ? y * f + a : y, yb = f > 1 ? y * f + b : y;
        const char *d0 = (const char *)depth + (int64_t)ya * depth_stride, *d1 = (const char *)depth + (int64_t)yb * depth_stride;
        for (int x = 0; x < W; x++) {
            const size_t xa = f > 1 ? (size_t)(x * f + a) : (size_t)x, xb = f > 1 ? (size_t)(x * f + b) : (size_t)x, o = (size_t)y * W + x;
            double d;
            if (f > 1) {
                if (depth_kind == 1) {
                    const float *p0 = (const float *)d0, *p1 = (const float *)d1;
                    const float tp = p0[xa] * 0.5f + p0[xb] * 0.5f, bt = p1[xa] * 0.5f + p1[xb] * 0.5f;
                    d = (double)(tp * 0.5f + bt * 0.5f);
                } else {
                    const double *p0 = (const double *)d0, *p1 = (const double *)d1;
                    const double tp = p0[xa] * 0.5 + p0[xb] * 0.5, bt = p1[xa] * 0.5 + p1[xb] * 0.5;
                    d = tp * 0.5 + bt * 0.5;
                }
            } else {
                d = depth_kind == 1 ? (double)((const float *)d0)[xa] : ((const double *)d0)[xa];
            }
            d = d * (body[o] ? 1.0 : 0.0);                           // target_depth *= body (predict.py:429)
            const double dl = d * (body_look[o] ? 1.0 : 0.0);        // lookup_depth = the copy, *= the lookup links' body (predict.py:432-438)
            for (int l = 0; l < n_links; l++)
                if ((bits[o] >> l) & 1) {
                    n_mask[l]++;
                    if (d != 0.0) n_depth[l]++;
                }
            if (tgt_depth) tgt_depth[o] = d;
            lookup_f32[o] = (float)dl;
            tq[o] = q32_of_depth(d) | ((uint64_t)bits[o] << 40);
        }
    }
    std::memset(flags, 0, 8);
    for (int l = 0; l < n_links; l++)
        if (present[l]) {
            flags[l] |= 1;
            if ((double)n_depth[l] > 0.05 * (double)n_mask[l]) flags[l] |= 2;
        }
    return ROPE_OK;
}

// Can a vertex of the robot get behind the near plane of camera PV (P·V, row-major doubles)?  z + w is affine in the world
// position; over the ball of radius `reach` about the base origin it is at least its value at the origin minus |gradient|
// times the radius.  Conservative with a centimetre to spare: "no" means the kernels without the clipping code draw every
// triangle exactly as the clipping ones would.
static bool near_plane_in_reach(const rope_ctx *c, const double *PV)
{
    const double g[3] = {PV[8] + PV[12], PV[9] + PV[13], PV[10] + PV[14]}, h = PV[11] + PV[15];
    const double gn = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    return !(h - gn * (c->reach + 0.01) > 0.0);                  // NaN counts as "in reach"
}

// the raster kernels that can clip, for this context's camera (or, on the camera-pose path, the current call's cameras)
static bool use_clip(const rope_ctx *c, bool views = false)
{
    if (c->strategy & STRATEGY_CLIP_KERNELS) return true;
    return views ? c->clip_views : near_plane_in_reach(c, c->h_PV);
}
static int ensure_capacity(rope_ctx *c, int C)
{
    if (C <= c->cap) return ROPE_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int cap = C < 64 ? 64 : C;
    HIP_TRY(c, realloc_dev(&c->d_cand, 6 * (size_t)cap));
    HIP_TRY(c, realloc_dev(&c->d_err, (size_t)cap + 2));      // errors, then best error and best index
    if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; }
    HIP_TRY(c, hipHostMalloc((void **)&c->h_stage, ((size_t)cap * 6 + 2) * sizeof(double), hipHostMallocDefault));
    HIP_TRY(c, realloc_dev(&c->d_mvp, (size_t)cap * ROPE_MAX_LINKS * 16));
    if (!c->have_robot || !c->have_camera) ARG_FAIL(c, "candidates: robot and camera must be set first");
    HIP_TRY(c, realloc_dev(&c->d_bounds, (size_t)cap * c->n_meshlets));
    HIP_TRY(c, realloc_dev(&c->d_mask_lo, (size_t)cap * c->mask_words));
    HIP_TRY(c, realloc_dev(&c->d_mask_hi, (size_t)cap * c->mask_words));
    HIP_TRY(c, realloc_dev(&c->d_layer_of, (size_t)cap));
    HIP_TRY(c, realloc_dev(&c->d_sums, (size_t)cap * ROPE_SUM_WORDS));
    // raster queue: QUEUE_CLASSES segments (pairs by weight, heaviest first) and the per-(candidate, tile) weights, when the frame
    // has few enough tiles for the weights and the segments stay small; otherwise one segment, pairs in candidate order
    const size_t seg = (size_t)cap * c->mask_words * 32;
    c->q_weighted = c->n_tiles <= QUEUE_WEIGHT_TILES && 2 * seg * QUEUE_CLASSES * sizeof(uint32_t) <= ((size_t)512 << 20);
    c->q_segment = c->q_weighted ? seg : 0;
    HIP_TRY(c, realloc_dev(&c->d_qitems, c->q_weighted ? 2 * seg * QUEUE_CLASSES : seg));      // weighted: the scoring queue, then the layer queue
    if (c->d_tile_tris) { (void)hipFree(c->d_tile_tris); c->d_tile_tris = nullptr; }
    if (c->d_tile_tris_lo) { (void)hipFree(c->d_tile_tris_lo); c->d_tile_tris_lo = nullptr; }
    if (c->q_weighted) {
        HIP_TRY(c, realloc_dev(&c->d_tile_tris, (size_t)cap * c->n_tiles));
        HIP_TRY(c, realloc_dev(&c->d_tile_tris_lo, (size_t)cap * c->n_tiles));
    }
    c->cap = cap;
    return ROPE_OK;
}

static int upload_candidates(rope_ctx *c, const double *cand, int C, bool group, const int32_t *frame_of = nullptr);

extern "C" int rope_candidates_upload(rope_ctx *c, const double *cand, int C)
{
    if (!c) return ROPE_E_ARG;
    return upload_candidates(c, cand, C, true);
}

// frame_of (rope_eval_targets): the frame every row is scored against; rows only share a layer inside one frame (a layer
// carries loss sums, and those belong to a target)
static int upload_candidates(rope_ctx *c, const double *cand, int C, bool group, const int32_t *frame_of)
{
    if (!cand || C < 1 || C > 65535) ARG_FAIL(c, "rope_candidates_upload: need 1 <= C <= 65535");
    for (size_t i = 0; i < 6 * (size_t)C; i++)
        if (!std::isfinite(cand[i]) || std::fabs(cand[i]) > 1.0e4) ARG_FAIL(c, "rope_candidates_upload: joint angle not finite or |q| > 1e4 rad");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_capacity(c, C);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // the staging buffer may still feed an earlier copy
    if (C <= (frame_of ? rope_ctx::TARGET_HOST_ROWS : rope_ctx::HOST_ERR_ROWS)) {            // nothing is in flight (synchronised above): the kernels of the next pass read it in place
        std::memcpy(c->h_cand, cand, 6 * (size_t)C * sizeof(double));
        c->cand_dev = c->d_cand_host;
        if (frame_of) { std::memcpy(c->h_frame_of, frame_of, (size_t)C * sizeof(int32_t)); c->frame_of_dev = c->d_frame_of_host; }
    } else {
        std::memcpy(c->h_stage, cand, 6 * (size_t)C * sizeof(double));
        HIP_TRY(c, hipMemcpyAsync(c->d_cand, c->h_stage, 6 * (size_t)C * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->cand_dev = c->d_cand;
        if (frame_of) {
            if (C > c->frame_of_cap) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                c->frame_of_cap = 0;
                HIP_TRY(c, realloc_dev(&c->d_frame_of, (size_t)std::max(C, 1024)));
                c->frame_of_cap = std::max(C, 1024);
            }
            // pageable source: the copy has read it when the call returns
            HIP_TRY(c, hipMemcpyAsync(c->d_frame_of, frame_of, (size_t)C * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            c->frame_of_dev = c->d_frame_of;
        }
    }
    if (!frame_of) c->frame_of_dev = nullptr;
    // group candidates whose first two joint angles are bit-identical: their base_link, link_1_s and
    // link_2_l transforms are the same bits, so those links are rasterised once per group (a "layer")
    c->n_layers = C;                               // "no sharing" unless the grouping below finds some
    if (group && C >= 8) {
        struct Key { uint64_t a, b; int idx, frame; };
        std::vector<Key> keys((size_t)C);
        for (int i = 0; i < C; i++) {
            std::memcpy(&keys[i].a, &cand[6 * (size_t)i], 8);
            std::memcpy(&keys[i].b, &cand[6 * (size_t)i + 1], 8);
            keys[i].idx = i;
            keys[i].frame = frame_of ? frame_of[i] : 0;
        }
        std::sort(keys.begin(), keys.end(), [](const Key &x, const Key &y) {
            return x.frame != y.frame ? x.frame < y.frame : (x.a != y.a ? x.a < y.a : (x.b != y.b ? x.b < y.b : x.idx < y.idx));
        });
        std::vector<int32_t> layer_of((size_t)C), layer_rep, parent_of, parent_rep;
        for (int i = 0; i < C; i++) {
            const bool new_frame = i == 0 || keys[i].frame != keys[i - 1].frame;
            if (new_frame || keys[i].a != keys[i - 1].a || keys[i].b != keys[i - 1].b) {
                layer_rep.push_back(keys[i].idx);
                // layers come out sorted by q0: a new q0 opens a new parent, represented by this layer's own candidate
                if (new_frame || keys[i].a != keys[i - 1].a) parent_rep.push_back(keys[i].idx);
                parent_of.push_back((int32_t)parent_rep.size() - 1);
            }
            layer_of[keys[i].idx] = (int32_t)layer_rep.size() - 1;
        }
        c->n_layers = (int)layer_rep.size();
        c->n_parents = (int)parent_rep.size();
        if (c->n_layers * 4 <= C) {                // same rule as want_layers(): only then are the arrays read
            if (c->n_layers > c->layer_rep_cap) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                HIP_TRY(c, realloc_dev(&c->d_layer_rep, (size_t)c->n_layers));
                c->layer_rep_cap = c->n_layers;
            }
            HIP_TRY(c, hipMemcpyAsync(c->d_layer_of, layer_of.data(), (size_t)C * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->d_layer_rep, layer_rep.data(), layer_rep.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            if (c->n_parents * 4 <= c->n_layers) {     // same rule one level up (want_parents())
                if (c->n_layers > c->parent_idx_cap) {
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    HIP_TRY(c, realloc_dev(&c->d_parent_of, (size_t)c->n_layers));
                    HIP_TRY(c, realloc_dev(&c->d_parent_rep, (size_t)c->n_layers));
                    c->parent_idx_cap = c->n_layers;
                }
                HIP_TRY(c, hipMemcpyAsync(c->d_parent_of, parent_of.data(), parent_of.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipMemcpyAsync(c->d_parent_rep, parent_rep.data(), parent_rep.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            }
            HIP_TRY(c, hipStreamSynchronize(c->stream));   // layer_of / layer_rep are stack-local
        }
    }
    c->C = C;
    c->cand_valid = true;
    c->results_valid = false;
    return ROPE_OK;
}

static int check_eval_args(rope_ctx *c, int n_render, int loss, const int32_t *crop, FrameParams &fp, double &n_pix)
{
    if (!c->have_robot || !c->have_camera) ARG_FAIL(c, "eval: robot and camera must be set first");
    if (c->C < 1 || !c->cand_valid) ARG_FAIL(c, "eval: no candidates resident (upload them again after rope_eval_views)");
    if (n_render < 1 || n_render > c->n_links) ARG_FAIL(c, "eval: n_render out of range");
    if (loss < 0 || loss > 3) ARG_FAIL(c, "eval: unknown loss");
    if (!c->have_target) ARG_FAIL(c, "eval: no target set");
    if (loss == ROPE_LOSS_LOOKUP && !c->have_t32) ARG_FAIL(c, "eval: this loss needs the float32 target plane");
    if (loss == ROPE_LOSS_TSWEEP && !c->have_t32 && !c->have_t32ts) ARG_FAIL(c, "eval: this loss needs a float32 target plane");
    fp = c->fp;
    n_pix = (double)fp.W * fp.H;
    if (loss == ROPE_LOSS_LOOKUP) {
        if (!crop) ARG_FAIL(c, "eval: lookup loss needs a crop");
        if (crop[0] < 0 || crop[1] >= fp.H || crop[0] > crop[1] || crop[2] < 0 || crop[3] >= fp.W || crop[2] > crop[3]) ARG_FAIL(c, "eval: crop outside the image");
        fp.r0 = crop[0]; fp.r1 = crop[1]; fp.c0 = crop[2]; fp.c1 = crop[3];
        n_pix = (double)(crop[1] - crop[0] + 1) * (double)(crop[3] - crop[2] + 1);
    }
    return ROPE_OK;
}

// the float32 plane a loss reads: TensorSweep takes the whole target depth when it was given (rope_set_target_tsweep)
static const float *t32_plane(const rope_ctx *c, int loss) { return (loss == ROPE_LOSS_TSWEEP && c->have_t32ts) ? c->d_t32ts : c->d_t32; }

static int ensure_empty(rope_ctx *c, int loss, const FrameParams &fp)
{
    const int cr[4] = {fp.r0, fp.r1, fp.c0, fp.c1};
    if (c->empty_version[loss] == c->target_version && std::memcmp(cr, c->empty_crop[loss], sizeof cr) == 0) return ROPE_OK;
    HIP_TRY(c, launch_empty(loss, c->stream, fp, c->d_tq, t32_plane(c, loss), nullptr, c->d_empty[loss], c->d_total[loss]));
    c->empty_version[loss] = c->target_version;
    std::memcpy(c->empty_crop[loss], cr, sizeof cr);
    return ROPE_OK;
}

// Layers pay off when many candidates share one upstream pose (lookup grids, sweeps of U).
static bool want_layers(const rope_ctx *c) { return c->n_layers * 4 <= c->C; }

// A small batch whose candidates share one or two upstream poses (a sweep of the third joint) would draw links 0-2 in a layer
// launch of a handful of workgroups — 59 000 triangles each, the rest of the chip idle (100 us at 160x90) — before the scoring
// launch: the split path draws all six links per candidate over many workgroups instead and is more than twice as fast there.
static bool layers_pay(const rope_ctx *c)
{
    const int busy_tiles = std::max(std::min(2, c->n_tiles), c->n_tiles / 3);
    return c->C > 256 || c->n_layers * busy_tiles >= c->layer_min_wg;
}

static int ensure_layers(rope_ctx *c)
{
    size_t need = (size_t)c->n_layers * c->n_tiles * (TILE_W * TILE_H);
    if (need <= c->layers_cap) return ROPE_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, realloc_dev(&c->d_layers, need));
    HIP_TRY(c, realloc_dev(&c->d_layer_sums, (size_t)c->n_layers * c->n_tiles * ROPE_SUM_WORDS));
    c->layers_cap = need;
    return ROPE_OK;
}

static bool want_parents(const rope_ctx *c) { return c->n_parents * 4 <= c->n_layers; }
static uint32_t *queue_weights(const rope_ctx *c);
// Second level of sharing (links 0-1 once per distinct q0): pays when the layers are a launch of one workgroup per (layer, tile)
// pair — not when they go through the weighted queue, which draws links 0-2 per layer faster than the extra launch of a
// hundred busy workgroups plus the merge of every layer tile with its parent's (0.264 against 0.283 ms on the bench workload).
static bool use_parents(const rope_ctx *c, int n_shared)
{
    if (n_shared != 3 || !want_parents(c) || (c->strategy & STRATEGY_NO_PARENTS)) return false;
    return !(queue_weights(c) && !(c->strategy & STRATEGY_NO_QUEUE));
}

// The shared links of every layer into c->d_layers (+ their loss sums when `la` carries targets and layer_sums).
// Two levels when many layers share their first joint angle: links 0-1 once per distinct q0, then link 2 per layer
// merged with its parent's tile.
static int enqueue_layers(rope_ctx *c, RasterArgs la, int loss, int n_shared, const FrameParams &fp)
{
    la.cand_of_row = c->d_layer_rep; la.layers = c->d_layers;
    if (use_parents(c, n_shared)) {
        const size_t need = (size_t)c->n_parents * c->n_tiles * (TILE_W * TILE_H);
        if (need > c->parents_cap) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, realloc_dev(&c->d_parents, need));
            c->parents_cap = need;
        }
        RasterArgs pa = la;
        pa.l_begin = 0; pa.l_end = 2; pa.cand_of_row = c->d_parent_rep; pa.layers = c->d_parents; pa.layer_sums = nullptr;
        HIP_TRY(c, launch_raster(MODE_LAYER, loss, c->n_parents, c->stream, fp, c->rp, pa, use_clip(c)));
        la.l_begin = 2; la.l_end = 3; la.base_layers = c->d_parents; la.base_of_row = c->d_parent_of; la.base_rep = c->d_parent_rep;
    } else {
        la.l_begin = 0; la.l_end = n_shared;
    }
    if (queue_weights(c) && !(c->strategy & STRATEGY_NO_QUEUE)) {
        // the (layer, tile) pairs that have anything to draw, heaviest first, to a grid that just fills the chip: of the one-
        // workgroup-per-pair launch's 266 us the last 100 ran with a third of the chip (a 140 us pair that started at 150 us)
        HIP_TRY(c, launch_layer_queue(loss, c->n_layers, 2 * c->n_cu, c->stream, fp, c->rp, la, c->d_qitems + QUEUE_CLASSES * c->q_segment,
                                      c->q_segment, c->d_qctr + QUEUE_COUNTERS, c->d_tile_tris_lo, use_clip(c)));
        return ROPE_OK;
    }
    HIP_TRY(c, launch_raster(MODE_LAYER, loss, c->n_layers, c->stream, fp, c->rp, la, use_clip(c)));
    return ROPE_OK;
}

static RasterArgs base_args(rope_ctx *c, int n_render)
{
    RasterArgs a{};
    a.l_begin = 0; a.l_end = n_render; a.n_render = n_render;
    a.mvp = c->d_mvp; a.bounds = c->d_bounds;
    a.mask_lo = c->d_mask_lo; a.mask_hi = c->d_mask_hi; a.mask_words = c->mask_words;
    return a;
}

// The raster queue hands out the heaviest (candidate, tile) pairs first when a workgroup gets few enough pairs for the last ones
// to matter: with 40 pairs per workgroup (4096 candidates at 640x480) a 20 000-triangle pair that starts late kept its workgroup
// busy 180 us after the others had left, 7 % of the launch; with hundreds per workgroup (32 768 candidates at 1280x720) the tail is
// nothing and candidate order, which keeps a candidate's boxes in L2 for its tiles, is 1.5 % faster.
static uint32_t *queue_weights(const rope_ctx *c)
{
    const size_t busy_tiles = (size_t)std::max(std::min(2, c->n_tiles), c->n_tiles / 3);
    return (c->q_weighted && c->C > 256 && (size_t)c->C * busy_tiles <= (size_t)256 * 2 * c->n_cu) ? c->d_tile_tris : nullptr;     // C <= 256: fk_bounds_kernel, no weights
}

// forward kinematics + link matrices + screen boxes + tile masks (+ cleared sums) of the resident candidates
enum { EVAL_SINGLE = 0, EVAL_VIEWS = 1, EVAL_TARGETS = 2 };    // one target / camera-pose path (a view and a frame per row) / a frame per row

static int enqueue_geometry(rope_ctx *c, int n_render, int n_shared, const FrameParams &fp, bool views)
{
    const double *PV = views ? c->dv_PV : c->d_PV, *cand = views ? c->dv_cand : c->cand_dev;
    const int32_t *view_of = views ? c->dv_view_of : nullptr;
    if (c->C <= 256) {                              // one fused launch, one workgroup per candidate
        HIP_TRY(c, launch_fk_bounds(c->stream, cand, c->C, fp, c->rp, n_render, n_shared, c->d_joint_fixed, c->d_joint_axes, PV,
                                    view_of, c->d_mvp, c->d_bounds, c->d_sums, c->d_mask_lo, c->d_mask_hi, c->mask_words));
        return ROPE_OK;
    }
    HIP_TRY(c, launch_fk(c->stream, cand, c->C, n_render, c->d_joint_fixed, c->d_joint_axes, PV, view_of, c->d_mvp, c->d_sums,
                         c->d_mask_lo, c->d_mask_hi, c->mask_words, c->d_qctr, queue_weights(c), c->d_tile_tris_lo, c->n_tiles));
    // weights of the shared links for the layer launch: with the second level in use it draws the last shared link only
    const int lo_first = use_parents(c, n_shared) ? 2 : 0;
    HIP_TRY(c, launch_bounds(c->stream, c->C, fp, c->rp, n_render, n_shared, c->d_mvp, c->d_bounds, c->d_mask_lo, c->d_mask_hi, c->mask_words,
                             n_shared > 0 ? c->d_layer_of : nullptr, n_shared > 0 ? c->d_layer_rep : nullptr, queue_weights(c), c->d_tile_tris_lo, lo_first));
    return ROPE_OK;
}

static int enqueue_eval(rope_ctx *c, int n_render, int loss, const FrameParams &fp, double n_pix,
                        hipEvent_t *ev /* 5 events or nullptr */, int mode = EVAL_SINGLE)
{
    const bool views = mode == EVAL_VIEWS, targets = mode == EVAL_TARGETS;
    const float *tg_t32 = !targets ? nullptr : ((loss == ROPE_LOSS_TSWEEP && c->targets_ts) ? c->d_fts32 : (c->targets_t32 ? c->d_ft32 : nullptr));
    const bool layers = !views && want_layers(c) && layers_pay(c) && !(c->strategy & STRATEGY_NO_LAYERS);
    const int n_shared = layers ? std::min(3, n_render) : 0;
    if (layers) { int rc = ensure_layers(c); if (rc) return rc; }
    if (ev) HIP_TRY(c, hipEventRecord(ev[0], c->stream));
    // Few candidates (descent pairs, flips): one workgroup per (tile, candidate) would leave most of the chip idle,
    // so the meshlets of each tile are split over several workgroups that merge into a tile in global memory.
    int split = 1;
    if (!layers && !(c->strategy & STRATEGY_NO_SPLIT)) {
        // about a third of a frame's tiles hold the robot (all of them when the frame is one or two tiles): the others' workgroups
        // leave at once.  Too few shares per tile make each workgroup's chain of meshlets long: a floor, lower once the
        // launch runs to several generations of workgroups anyway (every share pays for clearing and scanning its LDS tile).
        const int busy_tiles = std::max(std::min(2, c->n_tiles), c->n_tiles / 3);
        const int floor_ = c->C * busy_tiles <= 256 ? c->split_min : c->split_min_many;
        split = std::max(std::min(c->split_cap, c->split_target / std::max(1, c->C * busy_tiles)), std::min(floor_, c->split_cap));
        if (split < 2 || c->C * busy_tiles > 4 * c->split_target) split = 1;       // enough (tile, candidate) pairs to fill the chip whole
    }
    // With the split, every workgroup can work out what it needs of the geometry itself — its candidate's six link matrices and
    // the screen boxes of its own share of the meshlets: the fk + boxes launch (10 us of a 60 us evaluation, most of it the launch)
    // goes.  (Not for the camera-pose path, whose rows name their own view matrices.)
    // Up to a few dozen rows: from 64 on, the launch saved is no more than what the matrices cost when every share of every row repeats them.
    // And on frames of a few tiles only (the Predictor's 160x90 is two): on a 25-tile frame two thirds of the workgroups belong to tiles
    // the robot does not reach — the masks of the separate launch let them leave at once, here each would do the matrices first.
    const bool geo = split > 1 && !views && c->C <= c->geo_rows && c->n_tiles <= 4 && !(c->strategy & STRATEGY_SEPARATE_GEOMETRY);
    if (!geo) { RopeRange r("rope:fk+bounds"); int rc = enqueue_geometry(c, n_render, n_shared, fp, views); if (rc) return rc; }
    c->mvp_valid = !geo;
    if (ev) HIP_TRY(c, hipEventRecord(ev[1], c->stream));
    RasterArgs a = base_args(c, n_render);
    if (split > 1) {
        const size_t need = (size_t)c->C * c->n_tiles * (TILE_W * TILE_H);
        if (need > c->gtile_cap) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, realloc_dev(&c->d_gtile, need));
            c->gtile_cap = need;
            c->gtile_dirty = true;
        }
        // the scoring kernel hands the buffer back "empty"; clear it only when new, after a failed pass, or when a
        // profiling flag may have skipped that kernel's work
        if (c->gtile_dirty || ROPE_SKIP(fp, ~0)) HIP_TRY(c, hipMemsetAsync(c->d_gtile, 0xFF, c->gtile_cap * sizeof(uint32_t), c->stream));
        c->gtile_dirty = true;
        RasterArgs sa = a;
        sa.split = split; sa.gtile = c->d_gtile;
        if (geo) {
            if ((size_t)c->C * c->n_tiles > c->touched_cap) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                HIP_TRY(c, realloc_dev(&c->d_touched, (size_t)256 * c->n_tiles));
                HIP_TRY(c, hipMemsetAsync(c->d_touched, 0, (size_t)256 * c->n_tiles * sizeof(int), c->stream));     // ordered with the kernels that stamp it
                c->touched_cap = (size_t)256 * c->n_tiles;
            }
            sa.cand_q = c->cand_dev; sa.joint_fixed = c->d_joint_fixed; sa.joint_axes = c->d_joint_axes; sa.PV = c->d_PV;
            sa.sums = c->d_sums;
            sa.touched = c->d_touched;
            if (c->pass_id == 0x7FFFFFFF) {                 // the stamp wraps after 2^31 passes: start over on a clean array
                HIP_TRY(c, hipMemsetAsync(c->d_touched, 0, c->touched_cap * sizeof(int), c->stream));
                c->pass_id = 0;
            }
            sa.pass_id = ++c->pass_id;
            a.touched = sa.touched; a.pass_id = sa.pass_id;
        }
        HIP_TRY(c, launch_raster(geo ? MODE_SPLIT_GEO : MODE_SPLIT, loss, c->C, c->stream, fp, c->rp, sa, use_clip(c, views)));
        a.gtile = c->d_gtile;
    }
    if (layers) {
        RasterArgs la = a;
        la.layer_sums = c->d_layer_sums; la.tq = c->d_tq; la.t32 = t32_plane(c, loss);
        if (targets) { la.tq = c->d_ftq; la.t32 = tg_t32; la.frame_of = c->frame_of_dev; }
        { RopeRange r("rope:shared-layers"); int rc = enqueue_layers(c, la, loss, n_shared, fp); if (rc) return rc; }
        a.l_begin = n_shared; a.layer_of = c->d_layer_of; a.layer_rep = c->d_layer_rep; a.layers = c->d_layers; a.layer_sums = c->d_layer_sums;
    }
    a.tq = c->d_tq; a.t32 = t32_plane(c, loss); a.sums = c->d_sums;
    if (targets) { a.tq = c->d_ftq; a.t32 = tg_t32; a.frame_of = c->frame_of_dev; }
    if (views) { a.tq = c->d_ftq; a.t32 = c->frames_t32 ? c->d_ft32 : nullptr; a.tl = c->frames_tl ? c->d_ftl : nullptr; a.frame_of = c->dv_frame_of; }
    if (ev) HIP_TRY(c, hipEventRecord(ev[2], c->stream));
    {
        RopeRange r("rope:raster+score");
        if (split > 1) {
            int slices = 1;
            while (slices < 8 && c->C * c->n_tiles * slices * 2 <= c->score_target) slices *= 2;
            HIP_TRY(c, launch_score_gtile(loss, c->C, slices, c->stream, fp, a));
            c->gtile_dirty = ROPE_SKIP(fp, ~0);
        } else if (c->C > 256 && !(c->strategy & STRATEGY_NO_QUEUE)) {
            // fk_mvp_kernel ran (C > 256) and cleared the queue counters; two 12-wave workgroups fit a CU
            HIP_TRY(c, launch_raster_queue(loss, c->C, 2 * c->n_cu, c->stream, fp, c->rp, a, c->d_qitems, queue_weights(c) ? c->q_segment : 0, c->d_qctr, queue_weights(c),
                                           use_clip(c, views)));
        } else {
            HIP_TRY(c, launch_raster(MODE_SCORE, loss, c->C, c->stream, fp, c->rp, a, use_clip(c, views)));
        }
    }
    if (ev) HIP_TRY(c, hipEventRecord(ev[3], c->stream));
    RopeRange fin("rope:finalize");
    if (views) return ROPE_OK;                     // per-(view, frame) sums are finalised by the caller
    c->err_on_host = c->C <= (targets ? rope_ctx::TARGET_HOST_ROWS : rope_ctx::HOST_ERR_ROWS);
    if (targets)                                   // every row against its own frame's totals and link flags; no argmin (rows of many frames)
        HIP_TRY(c, launch_finalize_frames(c->stream, c->d_sums, c->d_tg_total[loss], c->frame_of_dev, c->d_fflags, c->C, loss, n_render, n_pix,
                                          c->err_on_host ? c->d_err_host : c->d_err));
    else
        HIP_TRY(c, launch_finalize(c->stream, c->d_sums, c->d_total[loss], c->C, loss, n_render, n_pix, c->lf,
                                   c->err_on_host ? c->d_err_host : c->d_err));
    if (ev) HIP_TRY(c, hipEventRecord(ev[4], c->stream));
    c->last_n_render = n_render;
    c->results_valid = true;
    return ROPE_OK;
}

extern "C" int rope_eval_resident(rope_ctx *c, int n_render, int loss, const int32_t *crop)
{
    if (!c) return ROPE_E_ARG;
    FrameParams fp; double n_pix;
    int rc = check_eval_args(c, n_render, loss, crop, fp, n_pix);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    rc = ensure_empty(c, loss, fp);
    if (rc) return rc;
    return enqueue_eval(c, n_render, loss, fp, n_pix, nullptr);
}

extern "C" int rope_sync(rope_ctx *c)
{
    if (!c) return ROPE_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ROPE_OK;
}

extern "C" int rope_results_download(rope_ctx *c, double *err_out, uint64_t *sums_out, int32_t *best_idx, double *best_err)
{
    if (!c) return ROPE_E_ARG;
    if (c->C < 1 || !c->results_valid) ARG_FAIL(c, "rope_results_download: nothing evaluated (results do not survive rope_eval_views / rope_lookup_score / a new upload)");
    HIP_TRY(c, hipSetDevice(c->device));
    // one copy brings the errors, the best error and the best index (pinned staging)
    const double *res = c->err_on_host ? c->h_err : c->h_stage;
    if (!c->err_on_host) HIP_TRY(c, hipMemcpyAsync(c->h_stage, c->d_err, ((size_t)c->C + 2) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (sums_out) HIP_TRY(c, hipMemcpyAsync(sums_out, c->d_sums, (size_t)c->C * ROPE_SUM_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (err_out) std::memcpy(err_out, res, (size_t)c->C * sizeof(double));
    if (best_err) *best_err = res[c->C];
    if (best_idx) *best_idx = (int32_t)res[c->C + 1];
    return ROPE_OK;
}

// One batch is at most MAX_ROWS candidates (the grid's y dimension).
static constexpr int MAX_ROWS = 65535;

extern "C" int rope_eval(rope_ctx *c, const double *cand, int C, int n_render, int loss, const int32_t *crop,
                         double *err_out, uint64_t *sums_out, int32_t *best_idx, double *best_err)
{
    if (!c) return ROPE_E_ARG;
    if (C <= MAX_ROWS) {
        int rc = rope_candidates_upload(c, cand, C);
        if (rc) return rc;
        rc = rope_eval_resident(c, n_render, loss, crop);
        if (rc) return rc;
        return rope_results_download(c, err_out, sums_out, best_idx, best_err);
    }
    // larger sets (lookup grids up to 200 divisions per joint, lookup.py:50): batch after batch, the argmin merged by the
    // rule of finalize_argmin_kernel — first index of the smallest error, a NaN never beats a number
    if (!cand) ARG_FAIL(c, "rope_eval: null candidates");
    int32_t best = -1;
    double be = 0.0;
    for (int lo = 0; lo < C; lo += MAX_ROWS) {
        const int n = std::min(MAX_ROWS, C - lo);
        int rc = rope_candidates_upload(c, cand + 6 * (size_t)lo, n);
        if (rc) return rc;
        rc = rope_eval_resident(c, n_render, loss, crop);
        if (rc) return rc;
        int32_t bi = 0;
        double e = 0.0;
        rc = rope_results_download(c, err_out ? err_out + lo : nullptr, sums_out ? sums_out + (size_t)lo * ROPE_SUM_WORDS : nullptr, &bi, &e);
        if (rc) return rc;
        if (best < 0 || e < be || (be != be && e == e)) { best = lo + bi; be = e; }
    }
    if (best_idx) *best_idx = best;
    if (best_err) *best_err = be;
    return ROPE_OK;
}

static int raster_only(rope_ctx *c, const double *cand, int C, int n_render, int mode)
{
    if (!c->have_robot || !c->have_camera) ARG_FAIL(c, "render: robot and camera must be set first");
    if (n_render < 1 || n_render > c->n_links) ARG_FAIL(c, "render: n_render out of range");
    int rc = rope_candidates_upload(c, cand, C);
    if (rc) return rc;
    rc = enqueue_geometry(c, n_render, 0, c->fp, false);
    if (rc) return rc;
    RasterArgs a = base_args(c, n_render);
    a.key_out = c->d_key; a.cover = c->d_cover;
    HIP_TRY(c, launch_raster(mode, ROPE_LOSS_DEPTH, c->C, c->stream, c->fp, c->rp, a, use_clip(c)));
    c->last_n_render = n_render;
    return ROPE_OK;
}

extern "C" int rope_render(rope_ctx *c, const double *q, int n_render, float *depth, uint8_t *ids)
{
    if (!c) return ROPE_E_ARG;
    if (!q || !depth || !ids) ARG_FAIL(c, "rope_render: null pointer");
    if (!c->have_camera) ARG_FAIL(c, "rope_render: camera not set");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = (size_t)c->fp.W * c->fp.H;
    HIP_TRY(c, hipMemsetAsync(c->d_key, 0xFF, n * sizeof(uint32_t), c->stream));
    int rc = raster_only(c, q, 1, n_render, MODE_DUMP);
    if (rc) return rc;
    HIP_TRY(c, launch_resolve(c->stream, c->d_key, (int)n, c->fp, c->d_depth, c->d_ids));
    rc = copy_d2h_staged(c, depth, c->d_depth, n * sizeof(float));
    if (rc) return rc;
    return copy_d2h_staged(c, ids, c->d_ids, n);
}

extern "C" int rope_coverage(rope_ctx *c, const double *cand, int C, int n_render, uint8_t *cover)
{
    if (!c) return ROPE_E_ARG;
    if (!cand || !cover) ARG_FAIL(c, "rope_coverage: null pointer");
    if (!c->have_camera) ARG_FAIL(c, "rope_coverage: camera not set");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = (size_t)c->fp.W * c->fp.H;
    HIP_TRY(c, hipMemsetAsync(c->d_cover, 0, n, c->stream));
    int rc = raster_only(c, cand, C, n_render, MODE_COVER);
    if (rc) return rc;
    return copy_d2h_staged(c, cover, c->d_cover, n);
}

extern "C" int rope_lookup_build(rope_ctx *c, const double *cand, int C, int n_render, const int32_t *crop)
{
    if (!c) return ROPE_E_ARG;
    if (!cand || !crop) ARG_FAIL(c, "rope_lookup_build: null pointer");
    if (C < 1) ARG_FAIL(c, "rope_lookup_build: empty grid");
    if (!c->have_robot || !c->have_camera) ARG_FAIL(c, "rope_lookup_build: robot and camera must be set first");
    if (n_render < 1 || n_render > c->n_links) ARG_FAIL(c, "rope_lookup_build: n_render out of range");
    if (crop[0] < 0 || crop[1] >= c->fp.H || crop[0] > crop[1] || crop[2] < 0 || crop[3] >= c->fp.W || crop[2] > crop[3])
        ARG_FAIL(c, "rope_lookup_build: crop outside the image");
    HIP_TRY(c, hipSetDevice(c->device));
    c->table_C = 0;
    const size_t px = (size_t)(crop[1] - crop[0] + 1) * (size_t)(crop[3] - crop[2] + 1), need = px * (size_t)C;
    if (need > c->table_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, realloc_dev(&c->d_table, need));
        c->table_cap = need;
    }
    if (C > c->tscore_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->tscore_cap = 0;
        HIP_TRY(c, realloc_dev(&c->d_tsums, (size_t)C * ROPE_SUM_WORDS));
        HIP_TRY(c, realloc_dev(&c->d_terr, (size_t)C + 2));
        c->tscore_cap = C;
    }
    if (!c->d_zero_total) {
        HIP_TRY(c, hipMalloc((void **)&c->d_zero_total, ROPE_SUM_WORDS * sizeof(uint64_t)));
        HIP_TRY(c, hipMemsetAsync(c->d_zero_total, 0, ROPE_SUM_WORDS * sizeof(uint64_t), c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(c->d_table, 0, need * sizeof(float), c->stream));
    FrameParams fp = c->fp;
    fp.r0 = crop[0]; fp.r1 = crop[1]; fp.c0 = crop[2]; fp.c1 = crop[3];
    // the reference allows 200 divisions per joint (lookup.py:50, constants.py:30): more rows than one batch holds, so the
    // grid is rendered batch after batch into its rows of the table
    for (int lo = 0; lo < C; lo += MAX_ROWS) {
        const int n = std::min(MAX_ROWS, C - lo);
        int rc = rope_candidates_upload(c, cand + 6 * (size_t)lo, n);
        if (rc) return rc;
        const bool layers = want_layers(c);
        const int n_shared = layers ? std::min(3, n_render) : 0;
        if (layers) { rc = ensure_layers(c); if (rc) return rc; }
        rc = enqueue_geometry(c, n_render, n_shared, fp, false);
        if (rc) return rc;
        RasterArgs a = base_args(c, n_render);
        if (layers) {
            // layer pass without loss sums (layer_sums == nullptr): no target is needed to build a table
            rc = enqueue_layers(c, a, ROPE_LOSS_DEPTH, n_shared, c->fp);
            if (rc) return rc;
            a.l_begin = n_shared; a.layer_of = c->d_layer_of; a.layer_rep = c->d_layer_rep; a.layers = c->d_layers;
        }
        a.table = c->d_table + (size_t)lo * px;
        HIP_TRY(c, launch_raster(MODE_TABLE, ROPE_LOSS_LOOKUP, c->C, c->stream, fp, c->rp, a, use_clip(c)));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    // keep of every row only the groups of samples that hold something (the crop is the box of all poses together, and a pose
    // fills a fraction of its own box), then let the dense rows go: this is what the per-frame pass reads
    if ((size_t)C > c->trow_cap) {
        c->trow_cap = 0;
        HIP_TRY(c, realloc_dev(&c->d_tcount, (size_t)C));
        HIP_TRY(c, realloc_dev(&c->d_toff, (size_t)C));
        c->trow_cap = (size_t)C;
    }
    if (!c->d_tused) HIP_TRY(c, realloc_dev(&c->d_tused, (size_t)1));
    if (!c->d_ttotal) HIP_TRY(c, realloc_dev(&c->d_ttotal, (size_t)ROPE_SUM_WORDS));
    const int cw = crop[3] - crop[2] + 1, ch = crop[1] - crop[0] + 1;
    unsigned long long used = 0;
    hipError_t e = hipMemsetAsync(c->d_tused, 0, sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = launch_table_count(c->stream, cw, ch, c->d_table, C, c->d_tcount, c->d_toff, c->d_tused);
    if (e == hipSuccess) e = hipMemcpyAsync(&used, c->d_tused, sizeof(used), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (c->d_tgoff) { (void)hipFree(c->d_tgoff); c->d_tgoff = nullptr; }
    if (c->d_tgval) { (void)hipFree(c->d_tgval); c->d_tgval = nullptr; }
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_tgoff, std::max<size_t>(used, 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_tgval, std::max<size_t>(used, 1) * sizeof(float4));
    if (e == hipSuccess) e = launch_table_fill(c->stream, cw, ch, c->d_table, C, c->d_toff, c->d_tgoff, c->d_tgval);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_table); c->d_table = nullptr; c->table_cap = 0;
    HIP_TRY(c, e);
    c->table_groups = used;
    c->table_C = C;
    std::memcpy(c->table_crop, crop, 4 * sizeof(int32_t));
    return ROPE_OK;
}

extern "C" int rope_lookup_score(rope_ctx *c, double *scores_out, int32_t *best_idx, double *best_score)
{
    if (!c) return ROPE_E_ARG;
    if (c->table_C < 1) ARG_FAIL(c, "rope_lookup_score: no table built");
    if (!c->have_target || !c->have_t32) ARG_FAIL(c, "rope_lookup_score: needs a target with the float32 plane");
    HIP_TRY(c, hipSetDevice(c->device));
    FrameParams fp = c->fp;
    fp.r0 = c->table_crop[0]; fp.r1 = c->table_crop[1]; fp.c0 = c->table_crop[2]; fp.c1 = c->table_crop[3];
    const double n_pix = (double)(fp.r1 - fp.r0 + 1) * (double)(fp.c1 - fp.c0 + 1);
    const size_t words = table_crop_words(fp);
    if (words > c->tc_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->tc_cap = 0;
        HIP_TRY(c, realloc_dev(&c->d_tc, words));
        c->tc_cap = words;
    }
    HIP_TRY(c, launch_table_score(c->stream, fp, c->d_tcount, c->d_toff, c->d_tgoff, c->d_tgval, c->table_C, c->d_t32, c->d_tc,
                                  c->d_ttotal, c->d_tsums));
    HIP_TRY(c, launch_finalize(c->stream, c->d_tsums, c->d_zero_total, c->table_C, ROPE_LOSS_LOOKUP, ROPE_MAX_LINKS, n_pix, c->lf, c->d_terr));
    if (scores_out) HIP_TRY(c, hipMemcpyAsync(scores_out, c->d_terr, (size_t)c->table_C * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    double tail[2];
    HIP_TRY(c, hipMemcpyAsync(tail, c->d_terr + c->table_C, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (best_score) *best_score = tail[0];
    if (best_idx) *best_idx = (int32_t)tail[1];
    return ROPE_OK;
}

extern "C" int rope_set_frames(rope_ctx *c, int n_frames, const double *q, const uint64_t *tq, const float *t32,
                               const uint64_t *link_planes)
{
    if (!c) return ROPE_E_ARG;
    if (!c->have_camera) ARG_FAIL(c, "rope_set_frames: call rope_set_camera first (image size)");
    if (!q || !tq || n_frames < 1 || n_frames > 4096) ARG_FAIL(c, "rope_set_frames: need 1 <= n_frames <= 4096, q and tq");
    for (size_t i = 0; i < 6 * (size_t)n_frames; i++)
        if (!std::isfinite(q[i]) || std::fabs(q[i]) > 1.0e4) ARG_FAIL(c, "rope_set_frames: joint angle not finite or |q| > 1e4 rad");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t plane = (size_t)c->fp.W * c->fp.H, n = plane * (size_t)n_frames;
    // buffers only ever grow: a predictor calls this once per run with the same shapes
    if (n > c->frames_cap) { HIP_TRY(c, realloc_dev(&c->d_ftq, n)); c->frames_cap = n; }
    int rc = copy_h2d_staged(c, c->d_ftq, tq, n * sizeof(uint64_t));
    if (rc) return rc;
    c->frames_t32 = (t32 != nullptr);
    if (t32) {
        if (n > c->frames_t32_cap) { HIP_TRY(c, realloc_dev(&c->d_ft32, n)); c->frames_t32_cap = n; }
        rc = copy_h2d_staged(c, c->d_ft32, t32, n * sizeof(float));
        if (rc) return rc;
    }
    c->frames_tl = (link_planes != nullptr);
    if (link_planes) {
        if (n * ROPE_MAX_LINKS > c->frames_tl_cap) { HIP_TRY(c, realloc_dev(&c->d_ftl, n * ROPE_MAX_LINKS)); c->frames_tl_cap = n * ROPE_MAX_LINKS; }
        rc = copy_h2d_staged(c, c->d_ftl, link_planes, n * ROPE_MAX_LINKS * sizeof(uint64_t));
        if (rc) return rc;
    }
    if (n_frames > c->ftotal_cap) { HIP_TRY(c, realloc_dev(&c->d_ftotal, 5 * (size_t)n_frames * ROPE_SUM_WORDS)); c->ftotal_cap = n_frames; }
    if (!c->d_fempty) HIP_TRY(c, realloc_dev(&c->d_fempty, (size_t)MAX_MASK_WORDS * 32 * ROPE_SUM_WORDS));   // any tile count
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->h_fq.assign(q, q + 6 * (size_t)n_frames);
    c->n_frames = n_frames;
    c->n_targets = 0;                             // the planes are the camera-pose path's now
    for (bool &v : c->ftotal_valid) v = false;
    return ROPE_OK;
}

extern "C" int rope_eval_views(rope_ctx *c, const double *PV, int K, int n_render, int loss, uint64_t *sums_out)
{
    if (!c) return ROPE_E_ARG;
    if (!c->have_robot || !c->have_camera) ARG_FAIL(c, "rope_eval_views: robot and camera must be set first");
    if (c->n_frames < 1) ARG_FAIL(c, "rope_eval_views: no frames set");
    if (!PV || !sums_out || K < 1) ARG_FAIL(c, "rope_eval_views: bad arguments");
    if ((long long)K * c->n_frames > 65535) ARG_FAIL(c, "rope_eval_views: views x frames must not exceed 65535");
    if (n_render < 1 || n_render > c->n_links) ARG_FAIL(c, "rope_eval_views: n_render out of range");
    if (loss != ROPE_LOSS_DEPTH && loss != ROPE_LOSS_TSWEEP && loss != ROPE_LOSS_CAMFULL) ARG_FAIL(c, "rope_eval_views: loss must be DEPTH, TSWEEP or CAMFULL");
    if (loss == ROPE_LOSS_TSWEEP && !c->frames_t32) ARG_FAIL(c, "rope_eval_views: this loss needs the frames' float32 planes");
    if (loss == ROPE_LOSS_CAMFULL && !c->frames_tl) ARG_FAIL(c, "rope_eval_views: this loss needs the frames' link planes");
    for (size_t i = 0; i < 16 * (size_t)K; i++)
        if (!std::isfinite(PV[i])) ARG_FAIL(c, "rope_eval_views: non-finite view matrix");
    HIP_TRY(c, hipSetDevice(c->device));
    c->clip_views = false;
    for (int k = 0; k < K; k++) c->clip_views = c->clip_views || near_plane_in_reach(c, PV + 16 * (size_t)k);
    const int N = c->n_frames, C = K * N;
    int rc = ensure_capacity(c, C);
    if (rc) return rc;
    // One pinned block -> one copy: candidate (k, i) = view k, frame i (joint vector of frame i, scored against frame
    // i's planes) | the K view matrices | view index | frame index
    const size_t off_pv = 6 * (size_t)C * sizeof(double), off_vo = off_pv + 16 * (size_t)K * sizeof(double),
                 off_fo = off_vo + (size_t)C * sizeof(int32_t), bytes = off_fo + (size_t)C * sizeof(int32_t);
    if (bytes > c->vstage_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->h_vstage) { (void)hipHostFree(c->h_vstage); c->h_vstage = nullptr; }
        HIP_TRY(c, hipHostMalloc((void **)&c->h_vstage, bytes, hipHostMallocDefault));
        HIP_TRY(c, realloc_dev(&c->d_vstage, bytes));
        c->vstage_cap = bytes;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // the block may still feed the previous call's copy
    double *hc = reinterpret_cast<double *>(c->h_vstage);
    int32_t *hvo = reinterpret_cast<int32_t *>(c->h_vstage + off_vo), *hfo = reinterpret_cast<int32_t *>(c->h_vstage + off_fo);
    for (int k = 0; k < K; k++)
        for (int i = 0; i < N; i++) {
            std::memcpy(hc + 6 * ((size_t)k * N + i), &c->h_fq[6 * (size_t)i], 6 * sizeof(double));
            hvo[(size_t)k * N + i] = k;
            hfo[(size_t)k * N + i] = i;
        }
    std::memcpy(c->h_vstage + off_pv, PV, 16 * (size_t)K * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(c->d_vstage, c->h_vstage, bytes, hipMemcpyHostToDevice, c->stream));
    c->dv_cand = reinterpret_cast<const double *>(c->d_vstage);
    c->dv_PV = reinterpret_cast<const double *>(c->d_vstage + off_pv);
    c->dv_view_of = reinterpret_cast<const int32_t *>(c->d_vstage + off_vo);
    c->dv_frame_of = reinterpret_cast<const int32_t *>(c->d_vstage + off_fo);
    c->C = C;
    c->cand_valid = false;                            // the rows are (view, frame) pairs now: cand_dev / err_on_host describe nothing
    c->results_valid = false;
    c->cand_dev = nullptr;
    c->n_layers = C;                                  // every (view, frame) candidate is its own layer: nothing shared
    const size_t plane = (size_t)c->fp.W * c->fp.H;
    uint64_t *ftotal = c->d_ftotal + (size_t)loss * c->ftotal_cap * ROPE_SUM_WORDS;     // one block of totals per loss kind
    if (!c->ftotal_valid[loss]) {
        // sums of every frame with nothing rendered: the raster launch only adds what the render changes
        for (int i = 0; i < N; i++)
            HIP_TRY(c, launch_empty(loss, c->stream, c->fp, c->d_ftq + (size_t)i * plane, c->frames_t32 ? c->d_ft32 + (size_t)i * plane : nullptr,
                                    c->frames_tl ? c->d_ftl + (size_t)i * plane * ROPE_MAX_LINKS : nullptr, c->d_fempty,
                                    ftotal + (size_t)i * ROPE_SUM_WORDS));
        c->ftotal_valid[loss] = true;
    }
    rc = enqueue_eval(c, n_render, loss, c->fp, (double)plane, nullptr, EVAL_VIEWS);
    if (rc) return rc;
    const size_t n_sums = (size_t)C * ROPE_SUM_WORDS, n_tot = (size_t)N * ROPE_SUM_WORDS;
    if (n_sums + n_tot > c->vsums_cap) {
        if (c->h_vsums) { (void)hipHostFree(c->h_vsums); c->h_vsums = nullptr; }
        HIP_TRY(c, hipHostMalloc((void **)&c->h_vsums, (n_sums + n_tot) * sizeof(uint64_t), hipHostMallocDefault));
        c->vsums_cap = n_sums + n_tot;
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_vsums, c->d_sums, n_sums * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_vsums + n_sums, ftotal, n_tot * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint64_t *total = c->h_vsums + n_sums;
    std::memcpy(sums_out, c->h_vsums, n_sums * sizeof(uint64_t));
    const bool links = (loss == ROPE_LOSS_CAMFULL);
    for (int k = 0; k < K; k++)
        for (int i = 0; i < N; i++)
            for (int w = 0; w < ROPE_SUM_WORDS; w++) {
                uint64_t &o = sums_out[((size_t)k * N + i) * ROPE_SUM_WORDS + w];
                const bool live = w < SUM_LINK0 || (links && w < SUM_LINK0 + 3 * n_render);
                o = live ? o + total[(size_t)i * ROPE_SUM_WORDS + w] : 0;      // modulo 2^64, as the kernel's deltas are
            }
    return ROPE_OK;
}

// per-frame scratch of the batched prediction, sized for n_frames resident targets
static int size_target_scratch(rope_ctx *c, int n_frames)
{
    if (n_frames > c->tg_total_cap) {
        c->tg_total_cap = 0;
        for (int k = 0; k < 4; k++) HIP_TRY(c, realloc_dev(&c->d_tg_total[k], (size_t)n_frames * ROPE_SUM_WORDS));
        HIP_TRY(c, realloc_dev(&c->d_tg_ltotal, (size_t)n_frames * ROPE_SUM_WORDS));
        c->tg_total_cap = n_frames;
    }
    const size_t empties = (size_t)n_frames * c->n_tiles * ROPE_SUM_WORDS;
    if (empties > c->tg_empty_cap) { c->tg_empty_cap = 0; HIP_TRY(c, realloc_dev(&c->d_tg_empty, empties)); c->tg_empty_cap = empties; }
    if (n_frames > c->tg_best_cap) { c->tg_best_cap = 0; HIP_TRY(c, realloc_dev(&c->d_tg_best, 2 * (size_t)n_frames)); c->tg_best_cap = n_frames; }
    return ROPE_OK;
}

// ---- batched prediction: the targets of many frames resident at once, every row of a batch scored against its own frame's
extern "C" int rope_set_targets(rope_ctx *c, int n_frames, const uint64_t *tq, const float *t32, const float *t32_tsweep, const uint8_t *link_flags)
{
    if (!c) return ROPE_E_ARG;
    if (!c->have_camera) ARG_FAIL(c, "rope_set_targets: call rope_set_camera first (image size)");
    if (!tq || !link_flags || n_frames < 1 || n_frames > 65535) ARG_FAIL(c, "rope_set_targets: need 1 <= n_frames <= 65535, tq and link_flags");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_targets = 0;
    c->n_frames = 0;                              // d_ftq / d_ft32 are shared with the camera-pose path's frames
    const size_t plane = (size_t)c->fp.W * c->fp.H, n = plane * (size_t)n_frames;
    if (n > c->frames_cap) { c->frames_cap = 0; HIP_TRY(c, realloc_dev(&c->d_ftq, n)); c->frames_cap = n; }
    int rc = copy_h2d_staged(c, c->d_ftq, tq, n * sizeof(uint64_t));
    if (rc) return rc;
    if (t32) {
        if (n > c->frames_t32_cap) { c->frames_t32_cap = 0; HIP_TRY(c, realloc_dev(&c->d_ft32, n)); c->frames_t32_cap = n; }
        rc = copy_h2d_staged(c, c->d_ft32, t32, n * sizeof(float));
        if (rc) return rc;
    }
    if (t32_tsweep) {
        if (n > c->fts_cap) { c->fts_cap = 0; HIP_TRY(c, realloc_dev(&c->d_fts32, n)); c->fts_cap = n; }
        rc = copy_h2d_staged(c, c->d_fts32, t32_tsweep, n * sizeof(float));
        if (rc) return rc;
    }
    if (n_frames > c->fflags_cap) { c->fflags_cap = 0; HIP_TRY(c, realloc_dev(&c->d_fflags, (size_t)n_frames)); c->fflags_cap = n_frames; }
    static_assert(sizeof(LinkFlags) == 8, "link flags travel as 8 bytes per frame");
    rc = copy_h2d_staged(c, c->d_fflags, link_flags, 8 * (size_t)n_frames);
    if (rc) return rc;
    rc = size_target_scratch(c, n_frames);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->targets_t32 = (t32 != nullptr);
    c->targets_ts = (t32_tsweep != nullptr);
    for (bool &v : c->tg_total_valid) v = false;
    for (bool &v : c->ftotal_valid) v = false;
    c->n_targets = n_frames;
    return ROPE_OK;
}

// The same targets as rope_set_targets, but into the context's SECOND set of planes and on a stream of its own: returns once the
// copies are enqueued (page-locked sources, rope_host_alloc) — the resident set stays in use meanwhile, from this or another thread
// (this call touches nothing the evaluation calls use).  The host buffers belong to the copy until rope_commit_targets returns.
extern "C" int rope_stage_targets(rope_ctx *c, int n_frames, const uint64_t *tq, const float *t32, const float *t32_tsweep, const uint8_t *link_flags)
{
    if (!c) return ROPE_E_ARG;
    // this call may run beside an evaluation on another thread: its failures are reported through the return code and a message of its own
    auto fail = [&](int code, const char *msg) { c->stage_err = msg; return code; };
    c->stage_err.clear();
    if (!c->have_camera) return fail(ROPE_E_ARG, "rope_stage_targets: call rope_set_camera first (image size)");
    if (!tq || !link_flags || n_frames < 1 || n_frames > 65535) return fail(ROPE_E_ARG, "rope_stage_targets: need 1 <= n_frames <= 65535, tq and link_flags");
    if (hipSetDevice(c->device) != hipSuccess) return fail(ROPE_E_HIP, "rope_stage_targets: hipSetDevice failed");
    if (!c->copy_stream && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) return fail(ROPE_E_HIP, "rope_stage_targets: no stream");
    if (hipStreamSynchronize(c->copy_stream) != hipSuccess) return fail(ROPE_E_HIP, "rope_stage_targets: the upload stream failed");
    c->staged_n = 0;
    const int W = c->fp.W, H = c->fp.H;
    const size_t plane = (size_t)W * H, n = plane * (size_t)n_frames;
    bool ok = true;
    auto grow = [&](auto **p, size_t &cap, size_t want) {
        if (want <= cap) return;
        cap = 0;
        if (realloc_dev(p, want) != hipSuccess) { ok = false; return; }
        cap = want;
    };
    grow(&c->s_ftq, c->s_frames_cap, n);
    if (t32) grow(&c->s_ft32, c->s_t32_cap, n);
    if (t32_tsweep) grow(&c->s_fts32, c->s_fts_cap, n);
    if (n_frames > c->s_fflags_cap) {
        c->s_fflags_cap = 0;
        if (realloc_dev(&c->s_fflags, (size_t)n_frames) == hipSuccess) c->s_fflags_cap = n_frames; else ok = false;
    }
    if (!ok) { (void)hipGetLastError(); return fail(ROPE_E_NOMEM, "rope_stage_targets: out of device memory"); }
    // the few bytes of flags first: a small (or pageable) source goes through the pinned block, which waits for the stream
    int rc = copy_h2d_on(c->copy_stream, &c->h_copy2, c->stage_err, c->s_fflags, link_flags, 8 * (size_t)n_frames);
    if (!rc) rc = copy_h2d_on(c->copy_stream, &c->h_copy2, c->stage_err, c->s_ftq, tq, n * sizeof(uint64_t));
    if (!rc && t32) rc = copy_h2d_on(c->copy_stream, &c->h_copy2, c->stage_err, c->s_ft32, t32, n * sizeof(float));
    if (!rc && t32_tsweep) rc = copy_h2d_on(c->copy_stream, &c->h_copy2, c->stage_err, c->s_fts32, t32_tsweep, n * sizeof(float));
    if (rc) return rc;
    c->staged_t32 = (t32 != nullptr);
    c->staged_ts = (t32_tsweep != nullptr);
    c->staged_W = W; c->staged_H = H;
    c->staged_n = n_frames;
    return ROPE_OK;
}

// The staged set becomes the resident one (and the resident planes the next staging space): waits for the upload and for the
// context's own stream, then swaps.
extern "C" int rope_commit_targets(rope_ctx *c)
{
    if (!c) return ROPE_E_ARG;
    if (c->staged_n < 1) { c->err = c->stage_err.empty() ? "rope_commit_targets: nothing staged (rope_stage_targets)" : c->stage_err; return ROPE_E_ARG; }
    if (c->staged_W != c->fp.W || c->staged_H != c->fp.H) { c->staged_n = 0; ARG_FAIL(c, "rope_commit_targets: the image size changed since rope_stage_targets"); }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int n_frames = c->staged_n;
    c->staged_n = 0;
    c->n_targets = 0;
    c->n_frames = 0;                              // d_ftq / d_ft32 are shared with the camera-pose path's frames
    std::swap(c->d_ftq, c->s_ftq);       std::swap(c->frames_cap, c->s_frames_cap);
    if (c->staged_t32) { std::swap(c->d_ft32, c->s_ft32);   std::swap(c->frames_t32_cap, c->s_t32_cap); }
    if (c->staged_ts)  { std::swap(c->d_fts32, c->s_fts32); std::swap(c->fts_cap, c->s_fts_cap); }
    std::swap(c->d_fflags, c->s_fflags); std::swap(c->fflags_cap, c->s_fflags_cap);
    int rc = size_target_scratch(c, n_frames);
    if (rc) return rc;
    c->targets_t32 = c->staged_t32;
    c->targets_ts = c->staged_ts;
    for (bool &v : c->tg_total_valid) v = false;
    for (bool &v : c->ftotal_valid) v = false;
    c->n_targets = n_frames;
    return ROPE_OK;
}

// "nothing rendered" totals of every resident target for this loss (and crop)
static int ensure_target_totals(rope_ctx *c, int loss, const FrameParams &fp)
{
    const int cr[4] = {fp.r0, fp.r1, fp.c0, fp.c1};
    if (c->tg_total_valid[loss] && std::memcmp(cr, c->tg_crop[loss], sizeof cr) == 0) return ROPE_OK;
    const float *t32 = (loss == ROPE_LOSS_TSWEEP && c->targets_ts) ? c->d_fts32 : (c->targets_t32 ? c->d_ft32 : nullptr);
    HIP_TRY(c, launch_empty(loss, c->stream, fp, c->d_ftq, t32, nullptr, c->d_tg_empty, c->d_tg_total[loss], c->n_targets));
    c->tg_total_valid[loss] = true;
    std::memcpy(c->tg_crop[loss], cr, sizeof cr);
    return ROPE_OK;
}

extern "C" int rope_eval_targets(rope_ctx *c, const double *cand, const int32_t *frame_of, int R, int n_render, int loss, const int32_t *crop,
                                 double *err_out)
{
    if (!c) return ROPE_E_ARG;
    if (!c->have_robot || !c->have_camera) ARG_FAIL(c, "rope_eval_targets: robot and camera must be set first");
    if (c->n_targets < 1) ARG_FAIL(c, "rope_eval_targets: no targets set (rope_set_targets)");
    if (!cand || !frame_of || !err_out || R < 1) ARG_FAIL(c, "rope_eval_targets: bad arguments");
    if (n_render < 1 || n_render > c->n_links) ARG_FAIL(c, "rope_eval_targets: n_render out of range");
    if (loss < 0 || loss > 3) ARG_FAIL(c, "rope_eval_targets: unknown loss");
    if (loss == ROPE_LOSS_LOOKUP && !c->targets_t32) ARG_FAIL(c, "rope_eval_targets: this loss needs the targets' float32 planes");
    if (loss == ROPE_LOSS_TSWEEP && !c->targets_t32 && !c->targets_ts) ARG_FAIL(c, "rope_eval_targets: this loss needs float32 planes");
    for (int i = 0; i < R; i++)
        if (frame_of[i] < 0 || frame_of[i] >= c->n_targets) ARG_FAIL(c, "rope_eval_targets: frame index outside the resident targets");
    FrameParams fp = c->fp;
    double n_pix = (double)fp.W * fp.H;
    if (loss == ROPE_LOSS_LOOKUP) {
        if (!crop) ARG_FAIL(c, "rope_eval_targets: lookup loss needs a crop");
        if (crop[0] < 0 || crop[1] >= fp.H || crop[0] > crop[1] || crop[2] < 0 || crop[3] >= fp.W || crop[2] > crop[3]) ARG_FAIL(c, "rope_eval_targets: crop outside the image");
        fp.r0 = crop[0]; fp.r1 = crop[1]; fp.c0 = crop[2]; fp.c1 = crop[3];
        n_pix = (double)(crop[1] - crop[0] + 1) * (double)(crop[3] - crop[2] + 1);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_target_totals(c, loss, fp);
    if (rc) return rc;
    for (int lo = 0; lo < R; lo += MAX_ROWS) {
        const int n = std::min(MAX_ROWS, R - lo);
        rc = upload_candidates(c, cand + 6 * (size_t)lo, n, true, frame_of + lo);
        if (rc) return rc;
        rc = enqueue_eval(c, n_render, loss, fp, n_pix, nullptr, EVAL_TARGETS);
        // the rows are scored against the frames' targets: nothing here is a result of the single-target calls
        c->cand_valid = c->results_valid = false;
        if (rc) return rc;
        if (!c->err_on_host) HIP_TRY(c, hipMemcpyAsync(c->h_stage, c->d_err, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::memcpy(err_out + lo, c->err_on_host ? c->h_err : c->h_stage, (size_t)n * sizeof(double));
    }
    return ROPE_OK;
}

extern "C" int rope_lookup_score_targets(rope_ctx *c, int32_t *best_idx, double *best_score, double *scores_out)
{
    if (!c) return ROPE_E_ARG;
    if (c->table_C < 1) ARG_FAIL(c, "rope_lookup_score_targets: no table built");
    if (c->n_targets < 1 || !c->targets_t32) ARG_FAIL(c, "rope_lookup_score_targets: needs targets with their float32 planes (rope_set_targets)");
    if (!best_idx) ARG_FAIL(c, "rope_lookup_score_targets: null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    FrameParams fp = c->fp;
    fp.r0 = c->table_crop[0]; fp.r1 = c->table_crop[1]; fp.c0 = c->table_crop[2]; fp.c1 = c->table_crop[3];
    const size_t crop_px = table_crop_words(fp), N = (size_t)c->n_targets;     // crop rows padded to whole groups of four samples
    if (N * crop_px > c->tg_t32c_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->tg_t32c_cap = 0;
        HIP_TRY(c, realloc_dev(&c->d_tg_t32c, N * crop_px));
        c->tg_t32c_cap = N * crop_px;
    }
    if (N * c->table_C > c->tg_scores_cap) { HIP_TRY(c, hipStreamSynchronize(c->stream)); c->tg_scores_cap = 0; HIP_TRY(c, realloc_dev(&c->d_tg_scores, N * c->table_C)); c->tg_scores_cap = N * c->table_C; }
    HIP_TRY(c, launch_table_score_frames(c->stream, fp, c->d_tcount, c->d_toff, c->d_tgoff, c->d_tgval, c->table_C, c->d_ft32, c->n_targets, c->d_tg_t32c, c->d_tg_ltotal,
                                         c->d_tg_scores, c->d_tg_best));
    std::vector<double> best(2 * N);
    HIP_TRY(c, hipMemcpyAsync(best.data(), c->d_tg_best, 2 * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (scores_out) HIP_TRY(c, hipMemcpyAsync(scores_out, c->d_tg_scores, N * c->table_C * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t f = 0; f < N; f++) {
        best_idx[f] = (int32_t)best[2 * f + 1];
        if (best_score) best_score[f] = best[2 * f];
    }
    return ROPE_OK;
}

extern "C" int rope_debug_mvp(rope_ctx *c, float *mvp_out, int C, int n_render)
{
    if (!c) return ROPE_E_ARG;
    if (!mvp_out || C < 1 || C > c->C || n_render < 1 || n_render > ROPE_MAX_LINKS) ARG_FAIL(c, "rope_debug_mvp: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->mvp_valid) {
        // the last pass worked the matrices out inside its raster workgroups: form them the way any other pass does
        if (!c->cand_valid) ARG_FAIL(c, "rope_debug_mvp: no resident candidates");
        const int keep = c->strategy;
        c->strategy |= STRATEGY_SEPARATE_GEOMETRY;
        const int rc = enqueue_geometry(c, n_render, 0, c->fp, false);
        c->strategy = keep;
        if (rc) return rc;
        c->results_valid = false;                     // that launch cleared the sums
        c->mvp_valid = true;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < C; i++)
        HIP_TRY(c, hipMemcpy(mvp_out + (size_t)i * n_render * 16, c->d_mvp + (size_t)i * ROPE_MAX_LINKS * 16,
                             (size_t)n_render * 16 * sizeof(float), hipMemcpyDeviceToHost));
    return ROPE_OK;
}

extern "C" int rope_set_strategy(rope_ctx *c, int flags)
{
    if (!c) return ROPE_E_ARG;
    if (flags & ~(STRATEGY_NO_LAYERS | STRATEGY_NO_SPLIT | STRATEGY_NO_PARENTS | STRATEGY_NO_QUEUE | STRATEGY_CLIP_KERNELS | STRATEGY_SEPARATE_GEOMETRY)) ARG_FAIL(c, "rope_set_strategy: unknown flag");
    c->strategy = flags;
    return ROPE_OK;
}

#ifdef ROPE_PROFILE
namespace rope { hipError_t read_clock_stamps(unsigned long long *out); hipError_t read_bounds_violations(unsigned int *out); }

// Profiling build: how many indices into the raster kernels' shared arrays / queue segments were out of range since the library
// was loaded (ROPE_CHECK_INDEX in rope_kernels.hip)
extern "C" int rope_debug_bounds(rope_ctx *c, int *violations)
{
    if (!c || !violations) return ROPE_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned int v = 0;
    HIP_TRY(c, read_bounds_violations(&v));
    *violations = (int)v;
    return ROPE_OK;
}

// Profiling build: the shader clock (GHz) during the last scoring launch of a large batch — median over its workgroups of
// delta s_memtime / delta s_memrealtime (100 MHz); ghz[1] = the launch's length by the same stamps in ms (first start to last end)
extern "C" int rope_debug_clock(rope_ctx *c, double *ghz)
{
    if (!c || !ghz) return ROPE_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> st(4 * 1024);
    HIP_TRY(c, read_clock_stamps(st.data()));
    std::vector<double> f;
    unsigned long long r_lo = ~0ull, r_hi = 0;
    const int n = std::min(1024, 2 * c->n_cu);
    for (int i = 0; i < n; i++) {
        const unsigned long long t0 = st[4 * i], r0 = st[4 * i + 1], t1 = st[4 * i + 2], r1 = st[4 * i + 3];
        if (r1 > r0 + 100 && t1 > t0) f.push_back((double)(t1 - t0) / ((double)(r1 - r0) * 10.0));       // workgroups that ran for more than a microsecond
        if (r1 > r0) { r_lo = std::min(r_lo, r0); r_hi = std::max(r_hi, r1); }
    }
    if (f.empty()) ARG_FAIL(c, "rope_debug_clock: no stamps (run a batch of more than 256 rows first)");
    std::sort(f.begin(), f.end());
    ghz[0] = f[f.size() / 2];
    ghz[1] = (double)(r_hi - r_lo) * 1.0e-5;
    return ROPE_OK;
}

extern "C" int rope_debug_skip(rope_ctx *c, int mask)
{
    if (!c) return ROPE_E_ARG;
    c->fp.debug = mask;
    return ROPE_OK;
}
#endif

extern "C" int rope_profile_eval(rope_ctx *c, int n_render, int loss, const int32_t *crop, int reps, float *ms)
{
    if (!c) return ROPE_E_ARG;
    if (!ms || reps < 1) ARG_FAIL(c, "rope_profile_eval: bad arguments");
    FrameParams fp; double n_pix;
    int rc = check_eval_args(c, n_render, loss, crop, fp, n_pix);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    rc = ensure_empty(c, loss, fp);
    if (rc) return rc;
    std::vector<hipEvent_t> ev(5 * (size_t)reps);
    for (auto &e : ev) HIP_TRY(c, hipEventCreate(&e));
    for (int r = 0; r < reps; r++) {
        rc = enqueue_eval(c, n_render, loss, fp, n_pix, &ev[5 * (size_t)r]);
        if (rc) break;
    }
    if (!rc) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        double acc[4] = {0, 0, 0, 0};
        for (int r = 0; r < reps; r++) {
            float t;
            for (int k = 0; k < 4; k++) {
                HIP_TRY(c, hipEventElapsedTime(&t, ev[5 * (size_t)r + k], ev[5 * (size_t)r + k + 1]));
                acc[k] += t;
            }
        }
        float total;
        HIP_TRY(c, hipEventElapsedTime(&total, ev[0], ev[5 * (size_t)reps - 1]));
        for (int k = 0; k < 4; k++) ms[k] = (float)(acc[k] / reps);
        ms[4] = total / (float)reps;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}
