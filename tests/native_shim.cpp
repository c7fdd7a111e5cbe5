// Test scaffolding only: lets rope_predict.cpp (pure host code over the C ABI) run WITHOUT the HIP engine, by
// answering its rope_eval calls through a callback the test installs (which scores the poses with the CPU oracle).
// Built by tests/test_native_host.py into a scratch directory; never part of librope_hip.so.
#include <cstdint>
#include <string>

#include "../include/rope_s3d.h"

typedef int (*shim_eval_cb)(const double *cand, int C, int n_render, int loss, const int32_t *crop, double *err_out, int32_t *best_idx);

static shim_eval_cb g_cb = nullptr;
static std::string g_err;

void rope_set_error(rope_ctx *, const std::string &msg) { g_err = msg; }
static int g_ranges = 0, g_range_depth = 0;
void rope_range_push(const char *) { g_ranges++; g_range_depth++; }
void rope_range_pop() { g_range_depth--; }

extern "C" {
void shim_set_callback(shim_eval_cb cb) { g_cb = cb; }
const char *shim_last_error() { return g_err.c_str(); }
int shim_ranges_opened() { return g_ranges; }
int shim_range_depth() { return g_range_depth; }

int rope_eval(rope_ctx *, const double *cand, int C, int n_render, int loss, const int32_t *crop, double *err_out, uint64_t *,
              int32_t *best_idx, double *)
{
    return g_cb ? g_cb(cand, C, n_render, loss, crop, err_out, best_idx) : ROPE_E_ARG;
}

int rope_lookup_score(rope_ctx *, double *, int32_t *, double *)
{
    g_err = "shim: no stored table";
    return ROPE_E_ARG;
}

// rope_predict_batch's rows carry the frame they belong to: the test's second callback scores each against that frame's target
typedef int (*shim_targets_cb)(const double *cand, const int32_t *frame_of, int R, int n_render, int loss, const int32_t *crop, double *err_out);
static shim_targets_cb g_tcb = nullptr;
void shim_set_targets_callback(shim_targets_cb cb) { g_tcb = cb; }

int rope_eval_targets(rope_ctx *, const double *cand, const int32_t *frame_of, int R, int n_render, int loss, const int32_t *crop, double *err_out)
{
    return g_tcb ? g_tcb(cand, frame_of, R, n_render, loss, crop, err_out) : ROPE_E_ARG;
}

int rope_lookup_score_targets(rope_ctx *, int32_t *, double *, double *)
{
    g_err = "shim: no stored table";
    return ROPE_E_ARG;
}

// rope_set_robot_mesh (rope_meshlets.cpp) ends in rope_set_robot: here every array it hands over is read once, end to end
// (so that a sanitised build sees any range it should not have), and the totals are kept for the test to look at.
static int64_t g_robot[4];       // meshlets, vertices, triangles, checksum
const int64_t *shim_last_robot() { return g_robot; }

int rope_set_robot(rope_ctx *, const uint32_t *ml_header, int n_meshlets, const float *ml_verts, int n_ml_verts, const uint32_t *ml_tris,
                   int n_ml_tris, const int32_t *link_first, int n_links, const double *joint_fixed, const double *joint_axes)
{
    int64_t sum = 0;
    for (int m = 0; m < n_meshlets; m++) {
        const uint32_t *h = ml_header + 8 * (size_t)m;
        const uint32_t v0 = h[4], t0 = h[5], nv = h[6] & 0xFFFF, nt = h[6] >> 16;
        if ((size_t)v0 + nv > (size_t)n_ml_verts || (size_t)t0 + nt > (size_t)n_ml_tris) return ROPE_E_ARG;
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t p = ml_tris[t0 + t];
            if ((p & 0xFF) >= nv || ((p >> 8) & 0xFF) >= nv || ((p >> 16) & 0xFF) >= nv) return ROPE_E_ARG;
            sum += p;
        }
        for (uint32_t v = 0; v < 3 * nv; v++) sum += (int64_t)(ml_verts[3 * (size_t)v0 + v] * 1.0e4f);
    }
    for (int l = 0; l <= n_links; l++) sum += link_first[l];
    for (int k = 0; k < 72; k++) sum += (int64_t)(joint_fixed[k] * 1.0e3);
    for (int k = 0; k < 18; k++) sum += (int64_t)(joint_axes[k] * 1.0e3);
    g_robot[0] = n_meshlets; g_robot[1] = n_ml_verts; g_robot[2] = n_ml_tris; g_robot[3] = sum;
    return ROPE_OK;
}
}
