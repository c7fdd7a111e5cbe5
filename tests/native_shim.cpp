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

extern "C" {
void shim_set_callback(shim_eval_cb cb) { g_cb = cb; }
const char *shim_last_error() { return g_err.c_str(); }

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
}
