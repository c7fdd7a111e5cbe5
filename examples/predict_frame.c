/*
 * predict_frame.c — a host that is not Python: one frame through the whole prediction path using nothing but the C ABI
 * of librope_hip.so (include/rope_s3d.h).
 *
 *     gcc -O2 -Iinclude examples/predict_frame.c -Lrope_s3d_amd/csrc -lrope_hip \
 *         -Wl,-rpath,$PWD/rope_s3d_amd/csrc -Wl,-rpath-link,/opt/rocm/lib -lm -o predict_frame
 *     ./predict_frame <bundle directory>        # prints the six joint angles, %.17g
 *
 * The bundle is a directory of raw little-endian arrays (tools/dump_frame_bundle.py writes one): the welded link
 * meshes and the joint chain (what URDFReader + MeshLoader give the reference), the camera, one prepared target
 * frame, the lookup pose grid and its crop.  Calls, in order: rope_create, rope_set_robot_mesh, rope_set_camera,
 * rope_set_target, rope_lookup_build, rope_predict — the reference's Predictor.__init__ + Predictor.run
 * (robotpose/prediction/predict.py:38-124,127-375) with the 'SLU' stage list of stages.py:152-168.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rope_s3d.h"

static void *load(const char *dir, const char *name, size_t elem, size_t *count)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *buf = malloc(bytes > 0 ? (size_t)bytes : 1);
    if (!buf || fread(buf, 1, (size_t)bytes, f) != (size_t)bytes || (size_t)bytes % elem) { fprintf(stderr, "bad file %s\n", path); exit(2); }
    fclose(f);
    if (count) *count = (size_t)bytes / elem;
    return buf;
}

static rope_stage stage(int kind, int to_render, int count, unsigned joints, double range)
{
    rope_stage s;
    memset(&s, 0, sizeof s);
    s.kind = kind; s.to_render = to_render; s.count = count; s.joints = joints;
    for (int i = 0; i < 6; i++) s.init_rate[i] = NAN;
    s.rate_reduction = 0.5; s.early_stop = 0.01; s.range = range;
    return s;
}

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rope_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <bundle directory>\n", argv[0]); return 2; }
    const char *dir = argv[1];
    size_t n_vo, n_grid, n_px;
    float *verts = load(dir, "verts.f32", 4, NULL);
    int32_t *faces = load(dir, "faces.i32", 4, NULL);
    int32_t *vtx_off = load(dir, "vtx_off.i32", 4, &n_vo), *tri_off = load(dir, "tri_off.i32", 4, NULL);
    double *joint_fixed = load(dir, "joint_fixed.f64", 8, NULL), *joint_axes = load(dir, "joint_axes.f64", 8, NULL);
    double *PV = load(dir, "PV.f64", 8, NULL), *clip = load(dir, "clip.f64", 8, NULL);
    int32_t *dims = load(dir, "dims.i32", 4, NULL);                  /* W, H */
    double *limits = load(dir, "limits.f64", 8, NULL), *camera_pose = load(dir, "camera_pose.f64", 8, NULL);
    uint64_t *tq = load(dir, "tq.u64", 8, &n_px);
    float *t32 = load(dir, "t32.f32", 4, NULL);
    uint8_t *flags = load(dir, "flags.u8", 1, NULL);
    double *grid = load(dir, "grid.f64", 8, &n_grid);
    int32_t *crop = load(dir, "crop.i32", 4, NULL);
    if (n_px != (size_t)dims[0] * dims[1]) { fprintf(stderr, "target plane does not match %dx%d\n", dims[0], dims[1]); return 2; }

    rope_ctx *ctx = NULL;
    CHECK(rope_create(&ctx, 0));
    CHECK(rope_set_robot_mesh(ctx, verts, faces, vtx_off, tri_off, (int)n_vo - 1, joint_fixed, joint_axes));
    CHECK(rope_set_camera(ctx, PV, dims[0], dims[1], clip[0], clip[1]));
    CHECK(rope_lookup_build(ctx, grid, (int)(n_grid / 6), 6, crop));           /* once per camera pose */
    CHECK(rope_set_target(ctx, tq, t32, flags));                             /* once per frame */

    enum { S = 1, L = 2, U = 4 };
    rope_stage stages[9];
    stages[0] = stage(ROPE_STAGE_LOOKUP, 6, 0, 0, NAN);
    stages[1] = stage(ROPE_STAGE_SFLIP, 4, 0, 0, NAN);
    stages[2] = stage(ROPE_STAGE_DESCENT, 4, 10, S | L, NAN);
    { const double r[6] = {0.05, 0.05, 0.1, 0.5, 0.5, 0.5}; memcpy(stages[2].init_rate, r, sizeof r); stages[2].early_stop = 0.1; }
    stages[3] = stage(ROPE_STAGE_SFLIP, 4, 0, 0, NAN);
    stages[4] = stage(ROPE_STAGE_ISWEEP, 6, 25, U, NAN);
    stages[5] = stage(ROPE_STAGE_SFLIP, 4, 0, 0, NAN);
    stages[6] = stage(ROPE_STAGE_SFLIP, 6, 0, 0, NAN);
    stages[7] = stage(ROPE_STAGE_ISWEEP, 6, 10, U, 0.1);
    stages[8] = stage(ROPE_STAGE_DESCENT, 6, 40, S | L | U, NAN);
    stages[8].early_stop = 0.0075;

    const double min_ang_inc[6] = {.005, .005, .005, .005, .005, .005};
    rope_predict_args a;
    memset(&a, 0, sizeof a);
    a.stages = stages; a.n_stages = 9; a.speculate = 3;
    a.limits = limits; a.camera_pose = camera_pose; a.min_ang_inc = min_ang_inc;
    a.lookup_angles = grid; a.n_lookup = (int)(n_grid / 6); a.use_table = 1; a.lookup_crop = crop;
    double angles[6];
    int64_t evals = 0;
    CHECK(rope_predict(ctx, &a, angles, NULL, &evals));
    for (int j = 0; j < 6; j++) printf("%.17g%c", angles[j], j == 5 ? '\n' : ' ');
    fprintf(stderr, "%lld candidate poses rendered and scored\n", (long long)evals);
    rope_destroy(ctx);
    return 0;
}
