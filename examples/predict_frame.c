/*
 * predict_frame.c — a host that is not Python: one RGB-D frame through the whole prediction path using nothing but the C ABI
 * of librope_hip.so (include/rope_s3d.h), starting where the reference's Predictor starts: the robot's meshes and joint chain,
 * a camera pose, the camera's intrinsics, and the frame as the camera delivers it.
 *
 *     gcc -O2 -Iinclude examples/predict_frame.c -Lrope_s3d_amd/csrc -lrope_hip \
 *         -Wl,-rpath,$PWD/rope_s3d_amd/csrc -Wl,-rpath-link,/opt/rocm/lib -lm -o predict_frame
 *     ./predict_frame <bundle directory>        # prints the six joint angles, %.17g
 *
 * The bundle is a directory of raw little-endian arrays (tools/dump_frame_bundle.py writes one) — INPUTS only: the welded link
 * meshes and the joint chain (what URDFReader + MeshLoader give the reference), the joint limits, the camera pose, the base
 * intrinsics and the down-sampling factor, the colour-coded frame and its depth, the links' colours, the lookup grid's
 * divisions.  No matrix, crop, grid or packed target comes from Python: this program derives them through the library.
 *
 *   Predictor.__init__ (robotpose/prediction/predict.py:38-124)
 *     Intrinsics.downscale + the string round trip of Renderer's constructor (projection.py:127-136,20-46; render.py:41)    here
 *     Renderer.setCameraPose + IntrinsicsCamera (render.py:107-111, projection.py:161-169)        rope_camera_matrix, rope_set_camera
 *     MeshLoader / Klampt world (render_utils.py:22-41, kinematics.py:23-33)                    rope_set_robot_mesh
 *     Crop (crop.py:50-146)                          rope_crop_divisions + rope_lookup_grid + rope_coverage, box of the covered pixels
 *     RobotLookupManager.get + the depth table (lookup.py:39-106,184-283)                      rope_lookup_grid, rope_lookup_build
 *   Predictor.run (predict.py:127-375)
 *     _downsample + _loadSynthetic + _load_target (predict.py:378-381,445-469,397-413)          rope_prepare_synthetic, rope_set_target
 *     the 'SLU' stage list (stages.py:152-168)                                                  rope_predict
 *   the frame loop of predict_dataset.py:43-44 (optional second argument)          rope_host_alloc, rope_stage_targets / rope_commit_targets,
 *                                                                                  rope_predict_batch
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

/* The reference's Renderer rebuilds the Predictor's down-scaled Intrinsics from their printed form (render.py:41,
 * projection.py:20-46,183-184): librealsense prints six significant digits, so that is what the renderer sees. */
static double as_printed(double x)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%g", x);
    return strtod(buf, NULL);
}

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rope_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <bundle directory> [frames per lockstep batch]\n", argv[0]); return 2; }
    const char *dir = argv[1];
    size_t n_vo, n_color, n_depth;
    float *verts = load(dir, "verts.f32", 4, NULL);
    int32_t *faces = load(dir, "faces.i32", 4, NULL);
    int32_t *vtx_off = load(dir, "vtx_off.i32", 4, &n_vo), *tri_off = load(dir, "tri_off.i32", 4, NULL);
    double *joint_fixed = load(dir, "joint_fixed.f64", 8, NULL), *joint_axes = load(dir, "joint_axes.f64", 8, NULL);
    double *limits = load(dir, "limits.f64", 8, NULL), *camera_pose = load(dir, "camera_pose.f64", 8, NULL);
    double *intr = load(dir, "intrinsics.f64", 8, NULL);            /* W0, H0, ppx, ppy, fx, fy of the camera's full-size image */
    int32_t *setup = load(dir, "setup.i32", 4, NULL);               /* down-sampling factor, lookup divisions per S/L/U joint */
    int32_t *link_blue = load(dir, "link_blue.i32", 4, NULL);       /* channel 0 of each of the six rendered links' colours */
    uint8_t *color = load(dir, "color.u8", 1, &n_color);
    float *depth = load(dir, "depth.f32", 4, &n_depth);
    const int W0 = (int)intr[0], H0 = (int)intr[1], ds = setup[0], div = setup[1];
    const double znear = 0.05, zfar = 100.0;                        /* pyrender's IntrinsicsCamera defaults (projection.py:161-169) */
    if (ds < 1 || W0 % ds || H0 % ds || n_depth != (size_t)W0 * H0 || n_color != 3 * n_depth) {
        fprintf(stderr, "frame does not match %dx%d / %d\n", W0, H0, ds);
        return 2;
    }
    const int W = W0 / ds, H = H0 / ds;
    const double cx = as_printed(intr[2] / ds), cy = as_printed(intr[3] / ds), fx = as_printed(intr[4] / ds), fy = as_printed(intr[5] / ds);
    double PV[16];
    rope_ctx *ctx = NULL;
    if (rope_camera_matrix(camera_pose, fx, fy, cx, cy, W, H, znear, zfar, PV)) { fprintf(stderr, "rope_camera_matrix: bad camera\n"); return 2; }

    CHECK(rope_create(&ctx, 0));
    CHECK(rope_set_robot_mesh(ctx, verts, faces, vtx_off, tri_off, (int)n_vo - 1, joint_fixed, joint_axes));
    CHECK(rope_set_camera(ctx, PV, W, H, znear, zfar));

    /* Crop of the six rendered links: the box of every pixel any pose of the crop grid covers, padded by 10 (crop.py:98-112) */
    int32_t cdiv[6], crop[4];
    if (rope_crop_divisions((int64_t)W * H, 6, cdiv)) { fprintf(stderr, "rope_crop_divisions failed\n"); return 1; }
    const int64_t n_crop = rope_lookup_grid(limits, cdiv, NULL, 0);
    double *crop_grid = malloc((size_t)n_crop * 6 * sizeof(double));
    uint8_t *cover = malloc((size_t)W * H), *any = calloc((size_t)W * H, 1);
    if (!crop_grid || !cover || !any || rope_lookup_grid(limits, cdiv, crop_grid, n_crop) != n_crop) { fprintf(stderr, "crop grid failed\n"); return 1; }
    for (int64_t lo = 0; lo < n_crop; lo += 32768) {
        const int n = (int)(n_crop - lo < 32768 ? n_crop - lo : 32768);
        CHECK(rope_coverage(ctx, crop_grid + 6 * lo, n, 6, cover));
        for (size_t i = 0; i < (size_t)W * H; i++) any[i] |= cover[i];
    }
    int r0 = H, r1 = -1, c0 = W, c1 = -1;
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++)
            if (any[(size_t)r * W + c]) { if (r < r0) r0 = r; if (r > r1) r1 = r; if (c < c0) c0 = c; if (c > c1) c1 = c; }
    if (r1 < 0) { fprintf(stderr, "the robot is not in view\n"); return 1; }
    crop[0] = r0 - 10 > 0 ? r0 - 10 : 0; crop[1] = r1 + 10 < H - 1 ? r1 + 10 : H - 1;
    crop[2] = c0 - 10 > 0 ? c0 - 10 : 0; crop[3] = c1 + 10 < W - 1 ? c1 + 10 : W - 1;

    /* Lookup grid over S, L, U and its table of cropped depth images, once per camera pose (lookup.py:39-106) */
    const int32_t ldiv[6] = {div, div, div, 0, 0, 0};
    const int64_t n_grid = rope_lookup_grid(limits, ldiv, NULL, 0);
    double *grid = malloc((size_t)n_grid * 6 * sizeof(double));
    if (!grid || rope_lookup_grid(limits, ldiv, grid, n_grid) != n_grid) { fprintf(stderr, "lookup grid failed\n"); return 1; }
    CHECK(rope_lookup_build(ctx, grid, (int)n_grid, 6, crop));

    /* The frame: down-sampled, link masks read off the colour render, packed (once per frame) */
    uint64_t *tq = malloc((size_t)W * H * sizeof(uint64_t));
    float *t32 = malloc((size_t)W * H * sizeof(float));
    uint8_t flags[8];
    if (!tq || !t32 || rope_prepare_synthetic(color, (int64_t)3 * W0, depth, 1, (int64_t)4 * W0, H0, W0, ds, link_blue, 6, 6, tq, t32, NULL, flags)) {
        fprintf(stderr, "rope_prepare_synthetic failed\n");
        return 1;
    }
    CHECK(rope_set_target(ctx, tq, t32, flags));

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
    a.lookup_angles = grid; a.n_lookup = (int)n_grid; a.use_table = 1; a.lookup_crop = crop;
    double angles[6];
    int64_t evals = 0;
    CHECK(rope_predict(ctx, &a, angles, NULL, &evals));
    for (int j = 0; j < 6; j++) printf("%.17g%c", angles[j], j == 5 ? '\n' : ' ');
    fprintf(stderr, "crop %d %d %d %d, %lld lookup poses, %lld candidate poses rendered and scored\n", crop[0], crop[1], crop[2], crop[3],
            (long long)n_grid, (long long)evals);

    /* Optional second argument B: the frame loop of predict_dataset.py:43-44 as lockstep batches (rope_predict_batch) — two groups of
     * B copies of the frame, each prepared into page-locked planes (rope_host_alloc), the second group's planes going up on the
     * library's second stream (rope_stage_targets) while the first group is predicted.  Every copy must come out as the frame did. */
    const int B = argc > 2 ? atoi(argv[2]) : 0;
    if (B > 0) {
        const size_t plane = (size_t)W * H;
        uint64_t *btq[2];
        float *bt32[2];
        uint8_t *bfl[2];
        for (int k = 0; k < 2; k++) {
            btq[k] = rope_host_alloc(B * plane * sizeof(uint64_t));
            bt32[k] = rope_host_alloc(B * plane * sizeof(float));
            bfl[k] = malloc((size_t)B * 8);
            if (!btq[k] || !bt32[k] || !bfl[k]) { fprintf(stderr, "no page-locked memory\n"); return 1; }
        }
        double *out = malloc((size_t)2 * B * 6 * sizeof(double));
        if (!out) return 1;
        a.speculate = B >= 16 ? 1 : 3;                    /* hundreds of rows per step anyway: the reference's own two renders at a time */
        for (int g = 0; g < 2; g++) {
            if (g == 0) {
                for (int i = 0; i < B; i++)
                    if (rope_prepare_synthetic(color, (int64_t)3 * W0, depth, 1, (int64_t)4 * W0, H0, W0, ds, link_blue, 6, 6, btq[0] + i * plane,
                                               bt32[0] + i * plane, NULL, bfl[0] + 8 * i)) return 1;
                CHECK(rope_stage_targets(ctx, B, btq[0], bt32[0], NULL, bfl[0]));
            }
            CHECK(rope_commit_targets(ctx));
            if (g == 0) {                                 /* group 1 on its way up while group 0 is predicted */
                for (int i = 0; i < B; i++)
                    if (rope_prepare_synthetic(color, (int64_t)3 * W0, depth, 1, (int64_t)4 * W0, H0, W0, ds, link_blue, 6, 6, btq[1] + i * plane,
                                               bt32[1] + i * plane, NULL, bfl[1] + 8 * i)) return 1;
                CHECK(rope_stage_targets(ctx, B, btq[1], bt32[1], NULL, bfl[1]));
            }
            CHECK(rope_predict_batch(ctx, &a, B, out + (size_t)g * B * 6, NULL, NULL));
        }
        int same = 1;
        for (int i = 0; i < 2 * B; i++) same = same && memcmp(out + 6 * i, angles, sizeof angles) == 0;
        fprintf(stderr, "%d frames in two lockstep batches: %s\n", 2 * B, same ? "every one equal to the single frame" : "MISMATCH");
        for (int k = 0; k < 2; k++) { rope_host_free(btq[k]); rope_host_free(bt32[k]); free(bfl[k]); }
        free(out);
        if (!same) return 3;
    }
    rope_destroy(ctx);
    return 0;
}
