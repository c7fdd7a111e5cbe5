/*
 * rope_s3d.h — C ABI of the MI355X render-and-compare pose engine (librope_hip.so).
 *
 * The reference (OSU-AIMS/RoPE-S3D) is pure Python; the calls it makes on this path
 * go to pyrender/OpenGL, Klampt, TensorFlow and numpy.  Each entry point below names
 * the reference call site(s) it stands in for.  Conventions: plain pointers and
 * sizes, caller-owned host buffers, return 0 on success or a negative ROPE_E_* code
 * (text via rope_last_error), no exceptions cross the boundary, one context per
 * GPU, a context is not thread-safe (the reference's Predictor is not re-entrant
 * either: one GL context and mutable target state, predict.py:127-148).
 */
#ifndef ROPE_S3D_H
#define ROPE_S3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rope_ctx rope_ctx;

enum {
    ROPE_OK = 0,
    ROPE_E_ARG = -1,      /* bad argument / wrong call order */
    ROPE_E_HIP = -2,      /* a HIP runtime call failed */
    ROPE_E_NOMEM = -3
};

/* Loss kinds (what is reduced over the pixels of one candidate render). */
enum {
    ROPE_LOSS_DEPTH = 0,  /* last term of Predictor._error only: mean(D[D!=0])*std(D)   predict.py:503-507 */
    ROPE_LOSS_FULL = 1,   /* whole Predictor._error                                      predict.py:475-509 */
    ROPE_LOSS_LOOKUP = 2, /* Lookup stage score mean|T-sqrt(D)|*std over the crop        predict.py:165-171 */
    ROPE_LOSS_TSWEEP = 3, /* TensorSweep score  mean|sqrt(T)-sqrt(D)| * -std, full frame predict.py:363-369 */
    ROPE_LOSS_CAMFULL = 4 /* CameraPredictor._error sums (rope_eval_views only)  camera_pose_prediction.py:933-970 */
};

#define ROPE_MAX_LINKS 6          /* link_6_t is never rendered: render_utils.py:31-32 */
#define ROPE_SUM_WORDS 23         /* uint64 words of integer sums per candidate */

/* Lifetime.  `device` is the HIP device ordinal. */
int rope_create(rope_ctx **out, int device);
void rope_destroy(rope_ctx *ctx);
const char *rope_last_error(rope_ctx *ctx);

/* Identity of the sources this library was built from (first 16 hex digits of their SHA-256, rope_s3d_amd/build.py:source_hash):
 * committed profiles carry the same string, so a benchmark can tell whether their counters belong to the build it is timing. */
const char *rope_build_id(void);

/* Robot geometry + kinematic chain, uploaded once.
 * Replaces MeshLoader.load + pyrender.Mesh.from_trimesh (render_utils.py:22-41) and
 * klampt.WorldModel(urdf) (kinematics.py:23-33).
 *   ml_header   n_meshlets x 8 uint32: centre xyz + radius (float bits), first vertex,
 *               first triangle, n_verts | n_tris<<16, link id
 *   ml_verts    n_ml_verts x 3 float32 (meshlet-local copies of link-frame vertices)
 *   ml_tris     n_ml_tris uint32, three 8-bit local indices each
 *   link_first  n_links+1 meshlet offsets
 *   joint_fixed 6 x 12 doubles, row-major 3x4 parent-link -> joint frame (<origin>)
 *   joint_axes  6 x 3 doubles, unit rotation axes (<axis>) */
int rope_set_robot(rope_ctx *ctx, const uint32_t *ml_header, int n_meshlets, const float *ml_verts,
                   int n_ml_verts, const uint32_t *ml_tris, int n_ml_tris, const int32_t *link_first,
                   int n_links, const double *joint_fixed, const double *joint_axes);

/* The same from plain meshes: partitions every link into meshlets on the host (rope_partition_mesh), builds the
 * arrays above and calls rope_set_robot — everything MeshLoader.load + from_trimesh leave to pyrender
 * (render_utils.py:22-41), for hosts that are not Python.
 *   verts    V x 3 float32, welded link meshes concatenated (link frame, metres)
 *   faces    T x 3 int32, vertex indices LOCAL to the triangle's link
 *   vtx_off / tri_off   n_links+1 offsets of every link into verts / faces */
int rope_set_robot_mesh(rope_ctx *ctx, const float *verts, const int32_t *faces, const int32_t *vtx_off, const int32_t *tri_off,
                        int n_links, const double *joint_fixed, const double *joint_axes);

/* One mesh -> surface patches of <= max_tris (<= 128) triangles over <= max_verts (<= 64) vertices.  Host only: no
 * context, no GPU.  tri_order receives the n_tris triangle ids grouped by patch, meshlet_first (n_tris+1 entries
 * available) the offsets of the patches into it; returns the number of patches or ROPE_E_ARG.  Any partition renders
 * the same image; this one keeps patches compact on the surface (small screen boxes). */
int rope_partition_mesh(const float *verts, int n_verts, const int32_t *faces, int n_tris, int max_tris, int max_verts,
                        int32_t *tri_order, int32_t *meshlet_first);

/* Host only (no context): camera pose + pinhole intrinsics -> the matrix P·V rope_set_camera takes (16 doubles, row-major).
 * Replaces Renderer.setCameraPose's pose convention (render.py:107-111: roll = a4 + pi/2, pitch = a3, yaw = a5; makePose /
 * angToPoseArr, render_utils.py:56-108: R = Rz(yaw)·Ry(pitch)·Rx(roll), camera looking along -Z with +Y up, V = pose^-1) and
 * Intrinsics.pyrender_camera (projection.py:161-169: pyrender 0.1.45's IntrinsicsCamera projection).
 *   pose  [x, y, z, a3, a4, a5]: position in metres, the three angles in radians, as the reference's camera_pose arrays */
int rope_camera_matrix(const double *pose, double fx, double fy, double cx, double cy, int W, int H, double znear, double zfar, double *PV);

/* Host only (no context): the Lookup stage's pose grid (RobotLookupCreator / RobotLookupManager, lookup.py:39-66): divisions[j] >= 1
 * samples of joint j between its limits (6 x 2 doubles), np.linspace's values, joint 0 varying fastest; divisions[j] <= 0: joint j
 * is not part of the grid (angle 0).  Returns the number of rows; writes rows x 6 doubles to `out` when it is not NULL and
 * `capacity` rows fit (else ROPE_E_ARG). */
int64_t rope_lookup_grid(const double *limits, const int32_t *divisions, double *out, int64_t capacity);

/* Host only (no context): the divisions (rope_lookup_grid's convention) of the pose grid Crop renders to find the image bounds of the
 * first num_links links (2..6) of the robot at an image of n_pixels pixels (robotpose/crop.py:114-146; constants.py:19-23).  With
 * rope_lookup_grid and rope_coverage this replaces Crop._create (crop.py:50-96): bounds = box of the covered pixels, padded by 10. */
int rope_crop_divisions(int64_t n_pixels, int num_links, int32_t *divisions);

/* Camera: PV = P·V (4x4 row-major doubles), image size and clip planes.
 * Replaces Renderer.setCameraPose (render.py:107-111) + Intrinsics.pyrender_camera
 * (projection.py:161-169) + pyrender.OffscreenRenderer(W,H) (render.py:60). */
int rope_set_camera(rope_ctx *ctx, const double *PV, int W, int H, double znear, double zfar);

/* Target of the current frame.
 * Replaces the cached state of Predictor._load_target/_loadSynthetic (predict.py:397-413,445-469).
 *   tq         H x W uint64: bits 0..38 target depth in Q32 metres, bits 40..47 link mask bits
 *   t32        H x W float32 plane for ROPE_LOSS_LOOKUP (lookup target) / ROPE_LOSS_TSWEEP
 *              (full target); may be NULL when those losses are not used
 *   link_flags 8 bytes: bit0 link present in the target, bit1 ">5 % of mask has depth" (predict.py:495) */
int rope_set_target(rope_ctx *ctx, const uint64_t *tq, const float *t32, const uint8_t *link_flags);

/* The float32 plane ROPE_LOSS_TSWEEP reads, when it is not the lookup plane: TensorSweep compares against the WHOLE target depth
 * (predict.py:358-363, `self._tgt_depth`), the Lookup stage against the depth under the lookup links (predict.py:165-169).  Call after
 * rope_set_target (which forgets it); NULL goes back to the one plane. */
int rope_set_target_tsweep(rope_ctx *ctx, const float *t32_full);

/* Host only (no context): the packing rope_set_target expects.  depth n float64 metres (NaN, inf and values <= 0 count
 * as "no depth"), mask_bits n bytes (bit l = link l's mask) or NULL -> out n uint64: Q32 depth, round half even,
 * clipped to 2^39-1, mask bits at 40..47. */
int rope_pack_target(const double *depth, const uint8_t *mask_bits, int64_t n, uint64_t *out);

/* Host only (no context): cv2.resize(img, (W/f, H/f)) with INTER_LINEAR for an even integer factor f — the frame
 * down-sampling of Predictor._downsample (predict.py:378-381); four taps per output sample, OpenCV's fixed-point
 * rounding for uint8.  kind 0 uint8, 1 float32, 2 float64; `channels` interleaved; rows `row_stride` bytes apart;
 * dst is dense (H/f x W/f x channels). */
int rope_downsample_even(const void *src, int H, int W, int channels, int64_t row_stride, int f, int kind, void *dst);

/* Host only (no context): one frame of the synthetic path, camera arrays in, target planes out, in one pass: the down-sampling
 * of Predictor._downsample (predict.py:378-381, as rope_downsample_even), the link masks read off channel 0 of the colour render
 * and the lookup depth of _loadSynthetic (predict.py:445-469), the per-link flags and the packing of _load_target (predict.py:397-413).
 *   color  H0 x W0 x 3 uint8 (BGR, only channel 0 is read), rows color_stride bytes apart
 *   depth  H0 x W0 float32 (depth_kind 1) or float64 (2) metres, rows depth_stride bytes apart
 *   f      down-sampling factor: 1 or even;  link_blue: n_links channel-0 values (DEFAULT_RENDER_COLORS, constants.py:65-91), the first
 *          n_lookup_links of them the links of the lookup depth
 *   out    tq, lookup_f32 (H0/f x W0/f), flags (8 bytes) as rope_set_target takes them; tgt_depth float64 or NULL */
int rope_prepare_synthetic(const uint8_t *color, int64_t color_stride, const void *depth, int depth_kind, int64_t depth_stride, int H0, int W0,
                           int f, const int32_t *link_blue, int n_links, int n_lookup_links, uint64_t *tq, float *lookup_f32, double *tgt_depth,
                           uint8_t *flags);

/* Page-locked host memory for the planes handed to rope_set_target / rope_set_targets / rope_set_frames: copies out of it go to the
 * device in one transfer at the link's rate, instead of chunk by chunk through the library's own staging block (what pageable
 * memory needs).  Optional — any host pointer is accepted everywhere.  NULL when the allocation fails. */
void *rope_host_alloc(size_t bytes);
void rope_host_free(void *p);

/* Host only (no context): one frame of the SEGMENTATION path, segmenter output in, target planes out: the merge of a class's
 * instances (_reorganize_by_link, predict.py:383-395), the body mask erode7(dilate8(sum of the masks)) applied to the depth — over
 * all detected classes for the target, over the lookup links for the lookup depth (predict.py:419-438; cv2's box kernels with their
 * default anchors) — the depth's down-sampling (predict.py:378-381), the flags and the packing of _load_target (predict.py:397-413).
 *   depth    H0 x W0 float32 (kind 1) / float64 (kind 2), rows depth_stride bytes apart;  f: 1 or even
 *   masks    (H0/f) x (W0/f) x K bytes, non-zero = inside instance k (the segmenter's `masks`);  link_of: K link indices
 *            (class id - 1), -1 for an instance that is none of the rendered links (it still widens the body mask)
 *   out      tq, lookup_f32, flags as rope_set_target takes them; tgt_depth float64 (the masked depth) or NULL */
int rope_prepare_segmented(const void *depth, int depth_kind, int64_t depth_stride, int H0, int W0, int f, const uint8_t *masks, int K,
                           const int32_t *link_of, int n_links, int n_lookup_links, uint64_t *tq, float *lookup_f32, double *tgt_depth,
                           uint8_t *flags);

/* Candidate joint vectors (C x 6 doubles) into HBM; they stay resident until replaced. */
int rope_candidates_upload(rope_ctx *ctx, const double *cand, int C);

/* FK + raster of the first n_render links + loss reduction + argmin for the resident
 * candidates, enqueued on the context's stream (no host sync).
 * Replaces, per candidate, render_at_pos + _error (predict.py:159-161,475-509) and, for the
 * lookup loss, the TF reduction over the pre-rendered table (predict.py:167-171).
 *   crop   r0,r1,c0,c1 inclusive image rows/cols; required for ROPE_LOSS_LOOKUP, else NULL */
int rope_eval_resident(rope_ctx *ctx, int n_render, int loss, const int32_t *crop);

int rope_sync(rope_ctx *ctx);

/* Results of the last rope_eval_resident.  Any pointer may be NULL.
 *   err_out  C doubles; sums_out C x ROPE_SUM_WORDS uint64; best_idx/best_err = first argmin */
int rope_results_download(rope_ctx *ctx, double *err_out, uint64_t *sums_out, int32_t *best_idx, double *best_err);

/* upload + eval + sync + download in one call (host buffers in, host buffers out). */
int rope_eval(rope_ctx *ctx, const double *cand, int C, int n_render, int loss, const int32_t *crop,
              double *err_out, uint64_t *sums_out, int32_t *best_idx, double *best_err);

/* Stored lookup table.  rope_lookup_build renders the pose grid once (first n_render links) and keeps the
 * cropped sqrt-depth images in HBM (C x crop_h x crop_w float32); rope_lookup_score streams the table against
 * the current target's float32 plane and returns the lookup score of every row and the first argmin.
 * Replaces RobotLookupCreator.run / RobotLookupManager.get (lookup.py:69-106,184-283), tf.pow(depth, 0.5)
 * (predict.py:117) and the TF reduction + argmin per frame (predict.py:167-171).  Scores are bit-identical to
 * rope_eval with ROPE_LOSS_LOOKUP on the same grid and crop. */
int rope_lookup_build(rope_ctx *ctx, const double *cand, int C, int n_render, const int32_t *crop);
int rope_lookup_score(rope_ctx *ctx, double *scores_out, int32_t *best_idx, double *best_score);

/* One pose to images.  Replaces Renderer.setJointAngles + Renderer.render (render.py:88-98).
 *   depth H x W float32 metres (0 = empty), ids H x W uint8 link id (255 = background) */
int rope_render(rope_ctx *ctx, const double *q, int n_render, float *depth, uint8_t *ids);

/* OR over candidates of "pixel covered" (H x W uint8 0/1).  Replaces the depth-sum loop
 * of Crop._create (crop.py:60-81). */
int rope_coverage(rope_ctx *ctx, const double *cand, int C, int n_render, uint8_t *cover);

/* Camera-pose path: N frames (a known joint vector + target planes each) scored under K candidate cameras.
 * Replaces do_renders_at_pose + _error of ModellessCameraPredictor / CameraPredictor
 * (camera_pose_prediction.py:116-124,389-427,656-664,933-970): K x N renders and reductions in one batch.
 *   q            N x 6 joint vectors (robot_poses)
 *   tq           N planes H x W uint64, bits 0..38 target depth in Q32 metres
 *   t32          N planes H x W float32 target depth (ROPE_LOSS_TSWEEP: |sqrt(T)-sqrt(D)|), or NULL
 *   link_planes  N x 6 planes H x W uint64 (ROPE_LOSS_CAMFULL): bit 40 = link mask, bits 0..38 = mask * depth in
 *                Q32 metres (_masked_targets / _target_masks, camera_pose_prediction.py:919-931), or NULL
 * rope_eval_views: PV = K matrices P·V (row-major doubles); sums_out = K x N x ROPE_SUM_WORDS exact integer sums,
 * candidate order view-major.  Words: ROPE_LOSS_DEPTH / ROPE_LOSS_TSWEEP as rope_results_download;
 * ROPE_LOSS_CAMFULL: [0] #(D!=0), [1] sum |T-D| (Q32), [2] sum floor(sqrt|T-D| * 2^32), per link l at 5+3l:
 * #(M_l != R_l), #(d_l != 0), sum floor(sqrt(d_l) * 2^32).  The float epilogue stays with the caller. */
int rope_set_frames(rope_ctx *ctx, int n_frames, const double *q, const uint64_t *tq, const float *t32,
                    const uint64_t *link_planes);
int rope_eval_views(rope_ctx *ctx, const double *PV, int K, int n_render, int loss, uint64_t *sums_out);

/* The whole per-frame stage machine on the host, driving device batches through the calls above.
 * Replaces the stage loop of Predictor.run (predict.py:144-375) for the stages of the reference's 'SL' and 'SLU'
 * lists (stages.py:128-178); the target must have been set with rope_set_target (full loss masks + lookup plane).
 * Decisions follow the reference operation for operation, quirks included (SFlip's aliasing and its comparison
 * outside the endpoint loop, ISweep's stale base error, Descent's error history of the last joint only); the
 * not-a-knot cubic of interp1d(kind='cubic') (predict.py:310) is solved directly.  TensorSweep (in neither list, but a stage the
 * reference defines) is ROPE_STAGE_TSWEEP. */
enum {
    ROPE_STAGE_LOOKUP = 0,   /* stages.py:16-24   predict.py:165-171 */
    ROPE_STAGE_DESCENT = 1,  /* stages.py:92-119  predict.py:173-230 */
    ROPE_STAGE_SFLIP = 2,    /* stages.py:30-41   predict.py:232-281 */
    ROPE_STAGE_ISWEEP = 3,   /* stages.py:50-69   predict.py:283-338 */
    ROPE_STAGE_TSWEEP = 4    /* stages.py:71-90   predict.py:340-373: `count` divisions over `joints`, `range` as InterpolativeSweep; scored
                                with ROPE_LOSS_TSWEEP against the plane of rope_set_target_tsweep (the *- sign kept: the largest mean*std wins) */
};

typedef struct rope_stage {
    int32_t kind;
    int32_t to_render;        /* links drawn: 4 or 6 (Lookup: LOOKUP_NUM_RENDERED) */
    int32_t count;            /* Descent: iterations; InterpolativeSweep: divisions (>= 4) */
    uint32_t joints;          /* bit j set: joint j (S=0 .. T=5) takes part */
    double init_rate[6];      /* Descent: starting step per joint, NaN = keep the running step (None) */
    double rate_reduction;    /* Descent */
    double early_stop;        /* Descent */
    double range;             /* InterpolativeSweep: rad about the current angle, NaN = the whole joint range */
} rope_stage;

typedef struct rope_predict_args {
    const rope_stage *stages;
    int32_t n_stages;
    int32_t speculate;            /* Descent: joints whose under/over pairs go out as one batch, 1..3 (1 = the
                                     reference's two renders at a time; the decisions are the same either way) */
    const double *limits;         /* 6 x 2 joint limits (URDFReader.joint_limits) */
    const double *camera_pose;    /* 6: x y z + the three angles, as Predictor.camera_pose (SFlip's axis, predict.py:245) */
    const double *min_ang_inc;    /* 6 (Predictor.min_ang_inc) */
    const double *lookup_angles;  /* n_lookup x 6 pose grid (lookup.py:56-66 order) */
    int32_t n_lookup;
    int32_t use_table;            /* 1: score the table of rope_lookup_build; 0: render and score the grid now */
    const int32_t *lookup_crop;   /* r0,r1,c0,c1 when use_table == 0 */
    double *lookup_angles_live;   /* NULL (default): every frame starts from the grid row itself and frames are independent.
                                     Non-NULL: the reference's table aliasing, opt-in — an n_lookup x 6 copy of the grid that the
                                     caller keeps from frame to frame.  The Lookup stage takes its row from HERE (scores still
                                     come from the grid / the stored table), and while the current angles are still that row —
                                     until an SFlip or a sweep rebinds them — Descent's steps are written back into it, as
                                     `angles = self.lookup_angles[argmin]` (a numpy view, predict.py:171) followed by
                                     `angles[idx] += rate` (predict.py:212-215) does in the reference.  Results then depend on
                                     the order of the frames: one context, frames in sequence. */
} rope_predict_args;

/*   angles_out  6 doubles
 *   trace_out   n_stages x 6 doubles, the angles after every stage; may be NULL
 *   n_evals     candidate poses rendered and scored (lookup rows included); may be NULL */
int rope_predict(rope_ctx *ctx, const rope_predict_args *args, double *angles_out, double *trace_out, int64_t *n_evals);

/* ---- Many frames at once.  The reference predicts a dataset frame by frame (predict_dataset.py:43-44), each frame a chain of ~25
 * dependent render batches of 2-26 poses (predict.py:173-338).  Frames are independent (fresh state per frame, predict.py:144-148),
 * so B of them can walk the stage list in lockstep: every step then is ONE device batch holding the rows of all B frames, each row
 * scored against its own frame's target.
 *
 * rope_set_targets: the targets of n_frames frames, resident until replaced (they share buffers with rope_set_frames: one or the other).
 *   tq          n_frames planes H x W uint64 (as rope_set_target)
 *   t32         n_frames planes H x W float32 (lookup planes) or NULL
 *   t32_tsweep  n_frames planes H x W float32 for ROPE_LOSS_TSWEEP (rope_set_target_tsweep) or NULL
 *   link_flags  n_frames x 8 bytes (as rope_set_target)
 * rope_eval_targets: R candidate rows, row i scored against frame frame_of[i]'s target -> err_out[i]; loss DEPTH / FULL / LOOKUP
 *   (crop required) / TSWEEP.  Every error has the bits rope_eval gives with that frame as the single target.
 * rope_lookup_score_targets: the stored table of rope_lookup_build against every resident target in one pass: per frame the first
 *   argmin row and (optionally) its score; scores_out: n_frames x table rows or NULL.  Same bits as rope_lookup_score per frame. */
int rope_set_targets(rope_ctx *ctx, int n_frames, const uint64_t *tq, const float *t32, const float *t32_tsweep, const uint8_t *link_flags);
int rope_eval_targets(rope_ctx *ctx, const double *cand, const int32_t *frame_of, int R, int n_render, int loss, const int32_t *crop,
                      double *err_out);
int rope_lookup_score_targets(rope_ctx *ctx, int32_t *best_idx, double *best_score, double *scores_out);

/* The next set of targets on its way up while the resident set is being predicted (a dataset in groups of frames: the upload of
 * group k+1 hidden behind the stages of group k).  The context holds two sets of target planes.
 * rope_stage_targets: arguments as rope_set_targets, into the set NOT in use, on a stream of its own; returns when the copies are
 *   enqueued (sources from rope_host_alloc; pageable sources are copied through a pinned block and the call returns when the last
 *   chunk is enqueued).  May be called from another thread than the one evaluating on this context.  The host buffers belong to
 *   the copy until rope_commit_targets returns.
 * rope_commit_targets: waits for the upload and for the context's work, then makes the staged set the resident one (as if
 *   rope_set_targets had been called with it).  ROPE_E_ARG when nothing is staged or the image size changed in between. */
int rope_stage_targets(rope_ctx *ctx, int n_frames, const uint64_t *tq, const float *t32, const float *t32_tsweep, const uint8_t *link_flags);
int rope_commit_targets(rope_ctx *ctx);

/* rope_predict for the n_frames resident targets of rope_set_targets, in lockstep: the same stage list, limits and camera for all of
 * them (args as rope_predict; lookup_angles_live must be NULL — the table aliasing makes frames depend on their order).  A frame
 * that leaves a Descent stage early simply contributes no rows to the later batches of that stage.  Every frame's angles and trace
 * are those of rope_predict on that frame alone.
 *   angles_out  n_frames x 6;  trace_out  n_frames x n_stages x 6 or NULL;  n_evals  total poses rendered and scored, or NULL */
int rope_predict_batch(rope_ctx *ctx, const rope_predict_args *args, int n_frames, double *angles_out, double *trace_out, int64_t *n_evals);

/* Device-side per-candidate link matrices of the last eval (C x n_render x 16 float32), for tests. */
int rope_debug_mvp(rope_ctx *ctx, float *mvp_out, int C, int n_render);

/* Time `reps` back-to-back rope_eval_resident passes with HIP events on the context's
 * stream; ms[0] = FK + bounds kernels, ms[1] = shared-layer raster launch (0 when layers are not in use),
 * ms[2] = raster+score launch, ms[3] = finalize+argmin, ms[4] = whole pass (averages per pass, milliseconds). */
int rope_profile_eval(rope_ctx *ctx, int n_render, int loss, const int32_t *crop, int reps, float *ms);

/* Execution strategy: how a batch is laid out over launches.  Every value gives bit-identical results (the depth test is
 * a minimum and the sums are exact integers); the equivalence tests and the "unshared" bench figure use it.
 *   1  do not share links 0-2 between candidates with equal (q0, q1): six links drawn per candidate
 *   2  do not split the meshlets of a tile over several workgroups for small batches
 *   4  no second level of sharing (links 0-1 per distinct q0)
 *   8  large batches as one workgroup per (tile, candidate) pair instead of a queue of the pairs that have work
 *  16  always the raster kernels that can clip triangles at the near plane (by default only when the camera is within the
 *      robot's reach of it: they are several per cent slower, and without a triangle at the plane they draw the same image)
 *  32  small batches: forward kinematics and screen boxes as a launch of their own instead of inside the split raster's workgroups */
enum { ROPE_STRATEGY_NO_LAYERS = 1, ROPE_STRATEGY_NO_SPLIT = 2, ROPE_STRATEGY_NO_PARENTS = 4, ROPE_STRATEGY_NO_QUEUE = 8, ROPE_STRATEGY_CLIP_KERNELS = 16,
       ROPE_STRATEGY_SEPARATE_GEOMETRY = 32 };
int rope_set_strategy(rope_ctx *ctx, int flags);

/* ---- Segmentation stage (robotpose/prediction/predict.py:94-98,416: the Matterport Mask R-CNN in front of the engine).
 * The convolutions are library calls from PyTorch-ROCm (rope_s3d_amd/maskrcnn.py); the two box-shaped steps between them
 * are kernels of this library.  No context: plain device pointers and the HIP stream to launch on (hipStream_t, NULL = default).
 *
 * rope_seg_nms: greedy non-maximum suppression (tf.image.non_max_suppression, as the ProposalLayer and
 * refine_detections_graph of mrcnn/model.py call it) of n_sets independent sets of n boxes each.
 *   boxes    n_sets x n x 4 float32 (y1, x1, y2, x2), each set sorted by DESCENDING score; 16-byte aligned
 *   groups   n_sets x n int32 or NULL: boxes only suppress boxes of the same group (per-class NMS)
 *   valid    n_sets x n bytes or NULL: 0 = padding (never kept, never suppressing)
 *   limit    at most this many boxes kept per set (the best ones)
 *   scratch  n_sets x n x ceil(n / 64) uint64 of device memory
 *   keep     n_sets x n bytes out: 1 = kept (in the order of `boxes`)
 *
 * rope_seg_roi_align: PyramidROIAlign — the pyramid level by box area, then tf.image.crop_and_resize (bilinear,
 * pool x pool samples spanning the box, corners included, 0 outside the map).
 *   rows_bf16       the pyramid levels P2..P5 of all frames as one table of channel rows (bfloat16, `channels` per
 *                   feature pixel), level l starting at row level_off[l], frames back to back inside a level
 *   boxes           n_boxes x 4 float32 normalised (y1, x1, y2, x2); 16-byte aligned;  frame: n_boxes int32 batch entry
 *   level_hw        4 x (H, W) int32;  level_off 4 int64
 *   inv_level_unit  1 / (224 / image_size) as float32;  t: `pool` float32 sample positions in [0, 1] (device)
 *   out_bf16        n_boxes x pool x pool x channels bfloat16 */
/* rope_seg_bias_act: what follows a convolution, in place and in one pass: y <- [relu](bf16(bf16(y + bias[channel]) [+ residual])),
 * with the roundings separate tensor operations would make.
 *   y_bf16    n bfloat16 (n a multiple of 8), NCHW (`inner` = H*W, a multiple of 8) or channels-last (`inner` = 1, channels a
 *             multiple of 8);  bias_bf16: `channels` bfloat16;  residual_bf16: n bfloat16 in y's layout, or NULL;  relu: 0 / 1 */
int rope_seg_bias_act(void *y_bf16, const void *bias_bf16, const void *residual_bf16, int64_t n, int channels, int64_t inner, int relu,
                      void *stream);
int rope_seg_nms(const float *boxes, const int32_t *groups, const uint8_t *valid, int n_sets, int n, float iou_thr, int limit,
                 uint64_t *scratch, uint8_t *keep, void *stream);
int rope_seg_roi_align(const void *rows_bf16, const float *boxes, const int32_t *frame, const int32_t *level_hw,
                       const int64_t *level_off, int n_boxes, int channels, int pool, float inv_level_unit, const float *t,
                       void *out_bf16, void *stream);

/* Phase-skipping switches for kernel ablations (rope_debug_skip) exist only in the profiling build of the library
 * (librope_hip_profile.so, `python tools/build_variants.py profile`, -DROPE_PROFILE); this library does not export them. */

#ifdef __cplusplus
}
#endif
#endif
