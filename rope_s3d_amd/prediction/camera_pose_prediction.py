"""Camera-pose prediction by render-and-compare: N frames of a robot at known joint angles, one unknown camera.

Keeps the constructors and `run` signatures of the reference's `ModellessCameraPredictor` and `CameraPredictor`
(robotpose/prediction/camera_pose_prediction.py:28-431,576-975).  The reference renders every frame under every
trial camera pose one GL draw at a time (do_renders_at_pose, :116-124,656-664) and reduces with numpy / TF
(:389-427,841-852,933-970).  Here a trial is a batch: all K poses a stage wants to compare times all N frames go
to the HIP engine as one `rope_eval_views` call (K·N candidates, each scored against its own frame's target
planes); what comes back are exact integer sums, and the short float epilogues below turn them into the
reference's error values.  The stage machine follows the reference decision for decision; its quirks are kept and
cited where they appear.
"""
from typing import Callable, List, Optional

import numpy as np

from ..constants import DEFAULT_CAMERA_POSE, DEFAULT_RENDER_COLORS, ZFAR, ZNEAR
from ..engine import LOSS_CAMFULL, LOSS_TSWEEP, pack_target
from ..imgproc import resize_linear
from ..projection import view_matrix
from ..simulation.render import Renderer
from ..urdf import URDFReader
from .predict import cubic_interp

S_CNT, S_S1, S_AA, S_AB, S_BB, S_LINK0 = 0, 1, 2, 3, 4, 5
_XYZ, _RPY = [True, True, True, False, False, False], [False, False, False, True, True, True]


# ------------------------------------------------------------------------------ float epilogues
def sweep_mean_std(sums: np.ndarray, n_pix: float):
    """(..., 23) integer sums of |sqrt(T) - sqrt(D)| in Q32 -> (mean, population std) over n_pix pixels."""
    s = sums.astype(np.float64)
    m1 = (s[..., S_S1] * 2.0 ** -32) / n_pix
    s2 = (s[..., S_AA] * 2.0 ** 40 + s[..., S_AB] * 2.0 ** 21) + s[..., S_BB]
    var = np.maximum((s2 * 2.0 ** -64) / n_pix - m1 * m1, 0.0)
    return m1, np.sqrt(var)


def modelless_error(sums: np.ndarray, n_pix: float) -> np.ndarray:
    """(K, N, 23) -> (K,): mean over frames of 1.1 ** (mean * -std) (camera_pose_prediction.py:404-408,420-424)."""
    m1, sd = sweep_mean_std(sums, n_pix)
    return np.mean(np.power(1.1, m1 * -sd), axis=-1)


def pooled_sweep_error(sums: np.ndarray, n_pix: float) -> np.ndarray:
    """(K, N, 23) -> (K,): mean * -std of the differences of all frames together (:841-846,880-884)."""
    pooled = sums.sum(axis=1, dtype=np.uint64)
    m1, sd = sweep_mean_std(pooled, n_pix * sums.shape[1])
    return m1 * -sd


def camfull_error(sums: np.ndarray, n_pix: float, flags: np.ndarray) -> np.ndarray:
    """(K, N, 23) CAMFULL sums + (N, 6) link flags -> (K,) CameraPredictor._error (:933-970).

    flags bit 0: link has a target in that frame; bit 1: more than 5 % of its mask carries depth (:955)."""
    s = sums.astype(np.float64)
    K, N = s.shape[:2]
    flags = np.asarray(flags)
    with np.errstate(all='ignore'):
        # all six links at once; the additions below then run term by term in the reference's order
        L = s[..., S_LINK0:S_LINK0 + 18].reshape(K, N, 6, 3)
        has = ((flags & 1) != 0)[None]
        cnt = L[..., 1]
        mism = np.where(has, L[..., 0] / n_pix, 0.0)
        depth = np.where(has & ((flags & 2) != 0)[None] & (cnt > 0), (L[..., 2] * 2.0 ** -32) / np.where(cnt > 0, cnt, 1.0), 0.0)
        err = np.zeros((K, N))
        for l in range(6):
            err += mism[..., l]
            err += depth[..., l]
        n = s[..., S_CNT]
        m = (s[..., S_AA] * 2.0 ** -32) / n              # mean of sqrt|T - D| over its non-zero entries
        var = np.maximum((s[..., S_S1] * 2.0 ** -32) / n - m * m, 0.0)
        err += m * -np.sqrt(var)
        tot = np.zeros(K)
        for i in range(N):                                  # tot_err += err ** 2, frame after frame (:968)
            tot = tot + err[:, i] ** 2
        return tot


def link_planes_of(masked_targets: dict, target_masks: dict, link_names: List[str], shape) -> np.ndarray:
    """{link: mask * depth}, {link: mask} -> (6, H, W) uint64 planes: bit 40 = mask, bits 0..38 = Q32 depth."""
    planes = np.zeros((6,) + tuple(shape), np.uint64)
    for l, link in enumerate(link_names[:6]):
        if link in masked_targets:
            planes[l] = pack_target(masked_targets[link], np.asarray(target_masks[link]).astype(bool).astype(np.uint64))
    return planes


# ------------------------------------------------------------------------------ stage tables
def modelless_stages() -> list:
    """The active list of ModellessCameraPredictor._setStages (:70-111).  `p_fix` is rebound to the yaw sweep
    before `combo` is built (:96-98), so the combo's second stage sweeps yaw, not pitch."""
    coarse = []
    for x in np.logspace(1, .05, 5) / 30:
        coarse += [('tensorsweep', 20, x, _XYZ), ('tensorsweep', 20, x / 2, _RPY)]
    zp_sweep = ('zp_sweep', 20, 0.1)
    ya_fix = ('smartsweep', 20, .03, [False, False, False, False, False, True])
    xyya_narrow = ('smartsweep', 20, .15, [True, True, False, False, False, True])
    fine_descent = ('descent', 50, 0.5, .001, [True] * 6, [0.01] * 6)
    quick_descent = ('descent', 15, 0.5, .001, [True] * 6, [0] * 6)
    combo = [zp_sweep, ya_fix, xyya_narrow] * 2
    return [*coarse, ('tensorsweep', 20, .2, _XYZ), ('tensorsweep', 20, .1, _RPY), fine_descent, *combo,
            quick_descent, quick_descent]


def segmented_stages() -> list:
    """The active list of CameraPredictor._setStages (:614-653)."""
    coarse = []
    for x in np.linspace(.25, .025, 10):
        coarse += [('smartsweep', 6, x, _XYZ), ('smartsweep', 6, x / 2, _RPY)]
    zp_sweep = ('zp_sweep', 20, 0.1)
    p_fix = ('smartsweep', 20, .03, [False, False, False, False, True, False])
    xyya_narrow = ('smartsweep', 5, .025, [True, True, False, False, False, True])
    fine_descent = ('descent', 50, 0.5, .001, [True] * 6, [0.01] * 6)
    quick_descent = ('descent', 15, 0.5, .001, [True] * 6, [0] * 6)
    combo = [zp_sweep, p_fix, xyya_narrow] * 2
    return [*coarse, ('tensorsweep', 20, .2, _XYZ), ('tensorsweep', 20, .1, _RPY), fine_descent, *combo, quick_descent]


# ------------------------------------------------------------------------------ shared machine
class _CameraStageMachine:
    """Stage loop common to both predictors (the reference repeats it in each class)."""

    zp_div_from_stage = True      # CameraPredictor's zp_sweep reads a stale `div` instead (see run_stages)

    def __init__(self, base_pose, ds_factor, preview, save_to, min_angle_inc, history_length, base_intrinsics, device):
        self.base_pose = np.array(base_pose, dtype=float)
        self.ds_factor, self.preview = ds_factor, preview
        self.min_ang_inc = np.asarray(min_angle_inc, dtype=float)
        self.history_length = history_length
        self.u_reader = URDFReader()
        self.renderer = Renderer('seg', None, base_intrinsics, intrinsic_ds_factor=ds_factor, device=device)
        self.engine = self.renderer.engine
        self.classes = ["BG"]
        self.classes.extend(self.u_reader.mesh_names[:6])
        self.link_names = self.classes[1:]
        self.renderer.setMaxParts(None)
        self._P = self.renderer.intrinsics.gl_projection(ZNEAR, ZFAR)
        self.evaluations = 0          # (pose, frame) renders, for throughput accounting
        self.stages = None
        if preview:                   # headless, as Predictor's: frames go to .viz.frame and, with save_to, an uncompressed AVI
            from . import viz
            self.viz = getattr(viz, self._viz_class)(save_to)

    _viz_class = 'ModellessProjectionViz'

    def _preview_targets(self, og_image, target_depth, links_image=None):
        """What the reference loads into its window before the stages start (:140-142, :678-691): the first frame."""
        if self.preview:
            self.viz.loadTargetColor(np.asarray(og_image).astype(np.uint8))
            self.viz.loadTargetDepth(target_depth)
            if links_image is not None:
                self.viz.loadSegmentedLinks(links_image)

    def _preview_poses(self, poses: np.ndarray):
        """preview_if_applicable (:160-167): one window frame per trial pose, showing the FIRST frame's render under it.  The
        trial poses of a step are scored as one batch here; their preview frames follow in the same order."""
        if not self.preview:
            return
        for pose in np.atleast_2d(poses):
            self.renderer.setCameraPose(pose)
            self.renderer.setJointAngles(self.robot_poses[0])
            color, depth = self.renderer.render()
            self.viz.loadRenderedColor(color)
            self.viz.loadRenderedDepth(depth)
            self.viz.show()

    # -- evaluation -------------------------------------------------------------------------------
    def _views(self, poses: np.ndarray) -> np.ndarray:
        return np.stack([self._P @ view_matrix(p) for p in np.atleast_2d(poses)])

    def _sums(self, poses: np.ndarray, loss: int) -> np.ndarray:
        """(K, 6) camera poses -> (K, N, 23) integer sums, in as few engine calls as the 65 535-row limit allows."""
        PV = self._views(poses)
        step = max(1, 65535 // self.number_of_poses)
        out = [self.engine.eval_views(PV[k:k + step], 6, loss) for k in range(0, len(PV), step)]
        self.evaluations += len(PV) * self.number_of_poses
        self._preview_poses(poses)
        return out[0] if len(out) == 1 else np.concatenate(out)

    def _errors(self, poses: np.ndarray) -> np.ndarray:          # `_error` of one pose's renders, per pose
        raise NotImplementedError

    def _sweep_errors(self, poses: np.ndarray) -> np.ndarray:    # the reduction tensorsweep / zp_sweep use
        raise NotImplementedError

    def error_at(self, pose) -> float:
        return float(self._errors(np.asarray(pose, float)[None])[0])

    def do_renders_at_pose(self, pose):
        """Colour and depth of every frame under one camera pose (:116-124); for callers that want images."""
        self.renderer.setCameraPose(pose)
        color_out = np.zeros((self.number_of_poses, *self.renderer.resolution, 3))
        depth_out = np.zeros((self.number_of_poses, *self.renderer.resolution))
        for idx in range(self.number_of_poses):
            self.renderer.setJointAngles(self.robot_poses[idx])
            color_out[idx], depth_out[idx] = self.renderer.render()
        return color_out, depth_out

    def _batch_downsample(self, base: np.ndarray, factor: int) -> np.ndarray:
        """cv2.resize of every frame to (W//f, H//f), results held as float64 (:376-386)."""
        dims = [x // factor for x in base.shape[1:3]]
        out = np.zeros((base.shape[0], *dims, 3)) if base.ndim == 4 else np.zeros((base.shape[0], *dims))
        for idx in range(base.shape[0]):
            out[idx] = resize_linear(base[idx], dims[1], dims[0])
        return out

    # -- stage loop ---------------------------------------------------------------------------------
    def run_stages(self, pose: np.ndarray) -> np.ndarray:
        learning_rates = np.zeros(6)
        history = np.zeros((self.history_length, 6))
        err_history = np.zeros(self.history_length)
        div = None
        self.trace = []

        def sweep_along(low, high, n, extra=None):
            space = np.linspace(low, high, n)
            if extra is not None:
                extra(space)
            return space

        for stage in self.stages:
            kind = stage[0]
            if kind == 'spiral' and self.zp_div_from_stage:
                # global search over shells of spiralling view points (:174-181); only the model-less predictor has it
                pose = SpiralRenderer(self.renderer, self._spiral_errors, None, *stage[1:]).run()

            elif kind == 'descent':
                for i in range(6):
                    if stage[5][i] is not None:
                        learning_rates[i] = stage[5][i]
                do_param = np.array(stage[4])
                over_err = under_err = None
                for _ in range(stage[1]):
                    for idx in np.where(do_param)[0]:
                        if abs(np.mean(history, 0)[idx] - pose[idx]) <= learning_rates[idx]:
                            learning_rates[idx] *= stage[2]
                        learning_rates = np.max((learning_rates, self.min_ang_inc), 0)
                        # under and over in one batch of two views
                        pair = np.stack([pose.copy(), pose.copy()])
                        pair[0, idx] -= learning_rates[idx]
                        pair[1, idx] = pair[0, idx] + 2 * learning_rates[idx]      # temp[idx] += 2*lr on the "under" pose (:213)
                        under_err, over_err = self._errors(pair)
                        if over_err < under_err:
                            pose[idx] += learning_rates[idx]
                        elif over_err > under_err:
                            pose[idx] -= learning_rates[idx]
                    history[1:] = history[:-1]
                    history[0] = pose
                    err_history[1:] = err_history[:-1]
                    err_history[0] = min(over_err, under_err)
                    with np.errstate(all='ignore'):
                        if abs(np.mean(err_history) - err_history[0]) / err_history[0] < stage[3]:
                            break
                    span = history.max(0) - history.min(0)
                    if ((span <= self.min_ang_inc) + np.isclose(span, self.min_ang_inc)).all():
                        break
                    if (history[:3] == history[0]).all():
                        break

            elif kind == 'smartsweep':
                do_param = np.array(stage[3])
                div = stage[1]
                base_err = self._errors(pose[None])[0]
                for idx in np.where(do_param)[0]:
                    temp_low, temp_high = pose.copy(), pose.copy()
                    temp_low[idx] = temp_low[idx] - stage[2]
                    temp_high[idx] = temp_low[idx] + stage[2]      # low + range, i.e. back at the pose: the sweep is one-sided (:250-251)
                    space = np.linspace(temp_low, temp_high, div)
                    space_err = list(self._errors(space))
                    x = np.linspace(temp_low[idx], temp_high[idx], div * 5)
                    predicted_errors = cubic_interp(space[:, idx], np.array(space_err), x)
                    temp_pose = pose.copy()
                    temp_pose[idx] = x[predicted_errors.argmin()]
                    pred_min_err = self._errors(temp_pose[None])[0]
                    errs = [base_err, min(space_err), pred_min_err]
                    min_type = errs.index(min(errs))
                    if min_type == 1:
                        pose = space[space_err.index(min(space_err))].copy()
                        err_history[1:] = err_history[:-1]
                        err_history[0] = min(space_err)
                    elif min_type == 2:
                        pose = temp_pose
                        err_history[1:] = err_history[:-1]
                        err_history[0] = pred_min_err
                    history[1:] = history[:-1]
                    history[0] = pose

            elif kind == 'tensorsweep':
                do_param = np.array(stage[3])
                div = stage[1]
                for idx in np.where(do_param)[0]:
                    temp_low, temp_high = pose.copy(), pose.copy()
                    temp_low[idx] -= stage[2]
                    temp_high[idx] += stage[2]
                    space = np.linspace(temp_low, temp_high, div)
                    pose = space[int(np.argmin(self._sweep_errors(space)))].copy()

            elif kind == 'zp_sweep':
                # z sweep with the pitch that keeps the same focus point (:318-345,858-886).  The segmented predictor
                # never assigns `div` here and sweeps with whatever the previous sweep stage left (:870).
                if self.zp_div_from_stage:
                    div = stage[1]
                temp_low, temp_high, temp_pose = pose.copy(), pose.copy(), pose.copy()
                temp_low[2] = temp_pose[2] - stage[2]
                temp_high[2] = temp_pose[2] + stage[2]
                space = np.linspace(temp_low, temp_high, div)
                space[:, 4] = np.arctan(np.tan(temp_pose[4]) - ((space[:, 2] - temp_pose[2]) / np.sqrt(temp_pose[0] ** 2 + temp_pose[1] ** 2)))
                pose = space[int(np.argmin(self._sweep_errors(space)))].copy()

            elif kind == 'xya_sweep' and self.zp_div_from_stage:
                # x sweep with the yaw that keeps the same focus point (:347-371; the segmented predictor has no such branch)
                div = stage[1]
                temp_low, temp_high = pose.copy(), pose.copy()
                temp_low[0] = pose[0] - stage[2]
                temp_high[0] = pose[0] + stage[2]
                space = np.linspace(temp_low, temp_high, div)
                space[:, 5] = -np.arctan(((space[:, 0] - pose[0]) / pose[0]) * np.tan(pose[5]))
                pose = space[int(np.argmin(self._sweep_errors(space)))].copy()

            self.trace.append((kind, pose.copy()))
        return pose

    def _start(self, og_images, target_depths, robot_poses, starting_camera_pose):
        og_images, target_depths = np.asarray(og_images), np.asarray(target_depths)
        if og_images.ndim == 3:
            og_images, target_depths, robot_poses = np.array([og_images]), np.array([target_depths]), np.array([robot_poses])
        self.robot_poses = np.array(robot_poses, dtype=float)
        assert og_images.shape[0] == target_depths.shape[0] == self.robot_poses.shape[0]
        self.number_of_poses = og_images.shape[0]
        pose = np.copy(self.base_pose) if starting_camera_pose is None else np.array(starting_camera_pose, dtype=float)
        return og_images, target_depths, pose


class SpiralRenderer:
    """Exhaustive search over view points on `shells` cylinders of radius r_limits, each a spiral of `turns` x
    `per_round` poses climbing through z_limits and looking at the axis (:434-497).  The reference renders and scores
    the poses one by one and plots the errors; here `error_func` takes the whole pose array (chunks of `batch`)."""

    def __init__(self, renderer, error_func, render_func=None, batch=10000, r_limits=[1, 3], shells=25, per_round=75,
                 z_limits=[0, 1], turns=10) -> None:
        self.renderer, self.error, self.render_func = renderer, error_func, render_func
        self.batch = batch
        self.r_min, self.r_max = min(r_limits), max(r_limits)
        self.shells, self.per_round, self.turns = shells, per_round, turns
        self.z_min, self.z_max = min(z_limits), max(z_limits)

    def poses(self) -> np.ndarray:
        num_per_spiral = self.turns * self.per_round
        base_spiral = np.zeros((num_per_spiral, 6))
        angles_full = np.tile(np.linspace(0, 2 * np.pi, self.per_round), self.turns)
        base_spiral[:, 5] = 2 * np.pi - angles_full
        base_spiral[:, 0] = -np.sin(angles_full)
        base_spiral[:, 1] = -np.cos(angles_full)
        base_spiral[:, 2] = np.linspace(self.z_min, self.z_max, num_per_spiral)
        full_space = np.tile(base_spiral, (self.shells, 1))
        r_full = np.repeat(np.linspace(self.r_min, self.r_max, self.shells), num_per_spiral)
        full_space[:, 0] *= r_full
        full_space[:, 1] *= r_full
        return full_space

    def run(self) -> np.ndarray:
        full_space = self.poses()
        errors = np.concatenate([self.error(full_space[i:i + self.batch]) for i in range(0, len(full_space), self.batch)])
        self.errors = errors
        return full_space[errors.argmin()].copy()


class ModellessCameraPredictor(_CameraStageMachine):
    """Camera pose from depth alone: every frame's render against its depth map, no segmentation (:28-431)."""

    def __init__(self, base_pose=DEFAULT_CAMERA_POSE, ds_factor: int = 8, preview: bool = False, save_to: str = None,
                 min_angle_inc=np.array([0.001, 0.001, 0.001, 0.002, 0.002, 0.002]), history_length=5,
                 base_intrinsics='1280_720_color', *, device: int = 0):
        super().__init__(base_pose, ds_factor, preview, save_to, min_angle_inc, history_length, base_intrinsics, device)

    def _setStages(self):
        self.stages = modelless_stages()

    def run(self, og_images, target_depths, robot_poses, starting_camera_pose=None) -> np.ndarray:
        og_images, target_depths, pose = self._start(og_images, target_depths, robot_poses, starting_camera_pose)
        self._preview_targets(og_images[0], target_depths[0])
        self._tgt_depths = self._batch_downsample(target_depths, self.ds_factor)
        self._n_pix = float(self._tgt_depths.shape[1] * self._tgt_depths.shape[2])
        self._frame_planes = (np.stack([pack_target(d) for d in self._tgt_depths]), self._tgt_depths.astype(np.float32))
        self.engine.set_frames(self.robot_poses, *self._frame_planes)
        if self.stages is None:
            self._setStages()
        return self.run_stages(pose)

    def _errors(self, poses):
        return modelless_error(self._sums(poses, LOSS_TSWEEP), self._n_pix)

    _sweep_errors = _errors       # _error treats a (div, poses, H, W) stack the same way (:393-408)

    def _spiral_errors(self, poses):
        """What SpiralRenderer.run feeds `_error`: ONE render per pose — `renderer.render()` after do_renders_at_pose
        leaves the robot at the last frame's joint vector (:485-488) — broadcast against every frame's target (:410-424)."""
        tq, t32 = self._frame_planes
        self.engine.set_frames(np.tile(self.robot_poses[-1], (self.number_of_poses, 1)), tq, t32)
        try:
            return self._errors(poses)
        finally:
            self.engine.set_frames(self.robot_poses, tq, t32)


class CameraPredictor(_CameraStageMachine):
    """Camera pose from segmented frames: per-link masks and depths plus the whole depth map (:576-975)."""

    zp_div_from_stage = False
    _viz_class = 'ProjectionViz'

    def __init__(self, base_pose=DEFAULT_CAMERA_POSE, ds_factor: int = 8, preview: bool = False, save_to: str = None,
                 min_angle_inc=np.array([0.001, 0.001, 0.001, 0.002, 0.002, 0.002]), history_length=5,
                 base_intrinsics='1280_720_color', *, device: int = 0, segmenter: Optional[Callable] = None):
        """`segmenter`: callable(colour uint8 HxWx3) -> {'class_ids', 'rois', 'scores', 'masks' (H,W,K) bool}; stands in
        for pixellib's segmentImage on "models/segmentation/multi/B.h5" (:603-606,687)."""
        super().__init__(base_pose, ds_factor, preview, save_to, min_angle_inc, history_length, base_intrinsics, device)
        if segmenter is None:
            raise ValueError("CameraPredictor needs a segmenter (e.g. rope_s3d_amd.maskrcnn.MaskRCNNSegmenter)")
        self.seg = segmenter

    def _setStages(self):
        self.stages = segmented_stages()

    def _reorganize_by_link(self, data: dict) -> dict:
        """Instances of one class merged: OR of masks, max of scores (:903-917)."""
        out = {}
        ids = list(data['class_ids'])
        for idx, cid in enumerate(ids):
            name = self.classes[cid]
            if cid not in ids[:idx]:
                out[name] = {'roi': data['rois'][idx] if 'rois' in data else None, 'confidence': data['scores'][idx],
                             'mask': np.array(data['masks'][..., idx], dtype=bool)}
            else:
                out[name]['mask'] |= np.asarray(data['masks'][..., idx], dtype=bool)
                out[name]['confidence'] = max(out[name]['confidence'], data['scores'][idx])
        return out

    def _load_targets(self, seg_data: List[dict], tgt_depths: np.ndarray) -> None:
        """(:919-931)  `[{}] * n` makes every frame share ONE dictionary, so each link ends up with the mask and masked
        depth of the last frame it was found in, for all frames.  Kept: the same planes and flags go to every frame."""
        masked, masks = {}, {}
        self._tgt_depths = tgt_depths
        for idx in range(len(seg_data)):
            for link in self.link_names:
                if link in seg_data[idx]:
                    m = seg_data[idx][link]['mask']
                    masked[link] = m * tgt_depths[idx]
                    masks[link] = m
        self._masked_targets = [masked] * len(seg_data)
        self._target_masks = [masks] * len(seg_data)
        planes = link_planes_of(masked, masks, self.link_names, tgt_depths.shape[1:])
        flags = np.zeros(6, np.uint8)
        for l, link in enumerate(self.link_names):
            if link in masked:
                flags[l] = 1 | (2 if np.sum(masked[link] != 0) > (.05 * np.sum(masks[link])) else 0)
        self._flags = np.tile(flags, (len(seg_data), 1))
        self._n_pix = float(tgt_depths.shape[1] * tgt_depths.shape[2])
        self.engine.set_frames(self.robot_poses, np.stack([pack_target(d) for d in tgt_depths]),
                               tgt_depths.astype(np.float32), np.tile(planes[None], (len(seg_data), 1, 1, 1)))

    def run(self, og_images, target_depths, robot_poses, starting_camera_pose=None) -> np.ndarray:
        og_images, target_depths, pose = self._start(og_images, target_depths, robot_poses, starting_camera_pose)
        full_color, full_depth = og_images[0], target_depths[0]
        target_depths = self._batch_downsample(target_depths, self.ds_factor)
        og_images = self._batch_downsample(og_images, self.ds_factor)
        segmentation_data = [self._reorganize_by_link(self.seg(og_images[idx].astype(np.uint8))) for idx in range(len(og_images))]
        if self.preview:              # the window's "detected links": frame 0 with its links tinted (pixellib's annotated frame, :687-691)
            tinted = og_images[0].astype(np.float64)
            for name, d in segmentation_data[0].items():
                if name in self.link_names:
                    tinted[d['mask']] = tinted[d['mask']] * .5 + np.array(DEFAULT_RENDER_COLORS[self.link_names.index(name)], np.float64) * .5
            self._preview_targets(full_color, full_depth, np.rint(tinted).astype(np.uint8))
        if self.stages is None:
            self._setStages()
        self._load_targets(segmentation_data, target_depths)
        return self.run_stages(pose)

    def _errors(self, poses):
        return camfull_error(self._sums(poses, LOSS_CAMFULL), self._n_pix, self._flags)

    def _sweep_errors(self, poses):
        return pooled_sweep_error(self._sums(poses, LOSS_TSWEEP), self._n_pix)
