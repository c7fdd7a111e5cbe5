"""Sequential CPU restatement of the reference's camera-pose predictors
(robotpose/prediction/camera_pose_prediction.py:28-431 ModellessCameraPredictor, :576-975 CameraPredictor).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  PARITY UNPINNED: the reference holds no fixtures or decision
traces for this path and pyrender / tensorflow / cv2 are absent here.

One trial camera pose at a time, one frame at a time, as the reference executes it: set the camera, render the
frame's joint vector with the C oracle rasteriser, reduce.  Two reductions are given for every error:
  * `*_sums`     the exact integer sums of the arithmetic contract — what the GPU must reproduce bit
                 for bit — and the float epilogue over them;
  * `*_literal`  the reference's numpy / TF expression term by term in floating point, to bound the distance
                 between the contract and the reference's own arithmetic.
"""
import numpy as np
from scipy.interpolate import interp1d

from . import oracle as orc

Q32 = 4294967296.0
MASK39 = np.uint64((1 << 39) - 1)
XYZ, RPY = [True, True, True, False, False, False], [False, False, False, True, True, True]


# ------------------------------------------------------------------ camera (render.py:107-111, render_utils.py:56-108)
def pose_matrix(pose6):
    """Renderer.setCameraPose adds pi/2 to element 4, then makePose(x,y,z,pitch=a3,roll=a4,yaw=a5) builds
    Rz(yaw)·Ry(pitch)·Rx(roll) from products of cosines and sines (angToPoseArr)."""
    x, y, z, a3, a4, a5 = [float(v) for v in pose6]
    ang = np.array([a5, a3, a4 + np.pi / 2])            # yaw, pitch, roll
    c, s = np.cos(ang), np.sin(ang)
    M = np.zeros((4, 4))
    M[0, 0] = c[0] * c[1]
    M[1, 0] = c[1] * s[0]
    M[2, 0] = -1 * s[1]
    M[0, 1] = c[0] * s[1] * s[2] - c[2] * s[0]
    M[1, 1] = c[0] * c[2] + np.prod(s)
    M[2, 1] = c[1] * s[2]
    M[0, 2] = s[0] * s[2] + c[0] * c[2] * s[1]
    M[1, 2] = c[2] * s[0] * s[1] - c[0] * s[2]
    M[2, 2] = c[1] * c[2]
    M[:3, 3] = x, y, z
    M[3, 3] = 1.0
    return M


def view_of_pose(pose6):
    M = pose_matrix(pose6)
    V = np.eye(4)
    V[:3, :3] = M[:3, :3].T
    V[:3, 3] = -M[:3, :3].T @ M[:3, 3]
    return V


def gl_projection(fx, fy, cx, cy, W, H, znear=0.05, zfar=100.0):
    """pyrender 0.1.45 IntrinsicsCamera.get_projection_matrix."""
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * fx / W
    P[1, 1] = 2.0 * fy / H
    P[0, 2] = 1.0 - 2.0 * cx / W
    P[1, 2] = 2.0 * cy / H - 1.0
    P[3, 2] = -1.0
    P[2, 2] = (zfar + znear) / (znear - zfar)
    P[2, 3] = (2.0 * zfar * znear) / (znear - zfar)
    return P


# ------------------------------------------------------------------ reductions
def q32_of_f32(z):
    return (np.asarray(z, np.float32).astype(np.float64) * Q32).astype(np.uint64)


def sqrt_q32(dq):
    return (np.sqrt(dq.astype(np.float64)) * 65536.0).astype(np.uint64)


def camfull_sums(depth_f32, ids, tq, link_planes):
    """Integer sums of CameraPredictor._error for one frame (word layout of include/rope_s3d.h)."""
    s = np.zeros(23, np.uint64)
    zq = q32_of_f32(depth_f32)
    T = tq & MASK39
    dq = np.where(T > zq, T - zq, zq - T)
    nz = dq != 0
    s[0] = nz.sum()
    s[1] = dq.sum(dtype=np.uint64)
    s[2] = sqrt_q32(dq[nz]).sum(dtype=np.uint64)
    for l in range(6):
        M = ((link_planes[l] >> np.uint64(40)) & np.uint64(1)).astype(bool)
        # render_color[..., 0] == color_dict[link][0] (:943-946): base_link's blue value 0 is the background's as well
        # (constants.py:82-89), so its render mask holds every empty pixel too; depth is 0 there, the depth term is unchanged
        R = (ids == l) | (ids == 255) if l == 0 else ids == l
        a = link_planes[l] & MASK39
        b = np.where(R, zq, np.uint64(0))
        dl = np.where(a > b, a - b, b - a)
        s[5 + 3 * l] = (M != R).sum()
        s[5 + 3 * l + 1] = (dl != 0).sum()
        s[5 + 3 * l + 2] = sqrt_q32(dl[dl != 0]).sum(dtype=np.uint64)
    return s


def sweep_mean_std(s, n_pix):
    s = np.asarray(s).astype(np.float64)
    m1 = (s[..., 1] * 2.0 ** -32) / n_pix
    s2 = (s[..., 2] * 2.0 ** 40 + s[..., 3] * 2.0 ** 21) + s[..., 4]
    var = np.maximum((s2 * 2.0 ** -64) / n_pix - m1 * m1, 0.0)
    return m1, np.sqrt(var)


def camfull_frame_error(s, n_pix, flags):
    s = s.astype(np.float64)
    err = 0.0
    with np.errstate(all='ignore'):
        for l in range(6):
            if not flags[l] & 1:
                continue
            err += s[5 + 3 * l] / n_pix
            if (flags[l] & 2) and s[5 + 3 * l + 1] > 0:
                err += (s[5 + 3 * l + 2] * 2.0 ** -32) / s[5 + 3 * l + 1]
        m = (s[2] * 2.0 ** -32) / s[0]
        var = np.maximum((s[1] * 2.0 ** -32) / s[0] - m * m, 0.0)
        return err + m * -np.sqrt(var)


def modelless_error_literal(render_depth_frames, tgt_depths):
    """ModellessCameraPredictor._error, 3-D input, float32 as the TF ops run (:410-424)."""
    rendered = np.sqrt(np.asarray(render_depth_frames).astype(np.float32))
    actual = np.sqrt(np.asarray(tgt_depths).astype(np.float32))
    diff = np.abs(actual - rendered)
    err = diff.mean((1, 2), dtype=np.float64) * -diff.std((1, 2), dtype=np.float64)
    return float(np.mean(np.power(1.1, err)))


def pooled_sweep_literal(depths, tgt_depths):
    """CameraPredictor tensorsweep / zp_sweep score of one pose: all frames pooled (:841-846)."""
    diff = np.abs(np.sqrt(np.asarray(tgt_depths).astype(np.float32)) - np.sqrt(np.asarray(depths).astype(np.float32)))
    return float(diff.mean(dtype=np.float64) * -diff.std(dtype=np.float64))


def camfull_error_literal(render_blue_frames, render_depth_frames, tgt_depths, masked_targets, target_masks, link_names, link_blue):
    """CameraPredictor._error term by term (:933-970); masked_targets / target_masks are per-frame lists of dicts."""
    tot_err = 0
    for idx in range(len(render_depth_frames)):
        err = 0
        for link in link_names:
            if link in masked_targets[idx]:
                target_masked, joint_mask = masked_targets[idx][link], target_masks[idx][link]
                render_mask = render_blue_frames[idx] == link_blue[link]
                render_masked = render_depth_frames[idx] * render_mask
                err += np.mean(joint_mask != render_mask)
                if np.sum(target_masked != 0) > (.05 * np.sum(joint_mask)):
                    diff = np.abs(target_masked - render_masked) ** .5
                    if diff[diff != 0].size > 0:
                        err += np.mean(diff[diff != 0])
        diff = np.abs(tgt_depths[idx] - render_depth_frames[idx]) ** 0.5
        with np.errstate(all='ignore'):
            err += np.mean(diff[diff != 0]) * -np.std(diff[diff != 0])
        tot_err += err ** 2
    return tot_err


# ------------------------------------------------------------------ stage tables (:70-111, :614-653)
def modelless_stages():
    out = []
    for x in np.logspace(1, .05, 5) / 30:
        out += [['tensorsweep', 20, x, XYZ], ['tensorsweep', 20, x / 2, RPY]]
    zp = ['zp_sweep', 20, 0.1]
    ya_fix = ['smartsweep', 20, .03, [False, False, False, False, False, True]]     # `ya_fix = p_fix = ...` rebinds p_fix (:96)
    xyya = ['smartsweep', 20, .15, [True, True, False, False, False, True]]
    out += [['tensorsweep', 20, .2, XYZ], ['tensorsweep', 20, .1, RPY],
            ['descent', 50, 0.5, .001, [True] * 6, [0.01] * 6]]
    out += [zp, ya_fix, xyya] * 2
    quick = ['descent', 15, 0.5, .001, [True] * 6, [0] * 6]
    return out + [quick, quick]


def segmented_stages():
    out = []
    for x in np.linspace(.25, .025, 10):
        out += [['smartsweep', 6, x, XYZ], ['smartsweep', 6, x / 2, RPY]]
    zp = ['zp_sweep', 20, 0.1]
    p_fix = ['smartsweep', 20, .03, [False, False, False, False, True, False]]
    xyya = ['smartsweep', 5, .025, [True, True, False, False, False, True]]
    out += [['tensorsweep', 20, .2, XYZ], ['tensorsweep', 20, .1, RPY],
            ['descent', 50, 0.5, .001, [True] * 6, [0.01] * 6]]
    out += [zp, p_fix, xyya] * 2
    return out + [['descent', 15, 0.5, .001, [True] * 6, [0] * 6]]


# ------------------------------------------------------------------ the predictors
class CameraReference:
    """mode 'modelless' or 'segmented'.  Targets are taken at render resolution (the caller does the cv2.resize
    step and, for 'segmented', the segmentation + _reorganize_by_link)."""

    def __init__(self, o: orc.Oracle, P, mode, robot_poses, tgt_depths, seg_data=None, link_names=None,
                 min_angle_inc=None, history_length=5, stages=None):
        self.o, self.P, self.mode = o, np.asarray(P, float), mode
        self.robot_poses = np.asarray(robot_poses, float).reshape(-1, 6)
        self.tgt = np.asarray(tgt_depths, np.float64)
        self.N = len(self.robot_poses)
        self.n_pix = float(self.tgt.shape[1] * self.tgt.shape[2])
        self.min_ang_inc = np.array([0.001, 0.001, 0.001, 0.002, 0.002, 0.002]) if min_angle_inc is None else np.asarray(min_angle_inc, float)
        self.history_length = history_length
        self.stages = stages if stages is not None else (modelless_stages() if mode == 'modelless' else segmented_stages())
        self.evaluations = 0
        self.tq = [orc.pack_target(d) for d in self.tgt]
        self.t32 = [np.ascontiguousarray(d, np.float32) for d in self.tgt]
        if mode == 'segmented':
            # _load_targets (:919-931): `[{}] * n` -> ONE dict shared by all frames; later frames overwrite earlier ones
            masked, masks = {}, {}
            for idx in range(self.N):
                for link in link_names:
                    if link in seg_data[idx]:
                        m = seg_data[idx][link]['mask']
                        masked[link] = m * self.tgt[idx]
                        masks[link] = m
            self.masked_targets, self.target_masks = [masked] * self.N, [masks] * self.N
            self.planes = np.zeros((6,) + self.tgt.shape[1:], np.uint64)
            self.flags = np.zeros(6, np.uint8)
            for l, link in enumerate(link_names[:6]):
                if link in masked:
                    self.planes[l] = orc.pack_target(masked[link], np.asarray(masks[link]).astype(bool).astype(np.uint64))
                    self.flags[l] = 1 | (2 if np.sum(masked[link] != 0) > (.05 * np.sum(masks[link])) else 0)

    # one (pose, frame) render
    def _render_key(self, pose, idx):
        self.o.PV = np.ascontiguousarray(self.P @ view_of_pose(pose))
        self.evaluations += 1
        return self.o.raster_key(self.robot_poses[idx], 6)

    def frame_sums(self, pose, loss):
        """(N, 23) integer sums of every frame under `pose`."""
        out = np.zeros((self.N, 23), np.uint64)
        for idx in range(self.N):
            key = self._render_key(pose, idx)
            if loss == 'sweep':
                out[idx] = self.o.sums(key, orc.LOSS_TSWEEP, 6, self.tq[idx], self.t32[idx])
            else:
                depth, ids = self.o.resolve(key)
                out[idx] = camfull_sums(depth, ids, self.tq[idx], self.planes)
        return out

    def error(self, pose):
        if self.mode == 'modelless':
            m1, sd = sweep_mean_std(self.frame_sums(pose, 'sweep'), self.n_pix)
            return float(np.mean(np.power(1.1, m1 * -sd)))
        s = self.frame_sums(pose, 'full')
        return float(sum(camfull_frame_error(s[i], self.n_pix, self.flags) ** 2 for i in range(self.N)))

    def sweep_error(self, pose):
        if self.mode == 'modelless':
            return self.error(pose)
        pooled = self.frame_sums(pose, 'sweep').sum(axis=0, dtype=np.uint64)
        m1, sd = sweep_mean_std(pooled, self.n_pix * self.N)
        return float(m1 * -sd)

    def spiral(self, batch=10000, r_limits=(1, 3), shells=25, per_round=75, z_limits=(0, 1), turns=10):
        """SpiralRenderer.run (:453-497) with ModellessCameraPredictor._error on a 2-D depth: one render per pose at
        the LAST frame's joint vector, broadcast against all frame targets (:410-424)."""
        r_min, r_max, z_min, z_max = min(r_limits), max(r_limits), min(z_limits), max(z_limits)
        num_per_spiral = turns * per_round
        base_spiral = np.zeros((num_per_spiral, 6))
        angles_partial = np.linspace(0, 2 * np.pi, per_round)
        angles_full = np.tile(angles_partial, turns)
        base_spiral[:, 5] = 2 * np.pi - angles_full
        base_spiral[:, 0] = -np.sin(angles_full)
        base_spiral[:, 1] = -np.cos(angles_full)
        base_spiral[:, 2] = np.linspace(z_min, z_max, num_per_spiral)
        full_space = np.tile(base_spiral, (shells, 1))
        r_partial = np.linspace(r_min, r_max, shells)
        r_full = np.repeat(r_partial, num_per_spiral)
        full_space[:, 0] *= r_full
        full_space[:, 1] *= r_full
        errors = np.zeros(full_space.shape[0])
        for k, pose in enumerate(full_space):
            key = self._render_key(pose, self.N - 1)
            s = np.stack([self.o.sums(key, orc.LOSS_TSWEEP, 6, self.tq[i], self.t32[i]) for i in range(self.N)])
            m1, sd = sweep_mean_std(s, self.n_pix)
            errors[k] = np.mean(np.power(1.1, m1 * -sd))
        return full_space[errors.argmin()], errors

    def run(self, starting_pose):
        pose = np.array(starting_pose, dtype=float)
        learning_rates = np.zeros(6)
        history = np.zeros((self.history_length, 6))
        err_history = np.zeros(self.history_length)
        div = None
        trace = []
        for stage in self.stages:
            if stage[0] == 'spiral' and self.mode == 'modelless':
                pose, _ = self.spiral(*stage[1:])
            elif stage[0] == 'descent':
                for i in range(6):
                    if stage[5][i] is not None:
                        learning_rates[i] = stage[5][i]
                do_param = np.array(stage[4])
                for _ in range(stage[1]):
                    for idx in np.where(do_param)[0]:
                        if abs(np.mean(history, 0)[idx] - pose[idx]) <= learning_rates[idx]:
                            learning_rates[idx] *= stage[2]
                        learning_rates = np.max((learning_rates, self.min_ang_inc), 0)
                        temp = pose.copy()
                        temp[idx] -= learning_rates[idx]
                        under_err = self.error(temp)
                        temp[idx] += 2 * learning_rates[idx]
                        over_err = self.error(temp)
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
                    if ((history.max(0) - history.min(0) <= self.min_ang_inc)
                            + np.isclose((history.max(0) - history.min(0)), self.min_ang_inc)).all():
                        break
                    if (history[:3] == history[0]).all():
                        break
            elif stage[0] == 'smartsweep':
                do_param = np.array(stage[3])
                div = stage[1]
                base_err = self.error(pose)
                for idx in np.where(do_param)[0]:
                    temp_low = pose.copy()
                    temp_high = pose.copy()
                    temp_low[idx] = temp_low[idx] - stage[2]
                    temp_high[idx] = temp_low[idx] + stage[2]
                    space = np.linspace(temp_low, temp_high, div)
                    space_err = [self.error(pose_val) for pose_val in space]
                    err_pred = interp1d(space[:, idx], np.array(space_err), kind='cubic')
                    x = np.linspace(temp_low[idx], temp_high[idx], div * 5)
                    predicted_errors = err_pred(x)
                    temp_pose = pose.copy()
                    temp_pose[idx] = x[predicted_errors.argmin()]
                    pred_min_err = self.error(temp_pose)
                    errs = [base_err, min(space_err), pred_min_err]
                    min_type = errs.index(min(errs))
                    if min_type == 1:
                        pose = space[space_err.index(min(space_err))]
                        err_history[1:] = err_history[:-1]
                        err_history[0] = min(space_err)
                    elif min_type == 2:
                        pose = temp_pose
                        err_history[1:] = err_history[:-1]
                        err_history[0] = pred_min_err
                    history[1:] = history[:-1]
                    history[0] = pose
            elif stage[0] == 'tensorsweep':
                do_param = np.array(stage[3])
                div = stage[1]
                for idx in np.where(do_param)[0]:
                    temp_low = pose.copy()
                    temp_high = pose.copy()
                    temp_low[idx] -= stage[2]
                    temp_high[idx] += stage[2]
                    space = np.linspace(temp_low, temp_high, div)
                    errs = np.array([self.sweep_error(p) for p in space])
                    pose = space[errs.argmin()]
            elif stage[0] == 'zp_sweep':
                if self.mode == 'modelless':
                    div = stage[1]                      # the segmented predictor leaves `div` as the last sweep set it (:870)
                temp_low = pose.copy()
                temp_high = pose.copy()
                temp_pose = pose.copy()
                temp_low[2] = temp_pose[2] - stage[2]
                temp_high[2] = temp_pose[2] + stage[2]
                space = np.linspace(temp_low, temp_high, div)
                space[:, 4] = np.arctan(np.tan(temp_pose[4]) - ((space[:, 2] - temp_pose[2]) / np.sqrt(temp_pose[0] ** 2 + temp_pose[1] ** 2)))
                errs = np.array([self.sweep_error(p) for p in space])
                pose = space[errs.argmin()]
            elif stage[0] == 'xya_sweep' and self.mode == 'modelless':
                div = stage[1]
                temp_low = pose.copy()
                temp_high = pose.copy()
                temp_low[0] = pose[0] - stage[2]
                temp_high[0] = pose[0] + stage[2]
                space = np.linspace(temp_low, temp_high, div)
                space[:, 5] = -np.arctan(((space[:, 0] - pose[0]) / pose[0]) * np.tan(pose[5]))
                errs = np.array([self.sweep_error(p) for p in space])
                pose = space[errs.argmin()]
            trace.append((stage[0], np.array(pose, dtype=float)))
        return np.array(pose, dtype=float), trace
