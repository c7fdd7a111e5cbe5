"""Sequential CPU restatement of Predictor.run (robotpose/prediction/predict.py:127-375).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  PARITY UNPINNED: the reference has no
decision traces to compare with; this file restates its control flow one evaluation at a
time — E(a) = render at a, then _error — exactly as the reference executes it, so that the
batched GPU Predictor can be checked decision by decision.

Inputs are taken at render resolution (the caller does the cv2.resize step): the target
depth, channel 0 of the target colour image, the link -> channel-0 map, the lookup grid and
the lookup crop.
"""
import numpy as np
from scipy.interpolate import interp1d

from . import oracle as orc

HISTORY_LENGTH = 5                                   # predict.py:30

# stage lists, written out as data (robotpose/prediction/stages.py:138-168)
STAGES = {
    'SL': [('lookup',), ('sflip', 4), ('isweep', 4, 10, 'L', 0.1), ('isweep', 4, 10, 'S', 0.1), ('sflip', 4)],
    'SLU': [('lookup',), ('sflip', 4),
            ('descent', 4, 10, 'SL', [0.05, 0.05, 0.1, 0.5, 0.5, 0.5], 0.5, 0.1),
            ('sflip', 4), ('isweep', 6, 25, 'U', None), ('sflip', 4), ('sflip', 6), ('isweep', 6, 10, 'U', 0.1),
            ('descent', 6, 40, 'SLU', [None] * 6, 0.5, 0.0075)],
}


def _mask(letters):
    return np.array([c in letters for c in 'SLURBT'])


def predict_reference(o: orc.Oracle, tgt_depth, tgt_blue, link_names, link_blue, joint_limits, camera_pose,
                      lookup_angles, lookup_crop, do_angles='SLU', min_ang_inc=None, stages=None, seg_masks=None,
                      lookup_depth=None, lookup_live=None):
    """-> (final angles, trace [(stage kind, angles after it)], number of E(a) evaluations).

    Synthetic mode (default): masks are read from `tgt_blue` as _loadSynthetic does.  Segmentation mode:
    pass `seg_masks` {link name: bool mask} and `lookup_depth` as _segmentLoad/_load_target leave them
    (predict.py:397-442); `tgt_depth` is then the body-masked depth.

    `lookup_live`: the reference's cross-frame state, restated literally.  Its run() does
    `angles = self.lookup_angles[argmin]` — a numpy VIEW of the table row (predict.py:171) — and Descent's
    `angles[idx] += rate` (predict.py:212-215) then edits the table in place until some stage rebinds `angles`.
    Pass the same (n, 6) float64 array for every frame of a sequence and this function does exactly that with it
    (the scores keep coming from `lookup_angles`: the reference's depth table is rendered once from the untouched grid).
    None: every frame starts from a copy of the grid row."""
    min_ang_inc = np.array([.005] * 6) if min_ang_inc is None else np.asarray(min_ang_inc, float)
    tgt_depth = np.asarray(tgt_depth, np.float64)
    H, W = tgt_depth.shape
    n_pix = float(H * W)

    if seg_masks is None:
        # --- _loadSynthetic (predict.py:445-469)
        new = np.zeros(tgt_depth.shape)
        for k in link_blue:
            if k in link_names[:6]:
                new += tgt_blue == link_blue[k]
        lookup_depth = tgt_depth * new.astype(bool).astype(float)
        seg_masks = {}
        for link in link_names:
            m = tgt_blue == link_blue[link]
            if np.sum(m.astype(float)) > 0:
                seg_masks[link] = m
    bits = np.zeros(tgt_depth.shape, np.uint64)
    flags = np.zeros(8, np.uint8)
    for l, link in enumerate(link_names):
        if link in seg_masks:                              # _load_target (predict.py:408-413)
            m = np.asarray(seg_masks[link], bool)
            bits |= m.astype(np.uint64) << np.uint64(l)
            flags[l] |= 1
            if np.sum((m * tgt_depth) != 0) > (.05 * np.sum(m)):
                flags[l] |= 2
    tq = orc.pack_target(tgt_depth, bits)
    t_lookup = np.ascontiguousarray(lookup_depth, np.float32)

    count = [0]

    def E(n, a):                                       # render_at_pos + _error (predict.py:159-161,475-509)
        count[0] += 1
        key = o.raster_key(a, n)
        return float(o.finalize(o.sums(key, orc.LOSS_FULL, n, tq), orc.LOSS_FULL, n, n_pix, flags))

    lr = np.ones(6) * 0.1
    history = np.zeros((HISTORY_LENGTH, 6))
    err_history = np.zeros(HISTORY_LENGTH)
    angles = np.array([0] * 6, dtype=float)
    trace = []
    lim = np.asarray(joint_limits, float)
    cam = np.asarray(camera_pose, float)

    with np.errstate(all='ignore'):
        for st in (stages or STAGES[do_angles]):
            kind = st[0]
            if kind == 'lookup':                       # predict.py:165-171
                crop = np.asarray(lookup_crop, np.int32)
                npx = float((crop[1] - crop[0] + 1) * (crop[3] - crop[2] + 1))
                score = np.empty(len(lookup_angles))
                for i, a in enumerate(lookup_angles):
                    key = o.raster_key(a, 6)
                    score[i] = o.finalize(o.sums(key, orc.LOSS_LOOKUP, 6, None, t_lookup, crop), orc.LOSS_LOOKUP, 6, npx, flags)
                count[0] += len(lookup_angles)
                if lookup_live is not None:
                    assert isinstance(lookup_live, np.ndarray) and lookup_live.dtype == np.float64 and lookup_live.shape == np.shape(lookup_angles)
                    angles = lookup_live[int(np.argmin(score))]            # a view, as predict.py:171: the stages below edit or rebind it
                else:
                    angles = np.array(lookup_angles[int(np.argmin(score))], dtype=float)

            elif kind == 'descent':                    # predict.py:173-230
                _, n, its, letters, init_rate, redux, early = st
                for i in range(6):
                    if init_rate[i] is not None:
                        lr[i] = init_rate[i]
                for _ in range(its):
                    for idx in np.where(_mask(letters))[0]:
                        if abs(np.mean(history, 0)[idx] - angles[idx]) <= lr[idx]:
                            lr[idx] *= redux
                        lr = np.max((lr, min_ang_inc), 0)
                        temp = angles.copy()
                        temp[idx] -= lr[idx]
                        under_err = E(n, temp) if lim[idx][0] <= temp[idx] <= lim[idx][1] else np.inf
                        temp[idx] += 2 * lr[idx]
                        over_err = E(n, temp) if lim[idx][0] <= temp[idx] <= lim[idx][1] else np.inf
                        if over_err < under_err:
                            angles[idx] += lr[idx]
                        elif over_err > under_err:
                            angles[idx] -= lr[idx]
                    history[1:] = history[:-1]
                    history[0] = angles
                    err_history[1:] = err_history[:-1]
                    err_history[0] = min(over_err, under_err)
                    if abs(np.mean(err_history) - err_history[0]) / err_history[0] < early:
                        break
                    sp = history.max(0) - history.min(0)
                    if ((sp <= min_ang_inc) + np.isclose(sp, min_ang_inc)).all():
                        break
                    if (history[:3] == history[0]).all():
                        break

            elif kind == 'sflip':                      # predict.py:232-281
                n = st[1]
                base_err = E(n, angles)
                temp = angles.copy()
                a = cam[5] * np.abs(np.cos(cam[3])) + cam[4] * np.abs(np.sin(cam[3]))
                temp[0] = -temp[0] + 2 * a * np.sign(temp[0])
                close = 0.15 > abs(lim[0, 0] - temp[0]) or 0.15 > abs(lim[0, 1] - temp[0])
                inside = lim[0, 0] <= temp[0] <= lim[0, 1]
                if inside:
                    err = E(n, temp)
                    if err < base_err:
                        angles = temp
                        base_err = err
                if not inside or close:
                    for endpoint in lim[0]:            # comparison is outside the loop in the reference
                        temp[0] = endpoint
                        err = E(n, temp)
                    if err < base_err:
                        angles = temp
                        base_err = err

            elif kind == 'isweep':                     # predict.py:283-338
                _, n, div, letters, rng = st
                base_err = E(n, angles)
                for idx in np.where(_mask(letters))[0]:
                    lo, hi = angles.copy(), angles.copy()
                    if rng is None:
                        lo[idx], hi[idx] = lim[idx, 0], lim[idx, 1]
                    else:
                        lo[idx] = max(lo[idx] - rng, lim[idx, 0])
                        hi[idx] = min(hi[idx] + rng, lim[idx, 1])
                    space = np.linspace(lo, hi, div)
                    space_err = [E(n, a_) for a_ in space]
                    x = np.linspace(lo[idx], hi[idx], div * 5)
                    pred = interp1d(space[:, idx], np.array(space_err), kind='cubic')(x)
                    angs = angles.copy()
                    angs[idx] = x[pred.argmin()]
                    pred_min_err = E(n, angs)
                    errs = [base_err, min(space_err), pred_min_err]
                    which = errs.index(min(errs))
                    if which == 1:
                        angles = space[space_err.index(min(space_err))]
                        err_history[1:] = err_history[:-1]
                        err_history[0] = min(space_err)
                    elif which == 2:
                        angles = angs
                        err_history[1:] = err_history[:-1]
                        err_history[0] = pred_min_err
                    history[1:] = history[:-1]
                    history[0] = angles
            elif kind == 'tsweep':                     # predict.py:340-373; not in the default lists
                _, n, div, letters, rng = st
                t_full = np.ascontiguousarray(tgt_depth, np.float32)       # the whole target, both sides sqrt-ed
                for idx in np.where(_mask(letters))[0]:
                    lo, hi = angles.copy(), angles.copy()
                    if rng is None:
                        lo[idx], hi[idx] = lim[idx, 0], lim[idx, 1]
                    else:
                        lo[idx] = max(lo[idx] - rng, lim[idx, 0])
                        hi[idx] = min(hi[idx] + rng, lim[idx, 1])
                    space = np.linspace(lo, hi, div)
                    score = np.empty(div)
                    for i, a_ in enumerate(space):
                        key = o.raster_key(a_, n)
                        score[i] = o.finalize(o.sums(key, orc.LOSS_TSWEEP, n, None, t_full), orc.LOSS_TSWEEP, n, n_pix, flags)
                    count[0] += div
                    angles = space[int(np.argmin(score))]               # mean * -std: the `*-` of predict.py:367
            else:
                raise ValueError(kind)
            trace.append((kind, np.array(angles, dtype=float)))
    return angles, trace, count[0]
