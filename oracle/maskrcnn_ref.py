"""CPU restatement of the DETERMINISTIC half of the reference's segmentation stage — test infrastructure, not product code.

The reference segments a frame with PixelLib 0.5.6's `custom_segmentation` (robotpose/prediction/predict.py:94-98,416), which is
Matterport's Keras Mask R-CNN.  Neither PixelLib nor TensorFlow is under /root/reference (requirements.txt:1-17 pins
pixellib==0.5.6, tensorflow-gpu==2.4.1; scikit-image comes in unpinned through PixelLib), and no trained weights exist offline,
so the network's convolutions cannot be pinned.  What CAN be restated from the published algorithm (matterport/Mask_RCNN
mrcnn/model.py + mrcnn/utils.py, the code PixelLib vendors) is everything between the convolutions — the steps that decide which
boxes, classes and mask pixels come out of given network outputs:

    apply_box_deltas_graph / clip_boxes_graph       model.py  (ProposalLayer, refine_detections_graph)
    tf.image.non_max_suppression                    TF 2.4 NonMaxSuppressionV3: candidates by descending score (equal scores: the
                                                    lower index first), a candidate is kept unless its IoU with an already kept box
                                                    is strictly greater than the threshold, stop at max_output_size
    ProposalLayer                                   top PRE_NMS_LIMIT scores -> decode -> clip to [0, 1] -> NMS 0.7 -> 1000
    PyramidROIAlign                                 level = min(5, max(2, 4 + round(log2(sqrt(h w) / (224 / sqrt(image area)))))),
                                                    tf.image.crop_and_resize (bilinear, corner-inclusive sample grid, 0 outside)
    refine_detections_graph                         argmax class, class-specific deltas x BBOX_STD_DEV, clip to the window, drop the
                                                    background and scores < DETECTION_MIN_CONFIDENCE, per-class NMS 0.3 (at most 100
                                                    per class), top 100 by score
    unmold_detections / unmold_mask                 utils.py: boxes back to image pixels (np.around), the 28x28 mask resized to its box
                                                    (skimage.transform.resize, order 1, mode 'constant', cval 0, half-pixel centres),
                                                    >= 0.5 -> integer mask pasted at the box

Everything here is numpy, one box at a time, in the order the published code works; parity unpinned in the sense of the task
statement: the reference holds no fixture for this stage and its libraries cannot run here.  tests/test_maskrcnn.py feeds these
functions the trunk / head outputs of the torch model (same random weights) and compares the decisions of the HIP + torch stage.
"""
import math

import numpy as np

RPN_BBOX_STD_DEV = np.array([0.1, 0.1, 0.2, 0.2], np.float32)
BBOX_STD_DEV = np.array([0.1, 0.1, 0.2, 0.2], np.float32)
PRE_NMS_LIMIT, POST_NMS_ROIS, RPN_NMS_THRESHOLD = 6000, 1000, 0.7
DETECTION_MAX_INSTANCES, DETECTION_NMS_THRESHOLD = 100, 0.3


def apply_box_delta(box, delta):
    """model.py apply_box_deltas_graph for one box [y1, x1, y2, x2] and one refinement [dy, dx, log(dh), log(dw)], float32."""
    box, delta = np.asarray(box, np.float32), np.asarray(delta, np.float32)
    h, w = box[2] - box[0], box[3] - box[1]
    cy, cx = box[0] + np.float32(0.5) * h, box[1] + np.float32(0.5) * w
    cy, cx = cy + delta[0] * h, cx + delta[1] * w
    h, w = h * np.exp(delta[2]), w * np.exp(delta[3])
    y1, x1 = cy - np.float32(0.5) * h, cx - np.float32(0.5) * w
    return np.array([y1, x1, y1 + h, x1 + w], np.float32)


def clip_box(box, window):
    """model.py clip_boxes_graph: every coordinate into the window [wy1, wx1, wy2, wx2]."""
    wy1, wx1, wy2, wx2 = [np.float32(v) for v in window]
    y1, x1, y2, x2 = [np.float32(v) for v in box]
    return np.array([max(min(y1, wy2), wy1), max(min(x1, wx2), wx1), max(min(y2, wy2), wy1), max(min(x2, wx2), wx1)], np.float32)


def iou(a, b) -> float:
    """TF NonMaxSuppression's IOU: corners may come in either order; a box without area overlaps nothing."""
    ay1, ax1, ay2, ax2 = min(a[0], a[2]), min(a[1], a[3]), max(a[0], a[2]), max(a[1], a[3])
    by1, bx1, by2, bx2 = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])
    area_a, area_b = (ay2 - ay1) * (ax2 - ax1), (by2 - by1) * (bx2 - bx1)
    if area_a <= 0 or area_b <= 0:
        return 0.0
    ih, iw = max(min(ay2, by2) - max(ay1, by1), np.float32(0)), max(min(ax2, bx2) - max(ax1, bx1), np.float32(0))
    inter = ih * iw
    return float(inter / (area_a + area_b - inter))


def non_max_suppression(boxes, scores, max_output_size: int, iou_threshold: float):
    """tf.image.non_max_suppression -> indices kept, best first.  One candidate at a time, against the boxes kept so far."""
    boxes, scores = np.asarray(boxes, np.float32), np.asarray(scores, np.float32)
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))          # descending score, equal scores: first come
    kept = []
    for i in order:
        if len(kept) >= max_output_size:
            break
        if all(not (iou(boxes[i], boxes[j]) > iou_threshold) for j in kept):
            kept.append(i)
    return kept


def proposal_layer(scores, deltas, anchors, proposal_count: int = POST_NMS_ROIS, nms_threshold: float = RPN_NMS_THRESHOLD,
                   pre_nms_limit: int = PRE_NMS_LIMIT):
    """model.py ProposalLayer.call for one image: foreground scores (A,), deltas (A, 4), normalised anchors (A, 4)
    -> (anchor indices of the proposals, their boxes), best first."""
    scores = np.asarray(scores, np.float32)
    k = min(pre_nms_limit, len(scores))
    top = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))[:k]        # tf.nn.top_k(sorted=True)
    boxes = np.stack([clip_box(apply_box_delta(anchors[i], np.asarray(deltas[i], np.float32) * RPN_BBOX_STD_DEV), (0, 0, 1, 1)) for i in top])
    keep = non_max_suppression(boxes, scores[top], proposal_count, nms_threshold)
    return [top[j] for j in keep], boxes[keep]


def roi_level(box, image_size: int) -> int:
    """model.py PyramidROIAlign: the pyramid level (2..5) of a normalised box for a square image of image_size pixels."""
    h, w = np.float32(box[2]) - np.float32(box[0]), np.float32(box[3]) - np.float32(box[1])
    image_area = np.float32(image_size * image_size)
    lv = np.log(np.sqrt(h * w) / (np.float32(224.0) / np.sqrt(image_area))) / np.log(np.float32(2.0))      # log2_graph
    return int(min(5, max(2, 4 + int(np.rint(lv)))))


def crop_and_resize(fmap, box, crop: int):
    """tf.image.crop_and_resize (bilinear, extrapolation_value 0) of one feature map (H, W, C) and one normalised box -> (crop, crop, C)."""
    H, W = fmap.shape[:2]
    y1, x1, y2, x2 = [float(v) for v in box]
    out = np.zeros((crop, crop, fmap.shape[2]), np.float64)
    hs = (y2 - y1) * (H - 1) / (crop - 1) if crop > 1 else 0.0
    ws = (x2 - x1) * (W - 1) / (crop - 1) if crop > 1 else 0.0
    for y in range(crop):
        in_y = y1 * (H - 1) + y * hs if crop > 1 else 0.5 * (y1 + y2) * (H - 1)
        if in_y < 0 or in_y > H - 1:
            continue
        top, bottom, ly = int(math.floor(in_y)), int(math.ceil(in_y)), in_y - math.floor(in_y)
        for x in range(crop):
            in_x = x1 * (W - 1) + x * ws if crop > 1 else 0.5 * (x1 + x2) * (W - 1)
            if in_x < 0 or in_x > W - 1:
                continue
            left, right, lx = int(math.floor(in_x)), int(math.ceil(in_x)), in_x - math.floor(in_x)
            t = fmap[top, left] + (fmap[top, right] - fmap[top, left]) * lx
            b = fmap[bottom, left] + (fmap[bottom, right] - fmap[bottom, left]) * lx
            out[y, x] = t + (b - t) * ly
    return out


def refine_detections(rois, probs, deltas, window, min_confidence: float, max_instances: int = DETECTION_MAX_INSTANCES,
                      nms_threshold: float = DETECTION_NMS_THRESHOLD):
    """model.py refine_detections_graph for one image: rois (N, 4), class probabilities (N, classes), class-specific deltas
    (N, classes, 4), the image window in normalised coordinates -> (indices into rois, class ids, scores, refined boxes), best first."""
    probs = np.asarray(probs, np.float32)
    n = len(rois)
    class_ids = [int(np.argmax(probs[i])) for i in range(n)]
    class_scores = np.array([probs[i, class_ids[i]] for i in range(n)], np.float32)
    refined = np.stack([clip_box(apply_box_delta(rois[i], np.asarray(deltas[i][class_ids[i]], np.float32) * BBOX_STD_DEV), window)
                        for i in range(n)]) if n else np.zeros((0, 4), np.float32)
    keep = [i for i in range(n) if class_ids[i] > 0 and (not min_confidence or class_scores[i] >= np.float32(min_confidence))]
    nms_keep = []
    for c in sorted(set(class_ids[i] for i in keep)):                        # per-class NMS over the boxes that passed the filters
        ixs = [i for i in keep if class_ids[i] == c]
        class_keep = non_max_suppression(refined[ixs], class_scores[ixs], max_instances, nms_threshold)
        nms_keep += [ixs[j] for j in class_keep]
    keep = [i for i in keep if i in set(nms_keep)]
    keep = sorted(keep, key=lambda i: (-float(class_scores[i]), i))[:max_instances]          # tf.nn.top_k(sorted=True)
    return keep, [class_ids[i] for i in keep], class_scores[keep], refined[keep]


def denorm_box(box, shape):
    """utils.denorm_boxes: normalised [y1, x1, y2, x2] -> pixel coordinates, (y2, x2) outside the box (np.around: halves to even)."""
    h, w = shape
    scale, shift = np.array([h - 1, w - 1, h - 1, w - 1], np.float64), np.array([0, 0, 1, 1], np.float64)
    return np.around(np.asarray(box, np.float64) * scale + shift).astype(np.int32)


def resize_bilinear_constant(img, out_h: int, out_w: int):
    """skimage.transform.resize(img, (out_h, out_w), order=1, mode='constant', cval=0, anti_aliasing=False), as Matterport's
    utils.resize calls it: output sample r looks at input coordinate (r + 0.5) * in / out - 0.5, bilinear, and what lies outside
    the input counts as 0 (so the outermost output rows fade towards 0)."""
    img = np.asarray(img, np.float64)
    in_h, in_w = img.shape
    out = np.zeros((out_h, out_w))

    def px(r, c):
        return img[r, c] if 0 <= r < in_h and 0 <= c < in_w else 0.0
    for r in range(out_h):
        y = (r + 0.5) * in_h / out_h - 0.5
        y0 = math.floor(y)
        fy = y - y0
        for c in range(out_w):
            x = (c + 0.5) * in_w / out_w - 0.5
            x0 = math.floor(x)
            fx = x - x0
            out[r, c] = (px(y0, x0) * (1 - fx) + px(y0, x0 + 1) * fx) * (1 - fy) + (px(y0 + 1, x0) * (1 - fx) + px(y0 + 1, x0 + 1) * fx) * fy
    return out


def unmold_mask(mask, bbox, image_shape, threshold: float = 0.5):
    """utils.unmold_mask: the network's (28, 28) mask of one detection -> a full-image boolean mask, and the resized values (for
    the caller to see how close to the threshold a pixel was)."""
    y1, x1, y2, x2 = [int(v) for v in bbox]
    full = np.zeros(image_shape[:2], bool)
    vals = resize_bilinear_constant(mask, y2 - y1, x2 - x1)
    full[y1:y2, x1:x2] = vals >= threshold
    return full, vals


def unmold_detections(boxes_norm, masks, original_shape, image_shape, window_px):
    """model.py MaskRCNN.unmold_detections for one image: detection boxes in normalised coordinates of the moulded (padded) image,
    their (28, 28) masks, the original image's (H, W), the moulded image's (H, W), the window of the original inside the moulded
    image in pixels [y1, x1, y2, x2] -> (pixel boxes, [(full-size boolean mask, resized values, pixel box)], indices of the
    detections kept: zero-area boxes go).  Dtypes as the published code has them: utils.norm_boxes returns float32, the detections
    are float32, denorm_boxes multiplies by integer arrays (float64) and rounds halves to even."""
    ih, iw = image_shape
    window = ((np.asarray(window_px, np.float64) - np.array([0, 0, 1, 1])) / np.array([ih - 1, iw - 1, ih - 1, iw - 1])).astype(np.float32)
    wy1, wx1, wy2, wx2 = window
    shift, scale = np.array([wy1, wx1, wy1, wx1]), np.array([wy2 - wy1, wx2 - wx1, wy2 - wy1, wx2 - wx1])          # float32
    out_boxes, out_masks, kept = [], [], []
    for i, b in enumerate(boxes_norm):
        rel = np.divide(np.asarray(b, np.float32) - shift, scale)                                                   # float32
        bx = denorm_box(rel, original_shape)
        if (bx[2] - bx[0]) * (bx[3] - bx[1]) <= 0:
            continue
        # Matterport pastes the resized mask at the box as it is (a box reaching past the image would raise there); the window clip
        # of refine_detections keeps boxes inside, and this restatement clips the paste like the product does
        y1, x1, y2, x2 = max(bx[0], 0), max(bx[1], 0), min(bx[2], original_shape[0]), min(bx[3], original_shape[1])
        full = np.zeros(original_shape, bool)
        vals = resize_bilinear_constant(masks[i], bx[2] - bx[0], bx[3] - bx[1])
        full[y1:y2, x1:x2] = (vals >= 0.5)[y1 - bx[0]:y2 - bx[0], x1 - bx[1]:x2 - bx[1]]
        out_boxes.append(bx)
        out_masks.append((full, vals, bx))
        kept.append(i)
    return out_boxes, out_masks, kept
