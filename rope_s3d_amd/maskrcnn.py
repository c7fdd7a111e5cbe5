"""Mask R-CNN inference on PyTorch-ROCm — the segmentation stage in front of the HIP engine.

The reference segments the RGB frame with PixelLib 0.5.6 `custom_segmentation` (Matterport's Keras
Mask R-CNN, ResNet-101 + FPN; robotpose/prediction/predict.py:94-98,416; trained in train.py:49).
This module restates that network's INFERENCE graph in plain torch (torchvision is not in the image):
ResNet-101 C1..C5, FPN P2..P6, RPN over 5 scales x 3 ratios, proposal layer (top 6000, NMS 0.7,
1000 kept), pyramid RoIAlign (7x7 head, 14x14 mask), two-layer classifier head, detection layer
(score >= 0.7, per-class NMS 0.3, 100 instances), mask head (4 convs + deconv, 28x28 sigmoid), and the
un-moulding of masks into the image.  PixelLib's inference config for this call: 512x512 square input,
detection threshold 0.7 (pixellib custom_segmentation.inferConfig defaults).

No trained weights exist offline (models/*.h5 are git-ignored upstream), so by default the
network is random-initialised: it exercises the dense-contraction path (MIOpen convolutions on the
matrix cores: bf16 weights, batch norms folded) and the adapter into `Predictor._segmentLoad`, but its masks
are not comparable with the reference's.  `load_matterport_weights` converts a trained Keras weight file of that
network (the `.h5` PixelLib / Matterport write, training/models.py:180-324 picks one) into this module's state_dict.
The dense work is the only part of the prediction path where MFMA is the right tool; everything after
it (FK, raster, loss) is the hand-written HIP engine.
"""
import math
import os
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

MEAN_PIXEL = (123.7, 116.8, 103.9)               # Matterport Config.MEAN_PIXEL (RGB)
BACKBONE_STRIDES = (4, 8, 16, 32, 64)
RPN_ANCHOR_SCALES = (32, 64, 128, 256, 512)
RPN_ANCHOR_RATIOS = (0.5, 1.0, 2.0)
RPN_BBOX_STD_DEV = (0.1, 0.1, 0.2, 0.2)
BBOX_STD_DEV = (0.1, 0.1, 0.2, 0.2)
PRE_NMS_LIMIT, POST_NMS_ROIS, RPN_NMS_THRESHOLD = 6000, 1000, 0.7
DETECTION_MAX_INSTANCES, DETECTION_NMS_THRESHOLD = 100, 0.3
POOL_SIZE, MASK_POOL_SIZE, MASK_SHAPE = 7, 14, 28
TOP_DOWN_PYRAMID_SIZE, FPN_CLASSIF_FC = 256, 1024


def _bn(c):
    return nn.BatchNorm2d(c, eps=1e-3)             # Keras BatchNormalization default epsilon; frozen at inference


class _Bottleneck(nn.Module):
    """Matterport identity_block / conv_block: 1x1 -> 3x3 -> 1x1 with a projection shortcut when the shape changes."""

    def __init__(self, cin, mid, stride, project):
        super().__init__()
        self.c1, self.b1 = nn.Conv2d(cin, mid, 1, stride), _bn(mid)
        self.c2, self.b2 = nn.Conv2d(mid, mid, 3, 1, 1), _bn(mid)
        self.c3, self.b3 = nn.Conv2d(mid, mid * 4, 1), _bn(mid * 4)
        self.short = nn.Sequential(nn.Conv2d(cin, mid * 4, 1, stride), _bn(mid * 4)) if project else None

    def forward(self, x):
        if isinstance(self.b1, nn.Identity) and _fusable(x):      # batch norms folded, bf16 on the GPU: one pass after every convolution
            y = _conv_act(self.c2, _conv_act(self.c1, x))
            return _conv_act(self.c3, y, res=x if self.short is None else self.short(x))
        y = F.relu(self.b1(self.c1(x)))
        y = F.relu(self.b2(self.c2(y)))
        y = self.b3(self.c3(y))
        return F.relu(y + (x if self.short is None else self.short(x)))


class _ResNet(nn.Module):
    def __init__(self, blocks=(3, 4, 23, 3)):       # resnet101
        super().__init__()
        # Keras MaxPooling2D((3,3), strides 2, 'same') on an even-sized map pads one row/column at the END only
        # (TensorFlow's SAME rule), so the windows start at 0 — not at -1 as MaxPool2d(padding=1) would have them
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3), _bn(64), nn.ReLU(), nn.ConstantPad2d((0, 1, 0, 1), 0.0), nn.MaxPool2d(3, 2))
        stages, cin = [], 64
        for i, (n, mid) in enumerate(zip(blocks, (64, 128, 256, 512))):
            layers = [_Bottleneck(cin, mid, 1 if i == 0 else 2, True)]
            layers += [_Bottleneck(mid * 4, mid, 1, False) for _ in range(n - 1)]
            stages.append(nn.Sequential(*layers))
            cin = mid * 4
        self.stages = nn.ModuleList(stages)

    def forward(self, x):
        x = self.stem(x)
        out = []
        for s in self.stages:
            x = s(x)
            out.append(x)
        return out                                   # C2, C3, C4, C5


class _FPN(nn.Module):
    def __init__(self, c=TOP_DOWN_PYRAMID_SIZE):
        super().__init__()
        self.lat = nn.ModuleList([nn.Conv2d(k, c, 1) for k in (256, 512, 1024, 2048)])
        self.smooth = nn.ModuleList([nn.Conv2d(c, c, 3, 1, 1) for _ in range(4)])

    def forward(self, feats):
        c2, c3, c4, c5 = feats
        p5 = self.lat[3](c5)
        p4 = self.lat[2](c4) + F.interpolate(p5, scale_factor=2, mode='nearest')
        p3 = self.lat[1](c3) + F.interpolate(p4, scale_factor=2, mode='nearest')
        p2 = self.lat[0](c2) + F.interpolate(p3, scale_factor=2, mode='nearest')
        p2, p3, p4, p5 = [s(p) for s, p in zip(self.smooth, (p2, p3, p4, p5))]
        return [p2, p3, p4, p5, F.max_pool2d(p5, 1, 2)]          # P6 for the RPN only


class _RPN(nn.Module):
    def __init__(self, c=TOP_DOWN_PYRAMID_SIZE, a=len(RPN_ANCHOR_RATIOS)):
        super().__init__()
        self.shared = nn.Conv2d(c, 512, 3, 1, 1)
        self.cls, self.box = nn.Conv2d(512, 2 * a, 1), nn.Conv2d(512, 4 * a, 1)

    def forward(self, p):
        h = _conv_act(self.shared, p) if _fusable(p) else F.relu(self.shared(p))
        logits = self.cls(h).permute(0, 2, 3, 1).reshape(p.shape[0], -1, 2)
        deltas = self.box(h).permute(0, 2, 3, 1).reshape(p.shape[0], -1, 4)
        return logits.float().softmax(-1)[..., 1], deltas.float()


_anchor_cache = {}


def _pyramid_anchors(size: int, device) -> torch.Tensor:
    """Matterport utils.generate_pyramid_anchors (anchor stride 1), normalised to [0,1] as norm_boxes does.
    A function of the input size alone: built once per (size, device)."""
    key = (size, str(device))
    if key not in _anchor_cache:
        _anchor_cache[key] = _build_pyramid_anchors(size, device)
    return _anchor_cache[key]


def _build_pyramid_anchors(size: int, device) -> torch.Tensor:
    out = []
    for scale, stride in zip(RPN_ANCHOR_SCALES, BACKBONE_STRIDES):
        n = int(math.ceil(size / stride))
        ratios = torch.tensor(RPN_ANCHOR_RATIOS, device=device)
        hs, ws = scale / ratios.sqrt(), scale * ratios.sqrt()
        ys = torch.arange(n, device=device, dtype=torch.float32) * stride
        cy, cx = torch.meshgrid(ys, ys, indexing='ij')
        cy, cx = cy[..., None].expand(n, n, 3), cx[..., None].expand(n, n, 3)
        b = torch.stack([cy - 0.5 * hs, cx - 0.5 * ws, cy + 0.5 * hs, cx + 0.5 * ws], -1).reshape(-1, 4)
        out.append(b)
    a = torch.cat(out)
    scale = torch.tensor([size - 1, size - 1, size - 1, size - 1], device=device, dtype=torch.float32)
    shift = torch.tensor([0, 0, 1, 1], device=device, dtype=torch.float32)
    return (a - shift) / scale


# row counts the classifier / mask head are run on are multiples of these: divisors of the usual counts (1000 proposals and up to
# 100 detections per frame), so the usual case runs unpadded
HEAD_ROW_STEP, MASK_ROW_STEP = 250, 50


def _pad_rows(t: torch.Tensor, step: int) -> torch.Tensor:
    """t with zero rows appended up to the next multiple of `step` (at least one step)."""
    n = max(-(-len(t) // step), 1) * step
    return t if n == len(t) else torch.cat([t, t.new_zeros((n - len(t),) + tuple(t.shape[1:]))])


_CONSTS = {}


def _const(values, device, dtype=None) -> torch.Tensor:
    """torch.tensor(values, device=device) made once per (values, device, dtype): creating a tensor from host data on a GPU is a
    blocking copy that waits for everything queued before it — a dozen of them per batch made the host and the device take turns."""
    key = (tuple(values), str(device), dtype)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(list(values), device=device, dtype=dtype)
    return t


_SEG_LIB = None


def _seg_lib():
    """librope_hip.so (rope_seg.hip: NMS and pyramid RoIAlign).  On a GPU the stage's box steps run there; a missing library is an
    error, not a reason to fall back (EngineUnavailable)."""
    global _SEG_LIB
    if _SEG_LIB is None:
        from .engine import load_library
        _SEG_LIB = load_library()
    return _SEG_LIB


def _seg_kernels(t: torch.Tensor) -> bool:
    """The HIP kernels serve tensors on the GPU; ROPE_SEG_HIP=0 keeps the tensor formulation there too (equality tests, timing)."""
    import os
    return t.is_cuda and os.environ.get('ROPE_SEG_HIP', '1') != '0'


def _fusable(x: torch.Tensor) -> bool:
    return x.dtype == torch.bfloat16 and _seg_kernels(x)


def _conv_act(conv: nn.Conv2d, x: torch.Tensor, res: torch.Tensor = None, relu: bool = True) -> torch.Tensor:
    """relu(conv(x) + bias [+ res]) with everything after the library convolution in ONE pass over its output (rope_seg_bias_act,
    in place) instead of a bias pass, an addition and a ReLU — the same roundings to bfloat16 as those separate operations."""
    y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    n, c, hw = y.numel(), y.shape[1], y.shape[2] * y.shape[3]
    if y.is_contiguous() and hw % 8 == 0:
        inner = hw
    elif y.is_contiguous(memory_format=torch.channels_last) and c % 8 == 0:
        inner = 1
    else:
        inner = 0
    if res is not None and (res.shape != y.shape or res.stride() != y.stride()):
        res = res.contiguous(memory_format=torch.channels_last if inner == 1 else torch.contiguous_format) if inner else res
        if inner and res.stride() != y.stride():
            inner = 0
    if inner == 0 or conv.bias is None or n % 8:
        if conv.bias is not None:
            y = y + conv.bias.view(1, -1, 1, 1)
        if res is not None:
            y = y + res
        return F.relu(y) if relu else y
    rc = _seg_lib().rope_seg_bias_act(y.data_ptr(), conv.bias.data_ptr(), None if res is None else res.data_ptr(), n, c, inner, int(relu),
                                      torch.cuda.current_stream(y.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"rope_seg_bias_act failed ({rc})")
    return y


def _apply_deltas(boxes, deltas):
    """Matterport apply_box_deltas_graph, step for step (the far corner is the near one plus the size, not centre plus half)."""
    h, w = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cy, cx = boxes[:, 0] + 0.5 * h, boxes[:, 1] + 0.5 * w
    cy, cx = cy + deltas[:, 0] * h, cx + deltas[:, 1] * w
    h, w = h * deltas[:, 2].exp(), w * deltas[:, 3].exp()
    y1, x1 = cy - 0.5 * h, cx - 0.5 * w
    return torch.stack([y1, x1, y1 + h, x1 + w], 1)


def _iou_over(a: torch.Tensor, b: torch.Tensor, thr: float) -> torch.Tensor:
    """(len(a), len(b)) bool: IoU(a_i, b_j) > thr, coordinate by coordinate (no stacked temporaries)."""
    ih = (torch.min(a[:, None, 2], b[None, :, 2]) - torch.max(a[:, None, 0], b[None, :, 0])).clamp_(min=0)
    iw = (torch.min(a[:, None, 3], b[None, :, 3]) - torch.max(a[:, None, 1], b[None, :, 1])).clamp_(min=0)
    inter = ih.mul_(iw)
    area_a = (a[:, 2] - a[:, 0]).clamp(min=0) * (a[:, 3] - a[:, 1]).clamp(min=0)
    area_b = (b[:, 2] - b[:, 0]).clamp(min=0) * (b[:, 3] - b[:, 1]).clamp(min=0)
    return inter / (area_a[:, None] + area_b[None, :] - inter).clamp_(min=1e-12) > thr


def _iou_over_b(a: torch.Tensor, b: torch.Tensor, thr: float) -> torch.Tensor:
    """(B, len(a), len(b)) bool: IoU(a[f, i], b[f, j]) > thr for every frame f."""
    ih = (torch.min(a[:, :, None, 2], b[:, None, :, 2]) - torch.max(a[:, :, None, 0], b[:, None, :, 0])).clamp_(min=0)
    iw = (torch.min(a[:, :, None, 3], b[:, None, :, 3]) - torch.max(a[:, :, None, 1], b[:, None, :, 1])).clamp_(min=0)
    inter = ih.mul_(iw)
    area_a = (a[..., 2] - a[..., 0]).clamp(min=0) * (a[..., 3] - a[..., 1]).clamp(min=0)
    area_b = (b[..., 2] - b[..., 0]).clamp(min=0) * (b[..., 3] - b[..., 1]).clamp(min=0)
    return inter / (area_a[:, :, None] + area_b[:, None, :] - inter).clamp_(min=1e-12) > thr


def _nms_batched(boxes: torch.Tensor, scores: torch.Tensor, thr: float, limit: int, block: int = 2048, valid: torch.Tensor = None,
                 groups: torch.Tensor = None) -> torch.Tensor:
    """Greedy non-maximum suppression (tf.image.non_max_suppression) of B independent box sets at once:
    boxes (B, N, 4), scores (B, N) -> keep (B, N) bool over the INPUT order, at most `limit` per set (the best ones).

    The greedy rule "box j stays unless an earlier box that stays overlaps it" has exactly one solution; instead of
    sweeping the boxes one by one it is iterated as a whole on the device — keep <- alive and not any(earlier & keep &
    overlap) — until nothing changes (after t rounds at least the first t boxes are final; a dozen rounds settle
    thousands).  Boxes go through in blocks of `block` by descending score: a block is first thinned by the boxes kept
    in earlier blocks, then settled internally, and the walk stops as soon as every set holds `limit` boxes — most of
    the pairwise overlaps of a 6000-proposal frame are never computed, and nothing runs per box or per frame on the host.
    `valid` (B, N) masks padding; `groups` (B, N) restricts suppression to boxes of the same group (per-class NMS)."""
    B, N = scores.shape
    keep_in = torch.zeros((B, N), dtype=torch.bool, device=boxes.device)
    if N == 0:
        return keep_in
    key = scores if valid is None else torch.where(valid, scores, torch.full_like(scores, -float('inf')))
    order = key.argsort(dim=1, descending=True, stable=True)
    b = boxes.gather(1, order[..., None].expand(-1, -1, 4)).float()
    ok = torch.ones((B, N), dtype=torch.bool, device=boxes.device) if valid is None else valid.gather(1, order)
    g = None if groups is None else groups.gather(1, order)
    if _seg_kernels(boxes):
        # librope_hip.so: the suppression bits of all pairs in one launch, one wave per set settling them in a second
        # (rope_seg.hip) — same overlaps, same greedy rule, no host synchronisation
        lib = _seg_lib()
        b = b.contiguous()
        words = (N + 63) // 64
        scratch = torch.empty((B, N, words), dtype=torch.int64, device=boxes.device)
        kept = torch.empty((B, N), dtype=torch.bool, device=boxes.device)
        g32 = None if g is None else g.to(torch.int32).contiguous()
        okc = None if valid is None else ok.contiguous()
        rc = lib.rope_seg_nms(b.data_ptr(), None if g32 is None else g32.data_ptr(), None if okc is None else okc.data_ptr(),
                              B, N, float(thr), int(min(limit, N)), scratch.data_ptr(), kept.data_ptr(),
                              torch.cuda.current_stream(boxes.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"rope_seg_nms failed ({rc})")
        return keep_in.scatter_(1, order, kept)
    kept = torch.zeros((B, N), dtype=torch.bool, device=boxes.device)          # over the sorted order
    for start in range(0, N, block):
        blk = b[:, start:start + block]
        m = blk.shape[1]
        alive = ok[:, start:start + m].clone()
        if start:
            over = _iou_over_b(b[:, :start], blk, thr) & kept[:, :start, None]
            if g is not None:
                over &= g[:, :start, None] == g[:, None, start:start + m]
            alive &= ~over.any(1)
        sup = _iou_over_b(blk, blk, thr).triu_(1)                              # [f, i, j]: earlier i suppresses j
        if g is not None:
            sup &= g[:, start:start + m, None] == g[:, None, start:start + m]
        keep = alive
        for _ in range(0, m, 4):                                               # four rounds between convergence checks: the
            prev = keep                                                        # fixed point is stable under further rounds,
            for _ in range(4):                                                 # and every check is a host synchronisation
                prev, keep = keep, alive & ~(sup & keep[:, :, None]).any(1)
            if torch.equal(prev, keep):
                break
        kept[:, start:start + m] = keep
        if bool((kept.sum(1) >= limit).all()):
            break
    kept &= kept.cumsum(1) <= limit                                            # the first `limit` of every set
    return keep_in.scatter_(1, order, kept)


def _nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float, limit: int, block: int = 2048) -> torch.Tensor:
    """One box set: indices kept, best first."""
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    keep = _nms_batched(boxes[None], scores[None], thr, limit, block)[0]
    idx = keep.nonzero().squeeze(1)
    return idx[scores[idx].argsort(descending=True, stable=True)]


def _pack_levels(feats: List[torch.Tensor]):
    """P2..P5 as one table of channel rows, one per feature pixel of every frame and level, plus where each level starts."""
    lv = feats[:4]
    C = lv[0].shape[1]
    rows = torch.cat([f.permute(0, 2, 3, 1).reshape(-1, C) for f in lv])
    dev = rows.device
    Hs = _const([f.shape[2] for f in lv], dev)
    Ws = _const([f.shape[3] for f in lv], dev)
    sizes = [f.shape[0] * f.shape[2] * f.shape[3] for f in lv]
    offs = _const([sum(sizes[:k]) for k in range(4)], dev)
    return rows, Hs, Ws, offs


def _roi_align(feats: List[torch.Tensor], boxes: torch.Tensor, pool: int, size: int, frame: torch.Tensor = None, packed=None) -> torch.Tensor:
    """PyramidROIAlign: level by box area, then tf.image.crop_and_resize (bilinear, pool x pool samples spanning
    the box, corners included) — for all boxes of all levels and frames in one pass: four gathers of whole channel
    rows from the packed level table (`_pack_levels`), no per-level loop, no per-ROI feature copies.
    `frame` (K,) names the batch entry every box belongs to (default: entry 0)."""
    rows, Hs, Ws, offs = packed if packed is not None else _pack_levels(feats)
    K, C = len(boxes), rows.shape[1]
    if frame is None:
        frame = torch.zeros(K, dtype=torch.long, device=boxes.device)
    if K and rows.dtype == torch.bfloat16 and _seg_kernels(boxes):
        # librope_hip.so: level choice, the four gathers and the bilinear blend in one launch (rope_seg.hip); the same float32
        # and bfloat16 steps as the tensor formulation below
        lib = _seg_lib()
        lv = feats[:4]
        sizes = [f.shape[0] * f.shape[2] * f.shape[3] for f in lv]
        level_hw = np.array([[f.shape[2], f.shape[3]] for f in lv], np.int32)
        level_off = np.array([sum(sizes[:k]) for k in range(4)], np.int64)
        rows_c, b = rows.contiguous(), boxes.float().contiguous()
        out = torch.empty((K, pool, pool, C), dtype=torch.bfloat16, device=boxes.device)
        t = torch.linspace(0, 1, pool, device=boxes.device)
        inv_unit = np.float32(1.0) / np.float32(224.0 / size)                 # the tensor division by a scalar multiplies by its inverse
        f32 = frame.to(torch.int32).contiguous()
        rc = lib.rope_seg_roi_align(rows_c.data_ptr(), b.data_ptr(), f32.data_ptr(), level_hw.ctypes.data, level_off.ctypes.data,
                                    K, C, pool, float(inv_unit), t.data_ptr(), out.data_ptr(),
                                    torch.cuda.current_stream(boxes.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"rope_seg_roi_align failed ({rc})")
        return out.permute(0, 3, 1, 2)
    b = boxes.float()
    h, w = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    li = (4 + torch.log2((h * w).clamp(min=1e-12).sqrt() / (224.0 / size)).round()).clamp(2, 5).long() - 2
    Hf, Wf = Hs[li], Ws[li]                                         # (K,) size of every box's level
    base = (offs[li] + frame * (Hf * Wf))[:, None, None]
    t = torch.linspace(0, 1, pool, device=boxes.device)
    ys = (b[:, 0:1] + t[None, :] * (b[:, 2:3] - b[:, 0:1])) * (Hf - 1)[:, None]      # (K, P) in feature pixels
    xs = (b[:, 1:2] + t[None, :] * (b[:, 3:4] - b[:, 1:2])) * (Wf - 1)[:, None]
    y0, x0 = ys.floor(), xs.floor()
    wy, wx = (ys - y0).to(rows.dtype)[:, :, None, None], (xs - x0).to(rows.dtype)[:, None, :, None]
    inside = (((ys >= 0) & (ys <= (Hf - 1)[:, None]))[:, :, None] & ((xs >= 0) & (xs <= (Wf - 1)[:, None]))[:, None, :])   # crop_and_resize: 0 outside the map
    hm, wm = (Hf - 1)[:, None], (Wf - 1)[:, None]
    y0c, y1c = y0.long().clamp(min=0).minimum(hm), (y0.long() + 1).clamp(min=0).minimum(hm)
    x0c, x1c = x0.long().clamp(min=0).minimum(wm), (x0.long() + 1).clamp(min=0).minimum(wm)
    Wk = Wf[:, None, None]

    def g(yi, xi):                                                  # -> (K, P, P, C): whole channel rows by flat pixel index
        return rows.index_select(0, (base + yi[:, :, None] * Wk + xi[:, None, :]).reshape(-1)).view(K, pool, pool, C)
    val = (g(y0c, x0c) * (1 - wy) + g(y1c, x0c) * wy) * (1 - wx) + (g(y0c, x1c) * (1 - wy) + g(y1c, x1c) * wy) * wx
    return (val * inside[..., None].to(rows.dtype)).permute(0, 3, 1, 2)


def _fold_batchnorm(module: nn.Module) -> nn.Module:
    """Inference only: fold every frozen BatchNorm2d into the convolution in front of it (fewer launches, same function)."""
    def fold(conv, bn):
        w = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        conv.weight.data = conv.weight.data * w.view(-1, 1, 1, 1)
        bias = conv.bias.data if conv.bias is not None else torch.zeros_like(bn.running_mean)
        conv.bias = nn.Parameter((bias - bn.running_mean) * w + bn.bias.data)

    for m in module.modules():
        if isinstance(m, _Bottleneck):
            for c, b in (('c1', 'b1'), ('c2', 'b2'), ('c3', 'b3')):
                fold(getattr(m, c), getattr(m, b))
                setattr(m, b, nn.Identity())
            if m.short is not None:
                fold(m.short[0], m.short[1])
                m.short[1] = nn.Identity()
        elif isinstance(m, nn.Sequential):
            kids = list(m.children())
            for i in range(len(kids) - 1):
                if isinstance(kids[i], nn.Conv2d) and isinstance(kids[i + 1], nn.BatchNorm2d):
                    fold(kids[i], kids[i + 1])
                    m[i + 1] = nn.Identity()
    return module


class MaskRCNN(nn.Module):

    def __init__(self, num_classes: int = 7, image_size: int = 512, min_confidence: float = 0.7):
        super().__init__()
        self.num_classes, self.size, self.min_conf = num_classes, image_size, min_confidence
        self.backbone, self.fpn, self.rpn = _ResNet(), _FPN(), _RPN()
        c = TOP_DOWN_PYRAMID_SIZE
        self.head = nn.Sequential(nn.Conv2d(c, FPN_CLASSIF_FC, POOL_SIZE), _bn(FPN_CLASSIF_FC), nn.ReLU(),
                                  nn.Conv2d(FPN_CLASSIF_FC, FPN_CLASSIF_FC, 1), _bn(FPN_CLASSIF_FC), nn.ReLU())
        self.cls, self.box = nn.Linear(FPN_CLASSIF_FC, num_classes), nn.Linear(FPN_CLASSIF_FC, num_classes * 4)
        mask = []
        for _ in range(4):
            mask += [nn.Conv2d(c, c, 3, 1, 1), _bn(c), nn.ReLU()]
        self.mask = nn.Sequential(*mask, nn.ConvTranspose2d(c, c, 2, 2), nn.ReLU(), nn.Conv2d(c, num_classes, 1))

    @torch.no_grad()
    def detect(self, image_rgb: torch.Tensor):
        """image_rgb (H,W,3) uint8 on the module's device -> (class_ids (K,), scores (K,), masks (H,W,K) bool)."""
        return self.detect_batch([image_rgb])[0]

    @torch.no_grad()
    def detect_batch(self, images):
        """Several frames of one size at once: the convolutional trunk (backbone, FPN, RPN heads — the dense contraction)
        runs on the whole batch, proposals / RoIAlign / heads / un-moulding then per frame.  Same results per frame as
        `detect` up to the batch-size dependence of the convolution algorithms the library picks."""
        x, geo = self._mould(images)
        feats_all, probs, deltas = self._trunk_replayed(x)
        return self._detect(feats_all, probs, deltas, *geo)

    def _mould(self, images):
        """resize_image(mode='square') + mold_image of a batch -> (B, 3, size, size) in the weights' dtype, and the geometry to undo it."""
        dev = images[0].device
        H, W = images[0].shape[:2]
        scale = self.size / max(H, W)
        nh, nw = round(H * scale), round(W * scale)
        top, left = (self.size - nh) // 2, (self.size - nw) // 2
        x = torch.stack([im.permute(2, 0, 1) for im in images]).float()
        x = F.interpolate(x, (nh, nw), mode='bilinear', align_corners=False)
        x = x - _const(MEAN_PIXEL, dev).view(1, 3, 1, 1)
        x = F.pad(x, (left, self.size - nw - left, top, self.size - nh - top))
        wdt = next(self.parameters()).dtype                              # bf16 on the GPU (cast once), f32 on CPU
        return x.to(wdt).contiguous(), (H, W, scale, top, left, nh, nw)

    @torch.no_grad()
    def detect_batches(self, batches):
        """`detect_batch` of every batch of an iterable, in order (a generator).  On the GPU the trunk of batch i+1 is put on a
        second stream before the box steps of batch i start: those steps are many short launches and a few waits of the host
        for the device, which leave the GPU idle most of the time — the next trunk runs in those gaps.  Same operations on the
        same data as batch by batch; the trunk's outputs are copied out of the replayed graph's buffers before it runs again."""
        it = iter(batches)
        first = next(it, None)
        if first is None:
            return
        if first[0].device.type != 'cuda':
            yield self.detect_batch(first)
            for b in it:
                yield self.detect_batch(b)
            return
        if not hasattr(self, '_side'):
            self._side = torch.cuda.Stream(first[0].device)
        main = torch.cuda.current_stream(first[0].device)

        def start(images):
            x, geo = self._mould(images)
            ready = torch.cuda.Event()
            ready.record(main)
            x.record_stream(self._side)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                feats, probs, deltas = self._trunk_replayed(x)
                out = ([f.clone() for f in feats], probs.clone(), deltas.clone())
                done = torch.cuda.Event()
                done.record(self._side)
            return out, geo, done

        def finish(token):
            (feats, probs, deltas), geo, done = token
            main.wait_event(done)
            for t in (*feats, probs, deltas):
                t.record_stream(main)
            return self._detect(feats, probs, deltas, *geo)

        token = start(first)
        for b in it:
            nxt = start(b)
            yield finish(token)
            token = nxt
        yield finish(token)

    def _trunk(self, x):
        """Backbone, FPN and the RPN heads: the static-shape, dense part -> (P2..P6, objectness (B, anchors), deltas (B, anchors, 4))."""
        # the ResNet trunk runs in NCHW, the 256-channel FPN / RPN convolutions in channels-last (each the faster layout
        # for its shapes on gfx950) — and channels-last maps make _pack_levels a view
        cl = (lambda t: t.contiguous(memory_format=torch.channels_last)) if x.is_cuda else (lambda t: t)
        feats = self.fpn([cl(c) for c in self.backbone(x)])
        rpn = [self.rpn(p) for p in feats]
        return feats, torch.cat([r[0] for r in rpn], 1), torch.cat([r[1] for r in rpn], 1)

    def _trunk_replayed(self, x):
        """On the GPU the trunk is ~400 small launches whose issue time exceeds their run time: it is captured once per
        input shape into a HIP graph and replayed with one launch (static input and output buffers; the outputs are
        consumed by _detect before the next replay, on the same stream).  ROPE_SEG_GRAPH=0, or a failed capture, runs
        it eagerly — the same operations either way."""
        import os
        if not x.is_cuda or os.environ.get('ROPE_SEG_GRAPH', '1') == '0' or getattr(self, '_graph_broken', False):
            return self._trunk(x)
        if not hasattr(self, '_graphs'):
            self._graphs = {}
        key = (tuple(x.shape), x.dtype)
        if key not in self._graphs:
            try:
                static_x = x.clone()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):                                   # MIOpen settles its kernel choice outside the capture
                        self._trunk(static_x)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = self._trunk(static_x)
                self._graphs[key] = (graph, static_x, static_out)
            except Exception as e:                                       # noqa: BLE001 — capture support varies; eager is the same math
                import warnings
                warnings.warn(f"HIP graph capture of the Mask R-CNN trunk failed ({e}); running it eagerly")
                self._graph_broken = True
                torch.cuda.synchronize()
                return self._trunk(x)
        graph, static_x, static_out = self._graphs[key]
        static_x.copy_(x)
        graph.replay()
        return static_out

    def _detect(self, feats, probs, deltas, H, W, scale, top, left, nh, nw):
        """Proposal layer, pyramid RoIAlign, classifier head, detection layer, mask head and un-moulding for all B frames
        of the batch together: every step works on the concatenation of the frames' boxes with a frame index beside
        them; only the final split of the masks is per frame."""
        dev, B = probs.device, probs.shape[0]
        window = _const([top, left, top + nh, left + nw], dev, torch.float32)
        anchors = _pyramid_anchors(self.size, dev)
        k = min(PRE_NMS_LIMIT, probs.shape[1])
        top_p, top_idx = probs.topk(k, dim=1)                               # (B, k)
        d = deltas.gather(1, top_idx[..., None].expand(-1, -1, 4)) * _const(RPN_BBOX_STD_DEV, dev)
        boxes = _apply_deltas(anchors[top_idx].reshape(-1, 4), d.reshape(-1, 4)).clamp(0, 1).view(B, k, 4)
        keep = _nms_batched(boxes, top_p, RPN_NMS_THRESHOLD, POST_NMS_ROIS)
        frame, pos = keep.nonzero(as_tuple=True)                            # proposals of all frames, frame-major, by anchor rank
        rois = boxes[frame, pos]
        packed = _pack_levels(feats)
        # The heads see row counts in steps (padding rows: empty boxes of frame 0, dropped again): the count of proposals that
        # survive NMS moves by a few from batch to batch, and every new count is a new convolution shape the library has to
        # find a kernel for first (50-200 ms the first time it meets one).
        rois_p, frame_p = _pad_rows(rois, HEAD_ROW_STEP), _pad_rows(frame, HEAD_ROW_STEP)
        h = self.head(_roi_align(feats, rois_p, POOL_SIZE, self.size, frame_p, packed)).flatten(1)
        cls_prob = self.cls(h)[:len(rois)].float().softmax(-1)
        box_delta = self.box(h)[:len(rois)].float().view(-1, self.num_classes, 4)
        cls_id = cls_prob.argmax(1)
        score = cls_prob.gather(1, cls_id[:, None])[:, 0]
        d = box_delta[torch.arange(len(rois), device=dev), cls_id] * _const(BBOX_STD_DEV, dev)
        nwin = (window - _const([0, 0, 1, 1], dev)) / (self.size - 1)
        refined = _apply_deltas(rois, d)
        refined = torch.stack([refined[:, 0].clamp(nwin[0], nwin[2]), refined[:, 1].clamp(nwin[1], nwin[3]),
                               refined[:, 2].clamp(nwin[0], nwin[2]), refined[:, 3].clamp(nwin[1], nwin[3])], 1)
        cand = ((cls_id > 0) & (score >= self.min_conf)).nonzero().squeeze(1)
        empty = (torch.zeros(0, dtype=torch.long), torch.zeros(0), torch.zeros((H, W, 0), dtype=torch.bool))
        if cand.numel() == 0:
            return [empty] * B
        # per-class NMS of every frame in one pass (refine_detections_graph): suppression only inside (frame, class) groups
        grp = frame[cand] * self.num_classes + cls_id[cand]
        kept = _nms_batched(refined[cand][None], score[cand][None], DETECTION_NMS_THRESHOLD, len(cand), groups=grp[None])[0]
        # at most DETECTION_MAX_INSTANCES per class, then per frame, by score
        sel = cand[kept]
        sel = sel[score[sel].argsort(descending=True, stable=True)]

        def first_n(keys, n):                                               # keep the first n (in the current, score order) of every key
            o = keys.argsort(stable=True)
            ks = keys[o]
            start = torch.cat([ks.new_zeros(1, dtype=torch.bool) | True, ks[1:] != ks[:-1]])
            rank = torch.arange(len(ks), device=dev) - torch.cummax(torch.where(start, torch.arange(len(ks), device=dev), 0), 0).values
            ok = torch.zeros(len(ks), dtype=torch.bool, device=dev)
            ok[o] = rank < n
            return ok
        sel = sel[first_n(frame[sel] * self.num_classes + cls_id[sel], DETECTION_MAX_INSTANCES)]
        sel = sel[first_n(frame[sel], DETECTION_MAX_INSTANCES)]
        sel = sel[frame[sel].argsort(stable=True)]                          # frame-major, best first inside a frame
        det_boxes, det_cls, det_score, det_frame = refined[sel], cls_id[sel], score[sel], frame[sel]
        sel_all, det_boxes_all = sel, det_boxes
        m = self.mask(_roi_align(feats, _pad_rows(det_boxes, MASK_ROW_STEP), MASK_POOL_SIZE, self.size, _pad_rows(det_frame, MASK_ROW_STEP),
                                 packed))[:len(sel)].float().sigmoid()
        m = m[torch.arange(len(sel), device=dev), det_cls]                  # (K, 28, 28) of each detection's class
        # un-mould (MaskRCNN.unmold_detections): boxes relative to the window, normalised as utils.norm_boxes does (float32), then
        # utils.denorm_boxes onto the ORIGINAL image: round-half-even of box * (H - 1, W - 1) + (0, 0, 1, 1) in float64
        wn = ((window.double() - _const([0, 0, 1, 1], dev, torch.float64)) / (self.size - 1)).float()
        shift = torch.stack([wn[0], wn[1], wn[0], wn[1]])
        span = torch.stack([wn[2] - wn[0], wn[3] - wn[1], wn[2] - wn[0], wn[3] - wn[1]])
        rel = (det_boxes - shift) / span
        px = (rel.double() * _const([H - 1, W - 1, H - 1, W - 1], dev, torch.float64) + _const([0, 0, 1, 1], dev, torch.float64)).round().long()
        ok = (px[:, 2] - px[:, 0]) * (px[:, 3] - px[:, 1]) > 0                     # zero-area boxes are dropped (exclude_ix)
        if not bool(ok.all()):
            px, m, det_cls, det_score, det_frame = px[ok], m[ok], det_cls[ok], det_score[ok], det_frame[ok]
        # utils.unmold_mask: every detection's 28x28 mask resized to its box (skimage.transform.resize, order 1, mode 'constant',
        # cval 0: output sample r looks at (r + 0.5) * 28 / h - 0.5, and what lies outside the mask counts as 0), >= 0.5, pasted at
        # the box — all detections in one sampling pass over the image grid: pixel (Y, X) of a box starting at (y1, x1) is its
        # sample (Y - y1, X - x1), which grid_sample's half-pixel convention with zero padding reads at 2 (Y + 0.5 - y1) / h - 1
        y1r, x1r, y2r, x2r = px[:, 0], px[:, 1], px[:, 2], px[:, 3]
        y1, x1 = y1r.clamp(min=0), x1r.clamp(min=0)
        y2, x2 = y2r.clamp(max=H), x2r.clamp(max=W)
        Y = torch.arange(H, device=dev, dtype=torch.float32)[None, :, None] + 0.5
        X = torch.arange(W, device=dev, dtype=torch.float32)[None, None, :] + 0.5
        hh, ww = (y2r - y1r).clamp(min=1).float()[:, None, None], (x2r - x1r).clamp(min=1).float()[:, None, None]
        gy = 2 * (Y - y1r.float()[:, None, None]) / hh - 1
        gx = 2 * (X - x1r.float()[:, None, None]) / ww - 1
        grid = torch.stack([gx.expand(-1, H, W), gy.expand(-1, H, W)], -1)
        val = F.grid_sample(m[:, None], grid, mode='bilinear', padding_mode='zeros', align_corners=False)[:, 0]
        inside = (Y > y1[:, None, None]) & (Y < y2[:, None, None]) & (X > x1[:, None, None]) & (X < x2[:, None, None])
        masks = (val >= 0.5) & inside                                       # (K, H, W)
        if getattr(self, 'keep_trace', False):          # tests hold every decision of the stage against oracle/maskrcnn_ref.py
            self.trace = dict(anchors=anchors, probs=probs, deltas=deltas, top_idx=top_idx, top_p=top_p, decoded=boxes, keep=keep, rois=rois,
                              roi_frame=frame, cls_prob=cls_prob, box_delta=box_delta, refined=refined, sel=sel_all, ok=ok, det_boxes=det_boxes_all,
                              det_cls=det_cls, det_score=det_score, det_frame=det_frame, mask28=m, px=px, resized=val, window=window,
                              feats=feats, geometry=(H, W, scale, top, left, nh, nw))
        if dev.type == 'cuda':                                              # one transfer for the whole batch, through pinned memory
            host = torch.empty(masks.shape, dtype=torch.bool, pin_memory=True)
            host.copy_(masks, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            masks = host
        else:
            masks = masks.cpu()
        det_cls, det_score, counts = det_cls.cpu(), det_score.cpu(), torch.bincount(det_frame, minlength=B).cpu().tolist()
        out, at = [], 0
        for n in counts:
            out.append((det_cls[at:at + n], det_score[at:at + n], masks[at:at + n].permute(1, 2, 0).contiguous()) if n else empty)
            at += n
        return out


def matterport_layer_map(num_classes: int = 7) -> dict:
    """state_dict module prefix -> (Keras layer name, kind) for Matterport's resnet101 Mask R-CNN (mrcnn/model.py:
    resnet_graph, fpn, build_rpn_model, fpn_classifier_graph, build_fpn_mask_graph)."""
    m = {'backbone.stem.0': ('conv1', 'conv'), 'backbone.stem.1': ('bn_conv1', 'bn')}
    for s, n in enumerate((3, 4, 23, 3)):
        for b in range(n):
            blk = 'abcdefghijklmnopqrstuvwxyz'[b]                # stage 4 of resnet101: 'a', then chr(98 + i) = 'b'..'w'
            tag = f'{s + 2}{blk}'
            pre = f'backbone.stages.{s}.{b}'
            for j, br in enumerate(('2a', '2b', '2c')):
                m[f'{pre}.c{j + 1}'] = (f'res{tag}_branch{br}', 'conv')
                m[f'{pre}.b{j + 1}'] = (f'bn{tag}_branch{br}', 'bn')
            if b == 0:
                m[f'{pre}.short.0'] = (f'res{tag}_branch1', 'conv')
                m[f'{pre}.short.1'] = (f'bn{tag}_branch1', 'bn')
    for i in range(4):
        m[f'fpn.lat.{i}'] = (f'fpn_c{i + 2}p{i + 2}', 'conv')
        m[f'fpn.smooth.{i}'] = (f'fpn_p{i + 2}', 'conv')
    m.update({'rpn.shared': ('rpn_conv_shared', 'conv'), 'rpn.cls': ('rpn_class_raw', 'conv'), 'rpn.box': ('rpn_bbox_pred', 'conv'),
              'head.0': ('mrcnn_class_conv1', 'conv'), 'head.1': ('mrcnn_class_bn1', 'bn'),
              'head.3': ('mrcnn_class_conv2', 'conv'), 'head.4': ('mrcnn_class_bn2', 'bn'),
              'cls': ('mrcnn_class_logits', 'dense'), 'box': ('mrcnn_bbox_fc', 'dense'),
              'mask.12': ('mrcnn_mask_deconv', 'deconv'), 'mask.14': ('mrcnn_mask', 'conv')})
    for i in range(4):
        m[f'mask.{3 * i}'] = (f'mrcnn_mask_conv{i + 1}', 'conv')
        m[f'mask.{3 * i + 1}'] = (f'mrcnn_mask_bn{i + 1}', 'bn')
    return m


def keras_to_torch(kind: str, var: str, value: np.ndarray) -> np.ndarray:
    """One Keras variable in torch's layout.  Conv2D kernels are (kh, kw, in, out) -> (out, in, kh, kw); Conv2DTranspose
    kernels (kh, kw, out, in) -> ConvTranspose2d's (in, out, kh, kw) — the same axis permutation, no flip: both
    frameworks define the op as the gradient of their cross-correlation; Dense kernels (in, out) -> (out, in)."""
    if var == 'kernel':
        return np.ascontiguousarray(value.T if kind == 'dense' else value.transpose(3, 2, 0, 1))
    return np.ascontiguousarray(value)


_KERAS_VARS = {'conv': (('weight', 'kernel'), ('bias', 'bias')), 'deconv': (('weight', 'kernel'), ('bias', 'bias')),
               'dense': (('weight', 'kernel'), ('bias', 'bias')),
               'bn': (('weight', 'gamma'), ('bias', 'beta'), ('running_mean', 'moving_mean'), ('running_var', 'moving_variance'))}


def load_matterport_weights(path: str, num_classes: int = 7) -> dict:
    """A Keras `save_weights` HDF5 file of Matterport's Mask R-CNN (what PixelLib trains and loads: predict.py:96-98,
    train.py:49) -> state_dict of `MaskRCNN(num_classes)`.

    Keras keeps every variable as a dataset `<layer>/<scope>/<var>:0`, possibly below a `model_weights` group and, for
    the RPN, below the nested `rpn_model`; only the last two path components identify a variable, so the file is walked
    and indexed by `<scope>/<var>`.  Missing layers and shape mismatches (e.g. a head trained for another class count)
    raise with the layer's name."""
    try:
        import h5py
        f, found = h5py.File(path, 'r'), {}
        f.visititems(lambda n, o: found.__setitem__(n, o) if isinstance(o, h5py.Dataset) else None)
        read = lambda n: np.asarray(found[n])
        names = list(found)
    except ImportError:
        from .data.hdf5 import H5File
        f = H5File(path)
        names = list(f.walk())
        read = lambda n: np.asarray(f[n])
    index = {}
    for n in names:
        parts = n.split('/')
        if len(parts) >= 2:
            index[f"{parts[-2]}/{parts[-1].split(':')[0]}"] = n
    ref = MaskRCNN(num_classes).state_dict()
    out = {k: v for k, v in ref.items() if k.endswith('num_batches_tracked')}
    for prefix, (layer, kind) in matterport_layer_map(num_classes).items():
        for tname, kname in _KERAS_VARS[kind]:
            key = f'{layer}/{kname}'
            if key not in index:
                f.close()
                raise KeyError(f"{path}: no '{key}' (layer {layer} of the Matterport Mask R-CNN); is this a resnet101 weight file?")
            value = keras_to_torch(kind, kname, read(index[key]).astype(np.float32))
            want = tuple(ref[f'{prefix}.{tname}'].shape)
            if value.shape != want:
                f.close()
                raise ValueError(f"{path}: {key} has shape {value.shape} here, the network wants {want}"
                                 + (f" (trained for another number of classes than {num_classes}?)" if layer.startswith('mrcnn_') else ''))
            out[f'{prefix}.{tname}'] = torch.from_numpy(value)
    f.close()
    missing = set(ref) - set(out)
    assert not missing, missing
    return out


class MaskRCNNSegmenter:
    """`segmenter=` adapter: colour frame (H,W,3 uint8, BGR as the dataset stores it) -> PixelLib-style result dict.

    Create it (or call torch.cuda.init()) BEFORE the first Engine / Predictor of the process: PyTorch ships its own
    copy of the HIP runtime, and it cannot take the GPU once librope_hip.so has initialised the system copy."""

    def __init__(self, num_classes: int = 7, device: str = 'cuda:0', state_dict: dict = None, seed: int = 0,
                 min_confidence: float = 0.7):
        torch.manual_seed(seed)
        self.device = torch.device(device)
        self.net = MaskRCNN(num_classes, min_confidence=min_confidence)
        if state_dict is not None:
            self.net.load_state_dict(state_dict)
        self.net = _fold_batchnorm(self.net.eval()).to(self.device)
        if self.device.type == 'cuda':                                   # weights to bf16 once: MFMA path of MIOpen / hipBLASLt
            from .utils import limit_host_threads
            limit_host_threads()                                           # see there: idle pool threads spinning freeze the process under a CPU quota
            self.net = self.net.to(torch.bfloat16)
            if os.environ.get('ROPE_SEG_FIND') == '1':
                # MIOpen's find step (timed trials per convolution shape) instead of its immediate-mode pick: the batch of eight
                # 13.4 instead of 15.7 ms (10.5 / 11.5 pipelined) for half a minute more of start-up on a machine without a
                # find database of these shapes (profiles/r03_seg_trace.txt) — for long-running services, not for a 1 000-frame set
                torch.backends.cudnn.benchmark = True
            for part in (self.net.fpn, self.net.rpn):                      # see detect_batch for the layouts
                part.to(memory_format=torch.channels_last)

    def __call__(self, color_bgr: np.ndarray) -> dict:
        return self.batch([color_bgr])[0]

    def batch(self, frames) -> list:
        """Several frames of one size in one pass of the convolutional trunk -> list of result dicts."""
        return self._results(self.net.detect_batch(self._upload(frames)))

    def batches(self, groups):
        """`batch` of every group of frames of an iterable, in order (a generator); on the GPU the next group's trunk overlaps
        the current group's box steps (MaskRCNN.detect_batches)."""
        for out in self.net.detect_batches(self._upload(g) for g in groups):
            yield self._results(out)

    def _upload(self, frames):
        return [torch.from_numpy(np.ascontiguousarray(f[..., ::-1])).to(self.device) for f in frames]

    @staticmethod
    def _results(out):
        return [{'class_ids': cls.numpy(), 'scores': score.numpy(), 'masks': masks.numpy()} for cls, score, masks in out]


class BatchAheadSegmenter:
    """Segments the frames a caller is about to predict in batches, then hands the results out one by one.

    `announce(frames)` takes the down-sampled colour images of the coming frames (any number) and runs them through
    `segmenter.batch` in groups of `batch` on a background thread; `__call__(frame)` returns the result of that frame
    (matched by content) as soon as its group is through — the network works ahead of the stage machine, which spends
    most of its time inside the library with the interpreter lock released — or segments the frame on the spot if it
    was not announced.  `background=False` does the batches inside announce()."""

    def __init__(self, segmenter: MaskRCNNSegmenter, batch: int = 8, background: bool = True):
        import threading
        self._seg, self._batch, self._background = segmenter, int(batch), background
        self._store, self._pending, self._error = {}, {}, None
        self._cv = threading.Condition()
        self._seg_lock = threading.Lock()                 # one user of the network (and of its graph buffers) at a time

    @staticmethod
    def _key(color: np.ndarray):
        a = np.ascontiguousarray(color)
        return (a.shape, a[::5, ::5].tobytes())

    def announce(self, frames):
        import threading
        frames = [np.asarray(f) for f in frames]
        keys = [self._key(f) for f in frames]
        with self._cv:
            for k in keys:
                self._pending[k] = self._pending.get(k, 0) + 1
        if self._background:
            threading.Thread(target=self._work, args=(frames, keys), daemon=True).start()
        else:
            self._work(frames, keys)

    def _work(self, frames, keys):
        try:
            groups = [frames[i:i + self._batch] for i in range(0, len(frames), self._batch)]
            # The lock is held for the whole announced chunk: the generator keeps the next group's trunk in flight between two
            # results, so the network cannot be lent out in between.  A frame that was NOT announced (__call__'s fall-through)
            # therefore waits for the chunk to finish, not for one batch — announce what you are going to ask for.
            with self._seg_lock:
                for n, results in enumerate(self._seg.batches(groups)):
                    i = n * self._batch
                    with self._cv:
                        for k, r in zip(keys[i:i + self._batch], results):
                            self._store.setdefault(k, []).append(r)
                            self._pending[k] -= 1
                        self._cv.notify_all()
        except BaseException as e:                        # noqa: BLE001 — handed to the consumer, which re-raises it
            with self._cv:
                self._error = e
                self._cv.notify_all()

    def __call__(self, color: np.ndarray) -> dict:
        k = self._key(color)
        with self._cv:
            while True:
                if self._error is not None:
                    raise self._error
                ready = self._store.get(k)
                if ready:
                    r = ready.pop(0)
                    if not ready:
                        del self._store[k]
                    return r
                if self._pending.get(k, 0) <= 0:
                    break                                 # never announced (or already handed out): segment it now
                self._cv.wait()
        with self._seg_lock:
            return self._seg(color)
