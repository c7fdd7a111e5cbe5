"""Segmentation stage (plain-torch Mask R-CNN, random weights): structure on CPU, execution + adapter on the GPU."""
import numpy as np
import pytest
import torch

from rope_s3d_amd.maskrcnn import MaskRCNN, MaskRCNNSegmenter, _nms, _pyramid_anchors, _roi_align


def test_anchor_pyramid_and_parameter_count():
    a = _pyramid_anchors(512, 'cpu')
    assert a.shape == ((128 ** 2 + 64 ** 2 + 32 ** 2 + 16 ** 2 + 8 ** 2) * 3, 4)
    # first anchor of the stride-4 level: centre (0,0), scale 32, ratio 0.5 -> h = 32/sqrt(.5), w = 32*sqrt(.5), normalised by 511
    h, w = 32 / np.sqrt(0.5), 32 * np.sqrt(0.5)
    assert np.allclose(a[0].numpy(), [(-h / 2) / 511, (-w / 2) / 511, (h / 2 - 1) / 511, (w / 2 - 1) / 511], atol=1e-6)
    net = MaskRCNN(7)
    n = sum(p.numel() for p in net.parameters())
    assert 63.0e6 < n < 64.5e6                 # ResNet-101 + FPN Mask R-CNN with 7 classes
    assert len(net.backbone.stages[2]) == 23   # resnet101's third stage


def test_greedy_nms_and_roi_align():
    b = torch.tensor([[0, 0, 1, 1], [0, 0, .9, .9], [.5, .5, 1, 1], [0, 0, .5, .5]], dtype=torch.float32)
    s = torch.tensor([.9, .8, .7, .6])
    assert _nms(b, s, 0.5, 10).tolist() == [0, 2, 3] and _nms(b, s, 0.5, 2).tolist() == [0, 2]
    feats = [torch.linspace(0, 1, n).view(1, 1, 1, n).expand(1, 1, n, n).contiguous() for n in (128, 64, 32, 16)]
    out = _roi_align(feats, torch.tensor([[0.0, 0.25, 1.0, 0.75]]), 7, 512)
    assert np.allclose(out[0, 0, 0].numpy(), np.linspace(0.25, 0.75, 7), atol=1e-6)   # samples span the box, corners included


def test_cpu_forward_is_well_formed_and_deterministic():
    img = np.random.default_rng(0).integers(0, 255, (60, 80, 3), dtype=np.uint8)
    a = MaskRCNNSegmenter(7, device='cpu', seed=3, min_confidence=0.0)(img)
    b = MaskRCNNSegmenter(7, device='cpu', seed=3, min_confidence=0.0)(img)
    k = len(a['class_ids'])
    assert 0 < k <= 100 and a['masks'].shape == (60, 80, k) and a['masks'].dtype == bool
    assert (a['class_ids'] > 0).all() and (a['class_ids'] < 7).all() and (np.diff(a['scores']) <= 1e-6).all()
    assert np.array_equal(a['class_ids'], b['class_ids']) and np.array_equal(a['masks'], b['masks'])


@pytest.mark.gpu
def test_gpu_forward_and_predictor_adapter():
    from rope_s3d_amd import Predictor, Renderer
    from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
    seg = MaskRCNNSegmenter(7, device='cuda:0', seed=1, min_confidence=0.0)
    r = Renderer('seg', DEFAULT_CAMERA_POSE, '640_480_color')
    r.setJointAngles([0.3, 0.4, 0.9, 0, 0, 0])
    color, depth = r.render()
    out = seg(color)
    k = len(out['class_ids'])
    assert 0 < k <= 100 and out['masks'].shape == (480, 640, k)
    # whole pipeline on the segmentation path: untrained masks are meaningless, the plumbing must still hold
    p = Predictor(DEFAULT_CAMERA_POSE, 4, base_intrin='640_480_color', segmenter=seg, lookup_divisions=4)
    angles = p.run(color, depth.astype(np.float64))
    assert angles.shape == (6,) and np.isfinite(angles).all()
    assert (angles[:3] >= r.robot.joint_limits[:3, 0] - 1e-9).all() and (angles[:3] <= r.robot.joint_limits[:3, 1] + 1e-9).all()


def test_device_nms_equals_the_one_by_one_sweep():
    """The fixed-point iteration of _nms against the textbook greedy sweep (tf.image.non_max_suppression's rule)."""
    from rope_s3d_amd.maskrcnn import _nms

    def greedy(boxes, scores, thr, limit):
        order = np.argsort(-scores, kind='stable')
        b = boxes[order]
        area = np.clip(b[:, 2] - b[:, 0], 0, None) * np.clip(b[:, 3] - b[:, 1], 0, None)
        keep, alive = [], np.ones(len(b), bool)
        for i in range(len(b)):
            if not alive[i]:
                continue
            keep.append(i)
            if len(keep) >= limit:
                break
            inter = np.clip(np.minimum(b[i, 2:], b[:, 2:]) - np.maximum(b[i, :2], b[:, :2]), 0, None).prod(-1)
            alive &= ~(inter / np.clip(area[i] + area - inter, 1e-12, None) > thr)
        return order[keep]

    rng = np.random.default_rng(0)
    for n, thr, limit in ((1500, 0.7, 300), (400, 0.3, 100), (40, 0.5, 1000), (1, 0.5, 10), (5000, 0.7, 1000), (3000, 0.5, 5000)):
        c = rng.random((n, 2)).astype(np.float32)
        wh = (rng.random((n, 2)) * 0.2 + 0.02).astype(np.float32)
        boxes, scores = np.concatenate([c - wh / 2, c + wh / 2], 1), rng.random(n).astype(np.float32)
        for block in (2048, 300):                    # several blocks, carried-over suppression, early stop at the limit
            got = _nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr, limit, block=block).numpy()
            assert np.array_equal(got, greedy(boxes, scores, thr, limit)), (n, thr, limit, block)
    assert len(_nms(torch.zeros((0, 4)), torch.zeros(0), 0.5, 10)) == 0


def test_batched_trunk_gives_the_single_frame_results():
    """detect_batch runs backbone/FPN/RPN on all frames at once; per frame the results equal detect()'s (CPU float32:
    the same convolution code path for batch 1 and 2 up to rounding, so boxes may move by a hair but not change class)."""
    from rope_s3d_amd.maskrcnn import BatchAheadSegmenter
    seg = MaskRCNNSegmenter(7, device='cpu', seed=0, min_confidence=0.0)
    rng = np.random.default_rng(1)
    a, b = rng.integers(0, 255, (45, 80, 3), dtype=np.uint8), rng.integers(0, 255, (45, 80, 3), dtype=np.uint8)
    ra, rb_ = seg(a), seg(b)
    both = seg.batch([a, b])
    for single, batched in ((ra, both[0]), (rb_, both[1])):
        assert len(single['class_ids']) == len(batched['class_ids'])
        assert np.array_equal(single['class_ids'], batched['class_ids'])
        assert np.allclose(single['scores'], batched['scores'], atol=1e-4)
        assert (single['masks'] != batched['masks']).mean() < 1e-3
    ahead = BatchAheadSegmenter(seg, batch=2)
    ahead.announce([a, b])
    assert np.array_equal(ahead(b)['class_ids'], rb_['class_ids']) and not ahead._store.get(ahead._key(b))
    assert np.array_equal(ahead(b)['class_ids'], rb_['class_ids'])      # not stored any more: segmented on the spot
