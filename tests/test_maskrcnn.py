"""Segmentation stage (plain-torch Mask R-CNN, random weights): structure on CPU, execution + adapter on the GPU."""
import numpy as np
import pytest
import torch

from rope_s3d_amd.maskrcnn import MaskRCNN, MaskRCNNSegmenter, _nms, _pyramid_anchors, _roi_align

import helpers


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


def test_keras_layouts_mean_the_same_operation():
    """keras_to_torch: the converted weight computes what Keras computes with the original (Conv2D 'valid', HWIO kernel;
    Conv2DTranspose 2x2 stride 2, (kh, kw, out, in) kernel; Dense (in, out))."""
    import torch.nn.functional as F
    from rope_s3d_amd.maskrcnn import keras_to_torch
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, 6, 3))                                        # H, W, C as Keras sees a feature map
    k = rng.normal(size=(3, 2, 3, 4))                                     # kh, kw, in, out
    want = np.zeros((3, 5, 4))
    for y in range(3):
        for xx in range(5):
            want[y, xx] = np.einsum('abi,abio->o', x[y:y + 3, xx:xx + 2], k)
    got = F.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(keras_to_torch('conv', 'kernel', k)))[0].permute(1, 2, 0)
    assert np.allclose(got.numpy(), want)
    kt = rng.normal(size=(2, 2, 4, 3))                                    # kh, kw, out, in
    want = np.zeros((10, 12, 4))
    for i in range(5):
        for j in range(6):
            for a in range(2):
                for b in range(2):
                    want[2 * i + a, 2 * j + b] = kt[a, b] @ x[i, j]
    got = F.conv_transpose2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(keras_to_torch('deconv', 'kernel', kt)), stride=2)[0].permute(1, 2, 0)
    assert np.allclose(got.numpy(), want)
    kd = rng.normal(size=(3, 4))
    assert np.allclose(F.linear(torch.from_numpy(x), torch.from_numpy(keras_to_torch('dense', 'kernel', kd))).numpy(), x @ kd)
    assert keras_to_torch('bn', 'gamma', kd[0]) is not kd[0] and np.array_equal(keras_to_torch('bn', 'gamma', kd[0]), kd[0])


def test_stem_pooling_has_tensorflow_same_alignment():
    """Keras MaxPooling2D(3, strides 2, 'same') on an even map: windows [2i, 2i+2], the pad sits at the end."""
    net = MaskRCNN(7)
    pool = torch.nn.Sequential(*list(net.backbone.stem)[3:])
    x = torch.rand(1, 2, 8, 10)
    want = torch.zeros(1, 2, 4, 5)
    for i in range(4):
        for j in range(5):
            want[..., i, j] = x[..., 2 * i:2 * i + 3, 2 * j:2 * j + 3].amax((-1, -2))
    assert torch.equal(pool(x), want)


def test_matterport_weight_file_round_trip(tmp_path):
    """load_matterport_weights on a Keras-layout file (variables at <layer>/<scope>/<var>:0, the RPN nested in rpn_model,
    everything under model_weights as model.save() puts it): every tensor of the network comes back, in torch layout."""
    from rope_s3d_amd.data import hdf5
    from rope_s3d_amd.maskrcnn import load_matterport_weights, matterport_layer_map
    if not hdf5.available():
        pytest.skip("no libhdf5 on this machine")
    torch.manual_seed(3)
    net = MaskRCNN(7)
    for m in net.modules():                                               # running statistics away from their 0 / 1 defaults
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2.0)
    sd = net.state_dict()
    lm = matterport_layer_map(7)
    assert lm['backbone.stages.2.22.c3'] == ('res4w_branch2c', 'conv') and lm['backbone.stages.3.0.short.1'] == ('bn5a_branch1', 'bn')
    assert lm['fpn.lat.3'] == ('fpn_c5p5', 'conv') and lm['mask.12'] == ('mrcnn_mask_deconv', 'deconv')
    path = helpers.write_keras_weights(sd, str(tmp_path / 'mask_rcnn_model.h5'))
    arrays = helpers.keras_arrays(sd)
    got = load_matterport_weights(path, 7)
    assert set(got) == set(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    MaskRCNN(7).load_state_dict(got)                                      # strict
    with pytest.raises(ValueError, match='another number of classes'):
        load_matterport_weights(path, 5)
    del arrays['model_weights/res3b_branch2b/res3b_branch2b/kernel:0']
    path2 = hdf5.write_arrays(str(tmp_path / 'broken.h5'), arrays)
    with pytest.raises(KeyError, match='res3b_branch2b/kernel'):
        load_matterport_weights(path2, 7)


@pytest.mark.gpu
def test_predictor_loads_the_trained_model_of_its_dataset(tmp_path):
    """Predictor(model_ds=...) without a segmenter: the newest model trained on that dataset is found under MODELS
    (ModelManager.dynamicLoad, predict.py:94-98), its Keras checkpoint is converted, and frames go through it."""
    import json
    from rope_s3d_amd.data import hdf5
    if not hdf5.available():
        pytest.skip("no libhdf5 on this machine")
    from rope_s3d_amd import Predictor, SyntheticPredictor
    from rope_s3d_amd.config import Paths
    from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
    torch.cuda.init()
    torch.manual_seed(1)
    folder = tmp_path / 'models' / 'QXZV'
    folder.mkdir(parents=True)
    (folder / 'ModelData.json').write_text(json.dumps({'id': 'QXZV', 'dataset': 'set10', 'dataset_size': 10, 'train_size': 8, 'valid_size': 2,
                                                       'classes': ['BG'], 'date_trained': '2021-03-01 10:00:00.000000'}))
    want = MaskRCNN(7).state_dict()
    helpers.write_keras_weights(want, str(folder / 'mask_rcnn_model.007-0.120.h5'))
    old = Paths().MODELS
    Paths().set('MODELS', str(tmp_path / 'models'))
    try:
        p = Predictor(DEFAULT_CAMERA_POSE, 4, base_intrin='640_480_color', model_ds='set10', lookup_divisions=4)
        none = Predictor(DEFAULT_CAMERA_POSE, 4, base_intrin='640_480_color', model_ds='set10', lookup_divisions=4, segmenter=lambda c: None)
    finally:
        Paths().set('MODELS', old)
    assert isinstance(p.seg, MaskRCNNSegmenter) and not isinstance(none.seg, MaskRCNNSegmenter)
    sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '640_480_color', 4, 'SLU', noise=False, seed=2, lookup_divisions=4)
    sp.renderer.setJointAngles([0.3, 0.2, 0.5, 0, 0, 0])
    color, depth = sp.renderer.render()
    out = p.run(color, depth.astype(np.float64))
    assert out.shape == (6,) and np.isfinite(out).all()


def test_batched_nms_equals_set_by_set():
    """_nms_batched: B box sets at once, padding masked out, per-class groups — against _nms run set by set / group by
    group (which test_device_nms_equals_the_one_by_one_sweep ties to the literal greedy sweep)."""
    from rope_s3d_amd.maskrcnn import _nms_batched
    g = torch.Generator().manual_seed(5)
    B, N = 3, 700
    c = torch.rand(B, N, 2, generator=g)
    wh = torch.rand(B, N, 2, generator=g) * 0.2 + 0.02
    boxes = torch.cat([c - wh / 2, c + wh / 2], -1)
    scores = torch.rand(B, N, generator=g)
    valid = torch.ones(B, N, dtype=torch.bool)
    valid[1, 400:] = False                                                  # a shorter set, padded
    keep = _nms_batched(boxes, scores, 0.3, 50, block=256, valid=valid)
    for f in range(B):
        n = int(valid[f].sum())
        want = _nms(boxes[f, :n], scores[f, :n], 0.3, 50, block=256)
        assert sorted(keep[f].nonzero().squeeze(1).tolist()) == sorted(want.tolist()) and not keep[f, n:].any()
    groups = torch.randint(0, 4, (1, N), generator=g)
    keep = _nms_batched(boxes[:1], scores[:1], 0.3, N, block=128, groups=groups)[0]
    want = []
    for k in range(4):
        ix = (groups[0] == k).nonzero().squeeze(1)
        want += ix[_nms(boxes[0, ix], scores[0, ix], 0.3, N)].tolist()
    assert sorted(keep.nonzero().squeeze(1).tolist()) == sorted(want)
    assert _nms_batched(boxes[:, :0], scores[:, :0], 0.3, 5).shape == (B, 0)


def test_inference_configuration_is_the_references():
    """What the reference's own calls fix about the network, asserted against this module's constants:
    train.py:49 — modelConfig(network_backbone="resnet101", num_classes=len(class_names)) with class_names = the renderer's
    colour dictionary (one entry per rendered link, render.py:155-163); predict.py:97 — inferConfig(num_classes=6,
    class_names=["BG"] + mesh_names[:6]).  PixelLib 0.5.6 (requirements.txt; absent here) turns both into Matterport's Config
    with NUM_CLASSES = 1 + num_classes and, for inference, its published defaults: 512 x 512 square input,
    detection_threshold 0.7, RPN (32..512) x (0.5, 1, 2), 6000 -> NMS 0.7 -> 1000 proposals, 100 detections at NMS 0.3,
    7 / 14 / 28 pooling and mask sizes, 256-channel pyramid, 1024-wide classifier head, MEAN_PIXEL (123.7, 116.8, 103.9)."""
    import inspect
    from rope_s3d_amd import maskrcnn as M
    from rope_s3d_amd.constants import NUM_RENDER_LINKS
    from rope_s3d_amd.urdf import URDFReader
    links = URDFReader().mesh_names[:6]                                   # predict.py:88-91: classes = ["BG"] + these
    assert len(links) == NUM_RENDER_LINKS == 6
    sig = inspect.signature(M.MaskRCNN.__init__).parameters
    assert sig['num_classes'].default == 1 + len(links) == 7             # NUM_CLASSES = 1 + num_classes (background)
    assert sig['image_size'].default == 512 and sig['min_confidence'].default == 0.7
    assert inspect.signature(M.MaskRCNNSegmenter.__init__).parameters['num_classes'].default == 7
    net = M.MaskRCNN()
    # ResNet-101: 3 + 4 + 23 + 3 bottleneck blocks (train.py:49 network_backbone="resnet101")
    blocks = [len(list(stage.children())) for stage in net.backbone.stages]
    assert blocks == [3, 4, 23, 3]
    assert net.cls.out_features == 7 and net.box.out_features == 7 * 4 and list(net.mask.children())[-1].out_channels == 7
    assert (M.RPN_ANCHOR_SCALES, M.RPN_ANCHOR_RATIOS) == ((32, 64, 128, 256, 512), (0.5, 1.0, 2.0))
    assert (M.PRE_NMS_LIMIT, M.POST_NMS_ROIS, M.RPN_NMS_THRESHOLD) == (6000, 1000, 0.7)
    assert (M.DETECTION_MAX_INSTANCES, M.DETECTION_NMS_THRESHOLD) == (100, 0.3)
    assert (M.POOL_SIZE, M.MASK_POOL_SIZE, M.MASK_SHAPE) == (7, 14, 28)
    assert (M.TOP_DOWN_PYRAMID_SIZE, M.FPN_CLASSIF_FC) == (256, 1024)
    assert M.MEAN_PIXEL == (123.7, 116.8, 103.9)
    assert M.RPN_BBOX_STD_DEV == M.BBOX_STD_DEV == (0.1, 0.1, 0.2, 0.2)


def _clustered_boxes(rng, B, N):
    """Boxes around a few centres, so that many overlap: (B, N, 4) float32 in [0, 1], and scores with ties."""
    c = rng.uniform(0.2, 0.8, (B, 12, 2))
    pick = rng.integers(0, 12, (B, N))
    cy, cx = (np.take_along_axis(c[..., k], pick, 1) + rng.normal(0, 0.03, (B, N)) for k in (0, 1))
    h, w = rng.uniform(0.02, 0.3, (B, N)), rng.uniform(0.02, 0.3, (B, N))
    boxes = np.clip(np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1), 0, 1).astype(np.float32)
    scores = np.round(rng.uniform(0, 1, (B, N)), 3).astype(np.float32)          # rounded: equal scores occur
    return torch.from_numpy(boxes), torch.from_numpy(scores)


@pytest.mark.gpu
def test_hip_nms_equals_the_tensor_formulation(monkeypatch):
    """rope_seg_nms (librope_hip.so) keeps exactly the boxes the tensor fixed-point iteration keeps: several sets at once, padding,
    per-class groups, limits that cut the walk short, sizes on both sides of a 64-box word."""
    from rope_s3d_amd.maskrcnn import _nms_batched
    rng = np.random.default_rng(11)
    for B, N, thr, limit, with_valid, with_groups in ((3, 3000, 0.7, 1000, False, False), (2, 6000, 0.7, 1000, False, False),
                                                       (1, 2500, 0.3, 2500, False, True), (4, 777, 0.5, 50, True, True),
                                                       (2, 64, 0.3, 64, True, False), (1, 65, 0.7, 3, False, False), (1, 1, 0.7, 1, False, False)):
        boxes, scores = _clustered_boxes(rng, B, N)
        boxes, scores = boxes.cuda(), scores.cuda()
        valid = torch.from_numpy(rng.uniform(size=(B, N)) < 0.8).cuda() if with_valid else None
        groups = torch.from_numpy(rng.integers(0, 7, (B, N))).cuda() if with_groups else None
        monkeypatch.setenv('ROPE_SEG_HIP', '0')
        ref = _nms_batched(boxes, scores, thr, limit, valid=valid, groups=groups)
        monkeypatch.setenv('ROPE_SEG_HIP', '1')
        got = _nms_batched(boxes, scores, thr, limit, valid=valid, groups=groups)
        assert torch.equal(ref, got), (B, N, thr, limit)
        assert int(got.sum(1).max()) <= limit and (valid is None or not bool((got & ~valid).any()))


@pytest.mark.gpu
def test_hip_roi_align_equals_the_tensor_formulation(monkeypatch):
    """rope_seg_roi_align: same pyramid level, same four taps, same bfloat16 roundings — bit for bit, for boxes inside, on the
    edge of and beyond the map, of every pyramid level, both pool sizes."""
    from rope_s3d_amd.maskrcnn import _pack_levels
    rng = np.random.default_rng(5)
    B, C = 3, 256
    feats = [torch.from_numpy(rng.normal(0, 1, (B, C, s, s)).astype(np.float32)).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
             for s in (128, 64, 32, 16)]
    K = 4000
    cy, cx = rng.uniform(-0.05, 1.05, K), rng.uniform(-0.05, 1.05, K)
    side = np.exp(rng.uniform(np.log(0.004), np.log(1.2), K))                    # sqrt(area) from below P2's to above P5's range
    ar = np.exp(rng.uniform(-1, 1, K))
    h, w = side * ar, side / ar
    boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], 1).astype(np.float32)
    boxes[:50] = np.clip(boxes[:50], 0, 1)
    boxes[50:60, 2:] = boxes[50:60, :2]                                          # empty boxes
    boxes[60:70] = [0, 0, 1, 1]
    boxes = torch.from_numpy(boxes).cuda()
    frame = torch.from_numpy(rng.integers(0, B, K)).cuda()
    packed = _pack_levels(feats)
    for pool in (7, 14):
        monkeypatch.setenv('ROPE_SEG_HIP', '0')
        ref = _roi_align(feats, boxes, pool, 512, frame, packed)
        monkeypatch.setenv('ROPE_SEG_HIP', '1')
        got = _roi_align(feats, boxes, pool, 512, frame, packed)
        assert got.shape == ref.shape == (K, C, pool, pool)
        same = (ref.contiguous().view(torch.int16) == got.contiguous().view(torch.int16)).flatten(1).all(1)
        assert bool(same.all()), f"pool {pool}: {int((~same).sum())} of {K} boxes differ, first {int((~same).nonzero()[0])}"


@pytest.mark.gpu
def test_detections_do_not_depend_on_which_box_kernels_run(monkeypatch):
    """Everything after the convolutional trunk (proposals, RoIAlign, heads, detection layer, masks) on the same trunk outputs of a
    batch, with the HIP box kernels and with the tensor formulation: same classes, same scores, same masks.  (The trunk's own
    library convolutions are not bit-reproducible from run to run, so the comparison starts behind it.)"""
    import torch.nn.functional as F
    from rope_s3d_amd import maskrcnn as M
    from rope_s3d_amd.maskrcnn import MEAN_PIXEL
    # no padding of the heads' row counts here: a padded count is another convolution shape, and the library's kernel for it
    # may add its partial sums in a different order every run (seen with 8192 rows) — the comparison needs a repeatable head
    monkeypatch.setattr(M, 'HEAD_ROW_STEP', 1)
    monkeypatch.setattr(M, 'MASK_ROW_STEP', 1)
    net = MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0).net
    H, W = 90, 160
    images = [torch.from_numpy(np.random.default_rng(i).integers(0, 255, (H, W, 3), dtype=np.uint8)).cuda() for i in range(4)]
    scale = net.size / max(H, W)
    nh, nw = round(H * scale), round(W * scale)
    top, left = (net.size - nh) // 2, (net.size - nw) // 2
    x = F.interpolate(torch.stack([im.permute(2, 0, 1) for im in images]).float(), (nh, nw), mode='bilinear', align_corners=False)
    x = F.pad(x - torch.tensor(MEAN_PIXEL, device='cuda').view(1, 3, 1, 1), (left, net.size - nw - left, top, net.size - nh - top))
    with torch.no_grad():
        feats, probs, deltas = net._trunk(x.to(torch.bfloat16).contiguous())
        out = {}
        for flag in ('0', '1', '0'):
            monkeypatch.setenv('ROPE_SEG_HIP', flag)
            out.setdefault(flag, []).append(net._detect(feats, probs, deltas, H, W, scale, top, left, nh, nw))
    for other in (out['0'][1], out['1'][0]):
        for a, b in zip(out['0'][0], other):
            assert len(a[0]) > 0 and all(torch.equal(p, q) for p, q in zip(a, b))


@pytest.mark.gpu
def test_pipelined_batches_give_the_batch_by_batch_results(monkeypatch):
    """MaskRCNNSegmenter.batches overlaps the next group's trunk (second stream) with the current group's box steps.  With the
    trunk's outputs fixed per input (its library convolutions are not bit-reproducible run to run) every group's classes,
    scores and masks equal those of `batch` on the group alone — whichever stream produced and copied the trunk outputs."""
    from rope_s3d_amd import maskrcnn as M
    monkeypatch.setattr(M, 'HEAD_ROW_STEP', 1)          # see test_detections_do_not_depend_on_which_box_kernels_run
    monkeypatch.setattr(M, 'MASK_ROW_STEP', 1)
    seg = MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
    net = seg.net
    groups = [[np.random.default_rng(10 * g + i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(4)] for g in range(4)]
    real, cache = net._trunk_replayed, {}

    def fixed_trunk(x):
        key = float(x.float().sum())
        if key not in cache:
            feats, probs, deltas = real(x)
            cache[key] = ([f.clone() for f in feats], probs.clone(), deltas.clone())
        return cache[key]
    net._trunk_replayed = fixed_trunk
    try:
        alone = [seg.batch(g) for g in groups]
        piped = list(seg.batches(groups))
        assert len(cache) == len(groups)
    finally:
        net._trunk_replayed = real
    assert len(piped) == len(groups)
    for a, b in zip(alone, piped):
        assert len(a) == len(b) == 4
        for ra, rb_ in zip(a, b):
            assert len(ra['class_ids']) > 0 and all(np.array_equal(ra[k], rb_[k]) for k in ra)
    # and with the real trunk: same structure (the values are those of a different run of the convolutions)
    for res in seg.batches(groups[:2]):
        assert len(res) == 4 and all(r['masks'].shape[:2] == (90, 160) for r in res)


def test_heads_run_on_row_counts_in_steps():
    """_pad_rows: the classifier and mask heads see multiples of a step (a new row count is a new convolution shape for the
    library); padding rows are empty boxes whose results are dropped — same detections as without padding (CPU float32)."""
    from rope_s3d_amd import maskrcnn as M
    assert M._pad_rows(torch.ones(7, 4), 5).shape == (10, 4) and M._pad_rows(torch.ones(10, 4), 5).shape == (10, 4)
    assert M._pad_rows(torch.ones(0, 4), 5).shape == (5, 4) and float(M._pad_rows(torch.ones(7, 4), 5)[7:].abs().sum()) == 0
    seg = MaskRCNNSegmenter(7, device='cpu', seed=0, min_confidence=0.0)
    frame = np.random.default_rng(2).integers(0, 255, (45, 80, 3), dtype=np.uint8)
    padded = seg(frame)
    old = M.HEAD_ROW_STEP, M.MASK_ROW_STEP
    M.HEAD_ROW_STEP, M.MASK_ROW_STEP = 1, 1
    try:
        plain = seg(frame)
    finally:
        M.HEAD_ROW_STEP, M.MASK_ROW_STEP = old
    assert np.array_equal(padded['class_ids'], plain['class_ids']) and np.allclose(padded['scores'], plain['scores'], atol=1e-5)
    assert (padded['masks'] != plain['masks']).mean() < 1e-3


@pytest.mark.gpu
def test_hip_bias_act_equals_the_separate_tensor_operations(monkeypatch):
    """rope_seg_bias_act after a convolution: bias, residual and ReLU in one pass, bit-equal to conv(x) + b (+ res) -> relu as
    separate bfloat16 operations — both memory layouts, with and without residual / ReLU, on the same convolution output."""
    from rope_s3d_amd import maskrcnn as M
    torch.manual_seed(3)
    for cl in (False, True):
        conv = torch.nn.Conv2d(64, 128, 3, 1, 1).cuda().to(torch.bfloat16)
        x = torch.randn(2, 64, 24, 40, device='cuda').to(torch.bfloat16)
        if cl:
            conv, x = conv.to(memory_format=torch.channels_last), x.contiguous(memory_format=torch.channels_last)
        raw = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding)
        res = torch.randn_like(raw)
        real_conv2d = torch.nn.functional.conv2d
        monkeypatch.setattr(M.F, 'conv2d', lambda *a, **k: raw.clone())          # the same convolution output for both sides
        try:
            for r, relu in ((None, True), (res, True), (res, False), (None, False)):
                want = raw + conv.bias.view(1, -1, 1, 1)
                if r is not None:
                    want = want + r
                if relu:
                    want = torch.relu(want)
                got = M._conv_act(conv, x, res=r, relu=relu)
                assert got.stride() == want.stride() and torch.equal(got.view(torch.int16), want.view(torch.int16)), (cl, r is not None, relu)
        finally:
            monkeypatch.setattr(M.F, 'conv2d', real_conv2d)


def _stage_against_oracle(device: str, n_frames: int, size=(90, 160), seed: int = 5):
    """Every decision of the post-trunk stage (MaskRCNN._detect) against oracle/maskrcnn_ref.py, the numpy restatement of Matterport's
    published steps, one box at a time.  Each step of the oracle is fed what the product fed ITS step (so that a last-digit
    difference in exp() cannot cascade), and compared: values to float32 round-off, decisions — which boxes survive, in which order,
    with which class, which pixels the integer masks hold — exactly."""
    from oracle import maskrcnn_ref as ref
    from rope_s3d_amd.maskrcnn import DETECTION_MAX_INSTANCES, MaskRCNNSegmenter, _roi_align
    seg = MaskRCNNSegmenter(7, device=device, seed=seed, min_confidence=0.0)
    net = seg.net
    net.keep_trace = True
    net.min_conf = 0.142                              # random weights put the class scores just above 1/7: a threshold that really filters
    rng = np.random.default_rng(seed)
    frames = [rng.integers(0, 255, size + (3,), dtype=np.uint8) for _ in range(n_frames)]
    out = seg.batch(frames)

    def host(v):
        if torch.is_tensor(v):
            return (v.detach().float() if v.dtype.is_floating_point else v.detach()).cpu().numpy()
        return v
    t = {k: host(v) for k, v in net.trace.items() if k != 'feats'}
    feats = [host(x) for x in net.trace['feats'][:4]]                                       # (B, C, h, w) each
    H, W, scale, top, left, nh, nw = t['geometry']
    anchors, sel, ok = t['anchors'], [int(i) for i in t['sel']], t['ok']
    nwin = (np.array([top, left, top + nh, left + nw], np.float32) - np.array([0, 0, 1, 1], np.float32)) / np.float32(net.size - 1)
    checked = dict(proposals=0, detections=0, mask_pixels=0, near=0)
    roi_at = det_at = 0
    for f in range(n_frames):
        # ---- ProposalLayer: decode + clip of the top anchors (values), then NMS on the product's boxes (decisions)
        top_idx, top_p = t['top_idx'][f], t['top_p'][f]
        order = sorted(range(len(t['probs'][f])), key=lambda i: (-float(t['probs'][f][i]), i))[:len(top_idx)]
        assert np.array_equal(np.sort(np.asarray(order)), np.sort(top_idx))              # the same anchors enter
        for j in rng.choice(len(top_idx), 200, replace=False):
            want = ref.clip_box(ref.apply_box_delta(anchors[top_idx[j]], t['deltas'][f][top_idx[j]] * ref.RPN_BBOX_STD_DEV), (0, 0, 1, 1))
            assert np.allclose(t['decoded'][f][j], want, rtol=0, atol=2e-6), (f, j)
        keep_ref = ref.non_max_suppression(t['decoded'][f], top_p, ref.POST_NMS_ROIS, ref.RPN_NMS_THRESHOLD)
        keep_got = np.flatnonzero(t['keep'][f])
        assert list(keep_got) == sorted(keep_ref) and len(keep_ref) > 50, f                # the same proposals, by anchor rank
        n_roi = len(keep_ref)
        rois = t['rois'][roi_at:roi_at + n_roi]
        assert np.array_equal(rois, t['decoded'][f][keep_got]) and (t['roi_frame'][roi_at:roi_at + n_roi] == f).all()
        checked['proposals'] += n_roi
        # ---- refine_detections_graph on the product's head outputs for these proposals
        keep, cls, sc, boxes = ref.refine_detections(rois, t['cls_prob'][roi_at:roi_at + n_roi], t['box_delta'][roi_at:roi_at + n_roi], nwin,
                                                     net.min_conf, DETECTION_MAX_INSTANCES)
        sel_f = [i - roi_at for i in sel if roi_at <= i < roi_at + n_roi]
        assert sel_f == keep, (f, sel_f[:10], keep[:10])                                  # the same detections in the same order
        n_det = len(keep)
        assert np.allclose(t['det_boxes'][det_at:det_at + n_det], boxes, rtol=0, atol=2e-6)
        # ---- unmold_detections: pixel boxes and integer masks
        rows = list(range(det_at, det_at + n_det))                                        # rows of this frame among all detections
        kept_rows = list(np.flatnonzero(ok))                                              # ... and the product's rows after the zero-area filter
        m28 = [t['mask28'][kept_rows.index(r)] if ok[r] else np.zeros((28, 28), np.float32) for r in rows]
        bx, masks, kept = ref.unmold_detections([t['det_boxes'][r] for r in rows], m28, (H, W), (net.size, net.size),
                                                [top, left, top + nh, left + nw])
        assert [rows[k] for k in kept] == [r for r in rows if ok[r]]                       # the same zero-area boxes are dropped
        got_cls, got_masks = out[f]['class_ids'], out[f]['masks']
        assert list(got_cls) == [cls[k] for k in kept] and got_masks.shape == (H, W, len(kept))
        for k, (full, vals, box) in enumerate(masks):
            assert np.array_equal(t['px'][kept_rows.index(rows[kept[k]])], box), (f, k)
            # float32 bilinear weights against float64 ones: a resized value within 1e-5 of the threshold may fall either side
            near = np.zeros((H, W), bool)
            y1, x1, y2, x2 = max(box[0], 0), max(box[1], 0), min(box[2], H), min(box[3], W)
            near[y1:y2, x1:x2] = (np.abs(vals - 0.5) < 1e-5)[y1 - box[0]:y2 - box[0], x1 - box[1]:x2 - box[1]]
            assert np.array_equal(got_masks[..., k][~near], full[~near]), (f, k, int((got_masks[..., k] != full).sum()))
            checked['mask_pixels'] += int(full.sum())
            checked['near'] += int(near.sum())
        checked['detections'] += n_det
        det_at += n_det
        roi_at += n_roi
    # (random weights leave the mask logits near 0, i.e. the values near the 0.5 threshold: a per cent of the pixels are too close to call)
    assert checked['detections'] >= 2 * n_frames and checked['mask_pixels'] > 1000 and checked['near'] < 0.05 * checked['mask_pixels'], checked
    # ---- PyramidROIAlign on a few proposals of frame 0: level rule exact, samples to the features' precision
    got = _roi_align(net.trace['feats'], torch.from_numpy(t['rois'][:12].copy()).to(device), 7, net.size).float().cpu().numpy()      # (12, C, 7, 7)
    tol = 3e-2 if device != 'cpu' else 1e-5                                                # bfloat16 features on the GPU
    for k in range(12):
        lv = ref.roi_level(t['rois'][k], net.size)
        want = ref.crop_and_resize(feats[lv - 2][0].transpose(1, 2, 0).astype(np.float64), t['rois'][k], 7).transpose(2, 0, 1)
        assert np.allclose(got[k], want, rtol=tol, atol=tol * max(1.0, float(np.abs(want).max()))), (k, lv)
    return checked


def test_post_trunk_stage_against_the_matterport_restatement_cpu():
    _stage_against_oracle('cpu', 1)


@pytest.mark.gpu
@pytest.mark.parametrize('n_frames', [1, 8])
def test_post_trunk_stage_against_the_matterport_restatement_gpu(n_frames):
    """The same with the HIP kernels of csrc/rope_seg.hip in the stage (NMS, RoIAlign, bias/activation), batches of 1 and 8."""
    _stage_against_oracle('cuda:0', n_frames)
