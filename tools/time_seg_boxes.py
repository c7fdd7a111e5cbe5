#!/usr/bin/env python3
"""The segmentation stage's box steps on the GPU, tensor formulation against the HIP kernels of rope_seg.hip, interleaved on one
box: proposal NMS (8 sets of 6000), detection NMS (one set of 8000 in 56 groups), RoIAlign 7x7 of 8000 boxes and 14x14 of 800,
and the whole stage on a batch of 8 frames."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd import maskrcnn as M

rng = np.random.default_rng(3)


def boxes_like(B, N):
    c = rng.uniform(0.2, 0.8, (B, 40, 2)); pick = rng.integers(0, 40, (B, N))
    cy, cx = (np.take_along_axis(c[..., k], pick, 1) + rng.normal(0, 0.05, (B, N)) for k in (0, 1))
    h, w = rng.uniform(0.02, 0.4, (B, N)), rng.uniform(0.02, 0.4, (B, N))
    b = np.clip(np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1), 0, 1).astype(np.float32)
    return torch.from_numpy(b).cuda(), torch.from_numpy(rng.uniform(0, 1, (B, N)).astype(np.float32)).cuda()


def timeit(f, reps=20):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3


pb, ps = boxes_like(8, 6000)
db, dsc = boxes_like(1, 8000)
dg = torch.from_numpy(rng.integers(0, 56, (1, 8000))).cuda()
feats = [torch.randn(8, 256, s, s, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for s in (128, 64, 32, 16)]
packed = M._pack_levels(feats)
rb = boxes_like(1, 8000)[0][0]; rf = torch.from_numpy(np.repeat(np.arange(8), 1000)).cuda()
mb = rb[:800].contiguous(); mf = rf[::10].contiguous()
seg = M.MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
frames = [np.random.default_rng(i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(8)]
cases = {
    'proposal NMS 8 x 6000 -> 1000': lambda: M._nms_batched(pb, ps, 0.7, 1000),
    'detection NMS 8000, 56 groups': lambda: M._nms_batched(db, dsc, 0.3, 8000, groups=dg),
    'RoIAlign 7x7, 8000 boxes': lambda: M._roi_align(feats, rb, 7, 512, rf, packed),
    'RoIAlign 14x14, 800 boxes': lambda: M._roi_align(feats, mb, 14, 512, mf, packed),
    'whole stage, batch of 8': lambda: seg.batch(frames),
}
for name, f in cases.items():
    res = {'0': [], '1': []}
    for rep in range(3):
        for flag in ('0', '1'):
            os.environ['ROPE_SEG_HIP'] = flag
            res[flag].append(timeit(f, 10 if 'whole' in name else 20))
    print(f"{name:32s} tensor ops {min(res['0']):7.2f} ms   HIP kernels {min(res['1']):7.2f} ms   (best of 3; all: {np.round(res['0'], 2)} / {np.round(res['1'], 2)})")
