import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from rope_s3d_amd.maskrcnn import MaskRCNNSegmenter
import torch.nn.functional as F
from rope_s3d_amd.maskrcnn import MEAN_PIXEL
seg = MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
net = seg.net
frames = [np.random.default_rng(i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(4)]
images = [torch.from_numpy(np.ascontiguousarray(f[..., ::-1])).to('cuda:0') for f in frames]
H, W = 90, 160
scale = net.size / max(H, W); nh, nw = round(H * scale), round(W * scale); top, left = (net.size - nh) // 2, (net.size - nw) // 2
x = torch.stack([im.permute(2, 0, 1) for im in images]).float()
x = F.interpolate(x, (nh, nw), mode='bilinear', align_corners=False)
x = x - torch.tensor(MEAN_PIXEL, device='cuda:0').view(1, 3, 1, 1)
x = F.pad(x, (left, net.size - nw - left, top, net.size - nh - top))
with torch.no_grad():
    feats, probs, deltas = net._trunk(x.to(torch.bfloat16).contiguous())
    feats = [f.clone() for f in feats]; probs = probs.clone(); deltas = deltas.clone()
    def run(flag):
        os.environ['ROPE_SEG_HIP'] = flag
        return net._detect(feats, probs, deltas, H, W, scale, top, left, nh, nw)
    def same(a, b):
        return [all(torch.equal(p, q) for p, q in zip(x_, y_)) for x_, y_ in zip(a, b)]
    r1, r2, g1, g2 = run('0'), run('0'), run('1'), run('1')
    import rope_s3d_amd.maskrcnn as M
    M.HEAD_ROW_STEP = 1; M.MASK_ROW_STEP = 1
    u1, u2, v1 = run('0'), run('0'), run('1')
    print('unpadded: ref vs ref', same(u1, u2), 'ref vs hip', same(u1, v1), 'padded vs unpadded', same(r1, u1))
print('ref vs ref', same(r1, r2)); print('hip vs hip', same(g1, g2)); print('ref vs hip', same(r1, g1))
