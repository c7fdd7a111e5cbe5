import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from rope_s3d_amd.maskrcnn import MaskRCNNSegmenter
seg = MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
frames = [np.random.default_rng(i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(8)]
for _ in range(5): seg.batch(frames)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): seg.batch(frames)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
st.sort_stats('cumtime').print_stats(18)
