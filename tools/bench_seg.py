#!/usr/bin/env python3
"""Throughput of the segmentation stage (Mask R-CNN, ResNet-101-FPN, 512x512 input, bf16 autocast, random weights)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd.maskrcnn import MaskRCNNSegmenter

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seg = MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
for shape in ((90, 160, 3), (480, 640, 3)):
    img = np.random.default_rng(0).integers(0, 255, shape, dtype=np.uint8)
    for _ in range(3):
        seg(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = seg(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"input {shape[1]}x{shape[0]}: {dt * 1e3:.1f} ms/frame = {1 / dt:.1f} frames/s ({len(out['class_ids'])} detections)")
# backbone + FPN + RPN convolutions alone (the dense contraction part)
x = torch.randn(1, 3, 512, 512, device='cuda').contiguous(memory_format=torch.channels_last)
with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
    for _ in range(3):
        f = seg.net.fpn(seg.net.backbone(x))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f = seg.net.fpn(seg.net.backbone(x)); [seg.net.rpn(p) for p in f]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
flops = 2 * 57.0e9          # ~57 GMAC for ResNet-101-FPN + RPN at 512x512 (counted from the layer shapes)
print(f"backbone+FPN+RPN 512x512 bf16: {dt * 1e3:.2f} ms = {flops / dt / 1e12:.1f} TFLOP/s")
