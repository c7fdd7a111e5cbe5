#!/usr/bin/env python3
"""BASELINE configs[2] shape: synthetic RGB-D set -> Mask R-CNN stage (PyTorch-ROCm) -> HIP engine -> stage machine.

    python tools/bench_pipeline.py [n_frames]

Random network weights (none exist offline): detections and therefore predictions are meaningless, the frame rate
of the whole pipeline is what this measures.  min_confidence 0 keeps all 100 detections per frame (worst case)."""
import argparse, os, sys, tempfile, time
os.environ['ROPE_TIMING'] = '1'      # predict_dataset prints set-up and frame time apart
import numpy as np
import torch
torch.cuda.init()        # torch's bundled HIP runtime has to come up before librope_hip.so brings in the system one
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
import predict_dataset as pd
from rope_s3d_amd.data.dataset import make_synthetic_dataset

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with tempfile.TemporaryDirectory() as tmp:
    d = make_synthetic_dataset(os.path.join(tmp, 'pipe'), n, base_intrin='1280_720_color', seed=7919)
    os.chdir(tmp)
    for seg in (None, 'color', 'maskrcnn'):
        args = argparse.Namespace(dataset=d, angs='SLU', ds_factor=8, segmenter=seg, weights=None, lookup_divisions=None, predictors=int(os.environ.get('ROPE_PREDICTORS', '1')), batch=None)
        pd.run(argparse.Namespace(**{**vars(args)}))                 # warm-up incl. construction
        t0 = time.perf_counter()
        out = pd.run(args)
        dt = time.perf_counter() - t0
        print(f"segmenter={seg}: {n} frames in {dt:.2f} s = {n / dt:.1f} frames/s (includes Predictor construction and lookup-table build)")
