#!/usr/bin/env python3
"""Synthetic render -> predict -> compare batches (same command line as the reference's synth.py:29-41)."""
import argparse

from robotpose import Dataset, SyntheticPredictor
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE


def run(args):
    pose = DEFAULT_CAMERA_POSE
    if args.dataset not in (None, 'none'):
        pose = Dataset(args.dataset).camera_pose[0]
    synth = SyntheticPredictor(pose, args.intrinsics, args.ds_factor, args.angs, noise=args.noise)
    res = synth.run_batch(args.num, args.file)
    from rope_s3d_amd.prediction.analysis import Grapher
    Grapher(args.angs, res[1], res[0]).plot()


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument('dataset', type=str, nargs='?', default='none', help="The dataset whose camera pose to use ('none' = default pose).")
    parser.add_argument('-num', type=int, default=2500, help="Number of synthetic poses to predict.")
    parser.add_argument('-file', type=str, default='synth_test', help="File to save results to.")
    parser.add_argument('-noise', action="store_true", help="Adds semi-realistic noise to depth images.")
    parser.add_argument('-ds_factor', type=int, default=8, choices=[1, 2, 4, 6, 8, 10, 12], help="Downsampling factor.")
    parser.add_argument('-angs', type=str, default='SLU', help="The joints to predict.")
    parser.add_argument('-intrinsics', type=str, default='1280_720_color', help="Base camera instrinsics to use.")
    run(parser.parse_args())
