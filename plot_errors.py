#!/usr/bin/env python3
"""Error tables of a saved prediction run (same command line as the reference's plot_errors.py:30-67; text only).

    python plot_errors.py predictions_<dataset>.npy [-angs SLU]      # or a synth.py result (2, N, 6)
"""
import argparse
import os
import re

import numpy as np

from robotpose import Dataset, Grapher
from rope_s3d_amd.prediction.analysis import JointDistance


def run(args):
    file = args.file if args.file.endswith('.npy') else args.file + '.npy'
    results = np.load(file)
    if results.shape[0] == 2 and results.ndim == 3:          # [actual, predicted] as SyntheticPredictor saves it
        angles, preds = results[0], results[1]
    else:
        name = re.search(r'predictions_(.+)\.npy$', os.path.basename(file))
        ds = Dataset(args.dataset or name.group(1))
        preds, angles = results, np.copy(ds.angles)
    order = np.argsort(angles[..., 0])                       # the reference sorts by S whatever -sort_by says (plot_errors.py:55)
    Grapher(args.angs, preds[order], angles[order]).plot(20)
    JointDistance().plot(preds[order], angles[order], .25)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument('file', type=str, help="The file to view.")
    parser.add_argument('-sort_by', type=str, default='S', help="Joint to sort by.")
    parser.add_argument('-angs', type=str, default='SLU', help="The joints to predict.")
    parser.add_argument('-dataset', type=str, default=None, help="Dataset directory holding the true angles (default: from the file name).")
    run(parser.parse_args())
