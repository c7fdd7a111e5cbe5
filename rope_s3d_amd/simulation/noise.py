"""Depth-hole noise for synthetic frames (reference: robotpose/simulation/noise.py:7-31)."""
import numpy as np

from ..imgproc import dilate, erode


class NoiseMaker:

    def __init__(self, rng: np.random.Generator = None):
        self.rng = rng if rng is not None else np.random.default_rng()

    def holes(self, arr, max_size=25, std=0.22, thresh_factor=1, connection_factor=20):
        """Union of thresholded |N(0,std)| fields dilated by 3,6,..; closed with a 20x20 box; zeroes depth there."""
        shape = arr.shape
        holes = np.zeros(shape)
        for dilation in np.arange(3, max_size, 3):
            thresh = -thresh_factor / dilation + 1
            noise = np.abs(self.rng.normal(0, std, shape))
            noise = np.clip(noise, 0, 1)
            noise[noise < thresh] = 0
            holes += dilate(noise, int(dilation))
        holes[holes != 0] = 1
        holes = erode(dilate(holes, connection_factor), connection_factor)
        return arr * (holes == 0).astype(float)
