"""Several Predictors on one GPU, frames dealt round-robin to worker threads.

A frame is ~25 short dependent device batches with host decisions between them, so one Predictor leaves the GPU idle
a good part of the time (launch latencies, host preparation, result copies).  Every Predictor owns its engine context
and HIP stream, the library and numpy both release the interpreter lock, and frames are independent
(Predictor.run starts from a fresh state, predict.py:144-148) — so k Predictors fed by k threads overlap one frame's
gaps with another frame's kernels.  Results are the single Predictor's, frame by frame, in the input order."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .predict import Predictor


class PredictorPool:

    def __init__(self, n: int, *args, **kwargs):
        """n Predictors with the same arguments (Predictor.__init__).  A `segmenter` is shared by all of them and must
        therefore tolerate calls from several threads."""
        assert n >= 1
        with ThreadPoolExecutor(max_workers=n) as pool:             # crops and lookup tables of the k contexts build side by side
            self.predictors = [f.result() for f in [pool.submit(Predictor, *args, **kwargs) for _ in range(n)]]
        self._tune()

    THROUGHPUT_FROM = 10         # Predictors on one GPU from which the device's throughput, not one chain's latency, is the limit

    def _tune(self):
        """Small batches work out their forward kinematics and screen boxes inside the raster workgroups by default: one launch
        fewer per evaluation, which shortens a single Predictor's chain (464 against 432 frames/s) but repeats the matrices in every
        workgroup.  With many Predictors keeping the GPU full that repetition costs more than the launch (1 194 against 1 273
        frames/s with twelve): they get the separate launch back.  Same results either way (rope_set_strategy)."""
        if len(self.predictors) >= self.THROUGHPUT_FROM:
            for p in self.predictors:
                e = p.renderer.engine
                e.set_strategy(e.SEPARATE_GEOMETRY)

    def __len__(self):
        return len(self.predictors)

    @property
    def evaluations(self) -> int:
        return sum(p.evaluations for p in self.predictors)

    def run_many(self, target_colors, target_depths, camera_poses=None) -> np.ndarray:
        n, k = len(target_colors), len(self.predictors)
        out = np.zeros((n, 6))
        if n == 0:
            return out

        def work(w):
            p = self.predictors[w]
            for i in range(w, n, k):
                out[i] = p.run(target_colors[i], target_depths[i], None if camera_poses is None else camera_poses[i])
        if k == 1 or n == 1:
            work(0) if k == 1 else [work(w) for w in range(k)]
            return out
        with ThreadPoolExecutor(max_workers=k) as pool:
            for f in [pool.submit(work, w) for w in range(k)]:
                f.result()
        return out
