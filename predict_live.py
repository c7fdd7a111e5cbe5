#!/usr/bin/env python3
"""Live deviation monitor: predicted joint angles against what the robot controller claims (reference:
predict_live.py:94-185).  Every frame the controller's pose is read, a colour + depth frame is taken, the
Predictor runs, and the Cartesian distance between the tool positions of the claimed and the predicted pose is
tracked; the state goes "out of range" when LENGTH consecutive frames deviate by more than ALLOWED_DEVIANCE metres.
Claims and predictions are saved to live_preds.npy as (2, n, 6) after every frame, as the reference does.

    python predict_live.py                       # RealSense + the controller's JSON file (needs both)
    python predict_live.py -replay <dataset>     # a recorded Dataset as camera, its angles as the controller's claims

No windows are opened (cv2.imshow in the reference): the state is printed when it changes."""
import argparse
import logging

import numpy as np

from robotpose import Dataset, Intrinsics, Predictor
from rope_s3d_amd.prediction.analysis import JointDistance
from rope_s3d_amd.prediction.feed import DatasetCamera, JSONCoupling, LiveCamera

LENGTH = 3
ALLOWED_DEVIANCE = 0.1


class Live:

    def __init__(self, base_intrin_str, parent_ds, angs, ds_factor, camera=None, link=None, save_to='live_preds.npy', **predictor_kwargs):
        base_intrin = Intrinsics(base_intrin_str)
        ds = parent_ds if isinstance(parent_ds, Dataset) else Dataset(parent_ds)
        self.cam = camera if camera is not None else LiveCamera(base_intrin.width, base_intrin.height)
        self.link = link if link is not None else JSONCoupling()
        self.pred = Predictor(ds.camera_pose[0], ds_factor, False, None, angs, base_intrin=base_intrin_str,
                              model_ds=getattr(ds, 'name', str(parent_ds)), **predictor_kwargs)
        self.jd = JointDistance()
        self.cam.start()
        self.claims = np.zeros((LENGTH, 6))
        self.predictions = np.zeros((LENGTH, 6))
        self.running_claims, self.running_predictions = [], []
        self.save_to = save_to
        self.out_of_range = np.zeros(LENGTH, bool)
        self._last_state = None

    def stop(self):
        self.cam.stop()

    def step(self) -> bool:
        """One pass of the reference's loop body; False when a source has run dry (replay)."""
        claimed = self.link.get_pose()
        frame = self.cam.get()
        if claimed is None or frame is None:
            return False
        color, depth = frame
        calculated = self.pred.run(color, depth)
        self.link.reset()
        self.shift_in(claimed, calculated)
        self.update_error()
        self.displayState()
        self.save()
        return True

    def run(self, max_frames: int = None):
        logging.info("Ready")
        n = 0
        while (max_frames is None or n < max_frames) and self.step():
            n += 1
        return n

    def shift_in(self, claim, prediction):
        self.claims[1:] = self.claims[:-1]
        self.predictions[1:] = self.predictions[:-1]
        self.claims[0] = claim
        self.predictions[0] = prediction
        self.running_claims.append(claim)
        self.running_predictions.append(prediction)

    def update_error(self):
        self.diff = self.jd.single(self.predictions, self.claims)       # tool-frame distance, metres, newest first
        self.out_of_range = self.diff > ALLOWED_DEVIANCE

    def save(self):
        if self.save_to:
            np.save(self.save_to, np.array([self.running_claims, self.running_predictions]))

    @property
    def state(self) -> bool:
        return bool(np.sum(self.out_of_range, 0) == LENGTH)

    def displayState(self):
        if self.state != self._last_state:
            self._last_state = self.state
            print(f"frame {len(self.running_claims)}: {'OUT OF RANGE' if self.state else 'in range'} "
                  f"(tool deviation {float(np.ravel(self.diff)[0]) * 1000:.1f} mm)")


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument('-replay', type=str, default=None, help="replay this recorded dataset instead of a live camera")
    parser.add_argument('-dataset', type=str, default='set91', help="parent dataset: camera pose (and segmentation model) come from it")
    parser.add_argument('-intrin', type=str, default='1280_720_color')
    parser.add_argument('-angs', type=str, default='SLU')
    parser.add_argument('-ds_factor', type=int, default=8)
    parser.add_argument('-frames', type=int, default=None)
    args = parser.parse_args()
    if args.replay:
        ds = Dataset(args.replay)
        cam = DatasetCamera(ds)
        kw = {'color_dict': ds.attrs['color_dict']} if ds.attrs.get('synthetic') else {}
        a = Live(str(ds.intrinsics), ds, args.angs, args.ds_factor, camera=cam, link=cam.claims(), **kw)
    else:
        a = Live(args.intrin, args.dataset, args.angs, args.ds_factor)
    n = a.run(args.frames)
    a.stop()
    print(f"{n} frames, saved to {a.save_to}")
