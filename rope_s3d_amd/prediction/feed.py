"""Sources of the live loop (predict_live.py): where frames and the controller's claimed joint angles come from.

The reference reads an Intel RealSense through pyrealsense2 (robotpose/prediction/feed.py:14-86) and the robot
controller's joint state from a JSON file the controller rewrites (robotpose/textfile_integration.py:19-69).  The
JSON link is plain file traffic and is kept as is; the camera needs pyrealsense2 and the device, neither of which
exists here, so `LiveCamera` says so loudly and `DatasetCamera` replays a recorded Dataset through the same
`start / get / stop` surface (its `claims()` pairs every frame with the recorded joint angles)."""
import json
import os
import time

import numpy as np

from ..constants import JSON_LINK_FILE


class JSONCoupling:
    """The controller writes {"position": [six joint angles]} to a file; get_pose polls for it, reset removes it so
    that the next pose read is a fresh one (textfile_integration.py:23-69)."""

    def __init__(self, path: str = JSON_LINK_FILE, poll: float = 0.0001):
        self.path, self.poll, self.data = path, poll, None

    def get_pose(self, timeout: float = None):
        start = time.time()
        while True:
            if os.path.isfile(self.path):
                try:
                    with open(self.path, 'r') as f:
                        self.data = json.load(f)
                    break
                except (OSError, ValueError):       # the controller is mid-write: try again
                    pass
            if timeout is not None and time.time() - start > timeout:
                return None
            time.sleep(self.poll)
        return np.array(self.data['position'])

    def reset(self, timeout: float = None):
        start = time.time()                      # the reference's `start = time.time` (no call) makes its timeout a TypeError
        while os.path.isfile(self.path):
            try:
                os.remove(self.path)
                break
            except OSError:
                pass
            if timeout is not None and time.time() - start > timeout:
                break
            time.sleep(self.poll)


class LiveCamera:
    """RealSense colour + aligned depth (feed.py:14-86).  Needs pyrealsense2 and the camera."""

    def __init__(self, width: int = 1280, height: int = 720, fps: int = 30):
        try:
            import pyrealsense2  # noqa: F401
        except ImportError as e:
            raise RuntimeError("LiveCamera needs pyrealsense2 and an Intel RealSense device; replay a recorded set "
                               "with DatasetCamera instead (predict_live.py -replay <dataset>)") from e
        raise NotImplementedError("RealSense capture is outside the prediction path (DESIGN.md §9)")


class DatasetCamera:
    """A recorded Dataset played back as a camera: get() returns (colour uint8 BGR, depth metres) of the next frame,
    None when the recording is over."""

    def __init__(self, dataset, start: int = 0, stop: int = None):
        self.ds, self.i, self.end = dataset, start, dataset.length if stop is None else min(stop, dataset.length)

    def start(self):
        pass

    def stop(self):
        pass

    def get(self):
        if self.i >= self.end:
            return None
        i, self.i = self.i, self.i + 1
        return np.copy(self.ds.og_img[i]), np.copy(self.ds.depthmaps[i])

    def claims(self):
        """The controller side of a replay: the recorded joint angles of the frame get() hands out next."""
        ds, cam = self.ds, self

        class _Recorded:
            def get_pose(self, timeout=None):
                return None if cam.i >= cam.end else np.array(ds.angles[cam.i], dtype=float)

            def reset(self, timeout=None):
                pass
        return _Recorded()
