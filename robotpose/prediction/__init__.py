"""Alias of the reference's sub-package: `from robotpose.prediction.camera_pose_prediction import CameraPredictor`."""
from rope_s3d_amd.prediction.predict import Predictor  # noqa: F401
