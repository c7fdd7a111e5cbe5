"""rope_s3d_amd — MI355X-native render-and-compare pose engine behind RoPE-S3D's Predictor/Dataset API.

Exports mirror the names the reference package exposes for this path (robotpose/__init__.py:1-9).
Importing the package does not touch the GPU; constructing a Renderer/Predictor does.
"""
from .config import Paths
from .data.dataset import Dataset
from .prediction.analysis import Grapher
from .prediction.camera_pose_prediction import CameraPredictor, ModellessCameraPredictor
from .prediction.predict import Predictor
from .prediction.synthetic import SyntheticPredictor
from .projection import Intrinsics
from .simulation.render import Renderer

__all__ = ['Paths', 'Dataset', 'Grapher', 'Predictor', 'SyntheticPredictor', 'Intrinsics', 'Renderer',
           'CameraPredictor', 'ModellessCameraPredictor']
