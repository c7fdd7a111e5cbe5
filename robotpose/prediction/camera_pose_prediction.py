from rope_s3d_amd.prediction.camera_pose_prediction import CameraPredictor, ModellessCameraPredictor  # noqa: F401
