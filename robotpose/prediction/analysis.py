"""robotpose.prediction.analysis (predict_live.py:1, plot_errors.py of the reference)."""
from rope_s3d_amd.prediction.analysis import Grapher, JointDistance, joint_error_stats, print_error_table  # noqa: F401
