"""robotpose.utils: the helpers the prediction path's callers import (predict_live.py:4)."""
from rope_s3d_amd.prediction.viz import color_array  # noqa: F401
from rope_s3d_amd.utils import get_extremes, str_to_arr  # noqa: F401
