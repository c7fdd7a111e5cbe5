"""Drop-in alias: `from robotpose import Dataset, Predictor, Grapher` (predict_dataset.py:13, synth.py:13 of the
reference) and `from robotpose import Predictor, JSONCoupling, LiveCamera, Dataset, Intrinsics` (predict_live.py:2)
resolve to the MI355X engine's host package."""
from rope_s3d_amd import Dataset, Grapher, Intrinsics, Paths, Predictor, Renderer, SyntheticPredictor  # noqa: F401
from rope_s3d_amd.prediction.feed import JSONCoupling, LiveCamera  # noqa: F401
