"""Drop-in alias: `from robotpose import Dataset, Predictor, Grapher` (predict_dataset.py:13,
synth.py:13 of the reference) resolves to the MI355X engine's host package."""
from rope_s3d_amd import Dataset, Grapher, Intrinsics, Paths, Predictor, Renderer, SyntheticPredictor  # noqa: F401
