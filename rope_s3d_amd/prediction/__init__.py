from .predict import Predictor
from .synthetic import SyntheticPredictor
