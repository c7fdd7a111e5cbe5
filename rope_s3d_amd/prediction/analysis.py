"""Error statistics of a prediction run (the numbers the reference prints from its plots,
robotpose/prediction/analysis.py:57-73): mean, std, median, p90/p95/p99, max of |error| per joint."""
import numpy as np


def joint_error_stats(predicted: np.ndarray, actual: np.ndarray) -> dict:
    err = np.abs(np.asarray(predicted, float) - np.asarray(actual, float))
    return {
        'mean': err.mean(0), 'std': err.std(0), 'median': np.median(err, 0),
        'p90': np.percentile(err, 90, 0), 'p95': np.percentile(err, 95, 0), 'p99': np.percentile(err, 99, 0),
        'max': err.max(0),
    }


class Grapher:
    """Text-only stand-in for the matplotlib Grapher (analysis.py:17-144): `plot` prints the table."""

    def __init__(self, joints: str, prediction: np.ndarray, real: np.ndarray = None):
        self.joints, self.prediction, self.real = joints.upper(), np.asarray(prediction), None if real is None else np.asarray(real)

    def plot(self, ylim=None):
        if self.real is None:
            return
        stats = joint_error_stats(self.prediction, self.real)
        names = 'SLURBT'
        print('joint ' + ' '.join(f'{k:>9s}' for k in stats))
        for j, nme in enumerate(names):
            if nme in self.joints:
                print(f'{nme:>5s} ' + ' '.join(f'{stats[k][j]:9.5f}' for k in stats))
