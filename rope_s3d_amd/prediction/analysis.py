"""Error statistics of a prediction run: the tables the reference prints under its plots
(robotpose/prediction/analysis.py:17-73,79-160), without matplotlib — `plot` prints, nothing is drawn.

`Grapher`: per-joint |error| in degrees (mean, std, median, 90th/95th/99th percentile, max) with the reference's
wrap-around correction of joint B (:96-105).  `JointDistance`: Cartesian distance between the predicted and the
actual position of every joint frame, from the forward kinematics of the URDF chain (:138-160)."""
import numpy as np

from ..utils import str_to_arr


def joint_error_stats(predicted: np.ndarray, actual: np.ndarray) -> dict:
    err = np.abs(np.asarray(predicted, float) - np.asarray(actual, float))
    return {
        'mean': err.mean(0), 'std': err.std(0), 'median': np.median(err, 0),
        'p90': np.percentile(err, 90, 0), 'p95': np.percentile(err, 95, 0), 'p99': np.percentile(err, 99, 0),
        'max': err.max(0),
    }


def print_error_table(joints, unit: str, err: np.ndarray) -> dict:
    """The 'Err Stats' block of general_plot (analysis.py:55-73); returns the numbers it printed."""
    st = joint_error_stats(err, np.zeros_like(err))
    w = 6
    print(f"\nErr Stats ({unit}):")
    print(f"\t   {' ' * (w - 4)}Mean {' ' * (w - 3)}Std | {' ' * (w - 3)}Med {' ' * (w - 4)}90th {' ' * (w - 4)}95th {' ' * (w - 4)}99th {' ' * (w - 3)}Max")
    for idx, joint in enumerate(joints):
        print(f"\t{joint}: {st['mean'][idx]:{w}.2f} {st['std'][idx]:{w}.2f} | {st['median'][idx]:{w}.2f} {st['p90'][idx]:{w}.2f} "
              f"{st['p95'][idx]:{w}.2f} {st['p99'][idx]:{w}.2f} {st['max'][idx]:{w}.2f}")
    return st


class Grapher:

    def __init__(self, joints_to_plot: str, predictions: np.ndarray, ds_angles: np.ndarray = None):
        self.compare = ds_angles is not None
        self.joints = [x for x in joints_to_plot.upper()]
        self.predictions = np.degrees(np.asarray(predictions, float))
        self.true = np.degrees(np.asarray(ds_angles, float)) if self.compare else None
        if self.compare:
            self._b_correction()
            self._cropComparison()

    def plot(self, ylim=None) -> dict:
        if not self.compare:
            return {}
        return print_error_table(self.joints, 'deg', self.predictions - self.true)

    def _b_correction(self):
        """Joint B may be predicted a half or a full turn off; take the closest of -360..360 in steps of 180 (:96-105)."""
        if 'B' not in self.joints:
            return
        offsets = [-360, -180, 0, 180, 360]
        for idx in range(len(self.predictions)):
            err = [abs((self.predictions[idx, 4] + x) - self.true[idx, 4]) for x in offsets]
            self.predictions[idx, 4] += offsets[err.index(min(err))]

    def _cropComparison(self):
        ang = ['S', 'L', 'U', 'R', 'B', 'T']
        n = len(self.predictions)
        cols = [ang.index(j) for j in self.joints]
        self.true, self.predictions = self.true[:n][:, cols].copy(), self.predictions[:, cols].copy()


class JointDistance:
    """Distance (m) between predicted and actual joint-frame origins; columns S..T as in the angle arrays."""

    def __init__(self):
        from ..simulation.kinematics import ForwardKinematics
        self._fk = ForwardKinematics()
        self.joints_str = 'LURBT'
        self.joints = [x for x in self.joints_str]

    def distance(self, predicted: np.ndarray, actual: np.ndarray) -> np.ndarray:
        predicted, actual = np.asarray(predicted, float), np.asarray(actual, float)
        assert predicted.shape[0] == actual.shape[0]
        distances = np.zeros(predicted.shape)
        for idx in range(predicted.shape[0]):
            a = self._fk.calc(actual[idx])[1:, :3, 3]
            p = self._fk.calc(predicted[idx])[1:, :3, 3]
            distances[idx] = np.sum((a - p) ** 2, -1) ** 0.5
        return distances

    def plot(self, predicted: np.ndarray, actual: np.ndarray, y_lim=None) -> dict:
        err = self.distance(predicted, actual)
        return print_error_table(self.joints, 'cm', err[:, str_to_arr(self.joints_str)] * 100)

    def single(self, predicted, actual, joint='T'):
        return self.distance(predicted, actual)[..., str_to_arr(joint)]
