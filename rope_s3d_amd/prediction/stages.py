"""Stage descriptors of the prediction state machine.

Same vocabulary, parameters and hard-coded stage lists as the reference
(robotpose/prediction/stages.py:16-178): Lookup, SFlip, Descent, InterpolativeSweep,
TensorSweep and getStages('SL' | 'SLU').
"""
from dataclasses import dataclass, field
from typing import List, Optional, Union

import numpy as np

from ..utils import str_to_arr


def _joints(j: Union[str, np.ndarray]) -> np.ndarray:
    return str_to_arr(j) if isinstance(j, str) else np.asarray(j, bool)


class Lookup:
    """Compare the target against the pre-rendered pose grid (stages.py:16-24).  Grid size and
    varying joints are global knobs (constants.LOOKUP_*), not per-instance."""


@dataclass
class SFlip:
    """Try the S angle mirrored about the camera-derived axis (stages.py:30-41)."""
    to_render: int


@dataclass
class _Sweep:
    to_render: int
    divs: int
    joints: Union[str, np.ndarray]
    range: Optional[float] = None          # rad about the current angle; None = full joint range

    def __post_init__(self):
        self.joints = _joints(self.joints)


class InterpolativeSweep(_Sweep):
    """Sample a joint's range, interpolate the error cubically, test the predicted minimum
    (stages.py:50-69)."""


class TensorSweep(_Sweep):
    """Sample a joint's range and pick by the batched sqrt-depth score (stages.py:71-90)."""


@dataclass
class Descent:
    """Coordinate descent with per-joint step halving (stages.py:92-119).

    init_rate: scalar, None, or six entries; a None entry keeps the running step size."""
    to_render: int
    its: int
    joints: Union[str, np.ndarray]
    init_rate: Union[float, int, None, List] = None
    rate_redux: float = 0.5
    early_stop: float = 0.01

    def __post_init__(self):
        self.joints = _joints(self.joints)
        if self.init_rate is None or type(self.init_rate) in (float, int):
            self.init_rate = [self.init_rate] * 6


IntSweep = InterpolativeSweep
ISweep = InterpolativeSweep
TSweep = TensorSweep


def getStages(angles: str):
    """Stage list for a joint set; None when the set is not defined (stages.py:128-178)."""
    if angles == 'SL':
        s_flip = SFlip(4)
        sweeps = [InterpolativeSweep(4, 10, 'L', 0.1), InterpolativeSweep(4, 10, 'S', 0.1)]
        return [Lookup(), s_flip, *sweeps, s_flip]
    if angles == 'SLU':
        s_flip_4 = SFlip(4)
        sl_tune = Descent(4, 10, 'SL', [0.05, 0.05, 0.1, 0.5, 0.5, 0.5], early_stop=0.1)
        u_stages = [InterpolativeSweep(6, 25, 'U'), s_flip_4, SFlip(6), InterpolativeSweep(6, 10, 'U', 0.1)]
        full_tune = Descent(6, 40, 'SLU', early_stop=0.0075)
        return [Lookup(), s_flip_4, sl_tune, s_flip_4, *u_stages, full_tune]
    return None
