"""Lookup pose grids (reference: robotpose/simulation/lookup.py:39-106,184-316).

The reference pre-renders the grid into an HDF5 table of cropped depth images and scores a
frame against it with TensorFlow.  The engine renders and scores the grid on the fly
(ROPE_LOSS_LOOKUP), so this module only has to produce the same grid, in the same
order, with the same size rule.
"""
from typing import Union

import numpy as np

from ..constants import GPU_MEMORY_ALLOWED_FOR_LOOKUP, LOOKUP_MAX_DIV_PER_LINK
from ..utils import str_to_arr

# The reference sizes the table from the GPU's VRAM read through nvidia-smi (utils.py:21-37):
# bits = MiB * 8.389e6.  nvidia-smi does not exist here, so the budget is explicit; the default
# is the 8 GiB class of card the reference was configured on (crop.py:121 "GTX 1070").
DEFAULT_LOOKUP_VRAM_MIB = 8192


def lookup_grid(joint_limits: np.ndarray, varying: Union[str, np.ndarray], divisions) -> np.ndarray:
    """Grid of joint vectors, joint 0 varying fastest (lookup.py:39-66)."""
    varying = str_to_arr(varying) if isinstance(varying, str) else np.asarray(varying, bool)
    divisions = np.clip(np.array(divisions, dtype=int), 0, LOOKUP_MAX_DIV_PER_LINK)
    divisions[~varying] = 1
    num = int(np.prod(divisions))
    angles = np.zeros((num, 6))
    for idx in np.where(varying)[0]:
        rng = np.linspace(joint_limits[idx, 0], joint_limits[idx, 1], divisions[idx])
        repeat = int(np.prod(divisions[:idx]))
        tile = num // (repeat * divisions[idx])
        angles[:, idx] = np.tile(np.repeat(rng, repeat), tile)
    return angles


def default_divisions(crop_size: int, varying: Union[str, np.ndarray], vram_mib: float = DEFAULT_LOOKUP_VRAM_MIB,
                      element_bits: int = 32) -> np.ndarray:
    """Equal split of the pose budget over the varying joints (lookup.py:224-225,266-274)."""
    varying = str_to_arr(varying) if isinstance(varying, str) else np.asarray(varying, bool)
    max_elements = int(vram_mib * 8.389e6 * GPU_MEMORY_ALLOWED_FOR_LOOKUP)
    max_poses = max_elements / (crop_size * element_bits)
    divisions = np.zeros(6, int)
    divisions[varying] = int(max_poses ** (1 / sum(varying)))
    return divisions


class RobotLookupManager:
    """`get` keeps the reference's call shape but returns (angles, None): there is no table."""

    def __init__(self, joint_limits: np.ndarray, element_bits: int = 32):
        self.joint_limits = joint_limits
        self.element_bits = element_bits

    def get(self, crop_size: int, varying_angles: Union[str, np.ndarray], max_poses: int = None,
            divisions: np.ndarray = None, vram_mib: float = DEFAULT_LOOKUP_VRAM_MIB):
        assert not (max_poses is not None and divisions is not None), "give at most one of max_poses / divisions"
        varying = str_to_arr(varying_angles) if isinstance(varying_angles, str) else np.asarray(varying_angles, bool)
        if divisions is None:
            if max_poses is not None:
                divisions = np.zeros(6, int)
                divisions[varying] = int(max_poses ** (1 / sum(varying)))
            else:
                divisions = default_divisions(crop_size, varying, vram_mib, self.element_bits)
        return lookup_grid(self.joint_limits, varying, divisions), None
