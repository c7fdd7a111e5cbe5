"""Small helpers shared by the prediction path (reference: robotpose/utils.py:51-58,83-97)."""
import numpy as np

from .constants import JOINT_LETTERS


def str_to_arr(string: str) -> np.ndarray:
    """'SLU' -> bool(6) over joints S,L,U,R,B,T.  Unknown letters raise ValueError
    exactly as list.index does in the reference (utils.py:51-58)."""
    out = np.zeros(6, bool)
    for letter in string.upper():
        out[JOINT_LETTERS.index(letter)] = True
    return out


def get_extremes(mat: np.ndarray):
    """[min row, max row, min col, max col] of the True cells (utils.py:83-97)."""
    r, c = np.where(mat)
    return [int(r.min()), int(r.max()), int(c.min()), int(c.max())]
