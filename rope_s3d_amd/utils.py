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


def cpu_budget() -> int:
    """CPUs this process can really use: its scheduler affinity, cut down to the cgroup's CPU quota where one is set
    (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us` / `cpu.cfs_period_us`).  A container often shows every core of the host
    while its quota is a fraction of them."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def limit_host_threads(most: int = 8) -> int:
    """Caps PyTorch's intra-op thread pool at min(`most`, half the CPU budget).  The host-side tensor work of this package
    is small (index bookkeeping, a mask transposition), but the pool's idle threads spin: with one thread per visible core
    (128 on a 256-core host) and a container quota of 16 CPUs, a few milliseconds of spinning use up the quota of the
    scheduler's 100 ms period and the whole process — kernel launches included — is frozen for the rest of it.  Measured on
    the segmentation stage: every other batch took 80 ms instead of 17.5.  ROPE_TORCH_THREADS overrides; returns the count set."""
    import os
    import torch
    want = int(os.environ.get('ROPE_TORCH_THREADS', 0)) or max(1, min(most, cpu_budget() // 2))
    if torch.get_num_threads() > want or 'ROPE_TORCH_THREADS' in os.environ:
        torch.set_num_threads(want)
    return torch.get_num_threads()
