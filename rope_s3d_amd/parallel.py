"""Frame sharding across ranks and the single gather of final joint angles.

The prediction path has no data-path collective: frames are independent
(Predictor.run resets its state per call, predict.py:144-148), so rank r of R takes a
contiguous block of frames and the only exchange is one all-gather of (n_local, 6)
float64 results — RCCL over xGMI with backend 'nccl', gloo on CPU in the tests.
"""
import os

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Frames [lo, hi) of rank `rank`: blocks of ceil(n/world)."""
    per = -(-n // world)
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def dist_env():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def gather_rows(local: np.ndarray, n_total: int, device=None, single_rank_too: bool = False) -> np.ndarray:
    """All-gather per-rank result blocks (n_local, k) into (n_total, k) on every rank.

    Blocks are padded to ceil(n/world) rows so that one fixed-size all_gather suffices.  A world of one returns its block as it
    is, unless single_rank_too asks for the collective anyway (one rank's rehearsal of the RCCL path)."""
    local = np.ascontiguousarray(local, np.float64)
    try:
        import torch
        import torch.distributed as dist
    except ImportError:
        return local[:n_total]
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not single_rank_too):
        return local[:n_total]
    world = dist.get_world_size()
    per = -(-n_total // world)
    k = local.shape[1]
    buf = torch.zeros((per, k), dtype=torch.float64, device=device or 'cpu')
    buf[:len(local)] = torch.from_numpy(local).to(buf.device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat(out).cpu().numpy()[:n_total]
