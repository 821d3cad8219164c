"""Observation sharding for multi-GPU bundle adjustment (one process per GPU).

Every residual row depends on one camera's parameters and four neighbouring control points, and
J^T J / J^T r are sums over rows, so observations shard freely.  Each camera's (time-ordered)
detections are cut into ``world`` contiguous pieces and rank r takes piece r of every camera: all
ranks then cover the same share of the time axis per camera and do equal work.  Camera
parameters, the spline and the motion-regulariser rows' inputs are replicated; the only exchange
is the sum of the per-rank normal-equation / J^T u partials (the all-reduce callback of mvus_amd/dist.py).
"""
import numpy as np


def shard_offsets(count, rank, world):
    """[lo, hi) of piece ``rank`` when ``count`` items are split into ``world`` near-equal pieces."""
    base, rem = divmod(int(count), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
