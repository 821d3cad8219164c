"""Observation sharding for multi-GPU bundle adjustment (one process per GPU).

Every residual row depends on one camera's parameters and four neighbouring control points, and
J^T J / J^T r are sums over rows, so observations shard freely.  Each camera's (time-ordered)
detections are cut into ``world`` contiguous pieces and rank r takes piece r of every camera: all
ranks then cover the same share of the time axis per camera and do equal work.  Camera
parameters, the spline and the motion-regulariser rows' inputs are replicated; the only exchange
is the sum of the per-rank normal-equation / J^T u partials (see ``Reducer``).
"""
import numpy as np


def shard_offsets(count, rank, world):
    """[lo, hi) of piece ``rank`` when ``count`` items are split into ``world`` near-equal pieces."""
    base, rem = divmod(int(count), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Reducer:
    """Sum-all-reduce of a float64 buffer across ranks through ``torch.distributed``.

    Backend ``nccl`` is RCCL on ROCm (GPU tensors over xGMI); ``gloo`` is used by the CPU tests.
    With ``world_size == 1`` (or no process group) it is the identity."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.calls = 0
        self.elements = 0

    def all_reduce_(self, tensor):
        if self.active:
            self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM, group=self.group)
            self.calls += 1
            self.elements += tensor.numel()
        return tensor
