"""One-process-per-GPU plumbing: observation shards + the all-reduce hook of the C ABI.

PyTorch is used here for exactly two things: ``torch.distributed`` (backend ``nccl`` = RCCL over
xGMI on ROCm) and the HIP stream the kernels and the collective are ordered on.  The device
buffers the library asks to reduce are aliased as torch tensors through
``__cuda_array_interface__`` -- no staging copy.
"""
import numpy as np

from . import sharding
from .ba import BAHandle


class _DeviceDoubles:
    """Zero-copy view of ``count`` float64 at a raw device pointer."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<f8', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def make_gpu_allreduce(device_index, group=None):
    """Returns fn(ptr, count, stream) summing the buffer across ranks with torch.distributed (RCCL)."""
    import torch
    import torch.distributed as dist
    stats = {'calls': 0, 'doubles': 0}
    views = {}                    # the library reduces the same few persistent buffers every iteration: alias them once

    streams = {}

    def fn(ptr, count, stream):
        t = views.get((ptr, count))
        if t is None:
            t = views[(ptr, count)] = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:%d' % device_index)
        # the collective must be ordered on the stream the library's kernels run on (its own stream unless the handle was
        # created on an explicit one): make that stream torch's current stream around the call
        if stream:
            ext = streams.get(stream)
            if ext is None:
                ext = streams[stream] = torch.cuda.ExternalStream(stream, device='cuda:%d' % device_index)
            with torch.cuda.stream(ext):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        stats['calls'] += 1
        stats['doubles'] += count

    fn.stats = stats
    return fn


def agree_on_rccl(rank, world, group, get_id, join, leave, probe=None):
    """The route of the sums is a COLLECTIVE decision: every rank returns the same (ok, reason).

    ``probe()`` (every rank; raises when this rank cannot open RCCL at all) is asked FIRST and the answers are exchanged: ``join`` is
    ncclCommInitRank, itself a blocking collective -- a rank that cannot even load the library must be found out before anybody
    enters it, or the others wait there for ever.

    ``get_id()`` (rank 0 only) returns the 128-byte RCCL id or raises; ``join(id)`` makes this rank join the library's communicator
    or raises; ``leave()`` tears down a communicator that did initialise.  Protocol, every step a collective of ``group`` that all
    ranks enter whatever happened locally: (1) rank 0 broadcasts (ok, id-or-error) -- it never raises before the broadcast;
    (2) if there is an id every rank tries to join, then all ranks exchange their outcome (all_gather_object); (3) unless ALL joined,
    the ranks that did join leave again and everyone reports failure.  No rank can end up in the native route while another one is
    in the callback route (mismatched collectives = a hang)."""
    import torch.distributed as dist
    if probe is not None:
        perr = None
        try:
            probe()
        except Exception as e:
            perr = '%s: %s' % (type(e).__name__, e)
        answers = [None] * world
        if world > 1:
            dist.all_gather_object(answers, perr, group=group)
        else:
            answers[0] = perr
        missing = [(r, o) for r, o in enumerate(answers) if o is not None]
        if missing:
            return False, '; '.join('rank %d cannot open RCCL (%s)' % f for f in missing)
    box = [None]
    if rank == 0:
        try:
            box[0] = (True, bytes(get_id()))
        except Exception as e:
            box[0] = (False, '%s: %s' % (type(e).__name__, e))
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    ok, payload = box[0]
    if not ok:
        return False, 'rank 0 could not get an RCCL id (%s)' % payload
    err = None
    try:
        join(payload)
    except Exception as e:
        err = '%s: %s' % (type(e).__name__, e)
    outcomes = [None] * world
    if world > 1:
        dist.all_gather_object(outcomes, err, group=group)
    else:
        outcomes[0] = err
    failed = [(r, o) for r, o in enumerate(outcomes) if o is not None]
    if failed:
        if err is None:
            leave()
        return False, '; '.join('rank %d: %s' % f for f in failed)
    return True, ''


def join_rccl(handle, rank, world, group=None, is_root=None):
    """Native route: rank 0 asks the library for an RCCL id, torch.distributed (whatever its backend) hands the 128 bytes to the
    other ranks -- once -- and every rank joins the library's own communicator; from then on the iteration's sums are
    ncclAllReduce calls made by the library on the handle's stream.  Returns (ok, reason), the same on every rank
    (``agree_on_rccl``); on failure the handle is left without a route."""
    from . import _lib
    root = (rank == 0) if is_root is None else is_root
    return agree_on_rccl(rank, world, group, _lib.rccl_unique_id,
                         lambda uid: handle.set_rccl(uid, rank, world, is_root=root),
                         lambda: handle.set_allreduce(None, is_root=root), probe=_lib.rccl_available)


def sharded_handle(prob, rank, world, device_index, group=None, time_x=None, halo=8, collective='torch'):
    """BAHandle over observation shard ``rank`` of ``world`` on ``cuda:device_index``.

    ``time_x`` given (the start parameters): shard by TIME (BAProblem.shard_time + mvus_ba_set_time_shard) -- the
    LM/Schur solver then exchanges a few MB per iteration instead of the whole cross block; every rank must pass the
    same ``time_x``.  Without it every camera's detections are cut into ``world`` pieces (any solver).

    ``collective``: 'auto' = 'rccl' if the library can open and initialise RCCL, else 'torch'; 'torch' = the all-reduce callback (torch.distributed on aliased device buffers: any backend, gloo on CPU
    boxes), 'rccl' = RCCL called from the library on its own communicator (the route bench.py takes on a multi-GPU node).

    The handle runs on torch's current stream of that device so that its kernels and the collective are
    ordered without extra events.  Motion-regulariser rows are replicated inputs but must be counted
    once: every shard keeps the same layout and `is_root` tells the library which rank evaluates them."""
    import torch
    torch.cuda.set_device(device_index)
    cuts = None
    if time_x is not None and world > 1:
        shard, keep, cuts = prob.shard_time(rank, world, time_x, halo)
    else:
        shard, keep = prob.shard(rank, world)   # motion samples stay in every shard (same layout on all ranks);
                                                # the library evaluates them on the root rank only (is_root)
    stream = torch.cuda.current_stream(device_index).cuda_stream
    h = BAHandle(shard, device=device_index, stream=stream)
    if cuts is not None:
        h.set_time_shard(rank, world, cuts, halo)
    if world > 1 or group is not None:
        use_rccl = collective in ('rccl', 'auto')
        if use_rccl:                   # ncclAllReduce called by the library itself (mvus_ba_set_rccl); all ranks agree on the outcome
            ok, why = join_rccl(h, rank, world, group)
            if not ok:
                if collective == 'rccl':
                    raise RuntimeError('sharded_handle: RCCL from the library is not available: ' + why)
                if rank == 0:
                    print('sharded_handle: RCCL from the library is not available (%s) -- every rank uses the torch.distributed callback' % why, flush=True)
                use_rccl = False
        if not use_rccl:               # the callback: torch.distributed on tensors aliasing the library's buffers (any backend)
            cb = make_gpu_allreduce(device_index, group)
            h.set_allreduce(cb, is_root=(rank == 0))
            h.allreduce_stats = cb.stats
        h.collective_used = 'rccl' if use_rccl else 'torch'
    return h, keep


def solve_time_sharded(make_handle, x0, max_nfev=10, max_recuts=4, **solve_kw):
    """LM on time shards that survives drifting time stamps.  ``make_handle(x)`` returns this rank's handle with the timeline cut at x
    (``sharded_handle(prob, rank, world, device, time_x=x, ...)[0]``, or a test's own construction); when the solver reports that a
    row has left its slice (``ReshardNeeded``: raised on every rank together, carrying the point reached so far) the timeline is cut
    again at that point and the solve continues with the evaluations that are left -- the reference re-evaluates visibility at every
    call (common.py:317, tools/util.py:90-116); here that costs one shard_time + handle per re-cut, counted in ``result.recuts``."""
    from .ba import ReshardNeeded
    from .ba import _Result
    x = np.array(x0, dtype=np.float64)
    left, recuts, used, first_cost = int(max_nfev), 0, 0, None
    while True:
        h = make_handle(x)
        try:
            r = h.solve(x, max_nfev=max(left, 1), **solve_kw)
            r.recuts, r.nfev_total = recuts, used + r.nfev
            # the caller's view is ONE solve: evaluations over all segments, the cost it started from
            r.nfev = r.nfev_total
            if first_cost is not None:
                r.initial_cost = first_cost
            return r
        except ReshardNeeded as e:
            recuts += 1
            if recuts > max_recuts:
                raise
            x = np.array(e.x)
            if first_cost is None:
                first_cost = getattr(e, 'initial_cost', None)
            # (the evaluation at the start of the next solve repeats the last one of this solve: it is not charged twice)
            used += max(e.nfev - 1, 0)
            left = int(max_nfev) - used
            if left < 2:
                # the evaluation budget is spent: the point reached so far is the answer (status 0, as scipy reports max_nfev) -- no
                # further segment is forced onto the caller's budget
                return _Result(x=x, cost=float(e.cost), fun=None, nfev=used + 1, njev=0, status=0, optimality=float('nan'), lin_iters=0,
                               solve_ms=0.0, initial_cost=first_cost if first_cost is not None else float(e.cost), success=False,
                               grad=None, jac=None, recuts=recuts, nfev_total=used + 1)
        finally:
            h.close()


def shard_sizes(prob, world):
    return [sum(sharding.shard_offsets(int(prob.det_offsets[c + 1] - prob.det_offsets[c]), r, world)[1]
                - sharding.shard_offsets(int(prob.det_offsets[c + 1] - prob.det_offsets[c]), r, world)[0]
                for c in range(prob.C)) for r in range(world)]
