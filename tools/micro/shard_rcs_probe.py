"""Two time shards driven by host threads on one GPU, for a given camera count: does the LM step of the sharded handles match the unsharded one?"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mvus_amd import _lib, problem as mp, synth
from mvus_amd.ba import BAHandle
from mvus_amd.dist import _DeviceDoubles
C = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc = synth.make_scene(C, 400 * C, seed=3, rolling_shutter=True, num_knots=300)
prob, x0 = mp.problem_from_scene(sc)
kw = dict(solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC)
with BAHandle(prob) as h:
    ref = h.solve(x0, max_nfev=4, **kw)
world = 2
barrier = threading.Barrier(world)
bufs, total, errors, out = [None] * world, [None], [], [None] * world
def make_cb(rank):
    def cb(ptr, count, stream):
        t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
        torch.cuda.synchronize(); bufs[rank] = t; barrier.wait(30)
        if rank == 0:
            total[0] = bufs[0] + bufs[1]; torch.cuda.synchronize()
        barrier.wait(30); t.copy_(total[0]); torch.cuda.synchronize(); barrier.wait(30)
    return cb
def run(rank):
    try:
        shard, keep, cuts = prob.shard_time(rank, world, x0)
        h = BAHandle(shard, device=0); h.set_time_shard(rank, world, cuts); h.set_allreduce(make_cb(rank), is_root=(rank == 0))
        out[rank] = h.solve(x0, max_nfev=4, **kw); h.close()
    except Exception as e:
        import traceback; errors.append(traceback.format_exc()); barrier.abort()
ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join(120) for t in ts]
print('errors', errors[:1]); print('C', C, 'nn', C * 9, 'cost ref %.9g shard %.9g' % (ref.cost, out[0].cost if out[0] else float('nan')))
