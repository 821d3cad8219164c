#!/usr/bin/env python3
"""Time shards on ONE GPU (host threads + in-process sum) against the unsharded LM solve at a BASELINE config:
   python tools/check_time_shards.py <config> <world> <max_nfev>"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mvus_amd import _lib, problem as mp, synth
from mvus_amd.ba import BAHandle
from mvus_amd.dist import _DeviceDoubles

cfg, world, nfev = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kw = dict(synth.BASELINE_CONFIGS[cfg])
kw['total_obs'] *= int(os.environ.get('OBS_SCALE', '1'))      # bench.py's weak scaling: detections x world, same cameras and knots
prob, x0 = mp.problem_from_scene(synth.make_scene(**kw))
opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, nfev)
opts.verbose = 2 if len(sys.argv) > 4 else 0
calls = int(os.environ.get('CALLS', '1'))          # CALLS > 1: a run of short solves on one handle, like bench.py

lm_lambda = float(os.environ['LMSTEP']) if 'LMSTEP' in os.environ else None      # LMSTEP=<lambda>: compare one damped step only


class _Step:
    def __init__(self, p):
        self.x, self.nfev, self.njev, self.initial_cost, self.cost = p, 0, 0, 0.0, 0.0


def run_calls(h):
    if lm_lambda is not None:
        h.residual_jacobian(x0, 0)
        return _Step(h.lm_step(lm_lambda))
    x, out = x0, None
    for _ in range(calls):
        out = h.solve(x, opts=opts)
        x = out.x
        print('   call: cost %.10e -> %.10e nfev %d' % (out.initial_cost, out.cost, out.nfev), flush=True)
    return out

with BAHandle(prob) as h0:
    ref = run_calls(h0)
print('unsharded: nfev %d njev %d cost %.10e -> %.10e' % (ref.nfev, ref.njev, ref.initial_cost, ref.cost))
barrier = threading.Barrier(world)
bufs, total, results, errors = [None] * world, [None], [None] * world, []

def make_cb(rank):
    def cb(ptr, count, stream):
        t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
        torch.cuda.synchronize()
        bufs[rank] = t
        barrier.wait(120)
        if rank == 0:
            total[0] = torch.stack(bufs).sum(0)
            torch.cuda.synchronize()
        barrier.wait(120)
        t.copy_(total[0])
        torch.cuda.synchronize()
        barrier.wait(120)
    return cb

def run(rank):
    try:
        shard, keep, cuts = prob.shard_time(rank, world, x0)
        h = BAHandle(shard, device=0)
        h.set_time_shard(rank, world, cuts)
        h.set_allreduce(make_cb(rank), is_root=(rank == 0))
        results[rank] = run_calls(h)
        h.close()
    except Exception as e:
        errors.append(e); barrier.abort()

ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
[t.start() for t in ts]; [t.join(600) for t in ts]
print('errors', errors)
for r, res in enumerate(results):
    if res is not None:
        print('rank %d: nfev %d njev %d cost %.10e -> %.10e  max|dx| %.3e' % (r, res.nfev, res.njev, res.initial_cost, res.cost, np.abs(res.x - ref.x).max()))
