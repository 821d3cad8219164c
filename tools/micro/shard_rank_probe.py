"""What ONE rank of a time-sharded run computes per LM step: rank R of WORLD on BASELINE configs[CFG], alone on the GPU, with an all-reduce
callback that leaves the buffers as they are (the sums are wrong, the kernels and their sizes are the real ones).  Under
`rocprofv3 --kernel-trace --stats` this gives the per-kernel times behind the multi-GPU time model of DESIGN section 6 -- measured for
the rank's slice instead of scaled from the one-GPU run.  usage: shard_rank_probe.py [cfg=3] [world=8] [rank=3] [steps=8] [obs multiplier=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from mvus_amd import ba, problem as mp, synth
cfg, world, rank, steps, mult = [int(a) for a in (sys.argv[1:] + ['3', '8', '3', '8', '1'][len(sys.argv) - 1:])][:5]
kw = dict(synth.BASELINE_CONFIGS[cfg])
kw['total_obs'] *= mult              # (bench.py's weak scaling: the configuration's detections PER GPU)
prob, x0 = mp.problem_from_scene(synth.make_scene(**kw))
shard, keep, cuts = prob.shard_time(rank, world, x0)
calls = []
with ba.BAHandle(shard, device=0) as h:
    h.set_time_shard(rank, world, cuts)
    h.set_allreduce(lambda ptr, count, stream: calls.append(count), is_root=(rank == 0))
    x = x0.copy()
    for it in range(steps + 2):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter(); n0 = len(calls)
        try:
            h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False)      # (x0 every time: the sums are not real)
        except Exception as e:
            print('solve:', type(e).__name__, str(e)[:100])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3 / steps
per = (len(calls) - n0) / steps
sizes = calls[n0:n0 + int(round(per))]
print('configs[%d] rank %d of %d: %d of %d detections, %d of %d control points; %.3f ms per step on this rank alone (callbacks: no-ops); '
      '%.1f collectives per step, doubles: %s; MVUS_SEP_TWO_LEVEL=%s' % (cfg, rank, world, shard.M, prob.M, cuts[rank + 1] - cuts[rank], int(prob.n_coef.sum()), dt, per, sizes,
                                                                        os.environ.get('MVUS_SEP_TWO_LEVEL', 'default')))
