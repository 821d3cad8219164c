"""Per-collective cost at world 1 on one GPU, through both routes of the C ABI: the callback (Python -> torch.distributed.all_reduce
on a tensor aliasing the library's buffer, backend nccl = RCCL) and RCCL called by the library itself (mvus_ba_set_rccl ->
ncclAllReduce).  Sizes: the five sums of one time-sharded LM iteration at configs[2] / configs[3] and the whole
packed normal equations of an observation shard.  Then an LM step of a (one-rank) observation-sharded handle through each route."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import numpy as np
import torch
import torch.distributed as dist
from mvus_amd import ba, problem as mp, synth
from mvus_amd.dist import make_gpu_allreduce, join_rccl

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kw = dict(synth.BASELINE_CONFIGS[cfg])
prob, x0 = mp.problem_from_scene(synth.make_scene(**kw))
sizes = [('2 scalars', 2), ('step p (n + 2)', 15299), ('Schur contributions CB x (CB+1), configs[2]', 288 * 289),
         ('camera blocks + halo + diag/g, configs[2]', 133000), ('separator system, configs[2]', 425000),
         ('Schur contributions, configs[3]', 576 * 577), ('packed normal equations of an observation shard, configs[2]', 4_600_000)]
stream = torch.cuda.current_stream(0).cuda_stream
out = {}
for route in ('torch callback', 'rccl native'):
    with ba.BAHandle(prob, device=0, stream=stream) as h:
        if route == 'torch callback':
            h.set_allreduce(make_gpu_allreduce(0), is_root=True)
        else:
            ok, why = join_rccl(h, 0, 1)
            assert ok, why
        for name, cnt in sizes:
            ms = h.time_allreduce(cnt, 200)
            out.setdefault(name, {})[route] = ms
        x = x0.copy()
        for _ in range(3):
            x = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False).x
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            x = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False).x
        torch.cuda.synchronize()
        out.setdefault('LM step, configs[%d], one-rank observation shard (1 sum of the packed blocks + 3 scalar sums per step)' % cfg, {})[route] = 1e3 * (time.perf_counter() - t0) / 20
with ba.BAHandle(prob, device=0, stream=stream) as h:
    x = x0.copy()
    for _ in range(3):
        x = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False).x
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        x = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False).x
    torch.cuda.synchronize()
    base = 1e3 * (time.perf_counter() - t0) / 20
print('# world 1, one MI355X: microseconds per all-reduce (HIP events over 200 back-to-back calls on the handle\'s stream)')
print('%-92s %16s %14s' % ('buffer', 'torch callback', 'rccl native'))
for name, d in out.items():
    unit = 1.0 if name.startswith('LM step') else 1e3
    print('%-92s %16.1f %14.1f' % (name + (' [ms -> us]' if unit == 1e3 else ' [us/step]').replace(' [ms -> us]', ''), d['torch callback'] * 1e3 if unit == 1e3 else d['torch callback'] * 1e3, d['rccl native'] * 1e3))
print('LM step without any all-reduce route (plain handle): %.1f us' % (base * 1e3))
dist.destroy_process_group()
