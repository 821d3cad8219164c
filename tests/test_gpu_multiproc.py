"""The product's collective path with REAL processes: one process per rank, torch.distributed, the all-reduce callback of
mvus_amd/dist.py on the library's device buffers, time shards (LM + Schur) and observation shards (TRF + LSMR).

  * backend ``nccl`` (= RCCL) with one GPU per rank when the box has >= 2 devices -- skipped, with the reason, on a 1-GPU box
    (RCCL refuses two ranks on one device); ``rccl``: the same with RCCL called by the LIBRARY on its own communicator
    (mvus_ba_set_rccl: the route bench.py takes on a multi-GPU node), torch.distributed only carries the 128-byte id;
  * backend ``gloo`` with both ranks on cuda:0 (the sums are staged through the host by gloo): the same product code --
    sharded_handle, mvus_ba_set_time_shard, every packed buffer the HIP path reduces -- in two processes on any box.

Each worker is a fresh interpreter (multiprocessing 'spawn'); nothing here re-executes a process that touched the GPU."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from mvus_amd import synth
    return synth.make_scene(4, 12000, seed=43, rolling_shutter=True, num_knots=360, motion_reg=True, motion_type='F', motion_weights=50.0)


def _worker(rank, world, port, backend, mode, out_dir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    dev = rank if backend in ('nccl', 'rccl') else 0
    torch.cuda.set_device(dev)
    collective = 'rccl' if backend == 'rccl' else 'torch'
    if backend in ('nccl', 'rccl'):
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mvus_amd import _lib, problem as mp
        from mvus_amd.dist import sharded_handle
        prob, x0 = mp.problem_from_scene(_scene())
        if mode == 'time_lm':
            h, keep = sharded_handle(prob, rank, world, dev, time_x=x0, collective=collective)
            res = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=8)
            # BA -> remove_outliers -> BA on the shards (every rank filters its own slice)
            kept = h.remove_outliers(res.x, 12.0)
            res2 = h.solve(res.x, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=4)
            extra = dict(kept=int(kept.sum()), n_local=int(kept.size), cost2=res2.cost, x2=res2.x)
        else:
            h, keep = sharded_handle(prob, rank, world, dev, collective=collective)
            opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, 5)
            opts.lsmr_maxiter = 4
            res = h.solve(x0, opts=opts, ties='canonical')
            extra = {}
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), x=res.x, cost=res.cost, nfev=res.nfev, njev=res.njev, status=res.status,
                 calls=getattr(h, 'allreduce_stats', {'calls': -1})['calls'], doubles=getattr(h, 'allreduce_stats', {'doubles': -1})['doubles'],
                 n_det=keep.size, **extra)
        h.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('backend', ['gloo', 'nccl', 'rccl'])
@pytest.mark.parametrize('mode', ['time_lm', 'obs_trf'])
def test_two_processes_match_unsharded(tmp_path, backend, mode):
    import torch
    import torch.multiprocessing as tmp
    from mvus_amd import _lib, problem as mp
    from mvus_amd.ba import BAHandle
    if backend in ('nccl', 'rccl') and torch.cuda.device_count() < 2:
        pytest.skip('RCCL needs one device per rank and this box has %d GPU(s); the gloo leg runs the same product code in '
                    'two processes on cuda:0' % torch.cuda.device_count())
    world = 2
    tmp.spawn(_worker, args=(world, _free_port(), backend, mode, str(tmp_path)), nprocs=world, join=True)
    ranks = [dict(np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))) for r in range(world)]
    prob, x0 = mp.problem_from_scene(_scene())
    assert sum(int(r['n_det']) for r in ranks) == prob.M                                     # every detection on exactly one rank
    with BAHandle(prob) as h:
        if mode == 'time_lm':
            ref = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=8)
            keep = h.remove_outliers(ref.x, 12.0)
            ref2 = h.solve(ref.x, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=4)
        else:
            opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, 5)
            opts.lsmr_maxiter = 4
            ref = h.solve(x0, opts=opts, ties='canonical')
    for r in ranks:
        assert (int(r['nfev']), int(r['njev']), int(r['status'])) == (ref.nfev, ref.njev, ref.status)
        np.testing.assert_allclose(float(r['cost']), ref.cost, rtol=1e-9)
        np.testing.assert_allclose(r['x'], ref.x, rtol=0, atol=1e-7 * max(1.0, np.abs(ref.x).max()))
        assert backend == 'rccl' or (int(r['calls']) > 0 and int(r['doubles']) > 0)
    np.testing.assert_array_equal(ranks[0]['x'], ranks[1]['x'])                              # lockstep: the same bits on both ranks
    if mode == 'time_lm':
        assert sum(int(r['kept']) for r in ranks) == int(keep.sum())                         # the same inliers survive, rank by rank
        for r in ranks:
            np.testing.assert_allclose(float(r['cost2']), ref2.cost, rtol=1e-8)
            np.testing.assert_allclose(r['x2'], ref2.x, rtol=0, atol=1e-6 * max(1.0, np.abs(ref2.x).max()))
        # what crossed the wire per linear solve: a few MB, not the cross block (C*B*3N doubles) -- SURVEY 8e
        if backend != 'rccl':                       # (the callback counts; RCCL called by the library does not pass through it)
            per_solve = float(ranks[0]['doubles']) / max(int(ranks[0]['calls']), 1)
            assert per_solve < prob.C * 9 * 3 * int(prob.n_coef.sum())
