"""world_size-2 CPU coverage of the observation-sharded path (gloo): shards + Reducer reproduce the
unsharded normal equations, gradient and cost; motion rows are counted once."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as tmp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, out_dir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from golden_util import load_case
        from hostcheck_util import HostHandle
        from mvus_amd import _lib, problem as mp, sharding
        scene, g = load_case(case)
        prob, x0 = mp.problem_from_scene(scene)
        shard, keep = prob.shard(rank, world)
        if rank != 0:
            shard.motion_reg = False                      # replicated rows are owned by rank 0 (mvus_amd/dist.py)
        h = HostHandle(shard)
        red = sharding.Reducer()
        assert red.active
        x = g['x0'] + g['delta']
        f, J = h.dense_jacobian(x, _lib.JAC_ANALYTIC)
        packed = torch.from_numpy(np.concatenate(([0.5 * f @ f, float(h.m)], J.T @ f, (J.T @ J).ravel())))
        red.all_reduce_(packed)
        assert red.calls == 1
        if rank == 0:
            np.save(os.path.join(out_dir, 'reduced.npy'), packed.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', ['rs_F_2int_3cam', 'calib_KE_bounds_3cam'])
def test_sharded_normal_equations_match_unsharded(tmp_path, case):
    import hostcheck_util
    hostcheck_util.load()                                  # build once, before forking workers
    world = 2
    tmp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), 'reduced.npy'))
    from golden_util import load_case
    from hostcheck_util import HostHandle
    from mvus_amd import _lib, problem as mp
    scene, g = load_case(case)
    prob, _ = mp.problem_from_scene(scene)
    h = HostHandle(prob)
    f, J = h.dense_jacobian(g['x0'] + g['delta'], _lib.JAC_ANALYTIC)
    n = h.n
    want = np.concatenate(([0.5 * f @ f, float(h.m)], J.T @ f, (J.T @ J).ravel()))
    assert got[1] == want[1]                               # global row count: motion rows counted once
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12)
    scale = np.abs(want[2:]).max()
    np.testing.assert_allclose(got[2:], want[2:], rtol=0, atol=1e-10 * scale)


def test_reducer_is_identity_without_process_group():
    from mvus_amd import sharding
    red = sharding.Reducer()
    t = torch.arange(4, dtype=torch.float64)
    assert not red.active
    assert torch.equal(red.all_reduce_(t.clone()), t)
