"""world_size-2 CPU coverage (gloo) of what the multi-GPU path relies on, with the host build of the device math standing in
for the kernels:

  * observation shards (`BAProblem.shard`): the packed per-rank sums [cost | rows | J^T f | J^T J] added over the ranks are the
    unsharded ones, the replicated motion rows counted once;
  * time shards (`BAProblem.shard_time`, what bench.py / mvus_amd.dist use with the LM solver): the same, PLUS the locality the
    time-shard design of SURVEY 8e rests on -- a rank's rows touch only control points of its own slice +- halo, so the big cross
    block never has to move -- and a whole Levenberg-Marquardt solve driven in lockstep by the two processes (every iteration:
    local linearisation, ONE all-reduce of the packed normal equations, identical damped solve on both ranks, local trial
    residual, one all-reduce of the trial cost) ending where the unsharded solve ends.
The GPU-side counterpart with the product's own packed buffers is tests/test_gpu_multiproc.py."""
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


def _setup(rank, world, port):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _sum(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


def _worker(rank, world, port, case, out_dir):
    _setup(rank, world, port)
    try:
        from golden_util import load_case
        from hostcheck_util import HostHandle
        from mvus_amd import _lib, problem as mp
        scene, g = load_case(case)
        prob, x0 = mp.problem_from_scene(scene)
        shard, keep = prob.shard(rank, world)
        if rank != 0:
            shard.motion_reg = False                      # replicated rows are owned by rank 0 (mvus_amd/dist.py)
        h = HostHandle(shard)
        x = g['x0'] + g['delta']
        f, J = h.dense_jacobian(x, _lib.JAC_ANALYTIC)
        packed = _sum(np.concatenate(([0.5 * f @ f, float(h.m)], J.T @ f, (J.T @ J).ravel())))
        if rank == 0:
            np.save(os.path.join(out_dir, 'reduced.npy'), packed)
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
    want = np.concatenate(([0.5 * f @ f, float(h.m)], J.T @ f, (J.T @ J).ravel()))
    assert got[1] == want[1]                               # global row count: motion rows counted once
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12)
    scale = np.abs(want[2:]).max()
    np.testing.assert_allclose(got[2:], want[2:], rtol=0, atol=1e-10 * scale)


# ---- time shards -------------------------------------------------------------------------------------------------------
HALO = 8


def _time_scene():
    from mvus_amd import synth
    return synth.make_scene(3, 1500, seed=47, rolling_shutter=True, num_knots=80)


def _lm(n, linearize, trial_cost, x, max_nfev=12):
    """Marquardt-scaled LM with Nielsen's update (the driver of csrc/ba_schur.h restated in numpy).  `linearize(x)` ->
    (cost, g, H) and `trial_cost(x)` -> cost are the places where a sharded run exchanges data."""
    cost, g, H = linearize(x)
    lam, nu, nfev = 1e-4, 2.0, 1
    while nfev < max_nfev:
        D = np.where(np.diag(H) > 0, np.diag(H), 1.0)
        p = -np.linalg.solve(H + lam * np.diag(D), g)
        c_new = trial_cost(x + p)
        nfev += 1
        pred = 0.5 * (lam * float(p @ (D * p)) - float(g @ p))
        if c_new < cost:
            ratio = (cost - c_new) / pred if pred > 0 else 1.0
            x = x + p
            lam *= max(1.0 / 3.0, 1.0 - (2.0 * ratio - 1.0) ** 3)
            nu = 2.0
            if nfev < max_nfev:
                cost, g, H = linearize(x)
            else:
                cost = c_new
        else:
            lam *= nu
            nu *= 2.0
    return x, cost


def _time_worker(rank, world, port, out_dir):
    _setup(rank, world, port)
    try:
        from hostcheck_util import HostHandle
        from mvus_amd import _lib, problem as mp
        prob, x0 = mp.problem_from_scene(_time_scene())
        shard, keep, cuts = prob.shard_time(rank, world, x0, HALO)
        h = HostHandle(shard)
        n = h.n
        N = int(prob.n_coef.sum())
        # locality: this rank's rows reach only the control points of its own slice +- halo
        f, J = h.dense_jacobian(x0, _lib.JAC_ANALYTIC)
        first_spline_col = prob.C * (3 + prob.P)
        touched_cols = np.nonzero(np.abs(J[:, first_spline_col:]).sum(axis=0))[0]
        ctrl = touched_cols % N                        # one spline: column = first + d * N + control point
        assert ctrl.min() >= cuts[rank] - HALO and ctrl.max() < cuts[rank + 1] + HALO
        n_det = _sum(np.array([float(keep.size)]))[0]
        assert int(n_det) == prob.M                    # every detection lives on exactly one rank

        exchanged = []

        def linearize(x):
            f, J = h.dense_jacobian(x, _lib.JAC_ANALYTIC)
            packed = _sum(np.concatenate(([0.5 * f @ f], J.T @ f, (J.T @ J).ravel())))      # ONE all-reduce per linearisation
            exchanged.append(packed.size)
            return packed[0], packed[1:1 + n], packed[1 + n:].reshape(n, n)

        def trial_cost(x):
            f = h.residual(x)
            return _sum(np.array([0.5 * f @ f]))[0]

        x, cost = _lm(n, linearize, trial_cost, x0.copy())
        np.savez(os.path.join(out_dir, 'time_rank%d.npz' % rank), x=x, cost=cost, cuts=cuts, n_local=keep.size)
    finally:
        dist.destroy_process_group()


def test_time_sharded_lm_in_two_processes_matches_unsharded(tmp_path):
    import hostcheck_util
    hostcheck_util.load()
    world = 2
    tmp.spawn(_time_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(os.path.join(str(tmp_path), 'time_rank%d.npz' % k))) for k in range(world)]
    np.testing.assert_array_equal(r[0]['x'], r[1]['x'])                     # lockstep: identical decisions, identical bits
    assert min(int(r[0]['n_local']), int(r[1]['n_local'])) > 0.3 * (int(r[0]['n_local']) + int(r[1]['n_local']))   # balanced cut
    from hostcheck_util import HostHandle
    from mvus_amd import _lib, problem as mp
    prob, x0 = mp.problem_from_scene(_time_scene())
    h = HostHandle(prob)

    def linearize(x):
        f, J = h.dense_jacobian(x, _lib.JAC_ANALYTIC)
        return 0.5 * f @ f, J.T @ f, J.T @ J

    x_ref, cost_ref = _lm(h.n, linearize, lambda x: 0.5 * float(h.residual(x) @ h.residual(x)), x0.copy())
    assert cost_ref < 0.95 * 0.5 * float(h.residual(x0) @ h.residual(x0))  # the solve did something (2 % gross outliers carry most of the cost)
    np.testing.assert_allclose(float(r[0]['cost']), cost_ref, rtol=1e-9)
    np.testing.assert_allclose(r[0]['x'], x_ref, rtol=0, atol=1e-7 * max(1.0, np.abs(x_ref).max()))


def _route_worker(rank, world, port, fail_at, out_dir):
    """agree_on_rccl with injected failures: get_id raises on rank 0 ('id'), join raises on ONE rank ('join0' / 'join1'), or nothing
    fails ('none').  Whatever happens locally every rank must come out with the same decision, a rank that did join must have left
    again, and the collective that follows (here: the callback route's all_reduce) must complete on both ranks."""
    _setup(rank, world, port)
    try:
        from mvus_amd.dist import agree_on_rccl
        state = {'joined': False, 'left': False}

        def get_id():
            if fail_at == 'id':
                raise OSError('librccl.so.1 cannot be opened (injected)')
            return bytes(range(128))

        def join(uid):
            assert uid == bytes(range(128))
            if fail_at == 'join%d' % rank:
                raise ValueError('ncclCommInitRank failed (injected)')
            state['joined'] = True

        def leave():
            state['left'] = True

        def probe():                                   # (mvus_rccl_available: this rank cannot even open the library)
            if fail_at == 'probe%d' % rank:
                raise OSError('librccl.so.1 cannot be opened on this rank (injected)')

        ok, why = agree_on_rccl(rank, world, None, get_id, join, leave, probe=probe)
        total = _sum(np.array([1.0 + rank]))          # the route both ranks fell back to: must not hang
        np.save(os.path.join(out_dir, 'route%d.npy' % rank),
                np.array([float(ok), float(state['joined']), float(state['left']), float(total[0])]))
        with open(os.path.join(out_dir, 'why%d.txt' % rank), 'w') as f:
            f.write(why)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize('fail_at', ['none', 'id', 'join0', 'join1', 'probe1'])
def test_rccl_route_is_a_collective_decision(tmp_path, fail_at):
    """collective='auto' (mvus_amd/dist.py): a librccl the library cannot open on rank 0 only, or a communicator that fails to
    initialise on one rank only, must send BOTH ranks to the callback route -- not one rank into its first all_reduce while the
    other one still sits in the id broadcast (the round-4 advisor's finding)."""
    world, port = 2, _free_port()
    tmp.spawn(_route_worker, args=(world, port, fail_at, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ('route%d.npy' % k)) for k in range(world)]
    why = [(tmp_path / ('why%d.txt' % k)).read_text() for k in range(world)]
    assert r[0][0] == r[1][0] == (1.0 if fail_at == 'none' else 0.0)        # the same decision on both ranks
    assert why[0] == why[1]
    assert r[0][3] == r[1][3] == 3.0                                            # the next collective completed on both
    for k in range(world):
        joined, left = r[k][1], r[k][2]
        if fail_at == 'none':
            assert joined == 1.0 and left == 0.0
        elif fail_at == 'id':
            assert joined == 0.0 and left == 0.0 and 'rank 0 could not get an RCCL id' in why[k]
        elif fail_at.startswith('probe'):
            # a rank that cannot open RCCL at all is found out BEFORE anybody enters the (blocking, collective) join: nobody joined
            assert joined == 0.0 and left == 0.0 and 'rank 1 cannot open RCCL' in why[k]
        else:
            failing = int(fail_at[-1])
            assert joined == (0.0 if k == failing else 1.0) and left == joined        # who joined has left again
            assert 'rank %d' % failing in why[k]
