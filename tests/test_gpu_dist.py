"""GPU check of the all-reduce hook: RCCL through torch.distributed on buffers aliased from the library's device
pointers (world_size 1 on the single test GPU; the N>1 arithmetic is covered by tests/test_dist_gloo.py)."""
import os
import socket

import numpy as np
import pytest

from golden_util import load_case
from mvus_amd import _lib, problem as mp

pytestmark = pytest.mark.gpu


def test_allreduce_hook_with_rccl_world1():
    import torch
    import torch.distributed as dist
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import sharded_handle
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        scene, g = load_case('rs_F_2int_3cam')
        prob, x0 = mp.problem_from_scene(scene)
        opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, 6)
        opts.lsmr_maxiter = 4
        with BAHandle(prob) as h0:
            r0 = h0.solve(g['x0'], opts=opts)
        h, keep = sharded_handle(prob, 0, 1, 0, group=dist.group.WORLD)
        r1 = h.solve(g['x0'], opts=opts)
        assert h.allreduce_stats['calls'] > 10                      # the hook really ran (J^T u, dot products)
        assert keep.size == prob.M
        np.testing.assert_allclose(r1.cost, r0.cost, rtol=1e-12)
        np.testing.assert_allclose(r1.x, r0.x, rtol=0, atol=1e-10)
        h.close()
    finally:
        dist.destroy_process_group()


def test_rccl_called_by_the_library_world1():
    """mvus_rccl_unique_id / mvus_ba_set_rccl: the library opens librccl.so.1, joins its own communicator (world 1 on these boxes) and
    every sum of both solvers is an ncclAllReduce on the handle's stream -- no callback, no torch.distributed in the iteration."""
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import join_rccl
    scene, g = load_case('rs_F_2int_3cam')
    prob, x0 = mp.problem_from_scene(scene)
    uid = _lib.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, 6)
    opts.lsmr_maxiter = 4
    with BAHandle(prob) as h0, BAHandle(prob) as h:
        ok, why = join_rccl(h, 0, 1)
        assert ok, why
        assert h.time_allreduce(1000, 5) >= 0.0
        r0, r1 = h0.solve(g['x0'], opts=opts), h.solve(g['x0'], opts=opts)
        np.testing.assert_allclose(r1.cost, r0.cost, rtol=1e-12)
        np.testing.assert_allclose(r1.x, r0.x, rtol=0, atol=1e-10)
        l0 = h0.solve(g['x0'], solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=6)
        l1 = h.solve(g['x0'], solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=6)
        np.testing.assert_allclose(l1.cost, l0.cost, rtol=1e-12)
        np.testing.assert_allclose(l1.x, l0.x, rtol=0, atol=1e-10)
    with BAHandle(prob) as h:                      # bad arguments are an error code, not a crash
        with pytest.raises(ValueError):
            h.set_rccl(uid, 3, 2)


@pytest.mark.parametrize('solver', ['trf', 'lm'])
@pytest.mark.parametrize('case', ['rs_F_2int_3cam', 'calib_KE_bounds_3cam'])
def test_two_shards_on_one_gpu_match_unsharded(solver, case):
    """Two observation shards (rank 0 = root owning the motion rows, rank 1) driven by two host threads on the
    one test GPU, with an in-process sum standing in for RCCL: the sharded solve must reproduce the unsharded one."""
    import threading
    import torch
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import _DeviceDoubles
    scene, g = load_case(case)
    prob, x0 = mp.problem_from_scene(scene)
    if solver == 'trf':
        opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, 6)
        opts.lsmr_maxiter = 4
    else:
        opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 5)
    with BAHandle(prob) as h0:
        ref = h0.solve(g['x0'], opts=opts)

    world = 2
    barrier = threading.Barrier(world)
    bufs, total, results, errors = [None] * world, [None], [None] * world, []

    def make_cb(rank):
        def cb(ptr, count, stream):
            t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
            torch.cuda.synchronize()
            bufs[rank] = t
            barrier.wait(60)
            if rank == 0:
                total[0] = bufs[0] + bufs[1]
                torch.cuda.synchronize()
            barrier.wait(60)
            t.copy_(total[0])
            torch.cuda.synchronize()
            barrier.wait(60)
        return cb

    def run(rank):
        try:
            shard, keep = prob.shard(rank, world)
            h = BAHandle(shard, device=0)
            h.set_allreduce(make_cb(rank), is_root=(rank == 0))
            results[rank] = h.solve(g['x0'], opts=opts)
            h.close()
        except Exception as e:                      # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]      # daemon: a failure must not hang the exit
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert not errors, errors
    for r in range(world):
        res = results[r]
        assert (res.nfev, res.njev, res.status) == (ref.nfev, ref.njev, ref.status)
        np.testing.assert_allclose(res.cost, ref.cost, rtol=1e-9)
        np.testing.assert_allclose(res.x, ref.x, rtol=0, atol=1e-7 * max(1.0, np.abs(ref.x).max()))
    np.testing.assert_array_equal(results[0].x, results[1].x)        # ranks stay in lockstep bit for bit


@pytest.mark.parametrize('world,motion', [(2, False), (3, False), (3, True), (4, 'calib'), (2, 'trf')])
def test_time_shards_on_one_gpu_match_unsharded(world, motion):
    """Time shards: every rank holds the detections of one time slice and only that slice of the spline
    blocks; per LM iteration the ranks sum a few small buffers (camera blocks + halo, separator system, Schur
    contributions, step) -- driven here by `world` host threads on the one test GPU with an in-process sum standing
    in for RCCL.  Must reproduce the unsharded LM solve."""
    import threading
    import torch
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import _DeviceDoubles
    if motion == 'calib':       # full parameter vector: intrinsics + distortion free, rs bounded, kinetic-energy prior (wider band)
        sc = synth.make_scene(3, 8000, seed=41, rolling_shutter=True, num_knots=400, distortion=True, opt_calib=True,
                              rs_bounds=True, motion_reg=True, motion_type='KE', motion_weights=20.0)
    else:
        sc = synth.make_scene(3, 6000, seed=41, rolling_shutter=True, num_knots=300, motion_reg=bool(motion), motion_type='F',
                              motion_weights=50.0)
    prob, x0 = mp.problem_from_scene(sc)
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 5)
    if motion == 'trf':         # the parity solver on time shards: any detection subset works for the J v / J^T u sums
        opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, 5)
        opts.lsmr_maxiter = 4
    with BAHandle(prob) as h0:
        ref = h0.solve(x0, opts=opts)

    barrier = threading.Barrier(world)
    bufs, total, results, errors, nbytes = [None] * world, [None], [None] * world, [], [0]

    def make_cb(rank):
        def cb(ptr, count, stream):
            t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
            torch.cuda.synchronize()
            bufs[rank] = t
            barrier.wait(60)
            if rank == 0:
                assert all(b.numel() == bufs[0].numel() for b in bufs)
                total[0] = torch.stack(bufs).sum(0)
                nbytes[0] += 8 * count
                torch.cuda.synchronize()
            barrier.wait(60)
            t.copy_(total[0])
            torch.cuda.synchronize()
            barrier.wait(60)
        return cb

    def run(rank):
        try:
            shard, keep, cuts = prob.shard_time(rank, world, x0)
            h = BAHandle(shard, device=0)
            h.set_time_shard(rank, world, cuts)
            h.set_allreduce(make_cb(rank), is_root=(rank == 0))
            results[rank] = (h.solve(x0, opts=opts), shard.M)
            h.close()
        except Exception as e:                      # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]      # daemon: a failure must not hang the exit
    [t.start() for t in threads]
    [t.join(180) for t in threads]
    assert not errors, errors
    assert sum(m for _, m in results) == prob.M
    for res, _ in results:
        assert (res.nfev, res.njev, res.status) == (ref.nfev, ref.njev, ref.status)
        np.testing.assert_allclose(res.cost, ref.cost, rtol=1e-9)
        np.testing.assert_allclose(res.x, ref.x, rtol=0, atol=1e-7 * max(1.0, np.abs(ref.x).max()))
    for r in range(1, world):
        np.testing.assert_array_equal(results[0][0].x, results[r][0].x)        # ranks stay in lockstep bit for bit
    # the cross block (3N x C(3+P) doubles) is never exchanged: the traffic per linearisation stays far below it
    cross_bytes = 8 * 3 * int(prob.n_coef.sum()) * prob.C * (3 + prob.P)
    if motion != 'trf':
        assert nbytes[0] / max(1, ref.njev + ref.nfev) < 40 * cross_bytes    # small scene: separator rhs dominates


@pytest.mark.parametrize('world,motion', [(3, False), (4, True)])
def test_two_level_separator_elimination_sums_the_cut_separators_only(world, motion, monkeypatch):
    """Round 6: on time shards every rank eliminates its LOCAL separators itself (cyclic reduction on its own sub-chain with the two
    coupling blocks as extra right-hand-side columns) and only the world - 1 cut separators are summed and solved redundantly
    (k_sep2_build / k_sep2_reduce / k_sep2_finish).  Against the one-level form of rounds 3-5 (MVUS_SEP_TWO_LEVEL=0: the WHOLE separator
    system summed, every rank solving all of it): the same solve (cost 1e-10, x 1e-8 -- another elimination order of the same system),
    ranks in lockstep bit for bit, and far fewer bytes through the collective."""
    import threading
    import torch
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import _DeviceDoubles
    sc = synth.make_scene(3, 9000, seed=43, rolling_shutter=True, num_knots=900, motion_reg=bool(motion), motion_type='F', motion_weights=50.0)
    prob, x0 = mp.problem_from_scene(sc)
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 5)
    with BAHandle(prob) as h0:
        ref = h0.solve(x0, opts=opts)

    def run_world(levels):
        monkeypatch.setenv('MVUS_SEP_TWO_LEVEL', '1' if levels == 2 else '0')
        barrier = threading.Barrier(world)
        bufs, total, results, errors, sizes = [None] * world, [None], [None] * world, [], []

        def make_cb(rank):
            def cb(ptr, count, stream):
                t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
                torch.cuda.synchronize()
                bufs[rank] = t
                barrier.wait(60)
                if rank == 0:
                    total[0] = torch.stack(bufs).sum(0)
                    sizes.append(count)
                    torch.cuda.synchronize()
                barrier.wait(60)
                t.copy_(total[0])
                torch.cuda.synchronize()
                barrier.wait(60)
            return cb

        def run(rank):
            try:
                shard, keep, cuts = prob.shard_time(rank, world, x0)
                h = BAHandle(shard, device=0)
                h.set_time_shard(rank, world, cuts)
                h.set_allreduce(make_cb(rank), is_root=(rank == 0))
                results[rank] = h.solve(x0, opts=opts)
                h.close()
            except Exception as e:                      # pragma: no cover
                errors.append(e)
                barrier.abort()

        threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
        [t.start() for t in threads]
        [t.join(180) for t in threads]
        assert not errors, errors
        for r in range(1, world):
            np.testing.assert_array_equal(results[0].x, results[r].x)
        return results[0], sizes

    monkeypatch.delenv('MVUS_NO_SPEC_SHARDS', raising=False)
    two, sizes2 = run_world(2)
    one, sizes1 = run_world(1)
    # the linearisation at the trial point is enqueued speculatively on shards too (its collective with it): no value may change
    monkeypatch.setenv('MVUS_NO_SPEC_SHARDS', '1')
    seq, sizes_seq = run_world(2)
    monkeypatch.delenv('MVUS_NO_SPEC_SHARDS')
    assert (two.nfev, two.njev) == (seq.nfev, seq.njev)
    if motion:      # (time shards add the motion rows with fp64 atomics -- k_assemble_motion: last-bit differences from run to run)
        np.testing.assert_allclose(two.cost, seq.cost, rtol=1e-11)
        np.testing.assert_allclose(two.x, seq.x, rtol=0, atol=1e-9 * max(1.0, np.abs(seq.x).max()))
    else:
        assert np.array_equal(two.x, seq.x) and two.cost == seq.cost
    assert len(sizes2) >= len(sizes_seq)          # (a speculative assembly after the last trial, or after a rejected one, is an extra sum)
    for res in (two, one):
        assert (res.nfev, res.njev, res.status) == (ref.nfev, ref.njev, ref.status)
        np.testing.assert_allclose(res.cost, ref.cost, rtol=1e-9)
        np.testing.assert_allclose(res.x, ref.x, rtol=0, atol=1e-7 * max(1.0, np.abs(ref.x).max()))
    np.testing.assert_allclose(two.cost, one.cost, rtol=1e-10)
    np.testing.assert_allclose(two.x, one.x, rtol=0, atol=1e-8 * max(1.0, np.abs(one.x).max()))
    # the separator sum is the largest collective of an iteration in the one-level form; with two levels it carries world - 1 separators
    s3 = 3 * (5 if motion else 3)
    ncols = prob.C * (3 + prob.P) + 1
    cut = (world - 1) * (2 * s3 * s3 + s3 * ncols)
    assert cut in sizes2 and cut not in sizes1
    assert max(sizes2) < max(sizes1) and sum(sizes2) < 0.6 * sum(sizes1), (sum(sizes2), sum(sizes1))
    print('world %d: doubles through the collective per solve: two-level %d, one-level %d (largest buffer %d against %d)'
          % (world, sum(sizes2), sum(sizes1), max(sizes2), max(sizes1)))


def test_time_shard_reports_detections_that_leave_its_slice():
    """Cuts made at x0, solve started from time shifts 60 frames away: rank 0's detections now reach control points its
    slice does not hold -- the solve must fail loudly (no silent dropping of rows)."""
    import threading
    import torch
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import _DeviceDoubles
    sc = synth.make_scene(3, 6000, seed=41, rolling_shutter=True, num_knots=300)
    prob, x0 = mp.problem_from_scene(sc)
    x_bad = x0.copy()
    x_bad[prob.C:2 * prob.C] += 60.0
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 3)
    world = 2
    barrier = threading.Barrier(world)
    bufs, total, errors = [None] * world, [None], []

    def make_cb(rank):
        def cb(ptr, count, stream):
            t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
            torch.cuda.synchronize()
            bufs[rank] = t
            barrier.wait(30)
            if rank == 0:
                total[0] = bufs[0] + bufs[1]
                torch.cuda.synchronize()
            barrier.wait(30)
            t.copy_(total[0])
            torch.cuda.synchronize()
            barrier.wait(30)
        return cb

    def run(rank):
        try:
            shard, keep, cuts = prob.shard_time(rank, world, x0)
            h = BAHandle(shard, device=0)
            h.set_time_shard(rank, world, cuts)
            h.set_allreduce(make_cb(rank), is_root=(rank == 0))
            try:
                h.solve(x_bad, opts=opts)
            finally:
                h.close()
        except Exception as e:
            errors.append(str(e))
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]      # daemon: a failure must not hang the exit
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert any('outside this rank' in e for e in errors), errors


@pytest.mark.parametrize('motion', [False, True])
def test_time_shard_every_cut_position(motion):
    """The closing interior of a rank's chain takes every length as the cut moves (a control point at a time over more
    than one partition period): the sharded damped step must equal the unsharded one for all of them -- an interior
    shorter than the band half-width between two separators would let them couple directly."""
    import threading
    import torch
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import _DeviceDoubles
    sc = synth.make_scene(3, 5000, seed=53, rolling_shutter=True, num_knots=260, motion_reg=motion, motion_type='F',
                          motion_weights=30.0)          # the motion prior widens the band: separators of 5 control points
    prob, x0 = mp.problem_from_scene(sc)
    N = int(prob.n_coef.sum())
    with BAHandle(prob) as h0:
        h0.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        p_ref = h0.lm_step(0.3)
    world = 2
    spans = prob.detection_spans(x0)
    worst = 0.0
    for cut in range(N // 2 - 20, N // 2 + 20):
        cuts = np.array([0, cut, N], dtype=np.int32)
        barrier = threading.Barrier(world)
        bufs, total, steps, errors = [None] * world, [None], [None] * world, []

        def make_cb(rank):
            def cb(ptr, count, stream):
                t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
                torch.cuda.synchronize()
                bufs[rank] = t
                barrier.wait(60)
                if rank == 0:
                    total[0] = bufs[0] + bufs[1]
                    torch.cuda.synchronize()
                barrier.wait(60)
                t.copy_(total[0])
                torch.cuda.synchronize()
                barrier.wait(60)
            return cb

        def run(rank):
            try:
                shard, keep, _ = prob.shard_time(rank, world, x0, cuts=cuts)
                h = BAHandle(shard, device=0)
                h.set_time_shard(rank, world, cuts)
                h.set_allreduce(make_cb(rank), is_root=(rank == 0))
                h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
                steps[rank] = h.lm_step(0.3)
                h.close()
            except Exception as e:                      # pragma: no cover
                errors.append(e)
                barrier.abort()

        threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
        [t.start() for t in threads]
        [t.join(120) for t in threads]
        assert not errors, (cut, errors)
        for p in steps:
            worst = max(worst, float(np.abs(p - p_ref).max()))
    assert worst < 1e-9 * np.abs(p_ref).max(), worst


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_time_shards_randomised_layouts(seed):
    """Three time shards of scenes with locally shuffled detections, dropped stretches and detections outside every
    interval: the damped step of the sharded chain equals the unsharded one."""
    import threading
    import torch
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import _DeviceDoubles
    rng = np.random.default_rng(300 + seed)
    sc = synth.make_scene(3, 6000, seed=400 + seed, rolling_shutter=True, num_knots=int(rng.choice([200, 500])))
    for c in range(3):
        d = sc.detections[c]
        n = d.shape[1]
        if (c + seed) % 3 == 0:
            d = d[:, np.concatenate([b + rng.permutation(min(16, n - b)) for b in range(0, n, 16)])]
        elif (c + seed) % 3 == 1:
            keep = np.ones(n, bool)
            for _ in range(4):
                a = rng.integers(0, n)
                keep[a:a + rng.integers(1, n // 8)] = False
            d = d[:, keep].copy()
            d[0, 100:110] += 1e6
        sc.detections[c] = d
    prob, x0 = mp.problem_from_scene(sc)
    with BAHandle(prob) as h0:
        h0.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        p_ref = h0.lm_step(0.1)
    world = 3
    barrier = threading.Barrier(world)
    bufs, total, steps, errors = [None] * world, [None], [None] * world, []

    def make_cb(rank):
        def cb(ptr, count, stream):
            t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
            torch.cuda.synchronize()
            bufs[rank] = t
            barrier.wait(60)
            if rank == 0:
                total[0] = torch.stack(bufs).sum(0)
                torch.cuda.synchronize()
            barrier.wait(60)
            t.copy_(total[0])
            torch.cuda.synchronize()
            barrier.wait(60)
        return cb

    def run(rank):
        try:
            shard, keep, cuts = prob.shard_time(rank, world, x0)
            h = BAHandle(shard, device=0)
            h.set_time_shard(rank, world, cuts)
            h.set_allreduce(make_cb(rank), is_root=(rank == 0))
            h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
            steps[rank] = h.lm_step(0.1)
            h.close()
        except Exception as e:                      # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert not errors, errors
    for p in steps:
        np.testing.assert_allclose(p, p_ref, rtol=0, atol=1e-9 * np.abs(p_ref).max())


def test_ba_outliers_ba_on_time_shards():
    """The BA -> remove_outliers -> BA sequence of main.py:49-62 with the detections resident on two time shards: every rank
    filters its own slice in place; masks, detection counts and the second BA equal the unsharded sequence."""
    import threading
    import torch
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import _DeviceDoubles
    sc = synth.make_scene(3, 6000, seed=61, rolling_shutter=True, num_knots=300)
    prob, x0 = mp.problem_from_scene(sc)
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 6)

    def sequence(h):
        r1 = h.solve(x0, opts=opts)
        keep = h.remove_outliers(r1.x, 10.0)
        r2 = h.solve(r1.x, opts=opts)
        return r1, keep, r2

    with BAHandle(prob) as h0:
        ref1, keep_ref, ref2 = sequence(h0)
    assert 0 < (~keep_ref).sum() < prob.M // 4
    world = 2
    barrier = threading.Barrier(world)
    bufs, total, results, errors = [None] * world, [None], [None] * world, []

    def make_cb(rank):
        def cb(ptr, count, stream):
            t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
            torch.cuda.synchronize()
            bufs[rank] = t
            barrier.wait(60)
            if rank == 0:
                total[0] = bufs[0] + bufs[1]
                torch.cuda.synchronize()
            barrier.wait(60)
            t.copy_(total[0])
            torch.cuda.synchronize()
            barrier.wait(60)
        return cb

    def run(rank):
        try:
            shard, idx, cuts = prob.shard_time(rank, world, x0)
            h = BAHandle(shard, device=0)
            h.set_time_shard(rank, world, cuts)
            h.set_allreduce(make_cb(rank), is_root=(rank == 0))
            results[rank] = sequence(h) + (idx,)
            h.close()
        except Exception as e:                      # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in threads]
    [t.join(180) for t in threads]
    assert not errors, errors
    keep_all = np.zeros(prob.M, dtype=bool)
    for r1, keep, r2, idx in results:
        keep_all[idx] = keep
        np.testing.assert_allclose(r1.cost, ref1.cost, rtol=1e-9)
        np.testing.assert_allclose(r2.cost, ref2.cost, rtol=1e-9)
        np.testing.assert_allclose(r2.x, ref2.x, rtol=0, atol=1e-7 * max(1.0, np.abs(ref2.x).max()))
    np.testing.assert_array_equal(keep_all, keep_ref)


def test_time_shards_are_cut_again_when_the_time_stamps_drift():
    """The 60-frame shift of the test above, through mvus_amd.dist.solve_time_sharded: MVUS_E_RESHARD hands the point reached so far
    back on both ranks, the timeline is cut again THERE, and the sharded solve ends where the unsharded solve from the same start ends
    (the reference re-evaluates visibility at every call: common.py:317, tools/util.py:90-116)."""
    import threading
    import torch
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import _DeviceDoubles, solve_time_sharded
    sc = synth.make_scene(3, 6000, seed=41, rolling_shutter=True, num_knots=300)
    prob, x0 = mp.problem_from_scene(sc)
    x_bad = x0.copy()
    x_bad[prob.C:2 * prob.C] += 60.0
    kw = dict(solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC)
    with BAHandle(prob) as h:
        ref = h.solve(x_bad, max_nfev=8, **kw)
    world = 2
    barrier = threading.Barrier(world)
    bufs, total, errors, out = [None] * world, [None], [], [None] * world

    def make_cb(rank):
        def cb(ptr, count, stream):
            t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
            torch.cuda.synchronize()
            bufs[rank] = t
            barrier.wait(30)
            if rank == 0:
                total[0] = bufs[0] + bufs[1]
                torch.cuda.synchronize()
            barrier.wait(30)
            t.copy_(total[0])
            torch.cuda.synchronize()
            barrier.wait(30)
        return cb

    def run(rank):
        try:
            first = [True]

            def make_handle(x):
                # the FIRST cuts are made at x0 (where the time stamps were when the job was set up), later ones at the point handed back
                shard, keep, cuts = prob.shard_time(rank, world, x0 if first[0] else x)
                first[0] = False
                h = BAHandle(shard, device=0)
                h.set_time_shard(rank, world, cuts)
                h.set_allreduce(make_cb(rank), is_root=(rank == 0))
                return h
            out[rank] = solve_time_sharded(make_handle, x_bad, max_nfev=8, **kw)
        except Exception as e:
            import traceback
            errors.append(traceback.format_exc())
            barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts: t.start()
    for t in ts: t.join(180)
    assert not errors, errors[0]
    r0, r1 = out
    assert r0.recuts == r1.recuts == 1                                            # cut once more, at the start point
    assert np.array_equal(r0.x, r1.x)
    np.testing.assert_allclose(r0.cost, ref.cost, rtol=1e-9)
    np.testing.assert_allclose(r0.x, ref.x, rtol=0, atol=1e-6 * max(1.0, np.abs(ref.x).max()))


@pytest.mark.parametrize('what', ['pivot', 'handover', 'reshard', 'span'])
def test_failure_flags_of_one_rank_reach_every_rank_with_the_step(what):
    """The two failure flags of a solve travel behind the step through its sum over the ranks (packed by k_back_substitute, read where the
    sum lands by the trial kernel and -- with the trial's scalars -- by the host; round 6: no pack / unpack launches).  A flag raised on ONE
    rank must be every rank's: here the sum of the first solve's step buffer gets a flag added as if rank 1 had raised it
      pivot   (slot 0): a lost pivot -> both ranks take no step, raise the damping together and go on in lockstep, bit for bit;
      handover (slot 0, travels as 2^20): a hand-over time-out inside the reduced solve -> both ranks repeat the solve at the SAME damping on
              the separate-launch route, whose results are the in-launch route's bits: the solve ends where an undisturbed one ends;
      reshard (slot 1, bit 1): both ranks hand the point back with MVUS_E_RESHARD;
      span    (slot 1, bit 2 travels as 4096): both ranks report the internal error -- none waits for the other in the next collective."""
    import threading
    import torch
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle, ReshardNeeded
    from mvus_amd.dist import _DeviceDoubles
    world = 2
    sc = synth.make_scene(3, 6000, seed=41, rolling_shutter=True, num_knots=300)
    prob, x0 = mp.problem_from_scene(sc)
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 5)
    n = prob.n_params
    slot, value = {'pivot': (0, 1.0), 'handover': (0, 1048576.0), 'reshard': (1, 1.0), 'span': (1, 4096.0)}[what]

    def sharded_run(inject):
        barrier = threading.Barrier(world)
        bufs, total, results, errors, injected, steps = [None] * world, [None], [None] * world, [None] * world, [0 if inject else 1], [0]
        return _flag_run(barrier, bufs, total, results, errors, injected, steps)

    def _flag_run(barrier, bufs, total, results, errors, injected, steps):

        def make_cb(rank):
            def cb(ptr, count, stream):
                t = torch.as_tensor(_DeviceDoubles(ptr, count), device='cuda:0')
                torch.cuda.synchronize()
                bufs[rank] = t
                barrier.wait(60)
                if rank == 0:
                    total[0] = torch.stack(bufs).sum(0)
                    if count == n + 2:
                        steps[0] += 1
                        if not injected[0]:
                            total[0][n + slot] += value
                            injected[0] = 1
                    torch.cuda.synchronize()
                barrier.wait(60)
                t.copy_(total[0])
                torch.cuda.synchronize()
                barrier.wait(60)
            return cb

        def run(rank):
            try:
                shard, keep, cuts = prob.shard_time(rank, world, x0)
                h = BAHandle(shard, device=0)
                h.set_time_shard(rank, world, cuts)
                h.set_allreduce(make_cb(rank), is_root=(rank == 0))
                try:
                    results[rank] = h.solve(x0, opts=opts)
                except Exception as e:
                    errors[rank] = e
                h.close()
            except Exception as e:                      # pragma: no cover
                errors[rank] = e
                barrier.abort()

        threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
        [t.start() for t in threads]
        [t.join(180) for t in threads]
        assert not any(t.is_alive() for t in threads)          # nobody is left waiting in a collective
        return results, errors, injected, steps

    results, errors, injected, steps = sharded_run(True)
    assert injected[0] == 1
    if what == 'pivot':
        assert errors == [None, None], errors
        a, b = results
        np.testing.assert_array_equal(a.x, b.x)
        assert (a.nfev, a.njev, a.status) == (b.nfev, b.njev, b.status) and a.cost == b.cost
        assert steps[0] >= a.nfev                          # the failed solve was repeated at a larger damping: one step sum more than trials
        with BAHandle(prob) as h0:
            ref = h0.solve(x0, opts=opts)
        assert a.cost < ref.initial_cost and abs(a.cost - ref.cost) < 0.05 * ref.cost      # (another damping sequence, the same basin)
    elif what == 'handover':
        assert errors == [None, None], errors
        plain, perr, _, psteps = sharded_run(False)
        assert perr == [None, None], perr
        for r in range(world):
            np.testing.assert_array_equal(results[r].x, plain[r].x)
            assert (results[r].nfev, results[r].njev, results[r].cost) == (plain[r].nfev, plain[r].njev, plain[r].cost)
        assert steps[0] == psteps[0] + 1                   # one solve was repeated, nothing else changed
    elif what == 'reshard':
        assert all(isinstance(e, ReshardNeeded) for e in errors), errors
    else:
        assert all(e is not None and 'span table' in str(e) for e in errors), errors
