#!/usr/bin/env python3
"""bench.py -- BA iterations on synthetic N-camera x M-observation scenes, 1..8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config I] [--solver trf|lm]

A *step* is one bundle-adjustment iteration = one trust-region trial of ``Scene.BA``:
residual + Jacobian at x, one linear solve for the step, residual at the trial point
(``mvus_ba_solve`` with ``max_nfev = 2``; comparable to one scipy ``nfev``).  ``value`` is the whole-job
throughput in residuals/s -- one residual = one observation's (x, y) pair going through one BA
iteration -- with every input already resident in HBM (the handle is created before the timed region;
per step only the n-vector x crosses PCIe).  BA iterations/s is reported beside it.

Workload at N=1: BASELINE.json configs[2] ("synthetic 32 cams x 500k obs, RS on, ~5k spline knots"), the
configuration the north star quotes its HBM target on; configs[1] (7 cams x 100k) via ``--config 1``.
For N>1 the observation count grows with N (weak scaling: ~500k observations per GPU; cameras and
spline knots fixed, so the timeline gets denser).  The LM solver shards by TIME: every rank holds the
detections of one time slice and that slice of the spline blocks, and the ranks sum a few MB per iteration (camera
blocks + halo, separator system, Schur contributions, step) over RCCL; ``--shard obs`` (always used by ``--solver trf``)
cuts every camera's detections into N pieces instead, with one all-reduce per J^T u / dot product / normal-equation set.

The JSON line also carries the roofline of the dominant kernel (residual+Jacobian), measured with
HIP events on the kernel's own stream, and a CPU baseline (the oracle's restatement of the scipy path,
one core) on a bounded sample, on rank 0 at N=1 only.  ``long_solve`` (LM): the same number of trials inside ONE call, measured after
the timed region -- what an iteration costs inside a long Scene.BA (no residual at the start of each call, no Python between trials).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(prob):
    """per observation read frame,v_raw,u_obs,v_obs (32 B; 24 B + in-kernel undistortion with
    opt_calib), write 2 residuals (16 B), 2*NS Jacobian slots and one int32 span; once per launch the
    parameter vector, the knots and the per-camera constants."""
    NS = 3 + prob.P + 12
    per_obs = (24 if prob.opt_calib else 32) + 16 + 2 * NS * 8 + 4
    once = (prob.n_params + prob.knots.size + 10 * prob.C) * 8
    return per_obs, once


def cpu_baseline(config_index, full_scene=None, full_ba=False):
    """The reference's CPU path restated (oracle: numpy residual + scipy least_squares with the sparsity
    pattern, exactly the call of common.py:670), single core, on a 1/32-scale sample of the workload; plus -- the anchor for that
    sample -- ONE evaluation of the oracle's residual (error_BA) on the full-size workload."""
    import numpy as np
    from mvus_amd import synth
    from oracle import ba_oracle as orc
    kw = dict(synth.BASELINE_CONFIGS[config_index])
    scale = 32 if kw['total_obs'] >= 200_000 else 8
    kw['total_obs'] = max(kw['total_obs'] // scale, 2000)
    if kw.get('num_knots'):
        kw['num_knots'] = max(kw['num_knots'] // scale, 16)
    sc = synth.make_scene(**kw)
    oprob, x0 = orc.problem_from_scene(sc)
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):                  # `cores: 1` must be true: no BLAS / OpenMP thread pools
        t0 = time.perf_counter()
        res = orc.solve(oprob, x0, max_iter=48)        # ~15 s of single-core work at the 1/32 sample
        dt = time.perf_counter() - t0
    M = sum(d.shape[1] for d in oprob.detections)
    iters = max(res.nfev - 1, 1)
    full = None
    if full_scene is not None:
        fprob, fx0 = orc.problem_from_scene(full_scene)
        with threadpool_limits(limits=1):
            orc.residual(fprob, fx0)                      # (first call: page faults, imports)
            t1 = time.perf_counter()
            orc.residual(fprob, fx0)
            dtf = time.perf_counter() - t1
        Mf = sum(d.shape[1] for d in fprob.detections)
        full_ba_rec = None
        if full_ba:                                       # the reference's BA on the FULL workload, 10 evaluations, one core (the default up to 600k observations)
            with threadpool_limits(limits=1):
                t2 = time.perf_counter()
                rfull = orc.solve(fprob, fx0, max_iter=10)
                dtb = time.perf_counter() - t2
            itb = max(rfull.nfev - 1, 1)
            full_ba_rec = {'value': Mf * itb / dtb, 'unit': 'residuals/s', 'ba_iters_per_s': itb / dtb, 'seconds': dtb, 'nfev': int(rfull.nfev),
                           'cost_last': float(rfull.cost),
                           'what': 'oracle restatement of Scene.BA (scipy least_squares, jac_sparsity, lsmr, 2-point FD) on the full-size workload, max_nfev = 10, 1 core'}
        full = {'value': Mf / dtf, 'unit': 'residuals/s', 'seconds_per_evaluation': dtf, 'obs': Mf, 'full_size_ba': full_ba_rec,
                'what': 'one evaluation of the oracle residual (error_BA restated, numpy, 1 core) on the FULL workload; a BA iteration of '
                        'the reference costs (column groups + 1) such evaluations for its 2-point Jacobian plus the LSMR solve'}
    import numpy, scipy
    blas = [(d.get('internal_api'), d.get('version'), d.get('num_threads')) for d in __import__('threadpoolctl').threadpool_info()]
    return {'value': M * iters / dt, 'unit': 'residuals/s', 'cores': 1, 'kind': 'port',
            'ba_iters_per_s': iters / dt, 'os_cpu_count': os.cpu_count(), 'numpy': numpy.__version__, 'scipy': scipy.__version__,
            'blas': blas, 'blas_threads_during_timing': 1, 'residual_only_full_size': full,
            'sample': '%d cams x %d obs (%d params), 1/%d-scale sample of the workload from the same generator; '
                      'oracle restatement of Scene.BA: scipy least_squares(jac_sparsity, lsmr, 2-point FD), '
                      '%d trial steps in %.1f s' % (oprob.C, M, x0.size, scale, iters, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-long-solve', action='store_true', help='skip the one-call leg (long_solve in the JSON line)')
    ap.add_argument('--config', type=int, default=2, help='index into BASELINE.json configs (2 = 32 cams x 500k obs)')
    ap.add_argument('--solver', choices=['trf', 'lm'], default=os.environ.get('MVUS_BENCH_SOLVER', 'lm'))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-full', action='store_true', help='N=1: time the full-size 10-evaluation BA of the oracle (scipy path, one core) even above 600k observations (minutes there)')
    ap.add_argument('--no-cpu-full', action='store_true', help='N=1: skip the full-size 10-evaluation BA of the oracle (about 50 s at configs[2]; on by default up to 600k observations '
                                                                '-> cpu_baseline.full_size_ba)')
    ap.add_argument('--no-parity-solver', action='store_true', help='skip the extra timing of the scipy-TRF+LSMR restatement')
    ap.add_argument('--no-strong-config3', action='store_true', help='N>1: skip the strong-scaling sub-record on BASELINE configs[3]')
    ap.add_argument('--obs', type=int, default=None, help='override the detection count of the config (kernel studies at other sizes; the workload string says so)')
    ap.add_argument('--shard', choices=['time', 'obs'], default=None, help='N>1: how observations are cut over the ranks (default: time for lm, obs for trf)')
    ap.add_argument('--collective', choices=['rccl', 'torch', 'auto'], default=None,
                    help='N>1: rccl = ncclAllReduce called by the library on its own communicator; auto (default on a multi-GPU node) = rccl, or the callback if the library cannot open RCCL; '
                         'torch = the callback through torch.distributed (the only route of the one-device gloo flow test)')
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d' % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the BA hot path has no CPU fallback)')
    # MVUS_BENCH_ONE_DEVICE=1 (flow test on a 1-GPU box): every rank uses cuda:0 and gloo carries the sums
    one_device = os.environ.get('MVUS_BENCH_ONE_DEVICE') == '1'
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_device:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    from mvus_amd import problem as mp, synth
    from mvus_amd import ba
    from mvus_amd.dist import sharded_handle

    kw = dict(synth.BASELINE_CONFIGS[args.config])
    if args.obs:
        kw['total_obs'] = args.obs
    per_gpu_obs = kw['total_obs']
    # configs[3] IS a multi-GPU configuration (64 cams x 2M obs sharded over the node): the fixed problem is cut over the
    # ranks (strong scaling).  Every other config scales weakly: fixed observations per GPU, cameras and knots as configured.
    strong = args.config == 3
    if not strong:
        kw['total_obs'] = per_gpu_obs * world
    scene = synth.make_scene(**kw)
    prob, x0 = mp.problem_from_scene(scene)
    shard_mode = args.shard or ('time' if args.solver == 'lm' else 'obs')
    if world > 1:
        collective = args.collective or ('torch' if one_device else 'auto')
        handle, _ = sharded_handle(prob, rank, world, local_rank, time_x=x0 if shard_mode == 'time' else None, collective=collective)
    else:
        handle = ba.BAHandle(prob, device=local_rank)
    solver = ba.SOLVER_LM_SCHUR if args.solver == 'lm' else ba.SOLVER_TRF_LSMR
    jac_mode = ba.JAC_ANALYTIC if args.solver == 'lm' else ba.JAC_PATTERN

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    x = x0.copy()
    lin_iters = 0
    for _ in range(args.warmup):
        r = handle.solve(x, solver=solver, jac_mode=jac_mode, max_nfev=2, return_fun=False, ties='canonical')
        x = r.x
    # (like timeit: no garbage-collector pass inside the timed regions -- a generation-2 collection of the interpreter's heap is milliseconds,
    # twenty steps are eight)
    import gc
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    cost0 = None
    for _ in range(args.steps):
        r = handle.solve(x, solver=solver, jac_mode=jac_mode, max_nfev=2, return_fun=False, ties='canonical')
        x = r.x
        lin_iters += r.lin_iters
        cost0 = r.initial_cost if cost0 is None else cost0
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # The same number of trials inside ONE call (what an iteration costs in a Scene.BA of many evaluations: no residual at the start of
    # every call -- the accepted trial's residual is the next iteration's -- and no return to Python between trials).  Reported beside
    # the headline, never as it; outside the timed region.
    long_solve = None
    if args.solver == 'lm' and not args.no_long_solve:
        barrier()
        tl = time.perf_counter()
        rl = handle.solve(x, solver=solver, jac_mode=jac_mode, max_nfev=args.steps + 1, return_fun=False, ties='canonical')
        barrier()
        dtl = time.perf_counter() - tl
        if world > 1:
            tmax = torch.tensor([dtl], dtype=torch.float64, device='cuda')
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dtl = float(tmax.item())
        long_solve = {'ms_per_trial': 1e3 * dtl / max(rl.nfev - 1, 1), 'trials': rl.nfev - 1, 'linearisations': rl.njev, 'status': rl.status,
                      'what': 'one mvus_ba_solve with max_nfev = steps + 1 continuing from the timed steps; a rejected trial is not followed by a '
                              'linearisation (trials > linearisations - 1 then)'}

    gc.enable()
    # roofline of the dominant kernel: residual + Jacobian, HIP events on the kernel's own stream
    handle.set_x(x)
    t_rj = handle.time_kernel(ba.KERNEL_RESIDUAL_JACOBIAN, 100)          # outputs rotate over >= 1 GiB: every launch goes to HBM
    t_rj_one = handle.time_kernel(ba.KERNEL_RESIDUAL_JACOBIAN_ONE_BUFFER, 20)   # same kernel into one (Infinity-Cache sized) set
    t_r = handle.time_kernel(ba.KERNEL_RESIDUAL, 50)
    t_jv = handle.time_kernel(ba.KERNEL_JV, 50)
    t_jtu = handle.time_kernel(ba.KERNEL_JTU, 50)
    t_asm = handle.time_kernel(ba.KERNEL_ASSEMBLY, 20)
    t_fused = handle.time_kernel(ba.KERNEL_FUSED_ASSEMBLY, 20)
    per_obs, once = algorithmic_bytes(handle.prob)
    bytes_launch = handle.prob.M * per_obs + once
    achieved = bytes_launch / (t_rj * 1e-3) / 1e9

    # the reference-faithful solver (scipy TRF + LSMR restated) on the same workload, a few steps, for the record: with the
    # Jacobian `Scene.BA` uses BY DEFAULT (scipy's grouped 2-point finite differences, MVUS_JAC_FD: what a user of an unmodified
    # config.json gets) and with the analytic Jacobian masked to the reference's pattern (MVUS_JAC_PATTERN)
    parity = None
    if world == 1 and not args.no_parity_solver and args.solver == 'lm':      # (an N=1 report, like the CPU baseline)
        def timed(jm, prepare, nsteps=3):
            xs = x0.copy()
            t_prep = time.perf_counter()
            prepare(xs)                                  # the pattern at x0 (and the column groups): once, like least_squares' jac_sparsity
            t_prep = time.perf_counter() - t_prep
            handle.solve(xs, solver=ba.SOLVER_TRF_LSMR, jac_mode=jm, max_nfev=2, return_fun=False, prepared=True)
            barrier()
            tp = time.perf_counter()
            its = 0
            for _ in range(nsteps):
                rp = handle.solve(xs, solver=ba.SOLVER_TRF_LSMR, jac_mode=jm, max_nfev=2, return_fun=False, prepared=True)
                xs = rp.x
                its += rp.lin_iters
            barrier()
            dtp = time.perf_counter() - tp
            return {'ms_per_step': 1e3 * dtp / nsteps, 'ba_iters_per_sec': nsteps / dtp, 'residuals_per_sec': prob.M * nsteps / dtp,
                    'lsmr_its_per_step': its / nsteps, 'one_time_pattern_setup_ms': 1e3 * t_prep}
        ngroups = [0]
        def prep_fd(xs):
            ngroups[0] = handle.prepare_fd(xs, ties='canonical')
        fd = timed(ba.JAC_FD, prep_fd)
        fd.update({'solver': 'trf_lsmr, jac = scipy 2-point finite differences with column groups (the DEFAULT of Scene.BA; common.py:670)',
                   'column_groups': ngroups[0]})
        # the same default with ONE pass over J per LSMR iteration (MVUS_LSMR_ONE_PASS=1: u kept unnormalised, J v and J^T u from one read
        # of J; opt-in because the unconverged 10-evaluation iterate of one parity fixture leaves its bar -- DESIGN section 7)
        os.environ['MVUS_LSMR_ONE_PASS'] = '1'
        try:
            fd1 = timed(ba.JAC_FD, prep_fd)
        finally:
            del os.environ['MVUS_LSMR_ONE_PASS']
        fd1['solver'] = 'as default_fd, MVUS_LSMR_ONE_PASS=1'
        pat = timed(ba.JAC_PATTERN, lambda xs: handle.prepare_pattern(xs, ties='canonical'))
        pat['solver'] = 'trf_lsmr (scipy restatement, analytic J masked to the reference pattern)'
        parity = dict(pat)
        parity['default_fd'] = fd
        parity['default_fd_one_pass'] = fd1

    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):
        key = 'config%d_calib%d' % (args.config, int(prob.opt_calib))
        traffic = json.load(open(tpath)).get(key, {}).get('bytes_per_launch')

    # N > 1 on a weak-scaling default: the node's strong-scaling figure on BASELINE configs[3] (64 cams x 2M obs cut over the ranks) as a
    # sub-record of the same line, so that a scaling run lands on a configuration BASELINE.json lists whichever default it was started with
    strong3 = None
    if world > 1 and not strong and args.solver == 'lm' and not args.no_strong_config3:
        kw3 = dict(synth.BASELINE_CONFIGS[3])
        scene3 = synth.make_scene(**kw3)
        prob3, x03 = mp.problem_from_scene(scene3)
        h3, _ = sharded_handle(prob3, rank, world, local_rank, time_x=x03 if shard_mode == 'time' else None, collective=collective)
        x3 = x03.copy()
        for _ in range(2):
            x3 = h3.solve(x3, solver=solver, jac_mode=jac_mode, max_nfev=2, return_fun=False, ties='canonical').x
        barrier()
        t3 = time.perf_counter()
        n3 = 5
        c3 = None
        for _ in range(n3):
            r3 = h3.solve(x3, solver=solver, jac_mode=jac_mode, max_nfev=2, return_fun=False, ties='canonical')
            x3 = r3.x
            c3 = r3.initial_cost if c3 is None else c3
        barrier()
        dt3 = time.perf_counter() - t3
        tm3 = torch.tensor([dt3], dtype=torch.float64, device='cuda')
        dist.all_reduce(tm3, op=dist.ReduceOp.MAX)
        dt3 = float(tm3.item())
        strong3 = {'workload': 'BASELINE configs[3]: %d cams x %d obs cut over %d ranks (%s shards), strong scaling' % (prob3.C, prob3.M, world, shard_mode),
                   'n_gpus': world, 'steps': n3, 'ms_per_step': 1e3 * dt3 / n3, 'residuals_per_sec': prob3.M * n3 / dt3,
                   'ba_iters_per_sec': n3 / dt3, 'collective': getattr(h3, 'collective_used', None), 'cost_first': c3, 'cost_last': r3.cost,
                   'one_gpu_ms_per_step': 'profiles/: bench.py --config 3 on one GPU'}
        h3.close()

    if rank == 0:
        M_total = prob.M
        out = {
            'metric': 'residuals/sec', 'value': M_total * args.steps / dt, 'unit': 'residuals/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'ba_iters_per_sec': args.steps / dt,
            'config': {'workload': ('BASELINE configs[%d]' + (' RESIZED with --obs' if args.obs else '') + ': %d cams x %d obs (%d per GPU), rolling shutter %s, %d spline '
                                    'control points, %d params, %d residual rows; step = 1 trust-region trial '
                                    '(residual+Jacobian, %s, trial residual)')
                                   % (args.config, prob.C, M_total, handle.prob.M, 'on' if prob.rs_free else 'off',
                                      int(prob.n_coef.sum()), prob.n_params, prob.n_residuals,
                                      'LM normal equations + Schur solve' if args.solver == 'lm'
                                      else 'scipy-TRF restatement with LSMR on the block-sparse J (%.0f LSMR its/step)' % (lin_iters / max(args.steps, 1))),
                       'solver': args.solver, 'parallelism': '%s-shard x%d' % (shard_mode, world),
                       'collective': getattr(handle, 'collective_used', None),      # 'rccl' = ncclAllReduce called by the library, 'torch' = callback
                       'cost_first': cost0, 'cost_last': r.cost},
            'roofline': {'bound': 'hbm', 'kernel': 'k_observations<calib=%s,jac=true>' % ('true' if prob.opt_calib else 'false'),
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': 'profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, gfx950 corrections applied)' if traffic else None,
                         'bytes_per_launch': bytes_launch, 'avg_launch_ms': t_rj,
                         'bytes_per_obs': per_obs, 'obs_per_launch': handle.prob.M,
                         'timing': 'HIP events over 100 launches whose outputs rotate over >= 3 buffer sets (>= 1 GiB, 4x the Infinity Cache)',
                         'one_buffer': {'avg_launch_ms': t_rj_one, 'achieved': bytes_launch / (t_rj_one * 1e-3) / 1e9,
                                        'frac': bytes_launch / (t_rj_one * 1e-3) / 1e9 / HBM_PEAK_GBS}},
            'parity_solver': parity,
            'strong_config3': strong3,
            'long_solve': long_solve,
            'kernels_ms': {'residual': t_r, 'residual_jacobian': t_rj, 'jv': t_jv, 'jtu': t_jtu, 'normal_eq_assembly_from_J': t_asm,
                           'fused_jacobian_normal_eq_assembly': t_fused},
        }
        if world == 1 and not args.no_cpu_baseline:
            # the SAME-size CPU figure by default: the oracle's 10-evaluation BA on the full workload (one core, ~50 s at configs[2]), beside the
            # bounded 1/32-scale sample; above 600k observations (configs[3]: minutes) only when asked for
            want_full = not args.no_cpu_full and (args.cpu_full or prob.M <= 600_000)
            out['cpu_baseline'] = cpu_baseline(args.config, scene, want_full)
            fs = (out['cpu_baseline'].get('residual_only_full_size') or {}).get('full_size_ba')
            out['cpu_baseline']['full_size_ba'] = fs
            if fs:
                out['cpu_baseline']['gpu_over_cpu_same_size'] = {'ba_iters_per_s': (args.steps / dt) / fs['ba_iters_per_s'],
                                                                 'what': 'this line\'s BA iterations/s over the full-size one-core CPU BA\'s; a reported ratio, not a target'}
        print(json.dumps(out))
    handle.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
