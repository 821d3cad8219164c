#!/usr/bin/env python3
"""The reference's incremental loop (main.py:43-82) end to end on the GPU, at BASELINE configs[1] shape: 7 cameras x ~100k
detections, rolling shutter, motion regulariser F -- 12 BAs, 6 outlier passes and 5 x (select_most_overlap, get_camera_pose,
triangulate + refit), starting from two posed cameras and their common piece of trajectory.

    python tools/incremental_loop.py [--obs 100000] [--solver trf|lm] [--max-iter 10] [--cpu-sample]

Prints the wall-clock split per stage (the reference's only own performance figure is this loop's "Total time",
main.py:68,81), the final per-camera reprojection error, the inlier bookkeeping against the generator's labels and the
recovered geometry against ground truth.  --cpu-sample times the oracle's restatement of the reference's BA (scipy
least_squares, one core) on the first 2-camera problem of the loop for a per-BA figure to put beside the GPU's.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--obs', type=int, default=100_000)
    ap.add_argument('--cams', type=int, default=7)
    ap.add_argument('--solver', choices=['trf', 'lm'], default='trf')
    ap.add_argument('--max-iter', type=int, default=10)
    ap.add_argument('--motion-weights', type=float, default=1e2, help='configs[1] quotes 1e4 (README); on this synthetic flight that lets the regulariser outweigh the data 100:1')
    ap.add_argument('--cpu-sample', action='store_true')
    ap.add_argument('--cpu-all', action='store_true', help="time the oracle (scipy least_squares, one core) on EVERY BA of the loop, on the state the GPU BA starts from -- use a down-scaled flight (--obs 20000)")
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--lambda-min', type=float, default=None, help="settings['ba_lambda_min'] (floor of the LM damping)")
    ap.add_argument('--trust-radius', type=float, default=None, help="settings['ba_trust_radius']: 0 = scipy's Delta_0 = |x0|, > 0 that radius, < 0 none")
    ap.add_argument('--lm-wide', choices=['lm', 'trf'], default=None, help="settings['ba_lm_wide_band']: what ba_solver=lm does when the motion rows reach over more than six control points")
    args = ap.parse_args()
    from mvus_amd import pipeline, synth
    kw = dict(synth.BASELINE_CONFIGS[1])
    kw.update(num_cam=args.cams, total_obs=args.obs, motion_weights=args.motion_weights)
    seed = kw.pop('seed') if args.seed is None else (kw.pop('seed'), args.seed)[1]
    nc, nobs = kw.pop('num_cam'), kw.pop('total_obs')
    kw.pop('num_intervals', None)
    t0 = time.perf_counter()
    st_extra = {'ba_solver': args.solver}
    if args.lambda_min is not None: st_extra['ba_lambda_min'] = args.lambda_min
    if args.lm_wide is not None: st_extra['ba_lm_wide_band'] = args.lm_wide
    if args.trust_radius is not None: st_extra['ba_trust_radius'] = args.trust_radius
    flight, sc = pipeline.staged_scene(nc, nobs, seed=seed, settings=st_extra, perturb=0.3, **kw)
    print('scene: %d cameras, %d detections (%s), start trajectory %.0f..%.0f of 0..%.0f, %d control points; set-up %.2f s'
          % (nc, sum(d.shape[1] for d in flight.detections), [d.shape[1] for d in flight.detections], flight.spline['int'][0, 0],
             flight.spline['int'][1, -1], sc.interval[1, -1], sum(len(t[0]) - 4 for t in flight.spline['tck']), time.perf_counter() - t0))
    oracle_ba = None
    if args.cpu_sample:
        from oracle import ba_oracle as orc
        from types import SimpleNamespace
        from threadpoolctl import threadpool_limits
        st = flight.settings
        view = SimpleNamespace(cameras=[dict(K=c.K, d=c.d, R=c.R, t=c.t, resolution=c.resolution) for c in flight.cameras[:2]],
                               detections=flight.detections[:2], alpha=flight.alpha, beta=flight.beta, rs=flight.rs,
                               tck=flight.spline['tck'], interval=flight.spline['int'], settings=st, num_cam=2)
        oprob, ox0 = orc.problem_from_scene(view, num_cam=2, rs=st['rolling_shutter'], motion_reg=st['motion_reg'],
                                            motion_weights=st['motion_weights'], rs_bounds=st['rs_bounds'])
        with threadpool_limits(limits=1):
            t1 = time.perf_counter()
            r = orc.solve(oprob, ox0, max_iter=args.max_iter)
            oracle_ba = (time.perf_counter() - t1, r.nfev, sum(d.shape[1] for d in oprob.detections), ox0.size)
    cpu_rows = []
    if args.cpu_all:
        # the reference's loop calls Scene.BA twelve times (main.py:49-62); before each GPU call the same problem -- cameras
        # sequence[:numCam], their detections, the current spline -- goes to the oracle's restatement of the reference's BA
        from oracle import ba_oracle as orc
        from types import SimpleNamespace
        from threadpoolctl import threadpool_limits
        gpu_ba = flight.BA

        def ba_with_cpu_clock(numCam, **kw):
            cams = list(flight.sequence[:numCam])
            view = SimpleNamespace(cameras=[dict(K=flight.cameras[i].K, d=flight.cameras[i].d, R=flight.cameras[i].R, t=flight.cameras[i].t,
                                                 resolution=flight.cameras[i].resolution) for i in cams],
                                   detections=[flight.detections[i] for i in cams], alpha=np.asarray(flight.alpha)[cams],
                                   beta=np.asarray(flight.beta)[cams], rs=np.asarray(flight.rs)[cams], tck=flight.spline['tck'],
                                   interval=flight.spline['int'], settings=flight.settings, num_cam=numCam)
            oprob, ox0 = orc.problem_from_scene(view, num_cam=numCam, rs=kw.get('rs', False), motion_reg=kw.get('motion_reg', False),
                                                motion_weights=kw.get('motion_weights', 1), rs_bounds=kw.get('rs_bounds', False))
            with threadpool_limits(limits=1):
                t1 = time.perf_counter()
                r = orc.solve(oprob, ox0, max_iter=kw.get('max_iter', 10))
                cpu = time.perf_counter() - t1
            t1 = time.perf_counter()
            res = gpu_ba(numCam, **kw)
            gpu = time.perf_counter() - t1
            cpu_rows.append((numCam, sum(d.shape[1] for d in oprob.detections), ox0.size, r.nfev, cpu, r.cost, gpu, res.cost, getattr(res, 'solver_used', '?')))
            return res
        flight.BA = ba_with_cpu_clock
    t0 = time.perf_counter()
    timer = pipeline.incremental_reconstruction(flight, max_iter=args.max_iter, verbose=False)
    total = time.perf_counter() - t0
    print('loop finished: sequence %s, total %.2f s (solver %s, max_iter %d)' % (flight.sequence, total, args.solver, args.max_iter))
    print('%-22s %6s %10s' % ('stage', 'calls', 'seconds'))
    counts = {}
    for stage, _, _ in timer.rows:
        counts[stage] = counts.get(stage, 0) + 1
    for stage, secs in sorted(timer.totals().items(), key=lambda kv: -kv[1]):
        print('%-22s %6d %10.3f' % (stage, counts[stage], secs))
    print('per call, in order: ' + ' '.join('%s[%d]=%.2f' % (s_[:4], n, dt) for s_, n, dt in timer.rows))
    if oracle_ba:
        print('CPU (oracle restatement of the reference BA, scipy least_squares, 1 core): first 2-camera BA, %d detections, %d '
              'parameters, %d evaluations: %.1f s' % (oracle_ba[2], oracle_ba[3], oracle_ba[1], oracle_ba[0]))
    if cpu_rows:
        print('every BA of the loop: the oracle (scipy least_squares as the reference calls it, ONE core of this host) and the GPU on the same start')
        print('%3s %5s %9s %7s | %5s %9s %14s | %9s %14s  %s' % ('BA', 'cams', 'detect.', 'params', 'nfev', 'CPU s', 'CPU cost', 'GPU s', 'GPU cost', 'GPU solver'))
        for k, (nc_, m_, n_, nfev, cpu, ccost, gpu, gcost, used) in enumerate(cpu_rows):
            print('%3d %5d %9d %7d | %5d %9.2f %14.6g | %9.3f %14.6g  %s' % (k + 1, nc_, m_, n_, nfev, cpu, ccost, gpu, gcost, used[:40]))
        print('sum over the %d BAs: CPU %.1f s, GPU %.2f s (ratio %.0f)' % (len(cpu_rows), sum(r[4] for r in cpu_rows), sum(r[6] for r in cpu_rows),
                                                                          sum(r[4] for r in cpu_rows) / max(sum(r[6] for r in cpu_rows), 1e-9)))
        print('(loop total above includes the CPU clock; GPU stages alone: %.2f s)' % (total - sum(r[4] for r in cpu_rows)))
    ev = pipeline.evaluate_against_truth(flight, sc)
    print('mean reprojection error per camera (px):', np.round(ev['mean_err'], 3))
    print('detections kept / clean by the generator / kept although not clean:', list(zip(ev['kept'], ev['clean'], ev['kept_dirty'])))
    print('trajectory: covers %.0f..%.0f of 0..%.0f; vs ground truth after similarity: rms %.3f m, max %.3f m, scale %.4f'
          % (ev['trajectory_extent'] + (ev['traj_rms'], ev['traj_max'], ev['scale'])))
    print('camera centres vs truth (m):', np.round(ev['centre_err'], 3), ' orientations (deg):', np.round(ev['rot_err_deg'], 3))


if __name__ == '__main__':
    main()
