#!/usr/bin/env python3
"""How far does the REAL reference reproduce its own converged answer?  (tests/golden/ens_<case>.npz)

Run in the build container only (it needs /root/reference):

    python tests/golden/make_golden_ensemble.py [--members 8] [--workers 1] [case ...]

For every golden case the reference's pipeline of main.py:49-62 is re-run -- first BA (max_iter=10), remove_outliers, second
BA to convergence (max_iter=200) -- with the residual vector the second least_squares call sees perturbed in its last
place; everything else (start, matrix, algorithm, scipy) is the reference's own.  Two noise models, eight members each:

  `ens_*`   f * (1 + 1e-15 N(0,1))         one unit in the last place of the RESIDUAL (relative noise)
  `ensu_*`  f + 1.14e-13 N(0,1) (f != 0)   one unit in the last place of the PIXEL COORDINATES (ulp(1000 px)): a residual is
                                           |u_projected - u_observed| with both ~1e3 px, so this is the size of the rounding
                                           error of the reference's own arithmetic and the least by which ANY re-implementation
                                           with another operation order differs from it (oracle and HIP kernels: <= 4.5e-13)

Stored per model: the eight converged parameter vectors (+ cost, nfev, status, RMSE).  Member 0 of the run is the
unperturbed one and must reproduce the `ba2_200_x` of <case>.npz bit for bit (asserted).  The tests derive the
gauge-invariant spread of trajectory, camera centres, beta differences, ... from these vectors (tests/gauge.py) and hold
the GPU's reference-algorithm mode to a multiple of it.  Data only: no reference source is copied.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg            # noqa: E402  (cv2 stand-in, reference import, scene construction)

MEMBERS = 8
NOISE = 1e-15          # relative (ens_*)
NOISE_ULP = 1.1368683772161603e-13   # absolute, ulp(1000.0) (ensu_*)


def run_member(common, sc, model, k, mi=10, mi2=200):
    """One run of the reference's pipeline with noise model `model` ('rel' / 'ulp'), member k (0 = unperturbed)."""
    st = sc.settings
    C = sc.num_cam
    kw = dict(rs=st['rolling_shutter'], motion_reg=st['motion_reg'], motion_weights=st['motion_weights'], rs_bounds=st['rs_bounds'])
    real_ls = common.least_squares
    ref = mg.build_reference_scene(common, sc)
    ref.BA(C, max_iter=mi, **kw)
    ref.remove_outliers(ref.sequence[:C], thres=st['thres_outlier'])
    rng = np.random.default_rng((1000 if model == 'rel' else 2000) + k)

    def spy(fn, x0, **kwargs):
        if k == 0:
            return real_ls(fn, x0, **kwargs)

        def noisy(x):
            f = np.asarray(fn(x))
            if model == 'rel':
                return f * (1.0 + NOISE * rng.standard_normal(f.size))
            return f + NOISE_ULP * rng.standard_normal(f.size) * (f != 0)
        return real_ls(noisy, x0, **kwargs)
    common.least_squares = spy
    try:
        res = ref.BA(C, max_iter=mi2, **kw)
    finally:
        common.least_squares = real_ls
    rmse = float(np.sqrt(np.mean(np.concatenate([ref.error_cam(i, 'dist') for i in range(C)]) ** 2)))
    return dict(x=np.array(res.x), cost=float(res.cost), nfev=int(res.nfev), status=int(res.status), rmse=rmse)


def run_ensemble(common, name, sc, golden=None, members=MEMBERS, workers=1):
    """Sequential (workers=1) or one child process per member (big cases: the reference needs ~20 min per run there)."""
    jobs = [('rel', 0)] + [(m, k) for m in ('rel', 'ulp') for k in range(1, members + 1)]
    results = {}
    if workers > 1:
        import subprocess
        import tempfile
        tmp = tempfile.mkdtemp(prefix='ens_')
        pending, running = list(jobs), []
        while pending or running:
            while pending and len(running) < workers:
                m, k = pending.pop(0)
                out = os.path.join(tmp, '%s_%d.npz' % (m, k))
                running.append((m, k, out, subprocess.Popen([sys.executable, os.path.abspath(__file__), '--one', name, m, str(k), out])))
            m, k, out, proc = running.pop(0)
            if proc.wait() != 0:
                raise SystemExit('member %s %d failed' % (m, k))
            results[(m, k)] = dict(np.load(out))
    else:
        for m, k in jobs:
            results[(m, k)] = run_member(common, sc, m, k)
    for (m, k), r in sorted(results.items()):
        print('  %s %s member %d: cost %.9g nfev %d status %d rmse %.7f' % (name, m, k, float(r['cost']), int(r['nfev']), int(r['status']), float(r['rmse'])), flush=True)
    r0 = results[('rel', 0)]
    if golden is not None:
        assert np.array_equal(r0['x'], golden['ba2_200_x']), 'the unperturbed run does not reproduce %s.npz' % name
    out = {}
    for m, pre in (('rel', 'ens_'), ('ulp', 'ensu_')):
        rs_ = [results[(m, k)] for k in range(1, members + 1)]
        out.update({pre + 'x': np.array([r['x'] for r in rs_]), pre + 'cost': np.array([float(r['cost']) for r in rs_]),
                    pre + 'nfev': np.array([int(r['nfev']) for r in rs_], dtype=np.int64),
                    pre + 'status': np.array([int(r['status']) for r in rs_], dtype=np.int64),
                    pre + 'rmse': np.array([float(r['rmse']) for r in rs_])})
        d = out[pre + 'rmse'] - float(r0['rmse'])
        print('  %s %s: rmse spread %+.2e .. %+.2e px' % (name, m, d.min(), d.max()), flush=True)
    path = os.path.join(HERE, 'ens_' + name + '.npz')
    np.savez_compressed(path, x_ref=np.array(r0['x']), rmse_ref=np.float64(float(r0['rmse'])), noise=np.float64(NOISE),
                        noise_ulp=np.float64(NOISE_ULP), **out)
    print('wrote %s' % path, flush=True)


def main():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from golden_util import load_case, CASES
    common = mg.import_reference()
    args = sys.argv[1:]
    if args and args[0] == '--one':                       # child of a parallel run: one member, result to a file
        _, name, model, k, out = args
        scene, g = load_case(name)
        np.savez(out, **run_member(common, scene, model, int(k)))
        return
    members, workers = MEMBERS, 1
    while args and args[0].startswith('--'):
        key, val = args[0], int(args[1])
        members, workers = (val, workers) if key == '--members' else (members, val)
        args = args[2:]
    for name in (args or CASES):
        scene, g = load_case(name)
        print('case %s' % name, flush=True)
        run_ensemble(common, name, scene, golden=g, members=members, workers=workers)


if __name__ == '__main__':
    main()
