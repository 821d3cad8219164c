#!/usr/bin/env python3
"""How far does the REAL reference reproduce its own converged answer?  (tests/golden/ens_<case>.npz)

Run in the build container only (it needs /root/reference):

    python tests/golden/make_golden_ensemble.py [case ...]

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


def run_ensemble(common, name, sc, golden=None, mi=10, mi2=200):
    st = sc.settings
    C = sc.num_cam
    kw = dict(rs=st['rolling_shutter'], motion_reg=st['motion_reg'], motion_weights=st['motion_weights'], rs_bounds=st['rs_bounds'])
    real_ls = common.least_squares
    out = {}
    x_ref = rmse_ref = None
    for model in ('rel', 'ulp'):
        xs, costs, nfevs, stats, rmses = [], [], [], [], []
        for k in range(0 if model == 'rel' else 1, MEMBERS + 1):
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
            print('  %s %s member %d: cost %.9g nfev %d status %d rmse %.7f' % (name, model, k, res.cost, res.nfev, res.status, rmse), flush=True)
            if k == 0:
                if golden is not None:
                    assert np.array_equal(res.x, golden['ba2_200_x']), 'the unperturbed run does not reproduce %s.npz' % name
                x_ref, rmse_ref = np.array(res.x), rmse
                continue
            xs.append(np.array(res.x)); costs.append(res.cost); nfevs.append(res.nfev); stats.append(res.status); rmses.append(rmse)
        pre = 'ens_' if model == 'rel' else 'ensu_'
        out.update({pre + 'x': np.array(xs), pre + 'cost': np.array(costs), pre + 'nfev': np.array(nfevs, dtype=np.int64),
                    pre + 'status': np.array(stats, dtype=np.int64), pre + 'rmse': np.array(rmses)})
        print('  %s %s: rmse spread %+.2e .. %+.2e px' % (name, model, min(rmses) - rmse_ref, max(rmses) - rmse_ref), flush=True)
    path = os.path.join(HERE, 'ens_' + name + '.npz')
    np.savez_compressed(path, x_ref=x_ref, rmse_ref=np.float64(rmse_ref), noise=np.float64(NOISE), noise_ulp=np.float64(NOISE_ULP), **out)
    print('wrote %s' % path, flush=True)


def main():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from golden_util import load_case, CASES
    common = mg.import_reference()
    only = sys.argv[1:] or CASES
    for name in only:
        scene, g = load_case(name)
        print('case %s' % name, flush=True)
        run_ensemble(common, name, scene, golden=g)


if __name__ == '__main__':
    main()
