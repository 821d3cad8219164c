#!/usr/bin/env python3
"""The x-level parity table of DESIGN.md section 2: how far the recovered solution (res.x -- poses, alpha/beta/rs, trajectory)
of every solver mode is from the reference's converged second BA, in gauge-invariant terms (tests/gauge.py), next to
what the reference reproduces of itself (tests/golden/ens_*.npz) and to the generator's ground truth.

    python tools/xlevel_table.py [--host] [case ...]        (GPU by default; --host: the g++ build of the same solvers)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np

import gauge
import golden_cases
from golden_util import GOLDEN_DIR
from test_fd_mode_host import filtered_case, golden_matrix
from mvus_amd import _lib, problem as mp
from oracle import ba_oracle as orc


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    host = '--host' in sys.argv
    cases = args or ['c1_pinhole_2cam', 'rs_F_2int_3cam', 'dist_fixed_2cam', 'calib_KE_bounds_3cam', 'calib_KE_wellposed_5cam', 'config1_shape_7cam']
    for name in cases:
        scene, g = filtered_case(name)
        prob, _ = mp.problem_from_scene(scene)
        oprob, _ = orc.problem_from_scene(scene)
        xr = g['ba2_200_x']
        tr = golden_cases.make(name).truth
        xt = orc.pack_x(oprob, tr['alpha'], tr['beta'], tr['rs'], tr['cameras'], [t[1] for t in tr['tck']])
        keys = list(gauge.METRICS) + (list(gauge.CALIB_METRICS) if oprob.opt_calib else [])
        fmt = lambda c: ' '.join('%s %.1e' % (k.replace('_max', '').replace('traj_', 't'), c[k]) for k in keys)
        print('== %s  (reference: cost %.9g rmse %.7f nfev %d status %d)' % (name, float(g['ba2_200_cost']), float(g['ba2_200_rmse']),
                                                                              int(g['ba2_200_nfev']), int(g['ba2_200_status'])))
        epath = os.path.join(GOLDEN_DIR, 'ens_' + name + '.npz')
        if os.path.exists(epath):
            ens = dict(np.load(epath))
            for pre, label in (('ens_', 'reference ensemble, 1e-15 relative '), ('ensu_', 'reference ensemble, ulp(1e3 px) abs')):
                sp = gauge.ensemble_spread(oprob, xr, ens[pre + 'x'])
                print('  %-36s rmse %.1e | %s' % (label, sp['rmse'], fmt(sp)))
        print('  %-36s          | %s' % ('reference, move x0 -> x', fmt(gauge.compare(oprob, xr, g['ba2_200_x0']))))
        print('  %-36s          | %s' % ('reference vs ground truth', fmt(gauge.compare(oprob, xt, xr))))
        runs = [('TRF+FD (reference algorithm)', _lib.SOLVER_TRF_LSMR, _lib.JAC_FD, True, 200, None),
                ('TRF+pattern (analytic, masked)', _lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, True, 200, None),
                ('LM+Schur, 10 evaluations', _lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, False, 10, None),
                ('LM+Schur, 200', _lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, False, 200, None),
                ('LM+Schur, 200, no damping floor', _lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, False, 200, 0.0)]
        for label, solver, jm, mat, nf, lmin in runs:
            opts = _lib.default_opts(solver, jm, nf)
            if lmin is not None:
                opts.lm_lambda_min = lmin
            matrix = golden_matrix(g, second=True) if mat else None
            if host:
                from hostcheck_util import HostHandle
                x, res, _ = HostHandle(prob).solve(g['ba2_200_x0'], opts, matrix=matrix)
            else:
                from mvus_amd.ba import BAHandle
                with BAHandle(prob) as h:
                    res = h.solve(g['ba2_200_x0'], opts=opts, matrix=matrix)
                    x = res.x
            c, ct = gauge.compare(oprob, xr, x), gauge.compare(oprob, xt, x)
            keep = np.concatenate(orc.outlier_keep_mask(oprob, x, float(g['thres_outlier']))).astype(np.uint8)
            print('  %-36s st %d nfev %3d cost %.9g rmse %+.1e flips %d | vs ref: %s | vs truth: trms %.1e centre %.1e'
                  % (label, res.status, res.nfev, res.cost, c['rmse_b'] - c['rmse_a'], int((keep != g['ba2_200_keep']).sum()), fmt(c),
                     ct['traj_rms'], ct['centre_max']), flush=True)


if __name__ == '__main__':
    main()
