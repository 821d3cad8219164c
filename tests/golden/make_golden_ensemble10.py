#!/usr/bin/env python3
"""How far does the REAL reference reproduce its own UNCONVERGED answer -- the 10-evaluation BA every real call returns
(Scene.BA's default max_iter=10, reconstruction/common.py:441; main.py:49 and :59)?  (tests/golden/ens10_<case>.npz)

Run in the build container only (it needs /root/reference):

    python tests/golden/make_golden_ensemble10.py [--members 8] [case ...]

The companion of make_golden_ensemble.py (which does the same for the CONVERGED second BA).  For every golden case the
reference's first BA (max_iter=10) is re-run with the residual vector its least_squares call sees perturbed in its last
place -- the same two noise models, eight members each:

  `rel`  f * (1 + 1e-15 N(0,1))            one unit in the last place of the residual
  `ulp`  f + 1.14e-13 N(0,1) (f != 0)      one unit in the last place of the pixel coordinates (ulp(1000 px)): the least by which
                                           ANY re-implementation with another operation order differs from the reference

and, from the unperturbed first BA + remove_outliers, the second BA (max_iter=10, main.py:59) likewise.  Stored per member: cost,
RMSE (Scene.error_cam(i, 'dist') over the visible detections), nfev, and -- first BA -- the inlier mask of remove_outliers at the
returned point; plus the spreads the tests use: the largest deviation of any member from the unperturbed run in relative cost,
in RMSE and in mask flips.  Member 0 must reproduce ba10_cost / ba10_rmse / outlier_keep of <case>.npz bit for bit (asserted).
tests/test_gpu_parity.py::test_fd_mode_ba_vs_reference_result holds the GPU's reference-algorithm mode to SPREAD_FACTOR x these
(they replace bars that were multiples of the GPU's own measured deviations).  Data only: no reference source is copied.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg            # noqa: E402  (cv2 stand-in, reference import, scene construction)
from make_golden_ensemble import NOISE, NOISE_ULP   # noqa: E402

MEMBERS = 8


def _noisy_ls(common, real_ls, model, rng):
    def spy(fn, x0, **kwargs):
        def noisy(x):
            f = np.asarray(fn(x))
            if model == 'rel':
                return f * (1.0 + NOISE * rng.standard_normal(f.size))
            return f + NOISE_ULP * rng.standard_normal(f.size) * (f != 0)
        return real_ls(noisy, x0, **kwargs)
    return spy


def _rmse(ref, C):
    return float(np.sqrt(np.mean(np.concatenate([ref.error_cam(i, 'dist') for i in range(C)]) ** 2)))


def run_member(common, sc, model, k, stage, mi=10):
    """stage 1: the first BA perturbed (member 0 = unperturbed), then remove_outliers -> cost, rmse, nfev, keep.
    stage 2: first BA + remove_outliers unperturbed, the SECOND BA (max_iter=10) perturbed -> cost, rmse, nfev."""
    st = sc.settings
    C = sc.num_cam
    kw = dict(rs=st['rolling_shutter'], motion_reg=st['motion_reg'], motion_weights=st['motion_weights'], rs_bounds=st['rs_bounds'])
    real_ls = common.least_squares
    ref = mg.build_reference_scene(common, sc)
    rng = np.random.default_rng((3000 if model == 'rel' else 4000) + 100 * stage + k)
    noisy = _noisy_ls(common, real_ls, model, rng)
    if stage == 1 and k > 0:
        common.least_squares = noisy
    try:
        res = ref.BA(C, max_iter=mi, **kw)
    finally:
        common.least_squares = real_ls
    out = {}
    if stage == 1:
        out.update(cost=float(res.cost), nfev=int(res.nfev), rmse=_rmse(ref, C))
        frames_before = [d[0].copy() for d in ref.detections]
        ref.remove_outliers(ref.sequence[:C], thres=st['thres_outlier'])
        out['keep'] = np.concatenate([np.isin(fb, d[0]) for fb, d in zip(frames_before, ref.detections)]).astype(np.uint8)
        return out
    ref.remove_outliers(ref.sequence[:C], thres=st['thres_outlier'])
    if k > 0:
        common.least_squares = noisy
    try:
        res2 = ref.BA(C, max_iter=mi, **kw)
    finally:
        common.least_squares = real_ls
    out.update(cost=float(res2.cost), nfev=int(res2.nfev), rmse=_rmse(ref, C))
    return out


def run_case(common, name, sc, golden, members):
    out = {}
    for stage, tag in ((1, 'ba10'), (2, 'ba2_10')):
        r0 = run_member(common, sc, 'rel', 0, stage)
        if stage == 1:
            assert r0['cost'] == float(golden['ba10_cost']) and r0['rmse'] == float(golden['ba10_rmse']), 'the unperturbed run does not reproduce %s.npz' % name
            assert np.array_equal(r0['keep'], golden['outlier_keep'])
        elif 'ba2_10_cost' in golden:
            assert r0['cost'] == float(golden['ba2_10_cost']), (r0['cost'], float(golden['ba2_10_cost']))
        rs_ = [(m, k, run_member(common, sc, m, k, stage)) for m in ('rel', 'ulp') for k in range(1, members + 1)]
        for m, k, r in rs_:
            extra = ' flips %d' % int(np.sum(r['keep'] != r0['keep'])) if stage == 1 else ''
            print('  %s %s %s member %d: cost %.9g (rel %+.2e) nfev %d rmse %.7f (%+.2e)%s'
                  % (name, tag, m, k, r['cost'], r['cost'] / r0['cost'] - 1.0, r['nfev'], r['rmse'], r['rmse'] - r0['rmse'], extra), flush=True)
        out[tag + '_cost_ref'] = np.float64(r0['cost'])
        out[tag + '_rmse_ref'] = np.float64(r0['rmse'])
        out[tag + '_nfev_ref'] = np.int64(r0['nfev'])
        out[tag + '_cost'] = np.array([r['cost'] for _, _, r in rs_])
        out[tag + '_rmse'] = np.array([r['rmse'] for _, _, r in rs_])
        out[tag + '_nfev'] = np.array([r['nfev'] for _, _, r in rs_], dtype=np.int64)
        out[tag + '_spread_cost_rel'] = np.float64(np.max(np.abs(out[tag + '_cost'] / r0['cost'] - 1.0)))
        out[tag + '_spread_rmse'] = np.float64(np.max(np.abs(out[tag + '_rmse'] - r0['rmse'])))
        if stage == 1:
            flips = np.array([int(np.sum(r['keep'] != r0['keep'])) for _, _, r in rs_], dtype=np.int64)
            out['ba10_flips'] = flips
            out['ba10_spread_flips'] = np.int64(flips.max())
        print('  %s %s: spread cost %.2e (relative), rmse %.2e px%s' % (name, tag, out[tag + '_spread_cost_rel'], out[tag + '_spread_rmse'],
              ', mask flips %d' % int(out['ba10_spread_flips']) if stage == 1 else ''), flush=True)
    path = os.path.join(HERE, 'ens10_' + name + '.npz')
    np.savez_compressed(path, noise=np.float64(NOISE), noise_ulp=np.float64(NOISE_ULP), members=np.int64(members), **out)
    print('wrote %s' % path, flush=True)


def main():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from golden_util import load_case, CASES
    common = mg.import_reference()
    args = sys.argv[1:]
    members = MEMBERS
    while args and args[0].startswith('--'):
        members = int(args[1])
        args = args[2:]
    for name in (args or CASES):
        scene, g = load_case(name)
        print('case %s' % name, flush=True)
        run_case(common, name, scene, g, members)


if __name__ == '__main__':
    main()
