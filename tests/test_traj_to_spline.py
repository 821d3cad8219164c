"""Scene.traj_to_spline (reference common.py:224-270) = FITPACK ``splprep`` inside the smooth_factor loop.

* the oracle (oracle/fitpack_oracle.py, a line-by-line restatement of fppara) is pinned against scipy's own splprep and against
  the golden fixture written by the REAL reference's traj_to_spline (tests/golden/traj_spline.npz, make_golden_spline.py);
* the GPU path (mvus_spline_smooth behind mvus_amd.spline.smooth_fit / traj_fit and Scene.traj_to_spline) is compared with
  the oracle, with scipy and with that fixture: identical knot vectors, coefficients to 1e-8 absolute (values ~10).
"""
import os
import sys

import numpy as np
import pytest
from scipy import interpolate

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golden_util import GOLDEN_DIR                      # noqa: E402
from mvus_amd.reconstruction import common              # noqa: E402
from oracle import fitpack_oracle as fo                 # noqa: E402

COEF_ATOL = 1e-8


def _trajectory(seed, m, noise=0.02, speed=1.0):
    rng = np.random.default_rng(seed)
    u = np.cumsum(rng.uniform(0.5, 1.5, m))
    X = np.vstack([10 * np.sin(u / 80 * speed), 10 * np.cos(u / 95), 30 + 3 * np.sin(u / 50)]) + rng.normal(0, noise, (3, m))
    return u, X


def _golden():
    return dict(np.load(os.path.join(GOLDEN_DIR, 'traj_spline.npz')))


def _parts(g):
    traj = g['traj']
    for i in range(int(g['n_int'])):
        keep = (traj[0] >= g['interval'][0, i]) & (traj[0] <= g['interval'][1, i])
        yield i, traj[:, keep]


def _reference_loop(part, smooth_factor, fit):
    """common.py:236-262 with ``fit(X, u, s) -> tck``; returns (tck, s of the accepted fit, number of fits)."""
    lo, hi = min(smooth_factor), max(smooth_factor)
    measure = part[0, -1] - part[0, 0]
    s = (1e-3) ** 2 * measure
    prev, direction, calls = 0, 0, 0
    while True:
        tck = fit(part[1:], part[0], s)
        calls += 1
        n = len(tck[0]) - 4
        if n == prev and n == 4 and direction == 2:
            break
        prev = n
        if measure / n > hi:
            s, direction = s / 1.5, 1
        elif measure / n < lo:
            s, direction = s * 2, 2
        else:
            break
    return tck, s, calls


CASES = [  # seed, samples, s as a multiple of the duration or absolute
    (0, 120, ('rel', 1e-6)), (1, 90, ('rel', 1e-4)), (2, 150, ('abs', 0.05)), (3, 200, ('abs', 1.0)), (4, 80, ('abs', 50.0)),
    (5, 60, ('abs', 5e3)),
]


def _s_of(u, spec):
    return spec[1] * (u[-1] - u[0]) if spec[0] == 'rel' else spec[1]


@pytest.mark.parametrize('seed,m,spec', CASES)
def test_oracle_is_scipy_splprep(seed, m, spec):
    """The restatement reproduces FITPACK itself: same knots, same ier, coefficients and fp to rounding -- in the interpolating
    regime (tiny s), with knots added and smoothing, and for a fit that stays the least-squares polynomial (ier = -2)."""
    u, X = _trajectory(seed, m, speed=1 + seed)
    s = _s_of(u, spec)
    ((t0, c0, _), _u), fp0, ier0, _msg = interpolate.splprep(X, u=u, s=s, k=3, full_output=1)
    (t, c, k), info = fo.splprep(X, u, s)
    np.testing.assert_array_equal(t, t0)
    assert info['ier'] == ier0 and k == 3
    np.testing.assert_allclose(np.asarray(c), np.asarray(c0), rtol=0, atol=COEF_ATOL)
    assert abs(info['fp'] - fp0) <= 1e-9 * max(fp0, s)


def test_oracle_reproduces_the_reference_fixture():
    """The accepted fit of every interval of the golden trajectory: the smoothing factor the reference's loop ends on is found
    with scipy (fast), the oracle's fit at that factor equals what the reference stored."""
    g = _golden()
    for i, part in _parts(g):
        tck_s, s, calls = _reference_loop(part, g['smooth_factor'], lambda X, u, s: interpolate.splprep(X, u=u, s=s, k=3)[0])
        assert calls > 3                                  # the loop really walks through several smoothing factors
        np.testing.assert_array_equal(tck_s[0], g['knots_%d' % i])
        (t, c, _), info = fo.splprep(part[1:], part[0], s)
        np.testing.assert_array_equal(t, g['knots_%d' % i])
        np.testing.assert_allclose(np.asarray(c), g['coefs_%d' % i], rtol=0, atol=COEF_ATOL)


def test_host_side_of_the_gpu_fit_matches_the_oracle():
    """The scalar FITPACK routines the GPU path runs on the host (csrc/spline_fit.hip.h: fpdisc, fprati, fpknot), compiled with
    g++ (tests/hostcheck), against the oracle's restatement -- bit for bit, they are the same arithmetic."""
    import ctypes
    from hostcheck_util import load
    from mvus_amd import _lib
    lib = load()
    rng = np.random.default_rng(3)
    u, X = _trajectory(3, 90)
    (t, _c, _k), _info = fo.splprep(X, u, 0.05)
    n = t.size
    b = np.zeros((n - 8, 5))
    lib.hostcheck_fpdisc(n, _lib.dptr(np.ascontiguousarray(t)), _lib.dptr(b))
    np.testing.assert_array_equal(b, fo.fpdisc(t, n, 5))
    for _ in range(50):
        p1, p2 = np.sort(rng.uniform(0.1, 10.0, 2))
        f1, f2 = rng.uniform(0.1, 5.0), rng.uniform(-5.0, 5.0)
        p3, f3 = (-1.0, rng.uniform(-9.0, -5.0)) if rng.random() < 0.5 else (p2 + rng.uniform(0.1, 5.0), rng.uniform(-9.0, -5.0))
        pf = np.array([p1, f1, p2, f2, p3, f3])
        got = lib.hostcheck_fprati(_lib.dptr(pf))
        want = fo.fprati(p1, f1, p2, f2, p3, f3)
        assert got == want[0] and tuple(pf[[0, 1, 4, 5]]) == want[1:]
    # a few knot insertions from the state fppara is in after its first least-squares pass
    m, nest = u.size, u.size + 6
    t1 = np.zeros(nest); t1[:4] = u[0]; t1[4:8] = u[-1]
    t2 = t1.copy()
    fp1 = np.zeros(nest); fp1[0] = 3.25
    fp2 = fp1.copy()
    nr1 = np.zeros(nest, dtype=np.int64); nr1[0] = m - 2
    nr2 = nr1.astype(np.int32)
    n1, ri1 = 8, 1
    n2, ri2 = ctypes.c_int32(8), ctypes.c_int32(1)
    for step in range(12):
        n1, ri1 = fo.fpknot(u, t1, n1, fp1, nr1, ri1)
        lib.hostcheck_fpknot(nest, _lib.dptr(u), ctypes.byref(n2), _lib.dptr(t2), _lib.dptr(fp2), nr2.ctypes.data_as(_lib.c_int32_p), ctypes.byref(ri2))
        assert (n2.value, ri2.value) == (n1, ri1)
        np.testing.assert_array_equal(t2, t1)
        np.testing.assert_array_equal(fp2, fp1)
        np.testing.assert_array_equal(nr2, nr1)
        fp1[:ri1] *= rng.uniform(0.5, 1.5, ri1)           # new residuals before the next insertion
        fp2[:] = fp1


def test_batched_knot_insertion_is_fpknot_repeated():
    """fpknot_batch (heap of candidate intervals + linked list, what the GPU path runs between passes) against `nplus` plain
    fpknot calls from the same state: knots, residual estimates, sample counts, n and nrint identical -- on random states with
    many exactly tied residuals (fpknot gives both halves of a split interval fp * n1 / n and fp * n2 / n: ties are the rule),
    intervals without samples, and the stop at nest."""
    import ctypes
    from hostcheck_util import load
    from mvus_amd import _lib
    lib = load()
    rng = np.random.default_rng(11)
    for trial in range(30):
        m = int(rng.integers(400, 3000))
        u = np.cumsum(rng.uniform(0.5, 1.5, m))
        nest = m + 6
        # a state as after some passes: nrint intervals with random sample counts that add up (every interior knot is a sample)
        nrint = int(rng.integers(64, 200))
        cuts = np.sort(rng.choice(np.arange(2, m - 1), nrint - 1, replace=trial % 3 == 0))       # (replace: empty intervals cannot occur; adjacent cuts give nr = 0)
        cuts = np.unique(cuts)
        nrint = cuts.size + 1
        t = np.zeros(nest); t[:4] = u[0]
        t[4:4 + cuts.size] = u[cuts - 1]                   # knot at sample index cuts (1-based)
        n = 8 + cuts.size
        t[n - 4:n] = u[-1]
        edges = np.concatenate([[1], cuts, [m]])
        nr = np.zeros(nest, dtype=np.int32)
        nr[:nrint] = np.diff(edges) - 1
        nr[0] = cuts[0] - 2 if cuts.size else m - 2
        fp = np.zeros(nest)
        vals = rng.uniform(0.0, 5.0, nrint)
        vals[rng.random(nrint) < 0.3] = 2.5                # exact ties between computed residuals too
        vals[rng.random(nrint) < 0.05] = 0.0
        fp[:nrint] = vals
        nplus = int(rng.integers(8, 150))
        nmax = m + 4
        nest_stop = n + nplus - 3 if trial % 5 == 4 else nest          # some trials run into "n == nest"
        t1, f1, r1 = t.copy(), fp.copy(), nr.copy()
        n1, ri1 = ctypes.c_int32(n), ctypes.c_int32(nrint)
        for _ in range(nplus):
            lib.hostcheck_fpknot(nest, _lib.dptr(u), ctypes.byref(n1), _lib.dptr(t1), _lib.dptr(f1), r1.ctypes.data_as(_lib.c_int32_p), ctypes.byref(ri1))
            if n1.value == nmax or n1.value == nest_stop:
                break
        t2, f2, r2 = t.copy(), fp.copy(), nr.copy()
        n2, ri2 = ctypes.c_int32(n), ctypes.c_int32(nrint)
        t2s, f2s, r2s = t2[:nest_stop].copy(), f2[:nest_stop].copy(), r2[:nest_stop].copy()
        lib.hostcheck_fpknot_batch(nest_stop, _lib.dptr(u), ctypes.byref(n2), _lib.dptr(t2s), _lib.dptr(f2s), r2s.ctypes.data_as(_lib.c_int32_p), ctypes.byref(ri2),
                                   nplus, nmax)
        assert (n2.value, ri2.value) == (n1.value, ri1.value)
        k = ri1.value
        np.testing.assert_array_equal(t2s[4:4 + k - 1], t1[4:4 + k - 1])
        np.testing.assert_array_equal(f2s[:k], f1[:k])
        np.testing.assert_array_equal(r2s[:k], r1[:k])


def test_double_double_arithmetic_on_the_host():
    """The double-double primitives of the ill-conditioned passes against exact rational arithmetic: ~1e-31 relative."""
    from fractions import Fraction
    from hostcheck_util import load
    from mvus_amd import _lib
    lib = load()
    rng = np.random.default_rng(4)

    def run(op, a, b):
        out = np.zeros(2)
        lib.hostcheck_dd(op, _lib.dptr(np.array(a)), _lib.dptr(np.array(b)), _lib.dptr(out))
        return Fraction(out[0]) + Fraction(out[1])

    for _ in range(200):
        ah, bh = rng.uniform(-3, 3), rng.uniform(0.5, 3)
        a, b = (ah, ah * rng.uniform(-1, 1) * 2.0 ** -54), (bh, bh * rng.uniform(-1, 1) * 2.0 ** -54)
        fa, fb = Fraction(a[0]) + Fraction(a[1]), Fraction(b[0]) + Fraction(b[1])
        assert run(4, a, b) == Fraction(a[0]) * Fraction(b[0])                       # the product of two doubles is exact
        for op, exact in ((0, fa + fb), (1, fa * fb), (2, fa / fb)):
            err = abs(run(op, a, b) - exact)
            assert err <= abs(exact) * Fraction(1, 2 ** 100) + Fraction(1, 2 ** 150)
        r = run(3, (abs(a[0]) + 0.1, 0.0), b)
        assert abs(r * r - (Fraction(abs(a[0]) + 0.1))) <= Fraction(1, 2 ** 98) * r * r


# ---- GPU ------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize('seed,m,spec', CASES + [(6, 349, ('rel', 1e-6)), (7, 276, ('abs', 0.05)), (8, 5000, ('abs', 1.0))])
def test_gpu_smooth_fit_matches_fitpack(seed, m, spec):
    """mvus_spline_smooth against scipy's splprep (and, at sizes it finishes, the oracle): identical knots, ier, coefficients."""
    from mvus_amd import spline
    u, X = _trajectory(seed, m, speed=1 + seed % 7)
    s = _s_of(u, spec)
    (t0, c0, _), _u = interpolate.splprep(X, u=u, s=s, k=3)
    full = interpolate.splprep(X, u=u, s=s, k=3, full_output=1)
    tck, fp, ier = spline.smooth_fit(u, X, s, full_output=True)
    np.testing.assert_array_equal(tck[0], t0)
    assert ier == full[2] and tck[2] == 3
    np.testing.assert_allclose(np.asarray(tck[1]), np.asarray(c0), rtol=0, atol=COEF_ATOL)
    assert abs(fp - full[1]) <= 1e-8 * max(full[1], s)
    if m <= 200:
        (t1, c1, _), info = fo.splprep(X, u, s)
        np.testing.assert_array_equal(tck[0], t1)
        np.testing.assert_allclose(np.asarray(tck[1]), np.asarray(c1), rtol=0, atol=COEF_ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize('seed,m,spec', [(9, 20000, ('abs', 24.0)), (10, 30000, ('abs', 40.0)), (11, 12000, ('abs', 10.0))])
def test_gpu_smooth_fit_partitioned_band_solver(seed, m, spec):
    """Several hundred knots: the banded systems go through k_band_solve_parts (interiors in parallel, separator system,
    back-substitution; from 192 coefficients on) and the initial p through k_band_diag_sum -- still FITPACK's knots and spline.
    (MVUS_BAND_PARTS_MIN=16 pushes every case of this file through the same kernels.)"""
    from mvus_amd import spline
    u, X = _trajectory(seed, m, speed=1 + seed % 7)
    s = _s_of(u, spec)
    (tck0, _u), fp0, ier0, _msg = interpolate.splprep(X, u=u, s=s, k=3, full_output=1)
    assert len(tck0[0]) > 300
    tck, fp, ier = spline.smooth_fit(u, X, s, full_output=True)
    np.testing.assert_array_equal(tck[0], tck0[0])
    assert ier == ier0
    np.testing.assert_allclose(np.asarray(tck[1]), np.asarray(tck0[1]), rtol=0, atol=1e-7)
    assert abs(fp - fp0) <= 1e-7 * max(fp0, s)


@pytest.mark.gpu
def test_gpu_smooth_fit_session_is_the_one_shot_fit():
    """mvus_spline_fit_open / _smooth / _close (the samples uploaded once, work arrays kept: what traj_fit's smooth_factor loop
    uses) gives, for every s and in any order of s, exactly what one-shot mvus_spline_smooth gives."""
    from mvus_amd import spline
    u, X = _trajectory(12, 6000, speed=3)
    ss = [40.0, 5.0, 800.0, 7.2, 0.5, 40.0]                 # knot counts up and down: the work arrays grow, are reused, hold stale data
    with spline.SmoothFit(u, X) as fit:
        for s in ss:
            tck, fp, ier = fit(s, full_output=True)
            tck1, fp1, ier1 = spline.smooth_fit(u, X, s, full_output=True)
            np.testing.assert_array_equal(tck[0], tck1[0])
            np.testing.assert_array_equal(np.asarray(tck[1]), np.asarray(tck1[1]))
            assert (fp, ier) == (fp1, ier1)
    with pytest.raises(ValueError):
        fit(1.0)                                           # closed
    with pytest.raises(ValueError):
        spline.SmoothFit(u[::-1].copy(), X)                # the checks of the one-shot call happen at open


@pytest.mark.gpu
def test_gpu_smooth_fit_ill_conditioned_knot_set():
    """Close to interpolation FITPACK's knot search can produce a knot set with cond(A) ~ 1e10 (cond of the normal equations
    1e20): the fp64 Cholesky loses a pivot, the pass is repeated in double-double and still lands on FITPACK's spline."""
    from mvus_amd import spline
    rng = np.random.default_rng(0)
    for case in range(3):                                  # the third draw of this stream (m = 80) is the ill-conditioned one
        m = int(rng.integers(60, 400))
        u = np.cumsum(rng.uniform(0.5, 1.5, m))
        X = np.vstack([10 * np.sin(u / 80 * (1 + case)), 10 * np.cos(u / 95), 30 + 3 * np.sin(u / 50)]) + rng.normal(0, 0.02, (3, m))
    s = 1e-4 * (u[-1] - u[0])
    (t0, c0, _), _u = interpolate.splprep(X, u=u, s=s, k=3)
    A = interpolate.BSpline.design_matrix(u, t0, 3).toarray()
    sv = np.linalg.svd(A, compute_uv=False)
    assert sv[0] / sv[-1] > 1e9                            # the premise of this test
    tck, fp, ier = spline.smooth_fit(u, X, s, full_output=True)
    np.testing.assert_array_equal(tck[0], t0)
    np.testing.assert_allclose(np.asarray(tck[1]), np.asarray(c0), rtol=0, atol=1e-6)
    assert abs(fp - s) < 1e-3 * s and ier == 0


@pytest.mark.gpu
def test_gpu_traj_to_spline_matches_the_reference_fixture():
    """Scene.traj_to_spline with the fit on the GPU (the default) against the REAL reference's output: two intervals, the whole
    smooth_factor loop (a dozen fits per interval from the interpolating regime upwards)."""
    from mvus_amd import spline
    g = _golden()
    s = common.Scene()
    s.settings = {}
    s.traj = g['traj'].copy()
    sp = s.traj_to_spline(smooth_factor=list(g['smooth_factor']))
    np.testing.assert_array_equal(sp['int'], g['interval'])
    assert len(sp['tck']) == int(g['n_int'])
    for i, tck in enumerate(sp['tck']):
        np.testing.assert_array_equal(tck[0], g['knots_%d' % i])
        np.testing.assert_allclose(np.asarray(tck[1]), g['coefs_%d' % i], rtol=0, atol=COEF_ATOL)
    # every fit of the loop, not only the accepted one: same knot count as scipy at every smoothing factor it walks through
    for i, part in _parts(g):
        seen = []
        _reference_loop(part, g['smooth_factor'], lambda X, u, s_: (seen.append((s_, len(spline.smooth_fit(u, X, s_)[0]))), interpolate.splprep(X, u=u, s=s_, k=3)[0])[1])
        for s_, n_gpu in seen:
            assert n_gpu == len(interpolate.splprep(part[1:], u=part[0], s=s_, k=3)[0][0])


def _short():
    return dict(np.load(os.path.join(GOLDEN_DIR, 'traj_spline_short.npz')))


def test_short_part_fallback_is_the_reference_curve():
    """A part with three samples (the only size below four that find_intervals lets through): the reference's splprep(k=3)
    raises and its bare `except` fits a k=1 spline (common.py:266-267).  mvus_amd.spline.linear_fit_as_cubic restates
    FITPACK's k=1 answer and returns the same curve degree-elevated to a cubic; against the REAL reference's tck
    (tests/golden/traj_spline_short.npz: a bent part -- smoothing spline with one knot -- and a straight one -- the
    least-squares line) and against scipy over a sweep of shapes and scales."""
    from mvus_amd import spline
    g = _short()
    assert list(g['degree']) == [1, 3, 1]
    traj = g['traj']
    for i in (0, 2):
        a, b = g['interval'][:, i]
        part = traj[:, (traj[0] >= a) & (traj[0] <= b)]
        assert part.shape[1] == 3
        tck = spline.linear_fit_as_cubic(part, 1e-6 * (b - a))
        assert tck[2] == 3 and len(tck[0]) == (11 if i == 0 else 8)
        ts = np.linspace(a, b, 101)
        ref = np.asarray(interpolate.splev(ts, [g['knots_%d' % i], list(g['coefs_%d' % i]), 1]))
        np.testing.assert_allclose(np.asarray(interpolate.splev(ts, tck)), ref, rtol=0, atol=1e-12)
    rng = np.random.default_rng(0)
    for trial in range(200):
        u = np.cumsum(rng.uniform(2.5, 4.9, 3))
        X = rng.normal(size=(3, 3)) * rng.choice([1e-3, 1e-1, 1, 10])
        if trial % 3 == 0:                                  # nearly straight: both sides of the fp0 < s decision
            X = X[:, :1] + np.outer(rng.normal(size=3), u - u[0]) + 1e-4 * rng.normal(size=(3, 3)) * rng.choice([1, 0.1, 10])
        s_ = 1e-6 * (u[-1] - u[0])
        tck1 = interpolate.splprep(X, u=u, s=s_, k=1)[0]
        tck = spline.linear_fit_as_cubic(np.vstack((u, X)), s_)
        assert len(tck[0]) == {4: 8, 5: 11}[len(tck1[0])]
        ts = np.linspace(u[0], u[-1], 40)
        np.testing.assert_allclose(np.asarray(interpolate.splev(ts, tck)), np.asarray(interpolate.splev(ts, tck1)), rtol=0, atol=1e-8)
    with pytest.raises(ValueError):
        spline.linear_fit_as_cubic(np.array([[0.0], [1.0], [2.0], [3.0]]), 1e-6)       # one sample: the reference raises too


@pytest.mark.gpu
def test_gpu_traj_to_spline_with_short_parts_matches_the_reference():
    """Scene.traj_to_spline / spline_to_traj on a trajectory with two three-sample parts around a regular one, against the
    REAL reference's outputs: same intervals, the regular part's knots and coefficients, and every sampled point of all three
    parts (the short ones through the cubic twin of the reference's k=1 spline, triple interior knot included)."""
    g = _short()
    s = common.Scene()
    s.settings = {}
    s.traj = g['traj'].copy()
    sp = s.traj_to_spline(smooth_factor=[10, 20])
    np.testing.assert_array_equal(sp['int'], g['interval'])
    np.testing.assert_array_equal(sp['tck'][1][0], g['knots_1'])
    np.testing.assert_allclose(np.asarray(sp['tck'][1][1]), g['coefs_1'], rtol=0, atol=COEF_ATOL)
    traj = s.spline_to_traj(sampling_rate=0.25)
    assert traj.shape == g['traj_rate'].shape
    np.testing.assert_array_equal(traj[0], g['traj_rate'][0])
    np.testing.assert_allclose(traj[1:], g['traj_rate'][1:], rtol=0, atol=1e-8)
    tq = s.spline_to_traj(t=g['t_query'])
    np.testing.assert_array_equal(tq[0], g['traj_query'][0])
    np.testing.assert_allclose(tq[1:], g['traj_query'][1:], rtol=0, atol=1e-8)
    # ... and a BA over such a spline is refused with a message that says why: the straight part ([a x4, b x4]) is an ordinary cubic,
    # the bent one has a triple interior knot, for which neither the reference's jac_BA nor the kernels' pattern codes are defined
    from mvus_amd import problem as mp
    from mvus_amd.ba import BAHandle
    from mvus_amd import synth
    sc = synth.make_scene(2, 400, seed=3, knot_spacing=15.0)
    for d in sc.detections:
        d[0] = np.linspace(-5.0, 150.0, d.shape[1])
    sc.alpha[:] = 1.0
    sc.beta[:] = 0.0
    sc.tck = [[t[0].copy(), [c.copy() for c in t[1]], 3] for t in sp['tck']]
    sc.interval = sp['int'].copy()
    prob, x0 = mp.problem_from_scene(sc)
    with pytest.raises(ValueError, match='repeated interior knots'):
        BAHandle(prob)
    sc.tck, sc.interval = sc.tck[1:], sc.interval[:, 1:]                   # without the bent part: regular + straight three-sample part
    prob, x0 = mp.problem_from_scene(sc)
    from oracle import ba_oracle as orc
    oprob, _ = orc.problem_from_scene(sc)
    with BAHandle(prob) as h:
        f = h.residual(x0)
    fo_ = orc.residual(oprob, x0)
    assert np.array_equal(f == 0, fo_ == 0) and (f != 0).sum() > 100
    np.testing.assert_allclose(f, fo_, rtol=1e-9, atol=1e-7)


@pytest.mark.gpu
def test_gpu_smooth_fit_rejects_bad_input():
    from mvus_amd import spline
    u, X = _trajectory(0, 50)
    with pytest.raises(ValueError):
        spline.smooth_fit(u[::-1].copy(), X, 1.0)          # not increasing
    with pytest.raises(ValueError):
        spline.smooth_fit(u, X, 0.0)                       # s must be positive (the reference never asks for interpolation)
    Xn = X.copy(); Xn[1, 7] = np.nan
    with pytest.raises(ValueError):
        spline.smooth_fit(u, Xn, 1.0)
    with pytest.raises(ValueError):
        spline.smooth_fit(u[:3], X[:, :3], 1.0)            # fewer samples than a cubic needs
    with pytest.raises(ValueError):
        spline.smooth_fit(u, X[:2], 1.0)


@pytest.mark.gpu
def test_gpu_smooth_fit_full_size_properties():
    """A 10-minute flight at 30 fps (18 000 samples; scipy needs minutes here, its smoothing iteration is O(n^2)): properties
    FITPACK guarantees -- |fp - s| within its tolerance, interior knots are data sites, strictly increasing, clamped ends --
    and the fit is what scipy's splev evaluates to within the noise."""
    import time
    from mvus_amd import spline
    u, X = _trajectory(11, 18000, noise=0.05)
    for s in (1e-6 * (u[-1] - u[0]), 18000 * 0.05 ** 2 * 3):
        t0 = time.time()
        tck, fp, ier = spline.smooth_fit(u, X, s, full_output=True)
        elapsed = time.time() - t0
        t = tck[0]
        assert ier == 0 and abs(fp - s) <= 1e-3 * s
        assert np.all(t[:4] == u[0]) and np.all(t[-4:] == u[-1]) and np.all(np.diff(t[3:-3]) > 0)
        assert np.all(np.isin(t[4:-4], u))
        fit = np.asarray(interpolate.splev(u, tck))
        assert abs(np.sum((fit - X) ** 2) - fp) <= 1e-6 * fp
        assert elapsed < 30.0
