"""GPU tests of the normal-equation assembly and the LM + Schur solver against dense host algebra."""
import numpy as np
import pytest

from oracle import ba_oracle as orc
from golden_util import CASES, load_case
from mvus_amd import _lib
from mvus_amd import problem as mp

pytestmark = pytest.mark.gpu


def _host(prob):
    from hostcheck_util import HostHandle
    return HostHandle(prob)


def internal_index(prob):
    """x index of every unknown in the solver's internal order: camera blocks (alpha,beta,rs,params), then 3*ctrl+xyz."""
    C, P = prob.C, prob.P
    cam = [[c, C + c, 2 * C + c] + list(range(3 * C + c * P, 3 * C + (c + 1) * P)) for c in range(C)]
    spl = []
    for s, n in enumerate(prob.n_coef):
        for j in range(int(n)):
            spl += [int(prob.spline_x_offsets[s]) + d * int(n) + j for d in range(3)]
    return np.array(cam), np.array(spl)


@pytest.mark.parametrize('name', CASES)
def test_normal_equations_match_dense(name):
    from mvus_amd.ba import BAHandle
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    x = g['x0'] + g['delta']
    f, D = _host(prob).dense_jacobian(x, _lib.JAC_ANALYTIC)
    H, grad = D.T @ D, D.T @ f
    cam_idx, spl_idx = internal_index(prob)
    with BAHandle(prob) as h:
        h.residual_jacobian(x, _lib.JAC_ANALYTIC)
        gg, A, band, cross = h.normal_equations()
    scale = np.abs(H).max()
    np.testing.assert_allclose(gg, grad, rtol=0, atol=1e-11 * np.abs(grad).max())
    for c in range(prob.C):
        np.testing.assert_allclose(A[c], H[np.ix_(cam_idx[c], cam_idx[c])], rtol=0, atol=1e-12 * scale)
    Hs = H[np.ix_(spl_idx, spl_idx)]
    N, W = band.shape[0], band.shape[1]
    covered = np.zeros_like(Hs, dtype=bool)
    for gi in range(N):
        for w in range(W):
            if gi + w < N:
                blk = Hs[3 * gi:3 * gi + 3, 3 * (gi + w):3 * (gi + w) + 3]
                np.testing.assert_allclose(band[gi, w], blk, rtol=0, atol=1e-12 * scale)
                covered[3 * gi:3 * gi + 3, 3 * (gi + w):3 * (gi + w) + 3] = True
                covered[3 * (gi + w):3 * (gi + w) + 3, 3 * gi:3 * gi + 3] = True
    assert not Hs[~covered].any()                                   # nothing outside the band
    E = H[np.ix_(cam_idx.ravel(), spl_idx)]
    np.testing.assert_allclose(cross.reshape(E.shape), E, rtol=0, atol=1e-12 * scale)


@pytest.mark.parametrize('name', CASES)
def test_lm_schur_gpu_matches_host_dense_lm(name):
    from mvus_amd.ba import BAHandle
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 8)
    xh, rh, fh = _host(prob).solve(g['x0'], opts)
    with BAHandle(prob) as h:
        r = h.solve(g['x0'], opts=opts)
    assert (r.nfev, r.njev, r.status) == (rh.nfev, rh.njev, rh.status)
    np.testing.assert_allclose(r.cost, rh.cost, rtol=1e-7)
    np.testing.assert_allclose(r.x, xh, rtol=0, atol=1e-5 * max(1.0, np.abs(xh).max()))


@pytest.mark.parametrize('name', CASES)
def test_lm_schur_final_fit_not_worse_than_reference(name):
    """Second BA of main.py:59 (inliers only): LM with the exact Jacobian must reach at least the reference's cost."""
    from mvus_amd.ba import BAHandle
    scene, g = load_case(name)
    keep, off = g['outlier_keep'].astype(bool), g['det_offsets']
    for i in range(scene.num_cam):
        scene.detections[i] = scene.detections[i][:, keep[off[i]:off[i + 1]]]
    prob, _ = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    with BAHandle(prob) as h:
        r = h.solve(g['ba2_10_x0'], solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=40)
        np.testing.assert_allclose(0.5 * np.sum(orc.residual(oprob, r.x) ** 2), r.cost, rtol=1e-9)
    assert r.cost < float(g['ba2_10_cost'])
    if prob.rs_bounds:
        rs = r.x[2 * prob.C:3 * prob.C]
        assert np.all((rs >= 0) & (rs <= 1))


@pytest.mark.parametrize('motion', [False, True])
def test_partitioned_band_solver_many_partitions(motion):
    """~300 control points -> ~9 interiors + separators; LM steps must equal the dense host LM."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(3, 4000, seed=17, rolling_shutter=True, num_knots=300, motion_reg=motion, motion_type='F',
                          motion_weights=50.0)
    prob, x0 = mp.problem_from_scene(sc)
    assert int(prob.n_coef.sum()) > 250
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 4)
    xh, rh, fh = _host(prob).solve(x0, opts)
    with BAHandle(prob) as h:
        r = h.solve(x0, opts=opts)
    assert (r.nfev, r.njev, r.status) == (rh.nfev, rh.njev, rh.status)
    np.testing.assert_allclose(r.cost, rh.cost, rtol=1e-7)
    np.testing.assert_allclose(r.x, xh, rtol=0, atol=1e-5 * max(1.0, np.abs(xh).max()))


@pytest.mark.parametrize('spacing,motion_type', [(0.8, 'F'), (0.45, 'F'), (0.3, 'KE'), (0.22, 'F')])
def test_wide_band_knots_closer_than_a_frame(spacing, motion_type):
    """FITPACK knots less than one frame apart (what traj_to_spline returns after a dense triangulate, common.py:224-270): the three
    samples of a motion row (common.py:959-1001) then span more than three knot spans and the spline block has more than six 3x3
    blocks per row.  The LM solver keeps the band as it is (general band Cholesky instead of the partitioned solver): normal equations
    against the dense J^T J, LM steps against the dense host LM."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(3, 420, seed=61, rolling_shutter=True, knot_spacing=spacing, motion_reg=True, motion_type=motion_type, motion_weights=40.0)
    prob, x0 = mp.problem_from_scene(sc)
    f, D = _host(prob).dense_jacobian(x0, _lib.JAC_ANALYTIC)
    H, grad = D.T @ D, D.T @ f
    cam_idx, spl_idx = internal_index(prob)
    with BAHandle(prob) as h:
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        gg, A, band, cross = h.normal_equations()
        N, W = band.shape[0], band.shape[1]
        assert W > 6 or spacing >= 0.8
        scale = np.abs(H).max()
        np.testing.assert_allclose(gg, grad, rtol=0, atol=1e-11 * np.abs(grad).max())
        Hs = H[np.ix_(spl_idx, spl_idx)]
        covered = np.zeros_like(Hs, dtype=bool)
        for gi in range(N):
            for w in range(W):
                if gi + w < N:
                    np.testing.assert_allclose(band[gi, w], Hs[3 * gi:3 * gi + 3, 3 * (gi + w):3 * (gi + w) + 3], rtol=0, atol=1e-12 * scale)
                    covered[3 * gi:3 * gi + 3, 3 * (gi + w):3 * (gi + w) + 3] = True
                    covered[3 * (gi + w):3 * (gi + w) + 3, 3 * gi:3 * gi + 3] = True
        assert not Hs[~covered].any()                                   # nothing outside the band
        p = h.lm_step(0.1)
    Hd = H + 0.1 * np.diag(np.where(np.diag(H) > 0, np.diag(H), 1.0))
    p_ref = -np.linalg.solve(Hd, grad)
    np.testing.assert_allclose(p, p_ref, rtol=0, atol=1e-7 * np.abs(p_ref).max())
    if spacing < 0.3:
        return                                                          # (the dense host LM below is O(n^3) per step)
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 5)
    opts.lm_lambda_min = 0.3               # (the library's floor for bands wider than six control points; the host harness takes it from here)
    xh, rh, fh = _host(prob).solve(x0, opts)
    with BAHandle(prob) as h:
        r = h.solve(x0, opts=opts)
    assert (r.nfev, r.njev, r.status) == (rh.nfev, rh.njev, rh.status)
    np.testing.assert_allclose(r.cost, rh.cost, rtol=1e-7)
    np.testing.assert_allclose(r.x, xh, rtol=0, atol=1e-5 * max(1.0, np.abs(xh).max()))


def test_band_beyond_sixteen_control_points_is_an_error_code():
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle, UnsupportedBySolver
    sc = synth.make_scene(2, 300, seed=62, knot_spacing=0.08, motion_reg=True, motion_type='F', motion_weights=40.0)
    prob, x0 = mp.problem_from_scene(sc)
    with BAHandle(prob) as h:
        with pytest.raises(UnsupportedBySolver):
            h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=3)
        r = h.solve(x0, solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=3)      # the other solver has no such limit
        assert np.isfinite(r.cost)


@pytest.mark.parametrize('num_knots', [75, 140, 210, 280, 420, 560])
@pytest.mark.parametrize('sequential', [False, True])
def test_separator_chain_lengths(num_knots, sequential, monkeypatch):
    """1, 2, 3, 4, 6 and 8 separators (odd, even, power of two): the cyclic-reduction separator solve and the
    sequential block-tridiagonal one (MVUS_SEP_SEQUENTIAL, also the path for very long chains) both reproduce the
    dense host LM."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    if sequential:
        monkeypatch.setenv('MVUS_SEP_SEQUENTIAL', '1')
    sc = synth.make_scene(3, 3000, seed=31, rolling_shutter=True, num_knots=num_knots)
    prob, x0 = mp.problem_from_scene(sc)
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 3)
    xh, rh, fh = _host(prob).solve(x0, opts)
    with BAHandle(prob) as h:
        r = h.solve(x0, opts=opts)
    assert (r.nfev, r.njev, r.status) == (rh.nfev, rh.njev, rh.status)
    np.testing.assert_allclose(r.cost, rh.cost, rtol=1e-7)
    np.testing.assert_allclose(r.x, xh, rtol=0, atol=1e-5 * max(1.0, np.abs(xh).max()))


def test_many_cameras_streaming_back_substitution():
    """40 cameras -> a 360 x 360 reduced camera system: more than the register-resident back substitution holds
    (10 panels of 32), so the streaming one runs; 12 panels of the dense factorisation."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(40, 6000, seed=37, rolling_shutter=True, num_knots=30)
    prob, x0 = mp.problem_from_scene(sc)
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 3)
    xh, rh, fh = _host(prob).solve(x0, opts)
    with BAHandle(prob) as h:
        r = h.solve(x0, opts=opts)
    assert (r.nfev, r.njev, r.status) == (rh.nfev, rh.njev, rh.status)
    np.testing.assert_allclose(r.cost, rh.cost, rtol=1e-7)
    np.testing.assert_allclose(r.x, xh, rtol=0, atol=1e-5 * max(1.0, np.abs(xh).max()))


def test_unsorted_and_sparse_detections_take_the_atomic_fallback():
    """Detections shuffled locally (knot spans interleave in index order: sorted by span in LDS), shuffled globally and a
    camera with huge frame gaps (more spans than a chunk window holds: deferred to the atomic kernels); results are
    unchanged."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(3, 3000, seed=23, rolling_shutter=True, num_knots=400)
    rng = np.random.default_rng(5)
    n0 = sc.detections[0].shape[1]
    local = np.concatenate([b + rng.permutation(min(16, n0 - b)) for b in range(0, n0, 16)])
    sc.detections[0] = sc.detections[0][:, local]                                            # spans interleave, window fits
    sc.detections[1] = sc.detections[1][:, rng.permutation(sc.detections[1].shape[1])]       # unsorted
    sc.detections[2] = sc.detections[2][:, ::9]                                              # sparse: ~27 frames apart
    prob, x0 = mp.problem_from_scene(sc)
    f, D = _host(prob).dense_jacobian(x0, _lib.JAC_ANALYTIC)
    H, grad = D.T @ D, D.T @ f
    cam_idx, spl_idx = internal_index(prob)
    with BAHandle(prob) as h:
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        gg, A, band, cross = h.normal_equations()
        u = rng.normal(size=h.m)
        z = h.jtu(u)
    scale = np.abs(H).max()
    np.testing.assert_allclose(gg, grad, rtol=0, atol=1e-11 * np.abs(grad).max())
    np.testing.assert_allclose(z, D.T @ u, rtol=0, atol=1e-11 * np.abs(D.T @ u).max())
    for c in range(prob.C):
        np.testing.assert_allclose(A[c], H[np.ix_(cam_idx[c], cam_idx[c])], rtol=0, atol=1e-12 * scale)
    E = H[np.ix_(cam_idx.ravel(), spl_idx)]
    np.testing.assert_allclose(cross.reshape(E.shape), E, rtol=0, atol=1e-12 * scale)
    Hs = H[np.ix_(spl_idx, spl_idx)]
    for gi in range(band.shape[0]):
        for w in range(band.shape[1]):
            if gi + w < band.shape[0]:
                np.testing.assert_allclose(band[gi, w], Hs[3 * gi:3 * gi + 3, 3 * (gi + w):3 * (gi + w) + 3], rtol=0, atol=1e-12 * scale)



def test_lm_step_matches_sparse_direct_solve_on_a_long_chain():
    """~2300 control points -> ~65 separators, 7 levels of cyclic reduction: the damped step of the whole GPU chain
    (mvus_ba_lm_step: assembly, band solver, Schur complement, reduced system) must solve (H + lambda diag H) p = -g with
    H and g taken from mvus_ba_normal_equations and the system solved by scipy's sparse LU on the CPU."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(3, 12000, seed=47, rolling_shutter=True, num_knots=2400)
    prob, x0 = mp.problem_from_scene(sc)
    N = int(prob.n_coef.sum())
    assert N > 2200
    cam_idx, spl_idx = internal_index(prob)
    lams = (1e-4, 3.0)
    with BAHandle(prob) as h:
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        g, A, band, cross = h.normal_equations()
        steps = [h.lm_step(lam) for lam in lams]
    n, C, B, W = prob.n_params, prob.C, 3 + prob.P, band.shape[1]
    rows, cols, vals = [], [], []
    for c in range(C):                                        # camera blocks
        ii, jj = np.meshgrid(cam_idx[c], cam_idx[c], indexing='ij')
        rows.append(ii.ravel()); cols.append(jj.ravel()); vals.append(A[c].ravel())
    E = cross.reshape(C * B, 3 * N)                           # cross block and its transpose
    ci = cam_idx.ravel()
    nz = np.nonzero(E)
    rows += [ci[nz[0]], spl_idx[nz[1]]]; cols += [spl_idx[nz[1]], ci[nz[0]]]; vals += [E[nz], E[nz]]
    for w in range(W):                                        # spline band (upper blocks + mirror)
        for gi in range(N - w):
            blk = band[gi, w]
            ri, cj = spl_idx[3 * gi:3 * gi + 3], spl_idx[3 * (gi + w):3 * (gi + w) + 3]
            ii, jj = np.meshgrid(ri, cj, indexing='ij')
            rows.append(ii.ravel()); cols.append(jj.ravel()); vals.append(blk.ravel())
            if w > 0:
                rows.append(jj.ravel()); cols.append(ii.ravel()); vals.append(blk.ravel())
    H = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    d = H.diagonal()
    d = np.where(d > 0, d, 1.0)
    for lam, p in zip(lams, steps):
        p_ref = spla.spsolve(H + lam * sp.diags(d), -g)
        np.testing.assert_allclose(p, p_ref, rtol=0, atol=1e-7 * np.abs(p_ref).max())


def test_damped_step_for_every_tail_length():
    """Spline lengths from 2 to 4 partitions, one control point at a time: the last interior takes every length (and the
    tail-merge rule triggers): the damped step must match a dense solve of the exported normal equations each time."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    worst = 0.0
    seen = set()
    for nk in range(72, 150):
        sc = synth.make_scene(2, 700, seed=59, rolling_shutter=True, num_knots=nk)
        prob, x0 = mp.problem_from_scene(sc)
        N = int(prob.n_coef.sum())
        if N in seen:
            continue
        seen.add(N)
        cam_idx, spl_idx = internal_index(prob)
        with BAHandle(prob) as h:
            h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
            g, A, band, cross = h.normal_equations()
            p = h.lm_step(0.2)
        n, C, B, W = prob.n_params, prob.C, 3 + prob.P, band.shape[1]
        H = np.zeros((n, n))
        for c in range(C):
            H[np.ix_(cam_idx[c], cam_idx[c])] = A[c]
        E = cross.reshape(C * B, 3 * N)
        H[np.ix_(cam_idx.ravel(), spl_idx)] = E
        H[np.ix_(spl_idx, cam_idx.ravel())] = E.T
        for w in range(W):
            for gi in range(N - w):
                ri, cj = spl_idx[3 * gi:3 * gi + 3], spl_idx[3 * (gi + w):3 * (gi + w) + 3]
                H[np.ix_(ri, cj)] = band[gi, w]
                H[np.ix_(cj, ri)] = band[gi, w].T
        d = np.diag(H).copy()
        p_ref = np.linalg.solve(H + 0.2 * np.diag(np.where(d > 0, d, 1.0)), -g)
        worst = max(worst, float(np.abs(p - p_ref).max() / np.abs(p_ref).max()))
    assert len(seen) > 60 and worst < 1e-8, (len(seen), worst)


@pytest.mark.parametrize('seed', range(8))
def test_assembly_randomised_layouts(seed):
    """Randomised detection layouts -- density from a fraction of a detection to ~50 per knot span, random local shuffles,
    random dropped stretches, detections outside the spline, with and without free calibration -- against the dense J^T J
    of the host build (ranges of one span of every length, the in-LDS sort, split runs, the many-ranges path)."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    rng = np.random.default_rng(100 + seed)
    calib = bool(seed % 2)
    nk = int(rng.choice([40, 90, 400, 1500]))
    sc = synth.make_scene(3, int(rng.choice([1500, 4000])), seed=200 + seed, rolling_shutter=True, num_knots=nk,
                          distortion=calib, opt_calib=calib)
    for c in range(3):
        d = sc.detections[c]
        n = d.shape[1]
        mode = rng.integers(0, 4)
        if mode == 1:                                    # local shuffles: spans interleave
            blk = int(rng.choice([4, 16, 64]))
            d = d[:, np.concatenate([b + rng.permutation(min(blk, n - b)) for b in range(0, n, blk)])]
        elif mode == 2:                                  # random stretches dropped + a block shifted outside every interval
            keep = np.ones(n, bool)
            for _ in range(5):
                a = rng.integers(0, n)
                keep[a:a + rng.integers(1, n // 6)] = False
            d = d[:, keep].copy()
            d[0, :7] += 1e6
        elif mode == 3:                                  # thinned: many spans per half chunk
            d = d[:, ::int(rng.choice([2, 5, 11]))]
        sc.detections[c] = d
    prob, x0 = mp.problem_from_scene(sc)
    if prob.n_params > 5000:
        pytest.skip('dense host Jacobian too large for this draw')
    f, D = _host(prob).dense_jacobian(x0, _lib.JAC_ANALYTIC)
    H, grad = D.T @ D, D.T @ f
    cam_idx, spl_idx = internal_index(prob)
    with BAHandle(prob) as h:
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        gg, A, band, cross = h.normal_equations()
        u, v = rng.normal(size=h.m), rng.normal(size=h.n)
        z, y = h.jtu(u), h.jv(v)                              # the operator pair of the TRF + LSMR path on the same layout
    np.testing.assert_allclose(z, D.T @ u, rtol=0, atol=1e-11 * np.abs(D.T @ u).max())
    np.testing.assert_allclose(y, D @ v, rtol=0, atol=1e-11 * np.abs(D @ v).max())
    scale = np.abs(H).max()
    np.testing.assert_allclose(gg, grad, rtol=0, atol=1e-11 * np.abs(grad).max())
    for c in range(prob.C):
        np.testing.assert_allclose(A[c], H[np.ix_(cam_idx[c], cam_idx[c])], rtol=0, atol=1e-12 * scale)
    E = H[np.ix_(cam_idx.ravel(), spl_idx)]
    np.testing.assert_allclose(cross.reshape(E.shape), E, rtol=0, atol=1e-12 * scale)
    Hs = H[np.ix_(spl_idx, spl_idx)]
    for gi in range(band.shape[0]):
        for w in range(band.shape[1]):
            if gi + w < band.shape[0]:
                np.testing.assert_allclose(band[gi, w], Hs[3 * gi:3 * gi + 3, 3 * (gi + w):3 * (gi + w) + 3], rtol=0, atol=1e-12 * scale)


@pytest.mark.parametrize('slabs', [None, 1, 3, 8, 11, 24])
def test_schur_product_slab_plans(slabs, monkeypatch):
    """k_schur_gemm deals the K-slabs of E^T Z to the XCDs (slab s -> XCD s mod 8) and every wavefront takes a share of the 16-row
    sets of the product; the slab count is planned per problem (HipSchur::plan_gemm).  Forced counts -- one slab, fewer than the
    XCDs, not a multiple of eight, more slabs than whole sets per wavefront -- on 17 cameras (153 columns: four 48-wide tiles, the last
    one ragged) and a row count that is not a multiple of 16: the damped step must match a dense solve of the exported normal equations."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    if slabs is None:
        monkeypatch.delenv('MVUS_GEMM_SLABS', raising=False)
    else:
        monkeypatch.setenv('MVUS_GEMM_SLABS', str(slabs))
    sc = synth.make_scene(17, 9000, seed=61, rolling_shutter=True, num_knots=171)
    prob, x0 = mp.problem_from_scene(sc)
    N = int(prob.n_coef.sum())
    assert (3 * N) % 16 != 0
    cam_idx, spl_idx = internal_index(prob)
    with BAHandle(prob) as h:
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        g, A, band, cross = h.normal_equations()
        p = h.lm_step(0.05)
    n, C, B, W = prob.n_params, prob.C, 3 + prob.P, band.shape[1]
    H = np.zeros((n, n))
    for c in range(C):
        H[np.ix_(cam_idx[c], cam_idx[c])] = A[c]
    E = cross.reshape(C * B, 3 * N)
    H[np.ix_(cam_idx.ravel(), spl_idx)] = E
    H[np.ix_(spl_idx, cam_idx.ravel())] = E.T
    for w in range(W):
        for gi in range(N - w):
            ri, cj = spl_idx[3 * gi:3 * gi + 3], spl_idx[3 * (gi + w):3 * (gi + w) + 3]
            H[np.ix_(ri, cj)] = band[gi, w]
            H[np.ix_(cj, ri)] = band[gi, w].T
    d = np.diag(H).copy()
    p_ref = np.linalg.solve(H + 0.05 * np.diag(np.where(d > 0, d, 1.0)), -g)
    assert float(np.abs(p - p_ref).max() / np.abs(p_ref).max()) < 1e-8


def test_lm_trust_region_gpu_matches_host_driver():
    """mvus_solve_opts.lm_trust_radius on the GPU (|p|^2 by two small launches, the cut applied by the trial kernel) against the host
    build of the same driver: a radius that binds on the first steps and is widened / narrowed by scipy's rule afterwards."""
    from mvus_amd.ba import BAHandle
    scene, g = load_case('rs_F_2int_3cam')
    prob, x0 = mp.problem_from_scene(scene)
    radius = 1e-4 * float(np.linalg.norm(g['x0']))
    opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 8)
    opts.lm_trust_radius = radius
    one = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 2)
    one.lm_trust_radius = radius
    with BAHandle(prob) as h:
        r1 = h.solve(g['x0'], opts=one)
    np.testing.assert_allclose(np.linalg.norm(r1.x - g['x0']), radius, rtol=1e-9)
    with BAHandle(prob) as h:                                  # (fresh handle: the damping carried over from a previous solve starts equal)
        r = h.solve(g['x0'], opts=opts)
    xh, rh, _ = _host(prob).solve(g['x0'], opts)
    assert (r.nfev, r.status) == (rh.nfev, rh.status)
    np.testing.assert_allclose(r.cost, rh.cost, rtol=1e-7)
    np.testing.assert_allclose(r.x, xh, rtol=0, atol=1e-6 * max(1.0, np.abs(xh).max()))


def test_lm_resumes_from_the_point_it_returned_without_an_upload():
    """The LM driver keeps its current / trial point on the device from solve to solve: a caller that continues from the x the previous
    solve returned finds it there (bitwise comparison with the host copy) and x is not uploaded again.  The continued solve must give
    the same bits as one that had to upload the same point (here: after an intervening one-evaluation solve from another point)."""
    from mvus_amd.ba import BAHandle
    scene, g = load_case('rs_F_2int_3cam')
    prob, x0 = mp.problem_from_scene(scene)
    kw = dict(solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC)
    with BAHandle(prob) as h:
        r1 = h.solve(g['x0'], max_nfev=8, **kw)
        r2 = h.solve(r1.x.copy(), max_nfev=8, **kw)                      # resumed: the point is on the device
    with BAHandle(prob) as h:
        q1 = h.solve(g['x0'], max_nfev=8, **kw)
        h.solve(g['x0'], max_nfev=1, **kw)                                # no trial, no change of the damping; the remembered point is x0 now
        q2 = h.solve(q1.x.copy(), max_nfev=8, **kw)                      # uploaded
    assert np.array_equal(r1.x, q1.x) and r1.cost == q1.cost
    assert np.array_equal(r2.x, q2.x) and r2.cost == q2.cost and r2.nfev == q2.nfev
    assert r2.cost < r1.cost


def test_a_failed_solve_leaves_no_stale_resume_point():
    """The remembered point is disarmed as soon as it has been consulted (HipBackend::lm_resume): a solve that leaves early -- a
    non-finite cost at its x0, after that x0 went into the driver's buffer -- must not leave 'buffer k holds the point I returned'
    behind.  solve OK, solve from a point with an infinite cost (ValueError, as scipy raises), solve from the FIRST result again: the
    same bits as on a handle that never saw the failure."""
    from mvus_amd.ba import BAHandle
    scene, g = load_case('rs_F_2int_3cam')
    prob, x0 = mp.problem_from_scene(scene)
    kw = dict(solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC)
    bad = g['x0'].copy()
    bad[-1] = 1e200                                    # a control point far away: finite x, |f|^2 overflows
    with BAHandle(prob) as h:
        q1 = h.solve(g['x0'], max_nfev=8, **kw)
        q2 = h.solve(q1.x.copy(), max_nfev=8, **kw)
    for x_fail in (bad, np.where(np.arange(bad.size) == bad.size - 1, np.nan, g['x0'])):
        with BAHandle(prob) as h:
            r1 = h.solve(g['x0'], max_nfev=8, **kw)
            with pytest.raises(ValueError):
                h.solve(x_fail, max_nfev=8, **kw)
            r2 = h.solve(r1.x.copy(), max_nfev=8, **kw)
        assert np.array_equal(r1.x, q1.x)
        assert np.array_equal(r2.x, q2.x) and r2.cost == q2.cost and r2.nfev == q2.nfev


@pytest.mark.parametrize('name', ['rs_F_2int_3cam', 'c1_pinhole_2cam', 'calib_KE_bounds_3cam'])
def test_speculative_linearisation_and_carry_over_change_no_bit(name, monkeypatch):
    """Round 6: (a) the linearisation at a trial point is enqueued into a second set of blocks before the host knows whether the trial
    is accepted (HipSchur::linearize_spec), the fetch waits behind an event; (b) a caller that continues from the returned point finds
    f(x), its cost and -- after a solve that ended on its budget with an accepted trial -- the normal equations on the device
    (HipBackend::lm_carry).  Both change WHEN work is done, never a value: one long solve and a chain of two-evaluation calls (bench.py's
    steps) give the same bits as the sequential driver (MVUS_NO_SPEC=1, MVUS_LM_NO_CARRY=1), accepted and rejected trials alike."""
    from mvus_amd.ba import BAHandle
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    kw = dict(solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC)

    def run():
        out = []
        with BAHandle(prob) as h:
            o = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 14)
            o.lm_lambda_min = 0.0                            # (no floor: the damping falls until trials are rejected)
            r = h.solve(g['x0'], opts=o)
            out.append((r.x.copy(), r.cost, r.nfev, r.njev, r.status))
        with BAHandle(prob) as h:
            x = g['x0'].copy()
            for k in range(7):
                r = h.solve(x, max_nfev=2 if k != 3 else 3, **kw)
                out.append((r.x.copy(), r.cost, r.nfev, r.njev, r.status))
                x = r.x
            f = h.residual(x)                                # (another call in between: what the solve left behind must not be used)
            r = h.solve(x, max_nfev=3, **kw)
            out.append((r.x.copy(), r.cost, r.nfev, r.njev, r.status, 0.5 * float(f @ f)))
        return out

    for k in ('MVUS_NO_SPEC', 'MVUS_LM_NO_CARRY', 'MVUS_SQ_DEVICE_SUM', 'MVUS_FETCH_EVENT'):
        monkeypatch.delenv(k, raising=False)
    new = run()
    again = run()
    monkeypatch.setenv('MVUS_NO_SPEC', '1')
    monkeypatch.setenv('MVUS_LM_NO_CARRY', '1')
    monkeypatch.setenv('MVUS_SQ_DEVICE_SUM', '1')          # |f|^2 by k_dot_final instead of the host-side sum of the partials
    old = run()
    for a, b, c in zip(new, old, again):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:], (a[1:], b[1:])
        assert np.array_equal(a[0], c[0]) and a[1:] == c[1:]
    assert new[0][2] - 1 > new[0][3] - 1 or name != 'rs_F_2int_3cam', 'the long solve was meant to contain a rejected trial'
    assert new[-1][1] <= new[-1][5]                           # the last solve started from the cost the residual call saw


@pytest.mark.parametrize('cams,knots', [(3, 2400), (20, 5000)])
def test_wide_cyclic_reduction_levels_in_one_launch(cams, knots, monkeypatch, capfd):
    """Round 6: the wide levels of the separators' cyclic reduction run in ONE launch (k_sep_bcr_levels: a survivor waits for the marks
    of the three nodes it reads, write-through stores, agent-scope loads) instead of a launch per level.  Same bits as the launch-per-level
    form (MVUS_BCR_FUSED=0); and a hand-over time-out (MVUS_RCS_SPIN_LIMIT=0: the first poll gives up) repeats the solve on the
    launch-per-level route instead of failing it numerically -- again the same bits."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(cams, 4000 * cams, seed=50 + cams, rolling_shutter=True, num_knots=knots)
    prob, x0 = mp.problem_from_scene(sc)
    lams = (1e-3, 0.5)

    def steps(env):
        for k in ('MVUS_BCR_FUSED', 'MVUS_RCS_SPIN_LIMIT', 'MVUS_DEBUG'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with BAHandle(prob) as h:
            h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
            return [h.lm_step(lam) for lam in lams] + [h.lm_step(lams[0])]

    fused = steps({})
    per_level = steps({'MVUS_BCR_FUSED': '0'})
    capfd.readouterr()
    timed_out = steps({'MVUS_RCS_SPIN_LIMIT': '0', 'MVUS_DEBUG': '1'})
    err = capfd.readouterr().err
    assert err.count('hand-over time-out') == 1, err
    for a, b, c in zip(fused, per_level, timed_out):
        assert np.isfinite(a).all() and np.array_equal(a, b) and np.array_equal(a, c)
    assert np.array_equal(fused[0], fused[2])
