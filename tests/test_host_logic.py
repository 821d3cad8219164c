"""Host-side logic that surrounds the GPU path: interval helpers, parameter codec, Scene/Camera surface,
config parsing, observation sharding (no GPU needed)."""
import json
import os

import numpy as np
import pytest

from oracle import ba_oracle as orc
from golden_util import CASES, load_case
from mvus_amd import problem as mp, sharding, synth
from mvus_amd.reconstruction import common
from mvus_amd.tools import util


def test_find_intervals_and_sampling_match_oracle():
    rng = np.random.default_rng(0)
    for trial in range(50):
        x = np.sort(rng.choice(np.arange(0, 400), size=rng.integers(2, 200), replace=False)).astype(float)
        a, ia = util.find_intervals(x, idx=True)
        b, ib = orc.find_intervals(x, idx=True)
        assert np.array_equal(a, b) and np.array_equal(ia, ib)
        if a.shape[1]:
            ts = rng.uniform(-5, 405, 300)
            ts[:a.shape[1]] = a[0]          # exactly on a start (inside)
            ts[a.shape[1]:2 * a.shape[1]] = a[1]   # exactly on an end (outside: half-open)
            _, ids = util.sampling(ts, a, belong=True)
            assert np.array_equal(ids, orc.sampling_idx(ts, a))
            picked, mask = util.sampling(np.vstack((ts, ts, ts)), a)
            assert mask.dtype == bool and picked.shape[1] == mask.sum()
    assert util.find_intervals(np.array([1.0, 2.0, 3.0])).shape == (2, 0)      # shorter than the gap: dropped
    with pytest.raises(AssertionError):
        util.find_intervals(np.array([2.0, 1.0]))


@pytest.mark.parametrize('name', CASES)
def test_pack_unpack_roundtrip(name):
    scene, g = load_case(name)
    prob, x0 = mp.problem_from_scene(scene)
    np.testing.assert_allclose(x0, g['x0'], rtol=0, atol=1e-12)
    assert prob.n_params == x0.size and prob.n_residuals == g['f_x0'].size
    alpha, beta, rs, cams, coefs = mp.unpack_x(prob, x0)
    x1 = mp.pack_x(prob, alpha, beta, rs, cams, [[None, c, 3] for c in coefs])
    np.testing.assert_allclose(x1, x0, rtol=0, atol=1e-12)
    lb, ub = prob.bounds()
    assert np.isinf(lb).all() != prob.rs_bounds


def test_camera_codec_and_projection():
    rng = np.random.default_rng(1)
    K = np.array([[1000.0, 0, 960], [0, 1010.0, 540], [0, 0, 1]])
    R = synth.rodrigues(rng.normal(size=3))
    cam = common.Camera(K=K, d=np.array([0.1, -0.02, 1e-4, -1e-4, 0.0]), R=R, t=rng.normal(size=3), fps=30, resolution=[1920, 1080])
    cam.compose()
    v6, v15 = cam.P2vector(), cam.P2vector(calib=True)
    assert v6.size == 6 and v15.size == 15
    cam2 = common.Camera(K=K.copy(), d=cam.d.copy(), fps=30, resolution=[1920, 1080])
    cam2.vector2P(v15, calib=True)
    np.testing.assert_allclose(cam2.P, cam.P, atol=1e-9)
    X = rng.normal(size=(3, 5)) + np.array([[0], [0], [30.0]])
    x = cam.projectPoint(X)
    np.testing.assert_allclose(x[2], 1.0)
    K2, R2, t2 = cam2.decompose()
    np.testing.assert_allclose(K2, K, atol=1e-6)
    np.testing.assert_allclose(R2, R, atol=1e-9)
    # undistortion is the inverse of the forward lens model
    xn = rng.uniform(-0.3, 0.3, (2, 20))
    xd, yd = synth.distort(xn[0], xn[1], cam.d)
    raw = np.vstack((K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]))
    und = cam.undist_point(raw)
    np.testing.assert_allclose(und, np.vstack((K[0, 0] * xn[0] + K[0, 2], K[1, 1] * xn[1] + K[1, 2])), atol=2e-3)
    np.testing.assert_allclose(und, orc.undist_point(raw, K, cam.d), atol=1e-12)
    # OpenCV's early-out (negative 1 / (1 + k1 r^2 + ...): the point stays where it started) -- host copy == oracle
    cam.d = np.array([-0.9, 0.0, 0.0, 0.0, 0.0])
    far = np.vstack((K[0, 2] + 1.6 * K[0, 0] * np.ones(3), K[1, 2] + np.array([-0.2, 0.0, 0.3]) * K[1, 1]))
    np.testing.assert_allclose(cam.undist_point(far), orc.undist_point(far, K, cam.d), atol=1e-12)
    np.testing.assert_allclose(cam.undist_point(far), far, atol=1e-9)


def test_create_scene_reads_reference_config(tmp_path):
    rng = np.random.default_rng(2)
    dets, cams = [], []
    for i in range(2):
        d = np.column_stack((rng.uniform(0, 1920, 30), rng.uniform(0, 1080, 30), np.arange(30) + 5))   # x y frame
        p = tmp_path / ('det%d.txt' % i)
        np.savetxt(p, d)
        dets.append(str(p))
        c = tmp_path / ('cam%d.json' % i)
        c.write_text(json.dumps({'K-matrix': [[1000, 0, 960], [0, 1000, 540], [0, 0, 1]], 'distCoeff': [0.1, 0.01, 0, 0],
                                 'fps': 30 - 5 * i, 'resolution': [1920, 1080]}))
        cams.append(str(c))
    cfg = {'necessary inputs': {'path_detections': dets, 'path_cameras': cams, 'corresponding_frames': [10, 20]},
           'optional inputs': {'ground_truth': {'filepath': 'gt.txt', 'frequency': 5}},
           'settings': {'num_detections': 20, 'opt_calib': False, 'cf_exact': True, 'undist_points': True,
                        'rolling_shutter': True, 'init_rs': [0.5, 0.6], 'rs_bounds': False, 'motion_reg': False,
                        'motion_weights': 100, 'camera_sequence': [], 'ref_cam': 0}}
    path = tmp_path / 'config.json'
    path.write_text(json.dumps(cfg))
    flight = common.create_scene(str(path))
    assert flight.numCam == 2 and flight.find_order
    assert flight.detections[0].shape == (3, 20)
    np.testing.assert_array_equal(flight.detections[0][0], np.arange(20) + 5)        # row 0 = frame id
    assert flight.cameras[0].d.size == 5 and flight.cameras[0].d[4] == 0            # 4 coefficients are zero-padded
    np.testing.assert_array_equal(flight.rs, [0.5, 0.6])
    flight.init_alpha()
    np.testing.assert_allclose(flight.alpha, [1.0, 30 / 25])
    flight.time_shift()
    np.testing.assert_allclose(flight.beta, 10 - flight.alpha * np.array([10.0, 20.0]))
    flight.detection_to_global()
    assert flight.detections_global[1].shape == (3, 20)
    assert flight.gt['frequency'] == 5


def test_out_of_scope_methods_say_so():
    s = common.Scene()
    for name in ('init_traj',):
        with pytest.raises(NotImplementedError):
            getattr(s, name)()


def test_shard_offsets_partition():
    for count in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            pieces = [sharding.shard_offsets(count, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(pieces[:-1], pieces[1:]))
            sizes = [hi - lo for lo, hi in pieces]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('world', [2, 3])
def test_problem_shards_cover_every_observation_once(world):
    scene, g = load_case('rs_F_2int_3cam')
    prob, x0 = mp.problem_from_scene(scene)
    seen = np.zeros(prob.M, dtype=int)
    for r in range(world):
        shard, keep = prob.shard(r, world)
        seen[keep] += 1
        assert shard.M == keep.size and shard.n_params == prob.n_params
        assert np.array_equal(shard.frame, prob.frame[keep])
    assert (seen == 1).all()


def test_motion_sample_times_match_oracle():
    scene, g = load_case('rs_F_2int_3cam')
    prob, x0 = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    _, _, _, _, tck = orc.unpack_x(oprob, x0)
    traj = orc.spline_to_traj(oprob, tck)
    ts, sid = prob.motion_sample_times()
    np.testing.assert_array_equal(ts, traj[0])
    assert prob.num_motion_rows == traj.shape[1]


@pytest.mark.parametrize('name', ['rs_F_2int_3cam', 'c1_pinhole_2cam'])
def test_detection_spans_match_the_jacobian_support(name):
    """BAProblem.detection_spans (what time shards are cut by) = first control point with a non-zero Jacobian entry in the
    detection's row, from the host restatement of the device math."""
    from hostcheck_util import HostHandle
    from mvus_amd import _lib
    scene, g = load_case(name)
    prob, x0 = mp.problem_from_scene(scene)
    x = g['x0'] + g['delta']
    f, D = HostHandle(prob).dense_jacobian(x, _lib.JAC_ANALYTIC)
    spans = prob.detection_spans(x)
    # x column -> global control point
    col2ctrl = np.full(prob.n_params, -1)
    goff = 0
    for s, n in enumerate(prob.n_coef):
        for d in range(3):
            base = int(prob.spline_x_offsets[s]) + d * int(n)
            col2ctrl[base:base + int(n)] = goff + np.arange(int(n))
        goff += int(n)
    row = 0
    checked = 0
    for c in range(prob.C):
        a, b = int(prob.det_offsets[c]), int(prob.det_offsets[c + 1])
        for i in range(a, b):
            r = row + (i - a)                      # |ex| row of detection i
            nz = np.nonzero(D[r])[0]
            ctrl = col2ctrl[nz]
            ctrl = ctrl[ctrl >= 0]
            if ctrl.size == 0:
                assert spans[i] == -1 or abs(f[r]) == 0.0
                continue
            assert spans[i] >= 0 and spans[i] <= ctrl.min() and ctrl.max() <= spans[i] + 3
            checked += 1
        row += 2 * (b - a)
    assert checked > 0.5 * prob.M


@pytest.mark.parametrize('world', [2, 3, 8])
def test_time_shards_cover_every_observation_once(world):
    from mvus_amd import synth
    sc = synth.make_scene(4, 8000, seed=43, rolling_shutter=True, num_knots=400)
    prob, x0 = mp.problem_from_scene(sc)
    halo = 8
    cuts = prob.time_cuts(x0, world, halo)
    assert cuts[0] == 0 and cuts[-1] == int(prob.n_coef.sum()) and np.all(np.diff(cuts) >= 2 * halo + 8)
    spans = prob.detection_spans(x0)
    seen = np.zeros(prob.M, dtype=int)
    sizes = []
    for r in range(world):
        shard, keep, cuts_r = prob.shard_time(r, world, x0, halo)
        assert np.array_equal(cuts_r, cuts)
        seen[keep] += 1
        sizes.append(keep.size)
        vis = spans[keep] >= 0
        assert np.all((spans[keep][vis] >= cuts[r]) & (spans[keep][vis] < cuts[r + 1]))     # every visible detection in its owner's range
        assert shard.M == keep.size and shard.n_params == prob.n_params
        assert np.all(np.diff(shard.det_offsets) >= 0)
    assert (seen == 1).all()
    assert max(sizes) < 1.5 * prob.M / world                                                # balanced by detection count


def test_time_cuts_refuse_too_few_control_points():
    scene, g = load_case('c1_pinhole_2cam')
    prob, x0 = mp.problem_from_scene(scene)
    with pytest.raises(ValueError):
        prob.time_cuts(x0, 8, halo=8)


def test_band_solver_partition_invariants():
    """ba_partition.h: for every chain length, separator width and both chain kinds -- interiors and separators tile the chain in
    order, separators have the band half-width, NO interior between two separators is shorter than that (its separators
    would couple directly, outside the block-tridiagonal separator system), interiors fit the kernels' row limit (and that of the
    interior length asked for: HipSchur takes half-length interiors when the reduced system has few columns), and a closed chain (a
    time shard that is not the last) ends with a separator."""
    import ctypes
    import hostcheck_util
    lib = hostcheck_util.load()
    I = ctypes.c_int * 512
    lib.hostcheck_partition.restype = ctypes.c_int
    rows_max = 3 * (32 + 6)
    for sctrl, close, length in [(s_, c_, l_) for s_ in (3, 5) for c_ in (0, 1) for l_ in (32, 16)]:
        if True:
            for n in range(2 * 8 + sctrl + 1, 400):
                for c0 in (0, 11):
                    i0, i1, sep, nsep = I(), I(), I(), ctypes.c_int(0)
                    P = lib.hostcheck_partition(c0, n, sctrl, close, length, i0, i1, sep, ctypes.byref(nsep))
                    ns = nsep.value
                    assert P >= 1 and ns == (P if close else P - 1), (n, sctrl, close)
                    pos = 3 * c0
                    for k in range(P):
                        assert i0[k] == pos and i1[k] > i0[k] and i1[k] - i0[k] <= 3 * (length + 6) <= rows_max, (n, sctrl, close, k)
                        if 0 < k and k < ns:                      # an interior with a separator on both sides
                            assert i1[k] - i0[k] >= 3 * sctrl, (n, sctrl, close, k, i1[k] - i0[k])
                        pos = i1[k]
                        if k < ns:
                            assert sep[k] == pos
                            pos += 3 * sctrl
                    assert pos == 3 * (c0 + n), (n, sctrl, close)
                    if close and ns >= 2:                     # the interior before the closing separator also sits between two
                        assert i1[P - 1] - i0[P - 1] >= 3 * sctrl


def test_scene_default_is_the_reference_algorithm():
    """A reference config.json carries no ba_* key: Scene.BA then runs the reference's own algorithm (scipy TRF + LSMR over
    grouped 2-point differences, restated on the GPU).  The LM + Schur solver is opt-in, never a silent default -- it reaches
    another point and has no meaningful answer on an ill-posed opt_calib scene."""
    from mvus_amd import _lib
    from mvus_amd.reconstruction.common import Scene
    s = Scene()
    s.settings = {'opt_calib': True, 'undist_points': True, 'rolling_shutter': True, 'motion_reg': True, 'motion_type': 'KE',
                  'rs_bounds': True, 'smooth_factor': [10, 20], 'thres_outlier': 10}         # keys of the reference's config.json only
    assert s.ba_mode() == (_lib.SOLVER_TRF_LSMR, _lib.JAC_FD)
    s.settings['ba_solver'] = 'lm'
    assert s.ba_mode() == (_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC)
    s.settings['ba_jacobian'] = 'pattern'
    assert s.ba_mode() == (_lib.SOLVER_LM_SCHUR, _lib.JAC_PATTERN)
    s.settings.update(ba_solver='trf', ba_jacobian='analytic')
    assert s.ba_mode() == (_lib.SOLVER_TRF_LSMR, _lib.JAC_ANALYTIC)
    import pytest
    for bad in ({'ba_solver': 'gn'}, {'ba_jacobian': 'exact'}):
        s.settings.update(ba_solver='trf', ba_jacobian='fd')
        s.settings.update(bad)
        with pytest.raises(ValueError):
            s.ba_mode()
    o = _lib.default_opts()
    assert o.lm_lambda_min == 3e-3


def test_group_columns_rejects_bad_input_and_matches_scipy_on_a_random_pattern():
    """mvus_group_columns (host side of the FD mode: scipy's group_columns without a scipy.sparse matrix): scipy's groups on a
    random pattern with duplicate entries in random order; a non-permutation and an entry outside the matrix are refused."""
    import ctypes
    from scipy import sparse
    from scipy.optimize._numdiff import group_columns
    from mvus_amd import _lib
    lib = _lib.load()
    m, n = 3000, 400
    A = sparse.random(m, n, density=0.006, random_state=5, format='csr')
    A.data[:] = 1
    coo = A.tocoo()
    rng = np.random.default_rng(5)
    rows = np.concatenate([coo.row, coo.row[:50]]).astype(np.int64)
    cols = np.concatenate([coo.col, coo.col[:50]]).astype(np.int64)
    p = rng.permutation(rows.size)
    rows, cols = np.ascontiguousarray(rows[p]), np.ascontiguousarray(cols[p])
    order = np.ascontiguousarray(np.random.RandomState(0).permutation(n), dtype=np.int64)

    def run(rows, cols, order):
        g = np.full(n, -7, dtype=np.int32)
        ng = lib.mvus_group_columns(m, n, rows.size, rows.ctypes.data_as(_lib.c_int64_p), cols.ctypes.data_as(_lib.c_int64_p),
                                    order.ctypes.data_as(_lib.c_int64_p), g.ctypes.data_as(_lib.c_int32_p))
        return ng, g

    ng, g = run(rows, cols, order)
    gs = group_columns(A)
    assert ng == int(gs.max()) + 1 and np.array_equal(g, gs)
    bad = order.copy(); bad[3] = bad[4]
    assert run(rows, cols, bad)[0] == _lib.MVUS_E_INVALID and b'permutation' in lib.mvus_last_error(None)
    r2 = rows.copy(); r2[7] = m
    assert run(r2, cols, order)[0] == _lib.MVUS_E_INVALID and b'outside' in lib.mvus_last_error(None)


def test_two_level_separator_elimination_algebra():
    """The algebra behind the time shards' two-level elimination of the separators (mvus_amd/csrc/ba_schur_hip.hip.h: k_sep2_build /
    k_sep2_reduce / k_sep2_finish), in numpy on a random SPD block-tridiagonal system: every "rank" solves its LOCAL separators with the
    two coupling blocks as extra right-hand-side columns, contributes T'_G, U'_G, R'_G, T'_K, R'_K to the (world - 1)-node cut system,
    the cut system is solved once, and X_L = Y - V X_G - W X_K.  Must equal the direct solve -- including ranks without local separators
    (the cuts couple directly), the first rank (no ghost) and the last (no cut)."""
    rng = np.random.default_rng(5)
    s3, ncols = 9, 7
    for sizes in ([3, 2, 4], [2, 0, 3, 1], [1, 1], [0, 2, 0], [5]):          # local separators per rank; ranks 0 .. W-2 are followed by a cut
        W = len(sizes)
        kinds = []                                                  # chain order: rank r's local nodes, then its cut (except the last rank)
        for r, k in enumerate(sizes):
            kinds += [('L', r)] * k + ([('K', r)] if r + 1 < W else [])
        m = len(kinds)
        if m == 0:
            continue
        U = [rng.normal(size=(s3, s3)) * 0.3 for _ in range(m - 1)]          # U[q] = T(q, q + 1)
        T = [np.eye(s3) * 4 + (lambda a: a @ a.T * 0.1)(rng.normal(size=(s3, s3))) for _ in range(m)]
        R = [rng.normal(size=(s3, ncols)) for _ in range(m)]
        A = np.zeros((m * s3, m * s3))
        for q in range(m):
            A[q * s3:(q + 1) * s3, q * s3:(q + 1) * s3] = T[q]
            if q + 1 < m:
                A[q * s3:(q + 1) * s3, (q + 1) * s3:(q + 2) * s3] = U[q]
                A[(q + 1) * s3:(q + 2) * s3, q * s3:(q + 1) * s3] = U[q].T
        X_direct = np.linalg.solve(A, np.vstack(R))
        # the cut system, summed over the "ranks"
        ncut = W - 1
        T2 = [np.zeros((s3, s3)) for _ in range(ncut)]
        U2 = [np.zeros((s3, s3)) for _ in range(ncut)]
        R2 = [np.zeros((s3, ncols)) for _ in range(ncut)]
        idx = {kind: q for q, kind in enumerate(kinds) if kind[0] == 'K'}
        local = {}
        for r, k in enumerate(sizes):
            g = idx.get(('K', r - 1))                                # the ghost: the cut closing rank r - 1
            c = idx.get(('K', r))                                    # this rank's own cut
            q0 = (g + 1) if g is not None else 0
            # a cut separator's T and R are the sum of what its two neighbouring ranks contribute (here: half each, any split works)
            if k == 0:
                if g is not None:
                    T2[r - 1] += 0.5 * T[g]; R2[r - 1] += 0.5 * R[g]
                    if c is not None:
                        U2[r - 1] += U[g]                            # no local separator: the two cuts couple directly
                if c is not None:
                    T2[r] += 0.5 * T[c]; R2[r] += 0.5 * R[c]
                continue
            L = np.zeros((k * s3, k * s3))
            for j in range(k):
                L[j * s3:(j + 1) * s3, j * s3:(j + 1) * s3] = T[q0 + j]
                if j + 1 < k:
                    L[j * s3:(j + 1) * s3, (j + 1) * s3:(j + 2) * s3] = U[q0 + j]
                    L[(j + 1) * s3:(j + 2) * s3, j * s3:(j + 1) * s3] = U[q0 + j].T
            rhs = np.zeros((k * s3, ncols + 2 * s3))
            rhs[:, :ncols] = np.vstack(R[q0:q0 + k])
            if g is not None:
                rhs[:s3, ncols:ncols + s3] = U[g].T                  # C_LG: T(s_1, G) = U_G^T in s_1's rows
            if c is not None:
                rhs[-s3:, ncols + s3:] = U[q0 + k - 1]               # C_LK: T(s_k, K) = U_{s_k} in s_k's rows
            YVW = np.linalg.solve(L, rhs)
            local[r] = (q0, k, YVW)
            Y1, V1, W1 = YVW[:s3, :ncols], YVW[:s3, ncols:ncols + s3], YVW[:s3, ncols + s3:]
            Yk, Wk = YVW[-s3:, :ncols], YVW[-s3:, ncols + s3:]
            if g is not None:
                T2[r - 1] += 0.5 * T[g] - U[g] @ V1
                R2[r - 1] += 0.5 * R[g] - U[g] @ Y1
                if c is not None:
                    U2[r - 1] += -U[g] @ W1
            if c is not None:
                T2[r] += 0.5 * T[c] - U[q0 + k - 1].T @ Wk
                R2[r] += 0.5 * R[c] - U[q0 + k - 1].T @ Yk
        X = np.zeros_like(X_direct)
        if ncut:
            C = np.zeros((ncut * s3, ncut * s3))
            for q in range(ncut):
                C[q * s3:(q + 1) * s3, q * s3:(q + 1) * s3] = T2[q]
                if q + 1 < ncut:
                    C[q * s3:(q + 1) * s3, (q + 1) * s3:(q + 2) * s3] = U2[q]
                    C[(q + 1) * s3:(q + 2) * s3, q * s3:(q + 1) * s3] = U2[q].T
            Xc = np.linalg.solve(C, np.vstack(R2))
            for r in range(ncut):
                X[idx[('K', r)] * s3:(idx[('K', r)] + 1) * s3] = Xc[r * s3:(r + 1) * s3]
        for r, (q0, k, YVW) in local.items():
            XL = YVW[:, :ncols].copy()
            if ('K', r - 1) in idx:
                XL -= YVW[:, ncols:ncols + s3] @ X[idx[('K', r - 1)] * s3:(idx[('K', r - 1)] + 1) * s3]
            if ('K', r) in idx:
                XL -= YVW[:, ncols + s3:] @ X[idx[('K', r)] * s3:(idx[('K', r)] + 1) * s3]
            X[q0 * s3:(q0 + k) * s3] = XL
        np.testing.assert_allclose(X, X_direct, rtol=0, atol=1e-11 * np.abs(X_direct).max())
