"""LM + Schur with the deterministic assembly (mvus_ba_set_deterministic / MVUS_DET_ASSEMBLY=1): every assembly workgroup leaves its
sums in a window of its own instead of adding with fp64 atomics, k_det_gather adds the windows per control point in a fixed order,
the motion rows are added per control point in row order.  Checked here: the normal equations equal the atomic assembly's to
rounding, and the assembly as well as a whole LM solve give the SAME BITS on every run -- on a pinhole scene, with rolling shutter +
motion regulariser F (two spline intervals), with opt_calib (P = 15) + KE, and on a dense flight (several windows of one camera
per control point).  Tracks so sparse that 128 consecutive detections of a camera span more than 64 control points keep the atomic
path for those workgroups (correct, not reproducible): not the regime of the BASELINE configurations."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scenes():
    from mvus_amd import synth
    yield 'pinhole_8cam', dict(num_cam=8, total_obs=60_000, seed=5)
    kw = dict(synth.BASELINE_CONFIGS[1]); kw.update(total_obs=40_000)
    yield 'rs_motion_F_dense_7cam', kw
    kw = dict(synth.BASELINE_CONFIGS[4]); kw.update(total_obs=30_000)
    yield 'calib_KE_7cam', kw
    # 300 detections per knot span and camera: a control point sits in ~10 windows of ONE camera -- the gather's second pass
    yield 'very_dense_2cam', dict(num_cam=2, total_obs=120_000, seed=7, num_knots=200, rolling_shutter=True)
    yield 'config2_32cam_504k', dict(synth.BASELINE_CONFIGS[2])       # full size: a thinner flight spreads a workgroup's 128 detections over
                                                                        # more than the 64 control points of a window and falls back to atomics


@pytest.mark.parametrize('name,kw', list(_scenes()), ids=[n for n, _ in _scenes()])
def test_deterministic_assembly_matches_and_repeats(name, kw):
    from mvus_amd import ba, problem as mp, synth
    scene = synth.make_scene(**kw)
    prob, x0 = mp.problem_from_scene(scene)

    def run(det):
        with ba.BAHandle(prob) as h:
            h.set_deterministic(det)
            h.residual_jacobian(x0)
            ne = h.normal_equations()
            res = h.solve(x0, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=6)
            return ne, res.cost, res.x.copy()

    ne_a, cost_a, _ = run(False)
    runs = [run(True) for _ in range(3)]
    for part_a, part_d, what in zip(ne_a, runs[0][0], ('gradient', 'camera blocks', 'band', 'cross block')):
        scale = np.max(np.abs(part_a))
        assert np.max(np.abs(part_d - part_a)) <= 1e-12 * scale, what          # measured 1e-20 ... 1e-15: summation order only
    for ne_d, cost_d, x_d in runs[1:]:
        for p0, p1 in zip(runs[0][0], ne_d):
            assert np.array_equal(p0, p1)                                       # the same bits, run to run
        assert cost_d == runs[0][1] and np.array_equal(x_d, runs[0][2])
    assert abs(runs[0][1] - cost_a) <= 1e-9 * cost_a                            # and the same optimisation as the atomic mode


def test_forty_fresh_handles_one_outcome():
    """tools/micro/loop_lm_configs.py as a test: 40 fresh handles, eight-evaluation LM solves on a rolling-shutter scene with motion
    regulariser -- ONE (cost, x) with the deterministic assembly (with atomics: as many outcomes as handles)."""
    import hashlib
    from mvus_amd import ba, problem as mp, synth
    kw = dict(synth.BASELINE_CONFIGS[1]); kw.update(total_obs=50_000)
    prob, x0 = mp.problem_from_scene(synth.make_scene(**kw))
    seen = set()
    for _ in range(40):
        with ba.BAHandle(prob) as h:
            h.set_deterministic(True)
            r = h.solve(x0, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=8)
            assert not h.deterministic_fallback()
            seen.add((repr(r.cost), hashlib.sha1(np.ascontiguousarray(r.x).tobytes()).hexdigest()))
    assert len(seen) == 1


def test_sparse_tracks_fall_back_and_say_so():
    """configs[2] thinned to 0.75 detections per knot span and camera: 128 consecutive detections of a camera reach over ~170 control
    points, more than a window holds -- those workgroups keep the atomic path, the handle reports it, the normal equations are the
    atomic mode's to rounding; the full-size configuration does not fall back."""
    from mvus_amd import ba, problem as mp, synth
    for obs, expect in ((120_000, True), (504_399, False)):
        kw = dict(synth.BASELINE_CONFIGS[2]); kw.update(total_obs=obs)
        prob, x0 = mp.problem_from_scene(synth.make_scene(**kw))
        with ba.BAHandle(prob) as h:
            assert h.deterministic_fallback() is False
            h.residual_jacobian(x0)
            ne_a = h.normal_equations()
            h.set_deterministic(True)
            ne_d = h.normal_equations()
            assert h.deterministic_fallback() is expect
            for a, d in zip(ne_a, ne_d):
                assert np.max(np.abs(a - d)) <= 1e-12 * np.max(np.abs(a))


def test_scene_setting_switches_the_mode():
    """settings['ba_deterministic'] with ba_solver 'lm': two Scene.BA calls from the same state end with identical parameters."""
    from mvus_amd import pipeline, synth
    kw = dict(synth.BASELINE_CONFIGS[1])
    for k in ('seed', 'num_cam', 'total_obs', 'num_intervals'):
        kw.pop(k, None)
    kw['motion_weights'] = 1e2
    ends = []
    for _ in range(2):
        flight, _sc = pipeline.staged_scene(4, 30_000, seed=4, settings={'ba_solver': 'lm', 'ba_deterministic': True}, perturb=0.3, **kw)
        flight.BA(2, max_iter=6, rs=True, motion_reg=True, motion_weights=1e2)
        ends.append(flight)
    for i in flights_cams(ends[0]):
        a, b = ends[0].cameras[i], ends[1].cameras[i]
        assert np.array_equal(a.R, b.R) and np.array_equal(a.t, b.t)
    assert all(np.array_equal(a[1], b[1]) for a, b in zip(ends[0].spline['tck'], ends[1].spline['tck']))
    assert np.array_equal(ends[0].alpha, ends[1].alpha) and np.array_equal(ends[0].beta, ends[1].beta)


def flights_cams(flight):
    return list(flight.sequence[:2])
