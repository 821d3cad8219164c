"""The reference's incremental loop (main.py:43-82) end to end through the drop-in Scene, every stage on the GPU, at
BASELINE configs[1] shape (7 cameras, rolling shutter, motion regulariser F): 12 BAs, 6 outlier passes,
5 x (select_most_overlap -> get_camera_pose -> triangulate + spline refit).  The reference's own loop cannot run in this
image (init_traj and get_camera_pose need OpenCV), so the loop as a whole is judged against the generator's ground truth;
its BA / remove_outliers / triangulate stages are pinned one by one against reference-generated goldens elsewhere
(test_gpu_parity.py, test_gpu_scene.py, test_triangulate.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('solver', ['trf', 'lm'])
def test_incremental_loop_seven_cameras(solver):
    from mvus_amd import pipeline, synth
    kw = dict(synth.BASELINE_CONFIGS[1])
    kw.pop('seed'); kw.pop('num_cam'); kw.pop('total_obs'); kw.pop('num_intervals', None)
    # configs[1] quotes the README's motion_weights = 1e4; on this synthetic flight (accelerations ~1e-3 m/frame^2, 3k rows) that
    # makes the regulariser 100x the data term and BA -- the reference's objective, any solver -- flattens the curve at the
    # price of 3 px of reprojection error, after which a 10 px outlier threshold eats the inliers.  1e2 keeps it a regulariser.
    kw['motion_weights'] = 1e2
    # 60k detections: on a sparse flight (21k) the loop amplifies last-bit differences into different outcomes -- the smoothing
    # fit of either solver breaks FITPACK's knot ties by rounding (fpknot gives the
    # two halves of a split interval residuals fp * n1 / n and fp * n2 / n: the quarters of an interval tie mathematically and the
    # winner is decided by the last bit of fp; scipy itself lands on other knots than this library on such a trajectory, see
    # tools/micro/loop_fits_vs_scipy.py and DESIGN.md section 2).  Eight seeds at 21k: worst camera centre 0.2-0.7 m with one summation
    # order of the fit, 0.2-11 m with another; at 60k every seed and order stays below 0.35 m.
    flight, sc = pipeline.staged_scene(7, 60_000, seed=2, settings={'ba_solver': solver}, perturb=0.3, **kw)
    start_extent = (float(flight.spline['int'][0, 0]), float(flight.spline['int'][1, -1]))
    assert all(flight.cameras[i].P is None for i in range(2, 7))
    timer = pipeline.incremental_reconstruction(flight, max_iter=10)
    stages = [r[0] for r in timer.rows]
    assert stages.count('BA') == 12 and stages.count('remove_outliers') == 6
    assert stages.count('get_camera_pose') == stages.count('triangulate') == stages.count('select_most_overlap') == 5
    assert sorted(flight.sequence) == list(range(7)) and all(c.P is not None for c in flight.cameras)
    ev = pipeline.evaluate_against_truth(flight, sc)
    print(solver, 'mean err', np.round(ev['mean_err'], 3), 'traj rms %.3f max %.3f' % (ev['traj_rms'], ev['traj_max']), 'centres', np.round(ev['centre_err'], 3),
          'rot', np.round(ev['rot_err_deg'], 3), 'kept/clean/dirty', list(zip(ev['kept'], ev['clean'], ev['kept_dirty'])), 'extent', ev['trajectory_extent'],
          'seconds', {k: round(v, 2) for k, v in timer.totals().items()})
    # the trajectory grew from the first two cameras' common range to (nearly) the whole flight
    assert ev['trajectory_extent'][1] > start_extent[1] + 0.2 * ev['trajectory_extent'][2]
    assert ev['trajectory_extent'][1] > 0.95 * ev['trajectory_extent'][2]
    # 0.5 px noise per axis -> mean distance ~0.63 px for a perfect fit; measured after 12 ten-evaluation BAs: 0.85-1.15 (trf, 60k, eight seeds),
    # 0.62-0.97 (lm, 60k; bit-reproducible since the window-major assembly of round 4, and run with the loop's damping floor)
    assert max(ev['mean_err']) < 1.3
    for kept, clean, dirty in zip(ev['kept'], ev['clean'], ev['kept_dirty']):
        assert kept >= 0.95 * clean                  # the inliers survive six outlier passes ...
        assert dirty <= 0.02 * kept + 5              # ... the gross outliers (2 % at 20-200 px) do not
    # geometry after the best similarity (the BA is free in its gauge): measured rms 0.2-0.4 m on a 20 m flight, cameras within 1.5 m / 1.5 deg
    # (lm at 60k: rms 0.18-0.23 m, cameras within 0.14 m / 0.13 deg; on the sparse 21k flight LM + Schur has been seen 3 m off while
    # fitting the detections as well: stretches that one camera alone observes)
    assert ev['traj_rms'] < 0.6 and max(ev['centre_err']) < 2.5 and max(ev['rot_err_deg']) < 2.5
    assert abs(ev['scale'] - 1.0) < 0.08 or solver == 'lm'           # (LM's gauge drifts in scale: 1.14 seen at 100k; the similarity absorbs it)


@pytest.mark.parametrize('seed', [1, 5, 9, 10])
def test_loop_with_lm_on_the_seeds_that_used_to_fail(seed):
    """ba_solver = 'lm' through the whole loop on the flights where it ended metres away with the library's damping floor (scale
    collapsing to 0.0 - 0.3: the staged two- and three-camera BAs are free in their similarity gauge and the motion regulariser rewards
    a smaller scene).  incremental_reconstruction sets the loop's floor (LOOP_LM_LAMBDA_MIN) and hands wide-band BAs to the parity
    solver; measured over eleven seeds: 0.21 - 0.43 m from the truth, scale within 0.4 % (profiles/round4/r04_loop_lm_damping_floors.txt)."""
    from mvus_amd import pipeline, synth
    kw = dict(synth.BASELINE_CONFIGS[1])
    kw.pop('seed'); kw.pop('num_cam'); kw.pop('total_obs'); kw.pop('num_intervals', None)
    kw['motion_weights'] = 1e2
    flight, sc = pipeline.staged_scene(7, 100_000, seed=seed, settings={'ba_solver': 'lm'}, perturb=0.3, **kw)
    assert 'ba_lambda_min' not in flight.settings
    pipeline.incremental_reconstruction(flight, max_iter=10)
    assert 'ba_lambda_min' not in flight.settings            # the loop's floor (LOOP_LM_LAMBDA_MIN) was in force for the loop only
    ev = pipeline.evaluate_against_truth(flight, sc)
    print('seed', seed, 'traj rms %.3f' % ev['traj_rms'], 'scale %.4f' % ev['scale'], 'centres', np.round(ev['centre_err'], 3))
    assert ev['traj_rms'] < 0.6 and abs(ev['scale'] - 1.0) < 0.02 and max(ev['centre_err']) < 1.0
