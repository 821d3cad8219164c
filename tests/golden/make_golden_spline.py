#!/usr/bin/env python3
"""Golden vector for the spline maintenance either side of BA: runs the REAL reference
``Scene.traj_to_spline`` / ``spline_to_traj`` (imported from /root/reference) on a seeded noisy trajectory with a gap
(two intervals) and stores inputs + outputs in tests/golden/traj_spline.npz.

    python tests/golden/make_golden_spline.py        (build container only)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                      # noqa: E402


def main():
    common = mg.import_reference()
    rng = np.random.default_rng(5)
    t = np.concatenate((np.arange(10.0, 410.0), np.arange(450.0, 800.0)))      # a gap of 40 frames: two intervals
    X = np.vstack((10 * np.sin(t / 80), 10 * np.cos(t / 95), 30 + 3 * np.sin(t / 50))) + rng.normal(scale=0.02, size=(3, t.size))
    traj = np.vstack((t, X))
    ref = common.Scene()
    ref.traj = traj.copy()
    smooth = [10, 20]
    spline = ref.traj_to_spline(smooth_factor=smooth)
    out = dict(traj=traj, smooth_factor=np.array(smooth, dtype=np.float64), interval=np.asarray(spline['int'], dtype=np.float64),
               n_int=np.int64(len(spline['tck'])))
    for i, tck in enumerate(spline['tck']):
        out['knots_%d' % i] = np.asarray(tck[0])
        out['coefs_%d' % i] = np.asarray(tck[1])
    out['traj_rate1'] = ref.spline_to_traj(sampling_rate=1).copy()
    ts = np.sort(rng.uniform(0.0, 820.0, size=300))
    out['t_query'] = ts
    out['traj_query'] = ref.spline_to_traj(t=ts).copy()
    np.savez_compressed(os.path.join(HERE, 'traj_spline.npz'), **out)
    print('intervals', spline['int'], 'knots', [len(t[0]) for t in spline['tck']], out['traj_rate1'].shape, out['traj_query'].shape)
    short_parts(common)


def short_parts(common):
    """Parts with fewer than four samples (reachable through find_intervals only with three: two gaps < 5 spanning >= 5):
    splprep(k=3) raises there and the reference's bare `except` fits a k=1 spline instead (common.py:266-267) -- a bent
    part (interpolating polyline), a straight one (the least-squares line is accepted) and a regular part in between."""
    rng = np.random.default_rng(6)
    t_long = np.arange(20.0, 120.0)
    X_long = np.vstack((10 * np.sin(t_long / 80), 10 * np.cos(t_long / 95), 30 + 3 * np.sin(t_long / 50))) + rng.normal(scale=0.02, size=(3, t_long.size))
    t_bent = np.array([0.0, 3.0, 7.5])
    X_bent = np.array([[1.0, 2.0, 2.5], [0.0, 1.5, 0.5], [30.0, 30.2, 31.0]])
    t_line = np.array([140.0, 143.5, 146.0])
    X_line = np.array([5.0, -2.0, 28.0])[:, None] + np.array([0.3, 0.1, -0.2])[:, None] * (t_line - 140.0) + 1e-6 * rng.normal(size=(3, 3))
    traj = np.hstack((np.vstack((t_bent, X_bent)), np.vstack((t_long, X_long)), np.vstack((t_line, X_line))))
    ref = common.Scene()
    ref.traj = traj.copy()
    spline = ref.traj_to_spline(smooth_factor=[10, 20])
    out = dict(traj=traj, interval=np.asarray(spline['int'], dtype=np.float64), n_int=np.int64(len(spline['tck'])),
               degree=np.array([tck[2] for tck in spline['tck']], dtype=np.int64))
    for i, tck in enumerate(spline['tck']):
        out['knots_%d' % i] = np.asarray(tck[0])
        out['coefs_%d' % i] = np.asarray(tck[1])
    out['traj_rate'] = ref.spline_to_traj(sampling_rate=0.25).copy()
    ts = np.sort(rng.uniform(-2.0, 150.0, size=200))
    out['t_query'] = ts
    out['traj_query'] = ref.spline_to_traj(t=ts).copy()
    np.savez_compressed(os.path.join(HERE, 'traj_spline_short.npz'), **out)
    print('short parts: intervals', spline['int'], 'degrees', out['degree'], 'knots', [len(t[0]) for t in spline['tck']])


if __name__ == '__main__':
    main()
