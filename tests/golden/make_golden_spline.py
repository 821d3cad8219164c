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


if __name__ == '__main__':
    main()
