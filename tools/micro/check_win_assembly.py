"""Window-major fused assembly (default) against the normal equations formed from the stored Jacobian blocks by the detection-major
kernel (MVUS_NE_FROM_J=1): max difference per part, run-to-run bits, and the time of both (mvus_ba_time_kernel 6 / 4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvus_amd import ba, problem as mp, synth

def scenes():
    yield 'pinhole_8cam_60k', dict(num_cam=8, total_obs=60_000, seed=5)
    kw = dict(synth.BASELINE_CONFIGS[1]); kw.update(total_obs=40_000); yield 'rs_F_7cam_40k', kw
    kw = dict(synth.BASELINE_CONFIGS[4]); kw.update(total_obs=30_000); yield 'calib_KE_7cam_30k', kw
    yield 'very_dense_2cam', dict(num_cam=2, total_obs=120_000, seed=7, num_knots=200, rolling_shutter=True)
    kw = dict(synth.BASELINE_CONFIGS[2]); kw.update(total_obs=120_000); yield 'config2_sparse_120k', kw
    yield 'config0', dict(synth.BASELINE_CONFIGS[0])
    yield 'config1', dict(synth.BASELINE_CONFIGS[1])
    yield 'config2', dict(synth.BASELINE_CONFIGS[2])
    yield 'config4', dict(synth.BASELINE_CONFIGS[4])

only = sys.argv[1:]
for name, kw in scenes():
    if only and name not in only: continue
    prob, x0 = mp.problem_from_scene(synth.make_scene(**kw))
    rng = np.random.default_rng(1)
    x = x0 + 1e-3 * rng.standard_normal(x0.size) * np.maximum(1.0, np.abs(x0)) * 0.01
    outs = {}
    for mode in ('win', 'win2', 'fromJ'):
        if mode == 'fromJ': os.environ['MVUS_NE_FROM_J'] = '1'
        else: os.environ.pop('MVUS_NE_FROM_J', None)
        with ba.BAHandle(prob) as h:
            h.residual_jacobian(x)
            outs[mode] = h.normal_equations()
            if mode == 'win':
                t6 = h.time_kernel(6, 20) if hasattr(h, 'time_kernel') else float('nan')
                t4 = h.time_kernel(4, 20) if hasattr(h, 'time_kernel') else float('nan')
    line = []
    for i, part in enumerate(('g', 'cam', 'band', 'cross')):
        ref = outs['fromJ'][i]; scale = np.max(np.abs(ref)) + 1e-300
        line.append('%s %.1e%s' % (part, np.max(np.abs(outs['win'][i] - ref)) / scale, '' if np.array_equal(outs['win'][i], outs['win2'][i]) else ' NOT-REPEATABLE'))
    print('%-22s n=%d M=%d: %s | fused window %.1f us, from J (atomics) %.1f us' % (name, x0.size, prob.M if hasattr(prob, 'M') else -1, '; '.join(line), 1e3 * t6, 1e3 * t4), flush=True)
