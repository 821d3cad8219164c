"""The generator calls behind the golden cases (tests/golden/<case>.npz): the seeded synthetic scene each fixture was made
from.  tests/golden/make_golden.py feeds these to the real reference; tests that need the generator's ground truth (which
the fixtures do not store) rebuild the scene from here -- mvus_amd.synth is deterministic in its arguments."""
import numpy as np

from mvus_amd import synth


def integer_intervals(sc):
    """Snap interval bounds (and the clamped end knots) to integers so that motion samples can land
    exactly on an interval end -- the closed/half-open corner of common.py:292 vs util.py:105."""
    for s in range(sc.interval.shape[1]):
        a, b = np.ceil(sc.interval[0, s]), np.floor(sc.interval[1, s])
        t = sc.tck[s][0]
        t[:4], t[-4:] = a, b
        inner = t[4:-4]
        t[4:-4] = np.clip(inner, a + 0.5, b - 0.5)
        sc.interval[0, s], sc.interval[1, s] = a, b
    return sc


GENERATORS = {
    # BASELINE config 1: 2 pinhole cams x 1k detections, global shutter, no motion reg
    'c1_pinhole_2cam': lambda: synth.make_scene(2, 2000, seed=1, knot_spacing=15.0),
    # rolling shutter + motion_reg F over two integer-aligned intervals
    'rs_F_2int_3cam': lambda: integer_intervals(synth.make_scene(3, 1500, seed=11, rolling_shutter=True, motion_reg=True,
                                                                 motion_type='F', motion_weights=1e4, num_intervals=2,
                                                                 knot_spacing=12.0, dropout=0.05)),
    # full parameter vector: K, dist, beta, RS (bounded), pose, spline; KE regulariser.  Cameras 60 m from a target that
    # stays within ~170 px of the image centre: the distortion coefficients are next to unobservable and the reference
    # itself wanders on it (its first BA takes k1 from -0.03 to 27.7) -- kept as the ILL-POSED calibration case
    'calib_KE_bounds_3cam': lambda: synth.make_scene(3, 1200, seed=21, rolling_shutter=True, distortion=True,
                                                     opt_calib=True, rs_bounds=True, motion_reg=True, motion_type='KE',
                                                     motion_weights=1e2, knot_spacing=14.0),
    # fixed calibration with lens distortion (observation-side undistortion only)
    'dist_fixed_2cam': lambda: synth.make_scene(2, 800, seed=31, rolling_shutter=True, distortion=True, knot_spacing=16.0),
    # BASELINE configs[4] made WELL POSED: five cameras 18 m from the target (it sweeps most of every image), ~3k detections
    # per camera, calibration started 0.3 sigma off, 0.5 % gross outliers: the reference keeps K, d physical and reproduces
    # its own 200-evaluation answer to ~1e-4 px
    'calib_KE_wellposed_5cam': lambda: synth.make_scene(5, 15000, seed=41, rolling_shutter=True, distortion=True,
                                                        opt_calib=True, rs_bounds=True, motion_reg=True, motion_type='KE',
                                                        motion_weights=1e2, knot_spacing=14.0, perturb=0.3, ring_radius=18.0,
                                                        outlier_frac=0.005),
    # BASELINE configs[1] in SHAPE (7 cameras, rolling shutter, motion_reg F with the README's weight 1e4, two intervals) at 1/14 of
    # its size, so that the real reference converges in minutes: the same generator arguments as synth.BASELINE_CONFIGS[1]
    'config1_shape_7cam': lambda: integer_intervals(synth.make_scene(**dict(synth.BASELINE_CONFIGS[1], total_obs=7000))),
}
MAX_ITERS = {'c1_pinhole_2cam': (10, 40)}          # first-BA budgets stored per case (default (10,))


def make(name):
    return GENERATORS[name]()
