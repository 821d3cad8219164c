"""Time of the fused assembly (mvus_ba_time_kernel 6: window-major kernel + camera-block sum + motion rows) per configuration, for the
library in MVUS_LIB_PATH and the window length in MVUS_WIN; usage: time_win.py [configs...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvus_amd import ba, problem as mp, synth
_cache = {}
for cfg in [int(a) for a in sys.argv[1:]] or [2, 1, 3]:
    if cfg not in _cache: _cache[cfg] = mp.problem_from_scene(synth.make_scene(**dict(synth.BASELINE_CONFIGS[cfg])))
    prob, x0 = _cache[cfg]
    for win in (os.environ.get('MVUS_WIN_LIST') or os.environ.get('MVUS_WIN') or '0').split(','):
        if win != '0': os.environ['MVUS_WIN'] = win
        else: os.environ.pop('MVUS_WIN', None)
        with ba.BAHandle(prob) as h:
            h.residual_jacobian(x0)
            g = h.normal_equations()[0]
            ts = [1e3 * h.time_kernel(6, 30) for _ in range(3)]
        print('config %d  lib %s  MVUS_WIN=%s: fused assembly %s us  (|g| %.6e)' % (cfg, os.path.basename(os.environ.get('MVUS_LIB_PATH', 'default')), win, ' '.join('%.1f' % t for t in ts), np.linalg.norm(g)), flush=True)
