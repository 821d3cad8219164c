"""Stress gate for the LM / Schur solve chain on fresh handles ("one incident": on one box 10 of 60
fresh-handle LM solves raised the pivot flag at every damping value; cause never found).  A few hundred fresh handles per
configuration, with the device memory they will be handed pre-filled with NaN / 1e300 every few handles (a read of
uninitialised memory then shows), every solve compared with the first one.  Runs on the driver's box every round."""
import collections

import numpy as np
import pytest

from mvus_amd import _lib
from mvus_amd import problem as mp
from mvus_amd import synth

pytestmark = pytest.mark.gpu


def _poison(value, gib=2):
    """Fill ``gib`` GiB of device memory with ``value`` and hand it back to HIP: the next hipMalloc blocks hold it."""
    import torch
    junk = [torch.full((1 << 27,), value, dtype=torch.float64, device='cuda') for _ in range(gib)]
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()


@pytest.mark.parametrize('index,handles', [(0, 300), (1, 300), (4, 60)])
def test_fresh_handle_lm_solves_never_fail(index, handles):
    """configs[0]- (2 cams x 2k), configs[1]- (7 cams x 100k, 63-unknown reduced system: the incident's) and configs[4]-size
    (opt_calib, 126 unknowns): every fresh handle must take the same LM path as the first -- same evaluation counts and
    status, cost equal to 1e-9 relative (the fused assembly's sums are order-dependent in the last bits only), never a
    pivot failure (which shows as a cost that does not decrease / status 0 with lambda blown up)."""
    from mvus_amd.ba import BAHandle
    sc = synth.baseline_scene(index)
    prob, x0 = mp.problem_from_scene(sc)
    first = None
    outcomes = collections.Counter()
    for rep in range(handles):
        if rep % 10 == 0:
            _poison(float('nan') if (rep // 10) % 2 == 0 else 1e300)
        with BAHandle(prob) as h:
            r = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=8, return_fun=False)
        key = (r.nfev, r.njev, r.status)
        outcomes[key] += 1
        if first is None:
            first = r
            assert np.isfinite(r.cost) and r.cost < r.initial_cost, (r.cost, r.initial_cost)      # (2 % gross outliers carry most of the cost)
            continue
        assert key == (first.nfev, first.njev, first.status), 'handle %d took another path: %r' % (rep, dict(outcomes))
        # the fused assembly adds with fp64 atomics: the last bits of the normal equations depend on the arrival order and
        # eight evaluations amplify them; a pivot failure or a read of poisoned memory changes the path or the cost by far more
        assert abs(r.cost - first.cost) <= 1e-7 * first.cost, 'handle %d: cost %.15g vs %.15g' % (rep, r.cost, first.cost)
        assert np.max(np.abs(r.x - first.x)) <= 1e-4 * max(1.0, float(np.max(np.abs(first.x))))


def test_one_handle_many_solves_all_descend():
    """The same handle solving from the same start 200 times (pool buffers recycled between solves, camera-state cache kept
    honest by release(); the damping is carried from one solve to the next, so the paths differ): every solve descends."""
    from mvus_amd.ba import BAHandle
    sc = synth.baseline_scene(1)
    prob, x0 = mp.problem_from_scene(sc)
    with BAHandle(prob) as h:
        costs = []
        for rep in range(200):
            r = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=6, return_fun=False)
            assert np.isfinite(r.cost) and r.cost < r.initial_cost, 'solve %d: %.9g -> %.9g (status %d)' % (rep, r.initial_cost, r.cost, r.status)
            costs.append(r.cost)
        assert max(costs) < (1 + 1e-2) * min(costs)
