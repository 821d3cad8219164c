"""The C-ABI library builds, loads, and exports every symbol include/mvus_ba.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from mvus_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as ge
    ge.build()
    return _lib.load()


def test_header_and_binding_agree(lib):
    hdr = open(os.path.join(ROOT, 'include', 'mvus_ba.h')).read()
    declared = set(re.findall(r'\b(mvus_[a-z_0-9]+)\s*\(', hdr)) - {'mvus_allreduce_fn'}
    bound = {name for name, _, _ in _lib.API}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name)


def test_struct_sizes_match_header(lib):
    # natural alignment of the C structs (see include/mvus_ba.h)
    assert ctypes.sizeof(_lib.MvusSolveOpts) == 88
    assert ctypes.sizeof(_lib.MvusResult) == 48
    assert ctypes.sizeof(_lib.MvusProblem) == 144


def test_default_opts_are_scipy_defaults(lib):
    o = _lib.MvusSolveOpts()
    lib.mvus_default_opts(ctypes.byref(o))
    assert (o.max_nfev, o.ftol, o.xtol, o.gtol) == (10, 1e-8, 1e-12, 1e-8)
    assert (o.lsmr_atol, o.lsmr_btol, o.lsmr_conlim) == (1e-6, 1e-6, 1e8)


def test_invalid_problem_is_rejected_before_touching_the_gpu(lib):
    from golden_util import load_case
    from mvus_amd import problem as mp
    scene, _ = load_case('c1_pinhole_2cam')
    prob, _ = mp.problem_from_scene(scene)
    prob.interval = prob.interval[::-1].copy()
    s, keep = _lib.make_problem_struct(prob)
    h = ctypes.c_void_p()
    rc = lib.mvus_ba_create(ctypes.byref(s), ctypes.byref(h))
    assert rc == _lib.MVUS_E_INVALID and not h
    assert b'interval' in lib.mvus_last_error(None)


def test_no_gpu_means_error_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from golden_util import load_case
    from mvus_amd import problem as mp
    scene, _ = load_case('c1_pinhole_2cam')
    prob, _ = mp.problem_from_scene(scene)
    s, keep = _lib.make_problem_struct(prob)
    h = ctypes.c_void_p()
    rc = lib.mvus_ba_create(ctypes.byref(s), ctypes.byref(h))
    assert rc == _lib.MVUS_E_HIP and not h
