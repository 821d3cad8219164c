"""The reference-side binding INTEGRATION.md documents is evidence, not prose.

CPU: the struct definitions of the stub (the code block is extracted from the markdown) are compared field by field with this
repo's own binding (mvus_amd/_lib.py) and with the library's sizeof (mvus_abi_sizes).
GPU: the stub is EXECUTED verbatim -- a fresh interpreter, a plain ctypes.CDLL, no mvus_amd import, a Scene-shaped object built
from a reference fixture -- and must return BAHandle.solve's x bit for bit (it stands where the reference calls
scipy.optimize.least_squares, reconstruction/common.py:670).
"""
import ctypes
import os
import re
import subprocess
import sys
import types

import numpy as np
import pytest

from mvus_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_source():
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', md, flags=re.S)
    stub = [b for b in blocks if '_gpu_least_squares' in b]
    assert len(stub) == 1, 'INTEGRATION.md must hold exactly one python block defining _gpu_least_squares'
    return stub[0]


def test_stub_structs_are_the_headers(monkeypatch):
    import __graft_entry__ as ge
    ge.build()
    monkeypatch.setenv('MVUS_LIB_PATH', _lib.LIB_PATH)
    ns = {}
    # importing the stub dlopens the library and runs its own load-time assertion against mvus_abi_sizes (no device is touched)
    exec(compile(stub_source(), 'INTEGRATION.md', 'exec'), ns)
    for theirs, mine in ((ns['mvus_problem'], _lib.MvusProblem), (ns['mvus_solve_opts'], _lib.MvusSolveOpts), (ns['mvus_result'], _lib.MvusResult)):
        assert [f[0] for f in theirs._fields_] == [f[0] for f in mine._fields_]
        assert ctypes.sizeof(theirs) == ctypes.sizeof(mine)
        for (name, t), (_, u) in zip(theirs._fields_, mine._fields_):
            assert ctypes.sizeof(t) == ctypes.sizeof(u) and getattr(theirs, name).offset == getattr(mine, name).offset, name
    # the byte counts the prose states
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    assert 'mvus_solve_opts (%d bytes)' % ctypes.sizeof(_lib.MvusSolveOpts) in md
    assert 'mvus_result (%d bytes)' % ctypes.sizeof(_lib.MvusResult) in md
    # the header's struct members, in order, are the binding's fields
    hdr = open(os.path.join(ROOT, 'include', 'mvus_ba.h')).read()
    for cname, mine in (('mvus_problem', _lib.MvusProblem), ('mvus_solve_opts', _lib.MvusSolveOpts), ('mvus_result', _lib.MvusResult)):
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), hdr, flags=re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        members = []
        for decl in body.split(';'):
            decl = decl.strip()
            if decl:
                members += [m.strip().lstrip('*').strip() for m in re.sub(r'^(const\s+)?[A-Za-z_0-9]+\s*\**', '', decl).split(',')]
        assert members == [f[0] for f in mine._fields_], cname


def test_a_stale_stub_fails_at_load(monkeypatch):
    """The round-5 defect (an 80-byte buffer for the 88-byte mvus_solve_opts) as a stub: the load-time assertion catches it."""
    import __graft_entry__ as ge
    ge.build()
    monkeypatch.setenv('MVUS_LIB_PATH', _lib.LIB_PATH)
    src = stub_source()
    stale = src.replace("('lm_trust_radius', ctypes.c_double)", "")
    assert stale != src
    stale = stale.replace("('lm_lambda_min', ctypes.c_double),", "('lm_lambda_min', ctypes.c_double)")
    with pytest.raises(AssertionError, match='does not match this stub'):
        exec(compile(stale, 'INTEGRATION.md (stale)', 'exec'), {})


_CHILD = r'''
import os, sys, types
import numpy as np
g = dict(np.load(sys.argv[1], allow_pickle=False))
src = open(sys.argv[2]).read()
ns = {}
exec(compile(src, 'INTEGRATION.md', 'exec'), ns)          # the stub, verbatim
assert 'mvus_amd' not in sys.modules and 'torch' not in sys.modules
C = int(g['num_cam']); off = g['det_offsets']; koff = g['knot_offsets']
scene = types.SimpleNamespace()                           # the attributes Scene.BA reads (common.py:441-697)
scene.sequence = list(range(C))
scene.detections = [g['detections'][:, off[i]:off[i + 1]].copy() for i in range(C)]
scene.cameras = [types.SimpleNamespace(K=g['cam_K'][i], d=g['cam_d'][i], resolution=[float(g['cam_res'][i][0]), float(g['cam_res'][i][1])]) for i in range(C)]
scene.spline = {'tck': [[g['knots'][koff[s]:koff[s + 1]].copy(), None, 3] for s in range(koff.size - 1)], 'int': g['interval']}
scene.settings = {'opt_calib': bool(g['opt_calib']), 'undist_points': bool(g['undist_points']), 'motion_type': str(g['motion_type'])}
x = ns['_gpu_least_squares'](scene, C, g['x0'], 10, bool(g['rolling_shutter']), bool(g['motion_reg']), float(g['motion_weights']), bool(g['rs_bounds']))
np.save(sys.argv[3], x)
'''


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['c1_pinhole_2cam', 'rs_F_2int_3cam'])
def test_stub_executed_verbatim_returns_the_bindings_x(name, tmp_path):
    from golden_util import GOLDEN_DIR, load_case
    from mvus_amd import problem as mp
    from mvus_amd.ba import BAHandle
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    x0 = g['x0']                                              # the reference's own start vector (what the child hands the stub)
    with BAHandle(prob, device=0) as h:
        want = h.solve(x0, max_nfev=10, ties='canonical').x   # mvus_default_opts: TRF + LSMR, pattern fixed on the GPU inside the call
    stub = tmp_path / 'stub.py'
    stub.write_text(stub_source())
    child = tmp_path / 'child.py'
    child.write_text(_CHILD)
    out = tmp_path / 'x.npy'
    env = dict(os.environ, MVUS_LIB_PATH=_lib.LIB_PATH)
    env.pop('PYTHONPATH', None)
    r = subprocess.run([sys.executable, str(child), os.path.join(GOLDEN_DIR, name + '.npz'), str(stub), str(out)],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.load(out)
    assert got.shape == want.shape and np.array_equal(got, want), 'stub x differs from BAHandle.solve: max |dx| = %g' % np.max(np.abs(got - want))
    assert not np.array_equal(got, x0)
