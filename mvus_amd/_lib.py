"""ctypes binding of include/mvus_ba.h (libmvusba.so).

The library is the only compute path of the package: if it is missing or cannot be loaded the
import of this module's :func:`load` raises -- there is no CPU fallback.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MVUS_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libmvusba.so')   # override: kernel experiments

MVUS_OK = 0
MVUS_E_INVALID, MVUS_E_HIP, MVUS_E_NUMERIC, MVUS_E_COMM, MVUS_E_UNSUPPORTED, MVUS_E_RESHARD = -1, -2, -3, -4, -5, -6
JAC_ANALYTIC, JAC_PATTERN, JAC_FD = 0, 1, 2
PAT_SHIFT, PAT_TIE = 25, 1 << 30          # pattern codes of mvus_ba_set_pattern (include/mvus_ba.h)
SOLVER_TRF_LSMR, SOLVER_LM_SCHUR = 0, 1

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)


class MvusProblem(ctypes.Structure):
    _fields_ = [
        ('num_cam', ctypes.c_int32), ('opt_calib', ctypes.c_int32), ('undist_points', ctypes.c_int32),
        ('rs_free', ctypes.c_int32), ('rs_bounds', ctypes.c_int32), ('motion_reg', ctypes.c_int32),
        ('motion_type', ctypes.c_int32), ('opt_sync', ctypes.c_int32), ('motion_weight', ctypes.c_double),
        ('det_offsets', c_int64_p), ('frame', c_double_p), ('u_raw', c_double_p), ('v_raw', c_double_p),
        ('img_height', c_double_p), ('K', c_double_p), ('dist', c_double_p),
        ('num_splines', ctypes.c_int32), ('interval', c_double_p), ('knot_offsets', c_int64_p),
        ('knots', c_double_p), ('device', ctypes.c_int32), ('stream', ctypes.c_void_p),
    ]


class MvusSolveOpts(ctypes.Structure):
    _fields_ = [
        ('solver', ctypes.c_int32), ('jac_mode', ctypes.c_int32), ('max_nfev', ctypes.c_int32),
        ('ftol', ctypes.c_double), ('xtol', ctypes.c_double), ('gtol', ctypes.c_double),
        ('lsmr_atol', ctypes.c_double), ('lsmr_btol', ctypes.c_double), ('lsmr_conlim', ctypes.c_double),
        ('lsmr_maxiter', ctypes.c_int32), ('verbose', ctypes.c_int32), ('lm_lambda_min', ctypes.c_double),
        ('lm_trust_radius', ctypes.c_double),
    ]


class MvusResult(ctypes.Structure):
    _fields_ = [
        ('cost', ctypes.c_double), ('optimality', ctypes.c_double), ('nfev', ctypes.c_int32),
        ('njev', ctypes.c_int32), ('status', ctypes.c_int32), ('lin_iters', ctypes.c_int32),
        ('solve_ms', ctypes.c_double), ('initial_cost', ctypes.c_double),
    ]


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)

# every symbol include/mvus_ba.h declares: (name, restype, argtypes)
API = [
    ('mvus_default_opts', None, [ctypes.POINTER(MvusSolveOpts)]),
    ('mvus_abi_sizes', ctypes.c_int32, [c_int32_p, c_int32_p, c_int32_p]),
    ('mvus_ba_create', ctypes.c_int, [ctypes.POINTER(MvusProblem), ctypes.POINTER(ctypes.c_void_p)]),
    ('mvus_ba_destroy', None, [ctypes.c_void_p]),
    ('mvus_last_error', ctypes.c_char_p, [ctypes.c_void_p]),
    ('mvus_ba_num_params', ctypes.c_int64, [ctypes.c_void_p]),
    ('mvus_ba_num_residuals', ctypes.c_int64, [ctypes.c_void_p]),
    ('mvus_ba_num_motion_rows', ctypes.c_int64, [ctypes.c_void_p]),
    ('mvus_ba_num_slots', ctypes.c_int32, [ctypes.c_void_p]),
    ('mvus_ba_residual', ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    ('mvus_ba_residual_jacobian', ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int32, c_double_p, c_double_p, c_int32_p]),
    ('mvus_ba_motion_rows', ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int32, c_double_p, c_double_p, c_int32_p]),
    ('mvus_ba_set_pattern', ctypes.c_int, [ctypes.c_void_p, c_double_p, c_int32_p]),
    ('mvus_ba_motion_pattern', ctypes.c_int, [ctypes.c_void_p, c_int32_p]),
    ('mvus_ba_upload_pattern', ctypes.c_int, [ctypes.c_void_p, c_int32_p, c_int32_p]),
    ('mvus_ba_set_deterministic', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    ('mvus_ba_deterministic_fallback', ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    ('mvus_ba_set_fd_groups', ctypes.c_int, [ctypes.c_void_p, c_int32_p, ctypes.c_int32]),
    ('mvus_group_columns', ctypes.c_int32, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_int64_p, c_int64_p, c_int64_p, c_int32_p]),
    ('mvus_fd_groups', ctypes.c_int32, [ctypes.POINTER(MvusProblem), c_int32_p, c_int32_p, c_int64_p, c_int32_p]),
    ('mvus_ba_jv', ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    ('mvus_ba_jtu', ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    ('mvus_ba_normal_equations', ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_int32_p]),
    ('mvus_ba_lm_step', ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, c_double_p]),
    ('mvus_ba_solve', ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.POINTER(MvusSolveOpts), ctypes.POINTER(MvusResult), c_double_p]),
    ('mvus_ba_outlier_mask', ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_double, c_uint8_p]),
    ('mvus_ba_remove_outliers', ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_double, c_uint8_p, c_int64_p]),
    ('mvus_ba_set_allreduce', ctypes.c_int, [ctypes.c_void_p, ALLREDUCE_FN, ctypes.c_void_p, ctypes.c_int32]),
    ('mvus_rccl_available', ctypes.c_int, []),
    ('mvus_rccl_unique_id', ctypes.c_int, [c_uint8_p]),
    ('mvus_ba_set_rccl', ctypes.c_int, [ctypes.c_void_p, c_uint8_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    ('mvus_ba_time_allreduce', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, c_double_p]),
    ('mvus_ba_set_time_shard', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, c_int32_p, ctypes.c_int32]),
    ('mvus_ba_time_kernel', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, c_double_p]),
    ('mvus_ba_set_x', ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    ('mvus_spline_eval', ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, c_double_p, c_int64_p, c_double_p, c_double_p, ctypes.c_int64, c_double_p, c_double_p, c_int32_p]),
    ('mvus_spline_lsq', ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, c_double_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p]),
    ('mvus_spline_smooth', ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, c_double_p, c_double_p, ctypes.c_double, c_int32_p, c_double_p, c_double_p,
                                          c_double_p, c_int32_p]),
    ('mvus_spline_fit_open', ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, c_double_p, c_double_p, ctypes.POINTER(ctypes.c_void_p)]),
    ('mvus_spline_fit_smooth', ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_int32), c_double_p, c_double_p,
                                              ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]),
    ('mvus_spline_fit_close', None, [ctypes.c_void_p]),
    ('mvus_pnp_ransac', ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_double, ctypes.c_int32,
                                       ctypes.c_uint64, c_double_p, c_double_p, c_uint8_p, c_int64_p]),
    ('mvus_triangulate', ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
]

_lib = None


def load(path=None):
    """dlopen libmvusba.so and declare the prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            'libmvusba.so is not built (%s). Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C mvus_amd/csrc`. mvus_amd has no CPU fallback for the BA hot path.' % p)
    # PyTorch-ROCm wheels bundle their own libamdhip64; if libmvusba.so pulls in /opt/rocm's copy first, a later
    # `import torch` in the same process finds no GPU (observed: "ProcessGroupNCCL ... no GPUs found").  Let torch's
    # runtime initialise first whenever torch is installed, so both share one HIP runtime.
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass
    lib = ctypes.CDLL(p)
    for name, restype, argtypes in API:
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    check_abi(lib)
    if path is None:
        _lib = lib
    return lib


ABI_VERSION = 7


def check_abi(lib):
    """The struct layouts of this binding against the library's own sizeof (mvus_abi_sizes): a stale binding fails at load."""
    so, sr, sp = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    ver = lib.mvus_abi_sizes(ctypes.byref(so), ctypes.byref(sr), ctypes.byref(sp))
    mine = (ctypes.sizeof(MvusSolveOpts), ctypes.sizeof(MvusResult), ctypes.sizeof(MvusProblem))
    if ver != ABI_VERSION or (so.value, sr.value, sp.value) != mine:
        raise RuntimeError('libmvusba.so ABI %d with struct sizes %s does not match this binding (ABI %d, %s): rebuild the library'
                           % (ver, (so.value, sr.value, sp.value), ABI_VERSION, mine))


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def rccl_available():
    """True when the library can open RCCL in this process (mvus_rccl_available: no device, no communicator touched); else raises."""
    lib = load()
    rc = lib.mvus_rccl_available()
    if rc != MVUS_OK:
        raise RuntimeError('mvus_rccl_available (%d): %s' % (rc, lib.mvus_last_error(None).decode()))
    return True


def rccl_unique_id():
    """128 bytes (RCCL's ncclUniqueId) from mvus_rccl_unique_id: obtained by ONE rank, handed to all of them (mvus_ba_set_rccl)."""
    lib = load()
    buf = (ctypes.c_uint8 * 128)()
    rc = lib.mvus_rccl_unique_id(buf)
    if rc != MVUS_OK:
        raise RuntimeError('mvus_rccl_unique_id failed (%d): %s' % (rc, lib.mvus_last_error(None).decode()))
    return bytes(buf)


def make_problem_struct(prob, device=0, stream=None):
    """BAProblem -> (MvusProblem, keepalive) -- keepalive holds the numpy buffers the struct points into."""
    c = lambda a, t: np.ascontiguousarray(a, dtype=t)
    keep = dict(
        det_offsets=c(prob.det_offsets, np.int64), frame=c(prob.frame, np.float64), u_raw=c(prob.u_raw, np.float64),
        v_raw=c(prob.v_raw, np.float64), img_height=c(prob.img_height, np.float64), K=c(prob.K, np.float64),
        dist=c(prob.dist, np.float64), interval=c(prob.interval, np.float64), knot_offsets=c(prob.knot_offsets, np.int64),
        knots=c(prob.knots, np.float64))
    s = MvusProblem()
    s.num_cam = prob.num_cam
    s.opt_calib = int(prob.opt_calib)
    s.undist_points = int(prob.undist_points)
    s.rs_free = int(prob.rs_free)
    s.rs_bounds = int(prob.rs_bounds)
    s.motion_reg = int(prob.motion_reg)
    s.motion_type = int(prob.motion_type)
    s.opt_sync = int(getattr(prob, 'opt_sync', True))
    s.motion_weight = float(prob.motion_weight)
    s.det_offsets = keep['det_offsets'].ctypes.data_as(c_int64_p)
    for k in ('frame', 'u_raw', 'v_raw', 'img_height', 'K', 'dist', 'interval', 'knots'):
        setattr(s, k, dptr(keep[k]))
    s.knot_offsets = keep['knot_offsets'].ctypes.data_as(c_int64_p)
    s.num_splines = prob.S
    s.device = int(device)
    s.stream = ctypes.c_void_p(stream) if stream else None
    return s, keep


def default_opts(solver=SOLVER_TRF_LSMR, jac_mode=JAC_PATTERN, max_nfev=10):
    """scipy least_squares defaults with the arguments Scene.BA passes (common.py:670)."""
    o = MvusSolveOpts()
    o.solver, o.jac_mode, o.max_nfev = solver, jac_mode, max_nfev
    o.ftol, o.xtol, o.gtol = 1e-8, 1e-12, 1e-8
    o.lsmr_atol, o.lsmr_btol, o.lsmr_conlim, o.lsmr_maxiter = 1e-6, 1e-6, 1e8, 0
    o.verbose = 0
    o.lm_lambda_min = 3e-3
    o.lm_trust_radius = -1.0
    return o
