/* mvus_ba.h -- C ABI of libmvusba.so, the MI355X (gfx950) bundle-adjustment core.
 *
 * The reference (CenekAlbl/mvus) has no FFI on this path: the boundary is the Python call
 *
 *     res = least_squares(fn, model, jac_sparsity=A, tr_solver='lsmr', xtol=1e-12,
 *                         max_nfev=max_iter, verbose=0, bounds=bounds_rs)
 *                                   (multiviewunsynch/reconstruction/common.py:670)
 *
 * inside Scene.BA (common.py:441-697) with fn = error_BA (common.py:448-487) and A = jac_BA()
 * (common.py:490-610), plus Scene.remove_outliers (common.py:700-717).  The entry points below are
 * what a ctypes binding placed at that call site needs; each one cites the reference code it
 * replaces.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MVUS_E_* code; nothing throws across the ABI;
 *     mvus_last_error() gives the message of the last failure on the handle (or, with NULL, of the
 *     last failed mvus_ba_create on this thread);
 *   - all `const double*` / `double*` arguments are HOST pointers unless the name ends in `_dev`;
 *     the caller owns every buffer it passes; the handle owns all device memory;
 *   - the parameter vector x has the reference layout (common.py:615-650):
 *       [alpha(C) beta(C) rs(C) cam_0(P) .. cam_{C-1}(P) spline_0: cx(n_0) cy(n_0) cz(n_0) spline_1 ..]
 *     P = 6 (rvec,t) or 15 (fx,fy,cx,cy,rvec,t,k1,k2,p1,p2,k3) when opt_calib (common.py:1113-1124);
 *   - the residual vector f has the reference row order (common.py:476-485): per camera
 *     [|ex|(M_c) |ey|(M_c)], then the motion-regulariser rows (T);
 *   - one handle per (host thread, GPU, stream); calls on one handle must not overlap.
 *   - there is NO CPU fallback: without a HIP device mvus_ba_create fails with MVUS_E_HIP.
 */
#ifndef MVUS_BA_H
#define MVUS_BA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVUS_OK 0
#define MVUS_E_INVALID (-1)  /* bad argument / inconsistent problem description */
#define MVUS_E_HIP (-2)      /* HIP runtime error (no device, allocation, launch) */
#define MVUS_E_NUMERIC (-3)  /* non-finite residuals at x0, x0 outside bounds (scipy raises ValueError) */
#define MVUS_E_COMM (-4)     /* the all-reduce callback reported a failure */
#define MVUS_E_UNSUPPORTED (-5)  /* the problem is outside what this solver handles (MVUS_SOLVER_LM_SCHUR: motion rows that couple control
                                 * points more than 16 apart, or more than 6 apart on a time shard): the other solver has no such limit */

#define MVUS_E_RESHARD (-6)  /* time shard (mvus_ba_set_time_shard): at the point the solver has reached, a detection or motion row touches
                              * control points outside this rank's slice +- halo (the time stamps drifted since the cuts were made).
                              * mvus_ba_solve then RETURNS that point in x (and nfev / cost so far in res): the caller re-cuts there
                              * and continues (mvus_amd.dist.solve_time_sharded).  Raised on every rank of the job together.
                              * The reference re-evaluates visibility at every call: common.py:317, tools/util.py:90-116 */

#define MVUS_MOTION_F 0  /* constant-force prior       common.py:984-998 */
#define MVUS_MOTION_KE 1 /* constant-kinetic-energy    common.py:976-981 */

/* Jacobian used by mvus_ba_solve */
#define MVUS_JAC_ANALYTIC 0 /* full analytic block-sparse Jacobian */
#define MVUS_JAC_PATTERN 1  /* analytic, masked to the reference sparsity pattern jac_BA builds at x0
                               (3 nearest knots per row, common.py:559-563,573-585) */
#define MVUS_JAC_FD 2       /* scipy's own estimate: sparse 2-point forward differences over that pattern, columns
                               perturbed group-wise (scipy/optimize/_numdiff.py:628-700), one residual launch per
                               group.  Needs mvus_ba_set_fd_groups. */

/* Solver used by mvus_ba_solve */
#define MVUS_SOLVER_TRF_LSMR 0 /* restatement of scipy trf + lsmr (the reference's optimiser), J kept as operator */
#define MVUS_SOLVER_LM_SCHUR 1 /* Levenberg-Marquardt on device-assembled normal equations: spline block eliminated
                                  by a partitioned band Cholesky + cyclic reduction, Schur complement on the fp64
                                  matrix cores, reduced camera system by block Gauss-Jordan; all vectors stay on the
                                  device, one 7-scalar read-back per iteration.  The damping (and its growth factor)
                                  is carried from one mvus_ba_solve on a handle to the next. */

typedef struct mvus_ba mvus_ba; /* opaque */

/* The Scene state Scene.BA closes over (common.py:441-697), cameras in sequence[:numCam] order. */
typedef struct mvus_problem {
  int32_t num_cam;       /* C = numCam */
  int32_t opt_calib;     /* settings['opt_calib']      common.py:621 */
  int32_t undist_points; /* settings['undist_points']  common.py:126 */
  int32_t rs_free;       /* `rs` argument of Scene.BA: rolling-shutter column in the pattern, common.py:518 */
  int32_t rs_bounds;     /* `rs_bounds`: 0 <= rs <= 1   common.py:655-662 */
  int32_t motion_reg;    /* `motion_reg`                common.py:483-485 */
  int32_t motion_type;   /* MVUS_MOTION_*  settings['motion_type'] common.py:416-420 */
  int32_t opt_sync;      /* settings['opt_sync'] (absent = 1): 0 freezes alpha and beta -- their columns leave the
                            pattern and every Jacobian, common.py:512-515 */
  double motion_weight;  /* `motion_weights`            common.py:413 */
  const int64_t* det_offsets; /* [C+1] detections of camera c are [det_offsets[c], det_offsets[c+1]) */
  const double* frame;        /* [M] detections[c][0]  (np.loadtxt usecols=(2,0,1), common.py:1190) */
  const double* u_raw;        /* [M] detections[c][1] */
  const double* v_raw;        /* [M] detections[c][2] */
  const double* img_height;   /* [C] cameras[c].resolution[1]  common.py:125 */
  const double* K;            /* [C*4] fx fy cx cy, used when !opt_calib */
  const double* dist;         /* [C*5] k1 k2 p1 p2 k3, used when !opt_calib */
  int32_t num_splines;        /* S = spline['int'].shape[1] */
  const double* interval;     /* [2*S] row-major spline['int']: S starts then S ends */
  const int64_t* knot_offsets; /* [S+1] */
  const double* knots;        /* concatenated spline['tck'][s][0] */
  int32_t device;             /* HIP device ordinal */
  void* stream;               /* hipStream_t to run on, or NULL for a stream owned by the handle */
} mvus_problem;

typedef struct mvus_solve_opts {
  int32_t solver;    /* MVUS_SOLVER_* */
  int32_t jac_mode;  /* MVUS_JAC_* */
  int32_t max_nfev;  /* max_iter of Scene.BA (10)           common.py:441,670 */
  double ftol;       /* 1e-8  scipy default                  */
  double xtol;       /* 1e-12 as passed at common.py:670     */
  double gtol;       /* 1e-8  scipy default                  */
  double lsmr_atol;  /* 1e-6  scipy lsmr default             */
  double lsmr_btol;  /* 1e-6                                 */
  double lsmr_conlim; /* 1e8                                 */
  int32_t lsmr_maxiter; /* 0 -> min(m, n)                    */
  int32_t verbose;
  double lm_lambda_min; /* MVUS_SOLVER_LM_SCHUR: floor of the Marquardt damping (3e-3; 0 = none).  Directions the data does
                           not determine (a control point seen by one camera: depth along its rays; rs against beta when the
                           image row hardly varies) have curvature << lambda * diag(H) and stay where they are instead of
                           following the noise -- the effect the reference gets from LSMR's truncated solves (common.py:670).
                           Problems whose motion rows reach over more than six control points (knots less than a frame apart: more
                           control points than detections) use at least 0.3 */
  double lm_trust_radius; /* MVUS_SOLVER_LM_SCHUR: the reference's trust region on top of the damping.  scipy's TRF (common.py:670,
                           x_scale = 1) bounds the Euclidean length of a step by Delta: Delta_0 = |x0|, Delta = 0.25 |step| after a
                           step whose actual / predicted reduction is below 0.25, Delta *= 2 after one above 0.75 that reached the
                           bound (scipy/optimize/_lsq/common.py update_tr_radius).  Here a damped step longer than Delta is cut back
                           to Delta along its direction and the damping is raised in proportion for the next solve.
                           0: Delta_0 = |x0| as scipy; > 0: this Delta_0; < 0: no trust region (damping only, rounds 2-4) */
} mvus_solve_opts;

/* scipy.optimize.OptimizeResult fields Scene.BA returns (common.py:670,697) */
typedef struct mvus_result {
  double cost;        /* 0.5*|f|^2 */
  double optimality;  /* |g|_inf (scaled by the Coleman-Li vector when bounded); LM_SCHUR stopped by max_nfev: at its last
                         linearisation point (the final accepted point is not re-linearised) */
  int32_t nfev, njev;
  int32_t status;     /* 0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 ftol&xtol (scipy codes) */
  int32_t lin_iters;  /* total LSMR iterations / Cholesky solves */
  double solve_ms;    /* wall time of the call */
  double initial_cost;
} mvus_result;

/* sum-all-reduce of `count` doubles at device address `buf_dev`, issued on `stream`; returns 0 on success.
 * Called by the solver once per normal-equation assembly / J^T u product when observations are sharded. */
typedef int (*mvus_allreduce_fn)(void* user, void* buf_dev, size_t count, void* stream);

/* Always start from mvus_default_opts: a zero-filled mvus_solve_opts is NOT the default (lm_trust_radius = 0 means "scipy's
 * Delta_0", i.e. ON; the defaults are listed in INTEGRATION.md). */
void mvus_default_opts(mvus_solve_opts* opts);

/* Layout check for bindings in other languages (ctypes / cgo / JNI stubs that restate the structs): writes sizeof(mvus_solve_opts),
 * sizeof(mvus_result), sizeof(mvus_problem) as this library was compiled (any pointer may be NULL) and returns MVUS_ABI_VERSION.  A
 * binding asserts these against its own struct definitions at load time -- mvus_solve_opts has grown over the rounds (lm_lambda_min,
 * lm_trust_radius) and a stale stub would otherwise hand the library a short buffer.  The version is raised whenever a struct or a
 * prototype of this header changes.  Stateless, no device is touched.  No reference counterpart (the reference has no FFI). */
#define MVUS_ABI_VERSION 7
int32_t mvus_abi_sizes(int32_t* solve_opts_size, int32_t* result_size, int32_t* problem_size);

/* Copies the problem to the GPU, undistorts observations once when calibration is fixed
 * (detection_to_global, common.py:126).  */
int mvus_ba_create(const mvus_problem* p, mvus_ba** out);
void mvus_ba_destroy(mvus_ba* h);
const char* mvus_last_error(const mvus_ba* h);

int64_t mvus_ba_num_params(const mvus_ba* h);      /* n = 3C + C*P + 3*sum(n_s)            common.py:652 */
int64_t mvus_ba_num_residuals(const mvus_ba* h);   /* m = 2M + T                                          */
int64_t mvus_ba_num_motion_rows(const mvus_ba* h); /* T = len(spline_to_traj()[0]) when motion_reg else 0 */
int32_t mvus_ba_num_slots(const mvus_ba* h);       /* NS = 3 + P + 12 Jacobian slots per residual row     */

/* error_BA(x): f[m]  (common.py:448-487) */
int mvus_ba_residual(mvus_ba* h, const double* x, double* f);

/* Residual plus block-sparse Jacobian of the 2M detection rows (analytic; replaces the 2-point finite
 * differences scipy takes over jac_BA's pattern).  Outputs (any may be NULL):
 *   f[m];  J[2*NS*M]: J[(a*NS + k)*M + i] = d|e_a(i)| / d slot k, a = 0 (x) / 1 (y), observation i in
 *   camera-segmented order; slots: 0 alpha, 1 beta, 2 rs, 3..3+P camera params, 3+P+3q+d control point
 *   ctrl[i]+q coordinate d;  ctrl[M]: global index of the first active control point, -1 if not visible. */
int mvus_ba_residual_jacobian(mvus_ba* h, const double* x, int32_t jac_mode, double* f, double* J, int32_t* ctrl);

/* Motion-regulariser rows and their Jacobian: mf[T], mJ[36*T] (mJ[k*T + j], k = 12*sample + 3*q + d for the
 * samples j-1, j, j+1), mctrl[3*T] first control point of each of the three samples (-1 = unused). */
int mvus_ba_motion_rows(mvus_ba* h, const double* x, int32_t jac_mode, double* mf, double* mJ, int32_t* mctrl);

/* Fix the reference sparsity pattern at x0 (jac_BA + compute_visibility, common.py:427-438,490-610), computed on the
 * GPU.  pat[M]: per detection row a PATTERN CODE  p | (mask << 25)  -- p = global index of the lowest in-pattern
 * control point, bit k of the 4-bit mask set when control point p+k is in the pattern (three bits set) -- or -1 for
 * an all-zero row.  The nearest-three rule of common.py:559-563 gives three consecutive points (mask 0x7).  In rows
 * where exactly one of two control points with the SAME centre knot (t[2:-2] repeats the interval start and end) is
 * among the three nearest, which twin np.argsort returns depends on numpy's sort kernel (scalar / AVX2 / AVX512
 * builds differ): the reference's pattern is implementation defined there.  Those rows get the canonical choice and
 * bit 30 (MVUS_PAT_TIE) set in pat_out. */
#define MVUS_PAT_SHIFT 25
#define MVUS_PAT_TIE (1 << 30)
int mvus_ba_set_pattern(mvus_ba* h, const double* x0, int32_t* pat_out);

/* The pattern codes of the T motion rows (common.py:573-585; parameter independent, canonical unless uploaded). */
int mvus_ba_motion_pattern(mvus_ba* h, int32_t* motion_pat_out);

/* Supply the pattern instead: the matrix A the reference passes as jac_sparsity at common.py:670 is an INPUT of the
 * call this library replaces.  pat[M] / motion_pat[T] (NULL = keep) are pattern codes as above (bit 30 ignored); every
 * point of a code must belong to one spline.  Stays in force -- MVUS_JAC_PATTERN solves do not recompute it -- until
 * the next mvus_ba_set_pattern or mvus_ba_remove_outliers.  Column groups for MVUS_JAC_FD must be set afterwards. */
int mvus_ba_upload_pattern(mvus_ba* h, const int32_t* pat, const int32_t* motion_pat);

/* MVUS_SOLVER_LM_SCHUR.  The normal equations of the analytic Jacobian are assembled window-major (ba_assemble_win.hip.h): every
 * entry has one writer and one order of additions, no floating-point atomics -- a solve gives the same bits on every run, on one
 * rank and on every rank of a sharded run (the sums over the ranks are the collective's).  This call is kept for the ABI and
 * changes nothing (round 3 had an opt-in deterministic mode beside an atomic default).  No counterpart in the reference (scipy is
 * deterministic; the LM solver keeps that property). */
int mvus_ba_set_deterministic(mvus_ba* h, int32_t on);
/* *fell_back = 1 when the handle's LAST assembly went through the detection-major kernel, which adds with fp64 atomics (last-bit
 * differences from run to run): a camera whose frames are not in non-decreasing order, more than 256 cameras, or normal equations
 * formed from a stored Jacobian that is not the analytic one (pattern-masked, finite differences).  0 otherwise. */
int mvus_ba_deterministic_fallback(mvus_ba* h, int32_t* fell_back);

/* Column groups for MVUS_JAC_FD: groups[n] in [0, num_groups), two columns share a group only if no row of the
 * reference pattern contains both (scipy.optimize._numdiff.group_columns on jac_BA's matrix). */
int mvus_ba_set_fd_groups(mvus_ba* h, const int32_t* groups, int32_t num_groups);

/* scipy.optimize._numdiff.group_columns(A, order) -- what least_squares runs on jac_sparsity (the reference passes
 * jac_BA's matrix, common.py:665-670) -- on the HOST, from the pattern's entries instead of a scipy.sparse matrix:
 * entries (rows[k], cols[k]), k < nnz, of an m x n pattern in any order, duplicates allowed; order[n] = the column
 * permutation (scipy: numpy.random.RandomState(0).permutation(n)); groups[n] out.  Same greedy pass as scipy's
 * group_sparse over the permuted columns, hence the same groups.  Returns the number of groups, or a negative MVUS_E_*.
 * Stateless; no device is touched. */
int32_t mvus_group_columns(int64_t m, int64_t n, int64_t nnz, const int64_t* rows, const int64_t* cols, const int64_t* order, int32_t* groups);

/* The same grouping straight from the pattern CODES of a problem (pat[M] as mvus_ba_set_pattern / mvus_ba_upload_pattern use them,
 * motion_pat[T] or NULL without motion rows): the entries of jac_BA's matrix (common.py:559-610 -- per detection row alpha, beta
 * [, rs], the camera's parameters and the three nearest control points' coordinates; per motion row its control points) are
 * generated here instead of being handed over as two nnz-long arrays.  Host only; p as for mvus_ba_create. */
int32_t mvus_fd_groups(const mvus_problem* p, const int32_t* pat, const int32_t* motion_pat, const int64_t* order, int32_t* groups);

/* y[m] = J v (v[n]);  z[n] = J^T u (u[m]) with the Jacobian currently held by the handle
 * (after mvus_ba_residual_jacobian / inside solve).  Test and integration hooks for the operator. */
int mvus_ba_jv(mvus_ba* h, const double* v, double* y);
int mvus_ba_jtu(mvus_ba* h, const double* u, double* z);

/* Gauss-Newton normal equations of the current Jacobian: dense copies for inspection.
 *   g[n] = J^T f;  JtJ_cam[C*B*B] camera diagonal blocks, B = 3+P (row-major);
 *   band: block-banded spline part, band[((g*W + w)*3 + a)*3 + b] = (J^T J)[3g+a, 3(g+w)+b], w < W;
 *   cross[C*B*3N]: cross[(c*B + k)*3N + 3g + d].  Any output may be NULL; *W_out receives W. */
int mvus_ba_normal_equations(mvus_ba* h, double* g, double* JtJ_cam, double* band, double* cross, int32_t* W_out);

/* One damped Gauss-Newton step of MVUS_SOLVER_LM_SCHUR with the Jacobian and residual currently held:
 * p[n] = -(J^T J + lambda D)^-1 J^T f, D = diag(J^T J) (1 where 0).  Inspection hook for the whole solve chain
 * (assembly, band solver, Schur complement, reduced system); no trial evaluation, no bounds.  No reference counterpart. */
int mvus_ba_lm_step(mvus_ba* h, double lambda, double* p_out);

/* The least_squares call of Scene.BA (common.py:670) -- x is read and overwritten with res.x.
 * lb/ub come from opts of the problem (rs_bounds).  f_out[m] may be NULL.  x, res and f_out are complete on return.  MVUS_SOLVER_LM_SCHUR
 * with f_out == NULL may return while device work for the NEXT call is still running on the handle's stream (the linearisation at the
 * returned point, enqueued before the accept / reject decision was known); every later call on the handle is ordered behind it, and a
 * solve that continues from the returned x reuses it together with f(x) and the cost instead of evaluating them again.
 * Termination tests, nfev/njev counting and status codes follow scipy for both solvers (max_nfev = the reference's
 * max_iter); MVUS_SOLVER_LM_SCHUR does not re-linearise the accepted point when max_nfev stops it. */
int mvus_ba_solve(mvus_ba* h, double* x, const mvus_solve_opts* opts, mvus_result* res, double* f_out);

/* Scene.remove_outliers (common.py:700-717): keep[i] = sqrt(ex^2 + ey^2) < thres, camera-segmented order. */
int mvus_ba_outlier_mask(mvus_ba* h, const double* x, double thres, uint8_t* keep);

/* Scene.remove_outliers applied to the handle itself (common.py:709-714: `detections[i] = detections[i][:, error<thres]`):
 * evaluates the mask at x, compacts the device-resident detection arrays in place (order kept) and rebuilds the launch
 * tables, so the next mvus_ba_solve -- the second BA of main.py:59 -- runs on the inliers without a new handle or any
 * re-upload.  keep_out[M_old] (may be NULL) receives the mask, det_offsets_out[C+1] the new camera offsets.
 * On a sharded handle every rank filters its own detections (all ranks must call it: the global row count is re-summed). */
int mvus_ba_remove_outliers(mvus_ba* h, const double* x, double thres, uint8_t* keep_out, int64_t* det_offsets_out);

/* Multi-GPU: observations sharded across ranks, this handle holds one shard.  `is_root` ranks add the
 * replicated terms (motion rows, damping) exactly once. */
int mvus_ba_set_allreduce(mvus_ba* h, mvus_allreduce_fn fn, void* user, int32_t is_root);

/* The same sums by RCCL called from the library itself (no host language in the iteration): every collective is one
 * ncclAllReduce(double, sum, in place) on the handle's stream.  One rank obtains an id with mvus_rccl_unique_id (128 bytes, RCCL's
 * ncclUniqueId) and hands it to the others by any means (a torch.distributed / MPI broadcast, a file); every rank then calls
 * mvus_ba_set_rccl, which joins the communicator (ncclCommInitRank: collective, blocks until all `world` ranks have called) on the
 * handle's device.  librccl.so.1 is opened at run time (the copy the process already holds -- e.g. PyTorch's -- else the loader's):
 * the library has no link-time dependency on it and MVUS_E_COMM reports its absence.  Replaces a callback set earlier; is_root as
 * above.  The sums of a given topology are RCCL's: the same bits on every run.  SURVEY 8e's collective; no reference counterpart. */
/* mvus_rccl_available: MVUS_OK when librccl.so.1 can be opened and its entry points resolved in this process, else MVUS_E_COMM (message:
 * mvus_last_error(NULL)).  It touches no device and no communicator: a job asks it on EVERY rank and exchanges the answers before any
 * rank calls mvus_ba_set_rccl -- ncclCommInitRank is itself a blocking collective, and a rank that cannot open RCCL would otherwise
 * leave the others waiting inside it (mvus_amd/dist.py::agree_on_rccl). */
int mvus_rccl_available(void);
int mvus_rccl_unique_id(uint8_t id_out[128]);
int mvus_ba_set_rccl(mvus_ba* h, const uint8_t id[128], int32_t rank, int32_t world, int32_t is_root);
/* Measurement hook: `reps` back-to-back sums of `count` doubles (a scratch buffer) through the route installed on the handle (the
 * callback or RCCL) between two hipEvents; average milliseconds per collective. */
int mvus_ba_time_allreduce(mvus_ba* h, int64_t count, int32_t reps, double* avg_ms);

/* Time sharding for the LM/Schur solver ("shard by time range"): this handle was created with the
 * detections of ONE time slice and owns the spline control points [ctrl_cuts[rank], ctrl_cuts[rank+1]) (global
 * control-point indices over all splines, ctrl_cuts[0] = 0, ctrl_cuts[world] = N; every rank passes the same array).
 * Each rank then assembles and factorises only its slice of the spline blocks; the cross block never leaves its GPU.
 * Per LM iteration the all-reduce callback (mvus_ba_set_allreduce, required) sums: the camera blocks + the blocks of the
 * control points within `halo` of a cut (rows of both neighbours land there), the separator system of the band solver,
 * the Schur complement contributions, the step, diag(H) and g -- a few MB in all, instead of the whole cross block.
 * A detection whose four control points leave [own range -+ halo) (time-stamp drift larger than the halo) makes the
 * next solve fail with MVUS_E_HIP and a message saying so.  Motion rows are evaluated by the rank owning their first
 * control point (is_root of mvus_ba_set_allreduce is ignored).  Call before mvus_ba_set_allreduce.  world = 1 turns
 * it off.  No reference counterpart (the reference is single-process). */
int mvus_ba_set_time_shard(mvus_ba* h, int32_t rank, int32_t world, const int32_t* ctrl_cuts, int32_t halo);

/* Measurement hook for bench.py: runs `launches` back-to-back launches of one kernel on the handle's
 * stream between two hipEvents and returns the average duration in milliseconds.
 *   which: 0 residual, 1 residual+Jacobian with the outputs ROTATING over >= 3 buffer sets (>= 1 GiB in rotation, so no
 *   launch writes into lines its predecessor left in the 256 MiB Infinity Cache), 2 J v, 3 J^T u, 4 normal-equation
 *   assembly from the materialised Jacobian, 5 residual+Jacobian re-launched into one buffer set (cache-resident variant,
 *   for comparison), 6 the fused Jacobian + normal-equation assembly of the LM path (no Jacobian in memory) */
int mvus_ba_time_kernel(mvus_ba* h, int32_t which, int32_t launches, double* avg_ms);

/* Upload x to the handle without evaluating anything (used with mvus_ba_time_kernel). */
int mvus_ba_set_x(mvus_ba* h, const double* x);

/* Two-view linear triangulation, epipolar.triangulate_matlab (reconstruction/epipolar.py:497-510) as called by
 * Scene.triangulate (common.py:783): for each of N point pairs the right singular vector of the smallest singular value
 * of the 4x4 matrix built from the two projections, divided by its last component.  x1, x2: [2*N] (u(N) then v(N)) pixel
 * coordinates in camera 1 / 2; P1, P2: [12] row-major 3x4 projection matrices; X: [4*N] rows x, y, z, 1.
 * err1 / err2 (NULL = skip): [N] reprojection distances of X in the two cameras (epipolar.reprojection_error of
 * Camera.projectPoint, common.py:786-787), the quantity Scene.triangulate thresholds.  Stateless: no handle; errors of
 * this call are reported by mvus_last_error(NULL).  No CPU fallback. */
int mvus_triangulate(int32_t device, int64_t N, const double* x1, const double* x2, const double* P1, const double* P2,
                     double* X, double* err1, double* err2);

/* Scene.spline_to_traj (common.py:273-301): evaluate the S trajectory splines at nt timestamps.  interval: [2*S] row-major
 * spline['int'] (starts then ends); knot_offsets [S+1] / knots: concatenated spline['tck'][s][0]; coefs: concatenated
 * spline['tck'][s][1] (cx(n_s) cy(n_s) cz(n_s) per spline).  which[nt] receives the interval each timestamp belongs to
 * (start <= t <= end, closed like common.py:292) or -1; X[3*nt] (x(nt) y(nt) z(nt)) the point, 0 where which = -1.
 * Same recurrence as scipy.interpolate.splev (FITPACK splev.f / fpbspl.f).  Stateless; no CPU fallback. */
int mvus_spline_eval(int32_t device, int32_t S, const double* interval, const int64_t* knot_offsets, const double* knots,
                     const double* coefs, int64_t nt, const double* t, double* X, int32_t* which);

/* Least-squares cubic spline on a FIXED knot vector (num_knots = n + 4, first/last four equal): coefs[3*n] (cx cy cz)
 * minimising sum_i |X(t_i) - X_i|^2 over the m data points t[m], X[3*m] (x(m) y(m) z(m)), t inside [knots[3], knots[n]].
 * The coefficient refit of Scene.traj_to_spline (common.py:224-270) once FITPACK's adaptive search has placed the knots (that
 * search stays on the host); equals scipy.interpolate.make_lsq_spline.  MVUS_E_NUMERIC when a coefficient has no data. */
int mvus_spline_lsq(int32_t device, int32_t num_knots, const double* knots, int64_t m, const double* t, const double* X, double* coefs);

/* The smoothing-spline fit inside Scene.traj_to_spline (common.py:247, :267): scipy.interpolate.splprep(X, u=u, s=s, k=3), i.e.
 * FITPACK parcur / fppara with iopt = 0, unit weights, ub = u[0], ue = u[m-1], nest = m + 6.  u[m] strictly increasing, X[3*m]
 * (x(m) y(m) z(m)), s > 0.  Outputs: *n_out knots in t_out (room for m + 6), the n - 4 coefficients of dimension d at
 * c_out[d * (m + 6) + j] (room for 3 * (m + 6)), *fp_out = the weighted residual sum of squares, *ier_out = FITPACK's ier
 * (0 smoothing spline, -1 interpolating spline, -2 least-squares polynomial, 1..3 its warnings).  FITPACK's knot search
 * (fpknot), root finding (fprati) and discontinuity jumps (fpdisc) are followed line by line; the least-squares problems are
 * solved through banded normal equations on the device instead of row-wise Givens rotations -- same knots, coefficients to
 * ~1e-9 relative on the fixtures (tests/test_traj_to_spline.py), and O(m + n) per pass where FITPACK's smoothing iteration is
 * O(n^2).  The smooth_factor loop around it (common.py:241-262) is mvus_amd.spline.traj_fit.  Stateless; no CPU fallback. */
int mvus_spline_smooth(int32_t device, int64_t m, const double* u, const double* X, double s, int32_t* n_out, double* t_out, double* c_out,
                       double* fp_out, int32_t* ier_out);

/* The same fit as a session: the samples are checked and uploaded once, the work arrays stay allocated, and every call of
 * mvus_spline_fit_smooth is one splprep(X, u=u, s=s, k=3) on them -- Scene.traj_to_spline's smooth_factor loop
 * (common.py:241-262) calls splprep about a dozen times on the same part with s doubled or divided by 1.5.  Same results
 * as mvus_spline_smooth (which is open + smooth + close).  t_out[m + 6], c_out[3][m + 6] as there. */
typedef struct mvus_spline_fit mvus_spline_fit;
int mvus_spline_fit_open(int32_t device, int64_t m, const double* u, const double* X, mvus_spline_fit** out);
int mvus_spline_fit_smooth(mvus_spline_fit* fit, double s, int32_t* n_out, double* t_out, double* c_out, double* fp_out, int32_t* ier_out);
void mvus_spline_fit_close(mvus_spline_fit* fit);

/* Scene.get_camera_pose's cv2.solvePnPRansac(objectPoints, imagePoints, K, d, reprojectionError=error) (common.py:744; OpenCV is
 * a third-party dependency absent from this image: parity unpinned, see pnp.hip.h).  X[3*N] object points (x(N) y(N) z(N)),
 * uv[2*N] raw pixels (u(N) v(N)), K = fx fy cx cy, d = k1 k2 p1 p2 k3.  `iterations` hypotheses (OpenCV's iterationsCount,
 * default 100) from six sampled points each (counter-based sampling from `seed`: deterministic), scored by the number of points
 * with reprojection error <= reproj_error pixels; the best pose is refined on its inliers by damped Gauss-Newton on the pixel
 * reprojection error (OpenCV's SOLVEPNP_ITERATIVE refinement minimises the same quantity).  Outputs: rvec[3], tvec[3]
 * (cv2.Rodrigues convention), inliers[N] (1 = inlier of the best hypothesis; may be NULL), *n_inliers.
 * MVUS_E_NUMERIC when no hypothesis is supported by six points.  Stateless; no CPU fallback. */
int mvus_pnp_ransac(int32_t device, int64_t N, const double* X, const double* uv, const double* K, const double* d, double reproj_error,
                    int32_t iterations, uint64_t seed, double* rvec, double* tvec, uint8_t* inliers, int64_t* n_inliers);

#ifdef __cplusplus
}
#endif
#endif /* MVUS_BA_H */
