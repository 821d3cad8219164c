// libmvusba.so: HIP backend of the templated optimiser + the C ABI of include/mvus_ba.h.
// There is no CPU compute path in this library: every entry point that evaluates anything
// launches HIP kernels, and mvus_ba_create fails with MVUS_E_HIP when no device is usable.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "ba_kernels.hip.h"
#include "ba_problem.h"
#include "ba_solver.h"
#include "ba_schur_hip.hip.h"
#include "triangulate.hip.h"
#include "spline_ops.hip.h"
#include "spline_fit.hip.h"
#include "pnp.hip.h"

namespace mvus {

static inline int grid_for(long long len) { return (int)std::min<long long>(2048, std::max<long long>(1, (len + kThreads - 1) / kThreads)); }

struct HipBackend {
  HostProblem hp;
  DevProblem dp{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int device = 0;
  std::vector<void*> owned;
  // evaluation state
  CamState* cams = nullptr;
  const double* cams_for = nullptr;   // device x buffer whose camera states `cams` holds (nullptr: none); cleared when that buffer is rewritten
  double *J = nullptr, *mJ = nullptr, *x_cur = nullptr, *f_cur = nullptr;
  int32_t *span = nullptr, *pat0 = nullptr, *mctrl = nullptr;
  // first control point per detection (-1: not visible) at the x of the last residual-only evaluation: the window-major fused
  // assembly evaluates with a known knot span (`span` belongs to the held Jacobian and must survive residual evaluations)
  int32_t* rspan = nullptr;
  const double* rspan_for = nullptr;
  bool has_pattern = false, has_jacobian = false;
  bool held_analytic_at_xcur = false; // the held Jacobian is the analytic one of x_cur (mvus_ba_residual_jacobian): its normal equations can be formed by the fused window-major assembly
  bool det_assembly = false;          // mvus_ba_set_deterministic (kept for the ABI: the window-major assembly is deterministic by construction)
  bool pattern_uploaded = false;     // the caller supplied the reference's pattern (mvus_ba_upload_pattern): solve keeps it
  int32_t* ms_pat_dev = nullptr;
  double* det_alt[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // second set of detection arrays (remove_outliers ping-pong)
  int64_t det_capacity = 0;
  // reductions
  double *partials = nullptr, *scal_dev = nullptr, *scal_host = nullptr;
  double* scal_map = nullptr;     // device address of scal_host (mapped pinned memory): scalar results a kernel writes there need no copy kernel
  // pinned staging ring for the n-vectors that cross PCIe every solve / iteration (x, gradient, bounds): a copy from pageable
  // memory costs ~30 us of host-side staging per call and a synchronisation; through a pinned slot the upload is asynchronous
  static constexpr int kStageSlots = 4;
  double* stage[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_ev[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  int64_t stage_cap = 0;
  int stage_next = 0;
  // multi-GPU
  mvus_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int is_root = 1;
  int64_t m_glob = 0;
  // time shard (mvus_ba_set_time_shard): this handle holds the detections of one time slice and owns the control points
  // cuts[rank] .. cuts[rank+1]; the LM/Schur path then keeps the spline blocks of that slice (+- halo) only
  struct TimeShard { bool on = false; int rank = 0, world = 1, halo = 8; std::vector<int> cuts; } tshard;
  double lm_lambda = 0, lm_nu = 0;   // LM damping and its growth factor, carried from one solve on this handle to the next
  // MVUS_JAC_FD
  int32_t* fd_groups = nullptr;
  int fd_ngroups = 0;
  double *fd_F = nullptr, *fd_h = nullptr, *fd_dx = nullptr, *fd_xg = nullptr;
  std::string err;

  template <class T>
  T* dalloc(size_t count) {
    void* p = nullptr;
    MVUS_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    owned.push_back(p);
    return static_cast<T*>(p);
  }
  template <class T>
  T* dupload(const std::vector<T>& v) {
    T* p = dalloc<T>(v.size());
    if (!v.empty()) MVUS_HIP(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    return p;
  }

  void init(const mvus_problem* p) {
    device = p->device;
    MVUS_HIP(hipSetDevice(device));
    if (p->stream) stream = static_cast<hipStream_t>(p->stream);
    else { MVUS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); own_stream = true; }
    dp.C = hp.C; dp.P = hp.P; dp.NS = hp.NS; dp.S = hp.S; dp.calib = hp.calib; dp.undist = hp.undist;
    dp.rs_free = hp.rs_free; dp.sync_free = hp.sync_free; dp.T = hp.T; dp.M = hp.M; dp.N = hp.N;
    det_capacity = std::max<int64_t>(hp.M, 1);
    dp.frame = dupload(hp.frame); dp.u_raw = dupload(hp.u_raw); dp.v_raw = dupload(hp.v_raw);
    dp.H = dupload(hp.H); dp.Kfix = dupload(hp.K); dp.dfix = dupload(hp.dist);
    dp.sp.S = hp.S; dp.sp.istart = dupload(hp.istart); dp.sp.iend = dupload(hp.iend); dp.sp.knots = dupload(hp.knots);
    dp.sp.knot_off = dupload(hp.knot_off); dp.sp.ctrl_off = dupload(hp.ctrl_off); dp.sp.xoff = dupload(hp.xoff);
    dp.sp.lut = dupload(hp.lut); dp.sp.lut_off = dupload(hp.lut_off); dp.sp.lut_scale = dupload(hp.lut_scale);
    dp.sp.info = dupload(hp.sinfo);
    dp.mv.T = hp.T; dp.mv.type = hp.motion_type; dp.mv.w = hp.w;
    dp.mv.t = dupload(hp.ms_t); dp.mv.basis = dupload(hp.ms_basis); dp.mv.ctrl = dupload(hp.ms_ctrl);
    dp.mv.part = dupload(hp.ms_part); ms_pat_dev = dupload(hp.ms_pat); dp.mv.pat = ms_pat_dev;
    dp.mv.ctrl_x0 = dupload(hp.ctrl_x0); dp.mv.ctrl_stride = dupload(hp.ctrl_stride);
    dp.mv.row_lo = dupload(hp.ms_row_lo); dp.mv.row_hi = dupload(hp.ms_row_hi);
    dp.chunk_cam = dupload(hp.chunk_cam); dp.chunk_count = dupload(hp.chunk_count); dp.chunks = dupload(hp.chunks);
    dp.cam_chunk_off = dupload(hp.cam_chunk_off);
    std::vector<long long> cs(hp.chunk_start.begin(), hp.chunk_start.end()), doff(hp.det_off.begin(), hp.det_off.end());
    dp.chunk_start = dupload(cs); dp.det_off = dupload(doff);
    dp.n_chunks = (int)hp.chunk_cam.size();
    dp.mot_lo = 0; dp.mot_hi = 0x7fffffff;
    double* uo = dalloc<double>(hp.M); double* vo = dalloc<double>(hp.M);
    dp.u_obs = uo; dp.v_obs = vo;
    cams = dalloc<CamState>(hp.C);
    span = dalloc<int32_t>(hp.M); pat0 = dalloc<int32_t>(hp.M); rspan = dalloc<int32_t>(hp.M);
    mJ = dalloc<double>((size_t)36 * hp.T); mctrl = dalloc<int32_t>((size_t)3 * hp.T);
    x_cur = dalloc<double>(hp.n); f_cur = alloc(hp.m);      // (from the pool: an LM solve swaps it with its trial buffer)
    partials = dalloc<double>(2048); scal_dev = dalloc<double>(32);
    MVUS_HIP(hipHostMalloc(reinterpret_cast<void**>(&scal_host), 32 * sizeof(double), hipHostMallocMapped));      // [kMarkSlot]: the start mark of fetch_poll_begin
    for (int i = 0; i < 32; ++i) scal_host[i] = 0.0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&scal_map), scal_host, 0) != hipSuccess) scal_map = nullptr;
    stage_cap = std::max<int64_t>(hp.n, 1024);
    for (int i = 0; i < kStageSlots; ++i) {
      MVUS_HIP(hipHostMalloc(reinterpret_cast<void**>(&stage[i]), stage_cap * sizeof(double), hipHostMallocDefault));
      MVUS_HIP(hipEventCreateWithFlags(&stage_ev[i], hipEventDisableTiming));
    }
    MVUS_HIP(hipMemsetAsync(span, 0xff, sizeof(int32_t) * std::max<int64_t>(hp.M, 1), stream));
    MVUS_HIP(hipMemsetAsync(pat0, 0xff, sizeof(int32_t) * std::max<int64_t>(hp.M, 1), stream));
    MVUS_HIP(hipMemsetAsync(mctrl, 0xff, sizeof(int32_t) * std::max<int64_t>(3 * hp.T, 1), stream));
    if (dp.n_chunks > 0) {
      if (!hp.calib && hp.undist) hipLaunchKernelGGL(k_undistort_fixed, dim3(dp.n_chunks), dim3(kThreads), 0, stream, dp, uo, vo);
      else {
        MVUS_HIP(hipMemcpyAsync(uo, dp.u_raw, sizeof(double) * hp.M, hipMemcpyDeviceToDevice, stream));
        MVUS_HIP(hipMemcpyAsync(vo, dp.v_raw, sizeof(double) * hp.M, hipMemcpyDeviceToDevice, stream));
      }
    }
    MVUS_HIP(hipGetLastError());
    MVUS_HIP(hipStreamSynchronize(stream));
    m_glob = hp.m;
  }

  ~HipBackend() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (void* p : owned) (void)hipFree(p);
    for (auto& kv : pool_size) (void)hipFree(kv.first);
    if (scal_host) (void)hipHostFree(scal_host);
    if (fetch_ev) (void)hipEventDestroy(fetch_ev);
    for (double* p : sqh_host) if (p) (void)hipHostFree(p);
    if (lsmr_host) (void)hipHostFree(lsmr_host);
    for (int i = 0; i < 2; ++i) if (xmir_host[i]) (void)hipHostFree(xmir_host[i]);
    for (int i = 0; i < kStageSlots; ++i) {
      if (stage[i]) (void)hipHostFree(stage[i]);
      if (stage_ev[i]) (void)hipEventDestroy(stage_ev[i]);
    }
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }

  // ---- Backend concept -------------------------------------------------------------------------
  int64_t n() const { return hp.n; }
  int64_t m_local() const { return hp.m; }
  int64_t m_global() const { return m_glob; }
  // work vectors come from a size-keyed free list: the solvers allocate and release the same handful of n- and
  // m-sized buffers on every call, and hipMalloc/hipFree cost far more than the kernels they feed
  std::map<size_t, std::vector<double*>> pool_free;
  std::map<double*, size_t> pool_size;
  double* alloc(int64_t len) {
    const size_t bytes = (size_t)std::max<int64_t>(len, 1) * sizeof(double);
    auto it = pool_free.find(bytes);
    if (it != pool_free.end() && !it->second.empty()) { double* p = it->second.back(); it->second.pop_back(); touch(p); return p; }
    void* p = nullptr;
    MVUS_HIP(hipMalloc(&p, bytes));
    pool_size[static_cast<double*>(p)] = bytes;
    return static_cast<double*>(p);
  }
  // stream-ordered reuse: one stream per handle.  A buffer that leaves or re-enters the pool no longer names an x whose camera
  // states are cached (not every writer of an n-vector calls touch(): J^T u, the Schur kernels); the invariant is kept HERE
  void release(double* p) { if (p) { touch(p); pool_free[pool_size[p]].push_back(p); } }
  // decoded camera states are reused while the x buffer they came from is untouched (the accepted point of an LM iteration
  // is the trial point whose states are already there; a 2-evaluation solve evaluates and linearises at the same x)
  void touch(const double* d) { if (d == cams_for) cams_for = nullptr; if (d == rspan_for) rspan_for = nullptr; }
  void ensure_cams(const double* x) {
    if (cams_for == x) return;
    hipLaunchKernelGGL(k_cam_states, dim3((hp.C + 63) / 64), dim3(64), 0, stream, dp, x, cams);
    cams_for = x;
  }
  int stage_slot() {                          // next pinned slot, free again once the copy that last used it has run
    const int slot = stage_next;
    stage_next = (stage_next + 1) % kStageSlots;
    MVUS_HIP(hipEventSynchronize(stage_ev[slot]));
    return slot;
  }
  void upload(double* d, const double* s, int64_t len) {      // the host buffer may be reused right away
    touch(d);
    if (d == x_cur) held_analytic_at_xcur = false;
    if (len <= stage_cap) {
      const int slot = stage_slot();
      std::memcpy(stage[slot], s, len * sizeof(double));
      MVUS_HIP(hipMemcpyAsync(d, stage[slot], len * sizeof(double), hipMemcpyHostToDevice, stream));
      MVUS_HIP(hipEventRecord(stage_ev[slot], stream));
      return;
    }
    MVUS_HIP(hipMemcpyAsync(d, s, len * sizeof(double), hipMemcpyHostToDevice, stream));
    MVUS_HIP(hipStreamSynchronize(stream));
  }
  void download(double* d, const double* s, int64_t len) {
    if (len <= stage_cap) {
      const int slot = stage_slot();
      MVUS_HIP(hipMemcpyAsync(stage[slot], s, len * sizeof(double), hipMemcpyDeviceToHost, stream));
      MVUS_HIP(hipStreamSynchronize(stream));
      std::memcpy(d, stage[slot], len * sizeof(double));
      return;
    }
    MVUS_HIP(hipMemcpyAsync(d, s, len * sizeof(double), hipMemcpyDeviceToHost, stream));
    MVUS_HIP(hipStreamSynchronize(stream));
  }
  void copy(double* d, const double* s, int64_t len) { touch(d); if (d != s) MVUS_HIP(hipMemcpyAsync(d, s, len * sizeof(double), hipMemcpyDeviceToDevice, stream)); }
  void fill(double* d, double v, int64_t len) {
    touch(d);
    const int tiles = grid_for(len);
    if (len > 0) hipLaunchKernelGGL(k_fill, dim3(xcd_grid(tiles)), dim3(kThreads), 0, stream, (long long)len, v, d, tiles);
  }
  void axpby(int64_t len, double a, const double* x, double b, const double* y, double* out) {
    touch(out);
    if (len > 0) hipLaunchKernelGGL(k_axpby, dim3(grid_for(len)), dim3(kThreads), 0, stream, (long long)len, a, x, b, y, out);
  }
  void mul(int64_t len, const double* x, const double* y, double* out) {
    touch(out);
    if (len > 0) hipLaunchKernelGGL(k_mul, dim3(grid_for(len)), dim3(kThreads), 0, stream, (long long)len, x, y, out);
  }
  // two launches (partials, then their sum): a single launch with a last-workgroup ticket was measured at 22.9 us against
  // 2 x 4.9 us -- the agent-scope release fence every workgroup needs before its ticket costs more than a launch
  void dot_into(const double* a, const double* b, int64_t len, double* out) {
    const int nb = grid_for(len);
    if (len > 0) {
      hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(kThreads), 0, stream, (long long)len, a, b, partials);
      hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kThreads), 0, stream, nb, partials, out);
    } else {
      MVUS_HIP(hipMemsetAsync(out, 0, sizeof(double), stream));
    }
  }
  // Where the scalars the host reads after every step are written.  One rank: straight into the mapped pinned mirror (the host
  // reads it after a stream synchronisation: no device-to-host copy kernel, ~4 us + a launch per fetch).  With an all-reduce
  // callback the scalars are summed over the ranks in device memory first, so they stay there and are copied.
  bool scal_direct() const { return scal_map != nullptr && !allreduce; }
  double* scal_out() { return scal_direct() ? scal_map : scal_dev; }
  void dot_to_slot(const double* a, const double* b, int64_t len, int slot) { dot_into(a, b, len, scal_out() + slot); }
  double read_slot(int slot) {
    if (!scal_direct()) MVUS_HIP(hipMemcpyAsync(scal_host + slot, scal_dev + slot, sizeof(double), hipMemcpyDeviceToHost, stream));
    MVUS_HIP(hipStreamSynchronize(stream));
    return scal_host[slot];
  }
  void reduce(double* buf, size_t count) {
    if (!allreduce) return;
    RoctxRange range("mvus all-reduce");
    if (allreduce(allreduce_user, buf, count, stream) != 0) throw HipError{"all-reduce callback failed", MVUS_E_COMM};
  }
  // device-resident LM driver (ba_schur.h)
  std::vector<double> lb_host, ub_host, lb_fixed, ub_fixed;
  bool fixed_bounds_on_device = false;
  double *lb_dev = nullptr, *ub_dev = nullptr;
  double* tr_pn2 = nullptr;         // |p|^2 of the damped step (trust region of the LM driver)
  bool reshard_flag = false;        // set by HipSchur::solve_ok when a row left the time slice
  bool reshard_pending() { const bool r = reshard_flag; reshard_flag = false; return r; }
  // current / trial point of the LM driver, kept from solve to solve (ba_schur.h: lm_resume)
  double* lm_x[2] = {nullptr, nullptr};
  std::vector<double> lm_last_host;
  const double* lm_last_ptr = nullptr;   // host copy of the point the last solve returned: the pinned mirror the trial kernel wrote (no copy), else lm_last_host
  int lm_last = -1;
  double* lm_xbuf(int k) { if (!lm_x[k]) lm_x[k] = dalloc<double>(std::max<int64_t>(hp.n, 1)); return lm_x[k]; }
  // Consulting the remembered point DISARMS it: from here on the solve writes into both buffers (upload, trial points), so only a solve
  // that reaches lm_remember -- a normal or a reshard exit -- re-arms it.  A solve that leaves early (non-finite cost at x0, an exception)
  // therefore cannot leave a stale "buffer k holds the point I returned" behind for a caller that retries from the last good point.
  int lm_resume(const double* x_host) {
    int r = -1;
    if (lm_last >= 0 && lm_last_ptr != nullptr && !allreduce && std::memcmp(lm_last_ptr, x_host, sizeof(double) * hp.n) == 0) r = lm_last;
    lm_last = -1; lm_last_ptr = nullptr;
    return r;
  }
  // mirror: the pinned host buffer that holds the returned point already (written by the accepted trial's kernel).  Every solve starts its
  // trials at mirror 0 again: the comparison in lm_resume happens before any trial of that solve is launched, and a resumed solve that
  // accepts no step re-remembers through the caller's x (mirror == nullptr) -- else the point is copied
  void lm_remember(const double* x_dev, const double* x_host, const double* mirror) {
    lm_last = x_dev == lm_x[0] ? 0 : (x_dev == lm_x[1] ? 1 : -1);
    lm_last_ptr = nullptr;
    if (lm_last < 0) return;
    if (mirror != nullptr) lm_last_ptr = mirror;
    else { lm_last_host.assign(x_host, x_host + hp.n); lm_last_ptr = lm_last_host.data(); }
  }
  // What the last LM solve left for the point it returned (ba_schur.h: LmCarry).  Valid for the NEXT call on the handle only: every entry
  // point bumps api_seq (guarded), so any call in between -- a residual, a pattern, an outlier removal -- disarms it; consulting it
  // disarms it as well (a solve that fails early must not leave it behind).
  uint64_t api_seq = 0, carry_seq = 0;
  LmCarry carry{};
  int carry_jac_mode = -1;
  LmCarry lm_carry(int jac_mode) {
    LmCarry c{};
    if (carry.f_valid && carry_seq + 1 == api_seq && !allreduce && !std::getenv("MVUS_LM_NO_CARRY")) {
      c = carry;
      if (carry_jac_mode != jac_mode) c.lin_valid = false;
    }
    carry = LmCarry{};
    return c;
  }
  void lm_keep(const LmCarry& c, int jac_mode) { carry = c; carry_seq = api_seq; carry_jac_mode = jac_mode; }
  // fetch_mark: the next fetch waits for the work enqueued so far only (an event), not for what is enqueued after the mark -- the
  // speculative linearisation of the LM driver.  Needs the scalars written straight into mapped memory (one rank).
  // fetch_poll_begin: the same without an event (hipEventRecord costs a 6 us bubble between the two kernels it separates): the FIRST
  // kernel enqueued after the point writes *mark = value when it starts -- it cannot start before everything in front of it has
  // finished and released its writes -- and the fetch spins on that word in mapped memory.
  static constexpr int kMarkSlot = 24;
  static constexpr int kFailSumSlot = 16;  // [16], [17]: a time shard's failure flags summed over the ranks (= lm_scalars() + 8: fetched with the trial's scalars)
  double fetch_seq = 0.0;
  bool fetch_polled = false;
  bool fetch_poll_begin(double** mark, double* value) {
    if (scal_map == nullptr || std::getenv("MVUS_FETCH_EVENT")) return false;      // (the mark itself is always written through mapped memory)
    fetch_seq += 1.0;
    *mark = scal_map + kMarkSlot; *value = fetch_seq;
    fetch_polled = true;
    return true;
  }
  // sharded handles (scalars in device memory, copied out): the copy a later fetch() reads is enqueued NOW, in front of speculative work
  bool fetch_queued = false;
  int64_t fq_off = 0;
  int fq_k = 0;
  void fetch_enqueue(const double* src, int k) {
    if (scal_direct()) return;
    fq_off = src - scal_out(); fq_k = k;
    MVUS_HIP(hipMemcpyAsync(scal_host + fq_off, src, sizeof(double) * k, hipMemcpyDeviceToHost, stream));
    fetch_queued = true;
  }
  bool spec_on_shards() const { return scal_map != nullptr && std::getenv("MVUS_NO_SPEC_SHARDS") == nullptr; }
  hipEvent_t fetch_ev = nullptr;
  bool fetch_marked = false;
  void fetch_mark() {
    if (!fetch_ev) MVUS_HIP(hipEventCreateWithFlags(&fetch_ev, hipEventDisableTiming));
    MVUS_HIP(hipEventRecord(fetch_ev, stream));
    fetch_marked = true;
  }
  unsigned* lm_counter = nullptr;   // ticket counter of the last-block reductions (k_lm_gnorm / k_lm_trial), kept at 0 between launches
  void set_bounds(const std::vector<double>& lb, const std::vector<double>& ub) {
    if (!lb_dev) {
      lb_dev = dalloc<double>(hp.n); ub_dev = dalloc<double>(hp.n); lm_counter = dalloc<unsigned>(1);
      MVUS_HIP(hipMemsetAsync(lm_counter, 0, sizeof(unsigned), stream));
    }
    if (&lb == &lb_fixed && &ub == &ub_fixed && fixed_bounds_on_device) return;      // (the handle's own vectors, uploaded before)
    if (lb != lb_host) { lb_host = lb; upload(lb_dev, lb_host.data(), hp.n); }
    if (ub != ub_host) { ub_host = ub; upload(ub_dev, ub_host.data(), hp.n); }
    fixed_bounds_on_device = &lb == &lb_fixed && &ub == &ub_fixed;
  }
  const double* lb_ptr() const { return lb_dev; }
  const double* ub_ptr() const { return ub_dev; }
  double* lm_scalars() { return scal_out() + 8; }
  void dot_m_into(const double* a, const double* b, double* out) {
    dot_into(a, b, hp.m, out);
    reduce(out, 1);
  }
  // the two vector reductions of an LM iteration: one workgroup up to 128k parameters (no cross-workgroup fences), else one per 1024
  unsigned lm_grid() const { return hp.n <= (1 << 17) ? 1u : (unsigned)std::min<int64_t>(2048 / 5, (hp.n + 1023) / 1024); }
  void lm_gnorm(const double* x, const double* lb, const double* ub, const double* g, double* out) {
    hipLaunchKernelGGL(k_lm_gnorm, dim3(lm_grid()), dim3(1024), 0, stream, (int)hp.n, x, lb, ub, g, out, partials, lm_counter);
  }
  // Two mapped pinned mirrors of the trial point (ping-pong with the accepted / trial roles of the LM driver): the trial kernel
  // writes x_new there as well, so that the host holds the accepted point after the fetch that decides on it and the solve ends
  // without a device-to-host copy and its synchronisation.  One rank only (like the mapped scalars); else nullptr -> download.
  static constexpr bool kSwapResiduals = true;          // f at an accepted point: the trial buffer changes roles with f_cur, no copy
  double* xmir_host[2] = {nullptr, nullptr};
  double* xmir_dev[2] = {nullptr, nullptr};
  int64_t xmir_cap = 0;
  double* mirror_dev(int k) {
    if (!scal_direct()) return nullptr;
    if (xmir_cap < hp.n) {
      for (int i = 0; i < 2; ++i) {
        if (xmir_host[i]) { (void)hipHostFree(xmir_host[i]); xmir_host[i] = xmir_dev[i] = nullptr; }
        MVUS_HIP(hipHostMalloc(reinterpret_cast<void**>(&xmir_host[i]), std::max<int64_t>(hp.n, 1) * sizeof(double), hipHostMallocMapped));
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&xmir_dev[i]), xmir_host[i], 0) != hipSuccess) xmir_dev[i] = nullptr;
      }
      xmir_cap = hp.n;
    }
    return xmir_dev[k];
  }
  // (the mirror of slot k holds the trial point only while the trial kernel writes it: not once an all-reduce route was installed on
  // a handle that had solved on one rank before -- the caller then downloads x)
  const double* mirror_host(int k) const { return scal_direct() ? xmir_host[k] : nullptr; }
  void adopt_residual(double*& f_dev, double*& f_new) {       // both are pool buffers of one size class
    if (f_dev == f_cur) f_cur = f_new;
    std::swap(f_dev, f_new);
  }
  void lm_trial(const double* x, const double* p, const double* lb, const double* ub, const double* g, const double* D,
                const int* fail, double* x_new, double* out, double* gnorm_out, double* x_mirror, double* pn2 = nullptr, double delta = 0.0,
                const double* fail_sum = nullptr) {
    touch(x_new);
    double* const fail_out = scal_out() + kFailSumSlot;
    // trust region: |p|^2 first (two small launches), the trial kernel then cuts the step back to delta along its direction
    double* pn2_dev = nullptr;
    if (pn2) { if (!tr_pn2) tr_pn2 = dalloc<double>(1); pn2_dev = tr_pn2; dot_into(p, p, hp.n, pn2_dev); }
    // one workgroup is limited by what one CU can load (six n-vectors: 18 us at n = 15k); a few workgroups and a second, tiny
    // launch for their partials take 10 us.  Beyond 128k parameters: the one-launch form with the last-workgroup hand-over.
    const unsigned g2 = hp.n > 2048 && hp.n <= (1 << 17) ? (unsigned)std::min<int64_t>(32, (hp.n + 1023) / 1024) : 0u;
    if (g2 > 1) {
      hipLaunchKernelGGL(k_lm_trial, dim3(g2), dim3(1024), 0, stream, (int)hp.n, x, p, lb, ub, g, D, fail, x_new, out, gnorm_out, partials, (unsigned*)nullptr, x_mirror, (const double*)pn2_dev, delta, fail_sum, fail_out);
      hipLaunchKernelGGL(k_lm_trial_sum, dim3(1), dim3(64), 0, stream, (int)g2, partials, out, gnorm_out, dp, (const double*)x_new, cams);
      cams_for = x_new;                      // decoded by that launch: the trial residual needs no k_cam_states
      return;
    }
    hipLaunchKernelGGL(k_lm_trial, dim3(lm_grid()), dim3(1024), 0, stream, (int)hp.n, x, p, lb, ub, g, D, fail, x_new, out, gnorm_out, partials, lm_counter, x_mirror, (const double*)pn2_dev, delta, fail_sum, fail_out);
  }
  // (spinning on a sentinel in the mapped scalars instead of hipStreamSynchronize was measured in round 5: 0.509 against 0.509 ms per
  // step, three A/B pairs on one box -- the runtime's own wait already spins; not kept)
  void fetch(const double* src, int k, double* host) {       // src inside scal_out(): the pinned mirror itself, or staged through it
    const int64_t off = src - scal_out();
    const bool queued = fetch_queued && off >= fq_off && off + k <= fq_off + fq_k;
    if (!scal_direct() && !queued) MVUS_HIP(hipMemcpyAsync(scal_host + off, src, sizeof(double) * k, hipMemcpyDeviceToHost, stream));
    fetch_queued = false;
    if (fetch_polled && (scal_direct() || queued)) {
      const volatile double* m = scal_host + kMarkSlot;
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (*m != fetch_seq) {
        if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { MVUS_HIP(hipStreamSynchronize(stream)); break; }   // (never seen; the stream then tells)
        __builtin_ia32_pause();
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    } else if (fetch_marked && (scal_direct() || queued)) MVUS_HIP(hipEventSynchronize(fetch_ev));
    else MVUS_HIP(hipStreamSynchronize(stream));
    fetch_marked = false; fetch_polled = false;
    for (SqPending& sp : sq_pend)                         // sums of squares left as per-workgroup partials in mapped memory (residual_sq)
      if (sp.slot >= off && sp.slot < off + k) { scal_host[sp.slot] = sq_host_sum(sqh_host[&sp - sq_pend], sp.n); sp.slot = -1; }
    for (int i = 0; i < k; ++i) host[i] = scal_host[off + i];
  }
  double dot_n(const double* a, const double* b, int64_t len) { dot_to_slot(a, b, len, 0); return read_slot(0); }
  double dot_m(const double* a, const double* b) { dot_to_slot(a, b, hp.m, 1); reduce(scal_out() + 1, 1); return read_slot(1); }

  // the slot Jacobian (2*NS*M doubles) is allocated on first use: residual-only handles (Scene.error_cam, outlier masks) never pay for it
  void ensure_J() { if (!J) J = dalloc<double>(j_doubles(hp.NS, dp.n_chunks)); }      // sized at first use; outlier removal only ever shrinks the chunk table
  void eval(const double* x, double* f, bool jac, int jac_mode) {
    RoctxRange range(jac ? "mvus residual+jacobian" : "mvus residual");
    if (jac) ensure_J();
    const bool masked = jac && jac_mode == MVUS_JAC_PATTERN;
    if (masked && !has_pattern) throw HipError{"MVUS_JAC_PATTERN needs mvus_ba_set_pattern (or solve) first"};
    ensure_cams(x);
    if (dp.n_chunks > 0) {
      const dim3 g(dp.n_chunks), gj(xcd_grid(dp.n_chunks)), b(kThreads);
      if (hp.calib) {
        if (jac) hipLaunchKernelGGL((k_observations<true, true>), gj, b, 0, stream, dp, cams, x, f, J, span, pat0, (int)masked);
        else hipLaunchKernelGGL((k_observations<true, false>), g, b, 0, stream, dp, cams, x, f, J, rspan, pat0, 0);
      } else {
        if (jac) hipLaunchKernelGGL((k_observations<false, true>), gj, b, 0, stream, dp, cams, x, f, J, span, pat0, (int)masked);
        else hipLaunchKernelGGL((k_observations<false, false>), g, b, 0, stream, dp, cams, x, f, J, rspan, pat0, 0);
      }
      if (!jac) rspan_for = x;
    }
    if (hp.T > 0) {
      if (is_root || tshard.on) {
        const dim3 g((hp.T + kThreads - 1) / kThreads), b(kThreads);
        if (jac) hipLaunchKernelGGL(k_motion<true>, g, b, 0, stream, dp, x, f + 2 * hp.M, mJ, mctrl, (int)masked);
        else hipLaunchKernelGGL(k_motion<false>, g, b, 0, stream, dp, x, f + 2 * hp.M, mJ, mctrl, 0);
      } else {
        // sharded run: the (replicated) motion rows are owned by the root rank; here they are rows of zeros
        // (mctrl stays -1 from initialisation, so they contribute nothing to J v, J^T u or the normal equations)
        MVUS_HIP(hipMemsetAsync(f + 2 * hp.M, 0, sizeof(double) * hp.T, stream));
      }
    }
    MVUS_HIP(hipGetLastError());
    if (jac) has_jacobian = true;
  }
  void residual(const double* x, double* f) { eval(x, f, false, 0); }
  // f = f(x) and out = |f|^2 (summed over the ranks): on one rank the residual kernels leave their workgroups' sums of squares
  // behind and one small launch adds them -- no separate pass over f
  double* sq_part = nullptr;
  size_t sq_cap = 0;
  // One rank (scalars in mapped memory): the workgroups' partial sums go to mapped pinned HOST memory and the host adds them after
  // the fetch's wait, in k_dot_final's own tree (sq_host_sum: the same bits) -- the 4.4 - 5 us launch of k_dot_final behind every
  // residual evaluation of the LM driver is gone.  Two sets: the start's |f|^2 and the first trial's are both outstanding at the
  // first fetch of a solve.  MVUS_SQ_DEVICE_SUM=1 keeps the launch (A/B).
  double* sqh_host[2] = {nullptr, nullptr};
  double* sqh_dev[2] = {nullptr, nullptr};
  size_t sqh_cap = 0;
  struct SqPending { int64_t slot = -1; int n = 0; } sq_pend[2];
  static double sq_host_sum(const double* part, int nb) {
    double red[kThreads / 64];
    for (int w = 0; w < kThreads / 64; ++w) {
      double lane[64];
      for (int l = 0; l < 64; ++l) {                       // k_dot_final: thread t adds partials t, t + kThreads, ...
        double a = 0.0;
        for (int i = w * 64 + l; i < nb; i += kThreads) a += part[i];
        lane[l] = a;
      }
      for (int off = 32; off > 0; off >>= 1)                // wave_sum as lane 0 sees it
        for (int l = 0; l < off; ++l) lane[l] += lane[l + off];
      red[w] = lane[0];
    }
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += red[w];
    return t;
  }
  // clr / clr_len: storage to zero beside the evaluation (HipSchur's normal-equation blocks); returns false if it was not done
  bool residual_sq(const double* x, double* f, double* out, double* clr = nullptr, int64_t clr_len = 0) {
    // (observation shards off the root rank: the replicated motion rows are rows of zeros there -- the general path)
    if (allreduce && hp.T > 0 && !(is_root || tshard.on)) { residual(x, f); dot_m_into(f, f, out); return false; }
    RoctxRange range("mvus residual");
    const int mb = hp.T > 0 ? (int)((hp.T + kThreads - 1) / kThreads) : 0;
    const size_t need = (size_t)dp.n_chunks + mb + 1;
    const bool host_sum = scal_direct() && out >= scal_out() && out < scal_out() + 16 && !std::getenv("MVUS_SQ_DEVICE_SUM");
    const int set = (out == lm_scalars()) ? 0 : 1;
    double* sq_part = this->sq_part;
    if (host_sum) {
      if (need > sqh_cap) {
        for (int i = 0; i < 2; ++i) {
          if (sqh_host[i]) (void)hipHostFree(sqh_host[i]);
          MVUS_HIP(hipHostMalloc(reinterpret_cast<void**>(&sqh_host[i]), need * sizeof(double), hipHostMallocMapped));
          MVUS_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&sqh_dev[i]), sqh_host[i], 0));
        }
        sqh_cap = need;
      }
      sq_part = sqh_dev[set];
    } else if (need > sq_cap) { this->sq_part = dalloc<double>(need); sq_cap = need; sq_part = this->sq_part; }
    ensure_cams(x);
    bool cleared = false;
    if (dp.n_chunks > 0) {
      const int fb = (clr && clr_len > 0) ? (int)std::min<int64_t>(2048, (clr_len + kThreads - 1) / kThreads) : 0;
      const dim3 g(dp.n_chunks + fb), b(kThreads);
      double* c = fb > 0 ? clr : (double*)nullptr;
      if (hp.calib) hipLaunchKernelGGL((k_observations<true, false>), g, b, 0, stream, dp, cams, x, f, J, rspan, pat0, 0, sq_part, c, (long long)clr_len);
      else hipLaunchKernelGGL((k_observations<false, false>), g, b, 0, stream, dp, cams, x, f, J, rspan, pat0, 0, sq_part, c, (long long)clr_len);
      rspan_for = x;
      cleared = fb > 0;
    }
    if (mb > 0) hipLaunchKernelGGL(k_motion<false>, dim3(mb), dim3(kThreads), 0, stream, dp, x, f + 2 * hp.M, mJ, mctrl, 0, sq_part + dp.n_chunks);
    if (host_sum) { sq_pend[set].slot = out - scal_out(); sq_pend[set].n = dp.n_chunks + mb; }      // added up by fetch()
    else if (dp.n_chunks + mb > 0) hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kThreads), 0, stream, dp.n_chunks + mb, sq_part, out);
    else MVUS_HIP(hipMemsetAsync(out, 0, sizeof(double), stream));
    reduce(out, 1);                  // (sharded: the workgroups' partial sums replace a pass of k_dot_partial over f; the ranks' sums are added here)
    MVUS_HIP(hipGetLastError());
    return cleared;
  }
  // motion rows only: their residuals and Jacobian blocks (the fused LM path evaluates the detection rows elsewhere)
  void motion_jacobian(const double* x, double* f) {
    if (hp.T <= 0) return;
    if (is_root || tshard.on) hipLaunchKernelGGL(k_motion<true>, dim3((hp.T + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, dp, x, f + 2 * hp.M, mJ, mctrl, 0);
    else MVUS_HIP(hipMemsetAsync(f + 2 * hp.M, 0, sizeof(double) * hp.T, stream));
    MVUS_HIP(hipGetLastError());
  }
  void jacobian(const double* x, double* f, int jac_mode) {
    if (jac_mode == MVUS_JAC_FD) jacobian_fd(x, f); else eval(x, f, true, jac_mode);
  }

  void set_fd_groups(const int32_t* groups, int ngroups) {
    if (!fd_groups) { fd_groups = dalloc<int32_t>(hp.n); fd_h = dalloc<double>(hp.n); fd_dx = dalloc<double>(hp.n); fd_xg = dalloc<double>(hp.n); }
    MVUS_HIP(hipMemcpyAsync(fd_groups, groups, sizeof(int32_t) * hp.n, hipMemcpyHostToDevice, stream));
    MVUS_HIP(hipStreamSynchronize(stream));
    if (ngroups > fd_ngroups || !fd_F) { fd_F = dalloc<double>((size_t)ngroups * hp.m); }
    fd_ngroups = ngroups;
  }
  // scipy's approx_derivative(..., method='2-point', sparsity=(pattern, groups)): one residual per column group
  void jacobian_fd(const double* x, double* f) {
    if (!has_pattern || fd_ngroups <= 0) throw HipError{"MVUS_JAC_FD needs mvus_ba_set_pattern and mvus_ba_set_fd_groups first"};
    const int n = (int)hp.n;
    ensure_J();
    eval(x, f, false, 0);
    hipLaunchKernelGGL(k_fd_steps, dim3((n + 255) / 256), dim3(256), 0, stream, n, hp.C, hp.rs_bounds, x, fd_h, fd_dx);
    for (int g = 0; g < fd_ngroups; ++g) {
      touch(fd_xg);
      hipLaunchKernelGGL(k_fd_perturb, dim3((n + 255) / 256), dim3(256), 0, stream, n, g, x, fd_h, fd_groups, fd_xg);
      eval(fd_xg, fd_F + (size_t)g * hp.m, false, 0);
    }
    if (dp.n_chunks > 0) {
      if (hp.calib) hipLaunchKernelGGL(k_fd_fill<30>, dim3(dp.n_chunks), dim3(kThreads), 0, stream, dp, (long long)hp.m, f, fd_F, fd_dx, fd_groups, pat0, J, span);
      else hipLaunchKernelGGL(k_fd_fill<21>, dim3(dp.n_chunks), dim3(kThreads), 0, stream, dp, (long long)hp.m, f, fd_F, fd_dx, fd_groups, pat0, J, span);
    }
    if (hp.T > 0) {
      if (is_root) hipLaunchKernelGGL(k_fd_fill_motion, dim3((hp.T + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, dp, (long long)hp.m, f, fd_F, fd_dx, fd_groups, mJ, mctrl);
    }
    MVUS_HIP(hipGetLastError());
    has_jacobian = true;
  }

  // Scene.remove_outliers on the resident data: mask at x, stable compaction, new launch tables.
  void remove_outliers(const double* x_dev, double thres, uint8_t* keep_out, int64_t* det_off_out) {
    residual(x_dev, f_cur);
    const int nc = dp.n_chunks;
    uint8_t* keep = nullptr;
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&keep), std::max<int64_t>(hp.M, 1)));
    int32_t* counts = nullptr;
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&counts), sizeof(int32_t) * std::max(nc, 1)));
    std::vector<int32_t> cnt_h(nc);
    std::vector<long long> off_h(nc);
    long long* off_d = nullptr;
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&off_d), sizeof(long long) * std::max(nc, 1)));
    auto cleanup = [&]() { (void)hipFree(keep); (void)hipFree(counts); (void)hipFree(off_d); };
    try {
      if (nc > 0) {
        hipLaunchKernelGGL(k_outlier_mask, dim3(nc), dim3(kThreads), 0, stream, dp, f_cur, thres, keep);
        hipLaunchKernelGGL(k_compact_count, dim3(nc), dim3(kThreads), 0, stream, dp, keep, counts);
        MVUS_HIP(hipMemcpyAsync(cnt_h.data(), counts, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, stream));
      }
      if (keep_out && hp.M > 0) MVUS_HIP(hipMemcpyAsync(keep_out, keep, hp.M, hipMemcpyDeviceToHost, stream));
      MVUS_HIP(hipStreamSynchronize(stream));
      std::vector<int64_t> new_off(hp.C + 1, 0);
      long long run = 0;
      for (int k = 0; k < nc; ++k) { off_h[k] = run; run += cnt_h[k]; new_off[hp.chunk_cam[k] + 1] += cnt_h[k]; }
      for (int c = 0; c < hp.C; ++c) new_off[c + 1] += new_off[c];
      const int64_t newM = run;
      // the compacted copy goes into a second set of detection arrays (allocated once, sized for the original problem);
      // the two sets swap roles on every call
      if (!det_alt[0]) for (double*& p : det_alt) p = dalloc<double>(det_capacity);
      double *nf = det_alt[0], *nu = det_alt[1], *nv = det_alt[2], *nuo = det_alt[3], *nvo = det_alt[4];
      det_alt[0] = const_cast<double*>(dp.frame); det_alt[1] = const_cast<double*>(dp.u_raw); det_alt[2] = const_cast<double*>(dp.v_raw);
      det_alt[3] = const_cast<double*>(dp.u_obs); det_alt[4] = const_cast<double*>(dp.v_obs);
      if (nc > 0) {
        MVUS_HIP(hipMemcpyAsync(off_d, off_h.data(), sizeof(long long) * nc, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_compact_scatter, dim3(nc), dim3(kThreads), 0, stream, dp, keep, off_d, nf, nu, nv, nuo, nvo);
        MVUS_HIP(hipGetLastError());
      }
      MVUS_HIP(hipStreamSynchronize(stream));
      // the handle now describes the filtered problem
      hp.det_off = new_off; hp.M = newM; hp.m = 2 * newM + hp.T;
      hp.frame.clear(); hp.u_raw.clear(); hp.v_raw.clear();          // host copies are no longer current
      hp.build_chunks();
      dp.frame = nf; dp.u_raw = nu; dp.v_raw = nv; dp.u_obs = nuo; dp.v_obs = nvo;
      dp.M = newM;
      // the launch tables shrink (never more chunks per camera than before): rewritten in place
      std::vector<long long> cs(hp.chunk_start.begin(), hp.chunk_start.end()), doff(hp.det_off.begin(), hp.det_off.end());
      const size_t nck = hp.chunk_cam.size();
      if (nck > 0) {
        MVUS_HIP(hipMemcpyAsync(const_cast<int32_t*>(dp.chunk_cam), hp.chunk_cam.data(), nck * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        MVUS_HIP(hipMemcpyAsync(const_cast<int32_t*>(dp.chunk_count), hp.chunk_count.data(), nck * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        MVUS_HIP(hipMemcpyAsync(const_cast<long long*>(dp.chunk_start), cs.data(), nck * sizeof(long long), hipMemcpyHostToDevice, stream));
        MVUS_HIP(hipMemcpyAsync(const_cast<ChunkInfo*>(dp.chunks), hp.chunks.data(), nck * sizeof(ChunkInfo), hipMemcpyHostToDevice, stream));
      }
      MVUS_HIP(hipMemcpyAsync(const_cast<long long*>(dp.det_off), doff.data(), doff.size() * sizeof(long long), hipMemcpyHostToDevice, stream));
      MVUS_HIP(hipMemcpyAsync(const_cast<int32_t*>(dp.cam_chunk_off), hp.cam_chunk_off.data(), hp.cam_chunk_off.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
      dp.n_chunks = (int)nck;
      MVUS_HIP(hipStreamSynchronize(stream));
      has_pattern = false; has_jacobian = false; fd_ngroups = 0; rspan_for = nullptr;
      if (pattern_uploaded && hp.T > 0) MVUS_HIP(hipMemcpyAsync(ms_pat_dev, hp.ms_pat.data(), sizeof(int32_t) * hp.T, hipMemcpyHostToDevice, stream));
      pattern_uploaded = false;
      m_glob = hp.m;
      if (det_off_out) std::memcpy(det_off_out, new_off.data(), sizeof(int64_t) * (hp.C + 1));
    } catch (...) { cleanup(); throw; }
    cleanup();
  }

  void set_pattern(const double* x0_dev) {
    ensure_cams(x0_dev);
    if (dp.n_chunks > 0) hipLaunchKernelGGL(k_pattern, dim3(dp.n_chunks), dim3(kThreads), 0, stream, dp, cams, pat0);
    MVUS_HIP(hipGetLastError());
    if (pattern_uploaded && hp.T > 0)      // back to the canonical motion-row codes
      MVUS_HIP(hipMemcpyAsync(ms_pat_dev, hp.ms_pat.data(), sizeof(int32_t) * hp.T, hipMemcpyHostToDevice, stream));
    has_pattern = true; pattern_uploaded = false;
  }
  // every point of a code must be a control point of ONE spline
  bool code_ok(int32_t code) const {
    if (code < 0) return code == -1;
    const int p = pattern_index(code), mk = pattern_mask(code);
    if (mk == 0 || p >= hp.N) return false;
    int top = 3; while (!((mk >> top) & 1)) --top;
    return p + top < hp.N && hp.ctrl_x0[p + top] == hp.ctrl_x0[p] + top;
  }
  std::string upload_pattern(const int32_t* pat, const int32_t* mpat) {
    for (int64_t i = 0; i < hp.M; ++i) if (!code_ok(pat[i])) return "upload_pattern: bad code in detection row " + std::to_string(i);
    if (hp.T > 0 && mpat) for (int j = 0; j < hp.T; ++j) if (mpat[j] < 0 || !code_ok(mpat[j])) return "upload_pattern: bad code in motion row " + std::to_string(j);
    if (hp.M > 0) MVUS_HIP(hipMemcpyAsync(pat0, pat, sizeof(int32_t) * hp.M, hipMemcpyHostToDevice, stream));
    if (hp.T > 0 && mpat) MVUS_HIP(hipMemcpyAsync(ms_pat_dev, mpat, sizeof(int32_t) * hp.T, hipMemcpyHostToDevice, stream));
    MVUS_HIP(hipStreamSynchronize(stream));
    has_pattern = true; pattern_uploaded = true; has_jacobian = false;
    return "";
  }

  void jv(const double* v, double* y) {
    const bool fused = dp.n_chunks > 0 && hp.T > 0;        // the motion rows in extra workgroups of k_jv
    if (dp.n_chunks > 0) {
      const unsigned grid = (unsigned)(xcd_grid(dp.n_chunks) + (fused ? (hp.T + kThreads - 1) / kThreads : 0));
      const double* mj = fused ? mJ : nullptr;
      if (hp.calib) hipLaunchKernelGGL(k_jv<30>, dim3(grid), dim3(kThreads), 0, stream, dp, J, span, v, y, mj, mctrl, y + 2 * hp.M);
      else hipLaunchKernelGGL(k_jv<21>, dim3(grid), dim3(kThreads), 0, stream, dp, J, span, v, y, mj, mctrl, y + 2 * hp.M);
    }
    if (hp.T > 0 && !fused) hipLaunchKernelGGL(k_motion_jv, dim3((hp.T + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, dp, mJ, mctrl, v, y + 2 * hp.M);
    MVUS_HIP(hipGetLastError());
  }
  // z = J^T u of this rank's rows: deterministic two-pass form (k_jtu_partial / k_jtu_reduce)
  double *zc = nullptr, *zs = nullptr;
  int32_t *zg0 = nullptr, *zfill = nullptr, *zext = nullptr;
  int* jt_bounds = nullptr;               // [2] measured by k_jtu_index: how far a chunk starts below the running maximum, how far a chunk reaches
  int* jt_nondet = nullptr;
  // z_is_zero: the caller guarantees z == 0 on entry (the device-resident LSMR loop clears it while consuming it);
  // reuse_index: zfill of the previous call is still valid (same Jacobian, hence the same window starts)
  int32_t* jt_first = nullptr;            // [N][C] first covering chunk per (control point, camera): valid while reuse_index calls follow one another
  void jtu_buffers() {
    if (!zc) {
      const size_t nc = std::max<size_t>(hp.chunks.size(), 1);
      zc = dalloc<double>(nc * (size_t)(hp.NS - 12)); zs = dalloc<double>(nc * 3 * (size_t)kJtWin); zg0 = dalloc<int32_t>(nc); zfill = dalloc<int32_t>(nc);
      zext = dalloc<int32_t>(nc); jt_bounds = dalloc<int>(2);
      jt_nondet = dalloc<int>(1);
      MVUS_HIP(hipMemsetAsync(jt_nondet, 0, sizeof(int), stream));
    }
  }
  void jtu_local(const double* u, double* z, bool z_is_zero = false, bool reuse_index = false, bool build_first = false) {
    jtu_buffers();
    if (!z_is_zero) MVUS_HIP(hipMemsetAsync(z, 0, sizeof(double) * hp.n, stream));
    if (!reuse_index) MVUS_HIP(hipMemsetAsync(jt_bounds, 0, 2 * sizeof(int), stream));
    const dim3 g2(hp.C + (hp.N + kThreads / 64 - 1) / (kThreads / 64)), b(kThreads);
    const int motion = hp.T > 0 ? 1 : 0;
    if (hp.calib) {
      if (dp.n_chunks > 0) hipLaunchKernelGGL(k_jtu_partial<30>, dim3(xcd_grid(dp.n_chunks)), b, 0, stream, dp, J, span, u, z, zc, zs, zg0, jt_nondet, zext);
      if (!reuse_index) hipLaunchKernelGGL(k_jtu_index, dim3(hp.C), dim3(64), 0, stream, dp, zg0, zfill, zext, jt_bounds);
      if (build_first) build_jt_first();
      hipLaunchKernelGGL(k_jtu_reduce<30>, g2, b, 0, stream, dp, zc, zs, zg0, zfill, mJ, mctrl, u + 2 * hp.M, motion, z, (reuse_index || build_first) ? jt_first : (int32_t*)nullptr, jt_bounds);
    } else {
      if (dp.n_chunks > 0) hipLaunchKernelGGL(k_jtu_partial<21>, dim3(xcd_grid(dp.n_chunks)), b, 0, stream, dp, J, span, u, z, zc, zs, zg0, jt_nondet, zext);
      if (!reuse_index) hipLaunchKernelGGL(k_jtu_index, dim3(hp.C), dim3(64), 0, stream, dp, zg0, zfill, zext, jt_bounds);
      if (build_first) build_jt_first();
      hipLaunchKernelGGL(k_jtu_reduce<21>, g2, b, 0, stream, dp, zc, zs, zg0, zfill, mJ, mctrl, u + 2 * hp.M, motion, z, (reuse_index || build_first) ? jt_first : (int32_t*)nullptr, jt_bounds);
    }
    MVUS_HIP(hipGetLastError());
  }
  void build_jt_first() {
    if (!jt_first) jt_first = dalloc<int32_t>((size_t)std::max<int64_t>(hp.N, 1) * hp.C);
    const long long e = (long long)hp.N * hp.C;
    if (e > 0) hipLaunchKernelGGL(k_jtu_first, dim3((unsigned)((e + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, dp, zfill, jt_first, jt_bounds);
  }
  void jtu(const double* u, double* z) { jtu_local(u, z); reduce(z, (size_t)hp.n); }

  // ---- device-resident LSMR iterations (Lsmr::run of ba_solver.h, unbounded case, one rank) ----
  static constexpr bool kDeviceLsmr = true;
  LsmrScalars* lsmr_state = nullptr;       // [2] device
  LsmrScalars* lsmr_host = nullptr;        // pinned
  double* lsmr_part = nullptr;             // [3][2048] partial sums (u.u, v.v, x.x) + beta
  double* lsmr_pu = nullptr;               // per-workgroup partials of |u'|^2 of the one-pass form (k_jvjtu)
  size_t lsmr_pu_cap = 0;
  bool lsmr_on_device() const { return !allreduce && std::getenv("MVUS_LSMR_HOST") == nullptr; }
  bool lsmr_scaled_on_device() const { return std::getenv("MVUS_LSMR_BOUNDED_HOST") == nullptr; }      // (A/B: the bounded problem on the host-driven loop, rounds 3-5)
  // D, E: the column scaling and the extra diagonal rows of the bounded problem (trf_bounds), device n-vectors or null; ub: the n extra rows of u
  void lsmr_iterations(LsmrScalars& sc, double* ut, double* tm, double* v, double* tn, double* h, double* hbar, double* x,
                       const double* D = nullptr, const double* E = nullptr, double* ub = nullptr) {
    if (!lsmr_state) {
      lsmr_state = dalloc<LsmrScalars>(2);
      lsmr_part = dalloc<double>(4 * 2048 + 8);      // u.u, v.v, x.x, (ub.ub) partials + beta
      MVUS_HIP(hipHostMalloc(reinterpret_cast<void**>(&lsmr_host), sizeof(LsmrScalars), hipHostMallocDefault));
    }
    const long long n = hp.n, m = hp.m;
    const int gm = grid_for(m), gn = grid_for(n);
    double *pu = lsmr_part, *pv = lsmr_part + 2048, *px = lsmr_part + 4096, *pub = lsmr_part + 6144, *beta_dev = lsmr_part + 8192;
    const bool scaled = D != nullptr || E != nullptr;
    double* dv = nullptr;                                   // D v (the scaled problem's argument of J)
    if (D) dv = alloc(n);
    *lsmr_host = sc;
    MVUS_HIP(hipMemcpyAsync(lsmr_state, lsmr_host, sizeof(LsmrScalars), hipMemcpyHostToDevice, stream));
    int cur = 0;
    const int batch = 8;
    long long launched = 0;
    MVUS_HIP(hipMemsetAsync(tn, 0, sizeof(double) * n, stream));        // J^T u adds into a zeroed vector; k_lsmr_v clears it again while reading it
    // MVUS_LSMR_ONE_PASS=1: one pass over J per iteration (k_jvjtu: u kept unnormalised, its norm in *ubeta).  Opt-in: the converged
    // answers stay inside the parity bars either way, but the unconverged 10-evaluation iterate of one fixture (dist_fixed_2cam) moves
    // outside the bars measured with the two-pass arithmetic
    const bool one_pass = dp.n_chunks > 0 && !scaled && std::getenv("MVUS_LSMR_ONE_PASS") != nullptr;
    const unsigned gj = (unsigned)(xcd_grid(dp.n_chunks) + (hp.T > 0 ? (hp.T + kThreads - 1) / kThreads : 0));
    double* ubeta = beta_dev + 1;
    if (one_pass) {
      if (lsmr_pu_cap < (size_t)gj) { lsmr_pu = dalloc<double>(gj); lsmr_pu_cap = gj; }
      jtu_buffers();
      const double one = 1.0;
      MVUS_HIP(hipMemcpyAsync(ubeta, &one, sizeof(double), hipMemcpyHostToDevice, stream));       // (ut arrives normalised)
      MVUS_HIP(hipMemsetAsync(jt_bounds, 0, 2 * sizeof(int), stream));
    }
    while (true) {
      for (int b = 0; b < batch && launched < sc.maxiter; ++b, ++launched) {
        const LsmrScalars* c = lsmr_state + cur;
        LsmrScalars* nx = lsmr_state + (cur ^ 1);
        if (one_pass) {
          const dim3 g2(hp.C + (hp.N + kThreads / 64 - 1) / (kThreads / 64)), bt(kThreads);
          const int motion = hp.T > 0 ? 1 : 0;
          if (hp.calib) hipLaunchKernelGGL(k_jvjtu<30>, dim3(gj), bt, 0, stream, dp, J, span, v, ut, c, ubeta, lsmr_pu, tn, zc, zs, zg0, jt_nondet, zext, mJ, mctrl);
          else hipLaunchKernelGGL(k_jvjtu<21>, dim3(gj), bt, 0, stream, dp, J, span, v, ut, c, ubeta, lsmr_pu, tn, zc, zs, zg0, jt_nondet, zext, mJ, mctrl);
          if (launched == 0) { hipLaunchKernelGGL(k_jtu_index, dim3(hp.C), dim3(64), 0, stream, dp, zg0, zfill, zext, jt_bounds); build_jt_first(); }
          if (hp.calib) hipLaunchKernelGGL(k_jtu_reduce<30>, g2, bt, 0, stream, dp, zc, zs, zg0, zfill, mJ, mctrl, ut + 2 * hp.M, motion, tn, jt_first, jt_bounds);
          else hipLaunchKernelGGL(k_jtu_reduce<21>, g2, bt, 0, stream, dp, zc, zs, zg0, zfill, mJ, mctrl, ut + 2 * hp.M, motion, tn, jt_first, jt_bounds);
          hipLaunchKernelGGL(k_lsmr_v1, dim3(gn), dim3(kThreads), 0, stream, n, tn, v, c, (int)gj, lsmr_pu, beta_dev, ubeta, pv);
        } else if (scaled) {
        // the bounded problem: A = [J diag(D); diag(E)] -- the host-driven loop's operations (Lsmr::run), launch for launch without its
        // three synchronisations per iteration
        if (D) { hipLaunchKernelGGL(k_lsmr_scale, dim3(gn), dim3(kThreads), 0, stream, n, D, (const double*)v, dv, c); jv(dv, tm); }
        else jv(v, tm);
        hipLaunchKernelGGL(k_lsmr_u, dim3(gm), dim3(kThreads), 0, stream, m, tm, ut, c, pu);
        if (E) {
          hipLaunchKernelGGL(k_lsmr_ub, dim3(gn), dim3(kThreads), 0, stream, n, E, (const double*)v, ub, c, pub);
          hipLaunchKernelGGL(k_lsmr_unorm2, dim3(gm), dim3(kThreads), 0, stream, m, ut, gm, pu, n, ub, gn, pub, c, beta_dev);
        } else hipLaunchKernelGGL(k_lsmr_unorm, dim3(gm), dim3(kThreads), 0, stream, m, ut, gm, pu, c, beta_dev);
        jtu_local(ut, tn, true, launched > 0, launched == 0);
        hipLaunchKernelGGL(k_lsmr_v_de, dim3(gn), dim3(kThreads), 0, stream, n, tn, v, D, E, (const double*)ub, c, beta_dev, pv);
        } else {
        jv(v, tm);
        hipLaunchKernelGGL(k_lsmr_u, dim3(gm), dim3(kThreads), 0, stream, m, tm, ut, c, pu);
        hipLaunchKernelGGL(k_lsmr_unorm, dim3(gm), dim3(kThreads), 0, stream, m, ut, gm, pu, c, beta_dev);
        jtu_local(ut, tn, true, launched > 0, launched == 0);
        hipLaunchKernelGGL(k_lsmr_v, dim3(gn), dim3(kThreads), 0, stream, n, tn, v, c, beta_dev, pv);
        }
        hipLaunchKernelGGL(k_lsmr_update, dim3(gn), dim3(kThreads), 0, stream, n, v, h, hbar, x, gn, pv, c, beta_dev, nx, px);
        hipLaunchKernelGGL(k_lsmr_test, dim3(1), dim3(kThreads), 0, stream, gn, px, c, nx);
        cur ^= 1;
      }
      MVUS_HIP(hipGetLastError());
      MVUS_HIP(hipMemcpyAsync(lsmr_host, lsmr_state + cur, sizeof(LsmrScalars), hipMemcpyDeviceToHost, stream));
      MVUS_HIP(hipStreamSynchronize(stream));
      if (lsmr_host->istop != 0 || launched >= sc.maxiter) break;
    }
    sc = *lsmr_host;
    if (dv) release(dv);
    touch(v); touch(x); touch(h); touch(hbar);
  }
};

}  // namespace mvus

using namespace mvus;

struct mvus_rccl_comm;
struct mvus_rccl_id { char internal[128]; };     // ncclUniqueId of rccl.h (passed by value to ncclCommInitRank)
struct mvus_rccl_handle { void* comm = nullptr; };
struct mvus_ba {
  HipBackend be;
  std::unique_ptr<HipSchur<HipBackend>> schur;   // normal-equation workspace, built on first use
  mvus_rccl_handle rccl;                         // communicator of mvus_ba_set_rccl (destroyed with the handle)
  ~mvus_ba();
};

static thread_local std::string g_create_error;

// stateless helpers share this: device buffers of one call, freed on every exit
namespace {
struct CallBuffers {
  std::vector<void*> bufs;
  hipStream_t st = nullptr;
  void open(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) throw HipError{"no usable HIP device (libmvusba has no CPU fallback)"};
    MVUS_HIP(hipSetDevice(device));
    MVUS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  }
  template <class T>
  T* get(size_t count) { void* p = nullptr; MVUS_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T))); bufs.push_back(p); return static_cast<T*>(p); }
  template <class T>
  T* put(const T* host, size_t count) { T* d = get<T>(count); if (count) MVUS_HIP(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, st)); return d; }
  ~CallBuffers() { for (void* p : bufs) (void)hipFree(p); if (st) (void)hipStreamDestroy(st); }
};
}  // namespace


// device work arrays of the smoothing fit in one precision (double, or double-double for ill-conditioned knot sets)
template <class T>
struct FitWork {
  T *SB = nullptr, *G5 = nullptr, *BtB = nullptr, *Mx = nullptr, *Lf = nullptr, *rhs = nullptr, *yw = nullptr, *YL = nullptr, *parts = nullptr;
  bool ready = false, penalty = false;
  void alloc(CallBuffers& cb, size_t nest) {
    if (ready) return;
    SB = cb.get<T>((size_t)kFitBlk * (nest + kFitSliceBlocks)); G5 = cb.get<T>(5 * nest); BtB = cb.get<T>(5 * nest); Mx = cb.get<T>(5 * nest);
    const size_t rows = nest + kBandPartsMax;            // the transposed copies hold length(0) rows for EVERY interior
    Lf = cb.get<T>(5 * rows); rhs = cb.get<T>(3 * nest); yw = cb.get<T>(3 * rows);
    YL = cb.get<T>(4 * rows); parts = cb.get<T>((size_t)kBandPartsWork);
    ready = true;
  }
};
// systems of at least this many rows go to k_band_solve_parts (MVUS_BAND_PARTS_MIN: the tests push the small fixtures through it too)
static int band_parts_min() {
  static const int v = [] { const char* e = std::getenv("MVUS_BAND_PARTS_MIN"); return e ? std::max(16, std::atoi(e)) : 192; }();
  return v;
}
// banded solve of the pass; returns true when the factor's diagonal in out[0] is FITPACK's (one chain in the natural order)
template <int HB, class T>
static bool fit_band_solve(hipStream_t st, FitWork<T>& w, int ncoef, const T* Mband, double* cd, double* out, int* fail) {
  const BandParts bp = band_parts(ncoef, HB, band_parts_min());
  if (bp.P < 2) {
    hipLaunchKernelGGL((k_band_solve<HB, T>), dim3(1), dim3(64), 0, st, ncoef, Mband, w.rhs, w.Lf, w.yw, cd, out, fail);
    return true;
  }
  hipLaunchKernelGGL((k_band_solve_parts<HB, T>), dim3(1), dim3(64 * ((bp.P + 63) / 64)), 0, st, ncoef, bp, Mband, w.rhs, w.Lf, w.yw, w.YL, w.parts, cd, out, fail);
  return false;
}
static dim3 fit_blocks(long long cnt) { return dim3((unsigned)((cnt + 255) / 256)); }
// least-squares spline on the current knots: normal equations from the span blocks, banded Cholesky, coefficients -> cd
template <class T>
static bool fit_lsq_pass(hipStream_t st, FitWork<T>& w, long long m, const long long* first, const double* q, const double* dX, int ncoef, int nrint,
                         double* cd, double* out, int* fail) {
  constexpr int NT = sizeof(T) == sizeof(double) ? 256 : 64;
  const int nslice = fit_slices(nrint);
  hipLaunchKernelGGL((k_fit_blocks<T, NT>), dim3(nrint, nslice), dim3(NT), 0, st, m, first, q, dX, w.SB);
  if (nslice > 1) hipLaunchKernelGGL(k_fit_slice_sum<T>, fit_blocks((long long)nrint * kFitBlk), dim3(256), 0, st, (long long)nrint * kFitBlk, nslice, w.SB);
  hipLaunchKernelGGL(k_fit_band<T>, fit_blocks(ncoef), dim3(256), 0, st, ncoef, nrint, w.SB, w.G5, w.rhs, band_parts(ncoef, 3, band_parts_min()), w.Lf, w.yw);
  w.penalty = false;
  return fit_band_solve<3, T>(st, w, ncoef, w.G5, cd, out, fail);
}
// the sum of the factor's diagonal in the natural elimination order (fppara's initial p) when the last pass was partitioned
template <class T>
static void fit_lsq_diag(hipStream_t st, FitWork<T>& w, int ncoef, double* out) {
  hipLaunchKernelGGL((k_band_diag_sum<3, T>), dim3(1), dim3(64), 0, st, ncoef, w.G5, out);
}
// smoothing spline for one value of p on the same knots (fit_lsq_pass has run in this precision)
template <class T>
static void fit_smooth_pass(hipStream_t st, FitWork<T>& w, int ncoef, int n8, const double* bd, double pinv, double* cd, double* out, int* fail) {
  if (!w.penalty) { hipLaunchKernelGGL(k_fit_penalty<T>, fit_blocks(ncoef), dim3(256), 0, st, ncoef, n8, bd, w.BtB); w.penalty = true; }
  hipLaunchKernelGGL(k_fit_combine<T>, fit_blocks(5ll * ncoef), dim3(256), 0, st, 5ll * ncoef, w.G5, w.BtB, pinv, w.Mx, ncoef, w.rhs,
                     band_parts(ncoef, 4, band_parts_min()), w.Lf, w.yw);
  fit_band_solve<4, T>(st, w, ncoef, w.Mx, cd, out, fail);
}

template <class F>
static int guarded(mvus_ba* h, F&& fn) {
  if (!h) return MVUS_E_INVALID;
  try {
    (void)hipSetDevice(h->be.device);
    ++h->be.api_seq;                 // (what an LM solve carries over to the next call is valid for the very next call only: HipBackend::lm_carry)
    return fn();
  } catch (const HipError& e) {
    h->be.err = e.msg;
    return e.code;
  } catch (const std::exception& e) {
    h->be.err = e.what();
    return MVUS_E_INVALID;
  }
}

extern "C" {

void mvus_default_opts(mvus_solve_opts* o) {
  if (!o) return;
  o->solver = MVUS_SOLVER_TRF_LSMR; o->jac_mode = MVUS_JAC_PATTERN; o->max_nfev = 10;
  o->ftol = 1e-8; o->xtol = 1e-12; o->gtol = 1e-8;
  o->lsmr_atol = 1e-6; o->lsmr_btol = 1e-6; o->lsmr_conlim = 1e8; o->lsmr_maxiter = 0; o->verbose = 0; o->lm_lambda_min = 3e-3; o->lm_trust_radius = -1.0;
}

int32_t mvus_abi_sizes(int32_t* solve_opts_size, int32_t* result_size, int32_t* problem_size) {
  if (solve_opts_size) *solve_opts_size = (int32_t)sizeof(mvus_solve_opts);
  if (result_size) *result_size = (int32_t)sizeof(mvus_result);
  if (problem_size) *problem_size = (int32_t)sizeof(mvus_problem);
  return MVUS_ABI_VERSION;
}

int mvus_ba_create(const mvus_problem* p, mvus_ba** out) {
  if (!out) return MVUS_E_INVALID;
  *out = nullptr;
  mvus_ba* h = nullptr;
  try {
    h = new mvus_ba();
    std::string msg = h->be.hp.build(p);
    if (!msg.empty()) { g_create_error = msg; delete h; return MVUS_E_INVALID; }
  } catch (const std::exception& e) {          // bad_alloc / length_error on absurd sizes: nothing throws across the ABI
    g_create_error = e.what();
    delete h;
    return MVUS_E_INVALID;
  }
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= p->device)
      throw HipError{"no usable HIP device (libmvusba has no CPU fallback)"};
    h->be.init(p);
  } catch (const HipError& e) {
    g_create_error = e.msg;
    delete h;
    return MVUS_E_HIP;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    delete h;
    return MVUS_E_INVALID;
  }
  *out = h;
  return MVUS_OK;
}

void mvus_ba_destroy(mvus_ba* h) { delete h; }

const char* mvus_last_error(const mvus_ba* h) { return h ? h->be.err.c_str() : g_create_error.c_str(); }

int64_t mvus_ba_num_params(const mvus_ba* h) { return h ? h->be.hp.n : -1; }
int64_t mvus_ba_num_residuals(const mvus_ba* h) { return h ? h->be.hp.m : -1; }
int64_t mvus_ba_num_motion_rows(const mvus_ba* h) { return h ? h->be.hp.T : -1; }
int32_t mvus_ba_num_slots(const mvus_ba* h) { return h ? h->be.hp.NS : -1; }

int mvus_ba_set_x(mvus_ba* h, const double* x) {
  return guarded(h, [&] { h->be.upload(h->be.x_cur, x, h->be.hp.n); return MVUS_OK; });
}

int mvus_ba_residual(mvus_ba* h, const double* x, double* f) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    be.upload(be.x_cur, x, be.hp.n);
    be.residual(be.x_cur, be.f_cur);
    if (f) be.download(f, be.f_cur, be.hp.m);
    return MVUS_OK;
  });
}

int mvus_ba_residual_jacobian(mvus_ba* h, const double* x, int32_t jac_mode, double* f, double* J, int32_t* ctrl) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    be.upload(be.x_cur, x, be.hp.n);
    // rows that stay invisible are never written by the kernel: clear so the host copy is well defined
    if (J) { be.ensure_J(); MVUS_HIP(hipMemsetAsync(be.J, 0, sizeof(double) * j_doubles(be.hp.NS, be.dp.n_chunks), be.stream)); }
    be.jacobian(be.x_cur, be.f_cur, jac_mode);
    be.held_analytic_at_xcur = jac_mode == MVUS_JAC_ANALYTIC;
    if (f) be.download(f, be.f_cur, be.hp.m);
    if (J) {                                 // the ABI's layout is slot-major (mvus_ba.h); the device keeps J chunk-major
      PoolGuard<HipBackend> pool(be);
      double* Jout = pool.get((int64_t)2 * be.hp.NS * std::max<int64_t>(be.hp.M, 1));
      MVUS_HIP(hipMemsetAsync(Jout, 0, sizeof(double) * 2 * be.hp.NS * be.hp.M, be.stream));
      if (be.dp.n_chunks > 0) {
        if (be.hp.calib) hipLaunchKernelGGL(k_j_export<30>, dim3(be.dp.n_chunks), dim3(kThreads), 0, be.stream, be.dp, be.J, Jout);
        else hipLaunchKernelGGL(k_j_export<21>, dim3(be.dp.n_chunks), dim3(kThreads), 0, be.stream, be.dp, be.J, Jout);
      }
      be.download(J, Jout, (int64_t)2 * be.hp.NS * be.hp.M);
    }
    if (ctrl) {
      MVUS_HIP(hipMemcpyAsync(ctrl, be.span, sizeof(int32_t) * be.hp.M, hipMemcpyDeviceToHost, be.stream));
      MVUS_HIP(hipStreamSynchronize(be.stream));
    }
    return MVUS_OK;
  });
}

int mvus_ba_motion_rows(mvus_ba* h, const double* x, int32_t jac_mode, double* mf, double* mJ, int32_t* mctrl) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    be.upload(be.x_cur, x, be.hp.n);
    be.jacobian(be.x_cur, be.f_cur, jac_mode);
    const int T = be.hp.T;
    if (mf && T) be.download(mf, be.f_cur + 2 * be.hp.M, T);
    if (mJ && T) be.download(mJ, be.mJ, (int64_t)36 * T);
    if (mctrl && T) {
      MVUS_HIP(hipMemcpyAsync(mctrl, be.mctrl, sizeof(int32_t) * 3 * T, hipMemcpyDeviceToHost, be.stream));
      MVUS_HIP(hipStreamSynchronize(be.stream));
    }
    return MVUS_OK;
  });
}

int mvus_ba_set_pattern(mvus_ba* h, const double* x0, int32_t* pat_out) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    be.upload(be.x_cur, x0, be.hp.n);
    be.set_pattern(be.x_cur);
    if (pat_out) {
      MVUS_HIP(hipMemcpyAsync(pat_out, be.pat0, sizeof(int32_t) * be.hp.M, hipMemcpyDeviceToHost, be.stream));
      MVUS_HIP(hipStreamSynchronize(be.stream));
    }
    return MVUS_OK;
  });
}

int mvus_ba_upload_pattern(mvus_ba* h, const int32_t* pat, const int32_t* motion_pat) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if ((!pat && be.hp.M > 0)) { be.err = "upload_pattern: NULL pattern"; return MVUS_E_INVALID; }
    const std::string msg = be.upload_pattern(pat, motion_pat);
    if (!msg.empty()) { be.err = msg; return MVUS_E_INVALID; }
    return MVUS_OK;
  });
}

int mvus_ba_motion_pattern(mvus_ba* h, int32_t* motion_pat_out) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (be.hp.T > 0) {
      if (!motion_pat_out) { be.err = "motion_pattern: NULL output"; return MVUS_E_INVALID; }
      MVUS_HIP(hipMemcpyAsync(motion_pat_out, be.ms_pat_dev, sizeof(int32_t) * be.hp.T, hipMemcpyDeviceToHost, be.stream));
      MVUS_HIP(hipStreamSynchronize(be.stream));
    }
    return MVUS_OK;
  });
}

int mvus_ba_set_deterministic(mvus_ba* h, int32_t on) {
  return guarded(h, [&] { h->be.det_assembly = on != 0; return MVUS_OK; });
}
int mvus_ba_deterministic_fallback(mvus_ba* h, int32_t* fell_back) {
  return guarded(h, [&] {
    if (!fell_back) { h->be.err = "bad arguments"; return MVUS_E_INVALID; }
    *fell_back = (h->schur && h->schur->last_atomic) ? 1 : 0;
    return MVUS_OK;
  });
}

int mvus_ba_set_fd_groups(mvus_ba* h, const int32_t* groups, int32_t num_groups) {
  return guarded(h, [&] {
    if (!groups || num_groups < 1) { h->be.err = "bad arguments"; return MVUS_E_INVALID; }
    for (int64_t j = 0; j < h->be.hp.n; ++j) if (groups[j] < 0 || groups[j] >= num_groups) { h->be.err = "group index out of range"; return MVUS_E_INVALID; }
    h->be.set_fd_groups(groups, num_groups);
    return MVUS_OK;
  });
}

}  // extern "C" (the two helpers below are C++)

// scipy/optimize/_group_columns.pyx: group_sparse over the CSC arrays of the permuted pattern; groups[order] = result
static int32_t group_sparse_permuted(int64_t m, int64_t n, const std::vector<int64_t>& indptr, const std::vector<int64_t>& indices, const int64_t* order, int32_t* groups) {
  std::vector<int32_t> gp((size_t)n, -1);
  std::vector<unsigned char> in_union((size_t)m);
  int32_t current = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (gp[(size_t)i] >= 0) continue;
    gp[(size_t)i] = current;
    bool all_grouped = true;
    std::fill(in_union.begin(), in_union.end(), (unsigned char)0);
    for (int64_t k = indptr[(size_t)i]; k < indptr[(size_t)i + 1]; ++k) in_union[(size_t)indices[(size_t)k]] = 1;
    for (int64_t j = 0; j < n; ++j) {
      if (gp[(size_t)j] >= 0) continue;
      all_grouped = false;
      bool intersect = false;
      for (int64_t k = indptr[(size_t)j]; k < indptr[(size_t)j + 1]; ++k) if (in_union[(size_t)indices[(size_t)k]]) { intersect = true; break; }
      if (!intersect) {
        for (int64_t k = indptr[(size_t)j]; k < indptr[(size_t)j + 1]; ++k) in_union[(size_t)indices[(size_t)k]] = 1;
        gp[(size_t)j] = current;
      }
    }
    if (all_grouped) break;
    ++current;
  }
  int32_t ng = 0;
  for (int64_t j = 0; j < n; ++j) { groups[(size_t)order[j]] = gp[(size_t)j]; ng = std::max(ng, gp[(size_t)j] + 1); }      // groups[order] = groups.copy()
  return ng;
}
// A[:, order] in CSC from a generator of entries: for_each(emit) calls emit(row, col) for every entry, the same sequence every time it is called
template <class ForEach>
static int32_t group_columns_of(int64_t m, int64_t n, const int64_t* order, int32_t* groups, ForEach&& for_each) {
  std::vector<int64_t> where((size_t)n, -1);
  for (int64_t j = 0; j < n; ++j) {
    if (order[j] < 0 || order[j] >= n || where[(size_t)order[j]] >= 0) { g_create_error = "group_columns: order is not a permutation"; return MVUS_E_INVALID; }
    where[(size_t)order[j]] = j;
  }
  std::vector<int64_t> indptr((size_t)n + 1, 0);
  bool bad = false;
  int64_t nnz = 0;
  for_each([&](int64_t r, int64_t c) { if (r < 0 || r >= m || c < 0 || c >= n) { bad = true; return; } ++indptr[(size_t)where[(size_t)c] + 1]; ++nnz; });
  if (bad) { g_create_error = "group_columns: entry outside the matrix"; return MVUS_E_INVALID; }
  for (int64_t j = 0; j < n; ++j) indptr[(size_t)j + 1] += indptr[(size_t)j];
  std::vector<int64_t> fill(indptr.begin(), indptr.end() - 1), indices((size_t)nnz);
  for_each([&](int64_t r, int64_t c) { indices[(size_t)fill[(size_t)where[(size_t)c]]++] = r; });
  return group_sparse_permuted(m, n, indptr, indices, order, groups);
}

extern "C" {

int32_t mvus_group_columns(int64_t m, int64_t n, int64_t nnz, const int64_t* rows, const int64_t* cols, const int64_t* order, int32_t* groups) {
  if (m < 0 || n < 1 || nnz < 0 || n > (1ll << 31) - 2 || (nnz > 0 && (!rows || !cols)) || !order || !groups) { g_create_error = "group_columns: bad arguments"; return MVUS_E_INVALID; }
  try {
    return group_columns_of(m, n, order, groups, [&](auto&& emit) { for (int64_t k = 0; k < nnz; ++k) emit(rows[k], cols[k]); });
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return MVUS_E_INVALID;
  }
}

int32_t mvus_fd_groups(const mvus_problem* p, const int32_t* pat, const int32_t* motion_pat, const int64_t* order, int32_t* groups) {
  if (!p || !pat || !order || !groups) { g_create_error = "fd_groups: bad arguments"; return MVUS_E_INVALID; }
  try {
    HostProblem hp;
    const std::string msg = hp.build(p);
    if (!msg.empty()) { g_create_error = msg; return MVUS_E_INVALID; }
    if (hp.T > 0 && !motion_pat) { g_create_error = "fd_groups: motion_reg needs the motion-row codes (mvus_ba_motion_pattern)"; return MVUS_E_INVALID; }
    const int C = hp.C, P = hp.P;
    const bool opt_sync = p->opt_sync != 0, rs_free = p->rs_free != 0;
    // control point p of the concatenated list -> its spline s, column of its x coefficient, stride between coordinates
    std::vector<int64_t> coff((size_t)p->num_splines + 1, 0), xoff((size_t)p->num_splines, 0), ncoef((size_t)p->num_splines, 0);
    {
      int64_t base = (int64_t)C * (3 + P);
      for (int s = 0; s < p->num_splines; ++s) {
        ncoef[(size_t)s] = p->knot_offsets[s + 1] - p->knot_offsets[s] - 4;
        coff[(size_t)s + 1] = coff[(size_t)s] + ncoef[(size_t)s];
        xoff[(size_t)s] = base;
        base += 3 * ncoef[(size_t)s];
      }
    }
    auto spline_cols = [&](int32_t code, auto&& emit_col) {          // the (<= 12) spline columns of a row with this pattern code
      const int32_t pc = pattern_index(code), mk = pattern_mask(code);
      const int s = (int)(std::upper_bound(coff.begin(), coff.end(), (int64_t)pc) - coff.begin()) - 1;
      if (s < 0 || s >= p->num_splines) return false;
      const int64_t base = xoff[(size_t)s] + (pc - coff[(size_t)s]);
      for (int k = 0; k < 4; ++k) if ((mk >> k) & 1) for (int d = 0; d < 3; ++d) emit_col(base + k + d * ncoef[(size_t)s]);
      return true;
    };
    bool bad_code = false;
    const int64_t M = hp.M;
    auto for_each = [&](auto&& emit) {
      for (int c = 0; c < C; ++c) {
        const int64_t a = p->det_offsets[c], b = p->det_offsets[c + 1];
        // Rows with the same set of columns are interchangeable for group_sparse (it only asks whether two columns share a row),
        // so ONE row stands for a run of them: the v row of a detection repeats its u row, and time-ordered detections of one
        // camera repeat the code of their neighbour for a whole knot span (200 k rows -> a few thousand at configs[1]; same groups).
        int32_t prev = -1;
        for (int64_t i = a; i < b; ++i) {
          if (pat[i] < 0 || pat[i] == prev) continue;
          prev = pat[i];
          const int64_t row = 2 * a + (i - a);
          if (opt_sync) { emit(row, (int64_t)c); emit(row, (int64_t)C + c); }
          if (rs_free) emit(row, 2 * (int64_t)C + c);
          for (int k = 0; k < P; ++k) emit(row, 3 * (int64_t)C + (int64_t)c * P + k);
          if (!spline_cols(pat[i], [&](int64_t col) { emit(row, col); })) bad_code = true;
        }
      }
      for (int64_t j = 0; j < hp.T; ++j) {
        if (motion_pat[j] < 0) continue;
        if (!spline_cols(motion_pat[j], [&](int64_t col) { emit(2 * M + j, col); })) bad_code = true;
      }
    };
    const int32_t ng = group_columns_of((int64_t)hp.m, (int64_t)hp.n, order, groups, for_each);
    if (bad_code) { g_create_error = "fd_groups: a pattern code points outside the control points"; return MVUS_E_INVALID; }
    return ng;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return MVUS_E_INVALID;
  }
}

int mvus_ba_jv(mvus_ba* h, const double* v, double* y) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (!be.has_jacobian) { be.err = "no Jacobian held: call mvus_ba_residual_jacobian first"; return MVUS_E_INVALID; }
    double* vd = be.alloc(be.hp.n); double* yd = be.alloc(be.hp.m);
    be.upload(vd, v, be.hp.n);
    be.jv(vd, yd);
    be.download(y, yd, be.hp.m);
    be.release(vd); be.release(yd);
    return MVUS_OK;
  });
}

int mvus_ba_jtu(mvus_ba* h, const double* u, double* z) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (!be.has_jacobian) { be.err = "no Jacobian held: call mvus_ba_residual_jacobian first"; return MVUS_E_INVALID; }
    double* ud = be.alloc(be.hp.m); double* zd = be.alloc(be.hp.n);
    be.upload(ud, u, be.hp.m);
    be.jtu(ud, zd);
    be.download(z, zd, be.hp.n);
    be.release(ud); be.release(zd);
    return MVUS_OK;
  });
}

int mvus_ba_normal_equations(mvus_ba* h, double* g, double* JtJ_cam, double* band, double* cross, int32_t* W_out) {
  return guarded(h, [&] {
    if (!h->schur) h->schur.reset(new HipSchur<HipBackend>(h->be));
    return schur_export(h->be, *h->schur, g, JtJ_cam, band, cross, W_out);
  });
}

int mvus_ba_lm_step(mvus_ba* h, double lambda, double* p_out) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (!p_out || !(lambda >= 0)) { be.err = "lm_step: bad arguments"; return MVUS_E_INVALID; }
    if (!be.has_jacobian) { be.err = "no Jacobian held: call mvus_ba_residual_jacobian first"; return MVUS_E_INVALID; }
    if (be.hp.C * (3 + be.hp.P) > 1152) throw HipError{"LM_SCHUR: reduced camera system larger than 1152 unknowns"};
    if (!h->schur) h->schur.reset(new HipSchur<HipBackend>(be));
    HipSchur<HipBackend>& sc = *h->schur;
    sc.assemble_held(be);
    sc.solve_async(lambda);
    be.download(p_out, sc.step_ptr(), be.hp.n);          // synchronises
    if (!sc.solve_ok() && sc.retry_same()) { sc.solve_async(lambda); be.download(p_out, sc.step_ptr(), be.hp.n); }
    if (!sc.solve_ok()) { be.err = "lm_step: the damped normal equations are not positive definite at this lambda"; return MVUS_E_NUMERIC; }
    return MVUS_OK;
  });
}

int mvus_ba_solve(mvus_ba* h, double* x, const mvus_solve_opts* opts, mvus_result* res, double* f_out) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (!x || !opts || !res) { be.err = "NULL argument"; return MVUS_E_INVALID; }
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t n = be.hp.n;
    std::vector<double> xv;                          // (the TRF path works on a vector; the LM driver on the caller's buffer itself)
    if (opts->solver != MVUS_SOLVER_LM_SCHUR || opts->jac_mode == MVUS_JAC_PATTERN) xv.assign(x, x + n);
    if ((int64_t)be.lb_fixed.size() != n) {          // the box of the rs block (common.py:652-668): fixed for the handle, built once
      be.lb_fixed.assign(n, -INFINITY); be.ub_fixed.assign(n, INFINITY);
      if (be.hp.rs_bounds) for (int c = 0; c < be.hp.C; ++c) { be.lb_fixed[2 * be.hp.C + c] = 0.0; be.ub_fixed[2 * be.hp.C + c] = 1.0; }
    }
    const std::vector<double>&lb = be.lb_fixed, &ub = be.ub_fixed;
    SolveOptions so;
    so.jac_mode = opts->jac_mode; so.max_nfev = opts->max_nfev; so.ftol = opts->ftol; so.xtol = opts->xtol; so.gtol = opts->gtol;
    so.lsmr_atol = opts->lsmr_atol; so.lsmr_btol = opts->lsmr_btol; so.lsmr_conlim = opts->lsmr_conlim;
    so.lsmr_maxiter = opts->lsmr_maxiter; so.verbose = opts->verbose; so.lm_lambda_min = opts->lm_lambda_min >= 0 ? opts->lm_lambda_min : 0.0;
    so.lm_trust_radius = opts->lm_trust_radius;
    if (so.jac_mode == MVUS_JAC_PATTERN && !be.pattern_uploaded) {
      be.upload(be.x_cur, xv.data(), n);
      be.set_pattern(be.x_cur);
    }
    if (so.jac_mode == MVUS_JAC_FD && (!be.has_pattern || be.fd_ngroups <= 0)) {
      be.err = "MVUS_JAC_FD: call mvus_ba_set_pattern(x0) and mvus_ba_set_fd_groups first";
      return MVUS_E_INVALID;
    }
    SolveResult sr;
    be.held_analytic_at_xcur = false;
    if (opts->solver == MVUS_SOLVER_LM_SCHUR) {
      if (be.hp.C * (3 + be.hp.P) > 1152) throw HipError{"LM_SCHUR: reduced camera system larger than 1152 unknowns"};
      if (!h->schur) h->schur.reset(new HipSchur<HipBackend>(be));
      so.lm_lambda0 = be.lm_lambda; so.lm_nu0 = be.lm_nu;
      // knots closer than a frame (band wider than six control points): more control points than detections -- the spline is
      // held by the motion regulariser alone between the data, and a converging LM whose damping falls to 3e-3 diag(H) follows
      // noise along those directions (the incremental loop then ends 3 m off; with 0.3 it ends where TRF + LSMR ends:
      // profiles/round4/r04_loop_lm_wide_band_damping.txt).  The floor of such problems is at least kLambdaMinWide.
      constexpr double kLambdaMinWide = 0.3;
      if (h->schur->wide && so.lm_trust_radius < 0) so.lm_lambda_min = std::max(so.lm_lambda_min, kLambdaMinWide);      // (a trust region bounds those steps itself)
      sr = lm_schur(be, *h->schur, x, lb, ub, so, be.f_cur);
      if (!sr.error) { be.lm_lambda = std::min(std::max(sr.lm_lambda, 1e-12), 1e6); be.lm_nu = std::min(sr.lm_nu, 1024.0); }
      if (sr.jac_stale) be.has_jacobian = false;      // mvus_ba_jv / jtu / lm_step must not pair J(x_old) with f(x_new)
    }
    else sr = trf_lsmr(be, xv, lb, ub, so, be.f_cur);
    const bool lm = opts->solver == MVUS_SOLVER_LM_SCHUR;
    if (sr.error == -4) {            // a row left the time slice: the point reached so far goes back to the caller, who re-cuts there (already in x)
      MVUS_HIP(hipStreamSynchronize(be.stream));
      h->schur.reset();              // (the device flag is raised: the next solve on this handle starts from fresh storage)
      res->cost = sr.cost; res->optimality = 0; res->nfev = sr.nfev; res->njev = sr.njev; res->status = 0; res->lin_iters = sr.lin_iters;
      res->initial_cost = sr.initial_cost; res->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      be.err = "time shard: a detection or motion row reaches control points outside this rank's slice +- halo (the time stamps have drifted since the cuts were made): re-cut at the returned point";
      return MVUS_E_RESHARD;
    }
    if (sr.error) { be.err = "residuals are not finite in the initial point, or x0 is outside of the bounds"; return MVUS_E_NUMERIC; }
    if (!lm) std::memcpy(x, xv.data(), sizeof(double) * n);
    if (f_out) be.download(f_out, be.f_cur, be.hp.m);
    // (an LM solve that ends on its evaluation budget may leave the linearisation at the returned point running: x and the scalars are
    // on the host already -- mapped memory, fetched behind an event -- and whatever reads those blocks is a later call on this stream)
    if (!(lm && sr.async_tail && !f_out)) MVUS_HIP(hipStreamSynchronize(be.stream));
    res->cost = sr.cost; res->optimality = sr.optimality; res->nfev = sr.nfev; res->njev = sr.njev; res->status = sr.status;
    res->lin_iters = sr.lin_iters; res->initial_cost = sr.initial_cost;
    res->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MVUS_OK;
  });
}

int mvus_ba_outlier_mask(mvus_ba* h, const double* x, double thres, uint8_t* keep) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    be.upload(be.x_cur, x, be.hp.n);
    be.residual(be.x_cur, be.f_cur);
    uint8_t* kd = nullptr;
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&kd), std::max<int64_t>(be.hp.M, 1)));
    if (be.dp.n_chunks > 0) hipLaunchKernelGGL(k_outlier_mask, dim3(be.dp.n_chunks), dim3(kThreads), 0, be.stream, be.dp, be.f_cur, thres, kd);
    hipError_t e = hipMemcpyAsync(keep, kd, be.hp.M, hipMemcpyDeviceToHost, be.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(be.stream);
    (void)hipFree(kd);
    MVUS_HIP(e);
    return MVUS_OK;
  });
}

int mvus_ba_remove_outliers(mvus_ba* h, const double* x, double thres, uint8_t* keep_out, int64_t* det_offsets_out) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    be.upload(be.x_cur, x, be.hp.n);
    h->schur.reset();                       // sized by the launch tables
    be.remove_outliers(be.x_cur, thres, keep_out, det_offsets_out);
    if (be.allreduce) {                     // a shard filters its own detections; only the global row count is shared
      const bool counts_motion = be.tshard.on ? be.tshard.rank == 0 : be.is_root != 0;
      be.scal_host[2] = (double)(2 * be.hp.M + (counts_motion ? be.hp.T : 0));
      MVUS_HIP(hipMemcpyAsync(be.scal_dev + 2, be.scal_host + 2, sizeof(double), hipMemcpyHostToDevice, be.stream));
      be.reduce(be.scal_dev + 2, 1);
      be.m_glob = (int64_t)(be.read_slot(2) + 0.5);
    }
    return MVUS_OK;
  });
}

// ---- RCCL from the library: librccl.so.1 opened at run time, five entry points ----
namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, mvus_rccl_id, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { err = std::string("librccl.so.1 cannot be opened: ") + dlerror(); return false; }
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) { err = "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy"; dlclose(lib); lib = nullptr; return false; }
    return true;
  }
  std::string what(int rc) const { return GetErrorString ? std::string(GetErrorString(rc)) : ("ncclResult " + std::to_string(rc)); }
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;
constexpr int kNcclFloat64 = 8, kNcclSum = 0;      // ncclDataType_t / ncclRedOp_t of rccl.h
}  // namespace

static int rccl_allreduce_cb(void* user, void* buf, size_t count, void* stream) {
  mvus_rccl_handle* c = static_cast<mvus_rccl_handle*>(user);
  return g_rccl.AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, c->comm, static_cast<hipStream_t>(stream));
}

int mvus_ba_set_allreduce(mvus_ba* h, mvus_allreduce_fn fn, void* user, int32_t is_root) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (fn != rccl_allreduce_cb && h->rccl.comm) {      // another route (or none) replaces the library's own communicator: tear it down
      MVUS_HIP(hipStreamSynchronize(be.stream));
      if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->rccl.comm);
      h->rccl.comm = nullptr;
    }
    be.allreduce = fn; be.allreduce_user = user; be.is_root = is_root;
    be.m_glob = be.hp.m;
    if (fn) {   // global row count = sum of the shards' detection rows + the motion rows once
      const bool counts_motion = be.tshard.on ? be.tshard.rank == 0 : is_root != 0;
      be.scal_host[2] = (double)(2 * be.hp.M + (counts_motion ? be.hp.T : 0));
      MVUS_HIP(hipMemcpyAsync(be.scal_dev + 2, be.scal_host + 2, sizeof(double), hipMemcpyHostToDevice, be.stream));
      be.reduce(be.scal_dev + 2, 1);
      be.m_glob = (int64_t)(be.read_slot(2) + 0.5);
    }
    return MVUS_OK;
  });
}

}  // extern "C"
mvus_ba::~mvus_ba() {
  schur.reset();
  if (rccl.comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(be.stream); (void)g_rccl.CommDestroy(rccl.comm); }
}
extern "C" {

int mvus_rccl_available(void) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.load()) { g_create_error = g_rccl.err; return MVUS_E_COMM; }
  return MVUS_OK;
}

int mvus_rccl_unique_id(uint8_t id_out[128]) {
  if (!id_out) return MVUS_E_INVALID;
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.load()) { g_create_error = g_rccl.err; return MVUS_E_COMM; }
  mvus_rccl_id id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc != 0) { g_create_error = "ncclGetUniqueId: " + g_rccl.what(rc); return MVUS_E_COMM; }
  std::memcpy(id_out, id.internal, 128);
  return MVUS_OK;
}

int mvus_ba_set_rccl(mvus_ba* h, const uint8_t id[128], int32_t rank, int32_t world, int32_t is_root) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (!id || world < 1 || rank < 0 || rank >= world) { be.err = "set_rccl: bad arguments"; return MVUS_E_INVALID; }
    {
      std::lock_guard<std::mutex> lock(g_rccl_mutex);
      if (!g_rccl.load()) { be.err = g_rccl.err; return MVUS_E_COMM; }
    }
    // the route is torn down BEFORE anything can fail: a failed call leaves the handle without a collective route (sums of one
    // rank), never with a callback that points at a destroyed or half-initialised communicator
    MVUS_HIP(hipStreamSynchronize(be.stream));
    if (be.allreduce == rccl_allreduce_cb) { be.allreduce = nullptr; be.allreduce_user = nullptr; }
    if (h->rccl.comm) { (void)g_rccl.CommDestroy(h->rccl.comm); h->rccl.comm = nullptr; }
    mvus_rccl_id uid;
    std::memcpy(uid.internal, id, 128);
    const int rc = g_rccl.CommInitRank(&h->rccl.comm, world, uid, rank);
    if (rc != 0) { h->rccl.comm = nullptr; be.err = "ncclCommInitRank: " + g_rccl.what(rc); return MVUS_E_COMM; }
    return mvus_ba_set_allreduce(h, rccl_allreduce_cb, &h->rccl, is_root);
  });
}

int mvus_ba_time_allreduce(mvus_ba* h, int64_t count, int32_t reps, double* avg_ms) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (count < 1 || reps < 1 || !avg_ms) { be.err = "bad arguments"; return MVUS_E_INVALID; }
    if (!be.allreduce) { be.err = "time_allreduce: no all-reduce route installed"; return MVUS_E_INVALID; }
    PoolGuard<HipBackend> pool(be);
    double* buf = pool.get(count);
    be.fill(buf, 0.0, count);
    hipEvent_t e0, e1;
    MVUS_HIP(hipEventCreate(&e0)); MVUS_HIP(hipEventCreate(&e1));
    float ms = 0;
    try {
      for (int i = 0; i < 3; ++i) be.reduce(buf, (size_t)count);
      MVUS_HIP(hipEventRecord(e0, be.stream));
      for (int i = 0; i < reps; ++i) be.reduce(buf, (size_t)count);
      MVUS_HIP(hipEventRecord(e1, be.stream));
      MVUS_HIP(hipEventSynchronize(e1));
      MVUS_HIP(hipEventElapsedTime(&ms, e0, e1));
    } catch (...) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); throw; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / reps;
    return MVUS_OK;
  });
}

int mvus_ba_set_time_shard(mvus_ba* h, int32_t rank, int32_t world, const int32_t* ctrl_cuts, int32_t halo) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (world < 1 || rank < 0 || rank >= world || !ctrl_cuts || halo < 1) { be.err = "set_time_shard: bad arguments"; return MVUS_E_INVALID; }
    if (ctrl_cuts[0] != 0 || ctrl_cuts[world] != be.hp.N) { be.err = "set_time_shard: cuts must run from 0 to the number of control points"; return MVUS_E_INVALID; }
    for (int r = 0; r < world; ++r)
      if (ctrl_cuts[r + 1] <= ctrl_cuts[r]) { be.err = "set_time_shard: cuts must increase"; return MVUS_E_INVALID; }
    h->schur.reset();
    be.tshard.on = world > 1; be.tshard.rank = rank; be.tshard.world = world; be.tshard.halo = halo;
    be.tshard.cuts.assign(ctrl_cuts, ctrl_cuts + world + 1);
    be.dp.mot_lo = be.tshard.on ? ctrl_cuts[rank] : 0;
    be.dp.mot_hi = be.tshard.on ? ctrl_cuts[rank + 1] : 0x7fffffff;
    be.has_jacobian = false;
    return MVUS_OK;
  });
}

int mvus_ba_time_kernel(mvus_ba* h, int32_t which, int32_t launches, double* avg_ms) {
  return guarded(h, [&] {
    HipBackend& be = h->be;
    if (launches < 1 || !avg_ms || which < 0 || which > 6) { be.err = "bad arguments"; return MVUS_E_INVALID; }
    be.ensure_J();
    hipEvent_t e0, e1;
    MVUS_HIP(hipEventCreate(&e0)); MVUS_HIP(hipEventCreate(&e1));
    PoolGuard<HipBackend> pool(be);
    double* vn = pool.get(be.hp.n); double* um = pool.get(be.hp.m); double* zn = pool.get(be.hp.n); double* ym = pool.get(be.hp.m);
    be.fill(vn, 1e-3, be.hp.n); be.fill(um, 1e-3, be.hp.m);
    be.ensure_cams(be.x_cur);
    if (which >= 2 && which <= 4 && !be.has_jacobian) be.jacobian(be.x_cur, be.f_cur, MVUS_JAC_ANALYTIC);
    if (which == 6) be.residual(be.x_cur, be.f_cur);
    const dim3 g(std::max(be.dp.n_chunks, 1)), b(kThreads);
    if ((which == 4 || which == 6) && !h->schur) h->schur.reset(new HipSchur<HipBackend>(be));
    HipSchur<HipBackend>* schur = h->schur.get();
    // which == 1: the outputs (J, span, f) rotate over enough buffer sets that the bytes written between two uses of a
    // set exceed the 256 MiB Infinity Cache several times -- every launch writes to HBM, not into lines the previous
    // launch left in the cache.  which == 5 is the same kernel re-launched into ONE set, for comparison.
    struct OutSet { double* J; int32_t* span; double* f; };
    std::vector<OutSet> sets{{be.J, be.span, be.f_cur}};
    std::vector<void*> extra;
    auto free_extra = [&]() { for (void* p : extra) (void)hipFree(p); extra.clear(); };
    if (which == 1) {
      const size_t jb = sizeof(double) * j_doubles(be.hp.NS, be.dp.n_chunks);
      const size_t want = (size_t)1 << 30;                                   // >= 1 GiB in rotation (4x the Infinity Cache)
      const int nrot = (int)std::min<size_t>(16, std::max<size_t>(3, (want + jb - 1) / jb));
      for (int r = 1; r < nrot; ++r) {
        void *pj = nullptr, *ps = nullptr, *pf = nullptr;
        hipError_t e = hipMalloc(&pj, jb);
        if (e == hipSuccess) { extra.push_back(pj); e = hipMalloc(&ps, sizeof(int32_t) * std::max<int64_t>(be.hp.M, 1)); }
        if (e == hipSuccess) { extra.push_back(ps); e = hipMalloc(&pf, sizeof(double) * std::max<int64_t>(be.hp.m, 1)); }
        if (e == hipSuccess) extra.push_back(pf);
        if (e != hipSuccess) { free_extra(); MVUS_HIP(e); }
        sets.push_back({static_cast<double*>(pj), static_cast<int32_t*>(ps), static_cast<double*>(pf)});
      }
    }
    int turn = 0;
    auto launch = [&]() {
      switch (which) {
        case 0:
          if (be.hp.calib) hipLaunchKernelGGL((k_observations<true, false>), g, b, 0, be.stream, be.dp, be.cams, be.x_cur, be.f_cur, be.J, be.rspan, be.pat0, 0);
          else hipLaunchKernelGGL((k_observations<false, false>), g, b, 0, be.stream, be.dp, be.cams, be.x_cur, be.f_cur, be.J, be.rspan, be.pat0, 0);
          break;
        case 1:
        case 5: {
          const OutSet& o = sets[turn];
          turn = (turn + 1) % (int)sets.size();
          const dim3 gj(xcd_grid(be.dp.n_chunks));
          if (be.hp.calib) hipLaunchKernelGGL((k_observations<true, true>), gj, b, 0, be.stream, be.dp, be.cams, be.x_cur, o.f, o.J, o.span, be.pat0, 0);
          else hipLaunchKernelGGL((k_observations<false, true>), gj, b, 0, be.stream, be.dp, be.cams, be.x_cur, o.f, o.J, o.span, be.pat0, 0);
          break;
        }
        case 2:
          if (be.hp.calib) hipLaunchKernelGGL(k_jv<30>, dim3(xcd_grid(be.dp.n_chunks)), b, 0, be.stream, be.dp, be.J, be.span, vn, ym);
          else hipLaunchKernelGGL(k_jv<21>, dim3(xcd_grid(be.dp.n_chunks)), b, 0, be.stream, be.dp, be.J, be.span, vn, ym);
          break;
        case 3:
          be.jtu_local(um, zn);
          break;
        case 6:
          schur->assemble_local(be.f_cur, be.x_cur);       // Jacobian evaluated inside the assembly (the LM path)
          break;
        default:
          schur->assemble_local(be.f_cur);
      }
    };
    if (be.dp.n_chunks == 0) { *avg_ms = 0; free_extra(); return MVUS_OK; }
    float ms = 0;
    try {
      for (size_t w = 0; w < sets.size(); ++w) launch();                       // warm-up: every set touched once
      MVUS_HIP(hipEventRecord(e0, be.stream));
      for (int i = 0; i < launches; ++i) launch();
      MVUS_HIP(hipEventRecord(e1, be.stream));
      MVUS_HIP(hipEventSynchronize(e1));
      MVUS_HIP(hipEventElapsedTime(&ms, e0, e1));
      MVUS_HIP(hipGetLastError());
    } catch (...) { free_extra(); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); throw; }
    *avg_ms = (double)ms / launches;
    free_extra();
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (which == 1 || which == 5) {            // the handle's own set holds J(x_cur) again (set 0 was written by one of the launches)
      be.has_jacobian = true;
    }
    return MVUS_OK;
  });
}

int mvus_triangulate(int32_t device, int64_t N, const double* x1, const double* x2, const double* P1, const double* P2,
                     double* X, double* err1, double* err2) {
  if (N < 0 || !P1 || !P2 || (N > 0 && (!x1 || !x2 || !X))) { g_create_error = "triangulate: bad arguments"; return MVUS_E_INVALID; }
  if (N == 0) return MVUS_OK;
  double *dx1 = nullptr, *dx2 = nullptr, *dX = nullptr, *de = nullptr;
  hipStream_t st = nullptr;
  auto cleanup = [&]() { (void)hipFree(dx1); (void)hipFree(dx2); (void)hipFree(dX); (void)hipFree(de); if (st) (void)hipStreamDestroy(st); };
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) throw HipError{"no usable HIP device (libmvusba has no CPU fallback)"};
    MVUS_HIP(hipSetDevice(device));
    MVUS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t b2 = sizeof(double) * 2 * (size_t)N;
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&dx1), b2)); MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&dx2), b2));
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&dX), 2 * b2)); MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&de), b2));
    MVUS_HIP(hipMemcpyAsync(dx1, x1, b2, hipMemcpyHostToDevice, st));
    MVUS_HIP(hipMemcpyAsync(dx2, x2, b2, hipMemcpyHostToDevice, st));
    TriCams cams;
    std::memcpy(cams.P1, P1, sizeof(cams.P1)); std::memcpy(cams.P2, P2, sizeof(cams.P2));
    hipLaunchKernelGGL(k_triangulate, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, cams, (long long)N, dx1, dx2, dX,
                       err1 ? de : (double*)nullptr, err2 ? de + N : (double*)nullptr);
    MVUS_HIP(hipGetLastError());
    MVUS_HIP(hipMemcpyAsync(X, dX, 2 * b2, hipMemcpyDeviceToHost, st));
    if (err1) MVUS_HIP(hipMemcpyAsync(err1, de, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    if (err2) MVUS_HIP(hipMemcpyAsync(err2, de + N, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    MVUS_HIP(hipStreamSynchronize(st));
  } catch (const HipError& e) {
    g_create_error = e.msg;
    cleanup();
    return e.code;
  } catch (const std::exception& e) {          // nothing throws across the ABI
    g_create_error = e.what();
    cleanup();
    return MVUS_E_INVALID;
  }
  cleanup();
  return MVUS_OK;
}

int mvus_spline_eval(int32_t device, int32_t S, const double* interval, const int64_t* knot_offsets, const double* knots,
                     const double* coefs, int64_t nt, const double* t, double* X, int32_t* which) {
  if (S < 1 || !interval || !knot_offsets || !knots || !coefs || nt < 0 || (nt > 0 && (!t || !X || !which))) { g_create_error = "spline_eval: bad arguments"; return MVUS_E_INVALID; }
  for (int s = 0; s < S; ++s)
    if (knot_offsets[s + 1] - knot_offsets[s] < 8) { g_create_error = "spline_eval: a cubic spline needs at least 8 knots"; return MVUS_E_INVALID; }
  if (nt == 0) return MVUS_OK;
  try {
    CallBuffers cb;
    cb.open(device);
    std::vector<long long> koff(knot_offsets, knot_offsets + S + 1), coff(S + 1, 0);
    for (int s = 0; s < S; ++s) coff[s + 1] = coff[s] + 3 * (koff[s + 1] - koff[s] - 4);
    SplineSet sp;
    sp.S = S;
    sp.istart = cb.put(interval, (size_t)S); sp.iend = cb.put(interval + S, (size_t)S);
    sp.knot_off = cb.put(koff.data(), koff.size()); sp.knots = cb.put(knots, (size_t)koff[S]);
    sp.coef_off = cb.put(coff.data(), coff.size()); sp.coefs = cb.put(coefs, (size_t)coff[S]);
    const double* dt = cb.put(t, (size_t)nt);
    double* dX = cb.get<double>(3 * (size_t)nt);
    int32_t* dw = cb.get<int32_t>((size_t)nt);
    hipLaunchKernelGGL(k_spline_eval, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, cb.st, sp, (long long)nt, dt, dX, dw);
    MVUS_HIP(hipGetLastError());
    MVUS_HIP(hipMemcpyAsync(X, dX, sizeof(double) * 3 * nt, hipMemcpyDeviceToHost, cb.st));
    MVUS_HIP(hipMemcpyAsync(which, dw, sizeof(int32_t) * nt, hipMemcpyDeviceToHost, cb.st));
    MVUS_HIP(hipStreamSynchronize(cb.st));
  } catch (const HipError& e) {
    g_create_error = e.msg;
    return e.code;
  } catch (const std::exception& e) {          // bad_alloc / length_error from the host-side tables: nothing throws across the ABI
    g_create_error = e.what();
    return MVUS_E_INVALID;
  }
  return MVUS_OK;
}

int mvus_spline_lsq(int32_t device, int32_t num_knots, const double* knots, int64_t m, const double* t, const double* X, double* coefs) {
  const int n = num_knots - 4;
  if (num_knots < 8 || !knots || m < 1 || !t || !X || !coefs) { g_create_error = "spline_lsq: bad arguments"; return MVUS_E_INVALID; }
  for (int k = 1; k < num_knots; ++k) if (knots[k] < knots[k - 1]) { g_create_error = "spline_lsq: knot vector must be non-decreasing"; return MVUS_E_INVALID; }
  for (int64_t i = 0; i < m; ++i) if (!(t[i] >= knots[3] && t[i] <= knots[n])) { g_create_error = "spline_lsq: data outside the knot interval"; return MVUS_E_INVALID; }
  try {
    CallBuffers cb;
    cb.open(device);
    const double* dk = cb.put(knots, (size_t)num_knots);
    const double* dt = cb.put(t, (size_t)m);
    const double* dX = cb.put(X, 3 * (size_t)m);
    double* G = cb.get<double>(4 * (size_t)n);
    double* rhs = cb.get<double>(3 * (size_t)n);
    int* fail = cb.get<int>(1);
    MVUS_HIP(hipMemsetAsync(G, 0, sizeof(double) * 4 * n, cb.st));
    MVUS_HIP(hipMemsetAsync(rhs, 0, sizeof(double) * 3 * n, cb.st));
    MVUS_HIP(hipMemsetAsync(fail, 0, sizeof(int), cb.st));
    hipLaunchKernelGGL(k_lsq_accumulate, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, cb.st, dk, n, (long long)m, dt, dX, G, rhs);
    hipLaunchKernelGGL(k_lsq_solve, dim3(1), dim3(64), 0, cb.st, n, G, rhs, fail);
    MVUS_HIP(hipGetLastError());
    int fh = 0;
    MVUS_HIP(hipMemcpyAsync(coefs, rhs, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, cb.st));
    MVUS_HIP(hipMemcpyAsync(&fh, fail, sizeof(int), hipMemcpyDeviceToHost, cb.st));
    MVUS_HIP(hipStreamSynchronize(cb.st));
    if (fh) { g_create_error = "spline_lsq: the normal equations are not positive definite (a coefficient without data: Schoenberg-Whitney violated)"; return MVUS_E_NUMERIC; }
  } catch (const HipError& e) {
    g_create_error = e.msg;
    return e.code;
  } catch (const std::exception& e) {          // bad_alloc / length_error from the host-side tables: nothing throws across the ABI
    g_create_error = e.what();
    return MVUS_E_INVALID;
  }
  return MVUS_OK;
}

/* scipy.interpolate.splprep(X, u=u, s=s, k=3) on the GPU (spline_fit.hip.h): fppara's control flow here, every pass over the
 * samples and every banded solve on the device.  A SESSION holds the samples (checked and uploaded once) and the work arrays:
 * traj_to_spline's smooth_factor loop fits the same samples a dozen times with different s. */
struct mvus_spline_fit {
  CallBuffers cb;
  int64_t m = 0;
  std::vector<double> hu;                                  // the timestamps on the host (fpknot places knots at samples)
  const double *du = nullptr, *dX = nullptr;
  int32_t* span = nullptr;
  double *q = nullptr, *term = nullptr, *tot_part = nullptr, *fp_part = nullptr;
  int* fail = nullptr;
  long long* first = nullptr;
  double *cd = nullptr, *td = nullptr, *bd = nullptr, *out = nullptr;
  FitWork<double> w1;
  FitWork<dd> w2;
  size_t cap = 0;
};
static int spline_fit_open_impl(mvus_spline_fit& S, int32_t device, int64_t m, const double* u, const double* X) {
  constexpr int k = 3;
  if (m <= k || m > (1ll << 30) || !u || !X) { g_create_error = "spline_smooth: bad arguments (m > 3 samples, s > 0)"; return MVUS_E_INVALID; }
  for (int64_t i = 1; i < m; ++i) if (!(u[i] > u[i - 1])) { g_create_error = "spline_smooth: the timestamps must be strictly increasing"; return MVUS_E_INVALID; }
  for (int64_t i = 0; i < 3 * m; ++i) if (!std::isfinite(X[i])) { g_create_error = "spline_smooth: non-finite sample"; return MVUS_E_INVALID; }
  try {
    S.m = m;
    S.hu.assign(u, u + m);
    CallBuffers& cb = S.cb;
    cb.open(device);
    S.du = cb.put(u, (size_t)m);
    S.dX = cb.put(X, 3 * (size_t)m);
    S.span = cb.get<int32_t>((size_t)m);
    S.q = cb.get<double>(4 * (size_t)m);
    S.term = cb.get<double>((size_t)m);
    S.tot_part = cb.get<double>(1024);
    S.fp_part = cb.get<double>(512 + kFitSliceBlocks);
    S.fail = cb.get<int>(1);
    MVUS_HIP(hipStreamSynchronize(cb.st));                 // u and X may go away after this call
  } catch (const HipError& e) {
    g_create_error = e.msg;
    return e.code;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return MVUS_E_INVALID;
  }
  return MVUS_OK;
}
static int spline_fit_run(mvus_spline_fit& S, double s, int32_t* n_out, double* t_out, double* c_out, double* fp_out, int32_t* ier_out) {
  constexpr int k = 3, k1 = 4, k2 = 5, nmin = 8, maxit = 20;
  constexpr double tol = 0.001;
  const int64_t m = S.m;
  const double* u = S.hu.data();
  if (!n_out || !t_out || !c_out || !(s > 0.0) || !std::isfinite(s)) { g_create_error = "spline_smooth: bad arguments (m > 3 samples, s > 0)"; return MVUS_E_INVALID; }
  const bool timing = std::getenv("MVUS_FIT_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t_begin = now();
  int passes = 0;
  const int nest = (int)m + 2 * k, nmax = (int)m + k1;
  const auto t_checked = now();
  auto t_ready = t_checked, t_fitted = t_checked;
  try {
    CallBuffers& cb = S.cb;
    const double* du = S.du;
    const double* dX = S.dX;
    int32_t* span = S.span;
    double* q = S.q;
    double* term = S.term;
    double* tot_part = S.tot_part;
    double* fp_part = S.fp_part;
    // Everything indexed by knots is sized by a CAPACITY that grows with the knot count (x4, up to FITPACK's nest = m + 6), not
    // by nest: the trajectories traj_to_spline fits are 50x oversampled (560k samples for ~600 knots), and allocating and
    // freeing ~40 arrays of nest doubles (430 MB with the double-double set) cost 70 of the 77 ms of such a fit
    long long*& first = S.first;
    double *&cd = S.cd, *&td = S.td, *&bd = S.bd, *&out = S.out;          // out: [0] sum diag(L), [1] f_p, [2] min diag(L), [3] max diag(L), [4..] residual per span
    int* fail = S.fail;
    FitWork<double>& w1 = S.w1;
    FitWork<dd>& w2 = S.w2;
    w1.penalty = false; w2.penalty = false;              // (the work arrays outlive a fit; what they hold does not)
    bool precise = false;                                  // double-double from the first ill-conditioned pass on
    bool diag_natural = true;                              // out[0] of the last least-squares pass is the sum FITPACK forms (see fit_band_solve)
    int lsq_dd_n = -1;                                     // knot count whose normal equations w2 holds
    MVUS_HIP(hipMemsetAsync(fail, 0, sizeof(int), cb.st));
    std::vector<double> t, fpint, host, b;
    std::vector<int> nrdata;
    size_t& cap = S.cap;
    auto ensure = [&](size_t need) {                      // between passes only: the device arrays hold nothing that outlives a pass
      if (need <= cap || cap >= (size_t)nest) return;     // (nest = m + 6 knots is all FITPACK can ever ask for: nothing to grow to)
      size_t c = std::max<size_t>(cap, 1024);
      while (c < need) c *= 4;
      cap = std::min<size_t>(c, (size_t)nest);
      first = cb.get<long long>(cap + 1); cd = cb.get<double>(3 * cap); td = cb.get<double>(cap); bd = cb.get<double>(5 * cap);
      out = cb.get<double>(cap + 4);
      w1 = FitWork<double>(); w2 = FitWork<dd>(); lsq_dd_n = -1;
      w1.alloc(cb, cap);
      if (t.size() < cap) { t.resize(cap, 0.0); fpint.resize(cap, 0.0); nrdata.resize(cap, 0); }
      if (host.size() < cap + 4) host.resize(cap + 4, 0.0);
    };
    ensure(nmin + 16);
    if (t.size() < cap) { t.resize(cap, 0.0); fpint.resize(cap, 0.0); nrdata.resize(cap, 0); }      // (capacity kept from an earlier fit of this session)
    if (host.size() < cap + 4) host.resize(cap + 4, 0.0);
    if (timing) { MVUS_HIP(hipStreamSynchronize(cb.st)); t_ready = now(); }
    const double ub = u[0], ue = u[m - 1], acc = tol * s;
    int n = nmin, nplus = 0, ier = 0, nrint = 1, failed = 0;
    double fpold = 0.0, fp0 = 0.0, fp = 0.0, p = -1.0;
    nrdata[0] = (int)m - 2;
    auto blocks = fit_blocks;
    auto residual = [&](int ncoef, bool spans, int nspan) {            // c -> f_p (and the per-span residuals), fetched
      hipLaunchKernelGGL(k_fit_residual, blocks(m), dim3(256), 0, cb.st, (long long)m, ncoef, span, q, dX, cd, term);
      if (m > 8192) {                                                 // two stages (still one fixed order)
        const int nbt = (int)std::min<long long>(1024, (m + 2047) / 2048);
        hipLaunchKernelGGL(k_fit_total_partial, dim3(nbt), dim3(256), 0, cb.st, (long long)m, term, tot_part);
        hipLaunchKernelGGL(k_fit_total, dim3(1), dim3(256), 0, cb.st, (long long)nbt, tot_part, out + 1);
      } else {
        hipLaunchKernelGGL(k_fit_total, dim3(1), dim3(256), 0, cb.st, (long long)m, term, out + 1);
      }
      if (spans) {
        const int nslice = fit_slices(nspan);
        hipLaunchKernelGGL(k_fit_fpint, dim3(nspan, nslice), dim3(256), 0, cb.st, nspan, first, term, out + 4, fp_part);
        if (nslice > 1) hipLaunchKernelGGL(k_fit_fpint_final, blocks(nspan), dim3(256), 0, cb.st, nspan, nslice, first, term, fp_part, out + 4);
      }
      MVUS_HIP(hipGetLastError());
      MVUS_HIP(hipMemcpyAsync(host.data(), out, sizeof(double) * (4 + (spans ? nspan : 0)), hipMemcpyDeviceToHost, cb.st));
      MVUS_HIP(hipMemcpyAsync(&failed, fail, sizeof(int), hipMemcpyDeviceToHost, cb.st));
      MVUS_HIP(hipStreamSynchronize(cb.st));
    };
    // a pass in fp64; when its Cholesky breaks down or the factor's diagonal spans more than four decades (cond(A^T A) >= 1e8)
    // the pass is repeated in double-double, and so is every later pass of this call
    auto ill = [&] { return failed != 0 || !(host[3] <= 1e4 * host[2]); };
    auto solve = [&](int ncoef, int nrint_, int n8, bool smoothing, double pinv) {
      for (int attempt = 0; attempt < 2; ++attempt) {
        if (!precise) {
          if (!smoothing) diag_natural = fit_lsq_pass<double>(cb.st, w1, (long long)m, first, q, dX, ncoef, nrint_, cd, out, fail);
          else fit_smooth_pass<double>(cb.st, w1, ncoef, n8, bd, pinv, cd, out, fail);
        } else {
          w2.alloc(cb, cap);
          if (lsq_dd_n != n) { diag_natural = fit_lsq_pass<dd>(cb.st, w2, (long long)m, first, q, dX, ncoef, nrint_, cd, out, fail); lsq_dd_n = n; }
          if (smoothing) fit_smooth_pass<dd>(cb.st, w2, ncoef, n8, bd, pinv, cd, out, fail);
        }
        residual(ncoef, !smoothing, nrint_);
        ++passes;
        if (std::getenv("MVUS_DEBUG")) std::fprintf(stderr, "spline_smooth: n=%d %s %s  diag(L) %.3e..%.3e  fp %.6e  fail %d\n", n, smoothing ? "smooth" : "lsq",
                                                    precise ? "dd" : "fp64", host[2], host[3], host[1], failed);
        if (precise) {                                    // floored pivots are accepted here (see k_band_solve)
          if (!std::isfinite(host[1])) throw HipError{"spline_smooth: a banded system is not positive definite", MVUS_E_NUMERIC};
          MVUS_HIP(hipMemsetAsync(fail, 0, sizeof(int), cb.st));
          return;
        }
        if (!ill()) return;
        precise = true;
        MVUS_HIP(hipMemsetAsync(fail, 0, sizeof(int), cb.st));
      }
    };
    int ncoef = 0;
    for (;;) {                                             // fppara: do 200 iter = 1, m
      ensure((size_t)n + 16);
      if (n == nmin) ier = -2;
      nrint = n - nmin + 1;
      ncoef = n - k1;
      for (int j = 0; j < k1; ++j) { t[j] = ub; t[n - 1 - j] = ue; }
      MVUS_HIP(hipMemcpyAsync(td, t.data(), sizeof(double) * n, hipMemcpyHostToDevice, cb.st));
      MVUS_HIP(hipStreamSynchronize(cb.st));               // t is modified on the host below
      hipLaunchKernelGGL(k_fit_basis, blocks(m), dim3(256), 0, cb.st, (long long)m, du, td, ncoef, span, q);
      hipLaunchKernelGGL(k_fit_first, blocks(nrint + 1), dim3(256), 0, cb.st, (long long)m, du, td, nrint, first);
      solve(ncoef, nrint, 0, false, 0.0);
      fp = host[1];
      if (ier == -2) fp0 = fp;
      double fpms = fp - s;
      if (std::fabs(fpms) < acc) break;
      if (fpms < 0.0) {
        if (ier == -2) break;                              // the least-squares polynomial is acceptable
        // ---- part 2: the smoothing spline, F(p) = s ----
        fitpack::fpdisc(t, n, b);
        const int n8 = n - nmin;
        MVUS_HIP(hipMemcpyAsync(bd, b.data(), sizeof(double) * b.size(), hipMemcpyHostToDevice, cb.st));
        double p1 = 0.0, f1 = fp0 - s, p3 = -1.0, f3 = fpms;
        if (!diag_natural) {                               // the pass above was partitioned: one chain over the same normal equations for sum a(i,1)
          if (precise) fit_lsq_diag<dd>(cb.st, w2, ncoef, out); else fit_lsq_diag<double>(cb.st, w1, ncoef, out);
          MVUS_HIP(hipGetLastError());
          MVUS_HIP(hipMemcpyAsync(host.data(), out, sizeof(double), hipMemcpyDeviceToHost, cb.st));
          MVUS_HIP(hipStreamSynchronize(cb.st));
        }
        p = (double)ncoef / host[0];
        int ich1 = 0, ich3 = 0;
        for (int iter = 1; iter <= maxit; ++iter) {
          solve(ncoef, nrint, n8, true, 1.0 / p);
          fp = host[1];
          fpms = fp - s;
          if (std::fabs(fpms) < acc) break;
          if (iter == maxit) { ier = 3; break; }
          const double p2 = p, f2 = fpms;
          if (ich3 == 0) {
            if (f2 - f3 <= acc) {                          // the initial choice of p is too large
              p3 = p2; f3 = f2;
              p = p * 0.04;
              if (p <= p1) p = p1 * 0.9 + p2 * 0.1;
              continue;
            }
            if (f2 < 0.0) ich3 = 1;
          }
          if (ich1 == 0) {
            if (f1 - f2 <= acc) {                          // the initial choice of p is too small
              p1 = p2; f1 = f2;
              p = p / 0.04;
              if (p3 < 0.0) continue;
              if (p >= p3) p = p2 * 0.1 + p3 * 0.9;
              continue;
            }
            if (f2 > 0.0) ich1 = 1;
          }
          if (f2 >= f1 || f2 <= f3) { ier = 2; break; }
          p = fitpack::fprati(p1, f1, p2, f2, p3, f3);
        }
        if (ier < 0) ier = 0;
        break;
      }
      if (n == nmax) { ier = -1; break; }
      if (n == nest) { ier = 1; break; }
      // ---- more knots ----
      if (ier == 0) {
        int npl1 = nplus * 2;
        const double rn = nplus;
        if (fpold - fp > acc) npl1 = (int)(rn * fpms / (fpold - fp));
        nplus = std::min(nplus * 2, std::max(std::max(npl1, nplus / 2), 1));
      } else {
        nplus = 1;
        ier = 0;
      }
      fpold = fp;
      {                                                    // room for the knots about to be added (host arrays; the device side follows at the top of the loop)
        const size_t need = std::min<size_t>((size_t)nest, (size_t)n + (size_t)nplus + 16);
        if (need > t.size()) { t.resize(need, 0.0); fpint.resize(need, 0.0); nrdata.resize(need, 0); }
      }
      for (int j = 0; j < nrint; ++j) fpint[j] = host[4 + j];
      fitpack::fpknot_batch(u, t, n, fpint, nrdata, nrint, nplus, nmax, nest);
      if (n == nmax) {                                      // fppara label 10: the knots of the interpolating spline
        if (t.size() < (size_t)nest) { t.resize((size_t)nest, 0.0); fpint.resize((size_t)nest, 0.0); nrdata.resize((size_t)nest, 0); }
        int i = k2, j = k / 2 + 2;
        for (int l = 0; l < (int)m - k1; ++l) { t[i - 1] = u[j - 1]; ++i; ++j; }
      }
    }
    t_fitted = now();
    std::vector<double> ch(3 * (size_t)ncoef);
    MVUS_HIP(hipMemcpyAsync(ch.data(), cd, sizeof(double) * 3 * ncoef, hipMemcpyDeviceToHost, cb.st));
    MVUS_HIP(hipStreamSynchronize(cb.st));
    for (int d = 0; d < 3; ++d) for (int j = 0; j < ncoef; ++j) c_out[(size_t)d * nest + j] = ch[(size_t)d * ncoef + j];
    for (int j = 0; j < n; ++j) t_out[j] = t[j];
    *n_out = n;
    if (fp_out) *fp_out = fp;
    if (ier_out) *ier_out = ier;
    if (timing) std::fprintf(stderr, "spline_smooth: m=%lld n=%d passes=%d | input checks %.2f ms, buffers+upload %.2f ms, passes %.2f ms", (long long)m, n, passes,
                             ms(t_begin, t_checked), ms(t_checked, t_ready), ms(t_ready, t_fitted));
  } catch (const HipError& e) {
    g_create_error = e.msg;
    return e.code;
  } catch (const std::exception& e) {          // bad_alloc / length_error on absurd sizes must not cross the C ABI
    g_create_error = e.what();
    return MVUS_E_INVALID;
  }
  if (timing) std::fprintf(stderr, ", total %.2f ms\n", ms(t_begin, now()));
  return MVUS_OK;
}

int mvus_spline_fit_open(int32_t device, int64_t m, const double* u, const double* X, mvus_spline_fit** out) {
  if (!out) { g_create_error = "spline_fit_open: bad arguments"; return MVUS_E_INVALID; }
  *out = nullptr;
  mvus_spline_fit* S = nullptr;
  try { S = new mvus_spline_fit(); } catch (const std::exception& e) { g_create_error = e.what(); return MVUS_E_INVALID; }
  const int rc = spline_fit_open_impl(*S, device, m, u, X);
  if (rc != MVUS_OK) { delete S; return rc; }
  *out = S;
  return MVUS_OK;
}
int mvus_spline_fit_smooth(mvus_spline_fit* S, double s, int32_t* n_out, double* t_out, double* c_out, double* fp_out, int32_t* ier_out) {
  if (!S) { g_create_error = "spline_fit_smooth: no session"; return MVUS_E_INVALID; }
  return spline_fit_run(*S, s, n_out, t_out, c_out, fp_out, ier_out);
}
void mvus_spline_fit_close(mvus_spline_fit* S) { delete S; }

int mvus_spline_smooth(int32_t device, int64_t m, const double* u, const double* X, double s, int32_t* n_out, double* t_out, double* c_out,
                       double* fp_out, int32_t* ier_out) {
  if (!n_out || !t_out || !c_out || !(s > 0.0) || !std::isfinite(s)) { g_create_error = "spline_smooth: bad arguments (m > 3 samples, s > 0)"; return MVUS_E_INVALID; }
  mvus_spline_fit S;
  const int rc = spline_fit_open_impl(S, device, m, u, X);
  if (rc != MVUS_OK) return rc;
  return spline_fit_run(S, s, n_out, t_out, c_out, fp_out, ier_out);
}

/* cv2.solvePnPRansac(objectPoints, imagePoints, K, d, reprojectionError) as Scene.get_camera_pose calls it (pnp.hip.h) */
int mvus_pnp_ransac(int32_t device, int64_t N, const double* X, const double* uv, const double* K, const double* d, double reproj_error,
                    int32_t iterations, uint64_t seed, double* rvec, double* tvec, uint8_t* inliers, int64_t* n_inliers) {
  if (N < 6 || N > (1ll << 30) || !X || !uv || !K || !d || !rvec || !tvec || !(reproj_error > 0.0) || iterations < 1 || iterations > 65536) {
    g_create_error = "pnp_ransac: bad arguments (at least 6 points, reprojection error > 0, 1..65536 iterations)";
    return MVUS_E_INVALID;
  }
  for (int64_t i = 0; i < 3 * N; ++i) if (!std::isfinite(X[i])) { g_create_error = "pnp_ransac: non-finite object point"; return MVUS_E_INVALID; }
  for (int64_t i = 0; i < 2 * N; ++i) if (!std::isfinite(uv[i])) { g_create_error = "pnp_ransac: non-finite image point"; return MVUS_E_INVALID; }
  // the object points are centred and scaled (the direct linear transform is badly conditioned otherwise); a pose (R, t')
  // of the scaled points is the pose (R, sigma t' - R m) of the original ones
  try {
  double m[3] = {0.0, 0.0, 0.0}, sigma = 0.0;
  for (int a = 0; a < 3; ++a) { for (int64_t i = 0; i < N; ++i) m[a] += X[a * N + i]; m[a] /= (double)N; }
  for (int a = 0; a < 3; ++a) for (int64_t i = 0; i < N; ++i) sigma += (X[a * N + i] - m[a]) * (X[a * N + i] - m[a]);
  sigma = std::sqrt(sigma / (3.0 * (double)N));
  if (!(sigma > 0.0)) { g_create_error = "pnp_ransac: all object points coincide"; return MVUS_E_INVALID; }
  std::vector<double> Xc(3 * (size_t)N);
  for (int a = 0; a < 3; ++a) for (int64_t i = 0; i < N; ++i) Xc[a * N + i] = (X[a * N + i] - m[a]) / sigma;
  double Kd[9] = {K[0], K[1], K[2], K[3], d[0], d[1], d[2], d[3], d[4]};
  {
    CallBuffers cb;
    cb.open(device);
    const double* dX = cb.put(Xc.data(), Xc.size());
    const double* duv = cb.put(uv, 2 * (size_t)N);
    const double* dK = cb.put(Kd, 9);
    double* xn = cb.get<double>(2 * (size_t)N);
    double* poses = cb.get<double>(13 * (size_t)iterations);
    int32_t* counts = cb.get<int32_t>((size_t)iterations);
    uint8_t* mask = cb.get<uint8_t>((size_t)N);
    double* pose_d = cb.get<double>(13);
    double* acc_d = cb.get<double>(29);
    const double thr2 = reproj_error * reproj_error;
    hipLaunchKernelGGL(k_pnp_normalise, fit_blocks(N), dim3(256), 0, cb.st, (long long)N, duv, dK, xn);
    hipLaunchKernelGGL(k_pnp_hypotheses, dim3((iterations + 63) / 64), dim3(64), 0, cb.st, iterations, (unsigned long long)seed, (long long)N, dX, xn, poses);
    hipLaunchKernelGGL(k_pnp_score, dim3(iterations), dim3(256), 0, cb.st, (long long)N, dX, duv, dK, poses, thr2, counts);
    MVUS_HIP(hipGetLastError());
    std::vector<int32_t> cnt((size_t)iterations);
    MVUS_HIP(hipMemcpyAsync(cnt.data(), counts, sizeof(int32_t) * iterations, hipMemcpyDeviceToHost, cb.st));
    MVUS_HIP(hipStreamSynchronize(cb.st));
    int best = 0;
    for (int h = 1; h < iterations; ++h) if (cnt[h] > cnt[best]) best = h;          // ties: the first hypothesis
    if (cnt[best] < 6) { g_create_error = "pnp_ransac: no hypothesis is supported by six points (reprojection error too small, or no consistent pose)"; return MVUS_E_NUMERIC; }
    double pose[13];
    MVUS_HIP(hipMemcpyAsync(pose, poses + 13ll * best, sizeof(double) * 13, hipMemcpyDeviceToHost, cb.st));
    hipLaunchKernelGGL(k_pnp_mask, fit_blocks(N), dim3(256), 0, cb.st, (long long)N, dX, duv, dK, poses + 13ll * best, thr2, mask);
    MVUS_HIP(hipStreamSynchronize(cb.st));
    // damped Gauss-Newton on the inliers; the normal equations come from the device, the 6x6 solve is done here
    double acc[29], cand[13], acc2[29];
    auto evaluate = [&](const double* ps, double* out) {
      MVUS_HIP(hipMemcpyAsync(pose_d, ps, sizeof(double) * 13, hipMemcpyHostToDevice, cb.st));
      hipLaunchKernelGGL(k_pnp_normal, dim3(1), dim3(256), 0, cb.st, (long long)N, dX, duv, dK, pose_d, mask, acc_d);
      MVUS_HIP(hipMemcpyAsync(out, acc_d, sizeof(double) * 29, hipMemcpyDeviceToHost, cb.st));
      MVUS_HIP(hipStreamSynchronize(cb.st));
    };
    evaluate(pose, acc);
    double lambda = 1e-3;
    for (int it = 0; it < 100; ++it) {
      double Hm[6][6], g[6], L[6][6], dlt[6];
      int e = 0;
      for (int a = 0; a < 6; ++a) for (int b = 0; b <= a; ++b) { Hm[a][b] = Hm[b][a] = acc[e++]; }
      for (int a = 0; a < 6; ++a) { g[a] = acc[21 + a]; Hm[a][a] += lambda * (Hm[a][a] > 0.0 ? Hm[a][a] : 1.0); }
      bool pd = true;
      for (int j = 0; j < 6 && pd; ++j) {
        double s = Hm[j][j];
        for (int k2 = 0; k2 < j; ++k2) s -= L[j][k2] * L[j][k2];
        if (!(s > 0.0)) { pd = false; break; }
        L[j][j] = std::sqrt(s);
        for (int i = j + 1; i < 6; ++i) { double v = Hm[i][j]; for (int k2 = 0; k2 < j; ++k2) v -= L[i][k2] * L[j][k2]; L[i][j] = v / L[j][j]; }
      }
      if (!pd) { lambda *= 10.0; if (lambda > 1e10) break; continue; }
      for (int i = 0; i < 6; ++i) { double v = -g[i]; for (int k2 = 0; k2 < i; ++k2) v -= L[i][k2] * dlt[k2]; dlt[i] = v / L[i][i]; }
      for (int i = 5; i >= 0; --i) { double v = dlt[i]; for (int k2 = i + 1; k2 < 6; ++k2) v -= L[k2][i] * dlt[k2]; dlt[i] = v / L[i][i]; }
      double dR[9], W[9];
      rodrigues(dlt, dR, W);
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cand[3 * a + b] = dR[3 * a] * pose[b] + dR[3 * a + 1] * pose[3 + b] + dR[3 * a + 2] * pose[6 + b];
      for (int a = 0; a < 3; ++a) cand[9 + a] = pose[9 + a] + dlt[3 + a];
      cand[12] = 1.0;
      evaluate(cand, acc2);
      double step = 0.0;
      for (int a = 0; a < 6; ++a) step = std::max(step, std::fabs(dlt[a]));
      if (acc2[28] == 0.0 && acc2[27] <= acc[27]) {
        const double gain = acc[27] - acc2[27];
        std::memcpy(pose, cand, sizeof(pose));
        std::memcpy(acc, acc2, sizeof(acc));
        lambda = std::max(lambda * 0.1, 1e-12);
        if (step < 1e-13 || gain <= 1e-15 * acc[27]) break;
      } else {
        lambda *= 10.0;
        if (lambda > 1e10 || step < 1e-14) break;
      }
    }
    // back to the scale of the original points
    for (int a = 0; a < 3; ++a) tvec[a] = sigma * pose[9 + a] - (pose[3 * a] * m[0] + pose[3 * a + 1] * m[1] + pose[3 * a + 2] * m[2]);
    rotation_to_rvec(pose, rvec);
    if (inliers) MVUS_HIP(hipMemcpyAsync(inliers, mask, (size_t)N, hipMemcpyDeviceToHost, cb.st));
    MVUS_HIP(hipStreamSynchronize(cb.st));
    if (n_inliers) *n_inliers = cnt[best];
  }
  } catch (const HipError& e) {
    g_create_error = e.msg;
    return MVUS_E_HIP;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return MVUS_E_INVALID;
  }
  return MVUS_OK;
}

}  // extern "C"
