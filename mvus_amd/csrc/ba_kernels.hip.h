// HIP kernels of the BA hot path for gfx950 (MI355X, CDNA4).  fp64 throughout.
//
// Data layout in HBM (all SoA, camera-segmented, detections of one camera in frame order):
//   frame[M] u_raw[M] v_raw[M]            raw detections (common.py:1190)
//   u_obs[M] v_obs[M]                     observed pixel after the one-time undistortion (fixed calibration)
//   f[m]                                  residuals in the reference row order (common.py:476-485)
//   J[(chunk*2*NS + a*NS + k)*256 + lane] Jacobian slot k of row a (0:x, 1:y) of the lane-th detection of a chunk (chunk-major:
//                                         one contiguous 2*NS*2 KB block per workgroup, every store / load of a wavefront
//                                         one contiguous 512-B segment; see j_chunk_offset)
//   span[M]                               first active control point (global index), -1 = row is zero
// Work decomposition: one 256-thread workgroup (4 wavefronts of 64) per chunk of <=256 consecutive
// detections of ONE camera, so the camera's decoded parameters (R, t, K, d, alpha, beta, rs and the
// rotation-derivative matrix W) are wave-uniform and come through the scalar cache into SGPRs;
// consecutive lanes hold consecutive frames, i.e. neighbouring timestamps, so their knot-span
// searches and control-point gathers hit the same L1/L2 lines.
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdlib>
#include <string>

#include "ba_math.h"
#include "ba_solver.h"

namespace mvus {

constexpr int kThreads = 256;

struct HipError { std::string msg; int code = -2; };   // code: the MVUS_E_* value the C ABI returns (-2 = MVUS_E_HIP)

#define MVUS_HIP(expr)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) throw ::mvus::HipError{std::string(#expr) + ": " + hipGetErrorString(e_)}; \
  } while (0)

// roctx ranges around the stages of a BA iteration (residual / linearise / solve / all-reduce) for rocprofv3 --marker-trace and the
// ROCm timeline tools: libroctx64.so is opened at run time when MVUS_ROCTX=1 (no link-time dependency, no cost otherwise)
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    if (!std::getenv("MVUS_ROCTX")) return;
    void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
inline RoctxApi& roctx_api() { static RoctxApi a; return a; }
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* name) : on(roctx_api().push != nullptr) { if (on) roctx_api().push(name); }
  ~RoctxRange() { if (on) roctx_api().pop(); }
  RoctxRange(const RoctxRange&) = delete;
  RoctxRange& operator=(const RoctxRange&) = delete;
};

struct DevProblem {  // trivially copyable: passed to kernels by value
  int C, P, NS, S, calib, undist, rs_free, sync_free, T, N;
  long long M;
  const double *frame, *u_raw, *v_raw, *u_obs, *v_obs, *H, *Kfix, *dfix;
  SplineView sp;
  MotionView mv;
  const int32_t *chunk_cam, *chunk_count;
  const ChunkInfo* chunks;    // the same launch table, one 32-byte record per chunk
  const int32_t* cam_chunk_off;   // [C+1] chunks of camera c are cam_chunk_off[c] .. cam_chunk_off[c+1]
  const long long *chunk_start, *det_off;
  int n_chunks;
  int mot_lo, mot_hi;   // motion rows whose first control point lies in [mot_lo, mot_hi) are evaluated (time shards), the rest are zero rows
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global load
// (vmcnt(0): gfx9 counts loads and stores together), which defeats prefetching across a barrier; kernels that keep
// global loads in flight over LDS-only phases use this instead.  Global data written before it is NOT made visible.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// wave_sum's tree for callers that read the total from lane 0 only: the two steps that cross 16-lane rows go through the LDS
// crossbar like __shfl_down, the four inside row 0 are DPP row shifts (no LDS trip).  Lane 0 holds the same bits as wave_sum's.
__device__ __forceinline__ double wave_sum_lane0(double v);
// The same reductions on the DPP data path (VALU lane shifts inside 16-lane rows, then the two row broadcasts): no trip through
// the LDS crossbar per step.  With 16 wavefronts of one workgroup each reducing five values, the ds_bpermute version of
// k_lm_trial spent 21-25 k cycles in its reductions (MVUS_TRIAL_PROBE); the summation tree differs from wave_sum's, so these are
// used by the LM driver's kernels only, not by anything the TRF + LSMR parity path sums.  Result valid in every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift(double v, double fill) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_lane0(double v) {
  v += __shfl_down(v, 32, 64);
  v += __shfl_down(v, 16, 64);
  v += dpp_shift<0x108, 0xF>(v, 0.0);      // row_shl:8
  v += dpp_shift<0x104, 0xF>(v, 0.0);      // row_shl:4
  v += dpp_shift<0x102, 0xF>(v, 0.0);      // row_shl:2
  v += dpp_shift<0x101, 0xF>(v, 0.0);      // row_shl:1
  return v;
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_shift<0x111, 0xF>(v, 0.0);      // row_shr:1
  v += dpp_shift<0x112, 0xF>(v, 0.0);      // row_shr:2
  v += dpp_shift<0x114, 0xF>(v, 0.0);      // row_shr:4
  v += dpp_shift<0x118, 0xF>(v, 0.0);      // row_shr:8   -> lane 15 of every row holds the row's sum
  v += dpp_shift<0x142, 0xA>(v, 0.0);      // row_bcast:15 into rows 1 and 3
  v += dpp_shift<0x143, 0xC>(v, 0.0);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_dpp(double v) {       // for values >= 0 (fill 0)
  v = fmax(v, dpp_shift<0x111, 0xF>(v, 0.0));
  v = fmax(v, dpp_shift<0x112, 0xF>(v, 0.0));
  v = fmax(v, dpp_shift<0x114, 0xF>(v, 0.0));
  v = fmax(v, dpp_shift<0x118, 0xF>(v, 0.0));
  v = fmax(v, dpp_shift<0x142, 0xA>(v, 0.0));
  v = fmax(v, dpp_shift<0x143, 0xC>(v, 0.0));
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// Decode every camera's parameters once per evaluation (C threads): Rodrigues + derivative matrix.
__global__ void k_cam_states(DevProblem dp, const double* __restrict__ x, CamState* __restrict__ cams) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= dp.C) return;
  CamState s;
  load_cam_state(x, dp.C, c, dp.calib != 0, dp.Kfix, dp.dfix, dp.H[c], s);
  cams[c] = s;
}

// Stage one camera state into LDS with the first sizeof(CamState)/8 lanes.
__device__ __forceinline__ void stage_cam(const CamState* __restrict__ cams, int c, CamState& lds_cam) {
  constexpr int kWords = sizeof(CamState) / sizeof(double);
  const double* src = reinterpret_cast<const double*>(cams + c);
  double* dst = reinterpret_cast<double*>(&lds_cam);
  if (threadIdx.x < kWords) dst[threadIdx.x] = src[threadIdx.x];
  __syncthreads();
}

// One-time undistortion of the observations when calibration is fixed (detection_to_global, common.py:126).
__global__ void k_undistort_fixed(DevProblem dp, double* __restrict__ u_obs, double* __restrict__ v_obs) {
  const int chunk = blockIdx.x;
  const int c = dp.chunk_cam[chunk];
  if ((int)threadIdx.x >= dp.chunk_count[chunk]) return;
  const long long i = dp.chunk_start[chunk] + threadIdx.x;
  const double fx = dp.Kfix[4 * c], fy = dp.Kfix[4 * c + 1], cx = dp.Kfix[4 * c + 2], cy = dp.Kfix[4 * c + 3];
  double xn, yn;
  undistort5<false>((dp.u_raw[i] - cx) / fx, (dp.v_raw[i] - cy) / fy, dp.dfix + 5 * c, xn, yn, nullptr, nullptr);
  u_obs[i] = fx * xn + cx;
  v_obs[i] = fy * yn + cy;
}

// Workgroups are handed to the 8 XCDs round robin (blockIdx % 8), each XCD with its own L2.  xcd_tile maps blockIdx so that
// one XCD works on runs of kXcdRun consecutive tiles (shorter runs for small grids); launch xcd_grid(tiles) workgroups and
// skip tiles >= the real count.
#ifndef MVUS_XCD_RUN
#define MVUS_XCD_RUN 64
#endif
constexpr int kXcdRun = MVUS_XCD_RUN;
MVUS_HD int xcd_run_for(int tiles) { const int r = (tiles + 7) / 8; return r < kXcdRun ? (r > 0 ? r : 1) : kXcdRun; }
MVUS_HD int xcd_grid(int tiles) { const int per = 8 * xcd_run_for(tiles); return (tiles + per - 1) / per * per; }
#if defined(__HIPCC__)
__device__ __forceinline__ int xcd_tile(int tiles) {
  const int run = xcd_run_for(tiles);
  const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;          // q-th workgroup of this XCD
  return ((q / run) * 8 + xcd) * run + q % run;
}
#endif


// Fused per-observation kernel: timestamp -> interval/span search -> de Boor -> R,t -> K -> |residual|
// and (JAC) the 2 x NS analytic Jacobian block.  masked != 0 keeps only the reference pattern (pat0).
#ifndef MVUS_JAC_WAVES
#define MVUS_JAC_WAVES 4
#endif
#ifndef MVUS_JAC_WAVES_CALIB
#define MVUS_JAC_WAVES_CALIB 3        // opt_calib (2 x 30 slots + the K, d tangents): at 4 wavefronts per SIMD (128 VGPRs) it spills 40 B per lane
#endif
// Chunk-major Jacobian: the 2*NS slot rows of one <=256-detection chunk are adjacent in memory,
//     J[(chunk * 2*NS + r) * kThreads + lane],   r = k (x row of slot k) or NS + k (y row),
// so a workgroup writes (and J v / J^T u / the assembly read) ONE contiguous 2*NS*2 KB block instead of 2*NS pieces M*8 bytes
// apart: the bare store pattern gains 8 % at 504k detections and 15 % at 2 M over the slot-major layout even with the XCD-aware
// order (tools/micro/store_bw.hip).  The C ABI keeps the slot-major layout (k_j_export converts on the way out).
template <int NS>
MVUS_HD long long j_chunk_offset(int chunk) { return (long long)chunk * (2 * NS * kThreads); }
inline size_t j_doubles(int NS, int n_chunks) { return (size_t)2 * NS * kThreads * (size_t)(n_chunks > 0 ? n_chunks : 1); }

// The Jacobian is written once and read by other kernels later: non-temporal stores.  With the chunk-major layout they take the
// kernel from 39.4 to 35.2 us when the outputs go to HBM (with the slot-major layout they cost +4 us, with plain stores into a
// cache-resident buffer they are 1 us slower; write-through sc1 stores were slower still): measured in rounds 2-3 with build variants
// (docs/NOTEBOOK.md section G); the shipped form is the only one in the source.
#define MVUS_JSTORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
// Sink of eval_observation_to that stores each value of the 2 x NS block straight to the chunk-major Jacobian
// (masked: spline slots outside the pattern become zeros; an all-zero pattern row stores nothing).
struct JStoreSink {
  static constexpr bool kFactored = false;
  double *Jx, *Jy;
  long long stride;
  int32_t pat;
  int base;
  bool masked, live;
  int32_t ctrl;
  __device__ __forceinline__ void begin(int32_t c) { ctrl = c; live = !(masked && pat < 0); }
  __device__ __forceinline__ double keep(int k, double v) const {
    return (masked && k >= base && !pattern_has(pat, ctrl + (k - base) / 3)) ? 0.0 : v;
  }
  // Jx / Jy are wave-uniform (the chunk's first column of slot 0), lane the 32-bit column inside the chunk: the address is
  // an SGPR base plus a zero-extended VGPR offset, no per-lane 64-bit arithmetic and no address registers per store
  unsigned lane;
  __device__ __forceinline__ void x(int k, double v) { if (live) MVUS_JSTORE(&(Jx + (long long)k * stride)[lane], keep(k, v)); }
  __device__ __forceinline__ void y(int k, double v) { if (live) MVUS_JSTORE(&(Jy + (long long)k * stride)[lane], keep(k, v)); }
};

template <bool CALIB, bool JAC>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(JAC ? (CALIB ? MVUS_JAC_WAVES_CALIB : MVUS_JAC_WAVES) : 4, 8))) void k_observations(DevProblem dp, const CamState* __restrict__ cams,
                                                           const double* __restrict__ x, double* __restrict__ f,
                                                           double* __restrict__ J, int32_t* __restrict__ span,
                                                           const int32_t* __restrict__ pat0, int masked, double* __restrict__ sq_part = nullptr,
                                                           double* __restrict__ clr = nullptr, long long clr_len = 0) {
  constexpr int NS = 3 + (CALIB ? 15 : 6) + 12;
  // JAC: the grid is xcd_grid(n_chunks) workgroups and each XCD works on runs of consecutive chunks (xcd_tile): the kernel
  // is bound by its store stream, and an L2 that writes back runs of consecutive lines of each slot row reaches 12-14 %
  // more of the HBM write bandwidth than eight L2s interleaving 2 KB segments (50.6 -> 44.5 us at 504k detections)
  const int chunk = JAC ? xcd_tile(dp.n_chunks) : (int)blockIdx.x;
  if constexpr (!JAC) {
    // workgroups past the chunks (the LM driver's first evaluation of a solve): they zero the normal-equation storage the
    // linearisation that follows adds into -- a bandwidth-bound pass riding beside a latency-bound kernel instead of a launch of its own
    if (clr != nullptr && chunk >= dp.n_chunks) {
      const long long nb = (long long)gridDim.x - dp.n_chunks;
      for (long long i = (chunk - dp.n_chunks) * (long long)kThreads + threadIdx.x; i < clr_len; i += nb * kThreads) clr[i] = 0.0;
      return;
    }
  }
  if (chunk >= dp.n_chunks) return;
  const ChunkInfo ci = dp.chunks[chunk];                 // wave-uniform: one scalar load
  // the camera state is wave-uniform too: it is read through the scalar cache into SGPRs (34 doubles that would occupy
  // 68 VGPRs of every lane as hoisted LDS broadcasts, and no LDS staging + barrier before the first useful load)
  const CamState& cam = cams[ci.cam];
  if constexpr (!JAC) {
    // residual-only launches of the LM driver: the chunk's sum of squares is left in sq_part[chunk] (the driver needs |f|^2 of
    // every trial point; a separate pass over f for it costs a launch and 8 MB of reads).  Every lane stays for the reduction.
    if (sq_part != nullptr) {
      __shared__ double sq_red[kThreads / 64];
      double sq = 0.0;
      if ((int)threadIdx.x < ci.count) {
        const long long i = ci.start + threadIdx.x;
        const long long a = ci.cam_start, Mc = ci.cam_count;
        const double uo = CALIB ? 0.0 : dp.u_obs[i], vo = CALIB ? 0.0 : dp.v_obs[i];
        const double ur = CALIB ? dp.u_raw[i] : 0.0;
        JStoreSink sink{J, J, kThreads, 0, NS - 12, false, true, -1, threadIdx.x};
        const ObsResult r = eval_observation_to<CALIB, false>(cam, dp.sp, x, dp.undist != 0, dp.rs_free != 0, dp.sync_free != 0, dp.frame[i], ur, dp.v_raw[i], uo, vo, sink);
        f[2 * a + (i - a)] = r.ex;
        f[2 * a + Mc + (i - a)] = r.ey;
        if (span != nullptr) span[i] = r.ctrl;           // (residual-only launches: the span table the fused assembly starts from)
        sq = r.ex * r.ex + r.ey * r.ey;
      }
      sq = wave_sum(sq);
      if ((threadIdx.x & 63) == 0) sq_red[threadIdx.x >> 6] = sq;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) t += sq_red[w];
        sq_part[chunk] = t;
      }
      return;
    }
  }
  if ((int)threadIdx.x >= ci.count) return;
  const long long i = ci.start + threadIdx.x;
  const long long a = ci.cam_start, Mc = ci.cam_count;
  const double uo = CALIB ? 0.0 : dp.u_obs[i], vo = CALIB ? 0.0 : dp.v_obs[i];
  const double ur = CALIB ? dp.u_raw[i] : 0.0;
  // every Jacobian value goes to memory as soon as it exists: the row never sits in registers as a whole
  double* __restrict__ Jc = J + j_chunk_offset<NS>(chunk);          // wave-uniform
  JStoreSink sink{Jc, Jc + NS * kThreads, kThreads, (JAC && masked) ? pat0[i] : 0, NS - 12, JAC && masked != 0, true, -1, threadIdx.x};
  ObsResult r = eval_observation_to<CALIB, JAC>(cam, dp.sp, x, dp.undist != 0, dp.rs_free != 0, dp.sync_free != 0, dp.frame[i], ur, dp.v_raw[i], uo, vo, sink);
  f[2 * a + (i - a)] = r.ex;
  f[2 * a + Mc + (i - a)] = r.ey;
  if (JAC) span[i] = (r.ctrl >= 0 && sink.live) ? r.ctrl : -1;
  else if (span != nullptr) span[i] = r.ctrl;
}

// Reference sparsity pattern of the detection rows at x0 (jac_BA + compute_visibility).
__global__ __launch_bounds__(kThreads) void k_pattern(DevProblem dp, const CamState* __restrict__ cams, int32_t* __restrict__ pat0) {
  __shared__ CamState cam;
  const int chunk = blockIdx.x;
  const int c = dp.chunk_cam[chunk];
  stage_cam(cams, c, cam);
  if ((int)threadIdx.x >= dp.chunk_count[chunk]) return;
  const long long i = dp.chunk_start[chunk] + threadIdx.x;
  pat0[i] = observation_pattern(cam, dp.sp, dp.frame[i], dp.v_raw[i]);
}

// Motion-regulariser rows (error_motion / motion_prior): one thread per sample.
template <bool JAC>
__global__ __launch_bounds__(kThreads) void k_motion(DevProblem dp, const double* __restrict__ x, double* __restrict__ fm,
                                                     double* __restrict__ mJ, int32_t* __restrict__ mctrl, int masked, double* __restrict__ sq_part = nullptr) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (!JAC) {
    if (sq_part != nullptr) {                            // residual-only, with the workgroup's sum of squares (see k_observations)
      __shared__ double sq_red[kThreads / 64];
      double v = 0.0;
      if (j < dp.T) {
        const int key = dp.mv.ctrl[j];
        if (key >= dp.mot_lo && key < dp.mot_hi) { double jrow[36]; int32_t cidx[3]; v = eval_motion_row<false>(dp.mv, x, j, false, jrow, cidx); }
        fm[j] = v;
      }
      double sq = wave_sum(v * v);
      if ((threadIdx.x & 63) == 0) sq_red[threadIdx.x >> 6] = sq;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) t += sq_red[w];
        sq_part[blockIdx.x] = t;
      }
      return;
    }
  }
  if (j >= dp.T) return;
  int32_t cidx[3];
  const int key = dp.mv.ctrl[j];
  if (key < dp.mot_lo || key >= dp.mot_hi) {          // owned by another time shard: a row of zeros here
    fm[j] = 0.0;
    if (JAC) {
#pragma unroll
      for (int k = 0; k < 3; ++k) mctrl[(long long)k * dp.T + j] = -1;
    }
    return;
  }
  if (JAC) {
    // every entry goes to memory as it is produced (zeros first: the row function hands over the non-zero ones) -- the row never
    // sits in a local array (36 doubles + index arithmetic were 304 bytes of scratch per lane)
#pragma unroll
    for (int k = 0; k < 36; ++k) mJ[(long long)k * dp.T + j] = 0.0;
    double* mj = mJ + j;
    const long long T = dp.T;
    fm[j] = eval_motion_row_to<true>(dp.mv, x, j, masked != 0, [&](int k, int q, int d, double v) { mj[(long long)(12 * k + 3 * q + d) * T] = v; }, cidx);
#pragma unroll
    for (int k = 0; k < 3; ++k) mctrl[(long long)k * dp.T + j] = cidx[k];
  } else {
    double jrow[1];
    fm[j] = eval_motion_row<false>(dp.mv, x, j, false, jrow, cidx);
  }
}

__device__ __forceinline__ int cam_col(int C, int P, int c, int k) { return k < 3 ? k * C + c : 3 * C + c * P + (k - 3); }

// the C ABI's slot-major copy of the Jacobian, Jout[r * M + i], from the chunk-major device layout (inspection calls only)
template <int NS>
__global__ __launch_bounds__(kThreads) void k_j_export(DevProblem dp, const double* __restrict__ J, double* __restrict__ Jout) {
  const int chunk = blockIdx.x;
  if ((int)threadIdx.x >= dp.chunk_count[chunk]) return;
  const long long i = dp.chunk_start[chunk] + threadIdx.x;
  const double* __restrict__ Jc = J + j_chunk_offset<NS>(chunk) + threadIdx.x;
  for (int r = 0; r < 2 * NS; ++r) Jout[(long long)r * dp.M + i] = Jc[r * kThreads];
}

// y = J v on the detection rows.  Camera/sync entries of v are staged in LDS once per workgroup.
// row j of the motion regulariser times v
__device__ __forceinline__ double motion_row_times(const DevProblem& dp, const double* __restrict__ mJ, const int32_t* __restrict__ mctrl,
                                                   const double* __restrict__ v, int j) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int g = mctrl[(long long)k * dp.T + j];
    if (g < 0) continue;
    const int x0 = dp.mv.ctrl_x0[g], st = dp.mv.ctrl_stride[g];
    for (int q = 0; q < 4; ++q)
      for (int d = 0; d < 3; ++d) s += mJ[(long long)(12 * k + 3 * q + d) * dp.T + j] * v[x0 + q + d * st];
  }
  return s;
}
// (workgroups past xcd_grid(n_chunks), when mJ is given: the motion rows -- k_motion_jv beside the detection rows instead of
// after them: one launch less in every LSMR iteration of the default solver, which is launch bound at the reference's sizes)
template <int NS>
__global__ __launch_bounds__(kThreads) void k_jv(DevProblem dp, const double* __restrict__ J, const int32_t* __restrict__ span,
                                                 const double* __restrict__ v, double* __restrict__ y, const double* __restrict__ mJ = nullptr,
                                                 const int32_t* __restrict__ mctrl = nullptr, double* __restrict__ ym = nullptr) {
  constexpr int B = NS - 12;
  __shared__ double vc[B];
  if (mJ != nullptr && (int)blockIdx.x >= xcd_grid(dp.n_chunks)) {
    const int j = ((int)blockIdx.x - xcd_grid(dp.n_chunks)) * kThreads + threadIdx.x;
    if (j < dp.T) ym[j] = motion_row_times(dp, mJ, mctrl, v, j);
    return;
  }
    const int chunk = xcd_tile(dp.n_chunks);        // grid = xcd_grid(n_chunks): every XCD streams runs of consecutive chunks of J
  if (chunk >= dp.n_chunks) return;
  const int c = dp.chunk_cam[chunk];
  if (threadIdx.x < B) vc[threadIdx.x] = v[cam_col(dp.C, dp.P, c, threadIdx.x)];
  __syncthreads();
  if ((int)threadIdx.x >= dp.chunk_count[chunk]) return;
  const long long i = dp.chunk_start[chunk] + threadIdx.x;
  const long long a = dp.det_off[c], Mc = dp.det_off[c + 1] - a;
  const int g = span[i];
  const double* __restrict__ Jc = J + j_chunk_offset<NS>(chunk) + threadIdx.x;
  double sx = 0.0, sy = 0.0;
  if (g >= 0) {
#pragma unroll
    for (int k = 0; k < B; ++k) {
      sx += Jc[k * kThreads] * vc[k];
      sy += Jc[(NS + k) * kThreads] * vc[k];
    }
    const int x0 = dp.mv.ctrl_x0[g], st = dp.mv.ctrl_stride[g];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double vv = v[x0 + q + d * st];
        sx += Jc[(B + 3 * q + d) * kThreads] * vv;
        sy += Jc[(NS + B + 3 * q + d) * kThreads] * vv;
      }
  }
  y[2 * a + (i - a)] = sx;
  y[2 * a + Mc + (i - a)] = sy;
}

constexpr int kJtWin = 144;     // control points per chunk window of the J^T u gather

// ---- z = J^T u, DETERMINISTIC two-pass form (what the TRF + LSMR parity solver uses) ---------------------------------
// LSMR on these Jacobians amplifies last-bit differences of its products by ~10x every 1-2 iterations, so a J^T u whose
// fp64 atomics arrive in a different order on every run makes the whole optimiser irreproducible from run to run.  Here
// every output is written by exactly one thread that adds its contributions in a fixed order:
//   pass 1 (k_jtu_partial, one workgroup per chunk): the chunk's B camera sums and the sums of its control-point window
//          (the gather of k_jtu_gather) go to per-chunk buffers, nothing is added to z;
//   pass 2 (k_jtu_reduce, one thread per column of z): camera columns add their camera's chunks in order; a spline
//          column adds, camera by camera and chunk by chunk, the window entries that cover its control point, then the
//          motion rows that touch it (a contiguous row range known in advance).
// Chunks whose spans interleave (a large rolling-shutter coefficient) take an O(256) loop per output instead of the range
// tables -- still ordered; only a chunk whose detections spread over more than kJtWin control points (very sparse
// tracks) falls back to atomics, and raises *nondet so the caller can know.
template <int NS>
__global__ __launch_bounds__(kThreads) void k_jtu_partial(DevProblem dp, const double* __restrict__ J, const int32_t* __restrict__ span,
                                                          const double* __restrict__ u, double* __restrict__ z, double* __restrict__ zc,
                                                          double* __restrict__ zs, int32_t* __restrict__ zg0, int* __restrict__ nondet,
                                                          int32_t* __restrict__ zext = nullptr) {
  constexpr int B = NS - 12;
  constexpr int PS = kThreads + 1;
  __shared__ double part[kThreads / 64][B];
  __shared__ double prod[12 * PS];
  __shared__ int lo[kJtWin], hi[kJtWin], skey[kThreads];
  __shared__ int gmin_s[kThreads / 64];
  __shared__ int bad_s, wide_s, lmax_s;
    const int chunk = xcd_tile(dp.n_chunks);        // grid = xcd_grid(n_chunks): every XCD streams runs of consecutive chunks of J
  if (chunk >= dp.n_chunks) return;
  const ChunkInfo ci = dp.chunks[chunk];
  const bool active = (int)threadIdx.x < ci.count;
  const long long i = ci.start + (active ? threadIdx.x : 0);
  const long long a = ci.cam_start, Mc = ci.cam_count;
  const int g = active ? span[i] : -1;
  const double ux = g >= 0 ? u[2 * a + (i - a)] : 0.0;
  const double uy = g >= 0 ? u[2 * a + Mc + (i - a)] : 0.0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* __restrict__ Jc = J + j_chunk_offset<NS>(chunk) + (active ? threadIdx.x : 0);
  for (int k = threadIdx.x; k < kJtWin; k += kThreads) { lo[k] = 0x7fffffff; hi[k] = 0; }
  if (threadIdx.x == 0) { bad_s = 0; wide_s = 0; lmax_s = -1; }
  skey[threadIdx.x] = g;
  int gm = g >= 0 ? g : 0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gm = min(gm, __shfl_xor(gm, off, 64));
  if (lane == 0) gmin_s[wave] = gm;
#pragma unroll
  for (int k = 0; k < B; ++k) {
    double val = 0.0;
    if (g >= 0) val = Jc[k * kThreads] * ux + Jc[(NS + k) * kThreads] * uy;
    val = wave_sum_lane0(val);
    if (lane == 0) part[wave][k] = val;
  }
  __syncthreads();
  int g0 = gmin_s[0];
#pragma unroll
  for (int w = 1; w < kThreads / 64; ++w) g0 = min(g0, gmin_s[w]);
  if (threadIdx.x < B) {
    double sacc = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) sacc += part[w][threadIdx.x];
    zc[(long long)chunk * B + threadIdx.x] = sacc;
  }
  if (threadIdx.x == 0) { zg0[chunk] = g0; if (zext != nullptr && g0 == 0x7fffffff) zext[chunk] = 0; }
  double* zw = zs + (long long)chunk * (3 * kJtWin);
  if (g0 == 0x7fffffff) return;                       // nothing visible (uniform): pass 2 skips the chunk
  const int l = g - g0;
  double pv[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    pv[e] = g >= 0 ? Jc[(B + e) * kThreads] * ux + Jc[(NS + B + e) * kThreads] * uy : 0.0;
    prod[e * PS + threadIdx.x] = pv[e];
  }
  if (g >= 0) {
    if (l + 3 >= kJtWin) atomicOr(&wide_s, 1);
    else {
      // only the ends of a run of equal span can set the minimum / maximum of their span's index range (LDS atomics serialise
      // per address: a run of r detections cost 2 r of them, now 2)
      const int tid = threadIdx.x;
      if (tid == 0 || skey[tid - 1] != g) atomicMin(&lo[l], tid);
      if (tid == kThreads - 1 || skey[tid + 1] != g) { atomicMax(&hi[l], tid + 1); atomicMax(&lmax_s, l); }
    }
  }
  __syncthreads();
  // do the index ranges of consecutive non-empty spans overlap (detections not in span order)?  Each non-empty span looks for the
  // next one -- up to the last span in use only: scanning the empty tail of the 144-wide window took the last span's thread
  // ~60 dependent LDS reads, 15 of the kernel's 49 us
  if (threadIdx.x < kJtWin && hi[threadIdx.x] > 0) {
    const int last = lmax_s;
    for (int t = threadIdx.x + 1; t <= last; ++t)
      if (hi[t] > 0) { if (lo[t] < hi[threadIdx.x]) atomicOr(&bad_s, 1); break; }
  }
  __syncthreads();
  // how many control points of the window this chunk really reaches (k_jtu_index takes the maximum over the chunks: pass 2 then
  // looks only at chunks that can cover its control point instead of at all whose 144-wide window does)
  if (threadIdx.x == 0 && zext != nullptr) zext[chunk] = wide_s ? 0 : lmax_s + 4;
  if (wide_s) {                                        // detections spread over more control points than the window holds
    if (threadIdx.x == 0) { zg0[chunk] = 0x7fffffff; atomicAdd(nondet, 1); }
    if (g >= 0) {
      const int x0 = dp.mv.ctrl_x0[g], st = dp.mv.ctrl_stride[g];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) unsafeAtomicAdd(&z[x0 + q + d * st], pv[3 * q + d]);
    }
    return;
  }
  for (int o = threadIdx.x; o < 3 * kJtWin; o += kThreads) {
    const int lc = o / 3, d = o % 3;
    double acc = 0.0;
    if (!bad_s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sidx = lc - q;
        if (sidx < 0) continue;
        const int b = lo[sidx], e = hi[sidx];
        for (int t = b; t < e; ++t) acc += prod[(3 * q + d) * PS + t];
      }
    } else {                                           // interleaved spans: every detection, in index order
      for (int t = 0; t < ci.count; ++t) {
        const int q = lc - (skey[t] - g0);
        if (skey[t] >= 0 && q >= 0 && q < 4) acc += prod[(3 * q + d) * PS + t];
      }
    }
    zw[o] = acc;
  }
}

// between the passes (one wavefront per camera): zfill[chunk] = running maximum of the window starts of the camera's chunks
// -- non-decreasing whatever the data, so pass 2 can binary-search it; chunks are time ordered, so zfill exceeds a
// chunk's own start by at most the few spans a rolling-shutter shift can reorder
// bounds[0] = max over the chunks of (running maximum - own window start): how far behind the running maximum a chunk can start;
// bounds[1] = max extent of a chunk's window in control points (zext).  Both zeroed by the caller before the launch.
__global__ __launch_bounds__(64) void k_jtu_index(DevProblem dp, const int32_t* __restrict__ zg0, int32_t* __restrict__ zfill,
                                                  const int32_t* __restrict__ zext = nullptr, int* __restrict__ bounds = nullptr) {
  const int c = blockIdx.x, lane = threadIdx.x;
  int carry = -0x7fffffff, slack = 0, ext = 0;
  for (int base = dp.cam_chunk_off[c]; base < dp.cam_chunk_off[c + 1]; base += 64) {
    const int ch = base + lane;
    const bool in = ch < dp.cam_chunk_off[c + 1];
    const int g0 = in ? zg0[ch] : 0x7fffffff;
    int v = g0 == 0x7fffffff ? -0x7fffffff : g0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(v, off, 64); if (lane >= off) v = max(v, o); }
    v = max(v, carry);
    if (in) zfill[ch] = v;
    if (in && g0 != 0x7fffffff) { slack = max(slack, v - g0); if (zext != nullptr) ext = max(ext, zext[ch]); }
    carry = __shfl(v, 63, 64);
  }
  if (bounds != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { slack = max(slack, __shfl_xor(slack, off, 64)); ext = max(ext, __shfl_xor(ext, off, 64)); }
    if (lane == 0) { atomicMax(&bounds[0], slack); atomicMax(&bounds[1], ext); }
  }
}

// pass 2.  Blocks [0, C): the B camera columns of camera blockIdx.x, each summed over the camera's chunks in order.
// Blocks [C, ..): one wavefront per control point, lane = camera (cameras lane, lane + 64, ...): every lane adds, chunk by
// chunk, the window entries of its camera that cover the control point; the lanes are then combined by a butterfly whose
// pairing is fixed (the same bits every run), lane 0 adds the motion rows (a contiguous, precomputed row range) and writes z.
// first chunk of camera c whose running-max window start is within reach of control point g: what pass 2 finds by binary
// search.  It depends on the Jacobian's spans only, so a run of products with one Jacobian (LSMR) looks it up once.
// (win: no chunk reaches further than win control points past its window start -- kJtWin, or the measured bounds[1])
__device__ __forceinline__ int jtu_first_chunk(const DevProblem& dp, const int32_t* __restrict__ zfill, int c, int g, int win = kJtWin) {
  int lo = dp.cam_chunk_off[c], hi = dp.cam_chunk_off[c + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (zfill[mid] + win > g) hi = mid; else lo = mid + 1;
  }
  return lo;
}
__device__ __forceinline__ int jtu_window(const int* __restrict__ bounds) { return bounds != nullptr ? min(kJtWin, max(1, bounds[1])) : kJtWin; }
__global__ __launch_bounds__(kThreads) void k_jtu_first(DevProblem dp, const int32_t* __restrict__ zfill, int32_t* __restrict__ first,
                                                        const int* __restrict__ bounds = nullptr) {
  const long long e = blockIdx.x * (long long)kThreads + threadIdx.x;
  if (e >= (long long)dp.N * dp.C) return;
  first[e] = jtu_first_chunk(dp, zfill, (int)(e % dp.C), (int)(e / dp.C), jtu_window(bounds));
}
template <int NS>
__global__ __launch_bounds__(kThreads) void k_jtu_reduce(DevProblem dp, const double* __restrict__ zc, const double* __restrict__ zs,
                                                         const int32_t* __restrict__ zg0, const int32_t* __restrict__ zfill, const double* __restrict__ mJ,
                                                         const int32_t* __restrict__ mctrl, const double* __restrict__ um, int motion,
                                                         double* __restrict__ z, const int32_t* __restrict__ first = nullptr,
                                                         const int* __restrict__ bounds = nullptr) {
  constexpr int B = NS - 12;
  if ((int)blockIdx.x < dp.C) {
    const int c = blockIdx.x, k = threadIdx.x;
    if (k >= B) return;
    double acc = 0.0;
    // (sixteen loads in flight, then their adds in chunk order: the same sum, a sixteenth of the memory round trips)
    const int ch1 = dp.cam_chunk_off[c + 1];
    for (int ch0 = dp.cam_chunk_off[c]; ch0 < ch1; ch0 += 16) {
      double t[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] = ch0 + q < ch1 ? zc[(long long)(ch0 + q) * B + k] : 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) if (ch0 + q < ch1) acc += t[q];
    }
    z[cam_col(dp.C, dp.P, c, k)] += acc;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int g = ((int)blockIdx.x - dp.C) * (kThreads / 64) + (threadIdx.x >> 6);
  if (g >= dp.N) return;
  double acc[3] = {0.0, 0.0, 0.0};
  const int mrow_lo = motion ? dp.mv.row_lo[g] : 0, mrow_hi = motion ? dp.mv.row_hi[g] : 0;      // (fetched beside the cameras' first loads)
  // a chunk starts at most bounds[0] control points below the running maximum of the starts: once that maximum is more than
  // bounds[0] past g, no later chunk starts at or before g (without the measured bound: a whole window)
  // (the MEASURED bound, not clipped to a window: a camera whose detections are not in time order -- nothing in the reference
  // forbids it -- has chunks that start arbitrarily far below the running maximum; the walk is then longer, never wrong)
  const int reach = bounds != nullptr ? bounds[0] : kJtWin;
  for (int c = lane; c < dp.C; c += 64) {
    // first chunk whose running-max window start is within reach of g (everything before ends left of g) ...
    const int end = dp.cam_chunk_off[c + 1];
    const int lo = first ? first[(long long)g * dp.C + c] : jtu_first_chunk(dp, zfill, c, g, jtu_window(bounds));
    // ... then forward until the running maximum is a whole window past g (a later chunk's own start is never that far
    // below the running maximum), adding the covering windows in chunk order
    // (four chunks at a time: their window starts together, then the entries of the covering ones together, then the adds in
    // chunk order -- the same sums as the one-chunk-at-a-time loop with a third of its dependent memory round trips; zfill is
    // non-decreasing, so "the loop has not stopped before chunk ch" is zfill[ch] <= g + kJtWin by itself)
    for (int ch0 = lo; ch0 < end; ch0 += 4) {
      int fill[4], g0[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool in = ch0 + q < end;
        fill[q] = in ? zfill[ch0 + q] : 0x7fffffff;
        g0[q] = in ? zg0[ch0 + q] : 0x7fffffff;
      }
      double w[4][3];
      bool use[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int lc = g - g0[q];
        use[q] = fill[q] <= g + reach && g0[q] != 0x7fffffff && lc >= 0 && lc < kJtWin;
        const double* wp = zs + (long long)(ch0 + q) * (3 * kJtWin) + 3 * (use[q] ? lc : 0);
#pragma unroll
        for (int d = 0; d < 3; ++d) w[q][d] = use[q] ? wp[d] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) if (use[q]) { acc[0] += w[q][0]; acc[1] += w[q][1]; acc[2] += w[q][2]; }
      if (!(fill[3] <= g + reach)) break;
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[d] += __shfl_xor(acc[d], off, 64);      // fixed pairing: deterministic
  if (motion) {
    // motion rows: lanes take rows row_lo + lane, + 64, ...; same butterfly
    double ma[3] = {0.0, 0.0, 0.0};
    for (int j = mrow_lo + lane; j < mrow_hi; j += 64) {
      const double uj = um[j];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int cg = mctrl[(long long)k * dp.T + j];
        if (cg >= 0 && g >= cg && g <= cg + 3) {
#pragma unroll
          for (int d = 0; d < 3; ++d) ma[d] += mJ[(long long)(12 * k + 3 * (g - cg) + d) * dp.T + j] * uj;
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ma[d] += __shfl_xor(ma[d], off, 64);
      acc[d] += ma[d];
    }
  }
  if (lane < 3) z[dp.mv.ctrl_x0[g] + lane * dp.mv.ctrl_stride[g]] += (lane == 0 ? acc[0] : (lane == 1 ? acc[1] : acc[2]));
}

__global__ __launch_bounds__(kThreads) void k_motion_jv(DevProblem dp, const double* __restrict__ mJ, const int32_t* __restrict__ mctrl,
                                                        const double* __restrict__ v, double* __restrict__ ym) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= dp.T) return;
  ym[j] = motion_row_times(dp, mJ, mctrl, v, j);
}

// Scene.remove_outliers (common.py:709-713): keep = sqrt(ex^2 + ey^2) < thres.  Explicit round-to-nearest
// multiplies/adds (no FMA contraction) so the integer result is the one numpy computes from the same ex, ey.
__global__ __launch_bounds__(kThreads) void k_outlier_mask(DevProblem dp, const double* __restrict__ f, double thres,
                                                           uint8_t* __restrict__ keep) {
  const int chunk = blockIdx.x;
  const int c = dp.chunk_cam[chunk];
  if ((int)threadIdx.x >= dp.chunk_count[chunk]) return;
  const long long i = dp.chunk_start[chunk] + threadIdx.x;
  const long long a = dp.det_off[c], Mc = dp.det_off[c + 1] - a;
  const double ex = f[2 * a + (i - a)], ey = f[2 * a + Mc + (i - a)];
  const double e = __dsqrt_rn(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)));
  keep[i] = e < thres ? 1 : 0;
}

// ---- small vector kernels (n- and m-sized) -------------------------------------------------------
__global__ void k_axpby(long long len, double a, const double* x, double b, const double* y, double* out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x)
    out[i] = a * x[i] + b * y[i];
}
__global__ void k_mul(long long len, const double* x, const double* y, double* out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x)
    out[i] = x[i] * y[i];
}
// grid = xcd_grid(tiles) workgroups; every pass of the grid-stride loop covers tiles * 256 consecutive elements, of which each
// XCD writes runs of consecutive 2 KB pieces (xcd_tile)
__global__ void k_fill(long long len, double v, double* out, int tiles) {
  const int t = xcd_tile(tiles);
  if (t >= tiles) return;
  for (long long i = t * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)tiles * blockDim.x) out[i] = v;
}
// deterministic two-stage dot product: per-workgroup partials, then one workgroup sums them
__global__ __launch_bounds__(kThreads) void k_dot_partial(long long len, const double* __restrict__ a, const double* __restrict__ b,
                                                          double* __restrict__ partials) {
  __shared__ double red[kThreads / 64];
  double s = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x)
    s += a[i] * b[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += red[w];
    partials[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(kThreads) void k_dot_final(int nb, const double* __restrict__ partials, double* __restrict__ out) {
  __shared__ double red[kThreads / 64];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += partials[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += red[w];
    *out = t;
  }
}

// ---- device-resident LSMR iteration (ba_solver.h: Lsmr::run, unbounded case) ------------------------------------------------
// The scalar state lives in device memory (two copies, read `cur` / write `nxt`, flipped per iteration), so a batch of
// iterations is launched without the three host synchronisations per iteration of the host-driven loop.  Every kernel is a
// no-op once cur->istop is set: iterations launched past convergence cost their launches only.  The element arithmetic and
// the summation trees are those of k_axpby / k_dot_partial / k_dot_final (same grids), so the iterates are the same bits.
__device__ __forceinline__ double dot_final_block(int nb, const double* __restrict__ partials) {     // k_dot_final's tree; blockDim.x == kThreads
  __shared__ double red_f[kThreads / 64];
  __shared__ double tot_f;
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += kThreads) s += partials[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red_f[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += red_f[w];
    tot_f = t;
  }
  __syncthreads();
  return tot_f;
}
__device__ __forceinline__ void dot_partial_store(double s, double* __restrict__ partials) {         // k_dot_partial's tree
  __shared__ double red_p[kThreads / 64];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red_p[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += red_p[w];
    partials[blockIdx.x] = t;
  }
}
// ---- ONE pass over J per LSMR iteration (round 5): u' = J v - alpha u and the first pass of z = J^T u' in the same kernel -------------
// The Golub-Kahan step of LSMR is  u <- (J v - alpha u) / beta,  v <- J^T u - beta v  -- two passes over J (k_jv, k_jtu_partial) with
// the normalisation of u between them.  Here u is kept UNNORMALISED (ut holds u' = beta u, *ubeta its norm): a row's slots are read
// once, give the row's entry of u' = J v - (alpha / ubeta_old) u'_old, which is stored, squared into the workgroup's partial of
// |u'|^2 and multiplied straight back into the row for the per-chunk sums of J^T u' (k_jtu_partial's second half, unchanged);
// k_lsmr_v1 then forms beta = |u'| and v <- (J^T u') / beta - beta v.  J is read once per iteration (183 MB at configs[2]) and three
// launches go (k_jv, k_lsmr_u, k_lsmr_unorm).  The iterates differ from the two-pass form's in the last bits ((J^T u') / beta against
// J^T (u' / beta)), so the form is opt-in: MVUS_LSMR_ONE_PASS=1.  Workgroups past xcd_grid(n_chunks): the motion rows of u'.
template <int NS>
__global__ __launch_bounds__(kThreads) void k_jvjtu(DevProblem dp, const double* __restrict__ J, const int32_t* __restrict__ span,
                                                    const double* __restrict__ v, double* __restrict__ ut, const LsmrScalars* __restrict__ cur,
                                                    const double* __restrict__ ubeta, double* __restrict__ pu, double* __restrict__ z,
                                                    double* __restrict__ zc, double* __restrict__ zs, int32_t* __restrict__ zg0,
                                                    int* __restrict__ nondet, int32_t* __restrict__ zext, const double* __restrict__ mJ,
                                                    const int32_t* __restrict__ mctrl) {
  constexpr int B = NS - 12;
  constexpr int PS = kThreads + 1;
  __shared__ double part[kThreads / 64][B];
  __shared__ double prod[12 * PS];
  __shared__ int lo[kJtWin], hi[kJtWin], skey[kThreads];
  __shared__ int gmin_s[kThreads / 64];
  __shared__ int bad_s, wide_s, lmax_s;
  __shared__ double vc[B];
  if (cur->istop != 0) return;
  const double cu = -cur->alpha / *ubeta;
  if ((int)blockIdx.x >= xcd_grid(dp.n_chunks)) {                    // motion rows
    const int j = ((int)blockIdx.x - xcd_grid(dp.n_chunks)) * kThreads + threadIdx.x;
    double w = 0.0;
    if (j < dp.T) {
      double* um = ut + 2 * (long long)dp.M;
      w = motion_row_times(dp, mJ, mctrl, v, j) + cu * um[j];
      um[j] = w;
    }
    dot_partial_store(w * w, pu);
    return;
  }
  const int chunk = xcd_tile(dp.n_chunks);
  if (chunk >= dp.n_chunks) { dot_partial_store(0.0, pu); return; }
  const ChunkInfo ci = dp.chunks[chunk];
  const bool active = (int)threadIdx.x < ci.count;
  const long long i = ci.start + (active ? threadIdx.x : 0);
  const long long a = ci.cam_start, Mc = ci.cam_count;
  const int g = active ? span[i] : -1;
  const double* __restrict__ Jc = J + j_chunk_offset<NS>(chunk) + (active ? threadIdx.x : 0);
  if (threadIdx.x < B) vc[threadIdx.x] = v[cam_col(dp.C, dp.P, dp.chunk_cam[chunk], threadIdx.x)];
  __syncthreads();
  double ux = 0.0, uy = 0.0;
  {
    double sx = 0.0, sy = 0.0;
    if (g >= 0) {
#pragma unroll
      for (int k = 0; k < B; ++k) {
        sx += Jc[k * kThreads] * vc[k];
        sy += Jc[(NS + k) * kThreads] * vc[k];
      }
      const int x0 = dp.mv.ctrl_x0[g], st = dp.mv.ctrl_stride[g];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const double vv = v[x0 + q + d * st];
          sx += Jc[(B + 3 * q + d) * kThreads] * vv;
          sy += Jc[(NS + B + 3 * q + d) * kThreads] * vv;
        }
    }
    if (active) {
      ux = sx + cu * ut[2 * a + (i - a)];
      uy = sy + cu * ut[2 * a + Mc + (i - a)];
      ut[2 * a + (i - a)] = ux;
      ut[2 * a + Mc + (i - a)] = uy;
    }
    dot_partial_store(ux * ux + uy * uy, pu);
    if (g < 0) { ux = 0.0; uy = 0.0; }                               // (rows without a visible span have no Jacobian entries)
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < kJtWin; k += kThreads) { lo[k] = 0x7fffffff; hi[k] = 0; }
  if (threadIdx.x == 0) { bad_s = 0; wide_s = 0; lmax_s = -1; }
  skey[threadIdx.x] = g;
  int gm = g >= 0 ? g : 0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gm = min(gm, __shfl_xor(gm, off, 64));
  if (lane == 0) gmin_s[wave] = gm;
#pragma unroll
  for (int k = 0; k < B; ++k) {
    double val = 0.0;
    if (g >= 0) val = Jc[k * kThreads] * ux + Jc[(NS + k) * kThreads] * uy;
    val = wave_sum_lane0(val);
    if (lane == 0) part[wave][k] = val;
  }
  __syncthreads();
  int g0 = gmin_s[0];
#pragma unroll
  for (int w = 1; w < kThreads / 64; ++w) g0 = min(g0, gmin_s[w]);
  if (threadIdx.x < B) {
    double sacc = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) sacc += part[w][threadIdx.x];
    zc[(long long)chunk * B + threadIdx.x] = sacc;
  }
  if (threadIdx.x == 0) { zg0[chunk] = g0; if (zext != nullptr && g0 == 0x7fffffff) zext[chunk] = 0; }
  double* zw = zs + (long long)chunk * (3 * kJtWin);
  if (g0 == 0x7fffffff) return;                       // nothing visible (uniform): pass 2 skips the chunk
  const int l = g - g0;
  double pv[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    pv[e] = g >= 0 ? Jc[(B + e) * kThreads] * ux + Jc[(NS + B + e) * kThreads] * uy : 0.0;
    prod[e * PS + threadIdx.x] = pv[e];
  }
  if (g >= 0) {
    if (l + 3 >= kJtWin) atomicOr(&wide_s, 1);
    else {
      // only the ends of a run of equal span can set the minimum / maximum of their span's index range (LDS atomics serialise
      // per address: a run of r detections cost 2 r of them, now 2)
      const int tid = threadIdx.x;
      if (tid == 0 || skey[tid - 1] != g) atomicMin(&lo[l], tid);
      if (tid == kThreads - 1 || skey[tid + 1] != g) { atomicMax(&hi[l], tid + 1); atomicMax(&lmax_s, l); }
    }
  }
  __syncthreads();
  // do the index ranges of consecutive non-empty spans overlap (detections not in span order)?  Each non-empty span looks for the
  // next one -- up to the last span in use only: scanning the empty tail of the 144-wide window took the last span's thread
  // ~60 dependent LDS reads, 15 of the kernel's 49 us
  if (threadIdx.x < kJtWin && hi[threadIdx.x] > 0) {
    const int last = lmax_s;
    for (int t = threadIdx.x + 1; t <= last; ++t)
      if (hi[t] > 0) { if (lo[t] < hi[threadIdx.x]) atomicOr(&bad_s, 1); break; }
  }
  __syncthreads();
  // how many control points of the window this chunk really reaches (k_jtu_index takes the maximum over the chunks: pass 2 then
  // looks only at chunks that can cover its control point instead of at all whose 144-wide window does)
  if (threadIdx.x == 0 && zext != nullptr) zext[chunk] = wide_s ? 0 : lmax_s + 4;
  if (wide_s) {                                        // detections spread over more control points than the window holds
    if (threadIdx.x == 0) { zg0[chunk] = 0x7fffffff; atomicAdd(nondet, 1); }
    if (g >= 0) {
      const int x0 = dp.mv.ctrl_x0[g], st = dp.mv.ctrl_stride[g];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) unsafeAtomicAdd(&z[x0 + q + d * st], pv[3 * q + d]);
    }
    return;
  }
  for (int o = threadIdx.x; o < 3 * kJtWin; o += kThreads) {
    const int lc = o / 3, d = o % 3;
    double acc = 0.0;
    if (!bad_s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sidx = lc - q;
        if (sidx < 0) continue;
        const int b = lo[sidx], e = hi[sidx];
        for (int t = b; t < e; ++t) acc += prod[(3 * q + d) * PS + t];
      }
    } else {                                           // interleaved spans: every detection, in index order
      for (int t = 0; t < ci.count; ++t) {
        const int q = lc - (skey[t] - g0);
        if (skey[t] >= 0 && q >= 0 && q < 4) acc += prod[(3 * q + d) * PS + t];
      }
    }
    zw[o] = acc;
  }
}

// u = A v - alpha u (A v is in tm), partial sums of u.u
__global__ __launch_bounds__(kThreads) void k_lsmr_u(long long m, const double* __restrict__ tm, double* __restrict__ ut,
                                                     const LsmrScalars* __restrict__ cur, double* __restrict__ partials) {
  if (cur->istop != 0) return;
  const double a = cur->one, b = -cur->alpha;
  double s = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
    const double u = a * tm[i] + b * ut[i];
    ut[i] = u;
    s += u * u;
  }
  dot_partial_store(s, partials);
}
// beta = |u| ; u /= beta.  Every workgroup sums the partials itself (same tree, same result); workgroup 0 records beta.
__global__ __launch_bounds__(kThreads) void k_lsmr_unorm(long long m, double* __restrict__ ut, int nb, const double* __restrict__ partials,
                                                         const LsmrScalars* __restrict__ cur, double* __restrict__ beta_out) {
  if (cur->istop != 0) return;
  const double beta = sqrt(dot_final_block(nb, partials));
  if (blockIdx.x == 0 && threadIdx.x == 0) *beta_out = beta;
  if (!(beta > 0)) return;
  const double a = 1.0 / beta, b = cur->zero;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x)
    ut[i] = a * ut[i] + b * ut[i];
}
// ---- the bounded problem (trf_bounds: A = [J diag(D); diag(E)], ba_solver.h Lsmr::run) on the device-resident iteration, round 6.  Each
// kernel evaluates the host-driven loop's elementwise operations in the same order and with the same roundings (mul: one product;
// axpby: a x + b y), so the iterates are that loop's bits.
// out = D v  (the argument of J in A v = J (D v))
__global__ __launch_bounds__(kThreads) void k_lsmr_scale(long long n, const double* __restrict__ D, const double* __restrict__ v, double* __restrict__ out,
                                                         const LsmrScalars* __restrict__ cur) {
  if (cur->istop != 0) return;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = D[i] * v[i];
}
// the n extra rows of u: ub = E v - alpha ub, partial sums of ub.ub (k_dot_partial's tree over grid_for(n) workgroups)
__global__ __launch_bounds__(kThreads) void k_lsmr_ub(long long n, const double* __restrict__ E, const double* __restrict__ v, double* __restrict__ ub,
                                                      const LsmrScalars* __restrict__ cur, double* __restrict__ partials) {
  if (cur->istop != 0) return;
  const double a = cur->one, b = -cur->alpha;
  double s = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double t = E[i] * v[i];
    const double u = a * t + b * ub[i];
    ub[i] = u;
    s += u * u;
  }
  dot_partial_store(s, partials);
}
// beta = sqrt(|u_top|^2 + |u_bottom|^2); both parts divided by it
__global__ __launch_bounds__(kThreads) void k_lsmr_unorm2(long long m, double* __restrict__ ut, int nb, const double* __restrict__ partials, long long n,
                                                          double* __restrict__ ub, int nb2, const double* __restrict__ partials2,
                                                          const LsmrScalars* __restrict__ cur, double* __restrict__ beta_out) {
  if (cur->istop != 0) return;
  double bsq = dot_final_block(nb, partials);
  bsq += dot_final_block(nb2, partials2);
  const double beta = sqrt(bsq);
  if (blockIdx.x == 0 && threadIdx.x == 0) *beta_out = beta;
  if (!(beta > 0)) return;
  const double a = 1.0 / beta, b = cur->zero;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x)
    ut[i] = a * ut[i] + b * ut[i];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    ub[i] = a * ub[i] + b * ub[i];
}
// v = D (J^T u_top) - beta v + E u_bottom, partial sums of v.v; tn is cleared as it is consumed (D / E / ub may be null: factor 1 / no rows)
__global__ __launch_bounds__(kThreads) void k_lsmr_v_de(long long n, double* __restrict__ tn, double* __restrict__ v, const double* __restrict__ D,
                                                        const double* __restrict__ E, const double* __restrict__ ub, const LsmrScalars* __restrict__ cur,
                                                        const double* __restrict__ beta_in, double* __restrict__ partials) {
  const double beta = *beta_in;
  if (cur->istop != 0 || !(beta > 0)) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) tn[i] = 0.0;
    if (cur->istop == 0 && threadIdx.x == 0) partials[blockIdx.x] = 0.0;
    return;
  }
  const double a = cur->one, b = -beta;
  double s = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double t = tn[i];
    if (D != nullptr) t = D[i] * t;
    double w = a * t + b * v[i];
    if (E != nullptr) { const double e = E[i] * ub[i]; w = a * w + a * e; }
    tn[i] = 0.0;
    v[i] = w;
    s += w * w;
  }
  dot_partial_store(s, partials);
}
// v = A^T u - beta v (A^T u is in tn), partial sums of v.v
// (tn is cleared as it is consumed: the next J^T u adds into it without a memset of its own)
__global__ __launch_bounds__(kThreads) void k_lsmr_v(long long n, double* __restrict__ tn, double* __restrict__ v,
                                                     const LsmrScalars* __restrict__ cur, const double* __restrict__ beta_in, double* __restrict__ partials) {
  const double beta = *beta_in;
  if (cur->istop != 0 || !(beta > 0)) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) tn[i] = 0.0;
    if (cur->istop == 0 && threadIdx.x == 0) partials[blockIdx.x] = 0.0;
    return;
  }
  const double a = cur->one, b = -beta;
  double s = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double w = a * tn[i] + b * v[i];
    tn[i] = 0.0;
    v[i] = w;
    s += w * w;
  }
  dot_partial_store(s, partials);
}
// one-pass form: beta = |u'| from the partials of k_jvjtu (every workgroup sums them itself: same tree, same bits; workgroup 0
// records beta and the new norm of ut), then v = (J^T u') / beta - beta v and the partial sums of v.v
__global__ __launch_bounds__(kThreads) void k_lsmr_v1(long long n, double* __restrict__ tn, double* __restrict__ v, const LsmrScalars* __restrict__ cur,
                                                      int nbu, const double* __restrict__ pu, double* __restrict__ beta_out, double* __restrict__ ubeta,
                                                      double* __restrict__ partials) {
  double beta = 0.0;
  if (cur->istop == 0) {
    beta = sqrt(dot_final_block(nbu, pu));
    if (blockIdx.x == 0 && threadIdx.x == 0) { *beta_out = beta; if (beta > 0) *ubeta = beta; }
  }
  if (cur->istop != 0 || !(beta > 0)) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) tn[i] = 0.0;
    if (cur->istop == 0 && threadIdx.x == 0) partials[blockIdx.x] = 0.0;
    return;
  }
  const double a = 1.0 / beta, b = -beta;
  double s = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double w = a * tn[i] + b * v[i];
    tn[i] = 0.0;
    v[i] = w;
    s += w * w;
  }
  dot_partial_store(s, partials);
}
// alpha = |v| ; v /= alpha ; plane rotations ; hbar, x, h updates ; partial sums of x.x.  Every workgroup evaluates the (cheap)
// scalar recurrences itself from the same inputs; workgroup 0 writes the new state.
__global__ __launch_bounds__(kThreads) void k_lsmr_update(long long n, double* __restrict__ v, double* __restrict__ h, double* __restrict__ hbar,
                                                          double* __restrict__ x, int nb, const double* __restrict__ partials_v,
                                                          const LsmrScalars* __restrict__ cur, const double* __restrict__ beta_in,
                                                          LsmrScalars* __restrict__ nxt, double* __restrict__ partials_x) {
  if (cur->istop != 0) { if (blockIdx.x == 0 && threadIdx.x == 0) *nxt = *cur; return; }
  LsmrScalars s = *cur;
  ++s.itn;
  s.beta = *beta_in;
  const double vv = dot_final_block(nb, partials_v);
  bool scale = false;
  if (s.beta > 0) { s.alpha = sqrt(vv); scale = s.alpha > 0; }
  detail::lsmr_rotations(s);
  if (blockIdx.x == 0 && threadIdx.x == 0) *nxt = s;
  const double inv = scale ? 1.0 / s.alpha : 1.0;
  const double one = cur->one, zero = cur->zero;
  double acc = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double vi = v[i];
    if (scale) { vi = inv * vi + zero * vi; v[i] = vi; }
    const double hb = one * h[i] + s.c_hbar * hbar[i];
    hbar[i] = hb;
    const double xi = one * x[i] + s.c_x * hb;
    x[i] = xi;
    h[i] = one * vi + s.c_h * h[i];
    acc += xi * xi;
  }
  dot_partial_store(acc, partials_x);
}
// |x| and the stopping tests
__global__ __launch_bounds__(kThreads) void k_lsmr_test(int nb, const double* __restrict__ partials_x, const LsmrScalars* __restrict__ cur, LsmrScalars* __restrict__ nxt) {
  if (cur->istop != 0) return;                       // (k_lsmr_update has carried the final state over)
  const double xx = dot_final_block(nb, partials_x);
  if (threadIdx.x == 0) {
    LsmrScalars s = *nxt;
    detail::lsmr_tests(s, sqrt(xx));
    nxt->istop = s.istop;
  }
}

// ---- LM driver (ba_schur.h): the two small vector reductions of an iteration --------------------------------
// One element per thread; every workgroup leaves its partial result in `part`, the last one to finish (ticket from an
// atomic counter, which it resets) combines them -- one launch, deterministic, no zero-initialised accumulator.
__device__ __forceinline__ bool last_block_done(unsigned* counter) {
  __shared__ bool last_s;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = atomicAdd(counter, 1u);
    last_s = ticket == gridDim.x - 1;
    if (last_s) *counter = 0u;
  }
  __syncthreads();
  if (last_s) __threadfence();
  return last_s;
}
// projected gradient norm: a component pushing against an active bound does not count
__global__ __launch_bounds__(1024) void k_lm_gnorm(int n, const double* __restrict__ x, const double* __restrict__ lb, const double* __restrict__ ub,
                                                   const double* __restrict__ g, double* __restrict__ out, double* part, unsigned* counter) {
  __shared__ double red[16];
  double gn = 0.0;
  // batches of 8 elements per thread: all loads of a batch are issued before the first is used (one memory round trip per
  // batch instead of one per element -- a single workgroup would otherwise be latency bound)
  for (int i0 = blockIdx.x * 1024 + threadIdx.x; i0 < n; i0 += gridDim.x * 1024 * 8) {
    double gi[8], xi[8], lo[8], hi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * gridDim.x * 1024;
      const bool in = i < n;
      gi[u] = in ? g[i] : 0.0; xi[u] = in ? x[i] : 0.0; lo[u] = in ? lb[i] : -INFINITY; hi[u] = in ? ub[i] : INFINITY;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool blocked = (xi[u] <= lo[u] && gi[u] > 0) || (xi[u] >= hi[u] && gi[u] < 0);
      if (!blocked) gn = fmax(gn, fabs(gi[u]));
    }
  }
  gn = wave_max_dpp(gn);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gn;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < 16; ++w) t = fmax(t, red[w]); part[blockIdx.x] = t; if (gridDim.x == 1) *out = t; }
  if (gridDim.x == 1) return;                              // one workgroup (n <= 128k): no cross-workgroup hand-off, no fences
  if (last_block_done(counter) && threadIdx.x == 0) {
    double t = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b) t = fmax(t, const_cast<volatile double*>(part)[b]);
    *out = t;
  }
}
// trial point x_new = P(x + p) and out = [g.step, step^T D step, |step|^2, |x|^2]; after a failed solve (fail flag)
// x_new = x, a non-finite component of p is not applied, and either way out[0] = NaN: the residual kernels never
// see a non-finite parameter and the driver rejects the trial
__global__ __launch_bounds__(1024) void k_lm_trial(int n, const double* __restrict__ x, const double* __restrict__ p, const double* __restrict__ lb,
                                                   const double* __restrict__ ub, const double* __restrict__ g, const double* __restrict__ D,
                                                   const int* __restrict__ fail, double* __restrict__ x_new, double* __restrict__ out,
                                                   double* __restrict__ gnorm_out, double* part, unsigned* counter, double* __restrict__ x_mirror,
                                                   const double* __restrict__ pn2 = nullptr, double delta = 0.0,
                                                   const double* __restrict__ fail_sum = nullptr, double* __restrict__ fail_out = nullptr) {
  // trust region (mvus_solve_opts.lm_trust_radius): a step longer than delta is cut back to delta along its direction
  const double cut = (pn2 != nullptr && delta > 0.0 && *pn2 > delta * delta) ? delta / sqrt(*pn2) : 1.0;
  if (pn2 != nullptr && blockIdx.x == 0 && threadIdx.x == 0) out[5] = *pn2;      // slot 7 of the driver's scalars
  // also delivers the projected gradient norm of k_lm_gnorm (x, g and the bounds are read here anyway): slot 4 = max
  __shared__ double red[5][16];
  // (time shards: fail_sum = the two failure flags of the solve SUMMED over the ranks -- they travelled with the step; any rank's failure
  // stops every rank's step, and the sums go on to the host with the trial's scalars)
  const bool dead = fail[0] != 0 || (fail_sum != nullptr && fail_sum[0] != 0.0);
  if (fail_sum != nullptr && blockIdx.x == 0 && threadIdx.x < 2) fail_out[threadIdx.x] = fail_sum[threadIdx.x];
  double s[5] = {dead ? __longlong_as_double(0x7ff8000000000000LL) : 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i0 = blockIdx.x * 1024 + threadIdx.x; i0 < n; i0 += gridDim.x * 1024 * 4) {      // batches of 4: loads first, see k_lm_gnorm
    double xv[4], pv[4], gv[4], lo[4], hi[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * gridDim.x * 1024;
      const bool in = i < n;
      xv[u] = in ? x[i] : 0.0; pv[u] = in ? p[i] : 0.0; gv[u] = in ? g[i] : 0.0;
      lo[u] = in ? lb[i] : -INFINITY; hi[u] = in ? ub[i] : INFINITY; dv[u] = in ? D[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * gridDim.x * 1024;
      if (i >= n) break;
      const double xi = xv[u], pi = pv[u], gi = gv[u];
      const bool ok = isfinite(pi);
      const double xn = (dead || !ok) ? xi : fmin(fmax(xi + cut * pi, lo[u]), hi[u]);
      const double st = xn - xi;
      x_new[i] = xn;
      if (x_mirror) x_mirror[i] = xn;                    // mapped pinned host memory: the accepted point needs no download
      s[0] += ok ? gi * st : pi * 0.0; s[1] += st * dv[u] * st; s[2] += st * st; s[3] += xi * xi;
      const bool blocked = (xi <= lo[u] && gi > 0) || (xi >= hi[u] && gi < 0);
      s[4] = fmax(s[4], blocked ? 0.0 : fabs(gi));
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double v = wave_sum_dpp(s[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  {
    const double v = wave_max_dpp(s[4]);
    if ((threadIdx.x & 63) == 0) red[4][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t = threadIdx.x < 4 ? t + red[threadIdx.x][w] : fmax(t, red[4][w]);
    part[blockIdx.x * 5 + threadIdx.x] = t;
    if (gridDim.x == 1) { if (threadIdx.x < 4) out[threadIdx.x] = t; else *gnorm_out = t; }
  }
  if (gridDim.x == 1 || counter == nullptr) return;        // one workgroup, or k_lm_trial_sum follows: nothing to hand over
  if (last_block_done(counter) && threadIdx.x < 5) {
    double t = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b) {
      const double v = const_cast<volatile double*>(part)[b * 5 + threadIdx.x];
      t = threadIdx.x < 4 ? t + v : fmax(t, v);
    }
    if (threadIdx.x < 4) out[threadIdx.x] = t; else *gnorm_out = t;
  }
}

// second stage of k_lm_trial launched over several workgroups with counter == nullptr: the five partials per workgroup -> out
// cams != nullptr: the workgroup also decodes the camera states of the trial point (complete in memory by now) for the residual
// evaluation that follows -- k_cam_states' work without its launch
__global__ __launch_bounds__(64) void k_lm_trial_sum(int nparts, const double* __restrict__ part, double* __restrict__ out, double* __restrict__ gnorm_out,
                                                     DevProblem dp, const double* __restrict__ x_new, CamState* __restrict__ cams) {
  if (cams != nullptr)
    for (int c = threadIdx.x; c < dp.C; c += 64) {
      CamState s;
      load_cam_state(x_new, dp.C, c, dp.calib != 0, dp.Kfix, dp.dfix, dp.H[c], s);
      cams[c] = s;
    }
  if (threadIdx.x >= 5) return;
  double t = 0.0;
  for (int b = 0; b < nparts; ++b) {
    const double v = part[b * 5 + threadIdx.x];
    t = threadIdx.x < 4 ? t + v : fmax(t, v);
  }
  if (threadIdx.x < 4) out[threadIdx.x] = t; else *gnorm_out = t;
}

// ---- MVUS_JAC_FD: scipy's sparse 2-point differences (scipy/optimize/_numdiff.py:628-700) ----------------
// steps: h_j and dx_j = (x_j + h_j) - x_j ; bounds are [0,1] on the rs block when rs_bounds, else infinite
__global__ void k_fd_steps(int n, int C, int rs_bounds, const double* __restrict__ x, double* __restrict__ h, double* __restrict__ dx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const bool bounded = rs_bounds && j >= 2 * C && j < 3 * C;
  const double hj = fd_step(x[j], bounded ? 0.0 : -INFINITY, bounded ? 1.0 : INFINITY);
  h[j] = hj;
  dx[j] = (x[j] + hj) - x[j];
}
__global__ void k_fd_perturb(int n, int g, const double* __restrict__ x, const double* __restrict__ h, const int32_t* __restrict__ groups,
                             double* __restrict__ xg) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) xg[j] = groups[j] == g ? x[j] + h[j] : x[j];
}
// J[i, j] = (f(x + h_group(j))[i] - f(x)[i]) / dx_j for (i, j) in the reference pattern; F[g*m + row]
template <int NS>
__global__ __launch_bounds__(kThreads) void k_fd_fill(DevProblem dp, long long m, const double* __restrict__ f0, const double* __restrict__ F,
                                                      const double* __restrict__ dx, const int32_t* __restrict__ groups,
                                                      const int32_t* __restrict__ pat0, double* __restrict__ J, int32_t* __restrict__ span) {
  constexpr int B = NS - 12;
  const int chunk = blockIdx.x;
  const int c = dp.chunk_cam[chunk];
  if ((int)threadIdx.x >= dp.chunk_count[chunk]) return;
  const long long i = dp.chunk_start[chunk] + threadIdx.x;
  const long long a = dp.det_off[c], Mc = dp.det_off[c + 1] - a;
  const long long rx = 2 * a + (i - a), ry = rx + Mc;
  const int p = pat0[i];
  span[i] = p;
  if (p < 0) return;
  const double fx = f0[rx], fy = f0[ry];
  // the row stores 4 consecutive control points base..base+3 of one spline; the pattern's points are among them
  const int base = pattern_fd_base(p, dp.N, dp.mv.ctrl_x0);
  span[i] = base;
  const int x0 = dp.mv.ctrl_x0[base], st = dp.mv.ctrl_stride[base];
  for (int k = 0; k < NS; ++k) {
    int col = -1;
    if (k < B) { if (!(k == 2 && !dp.rs_free) && !(k < 2 && !dp.sync_free)) col = cam_col(dp.C, dp.P, c, k); }
    else { const int q = (k - B) / 3, d = (k - B) % 3; if (pattern_has(p, base + q)) col = x0 + q + d * st; }
    double jx = 0.0, jy = 0.0;
    if (col >= 0) {
      const double* Fg = F + (long long)groups[col] * m;
      jx = (Fg[rx] - fx) / dx[col];
      jy = (Fg[ry] - fy) / dx[col];
    }
    J[j_chunk_offset<NS>(chunk) + k * kThreads + threadIdx.x] = jx;
    J[j_chunk_offset<NS>(chunk) + (NS + k) * kThreads + threadIdx.x] = jy;
  }
}
__global__ __launch_bounds__(kThreads) void k_fd_fill_motion(DevProblem dp, long long m, const double* __restrict__ f0, const double* __restrict__ F,
                                                             const double* __restrict__ dx, const int32_t* __restrict__ groups,
                                                             double* __restrict__ mJ, int32_t* __restrict__ mctrl) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= dp.T) return;
  for (int k = 0; k < 36; ++k) mJ[(long long)k * dp.T + j] = 0.0;
  mctrl[j] = -1; mctrl[(long long)2 * dp.T + j] = -1;
  if (dp.mv.part[j] < 0) { mctrl[(long long)dp.T + j] = -1; return; }     // row is identically zero
  const int pc = dp.mv.pat[j];
  const int base = pattern_fd_base(pc, dp.N, dp.mv.ctrl_x0);
  mctrl[(long long)dp.T + j] = base;
  const long long row = 2 * dp.M + j;
  const int x0 = dp.mv.ctrl_x0[base], st = dp.mv.ctrl_stride[base];
  for (int q = 0; q < 4; ++q) {
    if (!pattern_has(pc, base + q)) continue;
    for (int d = 0; d < 3; ++d) {
      const int col = x0 + q + d * st;
      mJ[(long long)(12 + 3 * q + d) * dp.T + j] = (F[(long long)groups[col] * m + row] - f0[row]) / dx[col];
    }
  }
}

// ---- in-place removal of outliers (stable stream compaction per chunk) ----------------------------------------
__global__ __launch_bounds__(kThreads) void k_compact_count(DevProblem dp, const uint8_t* __restrict__ keep, int32_t* __restrict__ counts) {
  const int chunk = blockIdx.x;
  const bool k = (int)threadIdx.x < dp.chunk_count[chunk] && keep[dp.chunk_start[chunk] + threadIdx.x];
  const int c = __syncthreads_count(k ? 1 : 0);
  if (threadIdx.x == 0) counts[chunk] = c;
}
__global__ __launch_bounds__(kThreads) void k_compact_scatter(DevProblem dp, const uint8_t* __restrict__ keep, const long long* __restrict__ out_off,
                                                              double* __restrict__ frame, double* __restrict__ u_raw, double* __restrict__ v_raw,
                                                              double* __restrict__ u_obs, double* __restrict__ v_obs) {
  __shared__ int wave_cnt[kThreads / 64];
  const int chunk = blockIdx.x;
  const bool act = (int)threadIdx.x < dp.chunk_count[chunk];
  const long long i = dp.chunk_start[chunk] + (act ? threadIdx.x : 0);
  const bool k = act && keep[i];
  const unsigned long long bal = __ballot(k);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = __popcll(bal);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wave_cnt[w];
  if (k) {
    const long long o = out_off[chunk] + base + __popcll(bal & ((1ull << lane) - 1ull));
    frame[o] = dp.frame[i]; u_raw[o] = dp.u_raw[i]; v_raw[o] = dp.v_raw[i]; u_obs[o] = dp.u_obs[i]; v_obs[o] = dp.v_obs[i];
  }
}

}  // namespace mvus
