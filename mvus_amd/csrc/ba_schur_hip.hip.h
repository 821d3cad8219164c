// HIP implementation of the `Schur` concept of ba_schur.h: on-device assembly of the normal equations
// (camera blocks, block-banded spline block, cross block, gradient) from the slot Jacobian, and the
// damped solve by elimination of the spline block.
//
// Layout (one packed buffer `NE`; observation shards sum all of it, time shards only its head):
//   A  [C][B][B]        camera diagonal blocks, B = 3+P (alpha, beta, rs, camera params)
//   gc [C][B]           camera part of g = J^T f
//   (time shards only: the halo exchange buffer, diag(H) and g in x order -- the part summed over the ranks)
//   Cb [N][W][3][3]     Cb[g][w] = H[3g.., 3(g+w)..]: upper block band of the spline block, W >= 4
//   gs [3N]             spline part of g, internal order 3*ctrl + xyz
//   Et [C][3N][B]       cross block, camera-major: the control points a half chunk touches are one contiguous run of
//                       doubles per camera, so its flush is coalesced; k_build_rhs transposes it into the row-major
//                       [3N][C*B] right-hand-side / GEMM operand
// N here is the number of control points of the slice this handle holds: all of them, or -- time shard -- the owned
// range plus a halo on each side.
//
// Solve chain (solve_async): k_build_rhs (right-hand sides + damped band pack + diag/gradient in x order) -> interiors (k_part_cholesky, k_part_solve) -> separator system
// (k_part_reduce, k_sep_bcr_level / k_sep_bcr_tail, k_sep_bcr_rhs | sequential k_sep_factor, k_sep_rhs) -> k_part_back -> Schur product
// on the fp64 matrix cores (k_schur_gemm, k_schur_finish) -> block Gauss-Jordan on the reduced camera system (k_gj_step)
// -> k_back_substitute.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "ba_kernels.hip.h"
#include "ba_partition.h"
#include "ba_schur.h"

namespace mvus {

constexpr int kNWin = 64;   // control points covered by a workgroup's LDS accumulation window

// N, N3 and every control-point index inside the kernels are LOCAL to this handle's slice, which starts at global
// control point row0 (0 and the whole spline unless the handle is a time shard)
struct NEView {
  double *A, *gc, *Cb, *gs, *Et;
  double* Apart;         // [assembly workgroups][8 wavefronts][(B+1)(B+2)/2] lower triangle of [camera slots; f] [..]^T, partial (k_cam_block_reduce sums them)
  int C, B, CB, N, N3, W;
  int row0;
  int* err;              // set when a row reaches outside the slice
};

}  // namespace mvus
#include "ba_assemble_win.hip.h"
namespace mvus {

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_down(v, off, 64));
  return v;
}

// ---- assembly of the detection rows ------------------------------------------------------------------------
// One workgroup per HALF chunk (<=128 consecutive detections of one camera).  The 2*NS Jacobian slots and the two
// residuals of those detections are staged in LDS (row stride padded to 129 doubles).  Detections of one knot span
// touch the same four control points and (sorted by time) occupy one contiguous index range.
//   1. accumulate: thread (range, xyz d) forms, in registers, the outer products of its range -- 4 control points x
//      (B camera columns + the gradient) for the cross block and the 10 control-point pairs x 3 coordinates of the
//      spline band -- reading LDS ~0.4 times per FMA (a plain one-output-per-thread gather reads it twice per FMA and
//      was LDS bound); the camera block is split four ways over the detections and summed with two shuffles;
//   2. flush: the staged Jacobian is dead, its LDS is reused for the per-range partial blocks; every output entry is
//      then owned by ONE thread, which sums the (at most four) ranges that touch its control point and issues one
//      global fp64 atomic, consecutive threads on consecutive addresses.  (Measured: the L2 retires ~35 G atomic
//      cache-line transactions/s, so scattered per-range atomics cost 1 ms here while coalesced ones are free; LDS
//      fp64 atomics cost ~8 cycles per lane and are not used at all.)
// When a large rolling-shutter coefficient reorders the time stamps (spans interleave in index order) the staged
// columns are first rank-sorted by span in LDS.  Half chunks with more than kGaMaxR ranges (very sparse detections)
// take the slow path: per-range atomics straight from the registers.
// FUSED (the LM path with the analytic Jacobian): the 2 x NS blocks are not read from a materialised J at all -- the first
// two wavefronts evaluate their detection's residual and Jacobian (eval_observation_to, the arithmetic of k_observations)
// straight into the LDS staging area.  Per LM iteration this removes the Jacobian kernel, its 183 MB of stores and the
// 197 MB this kernel used to read back (BASELINE configs[2]); HBM traffic per observation falls to the 32 B of inputs.
#ifndef MVUS_GA_OBS
#define MVUS_GA_OBS 128
#endif
constexpr int kGaObs = MVUS_GA_OBS, kGaStride = kGaObs + 1, kGaThreads = 4 * kGaObs, kGaMaxR = kGaThreads / 6, kGaSplit = 8;
constexpr int kGaShift = kGaObs == 128 ? 7 : 6, kGaParts = 256 / kGaObs;      // detections per workgroup: 128 or 64 (parts of a 256-chunk)
static_assert(kGaObs == 128 || kGaObs == 64, "the assembly tile is 128 or 64 detections");
struct LdsRowSink {          // eval_observation_to sink: slot k of the x / y row of staged column t
  static constexpr bool kFactored = false;
  double* col;               // Js + t
  int ns;
  __device__ __forceinline__ void begin(int32_t) {}
  __device__ __forceinline__ void x(int k, double v) { col[k * kGaStride] = v; }
  __device__ __forceinline__ void y(int k, double v) { col[(ns + k) * kGaStride] = v; }
};
template <int NS, bool FUSED>
__global__ __launch_bounds__(kGaThreads) void k_assemble_spans(DevProblem dp, const double* __restrict__ J, const int32_t* __restrict__ span,
                                                             const double* __restrict__ f, NEView ne, const CamState* __restrict__ cams,
                                                             const double* __restrict__ x) {
  constexpr int B = NS - 12;
  constexpr int kJs = (2 * NS + 2) * kGaStride;
  constexpr int kEp = 4 * 3 * B, kGp = 12, kCp = 10 * 9;        // partial block sizes per range: cross, gradient, band
  constexpr int kRbE = kJs / (kEp + kGp), kRbC = kJs / kCp;     // ranges per flush round
  __shared__ double Js[kJs];                             // rows 0..NS-1: x-row slots, NS..2NS-1: y-row slots, then fx, fy
  __shared__ int meta[kGaObs + kGaObs / 2], rg[kGaObs];
  __shared__ int any_s, sort_s, nr_s;
  // flush: per owned control point of the round -- its index, the first range that reaches it and how many consecutive ranges do
  __shared__ int pt_ctrl[4 * kGaMaxR + 4];
  __shared__ unsigned char pt_rl[4 * kGaMaxR + 4], pt_cnt[4 * kGaMaxR + 4], pt_q[4 * kGaMaxR + 4];
  int* key = meta;                                                   // span per staged column
  unsigned char* rs = reinterpret_cast<unsigned char*>(meta + kGaObs);   // first / one-past-last column of a range
  unsigned char* re = rs + kGaObs;
  const int chunk = blockIdx.x / kGaParts, half = blockIdx.x % kGaParts;
  const int c = dp.chunk_cam[chunk];
  const int cnt = min(kGaObs, dp.chunk_count[chunk] - half * kGaObs);
  if (cnt <= 0) return;
  const long long i0 = dp.chunk_start[chunk] + half * kGaObs;
  const long long a0 = dp.det_off[c], Mc = dp.det_off[c + 1] - a0;
  const int tid = threadIdx.x;
  if (tid == 0) { any_s = 0; sort_s = 0; nr_s = 0; }
  const int t = tid & (kGaObs - 1);                    // detection handled while staging
  constexpr int kRowsPer = (2 * NS + 2 + 3) / 4;
  int g, kt;
  if (FUSED) {
    // every staged value starts as zero (invisible detections, columns past the end); then one lane per detection fills its column
#pragma unroll
    for (int k = 0; k < kRowsPer; ++k) {
      const int r = (tid >> kGaShift) + 4 * k;
      if (r < 2 * NS + 2) Js[r * kGaStride + t] = 0.0;
    }
    lds_barrier();
    g = -1;
    if (tid < kGaObs && t < cnt) {
      constexpr bool CALIB = NS == 30;
      const CamState& cam = cams[c];                    // wave-uniform: scalar loads
      const long long i = i0 + t;
      const double uo = CALIB ? 0.0 : dp.u_obs[i], vo = CALIB ? 0.0 : dp.v_obs[i];
      const double ur = CALIB ? dp.u_raw[i] : 0.0;
      LdsRowSink sink{Js + t, NS};
      const ObsResult r = eval_observation_to<CALIB, true>(cam, dp.sp, x, dp.undist != 0, dp.rs_free != 0, dp.sync_free != 0,
                                                           dp.frame[i], ur, dp.v_raw[i], uo, vo, sink);
      g = r.ctrl;
      if (g >= 0) { Js[(2 * NS) * kGaStride + t] = r.ex; Js[(2 * NS + 1) * kGaStride + t] = r.ey; }
    }
    kt = g >= 0 ? g : 0x7fffffff;
    if (tid < kGaObs) key[tid] = kt;
  } else {
    g = t < cnt ? span[i0 + t] : -1;
    kt = g >= 0 ? g : 0x7fffffff;                      // invisible detections sort last
    if (tid < kGaObs) key[tid] = kt;
    // stage: thread (tid) loads rows tid/128, tid/128+4, ... for detection t
#pragma unroll
    for (int k = 0; k < kRowsPer; ++k) {
      const int r = (tid >> kGaShift) + 4 * k;
      if (r >= 2 * NS + 2) break;
      double v = 0.0;
      if (g >= 0) {
        if (r < 2 * NS) v = J[j_chunk_offset<NS>(chunk) + r * kThreads + half * kGaObs + t];
        else v = f[2 * a0 + (r - 2 * NS) * Mc + (i0 + t - a0)];
      }
      Js[r * kGaStride + t] = v;
    }
  }
  lds_barrier();
  if (FUSED) kt = key[t];                                // the key of column t, for the threads that did not evaluate it
  if (tid < kGaObs) {
    if (g >= 0) any_s = 1;
    if (tid > 0 && kt < key[tid - 1]) sort_s = 1;
  }
  lds_barrier();
  if (!any_s) return;      // nothing visible (uniform)
  if (sort_s) {
    int pos = 0;
    for (int u = 0; u < kGaObs; ++u) { const int ku = key[u]; pos += (ku < kt) || (ku == kt && u < t); }
    double tmp[kRowsPer];                                // this thread's rows: (tid >> 7) + 4k
#pragma unroll
    for (int k = 0; k < kRowsPer; ++k) { const int r = (tid >> kGaShift) + 4 * k; tmp[k] = r < 2 * NS + 2 ? Js[r * kGaStride + t] : 0.0; }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < kRowsPer; ++k) { const int r = (tid >> kGaShift) + 4 * k; if (r < 2 * NS + 2) Js[r * kGaStride + pos] = tmp[k]; }
    if (tid < kGaObs) key[pos] = kt;
    lds_barrier();
  }
  // ranges of equal span, in span order (slot = number of range starts before this one, from the wave ballots)
  {
    const int ks = tid < kGaObs ? key[tid] : 0x7fffffff;
    // a run of equal span longer than kGaSplit columns (dense timelines: many detections per knot span) is cut into
    // pieces of kGaSplit so that the outer products below are spread over more threads
    int back = 0;
    if (tid < kGaObs && ks != 0x7fffffff) while (back < tid && key[tid - back - 1] == ks) ++back;
    const bool start = tid < kGaObs && ks != 0x7fffffff && back % kGaSplit == 0;
    const unsigned long long mask = __ballot(start);
    if (tid == 0) nr_s = __popcll(mask);                 // starts in the first wavefront
    lds_barrier();
    if (start) {
      int e = tid + 1;
      while (e < kGaObs && key[e] == ks && e - tid < kGaSplit) ++e;
      const int slot = (tid >= 64 ? nr_s : 0) + __popcll(mask & ((1ull << (tid & 63)) - 1ull));
      const int kl = ks - ne.row0;                       // local control point; a time shard must hold all four
      const bool inr = kl >= 0 && kl + 3 < ne.N;
      if (!inr) atomicOr(ne.err, 1);
      rs[slot] = (unsigned char)tid; re[slot] = (unsigned char)(inr ? e : tid); rg[slot] = inr ? kl : 0;
    }
    lds_barrier();
    if (tid == 64) nr_s += __popcll(mask);
  }
  lds_barrier();
  const int nr = nr_s;
  const double* fxs = Js + (2 * NS) * kGaStride;
  const double* fys = fxs + kGaStride;
  // camera block (lower triangle) and camera gradient on the fp64 matrix cores: G = R R^T with R = [camera slots; f]
  // ((B+1) x 2*128, x rows then y rows).  For v_mfma_f64_16x16x4 the A fragment (lane l: R[l&15][k0 + (l>>4)]) IS the
  // B fragment of R^T, so one LDS read feeds both operands.  The LAST wavefront does all of it (64 MFMAs per tile pair)
  // while the others go on to the outer products below, where it would have been idle.
  {
    // every wavefront takes 1/8 of the k-steps (one wavefront doing all 64 matrix-core instructions shared its SIMD with
    // three busy wavefronts and finished last: 13k cycles while the others needed 6k and waited) and leaves its own partial
    // block; k_cam_block_reduce adds the 8 x (workgroups of the camera) partials
    using d4v = __attribute__((ext_vector_type(4))) double;
    constexpr int TI = (B + 1 + 15) / 16;                 // 16-row tiles of R
    constexpr int kWaves = kGaThreads / 64, kStepsPerWave = (kGaObs / 4) / kWaves;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    d4v cacc[TI][TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TI; ++j) cacc[i][j] = d4v{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int xy = 0; xy < 2; ++xy)
#pragma unroll
      for (int kk = 0; kk < kStepsPerWave; ++kk) {
        const int u = (wave * kStepsPerWave + kk) * 4 + lk;
        double a[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          // ONE unconditional LDS read per fragment (rows past B read row B and are zeroed by a select): a branch per case
          // serialises the unrolled reads
          const int row = 16 * i + lr;
          const int src = row < B ? xy * NS + row : 2 * NS + xy;
          const double v = Js[src * kGaStride + u];
          a[i] = row <= B ? v : 0.0;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) cacc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], a[j], cacc[i][j], 0, 0, 0);
      }
    double* mine = ne.Apart + ((long long)blockIdx.x * kWaves + wave) * ((B + 1) * (B + 2) / 2);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ra = 16 * i + lk + 4 * r, rb = 16 * j + lr;      // C/D layout: row = (lane>>4) + 4 reg, col = lane&15
          if (ra > B || rb > ra) continue;
          mine[ra * (ra + 1) / 2 + rb] = cacc[i][j][r];              // plain stores: no contention on the camera's few lines
        }
  }
  const bool fast = nr <= kGaMaxR;                        // uniform
  // one register file for both roles -- E role: EA(q, k) cross block + gradient (k = B) of (range, d);
  // C role: CA(qa, w, d2) band blocks (w = qb - qa), row coordinate d
  constexpr int kAcc = 4 * (B + 1) > 48 ? 4 * (B + 1) : 48;
  double acc_[kAcc];
#define EA(q, k) acc_[(q) * (B + 1) + (k)]
#define CA(qa, w, d2) acc_[((qa) * 4 + (w)) * 3 + (d2)]
  int myr = -1, myd = 0;
  bool crole = false;
  for (int item = tid; item < 6 * nr; item += kGaThreads) {
    crole = item >= 3 * nr;
    const int it2 = crole ? item - 3 * nr : item;
    const int r = it2 / 3, d = it2 % 3;
    const int b = rs[r], e = re[r];
    const double* sx = Js + B * kGaStride;               // spline slot (q, d2) x-row: sx[(3q+d2)*stride + u]
    const double* sy = Js + (NS + B) * kGaStride;
    myr = r; myd = d;
    if (!crole) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k <= B; ++k) EA(q, k) = 0.0;
      for (int u = b; u < e; ++u) {
        double ax[4], ay[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { ax[q] = sx[(3 * q + d) * kGaStride + u]; ay[q] = sy[(3 * q + d) * kGaStride + u]; }
#pragma unroll
        for (int k = 0; k <= B; ++k) {
          const double cx = k < B ? Js[k * kGaStride + u] : fxs[u];
          const double cy = k < B ? Js[(NS + k) * kGaStride + u] : fys[u];
#pragma unroll
          for (int q = 0; q < 4; ++q) EA(q, k) += cx * ax[q] + cy * ay[q];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int d2 = 0; d2 < 3; ++d2) CA(q, w, d2) = 0.0;
      for (int u = b; u < e; ++u) {
        double vx[12], vy[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) { vx[k] = sx[k * kGaStride + u]; vy[k] = sy[k * kGaStride + u]; }
#pragma unroll
        for (int qa = 0; qa < 4; ++qa) {                 // vx[3qa + d] with d a run-time value: select, keeps vx in registers
          const double ax = d == 0 ? vx[3 * qa] : (d == 1 ? vx[3 * qa + 1] : vx[3 * qa + 2]);
          const double ay = d == 0 ? vy[3 * qa] : (d == 1 ? vy[3 * qa + 1] : vy[3 * qa + 2]);
#pragma unroll
          for (int w = 0; qa + w < 4; ++w)
#pragma unroll
            for (int d2 = 0; d2 < 3; ++d2) CA(qa, w, d2) += ax * vx[3 * (qa + w) + d2] + ay * vy[3 * (qa + w) + d2];
        }
      }
    }
    if (!fast) {                                         // slow path: too many ranges to keep one per thread
      const int gq = rg[r];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!crole) {
          const int row = 3 * (gq + q) + d;
          double* er = ne.Et + ((long long)c * ne.N3 + row) * B;
#pragma unroll
          for (int k = 0; k < B; ++k) if (EA(q, k) != 0.0) unsafeAtomicAdd(&er[k], EA(q, k));
          if (EA(q, B) != 0.0) unsafeAtomicAdd(&ne.gs[row], EA(q, B));
        } else {
#pragma unroll
          for (int w = 0; q + w < 4; ++w)
#pragma unroll
            for (int d2 = 0; d2 < 3; ++d2)
              if (CA(q, w, d2) != 0.0) unsafeAtomicAdd(&ne.Cb[((long long)(gq + q) * ne.W + w) * 9 + 3 * d + d2], CA(q, w, d2));
        }
      }
    }
  }
  if (!fast) return;
  lds_barrier();                                        // every thread holds its partials in registers: Js is dead
  // ---- owner tables of ALL flush rounds, built once ----------------------------------------------------------------
  // A round covers kRb consecutive ranges (as many per-range partial blocks as fit the dead staging LDS).  Inside a round a
  // control point is owned by the first range that reaches it: range t owns its last min(4, rg[t] - rg[t-1]) points (all four
  // when it opens a round).  The thread of range t writes, for each point it owns: the control point, itself as the first
  // reaching range, how many consecutive ranges of the round reach the point (ranges are span ordered) and the slot q2 of the
  // point inside the first four of them -- so a flush thread reads one table entry and then its partial sums, with no chain of
  // dependent look-ups per output (measured per workgroup of 68 ranges: 4 owner builds x 3k cycles + 2 x 2 loops x 4k cycles
  // before; one build now).
  constexpr int kRb = kRbE < kRbC ? kRbE : kRbC;
  __shared__ int round_off[8];                            // first owned point of round rho; [rounds] = total
  {
    const int t = tid;
    const bool has = t < nr;
    const int rl = has ? t % kRb : 0;
    const int mine = has ? (rl == 0 ? 4 : min(4, rg[t] - rg[t - 1])) : 0;
    int incl = mine;
    if (tid < 128) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if ((tid & 63) >= off) incl += o; }
      if (tid == 63) round_off[7] = incl;                // total of the first wavefront (scratch)
    }
    lds_barrier();
    if (has) {
      const int bs = incl - mine + (tid >= 64 ? round_off[7] : 0);
      const int rend = min(nr, (t / kRb + 1) * kRb);     // one past the last range of this range's round
      for (int j = 0; j < mine; ++j) {
        const int ctrl = rg[t] + (4 - mine + j);
        int cnt = 0, qp = 0;
        while (t + cnt < rend && ctrl - rg[t + cnt] >= 0) { if (cnt < 4) qp |= (ctrl - rg[t + cnt]) << (2 * cnt); ++cnt; }
        pt_ctrl[bs + j] = ctrl; pt_rl[bs + j] = (unsigned char)rl; pt_cnt[bs + j] = (unsigned char)cnt; pt_q[bs + j] = (unsigned char)qp;
      }
      if (rl == 0) round_off[t / kRb] = bs;
      if (t == nr - 1) round_off[(nr - 1) / kRb + 1] = bs + mine;
    }
    lds_barrier();
  }
  // ---- flush of the cross block + gradient: Ep[rl][q][3][B], Gp[rl][q][3] ----
  for (int r0 = 0, rho = 0; r0 < nr; r0 += kRb, ++rho) {
    const int nb = min(kRb, nr - r0);
    double* Ep = Js;
    double* Gp = Js + kRb * kEp;
    if (!crole && myr >= r0 && myr < r0 + nb) {
      const int rl = myr - r0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int k = 0; k < B; ++k) Ep[(rl * 4 + q) * 3 * B + myd * B + k] = EA(q, k);
        Gp[(rl * 4 + q) * 3 + myd] = EA(q, B);
      }
    }
    lds_barrier();
    const int p0 = round_off[rho], nown = round_off[rho + 1] - p0;
    constexpr int per = 3 * B + 3;                       // entries per owned control point: 3 x B cross + 3 gradient
    for (int o = tid; o < nown * per; o += kGaThreads) {
      const int i = p0 + o / per, dk = o % per;
      const int ctrl = pt_ctrl[i], rl = pt_rl[i], cnt = pt_cnt[i], qp = pt_q[i];
      const bool grad = dk >= 3 * B;
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j)                        // the first four reaching ranges: slots from the table, reads independent
        if (j < cnt) { const int q2 = (qp >> (2 * j)) & 3; acc += grad ? Gp[((rl + j) * 4 + q2) * 3 + dk - 3 * B] : Ep[((rl + j) * 4 + q2) * 3 * B + dk]; }
      for (int j = 4; j < cnt; ++j) {                    // (a span cut into many pieces: rare)
        const int q2 = ctrl - rg[r0 + rl + j];
        acc += grad ? Gp[((rl + j) * 4 + q2) * 3 + dk - 3 * B] : Ep[((rl + j) * 4 + q2) * 3 * B + dk];
      }
      if (acc != 0.0) {
        if (grad) unsafeAtomicAdd(&ne.gs[3 * ctrl + dk - 3 * B], acc);
        else unsafeAtomicAdd(&ne.Et[((long long)c * ne.N3 + 3 * ctrl) * B + dk], acc);
      }
    }
    lds_barrier();
  }
  // ---- flush of the spline band: Cp[rl][pair (qa, w)][3][3], pair = 4 qa - qa (qa - 1) / 2 + w ----
  for (int r0 = 0, rho = 0; r0 < nr; r0 += kRb, ++rho) {
    const int nb = min(kRb, nr - r0);
    double* Cp = Js;
    if (crole && myr >= r0 && myr < r0 + nb) {
      const int rl = myr - r0;
#pragma unroll
      for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int w = 0; qa + w < 4; ++w)
#pragma unroll
          for (int d2 = 0; d2 < 3; ++d2) Cp[(rl * 10 + 4 * qa - qa * (qa - 1) / 2 + w) * 9 + 3 * myd + d2] = CA(qa, w, d2);
    }
    lds_barrier();
    const int p0 = round_off[rho], nown = round_off[rho + 1] - p0;
    for (int o = tid; o < nown * 36; o += kGaThreads) {
      const int i = p0 + o / 36, wd = o % 36, w = wd / 9, dd = wd % 9;
      const int ctrl = pt_ctrl[i], rl = pt_rl[i], cnt = pt_cnt[i], qp = pt_q[i];
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < cnt) { const int q2 = (qp >> (2 * j)) & 3; if (q2 + w < 4) acc += Cp[((rl + j) * 10 + 4 * q2 - q2 * (q2 - 1) / 2 + w) * 9 + dd]; }
      for (int j = 4; j < cnt; ++j) {
        const int q2 = ctrl - rg[r0 + rl + j];
        if (q2 + w < 4) acc += Cp[((rl + j) * 10 + 4 * q2 - q2 * (q2 - 1) / 2 + w) * 9 + dd];
      }
      if (acc != 0.0) unsafeAtomicAdd(&ne.Cb[((long long)ctrl * ne.W) * 9 + wd], acc);
    }
    lds_barrier();
  }
#undef EA
#undef CA
}

__device__ __forceinline__ void lds_wave_sync() {      // orders the LDS traffic of ONE wavefront (writes of some lanes read by others)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
// A[c] (both triangles) and gc[c] from the per-workgroup partial blocks of k_assemble_spans: one workgroup per camera, entry e
// of the packed lower triangle summed over the camera's assembly workgroups in index order by four threads (quarters, then a
// fixed combine) -- no atomics, the same bits every run.  Workgroups that had nothing to add left zeros (the buffer is
// cleared with the rest of the normal equations).
template <int B>
__global__ __launch_bounds__(1024) void k_cam_block_reduce(DevProblem dp, NEView ne) {
  constexpr int PSZ = (B + 1) * (B + 2) / 2, kLanes = PSZ <= 64 ? 64 : 256, kGroups = 1024 / kLanes;
  __shared__ double part[kGroups][kLanes];
  const int c = blockIdx.x;
  constexpr int kPer = kGaParts * (kGaThreads / 64);                      // partial blocks per 256-chunk: assembly workgroups x wavefronts
  // two workgroups per camera (blockIdx.y), each over half of the partial blocks; both add into the zeroed A / gc with atomics --
  // 0 + a + b has the same bits in either order, so the result stays deterministic
  const int wa = dp.cam_chunk_off[c] * kPer, wb = dp.cam_chunk_off[c + 1] * kPer, wm = wa + (wb - wa + 1) / 2;
  const int w0 = blockIdx.y == 0 ? wa : wm, w1 = blockIdx.y == 0 ? wm : wb;
  const int k = threadIdx.x % kLanes, grp = threadIdx.x / kLanes;
  // group grp adds the partial blocks w0 + grp, w0 + grp + kGroups, ... (independent loads, four in flight), then the groups are
  // added in index order: a fixed summation tree
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (k < PSZ) {
    int w = w0 + grp;
    for (; w + 3 * kGroups < w1; w += 4 * kGroups) {
      a0 += ne.Apart[(long long)w * PSZ + k];
      a1 += ne.Apart[(long long)(w + kGroups) * PSZ + k];
      a2 += ne.Apart[(long long)(w + 2 * kGroups) * PSZ + k];
      a3 += ne.Apart[(long long)(w + 3 * kGroups) * PSZ + k];
    }
    for (; w < w1; w += kGroups) a0 += ne.Apart[(long long)w * PSZ + k];
  }
  part[grp][k] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < PSZ) {
    double v = 0.0;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) v += part[g][threadIdx.x];
    const int kk = threadIdx.x;
    int ra = 0;
    while ((ra + 1) * (ra + 2) / 2 <= kk) ++ra;
    const int rb = kk - ra * (ra + 1) / 2;
    if (ra == B) { if (rb < B) unsafeAtomicAdd(&ne.gc[c * B + rb], v); }
    else {
      unsafeAtomicAdd(&ne.A[((long long)c * B + ra) * B + rb], v);
      if (ra != rb) unsafeAtomicAdd(&ne.A[((long long)c * B + rb) * B + ra], v);
    }
  }
}

// motion-regulariser rows: spline block and gradient only (the rows do not depend on camera parameters).
// 256 consecutive rows (= consecutive sample times) per workgroup; each row is first compacted to the <= 6
// distinct control points it touches, then accumulated into an LDS window like the detection rows.
constexpr int kMotW = 6;
__global__ __launch_bounds__(kThreads) void k_assemble_motion(DevProblem dp, const double* __restrict__ mJ, const int32_t* __restrict__ mctrl,
                                                              const double* __restrict__ fm, NEView ne) {
  __shared__ double Cw[kNWin * kMotW * 9];
  __shared__ double gsw[kNWin * 3];
  __shared__ int gmin_s[kThreads / 64];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < kNWin * kMotW * 9; k += kThreads) Cw[k] = 0.0;
  for (int k = threadIdx.x; k < kNWin * 3; k += kThreads) gsw[k] = 0.0;
  // compact the row: control points lo .. lo+kMotW-1, 3 coordinates each
  int lo = 0x7fffffff, hi = -1;
  int cid[3] = {-1, -1, -1};
  if (j < dp.T) {
    bool outside = false;
    for (int k = 0; k < 3; ++k) {
      cid[k] = mctrl[(long long)k * dp.T + j];
      if (cid[k] >= 0) {
        cid[k] -= ne.row0;                               // local control point
        if (cid[k] < 0 || cid[k] + 3 >= ne.N) outside = true;
        lo = min(lo, cid[k]); hi = max(hi, cid[k] + 3);
      }
    }
    if (outside) { atomicOr(ne.err, 1); lo = 0x7fffffff; hi = -1; cid[0] = cid[1] = cid[2] = -1; }
  }
  double rv[kMotW * 3];
#pragma unroll
  for (int e = 0; e < kMotW * 3; ++e) rv[e] = 0.0;
  const bool live = hi >= 0 && hi - lo < kMotW;      // the band width W (<= 6) guarantees this for valid rows
  if (live)
    for (int k = 0; k < 3; ++k) {
      if (cid[k] < 0) continue;
      const int o = cid[k] - lo;
      for (int q = 0; q < 4; ++q)
        for (int d = 0; d < 3; ++d) {
          const double v = mJ[(long long)(12 * k + 3 * q + d) * dp.T + j];
          // static indexing of rv: select by comparison
#pragma unroll
          for (int e = 0; e < kMotW * 3; ++e) if (e == 3 * (o + q) + d) rv[e] += v;
        }
    }
  int gm = wave_min_i(live ? lo : 0x7fffffff);
  if (lane == 0) gmin_s[wave] = gm;
  __syncthreads();
  int g0 = gmin_s[0];
#pragma unroll
  for (int w = 1; w < kThreads / 64; ++w) g0 = min(g0, gmin_s[w]);
  if (g0 == 0x7fffffff) return;
  if (live) {
    const double fj = fm[j];
    const int l = lo - g0;
    const bool inwin = l + kMotW <= kNWin;
#pragma unroll
    for (int a = 0; a < kMotW; ++a) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double va = rv[3 * a + d];
        if (va == 0.0) continue;
        if (inwin) unsafeAtomicAdd(&gsw[3 * (l + a) + d], va * fj);
        else unsafeAtomicAdd(&ne.gs[3 * (lo + a) + d], va * fj);
#pragma unroll
        for (int b = a; b < kMotW; ++b) {
          if (b - a >= ne.W) continue;
#pragma unroll
          for (int d2 = 0; d2 < 3; ++d2) {
            const double vb = rv[3 * b + d2];
            if (vb == 0.0) continue;
            if (inwin) unsafeAtomicAdd(&Cw[((l + a) * kMotW + (b - a)) * 9 + 3 * d + d2], va * vb);
            else unsafeAtomicAdd(&ne.Cb[((long long)(lo + a) * ne.W + (b - a)) * 9 + 3 * d + d2], va * vb);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kNWin * 3; k += kThreads) {
    const int r = 3 * g0 + k;
    if (r < ne.N3 && gsw[k] != 0.0) unsafeAtomicAdd(&ne.gs[r], gsw[k]);
  }
  for (int k = threadIdx.x; k < kNWin * kMotW * 9; k += kThreads) {
    const int gg = g0 + k / (kMotW * 9), w = (k / 9) % kMotW;
    const double v = Cw[k];
    if (gg < ne.N && w < ne.W && v != 0.0) unsafeAtomicAdd(&ne.Cb[((long long)gg * ne.W + w) * 9 + (k % 9)], v);
  }
}

// The motion rows in the deterministic mode: one wavefront per control point g.  The rows that can touch g (row_lo[g] <= j <
// row_hi[g], as in k_jtu_reduce) are taken 64 at a time: first lane = ROW -- its coefficients of (g + a, coordinate d), a < W, i.e.
// the sum of its (<= 3) sample blocks that reach that control point (what k_assemble_motion compacts into rv[]), and its residual,
// go to LDS; then lane = ENTRY (gradient coordinate, or band entry (w, d, d2)) adds the rows in row order.  One order of additions
// per entry, no atomics.  Rows k_assemble_motion skips (outside the slice, wider than kMotW control points) are zeros here.
constexpr int kDetMotW = 6;                              // >= ne.W (the band solver supports at most six 3x3 blocks)
__global__ __launch_bounds__(kThreads) void k_det_motion(DevProblem dp, const double* __restrict__ mJ, const int32_t* __restrict__ mctrl,
                                                         const double* __restrict__ fm, NEView ne) {
  constexpr int kWaves = kThreads / 64, kRv = kDetMotW * 3 + 1;       // per row: W x 3 coefficients + the residual
  __shared__ double rv[kWaves][64][kRv + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = (int)blockIdx.x * kWaves + wave;
  if (g >= ne.N) return;                                // (wave-uniform; no workgroup barrier below)
  const int gg = g + ne.row0;                            // row_lo / row_hi are indexed by the global control point
  const int j0 = dp.mv.row_lo[gg], j1 = dp.mv.row_hi[gg];
  const int per = 3 + ne.W * 9;
  const bool grad = lane < 3;
  const int w = grad ? 0 : (lane - 3) / 9, d = grad ? lane : ((lane - 3) % 9) / 3, d2 = grad ? 0 : (lane - 3) % 3;
  const bool mine = lane < per && g + w < ne.N;
  double acc = 0.0;
  for (int jb = j0; jb < j1; jb += 64) {
    const int j = jb + lane;
    double r[kRv];
#pragma unroll
    for (int e = 0; e < kRv; ++e) r[e] = 0.0;
    if (j < j1) {
      int cid[3], lo = 0x7fffffff, hi = -1;
      bool outside = false;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        cid[k] = mctrl[(long long)k * dp.T + j];
        if (cid[k] >= 0) {
          cid[k] -= ne.row0;
          if (cid[k] < 0 || cid[k] + 3 >= ne.N) outside = true;
          lo = min(lo, cid[k]); hi = max(hi, cid[k] + 3);
        }
      }
      if (!outside && hi >= 0 && hi - lo < kMotW) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (cid[k] < 0) continue;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int a = cid[k] + q - g;                // this block's control point q is g + a
            if (a < 0 || a >= kDetMotW) continue;
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
              const double v = mJ[(long long)(12 * k + 3 * q + dd) * dp.T + j];
#pragma unroll
              for (int e = 0; e < kDetMotW * 3; ++e) if (e == 3 * a + dd) r[e] += v;      // static indexing of r: select by comparison
            }
          }
        }
        r[kDetMotW * 3] = fm[j];
      }
    }
#pragma unroll
    for (int e = 0; e < kRv; ++e) rv[wave][lane][e] = r[e];
    lds_wave_sync();
    const int nrow = min(64, j1 - jb);
    if (mine) {
      for (int t = 0; t < nrow; ++t) {
        const double va = rv[wave][t][d];
        const double vb = grad ? rv[wave][t][kDetMotW * 3] : rv[wave][t][3 * w + d2];
        acc += va * vb;
      }
    }
    lds_wave_sync();
  }
  if (mine && acc != 0.0) {
    if (grad) ne.gs[3 * g + lane] += acc;
    else ne.Cb[((long long)g * ne.W) * 9 + (lane - 3)] += acc;
  }
}

// The same for motion rows that reach further than six control points (FITPACK knots less than a frame apart: the three samples of
// a row then span more than three knot spans; the band has W <= kWideW blocks per row): one wavefront per control point, the rows'
// coefficients built in LDS by the row's own lane, every lane adds up to three entries of the control point's band row.
constexpr int kWideW = 16;
__global__ __launch_bounds__(64) void k_det_motion_wide(DevProblem dp, const double* __restrict__ mJ, const int32_t* __restrict__ mctrl,
                                                        const double* __restrict__ fm, NEView ne) {
  constexpr int kRv = kWideW * 3 + 1;                      // per row: W x 3 coefficients + the residual
  __shared__ double rv[64][kRv + 1];
  const int lane = threadIdx.x, g = blockIdx.x;
  if (g >= ne.N) return;
  const int gg = g + ne.row0;
  const int j0 = dp.mv.row_lo[gg], j1 = dp.mv.row_hi[gg];
  const int per = 3 + ne.W * 9;
  double acc[3] = {0.0, 0.0, 0.0};
  for (int jb = j0; jb < j1; jb += 64) {
    const int j = jb + lane;
    for (int e = 0; e < kRv; ++e) rv[lane][e] = 0.0;
    if (j < j1) {
      int cid[3], lo = 0x7fffffff, hi = -1;
      bool outside = false;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        cid[k] = mctrl[(long long)k * dp.T + j];
        if (cid[k] >= 0) {
          cid[k] -= ne.row0;
          if (cid[k] < 0 || cid[k] + 3 >= ne.N) outside = true;
          lo = min(lo, cid[k]); hi = max(hi, cid[k] + 3);
        }
      }
      if (!outside && hi >= 0 && hi - lo < kWideW) {
        for (int k = 0; k < 3; ++k) {
          if (cid[k] < 0) continue;
          for (int q = 0; q < 4; ++q) {
            const int a = cid[k] + q - g;                  // this block's control point q is g + a
            if (a < 0 || a >= kWideW) continue;
            for (int dd = 0; dd < 3; ++dd) rv[lane][3 * a + dd] += mJ[(long long)(12 * k + 3 * q + dd) * dp.T + j];
          }
        }
        rv[lane][kWideW * 3] = fm[j];
      }
    }
    lds_wave_sync();
    const int nrow = min(64, j1 - jb);
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
      const int e = lane + 64 * t3;
      if (e >= per) continue;
      const bool grad = e < 3;
      const int w = grad ? 0 : (e - 3) / 9, d = grad ? e : ((e - 3) % 9) / 3, d2 = grad ? 0 : (e - 3) % 3;
      if (g + w >= ne.N) continue;
      double a = 0.0;
      for (int t = 0; t < nrow; ++t) a += rv[t][d] * (grad ? rv[t][kWideW * 3] : rv[t][3 * w + d2]);
      acc[t3] += a;
    }
    lds_wave_sync();
  }
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3) {
    const int e = lane + 64 * t3;
    if (e >= per || acc[t3] == 0.0) continue;
    if (e < 3) ne.gs[3 * g + e] += acc[t3];
    else if (g + (e - 3) / 9 < ne.N) ne.Cb[((long long)g * ne.W) * 9 + (e - 3)] += acc[t3];
  }
}

// ---- the spline block as a GENERAL band (W > 6): Cholesky in place, then every right-hand side by substitution ----------------
// Lb[i][j] = M(i, i - j), j = 0..BW (the damped band of k_band_pack).  One workgroup: a sliding window of BW + 1 rows in LDS (row
// k + r in slot (k + r) mod (BW + 1)), column by column -- pivot, the column's BW entries, the rank-one update of the window's
// trailing triangle -- the finished row goes back to memory and the next one comes in.  A fallback for the rare wide band
// (the partitioned solver below is built for W = 4 and 6): O(n BW^2) work on one CU.
__global__ __launch_bounds__(256) void k_band_chol_generic(int n, int BW, double* __restrict__ Lb, int* __restrict__ fail) {
  extern __shared__ double bwin[];                          // [(BW + 1)][(BW + 1)]
  __shared__ double dsh;
  const int R = BW + 1, tid = threadIdx.x;
  for (int e = tid; e < R * R; e += 256) { const int r = e / R, c = e % R; bwin[e] = r < n ? Lb[(long long)r * R + c] : 0.0; }
  __syncthreads();
  bool bad = false;
  for (int k = 0; k < n; ++k) {
    const int s0 = k % R;
    double* row0 = bwin + s0 * R;
    if (tid == 0) { const double p = row0[0]; bad = !(p > 0.0); dsh = sqrt(p > 0.0 ? p : 1.0); }
    __syncthreads();
    const double d = dsh;
    const int rows = min(BW, n - 1 - k);                    // rows k+1 .. k+rows hold column k
    if (tid == 0) row0[0] = d;
    for (int r = 1 + tid; r <= rows; r += 256) bwin[((k + r) % R) * R + r] /= d;     // M(k + r, k) sits at entry r of its row
    __syncthreads();
    for (int e = tid; e < rows * (rows + 1) / 2; e += 256) {                        // (r, s), 1 <= s <= r <= rows: M(k+r, k+s) -= l_r l_s
      int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while ((r + 1) * (r + 2) / 2 <= e) ++r;
      while (r * (r + 1) / 2 > e) --r;
      const int sidx = e - r * (r + 1) / 2 + 1, rr = r + 1;
      double* rowr = bwin + ((k + rr) % R) * R;
      rowr[rr - sidx] -= rowr[rr] * bwin[((k + sidx) % R) * R + sidx];
    }
    __syncthreads();
    // row k is final: back to memory; its slot takes row k + R
    for (int c = tid; c < R; c += 256) {
      Lb[(long long)k * R + c] = row0[c];
      const int nr = k + R;
      row0[c] = nr < n ? Lb[(long long)nr * R + c] : 0.0;
    }
    __syncthreads();
  }
  if (bad && tid == 0) fail[0] = 3;
}
// Z[i][c] <- (L L^T)^-1 Z[i][c]: one lane per right-hand side (64 per workgroup), the last BW solution entries in an LDS ring
__global__ __launch_bounds__(64) void k_band_solve_generic(int n, int BW, int ncols, const double* __restrict__ Lb, double* __restrict__ Z) {
  extern __shared__ double ring[];                          // [BW + 1][64]
  const int R = BW + 1, lane = threadIdx.x, c = blockIdx.x * 64 + lane;
  const bool on = c < ncols;
  for (int i = 0; i < n; ++i) {                             // forward: y_i = (b_i - sum_j L(i, i-j) y_{i-j}) / L(i, i)
    const double* Li = Lb + (long long)i * R;
    double sacc = on ? Z[(long long)i * ncols + c] : 0.0;
    const int jm = min(BW, i);
    for (int j = 1; j <= jm; ++j) sacc -= Li[j] * ring[((i - j) % R) * 64 + lane];
    const double y = sacc / Li[0];
    ring[(i % R) * 64 + lane] = y;
    if (on) Z[(long long)i * ncols + c] = y;
  }
  for (int i = n - 1; i >= 0; --i) {                        // backward: x_i = (y_i - sum_j L(i+j, i) x_{i+j}) / L(i, i)
    double sacc = on ? Z[(long long)i * ncols + c] : 0.0;
    const int jm = min(BW, n - 1 - i);
    for (int j = 1; j <= jm; ++j) sacc -= Lb[(long long)(i + j) * R + j] * ring[((i + j) % R) * 64 + lane];
    const double xv = sacc / Lb[(long long)i * R];
    ring[(i % R) * 64 + lane] = xv;
    if (on) Z[(long long)i * ncols + c] = xv;
  }
}

// D = diag(H) in x order (0 -> 1 so that unused columns stay put), and g in x order
__device__ __forceinline__ void ne_diag_grad_entry(const DevProblem& dp, const NEView& ne, int raw, int idx, double* __restrict__ D, double* __restrict__ gx) {
  // raw (time shards): this rank's PARTIAL diagonal, to be summed over the ranks before the unpacking k_halo_copy replaces zeros by 1
  if (idx < ne.CB) {
    const int c = idx / ne.B, k = idx % ne.B;
    const double h = ne.A[((long long)c * ne.B + k) * ne.B + k];
    const int col = cam_col(dp.C, dp.P, c, k);
    D[col] = (raw || h > 0.0) ? h : 1.0;
    gx[col] = ne.gc[idx];
  } else if (idx < ne.CB + ne.N3) {
    const int r = idx - ne.CB, g = r / 3, d = r % 3;
    const double h = ne.Cb[((long long)g * ne.W) * 9 + 4 * d];
    const int gg = g + ne.row0;
    const int col = dp.mv.ctrl_x0[gg] + d * dp.mv.ctrl_stride[gg];
    D[col] = (raw || h > 0.0) ? h : 1.0;
    gx[col] = ne.gs[r];
  }
}
__global__ void k_ne_diag_grad(DevProblem dp, NEView ne, int raw, double* __restrict__ D, double* __restrict__ gx) {
  ne_diag_grad_entry(dp, ne, raw, blockIdx.x * blockDim.x + threadIdx.x, D, gx);
}

// ---- time shards: the blocks of the control points within `halo` of a cut receive rows from both neighbours --------
// Packed exchange buffer: [boundary b = 1 .. world-1][2*halo control points from cut_b - halo][3*CB cross | W*9 band | 3 grad].
// pack: this rank's partial blocks of its (<= 2) boundaries, everything else stays zero; after the sum over the ranks
// unpack overwrites the local blocks with the totals.
// (round 6: the launch also carries what used to be two launches of their own either side of the sum -- pack: this rank's PARTIAL
// diag(H) and g in x order (k_ne_diag_grad, raw); unpack: zeros of the summed diagonal -> 1)
__global__ void k_halo_copy(NEView ne, int halo, int nb, const int* __restrict__ bcut, const int* __restrict__ bidx, int Ntot,
                            double* __restrict__ buf, int unpack, DevProblem dp, double* __restrict__ D, double* __restrict__ gx, long long n) {
  const int per = 3 * ne.CB + ne.W * 9 + 3;
  const long long total = (long long)nb * 2 * halo * per;
  if (!unpack) {
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < ne.CB + ne.N3; e += (long long)gridDim.x * blockDim.x)
      ne_diag_grad_entry(dp, ne, 1, (int)e, D, gx);
  } else {
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
      if (!(D[e] > 0.0)) D[e] = 1.0;
  }
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(e % per);
    const int ci = (int)((e / per) % (2 * halo)), b = (int)(e / ((long long)per * 2 * halo));
    const int ctrl = bcut[b] - halo + ci;              // global control point
    if (ctrl < 0 || ctrl >= Ntot) continue;
    const int l = ctrl - ne.row0;
    if (l < 0 || l >= ne.N) continue;                  // cannot happen: the slice spans its own range +- halo
    double* slot;
    if (f < 3 * ne.CB) { const int c = f / (3 * ne.B), dk = f % (3 * ne.B); slot = ne.Et + ((long long)c * ne.N3 + 3 * l) * ne.B + dk; }
    else if (f < 3 * ne.CB + ne.W * 9) slot = ne.Cb + (long long)l * ne.W * 9 + (f - 3 * ne.CB);
    else slot = ne.gs + 3 * l + (f - 3 * ne.CB - ne.W * 9);
    double* dst = buf + ((long long)bidx[b] * 2 * halo + ci) * per + f;
    if (unpack) *slot = *dst; else *dst = *slot;
  }
}
// time shards: the two failure flags travel with the step (px[n], px[n+1]) through its sum over the ranks, so that every
// rank takes the same decision (a rank whose own interiors factorise fine must still see its neighbour's failure)
// A hand-over time-out of the reduced solve (fail[0] = kFailHandoverCode, ba_rcs.hip.h) is not a numerical failure: it travels as a
// value no sum of the other codes (<= 8 each, <= 64 ranks) can reach, and EVERY rank then reports the time-out -- all of them repeat
// the solve on the separate-launch route together (HipSchur::retry_same), none raises the damping alone.
// (round 6: packed by the last kernel of the solve, k_back_substitute; the sum is read where it lands -- by the trial kernel, which takes
// no step when any rank failed, and by the host through the driver's scalars: no pack / unpack launches, no copy of its own)
constexpr int kFailHandoverCode = 4;
constexpr double kFailHandoverSum = 1048576.0;
constexpr double kFailSpanSum = 4096.0;                 // flag bit 2 (a span table that does not belong to the point) in the sum of the second flags
inline int fail_sum_code(double t0) { return t0 >= kFailHandoverSum ? kFailHandoverCode : (t0 != 0.0 ? 8 : 0); }
inline int fail_sum_flag(double t1) { return (t1 >= kFailSpanSum ? 2 : 0) | (std::fmod(t1, kFailSpanSum) != 0.0 ? 1 : 0); }
// (zero / nzero: the step vector every rank then writes its rows of, before it is summed -- cleared here, not by a launch of its own)
__global__ void k_sum_slabs(long long count, int nslab, const double* __restrict__ Gp, double* __restrict__ G0, double* __restrict__ zero, long long nzero) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < nzero) zero[i] = 0.0;
  if (i >= count) return;
  double t = 0.0;
  for (int sl = 0; sl < nslab; ++sl) t += Gp[sl * count + i];
  G0[i] = t;
}

// scalar lower band of (C + lambda D_s): Lb[i][j] = (C + lambda D)(i, i-j), j = 0..BW
__device__ __forceinline__ void ne_diag_grad_entry(const DevProblem& dp, const NEView& ne, int raw, int idx, double* __restrict__ D, double* __restrict__ gx);
// with_diag: the launch also writes D = diag(H) and g in x order (k_ne_diag_grad's work, folded in: one launch less per
// linearisation; the first solve after an assembly carries it)
__device__ __forceinline__ void band_pack_entry(const NEView& ne, double lambda, int BW, double* __restrict__ Lb, int* __restrict__ fail, const DevProblem& dp,
                                                int with_diag, double* __restrict__ D, double* __restrict__ gx, long long idx) {
  if (idx == 0) { fail[0] = 0; fail[2] = 0; }   // first kernel of a solve: clears the failure flag the later ones may raise, and the sticky hand-over mark
  if (with_diag && idx < ne.CB + ne.N3) ne_diag_grad_entry(dp, ne, 0, (int)idx, D, gx);
  const long long total = (long long)ne.N3 * (BW + 1);
  if (idx >= total) return;
  const int i = (int)(idx / (BW + 1)), j = (int)(idx % (BW + 1));
  const int cidx = i - j;
  double v = 0.0;
  if (cidx >= 0) {
    const int gi = i / 3, ai = i % 3, gc_ = cidx / 3, ac = cidx % 3;
    const int w = gi - gc_;
    if (w < ne.W) v = ne.Cb[((long long)gc_ * ne.W + w) * 9 + 3 * ac + ai];
    if (j == 0) { const double h = v; v = h + lambda * (h > 0.0 ? h : 1.0); }
  }
  Lb[idx] = v;
}

// ---- partitioned (separator-based) parallel solve of the banded spline system ---------------------------
// The chain of control points is cut into P interiors of kPartL control points separated by separators of
// W-1 control points (= the block half-bandwidth, so two interiors never couple directly):
//     I_0 S_0 I_1 S_1 ... I_{P-1}
// A) every interior is factorised and solved independently (one wavefront per interior; one thread per
//    right-hand side, plus the 2*(W-1)*3 coupling columns V_p = B_p^-1 H[I_p,S_{p-1}], W_p = B_p^-1 H[I_p,S_p]);
// B) the separator system (block tridiagonal, blocks of 3(W-1)) is formed from short dot products;
// C) it is solved sequentially by one workgroup (P-1 small steps, all right-hand sides in parallel);
// D) the interiors are corrected: X_I = Y - V X_{S_{p-1}} - W X_{S_p}.

// Rows are LOCAL scalar rows of this handle's slice of the spline system (the whole system unless the handle is a
// time shard); separators are numbered GLOBALLY along the whole chain, 0 .. m-1.
struct PartView {
  int P;                 // interiors of this slice
  int s3;                // separator size in scalars, 3*(W-1)
  const int* i0;         // [P]  first scalar row of interior p
  const int* i1;         // [P]  one past the last scalar row of interior p
  const int* sl;         // [P]  first scalar row of the separator left of interior p (global number q_off+p-1), -1: none
  const int* sr;         // [P]  first scalar row of the separator right of interior p (global number q_off+p), -1: none
  int q_off;             // global number of the separator right of interior 0
  int m;                 // separators of the whole chain
  // separator tasks of k_part_reduce, [nt] each: row of the separator, interior left / right of it held here (-1: held
  // by the neighbouring shard, which adds that part), global number, own = this slice adds H(S,S) and the rhs rows
  int nt;
  const int *tc0, *tpl, *tpr, *tgq, *town;
  double* VW;            // [P][kPartRowsMax][2*s3]
  // one rank, round 5 (no back-correction of the interiors' columns, see HipSchur::band_chain):
  double* Dl;            // [m][s3][CB] a copy of the separators' reduced right-hand sides R_S (pv.R is solved in place); nullptr: not kept
  const double* Et;      // with Dl: the separators' rows of Z are ZERO and their own right-hand sides are read from the assembled blocks:
  const double* gs;      //   the camera-major cross block [C][N3][B] and the spline gradient [N3]
  const unsigned char* seprow;   // [N3] 1 = the row belongs to a separator
  int CB, B, N3;
  int direct;            // 1: no right-hand-side copy -- the interior solves read the assembled blocks (part_solve_block), Z's separator rows stay zero
  double* T;             // [m][s3][s3] diagonal blocks of the separator system
  double* U;             // [m][s3][s3] U[q] = T(q, q+1)
  double *U2, *Ha, *Hc;  // [m][s3][s3] each: cyclic-reduction workspace (k_sep_bcr_*)
  double* R;             // [m][s3][ncols] reduced right-hand sides -> solution of the separator system
};

// original (damped) matrix entry H(i, c) read from the lower band; valid for separator rows/columns and for
// interior-separator couplings, which the interior factorisation never overwrites
template <int BW>
__device__ __forceinline__ double band_entry(const double* __restrict__ Lb, int i, int c) {
  const int hi = i > c ? i : c, d = i > c ? i - c : c - i;
  return d <= BW ? Lb[(long long)hi * (BW + 1) + d] : 0.0;
}
// the same without a conditional load (the row is always inside the band array): for the unconditioned batches of part_solve_block
template <int BW>
__device__ __forceinline__ double band_entry_nc(const double* __restrict__ Lb, int i, int c) {
  const int hi = i > c ? i : c, d = i > c ? i - c : c - i;
  const double v = Lb[(long long)hi * (BW + 1) + (d <= BW ? d : BW)];
  return d <= BW ? v : 0.0;
}

__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// banded Cholesky of interior p by ONE wavefront (lane = threadIdx.x & 63); T: (kPartRowsMax + 1) * (BW + 1) doubles of LDS private to it.
// (Only this wavefront touches T: the synchronisation is the wavefront-level one.)
__device__ __forceinline__ void part_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
template <int BW>
__device__ __forceinline__ void part_cholesky_body(const PartView& pv, double* __restrict__ Lb, int* __restrict__ fail, double* __restrict__ T, int p, int lane) {
  constexpr int R = BW + 1;
  const int r0 = pv.i0[p], n = pv.i1[p] - r0;
  for (int e0 = lane; e0 < n * R; e0 += 64 * 8) {          // eight loads in flight per lane
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = Lb[(long long)r0 * R + min(e0 + 64 * u, n * R - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (e0 + 64 * u < n * R) T[e0 + 64 * u] = v[u];
  }
  // trailing-update pairs (rr >= ss >= 1) owned by this lane, as offsets from the pivot row: fixed for all columns
  constexpr int kPairs = BW * (BW + 1) / 2, kSlots = (kPairs + 63) / 64;
  int prr[kSlots], offA[kSlots], offB[kSlots], offC[kSlots];
#pragma unroll
  for (int sl = 0; sl < kSlots; ++sl) {
    const int e = lane + 64 * sl;
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= e) ++r;
    const int rr = r + 1, ss = e - r * (r + 1) / 2 + 1;
    prr[sl] = e < kPairs ? rr : BW + 1;
    offA[sl] = rr * R + (rr - ss); offB[sl] = rr * R + rr; offC[sl] = ss * R + ss;
  }
  part_lds_sync();
  // One LDS round trip per column (round 4; it was four: pivot, scaled column, its read-back, trailing update).  The trailing
  // update reads the UNSCALED column, A(k+r, k+s) -= T_r T_s / piv, so it does not wait for the scaled column; the next pivot is
  // the updated (1, 1) entry, which lane 0 holds in a register (pair 0) -- a v_readlane instead of an LDS read, and its
  // reciprocal square root runs under the latency of the next column's reads.
  // The loop body has no branch: a lane without work in a column reads the pivot's slot and writes to a spare slot behind the
  // matrix (T holds one more row than the interior can have), so the reads carry no exec-mask joins for s_waitcnt to stop at.
  constexpr int kSpare = kPartRowsMax * R;
  bool bad = false;
  double piv = T[0];
  for (int k = 0; k < n; ++k) {
    const int nb = min(BW, n - 1 - k);
    const int base = k * R;
    double a[kSlots], b[kSlots], c[kSlots];
    bool on[kSlots];
#pragma unroll
    for (int sl = 0; sl < kSlots; ++sl) {
      on[sl] = prr[sl] <= nb;
      a[sl] = T[base + (on[sl] ? offA[sl] : 0)]; b[sl] = T[base + (on[sl] ? offB[sl] : 0)]; c[sl] = T[base + (on[sl] ? offC[sl] : 0)];
    }
    const bool incol = lane >= 1 && lane <= nb;
    const int icol = incol ? lane * R + lane : 0;
    const double colv = T[base + icol];
    bad |= !(piv > 0.0);
    piv = piv > 0.0 ? piv : 1.0;
    double inv = __builtin_amdgcn_rsq(piv);              // two Newton steps on v_rsq_f64
    inv = inv * (1.5 - 0.5 * piv * inv * inv);
    inv = inv * (1.5 - 0.5 * piv * inv * inv);
    const double invp = inv * inv;
    double first = 0.0;
#pragma unroll
    for (int sl = 0; sl < kSlots; ++sl) {
      const double v = a[sl] - (b[sl] * invp) * c[sl];
      T[on[sl] ? base + offA[sl] : kSpare] = v;
      if (sl == 0) first = v;
    }
    T[incol ? base + icol : kSpare + 1] = colv * inv;     // the column of L (every read of it above precedes this store in program order)
    T[lane == 0 ? base : kSpare + 2] = inv;               // reciprocal of L(k,k)
    piv = bcast_lane(first, 0);                           // A(k+1, k+1) after this column's update (pair 0 = (1, 1)); unused when nb = 0
    part_lds_sync();
  }
  if (bad && lane == 0) fail[0] = 1;
  // write back only entries whose column lies inside the interior (j <= row - r0); couplings stay original
  for (int e = lane; e < n * R; e += 64) {
    const int row = e / R, j = e % R;
    if (j <= row) Lb[(long long)r0 * R + e] = T[e];
  }
}
template <int BW>
__global__ __launch_bounds__(64) void k_part_cholesky(PartView pv, double* __restrict__ Lb, int* __restrict__ fail) {
  __shared__ double T[(kPartRowsMax + 1) * (BW + 1)];
  part_cholesky_body<BW>(pv, Lb, fail, T, (int)blockIdx.x, (int)threadIdx.x);
}

// interior solves, one thread per column, one wavefront per (interior, 64 columns).  COUPLING = false: right-hand-side
// columns of Z, in place.  COUPLING = true (one extra wavefront per interior): the 2*s3 coupling columns
// V_p = B_p^-1 H[I_p, S_{p-1}], W_p = B_p^-1 H[I_p, S_p] -> pv.VW.
// A lone wavefront per SIMD issues in order, so the kernel is bound by its instruction count: the factor is staged in
// LDS once (entries reaching outside the interior zeroed, rows padded to whole batches with identity rows: the
// substitution loops are branch free) and read as broadcasts, rows are addressed by pointer increments, and the rows of
// the right-hand side are fetched one batch of kPf rows ahead (they are ncols*8 bytes apart: every row is a fresh
// cache line, ~1-2 us away).
#ifndef MVUS_PS_PF
#define MVUS_PS_PF 16
#endif
constexpr int kPsPf = MVUS_PS_PF;       // rows of right-hand side in flight per lane (a batch)
template <int BW, bool COUPLING>
__device__ __forceinline__ void part_solve_block(const PartView& pv, int ncols, const double* __restrict__ Lb, double* __restrict__ Z, double* Ls, int p, int by) {
  constexpr int R = BW + 1, kPf = kPsPf;
  const int tid = threadIdx.x;
  const int s3 = pv.s3;
  const int r0 = pv.i0[p], nr = pv.i1[p] - r0;
  const int nrp = (nr + kPf - 1) / kPf * kPf;                // rows padded to whole batches
  // (eight loads in flight per lane, none behind a condition: rows past the interior repeat its last row and are replaced afterwards)
  constexpr int kStage = 8;
  const int total = (nrp + kPf) * R;
  for (int e0 = tid; e0 < total; e0 += 64 * kStage) {
    double v[kStage];
#pragma unroll
    for (int u = 0; u < kStage; ++u) {
      const int e = min(e0 + 64 * u, total - 1), row = e / R, jj = e - row * R;
      v[u] = Lb[(long long)(r0 + min(row, nr - 1)) * R + jj];
    }
#pragma unroll
    for (int u = 0; u < kStage; ++u) {
      const int e = e0 + 64 * u, row = e / R, jj = e - row * R;
      if (e < total) Ls[e] = row < nr ? ((jj <= row) ? v[u] : 0.0) : (jj == 0 ? 1.0 : 0.0);   // column row-jj inside the interior
    }
  }
  __syncthreads();
  int ccol = 0;
  bool live;
  if (COUPLING) {
    live = tid < 2 * s3 && (tid < s3 ? pv.sl[p] >= 0 : pv.sr[p] >= 0);
    if (live) ccol = tid < s3 ? pv.sl[p] + tid : pv.sr[p] + (tid - s3);
  } else {
    live = (int)(by * 64 + tid) < ncols;
  }
  if (!live) return;                                         // no barrier below
  double* out = COUPLING ? pv.VW + ((long long)p * kPartRowsMax) * (2 * s3) + tid : Z + (long long)r0 * ncols + by * 64 + tid;
  const long long ostride = COUPLING ? 2 * s3 : ncols;
  // where the forward pass reads its right-hand side: the column's rows of Z (written by the right-hand-side copy), or -- pv.direct, one rank,
  // many columns -- the assembled blocks themselves: column (c, k) of the camera-major cross block, the spline gradient for the last
  // column.  The copy (a pass over all of E and Z) is then not launched at all.
  const double* src = out;
  long long sstride = ostride;
  if (!COUPLING && pv.direct) {
    const int col = min((int)(by * 64 + tid), ncols - 1);
    if (col < pv.CB) { src = pv.Et + ((long long)(col / pv.B) * pv.N3 + r0) * pv.B + col % pv.B; sstride = pv.B; }
    else { src = pv.gs + r0; sstride = 1; }
  }
  // Batches of kPf rows.  A FULL batch (all its rows inside the interior) is loaded, solved and stored without a single condition:
  // with a condition per row (`i < nr ? load : 0`) every load and store sat in its own scalar branch and the compiler closed each
  // batch of prefetches with s_waitcnt vmcnt(0) -- the wavefront then waited out the memory latency once per batch in both passes
  // (round 4: 45 -> see DESIGN section 5).  Only the last, partial batch keeps the conditions.
  using Full = std::true_type;
  using Part = std::false_type;
  auto load_rhs = [&](auto full, double (&dst)[kPf], int ib) {
#pragma unroll
    for (int k = 0; k < kPf; ++k) {
      const int i = ib + k;
      if constexpr (decltype(full)::value) dst[k] = COUPLING ? band_entry_nc<BW>(Lb, r0 + i, ccol) : src[(long long)i * sstride];
      else dst[k] = i < nr ? (COUPLING ? band_entry<BW>(Lb, r0 + i, ccol) : src[(long long)i * sstride]) : 0.0;
    }
  };
  auto load_y = [&](auto full, double (&dst)[kPf], int ib) {
#pragma unroll
    for (int k = 0; k < kPf; ++k) {
      const int i = ib + k;
      if constexpr (decltype(full)::value) dst[k] = out[(long long)i * ostride];
      else dst[k] = i < nr ? out[(long long)i * ostride] : 0.0;
    }
  };
  auto take = [&](double (&dst)[kPf], const double (&src)[kPf]) {
#pragma unroll
    for (int k = 0; k < kPf; ++k) dst[k] = src[k];
  };
  double yw[BW];           // yw[0] = newest value
  // forward: y(i) = (b(i) - sum_j L(i, i-j) y(i-j)) / L(i,i)
  auto forward = [&](auto full, const double (&cur)[kPf], int ib) {
    const double* Lr = Ls + ib * R;
    double* yo = out + (long long)ib * ostride;
#pragma unroll
    for (int k = 0; k < kPf; ++k) {
      // only the j = 1 term depends on the previous row's result: summing from the far end of the band leaves a
      // two-operation dependent chain per row
      double acc = cur[k];
#pragma unroll
      for (int j = BW; j >= 1; --j) acc -= Lr[k * R + j] * yw[j - 1];
      const double y = acc * Lr[k * R];
      if constexpr (decltype(full)::value) yo[(long long)k * ostride] = y;
      else if (ib + k < nr) yo[(long long)k * ostride] = y;
#pragma unroll
      for (int j = BW - 1; j >= 1; --j) yw[j] = yw[j - 1];
      yw[0] = y;
    }
  };
  // backward: x(i) = (y(i) - sum_j L(i+j, i) x(i+j)) / L(i,i); the padded rows give x = 0
  auto backward = [&](auto full, const double (&cur)[kPf], int ib) {
#pragma unroll
    for (int k = kPf - 1; k >= 0; --k) {
      const int i = ib + k;
      double acc = cur[k];
#pragma unroll
      for (int j = BW; j >= 1; --j) acc -= Ls[(i + j) * R + j] * yw[j - 1];     // rows up to nrp + BW - 1 are staged
      const double xv = acc * Ls[i * R];
      if constexpr (decltype(full)::value) out[(long long)i * ostride] = xv;
      else if (i < nr) out[(long long)i * ostride] = xv;
#pragma unroll
      for (int j = BW - 1; j >= 1; --j) yw[j] = yw[j - 1];
      yw[0] = xv;
    }
  };
  const int nfull = nr / kPf, tail0 = nfull * kPf;
  const bool has_tail = tail0 < nr;
  double cur[kPf], nxt[kPf];
#pragma unroll
  for (int j = 0; j < BW; ++j) yw[j] = 0.0;
  if (nfull > 0) {
    load_rhs(Full{}, nxt, 0);
    for (int bt = 0; bt + 1 < nfull; ++bt) {               // the rows of the next batch are fetched while this one is solved
      take(cur, nxt);
      load_rhs(Full{}, nxt, (bt + 1) * kPf);
      forward(Full{}, cur, bt * kPf);
    }
    take(cur, nxt);
    if (has_tail) load_rhs(Part{}, nxt, tail0);
    forward(Full{}, cur, tail0 - kPf);
  } else if (has_tail) {
    load_rhs(Part{}, nxt, 0);
  }
  if (has_tail) { take(cur, nxt); forward(Part{}, cur, tail0); }
#pragma unroll
  for (int j = 0; j < BW; ++j) yw[j] = 0.0;
  if (has_tail) {
    load_y(Part{}, cur, tail0);
    if (nfull > 0) load_y(Full{}, nxt, tail0 - kPf);
    backward(Part{}, cur, tail0);
  } else if (nfull > 0) {
    load_y(Full{}, nxt, tail0 - kPf);
  }
  for (int bt = nfull - 1; bt >= 1; --bt) {
    take(cur, nxt);
    load_y(Full{}, nxt, (bt - 1) * kPf);
    backward(Full{}, cur, bt * kPf);
  }
  if (nfull > 0) { take(cur, nxt); backward(Full{}, cur, 0); }
}

// column blocks 0 .. gy-2: 64 right-hand-side columns each; the last one: the coupling columns (same launch, so that they run
// beside the others instead of after them)
template <int BW>
__global__ __launch_bounds__(64) void k_part_solve(PartView pv, int ncols, const double* __restrict__ Lb, double* __restrict__ Z, int gy) {
  __shared__ double Ls[(kPartRowsMax + 2 * kPsPf) * (BW + 1)];
  // 1-D grid: the gy column blocks of one interior are consecutive tiles, runs of tiles per XCD (xcd_tile): 48.9 -> 46.6 us
  const int tiles = pv.P * gy;
  const int tile = xcd_tile(tiles);
  if (tile >= tiles) return;
  const int p = tile / gy, by = tile % gy;
  if (by + 1 == gy) part_solve_block<BW, true>(pv, ncols, Lb, Z, Ls, p, by);
  else part_solve_block<BW, false>(pv, ncols, Lb, Z, Ls, p, by);
}

// separator system: T(q,q), T(q,q+1) and the reduced right-hand sides (in place in the separator rows of Z)
// right-hand sides of separator task t: entries e0, e0 + estride, ... of its s3 x ncols block
template <int BW>
__device__ __forceinline__ void part_reduce_rhs(const PartView& pv, int ncols, const double* __restrict__ Lb, const double* __restrict__ Z, int t, int e0, int estride) {
  const int s3 = pv.s3;
  const int c0 = pv.tc0[t], pl = pv.tpl[t], pr = pv.tpr[t], gq = pv.tgq[t];
  const bool own = pv.town[t] != 0;
  const int a0 = pl >= 0 ? pv.i0[pl] : 0, a1 = pl >= 0 ? pv.i1[pl] : 0;
  const int b0 = pr >= 0 ? pv.i0[pr] : 0, b1 = pr >= 0 ? pv.i1[pr] : 0;
  for (int e = e0; e < s3 * ncols; e += estride) {
    const int a = e / ncols, col = e % ncols;
    double r = 0.0;
    if (own) {
      if (pv.Dl == nullptr) r = Z[(long long)(c0 + a) * ncols + col];
      else r = col < pv.CB ? pv.Et[((long long)(col / pv.B) * pv.N3 + (c0 + a)) * pv.B + col % pv.B] : pv.gs[c0 + a];    // (its rows of Z hold zeros)
    }
    if (pl >= 0) {                                       // (clamped rows, dropped products: every load unconditional, as in k_part_reduce)
      double f[BW], z[BW];
#pragma unroll
      for (int jj = 0; jj < BW; ++jj) {
        const int ic = max(a1 - BW + jj, a0);
        f[jj] = band_entry_nc<BW>(Lb, ic, c0 + a);
        z[jj] = Z[(long long)ic * ncols + col];
      }
#pragma unroll
      for (int jj = 0; jj < BW; ++jj) r -= (a1 - BW + jj >= a0) ? f[jj] * z[jj] : 0.0;
    }
    if (pr >= 0) {
      double f[BW], z[BW];
#pragma unroll
      for (int jj = 0; jj < BW; ++jj) {
        const int ic = min(b0 + jj, b1 - 1);
        f[jj] = band_entry_nc<BW>(Lb, ic, c0 + a);
        z[jj] = Z[(long long)ic * ncols + col];
      }
#pragma unroll
      for (int jj = 0; jj < BW; ++jj) r -= (b0 + jj < b1) ? f[jj] * z[jj] : 0.0;
    }
    pv.R[((long long)gq * s3 + a) * ncols + col] = r;
    if (pv.Dl != nullptr && col < pv.CB) pv.Dl[((long long)gq * s3 + a) * pv.CB + col] = r;
  }
}
// what: 1 = the matrix blocks T, U (workgroups with blockIdx.y == 0), 2 = the right-hand sides, 3 = both
template <int BW>
__global__ __launch_bounds__(256) void k_part_reduce(PartView pv, int ncols, const double* __restrict__ Lb, const double* __restrict__ Z, int what = 3) {
  const int t = blockIdx.x, s3 = pv.s3, st = 2 * s3;
  const int c0 = pv.tc0[t], pl = pv.tpl[t], pr = pv.tpr[t], gq = pv.tgq[t];
  const bool own = pv.town[t] != 0;
  const int a0 = pl >= 0 ? pv.i0[pl] : 0, a1 = pl >= 0 ? pv.i1[pl] : 0;     // interior before the separator
  const int b0 = pr >= 0 ? pv.i0[pr] : 0, b1 = pr >= 0 ? pv.i1[pr] : 0;     // interior after it
  // only the last BW rows of the interior before and the first BW rows of the one after couple to the separator; the
  // loops run over exactly BW rows with the out-of-range ones masked, so that all their loads are issued together
  const double* VWa = pv.VW + ((long long)(pl >= 0 ? pl : 0) * kPartRowsMax) * st;
  const double* VWb = pv.VW + ((long long)(pr >= 0 ? pr : 0) * kPartRowsMax) * st;
  for (int e = threadIdx.x; (what & 1) && blockIdx.y == 0 && e < s3 * s3; e += blockDim.x) {
    const int a = e / s3, b = e % s3;
    // (rows outside the interior are clamped to its edge and their products dropped: no load sits behind a condition, all 2 x BW
    // x 2 - 3 of them are in flight together -- with a branch per row this loop was a chain of ~30 memory round trips, 12 us)
    double tt = own ? band_entry<BW>(Lb, c0 + a, c0 + b) : 0.0, u = 0.0;
    if (pl >= 0) {
      double f[BW], w[BW];
#pragma unroll
      for (int jj = 0; jj < BW; ++jj) {
        const int ic = max(a1 - BW + jj, a0);
        f[jj] = band_entry_nc<BW>(Lb, ic, c0 + a);
        w[jj] = VWa[(long long)(ic - a0) * st + s3 + b];
      }
#pragma unroll
      for (int jj = 0; jj < BW; ++jj) tt -= (a1 - BW + jj >= a0) ? f[jj] * w[jj] : 0.0;                             // F_right(q)^T W_q
    }
    if (pr >= 0) {
      double f[BW], wv[BW], ww[BW];
#pragma unroll
      for (int jj = 0; jj < BW; ++jj) {
        const int ic = min(b0 + jj, b1 - 1);
        f[jj] = band_entry_nc<BW>(Lb, ic, c0 + a);
        wv[jj] = VWb[(long long)(ic - b0) * st + b];
        ww[jj] = VWb[(long long)(ic - b0) * st + s3 + b];
      }
#pragma unroll
      for (int jj = 0; jj < BW; ++jj) {
        const bool in = b0 + jj < b1;
        tt -= in ? f[jj] * wv[jj] : 0.0;                                                                             // F_left(q+1)^T V_{q+1}
        u -= in ? f[jj] * ww[jj] : 0.0;                                                                              // F_left(q+1)^T W_{q+1}
      }
    }
    pv.T[((long long)gq * s3 + a) * s3 + b] = tt;
    pv.U[((long long)gq * s3 + a) * s3 + b] = (gq + 1 < pv.m) ? u : 0.0;
  }
  // the right-hand sides: gridDim.y workgroups per separator share the entries (few dependent load rounds each)
  if (what & 2) part_reduce_rhs<BW>(pv, ncols, Lb, Z, t, (int)(blockIdx.y * blockDim.x + threadIdx.x), (int)(gridDim.y * blockDim.x));
}

// block-tridiagonal Cholesky of the separator system by ONE wavefront (no workgroup barriers needed between
// the tiny dependent steps): T[q] <- C_q (lower factor of the q-th pivot block), U[q] <- L(q+1,q) = U_q^T C_q^-T
template <int S3>
__global__ __launch_bounds__(64) void k_sep_factor(PartView pv, int* __restrict__ fail) {
  __shared__ double Cq[S3 * S3];
  __shared__ double Lq[S3 * S3];
  __shared__ double Uq[S3 * S3];
  const int nq = pv.m, lane = threadIdx.x;
  constexpr int kPer = (S3 * S3 + 63) / 64;          // entries of T_q / U_q held per lane
  double tn[kPer], un[kPer];                         // next step's blocks, fetched one step ahead
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    const int e = lane + 64 * r;
    tn[r] = (nq > 0 && e < S3 * S3) ? pv.T[e] : 0.0;
    un[r] = (nq > 0 && e < S3 * S3) ? pv.U[e] : 0.0;
  }
  for (int q = 0; q < nq; ++q) {
    double* Tq = pv.T + (long long)q * S3 * S3;
    double* Ug = pv.U + (long long)q * S3 * S3;
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int e = lane + 64 * r;
      if (e < S3 * S3) {
        const int a = e / S3, b = e % S3;
        double v = tn[r];
        if (q > 0) for (int k = 0; k < S3; ++k) v -= Lq[a * S3 + k] * Lq[b * S3 + k];
        Cq[e] = v;
        Uq[e] = un[r];
      }
    }
    if (q + 1 < nq) {
#pragma unroll
      for (int r = 0; r < kPer; ++r) {
        const int e = lane + 64 * r;
        if (e < S3 * S3) { tn[r] = Tq[S3 * S3 + e]; un[r] = Ug[S3 * S3 + e]; }
      }
    }
    __syncthreads();
    // right-looking Cholesky of the S3 x S3 block, lanes over the trailing entries
    for (int k = 0; k < S3; ++k) {
      double d = Cq[k * S3 + k];
      if (!(d > 0.0)) { if (lane == 0) fail[0] = 3; d = 1.0; }
      d = sqrt(d);
      __syncthreads();
      if (lane == 0) Cq[k * S3 + k] = d;
      if (lane > k && lane < S3) Cq[lane * S3 + k] /= d;
      __syncthreads();
      for (int e = lane; e < S3 * S3; e += 64) {
        const int a = e / S3, b = e % S3;
        if (a > k && b > k && b <= a) Cq[e] -= Cq[a * S3 + k] * Cq[b * S3 + k];
      }
      __syncthreads();
    }
    for (int e = lane; e < S3 * S3; e += 64) Tq[e] = (e / S3 == e % S3) ? 1.0 / Cq[e] : Cq[e];      // diagonal inverted for k_sep_rhs
    if (q + 1 < nq) {
      if (lane < S3) {                       // row `lane` of L(q+1,q): solve C_q l = U_q[:, lane]
        double l[S3];
        for (int k = 0; k < S3; ++k) {
          double sacc = Uq[k * S3 + lane];
          for (int jj = 0; jj < k; ++jj) sacc -= Cq[k * S3 + jj] * l[jj];
          l[k] = sacc / Cq[k * S3 + k];
        }
        for (int k = 0; k < S3; ++k) Lq[lane * S3 + k] = l[k];
      }
      __syncthreads();
      for (int e = lane; e < S3 * S3; e += 64) Ug[e] = Lq[e];
    }
    __syncthreads();
  }
}

// forward / backward substitution of every right-hand side through the factored separator system: one thread per
// column, no synchronisation (the factors are read-only and identical for all lanes, so they are fetched as
// broadcasts through the caches); the next step's right-hand-side rows are fetched before the current step's
// dependent chain, and the diagonal of C_q is stored inverted (k_sep_factor) so the chain has no divisions.
template <int S3>
__global__ __launch_bounds__(64) void k_sep_rhs(PartView pv, int ncols) {
  double* __restrict__ Rr = pv.R;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncols) return;
  const int nq = pv.m;
  double prev[S3], rv[S3], nx[S3];
#pragma unroll
  for (int a = 0; a < S3; ++a) { prev[a] = 0.0; nx[a] = Rr[((long long)(0) * S3 + a) * ncols + col]; }
  for (int q = 0; q < nq; ++q) {
    const double* __restrict__ Cq = pv.T + (long long)q * S3 * S3;
    const double* __restrict__ Lq = pv.U + (long long)(q - 1) * S3 * S3;     // L(q, q-1)
#pragma unroll
    for (int a = 0; a < S3; ++a) rv[a] = nx[a];
    if (q + 1 < nq) {
#pragma unroll
      for (int a = 0; a < S3; ++a) nx[a] = Rr[((long long)(q + 1) * S3 + a) * ncols + col];
    }
    if (q > 0) {
#pragma unroll
      for (int a = 0; a < S3; ++a)
#pragma unroll
        for (int k = 0; k < S3; ++k) rv[a] -= Lq[a * S3 + k] * prev[k];
    }
#pragma unroll
    for (int a = 0; a < S3; ++a) {
      double sacc = rv[a];
#pragma unroll
      for (int jj = 0; jj < S3; ++jj) if (jj < a) sacc -= Cq[a * S3 + jj] * rv[jj];
      rv[a] = sacc * Cq[a * S3 + a];            // diagonal stored as 1 / C(a,a)
    }
#pragma unroll
    for (int a = 0; a < S3; ++a) { prev[a] = rv[a]; Rr[((long long)(q) * S3 + a) * ncols + col] = rv[a]; }
  }
#pragma unroll
  for (int a = 0; a < S3; ++a) nx[a] = prev[a];
  for (int q = nq - 1; q >= 0; --q) {
    const double* __restrict__ Cq = pv.T + (long long)q * S3 * S3;
    const double* __restrict__ Ln = pv.U + (long long)q * S3 * S3;           // L(q+1, q)
#pragma unroll
    for (int a = 0; a < S3; ++a) rv[a] = nx[a];
    if (q > 0) {
#pragma unroll
      for (int a = 0; a < S3; ++a) nx[a] = Rr[((long long)(q - 1) * S3 + a) * ncols + col];
    }
    if (q + 1 < nq) {
#pragma unroll
      for (int k = 0; k < S3; ++k)
#pragma unroll
        for (int a = 0; a < S3; ++a) rv[a] -= Ln[k * S3 + a] * prev[k];
    }
#pragma unroll
    for (int a = S3 - 1; a >= 0; --a) {
      double sacc = rv[a];
#pragma unroll
      for (int jj = 0; jj < S3; ++jj) if (jj > a) sacc -= Cq[jj * S3 + a] * rv[jj];
      rv[a] = sacc * Cq[a * S3 + a];
    }
#pragma unroll
    for (int a = 0; a < S3; ++a) { prev[a] = rv[a]; Rr[((long long)(q) * S3 + a) * ncols + col] = rv[a]; }
  }
}

// ---- block cyclic reduction of the separator system ---------------------------------------------------------
// The sequential block-tridiagonal Cholesky above walks P-1 dependent steps.  Cyclic reduction needs only
// ceil(log2(P)) of them: at stride h the nodes j with (j+1)/h odd are eliminated, x_j = D_j^-1 (r_j - A_j x_{j-h} -
// C_j x_{j+h}), and folded into their neighbours at distance h, which form the next (half as long) block-tridiagonal
// system.  Every Schur complement of an SPD matrix is SPD, so no pivoting is needed.  Stored per node, at the level
// that eliminates it: Dinv_j (in U2), Ha_j = Dinv_j A_j, Hc_j = Dinv_j C_j; only the coupling to the right neighbour,
// C_j, is carried through the levels (A_j = C_{j-h}^T).
//   survivors i:  D_i -= C_{i-h}^T Hc_{i-h} + C_i Ha_{i+h},   C_i <- -C_i Hc_{i+h},   r_i -= Hc_{i-h}^T r_{i-h} + Ha_{i+h}^T r_{i+h}
// k_sep_bcr_level / k_sep_bcr_tail: the matrix part, a wavefront per survivor (bcr_survivor below);
// k_sep_bcr_rhs: the right-hand sides, kBcrCols columns per workgroup staged in LDS through all levels and back.
// One S3 x S3 product on the fp64 matrix cores, one wavefront: acc += op(A) op(B), the blocks zero-padded to 16 x 16.
// ta: A is read transposed; tb: B is read transposed.  Fragment layout of v_mfma_f64_16x16x4: lane l holds
// A[l&15][k0 + (l>>4)] and B[k0 + (l>>4)][l&15].
using bcr_d4 = __attribute__((ext_vector_type(4))) double;
template <int S3>
struct BcrFrag { double a[(S3 + 3) / 4], b[(S3 + 3) / 4]; };
template <int S3>
__device__ __forceinline__ void bcr_load(BcrFrag<S3>& f, const double* __restrict__ A, bool ta, const double* __restrict__ Bm, bool tb, bool on) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s_ = 0; s_ < (S3 + 3) / 4; ++s_) {
    const int k = 4 * s_ + lk;
    const bool ok = on && lr < S3 && k < S3;
    f.a[s_] = ok ? (ta ? A[k * S3 + lr] : A[lr * S3 + k]) : 0.0;
    f.b[s_] = ok ? (tb ? Bm[lr * S3 + k] : Bm[k * S3 + lr]) : 0.0;
  }
}
template <int S3>
__device__ __forceinline__ bcr_d4 bcr_mma(const BcrFrag<S3>& f, bcr_d4 acc) {
#pragma unroll
  for (int s_ = 0; s_ < (S3 + 3) / 4; ++s_) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[s_], f.b[s_], acc, 0, 0, 0);
  return acc;
}

#ifndef MVUS_BCR_WAVES
#define MVUS_BCR_WAVES 16
#endif
#ifndef MVUS_BCR_TAIL_NS
#define MVUS_BCR_TAIL_NS 16
#endif
constexpr int kBcrWaves = MVUS_BCR_WAVES;      // wavefronts of the one workgroup that runs the last levels (S3 = 15: half, LDS)
constexpr int kBcrTailNs = MVUS_BCR_TAIL_NS;   // levels with more survivors than this get a launch of their own (one wavefront per survivor)


// One survivor of one level, by ONE wavefront and without any exchange with other wavefronts: the survivor i = 2h(s+1)-1
// inverts the diagonal blocks of BOTH eliminated neighbours jL = i-h and jR = i+h itself (its neighbours two places on
// do the same work again -- 2x redundant and free, the level is latency bound), forms Hc_jL, Ha_jR, Hc_jR on the matrix
// cores and applies them to its own D_i and C_i, in place: at one level D_i and C_i are touched by this wavefront only,
// the blocks of eliminated nodes are only read.  The inverses therefore go to a buffer of their own (U2), not over T.
// Stored for k_sep_bcr_rhs: Dinv_j, Ha_j, Hc_j of jR (and of jL for the first survivor, whose left neighbour nobody else
// owns).  s = 0 with no survivor at all is the last level: the one remaining node is inverted.
// w: 3 * S3 * S3 doubles of LDS private to the wavefront.
// The blocks read here may have been written by ANOTHER wavefront of the same workgroup one level earlier (k_sep_bcr_tail), and a
// 128-byte line can straddle two 648-byte blocks: a line fetched into the CU's L1 for one block before a neighbour's store to
// the other landed would be stale at the next level.  Agent-scope loads go to L2, where the stores are.
__device__ __forceinline__ double bcr_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// lane J of every row of 16 lanes, to all lanes of that row (v_mov_b64_dpp row_newbcast:J -- the one DPP control 64-bit operands have here)
template <int J> __device__ __forceinline__ double row_bcast(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, false); }
// Gauss-Jordan inverse of an S3 x S3 SPD block held a row per lane (lane a of a row of 16 lanes: block row a; lanes >= S3 carry
// zeros): step K broadcasts the pivot row with DPP row_newbcast:K.  Template recursion: the DPP control is an immediate.
template <int S3, int K>
struct BcrGaussJordan {
  static __device__ __forceinline__ void run(double (&row)[S3], int a, bool& bad) {
    double pr[S3];
#pragma unroll
    for (int b = 0; b < S3; ++b) pr[b] = row_bcast<K>(row[b]);
    double piv = pr[K];
    bad |= !(piv > 0.0);
    piv = piv > 0.0 ? piv : 1.0;
    double ip = __builtin_amdgcn_rcp(piv);
    ip = ip * (2.0 - piv * ip);
    ip = ip * (2.0 - piv * ip);
    if (a == K) {
#pragma unroll
      for (int b = 0; b < S3; ++b) row[b] = (b == K) ? ip : row[b] * ip;
    } else {
      const double f = row[K] * ip;
#pragma unroll
      for (int b = 0; b < S3; ++b) row[b] = (b == K) ? -f : row[b] - f * pr[b];
    }
    BcrGaussJordan<S3, K + 1>::run(row, a, bad);
  }
};
template <int S3>
struct BcrGaussJordan<S3, S3> { static __device__ __forceinline__ void run(double (&)[S3], int, bool&) {} };

template <int S3, bool WT = false>      // WT: T_i and C_i leave as agent-scope write-through stores (readers in other workgroups of the SAME launch: k_sep_bcr_levels)
__device__ __forceinline__ void bcr_survivor(const PartView& pv, int h, int s, double* __restrict__ w, int* __restrict__ fail) {
  constexpr int SS = S3 * S3, KS = (S3 + 3) / 4;
  const int m = pv.m, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int i = 2 * h * (s + 1) - 1, jL = i - h, jR = i + h;
  const bool has_i = i < m, right = jR < m, right2 = jR + h < m;
  double* __restrict__ Cc = pv.U;
  // ---- all global operands, issued together ----
  // (block row a of neighbour nl in lane 16 nl + a: each neighbour's rows inside one row of 16 lanes, so that the pivot row of the
  // elimination below is a DPP row broadcast -- v_mov_b64_dpp row_newbcast -- instead of two ds_bpermute per entry)
  static_assert(S3 <= 16, "a block row per lane inside one DPP row");
  const int nl = lane >> 4, a = lane & 15;
  const bool act = a < S3 && (nl == 0 || (nl == 1 && right));
  const double* Tsrc = pv.T + (long long)(nl == 0 ? jL : (right ? jR : 0)) * SS + (a < S3 ? a : 0) * S3;
  double row[S3];
#pragma unroll
  for (int b = 0; b < S3; ++b) row[b] = act ? bcr_ld(Tsrc + b) : (a == b ? 1.0 : 0.0);
  double cL[KS], ci[KS], cR[KS], ti[4];
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) {
    const int k = 4 * s_ + lk;
    const bool ok = lr < S3 && k < S3;
    cL[s_] = (ok && has_i) ? bcr_ld(Cc + (long long)jL * SS + k * S3 + lr) : 0.0;        // C_jL[k][lr]: B operand of Hc_jL, A operand (transposed) of the D update
    ci[s_] = (ok && right) ? bcr_ld(Cc + (long long)i * SS + lr * S3 + k) : 0.0;         // C_i[lr][k]: B operand (transposed) of Ha_jR, A operand of both updates
    cR[s_] = (ok && right2) ? bcr_ld(Cc + (long long)jR * SS + k * S3 + lr) : 0.0;       // C_jR[k][lr]
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int rw = lk + 4 * r; ti[r] = (has_i && rw < S3 && lr < S3) ? bcr_ld(pv.T + (long long)i * SS + rw * S3 + lr) : 0.0; }
  // ---- Dinv of both neighbours: a block row per lane, the pivot row of each Gauss-Jordan step broadcast with shuffles ----
  bool bad = false;
  BcrGaussJordan<S3, 0>::run(row, a, bad);
  if (bad && act) fail[0] = 3;
  if (nl < 2 && a < S3) {
#pragma unroll
    for (int b = 0; b < S3; ++b) w[nl * SS + a * S3 + b] = row[b];
    if (nl == 0 ? s == 0 : right) {
      double* Dg = pv.U2 + (long long)(nl == 0 ? jL : jR) * SS + a * S3;
#pragma unroll
      for (int b = 0; b < S3; ++b) Dg[b] = row[b];
    }
  }
  lds_wave_sync();
  // ---- Hc_jL = Dinv_jL C_jL, Ha_jR = Dinv_jR C_i^T, Hc_jR = Dinv_jR C_jR ----
  double aL[KS], aR[KS];
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) {
    const int k = 4 * s_ + lk;
    const bool ok = lr < S3 && k < S3;
    aL[s_] = ok ? w[lr * S3 + k] : 0.0;
    aR[s_] = (ok && right) ? w[SS + lr * S3 + k] : 0.0;
  }
  bcr_d4 hcl{0.0, 0.0, 0.0, 0.0}, har{0.0, 0.0, 0.0, 0.0}, hcr{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) {
    hcl = __builtin_amdgcn_mfma_f64_16x16x4f64(aL[s_], cL[s_], hcl, 0, 0, 0);
    har = __builtin_amdgcn_mfma_f64_16x16x4f64(aR[s_], ci[s_], har, 0, 0, 0);
    hcr = __builtin_amdgcn_mfma_f64_16x16x4f64(aR[s_], cR[s_], hcr, 0, 0, 0);
  }
  lds_wave_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rw = lk + 4 * r;
    if (rw < S3 && lr < S3) {
      const int e = rw * S3 + lr;
      w[e] = hcl[r]; w[SS + e] = har[r]; w[2 * SS + e] = hcr[r];
      if (s == 0) { pv.Ha[(long long)jL * SS + e] = 0.0; pv.Hc[(long long)jL * SS + e] = hcl[r]; }
      if (right) { pv.Ha[(long long)jR * SS + e] = har[r]; pv.Hc[(long long)jR * SS + e] = hcr[r]; }
    }
  }
  if (!has_i) return;                                  // wave-uniform: the last level has no survivor
  lds_wave_sync();
  // ---- D_i -= C_jL^T Hc_jL + C_i Ha_jR,  C_i <- -C_i Hc_jR ----
  bcr_d4 dacc{0.0, 0.0, 0.0, 0.0}, cacc{0.0, 0.0, 0.0, 0.0};
  double b1[KS], b2[KS], b3[KS];
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) {
    const int k = 4 * s_ + lk;
    const bool ok = lr < S3 && k < S3;
    b1[s_] = ok ? w[k * S3 + lr] : 0.0;
    b2[s_] = ok ? w[SS + k * S3 + lr] : 0.0;
    b3[s_] = ok ? w[2 * SS + k * S3 + lr] : 0.0;
  }
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(cL[s_], b1[s_], dacc, 0, 0, 0);
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(ci[s_], b2[s_], dacc, 0, 0, 0);
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(ci[s_], b3[s_], cacc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rw = lk + 4 * r;
    if (rw < S3 && lr < S3) {
      if constexpr (WT) {
        __hip_atomic_store(pv.T + (long long)i * SS + rw * S3 + lr, ti[r] - dacc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(Cc + (long long)i * SS + rw * S3 + lr, 0.0 - cacc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        pv.T[(long long)i * SS + rw * S3 + lr] = ti[r] - dacc[r];
        Cc[(long long)i * SS + rw * S3 + lr] = 0.0 - cacc[r];
      }
    }
  }
}

// one level in a launch of its own: one wavefront (= workgroup) per survivor
template <int S3>
__global__ __launch_bounds__(64) void k_sep_bcr_level(PartView pv, int h, int* __restrict__ fail) {
  __shared__ double w[3 * S3 * S3];
  bcr_survivor<S3>(pv, h, blockIdx.x, w, fail);
}
// ALL the wide levels in ONE launch (round 6): workgroup b is survivor s of level L (levels in launch order: the first ones have the
// lowest block indices), and a survivor of level L >= 1 waits until the three nodes it reads -- itself and its two neighbours, all
// survivors of level L - 1 -- carry the mark of that level.  Hand-over as in k_rcs_factor (the CDNA guide's Guideline 16): the
// producer's T_i and C_i go out as agent-scope (sc1, write-through) stores, s_waitcnt vmcnt(0), then ONE lane stores done[i] = epoch + L
// + 1 with a relaxed agent-scope atomic; the consumer polls with relaxed agent-scope loads (s_sleep, BOUNDED: a time-out raises
// fail[0] = kFailHandoverCode and the sticky fail[2], the host then repeats the solve with a launch per level) and reads the blocks with
// agent-scope loads (bcr_ld).  `epoch` grows by 8 per solve: the marks are never reset.  Three launches of ~5 us (a ramp and two or
// three memory round trips each) become one of ~7.
struct BcrLevels { int nlev; int first[4]; int h[4]; };      // first[l]: block index of level l's survivor 0
template <int S3>
__global__ __launch_bounds__(64) void k_sep_bcr_levels(PartView pv, BcrLevels lv, unsigned* __restrict__ done, unsigned epoch, unsigned spin_limit, int* __restrict__ fail) {
  __shared__ double w[3 * S3 * S3];
  int L = 0;
#pragma unroll
  for (int l = 1; l < 4; ++l) if (l < lv.nlev && (int)blockIdx.x >= lv.first[l]) L = l;
  const int h = lv.h[L], s = (int)blockIdx.x - lv.first[L];
  const int i = 2 * h * (s + 1) - 1, jL = i - h, jR = i + h;
  if (L > 0 && threadIdx.x == 0) {
    const unsigned want = epoch + (unsigned)L;
    unsigned spins = 0;
    while (true) {
      const unsigned a = __hip_atomic_load(done + jL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned b = i < pv.m ? __hip_atomic_load(done + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
      const unsigned c = jR < pv.m ? __hip_atomic_load(done + jR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
      if ((int)(a - want) >= 0 && (int)(b - want) >= 0 && (int)(c - want) >= 0) break;
      __builtin_amdgcn_s_sleep(4);
      if (++spins > spin_limit) { fail[0] = kFailHandoverCode; fail[2] = 1; break; }
    }
  }
  __builtin_amdgcn_wave_barrier();
  bcr_survivor<S3, true>(pv, h, s, w, fail);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && i < pv.m) __hip_atomic_store(done + i, epoch + (unsigned)L + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the remaining levels from stride h0 on, one workgroup: a wavefront per survivor, one barrier per level
constexpr int bcr_tail_waves(int s3) { return s3 > 9 ? kBcrWaves / 2 : kBcrWaves; }
template <int S3>
__device__ __forceinline__ void bcr_tail_body(const PartView& pv, int h0, int* __restrict__ fail, double* __restrict__ w) {
  constexpr int NW = bcr_tail_waves(S3);
  const int wave = threadIdx.x >> 6;
  for (int h = h0; h <= pv.m; h <<= 1) {
    const int ns = pv.m / (2 * h), nt = ns > 0 ? ns : 1;
    for (int s = wave; s < nt; s += NW) bcr_survivor<S3>(pv, h, s, w + wave * 3 * S3 * S3, fail);
    // The next level's loads (bcr_ld: agent scope, they bypass the CU's L1 and read L2) must see this level's plain stores of the
    // OTHER wavefronts of this workgroup.  All of them run on one CU, hence behind one L2: what is needed is that the stores have
    // ARRIVED there before anybody passes the barrier, i.e. that their acknowledgements are in (vmcnt = 0) -- a workgroup-scope
    // barrier by itself does not wait for that.  (A full agent-scope release, __threadfence(), also writes the XCD's L2 back for the
    // benefit of other XCDs -- `buffer_wbl2 sc1` -- which nothing here needs: measured +15 us per solve.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}
template <int S3>
__global__ __launch_bounds__(bcr_tail_waves(S3) * 64) void k_sep_bcr_tail(PartView pv, int h0, int* __restrict__ fail) {
  __shared__ double w[bcr_tail_waves(S3) * 3 * S3 * S3];
  bcr_tail_body<S3>(pv, h0, fail, w);
}
// The last levels of the cyclic reduction run in ONE workgroup (29 us of barriers); the right-hand sides of the separator system
// (k_part_reduce's second half, ~12 us over ~460 workgroups) need nothing from it: same launch, workgroup 0 = the tail.
template <int BW, int S3>
__global__ __launch_bounds__(bcr_tail_waves(S3) * 64) void k_bcr_tail_and_reduce_rhs(PartView pv, int h0, int run_tail, int* __restrict__ fail, int ncols,
                                                                                    const double* __restrict__ Lb, const double* __restrict__ Z, int ny) {
  __shared__ double w[bcr_tail_waves(S3) * 3 * S3 * S3];
  if (blockIdx.x == 0) { if (run_tail) bcr_tail_body<S3>(pv, h0, fail, w); return; }
  const int idx = (int)blockIdx.x - 1, t = idx / ny, by = idx % ny;
  part_reduce_rhs<BW>(pv, ncols, Lb, Z, t, (int)(by * blockDim.x + threadIdx.x), (int)(ny * blockDim.x));
}

#ifndef MVUS_BCR_COLS
#define MVUS_BCR_COLS 1
#endif
constexpr int kBcrCols = MVUS_BCR_COLS;       // columns per workgroup (one: more workgroups, fewer items per level and thread)
template <int S3, int TC>
__global__ __launch_bounds__(256) void k_sep_bcr_rhs(PartView pv, int ncols) {
  double* __restrict__ Rr = pv.R;
  constexpr int SS = S3 * S3;
  extern __shared__ double bcr_lds[];
  const int m = pv.m, tid = threadIdx.x;
  double* rs = bcr_lds;                       // [m][S3][TC]
  double* xs = bcr_lds + (size_t)m * S3 * TC;
  const int col0 = blockIdx.x * TC;
  for (int e = tid; e < m * S3 * TC; e += 256) {
    const int c = e % TC, a = (e / TC) % S3, q = e / (TC * S3);
    rs[e] = col0 + c < ncols ? Rr[((long long)q * S3 + a) * ncols + col0 + c] : 0.0;
  }
  __syncthreads();
  int h = 1;
  for (; 2 * h <= m; h <<= 1) {
    const int ns = m / (2 * h);
    for (int e = tid; e < ns * S3 * TC; e += 256) {
      const int c = e % TC, a = (e / TC) % S3, i = 2 * h * (e / (TC * S3) + 1) - 1;
      double acc = rs[(i * S3 + a) * TC + c];
      const double* Hl = pv.Hc + (long long)(i - h) * SS + a;
      const double* rl = rs + (i - h) * S3 * TC + c;
#pragma unroll
      for (int k = 0; k < S3; ++k) acc -= Hl[k * S3] * rl[k * TC];
      if (i + h < m) {
        const double* Hr = pv.Ha + (long long)(i + h) * SS + a;
        const double* rr = rs + (i + h) * S3 * TC + c;
#pragma unroll
        for (int k = 0; k < S3; ++k) acc -= Hr[k * S3] * rr[k * TC];
      }
      rs[(i * S3 + a) * TC + c] = acc;
    }
    __syncthreads();
  }
  for (; h >= 1; h >>= 1) {
    const int ne = (m / h + 1) / 2;
    for (int e = tid; e < ne * S3 * TC; e += 256) {
      const int c = e % TC, a = (e / TC) % S3, j = h * (2 * (e / (TC * S3)) + 1) - 1;
      const double* Di = pv.U2 + (long long)j * SS + a * S3;
      const double* rj = rs + j * S3 * TC + c;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < S3; ++k) acc += Di[k] * rj[k * TC];
      if (j - h >= 0) {
        const double* Hl = pv.Ha + (long long)j * SS + a * S3;
        const double* xl = xs + (j - h) * S3 * TC + c;
#pragma unroll
        for (int k = 0; k < S3; ++k) acc -= Hl[k] * xl[k * TC];
      }
      if (j + h < m) {
        const double* Hr = pv.Hc + (long long)j * SS + a * S3;
        const double* xr = xs + (j + h) * S3 * TC + c;
#pragma unroll
        for (int k = 0; k < S3; ++k) acc -= Hr[k] * xr[k * TC];
      }
      xs[(j * S3 + a) * TC + c] = acc;
    }
    __syncthreads();
  }
  for (int e = tid; e < m * S3 * TC; e += 256) {
    const int c = e % TC, a = (e / TC) % S3, q = e / (TC * S3);
    if (col0 + c < ncols) Rr[((long long)q * S3 + a) * ncols + col0 + c] = xs[e];
  }
}

// ---- time shards, round 6: TWO-LEVEL elimination of the separators ------------------------------------------------------------------
// Rank r's chain of separators is  [ghost G = the cut closing rank r-1]  s_1 .. s_k  [cut K closing rank r]  (global numbers q0-1, q0 ..
// q0+k-1, q0+k; no ghost on the first rank, no cut on the last).  k_part_reduce has left every block of it in the globally numbered
// [T | U | R]: T_q, U_q = T(q, q+1), the reduced right-hand sides R_q -- of G only the part interior 0 contributes, of K only the part
// the last interior contributes.  Rounds 3-5 summed the WHOLE system over the ranks (3.4 MB at configs[3]) and every rank solved all m
// separators.  Here the local separators L = {s_1 .. s_k} are a second-level interior:
//     L [Y | V | W] = [R_L | C_LG | C_LK]      the rank's own block-tridiagonal system, solved by the cyclic reduction as it is, on a
//                                               right-hand side with 2 s3 extra columns (C_LG = U_G^T in s_1's rows, C_LK = U_{s_k} in s_k's)
//     T'_G += T_G - U_G V_1      U'_G (= T'(G,K)) += -U_G W_1      R'_G += R_G - U_G Y_1            (subscript: the node's rows of Y, V, W)
//     T'_K += T_K - U_{s_k}^T W_k                                  R'_K += R_K - U_{s_k}^T Y_k
// the (world - 1)-node cut system [T' | U' | R'] is what the ranks sum (0.3 MB), every rank solves it (k_sep_factor, k_sep_rhs), and
//     X_L = Y - V X_G - W X_K
// goes into the globally numbered R beside X_G and X_K: k_back_correct, k_part_back and the Schur product's correction rows read it there
// as before.  The correction term of a cut separator splits by itself: rank r adds (its part of R_K)^T X_K, rank r+1 (its part)^T X_K.
template <int S3>
__global__ __launch_bounds__(256) void k_sep2_build(PartView pv, int q0, int k, int has_ghost, int has_cut, int ncols, double* __restrict__ Rloc,
                                                    double* __restrict__ CG, double* __restrict__ CK, int corr_q0 = 0, int corr_rows = 0, int CB = 0) {
  constexpr int SS = S3 * S3;
  const int nc2 = ncols + 2 * S3;
  const long long total = (long long)k * S3 * nc2;
  // (the correction rows of the Schur product need this rank's parts of the reduced right-hand sides R_S as they are BEFORE the solve
  // overwrites them: their camera columns go to pv.Dl here -- a rectangle copy of its own until round 6)
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)corr_rows * CB; e += (long long)gridDim.x * blockDim.x) {
    const long long r = (long long)corr_q0 * S3 + e / CB;
    pv.Dl[r * CB + e % CB] = pv.R[r * ncols + e % CB];
  }
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total + 2 * SS; e += (long long)gridDim.x * blockDim.x) {
    if (e >= total) {                                   // the two coupling blocks, saved: the cyclic reduction overwrites U of its last node
      const int t = (int)(e - total), which = t / SS, idx = t % SS;
      if (which == 0) CG[idx] = has_ghost ? pv.U[(long long)(q0 - 1) * SS + idx] : 0.0;
      else CK[idx] = (has_cut && k > 0) ? pv.U[(long long)(q0 + k - 1) * SS + idx] : 0.0;
      continue;
    }
    const int col = (int)(e % nc2), a = (int)((e / nc2) % S3), j = (int)(e / ((long long)nc2 * S3));
    double v = 0.0;
    if (col < ncols) v = pv.R[((long long)(q0 + j) * S3 + a) * ncols + col];
    else if (col < ncols + S3) { if (j == 0 && has_ghost) v = pv.U[(long long)(q0 - 1) * SS + (col - ncols) * S3 + a]; }            // U_G^T
    else if (j == k - 1 && has_cut) v = pv.U[(long long)(q0 + k - 1) * SS + a * S3 + (col - ncols - S3)];                          // U_{s_k}
    Rloc[e] = v;
  }
}
// the rank's contribution to the cut system (cut node c closes rank c: G = node rank - 1, K = node rank); EVERY entry of
// cutbuf = [T' | U' | R'] is written -- zeros for the nodes of other ranks (a memset of its own until round 6)
template <int S3>
__global__ __launch_bounds__(256) void k_sep2_reduce(PartView pv, int q0, int k, int has_ghost, int has_cut, int rank, int ncut, int ncols,
                                                     const double* __restrict__ Rloc, const double* __restrict__ CG, const double* __restrict__ CK,
                                                     double* __restrict__ cutbuf) {
  constexpr int SS = S3 * S3;
  const int nc2 = ncols + 2 * S3;
  const long long nT = (long long)ncut * SS, total = 2 * nT + (long long)ncut * S3 * ncols;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int sec = e < nT ? 0 : (e < 2 * nT ? 1 : 2);                         // T', U', R'
    const long long w = e - (sec == 0 ? 0 : (sec == 1 ? nT : 2 * nT));
    const int node = (int)(w / (sec == 2 ? S3 * ncols : SS)), o = (int)(w % (sec == 2 ? S3 * ncols : SS));
    const int side = (has_ghost && node == rank - 1) ? 0 : ((has_cut && node == rank) ? 1 : -1);      // 0: the ghost G, 1: the cut K
    double v = 0.0;
    if (side >= 0) {
      const long long gq = side == 0 ? q0 - 1 : q0 + k;
      if (sec == 0) {                                   // T'
        const int a = o / S3, b = o % S3;
        v = pv.T[gq * SS + o];
        if (k > 0) {
          if (side == 0) { for (int c = 0; c < S3; ++c) v -= CG[a * S3 + c] * Rloc[((long long)(0) * S3 + c) * nc2 + ncols + b]; }                   // U_G V_1
          else { for (int c = 0; c < S3; ++c) v -= CK[c * S3 + a] * Rloc[((long long)(k - 1) * S3 + c) * nc2 + ncols + S3 + b]; }                    // U_{s_k}^T W_k
        }
      } else if (sec == 1) {                            // U' = T'(G, K): through the local chain, or -- no local separator -- the direct coupling
        if (side == 0 && has_cut) {
          const int a = o / S3, b = o % S3;
          if (k > 0) { for (int c = 0; c < S3; ++c) v -= CG[a * S3 + c] * Rloc[((long long)(0) * S3 + c) * nc2 + ncols + S3 + b]; }                  // -U_G W_1
          else v = pv.U[gq * SS + o];
        }
      } else {                                          // R'
        const int a = o / ncols, col = o % ncols;
        v = pv.R[(gq * S3 + a) * ncols + col];
        if (k > 0) {
          if (side == 0) { for (int c = 0; c < S3; ++c) v -= CG[a * S3 + c] * Rloc[((long long)(0) * S3 + c) * nc2 + col]; }
          else { for (int c = 0; c < S3; ++c) v -= CK[c * S3 + a] * Rloc[((long long)(k - 1) * S3 + c) * nc2 + col]; }
        }
      }
    }
    cutbuf[e] = v;
  }
}
// X_L = Y - V X_G - W X_K and the two cut solutions, into the globally numbered R
template <int S3>
__global__ __launch_bounds__(256) void k_sep2_finish(PartView pv, int q0, int k, int has_ghost, int has_cut, int rank, int ncut, int ncols,
                                                     const double* __restrict__ Rloc, const double* __restrict__ cutbuf) {
  constexpr int SS = S3 * S3;
  const int nc2 = ncols + 2 * S3;
  const double* X2 = cutbuf + 2LL * ncut * SS;          // the solved cut system
  const long long total = (long long)(k + 2) * S3 * ncols;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(e % ncols), a = (int)((e / ncols) % S3), j = (int)(e / ((long long)ncols * S3));
    if (j >= k) {                                       // j = k: X_G, j = k + 1: X_K
      const bool g = j == k;
      if (g ? !has_ghost : !has_cut) continue;
      const int node = g ? rank - 1 : rank;
      const long long gq = g ? q0 - 1 : q0 + k;
      pv.R[(gq * S3 + a) * ncols + col] = X2[((long long)node * S3 + a) * ncols + col];
      continue;
    }
    const double* row = Rloc + ((long long)j * S3 + a) * nc2;
    double v = row[col];
    if (has_ghost) for (int b = 0; b < S3; ++b) v -= row[ncols + b] * X2[((long long)(rank - 1) * S3 + b) * ncols + col];
    if (has_cut) for (int b = 0; b < S3; ++b) v -= row[ncols + S3 + b] * X2[((long long)rank * S3 + b) * ncols + col];
    pv.R[((long long)(q0 + j) * S3 + a) * ncols + col] = v;
  }
}

// interiors: X = Y - V X_{S_{p-1}} - W X_{S_p}.  One thread per right-hand-side column: the 2*S3 separator values of
// its column stay in registers, the rows of V|W are the same for every lane (scalar loads), so each interior entry
// costs one coalesced load, 2*S3 FMAs and one store.
#ifndef MVUS_BACK_ROWS
#define MVUS_BACK_ROWS 16
#endif
constexpr int kBackRows = MVUS_BACK_ROWS;      // rows per workgroup: enough workgroups in flight to hide the load latency
template <int S3>
__global__ __launch_bounds__(64) void k_part_back(PartView pv, int ncols, double* __restrict__ Z, int gy, int gz) {
  // 1-D grid, tiles ordered column block fastest, then row block, then interior, handed to the XCDs in runs (xcd_tile):
  // one XCD reads and writes whole consecutive rows of Z (22.1 -> 20.3 us)
  const int tiles = pv.P * gy * gz;
  const int tile = xcd_tile(tiles);
  if (tile >= tiles) return;
  const int by = tile % gy, bz = (tile / gy) % gz, p = tile / (gy * gz);
  const int col = by * 64 + threadIdx.x;
  if (col >= ncols) return;
  constexpr int st = 2 * S3;
  const int r0 = pv.i0[p], n = pv.i1[p] - r0;
  const int ia = bz * kBackRows, ib = min(n, ia + kBackRows);      // rows of this workgroup
  if (ia >= n) return;
  const bool left = pv.sl[p] >= 0, right = pv.sr[p] >= 0;
  const double* __restrict__ VWp = pv.VW + ((long long)p * kPartRowsMax) * st;
  double zl[S3], zr[S3];
#pragma unroll
  for (int a = 0; a < S3; ++a) {
    zl[a] = left ? pv.R[((long long)(pv.q_off + p - 1) * S3 + a) * ncols + col] : 0.0;
    zr[a] = right ? pv.R[((long long)(pv.q_off + p) * S3 + a) * ncols + col] : 0.0;
  }
  if (right && bz == 0) {          // the solved separator right of this interior goes back into its rows of Z
#pragma unroll
    for (int a = 0; a < S3; ++a) Z[(long long)(pv.sr[p] + a) * ncols + col] = zr[a];
  }
  for (int i = ia; i < ib; ++i) {
    double v = Z[(long long)(r0 + i) * ncols + col];
    const double* __restrict__ vw = VWp + (long long)i * st;
    if (left) {
#pragma unroll
      for (int a = 0; a < S3; ++a) v -= vw[a] * zl[a];
    }
    if (right) {
#pragma unroll
      for (int a = 0; a < S3; ++a) v -= vw[S3 + a] * zr[a];
    }
    Z[(long long)(r0 + i) * ncols + col] = v;
  }
}

// Z[3N][ncols] = [E^T | gs]: the row-major copy of the camera-major cross block that the interior solves work on.  (Until round 5 a
// second copy, Erm[3N][CB], was written here as the Schur product's first operand: the product now reads the camera-major block
// itself -- 34 MB less written at configs[2], 138 MB at configs[3]: k_cholesky_and_rhs 69 -> 55 us there.)
// The same launch also packs the damped band (band_pack_entry: the two are independent element-wise passes over the assembled
// blocks, and every launch costs ~4.7 us before it does anything).
__global__ void k_build_rhs(NEView ne, int ncols, double* __restrict__ Z, const unsigned char* __restrict__ seprow, double lambda, int BW, double* __restrict__ Lb,
                            int* __restrict__ fail, DevProblem dp, int with_diag, double* __restrict__ D, double* __restrict__ gx, int tiles) {
  const long long idx = xcd_tile(tiles) * (long long)blockDim.x + threadIdx.x;      // XCD-aware order: 31.5 -> 25 us (see xcd_tile)
  band_pack_entry(ne, lambda, BW, Lb, fail, dp, with_diag, D, gx, idx);
  if (idx >= (long long)ne.N3 * ncols) return;
  const int r = (int)(idx / ncols), cidx = (int)(idx % ncols);
  if (seprow != nullptr && seprow[r]) Z[idx] = 0.0;            // (one rank: the separators' rows take no part in the product's main range)
  else if (cidx < ne.CB) {
    const int c = cidx / ne.B, k = cidx % ne.B;
    Z[idx] = ne.Et[((long long)c * ne.N3 + r) * ne.B + k];
  } else {
    Z[idx] = ne.gs[r];
  }
}

// The same work as two launches that OVERLAP: k_band_pack (the damped band, 0.2 M entries) first, then ONE launch whose first
// pad8(P) workgroups factorise the interiors (a lone wavefront each, ~37 us of dependent LDS round trips on 155 of 256 CUs)
// while all the others stream the right-hand-side copies (72 MB, bandwidth bound, ~22 us) -- the factorisation needs the
// band only, the copies need nothing from the factorisation.  25 + 37 us in sequence become ~40.
__global__ void k_band_pack(NEView ne, double lambda, int BW, double* __restrict__ Lb, int* __restrict__ fail, DevProblem dp, int with_diag,
                            double* __restrict__ D, double* __restrict__ gx) {
  band_pack_entry(ne, lambda, BW, Lb, fail, dp, with_diag, D, gx, blockIdx.x * (long long)blockDim.x + threadIdx.x);
}
template <int BW>
__global__ __launch_bounds__(256) void k_cholesky_and_rhs(PartView pv, double* __restrict__ Lb, int* __restrict__ fail, int chol_blocks,
                                                          NEView ne, int ncols, double* __restrict__ Z, int tiles, const unsigned char* __restrict__ seprow) {
  __shared__ double T[(kPartRowsMax + 1) * (BW + 1)];
  if ((int)blockIdx.x < chol_blocks) {                    // chol_blocks is a multiple of 8: the copy tiles keep their XCD mapping
    if ((int)blockIdx.x < pv.P && threadIdx.x < 64) part_cholesky_body<BW>(pv, Lb, fail, T, (int)blockIdx.x, (int)threadIdx.x);
    return;
  }
  const int b = (int)blockIdx.x - chol_blocks;
  const int run = xcd_run_for(tiles);
  const int xcd = b & 7, qq = b >> 3;
  const long long idx = (long long)(((qq / run) * 8 + xcd) * run + qq % run) * blockDim.x + threadIdx.x;     // xcd_tile for the shifted index
  if (idx >= (long long)ne.N3 * ncols) return;
  const int r = (int)(idx / ncols), cidx = (int)(idx % ncols);
  if (seprow != nullptr && seprow[r]) Z[idx] = 0.0;            // (one rank: the separators' rows take no part in the product's main range)
  else if (cidx < ne.CB) {
    const int c = cidx / ne.B, k = cidx % ne.B;
    Z[idx] = ne.Et[((long long)c * ne.N3 + r) * ne.B + k];        // (the product's first operand is read where the assembly left it)
  } else {
    Z[idx] = ne.gs[r];
  }
}

// Gp[slab][CB][ncols] = Et^T Z over one K-slab: the one dense contraction of the solve (2 * CB^2 * 3N flops), on the
// fp64 matrix cores (v_mfma_f64_16x16x4_f64).  Both operands are K-major (the cross block camera-major, [C][3N][B]: row stride B and a per-lane column offset; Z [3N][ncols]), which is
// exactly the MFMA operand layout (lane l: A[l&15][k = l>>4], B[k = l>>4][l&15]; every 16-lane group reads 128
// contiguous bytes), so fragments go from global memory straight to registers -- no LDS staging.  One wavefront owns a
// 48x48 output tile (3x3 MFMA tiles, 36 accumulator registers) over wave_k rows of K; the four wavefronts of a
// workgroup take consecutive K ranges of the same tile and are summed through LDS.  Et^T C^-1 Et is symmetric: only
// tiles on or below the block diagonal are computed (k_schur_finish mirrors), plus one 48x16 tile per block row for
// the right-hand-side column.  Partial sums per slab are written, not atomically added: no memset, deterministic.
using d4 = __attribute__((ext_vector_type(4))) double;
constexpr int kGemmT = 48;
constexpr int kGemmUn = 4;                                 // k-steps (of 4 rows) per operand set
constexpr int kGemmSetRows = 4 * kGemmUn;

// Operand sets of the Schur product: kGemmUn k-steps of a 48-wide A strip and a 16 NJ-wide B strip, one double per lane and
// 16 x 4 fragment.  Two sets are alive: the loads of set k+1 are issued before the matrix-core instructions of set k
// (a lone wavefront per SIMD otherwise waits out the full memory latency once per set: 58 us instead of 27 at configs[2]).
template <int NJ>
struct GemmSet {
  double a[kGemmUn][3], b[kGemmUn][NJ];
  // ap / bp point at this lane's row (k + lane / 16) and first column; every address is inside the arrays (columns beyond the
  // matrix are clamped by the caller: they only feed output rows / columns that are never stored), so the loads carry no
  // predicate and no branch -- the steady-state loop below is one basic block and s_waitcnt can count the younger set's loads.
  __device__ __forceinline__ void load(const double* __restrict__ ap, const double* __restrict__ bp, long long lda, long long ldb,
                                       const int (&ao)[3], const int (&bo)[NJ]) {
#pragma unroll
    for (int u = 0; u < kGemmUn; ++u) {
#pragma unroll
      for (int i = 0; i < 3; ++i) a[u][i] = ap[(long long)(4 * u) * lda + ao[i]];
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[u][j] = bp[(long long)(4 * u) * ldb + bo[j]];
    }
  }
  // the last, possibly partial set of the whole product: rows clamped to the last one, their values replaced by zero
  __device__ __forceinline__ void load_tail(const double* __restrict__ A, const double* __restrict__ B, long long lda, long long ldb,
                                            const int (&ao)[3], const int (&bo)[NJ], int k, int k1, int lk) {
#pragma unroll
    for (int u = 0; u < kGemmUn; ++u) {
      const int row = k + 4 * u + lk;
      const bool kv = row < k1;
      const long long rc = kv ? row : k1 - 1;
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double v = A[rc * lda + ao[i]]; a[u][i] = kv ? v : 0.0; }
#pragma unroll
      for (int j = 0; j < NJ; ++j) { const double v = B[rc * ldb + bo[j]]; b[u][j] = kv ? v : 0.0; }
    }
  }
  __device__ __forceinline__ void multiply(d4 (&acc)[3][NJ]) const {
#pragma unroll
    for (int u = 0; u < kGemmUn; ++u)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
  }
};

// sets [s_lo, s_hi) of 16 rows of ONE operand pair (A row stride lda with per-lane offsets ao, B row stride ldb with bo), rows
// [row_lo, row_hi): only the pair's very last set can be partial
template <int NJ>
__device__ __forceinline__ void schur_gemm_sets(d4 (&acc)[3][NJ], const double* __restrict__ A, const double* __restrict__ Bm, long long lda, long long ldb,
                                                const int (&ao)[3], const int (&bo)[NJ], int row_lo, int row_hi, int s_lo, int s_hi, int lk) {
  if (s_hi <= s_lo) return;
  const int nsets = (row_hi - row_lo + kGemmSetRows - 1) / kGemmSetRows;
  const bool tail = s_hi == nsets && (row_hi - row_lo) % kGemmSetRows != 0;
  const int nfull = s_hi - s_lo - (tail ? 1 : 0);
  const int k0 = row_lo + s_lo * kGemmSetRows;
  const double* ap = A + (long long)(k0 + lk) * lda;
  const double* bp = Bm + (long long)(k0 + lk) * ldb;
  const long long sa = kGemmSetRows * lda, sb = kGemmSetRows * ldb;
  GemmSet<NJ> s0, s1;
  if (nfull > 0) {
    s0.load(ap, bp, lda, ldb, ao, bo);
    int si = 0;
    for (; si + 2 < nfull; si += 2) {                    // steady state: one basic block, every load unconditional
      s1.load(ap + sa, bp + sb, lda, ldb, ao, bo);
      s0.multiply(acc);
      ap += 2 * sa; bp += 2 * sb;
      s0.load(ap, bp, lda, ldb, ao, bo);
      s1.multiply(acc);
    }
    if (si + 1 < nfull) {
      s1.load(ap + sa, bp + sb, lda, ldb, ao, bo);
      s0.multiply(acc);
      s1.multiply(acc);
    } else {
      s0.multiply(acc);
    }
  }
  if (tail) {
    s0.load_tail(A, Bm, lda, ldb, ao, bo, row_lo + (s_hi - 1) * kGemmSetRows, row_hi, lk);
    s0.multiply(acc);
  }
}
// The product's K range is the rows [row_lo, row_hi) of (Ecm, Z) followed by the rows [0, sep_rows) of the compact pair (Dl, Xs) -- the
// separators' correction term (one rank, round 5; sep_rows = 0: none) --, cut into sets of 16; wavefront g of nslab * 4 takes sets
// [g q + min(g, r), ...) with q, r = nsets / nwaves, nsets % nwaves of the WHOLE range: every wavefront within one set of the others.
template <int NJ>   // NJ = 3: 48x48 tile of the symmetric part, NJ = 1: 48x16 tile holding the rhs column
__device__ __forceinline__ void schur_gemm_tile(const NEView& ne, int ncols, const double* __restrict__ Ecm, const double* __restrict__ Z,
                                                double* __restrict__ Gp, int a0, int b0, int row_lo, int row_hi, int slab, int nslab, double* red,
                                                const double* __restrict__ Dl, const double* __restrict__ Xs, int sep_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  d4 acc[3][NJ];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  int ao[3], ac[3], bo[NJ];
#pragma unroll
  for (int i = 0; i < 3; ++i) {                                // A = the cross block where the assembly left it (camera-major [C][3N][B]):
    const int col = min(a0 + 16 * i + lr, ne.CB - 1);          // column (c, k) of row r is Et[(c N3 + r) B + k] -- row stride B, per-lane offset
    ao[i] = (col / ne.B) * ne.N3 * ne.B + col % ne.B;
    ac[i] = col;                                               // (the compact pair is row-major [rows][CB])
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) bo[j] = NJ == 3 ? min(b0 + 16 * j + lr, ne.CB - 1) : ne.CB;
  const int nmain = (row_hi - row_lo + kGemmSetRows - 1) / kGemmSetRows, ncorr = (sep_rows + kGemmSetRows - 1) / kGemmSetRows;
  const int nsets = nmain + ncorr, nwaves = nslab * 4, g = slab * 4 + wave;
  const int q = nsets / nwaves, r = nsets % nwaves;
  const int s_lo = g * q + min(g, r), s_hi = s_lo + q + (g < r ? 1 : 0);
  schur_gemm_sets<NJ>(acc, Ecm, Z, ne.B, ncols, ao, bo, row_lo, row_hi, min(s_lo, nmain), min(s_hi, nmain), lk);
  if (ncorr > 0) schur_gemm_sets<NJ>(acc, Dl, Xs, ne.CB, ncols, ac, bo, 0, sep_rows, max(s_lo, nmain) - nmain, max(s_hi, nmain) - nmain, lk);
  // sum the four wavefronts through LDS, then store: element (i, j, reg) of lane l is row 16i + (l>>4) + 4 reg, col 16j + (l&15)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double* dst = red + ((i * NJ + j) * 4 + r) * 64 + lane;
            *dst = (w == 0 ? 0.0 : *dst) + acc[i][j][r];
          }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < 3 * NJ * 4 * 64; e += 256) {
    const int l = e & 63, r = (e >> 6) & 3, ij = e >> 8, i = ij / NJ, j = ij % NJ;
    const int row = a0 + 16 * i + (l >> 4) + 4 * r, col = b0 + 16 * j + (l & 15);
    if (row < ne.CB && col < ncols && (NJ == 3 ? col < ne.CB : (l & 15) == 0)) Gp[(long long)row * ncols + col] = red[e];
  }
}

// Two workgroups per CU (<= 256 registers).  Every tile of a slab reads rows of the same K range, 12 tiles share each 48-column
// strip: all workgroups of a slab go to ONE XCD (workgroup L runs on XCD L % 8: slab = L % 8 + 8 * (L / 8 / tiles)), so a strip
// comes from HBM once and from that XCD's L2 afterwards (with the slabs dealt over all XCDs every XCD fetched every strip:
// 154 - 308 MB for 55 MB of operands, and the kernel ran at the fabric's rate, not the matrix cores').  Grid: 8 * tiles *
// ceil(nslab / 8) workgroups; the slab count is chosen by the host (HipSchur::plan_gemm) to fill whole rounds of the XCD's slots.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_schur_gemm(NEView ne, int ncols, int row_lo, int row_hi, int nslab, const double* __restrict__ Ecm, const double* __restrict__ Z, double* __restrict__ Gp,
                  const double* __restrict__ Dl = nullptr, const double* __restrict__ Xs = nullptr, int sep_rows = 0) {
  __shared__ double red[9 * 4 * 64];
  const int nbk = (ne.CB + kGemmT - 1) / kGemmT, nsym = nbk * (nbk + 1) / 2, tiles = nsym + nbk;
  const int L = blockIdx.x, j = L >> 3;
  const int slab = (L & 7) + 8 * (j / tiles), t = j % tiles;
  if (slab >= nslab) return;
  double* G = Gp + (long long)slab * ne.CB * ncols;
  // (Dl, Xs, sep_rows: one rank, round 5 -- the separators' share of E^T C^-1 E that is not in E_S^T X_S, (R_S - E_S)^T X_S over the compact
  // separator arrays, as further rows of K; with it the product may read the interiors' UNCORRECTED solutions: HipSchur, where ncorr is set)
  if (t < nsym) {
    int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    schur_gemm_tile<3>(ne, ncols, Ecm, Z, G, bi * kGemmT, bj * kGemmT, row_lo, row_hi, slab, nslab, red, Dl, Xs, sep_rows);
  } else {
    schur_gemm_tile<1>(ne, ncols, Ecm, Z, G, (t - nsym) * kGemmT, ne.CB, row_lo, row_hi, slab, nslab, red, Dl, Xs, sep_rows);
  }
}

// S = (A + lambda D_c) - sum_slabs Gp[:, :CB] (dense CB x CB, both triangles from the computed lower block triangle),
// rhs = gc - sum_slabs Gp[:, CB], in 32x32 tiles; the workgroup of tile (0,0) goes on to factorise the first pivot block of
// the Gauss-Jordan solve below (k_gj_pivot), saving a launch.
constexpr int kNB = 32;
__device__ __forceinline__ void pivot_inverse_wave(double (*Dm)[kNB + 1], double* __restrict__ W, double* __restrict__ Pout, int* __restrict__ fail);
constexpr int kPivScratch = 5 * 16 * 17;      // doubles of LDS scratch pivot_inverse_wave needs
constexpr int kFinThreads = 256;            // 32x32 tile, four rows per thread (256 threads: the pivot inverse needs > 128 registers)
__global__ __launch_bounds__(kFinThreads) void k_schur_finish(NEView ne, int ncols, int nslab, double lambda, const double* __restrict__ Gp, double* __restrict__ S,
                                                             double* __restrict__ Linv, int* __restrict__ fail) {
  __shared__ double Dm[kNB][kNB + 1];
  __shared__ double Wp[kPivScratch];
  constexpr int kRowsPer = kNB * kNB / kFinThreads, kRowStep = kFinThreads / kNB;
  const int r0 = threadIdx.x / kNB, c = threadIdx.x % kNB;
  const int b = blockIdx.x * kNB + c;
  const long long stride = (long long)ne.CB * ncols;
  const bool first = blockIdx.x == 0 && blockIdx.y == 0;
  const int nb = min(kNB, ne.CB);
  // the slabs' partial sums: every load of a thread is independent of the others -- eight slabs x four rows in flight, added in
  // slab order (a loop of load, add, load ... is a chain of nslab memory round trips: 19 us of this kernel at 16 slabs)
  double gsum[kRowsPer];
  long long goff[kRowsPer];
#pragma unroll
  for (int q = 0; q < kRowsPer; ++q) {
    const int a = blockIdx.y * kNB + r0 + kRowStep * q;
    const int hi = a / kGemmT >= b / kGemmT ? a : b, lo = a / kGemmT >= b / kGemmT ? b : a;
    goff[q] = (a < ne.CB && b < ne.CB) ? (long long)hi * ncols + lo : -1;
    gsum[q] = 0.0;
  }
  for (int sl = 0; sl < nslab; sl += 8) {
    double t[8][kRowsPer];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < kRowsPer; ++q) t[u][q] = (sl + u < nslab && goff[q] >= 0) ? Gp[(sl + u) * stride + goff[q]] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < kRowsPer; ++q) gsum[q] += t[u][q];
  }
#pragma unroll
  for (int q = 0; q < kRowsPer; ++q) {
    const int r = r0 + kRowStep * q, a = blockIdx.y * kNB + r;
    double v = 0.0;
    if (a < ne.CB && b < ne.CB) {
      v = -gsum[q];
      if (a / ne.B == b / ne.B) {
        const int cam = a / ne.B;
        double h = ne.A[((long long)cam * ne.B + a % ne.B) * ne.B + b % ne.B];
        if (a == b) h += lambda * (h > 0.0 ? h : 1.0);
        v += h;
      }
      S[(long long)a * ne.CB + b] = v;
    }
    if (first) Dm[r][c] = (r < nb && c < nb && c <= r) ? v : ((r < nb && c < nb) ? 0.0 : (r == c ? 1.0 : 0.0));
  }
  if (blockIdx.y == 0 && r0 == 0 && b < ne.CB) {               // rhs rides as row CB
    double gr = 0.0;
    for (int sl = 0; sl < nslab; sl += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = sl + u < nslab ? Gp[(sl + u) * stride + (long long)b * ncols + ne.CB] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) gr += t[u];
    }
    S[(long long)ne.CB * ne.CB + b] = ne.gc[b] - gr;
  }
  if (!first) return;
  __syncthreads();
  if (threadIdx.x < 64) pivot_inverse_wave(Dm, Wp, Linv, fail);
}

// Dense solve of the reduced camera system S x = b (nn <= 1152, SPD) by BLOCK GAUSS-JORDAN, 32-column panels, ONE
// launch per panel and no back substitution.  The right-hand side rides along as row nn of the (nn+1) x nn array
// (row-major, both triangles of S filled).  In column-operation form, with P_k the inverse of the current pivot block:
//     every row block I below the panel (the rhs row included), every column block J != k:
//         Q_I = M[I][k] P_k ;   M[I][J] -= Q_I M[k][J] ;   M[I][k] <- Q_I
// After the last panel the rhs row holds x.  Rows above the panel are never touched again, the trailing block stays the
// (symmetric positive definite) Schur complement.  Each step reads `src` and writes `dst` (ping-pong), so all tiles of a
// step are independent.  Against a blocked Cholesky this trades ~3x the (tiny, perfectly parallel) tile flops for the
// removal of the sequential back substitution (44 us at nn = 288: nine dependent round trips to data other XCDs wrote).
//
// The critical path per panel is the inverse of the 32x32 pivot block, done by ONE wavefront of the workgroup that
// produced that block (pivot_inverse_wave): 2x2 blocks of 16 -- Cholesky + inverse factor of a 16x16 block in registers
// (row of the block and of the identity per lane, multipliers by DPP row broadcast, 1/d by v_rcp_f64 + 2 Newton steps), the Schur complement, the off-diagonal block of L^-1 and P = L^-T L^-1 on the fp64 matrix cores.
// Round 2 history (cycle counters, MVUS_GJ_PROBE): a 32-step register Cholesky took 22k cycles and the tile update 13k
// (three 32-deep LDS dot products per thread: LDS-bandwidth bound); now 16-step halves + matrix-core tile products.


// L D L^T of a 16x16 SPD block, every row of 16 lanes on its own copy: lane i of the row holds row i of the block (a, lower
// triangle) and row i of the identity (x).  Column operations col_j -= col_k * (a_jk / d_k) on both: on return x[k] of lane c is
// X[k][c] = (L^-1)[k][c] (unit lower, zero above) and rd[k] = 1 / d_k in every lane, so that block^-1 = X^T diag(rd) X.  No square
// root.  The multiplier A[j][k] comes from lane j by DPP (v_mov_b64_dpp row_newbcast:j -- the one DPP control 64-bit operands have
// on this part): a vector move per multiplier instead of two v_readlane through an SGPR pair with its wait states, which is what
// the 16-column chain was made of (tools/micro/ldl16_bench.hip: cycles of the two forms on one wavefront).  One basic block.
template <int K, int J> struct LdlColOps {
  static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16], double ta, double tx) {
    const double m = row_bcast<J>(a[K]);
    a[J] -= ta * m;
    x[J] -= tx * m;
    LdlColOps<K, J + 1>::run(a, x, ta, tx);
  }
};
template <int K> struct LdlColOps<K, 16> { static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double, double) {} };
template <int K> struct LdlCols {
  static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16], double (&rd)[16], bool& bad) {
    double d = row_bcast<K>(a[K]);
    bad |= !(d > 0.0);
    d = d > 0.0 ? d : 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = r * (2.0 - d * r);
    r = r * (2.0 - d * r);
    rd[K] = r;
    LdlColOps<K, K + 1>::run(a, x, a[K] * r, x[K] * r);
    LdlCols<K + 1>::run(a, x, rd, bad);
  }
};
template <> struct LdlCols<16> { static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double (&)[16], bool&) {} };
__device__ __forceinline__ void ldl_inv16(double (&a)[16], double (&x)[16], double (&rd)[16], int lane, int* __restrict__ fail) {
  bool bad = false;
  LdlCols<0>::run(a, x, rd, bad);
  if (bad && lane == 0) fail[0] = 2;
}

}  // namespace mvus
#include "ba_rcs.hip.h"
namespace mvus {

// acc += op(A) op(B) for 16x16 blocks in LDS (stride lda / ldb), K = 16: ta: A is read transposed, tb: B is.
__device__ __forceinline__ bcr_d4 mma16(const double* __restrict__ A, int lda, bool ta, const double* __restrict__ B, int ldb, bool tb, bcr_d4 acc) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    const int k = 4 * s_ + lk;
    const double av = ta ? A[k * lda + lr] : A[lr * lda + k];
    const double bv = tb ? B[lr * ldb + k] : B[k * ldb + lr];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ void store16(double* __restrict__ M, int ld, bcr_d4 acc, double sign) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) M[(lk + 4 * r) * ld + lr] = sign * acc[r];
}

// first wavefront of the workgroup: inverse P of the SPD block in Dm (lower part valid, identity padded beyond the
// block) -> Pout[r*kNB+c], both triangles.  2x2 blocks of 16:  A11^-1 = X1^T D1^-1 X1 (ldl_inv16),  T = A21 A11^-1,
// S = A22 - T A21^T,  P22 = S^-1 (ldl_inv16 again),  P21 = -P22 T,  P11 = A11^-1 - T^T P21.  W: kPivScratch doubles of LDS.
__device__ __forceinline__ void pivot_inverse_wave(double (*Dm)[kNB + 1], double* __restrict__ W, double* __restrict__ Pout, int* __restrict__ fail) {
  constexpr int LD = 17, LDD = kNB + 1;
  const int lane = threadIdx.x & 63, row = lane & 15, lr = lane & 15, lk = lane >> 4;
  const bcr_d4 zero{0.0, 0.0, 0.0, 0.0};
  double* M1 = W;                 // X (unit lower inverse factor of the block being inverted)
  double* M2 = W + 16 * LD;       // D^-1 X
  double* M3 = W + 2 * 16 * LD;   // A11^-1
  double* M4 = W + 3 * 16 * LD;   // T, later P21
  double* M5 = W + 4 * 16 * LD;   // S, later P22
  double a[16], x[16], rd[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {          // unconditional LDS reads, then selects: no branch per element (every row of 16 lanes the same)
    const double t = Dm[row][k];
    a[k] = k <= row ? t : 0.0;
    x[k] = k == row ? 1.0 : 0.0;
  }
  rcs_ldl16(a, x, rd, lane, fail);
  if (lane < 32) {                        // lanes 0..15 store X, lanes 16..31 (the same values) D^-1 X
    double* dst = (lane < 16 ? M1 : M2) + row;
#pragma unroll
    for (int k = 0; k < 16; ++k) dst[k * LD] = lane < 16 ? x[k] : rd[k] * x[k];
  }
  lds_wave_sync();
  bcr_d4 a11 = mma16(M1, LD, true, M2, LD, false, zero);                 // A11^-1 = X^T (D^-1 X)
  store16(M3, LD, a11, 1.0);
  lds_wave_sync();
  bcr_d4 acc = mma16(&Dm[16][0], LDD, false, M3, LD, false, zero);       // T = A21 A11^-1
  store16(M4, LD, acc, 1.0);
  lds_wave_sync();
  acc = mma16(M4, LD, false, &Dm[16][0], LDD, true, zero);               // T A21^T
#pragma unroll
  for (int r = 0; r < 4; ++r) M5[(lk + 4 * r) * LD + lr] = Dm[16 + lk + 4 * r][16 + lr] - acc[r];
  lds_wave_sync();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double t = M5[row * LD + k];
    a[k] = k <= row ? t : 0.0;
    x[k] = k == row ? 1.0 : 0.0;
  }
  rcs_ldl16(a, x, rd, lane, fail);
  if (lane < 32) {
    double* dst = (lane < 16 ? M1 : M2) + row;
#pragma unroll
    for (int k = 0; k < 16; ++k) dst[k * LD] = lane < 16 ? x[k] : rd[k] * x[k];
  }
  lds_wave_sync();
  const bcr_d4 p22 = mma16(M1, LD, true, M2, LD, false, zero);           // S^-1
  store16(M5, LD, p22, 1.0);
  lds_wave_sync();
  const bcr_d4 p21 = mma16(M5, LD, false, M4, LD, false, zero);          // P22 T  (sign applied on use)
  lds_wave_sync();                                                       // every lane has read T before it is overwritten
  store16(M4, LD, p21, -1.0);
  lds_wave_sync();
  const bcr_d4 tp = mma16(&Dm[16][0], LDD, true, M4, LD, false, zero);   // A21^T P21;  T^T = A11^-1 A21^T  =>  T^T P21 = A11^-1 (A21^T P21)
  store16(M1, LD, tp, 1.0);
  lds_wave_sync();
  const bcr_d4 corr = mma16(M3, LD, false, M1, LD, false, zero);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rw = lk + 4 * r;
    Pout[rw * kNB + lr] = a11[r] - corr[r];
    Pout[(16 + rw) * kNB + lr] = -p21[r];
    Pout[lr * kNB + 16 + rw] = -p21[r];
    Pout[(16 + rw) * kNB + 16 + lr] = p22[r];
  }
}

// one panel step; grid (row tiles below the panel incl. the rhs row, all column tiles); pc != nullptr on the last step.
// Four wavefronts, one 16x16 quarter of the tile each; both tile products on the fp64 matrix cores.
constexpr int kGjThreads = 256;
__global__ __launch_bounds__(kGjThreads) void k_gj_step(int nn, int kb, const double* __restrict__ src, double* __restrict__ dst,
                                                        double* __restrict__ Pinv, int* __restrict__ fail, double* __restrict__ pc) {
  constexpr int kTile = kNB * (kNB + 1);
  __shared__ double lds[4 * kTile];
  double (*Ai)[kNB + 1] = reinterpret_cast<double (*)[kNB + 1]>(lds);
  double (*Pk)[kNB + 1] = reinterpret_cast<double (*)[kNB + 1]>(lds + kTile);
  double (*Q)[kNB + 1] = reinterpret_cast<double (*)[kNB + 1]>(lds + 2 * kTile);
  double (*Bk)[kNB + 1] = reinterpret_cast<double (*)[kNB + 1]>(lds + 3 * kTile);      // Q and Bk: one contiguous scratch area later
  const int nb = min(kNB, nn - kb), base = kb + nb;
  const int i0 = base + blockIdx.x * kNB, j0 = blockIdx.y * kNB;
  const bool pivcol = j0 == kb;                              // this tile is column block k: it stores Q_I
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const int br = (wave >> 1) * 16, bc = (wave & 1) * 16;     // this wavefront's quarter
#pragma unroll
  for (int q = 0; q < kNB * kNB / kGjThreads; ++q) {
    const int e = threadIdx.x + kGjThreads * q, r = e / kNB, c = e % kNB;
    // (indices clamped into the (nn + 1) x nn array, values outside the tile replaced afterwards: sixteen unconditional loads in
    // flight together instead of a branch around each -- the tile update is on the chain of every panel)
    const double ai = src[(long long)min(i0 + r, nn) * nn + min(kb + c, nn - 1)];
    const double bk = src[(long long)min(kb + r, nn) * nn + min(j0 + c, nn - 1)];
    Ai[r][c] = (i0 + r <= nn && c < nb) ? ai : 0.0;
    Pk[r][c] = Pinv[(long long)(kb / kNB) * kNB * kNB + e];
    Bk[r][c] = (r < nb && j0 + c < nn) ? bk : 0.0;
  }
  double v[4];                                               // the tile itself, in the accumulator layout
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + br + lk + 4 * r, j = j0 + bc + lr;
    const double t = src[(long long)min(i, nn) * nn + min(j, nn - 1)];
    v[r] = (i <= nn && j < nn && !pivcol) ? t : 0.0;
  }
  __syncthreads();
  bcr_d4 acc{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s_ = 0; s_ < kNB / 4; ++s_) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[br + lr][4 * s_ + lk], Pk[4 * s_ + lk][bc + lr], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) Q[br + lk + 4 * r][bc + lr] = acc[r];
  __syncthreads();
  if (pivcol) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r];
  } else {
    bcr_d4 upd{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s_ = 0; s_ < kNB / 4; ++s_) upd = __builtin_amdgcn_mfma_f64_16x16x4f64(Q[br + lr][4 * s_ + lk], Bk[4 * s_ + lk][bc + lr], upd, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] -= upd[r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + br + lk + 4 * r, j = j0 + bc + lr;
    if (i <= nn && j < nn && (!pivcol || bc + lr < nb)) {
      dst[(long long)i * nn + j] = v[r];
      if (pc && i == nn) pc[j] = -v[r];                      // last panel: the rhs row is the solution
    }
  }
  // the tile holding the next pivot block inverts it
  const int nb2 = min(kNB, nn - base);
  if (blockIdx.x != 0 || j0 != base || nb2 <= 0) return;
  __syncthreads();                                           // Ai is reused as the block to invert, Q and Bk as scratch
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = br + lk + 4 * r, cc = bc + lr;
    Ai[rr][cc] = (rr < nb2 && cc < nb2 && cc <= rr) ? v[r] : ((rr < nb2 && cc < nb2) ? 0.0 : (rr == cc ? 1.0 : 0.0));
  }
  __syncthreads();
  static_assert(kPivScratch <= 2 * kNB * (kNB + 1), "pivot scratch must fit Q and Bk");
  if (threadIdx.x < 64) pivot_inverse_wave(Ai, &Q[0][0], Pinv + (long long)(base / kNB) * kNB * kNB, fail);
}

// p (x order) from p_c and p_s = -(z_g + Z_E p_c)
__global__ __launch_bounds__(kThreads) void k_back_substitute(DevProblem dp, NEView ne, int ncols, int row_lo, int row_hi, int cams,
                                                              const double* __restrict__ Z, const double* __restrict__ pc, double* __restrict__ px,
                                                              int* __restrict__ fail = nullptr, int* __restrict__ fail_mirror = nullptr,
                                                              double* __restrict__ fail_tail = nullptr) {
  // one wavefront per owned spline row: lanes stride over the row of Z (coalesced), then a shuffle reduction
  const int lane = threadIdx.x & 63;
  // last kernel of a solve: the two failure flags go to the host's mapped copy here (a device-to-host copy of 8 bytes is a launch)
  // (a hand-over time-out inside the solve -- fail[2], raised by a workgroup that gave up waiting for another one of its launch: k_rcs_factor's
  // rows, k_sep_bcr_levels -- outranks whatever a later kernel made of the unfinished data: it is reported as its own code, HipSchur::retry_same)
  if (fail != nullptr && blockIdx.x == 0 && threadIdx.x < 2) {
    int v = fail[threadIdx.x];
    if (threadIdx.x == 0 && fail[2] != 0) { v = kFailHandoverCode; fail[0] = v; }
    if (fail_mirror != nullptr) fail_mirror[threadIdx.x] = v;
    // time shards: the flags travel with the step through its sum over the ranks (fail_tail = px + n; decoded by fail_sum_code / fail_sum_flag)
    if (fail_tail != nullptr) fail_tail[threadIdx.x] = threadIdx.x == 0 ? (v == kFailHandoverCode ? kFailHandoverSum : (double)v)
                                                                        : (double)(v & 1) + ((v & 2) ? kFailSpanSum : 0.0);
  }
  const int r = row_lo + blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && cams)
    for (int idx = threadIdx.x; idx < ne.CB; idx += kThreads) px[cam_col(dp.C, dp.P, idx / ne.B, idx % ne.B)] = pc[idx];
  if (r >= row_hi) return;
  const double* zr = Z + (long long)r * ncols;
  double acc = 0.0;
  for (int k = lane; k < ne.CB; k += 64) acc += zr[k] * pc[k];
  acc = wave_sum(acc);
  if (lane == 0) {
    const int g = r / 3 + ne.row0, d = r % 3;
    px[dp.mv.ctrl_x0[g] + d * dp.mv.ctrl_stride[g]] = -(acc + zr[ne.CB]);
  }
}

// One rank, round 5: the step's spline rows when the interiors' columns of Z were NOT corrected for the separators (k_part_back: a
// read-modify-write of all of Z) and the separators' rows of Z hold zeros.  k_back_substitute has written -(Z_i p_c + z_g,i) for the
// interiors' rows (and zeros for the separators'); one workgroup per interior finishes them:
//   s = X_S p_c + x_S,g  for the separators either side (rows of pv.R: 2 s3 dot products),
//   p_i += sum_a VW[i][a] s[a]  for its rows,   p_S = -s  for the separator after it.
template <int S3>
__global__ __launch_bounds__(256) void k_back_correct(DevProblem dp, NEView ne, PartView pv, int ncols, const double* __restrict__ pc, double* __restrict__ px) {
  const int p = blockIdx.x, i = threadIdx.x;
  constexpr int st = 2 * S3;
  const int r0 = pv.i0[p], nr = pv.i1[p] - r0;
  const int sl = pv.sl[p], sr = pv.sr[p];
  constexpr int nch = 256 / st;                               // threads per separator value: thread (a, ch) takes columns ch, ch + nch, ...
  __shared__ double part[st][nch];
  __shared__ double sx[st];
  // (everything the row needs at the end -- its coupling entries, its place in x, the value k_back_substitute left there -- is requested
  // before the dot products: the kernel is a chain of memory round trips)
  double vwv[st];
  const int ic = min(i, max(nr - 1, 0));
  {
    const double* __restrict__ vw = pv.VW + ((long long)p * kPartRowsMax + ic) * st;
#pragma unroll
    for (int a = 0; a < st; ++a) vwv[a] = vw[a];
  }
  const int rrow = r0 + ic, grow = rrow / 3 + ne.row0;
  const int xrow = dp.mv.ctrl_x0[grow] + (rrow % 3) * dp.mv.ctrl_stride[grow];
  const double pold = px[xrow];
  int xsep = 0;
  if (sr >= 0 && i < S3) { const int r = sr + i, g = r / 3 + ne.row0; xsep = dp.mv.ctrl_x0[g] + (r % 3) * dp.mv.ctrl_stride[g]; }
  {
    const int a = i / nch, ch = i - a * nch;
    if (a < st) {
      const bool on = (a < S3 ? sl : sr) >= 0;
      const long long gq = (long long)pv.q_off + p - (a < S3 ? 1 : 0);
      const double* __restrict__ xr = pv.R + ((on ? gq : 0) * S3 + (a % S3)) * ncols;
      double acc = 0.0;
      constexpr int kU = 12;
      for (int k0 = ch; k0 < ne.CB; k0 += nch * kU) {          // twelve products in flight per thread
        double xv[kU], pk[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const int k = min(k0 + nch * u, ne.CB - 1); xv[u] = xr[k]; pk[u] = pc[k]; }
#pragma unroll
        for (int u = 0; u < kU; ++u) acc += k0 + nch * u < ne.CB ? xv[u] * pk[u] : 0.0;
      }
      part[a][ch] = on ? acc : 0.0;
    }
    __syncthreads();
    if (i < st) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < nch; ++w) t += part[i][w];
      const bool o = (i < S3 ? sl : sr) >= 0;
      const long long gq = (long long)pv.q_off + p - (i < S3 ? 1 : 0);
      sx[i] = o ? t + pv.R[(gq * S3 + (i % S3)) * ncols + ne.CB] : 0.0;
    }
    __syncthreads();
  }
  if (sr >= 0 && i < S3) px[xsep] = -sx[S3 + i];
  if (i >= nr) return;
  double acc = 0.0;
  if (sl >= 0) {
#pragma unroll
    for (int a = 0; a < S3; ++a) acc += vwv[a] * sx[a];
  }
  if (sr >= 0) {
#pragma unroll
    for (int a = 0; a < S3; ++a) acc += vwv[S3 + a] * sx[S3 + a];
  }
  px[xrow] = pold + acc;
}
static_assert(kPartRowsMax <= 256, "k_back_correct: one thread per row of an interior");

template <class BE>
struct HipSchur {
  BE& be;
  NEView ne{};
  int ncols = 0, BW = 0;
  size_t ne_count = 0;
  double* NEset[2] = {nullptr, nullptr};   // two sets of normal-equation blocks: the solver reads NEset[ne_cur]; the other one takes the speculative linearisation (linearize_spec)
  int ne_cur = 0;
  size_t off_gc = 0, off_Cb = 0, off_gs = 0, off_Et = 0, off_Apart = 0;
  double *NE = nullptr, *Lb = nullptr, *Z = nullptr, *G = nullptr, *G0 = nullptr, *S = nullptr, *S2 = nullptr, *Linv = nullptr, *rhs = nullptr, *pc = nullptr,
         *DG = nullptr, *D = nullptr, *gx = nullptr, *px = nullptr, *sepbuf = nullptr;
  RcsView rcs{};            // reduced camera system in block-image form (ba_rcs.hip.h)
  unsigned* rcs_flags = nullptr;   // step counter of the in-launch hand-over (k_rcs_factor -> its row workgroups); zeroed by k_rcs_finish
  bool rcs_trsm_launch = false;
  bool bcr_fused = true;                     // the wide cyclic-reduction levels in one launch (k_sep_bcr_levels); MVUS_BCR_FUSED=0: a launch per level
  unsigned* bcr_done = nullptr;              // [m] per separator: the level mark of the hand-over
  unsigned bcr_epoch = 0;
  unsigned rcs_spin_limit = kRcsSpinLimit;   // MVUS_RCS_SPIN_LIMIT: test hook (0 = the first poll that finds the flag behind gives up)
  int handover_timeouts = 0;                 // solves repeated because a consumer workgroup of k_rcs_factor timed out (retry_same)
  int part_len = kPartL;    // control points per interior of the band solver (<= kPartL)
  bool use_rcs = true;      // MVUS_RCS=gj: the block Gauss-Jordan of rounds 1-4 (A/B)
  int* fail = nullptr;      // [0] numerical failure of a solve, [1] a row reached outside the slice (assembly)
  int* fail_host = nullptr;
  int* fail_map = nullptr;  // device address of fail_host (mapped pinned)
  PartView pv{};
  int* part_tables = nullptr;
  int nslab = 1;            // K-slabs of the Schur product (partial sums in G): HipSchur::plan_gemm
  int ncorr = 0;            // 1: the product carries the separators' correction term and the interiors are NOT back-corrected
  int n_own_sep = 0;        // separators of this slice (a time shard adds the correction rows of ITS separators: every separator once over the ranks)
  // time shards, round 6: two-level elimination of the separators (k_sep2_*): the local ones by this rank alone, the world - 1 cut separators summed
  bool two_level = false;
  int k_loc = 0, has_ghost = 0, has_cut = 0, ncut = 0;
  double *Rloc = nullptr, *CGK = nullptr, *cutbuf = nullptr, *cutws = nullptr;
  size_t cut_count = 0, bcr_lds_loc = 0;
  int bcr_cols_loc = kBcrCols;
  double* Dl = nullptr;
  int bcr_cols = kBcrCols;
  size_t bcr_lds = 0;       // dynamic LDS of k_sep_bcr_rhs; the sequential separator kernels remain for chains too long for it
  bool use_bcr = false;
  // slice of the spline system held by this handle (everything unless it is a time shard)
  bool shard = false;
  int Ntot = 0, own_lo = 0, own_hi = 0;        // owned control points, LOCAL indices (slice starts at ne.row0)
  size_t sep_count = 0, halo_count = 0, nAg = 0, n_apart = 0;
  int nbound = 0;
  bool diag_pending = false;                   // D / g in x order still to be written (folded into the next k_build_rhs)
  bool overlap_chol = true;                    // interiors factorised beside the right-hand-side copies (k_cholesky_and_rhs)
  int rhs_tiles_z = 0;
  int* halo_tables = nullptr;                  // [nbound] cut, [nbound] index in the packed buffer
  // window-major fused assembly (ba_assemble_win.hip.h): tables and the per-(window, camera) camera-block partials
  WinView wv{};
  bool use_win = false;
  bool wide = false;                           // band wider than six control points: the general band kernels instead of the partitioned solver
  void* win_tables = nullptr;
  size_t win_lds = 0;

  // K-slabs of the Schur product: a slab's tiles run on one XCD, two workgroups per CU, so the time is (rounds of the busiest XCD's
  // slots) x (row sets per wavefront); the slab count with the least of that (ties: fewer slabs = fewer partial sums for k_schur_finish)
  void plan_gemm(int rows) {
    const int nbk = (ne.CB + kGemmT - 1) / kGemmT, tiles = nbk * (nbk + 1) / 2 + nbk;
    int cus = 256;
    { int dev = 0; hipDeviceProp_t pr{}; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
    const int slots = 2 * cus, nsets = std::max(1, (rows + kGemmSetRows - 1) / kGemmSetRows);
    nslab = 1;
    const char* e = std::getenv("MVUS_GEMM_SLABS");
    if (e && std::atoi(e) > 0) { nslab = std::atoi(e); }
    else {
      double best = 1e300;
      for (int s = 1; s <= 128 && 4 * s <= nsets; ++s) {
        const int per_wave = (nsets + 4 * s - 1) / (4 * s);                       // sets of the busiest wavefront
        const int on_xcd = tiles * ((s + 7) / 8);                                 // workgroups of the busiest XCD
        const double cost = (double)((on_xcd + slots / 8 - 1) / (slots / 8)) * (per_wave + 1.5) + 0.002 * s;
        if (cost < best * 0.995) { best = cost; nslab = s; }
      }
    }
  }

  explicit HipSchur(BE& b) : be(b) {
    const HostProblem& hp = be.hp;
    const auto& ts = be.tshard;
    shard = ts.on && ts.world > 1;
    Ntot = hp.N;
    int glo = 0, ghi = hp.N, olo = 0, ohi = hp.N;          // slice and owned range, global control points
    if (shard) {
      olo = ts.cuts[ts.rank]; ohi = ts.cuts[ts.rank + 1];
      glo = std::max(0, olo - ts.halo); ghi = std::min(hp.N, ohi + ts.halo);
    }
    ne.C = hp.C; ne.B = 3 + hp.P; ne.CB = ne.C * ne.B; ne.N = ghi - glo; ne.N3 = 3 * ne.N; ne.row0 = glo;
    own_lo = olo - glo; own_hi = ohi - glo;
    int W = 4;
    for (int j = 1; j + 1 < hp.T; ++j) {
      if (hp.ms_part[j] < 0 || hp.ms_part[j - 1] != hp.ms_part[j]) continue;
      int lo = std::min(hp.ms_ctrl[j - 1], hp.ms_ctrl[j]), hi = std::max(hp.ms_ctrl[j - 1], hp.ms_ctrl[j]);
      if (hp.motion_type == MVUS_MOTION_F && hp.ms_part[j + 1] == hp.ms_part[j]) { lo = std::min(lo, hp.ms_ctrl[j + 1]); hi = std::max(hi, hp.ms_ctrl[j + 1]); }
      W = std::max(W, hi + 3 - lo + 1);
    }
    // W <= 6: the partitioned band solver (templates for W = 4 and 6).  Wider -- FITPACK knots less than a frame apart, the motion
    // rows then reach over more than three knot spans -- : the band as it is, factorised and solved by the general kernels
    // (k_band_chol_generic / k_band_solve_generic: one CU, for the small problems of the incremental loop where this happens)
    if (W > kWideW) throw HipError{"LM_SCHUR: motion rows couple control points " + std::to_string(W) + " apart (knots far below one frame) - unsupported band width (at most " + std::to_string(kWideW) + ")", MVUS_E_UNSUPPORTED};
    wide = W > 6;
    if (wide && shard) throw HipError{"LM_SCHUR: a band wider than six control points is not supported on a time shard", MVUS_E_UNSUPPORTED};
    if (!wide) W = W <= 4 ? 4 : 6;
    ne.W = W;
    BW = 3 * W - 1;
    ncols = ne.CB + 1;
    const int sctrl = W - 1;
    if (shard) {
      if (ts.halo < sctrl + 3) throw HipError{"time shard: halo must be at least band half-width + 3 control points"};
      for (int r = 0; r < ts.world; ++r)
        if (ts.cuts[r + 1] - ts.cuts[r] < 2 * ts.halo + sctrl + 1) throw HipError{"time shard: a rank owns fewer control points than 2 * halo + separator"};
    }
    // packed normal equations [A | gc | (halo exchange buffer) | Cb | gs | Et]: the head is what a time shard sums over the ranks
    nbound = shard ? (ts.rank > 0) + (ts.rank + 1 < ts.world) : 0;
    halo_count = shard ? (size_t)(ts.world - 1) * 2 * ts.halo * (3 * ne.CB + W * 9 + 3) : 0;
    const size_t nA = (size_t)ne.C * ne.B * ne.B, ngc = ne.CB, nCb = (size_t)ne.N * W * 9, ngs = ne.N3, nEt = (size_t)ne.N3 * ne.CB;
    nAg = nA + ngc;
    // time shards: diag(H) and g ride in the summed head too ([A | gc | halo | D | g])
    const size_t ndg = shard ? 2 * (size_t)hp.n : 0;
    ne_count = nA + ngc + halo_count + ndg + nCb + ngs + nEt;
    // + the per-workgroup camera-block partials of the assembly, behind the blocks (cleared with them, never summed over ranks)
    n_apart = (size_t)kGaParts * (kGaThreads / 64) * std::max<size_t>(hp.chunks.size(), 1) * (size_t)((ne.B + 1) * (ne.B + 2) / 2);
    NEset[0] = be.alloc(ne_count + n_apart);
    off_gc = nA; off_Cb = nA + ngc + halo_count + ndg; off_gs = off_Cb + nCb; off_Et = off_gs + ngs; off_Apart = off_Et + nEt;
    bind_ne(0);
    Lb = be.alloc((size_t)ne.N3 * (BW + 1));
    Z = be.alloc((size_t)ne.N3 * ncols);
    plan_gemm(3 * (own_hi - own_lo));
    G = be.alloc((size_t)nslab * ne.CB * ncols);
    G0 = be.alloc((size_t)ne.CB * ncols);
    S = be.alloc((size_t)(ne.CB + 1) * ne.CB);
    rhs = be.alloc(ne.CB); pc = be.alloc(ne.CB);
    S2 = be.alloc((size_t)(ne.CB + 1) * ne.CB);
    Linv = be.alloc((size_t)((ne.CB + kNB - 1) / kNB) * kNB * kNB);
    rcs.nn = ne.CB; rcs.nbk = (ne.CB + 15) / 16;
    rcs.Simg = be.alloc(rcs_doubles(ne.CB)); rcs.Tsc = be.alloc(rcs_doubles(ne.CB)); rcs.x = be.alloc((size_t)rcs.nbk * 16);
    // which solver of the reduced camera system: the blocked L D L^T of ba_rcs.hip.h (round 5) at every size -- one launch up to 144
    // unknowns (24 us against the block Gauss-Jordan's 32 at 63 unknowns), and with the rows below each super-block solved inside the
    // factor launch it is level with or ahead of the Gauss-Jordan beyond (configs[2]: 0.488 - 0.495 against 0.498 - 0.503 ms per
    // step, configs[3]: 1.297 against 1.299).  MVUS_RCS=gj keeps the Gauss-Jordan (A/B, tests).
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&rcs_flags), 4 * sizeof(unsigned)));
    MVUS_HIP(hipMemsetAsync(rcs_flags, 0, 4 * sizeof(unsigned), be.stream));
    { const char* e = std::getenv("MVUS_RCS_TRSM"); rcs_trsm_launch = e && std::strcmp(e, "launch") == 0; }
    { const char* e = std::getenv("MVUS_RCS_SPIN_LIMIT"); if (e) rcs_spin_limit = (unsigned)std::strtoul(e, nullptr, 10); }
    { const char* e = std::getenv("MVUS_RCS"); use_rcs = !(e && std::strcmp(e, "gj") == 0); }
    { const char* e = std::getenv("MVUS_BCR_FUSED"); if (e) bcr_fused = std::atoi(e) != 0; }
    MVUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rcs_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((rcs_stage_doubles(kRcsSP) + 512) * sizeof(double))));
    MVUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rcs_backsub), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(rcs_backsub_doubles(rcs.nbk) * sizeof(double))));
    if (shard) { DG = nullptr; D = NE + nAg + halo_count; gx = D + hp.n; }
    else { DG = be.alloc(2 * (size_t)hp.n); D = DG; gx = DG + hp.n; }
    px = be.alloc(hp.n + 2);                         // + the two failure flags of a time shard
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&fail), 4 * sizeof(int)));      // [2]: sticky hand-over time-out mark of the running solve
    MVUS_HIP(hipMemsetAsync(fail, 0, 4 * sizeof(int), be.stream));
    MVUS_HIP(hipHostMalloc(reinterpret_cast<void**>(&fail_host), 2 * sizeof(int), hipHostMallocMapped));
    fail_host[0] = fail_host[1] = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&fail_map), fail_host, 0) != hipSuccess) fail_map = nullptr;
    ne.err = fail + 1;
    // partition of the owned chain; the separators are numbered along the chain of ALL ranks (each rank can compute
    // every other rank's count from the cuts)
    const bool close = shard && ts.rank + 1 < ts.world;
    // Interior length: the factorisation of an interior is one dependent chain of its rows (0.3 us a row), the separator system one of
    // log2(separators) levels whose cost grows with the number of right-hand-side columns.  Few columns (<= 128: configs[1], [4]):
    // half-length interiors -- measured 0.333 -> 0.309 ms and 0.390 -> 0.362 ms a step; 289 columns: no difference; 577: 1.283 -> 1.363.
    // (Every rank of a sharded solve computes the same value: CB is global.)
    part_len = ne.CB <= 128 ? kPartL / 2 : kPartL;
    if (const char* e = std::getenv("MVUS_PART_LEN")) part_len = std::atoi(e);
    const ChainPart cp = partition_chain(own_lo, own_hi - own_lo, sctrl, close, part_len);
    pv.P = (int)cp.i0.size(); pv.s3 = 3 * sctrl;
    pv.q_off = 0; pv.m = (int)cp.sep.size();
    if (shard) {
      pv.m = 0;
      for (int r = 0; r < ts.world; ++r) {
        const ChainPart o = partition_chain(0, ts.cuts[r + 1] - ts.cuts[r], sctrl, r + 1 < ts.world, part_len);
        if (r == ts.rank) pv.q_off = pv.m;
        pv.m += (int)o.sep.size();
      }
    }
    const bool ghost = shard && ts.rank > 0;          // the separator that closes the previous rank's chain: left of interior 0
    std::vector<int> sl(pv.P, -1), sr(pv.P, -1), tc0, tpl, tpr, tgq, town;
    for (int k = 0; k < pv.P; ++k) {
      if (k < (int)cp.sep.size()) sr[k] = cp.sep[k];
      if (k > 0) sl[k] = cp.sep[k - 1];
    }
    if (ghost) {
      sl[0] = 3 * (own_lo - sctrl);
      tc0.push_back(sl[0]); tpl.push_back(-1); tpr.push_back(0); tgq.push_back(pv.q_off - 1); town.push_back(0);
    }
    for (int k = 0; k < (int)cp.sep.size(); ++k) {
      tc0.push_back(cp.sep[k]); tpl.push_back(k); tpr.push_back(k + 1 < pv.P ? k + 1 : -1); tgq.push_back(pv.q_off + k); town.push_back(1);
    }
    pv.nt = (int)tc0.size();
    std::vector<int> tab;
    for (const std::vector<int>* v : std::initializer_list<const std::vector<int>*>{&cp.i0, &cp.i1, &sl, &sr, &tc0, &tpl, &tpr, &tgq, &town}) tab.insert(tab.end(), v->begin(), v->end());
    const size_t seprow_at = tab.size();
    {
      std::vector<unsigned char> srow((size_t)(ne.N3 + 3) / 4 * 4, 0);      // 1 = the row belongs to a separator (bytes, packed into the int table)
      for (int k = 0; k < (int)cp.sep.size(); ++k) for (int a = 0; a < pv.s3; ++a) { const int r = cp.sep[k] + a; if (r >= 0 && r < ne.N3) srow[(size_t)r] = 1; }     // (rows as the tasks' tc0: local to the slice)
      tab.resize(tab.size() + srow.size() / 4);
      std::memcpy(tab.data() + seprow_at, srow.data(), srow.size());
    }
    tab.push_back(0);
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&part_tables), tab.size() * sizeof(int)));
    MVUS_HIP(hipMemcpyAsync(part_tables, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, be.stream));
    pv.i0 = part_tables; pv.i1 = pv.i0 + pv.P; pv.sl = pv.i1 + pv.P; pv.sr = pv.sl + pv.P;
    pv.tc0 = pv.sr + pv.P; pv.tpl = pv.tc0 + pv.nt; pv.tpr = pv.tpl + pv.nt; pv.tgq = pv.tpr + pv.nt; pv.town = pv.tgq + pv.nt;
    pv.seprow = reinterpret_cast<const unsigned char*>(part_tables + seprow_at); pv.CB = ne.CB; pv.B = ne.B; pv.N3 = ne.N3; pv.Dl = nullptr; pv.direct = 0;
    pv.Et = ne.Et; pv.gs = ne.gs;
    pv_ready = true;
    if (nbound > 0) {
      std::vector<int> hb;
      if (ts.rank > 0) hb.push_back(ts.cuts[ts.rank]);
      if (ts.rank + 1 < ts.world) hb.push_back(ts.cuts[ts.rank + 1]);
      if (ts.rank > 0) hb.push_back(ts.rank - 1);
      if (ts.rank + 1 < ts.world) hb.push_back(ts.rank);
      MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&halo_tables), hb.size() * sizeof(int)));
      MVUS_HIP(hipMemcpyAsync(halo_tables, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice, be.stream));
    }
    MVUS_HIP(hipStreamSynchronize(be.stream));
    pv.VW = be.alloc((size_t)pv.P * kPartRowsMax * 2 * pv.s3);
    // One rank: the interiors' columns of Z are NOT corrected for the separators after the separator solve (k_part_back: a
    // read-modify-write of all of Z, 20 us at configs[2], 41 at configs[3]).  Block elimination gives
    //   E^T C^-1 E = E_I^T (B^-1 E_I) + R_S^T X_S,   R_S = E_S - H_SI B^-1 E_I  (the separators' reduced right-hand sides),
    // so the Schur product may pair the cross block with the UNCORRECTED interior solutions if the separators contribute R_S^T X_S: their
    // rows of Z are written as ZEROS by the right-hand-side copy (nothing from the main K range), R_S is kept beside the in-place solve
    // (pv.Dl) and R_S^T X_S runs as further rows of the product's K range over the compact arrays pv.Dl, pv.R (dealt to the same
    // wavefronts: schur_gemm_tile).  The step's own back-substitution is corrected for ONE vector (k_back_correct).  Time shards keep the back-correction: their separator sums run over the ranks.
    ncorr = 0;
    n_own_sep = (int)cp.sep.size();
    if (!wide && pv.m > 0 && std::getenv("MVUS_PART_BACK") == nullptr) {      // (round 6: time shards too -- R_S is the SUMMED reduced right-hand side there, copied beside the in-place solve after the ranks' sum)
      ncorr = 1;
      Dl = be.alloc((size_t)pv.m * pv.s3 * ne.CB);
      pv.Dl = Dl;
      // ... and the right-hand-side copy (Z = row-major E: a pass over both, the longer half of k_cholesky_and_rhs at configs[3]) is not
      // made at all: the interior solves' forward pass reads the assembled blocks (part_solve_block).  Z's separator rows are then
      // written by nobody: zeroed once, here.  (configs[3] 1.245 -> 1.218 ms, configs[2] 0.468 -> 0.460, configs[1] level.)
      pv.direct = 1;
      if (const char* e = std::getenv("MVUS_DIRECT_RHS")) pv.direct = std::atoi(e) != 0;
      if (pv.direct) MVUS_HIP(hipMemsetAsync(Z, 0, (size_t)ne.N3 * ncols * sizeof(double), be.stream));
    }
    const size_t mm = (size_t)std::max(pv.m, 1), ss = (size_t)pv.s3 * pv.s3;
    sep_count = mm * (2 * ss + (size_t)pv.s3 * ncols);
    sepbuf = be.alloc(sep_count);                    // [T | U | R]: one sum over the ranks
    pv.T = sepbuf; pv.U = pv.T + mm * ss; pv.R = pv.U + mm * ss;
    pv.U2 = be.alloc(mm * ss);
    pv.Ha = be.alloc(mm * ss);
    pv.Hc = be.alloc(mm * ss);
    bcr_cols = kBcrCols;
    bcr_lds = (size_t)2 * mm * pv.s3 * bcr_cols * sizeof(double);
    if (bcr_lds > 64 * 1024) { bcr_cols = 1; bcr_lds /= kBcrCols; }
    use_bcr = bcr_lds <= 64 * 1024 && !std::getenv("MVUS_SEP_SEQUENTIAL");
    MVUS_HIP(hipMalloc(reinterpret_cast<void**>(&bcr_done), mm * sizeof(unsigned)));
    MVUS_HIP(hipMemsetAsync(bcr_done, 0, mm * sizeof(unsigned), be.stream));
    if (shard && use_bcr && !wide) {
      const char* e = std::getenv("MVUS_SEP_TWO_LEVEL");
      two_level = !(e && std::atoi(e) == 0);
    }
    if (two_level) {
      has_ghost = ts.rank > 0; has_cut = ts.rank + 1 < ts.world; ncut = ts.world - 1;
      k_loc = n_own_sep - has_cut;
      const size_t nc2 = (size_t)ncols + 2 * pv.s3;
      Rloc = be.alloc((size_t)std::max(k_loc, 1) * pv.s3 * nc2);
      CGK = be.alloc(2 * ss);
      cut_count = (size_t)ncut * (2 * ss + (size_t)pv.s3 * ncols);
      cutbuf = be.alloc(cut_count);
      cutws = be.alloc(3 * (size_t)std::max(ncut, 1) * ss);
      bcr_cols_loc = kBcrCols;
      bcr_lds_loc = (size_t)2 * std::max(k_loc, 1) * pv.s3 * bcr_cols_loc * sizeof(double);
      if (bcr_lds_loc > 64 * 1024) { bcr_cols_loc = 1; bcr_lds_loc /= kBcrCols; }
    }
    win_prepare();
  }
  // Window-major assembly: needs every camera's frames in non-decreasing order (HostProblem::frames_sorted; anything else keeps the
  // detection-major kernel with its atomics).  Window length: about one wavefront of detections per (window, camera) -- the
  // Wn + 3 spans that reach a window hold (Wn + 3) * M / (C * N) detections on average -- within [4, 16] control points.
  void win_prepare() {
    const HostProblem& hp = be.hp;
    use_win = hp.frames_sorted && hp.M > 0 && ne.N > 0 && hp.C <= 64 * kWinWaves && !std::getenv("MVUS_ASM_ATOMIC");
    // (the kernel keeps absolute detection indices in 32 bits and a camera's range length in 24: k_assemble_windows, cam_range)
    if (hp.M >= (int64_t)1 << 31) use_win = false;
    for (int c = 0; c < hp.C && use_win; ++c) if (hp.det_off[c + 1] - hp.det_off[c] >= (1 << 24)) use_win = false;
    if (!use_win) return;
    // Window length AND camera groups, from one cost model.  A (window, camera) pair costs its batches of 64 staged detections -- (Wn + 3)
    // spans reach a window, so camera c brings n_c = (Wn + 3) rho_c + 3 of them, rho_c = detections per knot span -- a fixed part per
    // batch (evaluation, matrix-core pass) and a part per detection (accumulation); a workgroup walks the cameras of its group (C / G of
    // them, dealt over its four wavefronts) and pays a prologue of its own (the window's spline records); the grid of nwin x G workgroups
    // runs two per CU at a time.  Short windows repeat more evaluations (the three spans below a window) but fill the machine; camera
    // groups (round 6) fill it when the windows alone cannot -- a time shard's slice, few control points -- as long as a wavefront still
    // walks more than one camera.  MVUS_WIN / MVUS_WIN_GROUPS override.
    int Wn = 8, G = 1;
    {
      int ncu = 256;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, be.device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
      const double slots = 2.0 * ncu;
      const double nspan = shard ? std::max(1, own_hi - own_lo) : std::max(1, hp.N);      // control points the held detections spread over
      auto wg_cost = [&](int w) {
        double wg = 0.0;
        for (int c = 0; c < hp.C; ++c) {
          const double nc = (w + 3) * (double)(hp.det_off[c + 1] - hp.det_off[c]) / nspan + 3.0;
          wg += 0.45 * std::ceil(nc / 64.0) + 0.55 * nc / 64.0 + 0.15;      // + the camera's own set-up and stores
        }
        return wg;
      };
      // one group: the model of round 4 (measured optima 10 / 3 / 8 / 3 at configs[2] / [1] / [3] / [4], picks within 6 %)
      double best = 1e300;
      for (int w = 3; w <= kWinMaxW; ++w) {
        const double t = std::max(1.0, std::ceil((double)ne.N / w) / slots) * wg_cost(w);
        if (t < best * 0.999) { best = t; Wn = w; }
      }
      // camera groups only where that choice leaves a quarter or more of the workgroup slots empty, and only in ONE round of workgroups
      // (measured, `tools/micro/win_group_sweep.sh`: with the slots full, groups + longer windows are level at configs[2] -- 95-97 us against
      // 91-93 -- and the model cannot tell 13 x 4 (114 us) from 21 x 4 (97 us) there)
      if (std::ceil((double)ne.N / Wn) <= 0.75 * slots) {
        double bt = wg_cost(Wn) + 0.5;
        for (int g = 2; g <= 4; g *= 2) {
          if (hp.C <= kWinWaves * (g / 2)) break;
          for (int w = 3; w <= kWinMaxW; ++w) {
            if (std::ceil((double)ne.N / w) * g > 1.03 * slots) continue;      // (a handful of late workgroups is no second round)
            const double t = wg_cost(w) / g + 0.5;
            if (t < bt * 0.999) { bt = t; Wn = w; G = g; }
          }
        }
      }
    }
    if (const char* e = std::getenv("MVUS_WIN")) { if (std::atoi(e) > 0) Wn = std::min(kWinMaxW, std::atoi(e)); }
    if (const char* e = std::getenv("MVUS_WIN_GROUPS")) { if (std::atoi(e) > 0) G = std::min(8, std::atoi(e)); }
    wv.Wn = Wn; wv.nwin = (ne.N + Wn - 1) / Wn; wv.Ntot = hp.N;
    wv.G = G;
    if (std::getenv("MVUS_DEBUG")) std::fprintf(stderr, "window-major assembly: %d control points per window, %d windows x %d camera group(s)\n", Wn, wv.nwin, G);
    wv.band_part = G > 1 ? be.alloc((size_t)G * ne.N * (3 + ne.W * 9)) : nullptr;
    const size_t psz = (size_t)(ne.B + 1) * (ne.B + 2) / 2;
    wv.Apart = be.alloc((size_t)wv.nwin * ne.C * psz);
    const size_t bytes_cw = sizeof(CamWin) * (size_t)hp.C, bytes_t = sizeof(double) * ((size_t)hp.N + 1), bytes_l = sizeof(int32_t) * (((size_t)hp.flut_len + 3) & ~(size_t)3);
    const size_t bytes_r = sizeof(int4) * (size_t)std::max(1, hp.N), bytes_p = sizeof(int32_t) * (size_t)hp.C;
    MVUS_HIP(hipMalloc(&win_tables, bytes_cw + 2 * bytes_t + bytes_l + bytes_r + bytes_p));
    // cameras dealt to the four wavefronts of a window by decreasing detection count, back and forth (0 1 2 3 3 2 1 0 ...): every
    // wavefront walks about the same number of detections whatever the cameras' frame rates
    std::vector<int32_t> perm((size_t)hp.C);
    {
      std::vector<int32_t> byc((size_t)hp.C);
      for (int c = 0; c < hp.C; ++c) byc[c] = c;
      std::stable_sort(byc.begin(), byc.end(), [&](int32_t u, int32_t v) { return hp.det_off[u + 1] - hp.det_off[u] > hp.det_off[v + 1] - hp.det_off[v]; });
      const int ns = kWinWaves * wv.G;                       // wavefront slots that share the cameras of a window (G workgroups of four)
      std::vector<std::vector<int32_t>> of(ns);
      for (int i = 0; i < hp.C; ++i) { const int r = i % (2 * ns); of[r < ns ? r : 2 * ns - 1 - r].push_back(byc[i]); }
      // slot u walks perm[u], perm[u + ns], ...: it takes ceil((C - u) / ns) cameras, the first slots one more than the last ones
      std::vector<int32_t> flat;
      for (int v = 0; v < ns; ++v) flat.insert(flat.end(), of[v].begin(), of[v].end());
      std::vector<size_t> take(ns);
      for (int v = 0; v < ns; ++v) take[v] = v < hp.C ? (size_t)(hp.C - v + ns - 1) / ns : 0;
      size_t pos = 0;
      for (int v = 0; v < ns; ++v) for (size_t i = 0; i < take[v]; ++i) perm[(size_t)v + ns * i] = flat[pos++];
    }
    std::vector<int4> crec((size_t)std::max(1, hp.N), int4{0, 0, 0, 0});
    for (int sI = 0; sI < hp.S; ++sI) {
      const int ns = hp.ctrl_off[sI + 1] - hp.ctrl_off[sI];
      for (int jj = 0; jj < ns; ++jj)        // span l = jj + 3: knots t[l-2 .. l+3] start at knot_off + jj + 1
        crec[(size_t)hp.ctrl_off[sI] + jj] = int4{hp.xoff[sI] + jj, ns, hp.knot_off[sI] + jj + 1, (jj == 0 ? 1 : 0) | (jj + 4 == ns ? 2 : 0) | (jj + 4 > ns ? 4 : 0)};
    }
    char* base = static_cast<char*>(win_tables);
    MVUS_HIP(hipMemcpyAsync(base, hp.cam_win.data(), bytes_cw, hipMemcpyHostToDevice, be.stream));
    MVUS_HIP(hipMemcpyAsync(base + bytes_cw, hp.win_tlo.data(), bytes_t, hipMemcpyHostToDevice, be.stream));
    MVUS_HIP(hipMemcpyAsync(base + bytes_cw + bytes_t, hp.win_thi.data(), bytes_t, hipMemcpyHostToDevice, be.stream));
    wv.cw = reinterpret_cast<const CamWin*>(base);
    wv.tlo = reinterpret_cast<const double*>(base + bytes_cw);
    wv.thi = reinterpret_cast<const double*>(base + bytes_cw + bytes_t);
    int32_t* flut = reinterpret_cast<int32_t*>(base + bytes_cw + 2 * bytes_t);
    wv.flut = flut;
    MVUS_HIP(hipMemcpyAsync(base + bytes_cw + 2 * bytes_t + bytes_l, crec.data(), bytes_r, hipMemcpyHostToDevice, be.stream));   // (synchronised below: crec outlives the copy)
    wv.crec = reinterpret_cast<const int4*>(base + bytes_cw + 2 * bytes_t + bytes_l);
    MVUS_HIP(hipMemcpyAsync(base + bytes_cw + 2 * bytes_t + bytes_l + bytes_r, perm.data(), bytes_p, hipMemcpyHostToDevice, be.stream));
    wv.cam_perm = reinterpret_cast<const int32_t*>(base + bytes_cw + 2 * bytes_t + bytes_l + bytes_r);
    hipLaunchKernelGGL(k_frame_lut, dim3((unsigned)((hp.flut_len + 255) / 256)), dim3(256), 0, be.stream, be.dp, wv.cw, flut, (long long)hp.flut_len);
    MVUS_HIP(hipGetLastError());
    MVUS_HIP(hipStreamSynchronize(be.stream));
    win_lds = (size_t)win_lds_doubles(ne.B) * sizeof(double);
    if (hp.calib) MVUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_windows<18>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_lds));
    else MVUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_windows<9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_lds));
  }
  ~HipSchur() {
    for (double* p : {wv.band_part, Rloc, CGK, cutbuf, cutws, Dl, NEset[0], NEset[1], Lb, Z, G, G0, S, S2, Linv, rhs, pc, DG, px, pv.VW, sepbuf, pv.U2, pv.Ha, pv.Hc, wv.Apart, rcs.Simg, rcs.Tsc, rcs.x}) if (p) be.release(p);
    if (win_tables) (void)hipFree(win_tables);
    if (part_tables) (void)hipFree(part_tables);
    if (halo_tables) (void)hipFree(halo_tables);
    if (fail) (void)hipFree(fail);
    if (rcs_flags) (void)hipFree(rcs_flags);
    if (bcr_done) (void)hipFree(bcr_done);
    if (fail_host) (void)hipHostFree(fail_host);
  }

  // ---- two sets of blocks ----
  bool pv_ready = false;
  void bind_ne(int k) {
    NE = NEset[k];
    ne.A = NE; ne.gc = NE + off_gc; ne.Cb = NE + off_Cb; ne.gs = NE + off_gs; ne.Et = NE + off_Et; ne.Apart = NE + off_Apart;
    if (pv_ready) { pv.Et = ne.Et; pv.gs = ne.gs; }
    if (shard) { D = NE + nAg + halo_count; gx = D + be.hp.n; }      // (a time shard's diag(H) and g ride inside the summed head of the set)
  }
  // Speculative linearisation (ba_schur.h, launch_trial): one rank, the fused window-major assembly (every entry written by one plain
  // store: the second set needs no clearing), scalars fetched behind an event.  MVUS_NO_SPEC=1 keeps the sequential form for an A/B.
  // Time shards (round 6): the speculative assembly carries its collective with it -- every rank takes the same decision from the same
  // summed scalars, so every rank enqueues the same sequence; the scalars' copy to the host is enqueued in front of it (fetch_enqueue).
  bool spec_ok(int jac_mode) {
    if (!use_win || jac_mode != MVUS_JAC_ANALYTIC) return false;
    if (shard ? !be.spec_on_shards() : !be.scal_direct()) return false;
    if (std::getenv("MVUS_NO_SPEC") || std::getenv("MVUS_LM_MATERIALIZE_J")) return false;
    if (!NEset[1]) NEset[1] = be.alloc(ne_count + n_apart);
    return true;
  }
  void linearize_spec(BE&, const double* x_dev, double* f_dev, int jac_mode) {
    const bool pending = diag_pending;      // D and g in x order belong to the set the solver reads: untouched until adopt_spec
    // where the fetch that follows stops waiting: the start of the assembly kernel (a word in mapped memory), else an event in front of it
    if (!be.fetch_poll_begin(&wv.mark, &wv.mark_val)) { wv.mark = nullptr; be.fetch_mark(); }
    bind_ne(ne_cur ^ 1);
    linearize(be, x_dev, f_dev, jac_mode, true);
    bind_ne(ne_cur);
    wv.mark = nullptr;
    diag_pending = pending;
  }
  void adopt_spec() { ne_cur ^= 1; bind_ne(ne_cur); diag_pending = !shard; }
  // a rejected trial's speculative assembly may have raised the "row left the slice" flag for a point nobody keeps: forget it (a flag
  // raised by the CURRENT point's assembly has been acted on before any trial)
  void drop_spec() { if (shard) MVUS_HIP(hipMemsetAsync(fail + 1, 0, sizeof(int), be.stream)); }

  // x_fused != nullptr: the detection rows' Jacobian is evaluated inside the assembly kernel at x_fused (no J in memory);
  // the motion rows (O(T), tiny) still go through k_motion
  // the storage the assembly adds into; the LM driver has it zeroed beside its first residual evaluation (mark_cleared)
  bool ne_cleared = false;
  double* clear_ptr() { return use_win ? nullptr : NE; }      // (the window-major assembly writes every entry: nothing to clear)
  int64_t clear_len() const { return (int64_t)(ne_count + n_apart); }
  void mark_cleared() { ne_cleared = true; }
  bool last_atomic = false;                                // the last assembly went through the detection-major kernel (fp64 atomics)
  void motion_rows(const double* f_dev) {
    // (one rank: the row-ordered kernel in both modes -- 22 us against 26 for the LDS-window one at configs[1], and one source of
    // run-to-run differences less; a time shard keeps k_assemble_motion, which also reports rows that leave the slice)
    if (be.hp.T > 0 && wide) {
      hipLaunchKernelGGL(k_det_motion_wide, dim3((unsigned)ne.N), dim3(64), 0, be.stream, be.dp, be.mJ, be.mctrl, f_dev + 2 * be.hp.M, ne);
    } else if (be.hp.T > 0 && !shard && ne.W <= kDetMotW) {
      hipLaunchKernelGGL(k_det_motion, dim3((unsigned)((ne.N + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads), 0, be.stream, be.dp, be.mJ, be.mctrl,
                         f_dev + 2 * be.hp.M, ne);
    } else if (be.hp.T > 0)
      hipLaunchKernelGGL(k_assemble_motion, dim3((be.hp.T + kThreads - 1) / kThreads), dim3(kThreads), 0, be.stream, be.dp, be.mJ, be.mctrl,
                         f_dev + 2 * be.hp.M, ne);
  }
  void assemble_local(const double* f_dev, const double* x_fused = nullptr, const int32_t* span_held = nullptr) {
    if (x_fused && use_win) {
      // window-major: every entry of A, gc, Cb, gs, Et is written by exactly one thread -- no clearing pass, no atomics.
      // It starts from the knot span of every detection at x: left behind by the residual evaluation at x that precedes every
      // linearisation (else evaluated now), or the held analytic Jacobian's own table
      ne_cleared = false; last_atomic = false;
      // (a time shard: the part of the packed head that is SUMMED over the ranks without being written in full here -- the halo exchange
      // buffer of the other ranks' cuts, diag(H) and g of the columns outside this slice -- starts from zero: ~1 MB, not the 36 MB of blocks)
      double* const zr = shard ? NE + nAg : (double*)nullptr;               // (cleared by k_cam_block_sum below: no launch of its own)
      const long long zn = shard ? (long long)(halo_count + 2 * (size_t)be.hp.n) : 0;
      if (span_held) wv.span = span_held;
      else {
        if (be.rspan_for != x_fused) be.residual(x_fused, const_cast<double*>(f_dev));      // (f_dev holds f(x) already: the same values again)
        wv.span = be.rspan;
      }
      be.ensure_cams(x_fused);
      if (be.hp.calib) {
        const unsigned sumg = (unsigned)be.hp.C + (wv.G > 1 ? (unsigned)(((long long)ne.N * (3 + ne.W * 9) + 1023) / 1024) : 0u);
        hipLaunchKernelGGL(k_assemble_windows<18>, dim3(wv.nwin * wv.G), dim3(kWinThreads), win_lds, be.stream, be.dp, ne, wv, be.cams, x_fused);
        hipLaunchKernelGGL(k_cam_block_sum<18>, dim3(sumg), dim3(1024), 0, be.stream, be.hp.C, wv.nwin, wv.Apart, ne, wv.G, (const double*)wv.band_part, zr, zn);
      } else {
        const unsigned sumg = (unsigned)be.hp.C + (wv.G > 1 ? (unsigned)(((long long)ne.N * (3 + ne.W * 9) + 1023) / 1024) : 0u);
        hipLaunchKernelGGL(k_assemble_windows<9>, dim3(wv.nwin * wv.G), dim3(kWinThreads), win_lds, be.stream, be.dp, ne, wv, be.cams, x_fused);
        hipLaunchKernelGGL(k_cam_block_sum<9>, dim3(sumg), dim3(1024), 0, be.stream, be.hp.C, wv.nwin, wv.Apart, ne, wv.G, (const double*)wv.band_part, zr, zn);
      }
      motion_rows(f_dev);
      MVUS_HIP(hipGetLastError());
      return;
    }
    if (!ne_cleared) be.fill(NE, 0.0, (int64_t)(ne_count + n_apart));      // one launch (hipMemsetAsync splits 36 MB into two fill kernels)
    ne_cleared = false;
    last_atomic = be.dp.n_chunks > 0;
    if (be.dp.n_chunks > 0) {
      const int nc = be.dp.n_chunks;
      const dim3 g(kGaParts * nc), b(kGaThreads);
      if (x_fused) {
        be.ensure_cams(x_fused);
        if (be.hp.calib) hipLaunchKernelGGL((k_assemble_spans<30, true>), g, b, 0, be.stream, be.dp, (const double*)nullptr, (const int32_t*)nullptr, f_dev, ne, be.cams, x_fused);
        else hipLaunchKernelGGL((k_assemble_spans<21, true>), g, b, 0, be.stream, be.dp, (const double*)nullptr, (const int32_t*)nullptr, f_dev, ne, be.cams, x_fused);
      } else {
        if (be.hp.calib) hipLaunchKernelGGL((k_assemble_spans<30, false>), g, b, 0, be.stream, be.dp, be.J, be.span, f_dev, ne, be.cams, (const double*)nullptr);
        else hipLaunchKernelGGL((k_assemble_spans<21, false>), g, b, 0, be.stream, be.dp, be.J, be.span, f_dev, ne, be.cams, (const double*)nullptr);
      }
    }
    if (be.dp.n_chunks > 0) {
      if (be.hp.calib) hipLaunchKernelGGL(k_cam_block_reduce<18>, dim3(be.hp.C, 2), dim3(1024), 0, be.stream, be.dp, ne);
      else hipLaunchKernelGGL(k_cam_block_reduce<9>, dim3(be.hp.C, 2), dim3(1024), 0, be.stream, be.dp, ne);
    }
    motion_rows(f_dev);
    MVUS_HIP(hipGetLastError());
  }
  // Linearise at x: residual f (unless the caller already holds f(x) in f_dev), Jacobian, normal equations.  With the
  // analytic Jacobian the detection rows are fused (assemble_local above); other Jacobian modes materialise J first.
  void linearize(BE&, const double* x_dev, double* f_dev, int jac_mode, bool f_valid) {
    RoctxRange range("mvus linearise");
    const bool fused = jac_mode == MVUS_JAC_ANALYTIC && !std::getenv("MVUS_LM_MATERIALIZE_J");
    if (!fused) { be.jacobian(x_dev, f_dev, jac_mode); assemble(be, f_dev); return; }
    if (!f_valid) be.residual(x_dev, f_dev);
    if (be.hp.T > 0) be.motion_jacobian(x_dev, f_dev);
    be.has_jacobian = false;                               // no materialised J belongs to this point
    assemble(be, f_dev, x_dev);
  }
  void assemble(BE&, const double* f_dev, const double* x_fused = nullptr, const int32_t* span_held = nullptr) {
    assemble_local(f_dev, x_fused, span_held);
    if (shard) {
      // time shard: sum the camera blocks and the blocks of the control points near a cut; the cross block never moves
      // ... and diag(H), g in x order: every rank adds its PARTIAL sums (rows of a control point near a cut sit on two
      // ranks), so they go into the same all-reduce, before the halo blocks are completed
      double* hb = NE + nAg;
      const int halo = be.tshard.halo;
      hipLaunchKernelGGL(k_halo_copy, dim3(256), dim3(256), 0, be.stream, ne, halo, nbound, halo_tables, halo_tables + nbound, Ntot, hb, 0, be.dp, D, gx, (long long)be.hp.n);
      be.reduce(NE, nAg + halo_count + 2 * (size_t)be.hp.n);
      hipLaunchKernelGGL(k_halo_copy, dim3(256), dim3(256), 0, be.stream, ne, halo, nbound, halo_tables, halo_tables + nbound, Ntot, hb, 1, be.dp, D, gx, (long long)be.hp.n);
    } else {
      be.reduce(NE, ne_count);          // observation shards: one sum-all-reduce of the packed normal-equation blocks per iteration
      diag_pending = true;              // D and g (x order) are written by the next solve's first kernel, or by flush_diag()
    }
    MVUS_HIP(hipGetLastError());
  }

  // normal equations of the Jacobian the backend holds: the analytic Jacobian of x_cur goes through the fused window-major assembly
  // (the LM path's kernel; MVUS_NE_FROM_J=1 forms them from the stored blocks instead), anything else is assembled from J
  void assemble_held(BE&) {
    const bool fused = be.held_analytic_at_xcur && use_win && !std::getenv("MVUS_NE_FROM_J");
    assemble(be, be.f_cur, fused ? be.x_cur : nullptr, fused ? be.span : nullptr);
  }

  void flush_diag() {
    if (!diag_pending) return;
    const int tot = ne.CB + ne.N3;
    hipLaunchKernelGGL(k_ne_diag_grad, dim3((tot + 255) / 256), dim3(256), 0, be.stream, be.dp, ne, 0, D, gx);
    diag_pending = false;
  }
  // valid once the stream reaches this point: written by the solve that follows an assembly, or flushed here when none has run
  const double* grad_ptr() { flush_diag(); return gx; }
  const double* diag_ptr() { flush_diag(); return D; }
  const double* step_ptr() const { return px; }
  const int* fail_ptr() const { return fail; }
  bool solve_ok() const {                 // valid after the stream has been synchronised (the driver's fetch)
    if (shard) { fail_host[0] = fail_sum_code(be.scal_host[be.kFailSumSlot]); fail_host[1] = fail_sum_flag(be.scal_host[be.kFailSumSlot + 1]); }
    if (fail_host[1] & 2) throw HipError{"fused assembly: the span table does not belong to the point being linearised (internal error)"};
    if (fail_host[1] != 0) {
      be.reshard_flag = true;        // (the LM driver hands the point it has reached back to the caller: MVUS_E_RESHARD)
      throw HipError{"time shard: a detection or motion row reaches control points outside this rank's slice +- halo (the time stamps have drifted since the cuts were made): re-cut at the returned point", MVUS_E_RESHARD};
    }
    if (fail_host[0] != 0 && std::getenv("MVUS_DEBUG")) std::fprintf(stderr, "schur solve: fail code %d\n", fail_host[0]);
    return fail_host[0] == 0;
  }
  // A failed solve that is NOT a numerical failure: a workgroup of k_rcs_factor gave up waiting for the factor workgroup's hand-over
  // flag (a GPU shared with other processes or streams may not schedule workgroup 0 of a launch before the others: forward progress
  // between the workgroups of one launch is assumed there, not guaranteed).  The handle then takes the separate-launch route for the
  // rows below a super-block (MVUS_RCS_TRSM=launch: bit-identical results, no spinning) for the rest of its life and the caller
  // repeats the solve at the SAME damping -- raising lambda, the answer to a lost pivot, would silently change the iterates.
  bool retry_same() {
    if (fail_host[0] != kFailHandover || (rcs_trsm_launch && !bcr_fused)) return false;
    rcs_trsm_launch = true; bcr_fused = false;
    ++handover_timeouts;
    if (std::getenv("MVUS_DEBUG")) std::fprintf(stderr, "schur solve: hand-over time-out in k_rcs_factor / k_sep_bcr_levels -> a launch per stage from now on\n");
    return true;
  }

  template <int BWT, int S3T>
  void band_chain() {
    const dim3 gsolve(pv.P, (ncols + 63) / 64 + 1);      // + one block row for the coupling columns
    if (overlap_chol && !pv.direct) {
      const int cb = (pv.P + 7) / 8 * 8;
      hipLaunchKernelGGL(k_cholesky_and_rhs<BWT>, dim3((unsigned)(cb + xcd_grid(rhs_tiles_z))), dim3(256), 0, be.stream, pv, Lb, fail, cb, ne, ncols, Z, rhs_tiles_z, pv.Dl ? pv.seprow : (const unsigned char*)nullptr);
    } else {
      hipLaunchKernelGGL(k_part_cholesky<BWT>, dim3(pv.P), dim3(64), 0, be.stream, pv, Lb, fail);
    }
    hipLaunchKernelGGL(k_part_solve<BWT>, dim3(xcd_grid(pv.P * (int)gsolve.y)), dim3(64), 0, be.stream, pv, ncols, Lb, Z, (int)gsolve.y);
    if (pv.m > 0) {
      // other ranks' separators: zero here (one level: the whole system is summed; two levels: k_part_reduce writes every block this rank reads)
      if (shard && !two_level) MVUS_HIP(hipMemsetAsync(sepbuf, 0, sep_count * sizeof(double), be.stream));
      // one rank, cyclic reduction: only the matrix blocks first; the right-hand sides ride beside the one-workgroup tail
      const bool split = overlap_chol && !shard && use_bcr && pv.nt > 0;
      if (pv.nt > 0) hipLaunchKernelGGL(k_part_reduce<BWT>, dim3(pv.nt, split ? 1 : (pv.s3 * ncols + 255) / 256), dim3(256), 0, be.stream, pv, ncols, Lb, Z, split ? 1 : 3);
      if (shard && two_level) {
        // round 6: the local separators are eliminated by this rank alone; only the world - 1 cut separators are summed (k_sep2_* above)
        const int q0 = pv.q_off, cq0 = q0 - has_ghost, cqn = n_own_sep + has_ghost;
        // (no back-correction: the correction rows of the Schur product need this rank's parts of the reduced right-hand sides R_S as
        // they are before the solve overwrites them)
        const int nc2 = ncols + 2 * S3T;
        double *CG = CGK, *CK = CGK + (size_t)S3T * S3T;
        {
          const long long tot = std::max((long long)k_loc * S3T * nc2 + 2 * S3T * S3T, ncorr > 0 ? (long long)cqn * S3T * ne.CB : 0LL);
          hipLaunchKernelGGL(k_sep2_build<S3T>, dim3((unsigned)std::min<long long>(2048, (tot + 255) / 256)), dim3(256), 0, be.stream, pv, q0, k_loc, has_ghost, has_cut, ncols, Rloc, CG, CK,
                             cq0, ncorr > 0 ? cqn * S3T : 0, ne.CB);
        }
        if (k_loc > 0) {
          PartView pl = pv;                  // the local chain: nodes q0 .. q0 + k - 1 of the global arrays, renumbered from 0
          const size_t ssz = (size_t)S3T * S3T;
          pl.m = k_loc; pl.T = pv.T + q0 * ssz; pl.U = pv.U + q0 * ssz; pl.U2 = pv.U2 + q0 * ssz; pl.Ha = pv.Ha + q0 * ssz; pl.Hc = pv.Hc + q0 * ssz; pl.R = Rloc;
          int h = 1;
          for (; h <= pl.m && pl.m / (2 * h) > kBcrTailNs; h <<= 1)
            hipLaunchKernelGGL(k_sep_bcr_level<S3T>, dim3(pl.m / (2 * h)), dim3(64), 0, be.stream, pl, h, fail);
          if (h <= pl.m) hipLaunchKernelGGL(k_sep_bcr_tail<S3T>, dim3(1), dim3(bcr_tail_waves(S3T) * 64), 0, be.stream, pl, h, fail);
          if (bcr_cols_loc == kBcrCols) hipLaunchKernelGGL((k_sep_bcr_rhs<S3T, kBcrCols>), dim3((nc2 + kBcrCols - 1) / kBcrCols), dim3(256), bcr_lds_loc, be.stream, pl, nc2);
          else hipLaunchKernelGGL((k_sep_bcr_rhs<S3T, 1>), dim3(nc2), dim3(256), bcr_lds_loc, be.stream, pl, nc2);
        }
        hipLaunchKernelGGL(k_sep2_reduce<S3T>, dim3((unsigned)((cut_count + 255) / 256)), dim3(256), 0, be.stream, pv, q0, k_loc, has_ghost, has_cut,
                           be.tshard.rank, ncut, ncols, (const double*)Rloc, (const double*)CG, (const double*)CK, cutbuf);
        be.reduce(cutbuf, cut_count);        // every rank now holds the cut system
        {
          PartView pc = pv;
          pc.m = ncut; pc.T = cutbuf; pc.U = cutbuf + (size_t)ncut * S3T * S3T; pc.R = cutbuf + 2 * (size_t)ncut * S3T * S3T;
          // (the cyclic reduction again, not the sequential block-tridiagonal kernels: 7 nodes are three levels in one workgroup -- 63 us of
          // k_sep_factor + k_sep_rhs measured at world 8, configs[3], against ~25)
          pc.U2 = cutws; pc.Ha = cutws + (size_t)ncut * S3T * S3T; pc.Hc = cutws + 2 * (size_t)ncut * S3T * S3T;
          if (ncut > 2 * kBcrTailNs + 1) {                 // (more than 33 ranks: the general kernels)
            hipLaunchKernelGGL(k_sep_factor<S3T>, dim3(1), dim3(64), 0, be.stream, pc, fail);
            hipLaunchKernelGGL(k_sep_rhs<S3T>, dim3((ncols + 63) / 64), dim3(64), 0, be.stream, pc, ncols);
          } else {
            hipLaunchKernelGGL(k_sep_bcr_tail<S3T>, dim3(1), dim3(bcr_tail_waves(S3T) * 64), 0, be.stream, pc, 1, fail);
            const size_t lds = (size_t)2 * ncut * S3T * kBcrCols * sizeof(double);
            hipLaunchKernelGGL((k_sep_bcr_rhs<S3T, kBcrCols>), dim3((ncols + kBcrCols - 1) / kBcrCols), dim3(256), lds, be.stream, pc, ncols);
          }
        }
        {
          const long long tot = (long long)(k_loc + 2) * S3T * ncols;
          hipLaunchKernelGGL(k_sep2_finish<S3T>, dim3((unsigned)std::min<long long>(2048, (tot + 255) / 256)), dim3(256), 0, be.stream, pv, q0, k_loc, has_ghost, has_cut,
                             be.tshard.rank, ncut, ncols, (const double*)Rloc, (const double*)cutbuf);
        }
      } else {
      if (shard) {
        be.reduce(sepbuf, sep_count);        // every rank now holds the whole separator system
        // (no back-correction: the Schur product needs the reduced right-hand sides R_S as they are BEFORE the in-place solve -- on one rank
        // part_reduce_rhs writes them to pv.Dl as it forms them, here they exist only after the sum)
        if (ncorr > 0) MVUS_HIP(hipMemcpy2DAsync(pv.Dl, (size_t)ne.CB * sizeof(double), pv.R, (size_t)ncols * sizeof(double), (size_t)ne.CB * sizeof(double),
                                                 (size_t)pv.m * pv.s3, hipMemcpyDeviceToDevice, be.stream));
      }
      if (use_bcr) {
        int h = 1;                                        // wide levels: one launch for all of them (or a launch each); the rest in one workgroup
        {
          BcrLevels lv{};
          int total = 0, hh = 1;
          for (; hh <= pv.m && pv.m / (2 * hh) > kBcrTailNs && lv.nlev < 4; hh <<= 1) { lv.first[lv.nlev] = total; lv.h[lv.nlev] = hh; total += pv.m / (2 * hh); ++lv.nlev; }
          if (bcr_fused && !shard && lv.nlev >= 2 && !(hh <= pv.m && pv.m / (2 * hh) > kBcrTailNs)) {
            bcr_epoch += 8;
            hipLaunchKernelGGL(k_sep_bcr_levels<S3T>, dim3(total), dim3(64), 0, be.stream, pv, lv, bcr_done, bcr_epoch, rcs_spin_limit, fail);
            h = hh;
          }
        }
        for (; h <= pv.m && pv.m / (2 * h) > kBcrTailNs; h <<= 1)
          hipLaunchKernelGGL(k_sep_bcr_level<S3T>, dim3(pv.m / (2 * h)), dim3(64), 0, be.stream, pv, h, fail);
        if (split) {
          const int tb = bcr_tail_waves(S3T) * 64, ny = (pv.s3 * ncols + tb - 1) / tb;
          hipLaunchKernelGGL((k_bcr_tail_and_reduce_rhs<BWT, S3T>), dim3(1 + pv.nt * ny), dim3(tb), 0, be.stream, pv, h, (int)(h <= pv.m), fail, ncols, Lb, Z, ny);
        } else if (h <= pv.m) hipLaunchKernelGGL(k_sep_bcr_tail<S3T>, dim3(1), dim3(bcr_tail_waves(S3T) * 64), 0, be.stream, pv, h, fail);
        if (bcr_cols == kBcrCols) hipLaunchKernelGGL((k_sep_bcr_rhs<S3T, kBcrCols>), dim3((ncols + kBcrCols - 1) / kBcrCols), dim3(256), bcr_lds, be.stream, pv, ncols);
        else hipLaunchKernelGGL((k_sep_bcr_rhs<S3T, 1>), dim3(ncols), dim3(256), bcr_lds, be.stream, pv, ncols);
      } else {
        hipLaunchKernelGGL(k_sep_factor<S3T>, dim3(1), dim3(64), 0, be.stream, pv, fail);
        hipLaunchKernelGGL(k_sep_rhs<S3T>, dim3((ncols + 63) / 64), dim3(64), 0, be.stream, pv, ncols);
      }
      }      // (one level: the whole separator system on every rank)
      if (ncorr == 0) {                                   // (one rank: no back-correction -- see where ncorr is set)
        const int gy = (ncols + 63) / 64, gz = (kPartRowsMax + kBackRows - 1) / kBackRows;
        const dim3 gback(xcd_grid(pv.P * gy * gz));
        hipLaunchKernelGGL(k_part_back<S3T>, gback, dim3(64), 0, be.stream, pv, ncols, Z, gy, gz);
      }
    }
  }

  // time shards: the failure flags of a solve are SUMMED with the step (px[n], px[n + 1]); the LM driver's trial kernel forwards the sums
  // to the scalars its fetch brings to the host (trial_follows), any other caller gets them by a copy to the same two slots
  bool fail_in_scalars() const { return shard; }
  const double* fail_sum_ptr() const { return shard ? px + be.hp.n : (const double*)nullptr; }
  void solve_async(double lambda, bool trial_follows = false) {
    RoctxRange range("mvus schur solve");
    const long long nLb = (long long)ne.N3 * (BW + 1);
    const long long nZ = (long long)ne.N3 * ncols;          // >= nLb: one launch covers both passes
    const int rhs_tiles = (int)((std::max(nZ, nLb) + 255) / 256);
    overlap_chol = std::getenv("MVUS_NO_OVERLAP") == nullptr && !wide;
    if (overlap_chol) {
      const long long nband = std::max<long long>(nLb, (long long)ne.CB + ne.N3);
      hipLaunchKernelGGL(k_band_pack, dim3((unsigned)((nband + 255) / 256)), dim3(256), 0, be.stream, ne, lambda, BW, Lb, fail, be.dp, (int)diag_pending, D, gx);
      rhs_tiles_z = (int)((nZ + 255) / 256);
    } else {
      hipLaunchKernelGGL(k_build_rhs, dim3((unsigned)xcd_grid(rhs_tiles)), dim3(256), 0, be.stream, ne, ncols, Z, (!wide && pv.Dl) ? pv.seprow : (const unsigned char*)nullptr, lambda, BW, Lb, fail, be.dp,
                         (int)diag_pending, D, gx, rhs_tiles);
    }
    diag_pending = false;
    if (wide) {
      const size_t lds_c = (size_t)(BW + 1) * (BW + 1) * sizeof(double), lds_s = (size_t)(BW + 1) * 64 * sizeof(double);
      hipLaunchKernelGGL(k_band_chol_generic, dim3(1), dim3(256), lds_c, be.stream, ne.N3, BW, Lb, fail);
      hipLaunchKernelGGL(k_band_solve_generic, dim3((unsigned)((ncols + 63) / 64)), dim3(64), lds_s, be.stream, ne.N3, BW, ncols, Lb, Z);
    } else if (BW == 11) band_chain<11, 9>(); else band_chain<17, 15>();
    const int row_lo = 3 * own_lo, row_hi = 3 * own_hi;
    {
      const int nbk = (ne.CB + kGemmT - 1) / kGemmT;
      const bool corr = !wide && ncorr > 0;
      // the separators whose R_S^T X_S this rank adds: all of them, or -- time shard -- its own (global numbers q_off ...)
      // (two levels: plus the ghost -- the cut separator's correction term splits into the two neighbours' own parts of R_S)
      const int cq0 = shard ? pv.q_off - (two_level ? has_ghost : 0) : 0, cqn = shard ? n_own_sep + (two_level ? has_ghost : 0) : pv.m;
      hipLaunchKernelGGL(k_schur_gemm, dim3(8 * (nbk * (nbk + 1) / 2 + nbk) * ((nslab + 7) / 8)), dim3(256), 0, be.stream, ne, ncols, row_lo, row_hi, nslab, ne.Et, Z, G,
                         corr ? (const double*)(pv.Dl + (size_t)cq0 * pv.s3 * ne.CB) : (const double*)nullptr, (const double*)(pv.R + (size_t)cq0 * pv.s3 * ncols),
                         corr ? cqn * pv.s3 : 0);
    }
    const int ntile = (ne.CB + kNB - 1) / kNB;
    const double* Gsum = G;
    int nsl = nslab;
    if (shard) {
      const long long cnt = (long long)ne.CB * ncols;
      hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((std::max<long long>(cnt, be.hp.n) + 255) / 256)), dim3(256), 0, be.stream, cnt, nslab, G, G0, px, (long long)be.hp.n);
      be.reduce(G0, (size_t)cnt);                   // the Schur complement contributions of all time slices
      Gsum = G0; nsl = 1;
    }
    if (use_rcs) {
      // blocked L D L^T in block-image form (ba_rcs.hip.h): per super-panel of 144 unknowns one factor launch (one workgroup, the pivot
      // chain inside one CU), the rows below, the trailing blocks; one descending substitution at the end
      const int nbk = rcs.nbk, nt = (16 * nbk + 31) / 32;
      hipLaunchKernelGGL(k_rcs_finish, dim3(nt, nt + 1), dim3(256), 0, be.stream, ne, ncols, nsl, lambda, Gsum, rcs, rcs_flags);
      for (int c0 = 0; c0 < nbk; c0 += kRcsSP) {
        const int nc = std::min(kRcsSP, nbk - c0), c1 = c0 + nc, m = nbk - c1;
        // (the block rows below the super-block are solved by m more workgroups of the same launch, one step behind the chain;
        // MVUS_RCS_TRSM=launch: by a launch of their own, for A/B)
        const bool fused_rows = m > 0 && !rcs_trsm_launch;
        hipLaunchKernelGGL(k_rcs_factor, dim3(1 + (fused_rows ? m : 0)), dim3(kRcsFactorThreads), 0, be.stream, rcs, c0, fail, (int)(m == 0), pc, rcs_flags, rcs_spin_limit);
        if (m > 0) {
          if (!fused_rows) hipLaunchKernelGGL(k_rcs_trsm, dim3(m), dim3(64 * kRcsTrsmWaves), (rcs_stage_doubles(nc) + 512) * sizeof(double), be.stream, rcs, c0);
          hipLaunchKernelGGL(k_rcs_syrk, dim3((m * (m + 1) / 2 + m + 3) / 4), dim3(256), 0, be.stream, rcs, c0);
        }
      }
      const int nsp = (nbk + kRcsSP - 1) / kRcsSP;                       // (the last super-panel is solved inside its factor launch)
      if (nsp > 1) hipLaunchKernelGGL(k_rcs_backsub, dim3(1), dim3(64 * kRcsBackWaves), rcs_backsub_doubles(nbk) * sizeof(double), be.stream, rcs, pc, nsp - 2);
    } else {
      hipLaunchKernelGGL(k_schur_finish, dim3(ntile, ntile), dim3(kFinThreads), 0, be.stream, ne, ncols, nsl, lambda, Gsum, S, Linv, fail);
      const int nn = ne.CB;
      double* a = S;
      double* b = S2;
      for (int kb = 0; kb < nn; kb += kNB) {
        const int nb = std::min(kNB, nn - kb), below = nn + 1 - (kb + nb);       // rows under the panel incl. the rhs row
        const bool last = kb + nb >= nn;
        hipLaunchKernelGGL(k_gj_step, dim3((below + kNB - 1) / kNB, (nn + kNB - 1) / kNB), dim3(kGjThreads), 0, be.stream, nn, kb, a, b, Linv, fail,
                           last ? pc : (double*)nullptr);
        std::swap(a, b);
      }
    }
    const int nrows = row_hi - row_lo, per = kThreads / 64;
    hipLaunchKernelGGL(k_back_substitute, dim3((unsigned)std::max(1, (nrows + per - 1) / per)), dim3(kThreads), 0, be.stream, be.dp, ne, ncols,
                       row_lo, row_hi, (int)(!shard || be.tshard.rank == 0), Z, pc, px, fail, (!shard && be.scal_direct()) ? fail_map : (int*)nullptr,
                       shard ? px + be.hp.n : (double*)nullptr);
    if (!wide && ncorr > 0) {                          // the interiors' rows were computed from uncorrected columns: one vector is corrected here
      if (BW == 11) hipLaunchKernelGGL(k_back_correct<9>, dim3(pv.P), dim3(256), 0, be.stream, be.dp, ne, pv, ncols, (const double*)pc, px);
      else hipLaunchKernelGGL(k_back_correct<15>, dim3(pv.P), dim3(256), 0, be.stream, be.dp, ne, pv, ncols, (const double*)pc, px);
    }
    if (shard) be.reduce(px, (size_t)be.hp.n + 2);             // every rank's part of the step (+ failure flags, packed by k_back_substitute)
    MVUS_HIP(hipGetLastError());
    if (shard) {
      if (!trial_follows) MVUS_HIP(hipMemcpyAsync(be.scal_host + be.kFailSumSlot, px + be.hp.n, 2 * sizeof(double), hipMemcpyDeviceToHost, be.stream));
    } else if (!be.scal_direct() || !fail_map) MVUS_HIP(hipMemcpyAsync(fail_host, fail, 2 * sizeof(int), hipMemcpyDeviceToHost, be.stream));
  }
};

// Gauss-Newton normal equations of the Jacobian currently held, copied out for inspection (mvus_ba_normal_equations)
template <class BE>
int schur_export(BE& be, HipSchur<BE>& sc, double* g, double* JtJ_cam, double* band, double* cross, int32_t* W_out) {
  if (!be.has_jacobian) { be.err = "no Jacobian held: call mvus_ba_residual_jacobian first"; return MVUS_E_INVALID; }
  if (sc.shard) { be.err = "normal_equations: not available on a time shard (every rank holds a slice of the spline blocks)"; return MVUS_E_INVALID; }
  if (W_out) *W_out = sc.ne.W;
  if (!g && !JtJ_cam && !band && !cross) return MVUS_OK;
  sc.assemble_held(be);
  sc.flush_diag();
  const NEView& ne = sc.ne;
  if (g) be.download(g, sc.gx, be.hp.n);
  if (JtJ_cam) be.download(JtJ_cam, ne.A, (int64_t)ne.C * ne.B * ne.B);
  if (band) be.download(band, ne.Cb, (int64_t)ne.N * ne.W * 9);
  if (cross) {
    std::vector<double> Ec((size_t)ne.N3 * ne.CB);
    be.download(Ec.data(), ne.Et, (int64_t)Ec.size());
    for (int c = 0; c < ne.C; ++c)
      for (int r = 0; r < ne.N3; ++r)
        for (int k = 0; k < ne.B; ++k) cross[((size_t)c * ne.B + k) * ne.N3 + r] = Ec[((size_t)c * ne.N3 + r) * ne.B + k];
  }
  return MVUS_OK;
}

}  // namespace mvus
