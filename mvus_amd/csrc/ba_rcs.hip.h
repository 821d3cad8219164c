// Reduced camera system S p_c = -rhs (dense, SPD, nn = C * (3 + P) <= 1152 unknowns): blocked L D L^T with the matrix held in the
// register layout of the fp64 matrix cores.  Included by ba_schur_hip.hip.h (needs NEView, bcr_d4, ldl_inv16, dpp_shift).
//
// No counterpart in the reference (Scene.BA hands the whole problem to scipy, reconstruction/common.py:670); this replaces the
// block Gauss-Jordan of rounds 1-4 (k_gj_step: one launch per 32 columns, 10.1 us each, of which the 32x32 pivot inverse in ONE
// wavefront was 5.8 us and the rest a kernel boundary plus a round trip of the pivot tile through memory).
//
// Storage.  The matrix is cut into 16x16 blocks.  A block M is kept as its IMAGE  img(M)[lane][r] = M[lk + 4r][lr]  (lane = 16 lk +
// lr, r = 0..3: four doubles = 32 contiguous bytes per lane, 2 KB per block) -- the accumulator layout of v_mfma_f64_16x16x4.  The same
// four values are at once the B operand of M (step s reads M[4s + lk][lr] = img(M)[s]) and the A operand of M^T (step s reads
// M^T[lr][4s + lk] = img(M)[s]): a product  D = P^T Q  of two blocks needs img(P) and img(Q) only -- no transposition, no LDS shuffle
// between the result of one product and the operand of the next.  Block (i, j), i >= j, lives at rcs_blk(i, j); block row nbk is the
// right-hand side (a block row whose first row is rhs^T).
//
// Algebra (right-looking, 16 columns per step).  U_ij = A_ij^T is what is stored.  Step k:  A_kk = L_k Delta_k L_k^T by ldl_inv16 (one
// wavefront, row per lane, multipliers by DPP) which also returns X_k = L_k^-1;  panel  T_ik = X_k U_ik  (i > k);  trailing update
// U_ij -= T_jk^T Delta_k^-1 T_ik  (i >= j > k).  The right-hand side rides as block row R = nbk; afterwards column 0 of T_Rk is
// t_k = (L^-1 rhs)_k and  x_k = X_k^T Delta_k^-1 (t_k - sum_{i>k} T_ik x_i)  descending in k.  The negated, scaled image -Delta_k^-1 T_ik
// is kept beside the raw one so that every update is a plain multiply-accumulate.
//
// Launches.  Super-panels of kRcsSP = 9 block columns (144 unknowns):  k_rcs_factor (ONE workgroup: the diagonal super-block and
// the right-hand-side row live in the registers of ten wavefronts (one block row each), an eleventh runs the 16-column pivot chain one step ahead of the
// trailing update: per 16 columns the chain is ldl_inv16 + one panel product + one block update, all inside one CU)  ->  k_rcs_trsm
// (block rows below: one wavefront per row, operands staged in LDS)  ->  k_rcs_syrk (trailing blocks, one wavefront each)  ->  next
// super-panel ...  ->  k_rcs_backsub (one workgroup, descending).  nn = 288: 6 launches instead of 1 + 9.
#pragma once

namespace mvus {

constexpr int kRcsSP = 9;                   // 16-column blocks per super-panel
constexpr int kRcsFactorThreads = 64 * (kRcsSP + 3);      // ten row-owning wavefronts (nine block rows + the right-hand side), the pivot wavefront, the publisher
constexpr int kRcsTrsmWaves = 4;            // wavefronts per block row (= workgroup) of k_rcs_trsm

struct RcsView {
  double* Simg;      // block images: U_ij, overwritten by T_ik (i > k) and by img(X_k^T) on the diagonal
  double* Tsc;       // -Delta_k^-1 T_ik images; diagonal block (k, k): 1/d of the 16 pivots in its first 16 doubles
  double* x;         // the solution, 16 nbk doubles (padding rows: 0)
  int nn, nbk;       // unknowns, 16-blocks (the right-hand side is block row nbk)
};

__host__ __device__ __forceinline__ long long rcs_blk(int i, int j) { return ((long long)i * (i + 1) / 2 + j) * 256; }
__host__ __device__ inline size_t rcs_doubles(int nn) { const int nbk = (nn + 15) / 16; return (size_t)rcs_blk(nbk + 1, 0); }
__device__ __forceinline__ bcr_d4 rcs_load(const double* __restrict__ p) { return *reinterpret_cast<const bcr_d4*>(p); }
__device__ __forceinline__ void rcs_store(double* __restrict__ p, bcr_d4 v) { *reinterpret_cast<bcr_d4*>(p) = v; }
// acc + P^T Q from img(P), img(Q)
__device__ __forceinline__ bcr_d4 rcs_mma(bcr_d4 p, bcr_d4 q, bcr_d4 acc) {
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(p[s], q[s], acc, 0, 0, 0);
  return acc;
}
// y = M w from img(M) and w[lr] (the value of this lane's column): the sums over the 16 lanes of a row by DPP row shifts;
// lane lr == 15 of row lk ends with y[lk + 4r] in out[r]
__device__ __forceinline__ bcr_d4 rcs_matvec(bcr_d4 img, double w) {
  bcr_d4 out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double v = img[r] * w;
    v += dpp_shift<0x111, 0xF>(v, 0.0);      // row_shr:1
    v += dpp_shift<0x112, 0xF>(v, 0.0);      // row_shr:2
    v += dpp_shift<0x114, 0xF>(v, 0.0);      // row_shr:4
    v += dpp_shift<0x118, 0xF>(v, 0.0);      // row_shr:8
    out[r] = v;
  }
  return out;
}

// L D L^T of a 16x16 block with the inverse factor, as ldl_inv16 (ba_schur_hip.hip.h: row i of the block and of the identity in lane i
// of every 16-lane row, multipliers by DPP row broadcast; same operations on every entry in the same order), with the instruction
// order fixed BY HAND.  The unit of work is one (column K, target column J) pair: broadcast of the multiplier, the update of a[J] and
// of x[J]; a scheduling barrier closes each.  Column K first updates column K + 1; the six dependent steps that turn the next pivot
// into its reciprocal (broadcast, v_rcp_f64, two Newton steps) are then dealt one per following pair, so that each waits under the
// ~30 issue cycles of a pair instead of stalling the wavefront.  Left to itself the compiler hoists the broadcasts of several
// columns, postpones the x updates, and spills the multipliers it keeps for them (256 registers and 300 - 700 bytes of scratch per
// lane inside the chain).
struct RcsPivotChain { double d, r, e; bool bad; };
// lane J's value to every lane of the 16-lane row, as two v_mov_b32_dpp with an UNDEFINED old value (llvm.amdgcn.mov.dpp): the 64-bit
// form the generic builtin produces (v_mov_b64_dpp, tied to its old operand) costs a register copy and two wait states per broadcast
template <int J> __device__ __forceinline__ double rcs_bcast(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + J, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + J, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int STAGE, int J> __device__ __forceinline__ void rcs_chain_stage(RcsPivotChain& c, const double (&a)[16]) {
  if constexpr (STAGE == 0) { double d = rcs_bcast<J>(a[J]); c.bad |= !(d > 0.0); c.d = d > 0.0 ? d : 1.0; }
  else if constexpr (STAGE == 1) c.r = __builtin_amdgcn_rcp(c.d);
  else if constexpr (STAGE == 2 || STAGE == 4) c.e = 2.0 - c.d * c.r;
  else if constexpr (STAGE == 3 || STAGE == 5) c.r = c.r * c.e;
}
template <int FROM, int KN> struct RcsChainRest {      // the stages that found no pair to hide under (the last columns)
  static __device__ __forceinline__ void run(RcsPivotChain& c, const double (&a)[16]) {
    if constexpr (FROM <= 5) { rcs_chain_stage<FROM, KN>(c, a); RcsChainRest<FROM + 1, KN>::run(c, a); }
  }
};
template <int K, int J> struct RcsColOps {
  // m: the multiplier of THIS pair, broadcast while the previous pair's FMAs were issued (a dependent VALU instruction waits for
  // its operand: with the two DPP moves directly in front of the FMAs that use them every pair cost ~32 cycles instead of ~16)
  static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16], double ta, double tx, double m, RcsPivotChain& c) {
    if constexpr (J < 16) {
      double mn = 0.0;
      if constexpr (J + 1 < 16) mn = rcs_bcast<J + 1>(a[K]);
      a[J] -= ta * m;
      x[J] -= tx * m;
      rcs_chain_stage<J - (K + 1), K + 1>(c, a);
      // (scheduling barriers order machine instructions only; the empty asm makes the pair's results exist HERE in the instruction
      // selector's order too -- without it the x updates sink to the end of the block and their multipliers go to scratch)
      asm volatile("" : "+v"(a[J]), "+v"(x[J]), "+v"(c.d), "+v"(c.r), "+v"(c.e), "+v"(mn));
      __builtin_amdgcn_sched_barrier(0);
      RcsColOps<K, J + 1>::run(a, x, ta, tx, mn, c);
    }
  }
};
template <int K> struct RcsCols {
  static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16], double (&rd)[16], RcsPivotChain& c) {
    const double r = c.r;
    rd[K] = r;
    if constexpr (K + 1 < 16) {
      const double m0 = rcs_bcast<K + 1>(a[K]);
      const double ta = a[K] * r, tx = x[K] * r;
      RcsColOps<K, K + 1>::run(a, x, ta, tx, m0, c);
      RcsChainRest<(15 - K < 6 ? 15 - K : 6), K + 1>::run(c, a);
      __builtin_amdgcn_sched_barrier(0);
      RcsCols<K + 1>::run(a, x, rd, c);
    }
  }
};
__device__ __forceinline__ void rcs_ldl16(double (&a)[16], double (&x)[16], double (&rd)[16], int lane, int* __restrict__ fail) {
  RcsPivotChain c{0.0, 0.0, 0.0, false};
  RcsChainRest<0, 0>::run(c, a);
  RcsCols<0>::run(a, x, rd, c);
  if (c.bad && lane == 0) fail[0] = 2;
}

// S = (A + lambda D_c) - sum_slabs Gp[:, :CB] and rhs = gc - sum_slabs Gp[:, CB] (k_schur_finish's sums), written as block images:
// lower block triangle, identity on the padding of the last block, the right-hand side as block row nbk.
// Grid (ntile, ntile + 1) of 32x32 tiles, the last row of the grid writes the right-hand side.
__global__ __launch_bounds__(256) void k_rcs_finish(NEView ne, int ncols, int nslab, double lambda, const double* __restrict__ Gp, RcsView rv,
                                                    unsigned* __restrict__ flags) {
  constexpr int kT = 32, kRowsPer = 4, kRowStep = 8;
  const int r0 = threadIdx.x / kT, c = threadIdx.x % kT;
  const int b = blockIdx.x * kT + c;
  const long long stride = (long long)ne.CB * ncols;
  const int nn = ne.CB, npad = rv.nbk * 16;
  if (blockIdx.y == gridDim.y - 1) {
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[0] = 0u;          // the hand-over counter of the factor launches that follow
    if (r0 != 0 || b >= npad) return;
    double gr = 0.0;
    if (b < nn) {
      for (int sl = 0; sl < nslab; sl += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = sl + u < nslab ? Gp[(sl + u) * stride + (long long)b * ncols + ne.CB] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) gr += t[u];
      }
    }
    const double val = b < nn ? ne.gc[b] - gr : 0.0;
    double* blk = rv.Simg + rcs_blk(rv.nbk, b / 16);
    const int cc = b % 16;
    for (int lr = 0; lr < 16; ++lr) blk[((cc % 4) * 16 + lr) * 4 + cc / 4] = lr == 0 ? val : 0.0;
    return;
  }
  if (blockIdx.x > blockIdx.y) return;
  double gsum[kRowsPer];
  long long goff[kRowsPer];
#pragma unroll
  for (int q = 0; q < kRowsPer; ++q) {
    const int a = blockIdx.y * kT + r0 + kRowStep * q;
    const int hi = a / kGemmT >= b / kGemmT ? a : b, lo = a / kGemmT >= b / kGemmT ? b : a;
    goff[q] = (a < nn && b < nn) ? (long long)hi * ncols + lo : -1;
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
    const int a = blockIdx.y * kT + r0 + kRowStep * q;
    if (a >= npad || b >= npad || a / 16 < b / 16) continue;
    double v = a == b ? 1.0 : 0.0;
    if (a < nn && b < nn) {
      v = -gsum[q];
      if (a / ne.B == b / ne.B) {
        const int cam = a / ne.B;
        double h = ne.A[((long long)cam * ne.B + a % ne.B) * ne.B + b % ne.B];
        if (a == b) h += lambda * (h > 0.0 ? h : 1.0);
        v += h;
      }
    }
    const int cc = b % 16;
    rv.Simg[rcs_blk(a / 16, b / 16) + ((cc % 4) * 16 + a % 16) * 4 + cc / 4] = v;
  }
}

// ---- hand-over of a finished panel step to the workgroups that solve the block rows BELOW the diagonal super-block, inside the
// launch (k_rcs_factor, blockIdx.x > 0).  The recipe of the CDNA programming guide (Guideline 16, R1): the producer stores the payload
// WRITE-THROUGH (sc1 -- other XCDs' L2s and the readers' L1s are not coherent with ours), waits for its stores (s_waitcnt vmcnt(0)),
// then one lane stores the step number with a relaxed agent-scope atomic; a consumer polls that one word (relaxed, s_sleep between
// polls, BOUNDED: a time-out raises fail[0] = kFailHandover -- a code of its own, NOT the 'not positive definite' of the pivots: the host
// then repeats the solve with the rows in a launch of their own instead of raising the damping) and reads the payload with sc1 loads.
using rcs_u4 = __attribute__((ext_vector_type(4))) unsigned;
struct RcsWide { rcs_u4 a, b; };
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rcs_rsrc(double* p, size_t doubles) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(doubles * sizeof(double)), 0x00020000);
}
__device__ __forceinline__ void rcs_store_sc1(__amdgpu_buffer_rsrc_t r, long long dbl_off, bcr_d4 v) {
  RcsWide w;
  __builtin_memcpy(&w, &v, sizeof(w));
  __builtin_amdgcn_raw_buffer_store_b128(w.a, r, (int)(dbl_off * 8), 0, 16);
  __builtin_amdgcn_raw_buffer_store_b128(w.b, r, (int)(dbl_off * 8 + 16), 0, 16);
}
__device__ __forceinline__ bcr_d4 rcs_load_sc1(__amdgpu_buffer_rsrc_t r, long long dbl_off) {
  RcsWide w;
  w.a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(dbl_off * 8), 0, 16);
  w.b = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(dbl_off * 8 + 16), 0, 16);
  bcr_d4 v;
  __builtin_memcpy(&v, &w, sizeof(v));
  return v;
}
__device__ __forceinline__ double rcs_load1_sc1(__amdgpu_buffer_rsrc_t r, long long dbl_off) {
  using u2 = __attribute__((ext_vector_type(2))) unsigned;
  const u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(dbl_off * 8), 0, 16);
  double v;
  __builtin_memcpy(&v, &w, sizeof(v));
  return v;
}
__device__ __forceinline__ void rcs_store1_sc1(__amdgpu_buffer_rsrc_t r, long long dbl_off, double v) {
  using u2 = __attribute__((ext_vector_type(2))) unsigned;
  u2 w;
  __builtin_memcpy(&w, &v, sizeof(w));
  __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)(dbl_off * 8), 0, 16);
}
constexpr unsigned kRcsSpinLimit = 1u << 21;                         // polls (each >= ~100 ns): far beyond any legitimate wait
constexpr int kFailHandover = kFailHandoverCode;                     // fail[0]: a consumer workgroup gave up waiting for the factor workgroup's flag

// x_i[c] = sum_j X[j][c] w[j] for lane c of a 16-lane row (col: the lane's column of X, 16 contiguous doubles in LDS; w: this lane's
// entry of w, handed round by DPP row broadcasts)
template <int J> struct RcsXtTimes {
  static __device__ __forceinline__ double run(const double* __restrict__ col, double w, double acc) {
    if constexpr (J < 16) return RcsXtTimes<J + 1>::run(col, w, acc + col[J] * rcs_bcast<J>(w));
    else return acc;
  }
};
__device__ __forceinline__ double rcs_xt_times(const double* __restrict__ col, double w) { return RcsXtTimes<0>::run(col, w, 0.0); }
// part = T_ik x_i for the block k of a row's register-resident blocks (k is wavefront-uniform but not a compile-time constant: the
// nine-way selection is a scalar branch)
__device__ __forceinline__ void rcs_part_product(const bcr_d4 (&sl)[kRcsSP], int k, const double* __restrict__ xi, double* __restrict__ out, int lr, int lk) {
  const double xv = xi[lr];
#pragma unroll
  for (int kk = 0; kk < kRcsSP; ++kk) {
    if (kk != k) continue;
    const bcr_d4 y = rcs_matvec(sl[kk], xv);
    if (lr == 15) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[lk + 4 * r] = y[r];
    }
  }
}

// the pivot wavefront of k_rcs_factor.  Nothing but LDS traffic inside the chain: img(X_k^T) and 1/d_k of every step stay in LDS
// (Xb, rdb: one copy per step) and go to global memory after the last step -- a vector-memory store inside the loop costs every later
// s_waitcnt vmcnt a store acknowledgement (stores and loads share the counter on this part).
__device__ __forceinline__ void rcs_pivot_role(int nc, int* __restrict__ fail, double* __restrict__ Dm, double* __restrict__ Xb, double* __restrict__ rdb) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_s_setprio(3);                                   // the chain's instructions go first on this SIMD
  for (int k = 0; k < nc; ++k) {
    if (k == 0) lds_barrier();                                     // B0: block (0, 0) staged
    double a[16], x[16], rd[16];
    // (the row index is made opaque per step: otherwise the sixteen identity values and the row's LDS addresses are hoisted out of the
    // loop and live -- or spilled -- across the whole chain.  The entries of a above the diagonal are never read by the lower ones.)
    int lr = lane & 15;
    asm volatile("" : "+v"(lr));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      a[j] = Dm[lr * 17 + j];
      x[j] = j == lr ? 1.0 : 0.0;
    }
    rcs_ldl16(a, x, rd, lane, fail);
    // lane c holds column c of X = L^-1 (x[j] = X[j][c]) and all sixteen 1/d: Xb[k] is X column-major (Xb[c * 16 + j] = X[j][c]) --
    // four 32-byte stores per lane of the first row instead of sixteen scattered 8-byte ones
    if (lane < 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) rcs_store(Xb + k * 256 + lane * 16 + 4 * q, bcr_d4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]});
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) rcs_store(rdb + k * 16 + 4 * q, bcr_d4{rd[4 * q], rd[4 * q + 1], rd[4 * q + 2], rd[4 * q + 3]});
    }
    lds_barrier();                                                 // b1: X_k, 1/d_k
    lds_barrier();                                                 // b2: panel k and block (k + 1, k + 1)
  }
}

// One super-panel: block columns c0 .. c0 + nc - 1 of the diagonal super-block and of the right-hand-side row.
// Wavefront w < 10 owns the local block row w (row nc = the right-hand side) -- its <= 9 blocks stay in registers from the load to the
// last update; wavefront 10 factorises the diagonal blocks (the pivot chain) and is handed block (k + 1, k + 1) as soon as its owner has
// applied panel k to it: the other updates of panel k run beside the chain.  Eleven wavefronts = three per SIMD, <= 168 registers each
// (with two row blocks per wavefront and 256 registers the compiler's scheduler let the pivot chain's pressure grow until it spilled).
// block rows below the diagonal super-block, solved INSIDE the factor launch (blockIdx.x >= 1: one block row per workgroup, its first
// four wavefronts): the algorithm of k_rcs_trsm, but each step waits for the factor workgroup's flag and reads that step's X_k, 1/d_k
// and T_jk with sc1 loads.  The rows' chain is shorter than the pivot chain, so these workgroups trail the factorisation by about one
// step and finish ~1 step after it: the 18 us launch of k_rcs_trsm (and its kernel boundary) disappear from the critical path.
__device__ __forceinline__ void rcs_trsm_role(RcsView rv, int c0, int nc, unsigned* __restrict__ flags, int* __restrict__ fail, double* __restrict__ xch,
                                              int* __restrict__ abort_s, unsigned spin_limit) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lk = lane >> 4;
  const int c1 = c0 + nc, gi = c1 + (int)blockIdx.x - 1;
  const size_t tot = rcs_doubles(rv.nn);
  const __amdgpu_buffer_rsrc_t rS = rcs_rsrc(rv.Simg, tot), rT = rcs_rsrc(rv.Tsc, tot);
  constexpr int kSlots = (kRcsSP + kRcsTrsmWaves - 1) / kRcsTrsmWaves;
  bcr_d4 s[kSlots];
  const bcr_d4 zero{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < kSlots; ++u) {
    const int jl = wave + kRcsTrsmWaves * u;
    s[u] = jl < nc ? rcs_load(rv.Simg + rcs_blk(gi, c0 + jl) + lane * 4) : zero;      // (written by earlier launches: plain loads)
  }
  if (threadIdx.x == 0) *abort_s = 0;
  lds_barrier();
#pragma unroll
  for (int k = 0; k < kRcsSP; ++k) {
    if (k >= nc) continue;
    if (threadIdx.x == 0 && *abort_s == 0) {                         // one lane polls one word
      const unsigned want = (unsigned)(c0 * 16 + k + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > spin_limit) { *abort_s = 1; fail[0] = kFailHandover; fail[2] = 1; break; }     // (fail[2]: sticky for this solve -- a later kernel that meets the unfinished rows may overwrite fail[0])
      }
    }
    lds_barrier();
    const bool dead = *abort_s != 0;
    bcr_d4 tj[kSlots];
    bool up[kSlots];
#pragma unroll
    for (int u = 0; u < kSlots; ++u) {
      const int jl = wave + kRcsTrsmWaves * u;
      up[u] = jl > k && jl < nc && !dead;
      tj[u] = up[u] ? rcs_load_sc1(rS, rcs_blk(c0 + jl, c0 + k) + lane * 4) : zero;
    }
    if (wave == k % kRcsTrsmWaves && !dead) {
      const bcr_d4 xi = rcs_load_sc1(rS, rcs_blk(c0 + k, c0 + k) + lane * 4);
      bcr_d4 nrd;
#pragma unroll
      for (int r = 0; r < 4; ++r) nrd[r] = -rcs_load1_sc1(rT, rcs_blk(c0 + k, c0 + k) + lk + 4 * r);
      const bcr_d4 t = rcs_mma(xi, s[k / kRcsTrsmWaves], zero);
      s[k / kRcsTrsmWaves] = t;
      rcs_store(xch + (k & 1) * 256 + lane * 4, t * nrd);
    }
    lds_barrier();
    if (k + 1 >= nc || dead) continue;
    const bcr_d4 sc = rcs_load(xch + (k & 1) * 256 + lane * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < kSlots; ++u)
        if (up[u]) s[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(tj[u][q], sc[q], s[u], 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < kSlots; ++u) {                                 // the row's T blocks and their scaled images, for k_rcs_syrk / the substitution
    const int jl = wave + kRcsTrsmWaves * u;
    if (jl >= nc) continue;
    bcr_d4 sc;
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[r] = -rcs_load1_sc1(rT, rcs_blk(c0 + jl, c0 + jl) + lk + 4 * r) * s[u][r];
    rcs_store(rv.Simg + rcs_blk(gi, c0 + jl) + lane * 4, s[u]);
    rcs_store(rv.Tsc + rcs_blk(gi, c0 + jl) + lane * 4, sc);
  }
}

// the twelfth wavefront of the factor workgroup: after the barrier that completes panel k it copies X_k, 1/d_k and the panel's T_jk
// from LDS to global memory with write-through stores; the step number is published one step LATER, when those stores are certainly
// acknowledged (s_waitcnt vmcnt(0) finds nothing to wait for) -- the wavefront must never make the others wait at a barrier
__device__ __forceinline__ void rcs_publisher_role(RcsView rv, int c0, int nc, unsigned* __restrict__ flags, const double* __restrict__ Xb,
                                                   const double* __restrict__ rdb, const double* __restrict__ panel) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const size_t tot = rcs_doubles(rv.nn);
  const __amdgpu_buffer_rsrc_t rS = rcs_rsrc(rv.Simg, tot), rT = rcs_rsrc(rv.Tsc, tot);
  lds_barrier();                                                     // B0
  for (int k = 0; k < nc; ++k) {
    lds_barrier();                                                   // b1
    if (k > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(flags, (unsigned)(c0 * 16 + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // steps 0 .. k - 1
    }
    lds_barrier();                                                   // b2: panel k complete
    bcr_d4 xi;
#pragma unroll
    for (int r = 0; r < 4; ++r) xi[r] = Xb[k * 256 + (lk + 4 * r) * 16 + lr];
    rcs_store_sc1(rS, rcs_blk(c0 + k, c0 + k) + lane * 4, xi);
    if (lane < 16) rcs_store1_sc1(rT, rcs_blk(c0 + k, c0 + k) + lane, rdb[k * 16 + lane]);
    for (int jl = k + 1; jl < nc; ++jl) rcs_store_sc1(rS, rcs_blk(c0 + jl, c0 + k) + lane * 4, rcs_load(panel + jl * 256 + lane * 4));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the LDS reads are done before the panel is overwritten: next b1)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(flags, (unsigned)(c0 * 16 + nc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kRcsFactorThreads) void k_rcs_factor(RcsView rv, int c0, int* __restrict__ fail, int last, double* __restrict__ pc,
                                                                  unsigned* __restrict__ flags, unsigned spin_limit) {
  __shared__ double Dm[16 * 17];
  __shared__ __attribute__((aligned(32))) double Xb[kRcsSP * 256];
  __shared__ __attribute__((aligned(32))) double rdb[kRcsSP * 16];
  __shared__ __attribute__((aligned(32))) double panel[(kRcsSP + 1) * 256];
  __shared__ int abort_s;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 15, lk = lane >> 4;
  const int nc = min(kRcsSP, rv.nbk - c0), R = rv.nbk;
  if (blockIdx.x > 0) {                                              // a block row below the super-block
    if (wave < kRcsTrsmWaves) rcs_trsm_role(rv, c0, nc, flags, fail, panel, &abort_s, spin_limit);
    return;
  }
  if (wave == kRcsSP + 2) {                                          // the publisher (it takes part in every barrier of the chain)
    if (gridDim.x > 1) rcs_publisher_role(rv, c0, nc, flags, Xb, rdb, panel);
    else { lds_barrier(); for (int k = 0; k < nc; ++k) { lds_barrier(); lds_barrier(); } }
    if (last) { lds_barrier(); for (int i = 0; i < nc; ++i) { lds_barrier(); lds_barrier(); } }
    return;
  }
  __shared__ double u0[kRcsSP * 16];                                 // t_k = column 0 of T_R,k
  __shared__ double part[kRcsSP * kRcsSP * 16];                      // part[i][k] = T_ik x_i
  __shared__ double xs[kRcsSP * 16];
  if (wave == kRcsSP + 1) {
    rcs_pivot_role(nc, fail, Dm, Xb, rdb);
    // the factors of the diagonal blocks, for the kernels that follow
    if (!last || c0 > 0)
      for (int k = 0; k < nc; ++k) {
        bcr_d4 xi;
#pragma unroll
        for (int r = 0; r < 4; ++r) xi[r] = Xb[k * 256 + (lk + 4 * r) * 16 + lr];
        rcs_store(rv.Simg + rcs_blk(c0 + k, c0 + k) + lane * 4, xi);
        if (lane < 16) rv.Tsc[rcs_blk(c0 + k, c0 + k) + lane] = rdb[k * 16 + lane];
      }
    if (!last) return;
    __builtin_amdgcn_s_setprio(0);
    // ---- back substitution of this (the last) super-panel, pivot wavefront's part: x_i = X_i^T Delta_i^-1 (t_i - sum_{i' > i} T_i'i x_i')
    lds_barrier();                                                   // S0: t in u0
    for (int i = nc - 1; i >= 0; --i) {
      double xi = 0.0;
      if (lane < 16) {
        double w = u0[i * 16 + lane];
        for (int ip = i + 1; ip < nc; ++ip) w -= part[(ip * kRcsSP + i) * 16 + lane];      // fixed order: the same bits every run
        w *= rdb[i * 16 + lane];
        xi = rcs_xt_times(Xb + i * 256 + lane * 16, w);
        xs[i * 16 + lane] = xi;
      }
      lds_barrier();                                                 // A_i: x_i
      lds_barrier();                                                 // B_i: part[i][i - 1] (and the deferred products of the rows above)
    }
    if (lane < 16)
      for (int k = 0; k < nc; ++k) {
        const int a = (c0 + k) * 16 + lane;
        const double v = xs[k * 16 + lane];
        rv.x[a] = v;
        if (a < rv.nn) pc[a] = -v;
      }
    return;
  }
  const int il = wave;                                               // local block row; il == nc: the right-hand side
  const bool has = il <= nc, isR = il == nc;
  const int gi = isR ? R : c0 + il;
  bcr_d4 sl[kRcsSP];
  const bcr_d4 zero{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int jl = 0; jl < kRcsSP; ++jl) sl[jl] = (has && jl < nc && (isR || jl <= il)) ? rcs_load(rv.Simg + rcs_blk(gi, c0 + jl) + lane * 4) : zero;
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Dm[(lk + 4 * r) * 17 + lr] = sl[0][r];
  }
  lds_barrier();                                                     // B0
#pragma unroll
  for (int k = 0; k < kRcsSP; ++k) {
    if (k < nc) {
      lds_barrier();                                                 // b1
      const bool act = has && il > k;                                // (the right-hand side: il = nc > k)
      const bool owner = act && il == k + 1 && !isR && k + 1 < nc;   // this row's diagonal block is the next pivot block
      if (owner) __builtin_amdgcn_s_setprio(2);                      // its two products are the chain: ahead of the other rows' on this SIMD
      bcr_d4 sc = zero;
      if (act) {
        bcr_d4 xi;                                                   // img(X_k^T)[lane][s] = X_k[lr][lk + 4s]; Xb holds X column-major
#pragma unroll
        for (int r = 0; r < 4; ++r) xi[r] = Xb[k * 256 + (lk + 4 * r) * 16 + lr];
        const bcr_d4 t = rcs_mma(xi, sl[k], zero);
        sl[k] = t;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[r] = -rdb[k * 16 + lk + 4 * r] * t[r];
        rcs_store(panel + il * 256 + lane * 4, t);
      }
      // the row whose diagonal block is the next pivot block applies panel k to it at once (its own T_k+1,k is in registers) and
      // stages it for the pivot wavefront: ONE barrier then publishes both the panel and the next pivot block
      const int kn = k + 1 < kRcsSP ? k + 1 : 0;                     // (k + 1 < nc <= kRcsSP when used; the clamp keeps the unrolled index static)
      if (owner) {
        sl[kn] = rcs_mma(sl[k], sc, sl[kn]);
#pragma unroll
        for (int r = 0; r < 4; ++r) Dm[(lk + 4 * r) * 17 + lr] = sl[kn][r];
        __builtin_amdgcn_s_setprio(0);
      }
      lds_barrier();                                                 // b2: panel k, block (k + 1, k + 1)
      {                                                              // the row's remaining blocks: K-slices outer, blocks inner
        bcr_d4 tj[kRcsSP];
        bool up[kRcsSP];
#pragma unroll
        for (int jl = k + 1; jl < kRcsSP; ++jl) {
          up[jl] = jl < nc && act && (isR || jl <= il) && !(owner && jl == k + 1);
          tj[jl] = up[jl] ? rcs_load(panel + jl * 256 + lane * 4) : zero;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int jl = k + 1; jl < kRcsSP; ++jl)
            if (up[jl]) sl[jl] = __builtin_amdgcn_mfma_f64_16x16x4f64(tj[jl][q], sc[q], sl[jl], 0, 0, 0);
      }
    }
  }
  // the row's panel blocks T_il,k (k < il) and their scaled images: stored once, after the chain (a last super-panel that is also the
  // first has no reader: the substitution below works from the registers)
  if (has && (!last || c0 > 0)) {
#pragma unroll
  for (int k = 0; k < kRcsSP; ++k) {
    if (k < nc && il > k) {
      bcr_d4 sc;
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[r] = -rdb[k * 16 + lk + 4 * r] * sl[k][r];
      rcs_store(rv.Simg + rcs_blk(gi, c0 + k) + lane * 4, sl[k]);
      rcs_store(rv.Tsc + rcs_blk(gi, c0 + k) + lane * 4, sc);
    }
  }
  }
  if (!last) return;
  // ---- back substitution, the row-owning wavefronts' part.  Row i hands T_ik x_i (k < i) to the pivot wavefront through part[i][k]:
  // the product for k = i - 1 -- the one the next x needs -- between the two barriers of its own step, the others one per barrier
  // interval afterwards (k = 2 i' - i in the first interval of step i' < i, k = 2 i' - i - 1 in the second: always a full step before
  // x_k is formed), so that no interval holds more than one product per wavefront and a step of the chain is two short intervals.
  if (has && isR && lr == 0) {
#pragma unroll
    for (int k = 0; k < kRcsSP; ++k)
      if (k < nc)
#pragma unroll
        for (int r = 0; r < 4; ++r) u0[k * 16 + lk + 4 * r] = sl[k][r];
  }
  lds_barrier();                                                     // S0
  for (int i = nc - 1; i >= 0; --i) {
    const bool mine = has && !isR && il > i;                         // a row above the current one: deferred products
    int kd = mine ? 2 * i - il : -1;
    if (kd >= 0) rcs_part_product(sl, kd, xs + il * 16, part + (il * kRcsSP + kd) * 16, lr, lk);
    lds_barrier();                                                   // A_i
    kd = (has && !isR && il == i) ? i - 1 : (mine ? 2 * i - il - 1 : -1);
    if (kd >= 0) rcs_part_product(sl, kd, xs + il * 16, part + (il * kRcsSP + kd) * 16, lr, lk);
    lds_barrier();                                                   // B_i
  }
}

// LDS staging of one factorised diagonal super-block for the kernels that apply it: img(X_k^T) (nc blocks), 1/d (nc x 16), the
// raw images T_jk, j > k (nc (nc - 1) / 2 blocks at tri(j, k)).  Every wavefront copies whole blocks, 32 bytes per lane.
__host__ __device__ inline int rcs_stage_doubles(int nc) { return nc * 256 + nc * 16 + nc * (nc - 1) / 2 * 256; }
__device__ __forceinline__ int rcs_tri(int j, int k) { return j * (j - 1) / 2 + k; }
// (all loads of a wavefront are issued before the first LDS store: one memory round trip, not one per block)
template <int NWAVES>
__device__ __forceinline__ void rcs_stage(const RcsView& rv, int c0, int nc, double* __restrict__ st, int lane, int wave) {
  constexpr int kMaxBlk = kRcsSP + kRcsSP * (kRcsSP - 1) / 2, kPer = (kMaxBlk + NWAVES - 1) / NWAVES;
  double* rds = st + nc * 256;
  double* ts = rds + nc * 16;
  const int nblk = nc + nc * (nc - 1) / 2;
  bcr_d4 v[kPer];
  double rdv[kPer];
  int dst[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int t = wave + NWAVES * u;
    dst[u] = -1; rdv[u] = 0.0;
    if (t < nc) {
      v[u] = rcs_load(rv.Simg + rcs_blk(c0 + t, c0 + t) + lane * 4);
      if (lane < 16) rdv[u] = rv.Tsc[rcs_blk(c0 + t, c0 + t) + lane];
      dst[u] = t * 256;
    } else if (t < nblk) {
      const int e = t - nc;
      int j = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);
      while (j * (j - 1) / 2 > e) --j;
      while (j * (j + 1) / 2 <= e) ++j;
      const int k = e - j * (j - 1) / 2;
      v[u] = rcs_load(rv.Simg + rcs_blk(c0 + j, c0 + k) + lane * 4);
      dst[u] = (int)(ts - st) + e * 256;
    }
  }
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int t = wave + NWAVES * u;
    if (dst[u] >= 0) rcs_store(st + dst[u] + lane * 4, v[u]);
    if (t < nc && lane < 16) rds[t * 16 + lane] = rdv[u];
  }
}

// block rows below the diagonal super-block: T_ik = X_k (U_ik - sum_{k' < k} T_kk'^T Delta_k'^-1 T_ik'), k ascending.  ONE block row per
// workgroup, four wavefronts = four SIMDs: wavefront w holds the row's blocks of the columns jl = w, w + 4, w + 8.  Step k: the
// wavefront that holds column k forms T_ik and hands its scaled image to the others through LDS (two buffers, one barrier per step),
// every wavefront then updates the blocks it holds.  (v_mfma_f64_16x16x4 issues in 64 cycles whether or not it accumulates into the
// previous result -- profiles/r05_fp64_issue_rates.txt -- so the order of the matrix-core instructions does not matter.)
__global__ __launch_bounds__(64 * kRcsTrsmWaves) void k_rcs_trsm(RcsView rv, int c0) {
  extern __shared__ __attribute__((aligned(32))) double rcs_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lk = lane >> 4;
  const int nc = min(kRcsSP, rv.nbk - c0), c1 = c0 + nc;
  rcs_stage<kRcsTrsmWaves>(rv, c0, nc, rcs_lds, lane, wave);
  const double* rds = rcs_lds + nc * 256;
  const double* ts = rds + nc * 16;
  double* xch = rcs_lds + rcs_stage_doubles(nc);                     // 2 x 256 doubles
  const int gi = c1 + blockIdx.x;
  constexpr int kSlots = (kRcsSP + kRcsTrsmWaves - 1) / kRcsTrsmWaves;
  bcr_d4 s[kSlots];
  const bcr_d4 zero{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < kSlots; ++u) {
    const int jl = wave + kRcsTrsmWaves * u;
    s[u] = jl < nc ? rcs_load(rv.Simg + rcs_blk(gi, c0 + jl) + lane * 4) : zero;
  }
  lds_barrier();
#pragma unroll
  for (int k = 0; k < kRcsSP; ++k) {
    if (k >= nc) continue;
    if (wave == k % kRcsTrsmWaves) {
      const bcr_d4 t = rcs_mma(rcs_load(rcs_lds + k * 256 + lane * 4), s[k / kRcsTrsmWaves], zero);
      s[k / kRcsTrsmWaves] = t;
      bcr_d4 sc;
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[r] = -rds[k * 16 + lk + 4 * r] * t[r];
      rcs_store(xch + (k & 1) * 256 + lane * 4, sc);
    }
    lds_barrier();
    if (k + 1 >= nc) continue;
    const bcr_d4 sc = rcs_load(xch + (k & 1) * 256 + lane * 4);
    bcr_d4 tj[kSlots];
    bool up[kSlots];
#pragma unroll
    for (int u = 0; u < kSlots; ++u) {
      const int jl = wave + kRcsTrsmWaves * u;
      up[u] = jl > k && jl < nc;
      tj[u] = up[u] ? rcs_load(ts + rcs_tri(jl, k) * 256 + lane * 4) : zero;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < kSlots; ++u)
        if (up[u]) s[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(tj[u][q], sc[q], s[u], 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < kSlots; ++u) {                                 // (stores after the chain: see rcs_pivot_role)
    const int jl = wave + kRcsTrsmWaves * u;
    if (jl >= nc) continue;
    bcr_d4 sc;
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[r] = -rds[jl * 16 + lk + 4 * r] * s[u][r];
    rcs_store(rv.Simg + rcs_blk(gi, c0 + jl) + lane * 4, s[u]);
    rcs_store(rv.Tsc + rcs_blk(gi, c0 + jl) + lane * 4, sc);
  }
}

// trailing blocks (i, j), c1 <= j <= i <= nbk (i = nbk: the right-hand side): U_ij -= sum_k T_jk^T Delta_k^-1 T_ik, one wavefront each
__global__ __launch_bounds__(256) void k_rcs_syrk(RcsView rv, int c0) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nc = min(kRcsSP, rv.nbk - c0), c1 = c0 + nc, m = rv.nbk - c1;
  const int t = blockIdx.x * 4 + wave, ntri = m * (m + 1) / 2;
  if (t >= ntri + m) return;
  int il, jl;
  if (t < ntri) {
    il = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((il + 1) * (il + 2) / 2 <= t) ++il;
    while (il * (il + 1) / 2 > t) --il;
    jl = t - il * (il + 1) / 2;
  } else { il = m; jl = t - ntri; }
  const int gi = c1 + il, gj = c1 + jl;                              // (c1 + m = nbk: the right-hand-side row)
  double* blk = rv.Simg + rcs_blk(gi, gj) + lane * 4;
  bcr_d4 acc = rcs_load(blk);
  bcr_d4 pj[kRcsSP], qi[kRcsSP];
#pragma unroll
  for (int k = 0; k < kRcsSP; ++k) {
    if (k >= nc) continue;
    pj[k] = rcs_load(rv.Simg + rcs_blk(gj, c0 + k) + lane * 4);
    qi[k] = rcs_load(rv.Tsc + rcs_blk(gi, c0 + k) + lane * 4);
  }
#pragma unroll
  for (int k = 0; k < kRcsSP; ++k)
    if (k < nc) acc = rcs_mma(pj[k], qi[k], acc);
  rcs_store(blk, acc);
}

// x = L^-T Delta^-1 t for the super-panels Kfirst .. 0 (the last one is solved inside its k_rcs_factor launch, its x is in rv.x),
// ONE workgroup of sixteen wavefronts; pc = -x.  Per super-panel: the staged factor of its diagonal super-block, the products
// T_i,jl x_i of the block rows below (x of the later super-panels) -- all loads of these three groups are issued before the first is
// waited for: one memory round trip --, then the chain  x_k = X_k^T Delta_k^-1 u_k;  u_jl -= T_k,jl x_k  (one wavefront per jl).
// Dynamic LDS: x (npad) | u (9 x 16) | partial sums per wavefront (16 x 9 x 16) | the staged factor (rcs_stage_doubles(9)).
constexpr int kRcsBackWaves = 16;
__host__ __device__ inline int rcs_backsub_doubles(int nbk) { return nbk * 16 + (1 + kRcsBackWaves) * kRcsSP * 16 + rcs_stage_doubles(kRcsSP); }
__global__ __launch_bounds__(64 * kRcsBackWaves) void k_rcs_backsub(RcsView rv, double* __restrict__ pc, int Kfirst) {
  extern __shared__ __attribute__((aligned(32))) double rcs_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 15, lk = lane >> 4;
  const int nbk = rv.nbk, R = nbk;
  double* xs = rcs_lds;
  double* u = xs + nbk * 16;
  double* up = u + kRcsSP * 16;
  double* st = up + kRcsBackWaves * kRcsSP * 16;
  for (int a = (Kfirst + 1) * kRcsSP * 16 + threadIdx.x; a < nbk * 16; a += 64 * kRcsBackWaves) xs[a] = rv.x[a];
  constexpr int kBatch = 4;
  for (int K = Kfirst; K >= 0; --K) {
    const int c0 = K * kRcsSP, nc = min(kRcsSP, nbk - c0), c1 = c0 + nc;
    lds_barrier();                                                   // (the previous super-panel's readers of st are done; xs is complete)
    // ---- issue: the staged factor (three blocks per wavefront), column 0 of T_R, the first batch of blocks below
    double* rds = st + nc * 256;
    double* ts = rds + nc * 16;
    const int nblk = nc + nc * (nc - 1) / 2;
    constexpr int kPer = (kRcsSP + kRcsSP * (kRcsSP - 1) / 2 + kRcsBackWaves - 1) / kRcsBackWaves;
    bcr_d4 sv[kPer];
    double srd[kPer];
    int sdst[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int t = wave + kRcsBackWaves * q;
      sdst[q] = -1; srd[q] = 0.0;
      if (t < nc) {
        sv[q] = rcs_load(rv.Simg + rcs_blk(c0 + t, c0 + t) + lane * 4);
        if (lane < 16) srd[q] = rv.Tsc[rcs_blk(c0 + t, c0 + t) + lane];
        sdst[q] = t * 256;
      } else if (t < nblk) {
        const int e = t - nc;
        int j = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);
        while (j * (j - 1) / 2 > e) --j;
        while (j * (j + 1) / 2 <= e) ++j;
        sv[q] = rcs_load(rv.Simg + rcs_blk(c0 + j, c0 + e - j * (j - 1) / 2) + lane * 4);
        sdst[q] = (int)(ts - st) + e * 256;
      }
    }
    double tR = 0.0;
    if (threadIdx.x < nc * 16) {
      const int jl = threadIdx.x / 16, row = threadIdx.x % 16;
      tR = rv.Simg[rcs_blk(R, c0 + jl) + ((row & 3) * 16) * 4 + (row >> 2)];            // column 0 of T_R,jl
    }
    const int nitem = (nbk - c1) * nc;
    bcr_d4 img[kBatch];
    double xv[kBatch];
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const int it = min(wave + kRcsBackWaves * q, max(nitem - 1, 0)), i = c1 + it / nc, jl = it % nc;
      img[q] = nitem > 0 ? rcs_load(rv.Simg + rcs_blk(i, c0 + jl) + lane * 4) : bcr_d4{0.0, 0.0, 0.0, 0.0};
      xv[q] = nitem > 0 ? xs[i * 16 + lr] : 0.0;
    }
    // ---- commit the stage, then the products of the rows below (partial sums per wavefront, added in wavefront order below)
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int t = wave + kRcsBackWaves * q;
      if (sdst[q] >= 0) rcs_store(st + sdst[q] + lane * 4, sv[q]);
      if (t < nc && lane < 16) rds[t * 16 + lane] = srd[q];
    }
    for (int e = lane; e < nc * 16; e += 64) up[wave * kRcsSP * 16 + e] = 0.0;
    lds_wave_sync();
    for (int it0 = wave; it0 < nitem; it0 += kRcsBackWaves * kBatch) {
      if (it0 != wave) {                                             // (the first batch is in flight already)
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
          const int it = min(it0 + kRcsBackWaves * q, nitem - 1), i = c1 + it / nc, jl = it % nc;
          img[q] = rcs_load(rv.Simg + rcs_blk(i, c0 + jl) + lane * 4);
          xv[q] = xs[i * 16 + lr];
        }
      }
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
        const int it = it0 + kRcsBackWaves * q;
        if (it >= nitem) continue;
        const int jl = it % nc;
        const bcr_d4 y = rcs_matvec(img[q], xv[q]);
        if (lr == 15) {
#pragma unroll
          for (int r = 0; r < 4; ++r) up[wave * kRcsSP * 16 + jl * 16 + lk + 4 * r] += y[r];
        }
      }
    }
    lds_barrier();
    if (threadIdx.x < nc * 16) {
      double v = tR;
#pragma unroll
      for (int w = 0; w < kRcsBackWaves; ++w) v -= up[w * kRcsSP * 16 + threadIdx.x];
      u[threadIdx.x] = v;
    }
    lds_barrier();
    for (int k = nc - 1; k >= 0; --k) {
      if (wave == 0) {
        const bcr_d4 y = rcs_matvec(rcs_load(st + k * 256 + lane * 4), rds[k * 16 + lr] * u[k * 16 + lr]);
        if (lr == 15) {
#pragma unroll
          for (int r = 0; r < 4; ++r) xs[(c0 + k) * 16 + lk + 4 * r] = y[r];
        }
      }
      lds_barrier();
      if (wave < k) {                                                // jl = wave
        const bcr_d4 y = rcs_matvec(rcs_load(ts + rcs_tri(k, wave) * 256 + lane * 4), xs[(c0 + k) * 16 + lr]);
        if (lr == 15) {
#pragma unroll
          for (int r = 0; r < 4; ++r) u[wave * 16 + lk + 4 * r] -= y[r];
        }
      }
      lds_barrier();
    }
  }
  const int top = min((Kfirst + 1) * kRcsSP * 16, rv.nn);
  for (int a = threadIdx.x; a < top; a += 64 * kRcsBackWaves) pc[a] = -xs[a];
}

}  // namespace mvus
