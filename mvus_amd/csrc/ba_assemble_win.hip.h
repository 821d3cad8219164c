// Window-major fused linearisation of the detection rows (the LM path): Jacobian blocks, J^T J and J^T f in one kernel, every
// entry of the normal equations written ONCE with a plain store -- no atomics, no clearing pass, the same bits on every run.
//
// Reference semantics: the rows are those of error_BA (multiviewunsynch/reconstruction/common.py:448-487; per camera |ex|, |ey| of
// every detection, zero rows for detections outside the spline intervals, common.py:357-358) and their derivatives; the reference
// never forms J^T J (scipy's trf/lsmr works on J), so the sums themselves have no counterpart there.
//
// Decomposition.  A workgroup owns a WINDOW of Wn consecutive control points [a, a + Wn) and its four wavefronts walk the cameras
// (wavefront v takes cameras v, v + 4, ...).  A detection row touches the four control points g .. g+3 of its knot span, so the
// rows that reach the window are those with g in [a - 3, a + Wn): per camera a contiguous range of the frame-sorted detections,
// found without a search -- the window's time range [win_tlo[a - 3], win_thi[a + Wn]) is turned into a frame range with the camera's
// alpha, beta, rs and looked up in a per-camera frame grid (CamWin: one table read per end, a superset by up to a cell on each
// side); every lane then evaluates one detection (eval_observation_to, the arithmetic of k_observations) and keeps it only if its
// span really is one of the window's Wn + 3.  The three spans below a are evaluated by the previous window as well (19 % more
// evaluations at Wn = 16): that is the price of giving every output entry exactly one writer.
//
// Inside a wavefront (no workgroup barrier until the very end):
//   stage      lane = detection.  The row pair is kept FACTORED: the twelve spline slots are h (x) [gu; gv] (four basis values, the
//              2 x 3 derivative w.r.t. the world point), so 4 + 6 + 2 B + 2 values per detection go to the wavefront's LDS region.
//   index      ds_or_b64 builds, per span of the window, the 64-bit mask of the lanes that hold one of its detections (any order:
//              a rolling-shutter coefficient that swaps neighbouring time stamps needs no sort).
//   camera blk G = R R^T, R = [camera slots; f] of the detections whose span is >= a (each detection is OWNED by one window), on the
//              fp64 matrix cores; one partial block per (window, camera), summed over the windows in order by k_cam_block_sum.
//   accumulate lane = (span j, coordinate d): walks its span's mask and adds, in registers,
//                  E[q][k]  += h_q (gu_d Jx_k + gv_d Jy_k)          cross block rows (g + q, d), this camera's B columns
//                  gq[q]    += h_q (gu_d fx + gv_d fy)              spline gradient
//                  C[qa][w] += h_qa h_{qa+w} (gu_d gu + gv_d gv)    band block (g + qa, g + qa + w), row d
//              -- 106 multiply-adds for 30 LDS reads per detection and lane; band and gradient accumulators live across the whole
//              camera walk, the cross-block accumulators across the batches of one camera.
//   flush E    the <= 4 spans that reach a control point are added in span order through the (now dead) staging region and the
//              camera's Wn x 3 x B block of Et leaves as one contiguous, coalesced run of plain stores.
//   finish     after the walk the four wavefronts' band / gradient sums are combined in a fixed order and stored.
// Every sum has one order: detections ascending inside a span, spans ascending per control point, cameras ascending, wavefronts
// ascending.  Sparse tracks, dense tracks, spans without detections need no special path (a span's mask is simply empty).
#pragma once
#include <hip/hip_runtime.h>

namespace mvus {

constexpr int kWinWaves = 4, kWinThreads = 64 * kWinWaves;
constexpr int kWinStr = 65;                     // staging row stride (odd: the matrix-core fragment reads walk the rows)
constexpr int kWinMaxW = 16, kWinMaxJ = kWinMaxW + 3;      // 3 * (Wn + 3) lane roles must fit a wavefront (and two workgroups the LDS of a CU)
constexpr int kWinSpl = 18;                                 // doubles of one span's record in LDS: six knots, 3 x 4 coefficients

struct WinView {
  const CamWin* cw;          // [C]
  const int32_t* flut;       // frame grids (CamWin::lut_off)
  const double* tlo;         // [Ntot + 1] by GLOBAL control point: <= every visible time stamp whose first control point is >= p
  const double* thi;         // [Ntot + 1]: >= every visible time stamp whose first control point is < p
  double* Apart;             // [nwin][C][(B+1)(B+2)/2] lower triangle of [camera slots; f][..]^T per (window, camera)
  const int32_t* span;       // [M] first control point (global) of every detection at the x being linearised, -1 = not visible
  const int4* crec;          // [Ntot] per control point g as a FIRST control point: {x index of coordinate 0, coefficients of its spline
                             //        (stride between coordinates), index of knot t[l-2] of its span, 1: first span | 2: last span of the interval}
  int Wn, nwin;
  int Ntot;                  // control points of the whole problem (a time shard's slice is shorter)
};

constexpr __host__ __device__ int win_region_doubles(int B) {       // LDS doubles per wavefront: staging, reused by the flushes
  const int stage = (12 + 2 * B) * kWinStr, eflush = 4 * kWinMaxW * 3 * B, cflush = kWinMaxW * (10 * 9 + 4 * 3);
  return stage > eflush ? (stage > cflush ? stage : cflush) : (eflush > cflush ? eflush : cflush);
}

// frame grid of every camera: one thread per cell (binary search over the camera's frames); again after detections were removed
__global__ void k_frame_lut(DevProblem dp, const CamWin* __restrict__ cw, int32_t* __restrict__ flut, long long total) {
  const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (e >= total) return;
  int lo = 0, hi = dp.C - 1;                   // camera whose table holds entry e
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cw[mid].lut_off <= e) lo = mid; else hi = mid - 1; }
  const CamWin w = cw[lo];
  const int k = (int)(e - w.lut_off);
  const long long a = dp.det_off[lo];
  const int Mc = (int)(dp.det_off[lo + 1] - a);
  int first = Mc;
  if (k < w.ncell) {
    const double edge = w.f0 + (double)k / w.scale;
    int p = 0, q = Mc;                          // first detection with frame >= edge
    while (p < q) { const int mid = (p + q) >> 1; if (dp.frame[a + mid] >= edge) q = mid; else p = mid + 1; }
    first = p;
  }
  if (k == 0) first = 0;
  flut[e] = first;
}

template <int B>
struct WinSink {             // eval_observation_to sink of the window-major assembly: camera slots as they come, the spline slots factored
  static constexpr bool kFactored = true;
  double* col;               // staging region + lane
  __device__ __forceinline__ void begin(int32_t) {}
  __device__ __forceinline__ void x(int k, double v) { if (k < B) col[(10 + k) * kWinStr] = v; }
  __device__ __forceinline__ void y(int k, double v) { if (k < B) col[(10 + B + k) * kWinStr] = v; }
  __device__ __forceinline__ void factored(const double h[4], double gu0, double gu1, double gu2, double gv0, double gv1, double gv2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) col[q * kWinStr] = h[q];
    col[4 * kWinStr] = gu0; col[5 * kWinStr] = gu1; col[6 * kWinStr] = gu2;
    col[7 * kWinStr] = gv0; col[8 * kWinStr] = gv1; col[9 * kWinStr] = gv2;
  }
};

__device__ __forceinline__ void win_wave_sync() {      // orders the LDS traffic of ONE wavefront (writes of some lanes read by others)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

#ifndef MVUS_WIN_MFMA_UNROLL
#define MVUS_WIN_MFMA_UNROLL 4
#endif
#ifndef MVUS_WIN_WAVES_PER_EU
#define MVUS_WIN_WAVES_PER_EU 2
#endif
MVUS_HD int win_pieces(int Wn) { return 21 / (Wn + 3); }      // pieces a span's detections are dealt into (dense tracks, short windows)
constexpr __host__ __device__ int win_wave_doubles(int B) { return win_region_doubles(B) + kWinMaxW * 3 * B + 24; }   // + E accumulator + span masks
constexpr __host__ __device__ int win_lds_doubles(int B) { return kWinWaves * win_wave_doubles(B) + kWinMaxJ * kWinSpl; }   // + the window's span records

template <int B>
__global__ __launch_bounds__(kWinThreads) __attribute__((amdgpu_waves_per_eu(B == 9 ? MVUS_WIN_WAVES_PER_EU : 1, B == 9 ? MVUS_WIN_WAVES_PER_EU : 1)))
void k_assemble_windows(DevProblem dp, NEView ne, WinView wv, const CamState* __restrict__ cams, const double* __restrict__ x) {
  constexpr bool CALIB = B == 18;
  constexpr int NV = 12 + 2 * B;                            // staged values per detection: h[4] gu[3] gv[3] Jx[B] Jy[B] fx fy
  constexpr int kFx = 10 + 2 * B, kFy = 11 + 2 * B;
  constexpr int PSZ = (B + 1) * (B + 2) / 2, TI = (B + 1 + 15) / 16;
  constexpr int REG = win_region_doubles(B), WAVE = win_wave_doubles(B);
  static_assert(NV * kWinStr <= REG, "staging fits the region");
  extern __shared__ double win_lds[];                       // per wavefront: [REG] staging / flush region, [kWinMaxW * 3 * B] E accumulator, [24] masks
  using d4v = __attribute__((ext_vector_type(4))) double;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* S = win_lds + wave * WAVE;
  double* Eacc = S + REG;
  unsigned long long* mk = reinterpret_cast<unsigned long long*>(Eacc + kWinMaxW * 3 * B);
  double* spl = win_lds + kWinWaves * WAVE;                 // [NJ][18]: knots t[l-2 .. l+3] and coefficients (x, y, z) x 4 of every span of the window
  const int win = blockIdx.x;
  const int a = win * wv.Wn;                                // first owned control point (local to the handle's slice)
  const int nown = min(wv.Wn, ne.N - a), NJ = nown + 3;     // spans a - 3 .. a + nown - 1 reach the window
  const int SP = win_pieces(wv.Wn);                         // lane = (span j, piece s, coordinate d)
  const int j = lane / (3 * SP), s = (lane / 3) % SP, d = lane % 3;
  const bool role = j < NJ;
  const int mslot = j * SP + s;
  // the window's time range; the first / last window of a time shard's slice also sees what lies beyond the slice (and flags it)
  const double inf = INFINITY;
  const int ga = a + ne.row0;
  const double T0 = (win == 0 && ne.row0 > 0) ? -inf : wv.tlo[max(ga - 3, 0)];
  const double T1 = (win == wv.nwin - 1 && ne.row0 + ne.N < wv.Ntot) ? inf : wv.thi[ga + nown];
  if (lane < 24) mk[lane] = 0ull;
  // the window's spline data, once per workgroup: every evaluation below reads knots and coefficients from LDS (no dependent global
  // loads on the per-detection path); a control point that cannot start a span (the last three of an interval) leaves zeros
  for (int e = threadIdx.x; e < NJ * kWinSpl; e += kWinThreads) {
    const int kk = e / kWinSpl, r = e - kk * kWinSpl;
    const int g = ga - 3 + kk;
    double v = 0.0;
    if (g >= 0 && g < wv.Ntot) {
      const int4 rec = wv.crec[g];
      if (!(rec.w & 4)) v = r < 6 ? dp.sp.knots[rec.z + r] : x[rec.x + (r - 6) % 4 + ((r - 6) / 4) * rec.y];
    }
    spl[e] = v;
  }
  unsigned flags = 0u;                                      // per span: 1 first / 2 last span of its interval (lane kk < NJ holds span kk's)
  if (lane < NJ) { const int g = ga - 3 + lane; if (g >= 0 && g < wv.Ntot) flags = (unsigned)wv.crec[g].w; }
  __syncthreads();

  // ---- detection range of every camera of this wavefront, all at once (lane i: camera wave + 4 i): frames that can carry a time
  //      stamp in [T0, T1), tau = alpha (frame + rs v / H) + beta with v in [vmin, vmax], looked up in the camera's frame grid ----
  const int ncam = (dp.C - wave + kWinWaves - 1) / kWinWaves;       // cameras of this wavefront: <= 64 (the reduced camera system limits C * B to 1152)
  int p0v = 0, p1v = 0;
  if (lane < ncam) {
    const int c = wave + kWinWaves * lane;
    const CamState* cs = cams + c;
    const CamWin cw = wv.cw[c];
    const double alpha = cs->alpha, beta = cs->beta, rs = cs->rs, H = cs->H;
    const int Mc = (int)(dp.det_off[c + 1] - dp.det_off[c]);
    int p0 = 0, p1 = Mc;
    const double ra = rs * cw.vmin / H, rb = rs * cw.vmax / H;
    const double FL = (T0 - beta) / alpha - fmax(ra, rb), FH = (T1 - beta) / alpha - fmin(ra, rb);
    if (alpha > 0.0 && FL <= FH) {                          // (anything else -- alpha <= 0, NaN -- : the whole camera, the span test decides)
      const double kl = fmin(fmax(floor((FL - cw.f0) * cw.scale) - 1.0, 0.0), (double)cw.ncell);
      const double kh = fmin(fmax(floor((FH - cw.f0) * cw.scale) + 2.0, 0.0), (double)cw.ncell);
      p0 = wv.flut[cw.lut_off + (int)kl];
      p1 = wv.flut[cw.lut_off + (int)kh];
    }
    p0v = p0; p1v = max(p0, p1);
  }
  auto cam_range = [&](int i, int& p0, int& p1) {           // (i wave-uniform)
    p0 = __builtin_amdgcn_readlane(p0v, i); p1 = __builtin_amdgcn_readlane(p1v, i);
  };

  double CA[30], gq[4];                                    // band blocks (pair (qa, w), column d2), gradient: across the camera walk
#pragma unroll
  for (int i = 0; i < 30; ++i) CA[i] = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) gq[q] = 0.0;
#define MVUS_WCA(qa, w, d2) CA[(4 * (qa) - (qa) * ((qa) - 1) / 2 + (w)) * 3 + (d2)]

  // inputs of one batch (lane = detection), fetched one batch ahead: the detection, its first control point from the span table
  // (key: its span's number in the window, -1 = not this window's / not visible) and that control point's record
  struct Inputs { double fr, vr, p, q; int g; };
  auto fetch = [&](long long a0, int pos, int p1) {
    Inputs in{0.0, 0.0, 0.0, 0.0, -1};
    if (pos < p1) {                                         // (five independent loads: nothing here waits for the span)
      const long long i = a0 + pos;
      in.g = wv.span[i];
      in.fr = dp.frame[i]; in.vr = dp.v_raw[i];
      in.p = CALIB ? dp.u_raw[i] : dp.u_obs[i];
      in.q = CALIB ? 0.0 : dp.v_obs[i];
    }
    return in;
  };
  auto span_key = [&](int g) {                              // number of the span in the window, -1: not this window's / not visible
    if (g < 0) return -1;
    const int gl = g - ne.row0;                             // local first control point; a time shard must hold all four
    if (gl < 0 || gl + 3 >= ne.N) { atomicOr(ne.err, 1); return -1; }
    const int kk = gl - (a - 3);
    return (kk >= 0 && kk < NJ) ? kk : -1;
  };

#ifdef MVUS_WIN_PROBE
  long long tpa[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tprev = clock64(); const long long tstart = tprev; int nbatch = 0;
#define MVUS_WTP(i) do { const long long t_ = clock64(); tpa[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define MVUS_WTP(i) ((void)0)
#endif
  int p0 = 0, p1 = 0;
  if (ncam > 0) cam_range(0, p0, p1);
  Inputs cur = ncam > 0 ? fetch(dp.det_off[wave], p0 + lane, p1) : Inputs{0.0, 0.0, 0.0, 0.0, -1};
  for (int ci = 0; ci < ncam; ++ci) {
    const int c = wave + kWinWaves * ci;
    const CamState& cam = cams[c];                          // wave-uniform: scalar loads
    const long long a0 = dp.det_off[c];
    int np0 = 0, np1 = 0;                                   // the next camera's range
    if (ci + 1 < ncam) cam_range(ci + 1, np0, np1);
    d4v cacc[2][TI][TI];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TI; ++jj) cacc[h2][i][jj] = d4v{0.0, 0.0, 0.0, 0.0};
    const int PS = nown * 3 * B;
    double* er = ne.Et + ((long long)c * ne.N3 + 3 * a) * B;
    if (!(p0 < p1)) {                                       // no detection of this camera near the window: its block of Et is zero
      for (int o = lane; o < PS; o += 64) er[o] = 0.0;
      cur = (ci + 1 < ncam) ? fetch(dp.det_off[c + kWinWaves], np0 + lane, np1) : Inputs{0.0, 0.0, 0.0, 0.0, -1};
    }

    for (int base = p0; base < p1; base += 64) {
      const bool first = base == p0, last = base + 64 >= p1;
      // the next batch's inputs (this camera's, else the first of the next camera): in flight over this batch's arithmetic
      Inputs nxt;
      if (!last) nxt = fetch(a0, base + 64 + lane, p1);
      else if (ci + 1 < ncam) nxt = fetch(dp.det_off[c + kWinWaves], np0 + lane, np1);
      else nxt = Inputs{0.0, 0.0, 0.0, 0.0, -1};
      MVUS_WTP(0);     // camera set-up / previous flush tail
      // ---- stage: lane = detection; knot span known (span table), its knots and coefficients from LDS ----
      const int key = span_key(cur.g);
      // staged column of a detection: its rank among the lanes that hold one of this window's (lane order kept) -- the columns in
      // use are contiguous, so the matrix-core pass below walks the owned ones only
      const unsigned long long valid = __ballot(key >= 0);
      const int slot = __popcll(valid & ((1ull << lane) - 1ull));
      if (key >= 0) {
        SpanLoc loc;
        loc.ctrl = ga - 3 + key; loc.n = 0; loc.xoff = 0;
        const double* sk = spl + key * kWinSpl;
#pragma unroll
        for (int t = 0; t < 6; ++t) loc.tt[t] = sk[t];
#pragma unroll
        for (int dd = 0; dd < 3; ++dd)
#pragma unroll
          for (int q = 0; q < 4; ++q) loc.c[dd][q] = sk[6 + 4 * dd + q];
        const double tau = cam.alpha * (cur.fr + cam.rs * cur.vr / cam.H) + cam.beta;
        // the table must be the one of THIS x (FITPACK rule t[l] <= tau < t[l+1], open at the clamped ends of an interval)
        const unsigned fl = (unsigned)__shfl((int)flags, key, 64);
        if (!(((fl & 1u) || loc.tt[2] <= tau) && ((fl & 2u) || tau < loc.tt[3]))) atomicOr(ne.err, 2);
        WinSink<B> sink{S + slot};
        const ObsResult r = eval_observation_at<CALIB, true>(cam, tau, loc, dp.undist != 0, dp.rs_free != 0, dp.sync_free != 0,
                                                             cur.fr, CALIB ? cur.p : 0.0, cur.vr, CALIB ? 0.0 : cur.p, cur.q, sink);
        S[kFx * kWinStr + slot] = r.ex; S[kFy * kWinStr + slot] = r.ey;
        atomicOr(&mk[key * SP + slot % SP], 1ull << slot);  // a span's detections are dealt to its pieces in turn
      }
      // spans >= a: the detections this window OWNS (camera block), as a mask over the staged columns (which keep the lane order:
      // the ballot shifted down when the lanes in use are one run -- anything else only with gaps in visibility -- else compressed)
      unsigned long long own = __ballot(key >= 3);
      if (own != 0ull) {
        const int x0 = __ffsll((long long)valid) - 1, nv = __popcll(valid);
        if ((valid >> x0) == (nv == 64 ? ~0ull : (1ull << nv) - 1ull)) own >>= x0;
        else {
          unsigned long long m = 0ull, va = valid;
          int col = 0;
          while (va != 0ull) { const int l = __ffsll((long long)va) - 1; va &= va - 1ull; if ((own >> l) & 1ull) m |= 1ull << col; ++col; }
          own = m;
        }
      }
      win_wave_sync();
      MVUS_WTP(1);     // evaluation + staging
      unsigned long long mym = 0ull;
      if (role) { mym = mk[mslot]; if (d == 0) mk[mslot] = 0ull; }   // (the three lanes of a slot read before lane d = 0 clears: LDS is in order)
#if defined(MVUS_WIN_STOP) && MVUS_WIN_STOP == 1
      mym = 0ull;                                           // timing probe: evaluation + staging only
#endif
      // ---- camera block of the owned detections on the matrix cores: two independent accumulation chains (x rows, y rows) ----
#if defined(MVUS_WIN_STOP) && (MVUS_WIN_STOP == 1 || MVUS_WIN_STOP == 3)
      if (false) {
#else
      if (own != 0ull) {
#endif
        const int lr = lane & 15, lk = lane >> 4;
        int srcs[2][TI];
        bool rowok[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int row = 16 * i + lr;
          rowok[i] = row <= B;
#pragma unroll
          for (int xy = 0; xy < 2; ++xy) srcs[xy][i] = (row < B ? 10 + xy * B + row : kFx + xy) * kWinStr + lk;
        }
        double nv[2][TI];                                   // operands of the next step: in flight over this step's matrix instructions
        const int k_lo = (__ffsll((long long)own) - 1) >> 2, k_hi = ((63 - __clzll((long long)own)) >> 2) + 1;   // steps of four columns that hold owned ones
#pragma unroll
        for (int xy = 0; xy < 2; ++xy)
#pragma unroll
          for (int i = 0; i < TI; ++i) nv[xy][i] = S[srcs[xy][i] + 4 * k_lo];
        for (int kk = k_lo; kk < k_hi; ++kk) {
          const bool on = (own >> (4 * kk + lk)) & 1ull;
          double av[2][TI];
#pragma unroll
          for (int xy = 0; xy < 2; ++xy)
#pragma unroll
            for (int i = 0; i < TI; ++i) av[xy][i] = (on && rowok[i]) ? nv[xy][i] : 0.0;
          if (kk + 1 < k_hi) {
#pragma unroll
            for (int xy = 0; xy < 2; ++xy)
#pragma unroll
              for (int i = 0; i < TI; ++i) nv[xy][i] = S[srcs[xy][i] + 4 * (kk + 1)];
          }
#pragma unroll
          for (int xy = 0; xy < 2; ++xy)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
              for (int jj = 0; jj <= i; ++jj) cacc[xy][i][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[xy][i], av[xy][jj], cacc[xy][i][jj], 0, 0, 0);
        }
      }
      MVUS_WTP(2);     // camera block on the matrix cores
      // ---- accumulate: lane = (span j, piece s, coordinate d); the cross-block sums of THIS batch (flushed below) ----
#if defined(MVUS_WIN_STOP) && MVUS_WIN_STOP == 2
      mym = 0ull;                                           // timing probe: no accumulation
#endif
      double E[4][B];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < B; ++k) E[q][k] = 0.0;
      while (mym != 0ull) {
        const int u = __ffsll((long long)mym) - 1;
        mym &= mym - 1ull;
        const double* su = S + u;
        double hb[4], gu[3], gv[3];
#pragma unroll
        for (int q = 0; q < 4; ++q) hb[q] = su[q * kWinStr];
#pragma unroll
        for (int e = 0; e < 3; ++e) { gu[e] = su[(4 + e) * kWinStr]; gv[e] = su[(7 + e) * kWinStr]; }
        const double gud = d == 0 ? gu[0] : (d == 1 ? gu[1] : gu[2]);
        const double gvd = d == 0 ? gv[0] : (d == 1 ? gv[1] : gv[2]);
#pragma unroll
        for (int k = 0; k < B; ++k) {
          const double V = gud * su[(10 + k) * kWinStr] + gvd * su[(10 + B + k) * kWinStr];
#pragma unroll
          for (int q = 0; q < 4; ++q) E[q][k] += hb[q] * V;
        }
        const double Vg = gud * su[kFx * kWinStr] + gvd * su[kFy * kWinStr];
#pragma unroll
        for (int q = 0; q < 4; ++q) gq[q] += hb[q] * Vg;
        double mm[3];
#pragma unroll
        for (int d2 = 0; d2 < 3; ++d2) mm[d2] = gud * gu[d2] + gvd * gv[d2];
#pragma unroll
        for (int qa = 0; qa < 4; ++qa)
#pragma unroll
          for (int w = 0; qa + w < 4; ++w) {
            const double bb = hb[qa] * hb[qa + w];
#pragma unroll
            for (int d2 = 0; d2 < 3; ++d2) MVUS_WCA(qa, w, d2) += bb * mm[d2];
          }
      }
      MVUS_WTP(3);     // accumulation
      // ---- the batch's part of the camera's block of Et: spans j = pl + 3 - q, q = 0..3 (and their pieces) reach the owned control
      //      point pl; the lane of span pl + 3, piece 0 collects them in (q, piece) order with lane shuffles (no trip through
      //      memory), adds the earlier batches' sum (the wavefront's accumulator: the lane's own nine entries) and the last batch
      //      stores the row ----
      {
        const bool owner = role && s == 0 && j >= 3;        // row (pl, d) = (j - 3, d)
        const int orow = ((j - 3) * 3 + d) * B;
#pragma unroll
        for (int k = 0; k < B; ++k) {
          double acc = (owner && !first) ? Eacc[orow + k] : 0.0;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            for (int sp = 0; sp < SP; ++sp) {
              const int src = ((j - q) * SP + sp) * 3 + d;
              acc += __shfl(E[q][k], src & 63, 64);
            }
          if (owner) { if (last) er[orow + k] = acc; else Eacc[orow + k] = acc; }
        }
      }
      win_wave_sync();                                      // the staging region is rewritten by the next batch
      MVUS_WTP(4);     // E flush
#ifdef MVUS_WIN_PROBE
      ++nbatch;
#endif
      cur = nxt;
    }

    // the camera block's partial (C/D layout of the 16x16 tile: row = (lane >> 4) + 4 reg, column = lane & 15)
    {
      double* mine = wv.Apart + ((long long)win * dp.C + c) * PSZ;
      const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj <= i; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ra = 16 * i + lk + 4 * r, rb = 16 * jj + lr;
            if (ra > B || rb > ra) continue;
            mine[ra * (ra + 1) / 2 + rb] = cacc[0][i][jj][r] + cacc[1][i][jj][r];
          }
    }
    p0 = np0; p1 = np1;
  }

  MVUS_WTP(5);
#ifdef MVUS_WIN_PROBE
  if ((blockIdx.x % 97) == 5 && lane == 0)
    printf("win %d wave %d ncam %d batches %d: total %lld | per batch: setup %lld eval %lld mfma %lld accum %lld eflush %lld | tail %lld\n", win, wave, ncam, nbatch,
           clock64() - tstart, tpa[0] / max(nbatch, 1), tpa[1] / max(nbatch, 1), tpa[2] / max(nbatch, 1), tpa[3] / max(nbatch, 1), tpa[4] / max(nbatch, 1), tpa[5]);
#endif
  // ---- band and gradient: overlap-add of the spans and pieces, then of the four wavefronts, one fixed order ----
  const int CP = nown * 9, GP = nown * 3, GP0 = 10 * SP * CP;
  if (role) {
#pragma unroll
    for (int qa = 0; qa < 4; ++qa) {
      const int pl = j + qa - 3;
      if (pl >= 0 && pl < nown) {
#pragma unroll
        for (int w = 0; qa + w < 4; ++w)
#pragma unroll
          for (int d2 = 0; d2 < 3; ++d2) S[((4 * qa - qa * (qa - 1) / 2 + w) * SP + s) * CP + pl * 9 + 3 * d + d2] = MVUS_WCA(qa, w, d2);
        S[GP0 + (qa * SP + s) * GP + pl * 3 + d] = gq[qa];
      }
    }
  }
#undef MVUS_WCA
  __syncthreads();
  const int per = 3 + ne.W * 9;
  for (int e = threadIdx.x; e < nown * per; e += kWinThreads) {
    const int pl = e / per, r = e - pl * per;
    double acc = 0.0;
    if (r < 3) {
      for (int v = 0; v < kWinWaves; ++v) {
        const double* Sv = win_lds + v * WAVE + GP0 + pl * 3 + r;
        double t = 0.0;
        for (int qs = 0; qs < 4 * SP; ++qs) t += Sv[qs * GP];
        acc += t;
      }
      ne.gs[3 * (a + pl) + r] = acc;
    } else {
      const int w = (r - 3) / 9, dd = (r - 3) - 9 * w;
      if (w < 4) {
        for (int v = 0; v < kWinWaves; ++v) {
          const double* Sv = win_lds + v * WAVE + pl * 9 + dd;
          double t = 0.0;
          for (int qa = 0; qa + w < 4; ++qa)
            for (int sp = 0; sp < SP; ++sp) t += Sv[((4 * qa - qa * (qa - 1) / 2 + w) * SP + sp) * CP];
          acc += t;
        }
      }
      ne.Cb[((long long)(a + pl) * ne.W) * 9 + (r - 3)] = acc;       // (w >= 4: the motion rows' blocks start from zero)
    }
  }
}

// A[c] (both triangles) and gc[c] from the per-(window, camera) partial blocks: one workgroup per camera, the windows added in
// index order (four interleaved chains per entry and group, the groups combined in order) -- plain stores, nothing to clear.
template <int B>
__global__ __launch_bounds__(1024) void k_cam_block_sum(int C, int nwin, const double* __restrict__ Apart, NEView ne) {
  constexpr int PSZ = (B + 1) * (B + 2) / 2, kLanes = PSZ <= 64 ? 64 : 256, kGroups = 1024 / kLanes;
  __shared__ double part[kGroups][kLanes];
  const int c = blockIdx.x;
  const int k = threadIdx.x % kLanes, grp = threadIdx.x / kLanes;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (k < PSZ) {
    int w = grp;
    for (; w + 3 * kGroups < nwin; w += 4 * kGroups) {
      a0 += Apart[((long long)w * C + c) * PSZ + k];
      a1 += Apart[((long long)(w + kGroups) * C + c) * PSZ + k];
      a2 += Apart[((long long)(w + 2 * kGroups) * C + c) * PSZ + k];
      a3 += Apart[((long long)(w + 3 * kGroups) * C + c) * PSZ + k];
    }
    for (; w < nwin; w += kGroups) a0 += Apart[((long long)w * C + c) * PSZ + k];
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
    if (ra == B) { if (rb < B) ne.gc[c * B + rb] = v; }
    else {
      ne.A[((long long)c * B + ra) * B + rb] = v;
      if (ra != rb) ne.A[((long long)c * B + rb) * B + ra] = v;
    }
  }
}

}  // namespace mvus
