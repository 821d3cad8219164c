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
//              -- 44 multiply-adds for 30 LDS reads per detection and lane (round 6: the band row through the 1 x 3 row of its point block); band and gradient accumulators live across the whole
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
constexpr int kWinMaxW = 21, kWinMaxJ = kWinMaxW + 3;      // 3 * Wn output rows (one lane each) must fit a wavefront
constexpr int kWinSpl = 18;                                 // doubles of one span's record in LDS: six knots, 3 x 4 coefficients

struct WinView {
  const CamWin* cw;          // [C]
  const int32_t* flut;       // frame grids (CamWin::lut_off)
  const double* tlo;         // [Ntot + 1] by GLOBAL control point: <= every visible time stamp whose first control point is >= p
  const double* thi;         // [Ntot + 1]: >= every visible time stamp whose first control point is < p
  double* Apart;             // [nwin][C][(B+1)(B+2)/2] lower triangle of [camera slots; f][..]^T per (window, camera)
  const int32_t* cam_perm;   // [C] the order in which the cameras are dealt to the four wavefronts (balanced by detection count)
  const int32_t* span;       // [M] first control point (global) of every detection at the x being linearised, -1 = not visible
  const int4* crec;          // [Ntot] per control point g as a FIRST control point: {x index of coordinate 0, coefficients of its spline
                             //        (stride between coordinates), index of knot t[l-2] of its span, 1: first span | 2: last span of the interval}
  int Wn, nwin;
  int Ntot;                  // control points of the whole problem (a time shard's slice is shorter)
  int G;                     // camera groups (round 6): the grid is nwin x G workgroups, group g walks the cameras perm[g * 4 + v + 4 G i] -- when the windows
                             // alone cannot fill the device (a time shard's slice, a few hundred control points) the cameras are dealt over more workgroups
  double* band_part;         // G > 1: [G][N * (3 + W * 9)] the groups' partial band rows and gradients (gs then Cb), summed in group order by k_cam_block_sum
  double* mark;              // non-null: workgroup 0 writes mark_val there when it starts (mapped host memory: the host's fetch spins on it --
  double mark_val;           //           this launch cannot start before everything enqueued in front of it has finished; HipBackend::fetch_poll_begin)
};

constexpr __host__ __device__ int win_region_doubles(int B) {       // LDS doubles per wavefront: staging, reused by the flushes
  const int stage = (15 + 2 * B) * kWinStr, eflush = kWinMaxW * 3 * B, cflush = kWinMaxW * 3 * 13;
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

// staged rows of one detection (column u of the wavefront's region, row stride kWinStr):
//   gu[3] gv[3] | Jx[B] Jy[B] | fx fy | h[4] | three rows of zeros (h[q + w] with q + w > 3: "no such control point")
template <int B>
struct WinRows {
  static constexpr int kGu = 0, kGv = 3, kJx = 6, kJy = 6 + B, kFx = 6 + 2 * B, kFy = 7 + 2 * B, kH = 8 + 2 * B, kZero = 12 + 2 * B, kRows = 15 + 2 * B;
};
template <int B>
struct WinSink {             // eval_observation_to sink of the window-major assembly: camera slots as they come, the spline slots factored
  static constexpr bool kFactored = true;
  using R = WinRows<B>;
  double* col;               // staging region + column
  __device__ __forceinline__ void begin(int32_t) {}
  __device__ __forceinline__ void x(int k, double v) { if (k < B) col[(R::kJx + k) * kWinStr] = v; }
  __device__ __forceinline__ void y(int k, double v) { if (k < B) col[(R::kJy + k) * kWinStr] = v; }
  __device__ __forceinline__ void factored(const double h[4], double gu0, double gu1, double gu2, double gv0, double gv1, double gv2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) col[(R::kH + q) * kWinStr] = h[q];
    col[(R::kGu + 0) * kWinStr] = gu0; col[(R::kGu + 1) * kWinStr] = gu1; col[(R::kGu + 2) * kWinStr] = gu2;
    col[(R::kGv + 0) * kWinStr] = gv0; col[(R::kGv + 1) * kWinStr] = gv1; col[(R::kGv + 2) * kWinStr] = gv2;
  }
};

__device__ __forceinline__ void win_wave_sync() {      // orders the LDS traffic of ONE wavefront (writes of some lanes read by others)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

#ifndef MVUS_WIN_WAVES_PER_EU
#define MVUS_WIN_WAVES_PER_EU 2
#endif
MVUS_HD int win_pieces(int Wn) { const int a = 21 / Wn, b = 32 / (Wn + 3); return a < b ? a : b; }          // lanes per output row: a span's detections are dealt to them in turn (dense tracks, short windows)
constexpr __host__ __device__ int win_wave_doubles(int B) { return win_region_doubles(B) + 32; }   // staging / flush region + span masks
constexpr __host__ __device__ int win_lds_doubles(int B) { return kWinWaves * win_wave_doubles(B) + kWinMaxJ * kWinSpl; }   // + the window's span records

template <int B>
__global__ __launch_bounds__(kWinThreads) __attribute__((amdgpu_waves_per_eu(B == 9 ? MVUS_WIN_WAVES_PER_EU : 1, B == 9 ? MVUS_WIN_WAVES_PER_EU : 1)))
void k_assemble_windows(DevProblem dp, NEView ne, WinView wv, const CamState* __restrict__ cams, const double* __restrict__ x) {
  constexpr bool CALIB = B == 18;
  using R = WinRows<B>;
  constexpr int NV = R::kRows;
  constexpr int kFx = R::kFx, kFy = R::kFy;
  constexpr int PSZ = (B + 1) * (B + 2) / 2, TI = (B + 1 + 15) / 16;
  constexpr int REG = win_region_doubles(B), WAVE = win_wave_doubles(B);
  static_assert(NV * kWinStr <= REG, "staging fits the region");
  extern __shared__ double win_lds[];                       // per wavefront: [REG] staging / flush region, [32] masks; then the window's span records
  if (wv.mark != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(wv.mark, wv.mark_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  using d4v = __attribute__((ext_vector_type(4))) double;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* S = win_lds + wave * WAVE;
  unsigned long long* mk = reinterpret_cast<unsigned long long*>(S + REG);
  double* spl = win_lds + kWinWaves * WAVE;                 // [NJ][18]: knots t[l-2 .. l+3] and coefficients (x, y, z) x 4 of every span of the window
  const int win = (int)blockIdx.x % wv.nwin, grp = (int)blockIdx.x / wv.nwin;       // (consecutive workgroups: consecutive windows of one camera group)
  const int slot = grp * kWinWaves + wave, nslot = wv.G * kWinWaves;                 // this wavefront's place among the G x 4 that share the cameras
  const int a = win * wv.Wn;                                // first owned control point (local to the handle's slice)
  const int nown = min(wv.Wn, ne.N - a), NJ = nown + 3;     // spans a - 3 .. a + nown - 1 reach the window
  const int SP = win_pieces(wv.Wn);                         // lane = (control point pl, coordinate d, piece s): an output ROW and a share of its detections (SP <= 5)
  const int s = lane % SP, pl = (lane / SP) / 3, d = (lane / SP) % 3;
  const bool role = pl < nown;
  // the window's time range; the first / last window of a time shard's slice also sees what lies beyond the slice (and flags it)
  const double inf = INFINITY;
  const int ga = a + ne.row0;
  const double T0 = (win == 0 && ne.row0 > 0) ? -inf : wv.tlo[max(ga - 3, 0)];
  const double T1 = (win == wv.nwin - 1 && ne.row0 + ne.N < wv.Ntot) ? inf : wv.thi[ga + nown];
  if (lane < 32) mk[lane] = 0ull;
#pragma unroll
  for (int t = 0; t < 3; ++t) S[(R::kZero + t) * kWinStr + lane] = 0.0;      // (columns 0..63; never written again)
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
  const int ncam = slot < dp.C ? (dp.C - slot + nslot - 1) / nslot : 0;       // cameras of this wavefront: <= 64 (the reduced camera system limits C * B to 1152)
  int p0v = 0, p1v = 0;
  if (lane < ncam) {
    const int c = wv.cam_perm[slot + nslot * lane];
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
    // ABSOLUTE detection indices (the host takes this path only with M < 2^31 and < 2^24 detections per camera: HipSchur::win_prepare)
    // and the camera's index in the top byte of the count: the camera loop below needs no load of cam_perm / det_off -- at the loop
    // head those were VECTOR loads (the loop stores, so the compiler cannot use the scalar cache) followed by s_waitcnt vmcnt(0): one
    // wait for the previous camera's stores to be acknowledged and one more for det_off[next camera] in front of every camera's first
    // prefetch, two dependent memory round trips per camera (found in the ISA, round 6)
    p0v = (int)dp.det_off[c] + p0; p1v = (c << 24) | (max(p0, p1) - p0);
  }
  auto cam_range = [&](int i, int& c, int& p0, int& p1) {   // (i wave-uniform)
    p0 = __builtin_amdgcn_readlane(p0v, i);
    const int pc = __builtin_amdgcn_readlane(p1v, i);
    c = (int)((unsigned)pc >> 24); p1 = p0 + (pc & 0xffffff);
  };

  // this lane's row of the band (blocks (pl, pl + w), w = 0..3, row d) and of the gradient: across the whole camera walk
  double Cw[4][3], gacc = 0.0;
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int d2 = 0; d2 < 3; ++d2) Cw[w][d2] = 0.0;

  // inputs of one batch (lane = detection), fetched one batch ahead: the detection and its first control point from the span table
  struct Inputs { double fr, vr, p, q; int g; };
  auto fetch = [&](int pos, int p1) {
    Inputs in{0.0, 0.0, 0.0, 0.0, -1};
    if (pos < p1) {                                         // (five independent loads: nothing here waits for the span)
      const long long i = pos;
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

  int p0 = 0, p1 = 0, cnext = 0;                            // (p0, p1: absolute detection indices)
  if (ncam > 0) cam_range(0, cnext, p0, p1);
  Inputs cur = ncam > 0 ? fetch(p0 + lane, p1) : Inputs{0.0, 0.0, 0.0, 0.0, -1};
  for (int ci = 0; ci < ncam; ++ci) {
    const int c = cnext;
    const CamState& cam = cams[c];                          // wave-uniform: scalar loads
    int np0 = 0, np1 = 0;                                   // the next camera's range
    if (ci + 1 < ncam) cam_range(ci + 1, cnext, np0, np1);
    d4v cacc[2][TI][TI];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TI; ++jj) cacc[h2][i][jj] = d4v{0.0, 0.0, 0.0, 0.0};
    double E[B];                                            // this lane's row (pl, d) of the camera's block of Et: across the camera's batches
#pragma unroll
    for (int k = 0; k < B; ++k) E[k] = 0.0;
    if (!(p0 < p1)) cur = (ci + 1 < ncam) ? fetch(np0 + lane, np1) : Inputs{0.0, 0.0, 0.0, 0.0, -1};   // (no batch: nothing was fetched ahead)

    for (int base = p0; base < p1; base += 64) {
      const bool last = base + 64 >= p1;
      // the next batch's inputs (this camera's, else the first of the next camera): in flight over this batch's arithmetic
      Inputs nxt;
      if (!last) nxt = fetch(base + 64 + lane, p1);
      else if (ci + 1 < ncam) nxt = fetch(np0 + lane, np1);
      else nxt = Inputs{0.0, 0.0, 0.0, 0.0, -1};
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
        atomicOr(&mk[key * SP + slot % SP], 1ull << slot);      // a span's detections are dealt to the pieces in turn
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
      // the masks of the four spans that reach this lane's control point (its piece of each): span pl + 3 - q touches it as ITS control point q
      unsigned long long mq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) mq[q] = role ? mk[(pl + 3 - q) * SP + s] : 0ull;
      // ---- camera block of the owned detections on the matrix cores: two independent accumulation chains (x rows, y rows) ----
      if (own != 0ull) {
        const int lr = lane & 15, lk = lane >> 4;
        int srcs[2][TI];
        bool rowok[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int row = 16 * i + lr;
          rowok[i] = row <= B;
#pragma unroll
          for (int xy = 0; xy < 2; ++xy) srcs[xy][i] = (row < B ? R::kJx + xy * B + row : kFx + xy) * kWinStr + lk;
        }
        // operands of the next two steps are in flight over this step's matrix instructions
        const int k_lo = (__ffsll((long long)own) - 1) >> 2, k_hi = ((63 - __clzll((long long)own)) >> 2) + 1;   // steps of four columns that hold owned ones
        double n1[2][TI], n2[2][TI];
#pragma unroll
        for (int xy = 0; xy < 2; ++xy)
#pragma unroll
          for (int i = 0; i < TI; ++i) { n1[xy][i] = S[srcs[xy][i] + 4 * k_lo]; n2[xy][i] = k_lo + 1 < k_hi ? S[srcs[xy][i] + 4 * (k_lo + 1)] : 0.0; }
        for (int kk = k_lo; kk < k_hi; ++kk) {
          const bool on = (own >> (4 * kk + lk)) & 1ull;
          double av[2][TI];
#pragma unroll
          for (int xy = 0; xy < 2; ++xy)
#pragma unroll
            for (int i = 0; i < TI; ++i) { av[xy][i] = (on && rowok[i]) ? n1[xy][i] : 0.0; n1[xy][i] = n2[xy][i]; }
          if (kk + 2 < k_hi) {
#pragma unroll
            for (int xy = 0; xy < 2; ++xy)
#pragma unroll
              for (int i = 0; i < TI; ++i) n2[xy][i] = S[srcs[xy][i] + 4 * (kk + 2)];
          }
#pragma unroll
          for (int xy = 0; xy < 2; ++xy)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
              for (int jj = 0; jj <= i; ++jj) cacc[xy][i][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[xy][i], av[xy][jj], cacc[xy][i][jj], 0, 0, 0);
        }
      }
      // ---- accumulate: lane = (control point pl, coordinate d, piece s).  The lane walks the staged detections of the four spans
      //      that reach its control point (its piece of them, in column = time order); a detection of span pl + 3 - q has the
      //      control point as ITS q-th, so the lane's spline slot is h[q] (gu_d, gv_d) and the band blocks (pl, pl + w) take
      //      h[q] h[q + w] (zero rows stand in for q + w > 3).  The sums of an output row never leave the lane: nothing to
      //      exchange, nothing to flush.  The values of the next detection are in flight while this one is added ----
      {
        unsigned long long mall = mq[0] | mq[1] | mq[2] | mq[3];
        // (two sets of plain locals and macros: value sets passed by reference end up in scratch memory)
#define MVUS_WIN_TAKE(P)                                                                                                   \
        do {                                                /* the lowest column of the walk: its values, one burst of reads */ \
          const int u_ = __ffsll((long long)mall) - 1;                                                                     \
          mall &= mall - 1ull;                                                                                             \
          const int q_ = (int)((mq[1] >> u_) & 1ull) + 2 * (int)((mq[2] >> u_) & 1ull) + 3 * (int)((mq[3] >> u_) & 1ull);  \
          const double* su_ = S + u_;                                                                                      \
          const double* sh_ = su_ + (R::kH + q_) * kWinStr;                                                                \
          _Pragma("unroll") for (int t = 0; t < 4; ++t) P##hb[t] = sh_[t * kWinStr];                                       \
          _Pragma("unroll") for (int e = 0; e < 3; ++e) { P##gu[e] = su_[(R::kGu + e) * kWinStr]; P##gv[e] = su_[(R::kGv + e) * kWinStr]; } \
          _Pragma("unroll") for (int k = 0; k < B; ++k) { P##jx[k] = su_[(R::kJx + k) * kWinStr]; P##jy[k] = su_[(R::kJy + k) * kWinStr]; } \
          P##fx = su_[kFx * kWinStr]; P##fy = su_[kFy * kWinStr];                                                          \
        } while (0)
#define MVUS_WIN_ADD(P)                                                                                                    \
        do {                                                                                                               \
          const double gud_ = d == 0 ? P##gu[0] : (d == 1 ? P##gu[1] : P##gu[2]);                                          \
          const double gvd_ = d == 0 ? P##gv[0] : (d == 1 ? P##gv[1] : P##gv[2]);                                          \
          const double hu_ = P##hb[0] * gud_, hv_ = P##hb[0] * gvd_;      /* this control point's spline slot (x row, y row) */ \
          _Pragma("unroll") for (int k = 0; k < B; ++k) E[k] += hu_ * P##jx[k] + hv_ * P##jy[k];                           \
          gacc += hu_ * P##fx + hv_ * P##fy;                                                                               \
          /* band row: h_q h_{q+w} (gu_d gu + gv_d gv) -- the 1 x 3 row m of the point block once per detection, then one  \
             product and three multiply-adds per block (22 operations instead of the 32 of (h_q g)(h_{q+w} g) per block) */ \
          double m_[3];                                                                                                    \
          _Pragma("unroll") for (int d2 = 0; d2 < 3; ++d2) m_[d2] = gud_ * P##gu[d2] + gvd_ * P##gv[d2];                   \
          _Pragma("unroll") for (int w = 0; w < 4; ++w) {                                                                  \
            const double hh_ = P##hb[0] * P##hb[w];                                                                        \
            _Pragma("unroll") for (int d2 = 0; d2 < 3; ++d2) Cw[w][d2] += hh_ * m_[d2];                                    \
          }                                                                                                                \
        } while (0)
        if (mall != 0ull) {                                 // two value sets in turn: the reads of one are in flight over the adds of the other
          double a_hb[4], a_gu[3], a_gv[3], a_jx[B], a_jy[B], a_fx, a_fy;
          double b_hb[4], b_gu[3], b_gv[3], b_jx[B], b_jy[B], b_fx, b_fy;
          MVUS_WIN_TAKE(a_);
          while (true) {
            bool more = mall != 0ull;
            if (more) MVUS_WIN_TAKE(b_);
            MVUS_WIN_ADD(a_);
            if (!more) break;
            more = mall != 0ull;
            if (more) MVUS_WIN_TAKE(a_);
            MVUS_WIN_ADD(b_);
            if (!more) break;
          }
        }
#undef MVUS_WIN_TAKE
#undef MVUS_WIN_ADD
      }
      win_wave_sync();                                      // the staging region is rewritten by the next batch
      if (lane < 32) mk[lane] = 0ull;
      cur = nxt;
    }

    // ---- the camera's block of Et: the rows' pieces added in order, every row stored by its lane (nine plain stores into the
    //      camera's contiguous nown x 3 x B block; the L2 merges the rows' lines) ----
    {
      double* er = ne.Et + ((long long)c * ne.N3 + 3 * a) * B;
      if (SP > 1) {
#pragma unroll
        for (int k = 0; k < B; ++k) {
          double t = E[k];
          for (int sp = 1; sp < SP; ++sp) t += __shfl(E[k], (lane - s + sp) & 63, 64);    // (lane - s: piece 0 of the row)
          E[k] = t;
        }
      }
      if (role && s == 0) {
        // (the row's offset is recomputed here, from a lane index the optimiser cannot see through: hoisted out of the camera loop, the
        // 64-bit offsets of this block's and the next block's stores are loop invariants that cost ten registers the loop does not have --
        // they were spilled to scratch, and a scratch reload in front of the stores waits for every store before it)
        int lh = lane;
        asm volatile("" : "+v"(lh));
        double* row = er + (lh / SP) * B;                   // lane / SP = pl * 3 + d
#pragma unroll
        for (int k = 0; k < B; ++k) row[k] = E[k];
      }
    }
    // the camera block's partial (C/D layout of the 16x16 tile: row = (lane >> 4) + 4 reg, column = lane & 15)
    {
      double* mine = wv.Apart + ((long long)win * dp.C + c) * PSZ;
      int lh = lane;
      asm volatile("" : "+v"(lh));                          // (as above: the entries' offsets are not to live across the loop)
      const int lr = lh & 15, lk = lh >> 4;
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

  // ---- band and gradient: the rows' pieces, then the four wavefronts, added in order; every entry stored once ----
  if (SP > 1) {
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int d2 = 0; d2 < 3; ++d2) {
        double t = Cw[w][d2];
        for (int sp = 1; sp < SP; ++sp) t += __shfl(Cw[w][d2], (lane - s + sp) & 63, 64);
        Cw[w][d2] = t;
      }
    double t = gacc;
    for (int sp = 1; sp < SP; ++sp) t += __shfl(gacc, (lane - s + sp) & 63, 64);
    gacc = t;
  }
  constexpr int kRow = 13;                                  // a row's sums: 4 x 3 band entries + the gradient
  if (role && s == 0) {
    double* row = S + (pl * 3 + d) * kRow;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int d2 = 0; d2 < 3; ++d2) row[w * 3 + d2] = Cw[w][d2];
    row[12] = gacc;
  }
  __syncthreads();
  const int per = 3 + ne.W * 9;
  // (several camera groups: the group's partial rows go to its slice of band_part -- gs then Cb, the layout of the blocks themselves --
  // and k_cam_block_sum adds the groups in order)
  double* gs_out = wv.G > 1 ? wv.band_part + (long long)grp * ((long long)ne.N * per) : ne.gs;
  double* Cb_out = wv.G > 1 ? gs_out + 3LL * ne.N : ne.Cb;
  for (int e = threadIdx.x; e < nown * per; e += kWinThreads) {
    const int p = e / per, r = e - p * per;
    double acc = 0.0;
    if (r < 3) {
      for (int v = 0; v < kWinWaves; ++v) acc += win_lds[v * WAVE + (p * 3 + r) * kRow + 12];
      gs_out[3 * (a + p) + r] = acc;
    } else {
      const int w = (r - 3) / 9, dd = (r - 3) - 9 * w;        // block (p, p + w), entry (row dd / 3, column dd % 3)
      if (w < 4)
        for (int v = 0; v < kWinWaves; ++v) acc += win_lds[v * WAVE + (p * 3 + dd / 3) * kRow + w * 3 + dd % 3];
      Cb_out[((long long)(a + p) * ne.W) * 9 + (r - 3)] = acc;       // (w >= 4: the motion rows' blocks start from zero)
    }
  }
}

// A[c] (both triangles) and gc[c] from the per-(window, camera) partial blocks: one workgroup per camera, the windows added in
// index order (four interleaved chains per entry and group, the groups combined in order) -- plain stores, nothing to clear.
template <int B>
__global__ __launch_bounds__(1024) void k_cam_block_sum(int C, int nwin, const double* __restrict__ Apart, NEView ne, int G = 1,
                                                        const double* __restrict__ band_part = nullptr, double* __restrict__ zero = nullptr,
                                                        long long zero_len = 0) {
  constexpr int PSZ = (B + 1) * (B + 2) / 2, kLanes = PSZ <= 64 ? 64 : 256, kGroups = 1024 / kLanes;
  __shared__ double part[kGroups][kLanes];
  // (a time shard: the part of the packed head that is SUMMED over the ranks without being written in full by this rank -- the other
  // cuts' halo blocks, diag(H) and g of the columns outside the slice -- starts from zero; cleared here instead of by a launch of its own)
  for (long long i = blockIdx.x * 1024LL + threadIdx.x; i < zero_len; i += gridDim.x * 1024LL) zero[i] = 0.0;
  if ((int)blockIdx.x >= C) {                               // workgroups past the cameras (G > 1): the camera groups' band rows and gradients, added in group order
    const long long len = (long long)ne.N * (3 + ne.W * 9), i = ((long long)blockIdx.x - C) * 1024 + threadIdx.x;
    if (i < len) {
      double t = band_part[i];
      for (int g = 1; g < G; ++g) t += band_part[(long long)g * len + i];
      if (i < 3LL * ne.N) ne.gs[i] = t; else ne.Cb[i - 3LL * ne.N] = t;
    }
    return;
  }
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
