// Do the fp64 vector unit and the fp64 matrix core of a gfx950 SIMD run at the same time?  (Round-5 review, "What's weak" #8: the
// round-5 table -- fp64_rates.hip -- printed the cycles ONE wavefront saw and never summed the work of all wavefronts of a SIMD.)
//
// One workgroup of 4 W wavefronts (W per SIMD; the SIMD of every wavefront is read back from HW_ID, not assumed).  Every wavefront is
// given a role: F = a loop of independent v_fma_f64 (8 accumulators), M = a loop of v_mfma_f64_16x16x4 (4 accumulators), D = one
// dependent v_fma_f64 chain, or idle.  Each wavefront records s_memtime at its start and end.  Reported per configuration:
//   - per role: cycles per instruction as seen by one wavefront (the round-5 figure),
//   - per SIMD: AGGREGATE instructions of each kind / (last end - first start) over the wavefronts of that SIMD -- what the SIMD delivered,
//   - for mixed configurations: elapsed against the same wavefronts' work run alone (sum = the units are shared, max = they overlap).
// hipcc --offload-arch=gfx950 -O3 -o fp64_coissue fp64_coissue.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
using d4 = __attribute__((ext_vector_type(4))) double;
constexpr int kFma = 8192;       // v_fma_f64 per F / D wavefront
constexpr int kMfma = 1024;      // v_mfma_f64_16x16x4 per M wavefront
struct Rec { long long t0, t1; unsigned hwid; int role; };

__global__ __launch_bounds__(1024) void k_roles(const int* __restrict__ roles, Rec* __restrict__ rec, double* __restrict__ sink, double seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = roles[wave];
  const double x = seed + lane * 1e-9;
  double out = 0.0;
  __syncthreads();                                   // all wavefronts start together
  const long long t0 = __builtin_readcyclecounter();
  if (role == 1) {                                   // F: eight independent chains
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1);
    for (int it = 0; it < kFma / 8; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], x, 1e-9);
    for (int i = 0; i < 8; ++i) out += a[i];
  } else if (role == 2) {                            // M: four independent accumulators
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const double p = x, q = x * 0.5;
    for (int it = 0; it < kMfma / 4; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(p, q, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i) out += acc[i][0] + acc[i][3];
  } else if (role == 3) {                            // D: one dependent chain
    double b = seed;
    for (int it = 0; it < kFma / 8; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) b = __builtin_fma(b, x, 1e-9);
    out = b;
  }
  const long long t1 = __builtin_readcyclecounter();
  sink[threadIdx.x] = out;
  if (lane == 0) rec[wave] = Rec{t0, t1, __builtin_amdgcn_s_getreg((31 << 11) | 4), role};   // HW_REG_HW_ID: simd_id = bits 5:4
}

struct Out { double elapsed; };
static int* d_roles; static Rec* d_rec; static double* d_sink;

static double run(const char* name, const std::vector<int>& roles) {        // returns elapsed cycles of the busiest SIMD
  const int nw = (int)roles.size();
  hipMemcpy(d_roles, roles.data(), sizeof(int) * nw, hipMemcpyHostToDevice);
  double best = 1e30;
  std::vector<Rec> r(nw), keep;
  for (int rep = 0; rep < 5; ++rep) {                 // (the first launch pays for the code fetch)
    hipLaunchKernelGGL(k_roles, dim3(1), dim3(64 * nw), 0, 0, d_roles, d_rec, d_sink, 1.0000001);
    hipDeviceSynchronize();
    hipMemcpy(r.data(), d_rec, sizeof(Rec) * nw, hipMemcpyDeviceToHost);
    long long lo = r[0].t0, hi = r[0].t1;
    for (auto& e : r) if (e.role) { lo = std::min(lo, e.t0); hi = std::max(hi, e.t1); }
    if ((double)(hi - lo) < best) { best = (double)(hi - lo); keep = r; }
  }
  printf("%s\n", name);
  for (int simd = 0; simd < 4; ++simd) {
    long long lo = 0, hi = 0; int nf = 0, nm = 0, nd = 0; bool any = false;
    double cf = 0, cm = 0, cd = 0;
    for (auto& e : keep) {
      if (!e.role || (int)((e.hwid >> 4) & 3) != simd) continue;
      if (!any) { lo = e.t0; hi = e.t1; any = true; }
      lo = std::min(lo, e.t0); hi = std::max(hi, e.t1);
      const double c = (double)(e.t1 - e.t0);
      if (e.role == 1) { ++nf; cf += c / kFma; } else if (e.role == 2) { ++nm; cm += c / kMfma; } else { ++nd; cd += c / kFma; }
    }
    if (!any) continue;
    const double el = (double)(hi - lo);
    printf("  SIMD %d: %d F + %d M + %d D wavefronts, %8.0f cycles;", simd, nf, nm, nd, el);
    if (nf) printf("  F: %.2f cyc/instr per wavefront, aggregate %.2f cycles per v_fma_f64 (%.1f flop/cycle)", cf / nf, el / (nf * (double)kFma), nf * (double)kFma * 128 / el);
    if (nd) printf("  D: %.2f cyc/instr per wavefront, aggregate %.2f cycles per v_fma_f64", cd / nd, el / (nd * (double)kFma));
    if (nm) printf("  M: %.2f cyc/instr per wavefront, aggregate %.2f cycles per v_mfma_f64_16x16x4 (%.1f flop/cycle)", cm / nm, el / (nm * (double)kMfma), nm * (double)kMfma * 2048 / el);
    printf("\n");
  }
  return best;
}

int main() {
  hipMalloc(&d_roles, 16 * sizeof(int)); hipMalloc(&d_rec, 16 * sizeof(Rec)); hipMalloc(&d_sink, 1024 * 8);
  auto mk = [](std::initializer_list<int> per_simd_roles) {     // roles of the wavefronts of ONE SIMD, replicated over the four (wave w -> SIMD w % 4)
    std::vector<int> v;
    for (int r : per_simd_roles) for (int s = 0; s < 4; ++s) v.push_back(r);
    return v;
  };
  printf("# fp64 vector (F: 8 independent v_fma_f64 chains, D: one dependent chain) and matrix (M: v_mfma_f64_16x16x4, 4 accumulators) work on one CU of gfx950;\n");
  printf("# %d v_fma_f64 per F/D wavefront, %d v_mfma per M wavefront; elapsed = last end - first start of the wavefronts of a SIMD (s_memtime)\n", kFma, kMfma);
  const double f1 = run("F x1 per SIMD", mk({1}));
  const double f2 = run("F x2 per SIMD", mk({1, 1}));
  const double f4 = run("F x4 per SIMD", mk({1, 1, 1, 1}));
  const double d1 = run("D x1 per SIMD", mk({3}));
  const double d2 = run("D x2 per SIMD", mk({3, 3}));
  const double d4_ = run("D x4 per SIMD", mk({3, 3, 3, 3}));
  const double m1 = run("M x1 per SIMD", mk({2}));
  const double m2 = run("M x2 per SIMD", mk({2, 2}));
  const double m4 = run("M x4 per SIMD", mk({2, 2, 2, 2}));
  const double fm = run("F x1 + M x1 per SIMD (different wavefronts)", mk({1, 2}));
  const double ffmm = run("F x2 + M x2 per SIMD", mk({1, 1, 2, 2}));
  const double fffm = run("F x3 + M x1 per SIMD", mk({1, 1, 1, 2}));
  const double dm = run("D x1 + M x1 per SIMD", mk({3, 2}));
  printf("# summary (cycles):  F1 %.0f  F2 %.0f  F4 %.0f | D1 %.0f D2 %.0f D4 %.0f | M1 %.0f  M2 %.0f  M4 %.0f\n", f1, f2, f4, d1, d2, d4_, m1, m2, m4);
  printf("# mixed F1+M1: %.0f  (alone: F1 %.0f, M1 %.0f; sum %.0f, max %.0f) -> %s\n", fm, f1, m1, f1 + m1, std::max(f1, m1),
         fm < 0.75 * (f1 + m1) ? "the two units OVERLAP" : "the two kinds of work add up: ONE fp64 pipe");
  printf("# mixed F2+M2: %.0f  (alone: F2 %.0f, M2 %.0f; sum %.0f)\n", ffmm, f2, m2, f2 + m2);
  printf("# mixed F3+M1: %.0f  (alone: F3 ~%.0f, M1 %.0f)\n", fffm, 0.75 * f4, m1);
  printf("# mixed D1+M1: %.0f  (alone: D1 %.0f, M1 %.0f)\n", dm, d1, m1);
  return 0;
}
