// Cycle cost of the 16x16 L D L^T + inverse factor that sits on the critical path of the block Gauss-Jordan (ldl_inv16 in
// ba_schur_hip.hip.h), one lone wavefront, three formulations; results checked against each other and against the host.
// hipcc --offload-arch=gfx950 -O3 -o ldl16_bench ldl16_bench.hip && ./ldl16_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long tick() { long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
// A: the shipped form -- lanes 0..15 rows of A, lanes 16..31 rows of I, multipliers by v_readlane (through SGPRs)
__device__ __forceinline__ void ldl_readlane(double (&a)[16], double (&rd)[16]) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    double d = bcast_lane(a[k], k);
    d = d > 0.0 ? d : 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = r * (2.0 - d * r);
    r = r * (2.0 - d * r);
    rd[k] = r;
    const double t = a[k] * r;
#pragma unroll
    for (int j = k + 1; j < 16; ++j) a[j] -= t * bcast_lane(a[k], j);
  }
}
template <int J> __device__ __forceinline__ double row_bcast(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, false); }
template <int K, int J> struct ColOps {
  static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16], double ta, double tx) {
    const double m = row_bcast<J>(a[K]);            // A[J][K] from lane J of this row of 16 lanes
    a[J] -= ta * m;
    x[J] -= tx * m;
    ColOps<K, J + 1>::run(a, x, ta, tx);
  }
};
template <int K> struct ColOps<K, 16> { static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double, double) {} };
template <int K> struct Cols {
  static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16], double (&rd)[16]) {
    double d = row_bcast<K>(a[K]);
    d = d > 0.0 ? d : 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = r * (2.0 - d * r);
    r = r * (2.0 - d * r);
    rd[K] = r;
    ColOps<K, K + 1>::run(a, x, a[K] * r, x[K] * r);
    Cols<K + 1>::run(a, x, rd);
  }
};
template <> struct Cols<16> { static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double (&)[16]) {} };
// B: 16 lanes, lane i holds row i of A (a) and row i of I (x); multipliers by DPP row_newbcast (v_mov_b64_dpp), no SGPR
__device__ __forceinline__ void ldl_dpp(double (&a)[16], double (&x)[16], double (&rd)[16]) { Cols<0>::run(a, x, rd); }

// out: [variant][ X (16x16 row-major: X[k][c]) | rd (16) ], cycles[variant]
__global__ __launch_bounds__(64) void k_bench(const double* __restrict__ A, double* __restrict__ out, long long* __restrict__ cycles, int reps) {
  const int lane = threadIdx.x, row = lane & 15;
  double a[16], x[16], rd[16];
  long long best = 1ll << 62;
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double t = A[row * 16 + k]; a[k] = lane < 16 ? (k <= row ? t : 0.0) : (lane < 32 && k == row ? 1.0 : 0.0); }
    const long long t0 = tick();
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(a[k]));
    ldl_readlane(a, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(a[k]), "+v"(rd[k]));
    const long long t1 = tick();
    best = t1 - t0 < best ? t1 - t0 : best;
  }
  if (lane >= 16 && lane < 32) for (int k = 0; k < 16; ++k) out[k * 16 + row] = a[k];          // lane 16+c holds X[k][c]
  if (lane == 0) { for (int k = 0; k < 16; ++k) out[256 + k] = rd[k]; cycles[0] = best; }
  best = 1ll << 62;
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double t = A[row * 16 + k]; a[k] = k <= row ? t : 0.0; x[k] = k == row ? 1.0 : 0.0; }
    const long long t0 = tick();
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(a[k]), "+v"(x[k]));
    ldl_dpp(a, x, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(a[k]), "+v"(x[k]), "+v"(rd[k]));
    const long long t1 = tick();
    best = t1 - t0 < best ? t1 - t0 : best;
  }
  if (lane < 16) for (int k = 0; k < 16; ++k) out[272 + k * 16 + row] = x[k];                    // lane c holds X[k][c]
  if (lane == 0) { for (int k = 0; k < 16; ++k) out[272 + 256 + k] = rd[k]; cycles[1] = best; }
}

int main() {
  std::vector<double> B(256), A(256, 0.0);
  unsigned s = 12345;
  for (auto& v : B) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double t = i == j ? 4.0 : 0.0; for (int k = 0; k < 16; ++k) t += B[i * 16 + k] * B[j * 16 + k]; A[i * 16 + j] = t; }
  double *dA, *dout; long long* dc;
  hipMalloc(&dA, 256 * 8); hipMalloc(&dout, 2 * 272 * 8); hipMalloc(&dc, 16);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, dA, dout, dc, 20);
  std::vector<double> out(2 * 272); long long c[2];
  hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
  // host check: A^-1 = X^T diag(rd) X
  for (int v = 0; v < 2; ++v) {
    const double* X = out.data() + 272 * v; const double* rd = X + 256;
    double worst = 0.0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double t = 0.0;                                  // (A * Ainv)[i][j]
      for (int m = 0; m < 16; ++m) { double ainv = 0.0; for (int k = 0; k < 16; ++k) ainv += X[k * 16 + m] * rd[k] * X[k * 16 + j]; t += A[i * 16 + m] * ainv; }
      worst = fmax(worst, fabs(t - (i == j ? 1.0 : 0.0)));
    }
    printf("%s: %lld cycles (best of 20), |A A^-1 - I| max %.2e\n", v == 0 ? "readlane (shipped)" : "dpp row_newbcast, 16 lanes", c[v], worst);
  }
  double diff = 0.0; for (int i = 0; i < 272; ++i) diff = fmax(diff, fabs(out[i] - out[272 + i]));
  printf("max difference between the two: %.2e\n", diff);
  return 0;
}
