// fp64 issue rates of one gfx950 SIMD, in shader cycles per wave64 instruction (clock64 inside one wavefront; 1, 2 or 4 wavefronts per
// SIMD): independent and dependent v_fma_f64, v_mov_b32_dpp pairs feeding v_fma_f64 (the 16-column LDL of ba_rcs.hip.h),
// v_mfma_f64_16x16x4 with independent accumulators and as one dependent chain.  hipcc --offload-arch=gfx950 -O3 fp64_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
constexpr int kIter = 512;
__global__ void k_rates(double* out, long long* cyc, double seed) {
  double a[8], x = seed + threadIdx.x * 1e-9;
  for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1);
  long long t0, t1;
  // 1: eight independent FMA chains
  t0 = clock64();
  for (int it = 0; it < kIter; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], x, 1e-9);
  t1 = clock64();
  long long c_indep = t1 - t0;
  // 2: one dependent chain
  double b = seed;
  t0 = clock64();
  for (int it = 0; it < kIter; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) b = __builtin_fma(b, x, 1e-9);
  t1 = clock64();
  long long c_dep = t1 - t0;
  // 3: pair = two DPP moves (row broadcast of a 64-bit value) + two FMAs that use it, eight independent pairs per iteration
  double m = seed;
  t0 = clock64();
  for (int it = 0; it < kIter; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int lo = __builtin_amdgcn_mov_dpp(__double2loint(a[i]), 0x155, 0xf, 0xf, true);
      const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(a[i]), 0x155, 0xf, 0xf, true);
      const double mm = __hiloint2double(hi, lo);
      a[(i + 1) & 7] = __builtin_fma(mm, x, a[(i + 1) & 7]);
      m = __builtin_fma(mm, b, m);
    }
  t1 = clock64();
  long long c_pair = t1 - t0;
  // 4: MFMA, four independent accumulators
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  t0 = clock64();
  for (int it = 0; it < kIter / 4; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], a[i + 4], acc[i], 0, 0, 0);
  t1 = clock64();
  long long c_mfma_indep = t1 - t0;
  // 5: MFMA, one dependent chain
  d4 accd = {0, 0, 0, 0};
  t0 = clock64();
  for (int it = 0; it < kIter / 4; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i) accd = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], a[i + 4], accd, 0, 0, 0);
  t1 = clock64();
  long long c_mfma_dep = t1 - t0;
  double s = b + m + accd[0] + accd[3];
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += acc[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    long long* c = cyc + (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 5;
    c[0] = c_indep; c[1] = c_dep; c[2] = c_pair; c[3] = c_mfma_indep; c[4] = c_mfma_dep;
  }
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 16 * 5 * 8);
  for (int threads : {256, 512, 1024}) {
    hipLaunchKernelGGL(k_rates, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0000001);
    hipDeviceSynchronize();
    long long h[16 * 5];
    hipMemcpy(h, cyc, sizeof(long long) * (threads / 64) * 5, hipMemcpyDeviceToHost);
    const double n_fma = kIter * 8.0, n_mfma = kIter;
    printf("%d wavefront(s) per SIMD (one workgroup of %d threads), cycles per wave64 instruction seen by wavefront 0:\n", threads / 256, threads);
    printf("  v_fma_f64, 8 independent chains   %.1f\n", h[0] / n_fma);
    printf("  v_fma_f64, one dependent chain    %.1f\n", h[1] / n_fma);
    printf("  2 x v_mov_b32_dpp + 2 x v_fma_f64 %.1f per group of four\n", h[2] / n_fma);
    printf("  v_mfma_f64_16x16x4, 4 accumulators %.1f\n", h[3] / n_mfma);
    printf("  v_mfma_f64_16x16x4, dependent      %.1f\n", h[4] / n_mfma);
  }
  return 0;
}
