// Issue rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on this GPU (what the fp64 products of the solve can reach).
// hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(int iters, double* out) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma(int iters, double* out) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  const double a = threadIdx.x * 1e-9 + 1.0, b = blockIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0.0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static float timed(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  double* out; (void)hipMalloc(&out, 4096 * 256 * 8);
  const int iters = 20000;
  for (int wgs : {256, 512, 1024}) {
    float ms = timed([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(wgs), dim3(256), 0, 0, iters, out); });
    double n = (double)wgs * 4 * iters * 8;                         // matrix instructions
    std::printf("mfma_f64_16x16x4, %4d workgroups of 4 wavefronts: %.1f TFLOP/s, %.1f cycles per instruction and SIMD at 2.4 GHz\n", wgs,
                n * 2048 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (n / 1024.0 / (wgs >= 256 ? 1 : 1)) );
    ms = timed([&] { hipLaunchKernelGGL(k_fma, dim3(wgs), dim3(256), 0, 0, iters, out); });
    n = (double)wgs * 4 * iters * 16;
    std::printf("v_fma_f64,        %4d workgroups of 4 wavefronts: %.1f TFLOP/s\n", wgs, n * 128 / ms * 1e-9);
  }
  return 0;
}
