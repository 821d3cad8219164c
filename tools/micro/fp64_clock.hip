// What clock do the fp64 pipes run at?  tools/micro/fp64_coissue.hip counts s_memtime cycles on ONE CU; this one times, with HIP events,
// a launch of G workgroups x 4 wavefronts (one per SIMD) that each issue N v_mfma_f64_16x16x4 (four accumulators: issue bound, 64 cycles
// each) or N x 8 independent v_fma_f64 -- for G = 1 (one CU busy) and G = 256 / 512 (the whole device busy).  ns per instruction and
// wavefront -> the clock the pipe really ran at (64 cycles per matrix instruction; 5.37 per v_fma_f64 with one wavefront per SIMD).
// build: hipcc -O3 --offload-arch=gfx950 -o fp64_clock tools/micro/fp64_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
// (__launch_bounds__(1024): at most 128 registers per wavefront, so the accumulators stay in VGPRs -- with a 512-register budget the compiler
// parks them in AGPRs between iterations and the loop measures v_accvgpr moves and pipeline drains, not the matrix cores)
__global__ __launch_bounds__(1024) void k_mfma(int n, double* sink, unsigned long long* ticks) {
  d4 a[8];
  for (int k = 0; k < 8; ++k) a[k] = d4{0, 0, 0, 0};
  const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[u & 7], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0; for (int k = 0; k < 8; ++k) s += a[k][0] + a[k][3];
  if (s == 12345.678) sink[0] = 1.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ __launch_bounds__(1024) void k_fma(int n, double* sink, unsigned long long* ticks) {
  double a[8];
  for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 1e-9 + k;
  const double x = 1.0000001, y = 1e-9;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], x, y);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0; for (int k = 0; k < 8; ++k) s += a[k];
  if (s == 12345.678) sink[0] = 1.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
  double* sink; unsigned long long* ticks;
  (void)hipMalloc(&sink, 8); (void)hipMalloc(&ticks, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nm = 1 << 16, nf = 1 << 16;
  printf("# G workgroups x W wavefronts per SIMD; HIP-event time of the launch; ticks = s_memtime of wavefront 0 over its loop\n");
  for (int kind = 0; kind < 2; ++kind)
    for (int W : {1, 2, 4})
      for (int G : {1, 256}) {
        float ms = 0; unsigned long long t = 0;
        for (int rep = 0; rep < 3; ++rep) {
          (void)hipEventRecord(e0, 0);
          if (kind == 0) hipLaunchKernelGGL(k_mfma, dim3(G), dim3(256 * W), 0, 0, nm, sink, ticks);
          else hipLaunchKernelGGL(k_fma, dim3(G), dim3(256 * W), 0, 0, nf, sink, ticks);
          (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
          (void)hipEventElapsedTime(&ms, e0, e1);
          (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        }
        const double ninstr = (kind == 0 ? (double)nm : 8.0 * nf) * W;       // per SIMD
        const double per = ms * 1e6 / ninstr;
        printf("%s G=%3d W=%d: %8.3f ms, %7.3f ns per instruction and SIMD (%5.1f TFLOP/s if all 1024 SIMDs ran like this);  wavefront 0: %.2f ticks per instruction and SIMD, %.0f MHz\n",
               kind == 0 ? "v_mfma_f64_16x16x4" : "v_fma_f64         ", G, W, ms, per, (kind == 0 ? 2048.0 : 128.0) / per * 1024 * 1e-3, (double)t / ninstr, t / (ms * 1e3));
      }
  return 0;
}
