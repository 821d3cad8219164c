// Micro-benchmark: how fast can a kernel with the Jacobian kernel's store pattern write to HBM on gfx950?
// Slot-major layout J[k*M + i], 42 slots, M observations; one lane per observation (8-B stores, 512 B per wave store)
// vs one lane per observation pair (16-B stores, 1 KiB per wave store).  Outputs rotate over NROT buffers (> Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NS2 = 42;
__global__ __launch_bounds__(256) void store8(double* __restrict__ J, long long M, double seed) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= M) return;
  double v = seed + (double)i;
#pragma unroll
  for (int k = 0; k < NS2; ++k) { v = v * 1.0000001 + 0.5; J[(long long)k * M + i] = v; }
}
// the same 8-byte pattern with the workgroups of one XCD (blockIdx % 8) working on runs of RUN consecutive 2 KB tiles
template <int RUN>
__global__ __launch_bounds__(256) void store8_xcd(double* __restrict__ J, long long M, double seed) {
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const long long tile = ((long long)(q / RUN) * 8 + xcd) * RUN + q % RUN;
  const long long i = tile * 256ll + threadIdx.x;
  if (i >= M) return;
  double v = seed + (double)i;
#pragma unroll
  for (int k = 0; k < NS2; ++k) { v = v * 1.0000001 + 0.5; J[(long long)k * M + i] = v; }
}
// chunk-major layout: the 42 slot rows of one 256-observation chunk are adjacent (one contiguous 86 KB block per workgroup)
template <int RUN>
__global__ __launch_bounds__(256) void store8_tile(double* __restrict__ J, long long M, double seed) {
  long long tile = blockIdx.x;
  if (RUN > 0) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; tile = ((long long)(q / RUN) * 8 + xcd) * RUN + q % RUN; }
  const long long i = tile * 256ll + threadIdx.x;
  if (i >= M) return;
  double v = seed + (double)i;
  double* base = J + tile * (long long)(NS2 * 256) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < NS2; ++k) { v = v * 1.0000001 + 0.5; base[k * 256] = v; }
}
__global__ __launch_bounds__(256) void store16(double* __restrict__ J, long long M, double seed) {
  const long long i = 2 * (blockIdx.x * 256ll + threadIdx.x);
  if (i + 1 >= M) return;
  double v = seed + (double)i;
#pragma unroll
  for (int k = 0; k < NS2; ++k) {
    v = v * 1.0000001 + 0.5;
    double2 w = {v, v + 1.0};
    *reinterpret_cast<double2*>(&J[(long long)k * M + i]) = w;
  }
}
// pair exchange by DPP as the real kernel would do it: each lane computes 42 values for ITS observation, then even/odd lanes swap halves
__device__ __forceinline__ double dpp_swap1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(256) void store16x(double* __restrict__ J, long long M, double seed) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= M) return;
  double v[NS2];
  double a = seed + (double)i;
#pragma unroll
  for (int k = 0; k < NS2; ++k) { a = a * 1.0000001 + 0.5; v[k] = a; }
  const bool odd = threadIdx.x & 1;
#pragma unroll
  for (int k = 0; k < NS2; k += 2) {
    const double send = odd ? v[k] : v[k + 1];
    const double recv = dpp_swap1(send);
    double2 w;
    w.x = odd ? recv : v[k];
    w.y = odd ? v[k + 1] : recv;
    const long long slot = odd ? k + 1 : k;
    *reinterpret_cast<double2*>(&J[slot * M + (i & ~1ll)]) = w;
  }
}
int main(int argc, char** argv) {
  const long long M = argc > 1 ? atoll(argv[1]) : 504400;   // even
  const int NROT = 6, reps = 60;
  std::vector<double*> bufs(NROT);
  for (auto& b : bufs) CK(hipMalloc(&b, sizeof(double) * NS2 * (M + 65536)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)NS2 * M * 8;
  for (int variant = 0; variant < 8; ++variant) {
    for (int rot = 0; rot < 2; ++rot) {
      const int nb = rot ? NROT : 1;
      auto launch = [&](int it) {
        double* J = bufs[it % nb];
        if (variant == 0) hipLaunchKernelGGL(store8, dim3((M + 255) / 256), dim3(256), 0, 0, J, M, 1.0);
        else if (variant == 1) hipLaunchKernelGGL(store16, dim3((M / 2 + 255) / 256), dim3(256), 0, 0, J, M, 1.0);
        else if (variant == 2) hipLaunchKernelGGL(store16x, dim3((M + 255) / 256), dim3(256), 0, 0, J, M, 1.0);
        else {
          const int run = variant == 3 ? 8 : variant == 4 ? 64 : 256;
          const long long tiles = (M + 255) / 256, per = 8 * run, grid = (tiles + per - 1) / per * per;
          if (variant == 3) hipLaunchKernelGGL(store8_xcd<8>, dim3(grid), dim3(256), 0, 0, J, M, 1.0);
          else if (variant == 4) hipLaunchKernelGGL(store8_xcd<64>, dim3(grid), dim3(256), 0, 0, J, M, 1.0);
          else if (variant == 5) hipLaunchKernelGGL(store8_xcd<256>, dim3(grid), dim3(256), 0, 0, J, M, 1.0);
          else if (variant == 6) hipLaunchKernelGGL(store8_tile<0>, dim3(tiles), dim3(256), 0, 0, J, M, 1.0);
          else { const long long g64 = (tiles + 511) / 512 * 512; hipLaunchKernelGGL(store8_tile<64>, dim3(g64), dim3(256), 0, 0, J, M, 1.0); }
        }
      };
      for (int it = 0; it < NROT; ++it) launch(it);
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < reps; ++it) launch(it);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = 1e3 * ms / reps;
      printf("%-8s %s: %.1f us per launch, %.0f GB/s\n", variant == 0 ? "store8" : variant == 1 ? "store16" : variant == 2 ? "store16x" : variant == 3 ? "xcd8" : variant == 4 ? "xcd64" : variant == 5 ? "xcd256" : variant == 6 ? "tile" : "tile+xcd64", rot ? "rotating 6 x 169 MB" : "one buffer        ", us, bytes / us * 1e-3);
    }
  }
  return 0;
}
