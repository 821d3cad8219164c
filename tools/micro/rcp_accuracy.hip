// Relative error of v_rcp_f64 / v_rsq_f64 raw and after one and two Newton steps (decides how many steps the small
// dense factorisations need).  hipcc --offload-arch=gfx950 -O3 -o rcp_accuracy rcp_accuracy.hip && ./rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(int n, const double* x, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  out[i] = r;
  r = r * (2.0 - d * r);
  out[n + i] = r;
  r = r * (2.0 - d * r);
  out[2 * n + i] = r;
  double q = __builtin_amdgcn_rsq(d);
  out[3 * n + i] = q;
  q = q * (1.5 - 0.5 * d * q * q);
  out[4 * n + i] = q;
  q = q * (1.5 - 0.5 * d * q * q);
  out[5 * n + i] = q;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(6 * n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (s >> 11) * 0x1p-53, (int)(s % 80) - 40); }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, n, dx, dout);
  hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
  const char* names[6] = {"rcp raw", "rcp + 1 Newton", "rcp + 2 Newton", "rsq raw", "rsq + 1 Newton", "rsq + 2 Newton"};
  for (int v = 0; v < 6; ++v) {
    double worst = 0.0;
    for (int i = 0; i < n; ++i) {
      const long double ex = v < 3 ? 1.0L / x[i] : 1.0L / sqrtl((long double)x[i]);
      worst = std::fmax(worst, (double)fabsl((o[v * n + i] - ex) / ex));
    }
    std::printf("%-16s max relative error %.3e\n", names[v], worst);
  }
  return 0;
}
