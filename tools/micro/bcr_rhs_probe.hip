// Where do the 29 us of k_sep_bcr_rhs go?  The kernel's loops on synthetic data of configs[2]'s shape (m = 330 separators of 9 unknowns, 289
// right-hand-side columns, one column per workgroup of 256 threads), with s_memtime stamps at every phase boundary of a few workgroups:
// load of the column, each forward level, each backward level, store.  build: hipcc -O3 --offload-arch=gfx950 -o bcr_rhs_probe tools/micro/bcr_rhs_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int S3 = 9, SS = 81, TC = 1, kMaxStamp = 64;
struct View { const double *Ha, *Hc, *U2; double* R; int m; };
template <int NT>
__global__ __launch_bounds__(NT) void k_probe(View pv, int ncols, unsigned long long* stamps, int variant) {
  extern __shared__ double lds[];
  const int m = pv.m, tid = threadIdx.x;
  double* rs = lds;
  double* xs = lds + (size_t)m * S3 * TC;
  const int col0 = blockIdx.x * TC;
  int ns_ = 0;
  unsigned long long* st = stamps + (size_t)blockIdx.x * kMaxStamp;
  auto stamp = [&]() { if (tid == 0 && ns_ < kMaxStamp) st[ns_] = __builtin_readcyclecounter(); ++ns_; };
  stamp();
  for (int e = tid; e < m * S3 * TC; e += NT) {
    const int c = e % TC, a = (e / TC) % S3, q = e / (TC * S3);
    rs[e] = col0 + c < ncols ? pv.R[((long long)q * S3 + a) * ncols + col0 + c] : 0.0;
  }
  __syncthreads();
  stamp();
  int h = 1;
  for (; 2 * h <= m; h <<= 1) {
    const int ns = m / (2 * h);
    for (int e = tid; e < ns * S3 * TC; e += NT) {
      const int c = e % TC, a = (e / TC) % S3, i = 2 * h * (e / (TC * S3) + 1) - 1;
      double acc = rs[(i * S3 + a) * TC + c];
      const double* Hl = pv.Hc + (long long)(i - h) * SS + a;
      const double* rl = rs + (i - h) * S3 * TC + c;
#pragma unroll
      for (int k = 0; k < S3; ++k) acc -= (variant & 1 ? 1e-3 * (k + 1) : Hl[k * S3]) * rl[k * TC];
      if (i + h < m) {
        const double* Hr = pv.Ha + (long long)(i + h) * SS + a;
        const double* rr = rs + (i + h) * S3 * TC + c;
#pragma unroll
        for (int k = 0; k < S3; ++k) acc -= (variant & 1 ? 1e-3 * (k + 1) : Hr[k * S3]) * rr[k * TC];
      }
      rs[(i * S3 + a) * TC + c] = acc;
    }
    __syncthreads();
    stamp();
  }
  for (; h >= 1; h >>= 1) {
    const int ne = (m / h + 1) / 2;
    for (int e = tid; e < ne * S3 * TC; e += NT) {
      const int c = e % TC, a = (e / TC) % S3, j = h * (2 * (e / (TC * S3)) + 1) - 1;
      const double* Di = pv.U2 + (long long)j * SS + a * S3;
      const double* rj = rs + j * S3 * TC + c;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < S3; ++k) acc += (variant & 1 ? 1e-3 * (k + 1) : Di[k]) * rj[k * TC];
      if (j - h >= 0) {
        const double* Hl = pv.Ha + (long long)j * SS + a * S3;
        const double* xl = xs + (j - h) * S3 * TC + c;
#pragma unroll
        for (int k = 0; k < S3; ++k) acc -= (variant & 1 ? 1e-3 * (k + 1) : Hl[k]) * xl[k * TC];
      }
      if (j + h < m) {
        const double* Hr = pv.Hc + (long long)j * SS + a * S3;
        const double* xr = xs + (j + h) * S3 * TC + c;
#pragma unroll
        for (int k = 0; k < S3; ++k) acc -= (variant & 1 ? 1e-3 * (k + 1) : Hr[k]) * xr[k * TC];
      }
      xs[(j * S3 + a) * TC + c] = acc;
    }
    __syncthreads();
    stamp();
  }
  for (int e = tid; e < m * S3 * TC; e += NT) {
    const int c = e % TC, a = (e / TC) % S3, q = e / (TC * S3);
    if (col0 + c < ncols) pv.R[((long long)q * S3 + a) * ncols + col0 + c] = xs[e];
  }
  stamp();
  if (tid == 0) st[kMaxStamp - 1] = ns_;
}
int main() {
  const int m = 330, ncols = 289;
  std::vector<double> H((size_t)m * SS), R((size_t)m * S3 * ncols);
  for (size_t i = 0; i < H.size(); ++i) H[i] = 1e-3 * ((i * 2654435761u) % 1000) / 1000.0;
  for (size_t i = 0; i < R.size(); ++i) R[i] = ((i * 40503u) % 1000) / 1000.0;
  double *Ha, *Hc, *U2, *Rd; unsigned long long* st;
  (void)hipMalloc(&Ha, H.size() * 8); (void)hipMalloc(&Hc, H.size() * 8); (void)hipMalloc(&U2, H.size() * 8); (void)hipMalloc(&Rd, R.size() * 8);
  (void)hipMalloc(&st, (size_t)ncols * kMaxStamp * 8);
  (void)hipMemcpy(Ha, H.data(), H.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(Hc, H.data(), H.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(U2, H.data(), H.size() * 8, hipMemcpyHostToDevice);
  View pv{Ha, Hc, U2, Rd, m};
  const size_t lds = (size_t)2 * m * S3 * TC * 8;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int nt : {256, 512, 1024})
  for (int variant = 0; variant < 2; ++variant)
    for (int G : {256, 289}) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemcpy(Rd, R.data(), R.size() * 8, hipMemcpyHostToDevice);
        (void)hipEventRecord(e0, 0);
        if (nt == 256) hipLaunchKernelGGL(k_probe<256>, dim3(G), dim3(256), lds, 0, pv, ncols, st, variant);
        else if (nt == 512) hipLaunchKernelGGL(k_probe<512>, dim3(G), dim3(512), lds, 0, pv, ncols, st, variant);
        else hipLaunchKernelGGL(k_probe<1024>, dim3(G), dim3(1024), lds, 0, pv, ncols, st, variant);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<unsigned long long> s((size_t)G * kMaxStamp);
      (void)hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
      printf("%4d threads, %s, %3d workgroups: %.1f us (HIP events).  cycles per phase", nt, variant ? "H from constants" : "H from memory   ", G, ms * 1e3);
      for (int b : {0, G - 1}) {
        const unsigned long long* q = s.data() + (size_t)b * kMaxStamp;
        const int n = (int)q[kMaxStamp - 1];
        printf("\n   workgroup %3d: load %5llu | down", b, q[1] - q[0]);
        int k = 2;
        for (int h = 1; 2 * h <= m; h <<= 1, ++k) printf(" %5llu", q[k] - q[k - 1]);
        printf(" | up");
        for (; k < n - 1; ++k) printf(" %5llu", q[k] - q[k - 1]);
        printf(" | store %5llu | total %llu", q[n - 1] - q[n - 2], q[n - 1] - q[0]);
        if (b == G - 1 || G == 1) break;
      }
      printf("\n");
    }
  return 0;
}
