// Accuracy of v_rcp_f64 and of one / two Newton steps on top of it (the radial network evaluates 20 reciprocals per pair).
//   hipcc --offload-arch=gfx950 -O3 -o rcp_probe.bin rcp_probe.hip && ./rcp_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double u = x[i];
  double r = __builtin_amdgcn_rcp(u);
  r0[i] = r;
  double e = __builtin_fma(-u, r, 1.0);
  r = __builtin_fma(r, e, r);
  r1[i] = r;
  e = __builtin_fma(-u, r, 1.0);
  r2[i] = __builtin_fma(r, e, r);
}
int main() {
  const int n = 1 << 20;
  double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double t = (double)(s >> 11) / 9007199254740992.0;      // [0, 1)
    hx[i] = 1.0 + t * ((i & 1) ? 1.0 : 1e6);                      // the argument range of 1 + c^2 |n|^2
  }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double ex = 1.0L / (long double)hx[i];
    m0 = fmax(m0, (double)fabsl(((long double)h0[i] - ex) / ex));
    m1 = fmax(m1, (double)fabsl(((long double)h1[i] - ex) / ex));
    m2 = fmax(m2, (double)fabsl(((long double)h2[i] - ex) / ex));
  }
  printf("max relative error: v_rcp_f64 %.3e   + 1 Newton step %.3e   + 2 Newton steps %.3e   (eps = 2.2e-16)\n", m0, m1, m2);
  return 0;
}
