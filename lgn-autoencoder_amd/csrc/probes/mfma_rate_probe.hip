// Issue-rate probe: cycles per v_mfma_f64_16x16x4_f64 and per v_fma_f64 (wave64), one or several waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe mfma_rate_probe.hip && ./mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void mfma_loop(double* out, long long* cyc, int iters) {
  v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 0.001, y = 1.0 + threadIdx.x * 1e-6;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void fma_loop(double* out, long long* cyc, int iters) {
  double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  double x = threadIdx.x * 0.001, y = 1.0 + threadIdx.x * 1e-6;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_fma(a0, x, y); a1 = __builtin_fma(a1, x, y); a2 = __builtin_fma(a2, x, y); a3 = __builtin_fma(a3, x, y);
    a4 = __builtin_fma(a4, x, y); a5 = __builtin_fma(a5, x, y); a6 = __builtin_fma(a6, x, y); a7 = __builtin_fma(a7, x, y);
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// Do fp64 MFMAs and fp64 VALU FMAs of two waves on the same SIMD overlap?  512 threads per workgroup = 8 waves, waves w and w + 4
// share a SIMD: role 0 = every wave MFMA, 1 = every wave FMA, 2 = waves 0..3 MFMA and waves 4..7 FMA (one of each per SIMD).
__global__ void mixed_loop(double* out, int iters, int role) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = role == 0 || (role == 2 && wave < 4);
  double x = threadIdx.x * 0.001, y = 1.0 + threadIdx.x * 1e-6, r = 0;
  if (do_mfma) {
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    for (int i = 0; i < 8 * iters; ++i) {       // 64 FMAs per iteration of the MFMA loop's 4 MFMAs: the same 256 cycles at peak
      a0 = __builtin_fma(a0, x, y); a1 = __builtin_fma(a1, x, y); a2 = __builtin_fma(a2, x, y); a3 = __builtin_fma(a3, x, y);
      a4 = __builtin_fma(a4, x, y); a5 = __builtin_fma(a5, x, y); a6 = __builtin_fma(a6, x, y); a7 = __builtin_fma(a7, x, y);
    }
    r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  double* out; long long* cyc; long long h;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int threads : {64, 256, 512}) {
    for (int pass = 0; pass < 2; ++pass) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(256 * 2), dim3(threads), 0, 0, out, cyc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      if (pass) printf("mfma f64 16x16x4: %d threads/WG, 512 WGs: %.1f clock64 ticks per MFMA per wave, %.3f ms -> %.1f TFLOP/s\n", threads,
                       (double)h / (4.0 * iters), ms, 512.0 * (threads / 64) * 4.0 * iters * 2048 / (ms * 1e-3) / 1e12);
    }
    for (int pass = 0; pass < 2; ++pass) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(fma_loop, dim3(256 * 2), dim3(threads), 0, 0, out, cyc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      if (pass) printf("v_fma_f64: %d threads/WG: %.1f ticks per FMA per wave, %.3f ms -> %.1f TFLOP/s\n", threads, (double)h / (8.0 * iters), ms,
                       512.0 * threads * 8.0 * iters * 2 / (ms * 1e-3) / 1e12);
    }
  }
  for (int role = 0; role < 3; ++role) {
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(mixed_loop, dim3(256), dim3(512), 0, 0, out, 5000, role);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("mixed: role %d (%s): %.3f ms\n", role, role == 0 ? "8 MFMA waves per CU" : role == 1 ? "8 FMA waves per CU" : "4 MFMA + 4 FMA waves per CU, one of each per SIMD", ms);
  }
  return 0;
}
