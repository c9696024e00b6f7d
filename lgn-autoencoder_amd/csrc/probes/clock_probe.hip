// Which shader clock does the chip hold under which instruction mix?  One wave per SIMD (or two) runs a loop for >= 2 s and reads
// both counters at its ends: s_memtime (clock64: shader cycles) and s_memrealtime (wall_clock64: constant rate, calibrated here
// against the host's steady clock).  cycles / seconds = the sustained clock.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void read_wall(long long* out) { out[0] = wall_clock64(); }

// mix: 0 = fp64 MFMA only, 1 = fp64 FMA only, 2 = MFMA + FMA + LDS traffic in one wave (the level kernels' diet), 3 = integer adds
__global__ void burn(double* out, long long* t, int iters, int mix) {
  __shared__ double lds[4096];
  v4d a0 = {0, 0, 0, 0}, a1 = a0;
  double x = threadIdx.x * 0.001, y = 1.0 + threadIdx.x * 1e-6, f0 = 0, f1 = 1, f2 = 2, f3 = 3;
  int n = threadIdx.x;
  for (int e = threadIdx.x; e < 4096; e += blockDim.x) lds[e] = e * 1e-3;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (mix == 0 || mix == 2) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
    }
    if (mix == 1 || mix == 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        f0 = __builtin_fma(f0, x, y); f1 = __builtin_fma(f1, x, y); f2 = __builtin_fma(f2, x, y); f3 = __builtin_fma(f3, x, y);
      }
    }
    if (mix == 2) {
      const double v = lds[(threadIdx.x * 2 + i) & 4095];
      lds[(threadIdx.x * 2 + i + 1024) & 4095] = v + f0;
      x = x + v * 1e-30;
    }
    if (mix == 3) n = n * 3 + i;
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + f0 + f1 + f2 + f3 + n + lds[threadIdx.x];
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

int main() {
  double* out; long long* t; long long h[2];
  hipMalloc(&out, 1 << 24); hipMalloc(&t, 16);
  // rate of wall_clock64 against the host
  read_wall<<<1, 1>>>(t); hipDeviceSynchronize(); hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
  const auto s0 = std::chrono::steady_clock::now(); const long long w0 = h[0];
  std::this_thread::sleep_for(std::chrono::milliseconds(500));
  read_wall<<<1, 1>>>(t); hipDeviceSynchronize(); hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();
  const double wall_hz = (h[0] - w0) / dt;
  printf("wall_clock64 rate: %.3f MHz (against the host clock over %.3f s)\n", wall_hz / 1e6, dt);
  const char* names[4] = {"fp64 MFMA only", "fp64 FMA only", "fp64 MFMA + FMA + LDS (one wave does all three)", "integer VALU only"};
  for (int waves : {4, 8}) {
    for (int mix = 0; mix < 4; ++mix) {
      int iters = 2000;
      for (int pass = 0; pass < 2; ++pass) {          // pass 0 sizes the loop for ~2 s, pass 1 measures
        burn<<<256 * 2 / (waves / 4), 64 * waves>>>(out, t, iters, mix);
        hipDeviceSynchronize(); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        const double sec = h[1] / wall_hz;
        if (pass == 0) iters = (int)(iters * 2.0 / sec);
        else printf("%d waves per workgroup, 2 workgroups per CU | %-48s | %.2f s | %.3f GHz sustained\n", waves, names[mix], sec, h[0] / sec / 1e9);
      }
    }
  }
  return 0;
}
