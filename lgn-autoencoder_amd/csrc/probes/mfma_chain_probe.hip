// Why does ONE wave per SIMD issuing 36 MFMAs per "layer" run at ~125 ticks per MFMA in the CGMLP kernel?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain_probe mfma_chain_probe.hip && ./mfma_chain_probe
// Variants (all: 256 WGs, one wave per SIMD when 256 threads):
//   0: same A/B registers, 4 chains (the rate probe)            1: 36 distinct A registers, 3 chains, B from 12 registers
//   2: as 1 with A fragments re-read from LDS every round      3: as 2 with a workgroup barrier per round
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256) void chain(double* out, long long* cyc, int rounds) {
  __shared__ double W[48 * 50];
  const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  for (int e = threadIdx.x; e < 48 * 50; e += blockDim.x) W[e] = 1e-3 * (e % 97);
  __syncthreads();
  v4d acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  v4d x[3] = {{1, 2, 3, 4}, {2, 3, 4, 5}, {3, 4, 5, 6}};
  double af[12][3];
  const double* wa = W + i * 50 + kq;
  for (int s = 0; s < 12; ++s)
    for (int t = 0; t < 3; ++t) af[s][t] = wa[16 * t * 50 + 4 * s];
  long long t0 = clock64();
  for (int r = 0; r < rounds; ++r) {
    if (VAR >= 2) {
#pragma unroll
      for (int s = 0; s < 12; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) af[s][t] = wa[16 * t * 50 + 4 * s + (r & 1)];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (VAR == 0) {
#pragma unroll
      for (int s = 0; s < 12; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][0], x[0][0], acc[t], 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < 12; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s][t], x[s >> 2][s & 3], acc[t], 0, 0, 0);
    }
    if (VAR == 3) __syncthreads();
    // feed the result back like the layers do (keeps the compiler from hoisting anything)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) x[t][q] = fmax(acc[t][q], 0.01 * acc[t][q]) * 1e-3;
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + x[0][0];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int VAR>
void run(int threads, int wgs, double* out, long long* cyc) {
  const int rounds = 2000;
  float ms = 0;
  long long h = 0;
  for (int pass = 0; pass < 2; ++pass) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<VAR>, dim3(wgs), dim3(threads), 0, 0, out, cyc, rounds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  const double mfmas = 36.0 * rounds;
  printf("variant %d, %3d threads/WG, %4d WGs: %.1f ticks per MFMA per wave; %.3f ms -> %.1f ns per 36-MFMA round, %.1f TFLOP/s\n", VAR, threads, wgs,
         (double)h / mfmas, ms, ms * 1e6 / rounds, (double)wgs * (threads / 64) * mfmas * 2048 / (ms * 1e-3) / 1e12);
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  for (int wgs : {256, 16}) {
    run<0>(256, wgs, out, cyc);
    run<1>(256, wgs, out, cyc);
    run<2>(256, wgs, out, cyc);
    run<3>(256, wgs, out, cyc);
  }
  run<1>(64, 256, out, cyc);
  run<1>(512, 256, out, cyc);
  run<2>(512, 256, out, cyc);
  return 0;
}
