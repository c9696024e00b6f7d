// What a kernel node of a replayed HIP graph costs beside its work (round 5: a cfg2 step is 25 dependent launches).
//   hipcc --offload-arch=gfx950 -O3 -o launch_overhead_probe.probe launch_overhead_probe.hip && ./launch_overhead_probe.probe
// A graph of 40 dependent launches of (a) an empty kernel, (b) a kernel whose workgroups each spin for ~10 us on the constant-rate
// counter, at 1 / 256 / 512 workgroups of 256 threads; printed: time per launch, and for (b) the part that is not the 10 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void empty_kernel(double* p) { if (p && threadIdx.x == 9999) p[0] = 1.0; }
__global__ void spin_kernel(double* p, long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (p && threadIdx.x == 9999) p[0] = 1.0;
}
int main() {
  double* buf;
  CK(hipMalloc(&buf, 1 << 20));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nl = 40;
  for (int spin = 0; spin < 2; ++spin)
    for (int wg : {1, 256, 512}) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < nl; ++i) {
        if (spin) hipLaunchKernelGGL(spin_kernel, dim3(wg), dim3(256), 0, st, buf, 1000LL);      // 1000 ticks of 100 MHz = 10 us
        else hipLaunchKernelGGL(empty_kernel, dim3(wg), dim3(256), 0, st, buf);
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      const int reps = 20;
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / (reps * nl);
      if (spin) printf("spin 10 us, %3d workgroups: %6.2f us per launch -> %5.2f us beside the work\n", wg, us, us - 10.0);
      else printf("empty kernel, %3d workgroups: %6.2f us per launch\n", wg, us);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  return 0;
}
