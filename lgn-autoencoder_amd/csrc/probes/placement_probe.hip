// Which workgroups of a one-round launch share a CU?  512 workgroups x 256 threads with 77 KB of LDS each (the shape of the level
// kernels at bs = 512: two workgroups per CU) record HW_ID / XCC_ID; the host prints the blockIdx pairs per (XCC, SE, CU).
//   hipcc --offload-arch=gfx950 -O3 -o placement_probe placement_probe.hip && ./placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void where(unsigned* out, int spin) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double x = lds[(threadIdx.x + 1) & 255];
  for (int i = 0; i < spin; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);      // stay resident while the rest of the grid is dispatched
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);        // HW_ID
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // XCC_ID [3:0]
  }
  if (x == 0.5) out[0] = 0;
}
int main() {
  const int B = 512;
  unsigned* d; hipMalloc(&d, B * 8);
  hipFuncSetAttribute((const void*)where, hipFuncAttributeMaxDynamicSharedMemorySize, 77 * 1024);
  where<<<B, 256, 77 * 1024>>>(d, 20000);
  std::vector<unsigned> h(2 * B);
  hipMemcpy(h.data(), d, B * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < B; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
    const unsigned cu_id = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;   // gfx9 HW_ID: cu [11:8], sh [12], se [15:13]
    cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
  }
  printf("%zu distinct (xcc, se, sh, cu); workgroups per CU:\n", cu.size());
  int shown = 0;
  std::map<int, int> delta;
  for (auto& kv : cu) {
    if (shown++ < 24) { printf("  xcc %u se %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, kv.first & 15); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
  }
  printf("blockIdx distance between the two workgroups of a CU -> number of CUs:\n");
  for (auto& kv : delta) printf("  %d: %d\n", kv.first, kv.second);
  return 0;
}
