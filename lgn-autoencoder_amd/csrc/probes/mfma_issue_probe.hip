// What ONE wave per SIMD can keep up beside a stream of v_mfma_f64_16x16x4_f64 (round 5: the chain CGMLP kernels, mlp_chain.hip).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_issue_probe mfma_issue_probe.hip && ./mfma_issue_probe
// 256 workgroups x 256 threads (one wave per SIMD), 12 matrix instructions per loop body, order pinned by sched_barrier(0).
// Variants: accumulator pattern (1 chain / 3 round robin), A operand from LDS through a 9-deep read queue, and per matrix
// instruction K extra instructions of one kind (32-bit VALU, fp64 VALU, ds_write_b64, global load, global store).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

enum Kind { NONE = 0, VALU32, VALU64, DSWRITE, GLOAD, GSTORE, DSREAD_ONLY };

template <int NACC, bool LDSA, int KIND, int K>
__global__ __launch_bounds__(256) void probe(double* out, const double* in, long long* cyc, int iters) {
  __shared__ double lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 4096; e += 256) lds[e] = 1e-3 * e;
  __syncthreads();
  v4d acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double x = tid * 0.001, y = 1.0 + tid * 1e-6;
  unsigned u = tid;
  double d = 1.0 + tid * 1e-9, gl = 0.0;
  const double* la = lds + (lane & 15) * 50 + (lane >> 4);
  double q[12];
#pragma unroll
  for (int i = 0; i < 9; ++i) q[i] = LDSA ? la[4 * i] : x;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (LDSA) q[(i + 9) % 12] = la[4 * ((i + 9) % 12) + 48 * (it & 7)];
      acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(LDSA ? q[i] : x, y, acc[i % NACC], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == VALU32) u = u * 3u + 7u;
        if (KIND == VALU64) d = d * 1.0000001;
        if (KIND == DSWRITE) lds[2048 + ((tid + 64 * k) & 1023)] = d;
        if (KIND == GLOAD) gl += in[(size_t)tid + 256 * ((it * 12 + i) & 63)];
        if (KIND == GSTORE) __builtin_nontemporal_store(d, &out[(size_t)blockIdx.x * 4096 + tid + 256 * ((i + k) & 15)]);
      }
      FENCE();
    }
  }
  const long long t1 = clock64();
  out[(size_t)blockIdx.x * 4096 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + u + d + gl + lds[2048 + tid];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// Activation of a finished accumulator tile in the shadow of the following matrix instructions: 36 matrix instructions per
// "layer" in three 12-instruction chains; the four values of chain t are activated DIST instructions after its last one.
//   FORM 0: t = 0.01 x, y = x > 0 ? x : t   1: the same with the sign test on the high dword   2: fmax(x, 0.01 x)
//   3: y = x * (sign ? 0.01 : 1)            4: none
template <int FORM, int DIST>
__global__ __launch_bounds__(256) void act_probe(double* out, long long* cyc, int iters) {
  __shared__ double lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 4096; e += 256) lds[e] = 1e-3 * e - 2.0;
  __syncthreads();
  const double* la = lds + (lane & 15) * 50 + (lane >> 4);
  v4d h[3] = {{1, -2, 3, -4}, {-1, 2, -3, 4}, {0.5, -0.5, 0.25, -0.25}};
  double q[36];
#pragma unroll
  for (int i = 0; i < 9; ++i) q[i] = la[4 * i];
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    v4d acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v4d hn[3];
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      const int t = i / 12, ks = i % 12;
      q[(i + 9) % 36] = la[4 * ((i + 9) % 36) + 48 * (it & 7)];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(q[i], h[ks >> 2][ks & 3], acc[t], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        if (i == 12 * (tt + 1) - 1 + DIST) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double x = acc[tt][r];
            double y;
            if (FORM == 0) { const double t_ = 0.01 * x; y = x > 0.0 ? x : t_; }
            else if (FORM == 1) { const double t_ = 0.01 * x; y = __double2hiint(x) < 0 ? t_ : x; }
            else if (FORM == 2) y = fmax(x, 0.01 * x);
            else if (FORM == 3) y = x * (__double2hiint(x) < 0 ? 0.01 : 1.0);
            else y = x;
            asm volatile("" : "+v"(y));
            hn[tt][r] = y;
          }
        }
      FENCE();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double x = acc[2][r];
      double y;
      if (FORM == 0) { const double t_ = 0.01 * x; y = x > 0.0 ? x : t_; }
      else if (FORM == 1) { const double t_ = 0.01 * x; y = __double2hiint(x) < 0 ? t_ : x; }
      else if (FORM == 2) y = fmax(x, 0.01 * x);
      else if (FORM == 3) y = x * (__double2hiint(x) < 0 ? 0.01 : 1.0);
      else y = x;
      hn[2][r] = y;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) h[t] = hn[t];
    FENCE();
  }
  const long long t1 = clock64();
  out[(size_t)blockIdx.x * 4096 + tid] = h[0][0] + h[1][1] + h[2][2];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int FORM, int DIST>
static void run_act(const char* what, double* out, long long* cyc) {
  const int iters = 1000;
  long long h = 0;
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL((act_probe<FORM, DIST>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("%-72s %6.0f cycles per 36-instruction layer (2 304 = the pipe)\n", what, (double)h / iters);
}

template <int NACC, bool LDSA, int KIND, int K>
static void run(const char* what, double* out, double* in, long long* cyc) {
  const int iters = 2000;
  long long h = 0;
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL((probe<NACC, LDSA, KIND, K>), dim3(256), dim3(256), 0, 0, out, in, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("%-72s %6.1f cycles per matrix instruction\n", what, (double)h / (12.0 * iters));
}

int main() {
  double *out, *in; long long* cyc;
  hipMalloc(&out, 256 * 4096 * 8); hipMalloc(&in, 1 << 20); hipMalloc(&cyc, 8);
  hipMemset(in, 0, 1 << 20);
  run<1, false, NONE, 0>("one accumulator chain, operands in registers", out, in, cyc);
  run<3, false, NONE, 0>("three accumulators round robin, operands in registers", out, in, cyc);
  run<1, true, NONE, 0>("one chain, A from LDS (9 reads ahead)", out, in, cyc);
  run<3, true, NONE, 0>("three accumulators, A from LDS (9 reads ahead)", out, in, cyc);
  run<3, true, VALU32, 2>("  + 2 x 32-bit VALU per matrix instruction", out, in, cyc);
  run<3, true, VALU32, 6>("  + 6 x 32-bit VALU", out, in, cyc);
  run<3, true, VALU32, 12>("  + 12 x 32-bit VALU", out, in, cyc);
  run<3, true, VALU64, 1>("  + 1 x fp64 VALU (v_mul_f64)", out, in, cyc);
  run<3, true, VALU64, 2>("  + 2 x fp64 VALU", out, in, cyc);
  run<3, true, VALU64, 4>("  + 4 x fp64 VALU", out, in, cyc);
  run<3, true, DSWRITE, 1>("  + 1 x ds_write_b64", out, in, cyc);
  run<3, true, DSWRITE, 2>("  + 2 x ds_write_b64", out, in, cyc);
  run<3, true, GLOAD, 1>("  + 1 x global_load_dwordx2", out, in, cyc);
  run<3, true, GSTORE, 1>("  + 1 x global_store_dwordx2 (nt)", out, in, cyc);
  run_act<4, 2>("layer of 3 x 12, no activation", out, cyc);
  run_act<0, 2>("mul + fp64 compare + select, 2 instructions after the tile", out, cyc);
  run_act<0, 6>("mul + fp64 compare + select, 6 after", out, cyc);
  run_act<1, 2>("mul + sign-bit compare + select, 2 after", out, cyc);
  run_act<1, 6>("mul + sign-bit compare + select, 6 after", out, cyc);
  run_act<2, 6>("fmax(x, 0.01 x), 6 after", out, cyc);
  run_act<3, 6>("x * (sign ? 0.01 : 1), 6 after", out, cyc);
  return 0;
}
