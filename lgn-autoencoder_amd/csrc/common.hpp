// lgn-autoencoder_amd/csrc/common.hpp -- shared device/host helpers for the gfx950 kernels.
//
// Layout conventions (identical to the reference's planar-complex GVec layout,
// lgn/g_lib/g_vec.py:30-48, so that level buffers can be handed back to Python as the
// reference's `nodes_all` without a transpose):
//   scalar irrep (0,0):  T s[2][B][N][C]
//   vector irrep (1,1):  T v[2][B][N][C][4]      canonical basis (E, (px-i py)/rt2, pz, (-px-i py)/rt2)
//   MixReps weight:      T w[2][C_out][C_in]
// plane 0 = real part, plane 1 = imaginary part.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace lgn {

constexpr int NB = 20;          // 2 * num_basis_fn radial basis functions (position_levels.py:63)
constexpr int BLOCK = 256;      // threads per workgroup (4 waves of 64)

// ---- error plumbing -------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define LGN_CHECK_ARG(cond, ...)                      \
  do {                                                \
    if (!(cond)) {                                    \
      lgn::set_error(__VA_ARGS__);                    \
      return -1;                                      \
    }                                                 \
  } while (0)
#define LGN_CHECK_LAUNCH()                                              \
  do {                                                                  \
    hipError_t e_ = hipGetLastError();                                  \
    if (e_ != hipSuccess) {                                             \
      lgn::set_error("HIP launch failed: %s", hipGetErrorString(e_));   \
      return (int)e_;                                                   \
    }                                                                   \
  } while (0)

// ---- tiny complex type living in registers ---------------------------------------------
template <typename T>
struct cx {
  T r, i;
};
template <typename T>
__device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
  return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r};
}
// a * conj(b)
template <typename T>
__device__ __forceinline__ cx<T> cmulc(cx<T> a, cx<T> b) {
  return {a.r * b.r + a.i * b.i, a.i * b.r - a.r * b.i};
}
// acc += a * b as two fused multiply-adds per component.  (Written "acc.r += a.r * b.r - a.i * b.i" the compiler forms the
// product first -- mul + fma -- and then needs a third instruction to add it to the accumulator.)
template <typename T>
__device__ __forceinline__ void cfma(cx<T>& acc, cx<T> a, cx<T> b) {
  acc.r = __builtin_fma(a.r, b.r, acc.r);
  acc.r = __builtin_fma(-a.i, b.i, acc.r);
  acc.i = __builtin_fma(a.r, b.i, acc.i);
  acc.i = __builtin_fma(a.i, b.r, acc.i);
}
// acc += a * conj(b)
template <typename T>
__device__ __forceinline__ void cfmac(cx<T>& acc, cx<T> a, cx<T> b) {
  acc.r = __builtin_fma(a.r, b.r, acc.r);
  acc.r = __builtin_fma(a.i, b.i, acc.r);
  acc.i = __builtin_fma(a.i, b.r, acc.i);
  acc.i = __builtin_fma(-a.r, b.i, acc.i);
}

template <typename T>
__device__ __forceinline__ T shfl_xor(T v, int m) {
  return __shfl_xor(v, m, 64);
}

// Sum over the JS consecutive lanes that share a row (JS power of two <= 64).
template <int JS, typename T>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
  for (int m = 1; m < JS; m <<= 1) v += shfl_xor(v, m);
  return v;
}

// Phase stamps (debug builds only: `make stamps` -> liblgn_amd_stamps.so, read by tools/kbench.py with KB_STAMPS=1).
// Thread 0 of workgroup 0 records s_memtime at each STAMP(i); compiled out of the shipped library.
// Slots 64 .. 127 hold the constant-rate counter (wall_clock64 = s_memrealtime) at the same points: the ratio of the two differences
// is the shader clock the kernel actually ran at (tools/kbench.py prints it).
#ifdef LGN_STAMPS
// Slots 128 .. 255: the same for the LAST workgroup of the grid (dispatch stagger, load imbalance between jets).
#define LGN_STAMP_DECL static __device__ long long g_stamps[256];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) { \
    const int o_ = blockIdx.x == 0 ? 0 : 128; g_stamps[o_ + (i)] = clock64(); g_stamps[o_ + 64 + (i)] = wall_clock64(); } } while (0)
#define LGN_STAMP_READER(name) \
  extern "C" int name(long long* out) {                                                                              \
    const int rc_ = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * 256);                      \
    const long long zero_ = 0;                       /* slot 63 starts over with every read */                        \
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &zero_, sizeof(long long), 63 * sizeof(long long));                   \
    return rc_;                                                                                                        \
  }
// slot 63: the LONGEST lifetime of any workgroup (ticks of the constant-rate counter, over all launches since the library was loaded)
#define STAMP_LIFE_BEGIN() const long long life0_ = wall_clock64()
#define STAMP_LIFE_END() do { if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(&g_stamps[63]), \
                                                           (unsigned long long)(wall_clock64() - life0_)); } while (0)
#else
#define LGN_STAMP_DECL
#define STAMP(i) do { } while (0)
#define STAMP_LIFE_BEGIN() do { } while (0)
#define STAMP_LIFE_END() do { } while (0)
#define LGN_STAMP_READER(name)
#endif

template <typename T>
__device__ __forceinline__ T leaky(T x) {
  return fmax(x, T(0.01) * x);         // == x > 0 ? x : 0.01 x; nn.LeakyReLU default slope (generic_levels.py:121-122)
}
// The CGMLP's activation (get_activation_fn, lgn/nn/generic_levels.py:119-135; ids = LGN_ACT_* of include/lgn_amd.h).  The choice is
// wave uniform; id 0 (LeakyReLU, the reference default) takes the first branch.
__device__ __forceinline__ double act_apply(double x, int act) {
  if (act == 0) return fmax(x, 0.01 * x);
  switch (act) {
    case 1: return fmax(x, 0.0);                                 // nn.ReLU
    case 2: return x > 0.0 ? x : expm1(x);                       // nn.ELU(alpha = 1)
    case 3: return 1.0 / (1.0 + exp(-x));                        // nn.Sigmoid
    case 4: return fmin(x, 0.0) - log1p(exp(-fabs(x)));          // nn.LogSigmoid (the stable form ATen uses)
    default: return atan(x);                                     // ATan (generic_levels.py:138-140)
  }
}
// GEN = false instantiations are the LeakyReLU kernels with nothing else compiled in (the run-time switch costs the default
// path 5 us per CGMLP launch: registers and code of the transcendental branches); GEN = true takes the switch.
template <bool GEN>
__device__ __forceinline__ double act_apply_t(double x, int act) {
  if constexpr (GEN) return act_apply(x, act);
  else return fmax(x, 0.01 * x);
}
// d act / d x expressed through the OUTPUT y = act(x): the backward keeps post-activations only
__device__ __forceinline__ double act_slope(double y, int act) {
  if (act == 0) return y > 0.0 ? 1.0 : 0.01;
  switch (act) {
    case 1: return y > 0.0 ? 1.0 : 0.0;
    case 2: return y > 0.0 ? 1.0 : y + 1.0;                      // e^x = y + 1 for x <= 0
    case 3: return y * (1.0 - y);
    case 4: return -expm1(y);                                    // 1 - sigmoid(x), sigmoid(x) = e^y
    default: { const double c = cos(y); return c * c; }          // 1 / (1 + tan^2 y)
  }
}
template <bool GEN>
__device__ __forceinline__ double act_slope_t(double y, int act) {
  if constexpr (GEN) return act_slope(y, act);
  else return y > 0.0 ? 1.0 : 0.01;
}

// 1/sqrt(2) used by the Cartesian <-> canonical change of basis (zonal_functions.py:266-283)
template <typename T>
__device__ __forceinline__ constexpr T rsqrt2() {
  return T(0.70710678118654752440084436210484903928);
}

// signed pseudo-norm of a real Cartesian 4-vector difference, exactly as the reference forms it:
// norm_sq = 2 E^2 - sum(p^2) + 1e-16 ; norm = norm_sq / sqrt(|norm_sq|)   (zonal_functions.py:142-144,201-218)
template <typename T>
__device__ __forceinline__ T signed_norm(T d0, T d1, T d2, T d3, T& nsq_out) {
  T q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
  T nsq = (T(2) * q0 - (((q0 + q1) + q2) + q3)) + T(1e-16);
  nsq_out = nsq;
  return nsq != T(0) ? nsq / sqrt(fabs(nsq)) : nsq;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace lgn
