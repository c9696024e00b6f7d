// lgn-autoencoder_amd/csrc/wave_sum.hpp -- sums of several per-lane fp64 values over the 64 lanes of a wave, in registers.
//
// A transposing butterfly: every stage halves the number of live values while it halves the lanes a partial sum is spread
// over.  The two cross-row stages use gfx950's v_permlane32_swap / v_permlane16_swap (3 instructions per pair of values),
// the in-row stages DPP (row_mirror, row_half_mirror, then the two quad_perm steps on the one value left).  About 50 VALU
// instructions for 12 values: no LDS, no waiting on another unit, a fixed summation order (deterministic).
//   wave_sum_core<NV>   -> one register per lane; the total of value k sits in lane wave_sum_owner(k) (and in its quad)
//   wave_sum_store<NV>  stores the NV totals to dst[0 .. NV)                    (one lane per value)
//   wave_allsum<NV>     broadcasts them: s[k] is wave-uniform (v_readlane -> scalar registers)
#pragma once
#include <hip/hip_runtime.h>

namespace lgn {

template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return v + __hiloint2double(hi, lo);
}
typedef unsigned ws_uint2_t __attribute__((ext_vector_type(2)));
template <bool ROW32>
__device__ __forceinline__ double swap_add(double a, double b) {
  ws_uint2_t lo, hi;
  if (ROW32) {
    lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  } else {
    lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  }
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// lanes with `first` keep a and fetch the partner's a; the others keep b and fetch the partner's b
template <int CTRL>
__device__ __forceinline__ double dpp_pair_add(double a, double b, bool first) {
  const double keep = first ? a : b, send = first ? b : a;
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(send), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(send), CTRL, 0xF, 0xF, true);
  return keep + __hiloint2double(hi, lo);
}

template <int NV>
__device__ __forceinline__ double wave_sum_core(const double (&v)[NV], int lane) {
  static_assert(NV == 8 || NV == 12 || NV == 16, "two cross-row stages need NV % 4 == 0");
  constexpr int N1 = NV / 2, N2 = NV / 4, N3 = (N2 + 1) / 2;
  double r[N1], t[N2], u[N3];
#pragma unroll
  for (int p = 0; p < N1; ++p) r[p] = swap_add<true>(v[2 * p], v[2 * p + 1]);       // 32-lane half h holds value 2p + h
#pragma unroll
  for (int q = 0; q < N2; ++q) t[q] = swap_add<false>(r[2 * q], r[2 * q + 1]);      // row rho holds value 4q + 2(rho & 1) + (rho >> 1)
  const int i = lane & 15;
#pragma unroll
  for (int j = 0; j < N2 / 2; ++j) u[j] = dpp_pair_add<0x140>(t[2 * j], t[2 * j + 1], i < 8);      // row_mirror
  if (N2 & 1) u[N3 - 1] = dpp_add<0x140>(t[N2 - 1]);
  double w;
  if (N3 == 2) w = dpp_pair_add<0x141>(u[0], u[1], (i & 4) == 0);      // row_half_mirror
  else w = dpp_add<0x141>(u[0]);
  w = dpp_add<0xB1>(w);
  return dpp_add<0x4E>(w);
}
// first lane that holds the total of value k:  k = 4 q + 2 (rho & 1) + (rho >> 1) in row rho, quad {0, 8, 4, 12}[q] of the row
__host__ __device__ constexpr int wave_sum_owner(int k) {
  const int kk = k & 3, q = k >> 2, rho = (kk >> 1) + 2 * (kk & 1);
  return 16 * rho + (q == 0 ? 0 : q == 1 ? 8 : q == 2 ? 4 : 12);
}
template <int NV>
__device__ __forceinline__ void wave_sum_store(const double (&v)[NV], double* __restrict__ dst, int lane) {
  constexpr int N2 = NV / 4, N3 = (N2 + 1) / 2;
  const double w = wave_sum_core<NV>(v, lane);
  const int i = lane & 15, rho = lane >> 4;
  const int jsel = N3 == 2 ? (i >> 2) & 1 : 0;
  const bool single = (N2 & 1) && jsel == N3 - 1;         // the unpaired value: both half rows hold it
  const int q = single ? N2 - 1 : 2 * jsel + (i >> 3);
  const bool owner = (i & 3) == 0 && (N3 == 2 || (i & 4) == 0) && (!single || i < 8);
  if (owner) dst[4 * q + 2 * (rho & 1) + (rho >> 1)] = w;
}
// the value whose total this lane is the FIRST holder of after wave_sum_core<NV> (wave_sum_store's owner rule), or -1
template <int NV>
__device__ __forceinline__ int wave_sum_slot(int lane) {
  constexpr int N2 = NV / 4, N3 = (N2 + 1) / 2;
  const int i = lane & 15, rho = lane >> 4;
  const int jsel = N3 == 2 ? (i >> 2) & 1 : 0;
  const bool single = (N2 & 1) && jsel == N3 - 1;
  const int q = single ? N2 - 1 : 2 * jsel + (i >> 3);
  const bool owner = (i & 3) == 0 && (N3 == 2 || (i & 4) == 0) && (!single || i < 8);
  return owner ? 4 * q + 2 * (rho & 1) + (rho >> 1) : -1;
}
template <int NV>
__device__ __forceinline__ void wave_allsum(const double (&v)[NV], double (&s)[NV], int lane) {
  const double w = wave_sum_core<NV>(v, lane);
#pragma unroll
  for (int k = 0; k < NV; ++k)
    s[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(w), wave_sum_owner(k)),
                            __builtin_amdgcn_readlane(__double2loint(w), wave_sum_owner(k)));
}

}  // namespace lgn
