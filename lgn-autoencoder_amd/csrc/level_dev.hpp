// lgn-autoencoder_amd/csrc/level_dev.hpp -- device helpers shared by the level forward/backward kernels.
#pragma once
#include "level.hpp"

namespace lgn {

// ---- LDS carve -------------------------------------------------------------------------
// All offsets are in units of T and are multiples of 2 (16-byte aligned for fp64 b128 reads).
template <int C, bool DEC>
struct Carve {
  static constexpr int R = 4 * C;                       // radial outputs per pair: [lin][2c+z]
  static constexpr int NS = node_stride(C);             // per-node scalars in the node tile
  static constexpr int PS = DEC ? 8 : 4;                // position scalars per node
  // radial parameter block
  static constexpr int RA = 0, RB = NB, RC = 2 * NB;    // a, b, c
  static constexpr int RW = 3 * NB;                     // Wt[k][R] (k-major)
  static constexpr int RBIAS = RW + NB * R;             // bias[R]
  static constexpr int RAD_SIZE = RBIAS + R;
  __host__ __device__ static constexpr int even(int x) { return (x + 1) & ~1; }
};

// Cooperative load of one jet's node features / positions / mask into LDS.
//   nd[j*NS + c*10 + {0: s_r, 1: s_i, 2..5: v_r[m], 6..9: v_i[m]}]
//   pj[j*PS + ...]  encoder: (E,px,py,pz); decoder: q_r[4], q_i[4] (complex canonical)
// Thread t takes item t of every array FIRST -- all its global loads are issued together (clamped addresses, no branches)
// and only then written to LDS: one memory round trip for a 30-particle jet instead of one per array (a loop "load, store"
// per array serialises the round trips: the stores of one loop wait for its loads before the next loop's loads are issued).
// Items beyond the first BLOCK of an array (N C > 256) take the plain loops.
template <typename T>
struct JetRegs {
  T s[2], v[8], p[2];
  uint8_t m;
};
template <typename T, int C, bool DEC>
__device__ __forceinline__ void load_jet_issue(const T* __restrict__ s_in, const T* __restrict__ v_in, const T* __restrict__ p,
                                               const uint8_t* __restrict__ mask, int B, int N, int b, JetRegs<T>& r) {
  const int tid = threadIdx.x;
  const size_t plane_s = (size_t)B * N * C, plane_p = (size_t)B * N * 4;
  const int e = tid < N * C ? tid : 0, ep = tid < N * 4 ? tid : 0, em = tid < N ? tid : 0;
  const T* s0 = s_in + (size_t)b * N * C;
  const T* v0 = v_in + (size_t)b * N * C * 4;
  const T* p0 = p + (size_t)b * N * 4;
  r.s[0] = s0[e];
  r.s[1] = s0[plane_s + e];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    r.v[m] = v0[e * 4 + m];
    r.v[4 + m] = v0[plane_s * 4 + e * 4 + m];
  }
  r.p[0] = p0[ep];
  r.p[1] = DEC ? p0[plane_p + ep] : T(0);
  r.m = DEC ? uint8_t(0) : mask[(size_t)b * N + em];
}
template <typename T, int C, bool DEC>
__device__ __forceinline__ void load_jet_commit(const T* __restrict__ s_in, const T* __restrict__ v_in, const T* __restrict__ p,
                                                const uint8_t* __restrict__ mask, int B, int N, int b, const JetRegs<T>& r,
                                                T* nd, T* pj, uint8_t* mk) {
  using L = Carve<C, DEC>;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const size_t plane_s = (size_t)B * N * C;
  if (tid < N * C) {
    const int j = tid / C, c = tid - j * C;
    T* d = nd + j * L::NS + c * 10;
    d[0] = r.s[0];
    d[1] = r.s[1];
#pragma unroll
    for (int m = 0; m < 8; ++m) d[2 + m] = r.v[m];
  }
  if (tid < N * 4) {
    if (DEC) {
      pj[(tid >> 2) * 8 + (tid & 3)] = r.p[0];
      pj[(tid >> 2) * 8 + 4 + (tid & 3)] = r.p[1];
    } else {
      pj[tid] = r.p[0];
    }
  }
  if (!DEC && tid < N) mk[tid] = r.m;
  // remainder (jets with more than BLOCK items per array; also the second half of a 512-thread workgroup's share)
  const T* s0 = s_in + (size_t)b * N * C;
  const T* v0 = v_in + (size_t)b * N * C * 4;
  for (int e = tid + nthr; e < N * C; e += nthr) {
    int j = e / C, c = e - j * C;
    T* d = nd + j * L::NS + c * 10;
    d[0] = s0[e];
    d[1] = s0[plane_s + e];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      d[2 + m] = v0[e * 4 + m];
      d[6 + m] = v0[plane_s * 4 + e * 4 + m];
    }
  }
  const T* p0 = p + (size_t)b * N * 4;
  if (DEC) {
    const size_t plane_p = (size_t)B * N * 4;
    for (int e = tid + nthr; e < N * 4; e += nthr) {
      pj[(e >> 2) * 8 + (e & 3)] = p0[e];
      pj[(e >> 2) * 8 + 4 + (e & 3)] = p0[plane_p + e];
    }
  } else {
    for (int e = tid + nthr; e < N * 4; e += nthr) pj[e] = p0[e];
    for (int e = tid + nthr; e < N; e += nthr) mk[e] = mask[(size_t)b * N + e];
  }
}
template <typename T, int C, bool DEC>
__device__ __forceinline__ void load_jet(const T* __restrict__ s_in, const T* __restrict__ v_in,
                                         const T* __restrict__ p, const uint8_t* __restrict__ mask, int B, int N,
                                         int b, T* nd, T* pj, uint8_t* mk) {
  JetRegs<T> r;
  load_jet_issue<T, C, DEC>(s_in, v_in, p, mask, B, N, b, r);
  load_jet_commit<T, C, DEC>(s_in, v_in, p, mask, B, N, b, r, nd, pj, mk);
}

// Radial parameters -> LDS.  encoder: a,b,c, Wt[k][R] (transposed so that the R outputs of one basis
// function are contiguous), bias[R] with R index = lin*2C + (2c+z).  decoder: bias[R] only, where both
// planes of channel c carry the same real bias (position_levels.py:184-188 with an all-zero mask).
template <typename T, int C, bool DEC>
__device__ __forceinline__ void load_radial(const T* ra, const T* rb, const T* rc, const T* w0, const T* b0,
                                            const T* w1, const T* b1, T* rp) {
  using L = Carve<C, DEC>;
  const int tid = threadIdx.x;
  if (DEC) {
    for (int e = tid; e < L::R; e += BLOCK) {
      int lin = e / (2 * C), c = (e - lin * 2 * C) >> 1;
      rp[L::RBIAS + e] = lin ? b1[c] : b0[c];
    }
  } else {
    for (int e = tid; e < NB; e += BLOCK) {
      rp[L::RA + e] = ra[e];
      rp[L::RB + e] = rb[e];
      rp[L::RC + e] = rc[e];
    }
    for (int e = tid; e < NB * L::R; e += BLOCK) {
      int k = e / L::R, r = e - k * L::R;
      int lin = r / (2 * C), f = r - lin * 2 * C;
      rp[L::RW + e] = (lin ? w1 : w0)[f * NB + k];
    }
    for (int e = tid; e < L::R; e += BLOCK) {
      int lin = e / (2 * C), f = e - lin * 2 * C;
      rp[L::RBIAS + e] = (lin ? b1 : b0)[f];
    }
  }
}

// Geometry of one ordered pair (i, j): q = canonical(p_i - p_j); encoder also the signed norm + mask.
template <typename T, bool DEC>
struct PairGeom {
  cx<T> q[4];
  T nrm;
  bool on;   // edge unmasked (encoder); decoder edges are always "masked" (radial == bias)
};

template <typename T, bool DEC>
__device__ __forceinline__ PairGeom<T, DEC> pair_geom(const T* pi, const T* pjj, bool mi, bool mj) {
  PairGeom<T, DEC> g;
  if (DEC) {
#pragma unroll
    for (int m = 0; m < 4; ++m) g.q[m] = {pi[m] - pjj[m], pi[4 + m] - pjj[4 + m]};
    g.nrm = T(0);
    g.on = false;
  } else {
    T d0 = pi[0] - pjj[0], d1 = pi[1] - pjj[1], d2 = pi[2] - pjj[2], d3 = pi[3] - pjj[3];
    T nsq;
    g.nrm = signed_norm(d0, d1, d2, d3, nsq);
    g.on = mi && mj && (g.nrm != T(0));
    const T h = rsqrt2<T>();
    g.q[0] = {d0, T(0)};
    g.q[1] = {d1 * h, -d2 * h};
    g.q[2] = {d3, T(0)};
    g.q[3] = {-d1 * h, -d2 * h};
  }
  return g;
}

// Radial network of one pair: rad[lin*2C + 2c + z].  Masked pairs keep the Linear bias
// (position_levels.py:144-149: the mask zeroes the basis, not the output).
template <typename T, int C, bool DEC>
__device__ __forceinline__ void radial_eval(const T* rp, T nrm, bool on, T (&rad)[4 * C]) {
  using L = Carve<C, DEC>;
#pragma unroll
  for (int r = 0; r < L::R; ++r) rad[r] = rp[L::RBIAS + r];
  if (!DEC) {
    if (on) {
      for (int k = 0; k < NB; ++k) {
        T t = rp[L::RC + k] * nrm;
        T u = (T(1) + t * t) + T(1e-16);
        T beta = rp[L::RB + k] * (T(1) / u) + rp[L::RA + k];
        const T* w = rp + L::RW + k * L::R;
#pragma unroll
        for (int r = 0; r < L::R; ++r) rad[r] += w[r] * beta;
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ cx<T> ld_cx(const T* base, int off_r, int off_i) {
  return {base[off_r], base[off_i]};
}

// <a, b> * 2 = a0 b0 + a1 b3 - a2 b2 + a3 b1  (the 1/2 of the CG coefficient is applied by the caller)
template <typename T>
__device__ __forceinline__ cx<T> bil2(const cx<T> (&a)[4], const cx<T> (&b)[4]) {
  cx<T> r = cmul(a[0], b[0]);
  cfma(r, a[1], b[3]);
  cx<T> t = cmul(a[2], b[2]);
  r.r -= t.r;
  r.i -= t.i;
  cfma(r, a[3], b[1]);
  return r;
}

// metric-permuted copy: tilde(x)[m] such that <a, b> = 1/2 sum_m a[m] * tilde(b)[m]
template <typename T>
__device__ __forceinline__ void metric_perm(const cx<T> (&x)[4], cx<T> (&y)[4]) {
  y[0] = x[0];
  y[1] = x[3];
  y[2] = {-x[2].r, -x[2].i};
  y[3] = x[1];
}

// Five reciprocals for the price of one: 1 / u_s from the reciprocal of the product (prefix products up, suffix products down).
// The five Lorentzian bells of a lane are 1 / (1 + c_s^2 |n|^2 + 1e-16), u_s >= 1 and far from overflow in the product; v_rcp_f64
// runs at a quarter of the fp64 rate and each refined reciprocal is 32 cycles of the datapath the pair sweeps are bound by:
// 5 x 32 -> 32 + 12 multiplies = 80 cycles per lane and tile.  Each result carries three or four roundings instead of one (a few
// 1e-16 relative).  A product that leaves the range (each u_s around 1e60: unnormalised momenta times a large c; or a NaN input)
// takes the five divisions one by one -- a branch no realistic jet enters; without it the overflow turned five values that
// should be ~0 into NaN (inf * 0 in the Newton step).
__device__ __forceinline__ void rcp5(const double (&u)[5], double (&r)[5]) {
  const double p1 = u[0] * u[1], p2 = p1 * u[2], p3 = p2 * u[3], p4 = p3 * u[4];
  if (__builtin_expect(!(p4 < 0x1p+1000), 0)) {
#pragma unroll
    for (int s = 0; s < 5; ++s) r[s] = 1.0 / u[s];
    return;
  }
  double t = __builtin_amdgcn_rcp(p4);
  double e = __builtin_fma(-p4, t, 1.0);
  t = __builtin_fma(t, e, t);
  e = __builtin_fma(-p4, t, 1.0);
  t = __builtin_fma(t, e, t);                  // 1 / (u0 u1 u2 u3 u4)
  r[4] = t * p3;  t *= u[4];                   // t = 1 / (u0 .. u3)
  r[3] = t * p2;  t *= u[3];
  r[2] = t * p1;  t *= u[2];                   // t = 1 / (u0 u1)
  r[1] = t * u[0];
  r[0] = t * u[1];
}

}  // namespace lgn
