// lgn-autoencoder_amd/csrc/generic_moments2.hip -- the N^2 part of a table-driven level (see generic_moments.hip for the
// operator), v2 for jets that fit in LDS (N <= 40): CHANNEL-OUTERMOST sweeps.
//
// v1 keeps all channels of a 16-pair tile in flight (lane = (pair, channel group)) and therefore has to cut the component
// axis q into chunks of 5 -- the radial network is evaluated once per chunk (4x at maxdim 3) -- and its backward kernels read
// the 200 doubles of dU of every (pair, channel) straight from global memory, i.e. N times each.  Here a workgroup owns a jet
// and loops over the channels; per channel c it
//   1. evaluates the radial functions of all N^2 pairs once into LDS:  R[i][j] = (R0r, R0i, R1r, R1i)    (one thread per pair)
//   2. stages that channel's slice of the features / of dU in LDS (10 KB / 48 KB at N = 30, Q = 20)
//   3. sweeps the pairs with every lane owning COMPLETE sums: wave = group of QPT components, lane = (row, half of the partner
//      range); the two halves meet by one DPP add at the end.  No quad reductions, no idle channel slots, no re-reads.
// Forward:   U[i][c][q][k]  = sum_j X_j[c][q] e_k(i,j)[c]
// Backward:  dX[j][c][q]   += sum_i sum_k dU[i][c][q][k] conj(e_k(i,j)[c])                           (j-centric sweep)
//            G_k(i,j)[c]    = sum_q dU[i][c][q][k] conj(X_j[c][q])  ->  G_R0, G_R1 of the pair          (i-centric sweep)
//            encoder: G_R0 / G_R1 go to a global buffer [B][N*N][4C] consumed by the radial-parameter reduction kernel
//            (the [4C x pairs] . [pairs x 42] matrix-core GEMM of v1, now without the q loop); decoder: bias sums and d p.
#include <stdlib.h>

#include "level_dev.hpp"
#include "ops.hpp"

namespace lgn {
namespace m2 {

constexpr int MAXN = 32;      // lane >> 1 indexes the row: 32 rows per sweep

__device__ __forceinline__ double fast_rcp(double u) {
  double r = __builtin_amdgcn_rcp(u);
  double e = __builtin_fma(-u, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-u, r, 1.0);
  return __builtin_fma(r, e, r);
}
// add the value held by the neighbouring lane (lane ^ 1)
__device__ __forceinline__ double pair_sum(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return v + __hiloint2double(hi, lo);
}

// LDS image of a jet: positions, mask, and per channel the pair table R
struct Jet {
  double* pj;        // [N][PS]
  uint8_t* mk;       // [N]
  double* Rl;        // [N][RP][4]   RP = row pitch in pairs (odd multiple keeps the 32-byte reads of a wave on distinct banks)
  double* wl;        // encoder: [20][8] = a_k, b_k, c_k^2, pad, w(R0r), w(R0i), w(R1r), w(R1i) of the current channel; [160..163] biases
};
__host__ __device__ inline int row_pitch(int N) { return (N | 1) + ((((N | 1) & 3) == 1) ? 0 : 2); }   // == 1 mod 4

// One channel of a tile-blocked tensor ([tile][C][Qx][2][64], node n = 64 tile + lane) -> LDS rows dst[node * pitch + 2 r + plane],
// r = 0 .. Qx - 1.  lane = particle of the jet (a jet's particles are consecutive lanes of one or two tiles: every load is a
// coalesced run), waves take rows r round robin; no index arithmetic per element, the loads of an unrolled group in flight together.
template <int NW = 4>      // waves of the workgroup
__device__ __forceinline__ void stage_tb(const double* __restrict__ src, int Qx, int C, int c, int b, int N, double* dst, int pitch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane >= N) return;
  const int n = b * N + lane;
  const double* __restrict__ s = src + ((size_t)(n >> 6) * C + c) * Qx * 128 + (n & 63);
  double* d = dst + (size_t)lane * pitch;
  int r = wave;
  for (; r + 3 * NW < Qx; r += 4 * NW) {
    double v[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u][0] = s[(r + NW * u) * 128]; v[u][1] = s[(r + NW * u) * 128 + 64]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { d[2 * (r + NW * u)] = v[u][0]; d[2 * (r + NW * u) + 1] = v[u][1]; }
  }
  for (; r < Qx; r += NW) { d[2 * r] = s[r * 128]; d[2 * r + 1] = s[r * 128 + 64]; }
}

template <bool DEC>
__device__ __forceinline__ void load_jet(const GenArgs& a, int b, Jet& J) {
  const int N = a.N, B = a.B;
  if (DEC) {
    const size_t plane_p = (size_t)B * N * 4;
    const double* p0 = a.p + (size_t)b * N * 4;
    for (int e = threadIdx.x; e < N * 4; e += BLOCK) {
      const int j = e >> 2, m = e & 3;
      J.pj[j * 8 + m] = p0[e];
      J.pj[j * 8 + 4 + m] = p0[plane_p + e];
    }
  } else {
    const double* p0 = a.p + (size_t)b * N * 4;
    for (int e = threadIdx.x; e < N * 4; e += BLOCK) J.pj[e] = p0[e];
    for (int e = threadIdx.x; e < N; e += BLOCK) J.mk[e] = a.mask[(size_t)b * N + e];
  }
}

// radial constants of channel c -> LDS (encoder)
__device__ __forceinline__ void load_channel_consts(const GenArgs& a, int c, double* wl) {
  for (int k = threadIdx.x; k < NB; k += BLOCK) {
    const double cc = a.rc[k];
    wl[k * 8 + 0] = a.ra[k];
    wl[k * 8 + 1] = a.rb[k];
    wl[k * 8 + 2] = cc * cc;
    wl[k * 8 + 3] = 0.0;
    wl[k * 8 + 4] = a.w0[(2 * c) * NB + k];
    wl[k * 8 + 5] = a.w0[(2 * c + 1) * NB + k];
    wl[k * 8 + 6] = a.w1[(2 * c) * NB + k];
    wl[k * 8 + 7] = a.w1[(2 * c + 1) * NB + k];
  }
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    wl[NB * 8 + q] = ((q >> 1) ? a.b1 : a.b0)[2 * c + (q & 1)];
  }
}

// R table of channel c for all ordered pairs (i, j): one thread per pair.  Masked pairs carry the Linear bias
// (position_levels.py:144-149); the decoder's radial functions ARE the biases (lgn_decoder.py:335-340).
template <bool DEC>
__device__ __forceinline__ void fill_R(const GenArgs& a, int c, const Jet& J, int RP) {
  const int N = a.N;
  if (DEC) {
    const double b0 = a.b0[c], b1 = a.b1[c];
    for (int p = threadIdx.x; p < N * N; p += BLOCK) {
      const int i = p / N, j = p - i * N;
      double* r = J.Rl + ((size_t)i * RP + j) * 4;
      r[0] = b0; r[1] = b0; r[2] = b1; r[3] = b1;
    }
    return;
  }
  for (int p = threadIdx.x; p < N * N; p += BLOCK) {
    const int i = p / N, j = p - i * N;
    const double* pi = J.pj + i * 4;
    const double* pq = J.pj + j * 4;
    const double d0 = pi[0] - pq[0], d1 = pi[1] - pq[1], d2 = pi[2] - pq[2], d3 = pi[3] - pq[3];
    const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
    const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;
    const double an = fabs(nsq);
    const bool on = J.mk[i] != 0 && J.mk[j] != 0 && nsq != 0.0;
    double r0 = J.wl[NB * 8 + 0], r1 = J.wl[NB * 8 + 1], r2 = J.wl[NB * 8 + 2], r3 = J.wl[NB * 8 + 3];
    if (on) {
#pragma unroll 4
      for (int k = 0; k < NB; ++k) {
        const double* w = J.wl + k * 8;
        const double beta = __builtin_fma(w[1], fast_rcp((1.0 + w[2] * an) + 1e-16), w[0]);
        r0 = __builtin_fma(w[4], beta, r0);
        r1 = __builtin_fma(w[5], beta, r1);
        r2 = __builtin_fma(w[6], beta, r2);
        r3 = __builtin_fma(w[7], beta, r3);
      }
    }
    double* r = J.Rl + ((size_t)i * RP + j) * 4;
    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
  }
}

// Encoder: the radial functions see a pair through |p_i - p_j|^2 and the two masks only, R(i, j) = R(j, i): the table for i <= j at
// tri_index(i, j), one thread per UNORDERED pair -- half the radial evaluations (a third of the forward kernel's instructions) and
// half the LDS of fill_R.  Same arithmetic per pair ((p_i - p_j)^2 is the same number either way round).
__device__ __forceinline__ int tri_index(int i, int j) {
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  return hi * (hi + 1) / 2 + lo;
}
__device__ __forceinline__ void fill_R_sym(const GenArgs& a, const Jet& J, int nthreads = BLOCK) {
  const int N = a.N, NU = N * (N + 1) / 2;
  for (int u = threadIdx.x; u < NU; u += nthreads) {
    int hi = (int)((sqrt(8.0 * (double)u + 1.0) - 1.0) * 0.5);
    while ((hi + 1) * (hi + 2) / 2 <= u) ++hi;
    while (hi * (hi + 1) / 2 > u) --hi;
    const int lo = u - hi * (hi + 1) / 2;
    const double* pi = J.pj + lo * 4;
    const double* pq = J.pj + hi * 4;
    const double d0 = pi[0] - pq[0], d1 = pi[1] - pq[1], d2 = pi[2] - pq[2], d3 = pi[3] - pq[3];
    const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
    const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;
    const double an = fabs(nsq);
    const bool on = J.mk[lo] != 0 && J.mk[hi] != 0 && nsq != 0.0;
    double r0 = J.wl[NB * 8 + 0], r1 = J.wl[NB * 8 + 1], r2 = J.wl[NB * 8 + 2], r3 = J.wl[NB * 8 + 3];
    if (on) {
#pragma unroll 4
      for (int k = 0; k < NB; ++k) {
        const double* w = J.wl + k * 8;
        const double beta = __builtin_fma(w[1], fast_rcp((1.0 + w[2] * an) + 1e-16), w[0]);
        r0 = __builtin_fma(w[4], beta, r0);
        r1 = __builtin_fma(w[5], beta, r1);
        r2 = __builtin_fma(w[6], beta, r2);
        r3 = __builtin_fma(w[7], beta, r3);
      }
    }
    double* r = J.Rl + (size_t)u * 4;
    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
  }
}

// canonical components of p_i - p_j (encoder: real Cartesian input; decoder: complex canonical input)
template <bool DEC>
__device__ __forceinline__ void rel_q(const double* pi, const double* pq, cx<double> (&q)[4]) {
  if (DEC) {
#pragma unroll
    for (int m = 0; m < 4; ++m) q[m] = {pi[m] - pq[m], pi[4 + m] - pq[4 + m]};
  } else {
    const double d0 = pi[0] - pq[0], d1 = pi[1] - pq[1], d2 = pi[2] - pq[2], d3 = pi[3] - pq[3];
    const double h = rsqrt2<double>();
    q[0] = {d0, 0.0};
    q[1] = {d1 * h, -d2 * h};
    q[2] = {d3, 0.0};
    q[3] = {-d1 * h, -d2 * h};
  }
}
__device__ __forceinline__ void edge_from_R(const double* r, const cx<double> (&q)[4], cx<double> (&e)[5]) {
  e[0] = {r[0] - r[1], r[0] + r[1]};               // R0 (1 + i)
  const cx<double> R1 = {r[2], r[3]};
#pragma unroll
  for (int m = 0; m < 4; ++m) e[1 + m] = cmul(R1, q[m]);
}

// =========================================================================================================
// forward
// =========================================================================================================
// (round 6: the QPT = 5 instantiation took 196 registers -- two waves per SIMD, PMC 1.6 resident, 42 % of a wave's life waiting; at
// three waves per SIMD (168 registers, 16 spilled outside the pair loop) 72.7 / 106.8 -> 69.3 / 94.8 us; four: 94 spilled, 2.4 x slower)
template <bool DEC, int QPT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(QPT >= 5 ? 3 : 1, QPT >= 5 ? 3 : 8))) void moments_fwd2_kernel(GenArgs a) {
  constexpr int PS = DEC ? 8 : 4;
  const int N = a.N, B = a.B, Q = a.Q, C = a.C, RP = row_pitch(N);
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  Jet J;
  J.Rl = reinterpret_cast<double*>(smem_raw);                  // decoder: N * RP * 4; encoder: the pairs i <= j only (fill_R_sym)
  double* xs = J.Rl + (DEC ? (size_t)N * RP * 4 : (size_t)(N * (N + 1) / 2) * 4);      // [N][Q][2] of the current channel
  J.pj = xs + (size_t)N * Q * 2;
  J.wl = J.pj + (size_t)N * PS;
  J.mk = reinterpret_cast<uint8_t*>(J.wl + NB * 8 + 4);
  load_jet<DEC>(a, b, J);
  const size_t plane = (size_t)B * N * C * Q;
  const int half = lane & 1, i = lane >> 1;                    // 32 rows x 2 halves of the partner range
  const bool iok = i < N;
  const int ii = iok ? i : N - 1;
  const int jmid = (N + 1) >> 1, jb = half ? jmid : 0, je = half ? N : jmid;
  for (int c = blockIdx.y; c < C; c += gridDim.y) {        // encoder: one workgroup per (jet, channel) -- grid.y = C; decoder sums run over c here
    __syncthreads();                                           // previous channel's sweeps are done with Rl / xs / wl
    if (!DEC) load_channel_consts(a, c, J.wl);
    if (a.tb) {
      stage_tb(a.X, Q, C, c, b, N, xs, 2 * Q);
    } else {
      for (int e = threadIdx.x; e < N * Q; e += BLOCK) {
        const int q = e % Q, j = e / Q;
        xs[2 * (j * Q + q)] = a.X[feat_index(false, plane, C, Q, b * N + j, c, q, 0)];
        xs[2 * (j * Q + q) + 1] = a.X[feat_index(false, plane, C, Q, b * N + j, c, q, 1)];
      }
    }
    __syncthreads();
    if (DEC) fill_R<DEC>(a, c, J, RP);
    else fill_R_sym(a, J);
    __syncthreads();
    for (int q0 = wave * QPT; q0 < Q; q0 += 4 * QPT) {         // wave-uniform
      cx<double> acc[QPT][5];
#pragma unroll
      for (int x = 0; x < QPT; ++x)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[x][k] = {0, 0};
      double pi[PS];
#pragma unroll
      for (int m = 0; m < PS; ++m) pi[m] = J.pj[ii * PS + m];
      for (int j = jb; j < je; ++j) {
        cx<double> q[4], e[5];
        rel_q<DEC>(pi, J.pj + j * PS, q);
        edge_from_R(J.Rl + (DEC ? ((size_t)ii * RP + j) : (size_t)tri_index(ii, j)) * 4, q, e);
        const double* xj = xs + ((size_t)j * Q + q0) * 2;
#pragma unroll
        for (int x = 0; x < QPT; ++x) {
          if (q0 + x < Q) {
            const cx<double> xv = {xj[2 * x], xj[2 * x + 1]};
#pragma unroll
            for (int k = 0; k < 5; ++k) cfma(acc[x][k], xv, e[k]);
          }
        }
      }
      double* u = a.U + ((((size_t)b * N + ii) * C + c) * Q + q0) * 10;
#pragma unroll
      for (int x = 0; x < QPT; ++x)
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const double sr = pair_sum(acc[x][k].r), si = pair_sum(acc[x][k].i);
          if (half == 0 && iok && q0 + x < Q) {
            if (a.tb) {
              a.U[feat_index(true, 0, C, 5 * Q, b * N + ii, c, (q0 + x) * 5 + k, 0)] = sr;
              a.U[feat_index(true, 0, C, 5 * Q, b * N + ii, c, (q0 + x) * 5 + k, 1)] = si;
            } else {
              u[(x * 5 + k) * 2] = sr;
              u[(x * 5 + k) * 2 + 1] = si;
            }
          }
        }
    }
  }
}

// =========================================================================================================
// backward, j-centric:  dX[j][c][q] += sum_i sum_k dU[i][c][q][k] conj(e_k(i,j)[c])      (+ decoder d p_j)
// =========================================================================================================
template <bool DEC, int QPT>
__global__ __launch_bounds__(BLOCK) void moments_bwd_nodes2_kernel(GenArgs a) {
  constexpr int PS = DEC ? 8 : 4;
  const int N = a.N, B = a.B, Q = a.Q, C = a.C, RP = N;          // lanes differ in j here: consecutive 32-byte entries, no padding
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  Jet J;
  J.Rl = reinterpret_cast<double*>(smem_raw);                  // N * RP * 4
  double* gu = J.Rl + (size_t)N * RP * 4;                      // [N][Q*5][2] of the current channel
  J.pj = gu + (size_t)N * Q * 10;
  J.wl = J.pj + (size_t)N * PS;
  double* gq = J.wl + NB * 8 + 4;                              // decoder: [4 waves][N][8]  d p_j partials of the component groups
  J.mk = reinterpret_cast<uint8_t*>(gq + (DEC ? 4 * N * 8 : 0));
  load_jet<DEC>(a, b, J);
  const size_t plane = (size_t)B * N * C * Q;
  const int half = lane & 1, j = lane >> 1;
  const bool jok = j < N;
  const int jj = jok ? j : N - 1;
  const int imid = (N + 1) >> 1, ib = half ? imid : 0, ie = half ? N : imid;
  cx<double> Gq[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};         // decoder: gradient w.r.t. q = p_i - p_j, summed over i, c, q
  for (int c = blockIdx.y; c < C; c += gridDim.y) {        // encoder: one workgroup per (jet, channel) -- grid.y = C; decoder sums run over c here
    __syncthreads();
    if (!DEC) load_channel_consts(a, c, J.wl);
    if (a.tb) {
      stage_tb(a.gU, 5 * Q, C, c, b, N, gu, Q * 10);
    } else {
      for (int e = threadIdx.x; e < N * Q * 10; e += BLOCK) {
        const int i = e / (Q * 10), r = e - i * Q * 10;
        gu[e] = a.gU[(((size_t)b * N + i) * C + c) * Q * 10 + r];
      }
    }
    __syncthreads();
    fill_R<DEC>(a, c, J, RP);
    __syncthreads();
    for (int q0 = wave * QPT; q0 < Q; q0 += 4 * QPT) {
      cx<double> acc[QPT], xme[QPT];
#pragma unroll
      for (int x = 0; x < QPT; ++x) {
        acc[x] = {0, 0};
        xme[x] = {0, 0};
        if (DEC && q0 + x < Q)
          xme[x] = {a.X[feat_index(a.tb, plane, C, Q, b * N + jj, c, q0 + x, 0)], a.X[feat_index(a.tb, plane, C, Q, b * N + jj, c, q0 + x, 1)]};
      }
      double pme[PS];
#pragma unroll
      for (int m = 0; m < PS; ++m) pme[m] = J.pj[jj * PS + m];
      for (int i = ib; i < ie; ++i) {
        cx<double> q[4], e[5];
        rel_q<DEC>(J.pj + i * PS, pme, q);
        const double* r = J.Rl + ((size_t)i * RP + jj) * 4;
        edge_from_R(r, q, e);
        const cx<double> R1 = {r[2], r[3]};
        const double* gi = gu + ((size_t)i * Q + q0) * 10;
#pragma unroll
        for (int x = 0; x < QPT; ++x) {
          if (q0 + x < Q) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              const cx<double> gv = {gi[(x * 5 + k) * 2], gi[(x * 5 + k) * 2 + 1]};
              cfmac(acc[x], gv, e[k]);
              if (DEC && k > 0) cfmac(Gq[k - 1], cmulc(gv, xme[x]), R1);   // G_e1[m] = dU[q][1+m] conj(X_j[q]);  G_q[m] += G_e1[m] conj(R1)
            }
          }
        }
      }
#pragma unroll
      for (int x = 0; x < QPT; ++x) {
        const double sr = pair_sum(acc[x].r), si = pair_sum(acc[x].i);
        if (half == 0 && jok && q0 + x < Q) {
          a.gX[feat_index(a.tb, plane, C, Q, b * N + j, c, q0 + x, 0)] += sr;
          a.gX[feat_index(a.tb, plane, C, Q, b * N + j, c, q0 + x, 1)] += si;
        }
      }
    }
  }
  if (DEC) {     // d p_j = - sum of G_q over partners, channels and components: halves by DPP, component groups (waves) through LDS
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double sr = pair_sum(Gq[m].r), si = pair_sum(Gq[m].i);
      if (half == 0 && jok) {
        gq[((size_t)wave * N + j) * 8 + m] = sr;
        gq[((size_t)wave * N + j) * 8 + 4 + m] = si;
      }
    }
    __syncthreads();
    const size_t plp = (size_t)B * N * 4;
    for (int e = threadIdx.x; e < N * 8; e += BLOCK) {
      const int jn = e >> 3, r = e & 7;
      const double s = (gq[((size_t)0 * N + jn) * 8 + r] + gq[((size_t)1 * N + jn) * 8 + r]) +
                       (gq[((size_t)2 * N + jn) * 8 + r] + gq[((size_t)3 * N + jn) * 8 + r]);
      a.g_p[(r >> 2) * plp + ((size_t)b * N + jn) * 4 + (r & 3)] -= s;
    }
  }
}

// =========================================================================================================
// backward, i-centric: gradient of the pair's radial values,  G_k(i,j) = sum_q dU[i][q][k] conj(X_j[q])
//   encoder: Gbuf[b][i*N + j][4C]: per channel (G_R0r, G_R0i, G_R1r, G_R1i), consumed by moments_rad_reduce2_kernel
//   decoder: bias-gradient sums (one partial row per jet) and d p_i
// wave = group of partners? no: lane = (row i, half of the partner range), wave = quarter of ... the q loop is the
// reduction here, so the four waves split the PARTNER range instead (each lane: one row, N/8 partners, all q).
// =========================================================================================================
__host__ __device__ inline int g2_row_pitch(int Q) { return Q * 10 + 2; }

template <bool DEC>
__global__ __launch_bounds__(BLOCK) void moments_bwd_G2_kernel(GenArgs a, double* Gbuf) {
  constexpr int PS = DEC ? 8 : 4;
  const int N = a.N, B = a.B, Q = a.Q, C = a.C;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  Jet J;                                                       // (no pair table: the gradient w.r.t. R does not depend on R)
  J.Rl = nullptr;
  // lane = row i reads ITS row of dU: a row pitch of Q * 10 doubles (1600 B at Q = 20) would put all 32 rows on two bank
  // groups; + 2 doubles makes the 16-byte reads of eight consecutive rows hit eight different bank quads
  const int GS = g2_row_pitch(Q);
  double* gu = reinterpret_cast<double*>(smem_raw);            // [N][GS]: (q, k, plane) of row n
  double* xs = gu + (size_t)N * GS;                            // [N][Q][2]
  J.pj = xs + (size_t)N * Q * 2;
  J.wl = J.pj + (size_t)N * PS;
  double* red = J.wl + NB * 8 + 4;                             // decoder: [8 groups][N][8] d p_i partials | [8][2] bias partials per channel
  J.mk = reinterpret_cast<uint8_t*>(red + (DEC ? 8 * N * 8 + 8 * 2 * 8 : 0));
  load_jet<DEC>(a, b, J);
  const size_t plane = (size_t)B * N * C * Q;
  const int grp = wave * 2 + (lane & 1), i = lane >> 1;        // 8 partner groups
  const bool iok = i < N;
  const int ii = iok ? i : N - 1;
  const int per = (N + 7) >> 3, jb = grp * per, je = min(N, jb + per);
  cx<double> Gq[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  for (int c = blockIdx.y; c < C; c += gridDim.y) {        // encoder: one workgroup per (jet, channel) -- grid.y = C; decoder sums run over c here
    __syncthreads();
    if (a.tb) {
      stage_tb(a.gU, 5 * Q, C, c, b, N, gu, GS);
      stage_tb(a.X, Q, C, c, b, N, xs, 2 * Q);
    } else {
      for (int e = threadIdx.x; e < N * Q * 10; e += BLOCK) {
        const int n = e / (Q * 10), r = e - n * Q * 10;
        gu[(size_t)n * GS + r] = a.gU[(((size_t)b * N + n) * C + c) * Q * 10 + r];
      }
      for (int e = threadIdx.x; e < N * Q; e += BLOCK) {
        const int q = e % Q, n = e / Q;
        xs[2 * (n * Q + q)] = a.X[feat_index(false, plane, C, Q, b * N + n, c, q, 0)];
        xs[2 * (n * Q + q) + 1] = a.X[feat_index(false, plane, C, Q, b * N + n, c, q, 1)];
      }
    }
    __syncthreads();
    double pi[PS];
#pragma unroll
    for (int m = 0; m < PS; ++m) pi[m] = J.pj[ii * PS + m];
    double dB0 = 0.0, dB1 = 0.0;
    const double* gi = gu + (size_t)ii * GS;
    // component loop outermost: the lane's own dU values of a component (5 complex) are read once for ALL of its partners
    // (read per partner, the LDS reads -- 6 per 20 fused multiply-adds -- were the limit, not the arithmetic)
    constexpr int JP = (MAXN + 7) / 8;                         // partners per lane, at most
    cx<double> ge[JP][5];
#pragma unroll
    for (int t = 0; t < JP; ++t)
#pragma unroll
      for (int k = 0; k < 5; ++k) ge[t][k] = {0, 0};
    for (int x = 0; x < Q; ++x) {
      cx<double> gv[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) gv[k] = {gi[x * 10 + 2 * k], gi[x * 10 + 2 * k + 1]};
#pragma unroll
      for (int t = 0; t < JP; ++t) {
        const int j = min(jb + t, N - 1);                      // (partners beyond the group's range are computed and dropped)
        const cx<double> xv = {xs[((size_t)j * Q + x) * 2], xs[((size_t)j * Q + x) * 2 + 1]};
#pragma unroll
        for (int k = 0; k < 5; ++k) cfmac(ge[t][k], gv[k], xv);
      }
    }
#pragma unroll
    for (int t = 0; t < JP; ++t) {
      const int j = jb + t;
      if (j >= je) continue;
      cx<double> q[4];
      rel_q<DEC>(pi, J.pj + j * PS, q);
      cx<double> gR1 = {0, 0};
#pragma unroll
      for (int m = 0; m < 4; ++m) cfmac(gR1, ge[t][1 + m], q[m]);
      const double G0r = ge[t][0].r + ge[t][0].i, G0i = ge[t][0].i - ge[t][0].r;     // e0 = R0 (1 + i)  ->  G_R0 = G_e0 conj(1 + i)
      if (DEC) {
        const cx<double> R1 = {a.b1[c], a.b1[c]};
#pragma unroll
        for (int m = 0; m < 4; ++m) cfmac(Gq[m], ge[t][1 + m], R1);
        if (iok) {
          dB0 += G0r + G0i;              // R0 = b0 (1 + i) on both planes: d b0 = Re G_R0 + Im G_R0
          dB1 += gR1.r + gR1.i;
        }
      } else if (iok) {
        double* g = Gbuf + (((size_t)b * N * N + (size_t)i * N + j) * C + c) * 4;
        g[0] = G0r; g[1] = G0i; g[2] = gR1.r; g[3] = gR1.i;
      }
    }
    if (DEC) {      // bias gradients of channel c: sum over all lanes of the workgroup
      for (int m = 1; m < 64; m <<= 1) { dB0 += shfl_xor(dB0, m); dB1 += shfl_xor(dB1, m); }
      double* bred = red + 8 * N * 8;
      if (lane == 0) { bred[(c * 4 + wave) * 2] = dB0; bred[(c * 4 + wave) * 2 + 1] = dB1; }
    }
  }
  if (DEC) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (iok) {
        red[((size_t)grp * N + i) * 8 + m] = Gq[m].r;
        red[((size_t)grp * N + i) * 8 + 4 + m] = Gq[m].i;
      }
    }
    __syncthreads();
    const size_t plp = (size_t)B * N * 4;
    for (int e = tid; e < N * 8; e += BLOCK) {
      const int n = e >> 3, r = e & 7;
      double s = 0.0;
      for (int g = 0; g < 8; ++g) s += red[((size_t)g * N + n) * 8 + r];
      a.g_p[(r >> 2) * plp + ((size_t)b * N + n) * 4 + (r & 3)] += s;
    }
    const double* bred = red + 8 * N * 8;
    double* part = a.part_rad + (size_t)b * rad_partial_size(C, true);
    if (tid < 2 * C) {
      const int lin = tid / C, ch = tid - lin * C;
      part[tid] = (bred[(ch * 4 + 0) * 2 + lin] + bred[(ch * 4 + 1) * 2 + lin]) + (bred[(ch * 4 + 2) * 2 + lin] + bred[(ch * 4 + 3) * 2 + lin]);
    }
  }
}


// =========================================================================================================
// encoder backward, both sweeps in ONE kernel (round 6).  moments_bwd_nodes2 and moments_bwd_G2 each staged the same 48 KB slice of
// dU (147 MB per level at cfg5) and ran one sweep over it; side by side their LDS images do not fit two workgroups per CU (the
// pair table alone is 29 KB).  The radial functions depend on the pair through |p_i - p_j|^2 and the two masks only: R(i, j) =
// R(j, i) -- the table is kept for i <= j (15 KB, half the radial evaluations), and one staging of dU and X serves
//   sweep 1 (j-centric):  dX[j][q] += sum_i sum_k dU[i][q][k] conj(e_k(i, j))
//   sweep 2 (i-centric):  G_k(i, j) = sum_q dU[i][q][k] conj(X_j[q])  ->  Gbuf (moments_rad_reduce2_kernel)
// with the arithmetic of the two kernels above, term for term.
// =========================================================================================================
// EIGHT waves per workgroup (round 6, second step): the LDS image allows two workgroups per CU, which at four waves each left two
// waves per SIMD and 58 % of a wave's life waiting (PMC) -- with eight, four.  Sweep 1: wave = (component group, half of the source
// range), the two halves of a (j, component) sum meet through LDS (in the pair table's place: sweep 2 does not need it); sweep 2:
// sixteen partner groups.
// (measured: 252.7 / 160.9 -> 238.4 / 156.0 us at Q = 20; the first level, Q = 5, has three component groups for four and LOSES with
// eight waves -- 58.2 -> 68.8 us -- and keeps four)
// doubles at the head of the LDS image: the symmetric pair table, or sweep 1's exchange block where that is larger (small jets)
__host__ __device__ inline size_t enc2_head_doubles(int N, int qpt) {
  const size_t tab = (size_t)(N * (N + 1) / 2) * 4, xch = (size_t)4 * N * qpt * 2;
  return tab > xch ? tab : xch;
}
template <int QPT, int ENC2_NW>
__global__ __launch_bounds__(64 * ENC2_NW) void moments_bwd_enc2_kernel(GenArgs a, double* Gbuf) {
  static_assert(ENC2_NW == 4 || ENC2_NW == 8, "four component groups x one or two halves of the source range");
  constexpr int NT = 64 * ENC2_NW, NPART = ENC2_NW / 2;          // source-range parts: lane halves x wave halves
  const int N = a.N, B = a.B, Q = a.Q, C = a.C;
  const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int GS = g2_row_pitch(Q);
  Jet J;
  J.Rl = reinterpret_cast<double*>(smem_raw);                  // [N (N + 1) / 2][4]: R(i, j) for i <= j at tri_index(i, j); then sweep 1's exchange
  double* gu = J.Rl + enc2_head_doubles(N, QPT);               // [N][GS]: (q, k, plane) of row n
  double* xs = gu + (size_t)N * GS;                            // [N][Q][2]
  J.pj = xs + (size_t)N * Q * 2;
  J.wl = J.pj + (size_t)N * 4;
  J.mk = reinterpret_cast<uint8_t*>(J.wl + NB * 8 + 4);
  {
    const double* p0 = a.p + (size_t)b * N * 4;
    for (int e = tid; e < N * 4; e += NT) J.pj[e] = p0[e];
    for (int e = tid; e < N; e += NT) J.mk[e] = a.mask[(size_t)b * N + e];
  }
  if (tid < BLOCK) load_channel_consts(a, c, J.wl);
  const size_t plane = (size_t)B * N * C * Q;
  if (a.tb) {
    stage_tb<ENC2_NW>(a.gU, 5 * Q, C, c, b, N, gu, GS);
    stage_tb<ENC2_NW>(a.X, Q, C, c, b, N, xs, 2 * Q);
  } else {
    for (int e = tid; e < N * Q * 10; e += NT) {
      const int n = e / (Q * 10), r = e - n * Q * 10;
      gu[(size_t)n * GS + r] = a.gU[(((size_t)b * N + n) * C + c) * Q * 10 + r];
    }
    for (int e = tid; e < N * Q; e += NT) {
      const int q = e % Q, n = e / Q;
      xs[2 * (n * Q + q)] = a.X[feat_index(false, plane, C, Q, b * N + n, c, q, 0)];
      xs[2 * (n * Q + q) + 1] = a.X[feat_index(false, plane, C, Q, b * N + n, c, q, 1)];
    }
  }
  __syncthreads();
  fill_R_sym(a, J, NT);
  __syncthreads();
  // ---- sweep 1, j-centric: lane = (node j, half), wave = (group of QPT components, half of the source range) ----
  const int half = lane & 1, j = lane >> 1;
  const bool jok = j < N;
  const int jj = jok ? j : N - 1;
  const int qg = wave & 3, ipart = wave >> 2, part = 2 * ipart + half;
  const int iper = (N + NPART - 1) / NPART, ib = part * iper, ie = min(N, ib + iper);
  constexpr int NQG = 1;        // component groups a wave runs through: Q <= 4 QPT for both instantiations (Q = 20 / QPT = 5, Q = 5 / QPT = 2)
  cx<double> acc[QPT];
  const int q0 = qg * QPT;
  {
    double pme[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pme[m] = J.pj[jj * 4 + m];
#pragma unroll
    for (int x = 0; x < QPT; ++x) acc[x] = {0, 0};
    if (q0 < Q) {
      for (int i = ib; i < ie; ++i) {
        cx<double> q[4], e[5];
        rel_q<false>(J.pj + i * 4, pme, q);
        edge_from_R(J.Rl + (size_t)tri_index(i, jj) * 4, q, e);
        const double* gi = gu + (size_t)i * GS + q0 * 10;
#pragma unroll
        for (int x = 0; x < QPT; ++x) {
          if (q0 + x < Q) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              const cx<double> gv = {gi[(x * 5 + k) * 2], gi[(x * 5 + k) * 2 + 1]};
              cfmac(acc[x], gv, e[k]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int x = 0; x < QPT; ++x) {
      acc[x].r = pair_sum(acc[x].r);
      acc[x].i = pair_sum(acc[x].i);
    }
  }
  (void)NQG;
  double* xch = J.Rl;                                          // [4 groups][N nodes][QPT][2]
  if constexpr (ENC2_NW == 8) {
    __syncthreads();                                           // every wave is done with the pair table
    if (ipart == 1 && half == 0 && jok) {
#pragma unroll
      for (int x = 0; x < QPT; ++x) {
        xch[((qg * N + j) * QPT + x) * 2] = acc[x].r;
        xch[((qg * N + j) * QPT + x) * 2 + 1] = acc[x].i;
      }
    }
    __syncthreads();
  }
  if (ipart == 0 && half == 0 && jok) {
#pragma unroll
    for (int x = 0; x < QPT; ++x) {
      if (q0 + x < Q) {
        const double or_ = ENC2_NW == 8 ? xch[((qg * N + j) * QPT + x) * 2] : 0.0, oi = ENC2_NW == 8 ? xch[((qg * N + j) * QPT + x) * 2 + 1] : 0.0;
        a.gX[feat_index(a.tb, plane, C, Q, b * N + j, c, q0 + x, 0)] += acc[x].r + or_;
        a.gX[feat_index(a.tb, plane, C, Q, b * N + j, c, q0 + x, 1)] += acc[x].i + oi;
      }
    }
  }
  // ---- sweep 2, i-centric: lane = (row i, partner group), the gradient of the pair's radial values ----
  {
    constexpr int NGRP = 2 * ENC2_NW;
    const int grp = wave * 2 + (lane & 1), i = lane >> 1;        // 8 or 16 partner groups
    const bool iok = i < N;
    const int ii = iok ? i : N - 1;
    const int per = (N + NGRP - 1) / NGRP, jb = grp * per, je = min(N, jb + per);
    double pi[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pi[m] = J.pj[ii * 4 + m];
    const double* gi = gu + (size_t)ii * GS;
    constexpr int JP = (MAXN + NGRP - 1) / NGRP;
    cx<double> ge[JP][5];
#pragma unroll
    for (int t = 0; t < JP; ++t)
#pragma unroll
      for (int k = 0; k < 5; ++k) ge[t][k] = {0, 0};
    for (int x = 0; x < Q; ++x) {
      cx<double> gv[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) gv[k] = {gi[x * 10 + 2 * k], gi[x * 10 + 2 * k + 1]};
#pragma unroll
      for (int t = 0; t < JP; ++t) {
        const int jx = min(jb + t, N - 1);
        const cx<double> xv = {xs[((size_t)jx * Q + x) * 2], xs[((size_t)jx * Q + x) * 2 + 1]};
#pragma unroll
        for (int k = 0; k < 5; ++k) cfmac(ge[t][k], gv[k], xv);
      }
    }
#pragma unroll
    for (int t = 0; t < JP; ++t) {
      const int jx = jb + t;
      if (jx >= je || !iok) continue;
      cx<double> q[4];
      rel_q<false>(pi, J.pj + jx * 4, q);
      cx<double> gR1 = {0, 0};
#pragma unroll
      for (int m = 0; m < 4; ++m) cfmac(gR1, ge[t][1 + m], q[m]);
      double* g = Gbuf + (((size_t)b * N * N + (size_t)i * N + jx) * C + c) * 4;
      g[0] = ge[t][0].r + ge[t][0].i;             // e0 = R0 (1 + i)  ->  G_R0 = G_e0 conj(1 + i)
      g[1] = ge[t][0].i - ge[t][0].r;
      g[2] = gR1.r;
      g[3] = gR1.i;
    }
  }
}

// =========================================================================================================
// decoder in separable form (SURVEY a-14; the same algebraic change as the SEP instantiations of level_fwd2 / level_bwd3 at
// maxdim 2, NOT a roofline gain).  The decoder's edge mask is identically zero (lgn_decoder.py:335-340), so its radial functions
// are the Linear biases: e0 = b0 (1 + i)^2 ... precisely R0 = b0 (1 + i), e0 = R0 (1 + i), R1 = b1 (1 + i), e_{1+m} = R1 (p_i - p_j)[m],
// constants per channel.  Every pair sum then separates into jet-level sums, O(N Q) per channel instead of O(N^2 Q):
//   forward    U[i][q][0] = e0 SX[q],   U[i][q][1+m] = R1 (P_i[m] SX[q] - SXP[q][m])        SX = sum_j X_j,  SXP = sum_j X_j P_j
//   backward   S[q][k] = sum_i dU[i][q][k],  SP[q][m] = sum_i dU[i][q][1+m] conj(P_i[m]):
//              dX[j][q] += conj(e0) S[q][0] + conj(R1) sum_m (SP[q][m] - conj(P_j[m]) S[q][1+m])
//              d p_i    += conj(R1) sum_q dU[i][q][1+m] conj(SX[q]),     d p_j -= conj(R1) sum_q S[q][1+m] conj(X_j[q])
//              d b0 = 2 Im A0,  A0 = sum_q S[q][0] conj(SX[q]);   d b1 = Re A1 + Im A1,  A1 = sum_{q,m} SP conj(SX) - S[1+m] conj(SXP)
// Momenta are centred on the jet mean first (differences are unchanged; the sums then do not cancel digits the pair sweep would
// keep under large boosts).  LGN_AMD_DEC_PAIRWISE=1 keeps the pair sweeps above.  One workgroup per jet, channels in a loop.
// =========================================================================================================
__device__ __forceinline__ void load_centred(const GenArgs& a, int b, double* pc) {       // pc [N][8]: centred complex canonical momenta
  const int N = a.N;
  const size_t plane_p = (size_t)a.B * N * 4;
  const double* p0 = a.p + (size_t)b * N * 4;
  __shared__ double mean[8];
  if (threadIdx.x < 8) {
    const int m = threadIdx.x & 3, z = threadIdx.x >> 2;
    double s = 0.0;
    for (int j = 0; j < N; ++j) s += p0[z * plane_p + j * 4 + m];
    mean[threadIdx.x] = s / N;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * 8; e += BLOCK) {
    const int j = e >> 3, r = e & 7;
    pc[e] = p0[(r >> 2) * plane_p + j * 4 + (r & 3)] - mean[r];
  }
}

__global__ __launch_bounds__(BLOCK) void moments_dec_sep_fwd_kernel(GenArgs a) {
  const int N = a.N, Q = a.Q, C = a.C, b = blockIdx.x;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* pc = reinterpret_cast<double*>(smem_raw);            // [N][8]
  double* xs = pc + (size_t)N * 8;                             // [N][Q][2]
  double* sx = xs + (size_t)N * Q * 2;                         // [Q][5][2]: SX | SXP[0..3]
  load_centred(a, b, pc);
  const size_t plane = (size_t)a.B * N * C * Q;
  for (int c = 0; c < C; ++c) {
    __syncthreads();
    for (int e = threadIdx.x; e < N * Q; e += BLOCK) {
      const int q = a.tb ? e / N : e % Q, j = a.tb ? e % N : e / Q;
      xs[2 * (j * Q + q)] = a.X[feat_index(a.tb, plane, C, Q, b * N + j, c, q, 0)];
      xs[2 * (j * Q + q) + 1] = a.X[feat_index(a.tb, plane, C, Q, b * N + j, c, q, 1)];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Q * 5; e += BLOCK) {         // jet-level sums, in particle order
      const int q = e / 5, k = e - q * 5;
      cx<double> s = {0, 0};
      for (int j = 0; j < N; ++j) {
        const cx<double> x = {xs[2 * (j * Q + q)], xs[2 * (j * Q + q) + 1]};
        if (k == 0) { s.r += x.r; s.i += x.i; }
        else cfma(s, x, cx<double>{pc[j * 8 + k - 1], pc[j * 8 + 4 + k - 1]});
      }
      sx[2 * e] = s.r;
      sx[2 * e + 1] = s.i;
    }
    __syncthreads();
    const double b0 = a.b0[c], b1 = a.b1[c];
    const cx<double> e0 = {0.0, 2.0 * b0}, R1 = {b1, b1};      // R0 (1 + i) with R0 = b0 (1 + i);  R1 = b1 (1 + i)
    for (int e = threadIdx.x; e < N * Q * 5; e += BLOCK) {
      const int i = e / (Q * 5), r = e - i * Q * 5, q = r / 5, k = r - q * 5;
      const cx<double> SX = {sx[2 * (q * 5)], sx[2 * (q * 5) + 1]};
      cx<double> u;
      if (k == 0) u = cmul(e0, SX);
      else {
        const cx<double> P = {pc[i * 8 + k - 1], pc[i * 8 + 4 + k - 1]};
        cx<double> t = cmul(P, SX);
        t.r -= sx[2 * (q * 5 + k)];
        t.i -= sx[2 * (q * 5 + k) + 1];
        u = cmul(R1, t);
      }
      if (a.tb) {
        a.U[feat_index(true, 0, C, 5 * Q, b * N + i, c, r, 0)] = u.r;
        a.U[feat_index(true, 0, C, 5 * Q, b * N + i, c, r, 1)] = u.i;
      } else {
        double* dst = a.U + ((((size_t)b * N + i) * C + c) * Q) * 10 + 2 * r;
        dst[0] = u.r;
        dst[1] = u.i;
      }
    }
  }
}

__global__ __launch_bounds__(BLOCK) void moments_dec_sep_bwd_kernel(GenArgs a) {
  const int N = a.N, Q = a.Q, C = a.C, b = blockIdx.x, tid = threadIdx.x;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* pc = reinterpret_cast<double*>(smem_raw);            // [N][8]
  double* xs = pc + (size_t)N * 8;                             // [N][Q][2]
  double* gu = xs + (size_t)N * Q * 2;                         // [N][Q*5][2]
  double* sx = gu + (size_t)N * Q * 10;                        // [Q][5][2]  SX | SXP
  double* sg = sx + (size_t)Q * 10;                            // [Q][5][2]  S
  double* sp = sg + (size_t)Q * 10;                            // [Q][4][2]  SP
  double* gp = sp + (size_t)Q * 8;                             // [N][8]     d p accumulated over the channels
  double* ab = gp + (size_t)N * 8;                             // [Q][4]     per-q terms of A0, A1
  load_centred(a, b, pc);
  for (int e = tid; e < N * 8; e += BLOCK) gp[e] = 0.0;
  const size_t plane = (size_t)a.B * N * C * Q;
  double* part = a.part_rad + (size_t)b * rad_partial_size(C, true);
  for (int c = 0; c < C; ++c) {
    __syncthreads();
    for (int e = tid; e < N * Q; e += BLOCK) {
      const int q = a.tb ? e / N : e % Q, j = a.tb ? e % N : e / Q;
      xs[2 * (j * Q + q)] = a.X[feat_index(a.tb, plane, C, Q, b * N + j, c, q, 0)];
      xs[2 * (j * Q + q) + 1] = a.X[feat_index(a.tb, plane, C, Q, b * N + j, c, q, 1)];
    }
    if (a.tb) {
      for (int e = tid; e < N * Q * 10; e += BLOCK) {
        const int r = e / N, i = e - r * N;
        gu[(size_t)i * Q * 10 + r] = a.gU[feat_index(true, 0, C, 5 * Q, b * N + i, c, r >> 1, r & 1)];
      }
    } else {
      for (int e = tid; e < N * Q * 10; e += BLOCK) {
        const int i = e / (Q * 10), r = e - i * Q * 10;
        gu[e] = a.gU[(((size_t)b * N + i) * C + c) * Q * 10 + r];
      }
    }
    __syncthreads();
    // jet-level sums (particle order): SX, SXP from X;  S, SP from dU
    for (int e = tid; e < Q * 14; e += BLOCK) {
      const int q = e / 14, w = e - q * 14;
      cx<double> s = {0, 0};
      if (w < 5) {                                             // SX (w = 0), SXP[w - 1]
        for (int j = 0; j < N; ++j) {
          const cx<double> x = {xs[2 * (j * Q + q)], xs[2 * (j * Q + q) + 1]};
          if (w == 0) { s.r += x.r; s.i += x.i; }
          else cfma(s, x, cx<double>{pc[j * 8 + w - 1], pc[j * 8 + 4 + w - 1]});
        }
        sx[2 * (q * 5 + w)] = s.r;  sx[2 * (q * 5 + w) + 1] = s.i;
      } else if (w < 10) {                                     // S[k], k = w - 5
        const int k = w - 5;
        for (int i = 0; i < N; ++i) { s.r += gu[((size_t)i * Q + q) * 10 + 2 * k]; s.i += gu[((size_t)i * Q + q) * 10 + 2 * k + 1]; }
        sg[2 * (q * 5 + k)] = s.r;  sg[2 * (q * 5 + k) + 1] = s.i;
      } else {                                                 // SP[m], m = w - 10
        const int mm = w - 10;
        for (int i = 0; i < N; ++i)
          cfmac(s, cx<double>{gu[((size_t)i * Q + q) * 10 + 2 * (1 + mm)], gu[((size_t)i * Q + q) * 10 + 2 * (1 + mm) + 1]},
                cx<double>{pc[i * 8 + mm], pc[i * 8 + 4 + mm]});
        sp[2 * (q * 4 + mm)] = s.r;  sp[2 * (q * 4 + mm) + 1] = s.i;
      }
    }
    __syncthreads();
    const double b0 = a.b0[c], b1 = a.b1[c];
    const cx<double> e0 = {0.0, 2.0 * b0}, R1 = {b1, b1};
    // d X[j][q] += conj(e0) S[q][0] + conj(R1) sum_m (SP[q][m] - conj(P_j[m]) S[q][1+m])
    for (int e = tid; e < N * Q; e += BLOCK) {
      const int q = a.tb ? e / N : e % Q, j = a.tb ? e % N : e / Q;
      cx<double> t = {0, 0};
#pragma unroll
      for (int mm = 0; mm < 4; ++mm) {
        t.r += sp[2 * (q * 4 + mm)];
        t.i += sp[2 * (q * 4 + mm) + 1];
        const cx<double> ps = cmulc(cx<double>{sg[2 * (q * 5 + 1 + mm)], sg[2 * (q * 5 + 1 + mm) + 1]}, cx<double>{pc[j * 8 + mm], pc[j * 8 + 4 + mm]});
        t.r -= ps.r;
        t.i -= ps.i;
      }
      cx<double> g = cmulc(cx<double>{sg[2 * (q * 5)], sg[2 * (q * 5) + 1]}, e0);
      cfmac(g, t, R1);
      a.gX[feat_index(a.tb, plane, C, Q, b * N + j, c, q, 0)] += g.r;
      a.gX[feat_index(a.tb, plane, C, Q, b * N + j, c, q, 1)] += g.i;
    }
    // d p_i[m] += conj(R1) sum_q dU[i][q][1+m] conj(SX[q]);   d p_j[m] -= conj(R1) sum_q S[q][1+m] conj(X_j[q])
    for (int e = tid; e < N * 4; e += BLOCK) {
      const int n = e >> 2, mm = e & 3;
      cx<double> t = {0, 0};
      for (int q = 0; q < Q; ++q) {
        cfmac(t, cx<double>{gu[((size_t)n * Q + q) * 10 + 2 * (1 + mm)], gu[((size_t)n * Q + q) * 10 + 2 * (1 + mm) + 1]},
              cx<double>{sx[2 * (q * 5)], sx[2 * (q * 5) + 1]});
        cx<double> u = cmulc(cx<double>{sg[2 * (q * 5 + 1 + mm)], sg[2 * (q * 5 + 1 + mm) + 1]}, cx<double>{xs[2 * (n * Q + q)], xs[2 * (n * Q + q) + 1]});
        t.r -= u.r;
        t.i -= u.i;
      }
      const cx<double> g = cmulc(t, R1);
      gp[n * 8 + mm] += g.r;
      gp[n * 8 + 4 + mm] += g.i;
    }
    // bias gradients: A0 = sum_q S[q][0] conj(SX[q]);  A1 = sum_{q,m} SP[q][m] conj(SX[q]) - S[q][1+m] conj(SXP[q][m])
    for (int q = tid; q < Q; q += BLOCK) {
      const cx<double> SX = {sx[2 * (q * 5)], sx[2 * (q * 5) + 1]};
      const cx<double> a0 = cmulc(cx<double>{sg[2 * (q * 5)], sg[2 * (q * 5) + 1]}, SX);
      cx<double> a1 = {0, 0};
#pragma unroll
      for (int mm = 0; mm < 4; ++mm) {
        cfmac(a1, cx<double>{sp[2 * (q * 4 + mm)], sp[2 * (q * 4 + mm) + 1]}, SX);
        const cx<double> u = cmulc(cx<double>{sg[2 * (q * 5 + 1 + mm)], sg[2 * (q * 5 + 1 + mm) + 1]},
                                   cx<double>{sx[2 * (q * 5 + 1 + mm)], sx[2 * (q * 5 + 1 + mm) + 1]});
        a1.r -= u.r;
        a1.i -= u.i;
      }
      ab[q * 4] = a0.r;  ab[q * 4 + 1] = a0.i;  ab[q * 4 + 2] = a1.r;  ab[q * 4 + 3] = a1.i;
    }
    __syncthreads();
    if (tid < 2) {
      double s = 0.0;
      for (int q = 0; q < Q; ++q) s += tid == 0 ? 2.0 * ab[q * 4 + 1] : ab[q * 4 + 2] + ab[q * 4 + 3];
      part[tid * C + c] = s;                                   // dB0[c] | dB1[c]
    }
  }
  __syncthreads();
  const size_t plp = (size_t)a.B * N * 4;
  for (int e = tid; e < N * 8; e += BLOCK) {
    const int n = e >> 3, r = e & 7;
    a.g_p[(r >> 2) * plp + ((size_t)b * N + n) * 4 + (r & 3)] += gp[e];
  }
}

// ---------------------------------------------------------------------------------------------------------
// encoder: radial-parameter sums from the pair gradients Gbuf -- the [4C x pairs] . [pairs x 42] GEMM of
// moments_bwd_rad_kernel (generic_moments.hip) on the matrix cores, without the q loop.
//   T1[r][k] = sum_p G[p][r] on rho_k    T2[r][k] = sum_p G[p][r] on n^2 rho_k^2    S[r] = sum_p G on    dB[r] = sum_p G
// ---------------------------------------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(BLOCK) void moments_rad_reduce2_kernel(GenArgs a, const double* Gbuf) {
  constexpr int NG = (C + 3) / 4;
  const int N = a.N;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* pj = reinterpret_cast<double*>(smem_raw);            // N * 4
  double* red = pj + (size_t)N * 4;                            // 4 waves * 64 lanes * NG * 12
  uint8_t* mk = reinterpret_cast<uint8_t*>(red + 4 * 64 * NG * 12);
  {
    const double* p0 = a.p + (size_t)b * N * 4;
    for (int e = tid; e < N * 4; e += BLOCK) pj[e] = p0[e];
    for (int e = tid; e < N; e += BLOCK) mk[e] = a.mask[(size_t)b * N + e];
  }
  __syncthreads();
  // MFMA operands: A[row r][k = pair slot], B[k = pair slot][col]; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15].
  // rows r' = cc + 4 * quantity (quantity = R0r, R0i, R1r, R1i) of channel group g; 4 pairs per MFMA (k dimension).
  double ck2[3], akk = 0;   // this lane's basis column(s): col = lane & 15 of the three column blocks [rho_0..15 | n2rho2_0..15 | rho_16..19, n2rho2_16..19, on, 1]
  (void)akk;
  const int col = lane & 15, kq = lane >> 4;
  {
    const double c0 = a.rc[col];
    ck2[0] = c0 * c0;
    const int k2 = col < 4 ? 16 + col : (col < 8 ? 16 + col - 4 : 0);
    const double c2 = a.rc[k2];
    ck2[1] = c2 * c2;
    ck2[2] = 0;
  }
  v4d T[NG][3];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < 3; ++t) T[g][t] = v4d{0, 0, 0, 0};
  // The basis functions see a pair only through |p_i - p_j|^2 and the two masks: rho(i, j) = rho(j, i), so the sum over ORDERED pairs
  // of G(i, j) rho(i, j) is the sum over UNORDERED pairs {i <= j} of (G(i, j) + G(j, i)) rho -- half the basis evaluations and matrix
  // instructions (level_bwd3.hip uses the same symmetry inside its sweep).  The pair gradients of the next step are requested before
  // this step's matrix instructions (the loop is a chain of global round trips otherwise).
  const int npairs = N * N, nuno = N * (N + 1) / 2;
  auto decode = [&](int u, int& i, int& j, bool& ok) {         // u -> (i <= j), row-major over the lower triangle of (j, i)
    ok = u < nuno;
    const int uu = ok ? u : nuno - 1;
    int r = (int)((sqrt(8.0 * (double)uu + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= uu) ++r;
    while (r * (r + 1) / 2 > uu) --r;
    j = r;
    i = uu - r * (r + 1) / 2;
  };
  auto fetch = [&](int i, int j, bool ok, double (&av)[2 * NG]) {        // (the two halves are added where they are used: no wait here)
    const double* g1 = Gbuf + ((size_t)b * npairs + (size_t)i * N + j) * C * 4;
    const double* g2 = Gbuf + ((size_t)b * npairs + (size_t)j * N + i) * C * 4;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int rr = lane & 15, quant = rr >> 2, ch = 4 * g + (rr & 3);
      const bool live = ok && ch < C;
      av[2 * g] = live ? g1[ch * 4 + quant] : 0.0;
      av[2 * g + 1] = (live && i != j) ? g2[ch * 4 + quant] : 0.0;
    }
  };
  // operands of the next PF steps in flight (one step ahead left the loop a chain of ~30 global round trips per wave: 34 - 46 us for
  // 60 - 90 MB of pair gradients; three ahead: 30 - 40 us)
  constexpr int PF = 6;
  int ci[PF], cj[PF];
  bool cok[PF];
  double cav[PF][2 * NG];
  // this lane's pairs are u = 4 wave + kq + 16 t: decoded once (the square root), then walked -- i += 16, rows j hold j + 1 pairs
  int wi, wj, wu = wave * 4 + kq;
  {
    bool ok0;
    decode(wu, wi, wj, ok0);
  }
  auto next_pair = [&](int& i, int& j, bool& ok) {
    ok = wu < nuno;
    i = ok ? wi : 0;
    j = ok ? wj : 0;
    wu += 16;
    wi += 16;
    while (wi > wj) { wi -= wj + 1; ++wj; }
  };
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    next_pair(ci[d], cj[d], cok[d]);
    fetch(ci[d], cj[d], cok[d], cav[d]);
  }
  for (int u0 = wave * 4; u0 < nuno; u0 += 16) {               // 4 pairs per MFMA step, waves interleaved
    const int i = ci[0], j = cj[0];
    const bool ok = cok[0];
    double av[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) av[g] = cav[0][2 * g] + cav[0][2 * g + 1];
#pragma unroll
    for (int d = 0; d + 1 < PF; ++d) {
      ci[d] = ci[d + 1]; cj[d] = cj[d + 1]; cok[d] = cok[d + 1];
#pragma unroll
      for (int g = 0; g < 2 * NG; ++g) cav[d][g] = cav[d + 1][g];
    }
    next_pair(ci[PF - 1], cj[PF - 1], cok[PF - 1]);                      // (beyond the last pair: ok = false, the loads are of pair (0, 0))
    fetch(ci[PF - 1], cj[PF - 1], cok[PF - 1], cav[PF - 1]);
    const double* pi = pj + i * 4;
    const double* pq = pj + j * 4;
    const double d0 = pi[0] - pq[0], d1 = pi[1] - pq[1], d2 = pi[2] - pq[2], d3 = pi[3] - pq[3];
    const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
    const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;
    const double an = fabs(nsq);
    const bool on = ok && mk[i] != 0 && mk[j] != 0 && nsq != 0.0;
    // B fragments: three column blocks of 16
    const double rho0 = on ? fast_rcp((1.0 + ck2[0] * an) + 1e-16) : 0.0;
    const double rho2 = on ? fast_rcp((1.0 + ck2[1] * an) + 1e-16) : 0.0;
    double bv[3];
    bv[0] = rho0;
    bv[1] = an * rho0 * rho0;
    bv[2] = col < 4 ? rho2 : (col < 8 ? an * rho2 * rho2 : (col == 8 ? (on ? 1.0 : 0.0) : (col == 9 ? (ok ? 1.0 : 0.0) : 0.0)));
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int t = 0; t < 3; ++t) T[g][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g], bv[t], T[g][t], 0, 0, 0);
  }
  // D fragment: lane holds rows (lane >> 4) + 4 * reg = channel cc = lane >> 4 of group g, quantity reg; column lane & 15
  {
    double* mine = red + (size_t)(wave * 64 + lane) * NG * 12;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) mine[(g * 3 + t) * 4 + q] = T[g][t][q];
  }
  __syncthreads();
  double* part = a.part_rad + (size_t)blockIdx.x * rad_partial_size(C, false);
  if (wave == 0) {
    constexpr int R = 4 * C;
    const int cg = lane >> 4;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int ch = 4 * g + cg;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = (g * 3 + t) * 4 + q;
          const double v = (red[(size_t)(0 * 64 + lane) * NG * 12 + e] + red[(size_t)(1 * 64 + lane) * NG * 12 + e]) +
                           (red[(size_t)(2 * 64 + lane) * NG * 12 + e] + red[(size_t)(3 * 64 + lane) * NG * 12 + e]);
          if (ch >= C) continue;
          const int r = (q >> 1) * 2 * C + 2 * ch + (q & 1);
          if (t == 0) part[r * NB + col] = v;
          else if (t == 1) part[R * NB + r * NB + col] = v;
          else {
            if (col < 4) part[r * NB + 16 + col] = v;
            else if (col < 8) part[R * NB + r * NB + 16 + (col - 4)] = v;
            else if (col == 8) part[2 * R * NB + r] = v;
            else if (col == 9) part[2 * R * NB + R + r] = v;
          }
        }
    }
  }
}

template <typename K>
static int set_smem(K kern, size_t smem, const char* what) {
  if (smem > 160 * 1024) {
    set_error("%s needs %zu B of LDS (> 160 KiB)", what, smem);
    return -1;
  }
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e)); return (int)e; }
  }
  return 0;
}

static size_t base_smem(const GenArgs& a, bool dec, int pitch) {     // pair table (pitch 0: none) + positions + radial constants + mask
  const int PS = dec ? 8 : 4;
  return sizeof(double) * ((size_t)a.N * pitch * 4 + (size_t)a.N * PS + NB * 8 + 4) + a.N + 16;
}

template <bool DEC>
static int launch(const GenArgs& a, int which, double* Gbuf, hipStream_t st) {
  const size_t xs = sizeof(double) * (size_t)a.N * a.Q * 2, gu = sizeof(double) * (size_t)a.N * a.Q * 10;
  // encoder: channels are independent -> one workgroup per (jet, channel): 3x .. 6x more workgroups in flight to hide the staging
  // and the barriers; the decoder kernels (pair-sweep cross-check only) accumulate d p and the bias gradients over the channels
  const dim3 grid(a.B, DEC ? 1 : a.C);
  int rc;
  if (which == 0) {
    // (encoder: the symmetric pair table, N (N + 1) / 2 entries instead of N x pitch)
    const size_t smem = base_smem(a, DEC, row_pitch(a.N)) + xs - (DEC ? 0 : sizeof(double) * 4 * ((size_t)a.N * row_pitch(a.N) - (size_t)(a.N * (a.N + 1) / 2)));
    if (a.Q <= 8) {
      auto k = moments_fwd2_kernel<DEC, 2>;
      if ((rc = set_smem(k, smem, "moments_fwd2"))) return rc;
      hipLaunchKernelGGL(k, grid, dim3(BLOCK), smem, st, a);
    } else {
      auto k = moments_fwd2_kernel<DEC, 5>;
      if ((rc = set_smem(k, smem, "moments_fwd2"))) return rc;
      hipLaunchKernelGGL(k, grid, dim3(BLOCK), smem, st, a);
    }
  } else if (which == 1) {
    const size_t smem = base_smem(a, DEC, a.N) + gu + (DEC ? sizeof(double) * 4 * a.N * 8 : 0);
    if (a.Q <= 8) {
      auto k = moments_bwd_nodes2_kernel<DEC, 2>;
      if ((rc = set_smem(k, smem, "moments_bwd_nodes2"))) return rc;
      hipLaunchKernelGGL(k, grid, dim3(BLOCK), smem, st, a);
    } else {
      auto k = moments_bwd_nodes2_kernel<DEC, 5>;
      if ((rc = set_smem(k, smem, "moments_bwd_nodes2"))) return rc;
      hipLaunchKernelGGL(k, grid, dim3(BLOCK), smem, st, a);
    }
  } else {
    const size_t smem = base_smem(a, DEC, 0) + sizeof(double) * (size_t)a.N * g2_row_pitch(a.Q) + xs + (DEC ? sizeof(double) * (8 * a.N * 8 + 8 * 2 * 8) : 0);
    auto k = moments_bwd_G2_kernel<DEC>;
    if ((rc = set_smem(k, smem, "moments_bwd_G2"))) return rc;
    hipLaunchKernelGGL(k, grid, dim3(BLOCK), smem, st, a, Gbuf);
  }
  LGN_CHECK_LAUNCH();
  return 0;
}

// encoder backward, both sweeps (moments_bwd_enc2_kernel): LDS = symmetric pair table + padded dU rows + X + jet data
static size_t enc2_smem(const GenArgs& a) {
  return sizeof(double) * (enc2_head_doubles(a.N, a.Q <= 8 ? 2 : 5) + (size_t)a.N * g2_row_pitch(a.Q) + (size_t)a.N * a.Q * 2 + (size_t)a.N * 4 + NB * 8 + 4) +
         a.N + 16;
}
static int launch_enc2(const GenArgs& a, double* Gbuf, hipStream_t st) {
  const size_t smem = enc2_smem(a);
  const dim3 grid(a.B, a.C);
  int rc;
  if (a.Q <= 8) {
    auto k = moments_bwd_enc2_kernel<2, 4>;
    if ((rc = set_smem(k, smem, "moments_bwd_enc2"))) return rc;
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a, Gbuf);
  } else {
    auto k = moments_bwd_enc2_kernel<5, 8>;
    if ((rc = set_smem(k, smem, "moments_bwd_enc2"))) return rc;
    hipLaunchKernelGGL(k, grid, dim3(512), smem, st, a, Gbuf);
  }
  LGN_CHECK_LAUNCH();
  return 0;
}

template <int C>
static int launch_reduce(const GenArgs& a, const double* Gbuf, hipStream_t st) {
  constexpr int NG = (C + 3) / 4;
  const size_t smem = sizeof(double) * ((size_t)a.N * 4 + 4 * 64 * NG * 12) + a.N + 16;
  auto k = moments_rad_reduce2_kernel<C>;
  if (int rc = set_smem(k, smem, "moments_rad_reduce2")) return rc;
  hipLaunchKernelGGL(k, dim3(a.B), dim3(BLOCK), smem, st, a, Gbuf);
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace m2

// Does the channel-outermost form apply?  (the jet's pair table, one channel's dU slice and its X slice must fit in LDS)
bool moments2_fits(const GenArgs& a, int decoder) {
  if (a.N > m2::MAXN || a.C > 8) return false;
  const size_t worst = m2::base_smem(a, decoder != 0, m2::row_pitch(a.N)) + sizeof(double) * ((size_t)a.N * a.Q * 12 + 2 * (size_t)a.N) +
                       (decoder ? sizeof(double) * (8 * a.N * 8 + 8 * 2 * 8) : 0);
  return worst <= 160 * 1024;
}
// scratch (doubles) the encoder's i-centric backward needs for the pair gradients
size_t moments2_gbuf_doubles(int B, int N, int C) { return (size_t)B * N * N * C * 4; }

// which: 0 forward, 1 backward j-centric, 2 backward i-centric (Gbuf: encoder only, moments2_gbuf_doubles)
int moments_dec_sep_tb_dispatch(const GenArgs& a, int which, hipStream_t st);      // generic_moments_sep.hip

int moments2_dispatch(const GenArgs& a, int decoder, int which, double* Gbuf, hipStream_t st) {
  LGN_CHECK_ARG(a.B > 0 && a.N > 0 && a.Q > 0, "moments: empty input (B=%d N=%d Q=%d)", a.B, a.N, a.Q);
  LGN_CHECK_ARG(a.C >= 1 && a.C <= 8, "moments: C=%d unsupported (1..8)", a.C);
  if (decoder) {
    if (a.flags & LVL_DEC_PAIRWISE) return m2::launch<true>(a, which, nullptr, st);   // O(N^2) pair sweeps (cross-check of the separable form)
    if (which == 2) return 0;                               // the separable backward does both passes in one launch
    if (const int rc = moments_dec_sep_tb_dispatch(a, which, st); rc != -2) return rc;      // tile-blocked layouts: generic_moments_sep.hip
    const size_t base = sizeof(double) * ((size_t)a.N * 8 + (size_t)a.N * a.Q * 2);
    if (which == 0) {
      const size_t smem = base + sizeof(double) * (size_t)a.Q * 10;
      if (int rc = m2::set_smem(m2::moments_dec_sep_fwd_kernel, smem, "moments_dec_sep_fwd")) return rc;
      hipLaunchKernelGGL(m2::moments_dec_sep_fwd_kernel, dim3(a.B), dim3(BLOCK), smem, st, a);
    } else {
      const size_t smem = base + sizeof(double) * ((size_t)a.N * a.Q * 10 + (size_t)a.Q * 28 + (size_t)a.N * 8 + (size_t)a.Q * 4);
      if (int rc = m2::set_smem(m2::moments_dec_sep_bwd_kernel, smem, "moments_dec_sep_bwd")) return rc;
      hipLaunchKernelGGL(m2::moments_dec_sep_bwd_kernel, dim3(a.B), dim3(BLOCK), smem, st, a);
    }
    LGN_CHECK_LAUNCH();
    return 0;
  }
  // encoder backward: ONE kernel for both sweeps (which = 1; which = 2 is then the radial-parameter reduction alone) unless the
  // caller asks for the two-kernel form (LVL_MOMENTS_SPLIT: cross-check) or has no pair-gradient scratch for pass 1
  const bool merged = Gbuf && !(a.flags & LVL_MOMENTS_SPLIT) && a.Q <= 20;       // (a wave of the merged kernel owns ONE group of <= 5 components)
  if (which == 0) return m2::launch<false>(a, 0, nullptr, st);
  if (which == 1) return merged ? m2::launch_enc2(a, Gbuf, st) : m2::launch<false>(a, 1, nullptr, st);
  LGN_CHECK_ARG(Gbuf, "moments: the encoder's radial backward needs the pair-gradient scratch buffer");
  if (!merged)
    if (int rc = m2::launch<false>(a, 2, Gbuf, st)) return rc;
#define LGN_CASE(CC) case CC: return m2::launch_reduce<CC>(a, Gbuf, st);
  switch (a.C) {
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
    default: return -1;
  }
#undef LGN_CASE
}

}  // namespace lgn
