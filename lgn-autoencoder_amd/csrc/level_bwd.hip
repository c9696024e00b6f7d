// lgn-autoencoder_amd/csrc/level_bwd.hip -- backward of the fused message-passing level (maxdim = 2).
//
// The reference obtains these gradients by autograd through RadPolyTrig / GScalar*GVec / cg_product /
// CatMixReps (lgn/nn/position_levels.py:118-209, lgn/models/lgn_cg.py:167, lgn/cg_lib/cg_ops.py:135-298,
// lgn/nn/g_nn.py:260-278) which stores the N x N edge tensors.  Here the edges are *recomputed*:
//
//   1. level_bwd_mix     per node: CatMix^T -> grad of aggregate (g_ag), direct + power-term grads of the
//                        node features, and CatMix weight gradient partials.
//   2. level_bwd_nodes   "j-centric" pass: g_node_j += sum_i g_ag_i (x) conj(edge_ij)   (+ decoder dp_j)
//   3. level_bwd_rad_*   "i-centric" pass: radial-network parameter gradients (+ decoder dp_i)
//
// Complex convention: every map is holomorphic (no conjugation in the forward), so for out = f(z) the
// planar gradient is G_z = G_out * conj(f'(z)).
#include <stdlib.h>

#include "level_dev.hpp"
#include <type_traits>
#include "ops.hpp"

namespace lgn {

// g_ag scratch layout per node (scalars): [A3: C x {r,i}] [A4: C x 2] [A1: C x 4 x 2] [A2: C x 4 x 2]
template <int C> struct GA {
  static constexpr int A3 = 0, A4 = 2 * C, A1 = 4 * C, A2 = 12 * C, SIZE = 20 * C;
};

// =========================================================================================
// 1. CatMix / power backward
// =========================================================================================
template <typename T, int C>
__global__ __launch_bounds__(BLOCK) void level_bwd_mix_kernel(LevelBwdArgs<T> a) {
  constexpr int IT = 32;
  constexpr int K = 5 * C;
  constexpr int XS = K * 10;                    // per node: x0[K][2], x1[K][4][2]
  const int N = a.N, B = a.B, CO = a.CO;
  const int b = blockIdx.x, tile = blockIdx.y, tid = threadIdx.x;
  const int GS = CO * 10;                       // per node: g_s[CO][2], g_v[CO][4][2]

  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  T* wm = sm;                                   // [irrep][z][CO][K]
  T* xt = wm + 4 * CO * K;                      // IT * XS
  T* gt = xt + IT * XS;                         // IT * GS

  for (int e = tid; e < 2 * CO * K; e += BLOCK) {
    wm[e] = a.wm0[e];
    wm[2 * CO * K + e] = a.wm1[e];
  }
  const size_t plo = (size_t)B * N * CO;
  for (int e = tid; e < IT * CO; e += BLOCK) {
    int rl = e / CO, o = e - rl * CO, r = tile * IT + rl;
    T* g = gt + rl * GS;
    if (r < N) {
      size_t idx = ((size_t)b * N + r) * CO + o;
      g[o * 2] = a.g_s_out[idx];
      g[o * 2 + 1] = a.g_s_out[plo + idx];
      for (int m = 0; m < 4; ++m) {
        g[2 * CO + (o * 4 + m) * 2] = a.g_v_out[idx * 4 + m];
        g[2 * CO + (o * 4 + m) * 2 + 1] = a.g_v_out[plo * 4 + idx * 4 + m];
      }
    } else {
      for (int q = 0; q < 2; ++q) g[o * 2 + q] = T(0);
      for (int q = 0; q < 8; ++q) g[2 * CO + o * 8 + q] = T(0);
    }
  }
  __syncthreads();

  // ---- per (node, channel): gradient of the concatenated input, then node / power / aggregate split
  {
    const int c = tid & 7, rl = tid >> 3;
    const int r = tile * IT + rl;
    if (c < C) {
      T* x = xt + rl * XS;
      if (r < N) {
        const size_t pls = (size_t)B * N * C;
        const size_t e = ((size_t)b * N + r) * C + c;
        cx<T> s = {a.s_in[e], a.s_in[pls + e]};
        cx<T> v[4], vt[4];
        for (int m = 0; m < 4; ++m) v[m] = {a.v_in[e * 4 + m], a.v_in[pls * 4 + e * 4 + m]};
        metric_perm(v, vt);
        // the five cat slots owned by this channel: k = c, C+c (aggregate), 2C+c (node), 3C+c, 4C+c (power)
        cx<T> gx0[5], gx1[5][4];
        const T* g = gt + rl * GS;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          const int k = q * C + c;
          cx<T> acc0 = {T(0), T(0)}, acc1[4];
          for (int m = 0; m < 4; ++m) acc1[m] = {T(0), T(0)};
          for (int o = 0; o < CO; ++o) {
            cx<T> w0 = {wm[(0 * CO + o) * K + k], wm[(1 * CO + o) * K + k]};
            cx<T> w1 = {wm[2 * CO * K + (0 * CO + o) * K + k], wm[2 * CO * K + (1 * CO + o) * K + k]};
            cfmac(acc0, cx<T>{g[o * 2], g[o * 2 + 1]}, w0);
            for (int m = 0; m < 4; ++m)
              cfmac(acc1[m], cx<T>{g[2 * CO + (o * 4 + m) * 2], g[2 * CO + (o * 4 + m) * 2 + 1]}, w1);
          }
          gx0[q] = acc0;
          for (int m = 0; m < 4; ++m) gx1[q][m] = acc1[m];
        }
        // aggregate gradient -> scratch
        T* ga = a.g_ag + ((size_t)b * N + r) * GA<C>::SIZE;
        ga[GA<C>::A3 + 2 * c] = gx0[0].r;  ga[GA<C>::A3 + 2 * c + 1] = gx0[0].i;
        ga[GA<C>::A4 + 2 * c] = gx0[1].r;  ga[GA<C>::A4 + 2 * c + 1] = gx0[1].i;
        for (int m = 0; m < 4; ++m) {
          ga[GA<C>::A1 + (c * 4 + m) * 2] = gx1[0][m].r;  ga[GA<C>::A1 + (c * 4 + m) * 2 + 1] = gx1[0][m].i;
          ga[GA<C>::A2 + (c * 4 + m) * 2] = gx1[1][m].r;  ga[GA<C>::A2 + (c * 4 + m) * 2 + 1] = gx1[1][m].i;
        }
        // node block + power blocks: sq(0,0) = [<v,v>, s^2], sq(1,1) = [v s, s v]
        cx<T> gs = gx0[2];
        cx<T> two_gss = {T(2) * gx0[4].r, T(2) * gx0[4].i};
        cfmac(gs, two_gss, s);
        cx<T> gv[4];
        for (int m = 0; m < 4; ++m) {
          cx<T> gvs = {gx1[3][m].r + gx1[4][m].r, gx1[3][m].i + gx1[4][m].i};
          cfmac(gs, gvs, v[m]);
          gv[m] = gx1[2][m];
          cfmac(gv[m], gvs, s);
          cfmac(gv[m], gx0[3], vt[m]);
        }
        a.g_s_in[e] = gs.r;
        a.g_s_in[pls + e] = gs.i;
        for (int m = 0; m < 4; ++m) {
          a.g_v_in[e * 4 + m] = gv[m].r;
          a.g_v_in[pls * 4 + e * 4 + m] = gv[m].i;
        }
        // stage the concatenated forward input x for the weight gradient
        const size_t pa = (size_t)B * N * 2 * C;
        const size_t ea = ((size_t)b * N + r) * 2 * C;
        cx<T> vv = bil2(v, v);
        vv.r *= T(0.5);  vv.i *= T(0.5);
        cx<T> ss = cmul(s, s);
        cx<T> x0v[5] = {{a.ag0[ea + c], a.ag0[pa + ea + c]}, {a.ag0[ea + C + c], a.ag0[pa + ea + C + c]}, s, vv, ss};
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          x[(q * C + c) * 2] = x0v[q].r;
          x[(q * C + c) * 2 + 1] = x0v[q].i;
        }
        for (int m = 0; m < 4; ++m) {
          cx<T> vs = cmul(v[m], s);
          cx<T> x1v[5] = {{a.ag1[(ea + c) * 4 + m], a.ag1[(pa + ea + c) * 4 + m]},
                          {a.ag1[(ea + C + c) * 4 + m], a.ag1[(pa + ea + C + c) * 4 + m]}, v[m], vs, vs};
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            x[2 * K + ((q * C + c) * 4 + m) * 2] = x1v[q].r;
            x[2 * K + ((q * C + c) * 4 + m) * 2 + 1] = x1v[q].i;
          }
        }
      } else {
        for (int q = 0; q < 5; ++q) {
          for (int z = 0; z < 2; ++z) x[(q * C + c) * 2 + z] = T(0);
          for (int z = 0; z < 8; ++z) x[2 * K + (q * C + c) * 8 + z] = T(0);
        }
      }
    }
  }
  __syncthreads();

  // ---- CatMix weight gradient partial: dW[o][k] = sum_nodes g[o] * conj(x[k])
  {
    T* part = a.part_mix + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (4 * CO * K);
    for (int e = tid; e < CO * K; e += BLOCK) {
      const int o = e / K, k = e - o * K;
      cx<T> d0 = {T(0), T(0)}, d1 = {T(0), T(0)};
      for (int rl = 0; rl < IT; ++rl) {
        const T* g = gt + rl * GS;
        const T* x = xt + rl * XS;
        cfmac(d0, cx<T>{g[o * 2], g[o * 2 + 1]}, cx<T>{x[k * 2], x[k * 2 + 1]});
#pragma unroll
        for (int m = 0; m < 4; ++m)
          cfmac(d1, cx<T>{g[2 * CO + (o * 4 + m) * 2], g[2 * CO + (o * 4 + m) * 2 + 1]},
                cx<T>{x[2 * K + (k * 4 + m) * 2], x[2 * K + (k * 4 + m) * 2 + 1]});
      }
      part[(0 * CO + o) * K + k] = d0.r;
      part[(1 * CO + o) * K + k] = d0.i;
      part[2 * CO * K + (0 * CO + o) * K + k] = d1.r;
      part[2 * CO * K + (1 * CO + o) * K + k] = d1.i;
    }
  }
}

// =========================================================================================
// 2. j-centric pass: gradient w.r.t. the node features that were *sources* of messages
//    G_v[j][c][m] += sum_i gA1[i][c][m] conj(e0_ij[c]) + 1/2 gA3[i][c] conj(tilde(e1_ij[c])[m])
//    G_s[j][c]    += sum_i sum_m gA2[i][c][m] conj(e1_ij[c][m]) + gA4[i][c] conj(e0_ij[c])
//    decoder:  g_p[j] -= sum_i G_q_ij,   G_q_ij[m] = sum_c G_e1_ij[c][m] conj(R1[c])
// =========================================================================================
template <typename T, int C, int IS, bool DEC>
__global__ __launch_bounds__(BLOCK) void level_bwd_nodes_kernel(LevelBwdArgs<T> a) {
  using L = Carve<C, DEC>;
  constexpr int JT = BLOCK / IS;
  constexpr int R = L::R;
  const int N = a.N, B = a.B;
  const int b = blockIdx.x, tile = blockIdx.y, tid = threadIdx.x;

  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  T* ga = sm;                                    // N * 20C
  T* pj = ga + N * GA<C>::SIZE;                  // N * PS
  T* rp = pj + L::even(N * L::PS);               // RAD_SIZE
  uint8_t* mk = reinterpret_cast<uint8_t*>(rp + L::even(L::RAD_SIZE));

  {
    const T* src = a.g_ag + (size_t)b * N * GA<C>::SIZE;
    for (int e = tid; e < N * GA<C>::SIZE; e += BLOCK) ga[e] = src[e];
    if (DEC) {
      const size_t plane_p = (size_t)B * N * 4;
      const T* p0 = a.p + (size_t)b * N * 4;
      for (int e = tid; e < N * 4; e += BLOCK) {
        int j = e >> 2, m = e & 3;
        pj[j * 8 + m] = p0[e];
        pj[j * 8 + 4 + m] = p0[plane_p + e];
      }
    } else {
      const T* p0 = a.p + (size_t)b * N * 4;
      for (int e = tid; e < N * 4; e += BLOCK) pj[e] = p0[e];
      for (int e = tid; e < N; e += BLOCK) mk[e] = a.mask[(size_t)b * N + e];
    }
    load_radial<T, C, DEC>(a.ra, a.rb, a.rc, a.w0, a.b0, a.w1, a.b1, rp);
  }
  __syncthreads();

  const int jl = tid / IS, is = tid % IS;
  const int j = tile * JT + jl;
  const bool ok = j < N;
  const int jj = ok ? j : 0;
  T pme[L::PS];
#pragma unroll
  for (int m = 0; m < L::PS; ++m) pme[m] = pj[jj * L::PS + m];
  const bool mj = DEC ? false : (mk[jj] != 0);

  cx<T> Gs[C], Gv[C][4], Gq[4];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    Gs[c] = {T(0), T(0)};
#pragma unroll
    for (int m = 0; m < 4; ++m) Gv[c][m] = {T(0), T(0)};
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) Gq[m] = {T(0), T(0)};

  // own node features (decoder position gradient only)
  cx<T> sj[C], vtj[C][4];
  if (DEC) {
    const size_t pls = (size_t)B * N * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const size_t e = ((size_t)b * N + jj) * C + c;
      sj[c] = {a.s_in[e], a.s_in[pls + e]};
      cx<T> v[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) v[m] = {a.v_in[e * 4 + m], a.v_in[pls * 4 + e * 4 + m]};
      metric_perm(v, vtj[c]);
    }
  }

  if (ok) {
    for (int i = is; i < N; i += IS) {
      // ordered pair (i, j): q = canonical(p_i - p_j)
      PairGeom<T, DEC> g = pair_geom<T, DEC>(pj + i * L::PS, pme, DEC ? false : (mk[i] != 0), mj);
      T rad[R];
      radial_eval<T, C, DEC>(rp, g.nrm, g.on, rad);
      const T* gi = ga + i * GA<C>::SIZE;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        cx<T> R0 = {rad[2 * c], rad[2 * c + 1]};
        cx<T> R1 = {rad[2 * C + 2 * c], rad[2 * C + 2 * c + 1]};
        cx<T> e0 = {R0.r - R0.i, R0.r + R0.i};
        cx<T> e1[4], e1t[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) e1[m] = cmul(R1, g.q[m]);
        metric_perm(e1, e1t);
        cx<T> gA3 = {T(0.5) * gi[GA<C>::A3 + 2 * c], T(0.5) * gi[GA<C>::A3 + 2 * c + 1]};
        cx<T> gA4 = {gi[GA<C>::A4 + 2 * c], gi[GA<C>::A4 + 2 * c + 1]};
        cfmac(Gs[c], gA4, e0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          cx<T> gA1 = {gi[GA<C>::A1 + (c * 4 + m) * 2], gi[GA<C>::A1 + (c * 4 + m) * 2 + 1]};
          cx<T> gA2 = {gi[GA<C>::A2 + (c * 4 + m) * 2], gi[GA<C>::A2 + (c * 4 + m) * 2 + 1]};
          cfmac(Gv[c][m], gA1, e0);
          cfmac(Gv[c][m], gA3, e1t[m]);
          cfmac(Gs[c], gA2, e1[m]);
          if (DEC) {
            cx<T> ge1 = cmulc(gA2, sj[c]);
            cfmac(ge1, gA3, vtj[c][m]);
            cfmac(Gq[m], ge1, R1);
          }
        }
      }
    }
  }

  const size_t pls = (size_t)B * N * C;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    T sr = group_sum<IS>(Gs[c].r), si = group_sum<IS>(Gs[c].i);
    const size_t e = ((size_t)b * N + jj) * C + c;
    if (ok && is == 0) {
      a.g_s_in[e] += sr;
      a.g_s_in[pls + e] += si;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      T vr = group_sum<IS>(Gv[c][m].r), vi = group_sum<IS>(Gv[c][m].i);
      if (ok && is == 0) {
        a.g_v_in[e * 4 + m] += vr;
        a.g_v_in[pls * 4 + e * 4 + m] += vi;
      }
    }
  }
  if (DEC) {
    const size_t plp = (size_t)B * N * 4;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      T qr = group_sum<IS>(Gq[m].r), qi = group_sum<IS>(Gq[m].i);
      if (ok && is == 0) {
        a.g_p[((size_t)b * N + jj) * 4 + m] -= qr;
        a.g_p[plp + ((size_t)b * N + jj) * 4 + m] -= qi;
      }
    }
  }
}

// =========================================================================================
// 3a. i-centric pass, encoder: radial-network parameter gradient partials.
//   rad[r] = sum_k W[r][k] beta_k + bias[r],  beta_k = on * (b_k rho_k + a_k),  rho_k = 1/(1 + (c_k n)^2 + 1e-16)
//   With G[p][r] = dL/drad[r] of pair p, everything follows from four pair-reductions
//     T1[r][k] = sum_p G on rho_k      T2[r][k] = sum_p G on n^2 rho_k^2     S[r] = sum_p G on     dB[r] = sum_p G
//   (finalised by rad_finalize_kernel).  G is produced lane-per-pair, staged in LDS, and reduced by a
//   register-tiled outer-product accumulation (lane = (r, k-group)).
// =========================================================================================
template <int C> struct Stage {
  static constexpr int R = 4 * C;
  static constexpr int X = NB + 2;               // rho_k (masked), n^2, on
  static constexpr int STRIDE = ((R + X) | 1);   // odd stride (in scalars) spreads lanes over banks
  static constexpr int RPL = (R + 15) / 16;      // radial rows per consumer lane
};

template <typename T, int C>
__global__ __launch_bounds__(BLOCK) void level_bwd_rad_enc_kernel(LevelBwdArgs<T> a, int JT) {
  using L = Carve<C, false>;
  using S = Stage<C>;
  constexpr int JS = 8, IT = BLOCK / JS;
  constexpr int R = S::R;
  const int N = a.N, B = a.B;
  const int b = blockIdx.x, it = blockIdx.y, jt = blockIdx.z, tid = threadIdx.x;
  const int j0 = jt * JT;
  const int nj = min(JT, N - j0);

  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  T* nd = sm;                                     // JT * NS
  T* pjt = nd + L::even(JT * L::NS);              // JT * 4
  T* pit = pjt + JT * 4;                          // IT * 4
  T* gat = pit + IT * 4;                          // IT * 20C
  T* rc = gat + IT * GA<C>::SIZE;                 // NB (c_k)
  T* stg = rc + NB;                               // BLOCK * STRIDE
  T* red = stg + BLOCK * S::STRIDE;               // 4 waves * 64 lanes * (RPL*10 + 2)
  uint8_t* mkj = reinterpret_cast<uint8_t*>(red + 4 * 64 * (S::RPL * 10 + 2));
  uint8_t* mki = mkj + JT;

  {
    const size_t pls = (size_t)B * N * C;
    for (int e = tid; e < nj * C; e += BLOCK) {
      int jl = e / C, c = e - jl * C;
      size_t src = ((size_t)b * N + j0 + jl) * C + c;
      nd[jl * L::NS + c * 10 + 0] = a.s_in[src];
      nd[jl * L::NS + c * 10 + 1] = a.s_in[pls + src];
      for (int m = 0; m < 4; ++m) {
        nd[jl * L::NS + c * 10 + 2 + m] = a.v_in[src * 4 + m];
        nd[jl * L::NS + c * 10 + 6 + m] = a.v_in[pls * 4 + src * 4 + m];
      }
    }
    for (int e = tid; e < nj * 4; e += BLOCK) pjt[e] = a.p[((size_t)b * N + j0) * 4 + e];
    for (int e = tid; e < nj; e += BLOCK) mkj[e] = a.mask[(size_t)b * N + j0 + e];
    for (int e = tid; e < IT; e += BLOCK) {
      int i = it * IT + e;
      mki[e] = i < N ? a.mask[(size_t)b * N + i] : 0;
      for (int m = 0; m < 4; ++m) pit[e * 4 + m] = i < N ? a.p[((size_t)b * N + i) * 4 + m] : T(0);
    }
    for (int e = tid; e < IT * GA<C>::SIZE; e += BLOCK) {
      int il = e / GA<C>::SIZE, i = it * IT + il;
      gat[e] = i < N ? a.g_ag[((size_t)b * N + it * IT) * GA<C>::SIZE + e] : T(0);
    }
    for (int e = tid; e < NB; e += BLOCK) rc[e] = a.rc[e];
  }
  __syncthreads();

  const int il = tid / JS, js = tid % JS;
  const int i = it * IT + il;
  const bool row_ok = i < N;
  const T pi[4] = {pit[il * 4], pit[il * 4 + 1], pit[il * 4 + 2], pit[il * 4 + 3]};
  const bool mi = mki[il] != 0;
  const T* gi = gat + il * GA<C>::SIZE;

  // consumer role of this lane in the outer-product accumulation
  const int lane = tid & 63, wave = tid >> 6;
  const int xg = lane & 3, rg = lane >> 2;
  T acc1[S::RPL][5], acc2[S::RPL][5], accx[S::RPL];
#pragma unroll
  for (int t = 0; t < S::RPL; ++t) {
    accx[t] = T(0);
#pragma unroll
    for (int q = 0; q < 5; ++q) acc1[t][q] = acc2[t][q] = T(0);
  }

  const int iters = (JT + JS - 1) / JS;
  for (int t = 0; t < iters; ++t) {
    const int jl = js + JS * t;
    T* my = stg + tid * S::STRIDE;
    if (row_ok && jl < nj) {
      PairGeom<T, false> g = pair_geom<T, false>(pi, pjt + jl * 4, mi, mkj[jl] != 0);
      const T n2 = g.nrm * g.nrm;
      for (int k = 0; k < NB; ++k) {
        T tt = rc[k] * g.nrm;
        T u = (T(1) + tt * tt) + T(1e-16);
        my[R + k] = g.on ? T(1) / u : T(0);
      }
      my[R + NB] = n2;
      my[R + NB + 1] = g.on ? T(1) : T(0);
      const T* njp = nd + jl * L::NS;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        cx<T> s = {njp[c * 10], njp[c * 10 + 1]};
        cx<T> v[4], vt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = {njp[c * 10 + 2 + m], njp[c * 10 + 6 + m]};
        metric_perm(v, vt);
        cx<T> gA3 = {T(0.5) * gi[GA<C>::A3 + 2 * c], T(0.5) * gi[GA<C>::A3 + 2 * c + 1]};
        cx<T> gA4 = {gi[GA<C>::A4 + 2 * c], gi[GA<C>::A4 + 2 * c + 1]};
        cx<T> ge0 = cmulc(gA4, s);
        cx<T> gR1 = {T(0), T(0)};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          cx<T> gA1 = {gi[GA<C>::A1 + (c * 4 + m) * 2], gi[GA<C>::A1 + (c * 4 + m) * 2 + 1]};
          cx<T> gA2 = {gi[GA<C>::A2 + (c * 4 + m) * 2], gi[GA<C>::A2 + (c * 4 + m) * 2 + 1]};
          cfmac(ge0, gA1, v[m]);
          cx<T> ge1 = cmulc(gA2, s);
          cfmac(ge1, gA3, vt[m]);
          cfmac(gR1, ge1, g.q[m]);
        }
        // e0 = R0 (1 + i)  ->  G_R0 = G_e0 (1 - i)
        my[2 * c] = ge0.r + ge0.i;
        my[2 * c + 1] = ge0.i - ge0.r;
        my[2 * C + 2 * c] = gR1.r;
        my[2 * C + 2 * c + 1] = gR1.i;
      }
    } else {
      for (int e = 0; e < R + S::X; ++e) my[e] = T(0);
    }
    __syncthreads();
    // each wave reduces the 64 pairs it just staged
    const T* base = stg + (wave * 64) * S::STRIDE;
    for (int p = 0; p < 64; ++p) {
      const T* row = base + p * S::STRIDE;
      T rho[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) rho[q] = row[R + xg * 5 + q];
      const T n2 = row[R + NB];
      const T extra = xg == 0 ? row[R + NB + 1] : (xg == 1 ? T(1) : T(0));
#pragma unroll
      for (int tt = 0; tt < S::RPL; ++tt) {
        const int r = rg + 16 * tt;
        const T G = r < R ? row[r] : T(0);
        accx[tt] += G * extra;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          T gr = G * rho[q];
          acc1[tt][q] += gr;
          acc2[tt][q] += gr * (n2 * rho[q]);
        }
      }
    }
    __syncthreads();
  }

  // cross-wave reduction and partial-row write
  constexpr int PER = S::RPL * 10 + 2;
  {
    T* mine = red + (wave * 64 + lane) * PER;
#pragma unroll
    for (int tt = 0; tt < S::RPL; ++tt) {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        mine[tt * 10 + q] = acc1[tt][q];
        mine[tt * 10 + 5 + q] = acc2[tt][q];
      }
    }
    mine[S::RPL * 10] = accx[0];
    mine[S::RPL * 10 + 1] = S::RPL > 1 ? accx[S::RPL - 1] : T(0);
  }
  __syncthreads();
  if (wave == 0) {
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    T* part = a.part_rad + blk * rad_partial_size(C, false);
    T tot[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e)
      tot[e] = red[(0 * 64 + lane) * PER + e] + red[(1 * 64 + lane) * PER + e] + red[(2 * 64 + lane) * PER + e] +
               red[(3 * 64 + lane) * PER + e];
#pragma unroll
    for (int tt = 0; tt < S::RPL; ++tt) {
      const int r = rg + 16 * tt;
      if (r < R) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          part[r * NB + xg * 5 + q] = tot[tt * 10 + q];
          part[R * NB + r * NB + xg * 5 + q] = tot[tt * 10 + 5 + q];
        }
        const T ex = tt == 0 ? tot[S::RPL * 10] : tot[S::RPL * 10 + 1];
        if (xg == 0) part[2 * R * NB + r] = ex;          // S
        if (xg == 1) part[2 * R * NB + R + r] = ex;      // dB
      }
    }
  }
}

// Turn the reduced pair-sums into parameter gradients (one tiny workgroup).
//   tot: T1[R][20] | T2[R][20] | S[R] | dB[R]
template <typename T>
__global__ void rad_finalize_kernel(const T* tot, int C, const T* ra, const T* rb, const T* rc, const T* w0, const T* w1,
                                    T* g_a, T* g_b, T* g_c, T* g_w0, T* g_b0, T* g_w1, T* g_b1) {
  const int R = 4 * C, F = 2 * C;
  const T* T1 = tot;
  const T* T2 = tot + R * NB;
  const T* S = tot + 2 * R * NB;
  const T* dB = S + R;
  for (int e = threadIdx.x; e < R * NB; e += blockDim.x) {
    int r = e / NB, k = e - r * NB;
    int lin = r / F, f = r - lin * F;
    T val = rb[k] * T1[e] + ra[k] * S[r];
    (lin ? g_w1 : g_w0)[f * NB + k] = val;
  }
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    int lin = r / F, f = r - lin * F;
    (lin ? g_b1 : g_b0)[f] = dB[r];
  }
  for (int k = threadIdx.x; k < NB; k += blockDim.x) {
    T da = T(0), db = T(0), dc = T(0);
    for (int r = 0; r < R; ++r) {
      int lin = r / F, f = r - lin * F;
      T w = (lin ? w1 : w0)[f * NB + k];
      da += w * S[r];
      db += w * T1[r * NB + k];
      dc += w * T2[r * NB + k];
    }
    g_a[k] = da;
    g_b[k] = db;
    g_c[k] = T(-2) * rb[k] * rc[k] * dc;
  }
}

__global__ void rad_finalize_batch_kernel(RadFinJob job) {
  const RadFinJob::Item it = job.it[blockIdx.x];
  const int C = it.C, R = 4 * C, F = 2 * C;
  const double* T1 = it.tot;
  const double* T2 = it.tot + R * NB;
  const double* S = it.tot + 2 * R * NB;
  const double* dB = S + R;
  for (int e = threadIdx.x; e < R * NB; e += blockDim.x) {
    int r = e / NB, k = e - r * NB;
    int lin = r / F, f = r - lin * F;
    (lin ? it.g_w1 : it.g_w0)[f * NB + k] = it.rb[k] * T1[e] + it.ra[k] * S[r];
  }
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    int lin = r / F, f = r - lin * F;
    (lin ? it.g_b1 : it.g_b0)[f] = dB[r];
  }
  for (int k = threadIdx.x; k < NB; k += blockDim.x) {
    double da = 0, db = 0, dc = 0;
    for (int r = 0; r < R; ++r) {
      int lin = r / F, f = r - lin * F;
      double w = (lin ? it.w1 : it.w0)[f * NB + k];
      da += w * S[r];
      db += w * T1[r * NB + k];
      dc += w * T2[r * NB + k];
    }
    it.g_a[k] = da;
    it.g_b[k] = db;
    it.g_c[k] = -2.0 * it.rb[k] * it.rc[k] * dc;
  }
}

int rad_finalize_batch(const RadFinJob& job, hipStream_t stream) {
  if (job.n <= 0) return 0;
  hipLaunchKernelGGL(rad_finalize_batch_kernel, dim3(job.n), dim3(256), 0, stream, job);
  LGN_CHECK_LAUNCH();
  return 0;
}

// =========================================================================================
// 3b. i-centric pass, decoder: radial bias gradients + position gradient of the receiving node.
//   R0[c] = b0[c] (1+i), R1[c] = b1[c] (1+i)  ->  d b0[c] = Re G_R0 + Im G_R0 (same for b1)
//   g_p[i] += sum_j G_q_ij
// =========================================================================================
template <typename T, int C>
__global__ __launch_bounds__(BLOCK) void level_bwd_rad_dec_kernel(LevelBwdArgs<T> a) {
  using L = Carve<C, true>;
  constexpr int JS = 8, IT = BLOCK / JS;
  const int N = a.N, B = a.B;
  const int b = blockIdx.x, it = blockIdx.y, tid = threadIdx.x;

  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  T* nd = sm;                                     // N * NS
  T* pj = nd + L::even(N * L::NS);                // N * 8
  T* gat = pj + N * 8;                            // IT * 20C
  T* bias = gat + IT * GA<C>::SIZE;               // 2C
  T* red = bias + L::even(2 * C);                 // BLOCK/64 * 2C

  load_jet<T, C, true>(a.s_in, a.v_in, a.p, nullptr, B, N, b, nd, pj, nullptr);
  for (int e = tid; e < IT * GA<C>::SIZE; e += BLOCK) {
    int il = e / GA<C>::SIZE, i = it * IT + il;
    gat[e] = i < N ? a.g_ag[((size_t)b * N + it * IT) * GA<C>::SIZE + e] : T(0);
  }
  for (int e = tid; e < 2 * C; e += BLOCK) bias[e] = e < C ? a.b0[e] : a.b1[e - C];
  __syncthreads();

  const int il = tid / JS, js = tid % JS;
  const int i = it * IT + il;
  const bool row_ok = i < N;
  const int ii = row_ok ? i : 0;
  T pi[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) pi[m] = pj[ii * 8 + m];
  const T* gi = gat + il * GA<C>::SIZE;

  T dB0[C], dB1[C];
  cx<T> Gq[4];
#pragma unroll
  for (int c = 0; c < C; ++c) dB0[c] = dB1[c] = T(0);
#pragma unroll
  for (int m = 0; m < 4; ++m) Gq[m] = {T(0), T(0)};

  if (row_ok) {
    for (int j = js; j < N; j += JS) {
      PairGeom<T, true> g = pair_geom<T, true>(pi, pj + j * 8, false, false);
      const T* njp = nd + j * L::NS;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        cx<T> R1 = {bias[C + c], bias[C + c]};
        cx<T> s = {njp[c * 10], njp[c * 10 + 1]};
        cx<T> v[4], vt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = {njp[c * 10 + 2 + m], njp[c * 10 + 6 + m]};
        metric_perm(v, vt);
        cx<T> gA3 = {T(0.5) * gi[GA<C>::A3 + 2 * c], T(0.5) * gi[GA<C>::A3 + 2 * c + 1]};
        cx<T> gA4 = {gi[GA<C>::A4 + 2 * c], gi[GA<C>::A4 + 2 * c + 1]};
        cx<T> ge0 = cmulc(gA4, s);
        cx<T> gR1 = {T(0), T(0)};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          cx<T> gA1 = {gi[GA<C>::A1 + (c * 4 + m) * 2], gi[GA<C>::A1 + (c * 4 + m) * 2 + 1]};
          cx<T> gA2 = {gi[GA<C>::A2 + (c * 4 + m) * 2], gi[GA<C>::A2 + (c * 4 + m) * 2 + 1]};
          cfmac(ge0, gA1, v[m]);
          cx<T> ge1 = cmulc(gA2, s);
          cfmac(ge1, gA3, vt[m]);
          cfmac(gR1, ge1, g.q[m]);
          cfmac(Gq[m], ge1, R1);
        }
        // G_R0 = G_e0 (1 - i); d bias0 = Re + Im = 2 Im(G_e0)... kept in the generic form
        dB0[c] += (ge0.r + ge0.i) + (ge0.i - ge0.r);
        dB1[c] += gR1.r + gR1.i;
      }
    }
  }

  const size_t plp = (size_t)B * N * 4;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    T qr = group_sum<JS>(Gq[m].r), qi = group_sum<JS>(Gq[m].i);
    if (row_ok && js == 0) {
      a.g_p[((size_t)b * N + i) * 4 + m] += qr;
      a.g_p[plp + ((size_t)b * N + i) * 4 + m] += qi;
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    T x0 = group_sum<64>(dB0[c]), x1 = group_sum<64>(dB1[c]);
    if (lane == 0) {
      red[wave * 2 * C + c] = x0;
      red[wave * 2 * C + C + c] = x1;
    }
  }
  __syncthreads();
  if (tid < 2 * C) {
    T* part = a.part_rad + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * rad_partial_size(C, true);
    T s = T(0);
    for (int w = 0; w < BLOCK / 64; ++w) s += red[w * 2 * C + tid];
    part[tid] = s;
  }
}

// =========================================================================================
// deterministic reduction of per-workgroup partial rows:  out[n] (+)= sum_blk part[blk][n]
// =========================================================================================
// 64 columns x 16 row-groups per workgroup: every column is summed by 16 threads (fixed row interleave),
// then the 16 partial sums are combined in a fixed order through LDS -> bitwise reproducible.
constexpr int RED_RG = 16;
template <typename T>
__global__ __launch_bounds__(64 * RED_RG) void reduce_partials_kernel(const T* __restrict__ part, int nblk, int stride, int n, T* out,
                                                                      int accumulate) {
  __shared__ T red[RED_RG][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
  if (col < n) {
    int r = rg;
    for (; r + 3 * RED_RG < nblk; r += 4 * RED_RG) {
      s0 += part[(size_t)r * stride + col];
      s1 += part[(size_t)(r + RED_RG) * stride + col];
      s2 += part[(size_t)(r + 2 * RED_RG) * stride + col];
      s3 += part[(size_t)(r + 3 * RED_RG) * stride + col];
    }
    for (; r < nblk; r += RED_RG) s0 += part[(size_t)r * stride + col];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && col < n) {
    T s = T(0);
#pragma unroll
    for (int q = 0; q < RED_RG; ++q) s += red[q][cl];
    out[col] = accumulate ? out[col] + s : s;
  }
}

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
template <typename K>
static int ensure_smem(K kern, size_t smem, const char* what) {
  if (smem > 160 * 1024) {
    set_error("%s needs %zu B of LDS (> 160 KiB)", what, smem);
    return -1;
  }
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
      return (int)e;
    }
  }
  return 0;
}

int level_bwd_nodes2_dispatch(const LevelBwdArgs<double>& a, int decoder, hipStream_t stream);   // level_bwd2.hip
int level_bwd_rad2_dispatch(const LevelBwdArgs<double>& a, hipStream_t stream);

int level_bwd3_dispatch(const LevelBwdArgs<double>& a, int decoder, hipStream_t stream);         // level_bwd3.hip
int level_bwd_dec_sep_dispatch(const LevelBwdArgs<double>& a, hipStream_t stream);               // level_bwd_dec_sep.hip
static bool dec_pairwise() {   // LGN_AMD_DEC_PAIRWISE=1 keeps the decoder on the O(N^2) pair sweeps (read per call: tests flip it)
  const char* e = getenv("LGN_AMD_DEC_PAIRWISE");
  return e && e[0] == '1';
}
bool level_bwd3_fits(int N);

static bool use_v1() {
  static const bool v = [] { const char* e = getenv("LGN_AMD_LEVEL_V1"); return e && e[0] == '1'; }();
  return v;
}
static bool use_v3(int N) {   // single-kernel backward (default when the jet fits in LDS); LGN_AMD_LEVEL_V2=1 forces the 3-kernel form
  static const bool off = [] { const char* e = getenv("LGN_AMD_LEVEL_V2"); return e && e[0] == '1'; }();
  return !off && !use_v1() && level_bwd3_fits(N);
}

int level_bwd_rad_jt(int N) { return N <= 64 ? ((N + 7) / 8) * 8 : 32; }

// number of partial rows the backward launch writes (host side must size the workspace with these)
void level_bwd_partial_rows(int B, int N, int decoder, int* rows_mix, int* rows_rad) {
  const int tiles = cdiv(N, 32);
  if (use_v3(N)) {                       // level_bwd3: one partial row per jet for both
    *rows_mix = B;
    *rows_rad = B;
    return;
  }
  *rows_mix = B * tiles;
  if (decoder && !use_v1() && !dec_pairwise()) {
    *rows_rad = B;                       // separable decoder backward: one partial row per jet
  } else if (decoder) {
    *rows_rad = B * tiles;
  } else if (!use_v1()) {
    *rows_rad = B;                       // level_bwd_rad2: one partial row per jet
  } else {
    const int JT = level_bwd_rad_jt(N);
    *rows_rad = B * tiles * cdiv(N, JT);
  }
}

template <typename T, int C, bool DEC>
static int launch_level_bwd(const LevelBwdArgs<T>& a, hipStream_t stream) {
  using L = Carve<C, DEC>;
  const int N = a.N, CO = a.CO, tiles = cdiv(N, 32);
  int rc;
  if (use_v3(N)) return level_bwd3_dispatch(a, DEC, stream);
  {  // 1. CatMix / power
    auto kern = level_bwd_mix_kernel<T, C>;
    size_t smem = sizeof(T) * (4 * CO * 5 * C + 32 * (5 * C * 10) + 32 * CO * 10);
    if ((rc = ensure_smem(kern, smem, "level_bwd_mix"))) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B, tiles), dim3(BLOCK), smem, stream, a);
    LGN_CHECK_LAUNCH();
  }
  if constexpr (DEC && std::is_same<T, double>::value) {
    if (!use_v1() && !dec_pairwise()) return level_bwd_dec_sep_dispatch(a, stream);   // 2+3. from jet-level sums, O(N C)
  }
  if (!use_v1()) {  // 2. j-centric pass, matrix-core version
    if ((rc = level_bwd_nodes2_dispatch(a, DEC, stream))) return rc;
  } else {
    constexpr int IS = 8;
    auto kern = level_bwd_nodes_kernel<T, C, IS, DEC>;
    size_t smem = sizeof(T) * (N * GA<C>::SIZE + L::even(N * L::PS) + L::even(L::RAD_SIZE)) + N + 16;
    if ((rc = ensure_smem(kern, smem, "level_bwd_nodes"))) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B, cdiv(N, BLOCK / IS)), dim3(BLOCK), smem, stream, a);
    LGN_CHECK_LAUNCH();
  }
  if (DEC) {  // 3b
    auto kern = level_bwd_rad_dec_kernel<T, C>;
    size_t smem = sizeof(T) * (L::even(N * L::NS) + N * 8 + 32 * GA<C>::SIZE + L::even(2 * C) + (BLOCK / 64) * 2 * C);
    if ((rc = ensure_smem(kern, smem, "level_bwd_rad_dec"))) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B, tiles), dim3(BLOCK), smem, stream, a);
    LGN_CHECK_LAUNCH();
  } else if (!use_v1()) {  // 3a, matrix-core version
    if ((rc = level_bwd_rad2_dispatch(a, stream))) return rc;
  } else {  // 3a
    using S = Stage<C>;
    const int JT = level_bwd_rad_jt(N);
    auto kern = level_bwd_rad_enc_kernel<T, C>;
    size_t smem = sizeof(T) * (L::even(JT * L::NS) + JT * 4 + 32 * 4 + 32 * GA<C>::SIZE + NB + BLOCK * S::STRIDE +
                               4 * 64 * (S::RPL * 10 + 2)) + JT + 32 + 16;
    if ((rc = ensure_smem(kern, smem, "level_bwd_rad_enc"))) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B, tiles, cdiv(N, JT)), dim3(BLOCK), smem, stream, a, JT);
    LGN_CHECK_LAUNCH();
  }
  return 0;
}

template <typename T>
int level_bwd_dispatch(const LevelBwdArgs<T>& a, int decoder, hipStream_t stream) {
  LGN_CHECK_ARG(a.B > 0 && a.N > 0, "level_bwd: empty batch (B=%d N=%d)", a.B, a.N);
  LGN_CHECK_ARG(a.CO >= 1 && a.CO <= 8, "level_bwd: C_out=%d unsupported (1..8)", a.CO);
#define LGN_CASE(CC)                                                              \
  case CC:                                                                        \
    return decoder ? launch_level_bwd<T, CC, true>(a, stream) : launch_level_bwd<T, CC, false>(a, stream);
  switch (a.C) {
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
    default:
      set_error("level_bwd: C_in=%d unsupported (1..8)", a.C);
      return -1;
  }
#undef LGN_CASE
}

template <typename T>
int reduce_partials(const T* part, int nblk, int n, T* out, int accumulate, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3(cdiv(n, 64)), dim3(64 * RED_RG), 0, stream, part, nblk, n, n, out, accumulate);
  LGN_CHECK_LAUNCH();
  return 0;
}

// several independent column ranges (possibly of different partial buffers) in ONE launch
template <typename T>
__global__ __launch_bounds__(64 * RED_RG) void reduce_segments_kernel(RedJob<T> job) {
  __shared__ T red[RED_RG][64];
  int k = 0;
  while (k + 1 < job.nseg && (int)blockIdx.x >= job.tile0[k + 1]) ++k;
  const RedSeg<T> sg = job.seg[k];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = ((int)blockIdx.x - job.tile0[k]) * 64 + cl;
  T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
  if (col < sg.n) {
    const T* p = sg.part + sg.col0 + col;
    int r = rg;
    for (; r + 3 * RED_RG < sg.rows; r += 4 * RED_RG) {
      s0 += p[(size_t)r * sg.stride];
      s1 += p[(size_t)(r + RED_RG) * sg.stride];
      s2 += p[(size_t)(r + 2 * RED_RG) * sg.stride];
      s3 += p[(size_t)(r + 3 * RED_RG) * sg.stride];
    }
    for (; r < sg.rows; r += RED_RG) s0 += p[(size_t)r * sg.stride];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && col < sg.n) {
    T s = T(0);
#pragma unroll
    for (int q = 0; q < RED_RG; ++q) s += red[q][cl];
    sg.out[col] = s;
  }
}

template <typename T>
int reduce_segments(RedJob<T>& job, hipStream_t stream) {
  int tiles = 0, live = 0;
  RedJob<T> packed{};
  for (int k = 0; k < job.nseg; ++k) {
    if (job.seg[k].n <= 0) continue;
    packed.seg[live] = job.seg[k];
    packed.tile0[live] = tiles;
    tiles += cdiv(job.seg[k].n, 64);
    ++live;
  }
  if (!live) return 0;
  packed.nseg = live;
  packed.tile0[live] = tiles;
  hipLaunchKernelGGL(reduce_segments_kernel<T>, dim3(tiles), dim3(64 * RED_RG), 0, stream, packed);
  LGN_CHECK_LAUNCH();
  return 0;
}
template int reduce_segments<double>(RedJob<double>&, hipStream_t);

// columns [col0, col0+n) of rows of length `stride`
template <typename T>
int reduce_partials_strided(const T* part, int nblk, int stride, int col0, int n, T* out, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3(cdiv(n, 64)), dim3(64 * RED_RG), 0, stream, part + col0, nblk, stride, n, out, 0);
  LGN_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int rad_finalize(const T* tot, int C, const T* ra, const T* rb, const T* rc, const T* w0, const T* w1, T* g_a, T* g_b,
                 T* g_c, T* g_w0, T* g_b0, T* g_w1, T* g_b1, hipStream_t stream) {
  hipLaunchKernelGGL(rad_finalize_kernel<T>, dim3(1), dim3(256), 0, stream, tot, C, ra, rb, rc, w0, w1, g_a, g_b, g_c, g_w0,
                     g_b0, g_w1, g_b1);
  LGN_CHECK_LAUNCH();
  return 0;
}

template int level_bwd_dispatch<double>(const LevelBwdArgs<double>&, int, hipStream_t);
template int reduce_partials<double>(const double*, int, int, double*, int, hipStream_t);
template int reduce_partials_strided<double>(const double*, int, int, int, int, double*, hipStream_t);
template int rad_finalize<double>(const double*, int, const double*, const double*, const double*, const double*,
                                  const double*, double*, double*, double*, double*, double*, double*, double*, hipStream_t);

}  // namespace lgn
