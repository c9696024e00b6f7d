// lgn-autoencoder_amd/csrc/level_bwd.hip -- backward of the fused message-passing level (maxdim = 2).
//
// The reference obtains these gradients by autograd through RadPolyTrig / GScalar*GVec / cg_product /
// CatMixReps (lgn/nn/position_levels.py:118-209, lgn/models/lgn_cg.py:167, lgn/cg_lib/cg_ops.py:135-298,
// lgn/nn/g_nn.py:260-278) which stores the N x N edge tensors.  Here the edges are *recomputed*:
//
//   1. level_bwd_mix     per node: CatMix^T -> grad of aggregate (g_ag), direct + power-term grads of the
//                        node features, and CatMix weight gradient partials.
//   2. level_bwd_nodes2  "j-centric" pass: g_node_j += sum_i g_ag_i (x) conj(edge_ij)   (+ decoder dp_j)   [level_bwd2.hip]
//   3. level_bwd_rad2 / level_bwd_rad_dec   "i-centric" pass: radial-network parameter gradients (+ decoder dp_i)
// This three-launch form serves N > 40 (and LGN_AMD_LEVEL_V2=1); smaller jets take the one-kernel level_bwd3.hip, the
// separable decoder level_bwd_dec_sep.hip.
//
// Complex convention: every map is holomorphic (no conjugation in the forward), so for out = f(z) the
// planar gradient is G_z = G_out * conj(f'(z)).
#include <stdlib.h>

#include "level_dev.hpp"
#include <type_traits>
#include "ops.hpp"
#include "tail_dev.hpp"

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
// 2. / 3a.  The j-centric pass (gradient w.r.t. the source nodes) and the encoder's i-centric radial-gradient pass
//    live in level_bwd2.hip (matrix-core kernels).  The radial pair sums they produce,
//      T1[r][k] = sum_p G on rho_k    T2[r][k] = sum_p G on n^2 rho_k^2    S[r] = sum_p G on    dB[r] = sum_p G,
//    are turned into parameter gradients by rad_finalize below.
// =========================================================================================

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

__global__ __launch_bounds__(256) void rad_finalize_batch_kernel(RadFinJob job) {
  // one workgroup per level.  The reduced sums and the two Linear weights are staged in LDS by all 256 threads first: the 20
  // threads that contract over the 4C rows would otherwise each walk 3 x 4C dependent global loads (8 us for ~2 k flops)
  __shared__ double sh[2 * 32 * NB + 2 * 32 + 32 * NB];      // T1 | T2 | S | dB | w (R <= 32 rows)
  const RadFinJob::Item it = job.it[blockIdx.x];
  const int C = it.C, R = 4 * C, F = 2 * C;
  double* T1 = sh;
  double* T2 = T1 + R * NB;
  double* S = T2 + R * NB;
  double* dB = S + R;
  double* w = dB + R;                                        // [R][NB]: rows 0..F-1 = w0, F..R-1 = w1
  for (int e = threadIdx.x; e < 2 * R * NB + 2 * R; e += blockDim.x) sh[e] = it.tot[e];
  for (int e = threadIdx.x; e < R * NB; e += blockDim.x) w[e] = e < F * NB ? it.w0[e] : it.w1[e - F * NB];
  __syncthreads();
  for (int e = threadIdx.x; e < R * NB; e += blockDim.x) {
    int r = e / NB, k = e - r * NB;
    int lin = r / F, f = r - lin * F;
    (lin ? it.g_w1 : it.g_w0)[f * NB + k] = radfin_weight(it.rb[k], T1[e], it.ra[k], S[r]);
  }
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    int lin = r / F, f = r - lin * F;
    (lin ? it.g_b1 : it.g_b0)[f] = dB[r];
  }
  for (int k = threadIdx.x; k < NB; k += blockDim.x) {
    it.g_a[k] = radfin_dot(w + k, NB, S, 1, R);
    it.g_b[k] = radfin_dot(w + k, NB, T1 + k, NB, R);
    it.g_c[k] = radfin_c(it.rb[k], it.rc[k], radfin_dot(w + k, NB, T2 + k, NB, R));
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
int level_bwd_sweep_enc_dispatch(const LevelBwdArgs<double>& a, hipStream_t stream);             // level_bwd2.hip
static bool dec_pairwise(int flags) { return (flags & LVL_DEC_PAIRWISE) != 0; }   // the decoder on the O(N^2) pair sweeps
bool level_bwd3_fits(int N);

// single-kernel backward (default when the jet fits in LDS); LVL_LEVEL_V2 forces the 3-kernel form also for small N
static bool use_v3(int N, int flags) { return !(flags & LVL_LEVEL_V2) && level_bwd3_fits(N); }

// does the level backward for N-particle jets run as the one kernel that can carry the input stage's backward (LevelBwdArgs::part_in0)?
bool level_bwd_carries_input(int N, int flags) { return use_v3(N, flags); }

// number of partial rows the backward launch writes (host side must size the workspace with these)
void level_bwd_partial_rows(int B, int N, int decoder, int flags, int* rows_mix, int* rows_rad) {
  const int tiles = cdiv(N, 32);
  if (use_v3(N, flags)) {                // level_bwd3: one partial row per workgroup for both (small batches: several per jet)
    const int split = (decoder && !dec_pairwise(flags)) ? 1 : level_jet_split(B, N);
    *rows_mix = B * split;
    *rows_rad = B * split;
    return;
  }
  *rows_mix = B * tiles;
  if (decoder && dec_pairwise(flags)) {
    *rows_rad = B * tiles;               // level_bwd_rad_dec: one partial row per 32-node tile
  } else {
    *rows_rad = B;                       // separable decoder backward / level_bwd_rad2: one partial row per jet
  }
}

template <typename T, int C, bool DEC>
static int launch_level_bwd(const LevelBwdArgs<T>& a, hipStream_t stream) {
  using L = Carve<C, DEC>;
  const int N = a.N, CO = a.CO, tiles = cdiv(N, 32);
  int rc;
  if (use_v3(N, a.flags)) return level_bwd3_dispatch(a, DEC, stream);
  {  // 1. CatMix / power
    auto kern = level_bwd_mix_kernel<T, C>;
    size_t smem = sizeof(T) * (4 * CO * 5 * C + 32 * (5 * C * 10) + 32 * CO * 10);
    if ((rc = ensure_smem(kern, smem, "level_bwd_mix"))) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B, tiles), dim3(BLOCK), smem, stream, a);
    LGN_CHECK_LAUNCH();
  }
  if constexpr (DEC && std::is_same<T, double>::value) {
    if (!dec_pairwise(a.flags)) return level_bwd_dec_sep_dispatch(a, stream);   // 2+3. from jet-level sums, O(N C)
  }
  if constexpr (!DEC && std::is_same<T, double>::value) {
    // 2+3. encoder: node gradients and radial sums in ONE pair sweep (level_bwd2.hip); LVL_LEVEL_V2 keeps the two sweeps below
    if (!(a.flags & LVL_LEVEL_V2)) return level_bwd_sweep_enc_dispatch(a, stream);
  }
  if ((rc = level_bwd_nodes2_dispatch(a, DEC, stream))) return rc;        // 2. j-centric pass
  if (DEC) {  // 3b
    auto kern = level_bwd_rad_dec_kernel<T, C>;
    size_t smem = sizeof(T) * (L::even(N * L::NS) + N * 8 + 32 * GA<C>::SIZE + L::even(2 * C) + (BLOCK / 64) * 2 * C);
    if ((rc = ensure_smem(kern, smem, "level_bwd_rad_dec"))) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B, tiles), dim3(BLOCK), smem, stream, a);
    LGN_CHECK_LAUNCH();
  } else {    // 3a
    if ((rc = level_bwd_rad2_dispatch(a, stream))) return rc;
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
