// lgn-autoencoder_amd/csrc/generic_local2.hip -- per-node part of a message-passing level for arbitrary irreps, v2:
// Clebsch-Gordan contraction of the neighbour moments (aggregate) and of node (x) node (power), concatenation with the
// node features and the complex CatMix -- forward and backward.  Same operator and tables as generic_local.hip
// (reference: cg_product lgn/cg_lib/cg_ops.py:177-218, CatReps / CatMixReps lgn/nn/g_nn.py:160-190,260-278).
//
// Mapping: lane = (node, channel).  A wave owns 64 / CP nodes (CP = 4 or 8 lanes per node, one per input channel) and
// walks the level's rows  (output irrep l, block, m)  in table order.  The walk is wave-uniform -- row bounds, term codes and
// CG coefficients are scalar loads, the loops have no divergence -- and every lane applies the same term to its own
// (node, channel) data:  cat[row][c] = sum_terms coef * { U[c][a] | X[c][a] | X[c][a] X[c][b] }.
// The CatMix sum over channels is a butterfly over the CP lanes of a node at the end of each output irrep; no LDS and no
// barrier in the forward, one lane-private LDS accumulator (d X) in the backward.  Nothing is staged: the 1.6 KB of moments of
// a (node, channel) are read straight from global memory, 16 B per term, and stay in L1/L2 while the rows are walked.
//   forward : out[o][q0_l + m] = sum_{blk, c} W_l[o][blk C + c] cat[(l, blk, m)][c]
//   backward: g_cat = W^H g_out  (recomputed where needed),  dW += g_out (x) conj(cat),  dU, dX through the term lists
#include "ops.hpp"

namespace lgn {
namespace {

constexpr int COMAX = 8;       // output channels (CatMix rows) per level
constexpr int TB = 4;          // terms fetched per batch (their loads are in flight together)

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

struct LaneCtx {
  const double* Un;            // U of this (node, channel): [Q*5][2]
  const double* Xr;            // X of this (node, channel), real plane [Q]
  const double* Xi;
};

// value of one term for this lane (all three operand kinds are loaded unconditionally: the branch-free form keeps the batch's
// loads independent of each other)
__device__ __forceinline__ cx<double> term_value(const LaneCtx& L, int ty, int ia, int ib) {
  const double* up = L.Un + 2 * (ty == 0 ? ia : 0);
  const int xa = ty == 0 ? 0 : ia, xb = ty == 2 ? ib : 0;
  const cx<double> u = {up[0], up[1]};
  const cx<double> x = {L.Xr[xa], L.Xi[xa]};
  const cx<double> y = {L.Xr[xb], L.Xi[xb]};
  cx<double> v = ty == 0 ? u : x;
  if (ty == 2) v = cmul(x, y);
  return v;
}

// cat value of one row: sum of its terms (in list order)
__device__ __forceinline__ cx<double> row_value(const LocalTables& t, const LaneCtx& L, int row) {
  const int beg = uni(t.row_ptr[row]), end = uni(t.row_ptr[row + 1]);
  cx<double> acc = {0, 0};
  for (int k = beg; k < end; k += TB) {
    int ty[TB], ia[TB], ib[TB];
    double cf[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int kk = min(k + j, end - 1);
      ty[j] = uni(t.t_type[kk]);
      ia[j] = uni(t.t_a[kk]);
      ib[j] = uni(t.t_b[kk]);
      cf[j] = k + j < end ? t.t_coef[kk] : 0.0;
    }
    cx<double> v[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) v[j] = term_value(L, ty[j], ia[j], ib[j]);
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      acc.r += cf[j] * v[j].r;
      acc.i += cf[j] * v[j].i;
    }
  }
  return acc;
}

template <int CP>
__device__ __forceinline__ double sum_over_channels(double v) {      // butterfly over the CP lanes of a node
  v += shfl_xor(v, 1);
  v += shfl_xor(v, 2);
  if (CP == 8) v += shfl_xor(v, 4);
  return v;
}
template <int CP>
__device__ __forceinline__ double sum_over_8_nodes(double v) {       // butterfly over a group of 8 nodes (CP * 8 consecutive lanes)
  v += shfl_xor(v, CP);
  v += shfl_xor(v, 2 * CP);
  v += shfl_xor(v, 4 * CP);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward, one output irrep of dimension D
// ---------------------------------------------------------------------------------------------------------------------
template <int CP, int D>
__device__ __forceinline__ void irrep_fwd(const LocalArgs& a, const LaneCtx& L, int l, int node, int c, bool live) {
  const int C = a.C, CO = a.CO, Qo = a.Qout;
  const int nb = uni(a.t.out_nblk[l]), row0 = uni(a.t.out_row0[l]), q0 = uni(a.t.out_q0[l]);
  const int K = nb * C;
  const double* wr = a.wcat + uni(a.t.out_w0[l]);
  const double* wi = wr + (size_t)CO * K;
  const int cc = c < C ? c : C - 1;
  cx<double> acc[COMAX][D];
#pragma unroll
  for (int o = 0; o < COMAX; ++o)
#pragma unroll
    for (int m = 0; m < D; ++m) acc[o][m] = {0, 0};
  for (int blk = 0; blk < nb; ++blk) {
    cx<double> w[COMAX];
#pragma unroll
    for (int o = 0; o < COMAX; ++o) {
      const int oo = o < CO ? o : CO - 1;
      w[o] = {wr[(size_t)oo * K + blk * C + cc], wi[(size_t)oo * K + blk * C + cc]};
      if (!live || o >= CO) w[o] = {0, 0};             // idle lanes (c >= C, node past the end) contribute nothing
    }
#pragma unroll
    for (int m = 0; m < D; ++m) {
      const cx<double> cat = row_value(a.t, L, row0 + blk * D + m);
#pragma unroll
      for (int o = 0; o < COMAX; ++o)
        if (o < CO) cfma(acc[o][m], w[o], cat);
    }
  }
  const size_t plo = (size_t)a.nodes * CO * Qo;
#pragma unroll
  for (int o = 0; o < COMAX; ++o) {
    if (o < CO) {
#pragma unroll
      for (int m = 0; m < D; ++m) {
        const double sr = sum_over_channels<CP>(acc[o][m].r), si = sum_over_channels<CP>(acc[o][m].i);
        if ((o % CP) == c && node < a.nodes) {
          const size_t oe = ((size_t)node * CO + o) * Qo + q0 + m;
          a.out[oe] = sr;
          a.out[plo + oe] = si;
          if (a.s_copy && q0 + m == a.q_s) {           // pre-MLP scalars kept for the CGMLP backward
            a.s_copy[(size_t)node * CO + o] = sr;
            a.s_copy[(size_t)a.nodes * CO + (size_t)node * CO + o] = si;
          }
        }
      }
    }
  }
}

template <int CP>
__global__ __launch_bounds__(BLOCK) void local_fwd2_kernel(LocalArgs a) {
  constexpr int NPW = 64 / CP;
  const int lane = threadIdx.x & 63, wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const int c = lane % CP, node = wave * NPW + lane / CP;
  if (wave * NPW >= a.nodes) return;                   // whole wave past the end (uniform)
  const bool live = node < a.nodes && c < a.C;
  const int nn = node < a.nodes ? node : a.nodes - 1, cc = c < a.C ? c : a.C - 1;
  const size_t e = ((size_t)nn * a.C + cc) * a.Q;
  LaneCtx L{a.U + e * 10, a.X + e, a.X + (size_t)a.nodes * a.C * a.Q + e};
  const int n_out = a.t.n_out;
  for (int l = 0; l < n_out; ++l) {
    switch (uni(a.t.out_dim[l])) {
      case 1: irrep_fwd<CP, 1>(a, L, l, node, c, live); break;
      case 3: irrep_fwd<CP, 3>(a, L, l, node, c, live); break;
      case 4: irrep_fwd<CP, 4>(a, L, l, node, c, live); break;
      case 9: irrep_fwd<CP, 9>(a, L, l, node, c, live); break;
      default: break;                                   // rejected on the host
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
// lane-private accumulator of the node-feature gradient in LDS: gx[q][lane] (re, im)
__device__ __forceinline__ void gx_add(double* gx, int q, int lane, cx<double> v) {
  double* p = gx + ((size_t)q * 64 + lane) * 2;
  p[0] += v.r;
  p[1] += v.i;
}

template <int CP, int D>
__device__ __forceinline__ void irrep_bwd(const LocalArgs& a, const LaneCtx& L, int l, int node, int c, bool live, int wave,
                                          int lane, double* gx) {
  const int C = a.C, CO = a.CO, Qo = a.Qout;
  const int nb = uni(a.t.out_nblk[l]), row0 = uni(a.t.out_row0[l]), q0 = uni(a.t.out_q0[l]), w0 = uni(a.t.out_w0[l]);
  const int K = nb * C;
  const double* wr = a.wcat + w0;
  const double* wi = wr + (size_t)CO * K;
  const int cc = c < C ? c : C - 1, nn = node < a.nodes ? node : a.nodes - 1;
  const size_t plo = (size_t)a.nodes * CO * Qo;
  // upstream gradient of this node's output irrep: go[o][m]
  cx<double> go[COMAX][D];
#pragma unroll
  for (int o = 0; o < COMAX; ++o) {
    const int oo = o < CO ? o : CO - 1;
#pragma unroll
    for (int m = 0; m < D; ++m) {
      const size_t oe = ((size_t)nn * CO + oo) * Qo + q0 + m;
      go[o][m] = {a.g_out[oe], a.g_out[plo + oe]};
      if (!live || o >= CO) go[o][m] = {0, 0};
    }
  }
  // one partial row per group of 8 nodes (a wave holds one group at CP = 8, two at CP = 4), irrep l at out_w0[l]
  const int prow = wave * (8 / CP) + lane / (8 * CP);
  double* part = a.part + (size_t)prow * 2 * a.t.n_w + w0;
  const bool pwrite = (lane % (8 * CP)) < CP && c < C && prow * 8 < a.nodes;
  for (int blk = 0; blk < nb; ++blk) {
    cx<double> w[COMAX], dw[COMAX];
#pragma unroll
    for (int o = 0; o < COMAX; ++o) {
      const int oo = o < CO ? o : CO - 1;
      w[o] = {wr[(size_t)oo * K + blk * C + cc], wi[(size_t)oo * K + blk * C + cc]};
      dw[o] = {0, 0};
    }
#pragma unroll
    for (int m = 0; m < D; ++m) {
      const int row = row0 + blk * D + m;
      cx<double> gc = {0, 0};                          // gradient of this cat row: sum_o g_out[o][q] conj(W[o][k])
#pragma unroll
      for (int o = 0; o < COMAX; ++o)
        if (o < CO) cfmac(gc, go[o][m], w[o]);
      // walk the row: value (for dW) and the scatter of gc into d X (node block and power blocks)
      const int beg = uni(a.t.row_ptr[row]), end = uni(a.t.row_ptr[row + 1]);
      cx<double> cat = {0, 0};
      for (int k = beg; k < end; k += TB) {
        int ty[TB], ia[TB], ib[TB];
        double cf[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          const int kk = min(k + j, end - 1);
          ty[j] = uni(a.t.t_type[kk]);
          ia[j] = uni(a.t.t_a[kk]);
          ib[j] = uni(a.t.t_b[kk]);
          cf[j] = k + j < end ? a.t.t_coef[kk] : 0.0;
        }
        cx<double> u[TB], x[TB], y[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          const double* up = L.Un + 2 * (ty[j] == 0 ? ia[j] : 0);
          const int xa = ty[j] == 0 ? 0 : ia[j], xb = ty[j] == 2 ? ib[j] : 0;
          u[j] = {up[0], up[1]};
          x[j] = {L.Xr[xa], L.Xi[xa]};
          y[j] = {L.Xr[xb], L.Xi[xb]};
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          if (k + j < end) {                           // uniform
            const cx<double> g = {cf[j] * gc.r, cf[j] * gc.i};
            cx<double> v = u[j];
            if (ty[j] == 1) {
              v = x[j];
              gx_add(gx, ia[j], lane, g);
            } else if (ty[j] == 2) {
              v = cmul(x[j], y[j]);
              gx_add(gx, ia[j], lane, cmulc(g, y[j]));
              gx_add(gx, ib[j], lane, cmulc(g, x[j]));
            }
            cat.r += cf[j] * v.r;
            cat.i += cf[j] * v.i;
          }
        }
      }
#pragma unroll
      for (int o = 0; o < COMAX; ++o)
        if (o < CO) cfmac(dw[o], go[o][m], cat);
    }
    // CatMix weight gradient of (l, blk, c): sum over the wave's nodes, one partial row per wave
#pragma unroll
    for (int o = 0; o < COMAX; ++o) {
      if (o < CO) {
        const double sr = sum_over_8_nodes<CP>(dw[o].r), si = sum_over_8_nodes<CP>(dw[o].i);
        if (pwrite) {
          part[(size_t)o * K + blk * C + c] = sr;
          part[(size_t)CO * K + (size_t)o * K + blk * C + c] = si;
        }
      }
    }
  }
}

template <int CP>
__global__ __launch_bounds__(BLOCK) void local_bwd2_kernel(LocalArgs a) {
  constexpr int NPW = 64 / CP;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6, wave = blockIdx.x * (BLOCK / 64) + wl;
  const int c = lane % CP, node = wave * NPW + lane / CP;
  if (wave * NPW >= a.nodes) return;                   // uniform; no barriers below
  const int C = a.C, CO = a.CO, Q = a.Q, Qo = a.Qout;
  double* gx = reinterpret_cast<double*>(smem_raw) + (size_t)wl * Q * 64 * 2;     // [Q][64 lanes][2], private to this wave
  for (int q = 0; q < Q; ++q) {
    gx[((size_t)q * 64 + lane) * 2] = 0.0;
    gx[((size_t)q * 64 + lane) * 2 + 1] = 0.0;
  }
  const bool live = node < a.nodes && c < C;
  const int nn = node < a.nodes ? node : a.nodes - 1, cc = c < C ? c : C - 1;
  const size_t e = ((size_t)nn * C + cc) * Q;
  LaneCtx L{a.U + e * 10, a.X + e, a.X + (size_t)a.nodes * C * Q + e};
  const int n_out = a.t.n_out;
  for (int l = 0; l < n_out; ++l) {
    switch (uni(a.t.out_dim[l])) {
      case 1: irrep_bwd<CP, 1>(a, L, l, node, c, live, wave, lane, gx); break;
      case 3: irrep_bwd<CP, 3>(a, L, l, node, c, live, wave, lane, gx); break;
      case 4: irrep_bwd<CP, 4>(a, L, l, node, c, live, wave, lane, gx); break;
      case 9: irrep_bwd<CP, 9>(a, L, l, node, c, live, wave, lane, gx); break;
      default: break;
    }
  }
  // gradient of the moments, gather form: dU[a] = sum over the rows it feeds of coef * g_cat(row); a moment feeds ~1 row
  // (114 terms for 100 moments at maxdim 3), so g_cat is recomputed from g_out and the weights instead of being stored
  const size_t plo = (size_t)a.nodes * CO * Qo;
  double* gu = a.gU + e * 10;
  for (int ua = 0; ua < 5 * Q; ++ua) {
    const int tb = uni(a.t.u_ptr[ua]), te = uni(a.t.u_ptr[ua + 1]);
    cx<double> acc = {0, 0};
    for (int t = tb; t < te; ++t) {
      const int row = uni(a.t.u_row[t]);
      const double cf = a.t.u_coef[t];
      int l = 0;
      while (l + 1 < n_out && row >= uni(a.t.out_row0[l + 1])) ++l;
      const int d = uni(a.t.out_dim[l]), rel = row - uni(a.t.out_row0[l]), blk = rel / d, m = rel - blk * d;
      const int K = uni(a.t.out_nblk[l]) * C, q = uni(a.t.out_q0[l]) + m;
      const double* wr = a.wcat + uni(a.t.out_w0[l]) + blk * C + cc;
      const double* wi = wr + (size_t)CO * K;
      cx<double> gc = {0, 0};
      for (int o = 0; o < CO; ++o) {
        const size_t oe = ((size_t)nn * CO + o) * Qo + q;
        cfmac(gc, cx<double>{a.g_out[oe], a.g_out[plo + oe]}, cx<double>{wr[(size_t)o * K], wi[(size_t)o * K]});
      }
      acc.r += cf * gc.r;
      acc.i += cf * gc.i;
    }
    if (live) {
      gu[2 * ua] = acc.r;
      gu[2 * ua + 1] = acc.i;
    }
  }
  // gradient of the node features (node block + power blocks); the N^2 backward adds the aggregate part afterwards
  if (live) {
    const size_t plx = (size_t)a.nodes * C * Q;
    for (int q = 0; q < Q; ++q) {
      a.gX[e + q] = gx[((size_t)q * 64 + lane) * 2];
      a.gX[plx + e + q] = gx[((size_t)q * 64 + lane) * 2 + 1];
    }
  }
}

}  // namespace

// partial rows of the CatMix weight gradient: one per group of 8 nodes (= local_partial_rows of generic_local.hip)

int local2_fwd(const LocalArgs& a, hipStream_t st) {
  LGN_CHECK_ARG(a.nodes > 0 && a.C >= 1 && a.C <= 8 && a.CO >= 1 && a.CO <= COMAX, "local_fwd: unsupported channels (C=%d CO=%d)", a.C, a.CO);
  if (a.C <= 4) {
    const int waves = cdiv(a.nodes, 16);
    hipLaunchKernelGGL(local_fwd2_kernel<4>, dim3(cdiv(waves, BLOCK / 64)), dim3(BLOCK), 0, st, a);
  } else {
    const int waves = cdiv(a.nodes, 8);
    hipLaunchKernelGGL(local_fwd2_kernel<8>, dim3(cdiv(waves, BLOCK / 64)), dim3(BLOCK), 0, st, a);
  }
  LGN_CHECK_LAUNCH();
  return 0;
}

int local2_bwd(const LocalArgs& a, hipStream_t st) {
  LGN_CHECK_ARG(a.nodes > 0 && a.C >= 1 && a.C <= 8 && a.CO >= 1 && a.CO <= COMAX, "local_bwd: unsupported channels (C=%d CO=%d)", a.C, a.CO);
  const size_t smem = sizeof(double) * (size_t)(BLOCK / 64) * a.Q * 64 * 2;
  LGN_CHECK_ARG(smem <= 160 * 1024, "local_bwd: Q=%d needs %zu B of LDS", a.Q, smem);
  if (a.C <= 4) {
    const int waves = cdiv(a.nodes, 16);
    auto k = local_bwd2_kernel<4>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(cdiv(waves, BLOCK / 64)), dim3(BLOCK), smem, st, a);
  } else {
    const int waves = cdiv(a.nodes, 8);
    auto k = local_bwd2_kernel<8>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(cdiv(waves, BLOCK / 64)), dim3(BLOCK), smem, st, a);
  }
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace lgn
