// lgn-autoencoder_amd/csrc/net_dev.hpp -- device code shared by net_kernels.hip (the O(N)-per-jet network ends) and the level
// kernels that carry one of those ends in their own launch: operand staging, block sums, and the decoder output + Chamfer
// loss of a jet (forward and backward), which is either a kernel of its own or the tail of the decoder's last level forward.
#pragma once
#include "common.hpp"

namespace lgn {

constexpr double NET_RSQRT2 = 0.70710678118654752440084436210484903928;

// gradient of cart_from_canon: G_c1 = h G_px - i h G_py, G_c3 = -h G_px - i h G_py
__device__ __forceinline__ void cart_from_canon_bwd(const cx<double> (&g)[4], cx<double> (&gc)[4]) {
  gc[0] = g[0];
  gc[1] = {(g[1].r + g[2].i) * NET_RSQRT2, (g[1].i - g[2].r) * NET_RSQRT2};
  gc[2] = g[3];
  gc[3] = {(-g[1].r + g[2].i) * NET_RSQRT2, (-g[1].i - g[2].r) * NET_RSQRT2};
}

// block-wide sum of one value per thread (BLOCK threads); result valid on thread 0
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = group_sum<64>(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return s;
}
// One operand on its way to LDS: issue() puts the first U * BLOCK elements into registers (loads only), commit() stores them
// through put(element, value) and fetches whatever lies beyond that window (large jets) with a plain loop.  A kernel issues ALL
// its operands before it commits the first one, so their HBM round trips overlap.
template <int U>
struct StageRegs {
  double r[U];
  __device__ __forceinline__ void issue(const double* __restrict__ src, int total) {
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = total > 0 ? src[min((int)threadIdx.x + u * BLOCK, total - 1)] : 0.0;   // clamped: no branch per load
  }
  template <class Put>
  __device__ __forceinline__ void commit(const double* __restrict__ src, int total, Put put) const {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((int)threadIdx.x + u * BLOCK < total) put((int)threadIdx.x + u * BLOCK, r[u]);
    for (int e = threadIdx.x + U * BLOCK; e < total; e += BLOCK) put(e, src[e]);
  }
};
__device__ __forceinline__ cx<double> cart_from_canon_m(const double* c, int m) {
  if (m == 0) return {c[0], c[4]};
  if (m == 3) return {c[2], c[6]};
  if (m == 1) return {(c[1] - c[3]) * NET_RSQRT2, (c[5] - c[7]) * NET_RSQRT2};
  return {-(c[5] + c[7]) * NET_RSQRT2, (c[1] + c[3]) * NET_RSQRT2};
}

// ============================================================================================
// decoder output + get_real('sum') + Chamfer loss, forward and backward in one pass per jet
//   recon [2][B][N][4]; loss_part [B]; g_v [2][B][N][C][4]; part row per jet: dWo1 [2][C]
// LDS: x [N][4] | tg [N][4] | rmin [N] | cmin [N] | gx [N][4] | ycl [N][8] | vl [N*C][8] | tmp [N*C][2] | wol [2C] | rarg, carg [N] ints
// ============================================================================================
__host__ __device__ inline size_t dec_out_loss_bytes(int N, int C) {
  return sizeof(double) * ((size_t)N * 22 + (size_t)N * C * 10 + 2 * (size_t)C) + sizeof(int) * 2 * (size_t)N;
}
// One workgroup of BLOCK threads per jet; lds = dec_out_loss_bytes(N, C) bytes of (dynamic) LDS.  A kernel of its own
// (dec_output_loss_kernel), or the tail of the decoder's last level forward (level_fwd2.hip: LevelArgs::loss_wo1).
__device__ __forceinline__ void dec_output_loss_body(int B, int N, int C, const double* __restrict__ v,
                                                     const double* __restrict__ wo1, const double* __restrict__ target,
                                                     double loss_scale, double* recon, double* loss_part, double* g_v,
                                                     double* part, unsigned char* smem_raw) {
  double* x = reinterpret_cast<double*>(smem_raw);       // [N][4] real reconstruction (re + im)
  double* tg = x + N * 4;                                // [N][4] target
  double* rmin = tg + N * 4;                             // [N]
  double* cmin = rmin + N;                               // [N]
  double* gx = cmin + N;                                 // [N][4]
  double* ycl = gx + N * 4;                              // [N][8] canonical output (re[4] | im[4])
  double* vl = ycl + N * 8;                              // [N*C][8] the last level's vectors (re[4] | im[4])
  double* tmp = vl + N * C * 8;                          // [N*C][2]
  double* wol = tmp + N * C * 2;                         // [2C]
  int* rarg = reinterpret_cast<int*>(wol + 2 * C);       // [N]
  int* carg = rarg + N;                                  // [N]
  __shared__ double red[4];
  const int b = blockIdx.x;
  const size_t plp = (size_t)B * N * 4, pl = (size_t)B * N * C, j4 = (size_t)b * N * 4, jc = (size_t)b * N * C;
  {
    StageRegs<4> vr, vi;
    StageRegs<1> tr, wr;
    vr.issue(v + jc * 4, N * C * 4); vi.issue(v + (pl + jc) * 4, N * C * 4); tr.issue(target + j4, N * 4); wr.issue(wo1, 2 * C);
    vr.commit(v + jc * 4, N * C * 4, [&](int e, double q) { vl[(e >> 2) * 8 + (e & 3)] = q; });
    vi.commit(v + (pl + jc) * 4, N * C * 4, [&](int e, double q) { vl[(e >> 2) * 8 + 4 + (e & 3)] = q; });
    tr.commit(target + j4, N * 4, [&](int e, double q) { tg[e] = q; });
    wr.commit(wo1, 2 * C, [&](int e, double q) { wol[e] = q; });
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * 4; e += BLOCK) {     // (node, component): mix_to_output on the (1,1) irrep
    const int n = e >> 2, m = e & 3;
    cx<double> yc = {0, 0};
#pragma unroll 4
    for (int c = 0; c < C; ++c)
      cfma(yc, cx<double>{wol[c], wol[C + c]}, cx<double>{vl[(n * C + c) * 8 + m], vl[(n * C + c) * 8 + 4 + m]});
    ycl[n * 8 + m] = yc.r;
    ycl[n * 8 + 4 + m] = yc.i;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * 4; e += BLOCK) {
    const cx<double> pc = cart_from_canon_m(ycl + (e >> 2) * 8, e & 3);
    recon[j4 + e] = pc.r;
    recon[plp + j4 + e] = pc.i;
    x[e] = pc.r + pc.i;                                  // get_real(., 'sum')
  }
  __syncthreads();
  // squared Euclidean distances d(i,j) = |t_j - x_i|^2; row minima (over j) and column minima (over i), first occurrence:
  // four lanes per row / column scan a quarter of the range each, then meet on (distance, index)
  const int nq = (N + 3) / 4;
  for (int e = threadIdx.x; e < 8 * N; e += BLOCK) {
    const bool rowwise = (e >> 2) < N;
    const int a = rowwise ? e >> 2 : (e >> 2) - N, q = e & 3;
    double best = 0;
    int arg = 0x7fffffff;
#pragma unroll 4
    for (int o = q * nq; o < min(N, (q + 1) * nq); ++o) {
      const double* xi = x + (rowwise ? a : o) * 4;
      const double* tj = tg + (rowwise ? o : a) * 4;
      double d0 = tj[0] - xi[0], d1 = tj[1] - xi[1], d2 = tj[2] - xi[2], d3 = tj[3] - xi[3];
      double d = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
      if (arg == 0x7fffffff || d < best) { best = d; arg = o; }
    }
#pragma unroll
    for (int off = 2; off; off >>= 1) {
      const double ob = __shfl_xor(best, off, 4);
      const int oa = __shfl_xor(arg, off, 4);
      if (oa != 0x7fffffff && (arg == 0x7fffffff || ob < best || (ob == best && oa < arg))) { best = ob; arg = oa; }
    }
    if (q == 0) {
      if (rowwise) { rmin[a] = best; rarg[a] = arg; } else { cmin[a] = best; carg[a] = arg; }
    }
  }
  __syncthreads();
  double lsum = 0;
  for (int n = threadIdx.x; n < N; n += BLOCK) lsum += (rmin[n] + cmin[n]) * 0.5;
  lsum = block_sum(lsum, red);
  if (threadIdx.x == 0) loss_part[b] = lsum;
  // d loss / d x_i = (x_i - t_{j*(i)}) + sum_{j : i*(j) = i} (x_i - t_j)      (two lanes per (i, component): a half of the j range each)
  const int nh = (N + 1) / 2;
  for (int e = threadIdx.x; e < N * 8; e += BLOCK) {
    const int im = e >> 1, h = e & 1, i = im >> 2, m = im & 3;
    const double xi = x[im];
    double g = h ? 0.0 : xi - tg[rarg[i] * 4 + m];
#pragma unroll 5
    for (int j = h * nh; j < min(N, (h + 1) * nh); ++j) g += carg[j] == i ? xi - tg[j * 4 + m] : 0.0;
    g += __shfl_xor(g, 1, 2);
    if (h == 0) gx[im] = g * loss_scale;
  }
  __syncthreads();
  // back through get_real (both planes receive g), rep_to_p and mix_to_output; dWo1[c] = sum_n sum_m G_yc[n][m] conj(v[n][c][m])
  for (int e = threadIdx.x; e < N * C; e += BLOCK) {
    const int n = e / C, c = e - n * C;
    cx<double> g[4], gc[4], d = {0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m) g[m] = {gx[n * 4 + m], gx[n * 4 + m]};
    cart_from_canon_bwd(g, gc);
    const cx<double> w = {wol[c], wol[C + c]};
    const size_t base = jc + e;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      cx<double> r = cmulc(gc[m], w);
      g_v[base * 4 + m] = r.r;
      g_v[pl * 4 + base * 4 + m] = r.i;
      cfmac(d, gc[m], cx<double>{vl[e * 8 + m], vl[e * 8 + 4 + m]});
    }
    tmp[e * 2] = d.r;
    tmp[e * 2 + 1] = d.i;
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * C) {
    const int k = threadIdx.x / C, c = threadIdx.x - k * C;
    double acc = 0.0;
#pragma unroll 6
    for (int n = 0; n < N; ++n) acc += tmp[(n * C + c) * 2 + k];
    part[(size_t)b * 2 * C + k * C + c] = acc;
  }
}


}  // namespace lgn
