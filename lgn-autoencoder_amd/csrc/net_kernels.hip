// lgn-autoencoder_amd/csrc/net_kernels.hip -- the O(N)-per-jet ends of the networks, fp64:
//   encoder input   (lgn_encoder.py:284-298,376)   mass + canonical momenta + input_func_node MixReps
//   encoder latent  (lgn_encoder.py:322-331,419-583) mix_reps MixReps + rep_to_p + 'min&max' pooling
//   decoder input   (lgn_decoder.py:305-345,257-265) latent_to_graph + p_cplx_to_rep + input_func_node
//   decoder output  (lgn_decoder.py:286-295) + get_real('sum') (utils/utils.py:194-207)
//                   + ChamferLoss (utils/losses/chamfer_loss/chamfer_loss.py:17-23) forward AND backward
//   L1 + Adam       (utils/train.py:484-487, utils/initialize.py:156-158)
// Each forward/backward pair is one workgroup per jet; weight-gradient partials are one row per jet.
#include "net.hpp"
#include "net_dev.hpp"
#include "tail_dev.hpp"

namespace lgn {

LGN_STAMP_DECL
LGN_STAMP_READER(lgn_debug_stamps_net)

namespace {
constexpr double H = 0.70710678118654752440084436210484903928;

__device__ __forceinline__ double minkowski_sq_ref(const double* p) {
  // 2 E^2 - sum p^2 with the left-to-right sum of the reference's CPU reduction (zonal_functions.py:201-218)
  double q0 = p[0] * p[0], q1 = p[1] * p[1], q2 = p[2] * p[2], q3 = p[3] * p[3];
  return 2.0 * q0 - (((q0 + q1) + q2) + q3);
}
// real Cartesian -> canonical (p_to_rep, zonal_functions.py:251-289)
__device__ __forceinline__ void canon_real(const double* p, cx<double> (&q)[4]) {
  q[0] = {p[0], 0.0};
  q[1] = {p[1] * H, -p[2] * H};
  q[2] = {p[3], 0.0};
  q[3] = {-p[1] * H, -p[2] * H};
}
// complex Cartesian -> canonical (p_cplx_to_rep, zonal_functions.py:292-341): c1 = (px - i py)/rt2, c3 = (-px - i py)/rt2
__device__ __forceinline__ void canon_cplx(const cx<double> (&p)[4], cx<double> (&c)[4]) {
  c[0] = p[0];
  c[1] = {(p[1].r + p[2].i) * H, (p[1].i - p[2].r) * H};
  c[2] = p[3];
  c[3] = {(-p[1].r + p[2].i) * H, (-p[1].i - p[2].r) * H};
}
// gradient of canon_cplx: G_p = J^H G_c
__device__ __forceinline__ void canon_cplx_bwd(const cx<double> (&g)[4], cx<double> (&gp)[4]) {
  gp[0] = g[0];
  gp[1] = {(g[1].r - g[3].r) * H, (g[1].i - g[3].i) * H};             // px: h (G1 - G3)
  gp[2] = {-(g[1].i + g[3].i) * H, (g[1].r + g[3].r) * H};            // py: i h (G1 + G3)
  gp[3] = g[2];
}
// canonical -> complex Cartesian (rep_to_p, zonal_functions.py:344-381): E=c0, px=(c1-c3)/rt2, py=i(c1+c3)/rt2, pz=c2
__device__ __forceinline__ void cart_from_canon(const cx<double> (&c)[4], cx<double> (&p)[4]) {
  p[0] = c[0];
  p[1] = {(c[1].r - c[3].r) * H, (c[1].i - c[3].i) * H};
  p[2] = {-(c[1].i + c[3].i) * H, (c[1].r + c[3].r) * H};
  p[3] = c[2];
}
}  // namespace

// ============================================================================================
// encoder input
// ============================================================================================
// z1/z2: two buffers the step wants zeroed before anything later in the stream touches them (the flat gradient buffer and
// the zero-initialised workspace block); folded into this first kernel instead of two memset launches.
// K input scalars per node (MixReps weight w0 [2][C][K]): x_0 = the mass, x_1 .. x_{K-1} from xs [B][N][K-1] (jet_features: the
// jet's un-rooted normsq4, then data['scalars']; lgn_encoder.py:372-411).  K = 1: xs unused.
__global__ void enc_input_fwd_kernel(int B, int N, int C, int K, const double* __restrict__ p4, const double* __restrict__ xs,
                                     const double* __restrict__ w0, const double* __restrict__ w1, double* s, double* v, double* z1,
                                     size_t n1, double* z2, size_t n2) {
  const size_t total = (size_t)B * N * C, pl = total;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n1; e += (size_t)gridDim.x * blockDim.x) z1[e] = 0.0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (size_t)gridDim.x * blockDim.x) z2[e] = 0.0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c = e % C;
    const size_t node = e / C;
    const double* p = p4 + node * 4;
    const double mass = sqrt(fabs(minkowski_sq_ref(p)));
    cx<double> q[4];
    canon_real(p, q);
    double sr = w0[c * K] * mass, si = w0[(C + c) * K] * mass;       // W00[c][0] * (mass + 0i)
    for (int k = 1; k < K; ++k) {
      const double x = xs[node * (K - 1) + (k - 1)];
      sr = __builtin_fma(w0[c * K + k], x, sr);
      si = __builtin_fma(w0[(C + c) * K + k], x, si);
    }
    s[e] = sr;
    s[pl + e] = si;
    const cx<double> w = {w1[c], w1[C + c]};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      cx<double> r = cmul(w, q[m]);
      v[e * 4 + m] = r.r;
      v[pl * 4 + e * 4 + m] = r.i;
    }
  }
}

// partial rows [nblk][(2K + 2) C]: dW00 (re [C][K], im [C][K]) then dW11 (re[C], im[C]).  One workgroup per jet:
// thread = (node, channel) writes its 2K + 2 terms to LDS, thread = (term, channel) adds them up in node order.
__global__ __launch_bounds__(BLOCK) void enc_input_bwd_kernel(int B, int N, int C, int K, const double* __restrict__ p4,
                                                             const double* __restrict__ xs, const double* __restrict__ g_s,
                                                             const double* __restrict__ g_v, double* part) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* tmp = reinterpret_cast<double*>(smem_raw);    // [N*C][2K + 2]: (g_s.r x_k)_k | (g_s.i x_k)_k | d1.r | d1.i
  const int T = 2 * K + 2;
  const int b = blockIdx.x;
  const size_t pl = (size_t)B * N * C;
  for (int i = threadIdx.x; i < N * C; i += BLOCK) {
    const int n = i / C;
    const double* p = p4 + ((size_t)b * N + n) * 4;
    const double mass = sqrt(fabs(minkowski_sq_ref(p)));
    cx<double> q[4];
    canon_real(p, q);
    const size_t e = (size_t)b * N * C + i;
    cx<double> d1 = {0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m) cfmac(d1, cx<double>{g_v[e * 4 + m], g_v[pl * 4 + e * 4 + m]}, q[m]);
    const double gr = g_s[e], gi = g_s[pl + e];
    tmp[i * T + 0] = gr * mass;
    tmp[i * T + K] = gi * mass;
    for (int k = 1; k < K; ++k) {
      const double x = xs[((size_t)b * N + n) * (K - 1) + (k - 1)];
      tmp[i * T + k] = gr * x;
      tmp[i * T + K + k] = gi * x;
    }
    tmp[i * T + 2 * K] = d1.r;
    tmp[i * T + 2 * K + 1] = d1.i;
  }
  __syncthreads();
  // output column o of the partial row: o < 2CK: (plane z, channel c, scalar k) = W00 layout; then dW11 re[C], im[C]
  for (int o = threadIdx.x; o < T * C; o += BLOCK) {
    int c, t;
    if (o < 2 * C * K) {
      const int z = o / (C * K), r = o - z * C * K;
      c = r / K;
      t = z * K + (r - c * K);
    } else {
      const int r = o - 2 * C * K, z = r / C;
      c = r - z * C;
      t = 2 * K + z;
    }
    double acc = 0.0;
    for (int n = 0; n < N; ++n) acc += tmp[(n * C + c) * T + t];
    part[(size_t)b * T * C + o] = acc;
  }
}

// ============================================================================================
// LDS staging.  The per-jet kernels below work on a few KB per jet: every global operand is copied to LDS first with
// all loads of a thread in flight together (one HBM round trip per kernel), and every later phase reads LDS only --
// round 2's versions walked global memory inside run-time loops (one round trip per channel) and ran their pooling /
// Chamfer phases on 18-60 threads: 17 us per junction kernel for a few kFLOP per jet.
// ============================================================================================
// component m of canon_cplx / cart_from_canon / their gradients from an LDS vector stored re[4] | im[4]
__device__ __forceinline__ cx<double> canon_cplx_m(const double* p, int m) {
  if (m == 0) return {p[0], p[4]};
  if (m == 2) return {p[3], p[7]};
  const double sg = m == 1 ? 1.0 : -1.0;
  return {(sg * p[1] + p[6]) * H, (sg * p[5] - p[2]) * H};
}
__device__ __forceinline__ cx<double> canon_cplx_bwd_m(const double* g, int m) {
  if (m == 0) return {g[0], g[4]};
  if (m == 3) return {g[2], g[6]};
  if (m == 1) return {(g[1] - g[3]) * H, (g[5] - g[7]) * H};
  return {-(g[5] + g[7]) * H, (g[1] + g[3]) * H};
}

// ============================================================================================
// encoder latent: MixReps -> Cartesian -> min&max pooling
//   lat_s [2][B][2Ts]  (min block, max block), lat_v [2][B][2Tv][4], idx [B][2][Ts+Tv][2] (plane, channel, min/max)
// LDS: y [N][2Ts + 8Tv] | sv [N][C][10] | w0l [2][Ts][C] | w1l [2][Tv][C]
// ============================================================================================
// the y / gy block: the latent channels of every particle -- under 'mix' there is no particle axis left, one row
__host__ __device__ inline size_t lat_y_doubles(int N, int Ts, int Tv, int pool = 0) {
  return (size_t)(pool_is_mix(pool) ? 1 : N) * (2 * Ts + 8 * Tv);
}
__host__ __device__ inline size_t lat_fwd_doubles(int N, int C, int Ts, int Tv, int pool = 0) {
  return lat_y_doubles(N, Ts, Tv, pool) + (size_t)N * C * 10 + 2 * (size_t)(Ts + Tv) * pool_mix_in(pool, N, C);
}
// node features of the jet -> sv [N][C][10] (s re, im, v re[4], im[4]) and the two mixing weights; zero_y: y / gy starts at zero
struct LatentStage {
  StageRegs<2> sr, si;
  StageRegs<4> vr, vi;
  StageRegs<1> w0, w1;
  const double *s0, *s1, *v0, *v1, *wl0, *wl1;
  int N, C, Ts, Tv, K, NY;                              // K: input channels of the two weights (C; N C under 'mix'); NY: y block
  __device__ __forceinline__ void issue(int B, int N_, int C_, int Ts_, int Tv_, const double* __restrict__ s, const double* __restrict__ v,
                                        const double* __restrict__ wl0_, const double* __restrict__ wl1_, int pool = 0) {
    N = N_; C = C_; Ts = Ts_; Tv = Tv_; wl0 = wl0_; wl1 = wl1_; K = pool_mix_in(pool, N_, C_); NY = (int)lat_y_doubles(N_, Ts_, Tv_, pool);
    const size_t pl = (size_t)B * N * C, j0 = (size_t)blockIdx.x * N * C;
    s0 = s + j0; s1 = s + pl + j0; v0 = v + j0 * 4; v1 = v + (pl + j0) * 4;
    vr.issue(v0, N * C * 4); vi.issue(v1, N * C * 4);
    sr.issue(s0, N * C); si.issue(s1, N * C);
    w0.issue(wl0, 2 * Ts * K); w1.issue(wl1, 2 * Tv * K);
  }
  __device__ __forceinline__ void commit(double* lds, bool zero_y) const {
    double* sv = lds + NY;
    double* w0l = sv + N * C * 10;
    double* w1l = w0l + 2 * Ts * K;
    if (zero_y)
      for (int e = threadIdx.x; e < NY; e += BLOCK) lds[e] = 0.0;
    vr.commit(v0, N * C * 4, [&](int e, double x) { sv[(e >> 2) * 10 + 2 + (e & 3)] = x; });
    vi.commit(v1, N * C * 4, [&](int e, double x) { sv[(e >> 2) * 10 + 6 + (e & 3)] = x; });
    sr.commit(s0, N * C, [&](int e, double x) { sv[e * 10] = x; });
    si.commit(s1, N * C, [&](int e, double x) { sv[e * 10 + 1] = x; });
    w0.commit(wl0, 2 * Ts * K, [&](int e, double x) { w0l[e] = x; });
    w1.commit(wl1, 2 * Tv * K, [&](int e, double x) { w1l[e] = x; });
  }
};
// after enc_latent_fwd_stage + a barrier.  TO_LDS: lat_l (LDS [2Tv][8]: re[4] | im[4]) also receives the latent vectors.
template <bool TO_LDS>
__device__ __forceinline__ void enc_latent_fwd_body(int B, int N, int C, int Ts, int Tv, int pool, double* lat_s, double* lat_v, int* idx,
                                                    double* lds, double* lat_l) {
  const int b = blockIdx.x, TT = Ts + Tv;
  const int P = pool_blocks(pool), PN = pool_n(pool);
  bool need_mean = false;
  for (int i = 0; i < PN; ++i) need_mean |= pool_op(pool, i) == LGN_POOL_MEAN;
  const int YS = 2 * Ts + 8 * Tv;                       // per-node stride
  double* y = lds;                                      // [n][t] : scalars 2 (re,im), vectors 8 (cart re[4], im[4])
  const double* sv = y + N * YS;
  const double* w0l = sv + N * C * 10;
  const double* w1l = w0l + 2 * Ts * C;
  STAMP(1);
  // MixReps to the latent channels + rep_to_p: one (particle, vector channel) per thread (N Tv = 240 at cfg2: one round), the
  // few scalar channels in a second, short loop
  for (int e = threadIdx.x; e < N * Tv; e += BLOCK) {
    const int n = e / Tv, tv = e - n * Tv;
    const double* x = sv + n * C * 10;
    cx<double> acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, pc[4];
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
      const cx<double> w = {w1l[tv * C + c], w1l[Tv * C + tv * C + c]};
#pragma unroll
      for (int m = 0; m < 4; ++m) cfma(acc[m], w, cx<double>{x[c * 10 + 2 + m], x[c * 10 + 6 + m]});
    }
    cart_from_canon(acc, pc);
    double* o = y + n * YS + 2 * Ts + 8 * tv;
#pragma unroll
    for (int m = 0; m < 4; ++m) { o[m] = pc[m].r; o[4 + m] = pc[m].i; }
  }
  for (int e = threadIdx.x; e < N * Ts; e += BLOCK) {
    const int n = e / Ts, t = e - n * Ts;
    const double* x = sv + n * C * 10;
    cx<double> acc = {0, 0};
#pragma unroll 4
    for (int c = 0; c < C; ++c) cfma(acc, cx<double>{w0l[t * C + c], w0l[Ts * C + t * C + c]}, cx<double>{x[c * 10], x[c * 10 + 1]});
    y[n * YS + 2 * t] = acc.r;
    y[n * YS + 2 * t + 1] = acc.i;
  }
  __syncthreads();
  STAMP(2);
  // arg-min / arg-max over the particles (first occurrence, padded particles included) per (plane, channel): 8 lanes per
  // item, lane = every 8th particle, then a butterfly over the 8 lanes on (value, index) -- ties go to the lower index, as the
  // sequential scan of the reference's torch.min / torch.max does.  (Round 2: one thread per item walked all N particles.)
  const int l8 = threadIdx.x & 7;
  constexpr int NONE = 0x7fffffff;
  for (int it = threadIdx.x >> 3; it < 2 * TT; it += BLOCK / 8) {
    const int z = it / TT, t = it - z * TT;
    int imin = NONE, imax = NONE;
    double smin = 0, smax = 0;
#pragma unroll 4
    for (int n = l8; n < N; n += 8) {
      double lo, hi;
      if (t < Ts) {
        const double val = y[n * YS + 2 * t + z];
        lo = val;                 // get_min_features: the value itself (lgn_encoder.py:544-545)
        hi = val * val;           // get_max_features: E^2 - |p|^2 with no spatial part = value^2 (lgn_encoder.py:568-569)
      } else {
        const double* o = y + n * YS + 2 * Ts + 8 * (t - Ts) + 4 * z;
        lo = hi = o[0] * o[0] - ((o[1] * o[1] + o[2] * o[2]) + o[3] * o[3]);
      }
      if (imin == NONE || lo < smin) { smin = lo; imin = n; }
      if (imax == NONE || hi > smax) { smax = hi; imax = n; }
    }
#pragma unroll
    for (int off = 4; off; off >>= 1) {
      const double om = __shfl_xor(smin, off, 8), oM = __shfl_xor(smax, off, 8);
      const int oi = __shfl_xor(imin, off, 8), oI = __shfl_xor(imax, off, 8);
      if (oi != NONE && (imin == NONE || om < smin || (om == smin && oi < imin))) { smin = om; imin = oi; }
      if (oI != NONE && (imax == NONE || oM > smax || (oM == smax && oI < imax))) { smax = oM; imax = oI; }
    }
    imin = __shfl(imin, 0, 8);        // (identical on all 8 lanes unless NaNs are present: lane 0 holds the sequential scan's answer)
    imax = __shfl(imax, 0, 8);
    if (l8 < 2) idx[(((size_t)b * 2 + z) * TT + t) * 2 + l8] = l8 ? imax : imin;
    // the item's components in y: one (scalar channel) or four (vector channel), at y[n * YS + y0 + m]
    const int nc = t < Ts ? 1 : 4, y0 = t < Ts ? 2 * t + z : 2 * Ts + 8 * (t - Ts) + 4 * z;
    double mean[4] = {0, 0, 0, 0};     // torch.mean over the particle axis, padded particles included (lgn_encoder.py:450-452)
    if (need_mean) {
      for (int n = l8; n < N; n += 8)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < nc) mean[m] += y[n * YS + y0 + m];
#pragma unroll
      for (int off = 4; off; off >>= 1)
#pragma unroll
        for (int m = 0; m < 4; ++m) mean[m] += __shfl_xor(mean[m], off, 8);      // (every lane ends with the same sum)
#pragma unroll
      for (int m = 0; m < 4; ++m) mean[m] /= (double)N;
    }
    // output blocks: '&' = one block per pooling, '+' = one block holding their average (lgn_encoder.py:454-496)
    for (int k = l8; k < P * nc; k += 8) {
      const int pb = k / nc, m = k - pb * nc;
      auto pick = [&](int op) -> double {
        if (op == LGN_POOL_MIN) return y[imin * YS + y0 + m];
        if (op == LGN_POOL_MAX) return y[imax * YS + y0 + m];
        return m == 0 ? mean[0] : (m == 1 ? mean[1] : (m == 2 ? mean[2] : mean[3]));
      };
      double val;
      if (pool_avg(pool)) {
        val = pick(pool_op(pool, 0));
        for (int i = 1; i < PN; ++i) val += pick(pool_op(pool, i));
        val /= (double)PN;
      } else {
        val = pick(pool_op(pool, pb));
      }
      if (t < Ts) {
        lat_s[((size_t)z * B + b) * P * Ts + pb * Ts + t] = val;
      } else {
        const int tv = t - Ts;
        lat_v[(((size_t)z * B + b) * P * Tv + pb * Tv + tv) * 4 + m] = val;
        if constexpr (TO_LDS) lat_l[(pb * Tv + tv) * 8 + 4 * z + m] = val;
      }
    }
  }
}
// map_to_latent = 'mix': latent channel t = sum over ALL (particle, channel) pairs k = n C + c of W[t][k] x[k] (the reference reshapes
// the node features to (2, B, 1, N C, d) before the MixReps, lgn_encoder.py:313-319), then rep_to_p.  8 lanes per latent channel.
// LDS as for the pooled maps, the weights being w0l [2][Ts][N C] | w1l [2][Tv][N C]; the y block is unused.
template <bool TO_LDS>
__device__ __forceinline__ void enc_latent_mix_fwd_body(int B, int N, int C, int Ts, int Tv, double* lat_s, double* lat_v, double* lds,
                                                        double* lat_l) {
  const int b = blockIdx.x, TT = Ts + Tv, K = N * C;
  const double* sv = lds + (2 * Ts + 8 * Tv);
  const double* w0l = sv + K * 10;
  const double* w1l = w0l + 2 * Ts * K;
  const int l8 = threadIdx.x & 7;
  for (int t = threadIdx.x >> 3; t < TT; t += BLOCK / 8) {
    if (t < Ts) {
      cx<double> acc = {0, 0};
      for (int k = l8; k < K; k += 8) cfma(acc, cx<double>{w0l[t * K + k], w0l[Ts * K + t * K + k]}, cx<double>{sv[k * 10], sv[k * 10 + 1]});
#pragma unroll
      for (int off = 4; off; off >>= 1) { acc.r += __shfl_xor(acc.r, off, 8); acc.i += __shfl_xor(acc.i, off, 8); }
      if (l8 < 2) lat_s[((size_t)l8 * B + b) * Ts + t] = l8 ? acc.i : acc.r;
    } else {
      const int tv = t - Ts;
      cx<double> acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, pc[4];
      for (int k = l8; k < K; k += 8) {
        const cx<double> w = {w1l[tv * K + k], w1l[Tv * K + tv * K + k]};
#pragma unroll
        for (int m = 0; m < 4; ++m) cfma(acc[m], w, cx<double>{sv[k * 10 + 2 + m], sv[k * 10 + 6 + m]});
      }
#pragma unroll
      for (int off = 4; off; off >>= 1)
#pragma unroll
        for (int m = 0; m < 4; ++m) { acc[m].r += __shfl_xor(acc[m].r, off, 8); acc[m].i += __shfl_xor(acc[m].i, off, 8); }
      cart_from_canon(acc, pc);
      const int z = l8 >> 2, m = l8 & 3;
      const cx<double> pm = m == 0 ? pc[0] : (m == 1 ? pc[1] : (m == 2 ? pc[2] : pc[3]));
      const double val = z ? pm.i : pm.r;
      lat_v[(((size_t)z * B + b) * Tv + tv) * 4 + m] = val;
      if constexpr (TO_LDS) lat_l[tv * 8 + 4 * z + m] = val;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void enc_latent_fwd_kernel(int B, int N, int C, int Ts, int Tv, int pool,
                                                              const double* __restrict__ s, const double* __restrict__ v,
                                                              const double* __restrict__ wl0, const double* __restrict__ wl1,
                                                              double* lat_s, double* lat_v, int* idx) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* lds = reinterpret_cast<double*>(smem_raw);
  LatentStage st;
  st.issue(B, N, C, Ts, Tv, s, v, wl0, wl1, pool);
  st.commit(lds, false);
  __syncthreads();
  if (pool_is_mix(pool)) enc_latent_mix_fwd_body<false>(B, N, C, Ts, Tv, lat_s, lat_v, lds, nullptr);
  else enc_latent_fwd_body<false>(B, N, C, Ts, Tv, pool, lat_s, lat_v, idx, lds, nullptr);
}

// backward: scatter the latent gradient to the selected particles, undo rep_to_p and the MixReps.
// part row per jet: dWl0 [2][Ts][C] then dWl1 [2][Tv][C].  The jet's node features and the mixing weights are staged
// in LDS once; the weight gradient runs over (channel pair, node part) items whose parts meet in LDS in a fixed order.
// LDS: gy [N][2Ts + 8Tv] | sv [N][C][10] | w0l | w1l | red [LAT_PARTS][TT*C][2]
constexpr int LAT_PARTS = 6;
__host__ __device__ inline size_t lat_bwd_doubles(int N, int C, int Ts, int Tv, int pool = 0) {
  return lat_fwd_doubles(N, C, Ts, Tv, pool) + (pool_is_mix(pool) ? 0 : (size_t)LAT_PARTS * (Ts + Tv) * C * 2);
}
// FROM_LDS: the latent-vector gradient of this jet comes from g_lat_l (LDS, [2Tv][8]: re[4] | im[4]) instead of g_lat_v
// `pre`: the pooling indices (and latent-scalar gradients) of this thread's (plane, channel), fetched with the staging loads
struct LatentBwdPrefetch {
  int n[2];
  double gs[4];
  bool valid;
  __device__ __forceinline__ void issue(int B, int Ts, int Tv, int pool, const int* __restrict__ idx, const double* __restrict__ g_lat_s) {
    const int TT = Ts + Tv, e = threadIdx.x, b = blockIdx.x, P = pool_blocks(pool);
    valid = 2 * TT <= BLOCK && !pool_is_mix(pool);
    if (valid && e < 2 * TT) {
      const int z = e / TT, t = e - z * TT;
#pragma unroll
      for (int kind = 0; kind < 2; ++kind) n[kind] = idx[(((size_t)b * 2 + z) * TT + t) * 2 + kind];
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) gs[pb] = (t < Ts && pb < P) ? g_lat_s[((size_t)z * B + b) * P * Ts + pb * Ts + t] : 0.0;
    }
  }
};
template <bool FROM_LDS>
__device__ __forceinline__ void enc_latent_bwd_body(int B, int N, int C, int Ts, int Tv, int pool, const double* __restrict__ g_lat_s,
                                                    const double* __restrict__ g_lat_v, const double* g_lat_l,
                                                    const int* __restrict__ idx, double* g_s, double* g_v, double* part, double* lds,
                                                    const LatentBwdPrefetch& pre) {
  const int b = blockIdx.x, TT = Ts + Tv;
  const int YS = 2 * Ts + 8 * Tv;
  double* gy = lds;                                     // [N][YS] same layout as y in the forward; vectors become canonical grads
  const double* sv = gy + N * YS;                       // [N][C][10]: s re, im, v re[4], im[4]
  const double* w0l = sv + N * C * 10;                  // [2][Ts][C]
  const double* w1l = w0l + 2 * Ts * C;                 // [2][Tv][C]
  double* red = const_cast<double*>(w1l) + 2 * Tv * C;  // [LAT_PARTS][TT*C][2]
  const size_t pl = (size_t)B * N * C;
  STAMP(14);
  const int P = pool_blocks(pool), PN = pool_n(pool);
  for (int e = threadIdx.x; e < 2 * TT; e += BLOCK) {    // (plane, channel) owners: no write conflicts
    const int z = e / TT, t = e - z * TT;
    int nmm[2];                                          // arg-min, arg-max of this (plane, channel)
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) nmm[kind] = pre.valid ? pre.n[kind] : idx[(((size_t)b * 2 + z) * TT + t) * 2 + kind];
    const int nc = t < Ts ? 1 : 4, y0 = t < Ts ? 2 * t + z : 2 * Ts + 8 * (t - Ts) + 4 * z;
    // '&': block pb carries the gradient of pooling pb; '+': the one block's gradient / PN goes to every pooling
    const int nscat = pool_avg(pool) ? PN : P;
    for (int i = 0; i < nscat; ++i) {
      const int pb = pool_avg(pool) ? 0 : i, op = pool_op(pool, i);
      double g[4] = {0, 0, 0, 0};
      if (t < Ts) {
        g[0] = pre.valid ? (pb == 0 ? pre.gs[0] : (pb == 1 ? pre.gs[1] : (pb == 2 ? pre.gs[2] : pre.gs[3])))
                         : g_lat_s[((size_t)z * B + b) * P * Ts + pb * Ts + t];
      } else {
        const int tv = t - Ts;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if constexpr (FROM_LDS) g[m] = g_lat_l[(pb * Tv + tv) * 8 + 4 * z + m];
          else g[m] = g_lat_v[(((size_t)z * B + b) * P * Tv + pb * Tv + tv) * 4 + m];
        }
      }
      if (pool_avg(pool))
#pragma unroll
        for (int m = 0; m < 4; ++m) g[m] /= (double)PN;
      if (op == LGN_POOL_MEAN) {
#pragma unroll
        for (int m = 0; m < 4; ++m) g[m] /= (double)N;
        for (int n = 0; n < N; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            if (m < nc) gy[n * YS + y0 + m] += g[m];
      } else {
        const int n = nmm[op == LGN_POOL_MAX];
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < nc) gy[n * YS + y0 + m] += g[m];
      }
    }
  }
  __syncthreads();
  STAMP(15);
  for (int e = threadIdx.x; e < N * Tv; e += BLOCK) {    // Cartesian gradient -> canonical gradient, in place
    const int n = e / Tv, tv = e - n * Tv;
    double* o = gy + n * YS + 2 * Ts + 8 * tv;
    cx<double> g[4], gc[4];
    for (int m = 0; m < 4; ++m) g[m] = {o[m], o[4 + m]};
    cart_from_canon_bwd(g, gc);
    for (int m = 0; m < 4; ++m) { o[m] = gc[m].r; o[4 + m] = gc[m].i; }
  }
  __syncthreads();
  STAMP(16);
  for (int e = threadIdx.x; e < N * C; e += BLOCK) {     // gradient w.r.t. the last level's node features
    const int n = e / C, c = e - n * C;
    const size_t base = ((size_t)b * N + n) * C + c;
    cx<double> as = {0, 0}, av[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll 2
    for (int t = 0; t < Ts; ++t)
      cfmac(as, cx<double>{gy[n * YS + 2 * t], gy[n * YS + 2 * t + 1]}, cx<double>{w0l[t * C + c], w0l[Ts * C + t * C + c]});
#pragma unroll 4
    for (int t = 0; t < Tv; ++t) {
      const cx<double> w = {w1l[t * C + c], w1l[Tv * C + t * C + c]};
      const double* o = gy + n * YS + 2 * Ts + 8 * t;
#pragma unroll
      for (int m = 0; m < 4; ++m) cfmac(av[m], cx<double>{o[m], o[4 + m]}, w);
    }
    g_s[base] = as.r;
    g_s[pl + base] = as.i;
#pragma unroll
    for (int m = 0; m < 4; ++m) { g_v[base * 4 + m] = av[m].r; g_v[pl * 4 + base * 4 + m] = av[m].i; }
  }
  // weight gradients of this jet: item = (node part, latent channel t, node channel c)
  STAMP(17);
  const int nper = (N + LAT_PARTS - 1) / LAT_PARTS;
  for (int e = threadIdx.x; e < LAT_PARTS * TT * C; e += BLOCK) {
    const int pi = e / (TT * C), r = e - pi * TT * C, t = r / C, c = r - t * C;
    const int n1 = min(N, (pi + 1) * nper);
    cx<double> acc = {0, 0};
    if (t < Ts) {
#pragma unroll 5
      for (int n = pi * nper; n < n1; ++n)
        cfmac(acc, cx<double>{gy[n * YS + 2 * t], gy[n * YS + 2 * t + 1]}, cx<double>{sv[(n * C + c) * 10], sv[(n * C + c) * 10 + 1]});
    } else {
      const int tv = t - Ts;
#pragma unroll 5
      for (int n = pi * nper; n < n1; ++n) {
        const double* o = gy + n * YS + 2 * Ts + 8 * tv;
        const double* x = sv + (n * C + c) * 10;
#pragma unroll
        for (int m = 0; m < 4; ++m) cfmac(acc, cx<double>{o[m], o[4 + m]}, cx<double>{x[2 + m], x[6 + m]});
      }
    }
    red[e * 2] = acc.r;
    red[e * 2 + 1] = acc.i;
  }
  __syncthreads();
  STAMP(18);
  double* row = part + (size_t)b * 2 * TT * C;
  for (int e = threadIdx.x; e < TT * C; e += BLOCK) {
    const int t = e / C, c = e - t * C;
    double ar = red[e * 2], ai = red[e * 2 + 1];
    for (int pi = 1; pi < LAT_PARTS; ++pi) { ar += red[(pi * TT * C + e) * 2]; ai += red[(pi * TT * C + e) * 2 + 1]; }
    if (t < Ts) {
      row[t * C + c] = ar;
      row[Ts * C + t * C + c] = ai;
    } else {
      row[2 * Ts * C + (t - Ts) * C + c] = ar;
      row[2 * Ts * C + Tv * C + (t - Ts) * C + c] = ai;
    }
  }
}
// backward of the 'mix' map.  part row per jet: dWl0 [2][Ts][N C] then dWl1 [2][Tv][N C] -- every entry is one product, no sum
// inside the jet.  The gradient of the latent channels (vectors: undone rep_to_p) sits in the first 2 Ts + 8 Tv doubles of the y block.
template <bool FROM_LDS>
__device__ __forceinline__ void enc_latent_mix_bwd_body(int B, int N, int C, int Ts, int Tv, const double* __restrict__ g_lat_s,
                                                        const double* __restrict__ g_lat_v, const double* g_lat_l, double* g_s, double* g_v,
                                                        double* part, double* lds) {
  const int b = blockIdx.x, TT = Ts + Tv, K = N * C;
  double* gy = lds;                                     // [2 Ts + 8 Tv]: scalars (re, im), vectors canonical re[4] | im[4]
  const double* sv = lds + (2 * Ts + 8 * Tv);
  const double* w0l = sv + K * 10;
  const double* w1l = w0l + 2 * Ts * K;
  const size_t pl = (size_t)B * K;
  for (int t = threadIdx.x; t < TT; t += BLOCK) {
    if (t < Ts) {
      gy[2 * t] = g_lat_s ? g_lat_s[((size_t)0 * B + b) * Ts + t] : 0.0;
      gy[2 * t + 1] = g_lat_s ? g_lat_s[((size_t)1 * B + b) * Ts + t] : 0.0;
    } else {
      const int tv = t - Ts;
      cx<double> g[4], gc[4];
      for (int m = 0; m < 4; ++m) {
        if constexpr (FROM_LDS) g[m] = {g_lat_l[tv * 8 + m], g_lat_l[tv * 8 + 4 + m]};
        else g[m] = {g_lat_v[(((size_t)0 * B + b) * Tv + tv) * 4 + m], g_lat_v[(((size_t)1 * B + b) * Tv + tv) * 4 + m]};
      }
      cart_from_canon_bwd(g, gc);
      double* o = gy + 2 * Ts + 8 * tv;
      for (int m = 0; m < 4; ++m) { o[m] = gc[m].r; o[4 + m] = gc[m].i; }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += BLOCK) {         // gradient w.r.t. the last level's node features
    cx<double> as = {0, 0}, av[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int t = 0; t < Ts; ++t) cfmac(as, cx<double>{gy[2 * t], gy[2 * t + 1]}, cx<double>{w0l[t * K + k], w0l[Ts * K + t * K + k]});
    for (int t = 0; t < Tv; ++t) {
      const cx<double> w = {w1l[t * K + k], w1l[Tv * K + t * K + k]};
      const double* o = gy + 2 * Ts + 8 * t;
#pragma unroll
      for (int m = 0; m < 4; ++m) cfmac(av[m], cx<double>{o[m], o[4 + m]}, w);
    }
    const size_t base = (size_t)b * K + k;
    g_s[base] = as.r;
    g_s[pl + base] = as.i;
#pragma unroll
    for (int m = 0; m < 4; ++m) { g_v[base * 4 + m] = av[m].r; g_v[pl * 4 + base * 4 + m] = av[m].i; }
  }
  double* row = part + (size_t)b * 2 * TT * K;
  for (int e = threadIdx.x; e < TT * K; e += BLOCK) {
    const int t = e / K, k = e - t * K;
    cx<double> acc = {0, 0};
    if (t < Ts) {
      cfmac(acc, cx<double>{gy[2 * t], gy[2 * t + 1]}, cx<double>{sv[k * 10], sv[k * 10 + 1]});
      row[t * K + k] = acc.r;
      row[Ts * K + t * K + k] = acc.i;
    } else {
      const int tv = t - Ts;
      const double* o = gy + 2 * Ts + 8 * tv;
#pragma unroll
      for (int m = 0; m < 4; ++m) cfmac(acc, cx<double>{o[m], o[4 + m]}, cx<double>{sv[k * 10 + 2 + m], sv[k * 10 + 6 + m]});
      row[2 * Ts * K + tv * K + k] = acc.r;
      row[2 * Ts * K + Tv * K + tv * K + k] = acc.i;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void enc_latent_bwd_kernel(int B, int N, int C, int Ts, int Tv, int pool,
                                                              const double* __restrict__ s, const double* __restrict__ v,
                                                              const double* __restrict__ wl0, const double* __restrict__ wl1,
                                                              const double* __restrict__ g_lat_s, const double* __restrict__ g_lat_v,
                                                              const int* __restrict__ idx, double* g_s, double* g_v, double* part) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* lds = reinterpret_cast<double*>(smem_raw);
  LatentStage st;
  LatentBwdPrefetch pre;
  st.issue(B, N, C, Ts, Tv, s, v, wl0, wl1, pool);
  pre.issue(B, Ts, Tv, pool, idx, g_lat_s);
  st.commit(lds, true);                                  // same regions as the forward; gy takes y's place and starts at zero
  __syncthreads();
  if (pool_is_mix(pool)) enc_latent_mix_bwd_body<false>(B, N, C, Ts, Tv, g_lat_s, g_lat_v, nullptr, g_s, g_v, part, lds);
  else enc_latent_bwd_body<false>(B, N, C, Ts, Tv, pool, g_lat_s, g_lat_v, nullptr, idx, g_s, g_v, part, lds, pre);
}

// ============================================================================================
// decoder input: latent vectors -> particles (latent_to_graph) -> canonical momenta -> input_func_node
//   pdec [2][B][N][4]; s0 [2][B][N][C] = W00[c] (1+1i); v0 [2][B][N][C][4] = W11[c] pc[n]
// LDS: wgl [2][N][Tin] | latl [Tin][8] | cartl [N][8] | w01 [4C] (W00 re, im, W11 re, im)
// ============================================================================================
__host__ __device__ inline size_t dec_in_fwd_doubles(int N, int C, int Tin) {
  return 2 * (size_t)N * Tin + (size_t)Tin * 8 + (size_t)N * 8 + 4 * (size_t)C;
}
// weights of the decoder input -> LDS (wgl, w01); with_lat: also this jet's latent vectors from global memory
struct DecInFwdStage {
  StageRegs<4> wg;
  StageRegs<1> a0, a1, lr, li;
  const double *wg1, *w0, *w1, *l0, *l1;
  int N, C, Tin;
  bool with_lat;
  __device__ __forceinline__ void issue(int B, int N_, int C_, int Tin_, const double* __restrict__ lat_v, const double* __restrict__ wg1_,
                                        const double* __restrict__ w0_, const double* __restrict__ w1_, bool with_lat_) {
    N = N_; C = C_; Tin = Tin_; wg1 = wg1_; w0 = w0_; w1 = w1_; with_lat = with_lat_;
    wg.issue(wg1, 2 * N * Tin);
    a0.issue(w0, 2 * C); a1.issue(w1, 2 * C);
    if (with_lat) {
      l0 = lat_v + (size_t)blockIdx.x * Tin * 4; l1 = lat_v + ((size_t)B + blockIdx.x) * Tin * 4;
      lr.issue(l0, Tin * 4); li.issue(l1, Tin * 4);
    }
  }
  __device__ __forceinline__ void commit(double* lds) const {
    double* wgl = lds;
    double* latl = wgl + 2 * N * Tin;
    double* w01 = latl + Tin * 8 + N * 8;
    wg.commit(wg1, 2 * N * Tin, [&](int e, double x) { wgl[e] = x; });
    a0.commit(w0, 2 * C, [&](int e, double x) { w01[e] = x; });
    a1.commit(w1, 2 * C, [&](int e, double x) { w01[2 * C + e] = x; });
    if (with_lat) {
      lr.commit(l0, Tin * 4, [&](int e, double x) { latl[(e >> 2) * 8 + (e & 3)] = x; });
      li.commit(l1, Tin * 4, [&](int e, double x) { latl[(e >> 2) * 8 + 4 + (e & 3)] = x; });
    }
  }
};
__device__ __forceinline__ void dec_input_fwd_body(int B, int N, int C, int Tin, double* pdec, double* s0, double* v0, double* lds) {
  const double* wgl = lds;                               // [2][N][Tin]
  const double* latl = wgl + 2 * N * Tin;                // [Tin][8]
  double* cartl = const_cast<double*>(latl) + Tin * 8;   // [N][8] complex Cartesian momenta of the decoder's particles
  const double* w01 = cartl + N * 8;
  const int b = blockIdx.x;
  const size_t plp = (size_t)B * N * 4, pl = (size_t)B * N * C;
  STAMP(3);
  for (int e = threadIdx.x; e < N * 4; e += BLOCK) {     // (particle, component): latent_to_graph
    const int n = e >> 2, m = e & 3;
    cx<double> acc = {0, 0};
#pragma unroll 8
    for (int t = 0; t < Tin; ++t)
      cfma(acc, cx<double>{wgl[n * Tin + t], wgl[N * Tin + n * Tin + t]}, cx<double>{latl[t * 8 + m], latl[t * 8 + 4 + m]});
    cartl[n * 8 + m] = acc.r;
    cartl[n * 8 + 4 + m] = acc.i;
  }
  __syncthreads();
  STAMP(4);
  for (int e = threadIdx.x; e < N * 4; e += BLOCK) {
    const cx<double> pc = canon_cplx_m(cartl + (e >> 2) * 8, e & 3);
    pdec[(size_t)b * N * 4 + e] = pc.r;
    pdec[plp + (size_t)b * N * 4 + e] = pc.i;
  }
  for (int e = threadIdx.x; e < N * C; e += BLOCK) {
    const int c = e % C;
    // W00 * (1 + 1i): the zonal (0,0) function is ones on both planes (zonal_functions.py:182-186)
    s0[(size_t)b * N * C + e] = w01[c] - w01[C + c];
    s0[pl + (size_t)b * N * C + e] = w01[C + c] + w01[c];
  }
  for (int e = threadIdx.x; e < N * C * 4; e += BLOCK) {  // (particle, channel, component)
    const int i = e >> 2, m = e & 3, n = i / C, c = i - n * C;
    const cx<double> r = cmul(cx<double>{w01[2 * C + c], w01[3 * C + c]}, canon_cplx_m(cartl + n * 8, m));
    v0[(size_t)b * N * C * 4 + e] = r.r;
    v0[pl * 4 + (size_t)b * N * C * 4 + e] = r.i;
  }
}
__global__ __launch_bounds__(BLOCK) void dec_input_fwd_kernel(int B, int N, int C, int Tin, const double* __restrict__ lat_v,
                                                             const double* __restrict__ wg1, const double* __restrict__ w0,
                                                             const double* __restrict__ w1, double* pdec, double* s0, double* v0) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* lds = reinterpret_cast<double*>(smem_raw);
  DecInFwdStage st;
  st.issue(B, N, C, Tin, lat_v, wg1, w0, w1, true);
  st.commit(lds);
  __syncthreads();
  dec_input_fwd_body(B, N, C, Tin, pdec, s0, v0, lds);
}

// backward.  g_p holds the gradient w.r.t. pdec accumulated by the levels.  part row per jet:
//   dW00 [2][C] | dW11 [2][C] | dWg1 [2][N][Tin]
// LDS: gcan [N][8] | gcart [N][8] | wgl [2][N][Tin] | latl [Tin][8] | gp_l [N][8] | pd_l [N][8] | gs_l [N*C][2] | gv_l [N*C][8] | w1l [2C]
// (round 6: the input-mixing terms tmp [N*C][4] take the first half of each gv_l row once the momenta gradient has read it -- one
//  barrier more, 29 KB less at N = 150, C = 6: that shape's decoder input stage now fits a CU's LDS, 150 of 160 KB)
__host__ __device__ inline size_t dec_in_bwd_doubles(int N, int C, int Tin) {
  return (size_t)N * 16 + 2 * (size_t)N * Tin + (size_t)Tin * 8 + (size_t)N * 16 + (size_t)N * C * 10 + 2 * (size_t)C;
}
struct DecInBwdStage {
  StageRegs<4> gvr, gvi, wg;
  StageRegs<1> gpr, gpi, pdr, pdi, gsr, gsi, lr, li, a1;
  const double *gv0, *gv1, *wg1, *gp0, *gp1, *pd0, *pd1, *gs0, *gs1, *l0, *l1, *w1;
  int N, C, Tin;
  __device__ __forceinline__ void issue(int B, int N_, int C_, int Tin_, const double* __restrict__ lat_v, const double* __restrict__ wg1_,
                                        const double* __restrict__ w1_, const double* __restrict__ pdec, const double* __restrict__ g_p,
                                        const double* __restrict__ g_s0, const double* __restrict__ g_v0) {
    N = N_; C = C_; Tin = Tin_; wg1 = wg1_; w1 = w1_;
    const int b = blockIdx.x;
    const size_t plp = (size_t)B * N * 4, pl = (size_t)B * N * C, j4 = (size_t)b * N * 4, jc = (size_t)b * N * C;
    gv0 = g_v0 + jc * 4; gv1 = g_v0 + (pl + jc) * 4; gp0 = g_p + j4; gp1 = g_p + plp + j4; pd0 = pdec + j4; pd1 = pdec + plp + j4;
    gs0 = g_s0 + jc; gs1 = g_s0 + pl + jc; l0 = lat_v + (size_t)b * Tin * 4; l1 = lat_v + ((size_t)B + b) * Tin * 4;
    gvr.issue(gv0, N * C * 4); gvi.issue(gv1, N * C * 4); wg.issue(wg1, 2 * N * Tin);
    gpr.issue(gp0, N * 4); gpi.issue(gp1, N * 4); pdr.issue(pd0, N * 4); pdi.issue(pd1, N * 4);
    gsr.issue(gs0, N * C); gsi.issue(gs1, N * C); lr.issue(l0, Tin * 4); li.issue(l1, Tin * 4); a1.issue(w1, 2 * C);
  }
  __device__ __forceinline__ void commit(double* lds) const {
    double* wgl = lds + N * 16;
    double* latl = wgl + 2 * N * Tin;
    double* gp_l = latl + Tin * 8;
    double* pd_l = gp_l + N * 8;
    double* gs_l = pd_l + N * 8;
    double* gv_l = gs_l + N * C * 2;
    double* w1l = gv_l + N * C * 8;
    gvr.commit(gv0, N * C * 4, [&](int e, double x) { gv_l[(e >> 2) * 8 + (e & 3)] = x; });
    gvi.commit(gv1, N * C * 4, [&](int e, double x) { gv_l[(e >> 2) * 8 + 4 + (e & 3)] = x; });
    wg.commit(wg1, 2 * N * Tin, [&](int e, double x) { wgl[e] = x; });
    gpr.commit(gp0, N * 4, [&](int e, double x) { gp_l[(e >> 2) * 8 + (e & 3)] = x; });
    gpi.commit(gp1, N * 4, [&](int e, double x) { gp_l[(e >> 2) * 8 + 4 + (e & 3)] = x; });
    pdr.commit(pd0, N * 4, [&](int e, double x) { pd_l[(e >> 2) * 8 + (e & 3)] = x; });
    pdi.commit(pd1, N * 4, [&](int e, double x) { pd_l[(e >> 2) * 8 + 4 + (e & 3)] = x; });
    gsr.commit(gs0, N * C, [&](int e, double x) { gs_l[e * 2] = x; });
    gsi.commit(gs1, N * C, [&](int e, double x) { gs_l[e * 2 + 1] = x; });
    lr.commit(l0, Tin * 4, [&](int e, double x) { latl[(e >> 2) * 8 + (e & 3)] = x; });
    li.commit(l1, Tin * 4, [&](int e, double x) { latl[(e >> 2) * 8 + 4 + (e & 3)] = x; });
    a1.commit(w1, 2 * C, [&](int e, double x) { w1l[e] = x; });
  }
};
// after dec_input_bwd_stage + a barrier.  TO_LDS: g_lat_l (LDS [Tin][8]) also receives this jet's latent-vector gradient.
template <bool TO_LDS>
__device__ __forceinline__ void dec_input_bwd_body(int B, int N, int C, int Tin, double* g_lat_v, double* part, double* lds, double* g_lat_l) {
  double* gcan = lds;                                    // [N][8] gradient w.r.t. the canonical momenta
  double* gcart = gcan + N * 8;                          // [N][8] gradient w.r.t. the complex Cartesian momenta
  const double* wgl = gcart + N * 8;                     // [2][N][Tin]
  const double* latl = wgl + 2 * N * Tin;                // [Tin][8]
  const double* gp_l = latl + Tin * 8;
  const double* pd_l = gp_l + N * 8;
  const double* gs_l = pd_l + N * 8;
  double* gv_l = const_cast<double*>(gs_l) + N * C * 2;  // [N*C][8]; then tmp: the input-mixing terms in [i][0..3]
  const double* w1l = gv_l + N * C * 8;
  const int b = blockIdx.x;
  double* row = part + (size_t)b * (4 * C + 2 * N * Tin);
  STAMP(11);
  for (int e = threadIdx.x; e < N * 4; e += BLOCK) {     // (n, m): G_pc[m] = g_p + sum_c g_v0[c][m] conj(W11[c])
    const int n = e >> 2, m = e & 3;
    cx<double> g = {gp_l[n * 8 + m], gp_l[n * 8 + 4 + m]};
#pragma unroll 4
    for (int c = 0; c < C; ++c)
      cfmac(g, cx<double>{gv_l[(n * C + c) * 8 + m], gv_l[(n * C + c) * 8 + 4 + m]}, cx<double>{w1l[c], w1l[C + c]});
    gcan[n * 8 + m] = g.r;
    gcan[n * 8 + 4 + m] = g.i;
  }
  __syncthreads();                                       // every read of gv_l by ANOTHER thread is done: row i now belongs to thread i
  for (int i = threadIdx.x; i < N * C; i += BLOCK) {     // (n, c): terms of dW00, dW11
    const int n = i / C;
    cx<double> d0 = {0, 0}, d1 = {0, 0};
    cfmac(d0, cx<double>{gs_l[i * 2], gs_l[i * 2 + 1]}, cx<double>{1.0, 1.0});
#pragma unroll
    for (int m = 0; m < 4; ++m)
      cfmac(d1, cx<double>{gv_l[i * 8 + m], gv_l[i * 8 + 4 + m]}, cx<double>{pd_l[n * 8 + m], pd_l[n * 8 + 4 + m]});
    gv_l[i * 8 + 0] = d0.r;  gv_l[i * 8 + 1] = d0.i;  gv_l[i * 8 + 2] = d1.r;  gv_l[i * 8 + 3] = d1.i;
  }
  __syncthreads();
  STAMP(12);
  for (int e = threadIdx.x; e < N * 4; e += BLOCK) {
    const cx<double> gc = canon_cplx_bwd_m(gcan + (e >> 2) * 8, e & 3);
    gcart[(e >> 2) * 8 + (e & 3)] = gc.r;
    gcart[(e >> 2) * 8 + 4 + (e & 3)] = gc.i;
  }
  if ((int)threadIdx.x >= BLOCK - 64 && (int)threadIdx.x < BLOCK - 64 + 4 * C) {   // input mixing weights (the last wave: idle above for N <= 48)
    const int k = (threadIdx.x - (BLOCK - 64)) / C, c = (threadIdx.x - (BLOCK - 64)) - k * C;
    double acc = 0.0;
#pragma unroll 6
    for (int n = 0; n < N; ++n) acc += gv_l[(n * C + c) * 8 + k];
    row[k * C + c] = acc;
  }
  __syncthreads();
  STAMP(13);
  for (int e = threadIdx.x; e < Tin * 4; e += BLOCK) {    // g_lat_v[t][m] = sum_n G_cart[n][m] conj(Wg1[n][t])
    const int t = e >> 2, m = e & 3;
    cx<double> acc = {0, 0};
#pragma unroll 6
    for (int n = 0; n < N; ++n)
      cfmac(acc, cx<double>{gcart[n * 8 + m], gcart[n * 8 + 4 + m]}, cx<double>{wgl[n * Tin + t], wgl[N * Tin + n * Tin + t]});
    g_lat_v[((size_t)b * Tin + t) * 4 + m] = acc.r;
    g_lat_v[(((size_t)B + b) * Tin + t) * 4 + m] = acc.i;
    if constexpr (TO_LDS) { g_lat_l[t * 8 + m] = acc.r; g_lat_l[t * 8 + 4 + m] = acc.i; }
  }
  for (int e = threadIdx.x; e < N * Tin; e += BLOCK) {    // dWg1[n][t] = sum_m G_cart[n][m] conj(lat[t][m])
    const int n = e / Tin, t = e - n * Tin;
    cx<double> acc = {0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m)
      cfmac(acc, cx<double>{gcart[n * 8 + m], gcart[n * 8 + 4 + m]}, cx<double>{latl[t * 8 + m], latl[t * 8 + 4 + m]});
    row[4 * C + e] = acc.r;
    row[4 * C + N * Tin + e] = acc.i;
  }
}
__global__ __launch_bounds__(BLOCK) void dec_input_bwd_kernel(int B, int N, int C, int Tin, const double* __restrict__ lat_v,
                                                             const double* __restrict__ wg1, const double* __restrict__ w1,
                                                             const double* __restrict__ pdec, const double* __restrict__ g_p,
                                                             const double* __restrict__ g_s0, const double* __restrict__ g_v0,
                                                             double* g_lat_v, double* part) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* lds = reinterpret_cast<double*>(smem_raw);
  DecInBwdStage st;
  st.issue(B, N, C, Tin, lat_v, wg1, w1, pdec, g_p, g_s0, g_v0);
  st.commit(lds);
  __syncthreads();
  dec_input_bwd_body<false>(B, N, C, Tin, g_lat_v, part, lds, nullptr);
}

// ============================================================================================
// encoder -> decoder junction, one launch per direction: the decoder input of a jet needs only that jet's latent
// vectors (and vice versa for the gradients), so the two per-jet kernels run back to back in the same workgroup --
// both stages' global operands are fetched together up front, the latent vectors (gradients) pass through LDS.
// ============================================================================================
// overlap = 0 (large jets: the two stages' LDS blocks would not fit side by side): the stages share LDS and fetch in turn.
__global__ __launch_bounds__(BLOCK) void junction_fwd_kernel(int B, int N, int CL, int Ts, int Tv, const double* s, const double* v,
                                                            const double* wl0, const double* wl1, double* lat_s, double* lat_v,
                                                            int* idx, int C0, const double* wg1, const double* w0, const double* w1,
                                                            double* pdec, double* s0, double* v0, int overlap, int pool) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* lat_lds = reinterpret_cast<double*>(smem_raw);
  // overlap: behind the encoder stage's block; else over its sv / weight regions, which are dead once y is complete
  double* dec_lds = lat_lds + (overlap ? lat_fwd_doubles(N, CL, Ts, Tv, pool) : lat_y_doubles(N, Ts, Tv, pool));
  const int Tin = pool_blocks(pool) * Tv;
  STAMP(0);
  LatentStage ls;
  DecInFwdStage ds;
  ls.issue(B, N, CL, Ts, Tv, s, v, wl0, wl1, pool);
  if (overlap) ds.issue(B, N, C0, Tin, nullptr, wg1, w0, w1, false);
  ls.commit(lat_lds, false);
  if (overlap) ds.commit(dec_lds);
  __syncthreads();
  // 'mix' reads the node features and weights WHILE it writes the latent vectors: when the stages share LDS they go to a spare
  // row behind both blocks first (the pooled maps write them after their last read of anything the decoder block overlays)
  const bool spare = pool_is_mix(pool) && !overlap;
  double* lat_l = dec_lds + 2 * N * Tin;
  if (spare) {
    const size_t nl = lat_fwd_doubles(N, CL, Ts, Tv, pool), nyd = lat_y_doubles(N, Ts, Tv, pool) + dec_in_fwd_doubles(N, C0, Tin);
    lat_l = lat_lds + (nl > nyd ? nl : nyd);
  }
  if (pool_is_mix(pool)) enc_latent_mix_fwd_body<true>(B, N, CL, Ts, Tv, lat_s, lat_v, lat_lds, lat_l);
  else enc_latent_fwd_body<true>(B, N, CL, Ts, Tv, pool, lat_s, lat_v, idx, lat_lds, lat_l);
  if (!overlap) {
    if (spare) __syncthreads();                          // every thread is done with the encoder stage's operands
    ds.issue(B, N, C0, Tin, nullptr, wg1, w0, w1, false);
    ds.commit(dec_lds);
    if (spare)
      for (int e = threadIdx.x; e < Tin * 8; e += BLOCK) dec_lds[2 * N * Tin + e] = lat_l[e];
  }
  __syncthreads();                                       // the latent vectors of the jet are in the decoder stage's LDS block
  dec_input_fwd_body(B, N, C0, Tin, pdec, s0, v0, dec_lds);
  STAMP(5);
}
__global__ __launch_bounds__(BLOCK) void junction_bwd_kernel(int B, int N, int C0, int Tin, const double* lat_v, const double* wg1,
                                                            const double* w1, const double* pdec, const double* g_p,
                                                            const double* g_s0, const double* g_v0, double* g_lat_v, double* part_dec,
                                                            int CL, int Ts, int Tv, const double* s, const double* v,
                                                            const double* wl0, const double* wl1, const double* g_lat_s,
                                                            const int* idx, double* g_s, double* g_v, double* part_enc, int overlap,
                                                            int pool) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* dec_lds = reinterpret_cast<double*>(smem_raw);
  const size_t nd = dec_in_bwd_doubles(N, C0, Tin), nl = lat_bwd_doubles(N, CL, Ts, Tv, pool);
  double* g_lat_l = dec_lds + (overlap ? nd : (nd > nl ? nd : nl));      // [Tin][8], outside both stages' blocks
  double* lat_lds = overlap ? g_lat_l + Tin * 8 : dec_lds;
  STAMP(10);
  DecInBwdStage ds;
  LatentStage ls;
  LatentBwdPrefetch pre;
  ds.issue(B, N, C0, Tin, lat_v, wg1, w1, pdec, g_p, g_s0, g_v0);
  pre.issue(B, Ts, Tv, pool, idx, g_lat_s);
  if (overlap) ls.issue(B, N, CL, Ts, Tv, s, v, wl0, wl1, pool);
  ds.commit(dec_lds);
  if (overlap) ls.commit(lat_lds, true);
  __syncthreads();
  dec_input_bwd_body<true>(B, N, C0, Tin, g_lat_v, part_dec, dec_lds, g_lat_l);
  __syncthreads();                                       // this jet's latent-vector gradient is in LDS; the decoder stage's block is free
  if (!overlap) {
    ls.issue(B, N, CL, Ts, Tv, s, v, wl0, wl1, pool);
    ls.commit(lat_lds, true);
    __syncthreads();
  }
  if (pool_is_mix(pool)) enc_latent_mix_bwd_body<true>(B, N, CL, Ts, Tv, g_lat_s, nullptr, g_lat_l, g_s, g_v, part_enc, lat_lds);
  else enc_latent_bwd_body<true>(B, N, CL, Ts, Tv, pool, g_lat_s, nullptr, g_lat_l, idx, g_s, g_v, part_enc, lat_lds, pre);
  STAMP(19);
}

// ============================================================================================
// decoder output + get_real('sum') + Chamfer loss, forward and backward in one pass per jet
//   recon [2][B][N][4]; loss_part [B]; g_v [2][B][N][C][4]; part row per jet: dWo1 [2][C]
// LDS: x [N][4] | tg [N][4] | rmin [N] | cmin [N] | gx [N][4] | ycl [N][8] | vl [N*C][8] | tmp [N*C][2] | wol [2C] | rarg, carg [N] ints
// ============================================================================================
__global__ __launch_bounds__(BLOCK) void dec_output_loss_kernel(int B, int N, int C, const double* __restrict__ v,
                                                               const double* __restrict__ wo1, const double* __restrict__ target,
                                                               double loss_scale, double* recon, double* loss_part, double* g_v,
                                                               double* part) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  dec_output_loss_body(B, N, C, v, wo1, target, loss_scale, recon, loss_part, g_v, part, smem_raw);
}

// ============================================================================================
// decoder output alone (module API: the loss is the caller's): mix_to_output on the (1,1) irrep + rep_to_p
// (lgn_decoder.py:286-295) and its backward from an arbitrary upstream gradient g_recon [2][B][N][4].
// ============================================================================================
__global__ __launch_bounds__(BLOCK) void dec_output_fwd_kernel(int B, int N, int C, const double* __restrict__ v,
                                                              const double* __restrict__ wo1, double* recon) {
  const size_t plp = (size_t)B * N * 4, pl = (size_t)B * N * C;
  for (size_t node = (size_t)blockIdx.x * BLOCK + threadIdx.x; node < (size_t)B * N; node += (size_t)gridDim.x * BLOCK) {
    cx<double> yc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, pc[4];
    for (int c = 0; c < C; ++c) {
      const cx<double> w = {wo1[c], wo1[C + c]};
      const size_t e = node * C + c;
#pragma unroll
      for (int m = 0; m < 4; ++m) cfma(yc[m], w, cx<double>{v[e * 4 + m], v[pl * 4 + e * 4 + m]});
    }
    cart_from_canon(yc, pc);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      recon[node * 4 + m] = pc[m].r;
      recon[plp + node * 4 + m] = pc[m].i;
    }
  }
}
// one workgroup per jet; g_v [2][B][N][C][4]; part row per jet: dWo1 [2][C]
// ============================================================================================
// Chamfer loss on its own (module API; the whole step has it inside dec_output_loss): ChamferLoss.forward of
// utils/losses/chamfer_loss/chamfer_loss.py:16-31 with cdist = sum over the 4 components of (x_i - y_j)^2 (distance_sq.py:263-304,
// even p: no eps).  One workgroup per jet computes the jet's term
//     (sum_i min_j d_ij + sum_j min_i d_ij) / 2   [+ sum_mu (sum_i x_i - sum_j y_j)_mu^2 / (4 B): the jet_features MSE term]
// and, in the same pass, its gradient w.r.t. x and y (first minimum on ties, as torch.min); autograd scales them by the upstream
// scalar.  LDS: x [N][4] | y [M][4] | rarg [N] | carg [M] (ints) | red [2 * BLOCK / 64 + 8]
// ============================================================================================
__global__ __launch_bounds__(BLOCK) void chamfer_kernel(int B, int N, int M, const double* __restrict__ x, const double* __restrict__ y,
                                                        int jet_features, double* loss_part, double* gx, double* gy) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* xl = reinterpret_cast<double*>(smem_raw);
  double* yl = xl + N * 4;
  double* red = yl + M * 4;                             // [BLOCK / 64] wave sums, then [8] jet sums
  int* rarg = reinterpret_cast<int*>(red + BLOCK / 64 + 8);
  int* carg = rarg + N;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int e = tid; e < N * 4; e += BLOCK) xl[e] = x[(size_t)b * N * 4 + e];
  for (int e = tid; e < M * 4; e += BLOCK) yl[e] = y[(size_t)b * M * 4 + e];
  __syncthreads();
  auto dist = [&](int i, int j) {
    const double d0 = xl[i * 4] - yl[j * 4], d1 = xl[i * 4 + 1] - yl[j * 4 + 1], d2 = xl[i * 4 + 2] - yl[j * 4 + 2],
                 d3 = xl[i * 4 + 3] - yl[j * 4 + 3];
    return ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;   // torch.sum over the last axis of 4: sequential
  };
  double mine = 0.0;
  for (int e = tid; e < N + M; e += BLOCK) {
    double best = 0.0;
    int arg = 0;
    if (e < N) {
      for (int j = 0; j < M; ++j) { const double d = dist(e, j); if (j == 0 || d < best) { best = d; arg = j; } }
      rarg[e] = arg;
    } else {
      const int j = e - N;
      for (int i = 0; i < N; ++i) { const double d = dist(i, j); if (i == 0 || d < best) { best = d; arg = i; } }
      carg[j] = arg;
    }
    mine += 0.5 * best;
  }
  if (jet_features && tid < 8) {                         // jet sums of x (tid 0..3) and y (4..7), component tid & 3
    const int m = tid & 3;
    double s = 0.0;
    if (tid < 4) for (int i = 0; i < N; ++i) s += xl[i * 4 + m];
    else for (int j = 0; j < M; ++j) s += yl[j * 4 + m];
    red[BLOCK / 64 + tid] = s;
  }
  mine = group_sum<64>(mine);
  if ((tid & 63) == 0) red[tid >> 6] = mine;
  __syncthreads();
  const double* js = red + BLOCK / 64;
  const double jscale = 1.0 / (4.0 * (double)B);         // nn.MSELoss(): mean over the (B, 4) jet momenta
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < BLOCK / 64; ++w) s += red[w];
    if (jet_features)
      for (int m = 0; m < 4; ++m) { const double d = js[m] - js[4 + m]; s += d * d * jscale; }
    loss_part[b] = s;
  }
  // gradients: row-minimum term of the own particle + every column (row) minimum that picked it, in index order
  for (int e = tid; e < (N + M) * 4; e += BLOCK) {
    const int p = e >> 2, m = e & 3;
    double g = 0.0;
    if (p < N) {
      const double xi = xl[p * 4 + m];
      g = xi - yl[rarg[p] * 4 + m];
      for (int j = 0; j < M; ++j) if (carg[j] == p) g += xi - yl[j * 4 + m];
      if (jet_features) g += 2.0 * jscale * (js[m] - js[4 + m]);
      gx[((size_t)b * N + p) * 4 + m] = g;
    } else {
      const int j = p - N;
      const double yj = yl[j * 4 + m];
      g = yj - xl[carg[j] * 4 + m];
      for (int i = 0; i < N; ++i) if (rarg[i] == j) g += yj - xl[i * 4 + m];
      if (jet_features) g -= 2.0 * jscale * (js[m] - js[4 + m]);
      gy[((size_t)b * M + j) * 4 + m] = g;
    }
  }
}

__global__ __launch_bounds__(BLOCK) void dec_output_bwd_kernel(int B, int N, int C, const double* __restrict__ v,
                                                              const double* __restrict__ wo1, const double* __restrict__ g_recon,
                                                              double* g_v, double* part) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* gc_l = reinterpret_cast<double*>(smem_raw);    // [N][8] gradient w.r.t. the canonical output (re[4] | im[4])
  double* tmp = gc_l + (size_t)N * 8;                    // [N*C][2]
  const int b = blockIdx.x;
  const size_t plp = (size_t)B * N * 4, pl = (size_t)B * N * C;
  for (int n = threadIdx.x; n < N; n += BLOCK) {
    const size_t node = (size_t)b * N + n;
    cx<double> g[4], gc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) g[m] = {g_recon[node * 4 + m], g_recon[plp + node * 4 + m]};
    cart_from_canon_bwd(g, gc);
#pragma unroll
    for (int m = 0; m < 4; ++m) { gc_l[n * 8 + m] = gc[m].r; gc_l[n * 8 + 4 + m] = gc[m].i; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * C; e += BLOCK) {
    const int n = e / C, c = e - n * C;
    const cx<double> w = {wo1[c], wo1[C + c]};
    const size_t base = (size_t)b * N * C + e;
    cx<double> d = {0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> gc = {gc_l[n * 8 + m], gc_l[n * 8 + 4 + m]};
      cx<double> r = cmulc(gc, w);
      g_v[base * 4 + m] = r.r;
      g_v[pl * 4 + base * 4 + m] = r.i;
      cfmac(d, gc, cx<double>{v[base * 4 + m], v[pl * 4 + base * 4 + m]});
    }
    tmp[e * 2] = d.r;
    tmp[e * 2 + 1] = d.i;
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * C) {
    const int k = threadIdx.x / C, c = threadIdx.x - k * C;
    double acc = 0.0;
    for (int n = 0; n < N; ++n) acc += tmp[(n * C + c) * 2 + k];
    part[(size_t)b * 2 * C + k * C + c] = acc;
  }
}

// ============================================================================================
// L1 regularisation + loss assembly, Adam
// ============================================================================================
// g += lambda * sign(w); Adam update (torch.optim.Adam defaults: no weight decay, no amsgrad).  The same pass adds up
// |w| of the weights BEFORE the update (the L1 term of this step's loss) into one partial per workgroup.
// The optimiser step counter lives on the device (graph replays stay correct): this step is number t = *step_dev + 1.
// The LAST workgroup to finish (a counter in the scratch block, bumped behind a device-scope fence) also assembles the loss
// -- loss_out[0] = chamfer + lambda * sum|w|, loss_out[1] = chamfer, loss_out[2] = sum|w| -- from the per-jet terms and the
// per-workgroup |w| sums IN INDEX ORDER (the result does not depend on which workgroup that was), bumps the device-side step
// counter (every workgroup read it at its start) and clears the counter for the next replay: one launch instead of two.
// Round 5 (the kernel took 11.7 us for 63 k parameters at ANY batch): the bias corrections 1 - beta^t were two fp64 pow() per
// THREAD; now one thread per workgroup takes beta1^t, beta2^t from the scratch block, where the previous step's last workgroup
// left them (slot t & 1 = {t, beta1^t, beta2^t}: written during step t - 1, read during step t, so no workgroup of a launch can
// see its own launch's update), and falls back to pow() when the slot does not name this step (first step, restored counter);
// four parameters per thread, a quarter of the workgroups at the counter.
constexpr int ADAM_PER_THREAD = 4;
__global__ __launch_bounds__(BLOCK) void l1_adam_kernel(long n, double* w, double* g, double* m, double* v, double lambda, double lr,
                                                       double beta1, double beta2, double eps, long* step_dev, int do_adam,
                                                       double* l1_part, double* powers, unsigned long long* done,
                                                       const double* __restrict__ loss_part, int nB, double* loss_out) {
  __shared__ double red[4];
  __shared__ double bc[4];
  __shared__ int last;
  if (threadIdx.x == 0) {
    bc[0] = bc[1] = bc[2] = bc[3] = 1.0;
    if (do_adam) {
      // counter and BOTH slots in one memory round trip (counter, then its slot, was two -- with every other thread at the barrier)
      double pw[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) pw[q] = powers[q];
      const long ti = *step_dev + 1;
      const double t = (double)ti;
      const int odd = (int)(ti & 1);
      const double sl0 = odd ? pw[3] : pw[0], sl1 = odd ? pw[4] : pw[1], sl2 = odd ? pw[5] : pw[2];
      double p1, p2;
      if (sl0 == t) { p1 = sl1; p2 = sl2; }
      else { p1 = pow(beta1, t); p2 = pow(beta2, t); }
      bc[0] = 1.0 - p1;
      bc[1] = sqrt(1.0 - p2);
      bc[2] = p1;
      bc[3] = p2;
    }
  }
  __syncthreads();
  const double bc1 = bc[0], bc2_sqrt = bc[1];
  double l1 = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double wi = w[i];
    l1 += fabs(wi);
    const AdamOut o = l1_adam_one(wi, g[i], do_adam ? m[i] : 0.0, do_adam ? v[i] : 0.0, lambda, lr, beta1, beta2, eps, bc1, bc2_sqrt);
    g[i] = o.g;
    if (do_adam) {
      m[i] = o.m;
      v[i] = o.v;
      w[i] = o.w;
    }
  }
  l1 = block_sum(l1, red);
  if (threadIdx.x == 0) {
    // The partial goes out as a device-scope atomic exchange (performed at the coherent level; its return is awaited, so it is
    // complete before the count below is issued) and the last workgroup reads the partials back with device-scope atomic loads:
    // no __threadfence() -- on this multi-XCD part a device-scope release / acquire pair writes back and invalidates the whole L2 of
    // the XCD, 3.5 of this kernel's 9 us.
    unsigned long long old = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(l1_part) + blockIdx.x,
                                                   (unsigned long long)__double_as_longlong(l1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) : : "memory");       // the exchange has been performed before the count is issued
    last = __hip_atomic_fetch_add(done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  double a = 0, l = 0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += BLOCK) {
    a += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(l1_part) + i, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
    // the slots go back to zero: step_tail.hip shares them and reads "zero = not yet written in this launch" (lgn_amd.h: the scratch
    // block is zero between calls, whichever of the two kernels ran last)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(l1_part) + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int i = threadIdx.x; i < nB; i += BLOCK) l += loss_part[i];
  a = block_sum(a, red);
  l = block_sum(l, red);
  if (threadIdx.x == 0) {
    loss_out[0] = l + lambda * a;
    loss_out[1] = l;
    loss_out[2] = a;
    if (do_adam) {
      const long tn = *step_dev + 2;                       // the NEXT step's number and powers, into the slot it will read
      double* slot = powers + 3 * (tn & 1);
      slot[1] = bc[2] * beta1;
      slot[2] = bc[3] * beta2;
      slot[0] = (double)tn;
      *step_dev += 1;
    }
    *done = 0ull;
  }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
#define LGN_LDS_LAUNCH(kernel, what, smem)                                                                            \
  LGN_CHECK_ARG((smem) <= 160 * 1024, what ": needs %zu B of LDS", (size_t)(smem));                                    \
  if ((smem) > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(smem))
static int grid_for(size_t total) {
  size_t g = (total + BLOCK - 1) / BLOCK;
  return (int)(g < 2048 ? (g ? g : 1) : 2048);
}

int enc_input_fwd(int B, int N, int C, int K, const double* p4, const double* xs, const double* w0, const double* w1, double* s, double* v,
                  hipStream_t st, double* z1, size_t n1, double* z2, size_t n2) {
  LGN_CHECK_ARG(K >= 1 && (K == 1 || xs), "enc_input: %d input scalars need the extra-scalar array", K);
  hipLaunchKernelGGL(enc_input_fwd_kernel, dim3(grid_for((size_t)B * N * C)), dim3(BLOCK), 0, st, B, N, C, K, p4, xs, w0, w1, s, v, z1,
                     n1, z2, n2);
  LGN_CHECK_LAUNCH();
  return 0;
}
int enc_input_bwd(int B, int N, int C, int K, const double* p4, const double* xs, const double* g_s, const double* g_v, double* part,
                  hipStream_t st) {
  LGN_CHECK_ARG(K >= 1 && (K == 1 || xs), "enc_input: %d input scalars need the extra-scalar array", K);
  const size_t smem = sizeof(double) * N * C * (2 * K + 2);
  LGN_LDS_LAUNCH(enc_input_bwd_kernel, "enc_input_bwd", smem);
  hipLaunchKernelGGL(enc_input_bwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, K, p4, xs, g_s, g_v, part);
  LGN_CHECK_LAUNCH();
  return 0;
}
int enc_latent_fwd(int B, int N, int C, int Ts, int Tv, int pool, const double* s, const double* v, const double* wl0, const double* wl1,
                   double* lat_s, double* lat_v, int* idx, hipStream_t st) {
  LGN_CHECK_ARG(pool_valid(pool), "enc_latent_fwd: bad latent pooling code %d", pool);
  const size_t smem = sizeof(double) * lat_fwd_doubles(N, C, Ts, Tv, pool_canon(pool));
  LGN_LDS_LAUNCH(enc_latent_fwd_kernel, "enc_latent_fwd", smem);
  hipLaunchKernelGGL(enc_latent_fwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, Ts, Tv, pool_canon(pool), s, v, wl0, wl1, lat_s, lat_v, idx);
  LGN_CHECK_LAUNCH();
  return 0;
}
int enc_latent_bwd(int B, int N, int C, int Ts, int Tv, int pool, const double* s, const double* v, const double* wl0, const double* wl1,
                   const double* g_lat_s, const double* g_lat_v, const int* idx, double* g_s, double* g_v, double* part,
                   hipStream_t st) {
  LGN_CHECK_ARG(pool_valid(pool), "enc_latent_bwd: bad latent pooling code %d", pool);
  const size_t smem = sizeof(double) * lat_bwd_doubles(N, C, Ts, Tv, pool_canon(pool));
  LGN_LDS_LAUNCH(enc_latent_bwd_kernel, "enc_latent_bwd", smem);
  hipLaunchKernelGGL(enc_latent_bwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, Ts, Tv, pool_canon(pool), s, v, wl0, wl1, g_lat_s, g_lat_v, idx,
                     g_s, g_v, part);
  LGN_CHECK_LAUNCH();
  return 0;
}
int dec_input_fwd(int B, int N, int C, int Tin, const double* lat_v, const double* wg1, const double* w0, const double* w1,
                  double* pdec, double* s0, double* v0, hipStream_t st) {
  const size_t smem = sizeof(double) * dec_in_fwd_doubles(N, C, Tin);
  LGN_LDS_LAUNCH(dec_input_fwd_kernel, "dec_input_fwd", smem);
  hipLaunchKernelGGL(dec_input_fwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, Tin, lat_v, wg1, w0, w1, pdec, s0, v0);
  LGN_CHECK_LAUNCH();
  return 0;
}
int dec_input_bwd(int B, int N, int C, int Tin, const double* lat_v, const double* wg1, const double* w1, const double* pdec,
                  const double* g_p, const double* g_s0, const double* g_v0, double* g_lat_v, double* part, hipStream_t st) {
  const size_t smem = sizeof(double) * dec_in_bwd_doubles(N, C, Tin);
  LGN_LDS_LAUNCH(dec_input_bwd_kernel, "dec_input_bwd", smem);
  hipLaunchKernelGGL(dec_input_bwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, Tin, lat_v, wg1, w1, pdec, g_p,
                     g_s0, g_v0, g_lat_v, part);
  LGN_CHECK_LAUNCH();
  return 0;
}
int dec_output_loss(int B, int N, int C, const double* v, const double* wo1, const double* target, double loss_scale, double* recon,
                    double* loss_part, double* g_v, double* part, hipStream_t st) {
  const size_t smem = dec_out_loss_bytes(N, C);
  LGN_LDS_LAUNCH(dec_output_loss_kernel, "dec_output_loss", smem);
  hipLaunchKernelGGL(dec_output_loss_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, v, wo1, target, loss_scale, recon, loss_part,
                     g_v, part);
  LGN_CHECK_LAUNCH();
  return 0;
}
int dec_output_fwd(int B, int N, int C, const double* v, const double* wo1, double* recon, hipStream_t st) {
  hipLaunchKernelGGL(dec_output_fwd_kernel, dim3(grid_for((size_t)B * N)), dim3(BLOCK), 0, st, B, N, C, v, wo1, recon);
  LGN_CHECK_LAUNCH();
  return 0;
}
int chamfer_fwd(int B, int N, int M, const double* x, const double* y, int jet_features, double* loss_part, double* gx, double* gy,
                hipStream_t st) {
  const size_t smem = sizeof(double) * ((size_t)(N + M) * 4 + BLOCK / 64 + 8) + sizeof(int) * (size_t)(N + M);
  LGN_LDS_LAUNCH(chamfer_kernel, "chamfer", smem);
  hipLaunchKernelGGL(chamfer_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, M, x, y, jet_features, loss_part, gx, gy);
  LGN_CHECK_LAUNCH();
  return 0;
}
// Plan-time fit queries (lgn_encoder_end_lds_bytes / lgn_decoder_end_lds_bytes / lgn_junction_lds_bytes): the LDS the per-jet end
// stages of a network need, from the same expressions as the launchers above -- so that the callers choose the per-operator path
// BEFORE the first launch instead of failing at it (a jet's latent stage is one workgroup: N C (Ts + Tv) grows past 160 KiB for
// 150-particle jets with three pooled blocks).
size_t encoder_end_lds_bytes(int N, int C0, int K, int CL, int Ts, int Tv, int pool) {
  pool = pool_canon(pool);
  size_t m = sizeof(double) * (size_t)N * C0 * (2 * K + 2);                                   // enc_input_bwd
  const size_t f = sizeof(double) * lat_fwd_doubles(N, CL, Ts, Tv, pool), b = sizeof(double) * lat_bwd_doubles(N, CL, Ts, Tv, pool);
  m = f > m ? f : m;
  return b > m ? b : m;
}
size_t decoder_end_lds_bytes(int N, int C0, int Tin, int CL) {
  size_t m = sizeof(double) * dec_in_fwd_doubles(N, C0, Tin);
  const size_t b = sizeof(double) * dec_in_bwd_doubles(N, C0, Tin), o = dec_out_loss_bytes(N, CL),
               ob = sizeof(double) * ((size_t)N * 8 + (size_t)N * CL * 2);
  m = b > m ? b : m;
  m = o > m ? o : m;
  return ob > m ? ob : m;
}
size_t junction_lds_bytes(int N, int CL, int Ts, int Tv, int pool, int C0) {
  pool = pool_canon(pool);
  const int Tin = pool_blocks(pool) * Tv;
  const size_t nl = lat_fwd_doubles(N, CL, Ts, Tv, pool), nd = dec_in_fwd_doubles(N, C0, Tin), ny = lat_y_doubles(N, Ts, Tv, pool);
  const bool ov = sizeof(double) * (nl + nd) <= 64 * 1024;
  const size_t fwd = sizeof(double) * (ov ? nl + nd : ny + (nl - ny > nd ? nl - ny : nd) + (pool_is_mix(pool) ? (size_t)Tin * 8 : 0));
  const size_t bd = dec_in_bwd_doubles(N, C0, Tin), bl = lat_bwd_doubles(N, CL, Ts, Tv, pool);
  const bool ovb = sizeof(double) * (bd + bl + (size_t)Tin * 8) <= 64 * 1024;
  const size_t bwd = sizeof(double) * ((ovb ? bd + bl : (bd > bl ? bd : bl)) + (size_t)Tin * 8);
  return fwd > bwd ? fwd : bwd;
}

int dec_output_bwd(int B, int N, int C, const double* v, const double* wo1, const double* g_recon, double* g_v, double* part,
                   hipStream_t st) {
  const size_t smem = sizeof(double) * ((size_t)N * 8 + (size_t)N * C * 2);
  LGN_CHECK_ARG(smem <= 160 * 1024, "dec_output_bwd: needs %zu B of LDS", smem);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_output_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(dec_output_bwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, v, wo1, g_recon, g_v, part);
  LGN_CHECK_LAUNCH();
  return 0;
}
// The two stages of a junction keep their LDS blocks side by side (all fetches up front) while that stays within 64 KB per
// workgroup; larger jets run them in turn on shared LDS.
int junction_fwd(int B, int N, int CL, int Ts, int Tv, int pool, const double* s, const double* v, const double* wl0, const double* wl1,
                 double* lat_s, double* lat_v, int* idx, int C0, const double* wg1, const double* w0, const double* w1, double* pdec,
                 double* s0, double* v0, hipStream_t st) {
  LGN_CHECK_ARG(pool_valid(pool), "junction_fwd: bad latent pooling code %d", pool);
  pool = pool_canon(pool);
  const size_t nl = lat_fwd_doubles(N, CL, Ts, Tv, pool), nd = dec_in_fwd_doubles(N, C0, pool_blocks(pool) * Tv), ny = lat_y_doubles(N, Ts, Tv, pool);
  const int overlap = sizeof(double) * (nl + nd) <= 64 * 1024;
  const size_t smem = sizeof(double) * (overlap ? nl + nd : ny + (nl - ny > nd ? nl - ny : nd) +
                                                            (pool_is_mix(pool) ? (size_t)pool_blocks(pool) * Tv * 8 : 0));
  LGN_LDS_LAUNCH(junction_fwd_kernel, "junction_fwd", smem);
  hipLaunchKernelGGL(junction_fwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, CL, Ts, Tv, s, v, wl0, wl1, lat_s, lat_v, idx, C0, wg1,
                     w0, w1, pdec, s0, v0, overlap, pool);
  LGN_CHECK_LAUNCH();
  return 0;
}
int junction_bwd(int B, int N, int C0, int Tin, const double* lat_v, const double* wg1, const double* w1, const double* pdec,
                 const double* g_p, const double* g_s0, const double* g_v0, double* g_lat_v, double* part_dec, int CL, int Ts, int Tv,
                 int pool, const double* s, const double* v, const double* wl0, const double* wl1, const double* g_lat_s, const int* idx,
                 double* g_s, double* g_v, double* part_enc, hipStream_t st) {
  LGN_CHECK_ARG(pool_valid(pool), "junction_bwd: bad latent pooling code %d", pool);
  pool = pool_canon(pool);
  LGN_CHECK_ARG(Tin == pool_blocks(pool) * Tv, "junction_bwd: the decoder takes the %d pooled latent vectors, got Tin = %d",
                pool_blocks(pool) * Tv, Tin);
  const size_t nd = dec_in_bwd_doubles(N, C0, Tin), nl = lat_bwd_doubles(N, CL, Ts, Tv, pool);
  const int overlap = sizeof(double) * (nd + nl + (size_t)Tin * 8) <= 64 * 1024;
  const size_t smem = sizeof(double) * ((overlap ? nd + nl : (nd > nl ? nd : nl)) + (size_t)Tin * 8);
  LGN_LDS_LAUNCH(junction_bwd_kernel, "junction_bwd", smem);
  hipLaunchKernelGGL(junction_bwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C0, Tin, lat_v, wg1, w1, pdec, g_p, g_s0, g_v0, g_lat_v,
                     part_dec, CL, Ts, Tv, s, v, wl0, wl1, g_lat_s, idx, g_s, g_v, part_enc, overlap, pool);
  LGN_CHECK_LAUNCH();
  return 0;
}

// loss_out: 3 results followed by LGN_FINALIZE_SCRATCH doubles of scratch (per-workgroup |w| partials)
int finalize_step(double* w, double* g, long n, const double* loss_part, int nB, double lambda, double* m, double* v, long* step_dev,
                  double lr, double beta1, double beta2, double eps, int do_adam, double* loss_out, hipStream_t st) {
  // scratch behind the 3 results: [0, nblk) per-workgroup |w| sums; the last slot = the finished-workgroup counter (zero between
  // calls: the caller allocates the block zero-filled, the kernel clears it again -- a launch that faulted part-way leaves it dirty:
  // re-zero the block before reusing it); the six slots before it = {t, beta1^t, beta2^t} for the next odd / even step
  int nblk = grid_for(((size_t)n + ADAM_PER_THREAD - 1) / ADAM_PER_THREAD);
  if (nblk > LGN_FINALIZE_SCRATCH - 12) nblk = LGN_FINALIZE_SCRATCH - 12;     // (slots -11 .. -8: the level counters of step_tail.hip)
  double* l1_part = loss_out + 3;
  double* powers = loss_out + 3 + LGN_FINALIZE_SCRATCH - 7;
  unsigned long long* done = reinterpret_cast<unsigned long long*>(loss_out + 3 + LGN_FINALIZE_SCRATCH - 1);
  hipLaunchKernelGGL(l1_adam_kernel, dim3(nblk), dim3(BLOCK), 0, st, n, w, g, m, v, lambda, lr, beta1, beta2, eps, step_dev, do_adam,
                     l1_part, powers, done, loss_part, nB, loss_out);
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace lgn
