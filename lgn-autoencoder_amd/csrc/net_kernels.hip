// lgn-autoencoder_amd/csrc/net_kernels.hip -- the O(N)-per-jet ends of the networks, fp64:
//   encoder input   (lgn_encoder.py:284-298,376)   mass + canonical momenta + input_func_node MixReps
//   encoder latent  (lgn_encoder.py:322-331,419-583) mix_reps MixReps + rep_to_p + 'min&max' pooling
//   decoder input   (lgn_decoder.py:305-345,257-265) latent_to_graph + p_cplx_to_rep + input_func_node
//   decoder output  (lgn_decoder.py:286-295) + get_real('sum') (utils/utils.py:194-207)
//                   + ChamferLoss (utils/losses/chamfer_loss/chamfer_loss.py:17-23) forward AND backward
//   L1 + Adam       (utils/train.py:484-487, utils/initialize.py:156-158)
// Each forward/backward pair is one workgroup per jet; weight-gradient partials are one row per jet.
#include "net.hpp"

namespace lgn {

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
// gradient of cart_from_canon: G_c1 = h G_px - i h G_py, G_c3 = -h G_px - i h G_py
__device__ __forceinline__ void cart_from_canon_bwd(const cx<double> (&g)[4], cx<double> (&gc)[4]) {
  gc[0] = g[0];
  gc[1] = {(g[1].r + g[2].i) * H, (g[1].i - g[2].r) * H};
  gc[2] = g[3];
  gc[3] = {(-g[1].r + g[2].i) * H, (-g[1].i - g[2].r) * H};
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
}  // namespace

// ============================================================================================
// encoder input
// ============================================================================================
// z1/z2: two buffers the step wants zeroed before anything later in the stream touches them (the flat gradient buffer and
// the zero-initialised workspace block); folded into this first kernel instead of two memset launches.
__global__ void enc_input_fwd_kernel(int B, int N, int C, const double* __restrict__ p4, const double* __restrict__ w0,
                                     const double* __restrict__ w1, double* s, double* v, double* z1, size_t n1, double* z2,
                                     size_t n2) {
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
    s[e] = w0[c] * mass;                  // W00[c] * (mass + 0i)
    s[pl + e] = w0[C + c] * mass;
    const cx<double> w = {w1[c], w1[C + c]};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      cx<double> r = cmul(w, q[m]);
      v[e * 4 + m] = r.r;
      v[pl * 4 + e * 4 + m] = r.i;
    }
  }
}

// partial rows [nblk][4C]: dW00 (re[C], im[C]) then dW11 (re[C], im[C]).  One workgroup per jet:
// thread = (node, channel) writes its four terms to LDS, thread = (term, channel) adds them up in node order.
__global__ __launch_bounds__(BLOCK) void enc_input_bwd_kernel(int B, int N, int C, const double* __restrict__ p4,
                                                             const double* __restrict__ g_s, const double* __restrict__ g_v,
                                                             double* part) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* tmp = reinterpret_cast<double*>(smem_raw);    // [N*C][4]
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
    tmp[i * 4 + 0] = g_s[e] * mass;
    tmp[i * 4 + 1] = g_s[pl + e] * mass;
    tmp[i * 4 + 2] = d1.r;
    tmp[i * 4 + 3] = d1.i;
  }
  __syncthreads();
  if ((int)threadIdx.x < 4 * C) {
    const int k = threadIdx.x / C, c = threadIdx.x - k * C;
    double acc = 0.0;
    for (int n = 0; n < N; ++n) acc += tmp[(n * C + c) * 4 + k];
    part[(size_t)b * 4 * C + k * C + c] = acc;
  }
}

// ============================================================================================
// encoder latent: MixReps -> Cartesian -> min&max pooling
//   lat_s [2][B][2Ts]  (min block, max block), lat_v [2][B][2Tv][4], idx [B][2][Ts+Tv][2] (plane, channel, min/max)
// ============================================================================================
__device__ __forceinline__ void enc_latent_fwd_body(int B, int N, int C, int Ts, int Tv,
                                                              const double* __restrict__ s, const double* __restrict__ v,
                                                              const double* __restrict__ wl0, const double* __restrict__ wl1,
                                                              double* lat_s, double* lat_v, int* idx, unsigned char* smem_raw) {
  double* y = reinterpret_cast<double*>(smem_raw);      // [n][t] : scalars 2 (re,im), vectors 8 (cart re[4], im[4])
  const int b = blockIdx.x, TT = Ts + Tv;
  const int YS = 2 * Ts + 8 * Tv;                       // per-node stride
  const size_t pl = (size_t)B * N * C;
  for (int e = threadIdx.x; e < N * TT; e += BLOCK) {
    const int n = e / TT, t = e - n * TT;
    const size_t base = ((size_t)b * N + n) * C;
    if (t < Ts) {
      cx<double> acc = {0, 0};
      for (int c = 0; c < C; ++c)
        cfma(acc, cx<double>{wl0[t * C + c], wl0[Ts * C + t * C + c]}, cx<double>{s[base + c], s[pl + base + c]});
      y[n * YS + 2 * t] = acc.r;
      y[n * YS + 2 * t + 1] = acc.i;
    } else {
      const int tv = t - Ts;
      cx<double> acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, pc[4];
      for (int c = 0; c < C; ++c) {
        const cx<double> w = {wl1[tv * C + c], wl1[Tv * C + tv * C + c]};
#pragma unroll
        for (int m = 0; m < 4; ++m) cfma(acc[m], w, cx<double>{v[(base + c) * 4 + m], v[(pl + base + c) * 4 + m]});
      }
      cart_from_canon(acc, pc);
      double* o = y + n * YS + 2 * Ts + 8 * tv;
#pragma unroll
      for (int m = 0; m < 4; ++m) { o[m] = pc[m].r; o[4 + m] = pc[m].i; }
    }
  }
  __syncthreads();
  // one thread per (plane, channel): arg-min / arg-max over particles (first occurrence), padded particles included
  for (int e = threadIdx.x; e < 2 * TT; e += BLOCK) {
    const int z = e / TT, t = e - z * TT;
    int imin = 0, imax = 0;
    double smin = 0, smax = 0;
    for (int n = 0; n < N; ++n) {
      double lo, hi;
      if (t < Ts) {
        const double val = y[n * YS + 2 * t + z];
        lo = val;                 // get_min_features: the value itself (lgn_encoder.py:544-545)
        hi = val * val;           // get_max_features: E^2 - |p|^2 with no spatial part = value^2 (lgn_encoder.py:568-569)
      } else {
        const double* o = y + n * YS + 2 * Ts + 8 * (t - Ts) + 4 * z;
        lo = hi = o[0] * o[0] - ((o[1] * o[1] + o[2] * o[2]) + o[3] * o[3]);
      }
      if (n == 0 || lo < smin) { smin = lo; imin = n; }
      if (n == 0 || hi > smax) { smax = hi; imax = n; }
    }
    idx[(((size_t)b * 2 + z) * TT + t) * 2 + 0] = imin;
    idx[(((size_t)b * 2 + z) * TT + t) * 2 + 1] = imax;
    if (t < Ts) {
      lat_s[((size_t)z * B + b) * 2 * Ts + t] = y[imin * YS + 2 * t + z];
      lat_s[((size_t)z * B + b) * 2 * Ts + Ts + t] = y[imax * YS + 2 * t + z];
    } else {
      const int tv = t - Ts;
      for (int m = 0; m < 4; ++m) {
        lat_v[(((size_t)z * B + b) * 2 * Tv + tv) * 4 + m] = y[imin * YS + 2 * Ts + 8 * tv + 4 * z + m];
        lat_v[(((size_t)z * B + b) * 2 * Tv + Tv + tv) * 4 + m] = y[imax * YS + 2 * Ts + 8 * tv + 4 * z + m];
      }
    }
  }
}
__global__ __launch_bounds__(BLOCK) void enc_latent_fwd_kernel(int B, int N, int C, int Ts, int Tv,
                                                              const double* __restrict__ s, const double* __restrict__ v,
                                                              const double* __restrict__ wl0, const double* __restrict__ wl1,
                                                              double* lat_s, double* lat_v, int* idx) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  enc_latent_fwd_body(B, N, C, Ts, Tv, s, v, wl0, wl1, lat_s, lat_v, idx, smem_raw);
}

// backward: scatter the latent gradient to the selected particles, undo rep_to_p and the MixReps.
// part row per jet: dWl0 [2][Ts][C] then dWl1 [2][Tv][C].  The jet's node features and the mixing weights are staged
// in LDS once; the weight gradient runs over (channel pair, node part) items whose parts meet in LDS in a fixed order.
constexpr int LAT_PARTS = 4;
__device__ __forceinline__ void enc_latent_bwd_body(int B, int N, int C, int Ts, int Tv,
                                                              const double* __restrict__ s, const double* __restrict__ v,
                                                              const double* __restrict__ wl0, const double* __restrict__ wl1,
                                                              const double* __restrict__ g_lat_s, const double* __restrict__ g_lat_v,
                                                              const int* __restrict__ idx, double* g_s, double* g_v, double* part, unsigned char* smem_raw) {
  const int b = blockIdx.x, TT = Ts + Tv;
  const int YS = 2 * Ts + 8 * Tv;
  double* gy = reinterpret_cast<double*>(smem_raw);     // [N][YS] same layout as y in the forward; vectors become canonical grads
  double* sv = gy + N * YS;                             // [N][C][10]: s re, im, v re[4], im[4]
  double* w0l = sv + N * C * 10;                        // [2][Ts][C]
  double* w1l = w0l + 2 * Ts * C;                       // [2][Tv][C]
  double* red = w1l + 2 * Tv * C;                       // [LAT_PARTS][TT*C][2]
  const size_t pl = (size_t)B * N * C;
  for (int e = threadIdx.x; e < N * YS; e += BLOCK) gy[e] = 0.0;
  for (int e = threadIdx.x; e < N * C; e += BLOCK) {
    const size_t base = (size_t)b * N * C + e;
    sv[e * 10] = s[base];
    sv[e * 10 + 1] = s[pl + base];
  }
  for (int e = threadIdx.x; e < N * C * 4; e += BLOCK) {
    const size_t base = (size_t)b * N * C * 4 + e;
    sv[(e >> 2) * 10 + 2 + (e & 3)] = v[base];
    sv[(e >> 2) * 10 + 6 + (e & 3)] = v[pl * 4 + base];
  }
  for (int e = threadIdx.x; e < 2 * Ts * C; e += BLOCK) w0l[e] = wl0[e];
  for (int e = threadIdx.x; e < 2 * Tv * C; e += BLOCK) w1l[e] = wl1[e];
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * TT; e += BLOCK) {    // (plane, channel) owners: no write conflicts
    const int z = e / TT, t = e - z * TT;
    for (int kind = 0; kind < 2; ++kind) {
      const int n = idx[(((size_t)b * 2 + z) * TT + t) * 2 + kind];
      if (t < Ts) {
        gy[n * YS + 2 * t + z] += g_lat_s[((size_t)z * B + b) * 2 * Ts + kind * Ts + t];
      } else {
        const int tv = t - Ts;
        for (int m = 0; m < 4; ++m)
          gy[n * YS + 2 * Ts + 8 * tv + 4 * z + m] += g_lat_v[(((size_t)z * B + b) * 2 * Tv + kind * Tv + tv) * 4 + m];
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * Tv; e += BLOCK) {    // Cartesian gradient -> canonical gradient, in place
    const int n = e / Tv, tv = e - n * Tv;
    double* o = gy + n * YS + 2 * Ts + 8 * tv;
    cx<double> g[4], gc[4];
    for (int m = 0; m < 4; ++m) g[m] = {o[m], o[4 + m]};
    cart_from_canon_bwd(g, gc);
    for (int m = 0; m < 4; ++m) { o[m] = gc[m].r; o[4 + m] = gc[m].i; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * C; e += BLOCK) {     // gradient w.r.t. the last level's node features
    const int n = e / C, c = e - n * C;
    const size_t base = ((size_t)b * N + n) * C + c;
    cx<double> as = {0, 0}, av[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int t = 0; t < Ts; ++t)
      cfmac(as, cx<double>{gy[n * YS + 2 * t], gy[n * YS + 2 * t + 1]}, cx<double>{w0l[t * C + c], w0l[Ts * C + t * C + c]});
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
  const int nper = (N + LAT_PARTS - 1) / LAT_PARTS;
  for (int e = threadIdx.x; e < LAT_PARTS * TT * C; e += BLOCK) {
    const int pi = e / (TT * C), r = e - pi * TT * C, t = r / C, c = r - t * C;
    const int n1 = min(N, (pi + 1) * nper);
    cx<double> acc = {0, 0};
    if (t < Ts) {
      for (int n = pi * nper; n < n1; ++n)
        cfmac(acc, cx<double>{gy[n * YS + 2 * t], gy[n * YS + 2 * t + 1]}, cx<double>{sv[(n * C + c) * 10], sv[(n * C + c) * 10 + 1]});
    } else {
      const int tv = t - Ts;
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
__global__ __launch_bounds__(BLOCK) void enc_latent_bwd_kernel(int B, int N, int C, int Ts, int Tv,
                                                              const double* __restrict__ s, const double* __restrict__ v,
                                                              const double* __restrict__ wl0, const double* __restrict__ wl1,
                                                              const double* __restrict__ g_lat_s, const double* __restrict__ g_lat_v,
                                                              const int* __restrict__ idx, double* g_s, double* g_v, double* part) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  enc_latent_bwd_body(B, N, C, Ts, Tv, s, v, wl0, wl1, g_lat_s, g_lat_v, idx, g_s, g_v, part, smem_raw);
}

// ============================================================================================
// decoder input: latent vectors -> particles (latent_to_graph) -> canonical momenta -> input_func_node
//   pdec [2][B][N][4]; s0 [2][B][N][C] = W00[c] (1+1i); v0 [2][B][N][C][4] = W11[c] pc[n]
// ============================================================================================
__device__ __forceinline__ void dec_input_fwd_body(int B, int N, int C, int Tin, const double* __restrict__ lat_v,
                                                             const double* __restrict__ wg1, const double* __restrict__ w0,
                                                             const double* __restrict__ w1, double* pdec, double* s0, double* v0, unsigned char* smem_raw) {
  const int b = blockIdx.x;
  const size_t plp = (size_t)B * N * 4, pl = (size_t)B * N * C;
  // the latent vectors of the jet and latent_to_graph's weights go to LDS first: the sum over the latent channels below
  // would otherwise be a chain of Tin dependent global round trips per particle
  double* wgl = reinterpret_cast<double*>(smem_raw);     // [2][N][Tin]
  double* latl = wgl + 2 * N * Tin;                      // [Tin][8]
  for (int e = threadIdx.x; e < 2 * N * Tin; e += BLOCK) wgl[e] = wg1[e];
  for (int e = threadIdx.x; e < Tin * 4; e += BLOCK) {
    latl[(e >> 2) * 8 + (e & 3)] = lat_v[(size_t)b * Tin * 4 + e];
    latl[(e >> 2) * 8 + 4 + (e & 3)] = lat_v[((size_t)B + b) * Tin * 4 + e];
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += BLOCK) {
    cx<double> cart[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, pc[4];
    for (int t = 0; t < Tin; ++t) {
      const cx<double> w = {wgl[n * Tin + t], wgl[N * Tin + n * Tin + t]};
#pragma unroll
      for (int m = 0; m < 4; ++m) cfma(cart[m], w, cx<double>{latl[t * 8 + m], latl[t * 8 + 4 + m]});
    }
    canon_cplx(cart, pc);
    const size_t node = (size_t)b * N + n;
#pragma unroll
    for (int m = 0; m < 4; ++m) { pdec[node * 4 + m] = pc[m].r; pdec[plp + node * 4 + m] = pc[m].i; }
    for (int c = 0; c < C; ++c) {
      const size_t e = node * C + c;
      // W00 * (1 + 1i): the zonal (0,0) function is ones on both planes (zonal_functions.py:182-186)
      s0[e] = w0[c] - w0[C + c];
      s0[pl + e] = w0[C + c] + w0[c];
      const cx<double> w = {w1[c], w1[C + c]};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        cx<double> r = cmul(w, pc[m]);
        v0[e * 4 + m] = r.r;
        v0[pl * 4 + e * 4 + m] = r.i;
      }
    }
  }
}
__global__ __launch_bounds__(BLOCK) void dec_input_fwd_kernel(int B, int N, int C, int Tin, const double* __restrict__ lat_v,
                                                             const double* __restrict__ wg1, const double* __restrict__ w0,
                                                             const double* __restrict__ w1, double* pdec, double* s0, double* v0) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  dec_input_fwd_body(B, N, C, Tin, lat_v, wg1, w0, w1, pdec, s0, v0, smem_raw);
}

// backward.  g_p holds the gradient w.r.t. pdec accumulated by the levels.  part row per jet:
//   dW00 [2][C] | dW11 [2][C] | dWg1 [2][N][Tin]
// Every global operand is read once up front; the reductions over the particles run on LDS data.
__device__ __forceinline__ void dec_input_bwd_body(int B, int N, int C, int Tin, const double* __restrict__ lat_v,
                                                             const double* __restrict__ wg1, const double* __restrict__ w1,
                                                             const double* __restrict__ pdec, const double* __restrict__ g_p,
                                                             const double* __restrict__ g_s0, const double* __restrict__ g_v0,
                                                             double* g_lat_v, double* part, unsigned char* smem_raw) {
  double* gcan = reinterpret_cast<double*>(smem_raw);    // [N][8] gradient w.r.t. the canonical momenta
  double* gcart = gcan + N * 8;                          // [N][8] gradient w.r.t. the complex Cartesian momenta
  double* tmp = gcart + N * 8;                           // [N*C][4] input-mixing terms
  double* wgl = tmp + N * C * 4;                         // [2][N][Tin]
  double* latl = wgl + 2 * N * Tin;                      // [Tin][8]
  const int b = blockIdx.x;
  const size_t plp = (size_t)B * N * 4, pl = (size_t)B * N * C;
  double* row = part + (size_t)b * (4 * C + 2 * N * Tin);
  for (int e = threadIdx.x; e < 2 * N * Tin; e += BLOCK) wgl[e] = wg1[e];
  for (int e = threadIdx.x; e < Tin * 4; e += BLOCK) {
    latl[(e >> 2) * 8 + (e & 3)] = lat_v[(size_t)b * Tin * 4 + e];
    latl[(e >> 2) * 8 + 4 + (e & 3)] = lat_v[((size_t)B + b) * Tin * 4 + e];
  }
  for (int e = threadIdx.x; e < N * 4; e += BLOCK) {     // (n, m): G_pc[m] = g_p + sum_c g_v0[c][m] conj(W11[c])
    const int n = e >> 2, m = e & 3;
    const size_t node = (size_t)b * N + n;
    cx<double> g = {g_p[node * 4 + m], g_p[plp + node * 4 + m]};
    for (int c = 0; c < C; ++c) {
      const size_t x = (node * C + c) * 4 + m;
      cfmac(g, cx<double>{g_v0[x], g_v0[pl * 4 + x]}, cx<double>{w1[c], w1[C + c]});
    }
    gcan[n * 8 + m] = g.r;
    gcan[n * 8 + 4 + m] = g.i;
  }
  for (int i = threadIdx.x; i < N * C; i += BLOCK) {     // (n, c): terms of dW00, dW11
    const int n = i / C;
    const size_t node = (size_t)b * N + n, e = (size_t)b * N * C + i;
    cx<double> d0 = {0, 0}, d1 = {0, 0};
    cfmac(d0, cx<double>{g_s0[e], g_s0[pl + e]}, cx<double>{1.0, 1.0});
#pragma unroll
    for (int m = 0; m < 4; ++m)
      cfmac(d1, cx<double>{g_v0[e * 4 + m], g_v0[pl * 4 + e * 4 + m]}, cx<double>{pdec[node * 4 + m], pdec[plp + node * 4 + m]});
    tmp[i * 4 + 0] = d0.r;  tmp[i * 4 + 1] = d0.i;  tmp[i * 4 + 2] = d1.r;  tmp[i * 4 + 3] = d1.i;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += BLOCK) {
    cx<double> g[4], gc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) g[m] = {gcan[n * 8 + m], gcan[n * 8 + 4 + m]};
    canon_cplx_bwd(g, gc);
#pragma unroll
    for (int m = 0; m < 4; ++m) { gcart[n * 8 + m] = gc[m].r; gcart[n * 8 + 4 + m] = gc[m].i; }
  }
  if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + 4 * C) {   // input mixing weights (a wave that is idle above)
    const int k = (threadIdx.x - 64) / C, c = (threadIdx.x - 64) - k * C;
    double acc = 0.0;
    for (int n = 0; n < N; ++n) acc += tmp[(n * C + c) * 4 + k];
    row[k * C + c] = acc;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < Tin * 4; e += BLOCK) {    // g_lat_v[t][m] = sum_n G_cart[n][m] conj(Wg1[n][t])
    const int t = e >> 2, m = e & 3;
    cx<double> acc = {0, 0};
    for (int n = 0; n < N; ++n)
      cfmac(acc, cx<double>{gcart[n * 8 + m], gcart[n * 8 + 4 + m]}, cx<double>{wgl[n * Tin + t], wgl[N * Tin + n * Tin + t]});
    g_lat_v[((size_t)b * Tin + t) * 4 + m] = acc.r;
    g_lat_v[(((size_t)B + b) * Tin + t) * 4 + m] = acc.i;
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
  dec_input_bwd_body(B, N, C, Tin, lat_v, wg1, w1, pdec, g_p, g_s0, g_v0, g_lat_v, part, smem_raw);
}

// ============================================================================================
// encoder -> decoder junction, one launch per direction: the decoder input of a jet needs only that jet's latent
// vectors (and vice versa for the gradients), so the two per-jet kernels run back to back in the same workgroup.
// ============================================================================================
__global__ __launch_bounds__(BLOCK) void junction_fwd_kernel(int B, int N, int CL, int Ts, int Tv, const double* s, const double* v,
                                                            const double* wl0, const double* wl1, double* lat_s, double* lat_v,
                                                            int* idx, int C0, const double* wg1, const double* w0, const double* w1,
                                                            double* pdec, double* s0, double* v0) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  enc_latent_fwd_body(B, N, CL, Ts, Tv, s, v, wl0, wl1, lat_s, lat_v, idx, smem_raw);
  __syncthreads();                                       // this jet's lat_v (global) is visible to the whole workgroup
  dec_input_fwd_body(B, N, C0, 2 * Tv, lat_v, wg1, w0, w1, pdec, s0, v0, smem_raw);
}
__global__ __launch_bounds__(BLOCK) void junction_bwd_kernel(int B, int N, int C0, int Tin, const double* lat_v, const double* wg1,
                                                            const double* w1, const double* pdec, const double* g_p,
                                                            const double* g_s0, const double* g_v0, double* g_lat_v, double* part_dec,
                                                            int CL, int Ts, int Tv, const double* s, const double* v,
                                                            const double* wl0, const double* wl1, const double* g_lat_s,
                                                            const int* idx, double* g_s, double* g_v, double* part_enc) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  dec_input_bwd_body(B, N, C0, Tin, lat_v, wg1, w1, pdec, g_p, g_s0, g_v0, g_lat_v, part_dec, smem_raw);
  __syncthreads();                                       // this jet's g_lat_v is visible; the LDS scratch is free again
  enc_latent_bwd_body(B, N, CL, Ts, Tv, s, v, wl0, wl1, g_lat_s, g_lat_v, idx, g_s, g_v, part_enc, smem_raw);
}

// ============================================================================================
// decoder output + get_real('sum') + Chamfer loss, forward and backward in one pass per jet
//   recon [2][B][N][4]; loss_part [B]; g_v [2][B][N][C][4]; part row per jet: dWo1 [2][C]
// ============================================================================================
__global__ __launch_bounds__(BLOCK) void dec_output_loss_kernel(int B, int N, int C, const double* __restrict__ v,
                                                               const double* __restrict__ wo1, const double* __restrict__ target,
                                                               double loss_scale, double* recon, double* loss_part, double* g_v,
                                                               double* part) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* x = reinterpret_cast<double*>(smem_raw);       // [N][4] real reconstruction (re + im)
  double* tg = x + N * 4;                                // [N][4] target
  double* rmin = tg + N * 4;                             // [N]
  double* cmin = rmin + N;                               // [N]
  double* gx = cmin + N;                                 // [N][4]
  int* rarg = reinterpret_cast<int*>(gx + N * 4);        // [N]
  int* carg = rarg + N;                                  // [N]
  __shared__ double red[4];
  const int b = blockIdx.x;
  const size_t plp = (size_t)B * N * 4, pl = (size_t)B * N * C;
  for (int n = threadIdx.x; n < N; n += BLOCK) {
    const size_t node = (size_t)b * N + n;
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
      x[n * 4 + m] = pc[m].r + pc[m].i;                  // get_real(., 'sum')
      tg[n * 4 + m] = target[node * 4 + m];
    }
  }
  __syncthreads();
  // squared Euclidean distances d(i,j) = |t_j - x_i|^2; row minima (over j) and column minima (over i)
  for (int e = threadIdx.x; e < 2 * N; e += BLOCK) {
    const bool rowwise = e < N;
    const int a = rowwise ? e : e - N;
    double best = 0;
    int arg = 0;
    for (int o = 0; o < N; ++o) {
      const double* xi = x + (rowwise ? a : o) * 4;
      const double* tj = tg + (rowwise ? o : a) * 4;
      double d0 = tj[0] - xi[0], d1 = tj[1] - xi[1], d2 = tj[2] - xi[2], d3 = tj[3] - xi[3];
      double d = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
      if (o == 0 || d < best) { best = d; arg = o; }
    }
    if (rowwise) { rmin[a] = best; rarg[a] = arg; } else { cmin[a] = best; carg[a] = arg; }
  }
  __syncthreads();
  double lsum = 0;
  for (int n = threadIdx.x; n < N; n += BLOCK) lsum += (rmin[n] + cmin[n]) * 0.5;
  lsum = block_sum(lsum, red);
  if (threadIdx.x == 0) loss_part[b] = lsum;
  // d loss / d x_i = (x_i - t_{j*(i)}) + sum_{j : i*(j) = i} (x_i - t_j)
  for (int i = threadIdx.x; i < N; i += BLOCK) {
    double g[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) g[m] = x[i * 4 + m] - tg[rarg[i] * 4 + m];
    for (int j = 0; j < N; ++j)
      if (carg[j] == i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) g[m] += x[i * 4 + m] - tg[j * 4 + m];
      }
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[i * 4 + m] = g[m] * loss_scale;
  }
  __syncthreads();
  // back through get_real (both planes receive g), rep_to_p and mix_to_output
  for (int e = threadIdx.x; e < N * C; e += BLOCK) {
    const int n = e / C, c = e - n * C;
    cx<double> g[4], gc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) g[m] = {gx[n * 4 + m], gx[n * 4 + m]};
    cart_from_canon_bwd(g, gc);
    const cx<double> w = {wo1[c], wo1[C + c]};
    const size_t base = ((size_t)b * N + n) * C + c;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      cx<double> r = cmulc(gc[m], w);
      g_v[base * 4 + m] = r.r;
      g_v[pl * 4 + base * 4 + m] = r.i;
    }
  }
  // dWo1[c] = sum_n sum_m G_yc[n][m] conj(v[n][c][m]): (n, c) terms to LDS (the distance scratch is dead), then 2C sums
  __syncthreads();
  double* tmp = x;                                        // [N*C][2] over x | tg (8N doubles >= 2NC for C <= 4) ...
  double* tmpbig = reinterpret_cast<double*>(carg + N);   // ... or the spill region for wider outputs
  if (2 * C > 8) tmp = tmpbig;
  for (int e = threadIdx.x; e < N * C; e += BLOCK) {
    const int n = e / C;
    cx<double> g[4], gc[4], d = {0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m) g[m] = {gx[n * 4 + m], gx[n * 4 + m]};
    cart_from_canon_bwd(g, gc);
    const size_t base = (size_t)b * N * C + e;
#pragma unroll
    for (int m = 0; m < 4; ++m) cfmac(d, gc[m], cx<double>{v[base * 4 + m], v[pl * 4 + base * 4 + m]});
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
// The optimiser step counter lives on the device (graph replays stay correct): this step is number *step_dev + 1.
__global__ __launch_bounds__(BLOCK) void l1_adam_kernel(long n, double* w, double* g, double* m, double* v, double lambda, double lr,
                                                       double beta1, double beta2, double eps, const long* step_dev, int do_adam,
                                                       double* l1_part) {
  __shared__ double red[4];
  double bc1 = 1.0, bc2_sqrt = 1.0;
  if (do_adam) {
    const double t = (double)(*step_dev + 1);
    bc1 = 1.0 - pow(beta1, t);
    bc2_sqrt = sqrt(1.0 - pow(beta2, t));
  }
  double l1 = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double wi = w[i];
    l1 += fabs(wi);
    const double gi = g[i] + lambda * ((wi > 0.0) - (wi < 0.0));
    g[i] = gi;
    if (do_adam) {
      const double mi = m[i] + (gi - m[i]) * (1.0 - beta1);
      const double vi = v[i] * beta2 + (1.0 - beta2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      const double denom = sqrt(vi) / bc2_sqrt + eps;
      w[i] = wi - (lr / bc1) * (mi / denom);
    }
  }
  l1 = block_sum(l1, red);
  if (threadIdx.x == 0) l1_part[blockIdx.x] = l1;
}

// loss_out[0] = chamfer + lambda * sum|w|, loss_out[1] = chamfer, loss_out[2] = sum|w|; bumps the step counter.
__global__ __launch_bounds__(BLOCK) void loss_final_kernel(const double* __restrict__ loss_part, int nB, const double* __restrict__ l1_part,
                                                          int nblk, double lambda, double* loss_out, long* step_dev, int bump) {
  __shared__ double red[4];
  double a = 0, l = 0;
  for (int i = threadIdx.x; i < nblk; i += BLOCK) a += l1_part[i];
  for (int i = threadIdx.x; i < nB; i += BLOCK) l += loss_part[i];
  a = block_sum(a, red);
  l = block_sum(l, red);
  if (threadIdx.x == 0) {
    loss_out[0] = l + lambda * a;
    loss_out[1] = l;
    loss_out[2] = a;
    if (bump) *step_dev += 1;
  }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static int grid_for(size_t total) {
  size_t g = (total + BLOCK - 1) / BLOCK;
  return (int)(g < 2048 ? (g ? g : 1) : 2048);
}

int enc_input_fwd(int B, int N, int C, const double* p4, const double* w0, const double* w1, double* s, double* v, hipStream_t st,
                  double* z1, size_t n1, double* z2, size_t n2) {
  hipLaunchKernelGGL(enc_input_fwd_kernel, dim3(grid_for((size_t)B * N * C)), dim3(BLOCK), 0, st, B, N, C, p4, w0, w1, s, v, z1, n1,
                     z2, n2);
  LGN_CHECK_LAUNCH();
  return 0;
}
int enc_input_bwd(int B, int N, int C, const double* p4, const double* g_s, const double* g_v, double* part, hipStream_t st) {
  hipLaunchKernelGGL(enc_input_bwd_kernel, dim3(B), dim3(BLOCK), sizeof(double) * N * C * 4, st, B, N, C, p4, g_s, g_v, part);
  LGN_CHECK_LAUNCH();
  return 0;
}
static size_t latent_smem(int N, int Ts, int Tv) { return sizeof(double) * (size_t)N * (2 * Ts + 8 * Tv); }
int enc_latent_fwd(int B, int N, int C, int Ts, int Tv, const double* s, const double* v, const double* wl0, const double* wl1,
                   double* lat_s, double* lat_v, int* idx, hipStream_t st) {
  const size_t smem = latent_smem(N, Ts, Tv);
  LGN_CHECK_ARG(smem <= 160 * 1024, "enc_latent: N=%d tau=(%d,%d) needs %zu B of LDS", N, Ts, Tv, smem);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_latent_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(enc_latent_fwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, Ts, Tv, s, v, wl0, wl1, lat_s, lat_v, idx);
  LGN_CHECK_LAUNCH();
  return 0;
}
int enc_latent_bwd(int B, int N, int C, int Ts, int Tv, const double* s, const double* v, const double* wl0, const double* wl1,
                   const double* g_lat_s, const double* g_lat_v, const int* idx, double* g_s, double* g_v, double* part,
                   hipStream_t st) {
  const size_t smem = latent_smem(N, Ts, Tv) + sizeof(double) * ((size_t)N * C * 10 + 2 * (Ts + Tv) * C + LAT_PARTS * (Ts + Tv) * C * 2);
  LGN_CHECK_ARG(smem <= 160 * 1024, "enc_latent_bwd: needs %zu B of LDS", smem);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_latent_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(enc_latent_bwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, Ts, Tv, s, v, wl0, wl1, g_lat_s, g_lat_v, idx,
                     g_s, g_v, part);
  LGN_CHECK_LAUNCH();
  return 0;
}
int dec_input_fwd(int B, int N, int C, int Tin, const double* lat_v, const double* wg1, const double* w0, const double* w1,
                  double* pdec, double* s0, double* v0, hipStream_t st) {
  const size_t smem = sizeof(double) * (2 * (size_t)N * Tin + (size_t)Tin * 8);
  LGN_CHECK_ARG(smem <= 64 * 1024, "dec_input_fwd: needs %zu B of LDS", smem);
  hipLaunchKernelGGL(dec_input_fwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, Tin, lat_v, wg1, w0, w1, pdec, s0, v0);
  LGN_CHECK_LAUNCH();
  return 0;
}
int dec_input_bwd(int B, int N, int C, int Tin, const double* lat_v, const double* wg1, const double* w1, const double* pdec,
                  const double* g_p, const double* g_s0, const double* g_v0, double* g_lat_v, double* part, hipStream_t st) {
  const size_t smem = sizeof(double) * ((size_t)N * 16 + (size_t)N * C * 4 + 2 * (size_t)N * Tin + (size_t)Tin * 8);
  LGN_CHECK_ARG(smem <= 160 * 1024, "dec_input_bwd: needs %zu B of LDS", smem);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_input_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(dec_input_bwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, Tin, lat_v, wg1, w1, pdec, g_p,
                     g_s0, g_v0, g_lat_v, part);
  LGN_CHECK_LAUNCH();
  return 0;
}
int dec_output_loss(int B, int N, int C, const double* v, const double* wo1, const double* target, double loss_scale, double* recon,
                    double* loss_part, double* g_v, double* part, hipStream_t st) {
  const size_t smem = sizeof(double) * (size_t)N * 14 + sizeof(int) * (size_t)N * 2 + 16 + (2 * C > 8 ? sizeof(double) * (size_t)N * C * 2 : 0);
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
int dec_output_bwd(int B, int N, int C, const double* v, const double* wo1, const double* g_recon, double* g_v, double* part,
                   hipStream_t st) {
  const size_t smem = sizeof(double) * ((size_t)N * 8 + (size_t)N * C * 2);
  LGN_CHECK_ARG(smem <= 160 * 1024, "dec_output_bwd: needs %zu B of LDS", smem);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_output_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(dec_output_bwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C, v, wo1, g_recon, g_v, part);
  LGN_CHECK_LAUNCH();
  return 0;
}
static size_t latent_bwd_smem(int N, int C, int Ts, int Tv) {
  return latent_smem(N, Ts, Tv) + sizeof(double) * ((size_t)N * C * 10 + 2 * (Ts + Tv) * C + LAT_PARTS * (Ts + Tv) * C * 2);
}
static size_t dec_input_bwd_smem(int N, int C, int Tin) {
  return sizeof(double) * ((size_t)N * 16 + (size_t)N * C * 4 + 2 * (size_t)N * Tin + (size_t)Tin * 8);
}
int junction_fwd(int B, int N, int CL, int Ts, int Tv, const double* s, const double* v, const double* wl0, const double* wl1,
                 double* lat_s, double* lat_v, int* idx, int C0, const double* wg1, const double* w0, const double* w1, double* pdec,
                 double* s0, double* v0, hipStream_t st) {
  const size_t s2 = sizeof(double) * (2 * (size_t)N * 2 * Tv + (size_t)2 * Tv * 8);   // dec_input_fwd part (Tin = 2 Tv)
  const size_t smem = latent_smem(N, Ts, Tv) > s2 ? latent_smem(N, Ts, Tv) : s2;
  LGN_CHECK_ARG(smem <= 160 * 1024, "junction_fwd: N=%d tau=(%d,%d) needs %zu B of LDS", N, Ts, Tv, smem);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(junction_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(junction_fwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, CL, Ts, Tv, s, v, wl0, wl1, lat_s, lat_v, idx, C0, wg1,
                     w0, w1, pdec, s0, v0);
  LGN_CHECK_LAUNCH();
  return 0;
}
int junction_bwd(int B, int N, int C0, int Tin, const double* lat_v, const double* wg1, const double* w1, const double* pdec,
                 const double* g_p, const double* g_s0, const double* g_v0, double* g_lat_v, double* part_dec, int CL, int Ts, int Tv,
                 const double* s, const double* v, const double* wl0, const double* wl1, const double* g_lat_s, const int* idx,
                 double* g_s, double* g_v, double* part_enc, hipStream_t st) {
  const size_t a = dec_input_bwd_smem(N, C0, Tin), b = latent_bwd_smem(N, CL, Ts, Tv), smem = a > b ? a : b;
  LGN_CHECK_ARG(smem <= 160 * 1024, "junction_bwd: needs %zu B of LDS", smem);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(junction_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(junction_bwd_kernel, dim3(B), dim3(BLOCK), smem, st, B, N, C0, Tin, lat_v, wg1, w1, pdec, g_p, g_s0, g_v0, g_lat_v,
                     part_dec, CL, Ts, Tv, s, v, wl0, wl1, g_lat_s, idx, g_s, g_v, part_enc);
  LGN_CHECK_LAUNCH();
  return 0;
}

// loss_out: 3 results followed by LGN_FINALIZE_SCRATCH doubles of scratch (per-workgroup |w| partials)
int finalize_step(double* w, double* g, long n, const double* loss_part, int nB, double lambda, double* m, double* v, long* step_dev,
                  double lr, double beta1, double beta2, double eps, int do_adam, double* loss_out, hipStream_t st) {
  int nblk = grid_for((size_t)n);
  if (nblk > LGN_FINALIZE_SCRATCH) nblk = LGN_FINALIZE_SCRATCH;
  double* l1_part = loss_out + 3;
  hipLaunchKernelGGL(l1_adam_kernel, dim3(nblk), dim3(BLOCK), 0, st, n, w, g, m, v, lambda, lr, beta1, beta2, eps, step_dev, do_adam,
                     l1_part);
  LGN_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(BLOCK), 0, st, loss_part, nB, l1_part, nblk, lambda, loss_out, step_dev, do_adam);
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace lgn
