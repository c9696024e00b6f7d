// lgn-autoencoder_amd/csrc/generic_moments_sep.hip -- decoder moments of the table-driven levels in separable form, on the
// tile-blocked layouts of generic_local_static.hip (X / dX [tile][C][Q][2][64], U / dU [tile][C][5 Q][2][64]).
//
// Same mathematics as moments_dec_sep_fwd/bwd_kernel of generic_moments2.hip (which stay for the node-major layout); see the
// derivation there: the decoder's edge mask is identically zero (reference: lgn/models/lgn_decoder.py:335-340), its radial
// functions are the Linear biases, and every pair sum separates into jet-level sums
//   SX[q] = sum_j X_j[q],  SXP[q][m] = sum_j X_j[q] P_j[m]                       (P: canonical momenta centred on the jet mean)
//   forward    U[i][q][0] = e0 SX[q],   U[i][q][1+m] = R1 (P_i[m] SX[q] - SXP[q][m])          e0 = 2 i b0,  R1 = b1 (1 + i)
//   backward   S[q][k] = sum_i dU[i][q][k],  SP[q][m] = sum_i dU[i][q][1+m] conj(P_i[m])
//              dX[j][q] += conj(e0) S[q][0] + conj(R1) sum_m (SP[q][m] - conj(P_j[m]) S[q][1+m])
//              d p_i   += conj(R1) sum_q (dU[i][q][1+m] conj(SX[q]) - S[q][1+m] conj(X_i[q]))
//              d b0 = 2 Im sum_q S[q][0] conj(SX[q]);   d b1 = (Re + Im) sum_{q,m} SP conj(SX) - S[1+m] conj(SXP)
// What changes is the mapping.  The work is pure streaming (147 MB of U per launch at cfg5, a few flops per byte), so:
//   wave = (jet b, channel c), lane = particle: every load / store is one coalesced run of the jet's N lanes inside a tile row
//   (N <= 64; a jet may straddle two tiles, the per-lane base address takes care of it); no LDS staging, no barriers in the
//   q loop; the jet-level sums are register butterflies (wave_sum.hpp) whose totals come back as wave-uniform scalars.
//   workgroup = the C waves of a jet: d p is the only quantity summed over channels (LDS, channel order: deterministic).
#include "ops.hpp"
#include "wave_sum.hpp"

namespace lgn {
namespace {

struct Jet {
  int n, lane, N;
  bool ok;
  cx<double> P[4];         // centred canonical momenta of this lane's particle (0 beyond the jet)
};
// momenta [2][B][N][4]; mean over the jet's particles subtracted (differences are unchanged; see generic_moments2.hip)
__device__ __forceinline__ Jet load_jet(const GenArgs& a, int b, int lane) {
  Jet J;
  J.N = a.N; J.lane = lane; J.ok = lane < a.N; J.n = b * a.N + (J.ok ? lane : 0);
  const size_t plane_p = (size_t)a.B * a.N * 4;
  double v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = J.ok ? a.p[(r >> 2) * plane_p + (size_t)J.n * 4 + (r & 3)] : 0.0;
  double s[8];
  wave_allsum<8>(v, s, lane);
  const double inv = 1.0 / a.N;
#pragma unroll
  for (int m = 0; m < 4; ++m) J.P[m] = J.ok ? cx<double>{v[m] - s[m] * inv, v[4 + m] - s[4 + m] * inv} : cx<double>{0, 0};
  return J;
}

template <int Q>
__global__ __launch_bounds__(512) void dec_sep_fwd_tb_kernel(GenArgs a) {
  const int b = blockIdx.x, c = threadIdx.x >> 6, lane = threadIdx.x & 63, C = a.C;
  const Jet J = load_jet(a, b, lane);
  const double* __restrict__ xc = a.X + ((size_t)(J.n >> 6) * C + c) * Q * 128 + (J.n & 63);
  double* __restrict__ uc = a.U + ((size_t)(J.n >> 6) * C + c) * Q * 640 + (J.n & 63);
  const double b0 = a.b0[c], b1 = a.b1[c];
  const cx<double> e0 = {0.0, 2.0 * b0}, R1 = {b1, b1};
#pragma unroll 2
  for (int q = 0; q < Q; ++q) {
    const cx<double> x = J.ok ? cx<double>{xc[q * 128], xc[q * 128 + 64]} : cx<double>{0, 0};
    double v[12], s[12];
    v[0] = x.r; v[1] = x.i;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> xp = cmul(x, J.P[m]);
      v[2 + 2 * m] = xp.r; v[3 + 2 * m] = xp.i;
    }
    v[10] = 0.0; v[11] = 0.0;
    wave_allsum<12>(v, s, lane);
    const cx<double> SX = {s[0], s[1]};
    if (J.ok) {
      const cx<double> u0 = cmul(e0, SX);
      uc[(q * 5) * 128] = u0.r;
      uc[(q * 5) * 128 + 64] = u0.i;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        cx<double> t = cmul(J.P[m], SX);
        t.r -= s[2 + 2 * m];
        t.i -= s[3 + 2 * m];
        const cx<double> u = cmul(R1, t);
        uc[(q * 5 + 1 + m) * 128] = u.r;
        uc[(q * 5 + 1 + m) * 128 + 64] = u.i;
      }
    }
  }
}

// Round 6: the jet-level sums as a TABLE instead of the moments tensor -- what local_fwd_sep / local_bwd_sep (generic_local_sep.hip)
// form U from.  Entry (jet b, channel c, component q) = SEP_TBL_STRIDE doubles (layout: local_static_dev.hpp), plus the centred
// canonical momenta of every node, pc [B N][8].  Same sums, same butterfly as dec_sep_fwd_tb_kernel; 5 MB written instead of 147.
template <int Q>
__global__ __launch_bounds__(512) void dec_sep_tab_kernel(GenArgs a, double* __restrict__ tbl, double* __restrict__ pc) {
  __shared__ double sraw[8 * Q * 12];                          // [c][q]: SX | SXP[0..3] | pad
  const int b = blockIdx.x, c = threadIdx.x >> 6, lane = threadIdx.x & 63, C = a.C;
  const Jet J = load_jet(a, b, lane);
  if (c == 0 && J.ok) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      pc[(size_t)J.n * 8 + 2 * m] = J.P[m].r;
      pc[(size_t)J.n * 8 + 2 * m + 1] = J.P[m].i;
    }
  }
  const double* __restrict__ xc = a.X + ((size_t)(J.n >> 6) * C + c) * Q * 128 + (J.n & 63);
#pragma unroll 2
  for (int q = 0; q < Q; ++q) {
    const cx<double> x = J.ok ? cx<double>{xc[q * 128], xc[q * 128 + 64]} : cx<double>{0, 0};
    double v[12];
    v[0] = x.r; v[1] = x.i;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> xp = cmul(x, J.P[m]);
      v[2 + 2 * m] = xp.r; v[3 + 2 * m] = xp.i;
    }
    v[10] = 0.0; v[11] = 0.0;
    wave_sum_store<12>(v, sraw + (c * Q + q) * 12, lane);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < C * Q; e += blockDim.x) {
    const int cc = e / Q;
    const double* s = sraw + e * 12;
    const double b0 = a.b0[cc], b1 = a.b1[cc];
    const cx<double> e0 = {0.0, 2.0 * b0}, R1 = {b1, b1}, SX = {s[0], s[1]};
    double* __restrict__ t = tbl + ((size_t)b * C * Q + e) * SEP_TBL_STRIDE;
    const cx<double> E = cmul(e0, SX), A = cmul(R1, SX);
    t[0] = E.r; t[1] = E.i; t[2] = A.r; t[3] = A.i;
    t[12] = SX.r; t[13] = SX.i;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> sxp = {s[2 + 2 * m], s[3 + 2 * m]}, Bm = cmul(R1, sxp);
      t[4 + 2 * m] = Bm.r; t[5 + 2 * m] = Bm.i;
      t[14 + 2 * m] = sxp.r; t[15 + 2 * m] = sxp.i;
    }
    t[22] = 0.0; t[23] = 0.0;
  }
}

template <int Q>
__global__ __launch_bounds__(512) void dec_sep_bwd_tb_kernel(GenArgs a) {
  const int b = blockIdx.x, c = threadIdx.x >> 6, lane = threadIdx.x & 63, C = a.C, N = a.N;
  __shared__ double gps[8 * 64 * 8];                           // [c][lane][8]  d p of the channel's wave
  const Jet J = load_jet(a, b, lane);
  const double* __restrict__ xc = a.X + ((size_t)(J.n >> 6) * C + c) * Q * 128 + (J.n & 63);
  double* __restrict__ gxc = a.gX + ((size_t)(J.n >> 6) * C + c) * Q * 128 + (J.n & 63);
  const double* __restrict__ guc = a.gU + ((size_t)(J.n >> 6) * C + c) * Q * 640 + (J.n & 63);
  const double b0 = a.b0[c], b1 = a.b1[c];
  const cx<double> e0 = {0.0, 2.0 * b0}, R1 = {b1, b1};
  cx<double> gp[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, A0 = {0, 0}, A1 = {0, 0};
#pragma unroll 2
  for (int q = 0; q < Q; ++q) {
    cx<double> x = {0, 0}, g[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) g[k] = {0, 0};
    if (J.ok) {
      x = {xc[q * 128], xc[q * 128 + 64]};
#pragma unroll
      for (int k = 0; k < 5; ++k) g[k] = {guc[(q * 5 + k) * 128], guc[(q * 5 + k) * 128 + 64]};
    }
    double v[12], sx[12], sg[12], w[8], sp[8];
    v[0] = x.r; v[1] = x.i;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> xp = cmul(x, J.P[m]);
      v[2 + 2 * m] = xp.r; v[3 + 2 * m] = xp.i;
    }
    v[10] = 0.0; v[11] = 0.0;
    wave_allsum<12>(v, sx, lane);                              // SX | SXP[0..3]
#pragma unroll
    for (int k = 0; k < 5; ++k) { v[2 * k] = g[k].r; v[2 * k + 1] = g[k].i; }
    wave_allsum<12>(v, sg, lane);                              // S[0..4]
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> t = cmulc(g[1 + m], J.P[m]);
      w[2 * m] = t.r; w[2 * m + 1] = t.i;
    }
    wave_allsum<8>(w, sp, lane);                               // SP[0..3]
    const cx<double> SX = {sx[0], sx[1]}, S0 = {sg[0], sg[1]};
    // d X
    cx<double> t = {0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> ps = cmulc(cx<double>{sg[2 + 2 * m], sg[3 + 2 * m]}, J.P[m]);
      t.r += sp[2 * m] - ps.r;
      t.i += sp[2 * m + 1] - ps.i;
    }
    cx<double> gx = cmulc(S0, e0);
    cfmac(gx, t, R1);
    if (J.ok) {
      gxc[q * 128] += gx.r;
      gxc[q * 128 + 64] += gx.i;
    }
    // d p (this channel's part), bias gradients (wave-uniform)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> Sm = {sg[2 + 2 * m], sg[3 + 2 * m]};
      cfmac(gp[m], g[1 + m], SX);
      const cx<double> u = cmulc(Sm, x);
      gp[m].r -= u.r;
      gp[m].i -= u.i;
      cfmac(A1, cx<double>{sp[2 * m], sp[2 * m + 1]}, SX);
      const cx<double> u2 = cmulc(Sm, cx<double>{sx[2 + 2 * m], sx[3 + 2 * m]});
      A1.r -= u2.r;
      A1.i -= u2.i;
    }
    cfmac(A0, S0, SX);
  }
  if (lane == 0) {
    double* part = a.part_rad + (size_t)b * rad_partial_size(C, true);
    part[c] = 2.0 * A0.i;                                      // dB0[c] | dB1[c]
    part[C + c] = A1.r + A1.i;
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const cx<double> gm = cmulc(gp[m], R1);
    gps[(c * 64 + lane) * 8 + m] = gm.r;
    gps[(c * 64 + lane) * 8 + 4 + m] = gm.i;
  }
  __syncthreads();
  const size_t plp = (size_t)a.B * N * 4;
  for (int e = threadIdx.x; e < N * 8; e += blockDim.x) {
    const int n = e >> 3, r = e & 7;
    double s = 0.0;
    for (int cc = 0; cc < C; ++cc) s += gps[(cc * 64 + n) * 8 + r];
    a.g_p[(r >> 2) * plp + ((size_t)b * N + n) * 4 + (r & 3)] += s;
  }
}

}  // namespace

// jet table + centred momenta of a decoder level in the tile-blocked layout (tbl: B C Q SEP_TBL_STRIDE doubles, pc: B N 8)
int dec_sep_tab(const GenArgs& a, double* tbl, double* pc, hipStream_t st) {
  LGN_CHECK_ARG(a.tb && a.N <= 64 && a.C <= 8 && (a.Q == 5 || a.Q == 20) && tbl && pc, "dec_sep_tab: unsupported shape (N=%d C=%d Q=%d)", a.N, a.C, a.Q);
  const dim3 grid(a.B), block(64 * a.C);
  if (a.Q == 5) hipLaunchKernelGGL(dec_sep_tab_kernel<5>, grid, block, 0, st, a, tbl, pc);
  else hipLaunchKernelGGL(dec_sep_tab_kernel<20>, grid, block, 0, st, a, tbl, pc);
  LGN_CHECK_LAUNCH();
  return 0;
}

// which: 0 forward, 1 backward.  Returns -2 when the shape is outside these kernels (the caller falls back to generic_moments2.hip).
int moments_dec_sep_tb_dispatch(const GenArgs& a, int which, hipStream_t st) {
  if (!a.tb || a.N > 64 || a.C > 8 || (a.Q != 5 && a.Q != 20)) return -2;
  const dim3 grid(a.B), block(64 * a.C);
  if (which == 0) {
    if (a.Q == 5) hipLaunchKernelGGL(dec_sep_fwd_tb_kernel<5>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(dec_sep_fwd_tb_kernel<20>, grid, block, 0, st, a);
  } else {
    if (a.Q == 5) hipLaunchKernelGGL(dec_sep_bwd_tb_kernel<5>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(dec_sep_bwd_tb_kernel<20>, grid, block, 0, st, a);
  }
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace lgn
