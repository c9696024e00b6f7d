// lgn-autoencoder_amd/csrc/level_bwd2.hip -- version 2 of the two N^2 backward passes of the fused level
// (same mathematics as level_bwd.hip, re-mapped like level_fwd2.hip: a wave owns a group of 4 particles and
// sweeps the other index in tiles of 4; lane = (pair slot = lane & 15, channel-in-group = lane >> 4); the radial
// Linear layers run on the fp64 matrix cores).
//
//   level_bwd_nodes2   "j-centric": g_node_j += sum_i g_ag_i (x) conj(edge_ij)            (+ decoder d p_j)
//   level_bwd_rad2     "i-centric", encoder: the four pair-reductions of the radial-network gradient
//                        T1[r][k] = sum_p G[p][r] on rho_k   T2[r][k] = sum_p G[p][r] on n^2 rho_k^2
//                        S[r] = sum_p G[p][r] on             dB[r] = sum_p G[p][r]
//                      are one GEMM  [r x pairs] x [pairs x 42 columns]  executed with v_mfma_f64_16x16x4_f64; the
//                      two operands are produced pair-per-lane and turned into A/B fragments by a 16x16 transpose
//                      through a wave-private LDS tile (4 writes + 4 reads per lane for A, 12 + 12 for B).
#include "level_dev.hpp"
#include "ops.hpp"

namespace lgn {

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ double dppq(double v, int xor2) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (xor2) {
    lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
  } else {
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += dppq(v, 0);
  v += dppq(v, 1);
  return v;
}
__device__ __forceinline__ double fast_rcp(double u) {
  double r = __builtin_amdgcn_rcp(u);
  double e = __builtin_fma(-u, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-u, r, 1.0);
  return __builtin_fma(r, e, r);
}
template <int C> struct GA2 {
  static constexpr int A3 = 0, A4 = 2 * C, A1 = 4 * C, A2 = 12 * C, SIZE = 20 * C;
};
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
LGN_STAMP_DECL
}  // namespace
LGN_STAMP_READER(lgn_debug_stamps_bwd2)

// =========================================================================================================
// j-centric pass
// =========================================================================================================
template <int C, bool DEC>
__global__ __launch_bounds__(3 * BLOCK) void level_bwd_nodes2_kernel(LevelBwdArgs<double> a) {
  constexpr int NG = (C + 3) / 4;
  constexpr int PS = DEC ? 8 : 4;
  using G = GA2<C>;
  const int N = a.N, B = a.B;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nw = nthr >> 6;            // 4 or 8 waves: 8 when the batch alone cannot fill the SIMDs

  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* ga = reinterpret_cast<double*>(smem_raw);          // N * 20C   gradient of the aggregate, all receivers i
  double* pj = ga + N * G::SIZE;                              // N * PS
  uint8_t* mk = reinterpret_cast<uint8_t*>(pj + N * PS);      // N

  {
    const double* src = a.g_ag + (size_t)b * N * G::SIZE;
    for (int e = tid; e < N * G::SIZE; e += nthr) ga[e] = src[e];
    if (DEC) {
      const size_t plane_p = (size_t)B * N * 4;
      const double* p0 = a.p + (size_t)b * N * 4;
      for (int e = tid; e < N * 4; e += nthr) {
        int j = e >> 2, m = e & 3;
        pj[j * 8 + m] = p0[e];
        pj[j * 8 + 4 + m] = p0[plane_p + e];
      }
    } else {
      const double* p0 = a.p + (size_t)b * N * 4;
      for (int e = tid; e < N * 4; e += nthr) pj[e] = p0[e];
      for (int e = tid; e < N; e += nthr) mk[e] = a.mask[(size_t)b * N + e];
    }
  }
  const int pr = lane & 15, cg = lane >> 4;
  const int tj = pr >> 2, ti = pr & 3;                        // which of the wave's 4 source particles j / slot in the i tile
  double ak[5], bk[5], ck2[5], wf[NG][5], bias[NG][4];
  if (!DEC) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int k = 4 * s + cg;
      ak[s] = a.ra[k];
      bk[s] = a.rb[k];
      const double c = a.rc[k];
      ck2[s] = c * c;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int rr = lane & 15, q = rr >> 2, ch = 4 * g + (rr & 3);
      const double* w = (q >> 1) ? a.w1 : a.w0;
#pragma unroll
      for (int s = 0; s < 5; ++s) wf[g][s] = ch < C ? w[(2 * ch + (q & 1)) * NB + 4 * s + cg] : 0.0;
    }
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int ch = 4 * g + cg;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* bb = (q >> 1) ? a.b1 : a.b0;
      bias[g][q] = ch < C ? (DEC ? bb[ch] : bb[2 * ch + (q & 1)]) : 0.0;
    }
  }
  __syncthreads();

  const size_t pls = (size_t)B * N * C;
  const int ngroups = (N + 3) >> 2;
  for (int rg = wave; rg < ngroups; rg += nw) {
    const int j = rg * 4 + tj;
    const bool jok = j < N;
    const int jj = jok ? j : N - 1;
    double pme[PS];
#pragma unroll
    for (int m = 0; m < PS; ++m) pme[m] = pj[jj * PS + m];
    const bool mj = DEC ? false : (mk[jj] != 0);

    cx<double> Gs[NG], Gv[NG][4], Gq[4];
    cx<double> sj[NG], vtj[NG][4];                             // own node features (decoder position gradient)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      Gs[g] = {0, 0};
#pragma unroll
      for (int m = 0; m < 4; ++m) Gv[g][m] = {0, 0};
      if (DEC) {
        const int ch = 4 * g + cg;
        const size_t e = ((size_t)b * N + jj) * C + (ch < C ? ch : 0);
        sj[g] = {a.s_in[e], a.s_in[pls + e]};
        cx<double> v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = {a.v_in[e * 4 + m], a.v_in[pls * 4 + e * 4 + m]};
        metric_perm(v, vtj[g]);
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) Gq[m] = {0, 0};

    for (int i0 = 0; i0 < N; i0 += 4) {
      const int i = i0 + ti;
      const bool ok = jok && i < N;
      const int ii = i < N ? i : N - 1;
      const double* pii = pj + ii * PS;
      cx<double> q[4];
      v4d R[NG];
      double qd0 = 0.0, qd3 = 0.0, qa = 0.0, qb = 0.0;       // encoder: q = [d0, a - ib, d3, -a - ib]
      if (DEC) {
#pragma unroll
        for (int m = 0; m < 4; ++m) q[m] = {pii[m] - pme[m], pii[4 + m] - pme[4 + m]};
#pragma unroll
        for (int g = 0; g < NG; ++g) R[g] = v4d{bias[g][0], bias[g][1], bias[g][2], bias[g][3]};
      } else {
        const double d0 = pii[0] - pme[0], d1 = pii[1] - pme[1], d2 = pii[2] - pme[2], d3 = pii[3] - pme[3];
        const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
        const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;
        const double an = fabs(nsq);
        const bool on = ok && mj && (mk[ii] != 0) && (nsq != 0.0);
        const double h = rsqrt2<double>();
        q[0] = {d0, 0.0};
        q[1] = {d1 * h, -d2 * h};
        q[2] = {d3, 0.0};
        q[3] = {-d1 * h, -d2 * h};
        qd0 = d0;  qd3 = d3;  qa = d1 * h;  qb = d2 * h;
        double beta[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const double u = (1.0 + ck2[s] * an) + 1e-16;
          const double bv = __builtin_fma(bk[s], fast_rcp(u), ak[s]);
          beta[s] = on ? bv : 0.0;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          R[g] = v4d{bias[g][0], bias[g][1], bias[g][2], bias[g][3]};
#pragma unroll
          for (int s = 0; s < 5; ++s) R[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[g][s], beta[s], R[g], 0, 0, 0);
        }
      }
      if (ok) {
        const double* gi = ga + ii * G::SIZE;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int ch = 4 * g + cg;
          if (ch < C) {
            const cx<double> R0 = {R[g][0], R[g][1]}, R1 = {R[g][2], R[g][3]};
            const cx<double> e0 = {R0.r - R0.i, R0.r + R0.i};
            const cx<double> gA3 = {0.5 * gi[G::A3 + 2 * ch], 0.5 * gi[G::A3 + 2 * ch + 1]};
            const cx<double> gA4 = {gi[G::A4 + 2 * ch], gi[G::A4 + 2 * ch + 1]};
            cx<double> gA1[4], gA2[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              gA1[m] = {gi[G::A1 + (ch * 4 + m) * 2], gi[G::A1 + (ch * 4 + m) * 2 + 1]};
              gA2[m] = {gi[G::A2 + (ch * 4 + m) * 2], gi[G::A2 + (ch * 4 + m) * 2 + 1]};
            }
            cfmac(Gs[g], gA4, e0);
            if (!DEC) {
              // the edge e1[m] = R1 q[m] enters through P2 = sum_m gA2[m] conj(q[m]) and Z = gA3 conj(R1) only; with real
              // momenta q = [d0, a - ib, d3, -a - ib] (level_bwd3.hip)
              const cx<double> dg = {gA2[1].r - gA2[3].r, gA2[1].i - gA2[3].i}, sg = {gA2[1].r + gA2[3].r, gA2[1].i + gA2[3].i};
              cx<double> P2;
              P2.r = __builtin_fma(gA2[0].r, qd0, __builtin_fma(gA2[2].r, qd3, __builtin_fma(qa, dg.r, -qb * sg.i)));
              P2.i = __builtin_fma(gA2[0].i, qd0, __builtin_fma(gA2[2].i, qd3, __builtin_fma(qa, dg.i, qb * sg.r)));
              cfmac(Gs[g], P2, R1);
              const cx<double> Z = cmulc(gA3, R1);
#pragma unroll
              for (int m = 0; m < 4; ++m) cfmac(Gv[g][m], gA1[m], e0);
              Gv[g][0].r = __builtin_fma(Z.r, qd0, Gv[g][0].r);   Gv[g][0].i = __builtin_fma(Z.i, qd0, Gv[g][0].i);
              Gv[g][2].r = __builtin_fma(-Z.r, qd3, Gv[g][2].r);  Gv[g][2].i = __builtin_fma(-Z.i, qd3, Gv[g][2].i);
              const double aZr = qa * Z.r, aZi = qa * Z.i, bZr = qb * Z.r, bZi = qb * Z.i;
              Gv[g][1].r -= aZr + bZi;  Gv[g][1].i += bZr - aZi;   // Z (-a + ib)
              Gv[g][3].r += aZr - bZi;  Gv[g][3].i += aZi + bZr;   // Z ( a + ib)
            } else {
              cx<double> e1[4], e1t[4];
#pragma unroll
              for (int m = 0; m < 4; ++m) e1[m] = cmul(R1, q[m]);
              metric_perm(e1, e1t);
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                cfmac(Gv[g][m], gA1[m], e0);
                cfmac(Gv[g][m], gA3, e1t[m]);
                cfmac(Gs[g], gA2[m], e1[m]);
                cx<double> ge1 = cmulc(gA2[m], sj[g]);
                cfmac(ge1, gA3, vtj[g][m]);
                cfmac(Gq[m], ge1, R1);
              }
            }
          }
        }
      }
    }

    // combine the 4 i-slots (quad lanes); the ti == 0 lane of each (j, channel) adds into the node gradient
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int ch = 4 * g + cg;
      const double sr = quad_sum(Gs[g].r), si = quad_sum(Gs[g].i);
      const size_t e = ((size_t)b * N + jj) * C + ch;
      const bool wr = jok && ti == 0 && ch < C;
      if (wr) {
        a.g_s_in[e] += sr;
        a.g_s_in[pls + e] += si;
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double vr = quad_sum(Gv[g][m].r), vi = quad_sum(Gv[g][m].i);
        if (wr) {
          a.g_v_in[e * 4 + m] += vr;
          a.g_v_in[pls * 4 + e * 4 + m] += vi;
        }
      }
    }
    if (DEC) {   // position gradient: also sum over the channel lanes (lane bits 4, 5)
      const size_t plp = (size_t)B * N * 4;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double qr = quad_sum(Gq[m].r), qi = quad_sum(Gq[m].i);
        qr += shfl_xor(qr, 16);  qr += shfl_xor(qr, 32);
        qi += shfl_xor(qi, 16);  qi += shfl_xor(qi, 32);
        if (jok && ti == 0 && cg == 0) {
          a.g_p[((size_t)b * N + jj) * 4 + m] -= qr;
          a.g_p[plp + ((size_t)b * N + jj) * 4 + m] -= qi;
        }
      }
    }
  }
}

// =========================================================================================================
// i-centric pass, encoder: radial-network gradient sums on the matrix cores
// =========================================================================================================
// B-operand columns (3 N-tiles of 16):  [0,16) X1[k=0..15] | [16,32) X2[k=0..15] | 32..35 X1[16..19], 36..39 X2[16..19], 40 on, 41 one
template <int C>
__global__ __launch_bounds__(2 * BLOCK) void level_bwd_rad2_kernel(LevelBwdArgs<double> a) {
  constexpr int NG = (C + 3) / 4;
  constexpr int NS = node_stride(C);
  using G = GA2<C>;
  constexpr int TS = 18;                                   // padded row stride of the 16 x 16 transpose tiles (scalars)
  const int N = a.N, B = a.B;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nw = blockDim.x >> 6;                            // 4 or 8 waves

  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* nd = reinterpret_cast<double*>(smem_raw);          // N * NS     source node features
  double* pj = nd + ((N * NS + 1) & ~1);                     // N * 4
  double* tr = pj + N * 4;                                   // 4 waves * (NG + 3) tiles * 16 * TS
  double* red = tr;                                          // 4 waves * 64 lanes * NG * 12, aliases the transpose tiles (used after the sweep)
  uint8_t* mk = reinterpret_cast<uint8_t*>(tr + nw * ((NG + 3) * 16 * TS > 64 * NG * 12 ? (NG + 3) * 16 * TS : 64 * NG * 12));

  load_jet<double, C, false>(a.s_in, a.v_in, a.p, a.mask, B, N, b, nd, pj, mk);
  const int pr = lane & 15, cg = lane >> 4;
  const int ti = pr >> 2, tj = pr & 3;
  double ck2[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const double c = a.rc[4 * s + cg];
    ck2[s] = c * c;
  }
  __syncthreads();

  double* trw = tr + wave * (NG + 3) * 16 * TS;              // this wave's transpose tiles: NG x G, 3 x X
  v4d T[NG][3];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < 3; ++t) T[g][t] = v4d{0, 0, 0, 0};

  const int ngroups = (N + 3) >> 2;
  for (int rg = wave; rg < ngroups; rg += nw) {
    const int i = rg * 4 + ti;
    const bool iok = i < N;
    const int ii = iok ? i : N - 1;
    const double pi0 = pj[ii * 4], pi1 = pj[ii * 4 + 1], pi2 = pj[ii * 4 + 2], pi3 = pj[ii * 4 + 3];
    const bool mi = mk[ii] != 0;
    // gradient of the aggregate of this lane's receiver i and channel(s): registers for the whole sweep
    cx<double> rA1[NG][4], rA2[NG][4], rA3[NG], rA4[NG];
    cx<double> dA2[NG], sA2[NG];                             // gA2[1] - gA2[3], gA2[1] + gA2[3]
    {
      const double* gi = a.g_ag + ((size_t)b * N + ii) * G::SIZE;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ch = 4 * g + cg;
        const bool live = iok && ch < C;
        const int cs = ch < C ? ch : 0;
        rA3[g] = live ? cx<double>{0.5 * gi[G::A3 + 2 * cs], 0.5 * gi[G::A3 + 2 * cs + 1]} : cx<double>{0, 0};
        rA4[g] = live ? cx<double>{gi[G::A4 + 2 * cs], gi[G::A4 + 2 * cs + 1]} : cx<double>{0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          rA1[g][m] = live ? cx<double>{gi[G::A1 + (cs * 4 + m) * 2], gi[G::A1 + (cs * 4 + m) * 2 + 1]} : cx<double>{0, 0};
          rA2[g][m] = live ? cx<double>{gi[G::A2 + (cs * 4 + m) * 2], gi[G::A2 + (cs * 4 + m) * 2 + 1]} : cx<double>{0, 0};
        }
        dA2[g] = {rA2[g][1].r - rA2[g][3].r, rA2[g][1].i - rA2[g][3].i};
        sA2[g] = {rA2[g][1].r + rA2[g][3].r, rA2[g][1].i + rA2[g][3].i};
      }
    }
    for (int j0 = 0; j0 < N; j0 += 4) {
      const int j = j0 + tj;
      const bool ok = iok && j < N;
      const int jj = j < N ? j : N - 1;
      const double* pjj = pj + jj * 4;
      const double d0 = pi0 - pjj[0], d1 = pi1 - pjj[1], d2 = pi2 - pjj[2], d3 = pi3 - pjj[3];
      const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
      const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;
      const double an = fabs(nsq);
      const bool on = ok && mi && (mk[jj] != 0) && (nsq != 0.0);
      const double h = rsqrt2<double>();
      cx<double> q[4];
      q[0] = {d0, 0.0};
      q[1] = {d1 * h, -d2 * h};
      q[2] = {d3, 0.0};
      q[3] = {-d1 * h, -d2 * h};

      // ---- B operand source: this lane's pair (row pr), columns k = 4s + cg ----------------------------
      double* xb = trw + NG * 16 * TS;                         // three 16 x 16 tiles [pair][col]
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const double rho = on ? fast_rcp((1.0 + ck2[s] * an) + 1e-16) : 0.0;
        const double x2 = an * rho * rho;
        if (s < 4) {
          xb[pr * TS + 4 * s + cg] = rho;                      // tile 0: X1[k], k = 4s + cg < 16
          xb[16 * TS + pr * TS + 4 * s + cg] = x2;             // tile 1: X2[k]
        } else {
          xb[32 * TS + pr * TS + cg] = rho;                    // tile 2: cols 0..3 X1[16 + cg]
          xb[32 * TS + pr * TS + 4 + cg] = x2;                 //         cols 4..7 X2[16 + cg]
        }
      }
      // cols 8 (on), 9 (one), 10..15 zero: each of the 4 channel lanes of a pair fills two of them
      xb[32 * TS + pr * TS + 8 + 2 * cg] = cg == 0 ? (on ? 1.0 : 0.0) : 0.0;
      xb[32 * TS + pr * TS + 9 + 2 * cg] = cg == 0 ? (ok ? 1.0 : 0.0) : 0.0;

      // ---- A operand source: dL/d rad of this lane's pair and channel -------------------------------------
      const double* njp = nd + jj * NS;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ch = 4 * g + cg;
        double G0r = 0, G0i = 0, G1r = 0, G1i = 0;
        if (ok && ch < C) {
          const cx<double> s = {njp[ch * 10], njp[ch * 10 + 1]};
          cx<double> v[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) v[m] = {njp[ch * 10 + 2 + m], njp[ch * 10 + 6 + m]};
          cx<double> ge0 = cmulc(rA4[g], s);
#pragma unroll
          for (int m = 0; m < 4; ++m) cfmac(ge0, rA1[g][m], v[m]);
          // G_R1 = sum_m ge1[m] conj(q[m]) = conj(s_j) P2 + gA3 conj(V), P2 = sum_m gA2[m] conj(q[m]), V = <v_j, q>, with the
          // real-momentum structure q = [d0, a - ib, d3, -a - ib] (level_bwd3.hip)
          const double qa = d1 * h, qb = d2 * h;
          cx<double> P2, V;
          P2.r = __builtin_fma(rA2[g][0].r, d0, __builtin_fma(rA2[g][2].r, d3, __builtin_fma(qa, dA2[g].r, -qb * sA2[g].i)));
          P2.i = __builtin_fma(rA2[g][0].i, d0, __builtin_fma(rA2[g][2].i, d3, __builtin_fma(qa, dA2[g].i, qb * sA2[g].r)));
          const cx<double> dv = {v[3].r - v[1].r, v[3].i - v[1].i}, sv = {v[1].r + v[3].r, v[1].i + v[3].i};
          V.r = __builtin_fma(v[0].r, d0, __builtin_fma(-v[2].r, d3, __builtin_fma(qa, dv.r, qb * sv.i)));
          V.i = __builtin_fma(v[0].i, d0, __builtin_fma(-v[2].i, d3, __builtin_fma(qa, dv.i, -qb * sv.r)));
          cx<double> gR1 = cmulc(P2, s);
          cfmac(gR1, rA3[g], V);
          G0r = ge0.r + ge0.i;  G0i = ge0.i - ge0.r;          // e0 = R0 (1+i)  ->  G_R0 = G_e0 (1-i)
          G1r = gR1.r;  G1i = gR1.i;
        }
        double* ta = trw + g * 16 * TS;                       // [pair][r' = cg + 4q]
        ta[pr * TS + cg] = G0r;
        ta[pr * TS + 4 + cg] = G0i;
        ta[pr * TS + 8 + cg] = G1r;
        ta[pr * TS + 12 + cg] = G1i;
      }
      wave_sync();
      // ---- T[r'][col] += sum_pairs G[pair][r'] X[pair][col] : A[i = r' = lane&15][k = pair], B[k = pair][j = col] ----
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int prow = 4 * s + cg;                          // pair index held by this lane for k-step s
        double bv[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bv[t] = xb[t * 16 * TS + prow * TS + pr];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const double av = trw[g * 16 * TS + prow * TS + pr];
#pragma unroll
          for (int t = 0; t < 3; ++t) T[g][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[t], T[g][t], 0, 0, 0);
        }
      }
      wave_sync();
    }
  }

  // ---- cross-wave reduction; D layout: lane holds T[r' = (lane>>4) + 4q][col = lane & 15] -----------------
  __syncthreads();                                           // every wave is done with its transpose tiles (aliased by red)
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
  if (wave == 0) {
    constexpr int R = 4 * C;
    double* part = a.part_rad + (size_t)blockIdx.x * rad_partial_size(C, false);
    const int col = lane & 15;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int ch = 4 * g + cg;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = (g * 3 + t) * 4 + q;
          double v = (red[(size_t)(0 * 64 + lane) * NG * 12 + e] + red[(size_t)(1 * 64 + lane) * NG * 12 + e]) +
                     (red[(size_t)(2 * 64 + lane) * NG * 12 + e] + red[(size_t)(3 * 64 + lane) * NG * 12 + e]);
          if (nw == 8)
            v += (red[(size_t)(4 * 64 + lane) * NG * 12 + e] + red[(size_t)(5 * 64 + lane) * NG * 12 + e]) +
                 (red[(size_t)(6 * 64 + lane) * NG * 12 + e] + red[(size_t)(7 * 64 + lane) * NG * 12 + e]);
          if (ch >= C) continue;
          const int r = (q >> 1) * 2 * C + 2 * ch + (q & 1);       // row of the partial layout: lin*2C + 2c + z
          if (t == 0) part[r * NB + col] = v;                      // T1[r][k = col]
          else if (t == 1) part[R * NB + r * NB + col] = v;        // T2[r][k = col]
          else {
            if (col < 4) part[r * NB + 16 + col] = v;
            else if (col < 8) part[R * NB + r * NB + 16 + (col - 4)] = v;
            else if (col == 8) part[2 * R * NB + r] = v;           // S
            else if (col == 9) part[2 * R * NB + R + r] = v;       // dB
          }
        }
    }
  }
}

// =========================================================================================================
// encoder, jets too large for level_bwd3: BOTH passes in ONE pair sweep (round 4; was nodes2 + rad2: two sweeps, each with its own
// geometry, basis functions and radial Linear on the matrix cores: 190 + 298 us per level at N = 150)
// =========================================================================================================
// j-centric like level_bwd3's phase 2, whose per-tile arithmetic this is: a wave owns 4 source particles j and sweeps the receivers i
// in tiles of 4; per pair (a) g_node_j += g_ag_i (x) conj(edge_ij) and (b) dL/d rad of the pair -> 16 x 16 LDS transposes -> the
// T1 | T2 | S | dB GEMM on the matrix cores.  What level_bwd3 keeps in LDS for the whole jet does not fit here (N = 150, C = 4: the
// gradient of the aggregate alone is 96 KB, beside 9 KB of transpose tiles per wave), so the receivers come in CHUNKS of `ichunk`
// particles: the chunk's g_ag rows (written by level_bwd_mix_kernel) are staged, every wave sweeps its row groups over the chunk
// and adds its share into the node gradient (which already holds the direct + power part from level_bwd_mix_kernel), next chunk.
// The source particle's own features come from global memory (10 reals per lane and row group); the radial sums live in the
// matrix-core accumulators across all chunks and leave as ONE partial row per jet, like level_bwd_rad2's.
template <int C, int NWV, bool SYM>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(2))) void level_bwd_sweep_enc_kernel(LevelBwdArgs<double> a, int ichunk) {
  constexpr bool sym = SYM;
  constexpr int NG = (C + 3) / 4;
  using G = GA2<C>;
  constexpr int TS = 18, BLK = 64 * NWV;
  constexpr int TRW = (NG + 3) * 16 * TS > 64 * NG * 12 ? (NG + 3) * 16 * TS : 64 * NG * 12;   // per wave: transpose tiles, later the reduction rows
  const int N = a.N, B = a.B;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* ga = reinterpret_cast<double*>(smem_raw);          // ichunk * 20C   gradient of the aggregate, receivers of the current chunk
  double* xn = ga + (size_t)ichunk * G::SIZE;                // ichunk * 10C   (sym) their node features [i][c][s2 | v8]
  double* pj = xn + (sym ? (size_t)ichunk * 10 * C : 0);     // N * 4
  double* tr = pj + N * 4;                                   // NWV * TRW
  double* gaj = tr + NWV * TRW;                              // (sym) NWV * 4 * 20C: g_ag rows of each wave's own 4 particles
  uint8_t* mk = reinterpret_cast<uint8_t*>(gaj + (sym ? NWV * 4 * G::SIZE : 0));  // N

  {
    const double* p0 = a.p + (size_t)b * N * 4;
    for (int e = tid; e < N * 4; e += BLK) pj[e] = p0[e];
    for (int e = tid; e < N; e += BLK) mk[e] = a.mask[(size_t)b * N + e];
  }
  const int pr = lane & 15, cg = lane >> 4;
  const int tj = pr >> 2, ti = pr & 3;                      // which of the wave's 4 source particles j / slot in the i tile
  double ak[5], bk[5], ck2[5], wf[NG][5], bias[NG][4];
  // sym: the 15 bell constants of the lane's five basis functions live in LDS (one row per lane group, read per tile) -- with them
  // in registers the second pass of the tiles below the diagonal spills into the hot loop
  __shared__ double rkt[4][16];
  const double* rkl = rkt[cg];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int k = 4 * s + cg;
    ak[s] = a.ra[k];
    bk[s] = a.rb[k];
    const double c = a.rc[k];
    ck2[s] = c * c;
    if (sym && pr == 0) {
      rkt[cg][s] = ak[s];
      rkt[cg][5 + s] = bk[s];
      rkt[cg][10 + s] = ck2[s];
    }
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int rr = lane & 15, q = rr >> 2, ch = 4 * g + (rr & 3);
    const double* w = (q >> 1) ? a.w1 : a.w0;
#pragma unroll
    for (int s = 0; s < 5; ++s) wf[g][s] = ch < C ? w[(2 * ch + (q & 1)) * NB + 4 * s + cg] : 0.0;
    const int chl = 4 * g + cg;
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      const double* bb = (q2 >> 1) ? a.b1 : a.b0;
      bias[g][q2] = chl < C ? bb[2 * chl + (q2 & 1)] : 0.0;
    }
  }

  double* trw = tr + wave * TRW;
  v4d T[NG][3];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < 3; ++t) T[g][t] = v4d{0, 0, 0, 0};
  const size_t pls = (size_t)B * N * C;
  const int ngroups = (N + 3) >> 2;
  // Which source groups this wave owns.  sym (R(i, j) = R(j, i): the radial-parameter GEMM runs once per UNORDERED tile, as in
  // level_bwd3.hip): a group's cost grows with its index -- tiles with receiver group I < J carry both directions' radial
  // gradient, tiles with I > J none -- so the groups are dealt by longest-processing-time-first on that cost model (every wave runs
  // the same few hundred scalar steps); else round robin.
  unsigned long long mine = 0;
  if (sym) {
    int load[NWV];
#pragma unroll
    for (int w = 0; w < NWV; ++w) load[w] = 0;
    for (int g = ngroups - 1; g >= 0; --g) {
      int best = 0, bestv = load[0];
#pragma unroll
      for (int w = 1; w < NWV; ++w)
        if (load[w] < bestv) { bestv = load[w]; best = w; }
#pragma unroll
      for (int w = 0; w < NWV; ++w)
        if (w == best) load[w] += 27 * g + 11 * (ngroups - 1 - g) + 23;      // ~cycles / 100 of a full / light / diagonal tile
      if (best == wave) mine |= 1ull << g;
    }
  }
  STAMP(0);

  for (int ilo = 0; ilo < N; ilo += ichunk) {
    const int ihi = min(N, ilo + ichunk);
    __syncthreads();                                         // (first chunk: pj / mk staged; later: the previous chunk is swept)
    STAMP(ilo ? 10 : 1);
    {
      const double* src = a.g_ag + ((size_t)b * N + ilo) * G::SIZE;
      for (int e = tid; e < (ihi - ilo) * G::SIZE; e += BLK) ga[e] = src[e];
      if (sym) {
        for (int e = tid; e < (ihi - ilo) * C; e += BLK) {
          const size_t ge = ((size_t)b * N + ilo) * C + e;
          double* x = xn + (size_t)e * 10;
          x[0] = a.s_in[ge];
          x[1] = a.s_in[pls + ge];
#pragma unroll
          for (int m = 0; m < 4; ++m) { x[2 + m] = a.v_in[ge * 4 + m]; x[6 + m] = a.v_in[pls * 4 + ge * 4 + m]; }
        }
      }
    }
    __syncthreads();
    STAMP(ilo ? 11 : 2);
    for (int rg = 0; rg < ngroups; ++rg) {
      // (ordered form: round robin, computed -- not through the 64-bit mask, which only covers the 64 groups of sym's N <= 256)
      if (sym ? !((mine >> rg) & 1) : (rg % NWV != wave)) continue;
      if (rg == 0) STAMP(ilo ? 12 : 3);
      if (rg == NWV) STAMP(ilo ? 15 : 6);
      double* gajw = gaj + wave * 4 * G::SIZE;
      if (sym) {                                             // the upstream gradient of the own particles' aggregates (reverse edges)
        for (int e = lane; e < 4 * G::SIZE; e += 64) {
          const int r = e / G::SIZE, jr = min(rg * 4 + r, N - 1);
          gajw[e] = a.g_ag[((size_t)b * N + jr) * G::SIZE + (e - r * G::SIZE)];
        }
        wave_sync();
      }
      const int j = rg * 4 + tj;
      const bool jok = j < N;
      const int jj = jok ? j : N - 1;
      double pme[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) pme[m] = pj[jj * 4 + m];
      const bool mj = mk[jj] != 0;
      cx<double> Gs[NG], Gv[NG][4];
      cx<double> sj[NG], vj[NG][4], dvj[NG], svj[NG];       // own (source) node features; v_j[3] - v_j[1], v_j[1] + v_j[3]
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        Gs[g] = {0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) Gv[g][m] = {0, 0};
        const int ch = 4 * g + cg;
        const size_t e = ((size_t)b * N + jj) * C + (ch < C ? ch : 0);
        sj[g] = {a.s_in[e], a.s_in[pls + e]};
#pragma unroll
        for (int m = 0; m < 4; ++m) vj[g][m] = {a.v_in[e * 4 + m], a.v_in[pls * 4 + e * 4 + m]};
        dvj[g] = {vj[g][3].r - vj[g][1].r, vj[g][3].i - vj[g][1].i};
        svj[g] = {vj[g][1].r + vj[g][3].r, vj[g][1].i + vj[g][3].i};
      }

      if (rg == 0) STAMP(ilo ? 13 : 4);
      for (int i0 = ilo; i0 < ihi; i0 += 4) {
        // 2: receiver group below the source group (the tile also carries the reverse edges' radial gradient), 1: diagonal tile,
        // 0: above (the owner of the other group adds this tile's radial gradient to its own)
        const int kind = sym ? ((i0 >> 2) < rg ? 2 : ((i0 >> 2) == rg ? 1 : 0)) : 1;
        const int i = i0 + ti;
        const bool ok = jok && i < ihi;
        const int ii = i < ihi ? i : ihi - 1;
        const double* pii = pj + ii * 4;
        v4d R[NG];
        double rho[5];
        const double d0 = pii[0] - pme[0], d1 = pii[1] - pme[1], d2 = pii[2] - pme[2], d3 = pii[3] - pme[3];
        const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
        const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;
        const double an = fabs(nsq);
        const bool on = ok && mj && (mk[ii] != 0) && (nsq != 0.0);
        const double h = rsqrt2<double>();
        const double qd0 = d0, qd3 = d3, qa = d1 * h, qb = d2 * h;   // q = [d0, a - ib, d3, -a - ib] (real momenta)
        double beta[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 5; ++s) rho[s] = 0.0;
        if (on) {                                            // (EXEC-masked block: no per-value selects)
#pragma unroll
          for (int s = 0; s < 5; ++s) beta[s] = 1.0 + (sym ? rkl[10 + s] : ck2[s]) * an;      // (+ 1e-16: absorbed, the sum is >= 1)
          rcp5(beta, rho);
#pragma unroll
          for (int s = 0; s < 5; ++s) beta[s] = __builtin_fma(sym ? rkl[5 + s] : bk[s], rho[s], sym ? rkl[s] : ak[s]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          R[g] = v4d{bias[g][0], bias[g][1], bias[g][2], bias[g][3]};
#pragma unroll
          for (int s = 0; s < 5; ++s) R[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[g][s], beta[s], R[g], 0, 0, 0);
        }
        // B-operand source of the radial GEMM: this lane's pair (row pr), columns k = 4s + cg
        double* xb = trw + NG * 16 * TS;
        if (kind != 0) {
#pragma unroll
          for (int s = 0; s < 5; ++s) {
            const double x2 = an * rho[s] * rho[s];
            if (s < 4) {
              xb[pr * TS + 4 * s + cg] = rho[s];
              xb[16 * TS + pr * TS + 4 * s + cg] = x2;
            } else {
              xb[32 * TS + pr * TS + cg] = rho[s];
              xb[32 * TS + pr * TS + 4 + cg] = x2;
            }
          }
          xb[32 * TS + pr * TS + 8 + 2 * cg] = cg == 0 ? (on ? 1.0 : 0.0) : 0.0;
          xb[32 * TS + pr * TS + 9 + 2 * cg] = cg == 0 ? (ok ? 1.0 : 0.0) : 0.0;
        }

        const double* gi = ga + (size_t)(ii - ilo) * G::SIZE;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int ch = 4 * g + cg;
          double G0r = 0, G0i = 0, G1r = 0, G1i = 0;
          if (ok && ch < C) {
            const cx<double> R0 = {R[g][0], R[g][1]}, R1 = {R[g][2], R[g][3]};
            const cx<double> e0 = {R0.r - R0.i, R0.r + R0.i};
            const cx<double> gA3 = {0.5 * gi[G::A3 + 2 * ch], 0.5 * gi[G::A3 + 2 * ch + 1]};
            const cx<double> gA4 = {gi[G::A4 + 2 * ch], gi[G::A4 + 2 * ch + 1]};
            cx<double> gA1[4], gA2[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              gA1[m] = {gi[G::A1 + (ch * 4 + m) * 2], gi[G::A1 + (ch * 4 + m) * 2 + 1]};
              gA2[m] = {gi[G::A2 + (ch * 4 + m) * 2], gi[G::A2 + (ch * 4 + m) * 2 + 1]};
            }
            cfmac(Gs[g], gA4, e0);
            cx<double> ge0 = cmulc(gA4, sj[g]);
            // the edge e1[m] = R1 q[m] enters through P2 = sum_m gA2[m] conj(q[m]), V = <v_j, q> and Z = gA3 conj(R1) only
            // (level_bwd3.hip, phase 2)
            const cx<double> dg = {gA2[1].r - gA2[3].r, gA2[1].i - gA2[3].i}, sg = {gA2[1].r + gA2[3].r, gA2[1].i + gA2[3].i};
            cx<double> P2;
            P2.r = __builtin_fma(gA2[0].r, qd0, __builtin_fma(gA2[2].r, qd3, __builtin_fma(qa, dg.r, -qb * sg.i)));
            P2.i = __builtin_fma(gA2[0].i, qd0, __builtin_fma(gA2[2].i, qd3, __builtin_fma(qa, dg.i, qb * sg.r)));
            cfmac(Gs[g], P2, R1);
            const cx<double> Z = cmulc(gA3, R1);
#pragma unroll
            for (int m = 0; m < 4; ++m) cfmac(Gv[g][m], gA1[m], e0);
            Gv[g][0].r = __builtin_fma(Z.r, qd0, Gv[g][0].r);   Gv[g][0].i = __builtin_fma(Z.i, qd0, Gv[g][0].i);
            Gv[g][2].r = __builtin_fma(-Z.r, qd3, Gv[g][2].r);  Gv[g][2].i = __builtin_fma(-Z.i, qd3, Gv[g][2].i);
            const double aZr = qa * Z.r, aZi = qa * Z.i, bZr = qb * Z.r, bZi = qb * Z.i;
            Gv[g][1].r -= aZr + bZi;  Gv[g][1].i += bZr - aZi;   // Z (-a + ib)
            Gv[g][3].r += aZr - bZi;  Gv[g][3].i += aZi + bZr;   // Z ( a + ib)
            cx<double> gR1 = {0, 0};
            if (kind != 0) {                                   // gradient w.r.t. the radial values of this pair
#pragma unroll
              for (int m = 0; m < 4; ++m) cfmac(ge0, gA1[m], vj[g][m]);
              cx<double> V;
              V.r = __builtin_fma(vj[g][0].r, qd0, __builtin_fma(-vj[g][2].r, qd3, __builtin_fma(qa, dvj[g].r, qb * svj[g].i)));
              V.i = __builtin_fma(vj[g][0].i, qd0, __builtin_fma(-vj[g][2].i, qd3, __builtin_fma(qa, dvj[g].i, -qb * svj[g].r)));
              gR1 = cmulc(P2, sj[g]);
              cfmac(gR1, gA3, V);
            }
            if (kind == 2) {
              // the reverse edge (receiver j, source i, momentum difference -q): level_bwd3.hip, SYM
              const double* gj = gajw + tj * G::SIZE;
              const double* ni = xn + ((size_t)(ii - ilo) * C + ch) * 10;
              const cx<double> si = {ni[0], ni[1]};
              cx<double> vi[4];
#pragma unroll
              for (int m = 0; m < 4; ++m) vi[m] = {ni[2 + m], ni[6 + m]};
              const cx<double> hA3 = {0.5 * gj[G::A3 + 2 * ch], 0.5 * gj[G::A3 + 2 * ch + 1]};
              const cx<double> hA4 = {gj[G::A4 + 2 * ch], gj[G::A4 + 2 * ch + 1]};
              cx<double> he0 = cmulc(hA4, si);
              cx<double> hA2[4];
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                const cx<double> hA1 = {gj[G::A1 + (ch * 4 + m) * 2], gj[G::A1 + (ch * 4 + m) * 2 + 1]};
                hA2[m] = {gj[G::A2 + (ch * 4 + m) * 2], gj[G::A2 + (ch * 4 + m) * 2 + 1]};
                cfmac(he0, hA1, vi[m]);
              }
              const cx<double> hd = {hA2[1].r - hA2[3].r, hA2[1].i - hA2[3].i}, hs = {hA2[1].r + hA2[3].r, hA2[1].i + hA2[3].i};
              cx<double> Q2;
              Q2.r = __builtin_fma(hA2[0].r, qd0, __builtin_fma(hA2[2].r, qd3, __builtin_fma(qa, hd.r, -qb * hs.i)));
              Q2.i = __builtin_fma(hA2[0].i, qd0, __builtin_fma(hA2[2].i, qd3, __builtin_fma(qa, hd.i, qb * hs.r)));
              const cx<double> dvi = {vi[3].r - vi[1].r, vi[3].i - vi[1].i}, svi = {vi[1].r + vi[3].r, vi[1].i + vi[3].i};
              cx<double> W;
              W.r = __builtin_fma(vi[0].r, qd0, __builtin_fma(-vi[2].r, qd3, __builtin_fma(qa, dvi.r, qb * svi.i)));
              W.i = __builtin_fma(vi[0].i, qd0, __builtin_fma(-vi[2].i, qd3, __builtin_fma(qa, dvi.i, -qb * svi.r)));
              cx<double> hR1 = cmulc(Q2, si);
              cfmac(hR1, hA3, W);
              ge0.r += he0.r;  ge0.i += he0.i;
              gR1.r -= hR1.r;  gR1.i -= hR1.i;
            }
            G0r = ge0.r + ge0.i;  G0i = ge0.i - ge0.r;        // e0 = R0 (1+i)  ->  G_R0 = G_e0 (1-i)
            G1r = gR1.r;  G1i = gR1.i;
          }
          if (kind != 0) {
            double* ta = trw + g * 16 * TS;                   // [pair][r' = cg + 4q]
            ta[pr * TS + cg] = G0r;
            ta[pr * TS + 4 + cg] = G0i;
            ta[pr * TS + 8 + cg] = G1r;
            ta[pr * TS + 12 + cg] = G1i;
          }
        }
        if (kind != 0) {
          wave_sync();
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int prow = 4 * s + cg;
            double bv[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) bv[t] = xb[t * 16 * TS + prow * TS + pr];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              const double av = trw[g * 16 * TS + prow * TS + pr];
#pragma unroll
              for (int t = 0; t < 3; ++t) T[g][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[t], T[g][t], 0, 0, 0);
            }
          }
          wave_sync();
        }
      }

      if (rg == 0) STAMP(ilo ? 14 : 5);
      // this chunk's share of the node gradient of the wave's 4 particles (quad sum over the i slots), added to what
      // level_bwd_mix_kernel (direct + power part) and the earlier chunks left there.  The ten old values are all requested
      // before the first store: written as ten "+=" the loads and stores alternate (the pointers may alias for all the compiler
      // knows) and the flush is ten memory round trips -- measured: a third of the kernel at N = 150.
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ch = 4 * g + cg;
        const size_t e = ((size_t)b * N + jj) * C + (ch < C ? ch : 0);
        const bool wr = jok && ti == 0 && ch < C;
        double old[10];
        if (wr) {
          old[0] = a.g_s_in[e];
          old[1] = a.g_s_in[pls + e];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            old[2 + m] = a.g_v_in[e * 4 + m];
            old[6 + m] = a.g_v_in[pls * 4 + e * 4 + m];
          }
        }
        double add[10];
        add[0] = quad_sum(Gs[g].r);
        add[1] = quad_sum(Gs[g].i);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          add[2 + m] = quad_sum(Gv[g][m].r);
          add[6 + m] = quad_sum(Gv[g][m].i);
        }
        if (wr) {
          a.g_s_in[e] = old[0] + add[0];
          a.g_s_in[pls + e] = old[1] + add[1];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            a.g_v_in[e * 4 + m] = old[2 + m] + add[2 + m];
            a.g_v_in[pls * 4 + e * 4 + m] = old[6 + m] + add[6 + m];
          }
        }
      }
    }
  }

  // ---- radial partial row of this jet: cross-wave sum of the accumulators (D layout: T[r' = (lane>>4) + 4q][col = lane & 15]) ----
  STAMP(20);
  __syncthreads();                                           // every wave is done with its transpose tiles (aliased by the rows below)
  STAMP(21);
  {
    double* mine = tr + (size_t)(wave * 64 + lane) * NG * 12;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) mine[(g * 3 + t) * 4 + q] = T[g][t][q];
  }
  __syncthreads();
  if (wave == 0) {
    constexpr int R = 4 * C;
    double* part = a.part_rad + (size_t)blockIdx.x * rad_partial_size(C, false);
    const int col = lane & 15;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int ch = 4 * g + cg;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = (g * 3 + t) * 4 + q;
          double v = 0.0;
          for (int w = 0; w < NWV; ++w) v += tr[(size_t)(w * 64 + lane) * NG * 12 + e];
          if (ch >= C) continue;
          const int r = (q >> 1) * 2 * C + 2 * ch + (q & 1);       // row of the partial layout: lin*2C + 2c + z
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

template <int C, int NWV>
static int launch_sweep_enc_w(const LevelBwdArgs<double>& a, hipStream_t stream) {
  constexpr int NG = (C + 3) / 4;
  constexpr size_t TRW = (NG + 3) * 16 * 18 > 64 * NG * 12 ? (NG + 3) * 16 * 18 : 64 * NG * 12;
  // sym (level_bwd3.hip: R(i, j) = R(j, i), the radial-parameter GEMM once per unordered tile): the chunk also holds the receivers'
  // node features, every wave the g_ag rows of its own four particles
  // (C <= 4: with two lane groups of channels the second pass does not fit the register file beside the first)
  const bool sym = !(a.flags & LVL_BWD_ORDERED) && C <= 4 && (a.N + 3) / 4 <= 64;     // (group ownership is a 64-bit mask)
  const size_t fixed = sizeof(double) * ((size_t)a.N * 4 + NWV * TRW + (sym ? NWV * 4 * 20 * C : 0)) + a.N + 16,
               row = sizeof(double) * (sym ? 30 : 20) * C;
  // receivers per chunk: the whole jet when it fits, else the largest multiple of 4 that does; chunks of equal size
  // (four waves: half the LDS, so that two workgroups share a CU -- unless not even four receivers fit then)
  // (1 KB below the CU's 160 KB / half of it: the kernel also has a few hundred bytes of static LDS)
  const size_t budget = (NWV == 4 && fixed + 4 * row <= 79 * 1024) ? 79 * 1024 : 159 * 1024;
  LGN_CHECK_ARG(fixed + 4 * row <= budget, "level_bwd_sweep: N=%d C=%d does not fit the LDS", a.N, a.C);
  int ichunk = (a.N + 3) & ~3;
  if (fixed + ichunk * row > budget) {
    const int cap = (int)((budget - fixed) / row) & ~3, nchunks = (a.N + cap - 1) / cap;
    ichunk = (((a.N + nchunks - 1) / nchunks) + 3) & ~3;
  }
  const size_t smem = fixed + (size_t)ichunk * row;
  auto kern = level_bwd_sweep_enc_kernel<C, NWV, false>;
  if constexpr (C <= 4) {
    if (sym) kern = level_bwd_sweep_enc_kernel<C, NWV, true>;
  }
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(64 * NWV), smem, stream, a, ichunk);
  LGN_CHECK_LAUNCH();
  return 0;
}

// encoder only; reads g_ag and adds into g_s_in / g_v_in (both written by level_bwd_mix_kernel), writes ONE radial partial row per jet
int level_bwd_sweep_enc_dispatch(const LevelBwdArgs<double>& a, hipStream_t stream) {
  // 8 waves per jet when the batch has no more jets than the chip has CUs (cfg4), else 4 (two workgroups per CU)
#define LGN_CASE(CC) case CC: return (a.B <= 320 && a.N >= 64) ? launch_sweep_enc_w<CC, 8>(a, stream) : launch_sweep_enc_w<CC, 4>(a, stream);
  switch (a.C) {
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
    default: set_error("level_bwd: C_in=%d unsupported (1..8)", a.C); return -1;
  }
#undef LGN_CASE
}

// ---------------------------------------------------------------------------------------------------------
// One workgroup per jet with 4 waves leaves a SIMD with a single wave when the batch has no more jets than the chip has
// CUs (cfg4: 256 jets of 150 particles = 38 row groups each): run 8 waves per jet then.
static int sweep_wave_factor(int B, int N) { return (B <= 320 && N >= 64) ? 2 : 1; }

template <int C, bool DEC>
static int launch_nodes2(const LevelBwdArgs<double>& a, hipStream_t stream) {
  constexpr int PS = DEC ? 8 : 4;
  const size_t smem = sizeof(double) * ((size_t)a.N * 20 * C + (size_t)a.N * PS) + a.N + 16;
  LGN_CHECK_ARG(smem <= 160 * 1024, "level_bwd_nodes: N=%d C=%d needs %zu B of LDS", a.N, a.C, smem);
  auto kern = level_bwd_nodes2_kernel<C, DEC>;
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  // (no per-wave LDS here and <= 168 VGPRs: 12 waves per jet when the batch leaves SIMDs idle)
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(BLOCK * (sweep_wave_factor(a.B, a.N) == 2 ? 3 : 1)), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

template <int C>
static int launch_rad2(const LevelBwdArgs<double>& a, hipStream_t stream) {
  constexpr int NG = (C + 3) / 4;
  auto bytes = [&](int nw) {
    return sizeof(double) * ((((size_t)a.N * node_stride(C) + 1) & ~size_t(1)) + (size_t)a.N * 4 +
                             (size_t)nw * ((NG + 3) * 16 * 18 > 64 * NG * 12 ? (NG + 3) * 16 * 18 : 64 * NG * 12)) + a.N + 16;
  };
  int f = sweep_wave_factor(a.B, a.N);
  if (f == 2 && bytes(8) > 160 * 1024) f = 1;
  const size_t smem = bytes(4 * f);
  LGN_CHECK_ARG(smem <= 160 * 1024, "level_bwd_rad: N=%d C=%d needs %zu B of LDS", a.N, a.C, smem);
  auto kern = level_bwd_rad2_kernel<C>;
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(BLOCK * f), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

int level_bwd_nodes2_dispatch(const LevelBwdArgs<double>& a, int decoder, hipStream_t stream) {
#define LGN_CASE(CC) case CC: return decoder ? launch_nodes2<CC, true>(a, stream) : launch_nodes2<CC, false>(a, stream);
  switch (a.C) {
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
    default: set_error("level_bwd: C_in=%d unsupported (1..8)", a.C); return -1;
  }
#undef LGN_CASE
}

// encoder only; writes ONE partial row per jet (rows_rad == B)
int level_bwd_rad2_dispatch(const LevelBwdArgs<double>& a, hipStream_t stream) {
#define LGN_CASE(CC) case CC: return launch_rad2<CC>(a, stream);
  switch (a.C) {
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
    default: set_error("level_bwd: C_in=%d unsupported (1..8)", a.C); return -1;
  }
#undef LGN_CASE
}

}  // namespace lgn
