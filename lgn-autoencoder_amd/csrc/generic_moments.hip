// lgn-autoencoder_amd/csrc/generic_moments.hip -- the N^2 part of a message-passing level for ARBITRARY node irreps
// (maxdim = 3: (0,0),(1,1),(2,0),(0,2),(2,2); reference lgn/cg_lib/cg_ops.py:135-298 with aggregate=True).
//
// The reference forms, per pair of irreps (r1 of the node, r2 of the edge), the Kronecker product
// node_j[r1] (x) edge_ij[r2], sums it over neighbours j FIRST and applies the Clebsch-Gordan matrix afterwards
// (cg_ops.py:287-297 then :195-204).  The neighbour sum therefore only ever needs the "moments"
//     U[i][c][q][0]     = sum_j X_j[c][q] * e0_ij[c]          (edge irrep (0,0),  e0 = R0 (1+1i))
//     U[i][c][q][1 + m] = sum_j X_j[c][q] * e1_ij[c][m]       (edge irrep (1,1),  e1 = R1 canonical(p_i - p_j))
// for every component q of the packed node feature vector X (Q = sum of the irrep dimensions).  These kernels
// compute the moments and their backward passes without any Clebsch-Gordan table; the (sparse, table driven)
// CG contraction is O(N) per jet and lives in generic_local.hip.  At maxdim = 2 (Q = 5) this reproduces the fused
// closed-form kernels, which the tests use as a cross-check.
//
// Mapping = level_fwd2 / level_bwd2: wave = 4 particles, tiles of 4 partners, lane = (pair slot, channel in group),
// radial Linear layers on v_mfma_f64_16x16x4_f64; the component axis q is processed in chunks of QC so that the
// accumulators stay in registers (the radial part is recomputed per chunk).
#include <stdlib.h>

#include "level_dev.hpp"
#include "ops.hpp"

namespace lgn {

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {
constexpr int QC = 5;      // components per sweep

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
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// per-lane constants of the radial network in MFMA fragment form (see level_fwd2.hip)
template <int C, bool DEC>
struct RadConst {
  static constexpr int NG = (C + 3) / 4;
  double ak[5], bk[5], ck2[5], wf[NG][5], bias[NG][4];
  __device__ __forceinline__ void load(const GenArgs& a, int lane) {
    const int cg = lane >> 4;
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
  }
};

// geometry + radial values of the ordered pair (i, j) held by this lane; returns R[g] = (R0r, R0i, R1r, R1i)
template <int C, bool DEC>
__device__ __forceinline__ void pair_radial(const RadConst<C, DEC>& rc, const double* pi, const double* pjj, bool ok, bool mi,
                                            bool mj, cx<double> (&q)[4], v4d (&R)[RadConst<C, DEC>::NG], double& an, bool& on) {
  constexpr int NG = RadConst<C, DEC>::NG;
  if (DEC) {
#pragma unroll
    for (int m = 0; m < 4; ++m) q[m] = {pi[m] - pjj[m], pi[4 + m] - pjj[4 + m]};
#pragma unroll
    for (int g = 0; g < NG; ++g) R[g] = v4d{rc.bias[g][0], rc.bias[g][1], rc.bias[g][2], rc.bias[g][3]};
    an = 0.0;
    on = false;
  } else {
    const double d0 = pi[0] - pjj[0], d1 = pi[1] - pjj[1], d2 = pi[2] - pjj[2], d3 = pi[3] - pjj[3];
    const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
    const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;
    an = fabs(nsq);
    on = ok && mi && mj && (nsq != 0.0);
    const double h = rsqrt2<double>();
    q[0] = {d0, 0.0};
    q[1] = {d1 * h, -d2 * h};
    q[2] = {d3, 0.0};
    q[3] = {-d1 * h, -d2 * h};
    double beta[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const double u = (1.0 + rc.ck2[s] * an) + 1e-16;
      const double bv = __builtin_fma(rc.bk[s], fast_rcp(u), rc.ak[s]);
      beta[s] = on ? bv : 0.0;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      R[g] = v4d{rc.bias[g][0], rc.bias[g][1], rc.bias[g][2], rc.bias[g][3]};
#pragma unroll
      for (int s = 0; s < 5; ++s) R[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(rc.wf[g][s], beta[s], R[g], 0, 0, 0);
    }
  }
}

// cooperative load of a jet's packed features X[2][B][N][C][Q] into LDS as xs[(j*C + c)*Q*2 + q*2 + z]
__device__ __forceinline__ void load_packed(const double* __restrict__ X, int B, int N, int C, int Q, int b, double* xs) {
  const size_t plane = (size_t)B * N * C * Q;
  const double* x0 = X + (size_t)b * N * C * Q;
  for (int e = threadIdx.x; e < N * C * Q; e += BLOCK) {
    xs[2 * e] = x0[e];
    xs[2 * e + 1] = x0[plane + e];
  }
}
template <bool DEC>
__device__ __forceinline__ void load_pos(const GenArgs& a, int b, double* pj, uint8_t* mk) {
  const int N = a.N, B = a.B;
  if (DEC) {
    const size_t plane_p = (size_t)B * N * 4;
    const double* p0 = a.p + (size_t)b * N * 4;
    for (int e = threadIdx.x; e < N * 4; e += BLOCK) {
      int j = e >> 2, m = e & 3;
      pj[j * 8 + m] = p0[e];
      pj[j * 8 + 4 + m] = p0[plane_p + e];
    }
  } else {
    const double* p0 = a.p + (size_t)b * N * 4;
    for (int e = threadIdx.x; e < N * 4; e += BLOCK) pj[e] = p0[e];
    for (int e = threadIdx.x; e < N; e += BLOCK) mk[e] = a.mask[(size_t)b * N + e];
  }
}
}  // namespace

// =========================================================================================================
// forward:  U[b][i][c][q][k][z]
// =========================================================================================================
// XL: the jet's packed features are staged in LDS (N C Q 16 bytes); XL = false (round 5: jets whose features do not fit -- 150
// particles at maxdim 3 are 288 KB -- used to be refused) reads them from global memory instead, every source row from L1 / L2.
template <int C, bool DEC, bool XL>
__global__ __launch_bounds__(BLOCK) void moments_fwd_kernel(GenArgs a) {
  constexpr int NG = (C + 3) / 4;
  constexpr int PS = DEC ? 8 : 4;
  const int N = a.N, B = a.B, Q = a.Q;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* xs = reinterpret_cast<double*>(smem_raw);           // N * C * Q * 2 (XL)
  double* pj = xs + (XL ? (size_t)N * C * Q * 2 : 0);         // N * PS
  uint8_t* mk = reinterpret_cast<uint8_t*>(pj + N * PS);
  const size_t xplane = (size_t)B * N * C * Q;
  if (XL) load_packed(a.X, B, N, C, Q, b, xs);
  load_pos<DEC>(a, b, pj, mk);
  RadConst<C, DEC> rc;
  rc.load(a, lane);
  __syncthreads();

  const int pr = lane & 15, cg = lane >> 4, ti = pr >> 2, tj = pr & 3;
  const int ngroups = (N + 3) >> 2;
  for (int rg = wave; rg < ngroups; rg += 4) {
    const int i = rg * 4 + ti;
    const bool iok = i < N;
    const int ii = iok ? i : N - 1;
    double pi[PS];
#pragma unroll
    for (int m = 0; m < PS; ++m) pi[m] = pj[ii * PS + m];
    const bool mi = DEC ? false : (mk[ii] != 0);
    for (int q0 = 0; q0 < Q; q0 += QC) {
      cx<double> acc[NG][QC][5];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int x = 0; x < QC; ++x)
#pragma unroll
          for (int k = 0; k < 5; ++k) acc[g][x][k] = {0, 0};
      for (int j0 = 0; j0 < N; j0 += 4) {
        const int j = j0 + tj;
        const bool ok = iok && j < N;
        const int jj = j < N ? j : N - 1;
        cx<double> q[4];
        v4d R[NG];
        double an;
        bool on;
        pair_radial<C, DEC>(rc, pi, pj + jj * PS, ok, mi, DEC ? false : (mk[jj] != 0), q, R, an, on);
        if (ok) {
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int ch = 4 * g + cg;
            if (ch < C) {
              const cx<double> R0 = {R[g][0], R[g][1]}, R1 = {R[g][2], R[g][3]};
              cx<double> e[5];
              e[0] = {R0.r - R0.i, R0.r + R0.i};
#pragma unroll
              for (int m = 0; m < 4; ++m) e[1 + m] = cmul(R1, q[m]);
              const double* xj = xs + ((size_t)jj * C + ch) * Q * 2;
              const double* xg = a.X + (((size_t)b * N + jj) * C + ch) * Q;
#pragma unroll
              for (int x = 0; x < QC; ++x) {
                if (q0 + x < Q) {
                  const cx<double> xv = XL ? cx<double>{xj[(q0 + x) * 2], xj[(q0 + x) * 2 + 1]} : cx<double>{xg[q0 + x], xg[xplane + q0 + x]};
#pragma unroll
                  for (int k = 0; k < 5; ++k) cfma(acc[g][x][k], xv, e[k]);
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ch = 4 * g + cg;
#pragma unroll
        for (int x = 0; x < QC; ++x)
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const double sr = quad_sum(acc[g][x][k].r), si = quad_sum(acc[g][x][k].i);
            if (tj == 0 && iok && ch < C && q0 + x < Q) {
              double* u = a.U + ((((size_t)b * N + i) * C + ch) * Q + q0 + x) * 10 + 2 * k;
              u[0] = sr;
              u[1] = si;
            }
          }
      }
    }
  }
}

// =========================================================================================================
// backward, j-centric:  gX[j][c][q] += sum_i sum_k gU[i][c][q][k] conj(e_k,ij[c])      (+ decoder d p_j)
// =========================================================================================================
template <int C, bool DEC>
__global__ __launch_bounds__(BLOCK) void moments_bwd_nodes_kernel(GenArgs a) {
  constexpr int NG = (C + 3) / 4;
  constexpr int PS = DEC ? 8 : 4;
  const int N = a.N, B = a.B, Q = a.Q;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* pj = reinterpret_cast<double*>(smem_raw);           // N * PS
  uint8_t* mk = reinterpret_cast<uint8_t*>(pj + N * PS);
  load_pos<DEC>(a, b, pj, mk);
  RadConst<C, DEC> rc;
  rc.load(a, lane);
  __syncthreads();

  const int pr = lane & 15, cg = lane >> 4, tj = pr >> 2, ti = pr & 3;
  const size_t plane = (size_t)B * N * C * Q;
  const int ngroups = (N + 3) >> 2;
  for (int rg = wave; rg < ngroups; rg += 4) {
    const int j = rg * 4 + tj;
    const bool jok = j < N;
    const int jj = jok ? j : N - 1;
    double pme[PS];
#pragma unroll
    for (int m = 0; m < PS; ++m) pme[m] = pj[jj * PS + m];
    const bool mj = DEC ? false : (mk[jj] != 0);
    cx<double> Gq[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int q0 = 0; q0 < Q; q0 += QC) {
      cx<double> acc[NG][QC], xme[NG][QC];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int x = 0; x < QC; ++x) {
          acc[g][x] = {0, 0};
          xme[g][x] = {0, 0};
          const int ch = 4 * g + cg;
          if (DEC && ch < C && q0 + x < Q) {
            const size_t e = (((size_t)b * N + jj) * C + ch) * Q + q0 + x;
            xme[g][x] = {a.X[e], a.X[plane + e]};
          }
        }
      for (int i0 = 0; i0 < N; i0 += 4) {
        const int i = i0 + ti;
        const bool ok = jok && i < N;
        const int ii = i < N ? i : N - 1;
        cx<double> q[4];
        v4d R[NG];
        double an;
        bool on;
        pair_radial<C, DEC>(rc, pj + ii * PS, pme, ok, DEC ? false : (mk[ii] != 0), mj, q, R, an, on);
        if (ok) {
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int ch = 4 * g + cg;
            if (ch < C) {
              const cx<double> R0 = {R[g][0], R[g][1]}, R1 = {R[g][2], R[g][3]};
              cx<double> e[5];
              e[0] = {R0.r - R0.i, R0.r + R0.i};
#pragma unroll
              for (int m = 0; m < 4; ++m) e[1 + m] = cmul(R1, q[m]);
              const double* gu = a.gU + (((size_t)b * N + ii) * C + ch) * Q * 10;
#pragma unroll
              for (int x = 0; x < QC; ++x) {
                if (q0 + x < Q) {
                  const double* gq = gu + (q0 + x) * 10;
#pragma unroll
                  for (int k = 0; k < 5; ++k) {
                    const cx<double> gv = {gq[2 * k], gq[2 * k + 1]};
                    cfmac(acc[g][x], gv, e[k]);
                    if (DEC && k > 0) {          // G_e1[m] += gU[q][1+m] conj(X_j[q]);  G_q[m] += G_e1[m] conj(R1)
                      cfmac(Gq[k - 1], cmulc(gv, xme[g][x]), R1);
                    }
                  }
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ch = 4 * g + cg;
#pragma unroll
        for (int x = 0; x < QC; ++x) {
          const double sr = quad_sum(acc[g][x].r), si = quad_sum(acc[g][x].i);
          if (ti == 0 && jok && ch < C && q0 + x < Q) {
            const size_t e = (((size_t)b * N + j) * C + ch) * Q + q0 + x;
            a.gX[e] += sr;
            a.gX[plane + e] += si;
          }
        }
      }
    }
    if (DEC) {
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
// backward, i-centric: radial parameter gradient sums (encoder: T1|T2|S|dB as in level_bwd2; decoder: bias sums)
// and the decoder's d p_i.   G_e_k[c] = sum_q gU[i][c][q][k] conj(X_j[c][q])
// =========================================================================================================
template <int C, bool DEC, bool XL>
__global__ __launch_bounds__(BLOCK) void moments_bwd_rad_kernel(GenArgs a) {
  constexpr int NG = (C + 3) / 4;
  constexpr int PS = DEC ? 8 : 4;
  constexpr int TS = 18;
  const int N = a.N, B = a.B, Q = a.Q;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* xs = reinterpret_cast<double*>(smem_raw);           // N * C * Q * 2 (XL; else read from global memory, see moments_fwd_kernel)
  double* pj = xs + (XL ? (size_t)N * C * Q * 2 : 0);         // N * PS
  const size_t xplane = (size_t)B * N * C * Q;
  double* tr = pj + N * PS;                                   // 4 waves * (NG + 3) * 16 * TS   (also the final reduction buffer)
  constexpr int TRSZ = 4 * (NG + 3) * 16 * TS > 4 * 64 * NG * 12 ? 4 * (NG + 3) * 16 * TS : 4 * 64 * NG * 12;
  uint8_t* mk = reinterpret_cast<uint8_t*>(tr + TRSZ);
  if (XL) load_packed(a.X, B, N, C, Q, b, xs);
  load_pos<DEC>(a, b, pj, mk);
  RadConst<C, DEC> rc;
  rc.load(a, lane);
  __syncthreads();

  const int pr = lane & 15, cg = lane >> 4, ti = pr >> 2, tj = pr & 3;
  double* trw = tr + wave * (NG + 3) * 16 * TS;
  v4d T[NG][3];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < 3; ++t) T[g][t] = v4d{0, 0, 0, 0};
  double dB0[NG], dB1[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) dB0[g] = dB1[g] = 0.0;

  const int ngroups = (N + 3) >> 2;
  for (int rg = wave; rg < ngroups; rg += 4) {
    const int i = rg * 4 + ti;
    const bool iok = i < N;
    const int ii = iok ? i : N - 1;
    double pi[PS];
#pragma unroll
    for (int m = 0; m < PS; ++m) pi[m] = pj[ii * PS + m];
    const bool mi = DEC ? false : (mk[ii] != 0);
    cx<double> Gq[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int j0 = 0; j0 < N; j0 += 4) {
      const int j = j0 + tj;
      const bool ok = iok && j < N;
      const int jj = j < N ? j : N - 1;
      cx<double> q[4];
      v4d R[NG];
      double an;
      bool on;
      pair_radial<C, DEC>(rc, pi, pj + jj * PS, ok, mi, DEC ? false : (mk[jj] != 0), q, R, an, on);
      double* xb = trw + NG * 16 * TS;
      if (!DEC) {
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const double rho = on ? fast_rcp((1.0 + rc.ck2[s] * an) + 1e-16) : 0.0;
          const double x2 = an * rho * rho;
          if (s < 4) {
            xb[pr * TS + 4 * s + cg] = rho;
            xb[16 * TS + pr * TS + 4 * s + cg] = x2;
          } else {
            xb[32 * TS + pr * TS + cg] = rho;
            xb[32 * TS + pr * TS + 4 + cg] = x2;
          }
        }
        xb[32 * TS + pr * TS + 8 + 2 * cg] = cg == 0 ? (on ? 1.0 : 0.0) : 0.0;
        xb[32 * TS + pr * TS + 9 + 2 * cg] = cg == 0 ? (ok ? 1.0 : 0.0) : 0.0;
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ch = 4 * g + cg;
        double G0r = 0, G0i = 0, G1r = 0, G1i = 0;
        if (ok && ch < C) {
          const double* xj = xs + ((size_t)jj * C + ch) * Q * 2;
          const double* xg = a.X + (((size_t)b * N + jj) * C + ch) * Q;
          const double* gu = a.gU + (((size_t)b * N + ii) * C + ch) * Q * 10;
          cx<double> ge[5] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
          for (int x = 0; x < Q; ++x) {
            const cx<double> xv = XL ? cx<double>{xj[2 * x], xj[2 * x + 1]} : cx<double>{xg[x], xg[xplane + x]};
#pragma unroll
            for (int k = 0; k < 5; ++k) cfmac(ge[k], cx<double>{gu[x * 10 + 2 * k], gu[x * 10 + 2 * k + 1]}, xv);
          }
          cx<double> gR1 = {0, 0};
          const cx<double> R1 = {R[g][2], R[g][3]};
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            cfmac(gR1, ge[1 + m], q[m]);
            if (DEC) cfmac(Gq[m], ge[1 + m], R1);
          }
          G0r = ge[0].r + ge[0].i;  G0i = ge[0].i - ge[0].r;
          G1r = gR1.r;  G1i = gR1.i;
        }
        if (DEC) {
          dB0[g] += G0r + G0i;          // R0 = b0 (1+i): d b0 = Re G_R0 + Im G_R0
          dB1[g] += G1r + G1i;
        } else {
          double* ta = trw + g * 16 * TS;
          ta[pr * TS + cg] = G0r;
          ta[pr * TS + 4 + cg] = G0i;
          ta[pr * TS + 8 + cg] = G1r;
          ta[pr * TS + 12 + cg] = G1i;
        }
      }
      if (!DEC) {
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
    if (DEC) {
      const size_t plp = (size_t)B * N * 4;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double qr = quad_sum(Gq[m].r), qi = quad_sum(Gq[m].i);
        qr += shfl_xor(qr, 16);  qr += shfl_xor(qr, 32);
        qi += shfl_xor(qi, 16);  qi += shfl_xor(qi, 32);
        if (iok && tj == 0 && cg == 0) {
          a.g_p[((size_t)b * N + i) * 4 + m] += qr;
          a.g_p[plp + ((size_t)b * N + i) * 4 + m] += qi;
        }
      }
    }
  }

  __syncthreads();
  double* part = a.part_rad + (size_t)blockIdx.x * rad_partial_size(C, DEC);
  if (DEC) {
    // sum over pair slots (lane & 15) and waves; lane>>4 = channel in group
    double* red = tr;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      double x0 = dB0[g], x1 = dB1[g];
      for (int m = 1; m < 16; m <<= 1) { x0 += shfl_xor(x0, m); x1 += shfl_xor(x1, m); }
      if (pr == 0) {
        red[(wave * NG + g) * 8 + cg] = x0;
        red[(wave * NG + g) * 8 + 4 + cg] = x1;
      }
    }
    __syncthreads();
    if (tid < 2 * C) {
      const int lin = tid / C, ch = tid - lin * C, g = ch >> 2, c4 = ch & 3;
      double s = 0;
      for (int w = 0; w < 4; ++w) s += red[(w * NG + g) * 8 + lin * 4 + c4];
      part[tid] = s;
    }
  } else {
    double* red = tr;
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
      const int col = lane & 15;
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
}

// ---------------------------------------------------------------------------------------------------------
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

template <int C, bool DEC>
static int launch_moments(const GenArgs& a, int which, hipStream_t st) {
  constexpr int PS = DEC ? 8 : 4;
  constexpr int NG = (C + 3) / 4;
  const size_t xs = sizeof(double) * (size_t)a.N * C * a.Q * 2, pos = sizeof(double) * (size_t)a.N * PS + a.N + 16;
  int rc;
  if (which == 0) {
    const bool xl = xs + pos <= 160 * 1024;          // (jets whose features do not fit read them from global memory)
    auto k = xl ? moments_fwd_kernel<C, DEC, true> : moments_fwd_kernel<C, DEC, false>;
    const size_t smem = (xl ? xs : 0) + pos;
    if ((rc = set_smem(k, smem, "moments_fwd"))) return rc;
    hipLaunchKernelGGL(k, dim3(a.B), dim3(BLOCK), smem, st, a);
  } else if (which == 1) {
    auto k = moments_bwd_nodes_kernel<C, DEC>;
    if ((rc = set_smem(k, pos, "moments_bwd_nodes"))) return rc;
    hipLaunchKernelGGL(k, dim3(a.B), dim3(BLOCK), pos, st, a);
  } else {
    constexpr int TRSZ = 4 * (NG + 3) * 16 * 18 > 4 * 64 * NG * 12 ? 4 * (NG + 3) * 16 * 18 : 4 * 64 * NG * 12;
    const bool xl = xs + pos + sizeof(double) * TRSZ <= 160 * 1024;
    auto k = xl ? moments_bwd_rad_kernel<C, DEC, true> : moments_bwd_rad_kernel<C, DEC, false>;
    const size_t smem = (xl ? xs : 0) + pos + sizeof(double) * TRSZ;
    if ((rc = set_smem(k, smem, "moments_bwd_rad"))) return rc;
    hipLaunchKernelGGL(k, dim3(a.B), dim3(BLOCK), smem, st, a);
  }
  LGN_CHECK_LAUNCH();
  return 0;
}

bool moments2_fits(const GenArgs& a, int decoder);                                               // generic_moments2.hip
int moments2_dispatch(const GenArgs& a, int decoder, int which, double* Gbuf, hipStream_t st);

// which: 0 forward, 1 backward j-centric, 2 backward i-centric (one partial row per jet).
// Jets that fit in LDS take the channel-outermost kernels of generic_moments2.hip (the encoder's i-centric pass then needs
// a.gbuf, moments2_gbuf_doubles(B, N, C) doubles of scratch); LGN_AMD_MOMENTS_V1=1 keeps this file's kernels.
int moments_dispatch(const GenArgs& a, int decoder, int which, hipStream_t st) {
  LGN_CHECK_ARG(a.B > 0 && a.N > 0 && a.Q > 0, "moments: empty input (B=%d N=%d Q=%d)", a.B, a.N, a.Q);
  {
    const bool v1 = (a.flags & LVL_MOMENTS_V1) != 0;
    if (!v1 && moments2_fits(a, decoder) && (decoder || which != 2 || a.gbuf)) return moments2_dispatch(a, decoder, which, a.gbuf, st);
  }
  LGN_CHECK_ARG(!a.tb, "moments: the tile-blocked layout is implemented by the channel-outermost kernels only (N <= 32)");
#define LGN_CASE(CC) case CC: return decoder ? launch_moments<CC, true>(a, which, st) : launch_moments<CC, false>(a, which, st);
  switch (a.C) {
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
    default: set_error("moments: C=%d unsupported (1..8)", a.C); return -1;
  }
#undef LGN_CASE
}

}  // namespace lgn
