// lgn-autoencoder_amd/csrc/level_fwd2.hip -- fused message-passing level, forward, maxdim = 2, version 2.
//
// Same operator as level_fwd.hip (RadPolyTrig + rad*zonal + CG aggregate + CG power + CatMix; reference
// lgn/nn/position_levels.py:118-209, lgn/models/lgn_cg.py:167, lgn/cg_lib/cg_ops.py:135-298,
// lgn/nn/g_nn.py:260-278) re-mapped for CDNA4:
//
//  * the radial network's Linear layers (20 basis functions -> 4C outputs per edge, 60 % of the edge flops)
//    run on the fp64 matrix cores:  D[r'][pair] = sum_k W'[r'][k] beta_k(pair) + bias[r']  with
//    v_mfma_f64_16x16x4_f64, A = weights (constant fragments held in registers for the whole kernel),
//    B = basis values.  Lane l evaluates beta_k for pair (l & 15) and k = 4s + (l >> 4), which IS the B
//    fragment layout: no basis value is computed twice and nothing is staged.  With the row order
//    r' = cc + 4*(2*lin + z) the D fragment hands lane (pair = l & 15, cc = l >> 4) exactly the four reals
//    (R0, R1 complex) of ITS pair and channel: producer and consumer roles coincide, no transpose.
//  * a wave owns a "row group" of 4 receiving particles i and sweeps the neighbours in tiles of 4 (16 pairs per
//    tile); the lane's accumulators (20 reals per channel group) stay in registers across the sweep and are
//    combined across the 4 lanes of a quad (the 4 neighbours of a tile) with two DPP quad permutes.
//  * CatMix is done by the same wave right after each row group from a wave-private LDS staging buffer,
//    so there is no workgroup barrier after the prologue.
#include "level_dev.hpp"
#include "ops.hpp"

namespace lgn {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double dpp_quad(double v, int ctrl_is_xor2) {
  // quad_perm [1,0,3,2] = 0xB1 (xor 1), [2,3,0,1] = 0x4E (xor 2)
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (ctrl_is_xor2) {
    lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
  } else {
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_quad(v, 0);
  v += dpp_quad(v, 1);
  return v;
}

// 1/u for u >= 1: hardware reciprocal seed + two Newton steps (the reference divides; relative difference <= 1 ulp)
__device__ __forceinline__ double fast_rcp(double u) {
  double r = __builtin_amdgcn_rcp(u);
  double e = __builtin_fma(-u, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-u, r, 1.0);
  return __builtin_fma(r, e, r);
}

template <int C, bool DEC>
struct Fwd2 {
  static constexpr int NG = (C + 3) / 4;                 // channel groups of 4
  static constexpr int NS = node_stride(C);
  static constexpr int PS = DEC ? 8 : 4;
  static constexpr int AGS = 20 * C;                     // staging per row: A3 | A4 | A1 | A2 (GA layout of level_bwd)
  static constexpr int A3 = 0, A4 = 2 * C, A1 = 4 * C, A2 = 12 * C;
};

template <int C, bool DEC>
__global__ __launch_bounds__(BLOCK) void level_fwd2_kernel(LevelArgs<double> a) {
  using F = Fwd2<C, DEC>;
  constexpr int NG = F::NG;
  const int N = a.N, B = a.B, CO = a.CO;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* nd = reinterpret_cast<double*>(smem_raw);                   // N * NS
  double* pj = nd + ((N * F::NS + 1) & ~1);                           // N * PS
  double* wm = pj + N * F::PS;                                        // 4 * CO * 5C
  double* agw = wm + 4 * CO * 5 * C;                                  // 4 waves * 4 rows * AGS
  uint8_t* mk = reinterpret_cast<uint8_t*>(agw + 4 * 4 * F::AGS);     // N

  load_jet<double, C, DEC>(a.s_in, a.v_in, a.p, a.mask, B, N, b, nd, pj, mk);
  for (int e = tid; e < 2 * CO * 5 * C; e += BLOCK) {
    wm[e] = a.wm0[e];
    wm[2 * CO * 5 * C + e] = a.wm1[e];
  }

  // ---- per-lane constants ---------------------------------------------------------------------------
  const int pr = lane & 15, cg = lane >> 4;                 // pair slot inside a tile / channel inside a group (== k group)
  const int ti = pr >> 2, tj = pr & 3;
  double ak[5], bk[5], ck2[5];                              // basis parameters of this lane's five basis functions
  double wf[NG][5];                                         // A fragments: W'[r' = lane & 15][k = 4s + (lane >> 4)]
  double bias[NG][4];                                       // D init: bias[r' = cg + 4q]
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
      const int rr = lane & 15, q = rr >> 2, ch = 4 * g + (rr & 3);        // row r' = cc + 4q, q = 2*lin + z
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

  double* stage = agw + wave * 4 * F::AGS;
  const int ngroups = (N + 3) >> 2;
  for (int rg = wave; rg < ngroups; rg += 4) {
    const int i0 = rg * 4;
    const int i = i0 + ti;
    const bool iok = i < N;
    const int ii = iok ? i : N - 1;
    double pi[F::PS];
#pragma unroll
    for (int m = 0; m < F::PS; ++m) pi[m] = pj[ii * F::PS + m];
    const bool mi = DEC ? false : (mk[ii] != 0);

    cx<double> A1[NG][4], A2[NG][4], A3[NG], A4[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      A3[g] = {0, 0};
      A4[g] = {0, 0};
#pragma unroll
      for (int m = 0; m < 4; ++m) { A1[g][m] = {0, 0}; A2[g][m] = {0, 0}; }
    }

    for (int j0 = 0; j0 < N; j0 += 4) {
      const int j = j0 + tj;
      const bool ok = iok && j < N;
      const int jj = j < N ? j : N - 1;
      const double* pjj = pj + jj * F::PS;
      cx<double> q[4];
      v4d R[NG];
      if (DEC) {
#pragma unroll
        for (int m = 0; m < 4; ++m) q[m] = {pi[m] - pjj[m], pi[4 + m] - pjj[4 + m]};
#pragma unroll
        for (int g = 0; g < NG; ++g) R[g] = v4d{bias[g][0], bias[g][1], bias[g][2], bias[g][3]};
      } else {
        const double d0 = pi[0] - pjj[0], d1 = pi[1] - pjj[1], d2 = pi[2] - pjj[2], d3 = pi[3] - pjj[3];
        const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
        const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;       // zonal_functions.py:142,201-218
        const double an = fabs(nsq);                                            // (c * norm)^2 == c^2 |norm_sq|
        const bool on = ok && mi && (mk[jj] != 0) && (nsq != 0.0);
        const double h = rsqrt2<double>();
        q[0] = {d0, 0.0};
        q[1] = {d1 * h, -d2 * h};
        q[2] = {d3, 0.0};
        q[3] = {-d1 * h, -d2 * h};
        double beta[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const double u = (1.0 + ck2[s] * an) + 1e-16;
          const double bv = __builtin_fma(bk[s], fast_rcp(u), ak[s]);
          beta[s] = on ? bv : 0.0;                           // masked edge: basis zeroed, Linear bias survives
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          R[g] = v4d{bias[g][0], bias[g][1], bias[g][2], bias[g][3]};
#pragma unroll
          for (int s = 0; s < 5; ++s) R[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[g][s], beta[s], R[g], 0, 0, 0);
        }
      }
      if (ok) {
        const double* nj = nd + jj * F::NS;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int ch = 4 * g + cg;
          if (ch < C) {
            const cx<double> R0 = {R[g][0], R[g][1]}, R1 = {R[g][2], R[g][3]};
            const cx<double> e0 = {R0.r - R0.i, R0.r + R0.i};
            const cx<double> sj = {nj[ch * 10], nj[ch * 10 + 1]};
            cx<double> vj[4], e1[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              vj[m] = {nj[ch * 10 + 2 + m], nj[ch * 10 + 6 + m]};
              e1[m] = cmul(R1, q[m]);
              cfma(A1[g][m], vj[m], e0);
              cfma(A2[g][m], sj, e1[m]);
            }
            cfma(A4[g], sj, e0);
            const cx<double> t = bil2(vj, e1);
            A3[g].r += t.r;
            A3[g].i += t.i;
          }
        }
      }
    }

    // ---- combine the 4 neighbour slots of a tile (quad lanes), stage + store the aggregate -----------------
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      A3[g].r = quad_sum(A3[g].r) * 0.5;  A3[g].i = quad_sum(A3[g].i) * 0.5;
      A4[g].r = quad_sum(A4[g].r);        A4[g].i = quad_sum(A4[g].i);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        A1[g][m].r = quad_sum(A1[g][m].r);  A1[g][m].i = quad_sum(A1[g][m].i);
        A2[g][m].r = quad_sum(A2[g][m].r);  A2[g][m].i = quad_sum(A2[g][m].i);
      }
    }
    if (tj == 0) {
      double* st = stage + ti * F::AGS;
      const size_t pl0 = (size_t)B * N * 2 * C;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ch = 4 * g + cg;
        if (ch < C) {
          st[F::A3 + 2 * ch] = A3[g].r;  st[F::A3 + 2 * ch + 1] = A3[g].i;
          st[F::A4 + 2 * ch] = A4[g].r;  st[F::A4 + 2 * ch + 1] = A4[g].i;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            st[F::A1 + (ch * 4 + m) * 2] = A1[g][m].r;  st[F::A1 + (ch * 4 + m) * 2 + 1] = A1[g][m].i;
            st[F::A2 + (ch * 4 + m) * 2] = A2[g][m].r;  st[F::A2 + (ch * 4 + m) * 2 + 1] = A2[g][m].i;
          }
          if (iok) {
            double* g0 = a.ag0 + ((size_t)b * N + i) * 2 * C;
            double* g1 = a.ag1 + ((size_t)b * N + i) * 2 * C * 4;
            g0[ch] = A3[g].r;  g0[pl0 + ch] = A3[g].i;
            g0[C + ch] = A4[g].r;  g0[pl0 + C + ch] = A4[g].i;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              g1[ch * 4 + m] = A1[g][m].r;  g1[pl0 * 4 + ch * 4 + m] = A1[g][m].i;
              g1[(C + ch) * 4 + m] = A2[g][m].r;  g1[pl0 * 4 + (C + ch) * 4 + m] = A2[g][m].i;
            }
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- power + CatMix for the 4 rows: item = (row, out channel, component m; m == 4 is the scalar) --------
    // Cat order per irrep: [aggregate (2C), node (C), power (2C)]; power (0,0) = [<v,v>, s*s], (1,1) = [v*s, s*v]
    {
      const int K = 5 * C;
      const size_t plo = (size_t)B * N * CO;
      for (int it = lane; it < 20 * CO; it += 64) {
        const int rl = it / (5 * CO), rem = it - rl * 5 * CO, o = rem / 5, m = rem - o * 5;
        const int r = i0 + rl;
        if (r >= N) continue;
        const double* st = stage + rl * F::AGS;
        const double* ni = nd + r * F::NS;
        cx<double> acc = {0, 0};
        if (m == 4) {
          const double* wr = wm + (0 * CO + o) * K;
          const double* wi = wm + (1 * CO + o) * K;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const cx<double> s = {ni[c * 10], ni[c * 10 + 1]};
            cx<double> v[4];
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) v[mm] = {ni[c * 10 + 2 + mm], ni[c * 10 + 6 + mm]};
            cx<double> vv = bil2(v, v);
            vv.r *= 0.5;  vv.i *= 0.5;
            cfma(acc, cx<double>{wr[c], wi[c]}, cx<double>{st[F::A3 + 2 * c], st[F::A3 + 2 * c + 1]});
            cfma(acc, cx<double>{wr[C + c], wi[C + c]}, cx<double>{st[F::A4 + 2 * c], st[F::A4 + 2 * c + 1]});
            cfma(acc, cx<double>{wr[2 * C + c], wi[2 * C + c]}, s);
            cfma(acc, cx<double>{wr[3 * C + c], wi[3 * C + c]}, vv);
            cfma(acc, cx<double>{wr[4 * C + c], wi[4 * C + c]}, cmul(s, s));
          }
          const size_t e = ((size_t)b * N + r) * CO + o;
          a.s_out[e] = acc.r;
          a.s_out[plo + e] = acc.i;
        } else {
          const double* wr = wm + 2 * CO * K + (0 * CO + o) * K;
          const double* wi = wm + 2 * CO * K + (1 * CO + o) * K;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const cx<double> s = {ni[c * 10], ni[c * 10 + 1]};
            const cx<double> v = {ni[c * 10 + 2 + m], ni[c * 10 + 6 + m]};
            cfma(acc, cx<double>{wr[c], wi[c]}, cx<double>{st[F::A1 + (c * 4 + m) * 2], st[F::A1 + (c * 4 + m) * 2 + 1]});
            cfma(acc, cx<double>{wr[C + c], wi[C + c]}, cx<double>{st[F::A2 + (c * 4 + m) * 2], st[F::A2 + (c * 4 + m) * 2 + 1]});
            cfma(acc, cx<double>{wr[2 * C + c], wi[2 * C + c]}, v);
            cfma(acc, cx<double>{wr[3 * C + c] + wr[4 * C + c], wi[3 * C + c] + wi[4 * C + c]}, cmul(v, s));
          }
          const size_t e = ((size_t)b * N + r) * CO + o;
          a.v_out[e * 4 + m] = acc.r;
          a.v_out[plo * 4 + e * 4 + m] = acc.i;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();       // staging buffer is reused by the next row group
  }
}

template <int C, bool DEC>
static int launch_level_fwd2(const LevelArgs<double>& a, hipStream_t stream) {
  using F = Fwd2<C, DEC>;
  const size_t smem = sizeof(double) * (((size_t)a.N * F::NS + 1 & ~size_t(1)) + (size_t)a.N * F::PS + 4 * a.CO * 5 * C + 16 * F::AGS) +
                      a.N + 16;
  LGN_CHECK_ARG(smem <= 160 * 1024, "level_fwd: N=%d C=%d needs %zu B of LDS (> 160 KiB)", a.N, a.C, smem);
  auto kern = level_fwd2_kernel<C, DEC>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(BLOCK), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

int level_fwd2_dispatch(const LevelArgs<double>& a, int decoder, hipStream_t stream) {
  LGN_CHECK_ARG(a.B > 0 && a.N > 0, "level_fwd: empty batch (B=%d N=%d)", a.B, a.N);
  LGN_CHECK_ARG(a.CO >= 1 && a.CO <= 8, "level_fwd: C_out=%d unsupported (1..8)", a.CO);
#define LGN_CASE(CC)                                                              \
  case CC:                                                                        \
    return decoder ? launch_level_fwd2<CC, true>(a, stream) : launch_level_fwd2<CC, false>(a, stream);
  switch (a.C) {
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
    default:
      set_error("level_fwd: C_in=%d unsupported (1..8)", a.C);
      return -1;
  }
#undef LGN_CASE
}

}  // namespace lgn
