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
//  * the aggregate of the whole jet is collected in LDS; after one barrier all 256 threads write it out (it is saved
//    for the backward) and run CatMix over the items (row, out channel, component).
#include "level_dev.hpp"
#include "net_dev.hpp"
#include "ops.hpp"
#include "wave_sum.hpp"

namespace lgn {

typedef double v4d __attribute__((ext_vector_type(4)));
LGN_STAMP_DECL
LGN_STAMP_READER(lgn_debug_stamps_fwd2)

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

// SEP (decoder only): the decoder's edge mask is identically zero, so its radial functions are the Linear biases --
// per-channel constants R0, R1 (position_levels.py:184-188) -- and the aggregate over j separates into jet-level sums:
//   A1_i = e0 sum_j v_j            A2_i = R1 (p_i sum_j s_j - sum_j s_j p_j)
//   A4_i = e0 sum_j s_j            A3_i = R1 (<sum_j v_j, p_i> - sum_j <v_j, p_j>) / 2
// O(N C) instead of O(N^2 C) work per jet, same values up to summation order.  SEP = false keeps the pair sweep.
template <int C, bool DEC, bool SEP>
__global__ __launch_bounds__(2 * BLOCK) void level_fwd2_kernel(LevelArgs<double> a, int chunk) {
  using F = Fwd2<C, DEC>;
  constexpr int NG = F::NG;
  const int N = a.N, B = a.B, CO = a.CO;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nw = nthr >> 6;            // 4 waves, or 8 when the batch alone cannot fill the SIMDs

  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* nd = reinterpret_cast<double*>(smem_raw);                   // N * NS
  double* pj = nd + ((N * F::NS + 1) & ~1);                           // N * PS
  double* wm = pj + N * F::PS;                                        // 4 * CO * 5C
  double* agl = wm + 4 * CO * 5 * C;                                  // chunk rows * AGS: aggregate of a chunk of rows
  double* sums = agl + chunk * F::AGS;                                // 20 C: jet-level sums of the separable form
  uint8_t* mk = reinterpret_cast<uint8_t*>(sums + 20 * C);            // N

  STAMP(0);
  if (!DEC && a.in_w0) {
    // first encoder level of a fused network: input features from the momenta (see LevelArgs::in_w0), buffers to clear
    {
      const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x, stride = (size_t)gridDim.x * gridDim.y * nthr;
      for (size_t e = wg * nthr + tid; e < a.z1n; e += stride) a.z1[e] = 0.0;
      for (size_t e = wg * nthr + tid; e < a.z2n; e += stride) a.z2[e] = 0.0;
    }
    const double* p0 = a.p + (size_t)b * N * 4;
    const size_t plane_s = (size_t)B * N * C;
    const int nw2 = 2 * CO * 5 * C;
    for (int e = tid; e < N * 4; e += nthr) pj[e] = p0[e];
    for (int e = tid; e < N; e += nthr) mk[e] = a.mask[(size_t)b * N + e];
    for (int e = tid; e < nw2; e += nthr) {
      wm[e] = a.wm0[e];
      wm[nw2 + e] = a.wm1[e];
    }
    for (int e = tid; e < N * C; e += nthr) {
      const int j = e / C, c = e - j * C;
      const double* p = p0 + j * 4;
      const double pe = p[0], px = p[1], py = p[2], pz = p[3];
      // 2 E^2 - sum p^2 with the left-to-right sum of the reference's CPU reduction (zonal_functions.py:201-218)
      const double q0 = pe * pe, q1 = px * px, q2 = py * py, q3 = pz * pz;
      const double mass = sqrt(fabs(2.0 * q0 - (((q0 + q1) + q2) + q3)));
      constexpr double H = 0.70710678118654752440084436210484903928;
      const cx<double> q[4] = {{pe, 0.0}, {px * H, -py * H}, {pz, 0.0}, {-px * H, -py * H}};     // p_to_rep (zonal_functions.py:251-289)
      const double w0r = a.in_w0[c], w0i = a.in_w0[C + c];
      const cx<double> w1 = {a.in_w1[c], a.in_w1[C + c]};
      double* d = nd + j * F::NS + c * 10;
      d[0] = w0r * mass;
      d[1] = w0i * mass;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const cx<double> r = cmul(w1, q[m]);
        d[2 + m] = r.r;
        d[6 + m] = r.i;
      }
      if (blockIdx.y == 0) {                                 // one copy for the backward (s_in / v_in of this level)
        const size_t ge = (size_t)b * N * C + e;
        a.in_s[ge] = d[0];
        a.in_s[plane_s + ge] = d[1];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          a.in_v[ge * 4 + m] = d[2 + m];
          a.in_v[plane_s * 4 + ge * 4 + m] = d[6 + m];
        }
      }
    }
  } else {  // the jet and the CatMix weights in one memory round trip (level_dev.hpp: load_jet_issue)
    JetRegs<double> jr;
    load_jet_issue<double, C, DEC>(a.s_in, a.v_in, a.p, a.mask, B, N, b, jr);
    const int nw2 = 2 * CO * 5 * C, ew = tid < nw2 ? tid : 0;
    const double w0v = a.wm0[ew], w1v = a.wm1[ew];
    load_jet_commit<double, C, DEC>(a.s_in, a.v_in, a.p, a.mask, B, N, b, jr, nd, pj, mk);
    if (tid < nw2) {
      wm[tid] = w0v;
      wm[nw2 + tid] = w1v;
    }
    for (int e = tid + nthr; e < nw2; e += nthr) {
      wm[e] = a.wm0[e];
      wm[nw2 + e] = a.wm1[e];
    }
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
  STAMP(1);

  if constexpr (DEC && SEP) {
    // Only differences p_i - p_j enter: centre the momenta on the jet mean first, so that the separated sums
    // p_i S - SP do not cancel digits the pair sweep would keep (boosted jets: |p| >> |p_i - p_j|).
    if (tid < 64) {                                       // lane = (node part, component): 8 x 8, parts meet by shuffles
      const int k = tid & 7, part = tid >> 3;
      double mean = 0.0;
      for (int n = part; n < N; n += 8) mean += pj[n * 8 + k];
      mean += shfl_xor(mean, 8);
      mean += shfl_xor(mean, 16);
      mean += shfl_xor(mean, 32);
      if (part == 0) sums[k] = mean / N;
    }
    __syncthreads();
    for (int e = tid; e < N * 8; e += nthr) pj[e] -= sums[e & 7];
    __syncthreads();
    // jet-level sums per channel: S | VS[4] | SP[4] | VP (10 complex numbers).  wave = channel, lane = node: the per-node terms
    // stay in registers and are summed over the lanes by transposing butterflies (wave_sum.hpp) -- no LDS staging, no barriers
#define LGN_PUT(k, val)                                           \
  do {                                                            \
    if ((k) < 12) a0[(k) < 12 ? (k) : 0] += (val);                \
    else a1[(k) >= 12 ? (k) - 12 : 0] += (val);                   \
  } while (0)
    for (int c = wave; c < C; c += nw) {
      double a0[12], a1[8];
#pragma unroll
      for (int k = 0; k < 12; ++k) a0[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) a1[k] = 0.0;
      for (int n = lane; n < N; n += 64) {
        const double* ni = nd + n * F::NS + c * 10;
        const double* pn = pj + n * 8;
        const cx<double> sn = {ni[0], ni[1]};
        cx<double> v[4], pc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          v[m] = {ni[2 + m], ni[6 + m]};
          pc[m] = {pn[m], pn[4 + m]};
        }
        LGN_PUT(0, sn.r);  LGN_PUT(1, sn.i);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          LGN_PUT(2 + 2 * m, v[m].r);  LGN_PUT(3 + 2 * m, v[m].i);
          const cx<double> sp = cmul(sn, pc[m]);
          LGN_PUT(10 + 2 * m, sp.r);  LGN_PUT(11 + 2 * m, sp.i);
        }
        const cx<double> vp = bil2(v, pc);
        LGN_PUT(18, vp.r);  LGN_PUT(19, vp.i);
      }
      wave_sum_store<12>(a0, sums + c * 20, lane);
      wave_sum_store<8>(a1, sums + c * 20 + 12, lane);
    }
#undef LGN_PUT
    __syncthreads();
  }

  // this workgroup's share of the jet's rows (level.hpp: level_jet_split; the whole jet unless the batch is small), processed
  // in chunks of `chunk` rows (a multiple of 16; all of them when they fit the LDS budget)
  const int ngroups = (N + 3) >> 2, gper = (ngroups + (int)gridDim.y - 1) / (int)gridDim.y;
  const int rlo = min(N, 4 * (int)blockIdx.y * gper), rhi = min(N, rlo + 4 * gper);
  // ---- power + CatMix.  A lane owns one row's scalar (wave 0) or one (row, component m) of the vectors (waves 1-3):
  // it builds that item's cat vector x[k] once in registers and runs all out channels over it; the weights are
  // wave-uniform reads.  Cat order per irrep: [aggregate (2C), node (C), power (2C)]; power (0,0) = [<v,v>, s*s],
  // (1,1) = [v*s, s*v]
  auto catmix = [&](int c0, int c1) {
    constexpr int K = 5 * C;
    const int nr = c1 - c0;
    const size_t plo = (size_t)B * N * CO;
    if (wave == 0) {
      for (int rl = lane; rl < nr; rl += 64) {
        const int r = c0 + rl;
        const double* st = agl + rl * F::AGS;
        const double* ni = nd + r * F::NS;
        cx<double> x[K];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const cx<double> sc = {ni[c * 10], ni[c * 10 + 1]};
          cx<double> v[4];
#pragma unroll
          for (int mm = 0; mm < 4; ++mm) v[mm] = {ni[c * 10 + 2 + mm], ni[c * 10 + 6 + mm]};
          cx<double> vv = bil2(v, v);
          vv.r *= 0.5;  vv.i *= 0.5;
          x[c] = {st[F::A3 + 2 * c], st[F::A3 + 2 * c + 1]};
          x[C + c] = {st[F::A4 + 2 * c], st[F::A4 + 2 * c + 1]};
          x[2 * C + c] = sc;
          x[3 * C + c] = vv;
          x[4 * C + c] = cmul(sc, sc);
        }
#pragma unroll 2
        for (int o = 0; o < CO; ++o) {
          const double* wr = wm + (0 * CO + o) * K;
          const double* wi = wm + (1 * CO + o) * K;
          cx<double> acc = {0, 0};
#pragma unroll
          for (int k = 0; k < K; ++k) cfma(acc, cx<double>{wr[k], wi[k]}, x[k]);
          const size_t e = ((size_t)b * N + r) * CO + o;
          a.s_out[e] = acc.r;
          a.s_out[plo + e] = acc.i;
        }
      }
    } else {
      for (int it = lane + 64 * (wave - 1); it < nr * 4; it += 64 * (nw - 1)) {
        const int rl = it >> 2, m = it & 3, r = c0 + rl;
        const double* st = agl + rl * F::AGS;
        const double* ni = nd + r * F::NS;
        cx<double> x[K];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const cx<double> sc = {ni[c * 10], ni[c * 10 + 1]};
          const cx<double> v = {ni[c * 10 + 2 + m], ni[c * 10 + 6 + m]};
          x[c] = {st[F::A1 + (c * 4 + m) * 2], st[F::A1 + (c * 4 + m) * 2 + 1]};
          x[C + c] = {st[F::A2 + (c * 4 + m) * 2], st[F::A2 + (c * 4 + m) * 2 + 1]};
          x[2 * C + c] = v;
          x[3 * C + c] = x[4 * C + c] = cmul(v, sc);
        }
#pragma unroll 2
        for (int o = 0; o < CO; ++o) {
          const double* wr = wm + 2 * CO * K + (0 * CO + o) * K;
          const double* wi = wm + 2 * CO * K + (1 * CO + o) * K;
          cx<double> acc = {0, 0};
#pragma unroll
          for (int k = 0; k < K; ++k) cfma(acc, cx<double>{wr[k], wi[k]}, x[k]);
          const size_t e = ((size_t)b * N + r) * CO + o;
          a.v_out[e * 4 + m] = acc.r;
          a.v_out[plo * 4 + e * 4 + m] = acc.i;
        }
      }
    }
    };
  for (int c0 = rlo; c0 < rhi; c0 += chunk) {
  const int c1 = min(rhi, c0 + chunk);
  if constexpr (DEC && SEP) {
    for (int e = tid; e < (c1 - c0) * C; e += nthr) {
      const int rl = e / C, c = e - rl * C, n = c0 + rl;
      const double* sm = sums + c * 20;
      const double* pn = pj + n * 8;
      const cx<double> R0 = {a.b0[c], a.b0[c]}, R1 = {a.b1[c], a.b1[c]};
      const cx<double> e0 = {R0.r - R0.i, R0.r + R0.i};
      const cx<double> S = {sm[0], sm[1]};
      cx<double> VS[4], pc[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        VS[m] = {sm[2 + 2 * m], sm[3 + 2 * m]};
        pc[m] = {pn[m], pn[4 + m]};
      }
      double* st = agl + rl * F::AGS;
      const cx<double> a4 = cmul(S, e0);
      cx<double> t = bil2(VS, pc);
      t.r -= sm[18];  t.i -= sm[19];
      cx<double> a3 = cmul(R1, t);
      st[F::A3 + 2 * c] = 0.5 * a3.r;  st[F::A3 + 2 * c + 1] = 0.5 * a3.i;
      st[F::A4 + 2 * c] = a4.r;        st[F::A4 + 2 * c + 1] = a4.i;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const cx<double> a1 = cmul(VS[m], e0);
        cx<double> u = cmul(pc[m], S);
        u.r -= sm[10 + 2 * m];  u.i -= sm[11 + 2 * m];
        const cx<double> a2 = cmul(R1, u);
        st[F::A1 + (c * 4 + m) * 2] = a1.r;  st[F::A1 + (c * 4 + m) * 2 + 1] = a1.i;
        st[F::A2 + (c * 4 + m) * 2] = a2.r;  st[F::A2 + (c * 4 + m) * 2 + 1] = a2.i;
      }
    }
  } else {
  // Small batches (level_jet_split: this workgroup has 1 or 2 row groups): the waves that would idle take a share of a row group's
  // PARTNER tiles -- rs = 4 or 2 waves per group; their partial aggregates land in slabs of the staging rows and are added up below
  const int ngr = (c1 - c0 + 3) >> 2;
  const int rs = (!DEC && nw == 4 && ngr >= 1 && ngr <= 2 && chunk >= 16) ? 4 / ngr : 1;      // (workgroup-uniform)
  const int rpart = rs > 1 ? wave % rs : 0, ntl = (N + 3) >> 2;
  // (the row group's sweep as a lambda called from two places: the whole-jet form keeps its compile-time-simple bounds -- with
  // run-time partner bounds in the one loop the cfg2 launches were 0.8 us slower)
  auto row_group = [&](const int rg, const int jlo, const int jhi, const int slab) __attribute__((always_inline)) {
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

    STAMP(2 + ((rg >> 2) & 3) * 4);
    // One tile = this row group x 4 neighbours (16 pairs).  prep(): geometry, basis functions and the radial Linear
    // (matrix cores) of a tile; agg(): the CG aggregate of a tile.  The loop is software pipelined over two tile
    // buffers: the MFMAs of tile t+1 are in flight while the VALU works through the aggregate of tile t.
    struct Tile {
      cx<double> q[4];                  // decoder: complex canonical difference
      double qd0, qd3, qa, qb;          // encoder: q = [d0, a - ib, d3, -a - ib] (real momenta)
      v4d R[NG];
      int jj;
      bool ok;
    };
    auto prep = [&](int j0, Tile& T) {
      const int j = j0 + tj;
      T.ok = iok && j < N;
      T.jj = j < N ? j : N - 1;
      const double* pjj = pj + T.jj * F::PS;
      if (DEC) {
#pragma unroll
        for (int m = 0; m < 4; ++m) T.q[m] = {pi[m] - pjj[m], pi[4 + m] - pjj[4 + m]};
#pragma unroll
        for (int g = 0; g < NG; ++g) T.R[g] = v4d{bias[g][0], bias[g][1], bias[g][2], bias[g][3]};
      } else {
        const double d0 = pi[0] - pjj[0], d1 = pi[1] - pjj[1], d2 = pi[2] - pjj[2], d3 = pi[3] - pjj[3];
        const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
        const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;       // zonal_functions.py:142,201-218
        const double an = fabs(nsq);                                            // (c * norm)^2 == c^2 |norm_sq|
        const bool on = T.ok && mi && (mk[T.jj] != 0) && (nsq != 0.0);
        const double h = rsqrt2<double>();
        T.qd0 = d0;  T.qd3 = d3;  T.qa = d1 * h;  T.qb = d2 * h;
        double beta[5] = {0.0, 0.0, 0.0, 0.0, 0.0};          // masked edge: basis zeroed, Linear bias survives
        if (on) {                                            // (EXEC-masked block: no per-value selects)
#pragma unroll
          for (int s = 0; s < 5; ++s) beta[s] = 1.0 + ck2[s] * an;      // (+ 1e-16 of position_levels.py:146: absorbed, the sum is >= 1)
          double rho5[5];
          rcp5(beta, rho5);
#pragma unroll
          for (int s = 0; s < 5; ++s) beta[s] = __builtin_fma(bk[s], rho5[s], ak[s]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          T.R[g] = v4d{bias[g][0], bias[g][1], bias[g][2], bias[g][3]};
#pragma unroll
          for (int s = 0; s < 5; ++s) T.R[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[g][s], beta[s], T.R[g], 0, 0, 0);
        }
      }
    };
    auto agg = [&](const Tile& T) {
      if (T.ok) {
        const double* nj = nd + T.jj * F::NS;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int ch = 4 * g + cg;
          if (ch < C) {
            const cx<double> R0 = {T.R[g][0], T.R[g][1]}, R1 = {T.R[g][2], T.R[g][3]};
            const cx<double> e0 = {R0.r - R0.i, R0.r + R0.i};
            const cx<double> sj = {nj[ch * 10], nj[ch * 10 + 1]};
            cx<double> vj[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              vj[m] = {nj[ch * 10 + 2 + m], nj[ch * 10 + 6 + m]};
              cfma(A1[g][m], vj[m], e0);
            }
            cfma(A4[g], sj, e0);
            if (!DEC) {
              // e1[m] = R1 q[m] with real momenta q = [d0, a - ib, d3, -a - ib]:
              //   A2[m] += (s_j R1) q[m],   A3 += R1 <v_j, q>,  <v_j, q> = v0 d0 - v2 d3 + a (v3 - v1) - ib (v1 + v3)
              const double qd0 = T.qd0, qd3 = T.qd3, qa = T.qa, qb = T.qb;
              const cx<double> Tm = cmul(sj, R1);
              A2[g][0].r = __builtin_fma(Tm.r, qd0, A2[g][0].r);  A2[g][0].i = __builtin_fma(Tm.i, qd0, A2[g][0].i);
              A2[g][2].r = __builtin_fma(Tm.r, qd3, A2[g][2].r);  A2[g][2].i = __builtin_fma(Tm.i, qd3, A2[g][2].i);
              const double aTr = qa * Tm.r, aTi = qa * Tm.i, bTr = qb * Tm.r, bTi = qb * Tm.i;
              A2[g][1].r += aTr + bTi;  A2[g][1].i += aTi - bTr;       // T ( a - ib)
              A2[g][3].r += bTi - aTr;  A2[g][3].i -= aTi + bTr;       // T (-a - ib)
              const cx<double> dv = {vj[3].r - vj[1].r, vj[3].i - vj[1].i}, sv = {vj[1].r + vj[3].r, vj[1].i + vj[3].i};
              cx<double> V;
              V.r = __builtin_fma(vj[0].r, qd0, __builtin_fma(-vj[2].r, qd3, __builtin_fma(qa, dv.r, qb * sv.i)));
              V.i = __builtin_fma(vj[0].i, qd0, __builtin_fma(-vj[2].i, qd3, __builtin_fma(qa, dv.i, -qb * sv.r)));
              cfma(A3[g], R1, V);
            } else {
              cx<double> e1[4];
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                e1[m] = cmul(R1, T.q[m]);
                cfma(A2[g][m], sj, e1[m]);
              }
              const cx<double> t = bil2(vj, e1);
              A3[g].r += t.r;
              A3[g].i += t.i;
            }
          }
        }
      }
    };
    if (jlo < jhi) {
      Tile ta, tb;
      prep(jlo, ta);
      int j0 = jlo;
      while (true) {
        if (j0 + 4 < jhi) prep(j0 + 4, tb);
        agg(ta);
        j0 += 4;
        if (j0 >= jhi) break;
        if (j0 + 4 < jhi) prep(j0 + 4, ta);
        agg(tb);
        j0 += 4;
        if (j0 >= jhi) break;
      }
    }

    STAMP(3 + ((rg >> 2) & 3) * 4);
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
      double* st = agl + (slab + i0 + ti - c0) * F::AGS;        // rows >= N of the last group land in the padding
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
        }
      }
    }
    STAMP(4 + ((rg >> 2) & 3) * 4);
  };
  if (rs > 1) {
    const int rg = (c0 >> 2) + wave / rs;
    if (rg * 4 < c1) row_group(rg, 4 * (rpart * ntl / rs), min(N, 4 * ((rpart + 1) * ntl / rs)), rpart * 4 * ngr);
  } else {
    for (int rg = (c0 >> 2) + wave; rg * 4 < c1; rg += nw) row_group(rg, 0, N, 0);
  }
  if (rs > 1) {                                             // the partner parts of a row: slabs 1 .. rs - 1 added to slab 0, in part order
    __syncthreads();
    const int tot = 4 * ngr * F::AGS;
    for (int e = tid; e < tot; e += nthr) {
      double v = agl[e];
      for (int pp = 1; pp < rs; ++pp) v += agl[pp * tot + e];
      agl[e] = v;
    }
  }
  }
  __syncthreads();
  STAMP(20);

  // ---- aggregate -> global (saved for the backward): ag0 [2][B][N][2C], ag1 [2][B][N][2C][4] -----------------------
  {
    const size_t pl0 = (size_t)B * N * 2 * C;
    for (int e = tid; e < (c1 - c0) * 2 * C; e += nthr) {        // e = n * 2C + (blk * C + ch), blk 0: A3, 1: A4
      const int n = e / (2 * C), r = e - n * 2 * C, blk = r / C, ch = r - blk * C;
      const double* st = agl + n * F::AGS + (blk ? F::A4 : F::A3) + 2 * ch;
      const size_t ge = ((size_t)b * N + c0) * 2 * C + e;
      a.ag0[ge] = st[0];
      a.ag0[pl0 + ge] = st[1];
    }
    for (int e = tid; e < (c1 - c0) * 2 * C * 4; e += nthr) {    // e = (n * 2C + blk * C + ch) * 4 + m, blk 0: A1, 1: A2
      const int n = e / (8 * C), r = e - n * 8 * C, blk = r / (4 * C), cm = r - blk * 4 * C;
      const double* st = agl + n * F::AGS + (blk ? F::A2 : F::A1) + 2 * cm;
      const size_t ge = ((size_t)b * N + c0) * 8 * C + e;
      a.ag1[ge] = st[0];
      a.ag1[pl0 * 4 + ge] = st[1];
    }
  }
  STAMP(21);

  catmix(c0, c1);
  STAMP(22);
  if (c1 < rhi) __syncthreads();                         // the chunk's aggregate rows are reused
  }
  STAMP(40);
  if constexpr (DEC) {
    if (a.loss_wo1) {      // (one workgroup of BLOCK threads per jet: checked on the host)
      __syncthreads();     // this jet's v_out is complete and visible to the workgroup; the level's LDS is free
      dec_output_loss_body(B, N, CO, a.v_out, a.loss_wo1, a.loss_target, a.loss_scale, a.loss_recon, a.loss_part, a.loss_gv,
                           a.loss_wpart, smem_raw);
    }
  }
}

template <int C, bool DEC, bool SEP>
static int launch_level_fwd2(const LevelArgs<double>& a, hipStream_t stream) {
  using F = Fwd2<C, DEC>;
  // aggregate rows kept in LDS: the whole jet if that still leaves room for two workgroups per CU (or nothing does),
  // else as many 16-row slabs as fit next to the node data
  const size_t fixed = sizeof(double) * (((size_t)a.N * F::NS + 1 & ~size_t(1)) + (size_t)a.N * F::PS + 4 * a.CO * 5 * C + 20 * C) + a.N + 16;
  // (a batch with no more jets than CUs runs one 8-wave workgroup per CU anyway: it takes the whole LDS)
  const bool wide = a.B <= 320 && a.N >= 64;
  const size_t row = sizeof(double) * F::AGS, budget1 = 160 * 1024, budget2 = wide ? budget1 : 78 * 1024;
  const int full = (a.N + 15) & ~15;
  int chunk = full;
  if (fixed + full * row > budget2) {
    const size_t room = fixed + 16 * row <= budget2 ? budget2 - fixed : (fixed < budget1 ? budget1 - fixed : 0);
    chunk = (int)(room / row) & ~15;
    if (chunk > full) chunk = full;
    if (chunk < 16) chunk = 16;
  }
  size_t smem = fixed + chunk * row;
  if (a.loss_wo1) {
    LGN_CHECK_ARG(DEC && SEP && a.N <= 40 && !wide && a.loss_target && a.loss_recon && a.loss_part && a.loss_gv && a.loss_wpart,
                  "level_fwd: the loss rides on the separable decoder forward of jets of <= 40 particles only");
    if (smem < dec_out_loss_bytes(a.N, a.CO)) smem = dec_out_loss_bytes(a.N, a.CO);
  }
  LGN_CHECK_ARG(smem <= 160 * 1024, "level_fwd: N=%d C=%d needs %zu B of LDS (> 160 KiB)", a.N, a.C, smem);
  auto kern = level_fwd2_kernel<C, DEC, SEP>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  // 8 waves per jet when the batch has no more jets than the chip has CUs and the jet has enough row groups (cfg4)
  const int nthreads = wide ? 2 * BLOCK : BLOCK;
  // small batches of small jets: several workgroups per jet, each with its own rows (level.hpp: level_jet_split)
  const int split = (SEP || a.N > 40) ? 1 : level_jet_split(a.B, a.N);
  hipLaunchKernelGGL(kern, dim3(a.B, split), dim3(nthreads), smem, stream, a, chunk);
  LGN_CHECK_LAUNCH();
  return 0;
}

bool level_fwd_carries_loss(int N, int flags) { return N <= 40 && !(flags & LVL_DEC_PAIRWISE); }

template <>
int level_fwd_dispatch<double>(const LevelArgs<double>& a, int decoder, hipStream_t stream) {
  LGN_CHECK_ARG(a.B > 0 && a.N > 0, "level_fwd: empty batch (B=%d N=%d)", a.B, a.N);
  LGN_CHECK_ARG(!a.in_w0 || (!decoder && a.in_w1 && a.in_s && a.in_v), "level_fwd: the input stage rides on encoder levels only");
  LGN_CHECK_ARG(a.CO >= 1 && a.CO <= 8, "level_fwd: C_out=%d unsupported (1..8)", a.CO);
  const bool pairwise = (a.flags & LVL_DEC_PAIRWISE) != 0;   // the decoder on the O(N^2) pair sweep (cross-check of the separable form)
#define LGN_CASE(CC)                                                                                         \
  case CC:                                                                                                   \
    if (!decoder) return launch_level_fwd2<CC, false, false>(a, stream);                                     \
    return pairwise ? launch_level_fwd2<CC, true, false>(a, stream) : launch_level_fwd2<CC, true, true>(a, stream);
  switch (a.C) {
#ifdef LGN_DEV_ONLY_C4      // development builds: one channel count (compile time)
    LGN_CASE(4)
#else
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
#endif
    default:
      set_error("level_fwd: C_in=%d unsupported (1..8)", a.C);
      return -1;
  }
#undef LGN_CASE
}

}  // namespace lgn
