// lgn-autoencoder_amd/csrc/local_static_dev.hpp -- device code shared by the compile-time-table local kernels:
// generic_local_static.hip (moments from global memory: encoder levels, per-operator API) and generic_local_sep.hip (decoder levels
// with the separable moments computed in the kernel, round 6).  See the head of generic_local_static.hip for the mapping.
#pragma once
#include "cg_static_tables.hpp"
#include "wave_sum.hpp"
#include "ops.hpp"

namespace lgn {
namespace lsd {

constexpr int COMAX = 8;

// XCD-aware workgroup index.  A tile's ny workgroups (its input channels in the backward kernels) all read the tile's upstream
// gradient; dealt as grid (tiles, ny) they are ny x tiles ids apart: the re-reads come from HBM (PMC: 118 of the 147 MB a decoder
// backward launch fetched at cfg5).  Consecutive linear ids go round-robin to the 8 XCDs, each with its own L2, so the ny workgroups
// of a tile get ids that are congruent mod 8 and at most 8 ny apart: id = (x / 8) 8 ny + y 8 + x % 8.  Grid: 1-D, xcd_grid(nx, ny)
// workgroups; xcd_index returns false for the padding ids of the last group of 8.
__host__ __device__ inline int xcd_grid(int nx, int ny) { return ((nx + 7) / 8) * 8 * ny; }
__device__ __forceinline__ bool xcd_index(int nx, int ny, int& x, int& y) {
  const int id = blockIdx.x, grp = id / (8 * ny), r = id - grp * 8 * ny;
  y = r >> 3;
  x = grp * 8 + (r & 7);
  return x < nx;
}

// Separable decoder moments (round 6): instead of the moments tensor U, a table of jet-level sums written by dec_sep_tab
// (generic_moments_sep.hip): entry (jet b, channel c, component q) = TBL_STRIDE doubles
//   [0,1] E = e0 SX    [2,3] A = R1 SX    [4 + 2m, 5 + 2m] B_m = R1 SXP_m    [12,13] SX    [14 + 2m, 15 + 2m] SXP_m      (m = 0..3)
// and the node's centred canonical momenta pc [node][8] = P_m (re, im).  Then U[i][q][0] = E[q], U[i][q][1 + m] = P_i[m] A[q] - B_m[q]:
// a 16-byte load (shared by the lanes of a jet) and two fused multiply-adds per component instead of two 8-byte loads of a
// tensor that is written once and read once (147 MB per level at cfg5).
constexpr int TBL_STRIDE = SEP_TBL_STRIDE;
struct SepLane {
  const double* tbc;       // table rows of this lane's jet and the current channel: tbc[q * TBL_STRIDE + ..]
  cx<double> P[4];
};
typedef double sep_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cx<double> sep_A(const SepLane& s, int q) {
  const sep_d2 A = reinterpret_cast<const sep_d2*>(s.tbc + q * TBL_STRIDE)[1];
  return {A.x, A.y};
}
// moment e = 5 q + k of the lane's node
__device__ __forceinline__ cx<double> sep_u(const SepLane& s, int e) {
  const int q = e / 5, k = e % 5;
  const sep_d2* t = reinterpret_cast<const sep_d2*>(s.tbc + q * TBL_STRIDE);
  if (k == 0) {
    const sep_d2 E = t[0];
    return {E.x, E.y};
  }
  const sep_d2 A = t[1], Bm = t[1 + k];
  cx<double> r = {-Bm.x, -Bm.y};
  cfma(r, s.P[k - 1], cx<double>{A.x, A.y});
  return r;
}

struct StaticArgs {
  int M, C, CO;
  const double* __restrict__ XT;      // [tile][C][Q][2][64]
  const double* __restrict__ UT;      // [tile][C][5 Q][2][64]                      (SEP: unused)
  const double* __restrict__ tbl;     // SEP: jet table [B][C][Q][TBL_STRIDE]
  const double* __restrict__ pc;      // SEP: centred momenta [M][8]
  int N;                              // SEP: particles per jet (node n belongs to jet n / N)
  const double* __restrict__ wp;      // packed CatMix weights: irrep l at wp + wp0[l]: [C][nblk_l][COT][2]
  int wp0[8];
  double* __restrict__ outT;          // [tile][CO][Qout][2][64]
  double* __restrict__ s_copy;        // optional [2][M][CO]: copy of output component q_s (pre-MLP scalars, dense layout for the MLP)
  int q_s;
};

// one item: rows M0 .. M0 + ROWS - 1 of output irrep L.  COT = compile-time bound on the output channels (4, 6 or 8; the packed
// weights are zero beyond CO), so the loops over o are branch-free and a block's weights are ONE scalar load.
template <class T, int L, int M0, int ROWS, int COT, bool SEP = false>
__device__ __forceinline__ void item_fwd(const StaticArgs& a, int tile, int lane) {
  constexpr int D = T::DIM[L], NB = T::NBLK[L], ROW0 = T::ROW0[L], Q = T::Q;
  const int C = a.C, CO = a.CO;
  // the weights are read through the constant address space: wave-uniform addresses then become scalar (SMEM) loads into SGPRs
  typedef const double __attribute__((address_space(4))) * cptr;
  cptr wl = (cptr)(a.wp + a.wp0[L]);
  const double* __restrict__ xt = a.XT + (size_t)tile * C * Q * 128 + lane;
  const double* __restrict__ ut = a.UT + (size_t)tile * C * Q * 640 + lane;
  cx<double> acc[COT][ROWS];
#pragma unroll
  for (int o = 0; o < COT; ++o)
#pragma unroll
    for (int mm = 0; mm < ROWS; ++mm) acc[o][mm] = {0, 0};
  SepLane sl{};
  const double* tb0 = nullptr;
  if constexpr (SEP) {
    const int n = tile * 64 + lane < a.M ? tile * 64 + lane : a.M - 1;      // (padding lanes of the last tile: any valid node)
    tb0 = a.tbl + (size_t)(n / a.N) * C * Q * TBL_STRIDE;
    const sep_d2* pp = reinterpret_cast<const sep_d2*>(a.pc + (size_t)n * 8);
#pragma unroll
    for (int m = 0; m < 4; ++m) { const sep_d2 v = pp[m]; sl.P[m] = {v.x, v.y}; }
  }
  for (int c = 0; c < C; ++c) {
    const double* __restrict__ xc = xt + (size_t)c * Q * 128;
    const double* __restrict__ uc = ut + (size_t)c * Q * 640;
    if constexpr (SEP) sl.tbc = tb0 + (size_t)c * Q * TBL_STRIDE;
    cptr wc = wl + (size_t)c * NB * COT * 2;
    cx<double> x[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) x[q] = {xc[q * 128], xc[q * 128 + 64]};
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      cx<double> cat[ROWS];
#pragma unroll
      for (int mm = 0; mm < ROWS; ++mm) {
        const int row = ROW0 + blk * D + M0 + mm;
        cat[mm] = {0, 0};
#pragma unroll
        for (int t = T::ROW_PTR[row]; t < T::ROW_PTR[row + 1]; ++t) {
          const int ty = T::T_TYPE[t], ia = T::T_A[t], ib = T::T_B[t];
          const double cf = T::T_COEF[t];
          cx<double> v;
          if (ty == 0) {
            if constexpr (SEP) v = sep_u(sl, ia);
            else v = {uc[ia * 128], uc[ia * 128 + 64]};
          }
          else if (ty == 1) v = x[ia];
          else v = cmul(x[ia], x[ib]);
          cat[mm].r = __builtin_fma(cf, v.r, cat[mm].r);
          cat[mm].i = __builtin_fma(cf, v.i, cat[mm].i);
        }
      }
#pragma unroll
      for (int o = 0; o < COT; ++o) {
        const cx<double> w = {wc[(blk * COT + o) * 2], wc[(blk * COT + o) * 2 + 1]};     // wave-uniform, immediate offset
#pragma unroll
        for (int mm = 0; mm < ROWS; ++mm) cfma(acc[o][mm], w, cat[mm]);
      }
      __builtin_amdgcn_sched_barrier(0);     // one block at a time: hoisting more weight loads only spills SGPRs
    }
  }
  constexpr int QO = T::QOUT, QBASE = T::Q0[L] + M0;
  const int node = tile * 64 + lane;
  double* __restrict__ ot = a.outT + (size_t)tile * CO * QO * 128 + lane;
#pragma unroll
  for (int o = 0; o < COT; ++o) {
    if (o < CO) {
#pragma unroll
      for (int mm = 0; mm < ROWS; ++mm) {
        double* __restrict__ dst = ot + (size_t)(o * QO + QBASE + mm) * 128;
        dst[0] = acc[o][mm].r;
        dst[64] = acc[o][mm].i;
        if (a.s_copy && QBASE + mm == a.q_s && node < a.M) {
          a.s_copy[(size_t)node * CO + o] = acc[o][mm].r;
          a.s_copy[(size_t)a.M * CO + (size_t)node * CO + o] = acc[o][mm].i;
        }
      }
    }
  }
}

struct PackArgs {
  int C, CO, COT, n_out;
  int nblk[8], w0[8], wp0[8];
};

// =====================================================================================================================
// backward.  Workgroup = (tile of 64 nodes, input channel c): the gradient of the node features of channel c depends on that
// channel alone (the power products are channel-wise).  lane = node; the walk is unrolled completely, block by block:
//   g_cat[row] = sum_o g_out[o][q] conj(W[o][blk, c])          (recomputed per row from the packed weights: scalar loads; the
//                upstream gradient rows of the irrep stay in registers over its blocks)
//   d X        : a block reads two contiguous feature ranges (the two factors of its Clebsch-Gordan product); their values and
//                the gradients the block's terms produce live in registers, the gradients are added to the wave's d X image
//                in LDS once per block (plain 16-byte read-modify-write of lane-private columns -- LDS float atomics per term
//                were the bottleneck of an earlier version: ~40 cycles per wave-wide ds_add_f64)
//   d U        : a moment feeds a few rows of a block; register accumulators per block, started from the value an earlier
//                irrep's block left in d U where there is one.  The moments (and those old values) of block b + 1 are loaded
//                while block b computes: the kernel runs one wave per SIMD (~500 registers), nothing else hides HBM latency
//   d W[o][blk, c] = sum_nodes sum_m g_out[o][q0 + m] conj(cat[row])   : the only cross-lane sum, a register butterfly
//                (wave_sum_store); one partial row per tile in the packed weight layout, reduced over tiles afterwards.
// =====================================================================================================================
struct StaticBwdArgs {
  int M, C, CO;
  const double* __restrict__ XT;      // [tile][C][Q][2][64]
  const double* __restrict__ UT;      // [tile][C][5 Q][2][64]
  const double* __restrict__ wp;      // packed weights (pack_weights_kernel)
  int wp0[8];
  const double* __restrict__ goT;     // upstream gradient [tile][CO][Qout][2][64]
  double* __restrict__ gUT;           // [tile][C][5 Q][2][64]   (every entry written)
  double* __restrict__ gXT;           // [tile][C][Q][2][64]     (overwritten; the N^2 backward adds the aggregate part)
  double* __restrict__ part;          // [tiles][n_packed]  partial CatMix weight gradients: packed layout, or (param_w0) the layout
  int n_packed;                       //                    of the CatMix parameters themselves, rows n_packed apart all the same
  // param_layout: the partial rows are written where the PARAMETERS sit relative to the level's CatMix base (irrep l at w0[l]:
  // [2][CO][nblk_l C]) -- the batch reduction then writes the gradient itself, nothing to unpack (whole-network / whole-step calls)
  int param_layout;
  int w0[8];
  // separable decoder form (generic_local_sep.hip): workgroup = (pair of jets, channel)
  const double* __restrict__ tbl;     // jet table (see TBL_STRIDE)
  const double* __restrict__ pc;      // centred momenta [M][8]
  const double *b0, *b1;              // the level's radial Linear biases [C] (decoder: they ARE the radial functions)
  double* __restrict__ gpb;           // [C][M][8]  per-channel d p (re m = 0..3 | im m = 0..3), every entry written
  double* __restrict__ part_rad;      // [B][2 C]   bias-gradient terms of the jet: dB0[c] | dB1[c]
  int B, N;
};


// feature index range [lo, hi) a block's terms touch through T_A (which = 0) / T_B (which = 1; product terms only)
template <class T>
constexpr int blk_lo(int row0, int rows, int which) {
  int lo = 1 << 30;
  for (int t = T::ROW_PTR[row0]; t < T::ROW_PTR[row0 + rows]; ++t) {
    if (T::T_TYPE[t] == 0 || (which == 1 && T::T_TYPE[t] != 2)) continue;
    const int v = which ? T::T_B[t] : T::T_A[t];
    if (v < lo) lo = v;
  }
  return lo == (1 << 30) ? 0 : lo;
}
template <class T>
constexpr int blk_hi(int row0, int rows, int which) {
  int hi = 0;
  for (int t = T::ROW_PTR[row0]; t < T::ROW_PTR[row0 + rows]; ++t) {
    if (T::T_TYPE[t] == 0 || (which == 1 && T::T_TYPE[t] != 2)) continue;
    const int v = (which ? T::T_B[t] : T::T_A[t]) + 1;
    if (v > hi) hi = v;
  }
  return hi;
}

// one row of the backward walk: gradient of the cat row, its scatter into the block's d X accumulators (registers) / d U
// (global), the row's value and its contribution to the block's weight gradient.  xa / ga cover the features [A0, A0 + NA)
// the block's terms read through T_A, xb / gb those read through T_B.
// where a row's moments come from: the block's preloaded values (UvRef: generic_local_static.hip) or, in the separable decoder form,
// formed on the spot from the LDS copies of the jet table and the lane's momenta (generic_local_sep.hip: UvLds)
template <int NU>
struct UvRef {
  const cx<double> (&uv)[NU];
  __device__ __forceinline__ cx<double> get(int /*e*/, int k) const { return uv[k]; }
};
template <class T, int ROW, int COT, int A0, int NA, int B0, int NB, int NU, class US>
__device__ __forceinline__ void row_bwd(const cx<double> (&gom)[COT], const cx<double> (&w)[COT], const US& us,
                                        cx<double> (&gu)[NU], const cx<double> (&xa)[NA], cx<double> (&ga)[NA],
                                        const cx<double> (&xb)[NB], cx<double> (&gb)[NB], double (&dw)[2 * COT]) {
  cx<double> gc = {0, 0};
#pragma unroll
  for (int o = 0; o < COT; ++o) cfmac(gc, gom[o], w[o]);
  cx<double> cat = {0, 0};
#pragma unroll
  for (int t = T::ROW_PTR[ROW]; t < T::ROW_PTR[ROW + 1]; ++t) {
    const int ty = T::T_TYPE[t], ia = T::T_A[t], ib = T::T_B[t];
    const double cf = T::T_COEF[t];
    const cx<double> g = {cf * gc.r, cf * gc.i};
    if (ty == 0) {
      const int k = T::T_USLOT[t];
      const cx<double> u = us.get(ia, k);
      cat.r = __builtin_fma(cf, u.r, cat.r);
      cat.i = __builtin_fma(cf, u.i, cat.i);
      gu[k].r += g.r;
      gu[k].i += g.i;
    } else if (ty == 1) {
      cat.r = __builtin_fma(cf, xa[ia - A0].r, cat.r);
      cat.i = __builtin_fma(cf, xa[ia - A0].i, cat.i);
      ga[ia - A0].r += g.r;
      ga[ia - A0].i += g.i;
    } else {
      const cx<double> p = cmul(xa[ia - A0], xb[ib - B0]);
      cat.r = __builtin_fma(cf, p.r, cat.r);
      cat.i = __builtin_fma(cf, p.i, cat.i);
      cfmac(ga[ia - A0], g, xb[ib - B0]);
      cfmac(gb[ib - B0], g, xa[ia - A0]);
    }
  }
#pragma unroll
  for (int o = 0; o < COT; ++o) {
    cx<double> d = {dw[2 * o], dw[2 * o + 1]};
    cfmac(d, gom[o], cat);
    dw[2 * o] = d.r;
    dw[2 * o + 1] = d.i;
  }
}

// rows ROWB + MM .. of a block (compile-time recursion: every row index is a template constant)
template <class T, int ROWB, int MM, int D, int COT, int A0, int NA, int B0, int NB, int NU, class US>
__device__ __forceinline__ void rows_bwd(const cx<double> (&go)[COT][D], const cx<double> (&w)[COT], const US& us,
                                         cx<double> (&gu)[NU], const cx<double> (&xa)[NA], cx<double> (&ga)[NA],
                                         const cx<double> (&xb)[NB], cx<double> (&gb)[NB], double (&dw)[2 * COT]) {
  if constexpr (MM < D) {
    cx<double> gom[COT];
#pragma unroll
    for (int o = 0; o < COT; ++o) gom[o] = go[o][MM];
    row_bwd<T, ROWB + MM, COT, A0, NA, B0, NB, NU>(gom, w, us, gu, xa, ga, xb, gb, dw);
    if constexpr (MM % 3 == 2) __builtin_amdgcn_sched_barrier(0);      // three rows at a time
    rows_bwd<T, ROWB, MM + 1, D, COT, A0, NA, B0, NB, NU>(go, w, us, gu, xa, ga, xb, gb, dw);
  }
}

// where a block's weight-gradient lane sums go: the total of value v = 2 o + plane (wave_sum_slot) of block BLK at p[BLK * stride]
struct DwOut {
  double* p;
  int stride;
  bool on;
};
template <int NV>
__device__ __forceinline__ void dw_store(const double (&dw)[NV], const DwOut& out, int blk, int lane) {
  const double w = wave_sum_core<NV>(dw, lane);
  if (out.on) out.p[blk * out.stride] = w;
}
// irrep L (nb blocks) of channel c of partial row `row`
template <int COT>
__device__ __forceinline__ DwOut dw_out(const StaticBwdArgs& a, double* row, int L, int nb, int c, int lane) {
  const int v = wave_sum_slot<2 * COT>(lane);
  if (a.param_layout) {
    const int K = nb * a.C, o = v >> 1;
    return DwOut{row + a.w0[L] + (size_t)(v & 1) * a.CO * K + (size_t)o * K + c, a.C, v >= 0 && o < a.CO};
  }
  return DwOut{row + a.wp0[L] + (size_t)c * nb * COT * 2 + (v >= 0 ? v : 0), COT * 2, v >= 0};
}

// d X accumulator of the wave in LDS, lane-private columns gxl[q * 128 + {0, 1}] (pointer already offset by the lane): plain
// read-modify-write per block, one 16-byte read and write per feature the block touches
template <int N>
__device__ __forceinline__ void gx_flush(double* gxl, int q0, const cx<double> (&g)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    gxl[(q0 + k) * 128] += g[k].r;
    gxl[(q0 + k) * 128 + 1] += g[k].i;
  }
}

// moments a block reads: BLK_UELEM[BLK_UPTR[bid] ..], bid = BLK0[l] + blk (array sizes: at least 1)
template <class T> constexpr int blk_u0(int l, int blk) { return blk < T::NBLK[l] ? T::BLK_UPTR[T::BLK0[l] + blk] : 0; }
template <class T> constexpr int blk_nu(int l, int blk) {
  return blk < T::NBLK[l] ? T::BLK_UPTR[T::BLK0[l] + blk + 1] - T::BLK_UPTR[T::BLK0[l] + blk] : 0;
}
template <class T> constexpr int blk_nu1(int l, int blk) { return blk_nu<T>(l, blk) > 0 ? blk_nu<T>(l, blk) : 1; }
// issue the loads of a block's moments and, where an earlier block (of another irrep) has written the gradient, of its value
// (valid = false on the padding lanes of the last tile: nothing ever wrote their moments -- the loads stay in bounds, the
// values are replaced by zeros so that 0 * garbage cannot put a NaN into the lane sums of the weight gradient)
template <class T, int L, int BLK, int N>
__device__ __forceinline__ void load_moments(const double* __restrict__ uc, const double* guc, cx<double> (&uv)[N], cx<double> (&gold)[N],
                                             bool valid) {
  constexpr int P0 = blk_u0<T>(L, BLK), NU = blk_nu<T>(L, BLK);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    uv[k] = {0, 0};
    gold[k] = {0, 0};
    if (k < NU) {
      const int e = T::BLK_UELEM[P0 + k];
      const double ur = uc[e * 128], ui = uc[e * 128 + 64];
      uv[k] = {valid ? ur : 0.0, valid ? ui : 0.0};
      if (!T::BLK_UFIRST[P0 + k]) gold[k] = {guc[e * 128], guc[e * 128 + 64]};
    }
  }
}

template <class T>
inline void fill_pack(PackArgs& p, int C, int CO, const int* w0) {
  p.C = C; p.CO = CO; p.COT = CO <= 4 ? 4 : (CO <= 6 ? 6 : 8); p.n_out = T::N_OUT;
  int off = 0;
  for (int l = 0; l < T::N_OUT; ++l) {
    p.nblk[l] = T::NBLK[l];
    p.w0[l] = w0[l];
    p.wp0[l] = off;
    off += C * T::NBLK[l] * p.COT * 2;
  }
}


}  // namespace lsd
}  // namespace lgn
