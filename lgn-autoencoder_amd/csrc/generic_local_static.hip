// lgn-autoencoder_amd/csrc/generic_local_static.hip -- per-node part of a table-driven level (sparse Clebsch-Gordan contraction
// of the neighbour moments and of node (x) node, concatenation, complex CatMix; reference: cg_product
// lgn/cg_lib/cg_ops.py:177-218, CatReps / CatMixReps lgn/nn/g_nn.py:160-190,260-278) for the two level kinds every maxdim = 3
// network is made of, with the walk tables as COMPILE-TIME constants (cg_static_tables.hpp, generated from lgn/plan.py).
//
// generic_local.hip walks the same tables at run time: every term costs index decoding and LDS / table round trips, and the
// work of one node is spread over a workgroup between barriers.  Here the contraction is unrolled completely:
//   * lane = node; a wave owns 64 consecutive nodes and one ITEM = (output irrep l, chunk of <= 4 of its rows m); the input
//     channels c are an ordinary loop inside the lane, so the CatMix sum over channels needs no cross-lane traffic at all;
//   * per channel the lane loads its 20 feature components into registers (coalesced: see the layout below); every term is
//     then "acc += constant * x[a] * x[b]" with a, b, constant known to the compiler; the moments are loaded at constant
//     offsets (all loads of a block in flight), the CatMix weights are wave-uniform scalar loads;
//   * no LDS, no barrier, no table memory traffic.
// Tile-blocked, node-innermost layouts ("TB64") make every access of a wave one contiguous 512-byte run at a COMPILE-TIME
// offset from the tile's base (no address arithmetic, no address registers):
//     XT [tile][C][Q][2][64]    UT [tile][C][5 Q][2][64]    outT [tile][CO][Qout][2][64]     node n = 64 tile + lane, plane 0 = real
// (generic_moments2.hip produces / consumes the same layouts when a network runs on the static kernels), and the CatMix
// weights are repacked per launch to  Wp [irrep l][c][blk][COT][2]  (zero-padded to the compile-time channel bound COT), so that
// one channel's weights of a block are one contiguous scalar load at an immediate offset.
#include "cg_static_tables.hpp"
#include "ops.hpp"

namespace lgn {
namespace {

constexpr int COMAX = 8;

struct StaticArgs {
  int M, C, CO;
  const double* __restrict__ XT;      // [tile][C][Q][2][64]
  const double* __restrict__ UT;      // [tile][C][5 Q][2][64]
  const double* __restrict__ wp;      // packed CatMix weights: irrep l at wp + wp0[l]: [C][nblk_l][COT][2]
  int wp0[8];
  double* __restrict__ outT;          // [tile][CO][Qout][2][64]
  double* __restrict__ s_copy;        // optional [2][M][CO]: copy of output component q_s (pre-MLP scalars, dense layout for the MLP)
  int q_s;
};

// one item: rows M0 .. M0 + ROWS - 1 of output irrep L.  COT = compile-time bound on the output channels (4, 6 or 8; the packed
// weights are zero beyond CO), so the loops over o are branch-free and a block's weights are ONE scalar load.
template <class T, int L, int M0, int ROWS, int COT>
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
  for (int c = 0; c < C; ++c) {
    const double* __restrict__ xc = xt + (size_t)c * Q * 128;
    const double* __restrict__ uc = ut + (size_t)c * Q * 640;
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
          if (ty == 0) v = {uc[ia * 128], uc[ia * 128 + 64]};
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

// ---- weight packing ----------------------------------------------------------------------------------------------------
// wcat (irrep l at w0[l]: [2][CO][nblk_l * C], the CatMix parameter layout)  ->  wp (irrep l at wp0[l]: [C][nblk_l][COT][2])
struct PackArgs {
  int C, CO, COT, n_out;
  int nblk[8], w0[8], wp0[8];
};
__global__ __launch_bounds__(BLOCK) void pack_weights_kernel(PackArgs p, const double* __restrict__ w, double* __restrict__ wp) {
  for (int l = 0; l < p.n_out; ++l) {
    const int nb = p.nblk[l], K = nb * p.C, total = p.C * nb * p.COT;
    for (int e = blockIdx.x * BLOCK + threadIdx.x; e < total; e += gridDim.x * BLOCK) {
      const int o = e % p.COT, blk = (e / p.COT) % nb, c = e / (p.COT * nb);
      double re = 0.0, im = 0.0;
      if (o < p.CO) {
        re = w[p.w0[l] + (size_t)o * K + blk * p.C + c];
        im = w[p.w0[l] + (size_t)p.CO * K + (size_t)o * K + blk * p.C + c];
      }
      wp[p.wp0[l] + 2 * e] = re;
      wp[p.wp0[l] + 2 * e + 1] = im;
    }
  }
}

// =====================================================================================================================
// backward.  Wave = (tile of 64 nodes, input channel c): the gradient of the node features of channel c depends on that
// channel alone (the power products are channel-wise), so the lane owns d X[c][.] in registers while it walks ALL rows:
//   g_cat[row] = sum_o g_out[o][q] conj(W[o][blk, c])          (recomputed per row from the packed weights: scalar loads)
//   d X, d U   : g_cat pushed through the row's terms -- compile-time indices, register accumulators for d X; a moment feeds
//                ~1 row, its gradient goes straight to dUT (store for the first contribution in walk order, else RMW)
//   d W[o][blk, c] = sum_nodes sum_m g_out[o][q0 + m] conj(cat[row])   : the only cross-lane sum.  Per block the 2 COT lane
//                values are transposed through a 6 KB LDS image and each (o, plane) is summed by four lanes (fixed order:
//                deterministic); one partial row per tile in the packed weight layout, reduced over tiles afterwards.
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
  double* __restrict__ part;          // [tiles][n_packed]  partial CatMix weight gradients, packed layout
  int n_packed;
};

constexpr int RED_PITCH = 65;

template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return v + __hiloint2double(hi, lo);
}
// d X accumulator: lane-private LDS column gxl[q * 128 + {0, 1}] (pointer already offset by the lane); LDS float atomics
// without return: fire and forget, executed in program order (one lane per address: deterministic)
__device__ __forceinline__ void gx_add(double* gxl, int q, cx<double> v) {
#ifdef SB_NO_GX
  if (v.r == 12345.678) gxl[q * 128] = v.i;
  return;
#endif
  __hip_atomic_fetch_add(gxl + q * 128, v.r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __hip_atomic_fetch_add(gxl + q * 128 + 1, v.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// sum the NV = 2 * COT values every lane holds over the 64 lanes; lanes 0, 4, 8, ... return the sum of value (lane >> 2)
template <int NV>
__device__ __forceinline__ double wave_transpose_sum(const double (&v)[NV], double* red, int lane) {
#pragma unroll
  for (int k = 0; k < NV; ++k) red[k * RED_PITCH + lane] = v[k];
  __builtin_amdgcn_wave_barrier();
  const int k = lane >> 2, s = lane & 3;
  double acc = 0.0;
  if (k < NV) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += red[k * RED_PITCH + i * 4 + s];
  }
  acc = dpp_add<0xB1>(acc);
  acc = dpp_add<0x4E>(acc);
  __builtin_amdgcn_wave_barrier();
  return acc;
}

template <class T, int L, int M0, int ROWS, int COT>
__device__ __forceinline__ void item_bwd(const StaticBwdArgs& a, int tile, int lane, int c, const cx<double> (&x)[T::Q],
                                         double* gxl, double* red) {
  constexpr int D = T::DIM[L], NB = T::NBLK[L], ROW0 = T::ROW0[L], Q = T::Q, QO = T::QOUT, QBASE = T::Q0[L] + M0;
  const int C = a.C, CO = a.CO;
  typedef const double __attribute__((address_space(4))) * cptr;
  cptr wc = (cptr)(a.wp + a.wp0[L]) + (size_t)c * NB * COT * 2;
  const double* __restrict__ uc = a.UT + ((size_t)tile * C + c) * Q * 640 + lane;
  double* __restrict__ guc = a.gUT + ((size_t)tile * C + c) * Q * 640 + lane;
  const double* __restrict__ got = a.goT + (size_t)tile * CO * QO * 128 + lane;
  double* __restrict__ part = a.part + (size_t)tile * a.n_packed + a.wp0[L] + (size_t)c * NB * COT * 2;
  cx<double> go[COT][ROWS];
#pragma unroll
  for (int o = 0; o < COT; ++o) {
    const int oo = o < CO ? o : CO - 1;
#pragma unroll
    for (int mm = 0; mm < ROWS; ++mm) {
      go[o][mm] = {got[(size_t)(oo * QO + QBASE + mm) * 128], got[(size_t)(oo * QO + QBASE + mm) * 128 + 64]};
      if (o >= CO) go[o][mm] = {0, 0};
    }
  }
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    cx<double> w[COT];
#pragma unroll
    for (int o = 0; o < COT; ++o) w[o] = {wc[(blk * COT + o) * 2], wc[(blk * COT + o) * 2 + 1]};
    double dw[2 * COT];
#pragma unroll
    for (int k = 0; k < 2 * COT; ++k) dw[k] = 0.0;
#pragma unroll
    for (int mm = 0; mm < ROWS; ++mm) {
      const int row = ROW0 + blk * D + M0 + mm;
      cx<double> gc = {0, 0};
#pragma unroll
      for (int o = 0; o < COT; ++o) cfmac(gc, go[o][mm], w[o]);
      cx<double> cat = {0, 0};
#pragma unroll
      for (int t = T::ROW_PTR[row]; t < T::ROW_PTR[row + 1]; ++t) {
        const int ty = T::T_TYPE[t], ia = T::T_A[t], ib = T::T_B[t];
        const double cf = T::T_COEF[t];
        const cx<double> g = {cf * gc.r, cf * gc.i};
        if (ty == 0) {
          cat.r = __builtin_fma(cf, uc[ia * 128], cat.r);
          cat.i = __builtin_fma(cf, uc[ia * 128 + 64], cat.i);
          if (T::T_UFIRST[t]) {
            guc[ia * 128] = g.r;
            guc[ia * 128 + 64] = g.i;
          } else {
            guc[ia * 128] += g.r;
            guc[ia * 128 + 64] += g.i;
          }
        } else if (ty == 1) {
          cat.r = __builtin_fma(cf, x[ia].r, cat.r);
          cat.i = __builtin_fma(cf, x[ia].i, cat.i);
          gx_add(gxl, ia, g);
        } else {
          const cx<double> p = cmul(x[ia], x[ib]);
          cat.r = __builtin_fma(cf, p.r, cat.r);
          cat.i = __builtin_fma(cf, p.i, cat.i);
          gx_add(gxl, ia, cmulc(g, x[ib]));
          gx_add(gxl, ib, cmulc(g, x[ia]));
        }
      }
#pragma unroll
      for (int o = 0; o < COT; ++o) {
        cx<double> d = {dw[2 * o], dw[2 * o + 1]};
        cfmac(d, go[o][mm], cat);
        dw[2 * o] = d.r;
        dw[2 * o + 1] = d.i;
      }
      __builtin_amdgcn_sched_barrier(0);               // one row at a time: keeps the live set (and the spills) bounded
    }
#ifdef SB_NO_DW
    const double sum = dw[0] + dw[1];
#else
    const double sum = wave_transpose_sum<2 * COT>(dw, red, lane);
#endif
    if ((lane & 3) == 0 && (lane >> 2) < 2 * COT) {
      double* dst = part + blk * COT * 2 + (lane >> 2);
      if (M0 == 0) *dst = sum;
      else *dst += sum;                                // later row chunks of the same irrep add to the first chunk's partial
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <class T, int COT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void local_bwd_static_kernel(StaticBwdArgs a) {
  static_assert(T::N_OUT == 5 && T::DIM[0] == 4 && T::DIM[1] == 3 && T::DIM[2] == 3 && T::DIM[3] == 9 && T::DIM[4] == 1 && T::CHUNK == 3,
                "item list / walk order of T_UFIRST");
  __shared__ double red[2 * COT * RED_PITCH];
  __shared__ double gxs[T::Q * 128];
  constexpr int Q = T::Q;
  const int tile = blockIdx.x, c = blockIdx.y, lane = threadIdx.x, C = a.C;
  const double* __restrict__ xc = a.XT + ((size_t)tile * C + c) * Q * 128 + lane;
  double* gxl = gxs + 2 * lane;
  cx<double> x[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    x[q] = {xc[q * 128], xc[q * 128 + 64]};
    gxl[q * 128] = 0.0;
    gxl[q * 128 + 1] = 0.0;
  }
  {  // moments that feed no row get a zero gradient
    double* __restrict__ guc = a.gUT + ((size_t)tile * C + c) * Q * 640 + lane;
#pragma unroll
    for (int k = 0; k < T::N_UNUSED; ++k) {
      guc[T::U_UNUSED[k] * 128] = 0.0;
      guc[T::U_UNUSED[k] * 128 + 64] = 0.0;
    }
  }
  // walk order = the order tools/gen_static_tables.py assumed for T_UFIRST: irrep, chunk of 3 rows, block, row, term
#ifndef SB_ONLY3
  item_bwd<T, 0, 0, 3, COT>(a, tile, lane, c, x, gxl, red);
  __builtin_amdgcn_sched_barrier(0);
  item_bwd<T, 0, 3, 1, COT>(a, tile, lane, c, x, gxl, red);
  __builtin_amdgcn_sched_barrier(0);
  item_bwd<T, 1, 0, 3, COT>(a, tile, lane, c, x, gxl, red);
  __builtin_amdgcn_sched_barrier(0);
  item_bwd<T, 2, 0, 3, COT>(a, tile, lane, c, x, gxl, red);
  __builtin_amdgcn_sched_barrier(0);
#endif
  item_bwd<T, 3, 0, 3, COT>(a, tile, lane, c, x, gxl, red);
  __builtin_amdgcn_sched_barrier(0);
  item_bwd<T, 3, 3, 3, COT>(a, tile, lane, c, x, gxl, red);
  __builtin_amdgcn_sched_barrier(0);
  item_bwd<T, 3, 6, 3, COT>(a, tile, lane, c, x, gxl, red);
  __builtin_amdgcn_sched_barrier(0);
  item_bwd<T, 4, 0, 1, COT>(a, tile, lane, c, x, gxl, red);
  __builtin_amdgcn_sched_barrier(0);
  double* __restrict__ gxc = a.gXT + ((size_t)tile * C + c) * Q * 128 + lane;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    gxc[q * 128] = gxl[q * 128];
    gxc[q * 128 + 64] = gxl[q * 128 + 1];
  }
}

// packed partial-gradient rows -> CatMix parameter layout (inverse of pack_weights_kernel), ADDED to gw
__global__ __launch_bounds__(BLOCK) void unpack_weight_grads_kernel(PackArgs p, const double* __restrict__ gp, double* __restrict__ gw) {
  for (int l = 0; l < p.n_out; ++l) {
    const int nb = p.nblk[l], K = nb * p.C, total = p.C * nb * p.COT;
    for (int e = blockIdx.x * BLOCK + threadIdx.x; e < total; e += gridDim.x * BLOCK) {
      const int o = e % p.COT, blk = (e / p.COT) % nb, c = e / (p.COT * nb);
      if (o < p.CO) {
        gw[p.w0[l] + (size_t)o * K + blk * p.C + c] += gp[p.wp0[l] + 2 * e];
        gw[p.w0[l] + (size_t)p.CO * K + (size_t)o * K + blk * p.C + c] += gp[p.wp0[l] + 2 * e + 1];
      }
    }
  }
}

// items of a level kind: every output irrep in chunks of <= 2 rows (few accumulators: two waves per SIMD); both kinds have output dims 4,3,3,9,1
constexpr int N_ITEMS = 12;
template <class T, int COT>
__global__ __launch_bounds__(64) void local_fwd_static_kernel(StaticArgs a) {
  static_assert(T::N_OUT == 5 && T::DIM[0] == 4 && T::DIM[1] == 3 && T::DIM[2] == 3 && T::DIM[3] == 9 && T::DIM[4] == 1, "item list");
  const int tile = blockIdx.x, lane = threadIdx.x;      // (the last tile's padding lanes compute on whatever the padding holds)
  switch (blockIdx.y) {      // item-major grid: neighbouring workgroups run the same straight-line code
    case 0: item_fwd<T, 3, 0, 2, COT>(a, tile, lane); break;
    case 1: item_fwd<T, 3, 2, 2, COT>(a, tile, lane); break;
    case 2: item_fwd<T, 3, 4, 2, COT>(a, tile, lane); break;
    case 3: item_fwd<T, 3, 6, 2, COT>(a, tile, lane); break;
    case 4: item_fwd<T, 3, 8, 1, COT>(a, tile, lane); break;
    case 5: item_fwd<T, 0, 0, 2, COT>(a, tile, lane); break;
    case 6: item_fwd<T, 0, 2, 2, COT>(a, tile, lane); break;
    case 7: item_fwd<T, 1, 0, 2, COT>(a, tile, lane); break;
    case 8: item_fwd<T, 1, 2, 1, COT>(a, tile, lane); break;
    case 9: item_fwd<T, 2, 0, 2, COT>(a, tile, lane); break;
    case 10: item_fwd<T, 2, 2, 1, COT>(a, tile, lane); break;
    default: item_fwd<T, 4, 0, 1, COT>(a, tile, lane); break;
  }
}

template <class T>
static void fill_pack(PackArgs& p, int C, int CO, const int* w0) {
  p.C = C; p.CO = CO; p.COT = CO <= 4 ? 4 : (CO <= 6 ? 6 : 8); p.n_out = T::N_OUT;
  int off = 0;
  for (int l = 0; l < T::N_OUT; ++l) {
    p.nblk[l] = T::NBLK[l];
    p.w0[l] = w0[l];
    p.wp0[l] = off;
    off += C * T::NBLK[l] * p.COT * 2;
  }
}

}  // namespace

// doubles of the packed weight image of a level
size_t local_static_packed_doubles(int kind, int C, int CO) {
  const int cot = CO <= 4 ? 4 : (CO <= 6 ? 6 : 8);
  int nb = 0;
  if (kind == 1) for (int l = 0; l < cgs::Kind1::N_OUT; ++l) nb += cgs::Kind1::NBLK[l];
  else for (int l = 0; l < cgs::Kind2::N_OUT; ++l) nb += cgs::Kind2::NBLK[l];
  return (size_t)C * nb * cot * 2;
}

// kind: 1 / 2 (cg_static_tables.hpp); w0: per-irrep weight offsets in wcat (host array of 5); wp: scratch of
// local_static_packed_doubles(kind, C, CO) doubles (written here); layouts: see the head of this file
int local_fwd_static(int kind, int M, int C, int CO, const double* XT, const double* UT, const double* w, const int* w0, double* wp,
                     double* outT, double* s_copy, int q_s, hipStream_t st) {
  LGN_CHECK_ARG(kind == 1 || kind == 2, "local_fwd_static: unknown level kind %d", kind);
  LGN_CHECK_ARG(M > 0 && C >= 1 && C <= 8 && CO >= 1 && CO <= COMAX, "local_fwd_static: unsupported shape (M=%d C=%d CO=%d)", M, C, CO);
  PackArgs p{};
  if (kind == 1) fill_pack<cgs::Kind1>(p, C, CO, w0); else fill_pack<cgs::Kind2>(p, C, CO, w0);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(8), dim3(BLOCK), 0, st, p, w, wp);
  LGN_CHECK_LAUNCH();
  StaticArgs a{M, C, CO, XT, UT, wp, {0, 0, 0, 0, 0, 0, 0, 0}, outT, s_copy, q_s};
  for (int l = 0; l < 5; ++l) a.wp0[l] = p.wp0[l];
  dim3 grid(cdiv(M, 64), N_ITEMS);
#define LGN_LAUNCH(KIND, COT) hipLaunchKernelGGL((local_fwd_static_kernel<cgs::KIND, COT>), grid, dim3(64), 0, st, a)
  if (kind == 1) { if (CO <= 4) LGN_LAUNCH(Kind1, 4); else if (CO <= 6) LGN_LAUNCH(Kind1, 6); else LGN_LAUNCH(Kind1, 8); }
  else { if (CO <= 4) LGN_LAUNCH(Kind2, 4); else if (CO <= 6) LGN_LAUNCH(Kind2, 6); else LGN_LAUNCH(Kind2, 8); }
#undef LGN_LAUNCH
  LGN_CHECK_LAUNCH();
  return 0;
}


// wp: packed weights of the level (as written by local_fwd_static; repacked here); part: [ceil(M / 64)][packed doubles] scratch
// (every entry written); gpacked: packed doubles (the caller reduces `part` over the tiles into it, then calls
// local_static_unpack_grads).  goT / gUT / gXT: see StaticBwdArgs.
int local_bwd_static(int kind, int M, int C, int CO, const double* XT, const double* UT, const double* w, const int* w0, double* wp,
                     const double* goT, double* gUT, double* gXT, double* part, hipStream_t st) {
  LGN_CHECK_ARG(kind == 1 || kind == 2, "local_bwd_static: unknown level kind %d", kind);
  LGN_CHECK_ARG(M > 0 && C >= 1 && C <= 8 && CO >= 1 && CO <= COMAX, "local_bwd_static: unsupported shape (M=%d C=%d CO=%d)", M, C, CO);
  PackArgs p{};
  if (kind == 1) fill_pack<cgs::Kind1>(p, C, CO, w0); else fill_pack<cgs::Kind2>(p, C, CO, w0);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(8), dim3(BLOCK), 0, st, p, w, wp);
  LGN_CHECK_LAUNCH();
  StaticBwdArgs a{M, C, CO, XT, UT, wp, {0, 0, 0, 0, 0, 0, 0, 0}, goT, gUT, gXT, part, (int)local_static_packed_doubles(kind, C, CO)};
  for (int l = 0; l < 5; ++l) a.wp0[l] = p.wp0[l];
  dim3 grid(cdiv(M, 64), C);
#define LGN_LAUNCH(KIND, COT) hipLaunchKernelGGL((local_bwd_static_kernel<cgs::KIND, COT>), grid, dim3(64), 0, st, a)
  if (kind == 1) { if (CO <= 4) LGN_LAUNCH(Kind1, 4); else if (CO <= 6) LGN_LAUNCH(Kind1, 6); else LGN_LAUNCH(Kind1, 8); }
  else { if (CO <= 4) LGN_LAUNCH(Kind2, 4); else if (CO <= 6) LGN_LAUNCH(Kind2, 6); else LGN_LAUNCH(Kind2, 8); }
#undef LGN_LAUNCH
  LGN_CHECK_LAUNCH();
  return 0;
}

// gw (CatMix parameter layout, irrep l at w0[l]) += unpacked gpacked
int local_static_unpack_grads(int kind, int C, int CO, const int* w0, const double* gpacked, double* gw, hipStream_t st) {
  PackArgs p{};
  if (kind == 1) fill_pack<cgs::Kind1>(p, C, CO, w0); else fill_pack<cgs::Kind2>(p, C, CO, w0);
  hipLaunchKernelGGL(unpack_weight_grads_kernel, dim3(8), dim3(BLOCK), 0, st, p, gpacked, gw);
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace lgn
