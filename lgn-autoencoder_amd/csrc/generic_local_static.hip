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
#include "local_static_dev.hpp"

namespace lgn {
namespace {
using namespace lsd;
LGN_STAMP_DECL
#ifdef LGN_STAMPS
#define SSTAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_stamps[i] = clock64(); } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif

// ---- weight packing ----------------------------------------------------------------------------------------------------
// wcat (irrep l at w0[l]: [2][CO][nblk_l * C], the CatMix parameter layout)  ->  wp (irrep l at wp0[l]: [C][nblk_l][COT][2])
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

template <class T, int L, int BLK, int BEND, int COT>
__device__ __forceinline__ void blocks_bwd(const cx<double> (&go)[COT][T::DIM[L]], cx<double> (&wn)[COT],
                                           const double __attribute__((address_space(4))) * wc, const double* __restrict__ uc,
                                           double* guc, const cx<double> (&uv)[blk_nu1<T>(L, BLK)],
                                           const cx<double> (&gold)[blk_nu1<T>(L, BLK)], const double* xl, double* gxl,
                                           const DwOut& part, int lane, bool valid) {
  constexpr int D = T::DIM[L], ROWB = T::ROW0[L] + BLK * D;
  if constexpr (BLK < BEND) {
    constexpr int A0 = blk_lo<T>(ROWB, D, 0), A1 = blk_hi<T>(ROWB, D, 0), B0 = blk_lo<T>(ROWB, D, 1), B1 = blk_hi<T>(ROWB, D, 1);
    constexpr int NA = A1 > A0 ? A1 - A0 : 1, NB = B1 > B0 ? B1 - B0 : 1, NU = blk_nu<T>(L, BLK), NU1 = blk_nu1<T>(L, BLK);
    cx<double> w[COT];
#pragma unroll
    for (int o = 0; o < COT; ++o) w[o] = wn[o];
    // software pipeline: the next block's weights and moments are in flight during this block
    cx<double> uvn[blk_nu1<T>(L, BLK + 1)], goldn[blk_nu1<T>(L, BLK + 1)];
    if constexpr (BLK + 1 < BEND) {
#pragma unroll
      for (int o = 0; o < COT; ++o) wn[o] = {wc[((BLK + 1) * COT + o) * 2], wc[((BLK + 1) * COT + o) * 2 + 1]};
      load_moments<T, L, BLK + 1>(uc, guc, uvn, goldn, valid);
      if constexpr (blk_nu<T>(L, BLK + 1) > 0) __builtin_amdgcn_sched_barrier(0);      // (issued here, not wherever the scheduler likes)
    }
    // the features the block reads, and register accumulators for the gradients of what it reads
    cx<double> xa[NA], xb[NB], ga[NA], gb[NB], gu[NU1];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      if (A1 > A0) xa[k] = {xl[(A0 + k) * 128], xl[(A0 + k) * 128 + 1]};
      ga[k] = {0, 0};
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if (B1 > B0) xb[k] = {xl[(B0 + k) * 128], xl[(B0 + k) * 128 + 1]};
      gb[k] = {0, 0};
    }
#pragma unroll
    for (int k = 0; k < NU1; ++k) gu[k] = gold[k];
    double dw[2 * COT];
#pragma unroll
    for (int k = 0; k < 2 * COT; ++k) dw[k] = 0.0;
    rows_bwd<T, ROWB, 0, D, COT, A0, NA, B0, NB, NU1>(go, w, UvRef<NU1>{uv}, gu, xa, ga, xb, gb, dw);
    if constexpr (A1 > A0) gx_flush<NA>(gxl, A0, ga);
    if constexpr (B1 > B0) gx_flush<NB>(gxl, B0, gb);
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const int e = T::BLK_UELEM[blk_u0<T>(L, BLK) + k];
      guc[e * 128] = gu[k].r;
      guc[e * 128 + 64] = gu[k].i;
    }
    dw_store<2 * COT>(dw, part, BLK, lane);
    __builtin_amdgcn_sched_barrier(0);
    blocks_bwd<T, L, BLK + 1, BEND, COT>(go, wn, wc, uc, guc, uvn, goldn, xl, gxl, part, lane, valid);
  }
}

// blocks BBEG .. BEND - 1 of one output irrep, all rows (walk order of T_UFIRST: irrep, block, row, term).  The upstream
// gradient rows are loaded once and stay in registers over the blocks.
template <class T, int L, int BBEG, int BEND, int COT>
__device__ __forceinline__ void irrep_bwd(const StaticBwdArgs& a, int tile, int lane, int c, const double* xl, double* gxl) {
  const bool valid = tile * 64 + lane < a.M;
  if constexpr (BBEG < BEND) {
    constexpr int D = T::DIM[L], NB = T::NBLK[L], Q = T::Q, QO = T::QOUT, QBASE = T::Q0[L];
    static_assert(BEND <= NB, "block range");
    const int C = a.C, CO = a.CO;
    typedef const double __attribute__((address_space(4))) * cptr;
    cptr wc = (cptr)(a.wp + a.wp0[L]) + (size_t)c * NB * COT * 2;
    const double* __restrict__ uc = a.UT + ((size_t)tile * C + c) * Q * 640 + lane;
    double* guc = a.gUT + ((size_t)tile * C + c) * Q * 640 + lane;
    const double* __restrict__ got = a.goT + (size_t)tile * CO * QO * 128 + lane;
    const DwOut part = dw_out<COT>(a, a.part + (size_t)tile * a.n_packed, L, NB, c, lane);
    cx<double> go[COT][D];
#pragma unroll
    for (int o = 0; o < COT; ++o) {
      const int oo = o < CO ? o : CO - 1;
#pragma unroll
      for (int mm = 0; mm < D; ++mm) {
        go[o][mm] = {got[(size_t)(oo * QO + QBASE + mm) * 128], got[(size_t)(oo * QO + QBASE + mm) * 128 + 64]};
        if (o >= CO || !valid) go[o][mm] = {0, 0};
      }
    }
    cx<double> wn[COT];
#pragma unroll
    for (int o = 0; o < COT; ++o) wn[o] = {wc[(BBEG * COT + o) * 2], wc[(BBEG * COT + o) * 2 + 1]};
    cx<double> uv[blk_nu1<T>(L, BBEG)], gold[blk_nu1<T>(L, BBEG)];
    load_moments<T, L, BBEG>(uc, guc, uv, gold, valid);
    blocks_bwd<T, L, BBEG, BEND, COT>(go, wn, wc, uc, guc, uv, gold, xl, gxl, part, lane, valid);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Workgroup = (tile of 64 nodes, input channel c), two waves that share the channel's features in LDS and split the blocks
// (the kernel holds one wave per SIMD -- 512 registers -- and the grid is ~1.4 such rounds: half-size jobs fill the tail):
//   wave 0: every moment / feature block (it alone touches U and d U), the product blocks of irreps 0 and 1 (but the last)
//   wave 1: the product blocks of irreps 2, 3, 4 and the last one of irrep 1        (balanced on measured wave times)
// Each wave accumulates d X in its own LDS image; the two are added at the end (fixed order: deterministic).
template <class T, int COT>
__global__ __launch_bounds__(128) void local_bwd_static_kernel(StaticBwdArgs a) {
  static_assert(T::N_OUT == 5 && T::NUBLK[1] < T::NBLK[1], "wave assignment");
  constexpr int Q = T::Q;
  __shared__ double xs[Q * 128];                       // this channel's features, lane-private columns [q][lane][2]
  __shared__ double gxs[2][Q * 128];                   // d X per wave, same layout
  int tile, c;
  if (!xcd_index((a.M + 63) >> 6, a.C, tile, c)) return;           // (workgroup-uniform: before any barrier)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, C = a.C;
  {
    const double* __restrict__ xc = a.XT + ((size_t)tile * C + c) * Q * 128;
    for (int e = threadIdx.x; e < Q * 128; e += 128) {
      // padding lanes of the last tile (node >= M): zeros, whatever the buffer holds there (see load_moments)
      xs[(e & ~127) + 2 * (e & 63) + ((e >> 6) & 1)] = tile * 64 + (e & 63) < a.M ? xc[e] : 0.0;
      gxs[0][e] = 0.0;
      gxs[1][e] = 0.0;
    }
    // moments that feed no row get a zero gradient
    double* __restrict__ guc = a.gUT + ((size_t)tile * C + c) * Q * 640;
    for (int e = threadIdx.x; e < T::N_UNUSED * 128; e += 128) guc[T::U_UNUSED[e >> 7] * 128 + (e & 127)] = 0.0;
  }
  __syncthreads();
  const double* xl = xs + 2 * lane;
  double* gxl = gxs[wave] + 2 * lane;
  // (round 6, in-kernel stamps of the 6 -> 6 level: wave 0 141 k cycles, wave 1 113 k -- the last four product blocks of irrep 0, 13 k,
  // moved to wave 1 for the later levels)
  // (first level, stamps of the encoder's level 0: wave 0 38 k cycles, wave 1 18 k -- its product blocks of irrep 0 go to wave 1 as
  // well; the moment blocks, two thirds of wave 0's time, cannot: dU is read-modify-written by one wave only)
  constexpr int B0 = T::Q == 20 ? T::NBLK[0] - 4 : T::NUBLK[0];
  static_assert(B0 >= T::NUBLK[0], "wave 1 takes product blocks only");
  if (wave == 0) {
    SSTAMP(0);
    irrep_bwd<T, 0, 0, B0, COT>(a, tile, lane, c, xl, gxl);
    SSTAMP(1);
    irrep_bwd<T, 1, 0, T::NBLK[1] - 1, COT>(a, tile, lane, c, xl, gxl);
    SSTAMP(2);
    irrep_bwd<T, 2, 0, T::NUBLK[2], COT>(a, tile, lane, c, xl, gxl);
    SSTAMP(3);
    irrep_bwd<T, 3, 0, T::NUBLK[3], COT>(a, tile, lane, c, xl, gxl);
    SSTAMP(4);
    irrep_bwd<T, 4, 0, T::NUBLK[4], COT>(a, tile, lane, c, xl, gxl);
    SSTAMP(5);
  } else {
    SSTAMP(10);
    irrep_bwd<T, 2, T::NUBLK[2], T::NBLK[2], COT>(a, tile, lane, c, xl, gxl);
    SSTAMP(11);
    irrep_bwd<T, 3, T::NUBLK[3], T::NBLK[3], COT>(a, tile, lane, c, xl, gxl);
    SSTAMP(12);
    irrep_bwd<T, 4, T::NUBLK[4], T::NBLK[4], COT>(a, tile, lane, c, xl, gxl);
    irrep_bwd<T, 1, T::NBLK[1] - 1, T::NBLK[1], COT>(a, tile, lane, c, xl, gxl);
    irrep_bwd<T, 0, B0, T::NBLK[0], COT>(a, tile, lane, c, xl, gxl);
    SSTAMP(13);
  }
  __syncthreads();
  SSTAMP(6);
  double* __restrict__ gxc = a.gXT + ((size_t)tile * C + c) * Q * 128;
  for (int e = threadIdx.x; e < Q * 128; e += 128) {
    const int k = (e & ~127) + 2 * (e & 63) + ((e >> 6) & 1);
    gxc[e] = gxs[0][k] + gxs[1][k];
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

// several levels per launch (blockIdx.y = job): the sequencers pack every level of a network before its forward and unpack
// every level's gradient after the deferred reductions -- one launch each instead of one per level
constexpr int PACK_MAX_JOBS = 8;
struct PackBatch {
  PackArgs p[PACK_MAX_JOBS];
  const double* src[PACK_MAX_JOBS];
  double* dst[PACK_MAX_JOBS];
};
__global__ __launch_bounds__(BLOCK) void pack_batch_kernel(PackBatch b, int unpack) {
  const PackArgs& p = b.p[blockIdx.y];
  const double* __restrict__ src = b.src[blockIdx.y];
  double* __restrict__ dst = b.dst[blockIdx.y];
  for (int l = 0; l < p.n_out; ++l) {
    const int nb = p.nblk[l], K = nb * p.C, total = p.C * nb * p.COT;
    for (int e = blockIdx.x * BLOCK + threadIdx.x; e < total; e += gridDim.x * BLOCK) {
      const int o = e % p.COT, blk = (e / p.COT) % nb, c = e / (p.COT * nb);
      const size_t ire = p.w0[l] + (size_t)o * K + blk * p.C + c, iim = ire + (size_t)p.CO * K;
      if (unpack) {
        if (o < p.CO) {
          dst[ire] += src[p.wp0[l] + 2 * e];
          dst[iim] += src[p.wp0[l] + 2 * e + 1];
        }
      } else {
        dst[p.wp0[l] + 2 * e] = o < p.CO ? src[ire] : 0.0;
        dst[p.wp0[l] + 2 * e + 1] = o < p.CO ? src[iim] : 0.0;
      }
    }
  }
}

// (the forward's twelve items of a tile re-read its features and moments as well, but giving THEM neighbouring ids was measured slower
// -- 49 / 79 -> 53 / 86 us at cfg5: a CU then holds workgroups of several items, i.e. several straight-line code images)
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

}  // namespace
LGN_STAMP_READER(lgn_debug_stamps_local_static)

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
                     double* outT, double* s_copy, int q_s, hipStream_t st, bool packed) {
  LGN_CHECK_ARG(kind == 1 || kind == 2, "local_fwd_static: unknown level kind %d", kind);
  LGN_CHECK_ARG(M > 0 && C >= 1 && C <= 8 && CO >= 1 && CO <= COMAX, "local_fwd_static: unsupported shape (M=%d C=%d CO=%d)", M, C, CO);
  PackArgs p{};
  if (kind == 1) fill_pack<cgs::Kind1>(p, C, CO, w0); else fill_pack<cgs::Kind2>(p, C, CO, w0);
  if (!packed) {
    hipLaunchKernelGGL(pack_weights_kernel, dim3(8), dim3(BLOCK), 0, st, p, w, wp);
    LGN_CHECK_LAUNCH();
  }
  StaticArgs a{M, C, CO, XT, UT, nullptr, nullptr, 0, wp, {0, 0, 0, 0, 0, 0, 0, 0}, outT, s_copy, q_s};
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
                     const double* goT, double* gUT, double* gXT, double* part, hipStream_t st, bool packed, bool param_layout) {
  LGN_CHECK_ARG(kind == 1 || kind == 2, "local_bwd_static: unknown level kind %d", kind);
  LGN_CHECK_ARG(M > 0 && C >= 1 && C <= 8 && CO >= 1 && CO <= COMAX, "local_bwd_static: unsupported shape (M=%d C=%d CO=%d)", M, C, CO);
  PackArgs p{};
  if (kind == 1) fill_pack<cgs::Kind1>(p, C, CO, w0); else fill_pack<cgs::Kind2>(p, C, CO, w0);
  if (!packed) {       // (the sequencers keep the forward's packed image)
    hipLaunchKernelGGL(pack_weights_kernel, dim3(8), dim3(BLOCK), 0, st, p, w, wp);
    LGN_CHECK_LAUNCH();
  }
  StaticBwdArgs a{M, C, CO, XT, UT, wp, {0, 0, 0, 0, 0, 0, 0, 0}, goT, gUT, gXT, part, (int)local_static_packed_doubles(kind, C, CO)};
  for (int l = 0; l < 5; ++l) { a.wp0[l] = p.wp0[l]; a.w0[l] = w0[l]; }
  a.param_layout = param_layout ? 1 : 0;
  dim3 grid(xcd_grid(cdiv(M, 64), C));
#define LGN_LAUNCH(KIND, COT) hipLaunchKernelGGL((local_bwd_static_kernel<cgs::KIND, COT>), grid, dim3(128), 0, st, a)
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

// n <= 8 jobs in one launch.  unpack = false: dst = packed image of the CatMix weights src;  unpack = true: dst (CatMix
// parameter layout) += unpacked src (packed gradients)
int local_static_pack_batch(const StaticPackJob* jobs, int n, bool unpack, hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += PACK_MAX_JOBS) {
    PackBatch b{};
    const int m = n - i0 < PACK_MAX_JOBS ? n - i0 : PACK_MAX_JOBS;
    for (int i = 0; i < m; ++i) {
      const StaticPackJob& j = jobs[i0 + i];
      LGN_CHECK_ARG((j.kind == 1 || j.kind == 2) && j.C >= 1 && j.C <= 8 && j.CO >= 1 && j.CO <= COMAX && j.src && j.dst,
                    "local_static_pack_batch: bad job %d (kind=%d C=%d CO=%d)", i0 + i, j.kind, j.C, j.CO);
      if (j.kind == 1) fill_pack<cgs::Kind1>(b.p[i], j.C, j.CO, j.w0); else fill_pack<cgs::Kind2>(b.p[i], j.C, j.CO, j.w0);
      b.src[i] = j.src;
      b.dst[i] = j.dst;
    }
    hipLaunchKernelGGL(pack_batch_kernel, dim3(8, m), dim3(BLOCK), 0, st, b, unpack ? 1 : 0);
    LGN_CHECK_LAUNCH();
  }
  return 0;
}

}  // namespace lgn
