// lgn-autoencoder_amd/csrc/mlp_mfma_wide.hip -- CGMLP forward / backward on the fp64 matrix cores for hidden widths
// 48 < H <= 96 (C = 5..8 at mlp_width 6; cfg5 has C = 6 -> H = 72).  Same operator and the same MFMA fragment layouts as
// mlp_mfma.hip (reference: CGMLP.forward, lgn/models/lgn_levels.py:191-227); what changes is the budget:
//   * NT = 4..6 N-tiles: a weight image is 34..75 KB, so it is single buffered and a layer costs two barriers
//     (operands visible / all reads done) instead of one; the tiles need no ping-pong then either;
//   * a workgroup runs 2 M-tiles x NT N-tiles = 8..12 waves (one 16x16 output tile per wave);
//   * the backward covers its workgroup's 64 rows in ONE pass: a wave owns the tiles (mp, nt) and (mp + 2, nt) of the four
//     16-row M-tiles (two accumulation chains that share every weight fragment).  Round 2 made two 32-row passes, the second
//     adding into the partial row the first had written: 450 MB of HBM traffic per launch at cfg5 (PMC, profiles/
//     r03_pmc_cfg5.json) for 54 MB of partial rows, and every weight image staged twice.  The hidden activations of the
//     recompute stay in registers (2 tiles x 6 layers; the LDS is taken by the 4 + 4 operand tiles of the weight gradient).
#include "ops.hpp"

namespace lgn {
namespace wide {

typedef double v4d __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int pad4(int x) { return (x + 3) & ~3; }
__host__ __device__ constexpr int lds_stride(int x) { return x + ((6 - (x & 3)) & 3); }   // == 2 (mod 4): conflict-free b64 reads

constexpr int MT = 2;

template <int NT>
struct Geo {
  static constexpr int HP = NT * 16;
  static constexpr int S = lds_stride(HP);
  static constexpr int S0 = 18;
  static constexpr int WSIZE = HP * S;
  static constexpr int TSIZE = 16 * S;
  static constexpr int T0SIZE = 16 * S0;
  static constexpr int THREADS = 64 * MT * NT;
  static constexpr int RPP = THREADS / HP;              // weight rows staged per pass (= 8)
  static constexpr int NPH = HP / RPP;                  // passes for a hidden / output layer image
  static constexpr int NPF = (HP * 16 + THREADS - 1) / THREADS;   // passes for the first layer image (16 columns)
  static_assert(NPF <= NPH, "prefetch registers");
  static constexpr size_t fwd_doubles() { return WSIZE + MT * TSIZE + MT * T0SIZE; }
  static constexpr int MTB = 4;                         // M-tiles of a backward workgroup (64 rows)
  //                                          image   X, G tiles       input tiles    column sums
  static constexpr size_t bwd_doubles() { return WSIZE + 2 * MTB * TSIZE + MTB * T0SIZE + MTB * HP; }
  static constexpr bool ONE_PASS = sizeof(double) * (WSIZE + 2 * MTB * TSIZE + MTB * T0SIZE + MTB * HP) <= 160 * 1024;
  // two-pass backward (NT = 6): the activations of the first NHL hidden layers stay in LDS tiles of their own
  static constexpr int BWD_BASE = WSIZE + 2 * MT * TSIZE + MT * T0SIZE + MT * HP;
  static constexpr int NHL_FIT = (160 * 1024 / 8 - BWD_BASE) / (MT * TSIZE);
  static constexpr int NHL = NHL_FIT > 5 ? 5 : NHL_FIT;
  static constexpr size_t bwd2p_doubles() { return BWD_BASE + NHL * MT * TSIZE; }
};

// ---- weight staging: thread-constant bases, wave-uniform strides (see mlp_mfma.hip) ---------------------------------
template <int NT>
__device__ __forceinline__ void prefetch_hidden(const double* __restrict__ W, const double* __restrict__ bias, int Hout, int Hin,
                                                double (&regs)[Geo<NT>::NPH], double& breg) {
  using G = Geo<NT>;
  const int o0 = (int)threadIdx.x / G::HP, k = (int)threadIdx.x - o0 * G::HP;
  const double* src = W + (o0 * Hin + k);
#pragma unroll
  for (int i = 0; i < G::NPH; ++i) regs[i] = (k < Hin && o0 + G::RPP * i < Hout) ? src[G::RPP * i * Hin] : 0.0;
  breg = ((int)threadIdx.x < Hout) ? bias[threadIdx.x] : 0.0;
}
template <int NT>
__device__ __forceinline__ void commit_hidden(double* Wl, const double (&regs)[Geo<NT>::NPH], double breg) {
  using G = Geo<NT>;
  const int o0 = (int)threadIdx.x / G::HP, k = (int)threadIdx.x - o0 * G::HP;
  double* dst = Wl + (o0 * G::S + k);
#pragma unroll
  for (int i = 0; i < G::NPH; ++i) dst[G::RPP * i * G::S] = regs[i];
  if ((int)threadIdx.x < G::HP) Wl[threadIdx.x * G::S + G::HP] = breg;
}
template <int NT>
__device__ __forceinline__ void prefetch_first(const double* __restrict__ W, const double* __restrict__ bias, int Hout, int Hin,
                                               double (&regs)[Geo<NT>::NPH], double& breg) {
  using G = Geo<NT>;
  const int o0 = (int)threadIdx.x >> 4, k = (int)threadIdx.x & 15;
#pragma unroll
  for (int i = 0; i < G::NPF; ++i) {
    const int o = o0 + (G::THREADS / 16) * i;
    regs[i] = (o < Hout && k < Hin) ? W[o * Hin + k] : 0.0;
  }
  breg = ((int)threadIdx.x < Hout) ? bias[threadIdx.x] : 0.0;
}
template <int NT>
__device__ __forceinline__ void commit_first(double* Wl, const double (&regs)[Geo<NT>::NPH], double breg) {
  using G = Geo<NT>;
  const int o0 = (int)threadIdx.x >> 4, k = (int)threadIdx.x & 15;
#pragma unroll
  for (int i = 0; i < G::NPF; ++i) {
    const int o = o0 + (G::THREADS / 16) * i;
    if (o < G::HP) Wl[o * G::S + k] = regs[i];
  }
  if ((int)threadIdx.x < G::HP) Wl[threadIdx.x * G::S + G::HP] = breg;
}
// rows of the scalar irrep [2][M][C] -> MT input tiles, feature k = 2c + z, zero padded to 16 columns
template <int NT, int MTILES = MT>
__device__ __forceinline__ void load_input_tiles(const double* __restrict__ s, int M, int C, int row0, double* X0) {
  using G = Geo<NT>;
  const int D = 2 * C;
  for (int e = threadIdx.x; e < 16 * MTILES * 16; e += G::THREADS) {
    const int r = e >> 4, k = e & 15, row = row0 + r;
    X0[(r >> 4) * G::T0SIZE + (r & 15) * G::S0 + k] = (row < M && k < D) ? s[(size_t)(k & 1) * M * C + (size_t)row * C + (k >> 1)] : 0.0;
  }
}

// pre-activation of this wave's tile: bias + X W^T  (l == 0: the 16-column input tiles)
template <int NT>
__device__ __forceinline__ v4d dense(bool first, const double* Wl, const double* X0, const double* X, int mt, int nt, int lane, int ksh) {
  using G = Geo<NT>;
  constexpr int S = G::S;
  const int c = lane & 15, g = lane >> 4;
  const double bias = Wl[(16 * nt + c) * S + G::HP];
  v4d acc = v4d{bias, bias, bias, bias};
  const double* wb = Wl + (16 * nt + c) * S + g;
  if (first) {
    const double* xa = X0 + mt * G::T0SIZE + c * G::S0 + g;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * s], wb[4 * s], acc, 0, 0, 0);
  } else {
    const double* xa = X + mt * G::TSIZE + c * S + g;
    for (int s = 0; s < ksh; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * s], wb[4 * s], acc, 0, 0, 0);
  }
  return acc;
}
template <int NT>
__device__ __forceinline__ void store_tile(double* T, int mt, int nt, int lane, const v4d& v) {
  using G = Geo<NT>;
  double* o = T + mt * G::TSIZE + (lane >> 4) * G::S + 16 * nt + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) o[4 * r * G::S] = v[r];
}

template <int NT, int NH, bool GEN>
__global__ __launch_bounds__(64 * MT * NT) void mlp_fwd_wide_kernel(MlpArgs<double> a) {
  using G = Geo<NT>;
  constexpr int S = G::S;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave % MT, nt = wave / MT;
  const int D = 2 * a.C, H = a.H, M = a.M;
  const int ksh = pad4(H) >> 2;
  const int row0 = blockIdx.x * 16 * MT;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);
  double* X = Wl + G::WSIZE;
  double* X0 = X + MT * G::TSIZE;

  double regs[G::NPH], breg;
  prefetch_first<NT>(a.w[0], a.b[0], H, D, regs, breg);
  load_input_tiles<NT>(a.s_in, M, a.C, row0, X0);
  commit_first<NT>(Wl, regs, breg);
  __syncthreads();
#pragma unroll
  for (int l = 0; l < NH; ++l) {
    prefetch_hidden<NT>(a.w[l + 1], a.b[l + 1], l + 1 == NH ? D : H, H, regs, breg);
    v4d acc = dense<NT>(l == 0, Wl, X0, X, mt, nt, lane, ksh);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = act_apply_t<GEN>(acc[r], a.act);
    __syncthreads();                                       // every read of the weight image and of the layer input is done
    store_tile<NT>(X, mt, nt, lane, acc);
    commit_hidden<NT>(Wl, regs, breg);
    __syncthreads();
  }
  if (nt == 0) {                                           // output layer: one N-tile (2C <= 16 neurons), no activation
    const int c = lane & 15, g = lane >> 4;
    const v4d acc = dense<NT>(false, Wl, X0, X, mt, 0, lane, ksh);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + mt * 16 + g + 4 * r;
      if (c < D && row < M) a.s_out[mlp_out_index(a, c & 1, row, c >> 1)] = acc[r];
    }
  }
  (void)S;
}

// pre-activations of the wave's two tiles (mp, nt) and (mp + 2, nt): bias + X W^T, every weight fragment used twice
template <int NT>
__device__ __forceinline__ void dense2(bool first, const double* Wl, const double* X0, const double* X, int mp, int nt, int lane, int ksh,
                                       v4d& acc0, v4d& acc1) {
  using G = Geo<NT>;
  constexpr int S = G::S;
  const int c = lane & 15, g = lane >> 4;
  const double bias = Wl[(16 * nt + c) * S + G::HP];
  acc0 = v4d{bias, bias, bias, bias};
  acc1 = acc0;
  const double* wb = Wl + (16 * nt + c) * S + g;
  if (first) {
    const double* xa = X0 + mp * G::T0SIZE + c * G::S0 + g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double w = wb[4 * s];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * s], w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2 * G::T0SIZE + 4 * s], w, acc1, 0, 0, 0);
    }
  } else {
    const double* xa = X + mp * G::TSIZE + c * S + g;
    for (int s = 0; s < ksh; ++s) {
      const double w = wb[4 * s];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * s], w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2 * G::TSIZE + 4 * s], w, acc1, 0, 0, 0);
    }
  }
}

template <int NT, int NH, bool GEN>
__global__ __launch_bounds__(64 * MT * NT) void mlp_bwd_wide_kernel(MlpArgs<double> a) {
  using G = Geo<NT>;
  constexpr int S = G::S, HP = G::HP, NW = MT * NT, MTB = G::MTB;
  static_assert(MT == 2 && MTB == 4, "a wave owns M-tiles mp and mp + 2");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mp = wave % MT, nt = wave / MT;
  const int c = lane & 15, g = lane >> 4;
  const int D = 2 * a.C, H = a.H, M = a.M;
  const int ksh = pad4(H) >> 2;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);
  double* X = Wl + G::WSIZE;                               // MTB layer-input tiles
  double* Gt = X + MTB * G::TSIZE;                         // MTB g_pre tiles
  double* X0 = Gt + MTB * G::TSIZE;                        // MTB MLP input tiles
  double* dbw = X0 + MTB * G::T0SIZE;                      // MTB x HP column sums
  double* part = a.part + (size_t)blockIdx.x * a.psize;
  const int row0 = blockIdx.x * 16 * MTB;

  // ---- forward recompute; h[q][l] = post-activation of hidden layer l, tile (mp + 2 q, nt), D layout ------------------------
  double regs[G::NPH], breg;
  prefetch_first<NT>(a.w[0], a.b[0], H, D, regs, breg);
  load_input_tiles<NT, MTB>(a.s_in, M, a.C, row0, X0);
  commit_first<NT>(Wl, regs, breg);
  __syncthreads();
  v4d h[2][NH];
#pragma unroll
  for (int l = 0; l < NH; ++l) {
    prefetch_hidden<NT>(a.w[l + 1], a.b[l + 1], l + 1 == NH ? D : H, H, regs, breg);   // after the last hidden layer: the output layer
    dense2<NT>(l == 0, Wl, X0, X, mp, nt, lane, ksh, h[0][l], h[1][l]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      h[0][l][r] = act_apply_t<GEN>(h[0][l][r], a.act);
      h[1][l][r] = act_apply_t<GEN>(h[1][l][r], a.act);
    }
    __syncthreads();                                       // every read of the input tiles and of the weight image is done
    if (l + 1 < NH) {
      store_tile<NT>(X, mp, nt, lane, h[0][l]);
      store_tile<NT>(X, mp + 2, nt, lane, h[1][l]);
    }
    commit_hidden<NT>(Wl, regs, breg);
    __syncthreads();
  }

  // ---- backward sweep ---------------------------------------------------------------------------------------
  v4d gpre[2] = {v4d{0, 0, 0, 0}, v4d{0, 0, 0, 0}};
  if (nt == 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + (mp + 2 * q) * 16 + g + 4 * r;
        gpre[q][r] = (c < D && row < M) ? a.g_out[mlp_out_index(a, c & 1, row, c >> 1)] : 0.0;
      }
  }
  int poff_end = a.psize;
#pragma unroll
  for (int l = NH; l >= 0; --l) {
    const int Hin = l == 0 ? D : H, Hout = l == NH ? D : H;
    poff_end -= Hout * Hin + Hout;
    double* pW = part + poff_end;
    double* pB = pW + Hout * Hin;
    if (l == 1) prefetch_first<NT>(a.w[0], a.b[0], H, D, regs, breg);
    else if (l > 1) prefetch_hidden<NT>(a.w[l - 1], a.b[l - 1], H, H, regs, breg);

#pragma unroll
    for (int q = 0; q < 2; ++q) {
      store_tile<NT>(Gt, mp + 2 * q, nt, lane, gpre[q]);
      if (l > 0) store_tile<NT>(X, mp + 2 * q, nt, lane, h[q][l > 0 ? l - 1 : 0]);
      double v = (gpre[q][0] + gpre[q][1]) + (gpre[q][2] + gpre[q][3]);   // bias gradient: column sums over the tile's 16 rows
      v += shfl_xor(v, 16);
      v += shfl_xor(v, 32);
      if (g == 0) dbw[(mp + 2 * q) * HP + 16 * nt + c] = v;
    }
    __syncthreads();

    // (a) g_in tiles (mp, nt), (mp + 2, nt) = g_pre W; the first layer has a single (16-column) input tile
    v4d gin[2] = {v4d{0, 0, 0, 0}, v4d{0, 0, 0, 0}};
    if (l > 0 || nt == 0) {
      const double* ga = Gt + mp * G::TSIZE + c * S + g;
      const double* wb = Wl + g * S + 16 * nt + c;
      const int ks = l == NH ? 4 : ksh;                    // K = output neurons of this layer
      for (int s = 0; s < ks; ++s) {
        const double w = wb[4 * s * S];
        gin[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[4 * s], w, gin[0], 0, 0, 0);
        gin[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[2 * G::TSIZE + 4 * s], w, gin[1], 0, 0, 0);
      }
    }
    // (b) dW tiles over the workgroup's 64 rows, dealt round-robin to the waves, two accumulation chains
    {
      const int nti = l == 0 ? 1 : NT, ntiles = (l == NH ? 1 : NT) * nti;
      for (int tile = wave; tile < ntiles; tile += NW) {
        const int t = tile / nti, u = tile - t * nti;
        const double* ga = Gt + g * S + 16 * t + c;
        const double* xb = l == 0 ? X0 + g * G::S0 + c : X + g * S + 16 * u + c;
        const int xts = l == 0 ? G::T0SIZE : G::TSIZE, xss = l == 0 ? G::S0 : S;
        v4d acc0 = v4d{0, 0, 0, 0}, acc1 = v4d{0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < MTB / 2; ++w)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[w * G::TSIZE + 4 * s * S], xb[w * xts + 4 * s * xss], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[(w + 2) * G::TSIZE + 4 * s * S], xb[(w + 2) * xts + 4 * s * xss], acc1, 0, 0, 0);
          }
        const int k = 16 * u + c;                          // D[i = o][j = k]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * t + g + 4 * r;
          if (o < Hout && k < Hin) __builtin_nontemporal_store(acc0[r] + acc1[r], &pW[o * Hin + k]);
        }
      }
      if (tid < Hout) pB[tid] = (dbw[tid] + dbw[HP + tid]) + (dbw[2 * HP + tid] + dbw[3 * HP + tid]);
    }
    if (l > 0) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) gpre[q][r] = gin[q][r] * act_slope_t<GEN>(h[q][l > 0 ? l - 1 : 0][r], a.act);
      __syncthreads();                                     // every read of the weight image and of the tiles is done
      if (l == 1) commit_first<NT>(Wl, regs, breg);
      else commit_hidden<NT>(Wl, regs, breg);
    } else if (nt == 0) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + (mp + 2 * q) * 16 + g + 4 * r;
          if (c < D && row < M) a.g_in[mlp_out_index(a, c & 1, row, c >> 1)] = gin[q][r];
        }
    }
  }
}

// H > 80 (NT = 6: C = 7, 8): the 4 + 4 operand tiles of the one-pass kernel do not fit the LDS beside a 75 KB weight image.
// Round-2 kernel: two 32-row passes per workgroup, the second adding into the partial row the first one wrote; the
// activations of the first NHL hidden layers stay in LDS tiles of their own.
template <int NT, int NH, bool GEN>
__global__ __launch_bounds__(64 * MT * NT) void mlp_bwd_wide_2pass_kernel(MlpArgs<double> a) {
  using G = Geo<NT>;
  constexpr int S = G::S, HP = G::HP, NW = MT * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave % MT, nt = wave / MT;
  const int c = lane & 15, g = lane >> 4;
  const int D = 2 * a.C, H = a.H, M = a.M;
  const int ksh = pad4(H) >> 2;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);
  double* X = Wl + G::WSIZE;                               // MT layer-input tiles
  double* Gt = X + MT * G::TSIZE;                          // MT g_pre tiles
  double* X0 = Gt + MT * G::TSIZE;                         // MT MLP input tiles
  double* dbw = X0 + MT * G::T0SIZE;                       // MT x HP column sums
  double* Hl = dbw + MT * HP;                              // NHL x MT tiles: post-activations of hidden layers 0 .. NHL - 1
  constexpr int NHL = G::NHL;
  double* part = a.part + (size_t)blockIdx.x * a.psize;

  for (int pass = 0; pass < 64 / (16 * MT); ++pass) {      // 32-row passes of this workgroup's 64 rows
    const int row0 = blockIdx.x * 64 + pass * 16 * MT;
    if (pass) __syncthreads();
    // ---- forward recompute; h[l] = post-activation of hidden layer l, this wave's tile, D layout ------------------------
    double regs[G::NPH], breg;
    prefetch_first<NT>(a.w[0], a.b[0], H, D, regs, breg);
    load_input_tiles<NT>(a.s_in, M, a.C, row0, X0);
    commit_first<NT>(Wl, regs, breg);
    __syncthreads();
    v4d h[NH];
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      prefetch_hidden<NT>(a.w[l + 1], a.b[l + 1], l + 1 == NH ? D : H, H, regs, breg);   // after the last hidden layer: the output layer
      h[l] = dense<NT>(l == 0, Wl, X0, (l >= 1 && l - 1 < NHL) ? Hl + (l - 1) * MT * G::TSIZE : X, mt, nt, lane, ksh);
#pragma unroll
      for (int r = 0; r < 4; ++r) h[l][r] = act_apply_t<GEN>(h[l][r], a.act);
      __syncthreads();
      if (l < NHL) store_tile<NT>(Hl + l * MT * G::TSIZE, mt, nt, lane, h[l]);
      else if (l + 1 < NH) store_tile<NT>(X, mt, nt, lane, h[l]);
      commit_hidden<NT>(Wl, regs, breg);
      __syncthreads();
    }

    // ---- backward sweep ---------------------------------------------------------------------------------------
    v4d gpre = v4d{0, 0, 0, 0};
    if (nt == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + mt * 16 + g + 4 * r;
        gpre[r] = (c < D && row < M) ? a.g_out[mlp_out_index(a, c & 1, row, c >> 1)] : 0.0;
      }
    }
    int poff_end = a.psize;
#pragma unroll
    for (int l = NH; l >= 0; --l) {
      const int Hin = l == 0 ? D : H, Hout = l == NH ? D : H;
      poff_end -= Hout * Hin + Hout;
      double* pW = part + poff_end;
      double* pB = pW + Hout * Hin;
      if (l == 1) prefetch_first<NT>(a.w[0], a.b[0], H, D, regs, breg);
      else if (l > 1) prefetch_hidden<NT>(a.w[l - 1], a.b[l - 1], H, H, regs, breg);

      store_tile<NT>(Gt, mt, nt, lane, gpre);
      const bool in_lds = l > 0 && l - 1 < NHL;            // layer input: already in its own tiles / from registers
      double* Xl = in_lds ? Hl + (l > 0 ? l - 1 : 0) * MT * G::TSIZE : X;
      if (l > 0 && !in_lds) store_tile<NT>(X, mt, nt, lane, h[l > 0 ? l - 1 : 0]);
      {
        double v = (gpre[0] + gpre[1]) + (gpre[2] + gpre[3]);   // bias gradient: column sums over this tile's 16 rows
        v += shfl_xor(v, 16);
        v += shfl_xor(v, 32);
        if (g == 0) dbw[mt * HP + 16 * nt + c] = v;
      }
      __syncthreads();

      // (a) g_in tile (mt, nt) = g_pre W; the first layer has a single (16-column) input tile
      v4d gin = v4d{0, 0, 0, 0};
      if (l > 0 || nt == 0) {
        const double* ga = Gt + mt * G::TSIZE + c * S + g;
        const double* wb = Wl + g * S + 16 * nt + c;
        const int ks = l == NH ? 4 : ksh;                    // K = output neurons of this layer
        for (int s = 0; s < ks; ++s) gin = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[4 * s], wb[4 * s * S], gin, 0, 0, 0);
      }
      // (b) dW tiles over this pass's 32 rows, dealt round-robin to the waves, one accumulation chain per M-tile
      {
        const int nti = l == 0 ? 1 : NT, ntiles = (l == NH ? 1 : NT) * nti;
        for (int tile = wave; tile < ntiles; tile += NW) {
          const int t = tile / nti, u = tile - t * nti;
          const double* ga = Gt + g * S + 16 * t + c;
          const double* xb = l == 0 ? X0 + g * G::S0 + c : Xl + g * S + 16 * u + c;
          const int xts = l == 0 ? G::T0SIZE : G::TSIZE, xss = l == 0 ? G::S0 : S;
          v4d acc0 = v4d{0, 0, 0, 0}, acc1 = v4d{0, 0, 0, 0};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[4 * s * S], xb[4 * s * xss], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[G::TSIZE + 4 * s * S], xb[xts + 4 * s * xss], acc1, 0, 0, 0);
          }
          const int k = 16 * u + c;                          // D[i = o][j = k]
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * t + g + 4 * r;
            if (o < Hout && k < Hin) {
              const double v = acc0[r] + acc1[r];
              pW[o * Hin + k] = pass ? pW[o * Hin + k] + v : v;
            }
          }
        }
        if (tid < Hout) {
          const double v = dbw[tid] + dbw[HP + tid];
          pB[tid] = pass ? pB[tid] + v : v;
        }
      }
      if (l > 0) {
        const double* hd = Xl + mt * G::TSIZE + g * S + 16 * nt + c;      // this wave's tile of the layer input, D layout
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double hv = in_lds ? hd[4 * r * S] : h[l > 0 ? l - 1 : 0][r];
          gpre[r] = gin[r] * act_slope_t<GEN>(hv, a.act);
        }
        __syncthreads();                                     // every read of the weight image and of the tiles is done
        if (l == 1) commit_first<NT>(Wl, regs, breg);
        else commit_hidden<NT>(Wl, regs, breg);
      } else if (nt == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + mt * 16 + g + 4 * r;
          if (c < D && row < M) a.g_in[mlp_out_index(a, c & 1, row, c >> 1)] = gin[r];
        }
      }
    }
  }
}

template <int NT, int NH = 6>
static int launch(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  using G = Geo<NT>;
  const size_t smem = sizeof(double) * (backward ? (G::ONE_PASS ? G::bwd_doubles() : G::bwd2p_doubles()) : G::fwd_doubles());
  static_assert(sizeof(double) * G::bwd2p_doubles() <= 160 * 1024 && sizeof(double) * G::fwd_doubles() <= 160 * 1024, "LDS budget");
  // (LeakyReLU, the reference default, has its own instantiation: common.hpp act_apply_t)
  // (mlp_depth 3 .. 5 share the instantiation with the activation switch)
  constexpr bool DEF = NH == 6;
  void (*kern)(MlpArgs<double>) = (DEF && a.act == 0) ? mlp_fwd_wide_kernel<NT, NH, !DEF> : mlp_fwd_wide_kernel<NT, NH, true>;
  if (backward) {
    if constexpr (G::ONE_PASS) kern = (DEF && a.act == 0) ? mlp_bwd_wide_kernel<NT, NH, !DEF> : mlp_bwd_wide_kernel<NT, NH, true>;
    else kern = (DEF && a.act == 0) ? mlp_bwd_wide_2pass_kernel<NT, NH, !DEF> : mlp_bwd_wide_2pass_kernel<NT, NH, true>;
  }
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int nblk = backward ? cdiv(a.M, 64) : cdiv(a.M, 16 * MT);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(G::THREADS), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace wide

template <int NH>
static int wide_launch_depth(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  const int nt = (a.H + 15) / 16;
  if (nt == 4) return wide::launch<4, NH>(a, backward, stream);
  if (nt == 5) return wide::launch<5, NH>(a, backward, stream);
  return wide::launch<6, NH>(a, backward, stream);
}
// 48 < H <= 96, 2C <= 16, 4 .. 7 Linear layers (mlp_depth 3 .. 6).  Returns -2 if the shape is outside this kernel's range.
int mlp_mfma_wide_dispatch(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  if (a.nlin < 4 || a.nlin > 7 || a.H <= 48 || a.H > 96 || 2 * a.C > 16 || a.H < 2 * a.C) return -2;
  if (a.nlin == 4) return wide_launch_depth<3>(a, backward, stream);
  if (a.nlin == 5) return wide_launch_depth<4>(a, backward, stream);
  if (a.nlin == 6) return wide_launch_depth<5>(a, backward, stream);
  return wide_launch_depth<6>(a, backward, stream);
}

}  // namespace lgn
