// lgn-autoencoder_amd/csrc/mlp_mfma.hip -- CGMLP forward / backward on the fp64 matrix cores.
//
// Same operator as csrc/mlp.hip (reference: CGMLP.forward, lgn/models/lgn_levels.py:191-227), for hidden
// widths H <= 48 (all BASELINE maxdim=2 configs: H = 6 * 2C, C <= 4).  Each Linear is a
// [rows x Hin] x [Hin x Hout] GEMM executed with v_mfma_f64_16x16x4_f64 (D(16x16) += A(16x4) B(4x16)):
//   A: lane l holds A[i = l&15][k = l>>4]      B: lane l holds B[k = l>>4][j = l&15]
//   D: lane l, register r holds D[i = (l>>4) + 4r][j = l&15]          (probed: csrc/probes/mfma_probe.hip)
//
// Work split: a workgroup owns 64 rows = 4 M-tiles of 16 rows and runs 4 * NT waves (NT = ceil(H/16) N-tiles);
// wave (mt, nt) owns ONE 16x16 output tile of every layer.  B*N rows give only ~one M-tile per SIMD of the
// chip, so splitting N over waves is what puts several waves on a SIMD: a layer is 12 dependent MFMAs per wave,
// and the other waves' MFMAs fill the gaps left by its LDS reads, LeakyReLU and stores.  It also keeps the
// backward's register footprint small (one tile of each hidden activation instead of NT).
//
// LDS: activations as padded [row][neuron] tiles per M-tile (row stride == 2 mod 4 scalars -> conflict-free
// ds_read_b64 of A fragments), ping-ponged between layers so that one barrier per layer suffices; layer weights
// row-major [out][in] with the same stride (bias in column HP), double buffered, the next layer's weights
// prefetched into registers while the current layer's MFMAs run.
//
// Backward (recompute; the hidden activations of the wave's own tile stay in registers in D layout):
//   g_in  = g_pre W        A = g_pre tile,            B = W[o][k] read "transposed" from the same LDS image
//   dW    = g_pre^T h_in   A = g_pre tile^T, B = h_in tile, K = the workgroup's 64 rows; output tile (o-tile, k-tile)
//                          number w is computed by wave w and stored into this workgroup's partial row
//                          (reduced deterministically by reduce_partials afterwards)
//   db    = column sums of g_pre (two wave shuffles + a sum over the 4 M-tiles through LDS)
#include "ops.hpp"

namespace lgn {

typedef double v4d __attribute__((ext_vector_type(4)));
namespace { LGN_STAMP_DECL }
LGN_STAMP_READER(lgn_debug_stamps_mlp)

__host__ __device__ constexpr int pad4(int x) { return (x + 3) & ~3; }
// smallest stride >= x with stride == 2 (mod 4): 16 rows x 2 k's of a ds_read_b64 group hit 32 distinct bank pairs
__host__ __device__ constexpr int lds_stride(int x) { return x + ((6 - (x & 3)) & 3); }

// MT = M-tiles (16 rows each) per workgroup: 4 (64 rows) in general; 1 when the batch has too few rows to give every CU a
// 64-row workgroup (mlp_rows_per_workgroup): 4x the workgroups, each with a quarter of the MFMAs per layer -- at 64 jets the
// layer time is the matrix-pipe time of ONE CU's 64 rows while 200 CUs idle.
template <int NT, int MT = 4>
struct Geo {
  static constexpr int HP = NT * 16;                  // padded hidden width
  static constexpr int S = lds_stride(HP);            // row stride of weight image and activation tiles
  static constexpr int S0 = 18;                       // row stride of the 16-column MLP input tiles
  static constexpr int WSIZE = HP * S;                // one weight image
  static constexpr int TSIZE = 16 * S;                // one 16-row activation tile
  static constexpr int T0SIZE = 16 * S0;              // one 16-row input tile
  static constexpr int THREADS = 64 * MT * NT;        // MT M-tiles x NT N-tiles waves
  static constexpr int NPH = 4 * NT / MT;             // staging passes of a hidden-layer image (4 MT rows per pass)
  static constexpr int NPF = 4 / MT;                  // staging passes of the first-layer image (4 MT NT rows per pass)
  static_assert(MT == 1 || MT == 2 || MT == 4, "M-tiles per workgroup");
  static constexpr size_t fwd_doubles() { return 2 * WSIZE + 2 * MT * TSIZE + MT * T0SIZE; }
  static constexpr size_t bwd_doubles() { return 2 * WSIZE + 4 * MT * TSIZE + MT * T0SIZE + 2 * MT * HP; }
};

// ---- weight staging ------------------------------------------------------------------------------
// Image of one Linear layer in LDS: W[o][k] at Wl[o*S + k] (zero padded to 16-multiples both ways; the k padding
// feeds the MFMAs zeros, padded neurons produce exact zeros) and its bias at Wl[o*S + HP] (S >= HP + 2).
// Hidden / output layers (HP input columns): thread tid stages column k = tid % HP of rows tid / HP + 16 i, so every
// address is a thread-constant base plus a compile-time (LDS) or wave-uniform (global) multiple of i.
template <int NT, int MT>
__device__ __forceinline__ void prefetch_hidden(const double* __restrict__ W, const double* __restrict__ bias, int Hout,
                                                int Hin, double (&regs)[Geo<NT, MT>::NPH], double& breg) {
  constexpr int HP = Geo<NT, MT>::HP, RPP = 4 * MT;
  const int o0 = (int)threadIdx.x / HP, k = (int)threadIdx.x - o0 * HP;
  const double* src = W + (o0 * Hin + k);
#pragma unroll
  for (int i = 0; i < Geo<NT, MT>::NPH; ++i) regs[i] = (k < Hin && o0 + RPP * i < Hout) ? src[RPP * i * Hin] : 0.0;
  breg = ((int)threadIdx.x < Hout) ? bias[threadIdx.x] : 0.0;
}
template <int NT, int MT>
__device__ __forceinline__ void commit_hidden(double* Wl, const double (&regs)[Geo<NT, MT>::NPH], double breg) {
  constexpr int HP = Geo<NT, MT>::HP, S = Geo<NT, MT>::S, RPP = 4 * MT;
  const int o0 = (int)threadIdx.x / HP, k = (int)threadIdx.x - o0 * HP;
  double* dst = Wl + (o0 * S + k);
#pragma unroll
  for (int i = 0; i < Geo<NT, MT>::NPH; ++i) dst[RPP * i * S] = regs[i];
  if ((int)threadIdx.x < HP) Wl[threadIdx.x * S + HP] = breg;
}
// First layer (2C <= 16 input columns): thread -> column tid % 16 of rows tid / 16 + 4 MT NT i.
template <int NT, int MT>
__device__ __forceinline__ void prefetch_first(const double* __restrict__ W, const double* __restrict__ bias, int Hout,
                                               int Hin, double (&regs)[Geo<NT, MT>::NPH], double& breg) {
  const int o0 = (int)threadIdx.x >> 4, k = (int)threadIdx.x & 15;
#pragma unroll
  for (int i = 0; i < Geo<NT, MT>::NPF; ++i) {
    const int o = o0 + 4 * MT * NT * i;
    regs[i] = (o < Hout && k < Hin) ? W[o * Hin + k] : 0.0;
  }
  breg = ((int)threadIdx.x < Hout) ? bias[threadIdx.x] : 0.0;
}
template <int NT, int MT>
__device__ __forceinline__ void commit_first(double* Wl, const double (&regs)[Geo<NT, MT>::NPH], double breg) {
  constexpr int HP = Geo<NT, MT>::HP, S = Geo<NT, MT>::S;
#pragma unroll
  for (int i = 0; i < Geo<NT, MT>::NPF; ++i) Wl[(((int)threadIdx.x >> 4) + 4 * MT * NT * i) * S + ((int)threadIdx.x & 15)] = regs[i];
  if ((int)threadIdx.x < HP) Wl[threadIdx.x * S + HP] = breg;
}

// rows of the scalar irrep [2][M][C] -> the workgroup's 4 input tiles, feature k = 2c + z, zero padded to 16 columns
template <int NT, int MT>
__device__ __forceinline__ void load_input_tiles(const double* __restrict__ s, int M, int C, int wg_row0, double* X0) {
  const int D = 2 * C;
  for (int e = threadIdx.x; e < 16 * MT * 16; e += Geo<NT, MT>::THREADS) {
    const int r = e >> 4, k = e & 15, row = wg_row0 + r;
    X0[(r >> 4) * Geo<NT, MT>::T0SIZE + (r & 15) * Geo<NT, MT>::S0 + k] =
        (row < M && k < D) ? s[(size_t)(k & 1) * M * C + (size_t)row * C + (k >> 1)] : 0.0;
  }
}

// acc (D layout) += X[16 x 4KS] W^T for one output tile.  xa -> A fragment base (row c, column g), wb -> B fragment base
// (W row 16 nt + c, column g); KS compile-time k-steps, or the run-time count ks when KS == 0.
template <int KS>
__device__ __forceinline__ void mma_rowmajor(const double* xa, const double* wb, int ks, v4d& acc) {
  if (KS > 0) {
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * s], wb[4 * s], acc, 0, 0, 0);
  } else {
    for (int s = 0; s < ks; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * s], wb[4 * s], acc, 0, 0, 0);
  }
}
// acc += G[16 x 4KS] W for one input tile: ga -> A fragment base of the g_pre tile, wb -> W image at (row g, column 16u + c)
template <int KS, int S>
__device__ __forceinline__ void mma_transposed(const double* ga, const double* wb, int ks, v4d& acc) {
  if (KS > 0) {
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[4 * s], wb[4 * s * S], acc, 0, 0, 0);
  } else {
    for (int s = 0; s < ks; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[4 * s], wb[4 * s * S], acc, 0, 0, 0);
  }
}

// One forward layer for this wave's tile (shared by the forward kernel and the backward's recompute).
//   l == 0 reads the input tiles (K = 16), hidden layers the activation buffer of parity l&1; writes LeakyReLU(pre)
//   into the buffer of parity (l+1)&1 when `store`.
template <int NT, int KSH, int MT, bool GEN>
__device__ __forceinline__ v4d forward_layer(int l, const double* Wcur, const double* X0, double* Xb, int mt, int nt, int lane,
                                             int ksh, bool store, int act) {
  using G = Geo<NT, MT>;
  constexpr int S = G::S;
  const int c = lane & 15, g = lane >> 4;
  const double bias = Wcur[(16 * nt + c) * S + G::HP];
  v4d acc = v4d{bias, bias, bias, bias};
  const double* wb = Wcur + (16 * nt + c) * S + g;
  if (l == 0) mma_rowmajor<4>(X0 + mt * G::T0SIZE + c * G::S0 + g, wb, 4, acc);
  else mma_rowmajor<KSH>(Xb + ((l & 1) * MT + mt) * G::TSIZE + c * S + g, wb, ksh, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = act_apply_t<GEN>(acc[r], act);
  if (store) {
    double* Xn = Xb + (((l + 1) & 1) * MT + mt) * G::TSIZE + g * S + 16 * nt + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) Xn[4 * r * S] = acc[r];
  }
  return acc;
}

template <int NT, int NH, int KSH, int MT, bool GEN>
__global__ __launch_bounds__(64 * MT * NT) void mlp_fwd_mfma_kernel(MlpArgs<double> a) {
  using G = Geo<NT, MT>;
  constexpr int S = G::S;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave % MT, nt = wave / MT;
  const int D = 2 * a.C, H = a.H, M = a.M;
  const int ksh = pad4(H) >> 2;
  const int wg_row0 = blockIdx.x * 16 * MT;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  double* Xb = Wl + 2 * G::WSIZE;                            // 2 x MT activation tiles
  double* X0 = Xb + 2 * MT * G::TSIZE;                       // MT input tiles

  double regs[G::NPH], breg;
  STAMP(0);
  prefetch_first<NT, MT>(a.w[0], a.b[0], H, D, regs, breg);
  load_input_tiles<NT, MT>(a.s_in, M, a.C, wg_row0, X0);
  commit_first<NT, MT>(Wl, regs, breg);
  __syncthreads();
#pragma unroll
  for (int l = 0; l < NH; ++l) {
    const double* Wcur = Wl + (l & 1) * G::WSIZE;
    prefetch_hidden<NT, MT>(a.w[l + 1], a.b[l + 1], l + 1 == NH ? D : H, H, regs, breg);
    const v4d hv = forward_layer<NT, KSH, MT, GEN>(l, Wcur, X0, Xb, mt, nt, lane, ksh, true, a.act);
    if (a.h_saved) {                                           // kept for the backward (rows beyond M: finite values of zero inputs)
      double* hs = a.h_saved + ((size_t)l * a.h_rows + wg_row0 + mt * 16 + (lane >> 4)) * G::HP + 16 * nt + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) hs[(size_t)4 * r * G::HP] = hv[r];
    }
    commit_hidden<NT, MT>(Wl + ((l + 1) & 1) * G::WSIZE, regs, breg);
    __syncthreads();
  }
  if (nt == 0) {                                             // output layer: one N-tile (2C <= 16 neurons), no activation
    const int c = lane & 15, g = lane >> 4;
    const double* Wcur = Wl + (NH & 1) * G::WSIZE;
    const double bias = Wcur[c * S + G::HP];
    v4d acc = v4d{bias, bias, bias, bias};
    mma_rowmajor<KSH>(Xb + ((NH & 1) * MT + mt) * G::TSIZE + c * S + g, Wcur + c * S + g, ksh, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wg_row0 + mt * 16 + g + 4 * r;
      if (c < D && row < M) a.s_out[mlp_out_index(a, c & 1, row, c >> 1)] = acc[r];
    }
  }
  STAMP(1);
}

template <int NT, int NH, int KSH, int MT, bool GEN>
__global__ __launch_bounds__(64 * MT * NT) void mlp_bwd_mfma_kernel(MlpArgs<double> a) {
  using G = Geo<NT, MT>;
  constexpr int S = G::S, HP = G::HP, NW = MT * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave % MT, nt = wave / MT;
  const int c = lane & 15, g = lane >> 4;
  const int D = 2 * a.C, H = a.H, M = a.M;
  const int ksh = pad4(H) >> 2;
  const int wg_row0 = blockIdx.x * 16 * MT;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  double* Xb = Wl + 2 * G::WSIZE;                            // 2 x MT layer-input tiles
  double* Gb = Xb + 2 * MT * G::TSIZE;                       // 2 x MT g_pre tiles
  double* X0 = Gb + 2 * MT * G::TSIZE;                       // MT MLP input tiles
  double* dbw = X0 + MT * G::T0SIZE;                         // 2 x MT x HP column sums
  double* part = a.part + (size_t)blockIdx.x * a.psize;

  // ---- h[l] = post-activation of hidden layer l, this wave's tile, D layout: read back from the forward's copy, or recomputed
  STAMP(2);
  double regs[G::NPH], breg;
  v4d h[NH];
  if (a.h_saved) {
    prefetch_hidden<NT, MT>(a.w[NH], a.b[NH], D, H, regs, breg);        // the output layer: first layer of the backward sweep
    load_input_tiles<NT, MT>(a.s_in, M, a.C, wg_row0, X0);
    const double* hs = a.h_saved + ((size_t)wg_row0 + mt * 16 + g) * HP + 16 * nt + c;
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[l][r] = hs[((size_t)l * a.h_rows + 4 * r) * HP];
    commit_hidden<NT, MT>(Wl + (NH & 1) * G::WSIZE, regs, breg);
    __syncthreads();
  } else {
    prefetch_first<NT, MT>(a.w[0], a.b[0], H, D, regs, breg);
    load_input_tiles<NT, MT>(a.s_in, M, a.C, wg_row0, X0);
    commit_first<NT, MT>(Wl, regs, breg);
    __syncthreads();
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      const double* Wcur = Wl + (l & 1) * G::WSIZE;
      // the weights of the next forward layer; after the last hidden layer: the output layer (first backward layer)
      prefetch_hidden<NT, MT>(a.w[l + 1], a.b[l + 1], l + 1 == NH ? D : H, H, regs, breg);
      h[l] = forward_layer<NT, KSH, MT, GEN>(l, Wcur, X0, Xb, mt, nt, lane, ksh, l + 1 < NH, a.act);
      commit_hidden<NT, MT>(Wl + ((l + 1) & 1) * G::WSIZE, regs, breg);
      __syncthreads();
    }
  }

  // ---- backward sweep; tile buffers of parity q alternate per layer -> one barrier per layer ------------
  v4d gpre = v4d{0, 0, 0, 0};
  if (nt == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wg_row0 + mt * 16 + g + 4 * r;
      gpre[r] = (c < D && row < M) ? a.g_out[mlp_out_index(a, c & 1, row, c >> 1)] : 0.0;
    }
  }
  int poff_end = a.psize;
#pragma unroll
  for (int l = NH; l >= 0; --l) {
    const int Hin = l == 0 ? D : H, Hout = l == NH ? D : H;
    const int q = (NH - l) & 1;
    poff_end -= Hout * Hin + Hout;
    double* pW = part + poff_end;
    double* pB = pW + Hout * Hin;
    const double* Wcur = Wl + (l & 1) * G::WSIZE;            // image of W_l (staged by the previous iteration)
    if (l == 1) prefetch_first<NT, MT>(a.w[0], a.b[0], H, D, regs, breg);
    else if (l > 1) prefetch_hidden<NT, MT>(a.w[l - 1], a.b[l - 1], H, H, regs, breg);

    // operands of this layer to LDS: g_pre tile and layer-input tile (h[l-1]; the MLP input tiles serve l == 0)
    double* Gq = Gb + q * MT * G::TSIZE;
    double* Xq = Xb + q * MT * G::TSIZE;
    {
      double* gt = Gq + mt * G::TSIZE + g * S + 16 * nt + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) gt[4 * r * S] = gpre[r];
      if (l > 0) {
        double* xt = Xq + mt * G::TSIZE + g * S + 16 * nt + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) xt[4 * r * S] = h[l > 0 ? l - 1 : 0][r];
      }
      // bias gradient: column sums over this tile's 16 rows
      double v = (gpre[0] + gpre[1]) + (gpre[2] + gpre[3]);
      v += shfl_xor(v, 16);
      v += shfl_xor(v, 32);
      if (g == 0) dbw[(q * MT + mt) * HP + 16 * nt + c] = v;
    }
    __syncthreads();

    // (a) g_in tile (mt, nt) = g_pre W; the first layer has a single (16-column) input tile
    v4d gin = v4d{0, 0, 0, 0};
    if (l > 0 || nt == 0) {
      const double* ga = Gq + mt * G::TSIZE + c * S + g;
      const double* wb = Wcur + g * S + 16 * nt + c;
      if (l == NH) mma_transposed<4, S>(ga, wb, 4, gin);     // K = 2C <= 16 output neurons
      else mma_transposed<KSH, S>(ga, wb, ksh, gin);
    }
    // (b) dW tiles over the workgroup's 16 MT rows, dealt round robin to the waves (MT = 4: 9 tiles on 12 waves, one each);
    //     two accumulation chains over the row blocks
    {
      const int nti = l == 0 ? 1 : NT, ntiles = (l == NH ? 1 : NT) * nti;
      for (int tile = wave; tile < ntiles; tile += NW) {
        const int t = tile / nti, u = tile - t * nti;
        const double* ga = Gq + g * S + 16 * t + c;
        const double* xb = l == 0 ? X0 + g * G::S0 + c : Xq + g * S + 16 * u + c;
        const int xts = l == 0 ? G::T0SIZE : G::TSIZE, xss = l == 0 ? G::S0 : S;
        v4d acc0 = v4d{0, 0, 0, 0}, acc1 = v4d{0, 0, 0, 0};
        if (MT >= 2) {
#pragma unroll
          for (int w = 0; w < MT / 2; ++w)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[w * G::TSIZE + 4 * s * S], xb[w * xts + 4 * s * xss], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[(w + MT / 2) * G::TSIZE + 4 * s * S], xb[(w + MT / 2) * xts + 4 * s * xss],
                                                          acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[4 * s * S], xb[4 * s * xss], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[4 * (s + 2) * S], xb[4 * (s + 2) * xss], acc1, 0, 0, 0);
          }
        }
        const int k = 16 * u + c;                            // D[i = o][j = k]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * t + g + 4 * r;
          if (o < Hout && k < Hin) __builtin_nontemporal_store(acc0[r] + acc1[r], &pW[o * Hin + k]);   // (read once, by the reduction at the end of the step)
        }
      }
      const double* dq = dbw + q * MT * HP;
      if (tid < Hout) {
        double v = dq[tid];
#pragma unroll
        for (int w = 1; w < MT; ++w) v += dq[w * HP + tid];
        __builtin_nontemporal_store(v, &pB[tid]);
      }
    }
    // next layer down
    if (l > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gpre[r] = gin[r] * act_slope_t<GEN>(h[l > 0 ? l - 1 : 0][r], a.act);
      // W_{l-1} goes into the image buffer last read two layers up; every wave is past that layer's barrier
      if (l == 1) commit_first<NT, MT>(Wl, regs, breg);
      else commit_hidden<NT, MT>(Wl + ((l - 1) & 1) * G::WSIZE, regs, breg);
    } else if (nt == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wg_row0 + mt * 16 + g + 4 * r;
        if (c < D && row < M) a.g_in[mlp_out_index(a, c & 1, row, c >> 1)] = gin[r];
      }
    }
  }
  STAMP(3);
}

template <int NT, int KSH, int MT, int NH>
static int launch_mlp_mfma_mt(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  using G = Geo<NT, MT>;
  const int nblk = cdiv(a.M, 16 * MT);
  const size_t smem = sizeof(double) * (backward ? G::bwd_doubles() : G::fwd_doubles());
  static_assert(sizeof(double) * G::bwd_doubles() <= 160 * 1024, "LDS budget");
  // (LeakyReLU, the reference default, has its own instantiation: common.hpp act_apply_t)
  // (the other depths -- mlp_depth 3 .. 5, round 5 -- share the instantiation with the activation switch)
  auto kern = backward ? mlp_bwd_mfma_kernel<NT, NH, KSH, MT, true> : mlp_fwd_mfma_kernel<NT, NH, KSH, MT, true>;
  if constexpr (NH == 6) {
    if (a.act == 0) kern = backward ? mlp_bwd_mfma_kernel<NT, NH, KSH, MT, false> : mlp_fwd_mfma_kernel<NT, NH, KSH, MT, false>;
  }
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(G::THREADS), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}
template <int NT, int KSH, int NH = 6>
static int launch_mlp_mfma(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  // (measured again in round 4 with today's staging: 32-row forward workgroups, two per CU with independent barrier domains --
  // MT = 2, 68 KB of LDS -- take the cfg2 step from 0.568 to 0.584 ms)
  return mlp_rows_per_workgroup(a.M, a.H) == 16 ? launch_mlp_mfma_mt<NT, KSH, 1, NH>(a, backward, stream)
                                                 : launch_mlp_mfma_mt<NT, KSH, 4, NH>(a, backward, stream);
}

// the depths besides the reference default (mlp_depth 3 .. 5 = 4 .. 6 Linear layers): run-time k-loops
template <int NH>
static int launch_mlp_mfma_depth(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  const int nt = (a.H + 15) / 16;
  if (nt == 1) return launch_mlp_mfma<1, 0, NH>(a, backward, stream);
  if (nt == 2) return launch_mlp_mfma<2, 0, NH>(a, backward, stream);
  return launch_mlp_mfma<3, 0, NH>(a, backward, stream);
}

// H <= 48, 2C <= 16, 4 .. 7 Linear layers (mlp_depth 3 .. 6).  Returns -2 if the shape is outside this kernel's range.
int mlp_mfma_dispatch(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  if (a.nlin < 4 || a.nlin > 7 || a.H > 48 || 2 * a.C > 16 || a.H < 2 * a.C) return -2;
  LGN_CHECK_ARG(!a.h_saved || a.h_rows >= mlp_saved_rows(a.M), "CGMLP: the saved-activation buffer has %d rows per layer, %d rows need %d",
                a.h_rows, a.M, mlp_saved_rows(a.M));
  if (a.nlin == 4) return launch_mlp_mfma_depth<3>(a, backward, stream);
  if (a.nlin == 5) return launch_mlp_mfma_depth<4>(a, backward, stream);
  if (a.nlin == 6) return launch_mlp_mfma_depth<5>(a, backward, stream);
  const int nt = (a.H + 15) / 16;
  // fully unrolled k-loops for the widths of the reference configs (H = 6 * 2C); anything else keeps the run-time loop
  if (a.H == 48) return launch_mlp_mfma<3, 12>(a, backward, stream);
  if (a.H == 36) return launch_mlp_mfma<3, 9>(a, backward, stream);
  if (a.H == 24) return launch_mlp_mfma<2, 6>(a, backward, stream);
  if (a.H == 12) return launch_mlp_mfma<1, 3>(a, backward, stream);
  if (nt == 1) return launch_mlp_mfma<1, 0>(a, backward, stream);
  if (nt == 2) return launch_mlp_mfma<2, 0>(a, backward, stream);
  return launch_mlp_mfma<3, 0>(a, backward, stream);
}

}  // namespace lgn
