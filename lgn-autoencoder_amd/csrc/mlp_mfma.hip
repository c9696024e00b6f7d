// lgn-autoencoder_amd/csrc/mlp_mfma.hip -- CGMLP forward / backward on the fp64 matrix cores.
//
// Same operator as csrc/mlp.hip (reference: CGMLP.forward, lgn/models/lgn_levels.py:191-227), for hidden
// widths H <= 48 (all BASELINE maxdim=2 configs: H = 6 * 2C, C <= 4).  Each Linear is a
// [rows x Hin] x [Hin x Hout] GEMM executed with v_mfma_f64_16x16x4_f64 (D(16x16) += A(16x4) B(4x16)):
//   A: lane l holds A[i = l&15][k = l>>4]      B: lane l holds B[k = l>>4][j = l&15]
//   D: lane l, register r holds D[i = (l>>4) + 4r][j = l&15]          (probed: csrc/probes/mfma_probe.hip)
// A workgroup owns 64 rows; each of its 4 waves owns a 16-row M-tile and all N-tiles of the layer.
// Activations stay wave-private: D fragments are written to a padded [row][neuron] LDS tile and read back
// as A fragments of the next layer (row stride == 2 mod 4 scalars -> conflict-free ds_read_b64).
// Layer weights are staged row-major [out][in] in LDS (same padded stride), double buffered, the next
// layer's weights prefetched into registers while the current layer's MFMAs run.
//
// Backward (recompute, hidden activations kept in registers in D layout):
//   g_in  = g_pre W        A = g_pre tile,            B = W[o][k] read "transposed" from the same LDS image
//   dW    = g_pre^T h_in   A = g_pre tile^T, B = h_in tile, K = the workgroup's 64 rows; the (o,k) output tiles
//                          are dealt round-robin to the 4 waves, which store them into this workgroup's
//                          partial row (reduced deterministically by reduce_partials afterwards)
//   db    = column sums of g_pre (two wave shuffles + a 4-wave LDS reduction)
#include "ops.hpp"

namespace lgn {

typedef double v4d __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int pad4(int x) { return (x + 3) & ~3; }
__host__ __device__ constexpr int pad16(int x) { return (x + 15) & ~15; }
// smallest stride >= x with stride == 2 (mod 4): 16 rows x 2 k's of a ds_read_b64 group hit 32 distinct bank pairs
__host__ __device__ constexpr int lds_stride(int x) { return x + ((6 - (x & 3)) & 3); }

template <int NT>
struct Geo {
  static constexpr int HP = NT * 16;                  // padded hidden width
  static constexpr int S = lds_stride(HP);            // row stride of weight image and activation tiles
  static constexpr int NPF = (HP * HP + BLOCK - 1) / BLOCK;   // prefetch registers per thread
  static constexpr int WSIZE = HP * S;                // one weight image
  static constexpr int TSIZE = 16 * S;                // one 16-row activation tile
};

// ---- weight staging ------------------------------------------------------------------------------
// Image of one Linear layer in LDS: W[o][k] at Wl[o*S + k] (zero padded to 16-multiples both ways, k padding feeds
// the MFMAs zeros) and its bias at Wl[o*S + HP] (S >= HP + 2).  KP = padded input width, a compile-time constant
// so that the element -> (o,k) split costs a multiply-shift, not an integer division.
template <int NT, int KP>
__device__ __forceinline__ void prefetch_weights(const double* __restrict__ W, const double* __restrict__ bias, int Hout,
                                                 int Hin, double (&regs)[Geo<NT>::NPF], double& breg) {
  const int HoP = pad16(Hout);
#pragma unroll
  for (int i = 0; i < Geo<NT>::NPF; ++i) {
    const int e = threadIdx.x + BLOCK * i;
    const int o = e / KP, k = e - o * KP;
    regs[i] = (e < HoP * KP && o < Hout && k < Hin) ? W[(size_t)o * Hin + k] : 0.0;
  }
  breg = ((int)threadIdx.x < Hout) ? bias[threadIdx.x] : 0.0;
}
template <int NT, int KP>
__device__ __forceinline__ void commit_weights(double* Wl, int Hout, const double (&regs)[Geo<NT>::NPF], double breg) {
  const int HoP = pad16(Hout);
#pragma unroll
  for (int i = 0; i < Geo<NT>::NPF; ++i) {
    const int e = threadIdx.x + BLOCK * i;
    const int o = e / KP, k = e - o * KP;
    if (e < HoP * KP) Wl[o * Geo<NT>::S + k] = regs[i];
  }
  if ((int)threadIdx.x < Geo<NT>::HP) Wl[threadIdx.x * Geo<NT>::S + Geo<NT>::HP] = breg;
}
// layer-kind dispatch: the first Linear has a 16-wide (padded 2C) input, all others a HP-wide one
template <int NT>
__device__ __forceinline__ void prefetch_layer(const MlpArgs<double>& a, int l, int NH, int D, int H, double (&regs)[Geo<NT>::NPF],
                                               double& breg) {
  if (l == 0) prefetch_weights<NT, 16>(a.w[0], a.b[0], H, D, regs, breg);
  else prefetch_weights<NT, Geo<NT>::HP>(a.w[l], a.b[l], l == NH ? D : H, H, regs, breg);
}
template <int NT>
__device__ __forceinline__ void commit_layer(double* Wl, int l, int NH, int D, int H, const double (&regs)[Geo<NT>::NPF], double breg) {
  if (l == 0) commit_weights<NT, 16>(Wl, H, regs, breg);
  else commit_weights<NT, Geo<NT>::HP>(Wl, l == NH ? D : H, regs, breg);
}

// ---- one dense layer on a wave's 16-row tile:  acc[t] (D layout) = bias + X W^T ---------------------
template <int NT>
__device__ __forceinline__ void load_bias(const double* Wl, int lane, double (&bv)[NT]) {
  const int c = lane & 15;
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = Wl[(16 * t + c) * Geo<NT>::S + Geo<NT>::HP];     // zero for padded neurons
}

// NU = number of live N-tiles, KS = number of k-steps (both compile time: a run-time guard around the MFMA makes hipcc
// shuttle the accumulators between register classes, and a fully unrolled k-loop lets it schedule the LDS operand
// reads of later steps under the MFMAs of earlier ones without loop-carried copies).  KS == 0: run-time k-loop.
template <int NT, int NU, int KS>
__device__ __forceinline__ void dense_tile(const double* Xt, const double* Wl, const double (&bv)[NT], int Hin,
                                           int lane, v4d (&acc)[NT]) {
  constexpr int S = Geo<NT>::S;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = v4d{bv[t], bv[t], bv[t], bv[t]};
  const double* xa = Xt + c * S + g;
  const double* wb = Wl + c * S + g;
  if (KS > 0) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double a = xa[4 * s];
#pragma unroll
      for (int t = 0; t < NU; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wb[16 * t * S + 4 * s], acc[t], 0, 0, 0);
    }
  } else {
    const int ks = pad4(Hin) >> 2;
    for (int s = 0; s < ks; ++s) {
      const double a = xa[4 * s];
#pragma unroll
      for (int t = 0; t < NU; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wb[16 * t * S + 4 * s], acc[t], 0, 0, 0);
    }
  }
}
// layer-kind dispatch of the k-step count: first layer K = 2C <= 16 (run time), hidden layers K = H = 4*KSH
template <int NT, int NU, int KSH>
__device__ __forceinline__ void dense_hidden(const double* Xt, const double* Wl, const double (&bv)[NT], int Hin, int lane,
                                             v4d (&acc)[NT], bool first) {
  if (first) dense_tile<NT, NU, 0>(Xt, Wl, bv, Hin, lane, acc);
  else dense_tile<NT, NU, KSH>(Xt, Wl, bv, Hin, lane, acc);
}

// g_in (D layout) += g_pre W for this wave's rows; NU live input tiles
template <int NT, int NU>
__device__ __forceinline__ void gin_tile(const double* Gt, const double* Wcur, int Hout, int lane, v4d (&gin)[NT]) {
  constexpr int S = Geo<NT>::S;
  const int c = lane & 15, g = lane >> 4;
  const int ks = pad4(Hout) >> 2;
  for (int s = 0; s < ks; ++s) {
    const double av = Gt[c * S + 4 * s + g];
#pragma unroll
    for (int u = 0; u < NU; ++u)
      gin[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Wcur[(4 * s + g) * S + 16 * u + c], gin[u], 0, 0, 0);
  }
}

// D-layout registers -> [row][neuron] tile (all NT tiles; padded neurons carry exact zeros)
template <int NT>
__device__ __forceinline__ void store_tile(double* Xt, const v4d (&v)[NT], int lane) {
  constexpr int S = Geo<NT>::S;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) Xt[(g + 4 * r) * S + 16 * t + c] = v[t][r];
}

// rows of the scalar irrep [2][M][C] -> wave tile, feature k = 2c + z, zero padded to 16 columns
template <int NT>
__device__ __forceinline__ void load_input_tile(const double* __restrict__ s, int M, int C, int row0, double* Xt, int lane) {
  constexpr int S = Geo<NT>::S;
  const int D = 2 * C;
  for (int e = lane; e < 16 * 16; e += 64) {
    const int r = e >> 4, k = e & 15, row = row0 + r;
    Xt[r * S + k] = (row < M && k < D) ? s[(size_t)(k & 1) * M * C + (size_t)row * C + (k >> 1)] : 0.0;
  }
}

template <int NT, int NH, int KSH>
__global__ __launch_bounds__(BLOCK) void mlp_fwd_mfma_kernel(MlpArgs<double> a) {
  using G = Geo<NT>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = 2 * a.C, H = a.H, M = a.M;
  const int row0 = blockIdx.x * 64 + wave * 16;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  double* Xt = Wl + 2 * G::WSIZE + wave * G::TSIZE;          // this wave's activation tile

  double regs[G::NPF], breg;
  prefetch_layer<NT>(a, 0, NH, D, H, regs, breg);
  load_input_tile<NT>(a.s_in, M, a.C, row0, Xt, lane);
  commit_layer<NT>(Wl, 0, NH, D, H, regs, breg);
  __syncthreads();
#pragma unroll
  for (int l = 0; l <= NH; ++l) {
    const int Hin = l == 0 ? D : H;
    double* Wcur = Wl + (l & 1) * G::WSIZE;
    double bv[NT];
    load_bias<NT>(Wcur, lane, bv);
    if (l < NH) prefetch_layer<NT>(a, l + 1, NH, D, H, regs, breg);
    v4d acc[NT];
    if (l < NH) dense_hidden<NT, NT, KSH>(Xt, Wcur, bv, Hin, lane, acc, l == 0);
    else dense_tile<NT, 1, KSH>(Xt, Wcur, bv, Hin, lane, acc);
    if (l < NH) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = leaky(acc[t][r]);
      store_tile<NT>(Xt, acc, lane);
      commit_layer<NT>(Wl + ((l + 1) & 1) * G::WSIZE, l + 1, NH, D, H, regs, breg);
      __syncthreads();
    } else {
      const int c = lane & 15, g = lane >> 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + g + 4 * r;
        if (c < D && row < M) a.s_out[(size_t)(c & 1) * M * a.C + (size_t)row * a.C + (c >> 1)] = acc[0][r];
      }
    }
  }
}

template <int NT, int NH, int KSH>
__global__ __launch_bounds__(BLOCK) void mlp_bwd_mfma_kernel(MlpArgs<double> a) {
  using G = Geo<NT>;
  constexpr int S = G::S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int D = 2 * a.C, H = a.H, M = a.M;
  const int row0 = blockIdx.x * 64 + wave * 16;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  double* Xall = Wl + 2 * G::WSIZE;                          // 4 waves x layer-input tile
  double* Gall = Xall + 4 * G::TSIZE;                        // 4 waves x g_pre tile
  double* X0all = Gall + 4 * G::TSIZE;                       // 4 waves x MLP input tile (16 columns)
  double* dbw = X0all + 4 * 16 * S;                          // 4 waves x HP column sums
  double* Xt = Xall + wave * G::TSIZE;
  double* Gt = Gall + wave * G::TSIZE;
  double* X0t = X0all + wave * 16 * S;
  double* part = a.part + (size_t)blockIdx.x * a.psize;

  // ---- forward recompute; h[l] = post-activation of hidden layer l in D layout -----------------------
  double regs[G::NPF], breg;
  prefetch_layer<NT>(a, 0, NH, D, H, regs, breg);
  load_input_tile<NT>(a.s_in, M, a.C, row0, X0t, lane);
  for (int e = lane; e < 16 * 16; e += 64) Xt[(e >> 4) * S + (e & 15)] = X0t[(e >> 4) * S + (e & 15)];
  commit_layer<NT>(Wl, 0, NH, D, H, regs, breg);
  __syncthreads();
  v4d h[NH][NT];
#pragma unroll
  for (int l = 0; l < NH; ++l) {
    const int Hin = l == 0 ? D : H;
    double* Wcur = Wl + (l & 1) * G::WSIZE;
    // the weights of the next forward layer; after the last hidden layer: the output layer (first backward layer)
    double bv[NT];
    load_bias<NT>(Wcur, lane, bv);
    prefetch_layer<NT>(a, l + 1, NH, D, H, regs, breg);
    dense_hidden<NT, NT, KSH>(Xt, Wcur, bv, Hin, lane, h[l], l == 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[l][t][r] = leaky(h[l][t][r]);
    if (l + 1 < NH) store_tile<NT>(Xt, h[l], lane);
    commit_layer<NT>(Wl + ((l + 1) & 1) * G::WSIZE, l + 1, NH, D, H, regs, breg);
    __syncthreads();
  }

  // ---- backward sweep ----------------------------------------------------------------------------
  v4d gpre[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) gpre[t] = v4d{0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + g + 4 * r;
    gpre[0][r] = (c < D && row < M) ? a.g_out[(size_t)(c & 1) * M * a.C + (size_t)row * a.C + (c >> 1)] : 0.0;
  }
  size_t poff_end = a.psize;
#pragma unroll
  for (int l = NH; l >= 0; --l) {
    const int Hin = l == 0 ? D : H, Hout = l == NH ? D : H;
    const int nto = (Hout + 15) >> 4, nti = (Hin + 15) >> 4;
    poff_end -= (size_t)Hout * Hin + Hout;
    double* pW = part + poff_end;
    double* pB = pW + (size_t)Hout * Hin;
    double* Wcur = Wl + (l & 1) * G::WSIZE;                   // image of W_l (staged by the previous iteration)
    if (l > 0) prefetch_layer<NT>(a, l - 1, NH, D, H, regs, breg);

    // operands of this layer to LDS: g_pre tile and layer-input tile (h[l-1]; the MLP input for l == 0)
    store_tile<NT>(Gt, gpre, lane);
    if (l > 0) store_tile<NT>(Xt, h[l > 0 ? l - 1 : 0], lane);
    // bias gradient: column sums over this wave's 16 rows
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double v = (gpre[t][0] + gpre[t][1]) + (gpre[t][2] + gpre[t][3]);
      v += shfl_xor(v, 16);
      v += shfl_xor(v, 32);
      if (g == 0) dbw[wave * G::HP + 16 * t + c] = v;
    }
    __syncthreads();

    // (a) g_in = g_pre W  (own rows), kept in registers for the next (lower) layer
    v4d gin[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) gin[u] = v4d{0, 0, 0, 0};
    if (l == 0) gin_tile<NT, 1>(Gt, Wcur, Hout, lane, gin);      // live input tiles known at compile time
    else gin_tile<NT, NT>(Gt, Wcur, Hout, lane, gin);
    // (b) dW tiles over the workgroup's 64 rows, dealt round-robin to the waves
    {
      const double* Xsrc = l > 0 ? Xall : X0all;
      const int xts = l > 0 ? G::TSIZE : 16 * S;
      for (int tile = wave; tile < nto * nti; tile += 4) {
        const int t = tile / nti, u = tile - t * nti;
        v4d acc = v4d{0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double av = Gall[w * G::TSIZE + (4 * s + g) * S + 16 * t + c];
            const double bv = Xsrc[w * xts + (4 * s + g) * S + 16 * u + c];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
          }
        // D[i = o][j = k]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * t + g + 4 * r, k = 16 * u + c;
          if (o < Hout && k < Hin) pW[(size_t)o * Hin + k] = acc[r];
        }
      }
      for (int o = tid; o < Hout; o += BLOCK)
        pB[o] = (dbw[o] + dbw[G::HP + o]) + (dbw[2 * G::HP + o] + dbw[3 * G::HP + o]);
    }
    // next layer down
    if (l > 0) {
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) gpre[u][r] = gin[u][r] * (h[l > 0 ? l - 1 : 0][u][r] > 0.0 ? 1.0 : 0.01);
      __syncthreads();                       // everyone is done with Wcur / the tiles
      commit_layer<NT>(Wl + ((l - 1) & 1) * G::WSIZE, l - 1, NH, D, H, regs, breg);
      // (the g_pre / input tiles are rewritten at the top of the next iteration, followed by a barrier)
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + g + 4 * r;
        if (c < D && row < M) a.g_in[(size_t)(c & 1) * M * a.C + (size_t)row * a.C + (c >> 1)] = gin[0][r];
      }
    }
  }
}

template <int NT, int KSH>
static int launch_mlp_mfma(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  using G = Geo<NT>;
  constexpr int NH = 6;
  const int nblk = cdiv(a.M, 64);
  if (!backward) {
    size_t smem = sizeof(double) * (2 * G::WSIZE + 4 * G::TSIZE);
    auto kern = mlp_fwd_mfma_kernel<NT, NH, KSH>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(BLOCK), smem, stream, a);
  } else {
    size_t smem = sizeof(double) * (2 * G::WSIZE + 8 * G::TSIZE + 4 * 16 * G::S + 4 * G::HP);
    auto kern = mlp_bwd_mfma_kernel<NT, NH, KSH>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(BLOCK), smem, stream, a);
  }
  LGN_CHECK_LAUNCH();
  return 0;
}

// H <= 48, 2C <= 16, 7 Linear layers.  Returns -2 if the shape is outside this kernel's range.
int mlp_mfma_dispatch(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  if (a.nlin != 7 || a.H > 48 || 2 * a.C > 16 || a.H < 2 * a.C) return -2;
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
