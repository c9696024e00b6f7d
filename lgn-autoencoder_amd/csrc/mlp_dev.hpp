// lgn-autoencoder_amd/csrc/mlp_dev.hpp -- the CGMLP of a level as a PHASE of the level kernels (device code).
//
// Reference: the level loop is node_level then mlp_level per row (lgn/models/lgn_cg.py:164-172; CGMLP.forward
// lgn/models/lgn_levels.py:191-227).  A jet's rows never meet another jet's, so the CGMLP of a level can run where the
// level's scalars already are: as the tail of level_fwd2_kernel and as the head of level_bwd3_kernel (one workgroup = one
// jet, 4 waves).  Same arithmetic as mlp_mfma.hip, different mapping:
//
//   * Layers are computed TRANSPOSED, h^T = W x^T, with v_mfma_f64_16x16x4_f64 (D(16x16) += A(16x4) B(4x16);
//       A: lane l holds A[i = l&15][k = l>>4]   B: lane l holds B[k = l>>4][j = l&15]   D: lane l, register r: D[i = (l>>4) + 4r][j = l&15]).
//     A = weight tile, B = activations of 16 ROWS.  With the k-steps of the next layer ordered (tile u, register r) ->
//     k = 16u + 4r + (l>>4), the D registers of a layer ARE the B operands of the next one: a wave that owns a 16-row tile
//     runs the whole chain out of registers -- activations never touch LDS.  The same holds for the backward chain
//     g_h^T = W^T g_pre^T, whose result lands on the lanes / registers that hold the matching activation (slope in place).
//   * 30 rows = two 16-row tiles = two "chain" waves.  Forward: the other two waves only help to stream the weights.
//     Backward: the other two waves run the weight-gradient GEMMs  dW_l = g_pre_l^T h_{l-1}  (K = the jet's rows) on the
//     tiles the chain waves publish in LDS.  Which waves take which role is chosen from the SIMDs they run on (HW_ID): the
//     two workgroups a CU holds at bs = 512 then put their chain waves on different SIMD pairs and every SIMD sees the same
//     matrix-pipe load.  The choice changes who computes a tile, never its value.
//   * Weights stream per layer from L2 into two alternating LDS images (18 KB each), prefetched into registers one layer
//     ahead; the small first-layer image is loaded once.  One barrier per forward layer, two per backward layer.
//   * Jets of 33..40 particles take a second pass over rows 32..; jets split over several workgroups (level_jet_split) run
//     the forward on their own rows and the backward chain on all rows, sharing the dW tiles between the shares.
//
// Shapes covered: 7 Linear layers, 2 C_out <= 16 inputs, H <= 48 (every BASELINE maxdim = 2 level); anything else keeps the
// separate CGMLP kernels (level.hpp: level_mlp_fusable).
#pragma once
#include "common.hpp"

namespace lgn {
namespace fm {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NT = 3;            // 16-neuron tiles of a hidden layer
constexpr int HP = 16 * NT;      // padded hidden width
constexpr int S = 50;            // row stride of hidden images and G / X tiles (== 2 mod 4: conflict-free ds_read_b64 of fragments)
constexpr int S0 = 18;           // row stride of 16-column tiles (MLP input rows, first-layer image); column 16 = bias
constexpr int IMG = HP * S;      // one hidden-layer image: W[o][k] at o*S + k, bias at o*S + HP
constexpr int IMG0 = HP * S0;    // first-layer image
constexpr int ROWS = 32;         // rows per pass (two chain waves x 16)
constexpr int NLIN = 7;
constexpr int TILE = ROWS * S;   // one G or X tile of the backward

__host__ __device__ constexpr int passes(int rows) { return (rows + ROWS - 1) / ROWS; }
// offset of W_l in the contiguous parameter block (W_0, b_0, W_1, b_1, ...); the partial rows use the same layout
__host__ __device__ constexpr int off_w(int l, int D, int H) { return l == 0 ? 0 : (H * D + H) + (l - 1) * (H * H + H); }
__host__ __device__ constexpr int psize(int D, int H) { return off_w(NLIN - 1, D, H) + D * H + D; }
// LDS of the phase, in doubles
__host__ __device__ constexpr int fwd_alias_doubles() { return 2 * IMG; }                                   // over the sweep's dead data
__host__ __device__ constexpr int fwd_own_doubles(int rows) { return IMG0 + passes(rows) * ROWS * S0; }     // first-layer image | input rows
__host__ __device__ constexpr int bwd_doubles() { return 2 * IMG + IMG0 + ROWS * S0 + 2 * TILE; }           // images | W0 | X0 | G | X

struct Dims {
  int D, H;      // MLP input / output width 2 C_out, hidden width
  bool full;     // H > 36: all 12 k-steps over a hidden activation; else the first 9 (H = 36: C_out = 3).  Images are zero padded
                 // to 48 x 48, so surplus k-steps and tiles multiply zeros: wasted matrix work, never a wrong value.
};
__device__ __forceinline__ Dims make_dims(int D, int H) { return Dims{D, H, H > 36}; }

// ---- roles -----------------------------------------------------------------------------------------------------------
// rank of this wave in the order "waves on the preferred SIMD pair first": ranks 0, 1 = chain waves.  ids: 4 ints of LDS.
// (HW_ID: wave slot [3:0], SIMD [5:4].  Two workgroups of a CU get different slots on a SIMD; slot parity picks the pair.)
__device__ __forceinline__ void role_publish(int* ids, int wave, int lane) {
  const unsigned hw = __builtin_amdgcn_s_getreg((6 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID bits [5:0]
  if (lane == 0) ids[wave] = (int)hw;
}
// (after a barrier behind role_publish)
__device__ __forceinline__ int role_resolve(const int* ids, int wave) {
  const int parity = ids[0] & 1;
  int key[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) key[w] = (((ids[w] >> 5) & 1) != parity) ? 1 : 0;
  int role = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) role += (key[w] < key[wave] || (key[w] == key[wave] && w < wave)) ? 1 : 0;
  return __builtin_amdgcn_readfirstlane(role);
}
__device__ __forceinline__ int wave_role(int* ids, int wave, int lane) {
  role_publish(ids, wave, lane);
  __syncthreads();
  return role_resolve(ids, wave);
}

// ---- weight staging: global -> registers (issue) -> LDS image (commit) ---------------------------------------------------
// NSW staging waves; wave sw takes rows sw, sw + NSW, ... of the image, lane = column: one coalesced run of H doubles per
// load, the row test is wave uniform, the column test one mask, every LDS address a lane constant plus an immediate.  The
// whole padded image is written (zeros outside [Hout x H]).  All loads of a thread are in flight before the first is used.
template <int NSW>
struct WRegs {
  static constexpr int NR = HP / NSW;
  double w[NR];
  double b;
};
// layers 1 .. 6 (H input columns); the output layer fills 16 rows only
template <int NSW>
__device__ __forceinline__ void stage_issue(const double* __restrict__ wb, int l, const Dims& d, int sw, int lane, WRegs<NSW>& r) {
  // (the weights are read-only, so nothing else stops the compiler from hoisting the loads of ALL layers to the top of the
  // phase -- measured: 102 -> 255 registers)
  asm volatile("" ::: "memory");
  const int Hout = l == NLIN - 1 ? d.D : d.H, H = d.H;
  const double* W = wb + off_w(l, d.D, H);
  const int rows = l == NLIN - 1 ? 16 : HP;
  const bool kok = lane < H;
  // one 32-bit byte offset per lane, advanced by a uniform step: scalar base + lane offset addressing, no per-row address registers
  const char* base = reinterpret_cast<const char*>(W);
  unsigned voff = (unsigned)(sw * H + (kok ? lane : 0)) * 8u;
  const unsigned step = (unsigned)(NSW * H) * 8u;
#pragma unroll
  for (int i = 0; i < WRegs<NSW>::NR; ++i) {
    const int o = sw + NSW * i;                            // (wave uniform)
    r.w[i] = 0.0;
    if (o < rows) {                                        // (compile time for all but the output layer)
      const bool ok = kok && o < Hout;
      const double v = *reinterpret_cast<const double*>(base + (o < Hout ? voff : 0u));
      r.w[i] = ok ? v : 0.0;
    }
    voff += step;
  }
  r.b = 0.0;
  if (sw == 0) {
    const double bv = W[Hout * H + (lane < Hout ? lane : 0)];
    r.b = lane < Hout ? bv : 0.0;
  }
}
template <int NSW>
__device__ __forceinline__ void stage_commit(double* img, int l, int sw, int lane, const WRegs<NSW>& r) {
  const int rows = l == NLIN - 1 ? 16 : HP;
  if (lane < HP) {
    double* dst = img + sw * S + lane;
#pragma unroll
    for (int i = 0; i < WRegs<NSW>::NR; ++i)
      if (sw + NSW * i < rows) dst[NSW * i * S] = r.w[i];
    if (sw == 0) img[lane * S + HP] = r.b;
  }
}
// layer 0 (16 input columns, stride S0): lane = (row in a group of 4, column); wave sw takes row groups sw, sw + NSW, ...
template <int NSW>
struct W0Regs {
  static constexpr int NR = HP / 4 / NSW;
  double w[NR];
  double b;
};
template <int NSW>
__device__ __forceinline__ void stage0_issue(const double* __restrict__ wb, const Dims& d, int sw, int lane, W0Regs<NSW>& r) {
  const int k = lane & 15, og = lane >> 4;
#pragma unroll
  for (int i = 0; i < W0Regs<NSW>::NR; ++i) {
    const int o = 4 * (sw + NSW * i) + og;
    const bool ok = o < d.H && k < d.D;
    const double v = wb[ok ? o * d.D + k : 0];
    r.w[i] = ok ? v : 0.0;
  }
  r.b = 0.0;
  if (sw == 0) {
    const double bv = wb[d.H * d.D + (lane < d.H ? lane : 0)];
    r.b = lane < d.H ? bv : 0.0;
  }
}
template <int NSW>
__device__ __forceinline__ void stage0_commit(double* img0, int sw, int lane, const W0Regs<NSW>& r) {
  const int k = lane & 15, og = lane >> 4;
#pragma unroll
  for (int i = 0; i < W0Regs<NSW>::NR; ++i) img0[(4 * (sw + NSW * i) + og) * S0 + k] = r.w[i];
  if (sw == 0 && lane < HP) img0[lane * S0 + 16] = r.b;
}

// ---- chain: forward layers ---------------------------------------------------------------------------------------------
// h[u][r] = activation of neuron o = 16u + (lane>>4) + 4r for row (lane & 15) of the wave's tile
template <bool GEN>
__device__ __forceinline__ void layer_first(const double* img0, const double* x0 /* the tile's 16 rows, stride S0 */, int lane, int act,
                                            v4d (&h)[NT]) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[u][r] = img0[(16 * u + g + 4 * r) * S0 + 16];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double bv = x0[c * S0 + 4 * t + g];
#pragma unroll
    for (int u = 0; u < NT; ++u) h[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(img0[(16 * u + c) * S0 + 4 * t + g], bv, h[u], 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[u][r] = act_apply_t<GEN>(h[u][r], act);
}
// k-steps (input tile up, register r) = 4 up + r: 0 .. 8 always, 9 .. 11 under ONE wave-uniform branch (a branch per k-step would
// cut the layer into 12 scheduling regions of three matrix instructions each)
#define LGN_FM_KSTEPS(BODY)                                  \
  _Pragma("unroll") for (int ks_ = 0; ks_ < 9; ++ks_) {      \
    const int up = ks_ >> 2, r = ks_ & 3;                    \
    BODY                                                     \
  }                                                          \
  if (full) {                                                \
    _Pragma("unroll") for (int ks_ = 9; ks_ < 12; ++ks_) {   \
      const int up = ks_ >> 2, r = ks_ & 3;                  \
      BODY                                                   \
    }                                                        \
  }
template <bool GEN>
__device__ __forceinline__ void layer_hidden(const double* img, bool full, int lane, int act, const v4d (&hin)[NT], v4d (&h)[NT]) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[u][r] = img[(16 * u + g + 4 * r) * S + HP];
  LGN_FM_KSTEPS({
    const double bv = hin[up][r];
    _Pragma("unroll") for (int u = 0; u < NT; ++u)
      h[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(img[(16 * u + c) * S + 16 * up + 4 * r + g], bv, h[u], 0, 0, 0);
  })
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[u][r] = act_apply_t<GEN>(h[u][r], act);
}
// output layer: out[r] = feature k = (lane>>4) + 4r (k < D) of row (lane & 15); no activation
__device__ __forceinline__ v4d layer_out(const double* img, bool full, int lane, const v4d (&hin)[NT]) {
  const int c = lane & 15, g = lane >> 4;
  v4d o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = img[(g + 4 * r) * S + HP];
  LGN_FM_KSTEPS({ o = __builtin_amdgcn_mfma_f64_16x16x4f64(img[c * S + 16 * up + 4 * r + g], hin[up][r], o, 0, 0, 0); })
  return o;
}

// ---- forward phase (tail of level_fwd2_kernel; every thread of the 256-thread workgroup calls it) ---------------------------
//   img   : 2 * IMG doubles of LDS (may alias anything dead after the caller's last barrier)
//   img0  : first-layer image, staged by the caller at kernel start (stage0_issue / stage0_commit) -- visible
//   x0    : the workgroup's `nrows` MLP input rows [row][S0] (k = 2c + z, zero padded to 16 columns and to whole passes) -- visible
//   wr    : registers holding W_1, issued by the caller (stage_issue<4>(wb, 1, ...))
//   s_out : &out[plane 0][first row of this workgroup][channel 0]; plane = stride between re / im
// The caller must put a barrier between this call and any reuse of img / x0.
template <bool GEN>
__device__ __forceinline__ void fwd_phase(const double* wb, const Dims& d, int act, double* img, const double* img0,
                                          const double* x0, int nrows, WRegs<4>& wr, int role, double* __restrict__ s_out, size_t plane, int CO) {
  const int tid = threadIdx.x, lane = tid & 63, sw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  for (int p = 0; p * ROWS < nrows; ++p) {
    // (an opaque copy of the pointer per pass: otherwise the address of every weight load of every layer -- 144 registers -- is
    // hoisted out of this loop as an invariant)
    asm volatile("" : "+s"(wb));
    const int row0 = p * ROWS + 16 * role;                 // this chain wave's tile
    const bool chain = role < 2 && row0 < nrows;
    if (p > 0) stage_issue<4>(wb, 1, d, sw, lane, wr);
    stage_commit<4>(img + IMG, 1, sw, lane, wr);
    stage_issue<4>(wb, 2, d, sw, lane, wr);
    v4d h[2][NT];
    if (chain) layer_first<GEN>(img0, x0 + row0 * S0, lane, act, h[0]);
    __syncthreads();
#pragma unroll
    for (int l = 1; l <= 5; ++l) {
      stage_commit<4>(img + ((l + 1) & 1) * IMG, l + 1, sw, lane, wr);
      if (l + 2 < NLIN) stage_issue<4>(wb, l + 2, d, sw, lane, wr);
      if (chain) layer_hidden<GEN>(img + (l & 1) * IMG, d.full, lane, act, h[(l - 1) & 1], h[l & 1]);
      __syncthreads();
    }
    if (chain) {
      const v4d o = layer_out(img, d.full, lane, h[1]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = g + 4 * r, row = row0 + c;
        if (k < d.D && row < nrows) s_out[(size_t)(k & 1) * plane + (size_t)row * CO + (k >> 1)] = o[r];
      }
    }
  }
}

// ---- backward phase (head of level_bwd3_kernel) ------------------------------------------------------------------------------
// gh[uk] = W_l^T g_pre:  A = W_l[o = 16u + 4r + (lane>>4)][k = 16uk + (lane&15)], B = g_pre registers
__device__ __forceinline__ void chain_gin(const double* img, bool full, int lane, const v4d (&gp)[NT], v4d (&gh)[NT]) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int uk = 0; uk < NT; ++uk) gh[uk] = v4d{0, 0, 0, 0};
  LGN_FM_KSTEPS({
    const double bv = gp[up][r];
    _Pragma("unroll") for (int uk = 0; uk < NT; ++uk)
      gh[uk] = __builtin_amdgcn_mfma_f64_16x16x4f64(img[(16 * up + 4 * r + g) * S + 16 * uk + c], bv, gh[uk], 0, 0, 0);
  })
}
// the output layer has <= 16 neurons: four k-steps
__device__ __forceinline__ void chain_gin_out(const double* img, int lane, const v4d& gp0, v4d (&gh)[NT]) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int uk = 0; uk < NT; ++uk) gh[uk] = v4d{0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int uk = 0; uk < NT; ++uk)
      gh[uk] = __builtin_amdgcn_mfma_f64_16x16x4f64(img[(4 * r + g) * S + 16 * uk + c], gp0[r], gh[uk], 0, 0, 0);
}
// gradient w.r.t. the MLP input: gx[r] = feature k = (lane>>4) + 4r of row (lane & 15)
__device__ __forceinline__ v4d chain_gin_first(const double* img0, bool full, int lane, const v4d (&gp)[NT]) {
  const int c = lane & 15, g = lane >> 4;
  v4d gx = v4d{0, 0, 0, 0};
  LGN_FM_KSTEPS({ gx = __builtin_amdgcn_mfma_f64_16x16x4f64(img0[(16 * up + 4 * r + g) * S0 + c], gp[up][r], gx, 0, 0, 0); })
  return gx;
}
// publish a chain wave's registers (D layout) as rows of a [row][neuron] tile
template <int NTL>
__device__ __forceinline__ void put_tile(double* tile16 /* the wave's 16 rows */, int lane, const v4d (&x)[NT]) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int u = 0; u < NTL; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) tile16[c * S + 16 * u + g + 4 * r] = x[u][r];
}
// dW_l tiles q, q + Q, ... of this pass's 32 rows -> the partial row; worker Q - 1 also sums the bias gradient
__device__ __forceinline__ void dw_tiles(int l, const double* Gt, const double* Xt /* l == 0: the input rows, stride S0 */, double* __restrict__ prow,
                                         const Dims& d, int q, int Q, int lane) {
  const int c = lane & 15, g = lane >> 4;
  const int Hin = l == 0 ? d.D : d.H, Hout = l == NLIN - 1 ? d.D : d.H;
  const int nto = (d.H + 15) >> 4;
  const int nto_l = l == NLIN - 1 ? 1 : nto, ntk_l = l == 0 ? 1 : nto, total = nto_l * ntk_l;
  const int xs = l == 0 ? S0 : S;
  double* pW = prow + off_w(l, d.D, d.H);
  for (int tix = q; tix < total; tix += Q) {
    const int u = tix / ntk_l, uk = tix - u * ntk_l;
    const double* ga = Gt + g * S + 16 * u + c;
    const double* xb = Xt + g * xs + 16 * uk + c;
    v4d a0 = v4d{0, 0, 0, 0}, a1 = v4d{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[4 * t * S], xb[4 * t * xs], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[(16 + 4 * t) * S], xb[(16 + 4 * t) * xs], a1, 0, 0, 0);
    }
    const int k = 16 * uk + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * u + g + 4 * r;
      if (o < Hout && k < Hin) pW[o * Hin + k] = a0[r] + a1[r];
    }
  }
  if (q == Q - 1 && lane < Hout) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int row = 0; row < ROWS; row += 2) {
      s0 += Gt[row * S + lane];
      s1 += Gt[(row + 1) * S + lane];
    }
    pW[Hout * Hin + lane] = s0 + s1;
  }
}

// Everything the backward phase needs.
struct BwdIo {
  const double* wb;        // parameter block
  const double* s_pre;     // [2][B][N][CO] MLP input (the level's scalars before the MLP)
  const double* g_out;     // [2][B][N][CO] gradient w.r.t. the MLP output
  double* part;            // [B * passes(N)][psize]: this jet's rows at (b * passes + p) * psize
  int B, N, CO, H, b, act;
  int share, nshare;       // level_jet_split: blockIdx.y / gridDim.y (every share runs the chain on all rows; the dW tiles are shared out)
};
// lds: bwd_doubles() doubles, dead afterwards; ids: 4 ints; gsx: [N][2 CO] doubles that stay alive for the caller -- the gradient
// w.r.t. the MLP input, i.e. the level's upstream scalar gradient (feature k = 2c + z).  Ends with a barrier: gsx is visible.
// The two roles are two PROGRAMS with the same barrier sequence (a wave-uniform branch around whole loops), so that the chain's
// 72 activation registers and the workers' 25 staging registers are never live in the same wave.
template <bool GEN>
__device__ __forceinline__ void bwd_phase(const BwdIo& io, double* lds, int* ids, double* gsx) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const Dims d = make_dims(2 * io.CO, io.H);
  double* img = lds;
  double* img0 = img + 2 * IMG;
  double* x0 = img0 + IMG0;
  double* Gt = x0 + ROWS * S0;
  double* Xt = Gt + TILE;
  const int role = wave_role(ids, wave, lane);
  const size_t plane = (size_t)io.B * io.N * io.CO;
  const int np = passes(io.N), ps = psize(d.D, d.H);
  const double* wb = io.wb;
  for (int p = 0; p < np; ++p) {
    asm volatile("" : "+s"(wb));                           // (see fwd_phase)
    // the pass's input rows (every thread: 2 of the 512 elements)
    {
      double xv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + BLOCK * i, r = e >> 4, k = e & 15, row = p * ROWS + r;
        const bool ok = row < io.N && k < d.D;
        const double v = io.s_pre[ok ? (size_t)(k & 1) * plane + ((size_t)io.b * io.N + row) * io.CO + (k >> 1) : 0];
        xv[i] = ok ? v : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) x0[((tid + BLOCK * i) >> 4) * S0 + ((tid + BLOCK * i) & 15)] = xv[i];
    }
    if (role < 2) {
      // ================= chain program: this wave's 16 rows through the whole MLP, forward then backward =================
      const int row = p * ROWS + 16 * role + c;            // the lane's row (D layout: j = lane & 15)
      v4d gp[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) gp[u] = v4d{0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = g + 4 * r;
        const bool ok = k < d.D && row < io.N;
        const double v = io.g_out[ok ? (size_t)(k & 1) * plane + ((size_t)io.b * io.N + row) * io.CO + (k >> 1) : 0];
        gp[0][r] = ok ? v : 0.0;
      }
      __syncthreads();                                      // x0, first-layer image
      v4d h[NLIN - 1][NT];
      layer_first<GEN>(img0, x0 + 16 * role * S0, lane, io.act, h[0]);
      __syncthreads();
#pragma unroll
      for (int l = 1; l <= 5; ++l) {
        layer_hidden<GEN>(img + (l & 1) * IMG, d.full, lane, io.act, h[l - 1], h[l]);
        __syncthreads();
      }
#pragma unroll
      for (int l = NLIN - 1; l >= 0; --l) {
        if (l == NLIN - 1) put_tile<1>(Gt + 16 * role * S, lane, gp);
        else put_tile<NT>(Gt + 16 * role * S, lane, gp);
        if (l >= 1) put_tile<NT>(Xt + 16 * role * S, lane, h[l >= 1 ? l - 1 : 0]);
        __syncthreads();                                    // (A) tiles of layer l published
        if (l >= 1) {
          v4d gh[NT];
          if (l == NLIN - 1) chain_gin_out(img + (l & 1) * IMG, lane, gp[0], gh);
          else chain_gin(img + (l & 1) * IMG, d.full, lane, gp, gh);
#pragma unroll
          for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[u][r] = gh[u][r] * act_slope_t<GEN>(h[l >= 1 ? l - 1 : 0][u][r], io.act);
        } else {
          const v4d gx = chain_gin_first(img0, d.full, lane, gp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = g + 4 * r;
            if (k < d.D && row < io.N) gsx[row * d.D + k] = gx[r];
          }
        }
        __syncthreads();                                    // (B) tiles consumed
      }
    } else {
      // ================= worker program: streams the weights, runs the weight-gradient GEMMs =================
      const int sw = role - 2;                             // staging wave 0 / 1
      const int q = io.share * 2 + (role - 2), Q = io.nshare * 2;
      double* prow = io.part + ((size_t)io.b * np + p) * ps;
      WRegs<2> wr;
      {
        W0Regs<2> w0;
        if (p == 0) stage0_issue<2>(wb, d, sw, lane, w0);
        stage_issue<2>(wb, 1, d, sw, lane, wr);
        if (p == 0) stage0_commit<2>(img0, sw, lane, w0);
      }
      __syncthreads();
      stage_commit<2>(img + IMG, 1, sw, lane, wr);
      stage_issue<2>(wb, 2, d, sw, lane, wr);
      __syncthreads();
#pragma unroll
      for (int l = 1; l <= 5; ++l) {
        // W_{l+1} into the image last read at layer l - 1; then the layer after it -- at l = 5 the first reload of the backward sweep
        stage_commit<2>(img + ((l + 1) & 1) * IMG, l + 1, sw, lane, wr);
        stage_issue<2>(wb, l + 2 < NLIN ? l + 2 : 4, d, sw, lane, wr);     // l = 4: W_6; l = 5: W_4
        __syncthreads();
      }
      // images now: [0] = W_6, [1] = W_5; registers: W_4
#pragma unroll
      for (int l = NLIN - 1; l >= 0; --l) {
        __syncthreads();                                    // (A)
        // every chain wave is past layer l + 1: its image takes W_{l-1}
        if (l <= 5 && l >= 2) stage_commit<2>(img + ((l - 1) & 1) * IMG, l - 1, sw, lane, wr);
        if (l <= 5 && l >= 3) stage_issue<2>(wb, l - 2, d, sw, lane, wr);
        dw_tiles(l, Gt, l == 0 ? x0 : Xt, prow, d, q, Q, lane);
        __syncthreads();                                    // (B)
      }
    }
  }
}

}  // namespace fm
}  // namespace lgn
