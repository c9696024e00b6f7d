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

// phase stamps of the debug build (make stamps; tools/kbench.py KB_STAMPS=lgn_debug_stamps_fm_*): lane 0 of every wave of workgroup
// 0 records the shader clock at FM_STAMP(role, i) -> slot role * 32 + i
#ifdef LGN_STAMPS
// (slots 128 .. 255: the same for the LAST workgroup of the grid)
static __device__ long long fm_stamps[256];
#define FM_STAMP(role, i) do { if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) \
    fm_stamps[(blockIdx.x == 0 ? 0 : 128) + (role) * 32 + (i)] = clock64(); } while (0)
#define FM_STAMP_READER(name) \
  extern "C" int name(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lgn::fm::fm_stamps), sizeof(long long) * 256); }
#else
#define FM_STAMP(role, i) do { } while (0)
#define FM_STAMP_READER(name)
#endif

// Workgroup barrier that orders LDS only.  __syncthreads() also waits for every global access in flight (s_waitcnt vmcnt(0)): the
// weight prefetch of the next layer and, in the backward, 100 KB of weight-gradient stores per workgroup -- measured: a backward layer
// step waits 2 - 3 us for its own stores.  Nothing the phases exchange between waves goes through global memory.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

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
__host__ __device__ constexpr int fwd_own_doubles(int rows) { return IMG0 + IMG + passes(rows) * ROWS * S0; }   // images of layers 0, 1 | input rows
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
#ifdef LGN_STAMPS
  if (lane == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) fm_stamps[(blockIdx.x == 0 ? 0 : 128) + wave * 32 + 31] = 0x1000 + (long long)hw;
#endif
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
// layers 1 .. 6 (H input columns); the output layer (OUT) fills 16 rows only.  Straight-line code: every row index is compared with
// compile-time bounds only, run-time tests (o < Hout, lane < H) select the ADDRESS (clamped) and the VALUE -- a branch around a load
// would put the load and its s_waitcnt into a block of its own and serialise the twelve round trips (measured: 3 us per layer).
template <int NSW, bool OUT>
__device__ __forceinline__ void stage_issue_t(const double* __restrict__ wb, int l, const Dims& d, int sw, int lane, WRegs<NSW>& r) {
  // (the weights are read-only: without this nothing stops the compiler from hoisting the loads of ALL layers to the top of the phase)
  asm volatile("" ::: "memory");
  const int Hout = OUT ? d.D : d.H, H = d.H;
  const double* W = wb + off_w(l, d.D, H);
  constexpr int rows = OUT ? 16 : HP;
  const bool kok = lane < H;
  // one 32-bit element offset per lane, advanced by a uniform step
  unsigned voff = (unsigned)(sw * H + (kok ? lane : 0));
  const unsigned step = (unsigned)(NSW * H);
#pragma unroll
  for (int i = 0; i < WRegs<NSW>::NR; ++i) {
    r.w[i] = 0.0;
    if (NSW * i < rows) {                                  // (compile time: sw < NSW)
      const bool ok = kok && (sw + NSW * i) < Hout;
      const double v = W[(sw + NSW * i) < Hout ? voff : 0u];
      r.w[i] = ok ? v : 0.0;
    }
    voff += step;
  }
  const double bv = W[Hout * H + (lane < Hout ? lane : 0)];
  r.b = lane < Hout ? bv : 0.0;
}
template <int NSW>
__device__ __forceinline__ void stage_issue(const double* __restrict__ wb, int l, const Dims& d, int sw, int lane, WRegs<NSW>& r) {
  if (l == NLIN - 1) stage_issue_t<NSW, true>(wb, l, d, sw, lane, r);
  else stage_issue_t<NSW, false>(wb, l, d, sw, lane, r);
}
template <int NSW>
__device__ __forceinline__ void stage_commit(double* img, int l, int sw, int lane, const WRegs<NSW>& r) {
  const int rows = l == NLIN - 1 ? 16 : HP;
  if (lane < HP) {
    double* dst = img + sw * S + lane;
#pragma unroll
    for (int i = 0; i < WRegs<NSW>::NR; ++i)
      if (NSW * i < rows) dst[NSW * i * S] = r.w[i];       // (compile time at every call site: l is a literal)
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
  const double bv = wb[d.H * d.D + (lane < d.H ? lane : 0)];
  r.b = lane < d.H ? bv : 0.0;
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
//   img   : 2 * IMG doubles of LDS, images A | B (may alias anything dead after the caller's last barrier)
//   img0, img1 : images of the first two layers, staged by the caller at kernel start (stage0_* / stage_*<4>(.., 1, ..)) -- visible
//   x0    : the workgroup's `nrows` MLP input rows [row][S0] (k = 2c + z, zero padded to 16 columns and to whole passes) -- visible
//   w2, w3 : registers with W_2, W_3, issued by the caller (all four waves, stage_issue<4>) BEFORE its last phase: a weight image is
//            4 - 5 k cycles away (512 workgroups pull the same 18 KB through L2 at once), more than a layer lasts
//   s_out : &out[plane 0][first row of this workgroup][channel 0]; plane = stride between re / im
// Two programs with the same barrier sequence: the chain waves (roles 0, 1: one 16-row tile each) only compute, the other two waves
// stream the weights, two layers ahead, from two register sets:
//   chain   first      | 1 (img1) | 2 (A) | 3 (B)        | 4 (A)     | 5 (B)     | out (A)
//   worker  ld W4, W5  | --       | --    | W4->A, ld W6 | W5->B     | W6->A     |
// (W_2 -> A and W_3 -> B are committed by everybody at the start.)  The caller must put a barrier between this call and any reuse
// of img / x0.
template <bool GEN>
__device__ __forceinline__ void fwd_phase(const double* __restrict__ wb_in, const Dims& d, int act, double* img, const double* img0,
                                          const double* img1, const double* x0, int nrows, WRegs<4>& w2, WRegs<4>& w3, int role,
                                          double* __restrict__ s_out, size_t plane, int CO) {
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* imgA = img;
  double* imgB = img + IMG;
  for (int p = 0; p * ROWS < nrows; ++p) {
    // (opaque per pass: every lane-derived LDS address and every weight address of the pass is otherwise hoisted out of this loop
    // as an invariant, all of them live at once -- hundreds of registers, spilled at the kernel's entry)
    int lane = tid & 63, opaque0 = 0;
    asm volatile("" : "+v"(lane), "+s"(opaque0));
    const double* __restrict__ wb = wb_in + opaque0;
    const int c = lane & 15, g = lane >> 4;
    if (p > 0) {
      stage_issue<4>(wb, 2, d, wave, lane, w2);
      stage_issue<4>(wb, 3, d, wave, lane, w3);
    }
    stage_commit<4>(imgA, 2, wave, lane, w2);
    stage_commit<4>(imgB, 3, wave, lane, w3);
    if (role < 2) {
      const int row0 = p * ROWS + 16 * role;               // this chain wave's tile
      const bool chain = row0 < nrows;
      v4d h[2][NT];
      FM_STAMP(role, 0);
      if (chain) layer_first<GEN>(img0, x0 + row0 * S0, lane, act, h[0]);
      FM_STAMP(role, 1);
      lds_barrier();
      FM_STAMP(role, 2);
#pragma unroll
      for (int l = 1; l <= 5; ++l) {
        if (chain) layer_hidden<GEN>(l == 1 ? img1 : ((l & 1) ? imgB : imgA), d.full, lane, act, h[(l - 1) & 1], h[l & 1]);
        FM_STAMP(role, 2 + 3 * l);
        lds_barrier();
        FM_STAMP(role, 3 + 3 * l);
      }
      if (chain) {
        const v4d o = layer_out(imgA, d.full, lane, h[1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = g + 4 * r, row = row0 + c;
          if (k < d.D && row < nrows) s_out[(size_t)(k & 1) * plane + (size_t)row * CO + (k >> 1)] = o[r];
        }
      }
      FM_STAMP(role, 19);
    } else {
      const int sw = role - 2;
      WRegs<2> r0, r1;
      FM_STAMP(role, 0);
      stage_issue<2>(wb, 4, d, sw, lane, r0);
      stage_issue<2>(wb, 5, d, sw, lane, r1);
      FM_STAMP(role, 1);
      lds_barrier();                                       // chain: first layer done
      lds_barrier();                                       // layer 1 (img1)
      lds_barrier();                                       // layer 2 (A)
      FM_STAMP(role, 9);
      stage_commit<2>(imgA, 4, sw, lane, r0);
      stage_issue<2>(wb, 6, d, sw, lane, r0);
      FM_STAMP(role, 10);
      lds_barrier();                                       // layer 3 (B)
      stage_commit<2>(imgB, 5, sw, lane, r1);
      lds_barrier();                                       // layer 4 (A)
      stage_commit<2>(imgA, 6, sw, lane, r0);
      FM_STAMP(role, 17);
      lds_barrier();                                       // layer 5 (B)
      FM_STAMP(role, 18);
    }
    if ((p + 1) * ROWS < nrows) lds_barrier();             // the next pass rewrites A while this pass's output layer may still read it
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
// one 16 x 16 tile (o-tile u, k-tile uk) of dW_l over this pass's 32 rows, two accumulation chains; stored into the partial row
__device__ __forceinline__ void dw_tile(const double* Gt, const double* Xt, int xs, int u, int uk, double* __restrict__ pW, int Hout, int Hin, int lane) {
  const int c = lane & 15, g = lane >> 4;
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
// dW_l tiles q, q + Q, ... of this pass's 32 rows -> the partial row; worker Q - 1 also sums the bias gradient.  L is a compile-time
// layer index.  Q == 2 (the jet is not split): the worker's tiles are a compile-time list, so that the fragment reads of the next tile
// are in flight under the matrix instructions of the current one.
template <int L>
__device__ __forceinline__ void dw_tiles(const double* Gt, const double* Xt /* L == 0: the input rows, stride S0 */, double* __restrict__ prow,
                                         const Dims& d, int q, int Q, int lane) {
  const int Hin = L == 0 ? d.D : d.H, Hout = L == NLIN - 1 ? d.D : d.H;
  constexpr int xs = L == 0 ? S0 : S;
  double* pW = prow + off_w(L, d.D, d.H);
  if (Q == 2 && d.H > 32) {                    // three live tiles per side
    constexpr int NU = L == NLIN - 1 ? 1 : NT, NK = L == 0 ? 1 : NT;
    if (q == 0) {
#pragma unroll
      for (int tix = 0; tix < NU * NK; tix += 2) dw_tile(Gt, Xt, xs, tix / NK, tix % NK, pW, Hout, Hin, lane);
    } else {
#pragma unroll
      for (int tix = 1; tix < NU * NK; tix += 2) dw_tile(Gt, Xt, xs, tix / NK, tix % NK, pW, Hout, Hin, lane);
    }
  } else {
    const int nto = (d.H + 15) >> 4;
    const int nto_l = L == NLIN - 1 ? 1 : nto, ntk_l = L == 0 ? 1 : nto, total = nto_l * ntk_l;
    for (int tix = q; tix < total; tix += Q) {
      const int u = tix / ntk_l, uk = tix - u * ntk_l;
      dw_tile(Gt, Xt, xs, u, uk, pW, Hout, Hin, lane);
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

// one backward layer of the worker program (L compile time): barrier (A), the weight image two layers down, the layer's dW tiles,
// barrier (B)
template <int L>
__device__ __forceinline__ void worker_bwd_layer(const double* __restrict__ wb, const Dims& d, double* img, const double* Gt, const double* Xt,
                                                 const double* x0, double* __restrict__ prow, WRegs<2>& wr, int sw, int q, int Q, int lane, int role) {
  lds_barrier();                                        // (A)
  FM_STAMP(role, 8 + 2 * (NLIN - 1 - L));
  // every chain wave is past layer L + 1: its image takes W_{L-1}
  if (L <= 5 && L >= 2) stage_commit<2>(img + ((L - 1) & 1) * IMG, L - 1, sw, lane, wr);
  if (L <= 5 && L >= 3) stage_issue<2>(wb, L - 2, d, sw, lane, wr);
  dw_tiles<L>(Gt, L == 0 ? x0 : Xt, prow, d, q, Q, lane);
  FM_STAMP(role, 22 + (NLIN - 1 - L));
  lds_barrier();                                        // (B)
  FM_STAMP(role, 9 + 2 * (NLIN - 1 - L));
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
  const int tid = threadIdx.x, wave = tid >> 6;
  const Dims d = make_dims(2 * io.CO, io.H);
  double* img = lds;
  double* img0 = img + 2 * IMG;
  double* x0 = img0 + IMG0;
  double* Gt = x0 + ROWS * S0;
  double* Xt = Gt + TILE;
  const int role = wave_role(ids, wave, tid & 63);
  const size_t plane = (size_t)io.B * io.N * io.CO;
  const int np = passes(io.N), ps = psize(d.D, d.H);
  for (int p = 0; p < np; ++p) {
    int opaque0 = 0, lane = tid & 63;
    asm volatile("" : "+s"(opaque0), "+v"(lane));          // (see fwd_phase: nothing of the pass may be hoisted out of this loop)
    const double* __restrict__ wb = io.wb + opaque0;
    const int c = lane & 15, g = lane >> 4;
    // Everything the pass needs from global memory is requested before anything is waited for (one round trip, ~5 k cycles, at the head
    // of the kernel): the workers' first two weight images, then the pass's input rows (every thread: 2 of the 512 elements)
    WRegs<2> wr;
    W0Regs<2> w0;
    if (role >= 2) {
      if (p == 0) stage0_issue<2>(wb, d, role - 2, lane, w0);
      stage_issue<2>(wb, 1, d, role - 2, lane, wr);
    }
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
      FM_STAMP(role, 0);
      lds_barrier();                                      // x0, first-layer image
      FM_STAMP(role, 1);
      v4d h[NLIN - 1][NT];
      layer_first<GEN>(img0, x0 + 16 * role * S0, lane, io.act, h[0]);
      lds_barrier();
      FM_STAMP(role, 2);
#pragma unroll
      for (int l = 1; l <= 5; ++l) {
        layer_hidden<GEN>(img + (l & 1) * IMG, d.full, lane, io.act, h[l - 1], h[l]);
        lds_barrier();
        FM_STAMP(role, 2 + l);
      }
#pragma unroll
      for (int l = NLIN - 1; l >= 0; --l) {
        if (l == NLIN - 1) put_tile<1>(Gt + 16 * role * S, lane, gp);
        else put_tile<NT>(Gt + 16 * role * S, lane, gp);
        if (l >= 1) put_tile<NT>(Xt + 16 * role * S, lane, h[l >= 1 ? l - 1 : 0]);
        lds_barrier();                                    // (A) tiles of layer l published
        FM_STAMP(role, 8 + 2 * (NLIN - 1 - l));
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
        FM_STAMP(role, 22 + (NLIN - 1 - l));
        lds_barrier();                                    // (B) tiles consumed
        FM_STAMP(role, 9 + 2 * (NLIN - 1 - l));
      }
    } else {
      // ================= worker program: streams the weights, runs the weight-gradient GEMMs =================
      const int sw = role - 2;                             // staging wave 0 / 1
      const int q = io.share * 2 + (role - 2), Q = io.nshare * 2;
      double* prow = io.part + ((size_t)io.b * np + p) * ps;
      if (p == 0) stage0_commit<2>(img0, sw, lane, w0);
      FM_STAMP(role, 0);
      lds_barrier();
      FM_STAMP(role, 1);
      stage_commit<2>(img + IMG, 1, sw, lane, wr);
      stage_issue<2>(wb, 2, d, sw, lane, wr);
      lds_barrier();
      FM_STAMP(role, 2);
#pragma unroll
      for (int l = 1; l <= 5; ++l) {
        // W_{l+1} into the image last read at layer l - 1; then the layer after it -- at l = 5 the first reload of the backward sweep
        stage_commit<2>(img + ((l + 1) & 1) * IMG, l + 1, sw, lane, wr);
        stage_issue<2>(wb, l + 2 < NLIN ? l + 2 : 4, d, sw, lane, wr);     // l = 4: W_6; l = 5: W_4
        lds_barrier();
        FM_STAMP(role, 2 + l);
      }
      // images now: [0] = W_6, [1] = W_5; registers: W_4
      worker_bwd_layer<6>(wb, d, img, Gt, Xt, x0, prow, wr, sw, q, Q, lane, role);
      worker_bwd_layer<5>(wb, d, img, Gt, Xt, x0, prow, wr, sw, q, Q, lane, role);
      worker_bwd_layer<4>(wb, d, img, Gt, Xt, x0, prow, wr, sw, q, Q, lane, role);
      worker_bwd_layer<3>(wb, d, img, Gt, Xt, x0, prow, wr, sw, q, Q, lane, role);
      worker_bwd_layer<2>(wb, d, img, Gt, Xt, x0, prow, wr, sw, q, Q, lane, role);
      worker_bwd_layer<1>(wb, d, img, Gt, Xt, x0, prow, wr, sw, q, Q, lane, role);
      worker_bwd_layer<0>(wb, d, img, Gt, Xt, x0, prow, wr, sw, q, Q, lane, role);
    }
  }
}

}  // namespace fm
}  // namespace lgn
