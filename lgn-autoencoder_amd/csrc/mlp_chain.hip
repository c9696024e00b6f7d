// lgn-autoencoder_amd/csrc/mlp_chain.hip -- CGMLP forward / backward for large batches: one wave carries a 16-row tile through
// ALL layers out of registers (round 5).
//
// Same operator as mlp_mfma.hip (reference: CGMLP.forward, lgn/models/lgn_levels.py:191-227), for the reference widths
// H = 6 * 2C, C <= 4, and batches that give every CU a 64-row workgroup.  Layers are computed TRANSPOSED,
//     h_l^T [H x rows] = W_l [H x Hin] . h_{l-1}^T [Hin x rows],
// with v_mfma_f64_16x16x4_f64 (D(16x16) += A(16x4) B(4x16)):
//   A: lane l holds A[i = l&15][k = l>>4]      B: lane l holds B[k = l>>4][j = l&15]
//   D: lane l, register r holds D[i = (l>>4) + 4r][j = l&15]
// With c = l&15, g = l>>4: the D register r of output tile u, D[neuron 16u + 4r + g][row c], IS the B operand of the next
// layer's k-step (u, r) (k = 16u + 4r + g): activations never leave the registers, the A operands are fragments of the weight
// image W[o][k] in LDS, one ds_read_b64 per matrix instruction.  A wave owns 16 rows; a workgroup = 4 waves = 64 rows shares
// the weight images (double buffered, staged through registers one layer ahead: ONE barrier per layer, and no wave ever waits
// for another wave's activations).  In mlp_mfma.hip a wave owns one 16 x 16 tile of every layer and the whole workgroup
// exchanges activations through LDS at every layer: thirteen latency-exposed layer steps in the backward.
//
// Backward, layer l (weights W_l [Hout x Hin], g_pre = d loss / d pre-activation, transposed like everything else):
//   g_in^T = W_l^T g_pre^T    A = fragments of the SAME image read transposed, B = the D registers of g_pre^T: registers only
//   dW_l   = g_pre^T h_in     K = rows: both operands have the row index on l&15, where a matrix instruction never contracts,
//                             so each wave publishes its 16 columns of g_pre^T and h_in^T into two LDS tiles [neuron][64 rows]
//                             (double buffered) and the tile rows of dW_l are dealt to the waves (owner = (t + l) mod 4, three
//                             16 x 16 tiles that share their A fragment): K = 64 rows per matrix chain, one partial row per
//                             64-row workgroup as before.  dW_l runs one step BEHIND the chain (after the barrier that follows
//                             the publication), so the one barrier per layer also covers the tiles.
//   db_l   = row sums of g_pre^T: the owner of a tile row adds up the A fragments it loads anyway (16 adds + 2 shuffles).
// The hidden activations are recomputed (six more layer steps at the head of the kernel) and stay in registers: 72 doubles.
#include "ops.hpp"

namespace lgn {

typedef double v4d __attribute__((ext_vector_type(4)));
namespace { LGN_STAMP_DECL }
#ifdef LGN_STAMPS
#define DSTAMP(i) do { if (threadIdx.x == 256 && blockIdx.x == 0) g_stamps[i] = clock64(); } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif
LGN_STAMP_READER(lgn_debug_stamps_mlp_chain)

namespace chain {

__host__ __device__ constexpr int stride2mod4(int x) { return x + ((6 - (x & 3)) & 3); }   // smallest >= x with == 2 (mod 4)

// LDS-only barrier: __syncthreads() also drains the global loads of the weight prefetch and the dW stores
// It is also a fence for the instruction scheduler, which would otherwise hoist the NEXT step's image commit (and the wait for its
// global loads) above this step's matrix instructions.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  __builtin_amdgcn_sched_barrier(0);
}
// A block of N matrix instructions whose A operands come from LDS, one ds_read per instruction, as an explicit software pipeline:
// P reads ahead of the instruction that consumes them, order pinned (one wave per SIMD: nobody else hides the LDS latency, and
// left alone the scheduler sinks every read to just above its use).
// side(i) runs in the shadow of matrix instruction i (64 cycles of the pipe each; the wave keeps issuing): everything a step has
// to do besides its matrix work -- image commits and requests, activations of finished tiles, tile publication, gradient
// stores -- is cut into pieces and dealt to the items, because with one wave per SIMD whatever is issued between two streams
// leaves the matrix pipe idle (measured: 1 200 - 2 200 cycles per step before this, for 2 300 - 5 400 cycles of pipe time).
template <int N, int P, class LoadF, class MmaF, class SideF>
__device__ __forceinline__ void mfma_stream(LoadF load, MmaF mma, SideF side) {
  double q[N];
#pragma unroll
  for (int i = 0; i < (P < N ? P : N); ++i) q[i] = load(i);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i + P < N) q[i + P] = load(i + P);
    mma(i, q[i]);
    side(i);
    __builtin_amdgcn_sched_barrier(0);
  }
}
constexpr int LOOKAHEAD = 9;
// pieces [i T / N, (i + 1) T / N) of a list of T pieces, for item i of N
template <int N, int T, class F>
__device__ __forceinline__ void deal(int i, F piece) {
  const int j0 = (i * T + N - 1) / N, j1 = ((i + 1) * T + N - 1) / N;       // (constants once the item loop is unrolled)
#pragma unroll
  for (int j = j0; j < j1; ++j) piece(j);
}
// keeps a value's computation on this side of the next scheduling fence (instruction selection places unchained arithmetic freely)
__device__ __forceinline__ void pin(double& x) { asm volatile("" : "+v"(x)); }

#ifndef LGN_DBG_NOSTAGE
#define LGN_DBG_NOSTAGE 0
#endif
#ifndef LGN_DBG_NOACT
#define LGN_DBG_NOACT 0
#endif
template <bool GEN>
__device__ __forceinline__ double act_t(double x, int act) {
  if (LGN_DBG_NOACT) return x;
  if constexpr (GEN) return act_apply(x, act);
  else return fmax(x, 0.01 * x);     // (probes/mfma_issue_probe: 165 cycles per 36-instruction layer; multiply + compare + select: 250 - 300)
}
template <int H_, int D_, int NST_ = 256>
struct Geo {
  static constexpr int H = H_, D = D_, NH = 6;
  static constexpr int NST = NST_;                          // threads that stage the weight images (256: a 64-row workgroup; 128: the 16-row kernels)
  static constexpr int NT = (H + 15) / 16, HP = 16 * NT;   // hidden tiles
  static constexpr int KSH = H / 4;                         // k-steps over a hidden layer's inputs
  static constexpr int KS0 = (D + 3) / 4;                   // k-steps over the MLP's inputs (2C features, zero padded to 4 KS0)
  static constexpr int S = stride2mod4(HP + 1);             // row stride of a weight image; column HP holds the bias
  static constexpr int WSIZE = HP * S;
  static constexpr int SR = NST == 256 ? 66 : 18;           // row stride of the [neuron][64 (16) rows] tiles of the backward (== 2 mod 4: conflict-free fragment reads)
  static constexpr int TSIZE = HP * SR;
  static constexpr int NREG = (HP * 2 * KSH + NST - 1) / NST;   // staging registers (PAIRS of doubles) per thread for a hidden image
  static_assert(H % 4 == 0 && D <= 16 && D <= H && H <= 96, "chain kernels: H = 12 .. 96 (multiple of 4; the backward: H <= 48), 2C <= 16");
  static constexpr size_t fwd_bytes() { return sizeof(double) * 2 * WSIZE; }
  static constexpr size_t bwd_bytes() { return sizeof(double) * (2 * WSIZE + 4 * TSIZE + (NST == 256 ? 2 * 1024 : 0)); }      // (+ the quarter sums of the two-role kernel)
  // offsets of (W_l, b_l) in a partial row / parameter block: concat_l (W_l, b_l)
  static constexpr int off_w(int l) { return l == 0 ? 0 : (H * D + H) + (l - 1) * (H * H + H); }
  static constexpr int hout(int l) { return l == NH ? D : H; }
  static constexpr int hin(int l) { return l == 0 ? D : H; }
  static constexpr int off_b(int l) { return off_w(l) + hout(l) * hin(l); }
  static constexpr int psize() { return off_b(NH) + D; }
};

// ---- weight images ---------------------------------------------------------------------------------
// Image of Linear l in LDS: W_l[o][k] at Wl[o * S + k], zero padded to RP x CP (what the matrix instructions read; beyond it a
// tile's unused rows / columns may hold anything: a row of A only reaches the same row of D), bias at Wl[o * S + HP].
template <class G, int L>
struct Img {
  static constexpr int HO = G::hout(L), HI = G::hin(L);
  static constexpr int RP = L == G::NH ? 16 : G::HP;                       // rows read: a full tile of outputs
  static constexpr int CP = L == 0 ? 4 * G::KS0 : 4 * G::KSH;              // columns read by the k-steps
  static constexpr int CP2 = CP / 2;                                       // ... in pairs: the images travel 16 bytes per lane
  static constexpr int NP = (RP * CP2 + G::NST - 1) / G::NST;              // (a lone wave pays ~40 cycles of issue per global load)
  static_assert(NP <= G::NREG && HI % 2 == 0, "staging registers; rows of an even number of doubles");
};
typedef double v2d __attribute__((ext_vector_type(2)));
template <class G>
struct WRegs {
  v2d v[G::NREG];
  double b;
};
// (a row of W_l holds an even number of doubles and every block of the parameter buffer starts on an even offset or not -- global
// loads only need 4-byte alignment; the LDS side, (o S + k) * 8 bytes with S and k even, is 16-byte aligned as ds_write_b128 requires)
// one register (a pair of doubles) of an image (j < NP) or its bias (j == NP)
template <class G, int L>
__device__ __forceinline__ void issue_piece(const MlpArgs<double>& a, WRegs<G>& wr, int tid, int j) {
  using I = Img<G, L>;
  if (j < I::NP) {
    const int e = tid + G::NST * j, o = e / I::CP2, k = 2 * (e - o * I::CP2);
    const bool ok = o < I::HO && k < I::HI;
    const v2d x = *reinterpret_cast<const v2d*>(a.w[L] + (ok ? o * I::HI + k : 0));   // clamped address + select: no branch around the load
    wr.v[j] = ok ? x : v2d{0.0, 0.0};
  } else if (j == I::NP) {
    const double bb = a.b[L][tid < I::HO ? tid : 0];
    wr.b = tid < I::HO ? bb : 0.0;
  }
}
template <class G, int L>
__device__ __forceinline__ void commit_piece(double* Wl, const WRegs<G>& wr, int tid, int j) {
  using I = Img<G, L>;
  if (j < I::NP) {
    const int e = tid + G::NST * j, o = e / I::CP2, k = 2 * (e - o * I::CP2);
    if (I::RP * I::CP2 % G::NST == 0 || e < I::RP * I::CP2) *reinterpret_cast<v2d*>(Wl + o * G::S + k) = wr.v[j];
  } else if (j == I::NP) {
    if (tid < I::RP) Wl[tid * G::S + G::HP] = wr.b;
  }
}
template <class G, int L>
__device__ __forceinline__ void issue_image(const MlpArgs<double>& a, WRegs<G>& wr, int tid) {
  using I = Img<G, L>;
#pragma unroll
  for (int i = 0; i < I::NP + 1; ++i) issue_piece<G, L>(a, wr, tid, i);
}
template <class G, int L>
__device__ __forceinline__ void commit_image(double* Wl, const WRegs<G>& wr, int tid) {
  using I = Img<G, L>;
#pragma unroll
  for (int i = 0; i < I::NP + 1; ++i) commit_piece<G, L>(Wl, wr, tid, i);
}
// Step q of a kernel reads the image of Linear seq(q) from buffer q & 1.  The images travel global -> registers -> LDS two steps
// ahead of their use (two register sets, images alternate between them): with one set, i.e. one layer of lead, every step waited
// for its successor's weights -- an L2 round trip under load is longer than the ~1 us a layer computes.
template <class G, bool BWD>
__host__ __device__ constexpr int seq(int q) { return BWD && q > G::NH ? 2 * G::NH - q : q; }
// a step's staging as pieces: 2 (NREG + 1) of them -- commit of the next step's image first (it frees the register set), then
// the request for the image two steps further on
template <class G>
constexpr int stage_pieces() { return 2 * (G::NREG + 1); }
template <class G, bool BWD, int Q>
__device__ __forceinline__ void stage_piece(const MlpArgs<double>& a, double* Wl, WRegs<G>& wrA, WRegs<G>& wrB, int tid, int j) {
  constexpr int LAST = BWD ? 2 * G::NH : G::NH, n1 = Q + 1, n3 = Q + 3;
  WRegs<G>& wr = (n1 & 1) ? wrB : wrA;
  if (j <= G::NREG) {
    if constexpr (n1 <= LAST) commit_piece<G, seq<G, BWD>(n1 <= LAST ? n1 : 0)>(Wl + (n1 & 1) * G::WSIZE, wr, tid, j);
  } else {
    if constexpr (n3 <= LAST) issue_piece<G, seq<G, BWD>(n3 <= LAST ? n3 : 0)>(a, wr, tid, j - G::NREG - 1);
  }
}
template <class G, bool BWD>
__device__ __forceinline__ void stage_prologue(const MlpArgs<double>& a, double* Wl, WRegs<G>& wrA, WRegs<G>& wrB, int tid) {
  issue_image<G, 0>(a, wrA, tid);
  issue_image<G, 1>(a, wrB, tid);
  commit_image<G, 0>(Wl, wrA, tid);
  issue_image<G, 2>(a, wrA, tid);
}

// ---- one layer of the chain ---------------------------------------------------------------------------
// hidden layer: hout^T = act(W hin^T + b).  KS = k-steps over the inputs (hin[u][r] <-> k = 16u + 4r + g).  Tile-major: tile t is
// complete after item (t + 1) KS - 1 and its activation runs two items later, under the next tile's matrix instructions; the
// bias of tile t + 1 is read while tile t computes.  ext(i) = the step's other work for item i.
template <class G, int KS, int NTI, bool GEN, int DBG = 0, class ExtF>
__device__ __forceinline__ void layer_fwd(const double* Wc, const v4d (&hin)[NTI], v4d (&hout)[G::NT], int c, int g, int act, ExtF ext) {
  constexpr int S = G::S, NT = G::NT;
  const double* wa = Wc + c * S + g;
  const double* wbias = Wc + g * S + G::HP;
  v4d acc[NT];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[0][r] = wbias[4 * r * S];
  if (DBG) STAMP(30);
  mfma_stream<KS * NT, LOOKAHEAD>(
      [&](int i) { return wa[16 * (i / KS) * S + 4 * (i % KS)]; },
      [&](int i, double av) {
        const int t = i / KS, ks = i % KS;
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, hin[ks >> 2][ks & 3], acc[t], 0, 0, 0);
      },
      [&](int i) {
        const int t = i / KS, ks = i % KS;
        if (ks == 0 && t + 1 < NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t + 1 < NT ? t + 1 : 0][r] = wbias[(16 * (t + 1) + 4 * r) * S];
        }
        constexpr int k1 = KS > 6 ? 5 : KS > 1 ? 1 : 0, k2 = KS > 6 ? 6 : KS > 2 ? 2 : k1;
#pragma unroll
        for (int r = 0; r < 4; ++r)                          // tile t - 1 finished a few items ago (its results have left the pipe)
          if (t > 0 && ks == (r < 2 ? k1 : k2)) {
            double y = act_t<GEN>(acc[t > 0 ? t - 1 : 0][r], act);
            pin(y);
            hout[t > 0 ? t - 1 : 0][r] = y;
          }
        ext(i);
        if (DBG && (i == 0 || i == 1 || i == 2 || i == 11 || i == 12 || i == 23 || i == 34 || i == 35)) STAMP(31 + i);
      });
  if (DBG) STAMP(28);
#pragma unroll
  for (int r = 0; r < 4; ++r) hout[NT - 1][r] = act_t<GEN>(acc[NT - 1][r], act);
  if (DBG) { double z_ = hout[NT - 1][3]; pin(z_); hout[NT - 1][3] = z_; STAMP(29); }
}
// output layer (one tile of 2C <= 16 neurons, no activation)
template <class G>
__device__ __forceinline__ v4d layer_out(const double* Wc, const v4d (&hin)[G::NT], int c, int g) {
  constexpr int S = G::S;
  const double* wa = Wc + c * S + g;
  v4d acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = Wc[(4 * r + g) * S + G::HP];
  mfma_stream<G::KSH, LOOKAHEAD>([&](int ks) { return wa[4 * ks]; },
                                 [&](int ks, double av) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, hin[ks >> 2][ks & 3], acc, 0, 0, 0); },
                                 [&](int) {});
  return acc;
}
// backward through a Linear: gin^T [NTO tiles of inputs] = W^T gpre^T.  KS = k-steps over the layer's OUTPUT neurons
// (gpre[t][r] <-> o = 16t + 4r + g); A fragment (i = input 16u + c, k = o): the image read transposed.  Tile-major over the
// input tiles u; fin(u, r) consumes gin[u][r] of a finished tile (activation slope) two items after the tile's last matrix
// instruction -- except the last tile, which the caller finishes under whatever it issues next.
template <class G, int KS, int NTO, int NTG, class FinF, class ExtF>
__device__ __forceinline__ void layer_bwd(const double* Wc, const v4d (&gpre)[NTG], v4d (&gin)[NTO], int c, int g, FinF fin, ExtF ext) {
  constexpr int S = G::S;
  const double* wa = Wc + g * S + c;
#pragma unroll
  for (int u = 0; u < NTO; ++u) gin[u] = v4d{0, 0, 0, 0};
  mfma_stream<KS * NTO, LOOKAHEAD>(
      [&](int i) { return wa[4 * (i % KS) * S + 16 * (i / KS)]; },
      [&](int i, double av) {
        const int u = i / KS, ks = i % KS;
        gin[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, gpre[ks >> 2][ks & 3], gin[u], 0, 0, 0);
      },
      [&](int i) {
        const int u = i / KS, ks = i % KS;
        constexpr int k1 = KS > 6 ? 5 : KS > 1 ? 1 : 0, k2 = KS > 6 ? 6 : KS > 2 ? 2 : k1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (u > 0 && ks == (r < 2 ? k1 : k2)) fin(u > 0 ? u - 1 : 0, r);
        ext(i);
      });
}

// Hidden activations kept for the backward (MlpArgs::h_saved): an opaque register image,
// [64-row block][wave][layer][tile][lane][register] -- a lane's four registers of a tile are 32 contiguous bytes, moved as two
// 16-byte pieces, a wave's piece = every other 16 bytes of a 2 KB run -- of exactly mlp_saved_doubles(M, H, 7) doubles.  Written and
// read by these kernels only (same H, same lane mapping).  The forward stores layer q - 1 under layer q's matrix instructions; the
// backward requests layer l - 1 TWO steps before the step that needs it (as it does for the weight images): loaded all at once in
// the prologue, the 35 MB of a 512-jet batch were a burst of ~6 us in front of every backward -- what the six recomputed layers cost.
template <class G>
__device__ __forceinline__ double* saved_ptr(const MlpArgs<double>& a, int wave, int lane) {
  return a.h_saved + ((size_t)blockIdx.x * 4 + wave) * (G::NH * G::NT * 256) + lane * 4;
}
template <class G>
constexpr int saved_pieces() { return 2 * G::NT; }
// piece j of layer l: tile j >> 1, registers 2 (j & 1) and 2 (j & 1) + 1
template <class G, int NTI>
__device__ __forceinline__ void save_piece(double* hs, int l, const v4d (&h)[NTI], int j) {
  const v2d v = {h[j >> 1][2 * (j & 1)], h[j >> 1][2 * (j & 1) + 1]};
  __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(hs + (l * G::NT + (j >> 1)) * 256 + 2 * (j & 1)));
}
template <class G>
__device__ __forceinline__ void load_piece(const double* hs, int l, v4d (&h)[G::NT], int j) {
  const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(hs + (l * G::NT + (j >> 1)) * 256 + 2 * (j & 1)));
  h[j >> 1][2 * (j & 1)] = v[0];
  h[j >> 1][2 * (j & 1) + 1] = v[1];
}

// MLP input rows in B layout: xb[r] = x[row0 + c][feature 4r + g], feature f = 2 ch + z of the planar [2][M][C] scalars
template <class G>
__device__ __forceinline__ void load_x(const MlpArgs<double>& a, int row, int g, v4d (&xb)[1]) {
  const bool rok = row < a.M;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 4 * r + g;
    const bool ok = r < G::KS0 && rok && f < G::D;
    const double x = a.s_in[ok ? (size_t)(f & 1) * a.M * a.C + (size_t)row * a.C + (f >> 1) : 0];
    xb[0][r] = ok ? x : 0.0;
  }
}

// TWO: eight waves per workgroup -- waves 4 - 7 stage the weight images (global -> registers -> LDS, two steps ahead) and nothing else,
// waves 0 - 3 carry the chain without the staging pieces in their matrix streams (round 6; see mlp_chain_bwd2_kernel)
template <int H, int D, bool GEN, bool SAVE, bool TWO = false>
__global__ __launch_bounds__(TWO ? 512 : 256) void mlp_chain_fwd_kernel(MlpArgs<double> a) {
  using G = Geo<H, D>;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int row = blockIdx.x * 64 + wave * 16 + c;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  WRegs<G> wrA, wrB;
  STAMP(0);
  STAMP_LIFE_BEGIN();
  if constexpr (TWO) {
    if (wave >= 4) {
      const int st = tid - 256;
      stage_prologue<G, false>(a, Wl, wrA, wrB, st);
      lds_barrier();
#define LGN_STAGE_STEP(Q)                                                                                            \
  _Pragma("unroll") for (int j = 0; j < stage_pieces<G>(); ++j) stage_piece<G, false, Q>(a, Wl, wrA, wrB, st, j);    \
  lds_barrier();
      LGN_STAGE_STEP(0) LGN_STAGE_STEP(1) LGN_STAGE_STEP(2) LGN_STAGE_STEP(3) LGN_STAGE_STEP(4) LGN_STAGE_STEP(5)
#undef LGN_STAGE_STEP
      return;
    }
  }
  v4d xb[1];
  load_x<G>(a, row, g, xb);
  if constexpr (!TWO) stage_prologue<G, false>(a, Wl, wrA, wrB, tid);
  lds_barrier();
  STAMP(1);
  v4d h0[G::NT], h1[G::NT];
  double* hs = SAVE ? saved_ptr<G>(a, wave, lane) : nullptr;
  // step q: commit the image of layer q + 1 (loaded during step q - 1), request layer q + 2, compute layer q
  // (SAVE: the activations of layer q - 1 are stored under layer q's matrix instructions)
#define LGN_CHAIN_STEP(Q, HIN, HOUT, KS, NTI)                                                                        \
  layer_fwd<G, KS, NTI, GEN, (Q == 3)>(Wl + (Q & 1) * G::WSIZE, HIN, HOUT, c, g, a.act, [&](int i) {               \
    if (!TWO && !LGN_DBG_NOSTAGE) deal<KS * G::NT, stage_pieces<G>()>(i, [&](int j) { stage_piece<G, false, Q>(a, Wl, wrA, wrB, tid, j); });      \
    if (SAVE && Q > 0) deal<KS * G::NT, 2 * NTI>(i, [&](int j) { save_piece<G>(hs, Q - 1, HIN, j); });              \
  });                                                                                                                \
  lds_barrier();                                                                                                     \
  STAMP(2 + Q);
  LGN_CHAIN_STEP(0, xb, h0, G::KS0, 1)
  LGN_CHAIN_STEP(1, h0, h1, G::KSH, G::NT)
  LGN_CHAIN_STEP(2, h1, h0, G::KSH, G::NT)
  LGN_CHAIN_STEP(3, h0, h1, G::KSH, G::NT)
  LGN_CHAIN_STEP(4, h1, h0, G::KSH, G::NT)
  LGN_CHAIN_STEP(5, h0, h1, G::KSH, G::NT)
#undef LGN_CHAIN_STEP
  if (SAVE) {
#pragma unroll
    for (int j = 0; j < saved_pieces<G>(); ++j) save_piece<G>(hs, G::NH - 1, h1, j);
  }
  const v4d y = layer_out<G>(Wl + (G::NH & 1) * G::WSIZE, h1, c, g);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * r + g;
    if (o < D && row < a.M) a.s_out[mlp_out_index(a, o & 1, row, o >> 1)] = y[r];
  }
  STAMP(8);
  STAMP_LIFE_END();
}

// one register (piece j = 4t + r) of this wave's columns 16 w .. 16 w + 15 of a [neuron][64 rows] tile: D layout, neuron
// 16t + 4r + g, row c
template <class G, int NTI>
__device__ __forceinline__ void publish_piece(double* T, const v4d (&v)[NTI], int wave, int c, int g, int j) {
  T[(4 * j + g) * G::SR + 16 * wave + c] = v[j >> 2][j & 3];
}
// dW tiles (t, u0 .. u0 + NU - 1) of Linear L over the workgroup's 64 rows, and (u0 == 0) the bias gradient of tile row t.
// Tile-major: the 16 A fragments (g_pre^T, shared by the NU tiles) are read during the first tile and kept; a finished tile is
// stored under the next tile's matrix instructions.  side(i): the caller's work for item i of 16 NU.
template <class G, int L, int NU, class SideF>
__device__ __forceinline__ void dw_unit(const double* Gt, const double* Xt, double* part, int t, int u0, int c, int g, SideF side) {
  constexpr int SR = G::SR, HO = G::hout(L), HI = G::hin(L), N = 16 * NU, P = 6;
  const double* ga = Gt + (16 * t + c) * SR + g;
  const double* xb = Xt + (16 * u0 + c) * SR + g;
  double* pW = part + G::off_w(L);
  v4d acc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[u] = v4d{0, 0, 0, 0};
  double dbs = 0.0, qa[16], qx[N];
  auto load = [&](int i) {
    qx[i] = xb[16 * (i >> 4) * SR + 4 * (i & 15)];
    if (i < 16) qa[i] = ga[4 * i];
  };
  auto store = [&](int u, int r) {
    const int o = 16 * t + 4 * r + g, k = 16 * (u0 + u) + c;
    if (o < HO && k < HI) __builtin_nontemporal_store(acc[u][r], &pW[o * HI + k]);
  };
#pragma unroll
  for (int i = 0; i < P; ++i) load(i);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int u = i >> 4, sk = i & 15;
    if (i + P < N) load(i + P);
    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[sk], qx[i], acc[u], 0, 0, 0);
    if (u > 0 && sk >= 1 && sk <= 4) store(u > 0 ? u - 1 : 0, sk - 1);
    side(i);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) store(NU - 1, r);
  if (u0 == 0) {
    // the bias gradient: row sums of g_pre^T = the sum of the 16 A fragments, which are still in registers -- added up HERE, in one
    // burst (an fp64 vector instruction between two matrix instructions costs ~19 cycles of the shared datapath for the first of
    // a burst and 4 for each further one: probes/mfma_issue_probe)
    double d4[4] = {0.0, 0.0, 0.0, 0.0};                     // (four chains: the adds are 8-cycle-latency instructions)
#pragma unroll
    for (int sk = 0; sk < 16; ++sk) d4[sk & 3] += qa[sk];
    dbs = (d4[0] + d4[1]) + (d4[2] + d4[3]);
    dbs += shfl_xor(dbs, 16);
    dbs += shfl_xor(dbs, 32);
    if (g == 0 && 16 * t + c < HO) __builtin_nontemporal_store(dbs, &part[G::off_b(L) + 16 * t + c]);
  }
}
// the weight gradient of Linear L from the tiles published one step earlier: hidden layers deal their tile ROWS to the waves
// (three tiles that share the A fragment; the owner rotates with the layer), the two end layers their single row / column of
// tiles.  Returns whether this wave had a unit (side ran); a wave without one runs side's work itself.
template <class G, int L, class SideF>
__device__ __forceinline__ bool dw_layer(const double* Gt, const double* Xt, double* part, int wave, int c, int g, SideF side) {
  if (L == G::NH) {
    if (wave < G::NT) { dw_unit<G, L, 1>(Gt, Xt, part, 0, wave, c, g, side); return true; }
  } else if (L == 0) {
    if (wave < G::NT) { dw_unit<G, L, 1>(Gt, Xt, part, wave, 0, c, g, side); return true; }
  } else {
    const int t = (wave + L) & 3;
    if (t < G::NT) { dw_unit<G, L, G::NT>(Gt, Xt, part, t, 0, c, g, side); return true; }
  }
  return false;
}

template <int H, int D, bool GEN, bool SAVE>
__global__ __launch_bounds__(256) void mlp_chain_bwd_kernel(MlpArgs<double> a) {
  using G = Geo<H, D>;
  constexpr int NT = G::NT, NH = G::NH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int row = blockIdx.x * 64 + wave * 16 + c;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  double* Gt = Wl + 2 * G::WSIZE;                            // 2 g_pre tiles [neuron][64 rows]
  double* Xt = Gt + 2 * G::TSIZE;                            // 2 layer-input tiles
  double* part = a.part + (size_t)blockIdx.x * a.psize;
  WRegs<G> wrA, wrB;
  STAMP(10);
  v4d xb[1];
  load_x<G>(a, row, g, xb);
  v4d gout[1];                                               // upstream gradient, D layout of the output tile (o = 4r + g, row c)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * r + g;
    const bool ok = o < D && row < a.M;
    const double x = a.g_out[ok ? mlp_out_index(a, o & 1, row, o >> 1) : 0];
    gout[0][r] = ok ? x : 0.0;
  }
  v4d h[NH][NT];
  const double* hs = SAVE ? saved_ptr<G>(a, wave, lane) : nullptr;
  if constexpr (SAVE) {
    // the forward's copy: the six recompute steps go away; the images start at the output layer (steps NH, NH + 1, ...: the same
    // buffers and register sets as in the recompute form)
    issue_image<G, NH>(a, wrA, tid);
    issue_image<G, NH - 1>(a, wrB, tid);
#pragma unroll
    for (int j = 0; j < saved_pieces<G>(); ++j) load_piece<G>(hs, NH - 1, h[NH - 1], j);      // needed by the first two steps; the
#pragma unroll
    for (int j = 0; j < saved_pieces<G>(); ++j) load_piece<G>(hs, NH - 2, h[NH - 2], j);      // others follow two steps ahead of their use
    commit_image<G, NH>(Wl, wrA, tid);
    issue_image<G, NH - 2>(a, wrA, tid);
    lds_barrier();
    STAMP(11);
  } else {
    stage_prologue<G, true>(a, Wl, wrA, wrB, tid);
    lds_barrier();
    STAMP(11);
    // ---- recompute the hidden activations: steps q = 0 .. 5, image q in buffer q & 1
#define LGN_CHAIN_STEP(Q, HIN, KS, NTI)                                                                              \
  layer_fwd<G, KS, NTI, GEN>(Wl + (Q & 1) * G::WSIZE, HIN, h[Q], c, g, a.act, [&](int i) {                          \
    deal<KS * NT, stage_pieces<G>()>(i, [&](int j) { stage_piece<G, true, Q>(a, Wl, wrA, wrB, tid, j); });          \
  });                                                                                                                \
  lds_barrier();                                                                                                     \
  STAMP(12 + Q);
    LGN_CHAIN_STEP(0, xb, G::KS0, 1)
    LGN_CHAIN_STEP(1, h[0], G::KSH, NT)
    LGN_CHAIN_STEP(2, h[1], G::KSH, NT)
    LGN_CHAIN_STEP(3, h[2], G::KSH, NT)
    LGN_CHAIN_STEP(4, h[3], G::KSH, NT)
    LGN_CHAIN_STEP(5, h[4], G::KSH, NT)
#undef LGN_CHAIN_STEP
  }
  // ---- backward sweep: step q = 6 .. 12 handles Linear l = 12 - q (image in buffer q & 1), tiles of parity l & 1.  The g_pre of
  // successive layers alternate between two register arrays: the running stream still reads the old one as its B operands.
  v4d gpA[NT], gpB[NT], gin[NT];
  {  // q = 6, l = 6: the output layer
    constexpr int NI = G::KS0 * NT, NSV = SAVE && NH >= 3 ? saved_pieces<G>() : 0, NPC = 4 + 4 * NT + stage_pieces<G>() + NSV;
    auto fin = [&](int u, int r) {
      double y = gin[u][r] * act_slope_t<GEN>(h[NH - 1][u][r], a.act);
      pin(y);
      gpA[u][r] = y;
    };
    layer_bwd<G, G::KS0, NT, 1>(Wl + 0 * G::WSIZE, gout, gin, c, g, fin, [&](int i) {
      deal<NI, NPC>(i, [&](int j) {
        if (j < 4) publish_piece<G, 1>(Gt, gout, wave, c, g, j);
        else if (j < 4 + 4 * NT) publish_piece<G, NT>(Xt, h[NH - 1], wave, c, g, j - 4);
        else if (j < 4 + 4 * NT + stage_pieces<G>()) stage_piece<G, true, NH>(a, Wl, wrA, wrB, tid, j - 4 - 4 * NT);
        else if constexpr (NSV > 0) load_piece<G>(hs, NH - 3, h[NH >= 3 ? NH - 3 : 0], j - 4 - 4 * NT - stage_pieces<G>());
      });
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) fin(NT - 1, r);
    lds_barrier();
    STAMP(18);
  }
#define LGN_CHAIN_BSTEP(L, GP, GN)                                                                                   \
  {                                                                                                                  \
    constexpr int q_ = 2 * NH - (L), NI = G::KSH * NT, NSV = SAVE && (L) >= 3 ? saved_pieces<G>() : 0,               \
                  NPC = 8 * NT + stage_pieces<G>() + NSV;                                                            \
    auto fin = [&](int u, int r) {                                                                                   \
      double y = gin[u][r] * act_slope_t<GEN>(h[(L) - 1][u][r], a.act);                                              \
      pin(y);                                                                                                        \
      GN[u][r] = y;                                                                                                  \
    };                                                                                                               \
    layer_bwd<G, G::KSH, NT, NT>(Wl + (q_ & 1) * G::WSIZE, GP, gin, c, g, fin, [&](int i) {                          \
      deal<NI, NPC>(i, [&](int j) {                                                                                  \
        if (j < 4 * NT) publish_piece<G, NT>(Gt + ((L) & 1) * G::TSIZE, GP, wave, c, g, j);                          \
        else if (j < 8 * NT) publish_piece<G, NT>(Xt + ((L) & 1) * G::TSIZE, h[(L) - 1], wave, c, g, j - 4 * NT);    \
        else if (j < 8 * NT + stage_pieces<G>()) stage_piece<G, true, q_>(a, Wl, wrA, wrB, tid, j - 8 * NT);         \
        else if constexpr (NSV > 0) load_piece<G>(hs, (L) - 3, h[(L) >= 3 ? (L) - 3 : 0], j - 8 * NT - stage_pieces<G>()); \
      });                                                                                                            \
    });                                                                                                              \
    auto tail = [&](int i) {                                                                                         \
      if (i == 1) { fin(NT - 1, 0); fin(NT - 1, 1); }                                                                \
      if (i == 2) { fin(NT - 1, 2); fin(NT - 1, 3); }                                                                \
    };                                                                                                               \
    if (!dw_layer<G, (L) + 1>(Gt + (((L) + 1) & 1) * G::TSIZE, Xt + (((L) + 1) & 1) * G::TSIZE, part, wave, c, g, tail)) {   \
      tail(1);                                                                                                       \
      tail(2);                                                                                                       \
    }                                                                                                                \
    lds_barrier();                                                                                                   \
    STAMP(12 + q_);                                                                                                  \
  }
  LGN_CHAIN_BSTEP(5, gpA, gpB)
  LGN_CHAIN_BSTEP(4, gpB, gpA)
  LGN_CHAIN_BSTEP(3, gpA, gpB)
  LGN_CHAIN_BSTEP(2, gpB, gpA)
  LGN_CHAIN_BSTEP(1, gpA, gpB)
#undef LGN_CHAIN_BSTEP
  {  // q = 12, l = 0: the first layer; its input tile holds the MLP's input rows (k-steps beyond KS0 never stored or read)
    v4d gx[1];
    layer_bwd<G, G::KSH, 1, NT>(Wl + 0 * G::WSIZE, gpB, gx, c, g, [&](int, int) {}, [&](int i) {
      deal<G::KSH, 4 * NT + 4>(i, [&](int j) {
        if (j < 4 * NT) publish_piece<G, NT>(Gt, gpB, wave, c, g, j);
        else publish_piece<G, 1>(Xt, xb, wave, c, g, j - 4 * NT);
      });
    });
    auto tail = [&](int i) {                                 // D layout of the input tile: feature f = 4r + g, row c
      if (i >= 1 && i <= 4) {
        const int f = 4 * (i - 1) + g;
        if (f < D && row < a.M) a.g_in[mlp_out_index(a, f & 1, row, f >> 1)] = gx[0][i - 1];
      }
    };
    if (!dw_layer<G, 1>(Gt + G::TSIZE, Xt + G::TSIZE, part, wave, c, g, tail)) {
#pragma unroll
      for (int i = 1; i <= 4; ++i) tail(i);
    }
    lds_barrier();
    STAMP(25);
  }
  dw_layer<G, 0>(Gt, Xt, part, wave, c, g, [&](int) {});
  STAMP(26);
}

// Weight gradient of a HIDDEN Linear L (NT = 3: nine 16 x 16 tiles over the workgroup's 64 rows = 144 matrix instructions) dealt evenly
// to the four dW waves of the two-role kernel: wave w takes tiles 2w and 2w + 1 (row-major) whole and k-steps 4w .. 4w + 3 of the
// ninth tile (2, 2) -- 36 instructions each, where whole tile rows gave 48 / 48 / 48 / 0.  The four quarter sums of the ninth tile
// meet through LDS (qs: [wave][register][lane]); dw_quarter_sum adds them in a fixed order one step later.
template <class G, int L, class SideF>
__device__ __forceinline__ void dw_hidden3(const double* Gt, const double* Xt, double* part, double* qs, int w, int lane, int c, int g, SideF side) {
  static_assert(G::NT == 3 && L >= 1 && L < G::NH, "hidden layers of three tiles");
  constexpr int SR = G::SR, H = G::H, N = 36, P = 6;
  const int i0 = 2 * w, i1 = 2 * w + 1;
  const int t0 = i0 / 3, u0 = i0 - 3 * t0, t1 = i1 / 3, u1 = i1 - 3 * t1;
  const double* ga0 = Gt + (16 * t0 + c) * SR + g;
  const double* xb0 = Xt + (16 * u0 + c) * SR + g;
  const double* ga1 = Gt + (16 * t1 + c) * SR + g;
  const double* xb1 = Xt + (16 * u1 + c) * SR + g;
  const double* ga2 = Gt + (32 + c) * SR + g + 16 * w;
  const double* xb2 = Xt + (32 + c) * SR + g + 16 * w;
  double* pW = part + G::off_w(L);
  v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  double qa[N], qx[N];
  auto load = [&](int i) {
    if (i < 16) { qa[i] = ga0[4 * i]; qx[i] = xb0[4 * i]; }
    else if (i < 32) { qa[i] = ga1[4 * (i - 16)]; qx[i] = xb1[4 * (i - 16)]; }
    else { qa[i] = ga2[4 * (i - 32)]; qx[i] = xb2[4 * (i - 32)]; }
  };
  auto store = [&](const v4d& acc, int t, int u, int r) {
    const int o = 16 * t + 4 * r + g, k = 16 * u + c;
    if (o < H && k < H) __builtin_nontemporal_store(acc[r], &pW[o * H + k]);
  };
#pragma unroll
  for (int i = 0; i < P; ++i) load(i);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i + P < N) load(i + P);
    if (i < 16) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[i], qx[i], acc0, 0, 0, 0);
    else if (i < 32) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[i], qx[i], acc1, 0, 0, 0);
    else acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[i], qx[i], acc2, 0, 0, 0);
    if (i >= 17 && i <= 20) store(acc0, t0, u0, i - 17);
    if (i >= 33) store(acc1, t1, u1, i - 33);
    side(i);
    __builtin_amdgcn_sched_barrier(0);
  }
  store(acc1, t1, u1, 3);
#pragma unroll
  for (int r = 0; r < 4; ++r) qs[(w * 4 + r) * 64 + lane] = acc2[r];
  if (w != 1) {      // bias gradient of tile row t0 (rows 0, 1, 2 for waves 0, 2, 3): the sum of the sixteen A fragments of the first tile
    double d4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sk = 0; sk < 16; ++sk) d4[sk & 3] += qa[sk];
    double dbs = (d4[0] + d4[1]) + (d4[2] + d4[3]);
    dbs += shfl_xor(dbs, 16);
    dbs += shfl_xor(dbs, 32);
    if (g == 0 && 16 * t0 + c < H) __builtin_nontemporal_store(dbs, &part[G::off_b(L) + 16 * t0 + c]);
  }
}
// the ninth tile of hidden Linear L from the four quarter sums (one wave; the quarters were written before the last barrier)
template <class G, int L>
__device__ __forceinline__ void dw_quarter_sum(const double* qs, double* part, int lane, int c, int g) {
  constexpr int H = G::H;
  double* pW = part + G::off_w(L);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double v = (qs[(0 * 4 + r) * 64 + lane] + qs[(1 * 4 + r) * 64 + lane]) + (qs[(2 * 4 + r) * 64 + lane] + qs[(3 * 4 + r) * 64 + lane]);
    const int o = 32 + 4 * r + g, k = 32 + c;
    if (o < H && k < H) __builtin_nontemporal_store(v, &pW[o * H + k]);
  }
}

// ---- backward, two roles per workgroup (round 6) ---------------------------------------------------------------------------------
// The kernel above runs ONE wave per SIMD: whatever a wave waits for -- an LDS fragment, the barrier, the datapath's turn-around
// after an fp64 vector instruction -- leaves the matrix pipe idle (0.40 busy at cfg2).  Here a workgroup is EIGHT waves over the same
// 64 rows: waves 0 - 3 carry the chain (recompute, g_in = W^T g_pre, tile publication) and nothing else; waves 4 - 7 take what the
// chain waves did on the side -- the weight gradients dW_l from the published tiles (one step behind the chain, as before) and the
// staging of the weight images.  Every SIMD then holds two waves with independent instruction streams and the pipe takes whichever is
// ready.  Same arithmetic, same summation order, same barriers (one per step, over all eight waves); 256 registers per wave.
template <int H, int D, bool GEN>
__global__ __launch_bounds__(512) void mlp_chain_bwd2_kernel(MlpArgs<double> a) {
  using G = Geo<H, D>;
  constexpr int NT = G::NT, NH = G::NH;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int w4 = wave & 3;                                   // chain waves: the 16-row tile; dW waves: the owner index of dw_layer
  const int c = lane & 15, g = lane >> 4;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  double* Gt = Wl + 2 * G::WSIZE;                            // 2 g_pre tiles [neuron][64 rows]
  double* Xt = Gt + 2 * G::TSIZE;                            // 2 layer-input tiles
  STAMP(10);
  if (wave >= 4) {
    // ================= weight gradients + image staging =================
    // (the SIMD arbiter serves the OLDER wave first: the chain wave of the SIMD ran its 36 matrix instructions in ~4 100 cycles and the dW
    // wave finished alone 2 000 later; with the dW waves at a higher priority both end within ~700 cycles of each other and a
    // backward step is 150 - 450 cycles shorter -- stamps 44.., 50.., 56.. of the debug build)
    __builtin_amdgcn_s_setprio(2);
    const int tid = (int)threadIdx.x - 256;
    double* part = a.part + (size_t)blockIdx.x * a.psize;
    WRegs<G> wrA, wrB;
    stage_prologue<G, true>(a, Wl, wrA, wrB, tid);
    lds_barrier();
#define LGN_STAGE_ALL(Q)                                                                                             \
  _Pragma("unroll") for (int j = 0; j < stage_pieces<G>(); ++j) stage_piece<G, true, Q>(a, Wl, wrA, wrB, tid, j);
#define LGN_STAGE_STEP(Q) LGN_STAGE_ALL(Q) lds_barrier();
    LGN_STAGE_STEP(0) LGN_STAGE_STEP(1) LGN_STAGE_STEP(2) LGN_STAGE_STEP(3) LGN_STAGE_STEP(4) LGN_STAGE_STEP(5)
    LGN_STAGE_STEP(6)                                        // the output layer's step: nothing published yet
    // step q = 2 NH - L: weight gradient of Linear L + 1 from the tiles of parity (L + 1) & 1, staging dealt under its matrix stream
    // (NT = 3: the hidden layers' nine tiles dealt evenly, dw_hidden3; the ninth tile's quarter sums of Linear L + 2 are added up by
    // wave 1 at the head of the step that follows their barrier)
    constexpr bool EVEN = NT == 3;
    double* qs = Xt + 2 * G::TSIZE;                          // [2 parities][4 waves][4 registers][64 lanes]
#define LGN_DW_STEP(L)                                                                                               \
  {                                                                                                                  \
    constexpr int q_ = 2 * NH - (L), NS = stage_pieces<G>(), L1 = (L) + 1;                                           \
    constexpr bool HID = EVEN && L1 >= 1 && L1 < NH;                                                                 \
    constexpr int NI = HID ? 36 : (L1 == NH || L1 == 0) ? 16 : 16 * NT;                                              \
    auto side = [&](int i) { deal<NI, NS>(i, [&](int j) { stage_piece<G, true, q_>(a, Wl, wrA, wrB, tid, j); }); };  \
    if constexpr (EVEN && L1 + 1 >= 1 && L1 + 1 < NH) {                                                              \
      if (w4 == 1) dw_quarter_sum<G, (L1 + 1 < NH ? L1 + 1 : 1)>(qs + ((L1 + 1) & 1) * 1024, part, lane, c, g);     \
    }                                                                                                                \
    if constexpr (HID) {                                                                                             \
      dw_hidden3<G, (HID ? L1 : 1)>(Gt + (L1 & 1) * G::TSIZE, Xt + (L1 & 1) * G::TSIZE, part, qs + (L1 & 1) * 1024, w4, lane, c, g, side); \
    } else if (!dw_layer<G, L1>(Gt + (L1 & 1) * G::TSIZE, Xt + (L1 & 1) * G::TSIZE, part, w4, c, g, side)) {         \
      LGN_STAGE_ALL(q_)                                                                                              \
    }                                                                                                                \
    DSTAMP(44 + (L));                                                                                                \
    lds_barrier();                                                                                                   \
  }
    LGN_DW_STEP(5) LGN_DW_STEP(4) LGN_DW_STEP(3) LGN_DW_STEP(2) LGN_DW_STEP(1) LGN_DW_STEP(0)
#undef LGN_DW_STEP
#undef LGN_STAGE_STEP
#undef LGN_STAGE_ALL
    if constexpr (EVEN && NH >= 2) {
      if (w4 == 1) dw_quarter_sum<G, 1>(qs + 1024, part, lane, c, g);      // (Linear 1: its quarters were written in the last step)
    }
    dw_layer<G, 0>(Gt, Xt, part, w4, c, g, [&](int) {});
    return;
  }
  // ================= the chain =================
  const int row = blockIdx.x * 64 + wave * 16 + c;
  v4d xb[1];
  load_x<G>(a, row, g, xb);
  v4d gout[1];                                               // upstream gradient, D layout of the output tile (o = 4r + g, row c)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * r + g;
    const bool ok = o < D && row < a.M;
    const double x = a.g_out[ok ? mlp_out_index(a, o & 1, row, o >> 1) : 0];
    gout[0][r] = ok ? x : 0.0;
  }
  v4d h[NH][NT];
  lds_barrier();
  STAMP(11);
#define LGN_CHAIN_STEP(Q, HIN, KS, NTI)                                                                              \
  layer_fwd<G, KS, NTI, GEN>(Wl + (Q & 1) * G::WSIZE, HIN, h[Q], c, g, a.act, [&](int) {});                          \
  lds_barrier();                                                                                                     \
  STAMP(12 + Q);
  LGN_CHAIN_STEP(0, xb, G::KS0, 1)
  LGN_CHAIN_STEP(1, h[0], G::KSH, NT)
  LGN_CHAIN_STEP(2, h[1], G::KSH, NT)
  LGN_CHAIN_STEP(3, h[2], G::KSH, NT)
  LGN_CHAIN_STEP(4, h[3], G::KSH, NT)
  LGN_CHAIN_STEP(5, h[4], G::KSH, NT)
#undef LGN_CHAIN_STEP
  v4d gpA[NT], gpB[NT], gin[NT];
  {  // q = 6, l = 6: the output layer
    constexpr int NI = G::KS0 * NT, NPC = 4 + 4 * NT;
    auto fin = [&](int u, int r) {
      double y = gin[u][r] * act_slope_t<GEN>(h[NH - 1][u][r], a.act);
      pin(y);
      gpA[u][r] = y;
    };
    layer_bwd<G, G::KS0, NT, 1>(Wl + 0 * G::WSIZE, gout, gin, c, g, fin, [&](int i) {
      deal<NI, NPC>(i, [&](int j) {
        if (j < 4) publish_piece<G, 1>(Gt, gout, wave, c, g, j);
        else publish_piece<G, NT>(Xt, h[NH - 1], wave, c, g, j - 4);
      });
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) fin(NT - 1, r);
    lds_barrier();
    STAMP(18);
  }
#define LGN_CHAIN_BSTEP(L, GP, GN)                                                                                   \
  {                                                                                                                  \
    constexpr int q_ = 2 * NH - (L), NI = G::KSH * NT, NPC = 8 * NT;                                                 \
    auto fin = [&](int u, int r) {                                                                                   \
      double y = gin[u][r] * act_slope_t<GEN>(h[(L) - 1][u][r], a.act);                                              \
      pin(y);                                                                                                        \
      GN[u][r] = y;                                                                                                  \
    };                                                                                                               \
    layer_bwd<G, G::KSH, NT, NT>(Wl + (q_ & 1) * G::WSIZE, GP, gin, c, g, fin, [&](int i) {                          \
      deal<NI, NPC>(i, [&](int j) {                                                                                  \
        if (j < 4 * NT) publish_piece<G, NT>(Gt + ((L) & 1) * G::TSIZE, GP, wave, c, g, j);                          \
        else publish_piece<G, NT>(Xt + ((L) & 1) * G::TSIZE, h[(L) - 1], wave, c, g, j - 4 * NT);                    \
      });                                                                                                            \
    });                                                                                                              \
    STAMP(56 + (L));                                                                                                 \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) fin(NT - 1, r);                                                    \
    STAMP(50 + (L));                                                                                                 \
    lds_barrier();                                                                                                   \
    STAMP(12 + q_);                                                                                                  \
  }
  LGN_CHAIN_BSTEP(5, gpA, gpB)
  LGN_CHAIN_BSTEP(4, gpB, gpA)
  LGN_CHAIN_BSTEP(3, gpA, gpB)
  LGN_CHAIN_BSTEP(2, gpB, gpA)
  LGN_CHAIN_BSTEP(1, gpA, gpB)
#undef LGN_CHAIN_BSTEP
  {  // q = 12, l = 0: the first layer; its input tile holds the MLP's input rows (k-steps beyond KS0 never stored or read)
    v4d gx[1];
    layer_bwd<G, G::KSH, 1, NT>(Wl + 0 * G::WSIZE, gpB, gx, c, g, [&](int, int) {}, [&](int i) {
      deal<G::KSH, 4 * NT + 4>(i, [&](int j) {
        if (j < 4 * NT) publish_piece<G, NT>(Gt, gpB, wave, c, g, j);
        else publish_piece<G, 1>(Xt, xb, wave, c, g, j - 4 * NT);
      });
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) {                            // D layout of the input tile: feature f = 4r + g, row c
      const int f = 4 * r + g;
      if (f < D && row < a.M) a.g_in[mlp_out_index(a, f & 1, row, f >> 1)] = gx[0][r];
    }
    lds_barrier();
    STAMP(25);
  }
}

// ---- small batches: 16-row workgroups (round 6) ---------------------------------------------------------------------------------
// Below 128 workgroups of 64 rows the CGMLP ran the 12-wave kernels of mlp_mfma.hip on 16-row workgroups: every layer split over
// three waves, activations exchanged through LDS, two barriers per layer -- 9 / 21 us per forward / backward at 64 jets for
// 0.05 GFLOP.  The chain form with roles needs no exchange: a workgroup is ONE chain wave (16 rows through all layers out of
// registers), one wave for the weight gradients (K = 16 rows: four k-steps per tile, 36 matrix instructions per hidden layer: what
// the chain wave computes meanwhile) and two waves staging the weight images -- each alone on its SIMD.  The forward keeps the
// activations for the backward where the step asks for it (MlpArgs::h_saved, <= 8 128 rows since the split kernels below; same register image as the 64-row
// kernels, one block per workgroup).
template <class G>
__device__ __forceinline__ double* saved_ptr16(const MlpArgs<double>& a, int lane) {
  return a.h_saved + (size_t)blockIdx.x * (G::NH * G::NT * 256) + lane * 4;
}
template <int H, int D, bool GEN, bool SAVE>
__global__ __launch_bounds__(192) void mlp_chain_fwd16_kernel(MlpArgs<double> a) {
  using G = Geo<H, D, 128>;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  if (wave >= 1) {
    const int st = tid - 64;
    WRegs<G> wrA, wrB;
    stage_prologue<G, false>(a, Wl, wrA, wrB, st);
    lds_barrier();
#define LGN_STAGE_STEP(Q)                                                                                            \
  _Pragma("unroll") for (int j = 0; j < stage_pieces<G>(); ++j) stage_piece<G, false, Q>(a, Wl, wrA, wrB, st, j);    \
  lds_barrier();
    LGN_STAGE_STEP(0) LGN_STAGE_STEP(1) LGN_STAGE_STEP(2) LGN_STAGE_STEP(3) LGN_STAGE_STEP(4) LGN_STAGE_STEP(5)
#undef LGN_STAGE_STEP
    return;
  }
  const int row = blockIdx.x * 16 + c;
  v4d xb[1];
  load_x<G>(a, row, g, xb);
  lds_barrier();
  v4d h0[G::NT], h1[G::NT];
  double* hs = SAVE ? saved_ptr16<G>(a, lane) : nullptr;
#define LGN_CHAIN_STEP(Q, HIN, HOUT, KS, NTI)                                                                        \
  layer_fwd<G, KS, NTI, GEN>(Wl + (Q & 1) * G::WSIZE, HIN, HOUT, c, g, a.act, [&](int i) {                          \
    if (SAVE && Q > 0) deal<KS * G::NT, 2 * NTI>(i, [&](int j) { save_piece<G>(hs, Q - 1, HIN, j); });              \
  });                                                                                                                \
  lds_barrier();
  LGN_CHAIN_STEP(0, xb, h0, G::KS0, 1)
  LGN_CHAIN_STEP(1, h0, h1, G::KSH, G::NT)
  LGN_CHAIN_STEP(2, h1, h0, G::KSH, G::NT)
  LGN_CHAIN_STEP(3, h0, h1, G::KSH, G::NT)
  LGN_CHAIN_STEP(4, h1, h0, G::KSH, G::NT)
  LGN_CHAIN_STEP(5, h0, h1, G::KSH, G::NT)
#undef LGN_CHAIN_STEP
  if (SAVE) {
#pragma unroll
    for (int j = 0; j < saved_pieces<G>(); ++j) save_piece<G>(hs, G::NH - 1, h1, j);
  }
  const v4d y = layer_out<G>(Wl + (G::NH & 1) * G::WSIZE, h1, c, g);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * r + g;
    if (o < D && row < a.M) a.s_out[mlp_out_index(a, o & 1, row, o >> 1)] = y[r];
  }
}

// weight gradient of Linear L over the workgroup's 16 rows (columns 0 .. 15 of the published tiles): every tile, four k-steps each
template <class G, int L>
__device__ __forceinline__ void dw16(const double* Gt, const double* Xt, double* part, int c, int g) {
  constexpr int SR = G::SR, HO = G::hout(L), HI = G::hin(L);
  constexpr int NTO = L == G::NH ? 1 : G::NT, NTI = L == 0 ? 1 : G::NT, NTL = NTO * NTI;
  double* pW = part + G::off_w(L);
  double qa[NTO][4], qx[NTI][4];
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) qa[t][sk] = Gt[(16 * t + c) * SR + g + 4 * sk];
#pragma unroll
  for (int u = 0; u < NTI; ++u)
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) qx[u][sk] = Xt[(16 * u + c) * SR + g + 4 * sk];
  __builtin_amdgcn_sched_barrier(0);
  v4d acc[NTL];
  auto store = [&](int tile, int r) {
    const int t = tile / NTI, u = tile - t * NTI, o = 16 * t + 4 * r + g, k = 16 * u + c;
    if (o < HO && k < HI) __builtin_nontemporal_store(acc[tile][r], &pW[o * HI + k]);
  };
#pragma unroll
  for (int i = 0; i < 4 * NTL; ++i) {
    const int tile = i >> 2, sk = i & 3, t = tile / NTI, u = tile - t * NTI;
    if (sk == 0) acc[tile] = v4d{0, 0, 0, 0};
    acc[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t][sk], qx[u][sk], acc[tile], 0, 0, 0);
    if (tile > 0) store(tile > 0 ? tile - 1 : 0, sk);       // the previous tile has left the pipe
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) store(NTL - 1, r);
#pragma unroll
  for (int t = 0; t < NTO; ++t) {                            // bias gradient: row sums of g_pre^T over the 16 rows
    double dbs = (qa[t][0] + qa[t][1]) + (qa[t][2] + qa[t][3]);
    dbs += shfl_xor(dbs, 16);
    dbs += shfl_xor(dbs, 32);
    if (g == 0 && 16 * t + c < HO) __builtin_nontemporal_store(dbs, &part[G::off_b(L) + 16 * t + c]);
  }
}

template <int H, int D, bool GEN, bool SAVE>
__global__ __launch_bounds__(256) void mlp_chain_bwd16_kernel(MlpArgs<double> a) {
  using G = Geo<H, D, 128>;
  constexpr int NT = G::NT, NH = G::NH;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  double* Gt = Wl + 2 * G::WSIZE;                            // 2 g_pre tiles [neuron][rows] (columns 0 .. 15 in use)
  double* Xt = Gt + 2 * G::TSIZE;                            // 2 layer-input tiles
  if (wave >= 2) {
    // ================= image staging =================
    const int st = (int)threadIdx.x - 128;
    WRegs<G> wrA, wrB;
#define LGN_STAGE_STEP(Q)                                                                                            \
  _Pragma("unroll") for (int j = 0; j < stage_pieces<G>(); ++j) stage_piece<G, true, Q>(a, Wl, wrA, wrB, st, j);     \
  lds_barrier();
    if constexpr (SAVE) {      // the images start at the output layer (steps NH, NH + 1, ...: the same buffers and register sets)
      issue_image<G, NH>(a, wrA, st);
      issue_image<G, NH - 1>(a, wrB, st);
      commit_image<G, NH>(Wl, wrA, st);
      issue_image<G, NH - 2>(a, wrA, st);
      lds_barrier();
    } else {
      stage_prologue<G, true>(a, Wl, wrA, wrB, st);
      lds_barrier();
      LGN_STAGE_STEP(0) LGN_STAGE_STEP(1) LGN_STAGE_STEP(2) LGN_STAGE_STEP(3) LGN_STAGE_STEP(4) LGN_STAGE_STEP(5)
    }
    LGN_STAGE_STEP(6) LGN_STAGE_STEP(7) LGN_STAGE_STEP(8) LGN_STAGE_STEP(9) LGN_STAGE_STEP(10) LGN_STAGE_STEP(11) LGN_STAGE_STEP(12)
#undef LGN_STAGE_STEP
    return;
  }
  if (wave == 1) {
    // ================= weight gradients, one step behind the chain =================
    double* part = a.part + (size_t)blockIdx.x * a.psize;
    lds_barrier();
    if constexpr (!SAVE) {
#pragma unroll
      for (int q = 0; q < 6; ++q) lds_barrier();
    }
    lds_barrier();                                           // q = 6: nothing published yet
    dw16<G, 6>(Gt + 0 * G::TSIZE, Xt + 0 * G::TSIZE, part, c, g);  lds_barrier();
    dw16<G, 5>(Gt + 1 * G::TSIZE, Xt + 1 * G::TSIZE, part, c, g);  lds_barrier();
    dw16<G, 4>(Gt + 0 * G::TSIZE, Xt + 0 * G::TSIZE, part, c, g);  lds_barrier();
    dw16<G, 3>(Gt + 1 * G::TSIZE, Xt + 1 * G::TSIZE, part, c, g);  lds_barrier();
    dw16<G, 2>(Gt + 0 * G::TSIZE, Xt + 0 * G::TSIZE, part, c, g);  lds_barrier();
    dw16<G, 1>(Gt + 1 * G::TSIZE, Xt + 1 * G::TSIZE, part, c, g);  lds_barrier();
    dw16<G, 0>(Gt, Xt, part, c, g);
    return;
  }
  // ================= the chain =================
  const int row = blockIdx.x * 16 + c;
  v4d xb[1];
  load_x<G>(a, row, g, xb);
  v4d gout[1];                                               // upstream gradient, D layout of the output tile (o = 4r + g, row c)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * r + g;
    const bool ok = o < D && row < a.M;
    const double x = a.g_out[ok ? mlp_out_index(a, o & 1, row, o >> 1) : 0];
    gout[0][r] = ok ? x : 0.0;
  }
  v4d h[NH][NT];
  const double* hs = SAVE ? saved_ptr16<G>(a, lane) : nullptr;
  if constexpr (SAVE) {
#pragma unroll
    for (int j = 0; j < saved_pieces<G>(); ++j) load_piece<G>(hs, NH - 1, h[NH - 1], j);      // needed by the first two steps; the
#pragma unroll
    for (int j = 0; j < saved_pieces<G>(); ++j) load_piece<G>(hs, NH - 2, h[NH - 2], j);      // others follow two steps ahead of their use
    lds_barrier();
  } else {
    lds_barrier();
#define LGN_CHAIN_STEP(Q, HIN, KS, NTI)                                                                              \
  layer_fwd<G, KS, NTI, GEN>(Wl + (Q & 1) * G::WSIZE, HIN, h[Q], c, g, a.act, [&](int) {});                          \
  lds_barrier();
    LGN_CHAIN_STEP(0, xb, G::KS0, 1)
    LGN_CHAIN_STEP(1, h[0], G::KSH, NT)
    LGN_CHAIN_STEP(2, h[1], G::KSH, NT)
    LGN_CHAIN_STEP(3, h[2], G::KSH, NT)
    LGN_CHAIN_STEP(4, h[3], G::KSH, NT)
    LGN_CHAIN_STEP(5, h[4], G::KSH, NT)
#undef LGN_CHAIN_STEP
  }
  v4d gpA[NT], gpB[NT], gin[NT];
  {  // q = 6, l = 6: the output layer
    constexpr int NI = G::KS0 * NT, NSV = SAVE && NH >= 3 ? saved_pieces<G>() : 0, NPC = 4 + 4 * NT + NSV;
    auto fin = [&](int u, int r) {
      double y = gin[u][r] * act_slope_t<GEN>(h[NH - 1][u][r], a.act);
      pin(y);
      gpA[u][r] = y;
    };
    layer_bwd<G, G::KS0, NT, 1>(Wl + 0 * G::WSIZE, gout, gin, c, g, fin, [&](int i) {
      deal<NI, NPC>(i, [&](int j) {
        if (j < 4) publish_piece<G, 1>(Gt, gout, 0, c, g, j);
        else if (j < 4 + 4 * NT) publish_piece<G, NT>(Xt, h[NH - 1], 0, c, g, j - 4);
        else if constexpr (NSV > 0) load_piece<G>(hs, NH - 3, h[NH >= 3 ? NH - 3 : 0], j - 4 - 4 * NT);
      });
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) fin(NT - 1, r);
    lds_barrier();
  }
#define LGN_CHAIN_BSTEP(L, GP, GN)                                                                                   \
  {                                                                                                                  \
    constexpr int q_ = 2 * NH - (L), NI = G::KSH * NT, NSV = SAVE && (L) >= 3 ? saved_pieces<G>() : 0, NPC = 8 * NT + NSV; \
    auto fin = [&](int u, int r) {                                                                                   \
      double y = gin[u][r] * act_slope_t<GEN>(h[(L) - 1][u][r], a.act);                                              \
      pin(y);                                                                                                        \
      GN[u][r] = y;                                                                                                  \
    };                                                                                                               \
    layer_bwd<G, G::KSH, NT, NT>(Wl + (q_ & 1) * G::WSIZE, GP, gin, c, g, fin, [&](int i) {                          \
      deal<NI, NPC>(i, [&](int j) {                                                                                  \
        if (j < 4 * NT) publish_piece<G, NT>(Gt + ((L) & 1) * G::TSIZE, GP, 0, c, g, j);                             \
        else if (j < 8 * NT) publish_piece<G, NT>(Xt + ((L) & 1) * G::TSIZE, h[(L) - 1], 0, c, g, j - 4 * NT);       \
        else if constexpr (NSV > 0) load_piece<G>(hs, (L) - 3, h[(L) >= 3 ? (L) - 3 : 0], j - 8 * NT);               \
      });                                                                                                            \
    });                                                                                                              \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) fin(NT - 1, r);                                                    \
    lds_barrier();                                                                                                   \
  }
  LGN_CHAIN_BSTEP(5, gpA, gpB)
  LGN_CHAIN_BSTEP(4, gpB, gpA)
  LGN_CHAIN_BSTEP(3, gpA, gpB)
  LGN_CHAIN_BSTEP(2, gpB, gpA)
  LGN_CHAIN_BSTEP(1, gpA, gpB)
#undef LGN_CHAIN_BSTEP
  {  // q = 12, l = 0: the first layer
    v4d gx[1];
    layer_bwd<G, G::KSH, 1, NT>(Wl + 0 * G::WSIZE, gpB, gx, c, g, [&](int, int) {}, [&](int i) {
      deal<G::KSH, 4 * NT + 4>(i, [&](int j) {
        if (j < 4 * NT) publish_piece<G, NT>(Gt, gpB, 0, c, g, j);
        else publish_piece<G, 1>(Xt, xb, 0, c, g, j - 4 * NT);
      });
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 4 * r + g;
      if (f < D && row < a.M) a.g_in[mlp_out_index(a, f & 1, row, f >> 1)] = gx[0][r];
    }
    lds_barrier();
  }
}

// ---- 16-row workgroups, every layer split over THREE chain waves (NT = 3; kept activations) ---------------------------------------
// At small batches a CU holds one workgroup and most SIMDs idle: a layer's three output tiles go to three waves (12 matrix
// instructions each instead of 36 in a row).  Wave t keeps its own tile of the previous layer in registers (it multiplies those
// k-steps first) and reads the other two from the [neuron][16 rows] tile the waves publish into before the layer's barrier -- in the
// backward those are the tiles the weight gradients are computed from anyway.  One barrier per layer; weight images by two staging
// waves; the backward's weight gradients by two more, concurrent with the chain step of the SAME layer.
// B operand of k-step ks (neuron 4 ks + g = tile ks >> 2, register ks & 3) for wave t in ROTATED order i -> ks = (4 t + i) mod KS
template <class G, int KS>
__device__ __forceinline__ void split_operands(const double* T, const double (&own)[4], int t, int c, int g, double (&hb)[KS]) {
  const int nown = KS - 4 * t < 4 ? KS - 4 * t : 4;          // (H = 36: the last tile has one k-step)
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    int ks = 4 * t + i;
    if (ks >= KS) ks -= KS;
    const double v = T[(4 * ks + g) * G::SR + c];
    hb[i] = (i < 4 && i < nown) ? own[i < 4 ? i : 0] : v;
  }
}
template <int H, int D, bool GEN, bool SAVE>
__global__ __launch_bounds__(320) void mlp_chain_fwd16s_kernel(MlpArgs<double> a) {
  using G = Geo<H, D, 128>;
  static_assert(G::NT == 3, "three chain waves");
  constexpr int S = G::S, SR = G::SR, NH = G::NH, KSH = G::KSH, KS0 = G::KS0, HP = G::HP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  double* At = Wl + 2 * G::WSIZE;                            // 2 activation tiles [neuron][16 rows]
  if (wave >= 3) {
    const int st = tid - 192;
    WRegs<G> wrA, wrB;
    stage_prologue<G, false>(a, Wl, wrA, wrB, st);
    lds_barrier();
#define LGN_STAGE_STEP(Q)                                                                                            \
  _Pragma("unroll") for (int j = 0; j < stage_pieces<G>(); ++j) stage_piece<G, false, Q>(a, Wl, wrA, wrB, st, j);    \
  lds_barrier();
    LGN_STAGE_STEP(0) LGN_STAGE_STEP(1) LGN_STAGE_STEP(2) LGN_STAGE_STEP(3) LGN_STAGE_STEP(4) LGN_STAGE_STEP(5)
#undef LGN_STAGE_STEP
    return;
  }
  const int t = wave, row = blockIdx.x * 16 + c;
  v4d xb[1];
  load_x<G>(a, row, g, xb);
  double* hs = SAVE ? saved_ptr16<G>(a, lane) + t * 256 : nullptr;
  lds_barrier();
  double own[4];
  auto finish = [&](int q, v4d& acc) {                      // activation, publication (and the backward's copy) of this wave's tile of layer q
    double* T = At + (q & 1) * G::TSIZE;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      own[r] = act_t<GEN>(acc[r], a.act);
      T[(16 * t + 4 * r + g) * SR + c] = own[r];
    }
    if (SAVE) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
        *reinterpret_cast<v2d*>(hs + q * G::NT * 256 + 2 * hf) = v2d{own[2 * hf], own[2 * hf + 1]};      // (plain store: a small batch's copy -- 4.4 MB
                                                                                                        // at 64 jets -- is read back from the same XCD's L2)
    }
  };
  {  // layer 0: inputs from registers
    const double* Wc = Wl;
    const double* wa = Wc + (16 * t + c) * S + g;
    v4d acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Wc[(16 * t + 4 * r + g) * S + HP];
    mfma_stream<KS0, LOOKAHEAD>([&](int i) { return wa[4 * i]; },
                                [&](int i, double av) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, xb[0][i], acc, 0, 0, 0); }, [&](int) {});
    finish(0, acc);
    lds_barrier();
  }
#pragma unroll
  for (int q = 1; q < NH; ++q) {
    const double* Wc = Wl + (q & 1) * G::WSIZE;
    const double* wa = Wc + (16 * t + c) * S + g;
    double hb[KSH];
    split_operands<G, KSH>(At + ((q - 1) & 1) * G::TSIZE, own, t, c, g, hb);
    v4d acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Wc[(16 * t + 4 * r + g) * S + HP];
    mfma_stream<KSH, LOOKAHEAD>([&](int i) { int ks = 4 * t + i; if (ks >= KSH) ks -= KSH; return wa[4 * ks]; },
                                [&](int i, double av) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, hb[i], acc, 0, 0, 0); }, [&](int) {});
    finish(q, acc);
    lds_barrier();
  }
  if (t != 0) return;
  {  // output layer: one tile, wave 0
    const double* Wc = Wl + (NH & 1) * G::WSIZE;
    const double* wa = Wc + c * S + g;
    double hb[KSH];
    split_operands<G, KSH>(At + ((NH - 1) & 1) * G::TSIZE, own, 0, c, g, hb);
    v4d y;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = Wc[(4 * r + g) * S + HP];
    mfma_stream<KSH, LOOKAHEAD>([&](int i) { return wa[4 * i]; },
                                [&](int i, double av) { y = __builtin_amdgcn_mfma_f64_16x16x4f64(av, hb[i], y, 0, 0, 0); }, [&](int) {});
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * r + g;
      if (o < D && row < a.M) a.s_out[mlp_out_index(a, o & 1, row, o >> 1)] = y[r];
    }
  }
}

// weight gradient of Linear L over 16 rows, tiles w, w + NW, ... (NW dW waves); wave 0 also the bias gradient
template <class G, int L, int NW = 4>
__device__ __forceinline__ void dw16_part(const double* Gt, const double* Xt, double* part, int w, int c, int g) {
  constexpr int SR = G::SR, HO = G::hout(L), HI = G::hin(L);
  constexpr int NTO = L == G::NH ? 1 : G::NT, NTI = L == 0 ? 1 : G::NT, NTL = NTO * NTI;
  double* pW = part + G::off_w(L);
  double qa[NTO][4], qx[NTI][4];
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) qa[t][sk] = Gt[(16 * t + c) * SR + g + 4 * sk];
#pragma unroll
  for (int u = 0; u < NTI; ++u)
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) qx[u][sk] = Xt[(16 * u + c) * SR + g + 4 * sk];
  __builtin_amdgcn_sched_barrier(0);
  v4d acc[NTL];
  auto store = [&](int tile) {
    const int t = tile / NTI, u = tile - t * NTI;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * t + 4 * r + g, k = 16 * u + c;
      if (o < HO && k < HI) __builtin_nontemporal_store(acc[tile][r], &pW[o * HI + k]);
    }
  };
  // a wave's tiles are w, w + NW, ...: the stores of one run under the matrix instructions of the next
#pragma unroll
  for (int tile = 0; tile < NTL; ++tile) {
    if (tile % NW == w) {
      const int t = tile / NTI, u = tile - t * NTI;
      acc[tile] = v4d{0, 0, 0, 0};
#pragma unroll
      for (int sk = 0; sk < 4; ++sk) acc[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t][sk], qx[u][sk], acc[tile], 0, 0, 0);
      if (tile >= NW) store(tile >= NW ? tile - NW : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int tile = (NTL > NW ? NTL - NW : 0); tile < NTL; ++tile)      // each wave's last tile
    if (tile % NW == w) store(tile);
  if (w == 0) {
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
      double dbs = (qa[t][0] + qa[t][1]) + (qa[t][2] + qa[t][3]);
      dbs += shfl_xor(dbs, 16);
      dbs += shfl_xor(dbs, 32);
      if (g == 0 && 16 * t + c < HO) __builtin_nontemporal_store(dbs, &part[G::off_b(L) + 16 * t + c]);
    }
  }
}

// backward with kept activations (the forward above, SAVE): 3 chain waves | 4 weight-gradient waves | 2 staging waves
template <int H, int D, bool GEN>
__global__ __launch_bounds__(576) void mlp_chain_bwd16s_kernel(MlpArgs<double> a) {
  using G = Geo<H, D, 128>;
  static_assert(G::NT == 3, "three chain waves");
  constexpr int S = G::S, SR = G::SR, NT = G::NT, NH = G::NH, KSH = G::KSH, KS0 = G::KS0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* Wl = reinterpret_cast<double*>(smem_raw);          // 2 weight images
  double* Gt = Wl + 2 * G::WSIZE;                            // 2 g_pre tiles [neuron][16 rows]
  double* Xt = Gt + 2 * G::TSIZE;                            // 2 layer-input tiles
  if (wave >= 7) {
    // ================= image staging: the images start at the output layer =================
    const int st = (int)threadIdx.x - 448;
    WRegs<G> wrA, wrB;
    issue_image<G, NH>(a, wrA, st);
    issue_image<G, NH - 1>(a, wrB, st);
    commit_image<G, NH>(Wl, wrA, st);
    issue_image<G, NH - 2>(a, wrA, st);
    lds_barrier();
#define LGN_STAGE_STEP(Q)                                                                                            \
  _Pragma("unroll") for (int j = 0; j < stage_pieces<G>(); ++j) stage_piece<G, true, Q>(a, Wl, wrA, wrB, st, j);     \
  lds_barrier();
    LGN_STAGE_STEP(6) LGN_STAGE_STEP(7) LGN_STAGE_STEP(8) LGN_STAGE_STEP(9) LGN_STAGE_STEP(10) LGN_STAGE_STEP(11)
#undef LGN_STAGE_STEP
    return;
  }
  if (wave >= 3) {
    // ================= weight gradients of Linear L during the chain's step for Linear L =================
    const int w = wave - 3;
    double* part = a.part + (size_t)blockIdx.x * a.psize;
    lds_barrier();
    dw16_part<G, 6>(Gt + 0 * G::TSIZE, Xt + 0 * G::TSIZE, part, w, c, g);  lds_barrier();
    dw16_part<G, 5>(Gt + 1 * G::TSIZE, Xt + 1 * G::TSIZE, part, w, c, g);  lds_barrier();
    dw16_part<G, 4>(Gt + 0 * G::TSIZE, Xt + 0 * G::TSIZE, part, w, c, g);  lds_barrier();
    dw16_part<G, 3>(Gt + 1 * G::TSIZE, Xt + 1 * G::TSIZE, part, w, c, g);  lds_barrier();
    dw16_part<G, 2>(Gt + 0 * G::TSIZE, Xt + 0 * G::TSIZE, part, w, c, g);  lds_barrier();
    dw16_part<G, 1>(Gt + 1 * G::TSIZE, Xt + 1 * G::TSIZE, part, w, c, g);  lds_barrier();
    dw16_part<G, 0>(Gt, Xt, part, w, c, g);
    return;
  }
  // ================= the chain: wave u owns tile u of every layer =================
  const int u = wave, row = blockIdx.x * 16 + c;
  STAMP(10);
  const double* hs = saved_ptr16<G>(a, lane) + u * 256;
  double hown[NH][4];                                        // this wave's tile of the six hidden activations
#pragma unroll
  for (int l = 0; l < NH; ++l)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const v2d v = *reinterpret_cast<const v2d*>(hs + l * NT * 256 + 2 * hf);
      hown[l][2 * hf] = v[0];
      hown[l][2 * hf + 1] = v[1];
    }
  v4d xb[1], gout[1];
  load_x<G>(a, row, g, xb);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * r + g;
    const bool ok = o < D && row < a.M;
    const double x = a.g_out[ok ? mlp_out_index(a, o & 1, row, o >> 1) : 0];
    gout[0][r] = ok ? x : 0.0;
  }
  // publication of this wave's tile (rows 16 u + 4 r + g) / of a single-tile operand (rows 4 r + g, wave 0)
  auto publish = [&](double* T, const double (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) T[(16 * u + 4 * r + g) * SR + c] = v[r];
  };
  auto publish1 = [&](double* T, const v4d& v) {
    if (u == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(4 * r + g) * SR + c] = v[r];
    }
  };
  STAMP(27);
  publish1(Gt, gout[0]);                                     // dW of the output layer: g_out and h_5
  publish(Xt, hown[NH - 1]);
  lds_barrier();
  STAMP(11);
  double gp[4];                                              // this wave's tile of g_pre of the layer below
  {  // l = 6: the output layer -- K = its 2C outputs, from registers
    const double* Wc = Wl;
    const double* wa = Wc + g * S + 16 * u + c;
    v4d gin = {0, 0, 0, 0};
    mfma_stream<KS0, LOOKAHEAD>([&](int i) { return wa[4 * i * S]; },
                                [&](int i, double av) { gin = __builtin_amdgcn_mfma_f64_16x16x4f64(av, gout[0][i], gin, 0, 0, 0); }, [&](int) {});
#pragma unroll
    for (int r = 0; r < 4; ++r) gp[r] = gin[r] * act_slope_t<GEN>(hown[NH - 1][r], a.act);
    publish(Gt + G::TSIZE, gp);                              // g_pre of Linear 5 (parity 1) and its input h_4
    publish(Xt + G::TSIZE, hown[NH - 2]);
    lds_barrier();
    STAMP(12);
  }
#pragma unroll
  for (int L = NH - 1; L >= 1; --L) {
    // Linear L: g_in = W_L^T g_pre_L (this wave's tile of the inputs), image in buffer (2 NH - L) & 1 = L & 1
    const double* Wc = Wl + (L & 1) * G::WSIZE;
    const double* wa = Wc + g * S + 16 * u + c;
    double hb[KSH];
    split_operands<G, KSH>(Gt + (L & 1) * G::TSIZE, gp, u, c, g, hb);
    v4d gin = {0, 0, 0, 0};
    mfma_stream<KSH, LOOKAHEAD>([&](int i) { int ks = 4 * u + i; if (ks >= KSH) ks -= KSH; return wa[4 * ks * S]; },
                                [&](int i, double av) { gin = __builtin_amdgcn_mfma_f64_16x16x4f64(av, hb[i], gin, 0, 0, 0); }, [&](int) {});
#pragma unroll
    for (int r = 0; r < 4; ++r) gp[r] = gin[r] * act_slope_t<GEN>(hown[L - 1][r], a.act);
    publish(Gt + ((L - 1) & 1) * G::TSIZE, gp);              // g_pre of Linear L - 1 and its input (h_{L-2}, or the MLP's input rows)
    if (L >= 2) publish(Xt + ((L - 1) & 1) * G::TSIZE, hown[L >= 2 ? L - 2 : 0]);
    else publish1(Xt, xb[0]);
    STAMP(20 + L);
    lds_barrier();
    STAMP(12 + NH - L);
  }
  if (u != 0) return;
  {  // Linear 0: the gradient of the MLP's input rows (one tile), wave 0
    const double* Wc = Wl;
    const double* wa = Wc + g * S + c;
    double hb[KSH];
    split_operands<G, KSH>(Gt, gp, 0, c, g, hb);
    v4d gx = {0, 0, 0, 0};
    mfma_stream<KSH, LOOKAHEAD>([&](int i) { return wa[4 * i * S]; },
                                [&](int i, double av) { gx = __builtin_amdgcn_mfma_f64_16x16x4f64(av, hb[i], gx, 0, 0, 0); }, [&](int) {});
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 4 * r + g;
      if (f < D && row < a.M) a.g_in[mlp_out_index(a, f & 1, row, f >> 1)] = gx[r];
    }
    STAMP(19);
  }
}

template <int H, int D>
static int launch16s(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  using G = Geo<H, D, 128>;
  const int nblk = cdiv(a.M, 16);
  const size_t smem = sizeof(double) * (2 * G::WSIZE + (backward ? 4 : 2) * G::TSIZE);
  if (backward) LGN_CHECK_ARG(a.psize == G::psize(), "cgmlp: psize %d, expected %d", a.psize, G::psize());
  LGN_CHECK_ARG(!a.h_saved || a.h_rows >= nblk * 16, "cgmlp: the saved-activation buffer has %d rows per layer, %d rows need %d",
                a.h_rows, a.M, nblk * 16);
  void (*kern)(MlpArgs<double>);
  if (backward) kern = a.act == 0 ? mlp_chain_bwd16s_kernel<H, D, false> : mlp_chain_bwd16s_kernel<H, D, true>;
  else kern = a.h_saved ? (a.act == 0 ? mlp_chain_fwd16s_kernel<H, D, false, true> : mlp_chain_fwd16s_kernel<H, D, true, true>)
                        : (a.act == 0 ? mlp_chain_fwd16s_kernel<H, D, false, false> : mlp_chain_fwd16s_kernel<H, D, true, false>);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(backward ? 576 : 320), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

template <int H, int D>
static int launch16(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  using G = Geo<H, D, 128>;
  const int nblk = cdiv(a.M, 16);
  const size_t smem = backward ? G::bwd_bytes() : G::fwd_bytes();
  if (backward) LGN_CHECK_ARG(a.psize == G::psize(), "cgmlp: psize %d, expected %d", a.psize, G::psize());
  LGN_CHECK_ARG(!a.h_saved || a.h_rows >= nblk * 16, "cgmlp: the saved-activation buffer has %d rows per layer, %d rows need %d",
                a.h_rows, a.M, nblk * 16);
  void (*kern)(MlpArgs<double>);
  if (backward) kern = a.h_saved ? (a.act == 0 ? mlp_chain_bwd16_kernel<H, D, false, true> : mlp_chain_bwd16_kernel<H, D, true, true>)
                                 : (a.act == 0 ? mlp_chain_bwd16_kernel<H, D, false, false> : mlp_chain_bwd16_kernel<H, D, true, false>);
  else kern = a.h_saved ? (a.act == 0 ? mlp_chain_fwd16_kernel<H, D, false, true> : mlp_chain_fwd16_kernel<H, D, true, true>)
                        : (a.act == 0 ? mlp_chain_fwd16_kernel<H, D, false, false> : mlp_chain_fwd16_kernel<H, D, true, false>);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(backward ? 256 : 192), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

// 48 < H <= 96 (C = 5 .. 8): the FORWARD chain only -- its two weight images (up to 150 KB) and ping-ponged activations fit; the
// backward's tiles and six kept activations do not (LDS 274 KB, > 512 registers), it stays with mlp_mfma_wide.hip
template <int H, int D>
static int launch_fwd(const MlpArgs<double>& a, hipStream_t stream) {
  using G = Geo<H, D>;
  static_assert(G::fwd_bytes() <= 160 * 1024, "LDS budget");
  const bool two = !(a.flags & LVL_MLP_BWD1);
  auto kern = two ? (a.act == 0 ? mlp_chain_fwd_kernel<H, D, false, false, true> : mlp_chain_fwd_kernel<H, D, true, false, true>)
                  : (a.act == 0 ? mlp_chain_fwd_kernel<H, D, false, false> : mlp_chain_fwd_kernel<H, D, true, false>);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::fwd_bytes());
  hipLaunchKernelGGL(kern, dim3(cdiv(a.M, 64)), dim3(two ? 512 : 256), G::fwd_bytes(), stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

template <int H, int D>
static int launch(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  using G = Geo<H, D>;
  const int nblk = cdiv(a.M, 64);
  const size_t smem = backward ? G::bwd_bytes() : G::fwd_bytes();
  static_assert(G::bwd_bytes() <= 160 * 1024, "LDS budget");
  if (backward) LGN_CHECK_ARG(a.psize == G::psize(), "cgmlp: psize %d, expected %d", a.psize, G::psize());
  LGN_CHECK_ARG(!a.h_saved || a.h_rows >= nblk * 64, "cgmlp: the saved-activation buffer has %d rows per layer, %d rows need %d",
                a.h_rows, a.M, nblk * 64);
  if (backward && !a.h_saved && !(a.flags & LVL_MLP_BWD1)) {      // (kept activations: the one-role kernel reads them)
    auto k2 = a.act == 0 ? mlp_chain_bwd2_kernel<H, D, false> : mlp_chain_bwd2_kernel<H, D, true>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k2, dim3(nblk), dim3(512), smem, stream, a);
    LGN_CHECK_LAUNCH();
    return 0;
  }
  if (!backward && !a.h_saved && !(a.flags & LVL_MLP_BWD1)) {
    auto k2 = a.act == 0 ? mlp_chain_fwd_kernel<H, D, false, false, true> : mlp_chain_fwd_kernel<H, D, true, false, true>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k2, dim3(nblk), dim3(512), smem, stream, a);
    LGN_CHECK_LAUNCH();
    return 0;
  }
  auto kern = a.h_saved ? (a.act == 0 ? (backward ? mlp_chain_bwd_kernel<H, D, false, true> : mlp_chain_fwd_kernel<H, D, false, true>)
                                      : (backward ? mlp_chain_bwd_kernel<H, D, true, true> : mlp_chain_fwd_kernel<H, D, true, true>))
                        : (a.act == 0 ? (backward ? mlp_chain_bwd_kernel<H, D, false, false> : mlp_chain_fwd_kernel<H, D, false, false>)
                                      : (backward ? mlp_chain_bwd_kernel<H, D, true, false> : mlp_chain_fwd_kernel<H, D, true, false>));
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace chain

// The reference widths H = 6 * 2C (C = 1 .. 4): 64-row workgroups (large batches, activations recomputed) or 16-row ones (small batches); -2 = not this kernel's shape.
int mlp_chain_dispatch(const MlpArgs<double>& a, bool backward, hipStream_t stream) {
  if ((a.flags & LVL_MLP_V1) || a.nlin != 7) return -2;
  if (mlp_rows_per_workgroup(a.M, a.H) != 64) {             // small batches: 16-row workgroups (one chain wave + helpers)
    // three chain waves per 16 rows: the forward always, the backward where the forward kept its activations (LGN_AMD_MLP_BWD1: the
    // one-chain-wave kernels everywhere)
    if (!(a.flags & LVL_MLP_BWD1) && (!backward || a.h_saved)) {
      if (a.H == 48 && a.C == 4) return chain::launch16s<48, 8>(a, backward, stream);
      if (a.H == 36 && a.C == 3) return chain::launch16s<36, 6>(a, backward, stream);
    }
    if (a.H == 48 && a.C == 4) return chain::launch16<48, 8>(a, backward, stream);
    if (a.H == 36 && a.C == 3) return chain::launch16<36, 6>(a, backward, stream);
    if (a.H == 24 && a.C == 2) return chain::launch16<24, 4>(a, backward, stream);
    if (a.H == 12 && a.C == 1) return chain::launch16<12, 2>(a, backward, stream);
    return -2;
  }
  if (a.H == 48 && a.C == 4) return chain::launch<48, 8>(a, backward, stream);
  if (a.H == 36 && a.C == 3) return chain::launch<36, 6>(a, backward, stream);
  if (a.H == 24 && a.C == 2) return chain::launch<24, 4>(a, backward, stream);
  if (a.H == 12 && a.C == 1) return chain::launch<12, 2>(a, backward, stream);
  if (!backward && !a.h_saved) {
    if (a.H == 60 && a.C == 5) return chain::launch_fwd<60, 10>(a, stream);
    if (a.H == 72 && a.C == 6) return chain::launch_fwd<72, 12>(a, stream);
    if (a.H == 84 && a.C == 7) return chain::launch_fwd<84, 14>(a, stream);
    if (a.H == 96 && a.C == 8) return chain::launch_fwd<96, 16>(a, stream);
  }
  return -2;
}

}  // namespace lgn
