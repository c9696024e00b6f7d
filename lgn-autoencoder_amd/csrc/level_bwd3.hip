// lgn-autoencoder_amd/csrc/level_bwd3.hip -- the whole backward of a fused maxdim=2 level in ONE kernel per jet
// (N <= 40): CatMix/power backward, the j-centric pass (node gradients), and the radial-parameter pass share one
// staging of the jet and ONE sweep over the particle pairs; the gradient of the aggregate never leaves LDS and the
// node gradient is written once.  Same mathematics as level_bwd.hip / level_bwd2.hip (reference: autograd through
// lgn/nn/position_levels.py:118-209, lgn/models/lgn_cg.py:167, lgn/cg_lib/cg_ops.py:135-298, lgn/nn/g_nn.py:260-278).
//
//   phase 1  thread = (node, channel): g_cat = W^H g_out -> g_ag (LDS), direct + power terms of the node gradient (LDS);
//            thread = (out channel, cat slot): CatMix weight gradient over the jet's nodes -> partial row
//   phase 2  wave = 4 source particles j, tiles of 4 receivers i, lane = (pair, channel in group), radial Linear on the
//            matrix cores.  Per pair: (a) g_node_j += g_ag_i (x) conj(edge_ij); (b) encoder: dL/d rad of the pair ->
//            16x16 LDS transposes -> T1|T2|S|dB GEMM on the matrix cores; decoder: bias sums and d p_j
//   phase 3  decoder only: i-centric sweep for d p_i
#include "level_dev.hpp"
#include "ops.hpp"
#include "wave_sum.hpp"

namespace lgn {

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ double dppq(double v, int xor2) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (xor2) {
    lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
  } else {
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += dppq(v, 0);
  v += dppq(v, 1);
  return v;
}
__device__ __forceinline__ double fast_rcp(double u) {
  double r = __builtin_amdgcn_rcp(u);
  double e = __builtin_fma(-u, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-u, r, 1.0);
  return __builtin_fma(r, e, r);
}
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <int C> struct GA3 {
  // (+ 2: rows of 20 C doubles put every second node of C = 4 on the same LDS banks -- the pair sweep reads four nodes' rows at once)
  static constexpr int A3 = 0, A4 = 2 * C, A1 = 4 * C, A2 = 12 * C, SIZE = 20 * C + 2;
};
constexpr int TS = 18;
LGN_STAMP_DECL
}  // namespace
LGN_STAMP_READER(lgn_debug_stamps_bwd3)

// NWV = waves per workgroup (4; 8 was measured for small batches: 31 -> 28.5 us at 64 jets -- a lone workgroup already keeps
// its CU's SIMDs two thirds busy, the idle CUs are what a small batch wastes: level_jet_split spreads a jet over several CUs)
template <int C, bool DEC, int NWV = 4>
struct Bwd3 {
  static constexpr int BLK = 64 * NWV;
  static constexpr int NG = (C + 3) / 4;
  static constexpr int NS = node_stride(C);
  static constexpr int PS = DEC ? 8 : 4;
  static constexpr int TRSZ = DEC ? NWV * 64 : (NWV * (NG + 3) * 16 * TS > NWV * 64 * NG * 12 ? NWV * (NG + 3) * 16 * TS : NWV * 64 * NG * 12);
  // phase-1 scratch (upstream gradient tile + CatMix weights) and phase-2 scratch (transpose tiles) share one region
  // phase 1: upstream gradient tile | CatMix weights | aggregate saved by the forward; then (aliased, after a barrier)
  // the per-part partial sums of the CatMix weight gradient
  static constexpr int MIXP = BLK / (5 * C) < 16 ? BLK / (5 * C) : 16;       // node parts of the CatMix weight gradient
  __host__ __device__ static size_t scratch(int N, int CO) {
    size_t p1 = (size_t)N * 10 * CO + 4 * CO * 5 * C + (size_t)N * 20 * C;
    const size_t red = (size_t)MIXP * CO * 5 * C * 4;
    if (red > p1) p1 = red;
    return p1 > (size_t)TRSZ ? p1 : (size_t)TRSZ;
  }
  static constexpr int SMS = 52;                         // 50 jet-level sums per channel (+ 2 of padding: 16 + 16 + 12 + 8 lane sums)
  static constexpr int SEPSZ = DEC ? SMS * C : 0;        // jet-level sums of the separable decoder form
  static size_t smem(int N, int CO) {
    return sizeof(double) * ((((size_t)N * NS + 1) & ~size_t(1)) + (size_t)N * (20 * C + 2) + (size_t)N * 10 * C + (size_t)N * PS +
                             scratch(N, CO) + SEPSZ) + N + 16;
  }
};

// SEP (decoder only): with the decoder's all-zero edge mask the radial weights are the per-channel constants R0, R1, so
// every sum over pairs separates into jet-level sums (see level_fwd2.hip); the pair sweep (phases 2, 3) is replaced by
//   stage 1   per (node, channel): the node's terms of S, VS, SP, VP (forward) and of the sums of g_ag (backward)
//   stage 2   per (channel, term): sum over the nodes in node order
//   outputs   node gradient, position gradient and bias gradients from O(N C) closed forms.
// (amdgpu_waves_per_eu(2): two workgroups per CU is what the 77 KB of LDS are sized for.  Round 6: the encoder instantiations with
// C >= 5 spilled 70 - 76 registers into the pair loop at that budget -- they get one wave per SIMD and ~312 registers instead:
// 142 - 159 us -> ~115 - 135 us per launch at 512 jets, tools/wide_levels.py, profiles/r06_c5to8_levels.txt)
// SYM (encoder, whole jet in one workgroup): the radial network sees a pair only through |p_i - p_j|^2 and the masks, so R(i, j) =
// R(j, i) and the gradient w.r.t. the pair's radial values is the SUM of what the two directed edges i <- j and j <- i send back.
// The wave that owns the source group J therefore adds, on its tiles with receiver group I < J, the reverse edge's share (receiver
// j, source i: a second, shorter pass over the same lanes) before the radial-parameter GEMM, and skips that GEMM -- 12 of a tile's 17
// matrix instructions, with the basis rows and transposes that feed it -- on its tiles with I > J, whose share the owner of I adds.
// 36 instead of 64 radial GEMM tiles per 30-particle jet; waves own the groups in pairs (p, G - 1 - p) so that each gets the same
// number of them.  Node gradients flow exactly as before (every ordered tile still evaluates R and its edge).
template <int C, bool DEC, bool SEP, int NWV, bool SYM = false>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(C <= 4 || DEC ? 2 : 1, C <= 4 || DEC ? 2 : 1))) void level_bwd3_kernel(LevelBwdArgs<double> a) {
  static_assert(!SYM || !DEC, "the symmetric sweep is the encoder's");
  using F = Bwd3<C, DEC, NWV>;
  constexpr int BLK = F::BLK;
  using G = GA3<C>;
  constexpr int NG = F::NG, NS = F::NS, PS = F::PS, K = 5 * C;
  const int N = a.N, B = a.B, CO = a.CO;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this workgroup's share of the jet (level_jet_split): groups [glo, ghi) of 4 particles = nodes [nlo, nhi)
  const int ngroups = (N + 3) >> 2, gper = (ngroups + (int)gridDim.y - 1) / (int)gridDim.y;
  const int glo = min(ngroups, (int)blockIdx.y * gper), ghi = min(ngroups, glo + gper);
  const int nlo = min(N, 4 * glo), nhi = min(N, 4 * ghi);
  const size_t prow = (size_t)b * gridDim.y + blockIdx.y;       // partial row of this workgroup

  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* nd = reinterpret_cast<double*>(smem_raw);            // N * NS      node features entering the level
  double* ga = nd + ((N * NS + 1) & ~1);                       // N * 20C     gradient of the aggregate
  double* gd = ga + N * G::SIZE;                               // N * 10C     direct + power part of the node gradient [n][c][s2|v8]
  double* pj = gd + N * 10 * C;                                // N * PS
  double* tr = pj + N * PS;                                    // phase 2: transpose tiles / reduction scratch ...
  double* go = tr;                                             // ... phase 1: N * 10CO upstream gradient [n][o][s2|v8]
  double* wm = go + N * 10 * CO;                               //              4 * CO * K CatMix weights
  double* agl = wm + 4 * CO * K;                               //              N * 2C * 10 aggregate [n][q*C+c][s2|v8]
  double* sm = tr + F::scratch(N, CO);                         // decoder: 50 C jet-level sums
  uint8_t* mk = reinterpret_cast<uint8_t*>(sm + F::SEPSZ);

  // ---------------- staging ----------------------------------------------------------------------------
  STAMP(0);
  // Every thread's first item of every array is requested before anything is written to LDS (see load_jet_issue): the jet, the
  // upstream gradient, the saved aggregate and the CatMix weights arrive in ONE memory round trip instead of six.
  {
    const size_t plo = (size_t)B * N * CO, pa = (size_t)B * N * 2 * C;
    JetRegs<double> jr;
    load_jet_issue<double, C, DEC>(a.s_in, a.v_in, a.p, a.mask, B, N, b, jr);
    const int eg = tid < N * CO ? tid : 0, ea_ = tid < N * 2 * C ? tid : 0, ew = tid < 2 * CO * K ? tid : 0;
    const size_t ig = (size_t)b * N * CO + eg, ia = (size_t)b * N * 2 * C + ea_;
    double rg[10], ra[10];
    rg[0] = a.g_s_out[ig];
    rg[1] = a.g_s_out[plo + ig];
    ra[0] = a.ag0[ia];
    ra[1] = a.ag0[pa + ia];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      rg[2 + m] = a.g_v_out[ig * 4 + m];
      rg[6 + m] = a.g_v_out[plo * 4 + ig * 4 + m];
      ra[2 + m] = a.ag1[ia * 4 + m];
      ra[6 + m] = a.ag1[(pa + ia) * 4 + m];
    }
    const double w0v = a.wm0[ew], w1v = a.wm1[ew];
    load_jet_commit<double, C, DEC>(a.s_in, a.v_in, a.p, a.mask, B, N, b, jr, nd, pj, mk);
    if (tid < N * CO) {
#pragma unroll
      for (int m = 0; m < 10; ++m) go[tid * 10 + m] = rg[m];
    }
    if (tid < N * 2 * C) {
#pragma unroll
      for (int m = 0; m < 10; ++m) agl[tid * 10 + m] = ra[m];
    }
    if (tid < 2 * CO * K) {
      wm[tid] = w0v;
      wm[2 * CO * K + tid] = w1v;
    }
    // remainders (more than one workgroup's worth of items: wide levels, jets of more than 32 particles)
    for (int e = tid + BLK; e < 2 * CO * K; e += BLK) {
      wm[e] = a.wm0[e];
      wm[2 * CO * K + e] = a.wm1[e];
    }
    for (int e = tid + BLK; e < N * CO; e += BLK) {
      const size_t idx = (size_t)b * N * CO + e;
      double* g = go + e * 10;
      g[0] = a.g_s_out[idx];
      g[1] = a.g_s_out[plo + idx];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        g[2 + m] = a.g_v_out[idx * 4 + m];
        g[6 + m] = a.g_v_out[plo * 4 + idx * 4 + m];
      }
    }
    for (int e = tid + BLK; e < N * 2 * C; e += BLK) {
      const size_t ea = (size_t)b * N * 2 * C + e;
      double* x = agl + e * 10;
      x[0] = a.ag0[ea];
      x[1] = a.ag0[pa + ea];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        x[2 + m] = a.ag1[ea * 4 + m];
        x[6 + m] = a.ag1[(pa + ea) * 4 + m];
      }
    }
  }
  __syncthreads();
  STAMP(1);

  // ---------------- phase 1a: per (node, channel) CatMix^H, power backward ---------------------------------
  // waves 0,1: aggregate blocks (cat slots q = 0,1) -> g_ag;  waves 2,3: node + power blocks (q = 2,3,4) -> direct part
  {
    const int half = tid >= BLK / 2;
    for (int e = half ? tid - BLK / 2 : tid; e < N * C; e += BLK / 2) {
      const int n = e / C, c = e - n * C;
      const double* w0r = wm + c;                          // [z][o][k]: + (z * CO + o) * K + q * C
      const double* w1r = wm + 2 * CO * K + c;
      if (half == 0) {
        cx<double> gx0[2] = {{0, 0}, {0, 0}}, gx1[2][4] = {{{0, 0}, {0, 0}, {0, 0}, {0, 0}}, {{0, 0}, {0, 0}, {0, 0}, {0, 0}}};
        for (int o = 0; o < CO; ++o) {
          const double* g = go + (n * CO + o) * 10;
          const cx<double> gs = {g[0], g[1]};
          cx<double> gv[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) gv[m] = {g[2 + m], g[6 + m]};
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const cx<double> w0 = {w0r[(0 * CO + o) * K + q * C], w0r[(1 * CO + o) * K + q * C]};
            const cx<double> w1 = {w1r[(0 * CO + o) * K + q * C], w1r[(1 * CO + o) * K + q * C]};
            cfmac(gx0[q], gs, w0);
#pragma unroll
            for (int m = 0; m < 4; ++m) cfmac(gx1[q][m], gv[m], w1);
          }
        }
        double* gan = ga + n * G::SIZE;
        gan[G::A3 + 2 * c] = gx0[0].r;  gan[G::A3 + 2 * c + 1] = gx0[0].i;
        gan[G::A4 + 2 * c] = gx0[1].r;  gan[G::A4 + 2 * c + 1] = gx0[1].i;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          gan[G::A1 + (c * 4 + m) * 2] = gx1[0][m].r;  gan[G::A1 + (c * 4 + m) * 2 + 1] = gx1[0][m].i;
          gan[G::A2 + (c * 4 + m) * 2] = gx1[1][m].r;  gan[G::A2 + (c * 4 + m) * 2 + 1] = gx1[1][m].i;
        }
      } else {
        // a2/a3/a4: scalar slots q = 2,3,4;  b2: vector slot q = 2;  b34: vector slots 3 and 4 (both multiply v s)
        cx<double> a2 = {0, 0}, a3 = {0, 0}, a4 = {0, 0};
        cx<double> b2[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, b34[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        for (int o = 0; o < CO; ++o) {
          const double* g = go + (n * CO + o) * 10;
          const cx<double> gs = {g[0], g[1]};
          const double* wr0 = w0r + (0 * CO + o) * K;
          const double* wi0 = w0r + (1 * CO + o) * K;
          const double* wr1 = w1r + (0 * CO + o) * K;
          const double* wi1 = w1r + (1 * CO + o) * K;
          cfmac(a2, gs, cx<double>{wr0[2 * C], wi0[2 * C]});
          cfmac(a3, gs, cx<double>{wr0[3 * C], wi0[3 * C]});
          cfmac(a4, gs, cx<double>{wr0[4 * C], wi0[4 * C]});
          const cx<double> w2 = {wr1[2 * C], wi1[2 * C]};
          const cx<double> w34 = {wr1[3 * C] + wr1[4 * C], wi1[3 * C] + wi1[4 * C]};
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const cx<double> gv = {g[2 + m], g[6 + m]};
            cfmac(b2[m], gv, w2);
            cfmac(b34[m], gv, w34);
          }
        }
        const double* ni = nd + n * NS + c * 10;
        const cx<double> s = {ni[0], ni[1]};
        cx<double> v[4], vt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = {ni[2 + m], ni[6 + m]};
        metric_perm(v, vt);
        // node block + power blocks: sq(0,0) = [<v,v>, s^2], sq(1,1) = [v s, s v]
        cx<double> gs = a2;
        cfmac(gs, cx<double>{2.0 * a4.r, 2.0 * a4.i}, s);
        double* gdn = gd + (n * C + c) * 10;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          cfmac(gs, b34[m], v[m]);
          cx<double> gv = b2[m];
          cfmac(gv, b34[m], s);
          cfmac(gv, a3, vt[m]);
          gdn[2 + m] = gv.r;
          gdn[6 + m] = gv.i;
        }
        gdn[0] = gs.r;
        gdn[1] = gs.i;
      }
    }
  }
  STAMP(2);
  // ---------------- phase 1b: CatMix weight gradient over the jet's nodes -> this jet's partial row --------
  // lane = (node part, cat slot k): builds the slot's cat entry x of each of its nodes once and updates all out
  // channels; the parts' sums meet in LDS in a fixed order (deterministic)
  {
    double* part = a.part_mix + prow * (4 * CO * K);
    constexpr int NPART = F::MIXP;
    const int OK = CO * K, nper = (nhi - nlo + NPART - 1) / NPART;
    const int pi = tid / K, k = tid - pi * K;
    cx<double> d0[8], d1[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) d0[o] = d1[o] = {0, 0};
    if (pi < NPART) {
      const int q = k / C, c = k - q * C;
      const int n1 = min(nhi, nlo + (pi + 1) * nper);
      for (int n = nlo + pi * nper; n < n1; ++n) {
        cx<double> x0, x1[4];
        if (q < 2) {                                  // aggregate blocks, saved by the forward
          const double* x = agl + (n * 2 * C + k) * 10;
          x0 = {x[0], x[1]};
#pragma unroll
          for (int m = 0; m < 4; ++m) x1[m] = {x[2 + m], x[6 + m]};
        } else {
          const double* ni = nd + n * NS + c * 10;
          const cx<double> s = {ni[0], ni[1]};
          cx<double> v[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) v[m] = {ni[2 + m], ni[6 + m]};
          if (q == 2) {                               // node block
            x0 = s;
#pragma unroll
            for (int m = 0; m < 4; ++m) x1[m] = v[m];
          } else {                                    // power blocks: (0,0): <v,v> | s^2 ; (1,1): v s | s v
            if (q == 3) { x0 = bil2(v, v); x0.r *= 0.5; x0.i *= 0.5; } else x0 = cmul(s, s);
#pragma unroll
            for (int m = 0; m < 4; ++m) x1[m] = cmul(v[m], s);
          }
        }
#pragma unroll
        for (int o = 0; o < 8; ++o)
          if (o < CO) {
            const double* g = go + (n * CO + o) * 10;
            cfmac(d0[o], cx<double>{g[0], g[1]}, x0);
#pragma unroll
            for (int m = 0; m < 4; ++m) cfmac(d1[o], cx<double>{g[2 + m], g[6 + m]}, x1[m]);
          }
      }
    }
    STAMP(48);
    __syncthreads();                                  // every read of go / wm / agl is done: reuse the region
    STAMP(49);
    double* red = tr;
    if (pi < NPART) {
#pragma unroll
      for (int o = 0; o < 8; ++o)
        if (o < CO) {
          double* r4 = red + ((pi * CO + o) * K + k) * 4;
          r4[0] = d0[o].r;  r4[1] = d0[o].i;  r4[2] = d1[o].r;  r4[3] = d1[o].i;
        }
    }
    STAMP(50);
    __syncthreads();
    STAMP(51);
    // thread = (entry e = o K + k, component x): the parts in index order, as before, but 4 OK threads at work instead of OK
    // (partial row: [x][o][k], x = re / im of the (0,0) weights, re / im of the (1,1) weights)
    for (int ex = tid; ex < 4 * OK; ex += BLK) {
      const int e = ex >> 2, x = ex & 3;
      double v = red[ex];
#pragma unroll
      for (int pp = 1; pp < NPART; ++pp) v += red[pp * OK * 4 + ex];
      part[x * OK + e] = v;
    }
  }

  if constexpr (DEC && SEP) {
    STAMP(3);
    __syncthreads();                                    // g_ag / gd are complete; the phase-1 scratch is dead
    STAMP(4);
    // centre the momenta on the jet mean (only differences enter; see level_fwd2.hip)
    if (tid < 64) {                                       // lane = (node part, component): 8 x 8, parts meet by shuffles
      const int k = tid & 7, part = tid >> 3;
      double mean = 0.0;
      for (int n = part; n < N; n += 8) mean += pj[n * 8 + k];
      mean += shfl_xor(mean, 8);
      mean += shfl_xor(mean, 16);
      mean += shfl_xor(mean, 32);
      if (part == 0) sm[k] = mean / N;
    }
    __syncthreads();
    for (int e = tid; e < N * 8; e += BLK) pj[e] -= sm[e & 7];
    __syncthreads();
    // ---- jet-level sums: wave = channel, lane = node; the 50 per-node terms stay in registers and are summed over the lanes
    //      by transposing butterflies (wave_sum.hpp) -- no LDS round trip, no barrier per round (round 2: three rounds of 20
    //      reals per (node, channel) through LDS, each a barrier, a 30-deep serial sum by 80 threads and another barrier)
    //   sm[c*SMS + ..]: S 0 | VS[m] 2+2m | SP[m] 10+2m | VP 18 | SG4 20 | SG3 22 | SG1[m] 24+2m | SG2[m] 32+2m | GP2 40 | GP3[m] 42+2m
    //   (g_A3 enters with its factor 1/2 everywhere)
#define LGN_PUT(k, val)                                       \
  do {                                                        \
    if ((k) < 16) a0[(k) < 16 ? (k) : 0] += (val);                                        \
    else if ((k) < 32) a1[(k) >= 16 && (k) < 32 ? (k) - 16 : 0] += (val);                 \
    else if ((k) < 44) a2[(k) >= 32 && (k) < 44 ? (k) - 32 : 0] += (val);                 \
    else a3[(k) >= 44 ? (k) - 44 : 0] += (val);                                           \
  } while (0)
    for (int c = wave; c < C; c += NWV) {
      double a0[16], a1[16], a2[12], a3[8];
#pragma unroll
      for (int k = 0; k < 16; ++k) a0[k] = a1[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 12; ++k) a2[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) a3[k] = 0.0;
      for (int n = lane; n < N; n += 64) {
        const double* ni = nd + n * NS + c * 10;
        const double* pn = pj + n * 8;
        const double* gi = ga + n * G::SIZE;
        cx<double> pc[4], v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          pc[m] = {pn[m], pn[4 + m]};
          v[m] = {ni[2 + m], ni[6 + m]};
        }
        const cx<double> sn = {ni[0], ni[1]};
        LGN_PUT(0, sn.r);  LGN_PUT(1, sn.i);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          LGN_PUT(2 + 2 * m, v[m].r);  LGN_PUT(3 + 2 * m, v[m].i);
          const cx<double> sp = cmul(sn, pc[m]);
          LGN_PUT(10 + 2 * m, sp.r);  LGN_PUT(11 + 2 * m, sp.i);
        }
        const cx<double> vp = bil2(v, pc);
        LGN_PUT(18, vp.r);  LGN_PUT(19, vp.i);
        const cx<double> g3 = {0.5 * gi[G::A3 + 2 * c], 0.5 * gi[G::A3 + 2 * c + 1]};
        LGN_PUT(20, gi[G::A4 + 2 * c]);  LGN_PUT(21, gi[G::A4 + 2 * c + 1]);
        LGN_PUT(22, g3.r);  LGN_PUT(23, g3.i);
        cx<double> gp2 = {0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const cx<double> g1 = {gi[G::A1 + (c * 4 + m) * 2], gi[G::A1 + (c * 4 + m) * 2 + 1]};
          const cx<double> g2 = {gi[G::A2 + (c * 4 + m) * 2], gi[G::A2 + (c * 4 + m) * 2 + 1]};
          LGN_PUT(24 + 2 * m, g1.r);  LGN_PUT(25 + 2 * m, g1.i);
          LGN_PUT(32 + 2 * m, g2.r);  LGN_PUT(33 + 2 * m, g2.i);
          cfmac(gp2, g2, pc[m]);
          const cx<double> gp3 = cmulc(g3, pc[m]);
          LGN_PUT(42 + 2 * m, gp3.r);  LGN_PUT(43 + 2 * m, gp3.i);
        }
        LGN_PUT(40, gp2.r);  LGN_PUT(41, gp2.i);
      }
      double* q = sm + c * F::SMS;
      wave_sum_store<16>(a0, q, lane);
      wave_sum_store<16>(a1, q + 16, lane);
      wave_sum_store<12>(a2, q + 32, lane);
      wave_sum_store<8>(a3, q + 44, lane);
    }
#undef LGN_PUT
    __syncthreads();

    STAMP(5);
    // ---- node gradient: neighbour part from the sums + direct part, written once --------------------------------
    const size_t pls = (size_t)B * N * C;
    for (int e = tid; e < N * C; e += BLK) {
      const int n = e / C, c = e - n * C;
      const double* q = sm + c * F::SMS;
      const double* pn = pj + n * 8;
      const double* gdn = gd + e * 10;
      const cx<double> R0 = {a.b0[c], a.b0[c]}, R1 = {a.b1[c], a.b1[c]};
      const cx<double> e0 = {R0.r - R0.i, R0.r + R0.i};
      cx<double> pc[4], pt[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) pc[m] = {pn[m], pn[4 + m]};
      metric_perm(pc, pt);
      const cx<double> SG4 = {q[20], q[21]}, SG3 = {q[22], q[23]}, GP2 = {q[40], q[41]};
      cx<double> GP3[4], GPT3[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) GP3[m] = {q[42 + 2 * m], q[43 + 2 * m]};
      metric_perm(GP3, GPT3);
      // Gs = conj(e0) SG4 + conj(R1) (GP2 - sum_m SG2[m] conj(p[m]))
      cx<double> u = GP2;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const cx<double> t = cmulc(cx<double>{q[32 + 2 * m], q[33 + 2 * m]}, pc[m]);
        u.r -= t.r;  u.i -= t.i;
      }
      cx<double> gs = cmulc(SG4, e0);
      cfmac(gs, u, R1);
      const size_t ge = ((size_t)b * N + n) * C + c;
      a.g_s_in[ge] = gdn[0] + gs.r;
      a.g_s_in[pls + ge] = gdn[1] + gs.i;
      // Gv[m] = conj(e0) SG1[m] + conj(R1) (GPT3[m] - SG3 conj(pt[m]))
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        cx<double> w = GPT3[m];
        const cx<double> t = cmulc(SG3, pt[m]);
        w.r -= t.r;  w.i -= t.i;
        cx<double> gv = cmulc(cx<double>{q[24 + 2 * m], q[25 + 2 * m]}, e0);
        cfmac(gv, w, R1);
        a.g_v_in[ge * 4 + m] = gdn[2 + m] + gv.r;
        a.g_v_in[pls * 4 + ge * 4 + m] = gdn[6 + m] + gv.i;
      }
    }
    STAMP(6);
    // ---- position gradient: d p_n[m] += sum_c conj(R1) [ gA2_n[m] conj(S) + gA3_n conj(VSt[m]) - SG2[m] conj(s_n) - SG3 conj(vt_n[m]) ]
    {
      const size_t plp = (size_t)B * N * 4;
      for (int e = tid; e < N * 4; e += BLK) {
        const int n = e >> 2, m = e & 3;
        const int mp = m == 1 ? 3 : (m == 3 ? 1 : m);      // metric_perm index; component 2 changes sign
        const double sg = m == 2 ? -1.0 : 1.0;
        const double* gi = ga + n * G::SIZE;
        cx<double> acc = {0, 0};
        for (int c = 0; c < C; ++c) {
          const double* q = sm + c * F::SMS;
          const double* ni = nd + n * NS + c * 10;
          const cx<double> R1 = {a.b1[c], a.b1[c]};
          const cx<double> S = {q[0], q[1]}, SG3 = {q[22], q[23]};
          const cx<double> VSt = {sg * q[2 + 2 * mp], sg * q[3 + 2 * mp]};
          const cx<double> vt = {sg * ni[2 + mp], sg * ni[6 + mp]};
          const cx<double> g2 = {gi[G::A2 + (c * 4 + m) * 2], gi[G::A2 + (c * 4 + m) * 2 + 1]};
          const cx<double> g3 = {0.5 * gi[G::A3 + 2 * c], 0.5 * gi[G::A3 + 2 * c + 1]};
          cx<double> t = cmulc(g2, S);
          cfmac(t, g3, VSt);
          const cx<double> t2 = cmulc(cx<double>{q[32 + 2 * m], q[33 + 2 * m]}, cx<double>{ni[0], ni[1]});
          t.r -= t2.r;  t.i -= t2.i;
          const cx<double> t3 = cmulc(SG3, vt);
          t.r -= t3.r;  t.i -= t3.i;
          cfmac(acc, t, R1);
        }
        a.g_p[((size_t)b * N + n) * 4 + m] += acc.r;
        a.g_p[plp + ((size_t)b * N + n) * 4 + m] += acc.i;
      }
    }
    // ---- bias gradients of this jet ----------------------------------------------------------------------------
    if (tid < 2 * C) {
      const int lin = tid / C, c = tid - lin * C;
      const double* q = sm + c * F::SMS;
      const cx<double> S = {q[0], q[1]};
      double* part = a.part_rad + (size_t)blockIdx.x * rad_partial_size(C, true);
      if (lin == 0) {
        cx<double> E0 = cmulc(cx<double>{q[20], q[21]}, S);
#pragma unroll
        for (int m = 0; m < 4; ++m) cfmac(E0, cx<double>{q[24 + 2 * m], q[25 + 2 * m]}, cx<double>{q[2 + 2 * m], q[3 + 2 * m]});
        part[tid] = (E0.r + E0.i) + (E0.i - E0.r);          // R0 = b0 (1+i): d b0 = Re G_R0 + Im G_R0, G_R0 = G_e0 (1-i)
      } else {
        cx<double> VS[4], VSt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) VS[m] = {q[2 + 2 * m], q[3 + 2 * m]};
        metric_perm(VS, VSt);
        cx<double> E1 = cmulc(cx<double>{q[40], q[41]}, S);
        cx<double> neg = cmulc(cx<double>{q[22], q[23]}, cx<double>{q[18], q[19]});
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          cfmac(neg, cx<double>{q[32 + 2 * m], q[33 + 2 * m]}, cx<double>{q[10 + 2 * m], q[11 + 2 * m]});
          cfmac(E1, cx<double>{q[42 + 2 * m], q[43 + 2 * m]}, VSt[m]);
        }
        part[tid] = (E1.r - neg.r) + (E1.i - neg.i);
      }
    }
    STAMP(7);
    return;
  }

  STAMP(3);
  // ---------------- per-lane constants of the pair sweep --------------------------------------------------------
  const int pr = lane & 15, cg = lane >> 4;
  const int tj = pr >> 2, ti = pr & 3;                  // which of the wave's 4 source particles j / slot in the i tile
  double ak[5], bk[5], ck2[5], wf[NG][5], bias[NG][4];
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
      const int rr = lane & 15, q = rr >> 2, ch = 4 * g + (rr & 3);
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
  STAMP(4);
  __syncthreads();                                      // g_ag / gd of the whole jet are in LDS; phase-1 scratch is dead
  STAMP(5);

  // ---------------- phase 2: one sweep over the ordered pairs (i, j), j-centric ------------------------------------
  double* trw = tr + wave * (NG + 3) * 16 * TS;         // encoder: transpose tiles of this wave
  v4d T[NG][3];
  double dB0[NG], dB1[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    dB0[g] = dB1[g] = 0.0;
#pragma unroll
    for (int t = 0; t < 3; ++t) T[g][t] = v4d{0, 0, 0, 0};
  }
  const size_t pls = (size_t)B * N * C;
  double in0[NG][4];                                     // input-stage weight-gradient terms of this lane's nodes (LevelBwdArgs::part_in0)
#pragma unroll
  for (int g = 0; g < NG; ++g) in0[g][0] = in0[g][1] = in0[g][2] = in0[g][3] = 0.0;
  // (Two workgroups share a CU and the SIMD arbiter serves the OLDER wave first: the workgroup that arrived first runs 30 % ahead of
  // the other -- 36 against 46 us at 512 jets, whatever the jets hold -- which then finishes alone.  s_setprio does move the
  // advantage (priority 3 on the younger one swaps the two times exactly), but alternating it per tile only brought the two to
  // 40.5 / 46.5 us and the kernel from 50.2 to 49.4 - 50.1 us: the makespan is the CU's total work, not the order.  Not kept.)
  // Small batches (level_jet_split: a workgroup owns 1 or 2 source groups of the jet): the waves that would idle take a share of the
  // RECEIVER tiles of a group instead -- rs = 4 or 2 waves per group, each sweeps ntiles / rs of them; the partial node gradients meet
  // in LDS below (round 6: at 64 jets the sweep was one wave running eight tiles in a row, 25 k of the workgroup's 43 k cycles).
  const int gcount = ghi - glo;
  const int rs = (!SYM && !DEC && NWV == 4 && gcount >= 1 && gcount <= 2) ? NWV / gcount : 1;      // (workgroup-uniform)
  const int rpart = rs > 1 ? wave % rs : 0;
  const int ntl = (N + 3) >> 2, i0lo = rs > 1 ? 4 * (rpart * ntl / rs) : 0, i0hi = rs > 1 ? min(N, 4 * ((rpart + 1) * ntl / rs)) : N;
  for (int u = 0;; ++u) {
    int rg;
    if constexpr (SYM) {                                   // groups in pairs (p, G - 1 - p): u = 2 k -> p, 2 k + 1 -> its partner
      const int p = wave + NWV * (u >> 1);
      if (p >= (ngroups + 1) >> 1) break;
      rg = (u & 1) ? ngroups - 1 - p : p;
      if ((u & 1) && rg == p) continue;
    } else if (rs > 1) {
      if (u > 0) break;
      rg = glo + wave / rs;
    } else {
      rg = glo + wave + NWV * u;
      if (rg >= ghi) break;
    }
    const int j = rg * 4 + tj;
    const bool jok = j < N;
    const int jj = jok ? j : N - 1;
    double pme[PS];
#pragma unroll
    for (int m = 0; m < PS; ++m) pme[m] = pj[jj * PS + m];
    const bool mj = DEC ? false : (mk[jj] != 0);
    cx<double> Gs[NG], Gv[NG][4], Gq[4];
    cx<double> sj[NG], vj[NG][4], vtj[NG][4];            // own (source) node features
    cx<double> dvj[NG], svj[NG];                         // v_j[3] - v_j[1], v_j[1] + v_j[3]
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      Gs[g] = {0, 0};
#pragma unroll
      for (int m = 0; m < 4; ++m) Gv[g][m] = {0, 0};
      const int ch = 4 * g + cg, cs = ch < C ? ch : 0;
      const double* nj = nd + jj * NS + cs * 10;
      sj[g] = {nj[0], nj[1]};
#pragma unroll
      for (int m = 0; m < 4; ++m) vj[g][m] = {nj[2 + m], nj[6 + m]};
      metric_perm(vj[g], vtj[g]);
      dvj[g] = {vj[g][3].r - vj[g][1].r, vj[g][3].i - vj[g][1].i};
      svj[g] = {vj[g][1].r + vj[g][3].r, vj[g][1].i + vj[g][3].i};
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) Gq[m] = {0, 0};

    for (int i0 = i0lo; i0 < i0hi; i0 += 4) {
      if (rg == glo && i0 < 32) STAMP(16 + (i0 >> 2) * 4);
      // SYM: 2 = receiver group below the source group (this tile also carries the reverse edges' radial gradient), 1 = diagonal
      // tile (both directions are lanes of the tile), 0 = above (radial gradient left to the owner of the other group)
      const int kind = SYM ? ((i0 >> 2) < rg ? 2 : ((i0 >> 2) == rg ? 1 : 0)) : 1;
      const int i = i0 + ti;
      const bool ok = jok && i < N;
      const int ii = i < N ? i : N - 1;
      const double* pii = pj + ii * PS;
      cx<double> q[4];
      v4d R[NG];
      double rho[5], an = 0.0;
      double qd0 = 0.0, qd3 = 0.0, qa = 0.0, qb = 0.0;   // encoder: q = [d0, a - ib, d3, -a - ib] (real momenta)
      bool on = false;
      if (DEC) {
#pragma unroll
        for (int m = 0; m < 4; ++m) q[m] = {pii[m] - pme[m], pii[4 + m] - pme[4 + m]};
#pragma unroll
        for (int g = 0; g < NG; ++g) R[g] = v4d{bias[g][0], bias[g][1], bias[g][2], bias[g][3]};
      } else {
        const double d0 = pii[0] - pme[0], d1 = pii[1] - pme[1], d2 = pii[2] - pme[2], d3 = pii[3] - pme[3];
        const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
        const double nsq = (2.0 * q0 - (((q0 + q1) + q2) + q3)) + 1e-16;
        an = fabs(nsq);
        on = ok && mj && (mk[ii] != 0) && (nsq != 0.0);
        const double h = rsqrt2<double>();
        q[0] = {d0, 0.0};
        q[1] = {d1 * h, -d2 * h};
        q[2] = {d3, 0.0};
        q[3] = {-d1 * h, -d2 * h};
        qd0 = d0;  qd3 = d3;  qa = d1 * h;  qb = d2 * h;
        double beta[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 5; ++s) rho[s] = 0.0;
        if (on) {                                            // (EXEC-masked block: no per-value selects)
#pragma unroll
          for (int s = 0; s < 5; ++s) beta[s] = 1.0 + ck2[s] * an;      // (+ 1e-16 of position_levels.py:146: absorbed, the sum is >= 1)
          rcp5(beta, rho);
#pragma unroll
          for (int s = 0; s < 5; ++s) beta[s] = __builtin_fma(bk[s], rho[s], ak[s]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          R[g] = v4d{bias[g][0], bias[g][1], bias[g][2], bias[g][3]};
#pragma unroll
          for (int s = 0; s < 5; ++s) R[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[g][s], beta[s], R[g], 0, 0, 0);
        }
        // B-operand source of the radial GEMM: this lane's pair (row pr), columns k = 4s + cg
        if (kind != 0) {
          double* xb = trw + NG * 16 * TS;
#pragma unroll
          for (int s = 0; s < 5; ++s) {
            const double x2 = an * rho[s] * rho[s];
            if (s < 4) {
              xb[pr * TS + 4 * s + cg] = rho[s];
              xb[16 * TS + pr * TS + 4 * s + cg] = x2;
            } else {
              xb[32 * TS + pr * TS + cg] = rho[s];
              xb[32 * TS + pr * TS + 4 + cg] = x2;
            }
          }
          xb[32 * TS + pr * TS + 8 + 2 * cg] = cg == 0 ? (on ? 1.0 : 0.0) : 0.0;
          xb[32 * TS + pr * TS + 9 + 2 * cg] = cg == 0 ? (ok ? 1.0 : 0.0) : 0.0;
        }
      }
      if (rg == glo && i0 < 32) STAMP(17 + (i0 >> 2) * 4);
      const double* gi = ga + ii * G::SIZE;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ch = 4 * g + cg;
        double G0r = 0, G0i = 0, G1r = 0, G1i = 0;
        if (ok && ch < C) {
          const cx<double> R0 = {R[g][0], R[g][1]}, R1 = {R[g][2], R[g][3]};
          const cx<double> e0 = {R0.r - R0.i, R0.r + R0.i};
          const cx<double> gA3 = {0.5 * gi[G::A3 + 2 * ch], 0.5 * gi[G::A3 + 2 * ch + 1]};
          const cx<double> gA4 = {gi[G::A4 + 2 * ch], gi[G::A4 + 2 * ch + 1]};
          cx<double> gA1[4], gA2[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            gA1[m] = {gi[G::A1 + (ch * 4 + m) * 2], gi[G::A1 + (ch * 4 + m) * 2 + 1]};
            gA2[m] = {gi[G::A2 + (ch * 4 + m) * 2], gi[G::A2 + (ch * 4 + m) * 2 + 1]};
          }
          cfmac(Gs[g], gA4, e0);
          cx<double> ge0 = cmulc(gA4, sj[g]);
          cx<double> gR1;
          if (!DEC) {
            // The edge e1[m] = R1 q[m] enters only through P2 = sum_m gA2[m] conj(q[m]), V = <v_j, q> and Z = gA3 conj(R1);
            // with real momenta q = [d0, a - ib, d3, -a - ib] these cost 8 + 8 + 12 flops instead of four complex products each.
            const cx<double> dg = {gA2[1].r - gA2[3].r, gA2[1].i - gA2[3].i}, sg = {gA2[1].r + gA2[3].r, gA2[1].i + gA2[3].i};
            cx<double> P2;
            P2.r = __builtin_fma(gA2[0].r, qd0, __builtin_fma(gA2[2].r, qd3, __builtin_fma(qa, dg.r, -qb * sg.i)));
            P2.i = __builtin_fma(gA2[0].i, qd0, __builtin_fma(gA2[2].i, qd3, __builtin_fma(qa, dg.i, qb * sg.r)));
            cfmac(Gs[g], P2, R1);                              // sum_m gA2[m] conj(e1[m]) = conj(R1) P2
            const cx<double> Z = cmulc(gA3, R1);              // gA3 conj(e1t[m]) = Z conj(qt[m])
#pragma unroll
            for (int m = 0; m < 4; ++m) cfmac(Gv[g][m], gA1[m], e0);
            Gv[g][0].r = __builtin_fma(Z.r, qd0, Gv[g][0].r);   Gv[g][0].i = __builtin_fma(Z.i, qd0, Gv[g][0].i);
            Gv[g][2].r = __builtin_fma(-Z.r, qd3, Gv[g][2].r);  Gv[g][2].i = __builtin_fma(-Z.i, qd3, Gv[g][2].i);
            const double aZr = qa * Z.r, aZi = qa * Z.i;
            const double bZr = qb * Z.r, bZi = qb * Z.i;
            Gv[g][1].r -= aZr + bZi;  Gv[g][1].i += bZr - aZi;   // Z (-a + ib)
            Gv[g][3].r += aZr - bZi;  Gv[g][3].i += aZi + bZr;   // Z ( a + ib)
            gR1 = {0, 0};
            if (kind != 0) {                                   // (b) gradient w.r.t. the radial values of this pair
#pragma unroll
              for (int m = 0; m < 4; ++m) cfmac(ge0, gA1[m], vj[g][m]);
              // V = <v_j, q> = v0 d0 - v2 d3 + a (v3 - v1) - ib (v1 + v3)
              cx<double> V;
              V.r = __builtin_fma(vj[g][0].r, qd0, __builtin_fma(-vj[g][2].r, qd3, __builtin_fma(qa, dvj[g].r, qb * svj[g].i)));
              V.i = __builtin_fma(vj[g][0].i, qd0, __builtin_fma(-vj[g][2].i, qd3, __builtin_fma(qa, dvj[g].i, -qb * svj[g].r)));
              gR1 = cmulc(P2, sj[g]);                          // sum_m ge1[m] conj(q[m]) = conj(s_j) P2 + gA3 conj(V)
              cfmac(gR1, gA3, V);
            }
            if constexpr (SYM) {
              if (kind == 2) {
                // the reverse edge (receiver j, source i, momentum difference -q): the same two gradients with the roles swapped --
                // the upstream gradient of j's aggregate, i's features; P2 and V change sign with q
                const double* gj = ga + jj * G::SIZE;
                const double* ni = nd + ii * NS + ch * 10;
                const cx<double> si = {ni[0], ni[1]};
                cx<double> vi[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) vi[m] = {ni[2 + m], ni[6 + m]};
                const cx<double> hA3 = {0.5 * gj[G::A3 + 2 * ch], 0.5 * gj[G::A3 + 2 * ch + 1]};
                const cx<double> hA4 = {gj[G::A4 + 2 * ch], gj[G::A4 + 2 * ch + 1]};
                cx<double> he0 = cmulc(hA4, si);
                cx<double> hA2[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                  const cx<double> hA1 = {gj[G::A1 + (ch * 4 + m) * 2], gj[G::A1 + (ch * 4 + m) * 2 + 1]};
                  hA2[m] = {gj[G::A2 + (ch * 4 + m) * 2], gj[G::A2 + (ch * 4 + m) * 2 + 1]};
                  cfmac(he0, hA1, vi[m]);
                }
                const cx<double> hd = {hA2[1].r - hA2[3].r, hA2[1].i - hA2[3].i}, hs = {hA2[1].r + hA2[3].r, hA2[1].i + hA2[3].i};
                cx<double> Q2;                                 // sum_m hA2[m] conj(q[m]) (the reverse edge's is its negative)
                Q2.r = __builtin_fma(hA2[0].r, qd0, __builtin_fma(hA2[2].r, qd3, __builtin_fma(qa, hd.r, -qb * hs.i)));
                Q2.i = __builtin_fma(hA2[0].i, qd0, __builtin_fma(hA2[2].i, qd3, __builtin_fma(qa, hd.i, qb * hs.r)));
                const cx<double> dvi = {vi[3].r - vi[1].r, vi[3].i - vi[1].i}, svi = {vi[1].r + vi[3].r, vi[1].i + vi[3].i};
                cx<double> W;                                  // <v_i, q>
                W.r = __builtin_fma(vi[0].r, qd0, __builtin_fma(-vi[2].r, qd3, __builtin_fma(qa, dvi.r, qb * svi.i)));
                W.i = __builtin_fma(vi[0].i, qd0, __builtin_fma(-vi[2].i, qd3, __builtin_fma(qa, dvi.i, -qb * svi.r)));
                cx<double> hR1 = cmulc(Q2, si);
                cfmac(hR1, hA3, W);
                ge0.r += he0.r;  ge0.i += he0.i;
                gR1.r -= hR1.r;  gR1.i -= hR1.i;
              }
            }
          } else {
            cx<double> e1[4], e1t[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) e1[m] = cmul(R1, q[m]);
            metric_perm(e1, e1t);
            gR1 = {0, 0};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              // (a) gradient w.r.t. the source node j
              cfmac(Gv[g][m], gA1[m], e0);
              cfmac(Gv[g][m], gA3, e1t[m]);
              cfmac(Gs[g], gA2[m], e1[m]);
              // (b) gradient w.r.t. the edge of this pair
              cfmac(ge0, gA1[m], vj[g][m]);
              cx<double> ge1 = cmulc(gA2[m], sj[g]);
              cfmac(ge1, gA3, vtj[g][m]);
              cfmac(gR1, ge1, q[m]);
              cfmac(Gq[m], ge1, R1);
            }
          }
          G0r = ge0.r + ge0.i;  G0i = ge0.i - ge0.r;        // e0 = R0 (1+i)  ->  G_R0 = G_e0 (1-i)
          G1r = gR1.r;  G1i = gR1.i;
        }
        if (DEC) {
          dB0[g] += G0r + G0i;                              // R0 = b0 (1+i): d b0 = Re G_R0 + Im G_R0
          dB1[g] += G1r + G1i;
        } else if (kind != 0) {
          double* ta = trw + g * 16 * TS;                   // [pair][r' = cg + 4q]
          ta[pr * TS + cg] = G0r;
          ta[pr * TS + 4 + cg] = G0i;
          ta[pr * TS + 8 + cg] = G1r;
          ta[pr * TS + 12 + cg] = G1i;
        }
      }
      if (rg == glo && i0 < 32) STAMP(18 + (i0 >> 2) * 4);
      if (!DEC && kind != 0) {
        wave_sync();
        const double* xb = trw + NG * 16 * TS;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int prow = 4 * s + cg;
          double bv[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) bv[t] = xb[t * 16 * TS + prow * TS + pr];
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const double av = trw[g * 16 * TS + prow * TS + pr];
#pragma unroll
            for (int t = 0; t < 3; ++t) T[g][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[t], T[g][t], 0, 0, 0);
          }
        }
        wave_sync();
      }
    }

    if (rg == glo) STAMP(6);
    if (rs > 1) {
      // receiver parts of one source group: parts 1 .. rs - 1 leave their quad sums in their own transpose tile (free after the
      // sweep), part 0 adds them in part order and finishes the group alone (the final sums of a lane sit where Gs / Gv were)
      double* mine = trw + lane;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        Gs[g].r = quad_sum(Gs[g].r);  Gs[g].i = quad_sum(Gs[g].i);
#pragma unroll
        for (int m = 0; m < 4; ++m) { Gv[g][m].r = quad_sum(Gv[g][m].r);  Gv[g][m].i = quad_sum(Gv[g][m].i); }
        if (rpart != 0) {
          mine[(g * 10 + 0) * 64] = Gs[g].r;  mine[(g * 10 + 1) * 64] = Gs[g].i;
#pragma unroll
          for (int m = 0; m < 4; ++m) { mine[(g * 10 + 2 + m) * 64] = Gv[g][m].r;  mine[(g * 10 + 6 + m) * 64] = Gv[g][m].i; }
        }
      }
      __syncthreads();                                       // (rs > 1 is workgroup-uniform and every wave runs this iteration)
      if (rpart != 0) break;
      for (int pp = 1; pp < rs; ++pp) {
        const double* oth = trw + pp * (NG + 3) * 16 * TS + lane;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          Gs[g].r += oth[(g * 10 + 0) * 64];  Gs[g].i += oth[(g * 10 + 1) * 64];
#pragma unroll
          for (int m = 0; m < 4; ++m) { Gv[g][m].r += oth[(g * 10 + 2 + m) * 64];  Gv[g][m].i += oth[(g * 10 + 6 + m) * 64]; }
        }
      }
    }
    // node gradient of the wave's 4 particles: neighbour part (quad sum over the i slots) + direct part, written once
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int ch = 4 * g + cg;
      const double sr = rs > 1 ? Gs[g].r : quad_sum(Gs[g].r), si = rs > 1 ? Gs[g].i : quad_sum(Gs[g].i);
      const bool wr = jok && ti == 0 && ch < C;
      const size_t e = ((size_t)b * N + jj) * C + (ch < C ? ch : 0);
      const double* gdn = gd + (jj * C + (ch < C ? ch : 0)) * 10;
      const double gsr = gdn[0] + sr, gsi = gdn[1] + si;
      if (wr) {
        a.g_s_in[e] = gsr;
        a.g_s_in[pls + e] = gsi;
      }
      cx<double> gvn[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double vr = rs > 1 ? Gv[g][m].r : quad_sum(Gv[g][m].r), vi = rs > 1 ? Gv[g][m].i : quad_sum(Gv[g][m].i);
        gvn[m] = {gdn[2 + m] + vr, gdn[6 + m] + vi};
        if (wr) {
          a.g_v_in[e * 4 + m] = gvn[m].r;
          a.g_v_in[pls * 4 + e * 4 + m] = gvn[m].i;
        }
      }
      if constexpr (!DEC) {
        if (a.part_in0 && wr) {                              // (wave-uniform pointer test; the terms of node jj, channel ch)
          // 2 E^2 - sum p^2 left to right, canonical momenta: the arithmetic of enc_input_bwd_kernel (net_kernels.hip)
          const double q0 = pme[0] * pme[0], q1 = pme[1] * pme[1], q2 = pme[2] * pme[2], q3 = pme[3] * pme[3];
          const double mass = sqrt(fabs(2.0 * q0 - (((q0 + q1) + q2) + q3)));
          constexpr double H = 0.70710678118654752440084436210484903928;
          const cx<double> q[4] = {{pme[0], 0.0}, {pme[1] * H, -pme[2] * H}, {pme[3], 0.0}, {-pme[1] * H, -pme[2] * H}};
          cx<double> d1 = {0, 0};
#pragma unroll
          for (int m = 0; m < 4; ++m) cfmac(d1, gvn[m], q[m]);
          in0[g][0] += gsr * mass;
          in0[g][1] += gsi * mass;
          in0[g][2] += d1.r;
          in0[g][3] += d1.i;
        }
      }
    }
    if (DEC) {
      const size_t plp = (size_t)B * N * 4;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double qr = quad_sum(Gq[m].r), qi = quad_sum(Gq[m].i);
        qr += shfl_xor(qr, 16);  qr += shfl_xor(qr, 32);
        qi += shfl_xor(qi, 16);  qi += shfl_xor(qi, 32);
        if (jok && ti == 0 && cg == 0) {
          a.g_p[((size_t)b * N + jj) * 4 + m] -= qr;
          a.g_p[plp + ((size_t)b * N + jj) * 4 + m] -= qi;
        }
      }
    }
  }

  STAMP(7);
  // ---------------- phase 3 (decoder): i-centric sweep for d p_i = sum_j G_q(i, j) ---------------------------------
  if (DEC) {
    const int ti2 = pr >> 2, tj2 = pr & 3;
    const size_t plp = (size_t)B * N * 4;
    for (int rg = glo + wave; rg < ghi; rg += NWV) {
      const int i = rg * 4 + ti2;
      const bool iok = i < N;
      const int ii = iok ? i : N - 1;
      const double* gi = ga + ii * G::SIZE;
      cx<double> Gq[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
      for (int j0 = 0; j0 < N; j0 += 4) {
        const int j = j0 + tj2;
        if (iok && j < N) {
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int ch = 4 * g + cg;
            if (ch < C) {
              const double* nj = nd + j * NS + ch * 10;
              const cx<double> s = {nj[0], nj[1]};
              cx<double> v[4], vt[4];
#pragma unroll
              for (int m = 0; m < 4; ++m) v[m] = {nj[2 + m], nj[6 + m]};
              metric_perm(v, vt);
              const cx<double> R1 = {bias[g][2], bias[g][3]};
              const cx<double> gA3 = {0.5 * gi[G::A3 + 2 * ch], 0.5 * gi[G::A3 + 2 * ch + 1]};
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                const cx<double> gA2 = {gi[G::A2 + (ch * 4 + m) * 2], gi[G::A2 + (ch * 4 + m) * 2 + 1]};
                cx<double> ge1 = cmulc(gA2, s);
                cfmac(ge1, gA3, vt[m]);
                cfmac(Gq[m], ge1, R1);
              }
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double qr = quad_sum(Gq[m].r), qi = quad_sum(Gq[m].i);
        qr += shfl_xor(qr, 16);  qr += shfl_xor(qr, 32);
        qi += shfl_xor(qi, 16);  qi += shfl_xor(qi, 32);
        if (iok && tj2 == 0 && cg == 0) {
          a.g_p[((size_t)b * N + i) * 4 + m] += qr;
          a.g_p[plp + ((size_t)b * N + i) * 4 + m] += qi;
        }
      }
    }
  }

  // ---------------- input-stage partial row (first encoder level of a fused network) -----------------------------
  __shared__ double in0l[NWV][8][4];
  if constexpr (!DEC) {
    if (a.part_in0) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double x = in0[g][k];                              // non-zero on the lanes (ti == 0, tj = pr >> 2) only: sum over the 4 nodes of a group
          x += shfl_xor(x, 4);
          x += shfl_xor(x, 8);
          if (pr == 0 && 4 * g + cg < C) in0l[wave][4 * g + cg][k] = x;
        }
    }
  }

  // ---------------- radial partial row of this jet ------------------------------------------------------------
  STAMP(8);
  __syncthreads();
  STAMP(9);
  if constexpr (!DEC) {
    if (a.part_in0 && tid < 4 * C) {
      const int k = tid / C, c = tid - k * C;
      double s = 0;
#pragma unroll
      for (int w = 0; w < NWV; ++w) s += in0l[w][c][k];
      a.part_in0[prow * 4 * C + k * C + c] = s;
    }
  }
  double* part = a.part_rad + prow * rad_partial_size(C, DEC);
  if (DEC) {
    double* red = tr;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      double x0 = dB0[g], x1 = dB1[g];
      for (int m = 1; m < 16; m <<= 1) { x0 += shfl_xor(x0, m); x1 += shfl_xor(x1, m); }
      if (pr == 0) {
        red[(wave * NG + g) * 8 + cg] = x0;
        red[(wave * NG + g) * 8 + 4 + cg] = x1;
      }
    }
    __syncthreads();
    if (tid < 2 * C) {
      const int lin = tid / C, ch = tid - lin * C, g = ch >> 2, c4 = ch & 3;
      double s = 0;
      for (int w = 0; w < NWV; ++w) s += red[(w * NG + g) * 8 + lin * 4 + c4];
      part[tid] = s;
    }
  } else {
    double* red = tr;
    {
      double* mine = red + (size_t)(wave * 64 + lane) * NG * 12;
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) mine[(g * 3 + t) * 4 + q] = T[g][t][q];
    }
    __syncthreads();
    // the 12 NG sums of a lane position are dealt to the first four waves by q (every wave reads LDS only: one wave doing all of them
    // was 4 000 cycles of the kernel's tail with the other waves idle)
    const int wq = __builtin_amdgcn_readfirstlane(wave);
    if (wq < 4) {
      constexpr int R = 4 * C;
      const int col = lane & 15;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ch = 4 * g + cg;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (q != wq) continue;
            const int e = (g * 3 + t) * 4 + q;
            double v = (red[(size_t)(0 * 64 + lane) * NG * 12 + e] + red[(size_t)(1 * 64 + lane) * NG * 12 + e]) +
                       (red[(size_t)(2 * 64 + lane) * NG * 12 + e] + red[(size_t)(3 * 64 + lane) * NG * 12 + e]);
            if (NWV == 8)
              v += (red[(size_t)(4 * 64 + lane) * NG * 12 + e] + red[(size_t)(5 * 64 + lane) * NG * 12 + e]) +
                   (red[(size_t)(6 * 64 + lane) * NG * 12 + e] + red[(size_t)(7 * 64 + lane) * NG * 12 + e]);
            if (ch >= C) continue;
            const int r = (q >> 1) * 2 * C + 2 * ch + (q & 1);
            if (t == 0) part[r * NB + col] = v;
            else if (t == 1) part[R * NB + r * NB + col] = v;
            else {
              if (col < 4) part[r * NB + 16 + col] = v;
              else if (col < 8) part[R * NB + r * NB + 16 + (col - 4)] = v;
              else if (col == 8) part[2 * R * NB + r] = v;
              else if (col == 9) part[2 * R * NB + R + r] = v;
            }
          }
      }
    }
  }
  STAMP(10);
}

bool level_bwd3_fits(int N) { return N <= 40; }

template <int C, bool DEC, bool SEP, int NWV>
static int launch_bwd3_w(const LevelBwdArgs<double>& a, int split, hipStream_t stream) {
  size_t smem = Bwd3<C, DEC, NWV>::smem(a.N, a.CO);
  LGN_CHECK_ARG(smem <= 160 * 1024, "level_bwd: N=%d C=%d needs %zu B of LDS", a.N, a.C, smem);
  LGN_CHECK_ARG(a.CO <= 8, "level_bwd: C_out=%d unsupported (1..8)", a.CO);
  auto kern = level_bwd3_kernel<C, DEC, SEP, NWV>;
  if constexpr (!DEC && C <= 4) {       // whole jets per workgroup: the sweep that uses R(i, j) = R(j, i) (LVL_BWD_ORDERED: the plain one;
                                        // C > 4: two lane groups of channels, the second pass would double the spills)
    if (split == 1 && !(a.flags & LVL_BWD_ORDERED)) kern = level_bwd3_kernel<C, DEC, SEP, NWV, true>;
  }
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3(a.B, split), dim3(64 * NWV), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}
template <int C, bool DEC, bool SEP>
static int launch_bwd3(const LevelBwdArgs<double>& a, hipStream_t stream) {
  // small batches: several workgroups per jet (level.hpp: level_jet_split); the separable decoder form has no sweep to split
  return launch_bwd3_w<C, DEC, SEP, 4>(a, SEP ? 1 : level_jet_split(a.B, a.N), stream);
}

// whole level backward in one launch; one CatMix partial row and one radial partial row per jet
int level_bwd3_dispatch(const LevelBwdArgs<double>& a, int decoder, hipStream_t stream) {
  const bool pairwise = (a.flags & LVL_DEC_PAIRWISE) != 0;   // the decoder on the O(N^2) pair sweep (cross-check of the separable form)
#define LGN_CASE(CC)                                                                \
  case CC:                                                                          \
    if (!decoder) return launch_bwd3<CC, false, false>(a, stream);                  \
    return pairwise ? launch_bwd3<CC, true, false>(a, stream) : launch_bwd3<CC, true, true>(a, stream);
  switch (a.C) {
#ifdef LGN_DEV_ONLY_C4      // development builds: one channel count (compile time)
    LGN_CASE(4)
#else
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
#endif
    default: set_error("level_bwd: C_in=%d unsupported (1..8)", a.C); return -1;
  }
#undef LGN_CASE
}

}  // namespace lgn
