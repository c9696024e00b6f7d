// lgn-autoencoder_amd/csrc/generic_local_sep.hip -- per-node part of a table-driven DECODER level with the separable moments kept
// on chip (round 6).  Same walk, same compile-time tables and same arithmetic per term as generic_local_static.hip; what changes
// is where the neighbour moments U and their gradient dU live.
//
// The decoder's edge mask is identically zero (lgn/models/lgn_decoder.py:335-340), so its moments are separable
// (generic_moments_sep.hip):  U[i][q][0] = e0 SX[q],  U[i][q][1 + m] = R1 (P_i[m] SX[q] - SXP[q][m])  with jet-level sums SX, SXP.
// Round 5 wrote U (147 MB per level at cfg5) with one kernel and read it with the next, and likewise dU on the way back: 4 x 147 MB
// per level for numbers that are a few flops away from a 4 KB table per (jet, channel).  Here
//   forward   dec_sep_tab (generic_moments_sep.hip) leaves the table; local_fwd_static<.., SEP> forms U[i][q][k] where a term reads it;
//   backward  ONE kernel per level (was local_bwd_static + dec_sep_bwd_tb): workgroup = (PAIR OF JETS, channel), lane = (jet of the
//             pair, particle) -- jets are the unit every reduction of the separable backward runs over, so they are aligned to the
//             two 32-lane halves of a wave (N <= 32; the tile-blocked tensors are addressed through the lane's node index, a jet's
//             particles stay a coalesced run).  dU never exists as a tensor: a block's contributions gu are, while still in
//             registers, (1) contracted with conj(A[q]) into the lane's d p, (2) summed over the jet -- S[q][k] = sum_i dU,
//             SP[q][m] = sum_i dU conj(P_i[m]) -- by a half-wave butterfly, 8 values at a time, the totals parked in LDS in the
//             order they are produced.  After the walk every lane adds the aggregate part of d X, the second part of d p and the
//             bias-gradient terms from those sums (what dec_sep_bwd_tb did from a 147 MB dU).
// Reference: cg_product aggregate lgn/cg_lib/cg_ops.py:281-297 (sum over j BEFORE the CG matrix: the order kept here), the radial
// functions of the decoder lgn/nn/position_levels.py:144-149 (masked edges carry the Linear bias).
#include "local_static_dev.hpp"
#include <algorithm>

namespace lgn {
namespace {
using namespace lsd;
LGN_STAMP_DECL
#ifdef LGN_STAMPS
// (first workgroup of the Q = 20 levels: the heavy ones -- -DLGN_PSTAMP_Q=5: the first level; wave 0 stamps 0.., wave 1 stamps 16..)
#ifndef LGN_PSTAMP_Q
#define LGN_PSTAMP_Q 20
#endif
#define PSTAMP(i) do { if (T::Q == LGN_PSTAMP_Q && (threadIdx.x & 63) == 0 && blockIdx.x == 0) g_stamps[i] = clock64(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

// sums of 8 per-lane values over each 32-lane half of the wave (wave_sum.hpp's butterfly without its cross-half stage): the total of
// value half_sum8_k(lane) comes back in every lane of a quad
__device__ __forceinline__ double half_sum8(const double (&v)[8], int lane) {
  double t[4], u[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = swap_add<false>(v[2 * q], v[2 * q + 1]);      // even / odd row of the half: value 2q / 2q + 1
  const int i = lane & 15;
#pragma unroll
  for (int j = 0; j < 2; ++j) u[j] = dpp_pair_add<0x140>(t[2 * j], t[2 * j + 1], i < 8);
  double w = dpp_pair_add<0x141>(u[0], u[1], (i & 4) == 0);
  w = dpp_add<0xB1>(w);
  return dpp_add<0x4E>(w);
}
__device__ __forceinline__ int half_sum8_k(int lane) {
  const int i = lane & 15, rho = lane >> 4;
  return 2 * (2 * ((i >> 2) & 1) + (i >> 3)) + (rho & 1);
}

// ---- compile-time layout of the parked sums: block bid (walk order) owns slots [sep_vbase(bid), + 8 ceil(nv / 8)); its moment k
// sits at sep_vpos(bid, k): (gu.r, gu.i) and, for a vector moment (e % 5 >= 1), (gu conj P).r, .i behind it
template <class T> constexpr int sep_nblocks() { return T::BLK0[T::N_OUT - 1] + T::NBLK[T::N_OUT - 1]; }
template <class T> constexpr int sep_nv(int bid) {
  int n = 0;
  for (int p = T::BLK_UPTR[bid]; p < T::BLK_UPTR[bid + 1]; ++p) n += (T::BLK_UELEM[p] % 5) ? 4 : 2;
  return n;
}
template <class T> constexpr int sep_vbase(int bid) {
  int s = 0;
  for (int b = 0; b < bid; ++b) s += (sep_nv<T>(b) + 7) / 8 * 8;
  return s;
}
template <class T> constexpr int sep_vpos(int bid, int k) {
  int n = 0;
  for (int p = T::BLK_UPTR[bid]; p < T::BLK_UPTR[bid] + k; ++p) n += (T::BLK_UELEM[p] % 5) ? 4 : 2;
  return n;
}
template <class T> constexpr int sep_vtotal() { return sep_vbase<T>(sep_nblocks<T>()); }

// What wave 0 needs through the walk in the separable form lives in LDS, not in registers (the walk itself holds ~500 of the 512
// registers of a lone wave per SIMD): the two jets' table rows E | A | B_0..3 (tab: [half][Q][12], broadcast reads), the lanes'
// momenta (pl: [lane][8]), the lanes' d p accumulators (gpl: [lane][8], flushed once per block like d X) and the parked sums.
struct SepWalk {
  const double* tab;       // LDS: table rows of this lane's half
  const double* pl;        // LDS: this lane's P[m] (re, im)
  double* gpl;             // LDS: this lane's d p, first part: sum_q dU[q][1 + m] conj(A[q])
  double* sraw;            // LDS: parked sums of this lane's half, already offset by half_sum8_k(lane)
  bool owner;              // first lane of its quad: stores the quad's total
};
__device__ __forceinline__ cx<double> lds_cx(const double* p) {
  const sep_d2 v = *reinterpret_cast<const sep_d2*>(p);
  return {v.x, v.y};
}
__device__ __forceinline__ void lds_cx_add(double* p, const cx<double>& g) {
  sep_d2 v = *reinterpret_cast<sep_d2*>(p);
  v.x += g.r;
  v.y += g.i;
  *reinterpret_cast<sep_d2*>(p) = v;
}
// moment e = 5 q + k of the lane's node from the LDS copies
__device__ __forceinline__ cx<double> sepw_u(const SepWalk& w, int e) {
  const int q = e / 5, k = e % 5;
  if (k == 0) return lds_cx(w.tab + q * 12);
  const cx<double> A = lds_cx(w.tab + q * 12 + 2), Bm = lds_cx(w.tab + q * 12 + 2 + 2 * k);
  cx<double> r = {-Bm.r, -Bm.i};
  cfma(r, lds_cx(w.pl + 128 * (k - 1)), A);
  return r;
}

struct UvLds {
  const SepWalk& w;
  // (no masking of idle lanes: their upstream gradient is zero, so whatever moment they form reaches nothing)
  __device__ __forceinline__ cx<double> get(int e, int /*k*/) const { return sepw_u(w, e); }
};

// absolute slot of entry p of BLK_UELEM (p counts over all blocks in walk order)
template <class T> constexpr int sep_bid_of(int p) {
  int bid = 0;
  while (T::BLK_UPTR[bid + 1] <= p) ++bid;
  return bid;
}
template <class T> constexpr int sep_at(int p) { return sep_vbase<T>(sep_bid_of<T>(p)) + sep_vpos<T>(sep_bid_of<T>(p), p - T::BLK_UPTR[sep_bid_of<T>(p)]); }

// does block (L, BLK) read a vector moment with component m?
template <class T> constexpr bool sep_blk_has_m(int l, int blk, int m) {
  for (int p = blk_u0<T>(l, blk); p < blk_u0<T>(l, blk) + blk_nu<T>(l, blk); ++p)
    if (T::BLK_UELEM[p] % 5 == m + 1) return true;
  return false;
}
// moment of block bid whose values cover position pos of the block's value list (-1: padding behind the last one)
template <class T> constexpr int sep_k_at(int bid, int pos) {
  int at = 0;
  for (int p = T::BLK_UPTR[bid]; p < T::BLK_UPTR[bid + 1]; ++p) {
    const int n = (T::BLK_UELEM[p] % 5) ? 4 : 2;
    if (pos < at + n) return p - T::BLK_UPTR[bid];
    at += n;
  }
  return -1;
}
// value POS of block (L, BLK): (gu.r, gu.i) of a moment and, for a vector moment, (gu conj P).r, .i -- formed when a batch needs it
template <class T, int L, int BLK, int POS, int NU1>
__device__ __forceinline__ double sep_val(const SepWalk& w, const cx<double> (&gu)[NU1]) {
  constexpr int BID = T::BLK0[L] + BLK, K = sep_k_at<T>(BID, POS);
  if constexpr (K < 0) return 0.0;
  else {
    constexpr int comp = POS - sep_vpos<T>(BID, K), e = T::BLK_UELEM[blk_u0<T>(L, BLK) + K];
    if constexpr (comp == 0) return gu[K].r;
    else if constexpr (comp == 1) return gu[K].i;
    else {
      const cx<double> sp = cmulc(gu[K], lds_cx(w.pl + 128 * (e % 5 - 1)));
      return comp == 2 ? sp.r : sp.i;
    }
  }
}
template <class T, int L, int BLK, int B8, int NB8, int NU1>
__device__ __forceinline__ void sep_batches(const SepWalk& w, const cx<double> (&gu)[NU1], int lane) {
  if constexpr (B8 < NB8) {
    const double v8[8] = {sep_val<T, L, BLK, 8 * B8 + 0, NU1>(w, gu), sep_val<T, L, BLK, 8 * B8 + 1, NU1>(w, gu),
                          sep_val<T, L, BLK, 8 * B8 + 2, NU1>(w, gu), sep_val<T, L, BLK, 8 * B8 + 3, NU1>(w, gu),
                          sep_val<T, L, BLK, 8 * B8 + 4, NU1>(w, gu), sep_val<T, L, BLK, 8 * B8 + 5, NU1>(w, gu),
                          sep_val<T, L, BLK, 8 * B8 + 6, NU1>(w, gu), sep_val<T, L, BLK, 8 * B8 + 7, NU1>(w, gu)};
    const double tot = half_sum8(v8, lane);
    constexpr int SLOT = sep_vbase<T>(T::BLK0[L] + BLK) + 8 * B8;      // (constexpr variable: evaluated by the compiler, not by the wave)
    if (w.owner) w.sraw[SLOT] = tot;
    __builtin_amdgcn_sched_barrier(0);        // one batch at a time: the next one's values stay where they are until then
    sep_batches<T, L, BLK, B8 + 1, NB8, NU1>(w, gu, lane);
  }
}
// vector moment K of the block: its share of the lane's d p
template <class T, int L, int BLK, int K, int NU1>
__device__ __forceinline__ void sep_gp(const SepWalk& w, const cx<double> (&gu)[NU1], cx<double> (&gpb)[4]) {
  if constexpr (K < blk_nu<T>(L, BLK)) {
    constexpr int e = T::BLK_UELEM[blk_u0<T>(L, BLK) + K];
    if constexpr (e % 5 != 0) cfmac(gpb[e % 5 - 1], gu[K], lds_cx(w.tab + (e / 5) * 12 + 2));
    sep_gp<T, L, BLK, K + 1, NU1>(w, gu, gpb);
  }
}
// the block's moment gradients leave the registers: d p, and the jet sums S / SP
template <class T, int L, int BLK, int NU1>
__device__ __forceinline__ void sep_reduce(const SepWalk& w, const cx<double> (&gu)[NU1], int lane) {
  constexpr int BID = T::BLK0[L] + BLK, NU = blk_nu<T>(L, BLK), NV = sep_nv<T>(BID), NB8 = (NV + 7) / 8;
  if constexpr (NU > 0) {
    cx<double> gpb[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    sep_gp<T, L, BLK, 0, NU1>(w, gu, gpb);
    // the lane's d p: plain read-modify-write of lane-private slots, once per block
    if constexpr (sep_blk_has_m<T>(L, BLK, 0)) lds_cx_add(w.gpl + 0, gpb[0]);
    if constexpr (sep_blk_has_m<T>(L, BLK, 1)) lds_cx_add(w.gpl + 128, gpb[1]);
    if constexpr (sep_blk_has_m<T>(L, BLK, 2)) lds_cx_add(w.gpl + 256, gpb[2]);
    if constexpr (sep_blk_has_m<T>(L, BLK, 3)) lds_cx_add(w.gpl + 384, gpb[3]);
    __builtin_amdgcn_sched_barrier(0);
    sep_batches<T, L, BLK, 0, NB8, NU1>(w, gu, lane);
  }
}

template <class T, int L, int BLK, int BEND, int COT>
__device__ __forceinline__ void blocks_bwd_sep(const cx<double> (&go)[COT][T::DIM[L]], cx<double> (&wn)[COT],
                                               const double __attribute__((address_space(4))) * wc, const SepWalk& w,
                                               const double* xl, double* gxl, const DwOut& part, int lane, bool valid) {
  constexpr int D = T::DIM[L], ROWB = T::ROW0[L] + BLK * D;
  if constexpr (BLK < BEND) {
    constexpr int A0 = blk_lo<T>(ROWB, D, 0), A1 = blk_hi<T>(ROWB, D, 0), B0 = blk_lo<T>(ROWB, D, 1), B1 = blk_hi<T>(ROWB, D, 1);
    constexpr int NA = A1 > A0 ? A1 - A0 : 1, NB = B1 > B0 ? B1 - B0 : 1, NU1 = blk_nu1<T>(L, BLK);
    cx<double> wv[COT];
#pragma unroll
    for (int o = 0; o < COT; ++o) wv[o] = wn[o];
    if constexpr (BLK + 1 < BEND) {      // the next block's weights are in flight during this block (its moments come from LDS: no pipeline)
#pragma unroll
      for (int o = 0; o < COT; ++o) wn[o] = {wc[((BLK + 1) * COT + o) * 2], wc[((BLK + 1) * COT + o) * 2 + 1]};
    }
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
    for (int k = 0; k < NU1; ++k) gu[k] = {0, 0};
    double dw[2 * COT];
#pragma unroll
    for (int k = 0; k < 2 * COT; ++k) dw[k] = 0.0;
    rows_bwd<T, ROWB, 0, D, COT, A0, NA, B0, NB, NU1>(go, wv, UvLds{w}, gu, xa, ga, xb, gb, dw);
    if constexpr (A1 > A0) gx_flush<NA>(gxl, A0, ga);
    if constexpr (B1 > B0) gx_flush<NB>(gxl, B0, gb);
    dw_store<2 * COT>(dw, part, BLK, lane);
    __builtin_amdgcn_sched_barrier(0);
    sep_reduce<T, L, BLK, NU1>(w, gu, lane);
    __builtin_amdgcn_sched_barrier(0);
    blocks_bwd_sep<T, L, BLK + 1, BEND, COT>(go, wn, wc, w, xl, gxl, part, lane, valid);
  }
}

// blocks BBEG .. BEND - 1 of output irrep L (as irrep_bwd of generic_local_static.hip); got / part: this lane's / this workgroup's rows
template <class T, int L, int BBEG, int BEND, int COT>
__device__ __forceinline__ void irrep_bwd_sep(const StaticBwdArgs& a, int c, const double* __restrict__ got, double* __restrict__ part0,
                                              const SepWalk& w, const double* xl, double* gxl, int lane, bool valid) {
  if constexpr (BBEG < BEND) {
    constexpr int D = T::DIM[L], NB = T::NBLK[L], QO = T::QOUT, QBASE = T::Q0[L];
    static_assert(BEND <= NB, "block range");
    const int CO = a.CO;
    typedef const double __attribute__((address_space(4))) * cptr;
    cptr wc = (cptr)(a.wp + a.wp0[L]) + (size_t)c * NB * COT * 2;
    const DwOut part = dw_out<COT>(a, part0, L, NB, c, lane);
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
    blocks_bwd_sep<T, L, BBEG, BEND, COT>(go, wn, wc, w, xl, gxl, part, lane, valid);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- after the walk: the parked block totals are added up into the jet sums, cmp [half][q][18] = S_0..4 (re, im) | SP_0..3 (re, im).
// Which parked slots feed which sum is a compile-time table (CSR: sums in walk order, i.e. a fixed summation order); the threads of
// the workgroup share the 36 Q sums of the two jets.
template <class T>
struct SepMap {
  static constexpr int NOUT = T::Q * 18, NENT = T::BLK_UPTR[sep_nblocks<T>()], ZERO = sep_vtotal<T>();      // ZERO: a slot that holds 0.0
  struct Tab { short src[NOUT][4]; };       // the parked slots of a sum, in walk order, padded with ZERO (no element feeds more than 4 blocks)
  static constexpr int out_of(int e, int part, int comp) {         // part 0: S_k, part 1: SP_m
    return (e / 5) * 18 + (part ? 10 + 2 * (e % 5 - 1) : 2 * (e % 5)) + comp;
  }
  static constexpr Tab make() {
    Tab t{};
    int fill[NOUT] = {};
    for (int o = 0; o < NOUT; ++o)
      for (int i = 0; i < 4; ++i) t.src[o][i] = (short)ZERO;
    for (int p = 0; p < NENT; ++p) {
      const int e = T::BLK_UELEM[p];
      for (int part = 0; part < (e % 5 ? 2 : 1); ++part)
        for (int comp = 0; comp < 2; ++comp) {
          const int o = out_of(e, part, comp);
          t.src[o][fill[o] < 4 ? fill[o] : 3] = (short)(sep_at<T>(p) + 2 * part + comp);
          ++fill[o];
        }
    }
    return t;
  }
  static constexpr bool fits() {
    int cnt[NOUT] = {};
    for (int p = 0; p < NENT; ++p) {
      const int e = T::BLK_UELEM[p];
      ++cnt[out_of(e, 0, 0)];
    }
    for (int o = 0; o < NOUT; ++o)
      if (cnt[o] > 4) return false;
    return true;
  }
  static constexpr Tab tab = make();
};

struct SepTail {
  cx<double> e0, R1;
  cx<double> gp2[4], A0, A1;
};
// components q0 .. q1 - 1 of the lane's node: aggregate part of d X (written with the two waves' own parts), second part of d p,
// bias-gradient terms.  cm: the compact sums of the lane's jet
__device__ __forceinline__ void sep_tail(int q0, int q1, const double* cm, const double* raw, const cx<double> (&P)[4], const double* xl,
                                         const double* gx0, const double* gx1, double* __restrict__ gxo, bool valid, SepTail& t) {
  for (int q = q0; q < q1; ++q) {
    const double* c18 = cm + q * 18;
    cx<double> S[5], SP[4];
#pragma unroll
    for (int k = 0; k < 5; ++k) S[k] = lds_cx(c18 + 2 * k);
#pragma unroll
    for (int m = 0; m < 4; ++m) SP[m] = lds_cx(c18 + 10 + 2 * m);
    const double* r10 = raw + q * 10;          // the jet's SX | SXP_0..3 (LDS)
    const cx<double> SX = lds_cx(r10), x = {xl[q * 128], xl[q * 128 + 1]};
    cx<double> u = {0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> ps = cmulc(S[1 + m], P[m]);
      u.r += SP[m].r - ps.r;
      u.i += SP[m].i - ps.i;
      cfmac(t.gp2[m], S[1 + m], x);
      cfmac(t.A1, SP[m], SX);
      const cx<double> u2 = cmulc(S[1 + m], lds_cx(r10 + 2 + 2 * m));
      t.A1.r -= u2.r;
      t.A1.i -= u2.i;
    }
    cfmac(t.A0, S[0], SX);
    cx<double> gx = cmulc(S[0], t.e0);
    cfmac(gx, u, t.R1);
    if (valid) {
      gxo[q * 128] = (gx0[q * 128] + gx1[q * 128]) + gx.r;
      gxo[q * 128 + 64] = (gx0[q * 128 + 1] + gx1[q * 128 + 1]) + gx.i;
    }
  }
}

// Workgroup = (jets 2 t and 2 t + 1, input channel c), two waves that split the blocks as in local_bwd_static_kernel (wave 0 alone
// touches the moments).  LDS (76.4 KB at Q = 20: two workgroups per CU): the channel's features xs [Q][64][2], the two waves' d X
// images, the table rows of the two jets, the lanes' momenta and d p, and the parked sums (whose space the tail's exchange reuses).
template <class T> constexpr int sep_un_doubles() { return 2 * (sep_vtotal<T>() + 2) > 64 * 8 + 16 ? 2 * (sep_vtotal<T>() + 2) : 64 * 8 + 16; }
// (the first level kind -- Q = 5, a small walk -- lets wave 1 take moment blocks too: it then has d p accumulators of its own)
template <class T> constexpr int sep_gpl_waves() { return T::Q == 5 ? 2 : 1; }
template <class T> constexpr size_t sep_lds_bytes() {
  return sizeof(double) * (size_t)(3 * T::Q * 128 + 2 * T::Q * 12 + (1 + sep_gpl_waves<T>()) * 64 * 8 + sep_un_doubles<T>());
}

// Wave assignment (balanced on in-kernel stamps of the 6 -> 4 level of cfg5, tools/sep_stamps.py: walk 100 k / 96 k cycles, sums +
// tail 15 k):  wave 0: the moment / feature blocks of every irrep, all of irrep 4, irrep 1 but its last block, the first product block
// of irrep 3;  wave 1: the product blocks of irreps 0, 2, 3 (but the first) and the last block of irrep 1.
template <class T, int COT>
__global__ __launch_bounds__(128) void local_bwd_sep_kernel(StaticBwdArgs a) {
  static_assert(T::N_OUT == 5 && T::NUBLK[1] < T::NBLK[1] && T::NUBLK[3] < T::NBLK[3], "wave assignment");
  constexpr int Q = T::Q, QO = T::QOUT, VT = sep_vtotal<T>() + 2;      // (+ the slot that holds 0.0: SepMap::ZERO)
  static_assert(SepMap<T>::fits(), "a jet sum with more than 4 parked slots");
  extern __shared__ __align__(16) double lds[];      // (16 bytes: the (re, im) pairs travel as ds_read_b128 / ds_write_b128, not as two b64 halves)
  double* xs = lds;                       // [Q][64][2]
  double* gxs = xs + Q * 128;             // [2 waves][Q][64][2]
  double* tab = gxs + 2 * Q * 128;        // [2 halves][Q][12]: E | A | B_0..3
  double* pl = tab + 2 * Q * 12;          // [4 m][64 lanes][2]: a lane's pairs 16 bytes apart (stride 64 bytes: 4-way bank conflicts on every read)
  double* gpl = pl + 64 * 8;              // [1 or 2 waves][4 m][64 lanes][2]
  double* sraw = gpl + sep_gpl_waves<T>() * 64 * 8;      // [2 halves][VT]; after the walk: the jet sums, then wave 1's tail sums [64][8] + [2 halves][4]
  int pair, c;
  if (!xcd_index((a.B + 1) >> 1, a.C, pair, c)) return;            // (workgroup-uniform: before any barrier)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, C = a.C, CO = a.CO, N = a.N;
  const int half = lane >> 5, j = lane & 31, jet = 2 * pair + half;
  const bool valid = jet < a.B && j < N;
  const int node = valid ? jet * N + j : (jet < a.B ? jet * N : a.M - 1);      // (an idle lane points at a valid node; it never stores)
  const size_t tile = (size_t)(node >> 6), l64 = (size_t)(node & 63);
  {
    const double* __restrict__ xc = a.XT + (tile * C + c) * Q * 128 + l64;
    double* gxw = gxs + wave * Q * 128;
    for (int q = wave; q < Q; q += 2) {
      xs[q * 128 + 2 * lane] = valid ? xc[q * 128] : 0.0;
      xs[q * 128 + 2 * lane + 1] = valid ? xc[q * 128 + 64] : 0.0;
    }
    for (int q = 0; q < Q; ++q) {
      gxw[q * 128 + 2 * lane] = 0.0;
      gxw[q * 128 + 2 * lane + 1] = 0.0;
    }
    for (int e = threadIdx.x; e < 2 * Q * 12; e += 128) {
      const int h = e / (Q * 12), r = e - h * Q * 12, jh = 2 * pair + h < a.B ? 2 * pair + h : a.B - 1;
      tab[e] = a.tbl[(((size_t)jh * C + c) * Q + r / 12) * TBL_STRIDE + r % 12];
    }
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) pl[(r >> 1) * 128 + 2 * lane + (r & 1)] = valid ? a.pc[(size_t)node * 8 + r] : 0.0;
    }
    if (wave < sep_gpl_waves<T>()) {
#pragma unroll
      for (int r = 0; r < 8; ++r) gpl[wave * 512 + (r >> 1) * 128 + 2 * lane + (r & 1)] = 0.0;
    }
    if (threadIdx.x < 2) sraw[threadIdx.x * VT + SepMap<T>::ZERO] = 0.0;
  }
  SepWalk w{tab + half * Q * 12, pl + 2 * lane, gpl + (sep_gpl_waves<T>() == 2 ? wave : 0) * 512 + 2 * lane,
            sraw + half * VT + half_sum8_k(lane), (lane & 3) == 0};
  __syncthreads();
  const double* xl = xs + 2 * lane;
  {
    double* gxl = gxs + wave * Q * 128 + 2 * lane;
    const double* __restrict__ got = a.goT + tile * CO * QO * 128 + l64;
    double* __restrict__ part0 = a.part + (size_t)pair * a.n_packed;
    if constexpr (T::Q == 5) {
      // first level (stamps, -DLGN_PSTAMP_Q=5: the round-3 split left wave 1 idle for 44 k of 68 k cycles): wave 0 the moment blocks of
      // irreps 0, 1, 3, 4; wave 1 irrep 2 whole -- its moment block included -- and every product block
      if (wave == 0) {
        PSTAMP(0);
        irrep_bwd_sep<T, 0, 0, T::NUBLK[0], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
        PSTAMP(1);
        irrep_bwd_sep<T, 1, 0, T::NUBLK[1], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
        PSTAMP(2);
        irrep_bwd_sep<T, 3, 0, T::NUBLK[3], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
        PSTAMP(4);
        irrep_bwd_sep<T, 4, 0, T::NUBLK[4], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
        PSTAMP(5);
      } else {
        PSTAMP(16);
        irrep_bwd_sep<T, 2, 0, T::NBLK[2], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
        PSTAMP(17);
        irrep_bwd_sep<T, 3, T::NUBLK[3], T::NBLK[3], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
        PSTAMP(18);
        irrep_bwd_sep<T, 4, T::NUBLK[4], T::NBLK[4], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
        PSTAMP(19);
        irrep_bwd_sep<T, 1, T::NUBLK[1], T::NBLK[1], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
        irrep_bwd_sep<T, 0, T::NUBLK[0], T::NBLK[0], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
        PSTAMP(20);
      }
    } else if (wave == 0) {
      PSTAMP(0);
      irrep_bwd_sep<T, 0, 0, T::NUBLK[0], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
      PSTAMP(1);
      irrep_bwd_sep<T, 1, 0, T::NBLK[1] - 1, COT>(a, c, got, part0, w, xl, gxl, lane, valid);
      PSTAMP(2);
      irrep_bwd_sep<T, 2, 0, T::NUBLK[2], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
      PSTAMP(3);
      irrep_bwd_sep<T, 3, 0, T::NUBLK[3] + 1, COT>(a, c, got, part0, w, xl, gxl, lane, valid);
      PSTAMP(4);
      irrep_bwd_sep<T, 4, 0, T::NBLK[4], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
      PSTAMP(5);
    } else {
      PSTAMP(16);
      irrep_bwd_sep<T, 2, T::NUBLK[2], T::NBLK[2], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
      PSTAMP(17);
      irrep_bwd_sep<T, 3, T::NUBLK[3] + 1, T::NBLK[3], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
      PSTAMP(18);
      PSTAMP(19);
      irrep_bwd_sep<T, 1, T::NBLK[1] - 1, T::NBLK[1], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
      irrep_bwd_sep<T, 0, T::NUBLK[0], T::NBLK[0], COT>(a, c, got, part0, w, xl, gxl, lane, valid);
      PSTAMP(20);
    }
  }
  __syncthreads();
  if (wave == 0) PSTAMP(6);
  // ---- jet sums: parked totals -> cmp [half][Q][18], in place (every thread holds its sums before anybody overwrites a slot); the
  // jets' raw sums SX | SXP (tail: bias gradients) come in from the table meanwhile and take the place of the walk's table rows ----
  double* raw = tab;                     // [2 halves][Q][10]
  {
    constexpr int NOUT = SepMap<T>::NOUT, PER = (2 * NOUT + 127) / 128, PERR = (2 * Q * 10 + 127) / 128;
    static_assert(NOUT <= VT, "the compact sums take the place of the parked ones");
    double rawv[PERR];
#pragma unroll
    for (int r = 0; r < PERR; ++r) {
      const int e = threadIdx.x + 128 * r, ee = e < 2 * Q * 10 ? e : 0, h = ee / (Q * 10), x = ee - h * Q * 10;
      const int jh = 2 * pair + h < a.B ? 2 * pair + h : a.B - 1;
      rawv[r] = a.tbl[(((size_t)jh * C + c) * Q + x / 10) * TBL_STRIDE + 12 + x % 10];
    }
    double mine[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int o = threadIdx.x + 128 * r, oc = o < 2 * NOUT ? o : 0, h = oc >= NOUT ? 1 : 0, oo = oc - h * NOUT;
      const short* sp = SepMap<T>::tab.src[oo];
      const double* sh = sraw + h * VT;
      mine[r] = ((sh[sp[0]] + sh[sp[1]]) + sh[sp[2]]) + sh[sp[3]];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int o = threadIdx.x + 128 * r, h = o >= NOUT ? 1 : 0;
      if (o < 2 * NOUT) sraw[h * VT + (o - h * NOUT)] = mine[r];
    }
#pragma unroll
    for (int r = 0; r < PERR; ++r)
      if (threadIdx.x + 128 * r < 2 * Q * 10) raw[threadIdx.x + 128 * r] = rawv[r];
    __syncthreads();
  }
  if (wave == 0) PSTAMP(9);
  // ---- tail: both waves, half of the components each ----
  SepTail t{};
  const double b0 = a.b0[c], b1 = a.b1[c];
  t.e0 = {0.0, 2.0 * b0};
  t.R1 = {b1, b1};
  {
    cx<double> P[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) P[m] = lds_cx(pl + 128 * m + 2 * lane);
    double* __restrict__ gxo = a.gXT + (tile * C + c) * Q * 128 + l64;
    constexpr int QH = (Q + 1) / 2;
    sep_tail(wave == 0 ? 0 : QH, wave == 0 ? QH : Q, sraw + half * VT, raw + half * Q * 10, P, xl, gxs + 2 * lane, gxs + Q * 128 + 2 * lane,
             gxo, valid, t);
  }
  if (wave == 0) PSTAMP(7); else PSTAMP(21);
  __syncthreads();                       // every read of the sums is done: their space takes wave 1's share of the tail
  double* xch = sraw;
  if (wave == 1) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      xch[m * 64 + lane] = t.gp2[m].r;
      xch[(4 + m) * 64 + lane] = t.gp2[m].i;
    }
    if (j == 0) {
      double* ax = xch + 64 * 8 + half * 4;
      ax[0] = t.A0.r; ax[1] = t.A0.i; ax[2] = t.A1.r; ax[3] = t.A1.i;
    }
  }
  __syncthreads();
  if (wave == 0) {
    if (valid) {
      double* __restrict__ gp = a.gpb + ((size_t)c * a.M + node) * 8;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const cx<double> g2 = {t.gp2[m].r + xch[m * 64 + lane], t.gp2[m].i + xch[(4 + m) * 64 + lane]};
        const cx<double> r = cmulc(g2, t.R1);              // g2 conj(R1)
        double g1r = gpl[128 * m + 2 * lane], g1i = gpl[128 * m + 2 * lane + 1];
        if constexpr (sep_gpl_waves<T>() == 2) { g1r += gpl[512 + 128 * m + 2 * lane]; g1i += gpl[512 + 128 * m + 2 * lane + 1]; }
        gp[m] = g1r - r.r;
        gp[4 + m] = g1i - r.i;
      }
    }
    if (j == 0 && jet < a.B) {
      const double* a1 = xch + 64 * 8 + half * 4;
      double* pr = a.part_rad + (size_t)jet * 2 * C;
      pr[c] = 2.0 * (t.A0.i + a1[1]);                                    // dB0[c] = 2 Im A0
      pr[C + c] = (t.A1.r + a1[2]) + (t.A1.i + a1[3]);                   // dB1[c] = Re A1 + Im A1
    }
    PSTAMP(8);
  }
}

template <class T, int COT>
__global__ __launch_bounds__(64) void local_fwd_sep_kernel(StaticArgs a) {
  static_assert(T::N_OUT == 5 && T::DIM[0] == 4 && T::DIM[1] == 3 && T::DIM[2] == 3 && T::DIM[3] == 9 && T::DIM[4] == 1, "item list");
  const int tile = blockIdx.x, lane = threadIdx.x;
  switch (blockIdx.y) {
    case 0: item_fwd<T, 3, 0, 2, COT, true>(a, tile, lane); break;
    case 1: item_fwd<T, 3, 2, 2, COT, true>(a, tile, lane); break;
    case 2: item_fwd<T, 3, 4, 2, COT, true>(a, tile, lane); break;
    case 3: item_fwd<T, 3, 6, 2, COT, true>(a, tile, lane); break;
    case 4: item_fwd<T, 3, 8, 1, COT, true>(a, tile, lane); break;
    case 5: item_fwd<T, 0, 0, 2, COT, true>(a, tile, lane); break;
    case 6: item_fwd<T, 0, 2, 2, COT, true>(a, tile, lane); break;
    case 7: item_fwd<T, 1, 0, 2, COT, true>(a, tile, lane); break;
    case 8: item_fwd<T, 1, 2, 1, COT, true>(a, tile, lane); break;
    case 9: item_fwd<T, 2, 0, 2, COT, true>(a, tile, lane); break;
    case 10: item_fwd<T, 2, 2, 1, COT, true>(a, tile, lane); break;
    default: item_fwd<T, 4, 0, 1, COT, true>(a, tile, lane); break;
  }
}

// d p of a network's decoder levels: sum of the per-channel parts of up to 4 levels, added to g_p [2][B][N][4]
struct GpJob { const double* gpb[4]; int C[4]; int n; };
__global__ __launch_bounds__(BLOCK) void gp_reduce_kernel(GpJob job, int M, double* __restrict__ g_p) {
  const size_t plane = (size_t)M * 4;
  for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < (size_t)M * 8; e += (size_t)gridDim.x * BLOCK) {
    const size_t node = e >> 3;
    const int r = (int)(e & 7);
    double s = 0.0;
    for (int l = 0; l < job.n; ++l)
      for (int c = 0; c < job.C[l]; ++c) s += job.gpb[l][((size_t)c * M + node) * 8 + r];
    g_p[(r >> 2) * plane + node * 4 + (r & 3)] += s;
  }
}

}  // namespace
LGN_STAMP_READER(lgn_debug_stamps_local_sep)

// rows of the packed CatMix partial gradients the separable backward writes (one per pair of jets)
int local_sep_part_rows(int B) { return (B + 1) / 2; }

// forward of a decoder level's per-node part with the moments formed from the jet table (tbl, pc: dec_sep_tab)
int local_fwd_sep(int kind, int B, int N, int C, int CO, const double* XT, const double* tbl, const double* pc, const double* wp,
                  double* outT, double* s_copy, int q_s, hipStream_t st) {
  LGN_CHECK_ARG(kind == 1 || kind == 2, "local_fwd_sep: unknown level kind %d", kind);
  LGN_CHECK_ARG(B > 0 && N > 0 && C >= 1 && C <= 8 && CO >= 1 && CO <= COMAX, "local_fwd_sep: unsupported shape (B=%d N=%d C=%d CO=%d)", B, N, C, CO);
  const int M = B * N;
  PackArgs p{};
  const int w0[5] = {0, 0, 0, 0, 0};
  if (kind == 1) fill_pack<cgs::Kind1>(p, C, CO, w0); else fill_pack<cgs::Kind2>(p, C, CO, w0);
  StaticArgs a{M, C, CO, XT, nullptr, tbl, pc, N, wp, {0, 0, 0, 0, 0, 0, 0, 0}, outT, s_copy, q_s};
  for (int l = 0; l < 5; ++l) a.wp0[l] = p.wp0[l];
  dim3 grid(cdiv(M, 64), 12);
#define LGN_LAUNCH(KIND, COT) hipLaunchKernelGGL((local_fwd_sep_kernel<cgs::KIND, COT>), grid, dim3(64), 0, st, a)
  if (kind == 1) { if (CO <= 4) LGN_LAUNCH(Kind1, 4); else if (CO <= 6) LGN_LAUNCH(Kind1, 6); else LGN_LAUNCH(Kind1, 8); }
  else { if (CO <= 4) LGN_LAUNCH(Kind2, 4); else if (CO <= 6) LGN_LAUNCH(Kind2, 6); else LGN_LAUNCH(Kind2, 8); }
#undef LGN_LAUNCH
  LGN_CHECK_LAUNCH();
  return 0;
}

// backward of a decoder level: per-node part AND separable moments in one launch.  part: local_sep_part_rows(B) rows of packed
// CatMix partial gradients; gpb [C][B N][8] and part_rad [B][2 C]: every entry written; gXT: written for every node
int local_bwd_sep(int kind, int B, int N, int C, int CO, const double* XT, const double* tbl, const double* pc, const double* b0,
                  const double* b1, const double* wp, const int* w0p, const double* goT, double* gXT, double* part, double* gpb,
                  double* part_rad, hipStream_t st) {
  LGN_CHECK_ARG(kind == 1 || kind == 2, "local_bwd_sep: unknown level kind %d", kind);
  LGN_CHECK_ARG(B > 0 && N > 0 && N <= 32 && C >= 1 && C <= 8 && CO >= 1 && CO <= COMAX,
                "local_bwd_sep: unsupported shape (B=%d N=%d C=%d CO=%d)", B, N, C, CO);
  PackArgs p{};
  const int w0[5] = {0, 0, 0, 0, 0};
  if (kind == 1) fill_pack<cgs::Kind1>(p, C, CO, w0); else fill_pack<cgs::Kind2>(p, C, CO, w0);
  StaticBwdArgs a{};
  a.M = B * N; a.C = C; a.CO = CO; a.XT = XT; a.wp = wp; a.goT = goT; a.gXT = gXT; a.part = part;
  a.n_packed = (int)local_static_packed_doubles(kind, C, CO);
  a.tbl = tbl; a.pc = pc; a.b0 = b0; a.b1 = b1; a.gpb = gpb; a.part_rad = part_rad; a.B = B; a.N = N;
  for (int l = 0; l < 5; ++l) { a.wp0[l] = p.wp0[l]; a.w0[l] = w0p ? w0p[l] : 0; }
  a.param_layout = w0p ? 1 : 0;          // (w0p: the irreps' offsets in the CatMix parameter block -> partial rows in parameter layout)
  dim3 grid(xcd_grid(local_sep_part_rows(B), C));
#define LGN_LAUNCH(KIND, COT)                                                                                                   \
  do {                                                                                                                          \
    const size_t smem = sep_lds_bytes<cgs::KIND>();                                                                             \
    LGN_CHECK_ARG(smem <= 80 * 1024, "local_bwd_sep: %zu B of LDS", smem);                                                      \
    if (smem > 64 * 1024)                                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(local_bwd_sep_kernel<cgs::KIND, COT>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                                         \
    hipLaunchKernelGGL((local_bwd_sep_kernel<cgs::KIND, COT>), grid, dim3(128), smem, st, a);                                   \
  } while (0)
  if (kind == 1) { if (CO <= 4) LGN_LAUNCH(Kind1, 4); else if (CO <= 6) LGN_LAUNCH(Kind1, 6); else LGN_LAUNCH(Kind1, 8); }
  else { if (CO <= 4) LGN_LAUNCH(Kind2, 4); else if (CO <= 6) LGN_LAUNCH(Kind2, 6); else LGN_LAUNCH(Kind2, 8); }
#undef LGN_LAUNCH
  LGN_CHECK_LAUNCH();
  return 0;
}

// g_p [2][B][N][4] += sum over levels and channels of gpb_l [C_l][M][8]
int local_sep_gp_reduce(const double* const* gpb, const int* C, int n, int M, double* g_p, hipStream_t st) {
  LGN_CHECK_ARG(n >= 1 && n <= 4 && M > 0 && g_p, "local_sep_gp_reduce: bad arguments");
  GpJob job{};
  job.n = n;
  for (int l = 0; l < n; ++l) { job.gpb[l] = gpb[l]; job.C[l] = C[l]; }
  const int blocks = (int)std::min<size_t>(((size_t)M * 8 + BLOCK - 1) / BLOCK, 1024);
  hipLaunchKernelGGL(gp_reduce_kernel, dim3(blocks), dim3(BLOCK), 0, st, job, M, g_p);
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace lgn
