// lgn-autoencoder_amd/csrc/level.hpp -- argument block + LDS carve shared by the maxdim=2 level kernels.
//
// One "level" = LGNNodeLevel of the reference (lgn/models/lgn_levels.py:96-121) with its edge
// network fused in:  edge_ij = RadPolyTrig(|p_i-p_j|) * zonal(p_i-p_j)      (position_levels.py:118-209,
//                                                                          lgn_cg.py:167)
//                    ag_i    = sum_j CG(node_j (x) edge_ij)                 (cg_ops.py:135-298, aggregate)
//                    sq_i    = CG(node_i (x) node_i)                        (cg_ops.py, power)
//                    out_i   = MixReps([ag_i, node_i, sq_i])                (g_nn.py:260-278)
// The N x N edge tensors are never written to memory.
//
// Closed form at maxdim = 2 (the only non-trivial CG block is
// (1,1)x(1,1)->(0,0) = 1/2 [e00 + e13 - e22 + e31]):
//   A1[c][m] = sum_j v_j[c][m] * e0_ij[c]            -> ag(1,1) channel c        pair ((1,1),(0,0))
//   A2[c][m] = sum_j s_j[c]    * e1_ij[c][m]         -> ag(1,1) channel C+c      pair ((0,0),(1,1))
//   A3[c]    = sum_j <v_j[c], e1_ij[c]>              -> ag(0,0) channel c        pair ((1,1),(1,1))
//   A4[c]    = sum_j s_j[c]    * e0_ij[c]            -> ag(0,0) channel C+c      pair ((0,0),(0,0))
//   e0 = R0 * (1+1i)   (zonal (0,0) is ones on BOTH planes, zonal_functions.py:150-154)
//   e1[m] = R1 * q[m], q = canonical(p_i - p_j)
//   <a,b> = 1/2 (a0 b0 + a1 b3 - a2 b2 + a3 b1)       (no conjugation)
#pragma once
#include "common.hpp"

namespace lgn {

// Kernel-selecting debug switches (cross-checks).  ONE decision per call: the per-operator C entry points read the environment once
// (api_common.cpp: level_flags_from_env), whole-network calls take the bits frozen into lgn_net_desc.flags (same values as LGN_NET_*)
// -- every launcher and every sizing helper below receives them as an argument and none reads the environment itself.
constexpr int LVL_DEC_PAIRWISE = 2;    // decoder levels as O(N^2) pair sweeps instead of the separable form
constexpr int LVL_LEVEL_V2 = 4;        // three-kernel level backward also for N <= 40
constexpr int LVL_MLP_V1 = 8;          // CGMLP: the 12-wave kernels of mlp_mfma.hip also where the chain kernels (mlp_chain.hip) apply
// (8 and 32 were round 4's switches of the CGMLP riding on the level kernels: built, measured slower in every regime, removed in round 5)
constexpr int LVL_MOMENTS_V1 = 16;     // table-driven levels: component-chunked moments kernels
constexpr int LVL_MLP_BWD1 = 1024;      // CGMLP chain backward: the one-role kernel (four waves: chain + weight gradients + staging) instead of the two-role one
constexpr int LVL_MOMENTS_SPLIT = 512; // table-driven encoder levels: the backward's two pair sweeps as two kernels (cross-check of the merged one)
constexpr int LVL_BWD_ORDERED = 64;    // encoder level backward (N <= 40): radial-gradient GEMM per ORDERED pair tile (cross-check of the symmetric sweep)
int level_flags_from_env();

template <typename T>
struct LevelArgs {
  int B, N, C, CO;
  // node features entering the level
  const T* s_in;   // [2][B][N][C]
  const T* v_in;   // [2][B][N][C][4]
  // positions: encoder real Cartesian [B][N][4]; decoder complex canonical [2][B][N][4]
  const T* p;
  const uint8_t* mask;  // encoder [B][N]; decoder: nullptr (all edges are "masked": radial == bias)
  // radial network.  encoder: ra,rb,rc [20]; w0,w1 [2C][20]; b0,b1 [2C] (feature 2c+z).
  //                  decoder: b0,b1 [C] only (same real bias on both planes, position_levels.py:184-188)
  const T *ra, *rb, *rc, *w0, *b0, *w1, *b1;
  // CatMix weights [2][CO][5C] per irrep
  const T *wm0, *wm1;
  // saved aggregate (needed by the backward CatMix-weight gradient) in reference layout
  T* ag0;    // [2][B][N][2C]
  T* ag1;    // [2][B][N][2C][4]
  // level output before the CGMLP
  T* s_out;  // [2][B][N][CO]
  T* v_out;  // [2][B][N][CO][4]
  // Encoder, first level of a whole-network / whole-step call (in_w0 != nullptr): the level forms its own input features from
  // the momenta -- s = W00[c] * mass, v = W11[c] * canon(p): input_func_node of lgn_encoder.py:284-298,376, the arithmetic of
  // enc_input_fwd_kernel -- instead of reading s_in / v_in, and writes them to in_s / in_v (== s_in / v_in) for the backward.
  // It is then the FIRST kernel of the step and also clears z1[0 .. z1n) and z2[0 .. z2n) (gradient buffer, zero block).
  const T* in_w0 = nullptr;   // [2][C] input_func_node weights of (0,0)
  const T* in_w1 = nullptr;   // [2][C]                          (1,1)
  T* in_s = nullptr;
  T* in_v = nullptr;
  T* z1 = nullptr;
  size_t z1n = 0;
  T* z2 = nullptr;
  size_t z2n = 0;
  // Decoder, last level of a whole-step call (loss_wo1 != nullptr; separable form, N <= 40: level_fwd_carries_loss): the decoder
  // output + get_real('sum') + Chamfer loss of the jet, forward and backward (net_dev.hpp: dec_output_loss_body), run as the tail
  // of this kernel on the v_out it has just written.
  const T* loss_wo1 = nullptr;     // mix_to_output weights of (1,1): [2][CO]
  const T* loss_target = nullptr;  // [B][N][4]
  T loss_scale = T(1);
  T* loss_recon = nullptr;         // [2][B][N][4]
  T* loss_part = nullptr;          // [B]
  T* loss_gv = nullptr;            // [2][B][N][CO][4] gradient w.r.t. v_out
  T* loss_wpart = nullptr;         // [B][2 CO]
  int flags = 0;                   // LVL_*
};

template <typename T>
struct LevelBwdArgs {
  int B, N, C, CO;
  const T *s_in, *v_in, *p;
  const uint8_t* mask;
  const T *ra, *rb, *rc, *w0, *b0, *w1, *b1;
  const T *wm0, *wm1;
  const T *ag0, *ag1;
  // upstream gradients w.r.t. the level output (pre-MLP scalars, vectors)
  const T* g_s_out;  // [2][B][N][CO]
  const T* g_v_out;  // [2][B][N][CO][4]
  // gradient of the aggregate, scratch [B][N][20C]: per node [A3(2C: c,z)][A4][A1 (C,4,2)][A2]
  T* g_ag;
  // outputs
  T* g_s_in;   // [2][B][N][C]
  T* g_v_in;   // [2][B][N][C][4]
  T* g_p;      // decoder only: [2][B][N][4]; accumulated (+=)
  // per-workgroup partial sums of parameter gradients, reduced by reduce_partials afterwards
  T* part_mix;   // [nblk_mix][2*2*CO*5C]      wm0 then wm1
  T* part_rad;   // [nblk_edge][rad_partial_size(C, decoder)]
  // Encoder, first level of a whole-network / whole-step call (N <= 40 kernel only): the backward of the input stage
  // (input_func_node: dW00[c] = sum_n g_s[n][c] mass_n, dW11[c] = sum_n sum_m g_v[n][c][m] conj(q_n[m]); enc_input_bwd_kernel's
  // arithmetic) rides on this kernel, one partial row [4C] = (dW00 re, im, dW11 re, im) per workgroup like part_mix.
  T* part_in0 = nullptr;
  int flags = 0;             // LVL_*
};

// Per-node stride (in scalars) of the node tile kept in LDS: [c][ s_r s_i v_r[4] v_i[4] ] + 2 pad.
__host__ __device__ constexpr int node_stride(int C) { return 10 * C + 4; }

// encoder radial partial layout: T1[4C][20] | T2[4C][20] | S[4C] | dB[4C]
// decoder: dB0[C] | dB1[C]
__host__ __device__ constexpr int rad_partial_size(int C, bool dec) {
  return dec ? 2 * C : (2 * 4 * C * NB + 8 * C);
}

// Workgroups per jet of the pair-sweep level kernels for jets of <= 40 particles (level_fwd2.hip, level_bwd3.hip).  A jet is
// one workgroup whose waves each sweep a group of 4 particles against the whole jet; with 512 jets that fills the chip (two
// workgroups per CU), with 64 jets three quarters of the CUs idle while each busy one works through a whole jet (31 us for the
// backward whatever the batch).  Small batches therefore give the groups of a jet to several workgroups -- every one stages the
// jet, sweeps its own groups and writes its own rows of the outputs and its own partial rows of the parameter gradients.
inline int level_jet_split(int B, int N) {
  const int groups = (N + 3) / 4;
  int s = 1;
  while (s * 2 <= groups && B * s * 2 <= 512) s *= 2;      // (measured at 512 jets: 2 / 4 workgroups per jet cost 6 / 44 us per backward launch)
  return s;
}

}  // namespace lgn
