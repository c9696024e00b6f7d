/*
 * include/lgn_amd.h -- C ABI of liblgn_amd.so, the MI355X (gfx950) implementation of the LGN
 * message-passing hot path of zichunhao/lgn-autoencoder.
 *
 * The reference has no FFI layer: the path sits behind the Python nn.Module API of
 * lgn.models.LGNEncoder / LGNDecoder (lgn/models/__init__.py:1-5).  This header is the boundary the
 * build introduces *below* that API; each entry point names the reference code it replaces.
 *
 * Conventions
 *  - All pointers are DEVICE pointers owned by the caller (PyTorch); the library never allocates,
 *    frees or retains device memory.  Tensors are contiguous in the reference's planar-complex
 *    layouts:  scalar irrep (0,0): T[2][B][N][C];  vector irrep (1,1): T[2][B][N][C][4];
 *    MixReps weight: T[2][C_out][C_in]  (lgn/g_lib/g_vec.py:30-48, g_weight.py:38-40).
 *  - `stream` is a hipStream_t (0 = default stream).  Calls only enqueue work; no host sync.
 *  - Return value: 0 success; < 0 argument/shape error detected on the host before any launch;
 *    > 0 a hipError_t from a launch.  lgn_last_error() gives the message (thread-local).
 *  - Suffix _f64: IEEE double arithmetic (the reference is fp64-only, lgn/cg_lib/cg_module.py:62-73).
 *  - Parameter-gradient reductions over the batch are deterministic: kernels write per-workgroup
 *    partial rows into caller-provided workspaces which lgn_reduce_partials_f64 sums in a fixed order.
 */
#ifndef LGN_AMD_H
#define LGN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGN_AMD_ABI_VERSION 17   /* bump on ANY struct or signature change (lgn/_native.py: ABI_VERSION) */

int lgn_abi_version(void);
const char* lgn_last_error(void);

/* ---- message-passing level, maxdim = 2 ------------------------------------------------------
 * Replaces, fused: RadPolyTrig.forward (lgn/nn/position_levels.py:118-209), edge = rad * zonal
 * (lgn/models/lgn_cg.py:167; zonal functions lgn/cg_lib/zonal_functions.py:123-248),
 * CGProduct aggregate + power (lgn/cg_lib/cg_ops.py:135-298; LGNNodeLevel.forward
 * lgn/models/lgn_levels.py:96-121) and CatMixReps (lgn/nn/g_nn.py:260-278).
 *
 *  decoder = 0: p = real Cartesian momenta [B][N][4], mask = node mask [B][N] (uint8);
 *               ra,rb,rc [20]; w0,w1 [2C][20]; b0,b1 [2C]   (Linear feature index 2c+z)
 *  decoder = 1: p = complex canonical momenta [2][B][N][4], mask ignored (all edges masked:
 *               lgn/models/lgn_decoder.py:335-340); only the Linear biases b0,b1 [C] are used.
 *  wm0, wm1: CatMix weights [2][CO][5C] of irreps (0,0) and (1,1); cat order [aggregate, node, power].
 *  outputs: ag0 [2][B][N][2C], ag1 [2][B][N][2C][4] (the aggregate CG product, kept for backward),
 *           s_out [2][B][N][CO], v_out [2][B][N][CO][4].
 */
int lgn_level_fwd_f64(int B, int N, int C, int CO, int decoder,
                      const double* s_in, const double* v_in, const double* p, const uint8_t* mask,
                      const double* ra, const double* rb, const double* rc,
                      const double* w0, const double* b0, const double* w1, const double* b1,
                      const double* wm0, const double* wm1,
                      double* ag0, double* ag1, double* s_out, double* v_out, void* stream);

/* Workspace sizing for lgn_level_bwd_f64: number of partial rows written for the CatMix weights
 * (row length 4*CO*5C) and for the radial network (row length lgn_level_rad_partial_len). */
int lgn_level_bwd_partial_rows(int B, int N, int decoder, int* rows_mix, int* rows_rad);
int lgn_level_rad_partial_len(int C, int decoder);
/* Workgroups per jet of the pair-sweep level kernels at this batch (jets of <= 40 particles; small batches split a jet's row groups
 * over several workgroups; 1 otherwise): what decides the partial-row counts above and which backward instantiation runs. */
int lgn_level_jet_split(int B, int N);

/* Backward of lgn_level_fwd_f64 (autograd of the same reference functions).  Edges are recomputed.
 *  in : forward inputs + ag0/ag1 + g_s_out [2][B][N][CO], g_v_out [2][B][N][CO][4]
 *  out: g_s_in [2][B][N][C], g_v_in [2][B][N][C][4] (overwritten);
 *       g_p [2][B][N][4] (decoder only; ACCUMULATED into, caller zero-initialises);
 *       g_ag scratch [B][N][20C]; part_mix [rows_mix][4*CO*5C]; part_rad [rows_rad][rad_partial_len].
 *  The caller then reduces the partial rows (lgn_reduce_partials_f64) and, for the encoder, converts
 *  the reduced radial sums into parameter gradients with lgn_radial_finalize_f64.
 */
int lgn_level_bwd_f64(int B, int N, int C, int CO, int decoder,
                      const double* s_in, const double* v_in, const double* p, const uint8_t* mask,
                      const double* ra, const double* rb, const double* rc,
                      const double* w0, const double* b0, const double* w1, const double* b1,
                      const double* wm0, const double* wm1, const double* ag0, const double* ag1,
                      const double* g_s_out, const double* g_v_out,
                      double* g_ag, double* g_s_in, double* g_v_in, double* g_p,
                      double* part_mix, double* part_rad, void* stream);

/* ---- Clebsch-Gordan product of two irreps, standalone (lgn/cg_lib/cg_ops.py:135-218: cg_product; :221-297: complex_kron_product) --
 * Per channel: out = H (x1 (x) x2), H = the stacked Clebsch-Gordan matrix [DO][D1 * D2] of the irrep pair in CSR form (device arrays
 * row_ptr [DO + 1], col [nnz] = m1 * D2 + m2, coef [nnz]).  mode 0: x1 [2][R][C][D1], x2 [2][R][C][D2].  Aggregate (sum over the
 * neighbour index before H, cg_ops.py:281-291), rows R = B * N: mode 1: x1 [2][B][N][N][C][D1] edge-like, x2 [2][B][N][C][D2];
 * mode 2: the operands the other way round.  out [2][R][C][DO].  Backward: g_x1 / g_x2 are ACCUMULATED into (zero-filled by the
 * caller); either may be NULL (that operand is data: its gradient is not computed).  What lgn.cg_lib.cg_product / CGProduct bind, one call per pair of irreps; inside the networks the product is fused into
 * the level kernels and never materialised. */
int lgn_cg_product_fwd_f64(int R, int N, int C, int D1, int D2, int DO, int mode, int nnz, const int* row_ptr, const int* col,
                           const double* coef, const double* x1, const double* x2, double* out, void* stream);
int lgn_cg_product_bwd_f64(int R, int N, int C, int D1, int D2, int DO, int mode, int nnz, const int* row_ptr, const int* col,
                           const double* coef, const double* x1, const double* x2, const double* g_out, double* g_x1, double* g_x2,
                           void* stream);

/* out[n] = (accumulate ? out[n] : 0) + sum_r part[r][n], fixed summation order. */
int lgn_reduce_partials_f64(const double* part, int rows, int n, double* out, int accumulate, void* stream);

/* Encoder radial network: reduced pair sums (T1|T2|S|dB, see csrc/level_bwd.hip) -> gradients of
 * RadPolyTrig's a, b, c [20], linear.{0,1}.weight [2C][20] and .bias [2C]
 * (parameters of lgn/nn/position_levels.py:67-97). */
int lgn_radial_finalize_f64(const double* tot, int C, const double* ra, const double* rb, const double* rc,
                            const double* w0, const double* w1,
                            double* g_a, double* g_b, double* g_c,
                            double* g_w0, double* g_b0, double* g_w1, double* g_b1, void* stream);

/* ---- CGMLP (lgn/models/lgn_levels.py:191-227) ------------------------------------------------
 * rows M = B*N, features 2C (index 2c+z) taken from / written to the scalar irrep [2][M][C];
 * nlin Linear layers (nn.Linear weight [out][in], bias [out]) of hidden width H, the activation
 * (LGN_ACT_*: get_activation_fn, lgn/nn/generic_levels.py:119-135) after all but the last.
 * w / b: host arrays of nlin device pointers. */
#define LGN_ACT_LEAKYRELU 0   /* nn.LeakyReLU() (slope 0.01): the reference default */
#define LGN_ACT_RELU 1
#define LGN_ACT_ELU 2         /* alpha = 1 */
#define LGN_ACT_SIGMOID 3
#define LGN_ACT_LOGSIGMOID 4
#define LGN_ACT_ATAN 5
#define LGN_ACT_COUNT 6
int lgn_cgmlp_fwd_f64(int M, int C, int H, int nlin, int activation, const double* const* w, const double* const* b,
                      const double* s_in, double* s_out, void* stream);
/* rows of the partial weight-gradient buffer for M rows at hidden width H: one per workgroup of the backward (64 rows; 16 rows
 * for H <= 48 when M is small enough that 64-row workgroups would leave most CUs idle). */
int lgn_cgmlp_partial_rows(int M, int H);
/* part [lgn_cgmlp_partial_rows(M, H)][psize], psize = sum_l (out_l*in_l + out_l), layout concat_l (W_l, b_l). */
int lgn_cgmlp_bwd_f64(int M, int C, int H, int nlin, int activation, const double* const* w, const double* const* b,
                      const double* s_in, const double* g_out, double* g_in, double* part, int psize, void* stream);

/* ---- MixReps (lgn/nn/g_nn.py:95-117, lgn/g_lib/cplx_lib.py:7-25) -------------------------------
 * y[z][row][o][m] = sum_i W[o][i] x[row][i][m]   complex, d = irrep dimension. */
int lgn_mixreps_fwd_f64(int rows, int Cin, int Cout, int d, const double* w, const double* x, double* y, void* stream);
int lgn_mixreps_partial_rows(int rows);
/* g_x may be NULL (input is data).  part [rows][2*Cout*Cin]. */
int lgn_mixreps_bwd_f64(int rows, int Cin, int Cout, int d, const double* w, const double* x, const double* g_y,
                        double* g_x, double* part, void* stream);

/* ---- arbitrary irreps (maxdim = 3): table-driven level --------------------------------------------------
 * Same operator as lgn_level_fwd/bwd for node features carrying any set of irreps with k, n < maxdim, packed as
 * X [2][B][N][C][Q] (Q = sum of irrep dimensions, irreps in the level's GVec order).  Split in two stages:
 *  (1) "moments" (O(N^2), no CG tables):  U[b][i][c][q][0] = sum_j X_j[c][q] e0_ij[c],  U[..][1+m] = sum_j X_j[c][q] e1_ij[c][m]
 *      -- the neighbour sum of the Kronecker products of lgn/cg_lib/cg_ops.py:281-297 before the CG matrix is applied;
 *  (2) per node: sparse CG contraction of U (aggregate) and of X (x) X (power), concatenation with X, CatMix
 *      (cg_ops.py:195-215, lgn/nn/g_nn.py:160-190,260-278), driven by the CSR tables of lgn_local_tables.
 * U / gU layout: [B][N][C][Q][5][2] (re, im innermost).  Radial parameters as for lgn_level_fwd_f64. */
int lgn_moments_fwd_f64(int B, int N, int C, int Q, int decoder, const double* X, const double* p, const uint8_t* mask,
                        const double* ra, const double* rb, const double* rc, const double* w0, const double* b0,
                        const double* w1, const double* b1, double* U, void* stream);
/* gX [2][B][N][C][Q] and g_p (decoder) are ACCUMULATED into; part_rad [B][lgn_level_rad_partial_len(C, decoder)];
 * scratch: lgn_moments_scratch_doubles(B, N, C, decoder) doubles (0 = none needed, NULL allowed: jets of up to 32 particles
 * run channel-outermost kernels whose encoder radial backward parks the per-pair gradients there). */
long long lgn_moments_scratch_doubles(int B, int N, int C, int decoder);
int lgn_moments_bwd_f64(int B, int N, int C, int Q, int decoder, const double* X, const double* p, const uint8_t* mask,
                        const double* ra, const double* rb, const double* rc, const double* w0, const double* b0,
                        const double* w1, const double* b1, const double* gU, double* gX, double* g_p,
                        double* part_rad, double* scratch, void* stream);

typedef struct lgn_local_tables {
  int n_rows, n_out, n_w;          /* concatenated rows (irrep, block, m); output irreps; complex CatMix weights */
  int n_terms, n_u, n_x;           /* lengths of the three CSR term lists (= row_ptr[n_rows], u_ptr[5 Q], x_ptr[Q]); the backward keeps
                                      the U list in LDS and the host sizes it from n_u */
  int n_units, static_kind;        /* n_units: reserved (0).  static_kind: 1 / 2 when these tables are exactly the ones compiled into
                                      the library for the first / later levels of maxdim = 3 networks (csrc/cg_static_tables.hpp,
                                      lgn/plan.py: static_kind), else 0: whole-network calls then run the compile-time-table kernels */
  const int *row_ptr, *t_type, *t_a, *t_b;      /* CSR terms per row: type (t_type & 3) 0: U[a = q*5+k], 1: X[a], 2: X[a]*X[b];
                                                   t_type & 4 marks the last term of a row (every row has >= 1 term) */
  const double* t_coef;
  const int *out_dim, *out_nblk, *out_row0, *out_q0, *out_w0;   /* per output irrep */
  const int *u_ptr, *u_row;                     /* transposed lists for the backward */
  const double* u_coef;
  const int *x_ptr, *x_row, *x_other;
  const double* x_coef;
  int h_out_w0[8];                 /* host copy of out_w0 (first n_out entries): the compile-time-table kernels take the offsets by value */
} lgn_local_tables;

/* X [2][nodes][C][Q], U [nodes][C][Q][5][2], wcat = CatMix weights of all irreps ([2][CO][nblk*C] each, at out_w0),
 * out [2][nodes][CO][Qout]. */
int lgn_local_fwd_f64(int nodes, int C, int CO, int Q, int Qout, const lgn_local_tables* t, const double* X, const double* U,
                      const double* wcat, double* out, void* stream);
int lgn_local_partial_rows(int nodes);
/* gU, gX overwritten; part [lgn_local_partial_rows][2*n_w] (layout like wcat). */
int lgn_local_bwd_f64(int nodes, int C, int CO, int Q, int Qout, const lgn_local_tables* t, const double* X, const double* U,
                      const double* wcat, const double* g_out, double* gU, double* gX, double* part, void* stream);

/* The two level kinds of maxdim = 3 networks (kind 1: first level, node irreps (1,1), (0,0); kind 2: later levels, all five
 * irreps) have their tables compiled in (csrc/cg_static_tables.hpp): same operator as lgn_local_fwd_f64 on tile-blocked,
 * node-innermost layouts (node n = 64 tile + lane; buffers cover ceil(nodes / 64) whole tiles):
 *   XT [tile][C][Q][2][64], UT [tile][C][5 Q][2][64], outT [tile][CO][Qout][2][64];
 * w0[5] = offset of each output irrep's weights in wcat (host array); wpacked: scratch of
 * lgn_local_static_packed_doubles(kind, C, CO) doubles (the call repacks the weights into it);
 * s_copy optional [2][nodes][CO] copy of output component q_s (dense). */
long long lgn_local_static_packed_doubles(int kind, int C, int CO);
/* Backward of lgn_local_fwd_static_f64: goT [tile][CO][Qout][2][64] -> gUT [tile][C][5 Q][2][64], gXT [tile][C][Q][2][64]
 * (both overwritten) and g_wcat += CatMix weight gradient (wcat layout; caller zero-initialises).  Scratch: wpacked and
 * gpacked (lgn_local_static_packed_doubles doubles each), part (ceil(nodes / 64) rows of that length). */
int lgn_local_bwd_static_f64(int kind, int nodes, int C, int CO, const double* XT, const double* UT, const double* wcat,
                             const int* w0, double* wpacked, const double* goT, double* gUT, double* gXT, double* part,
                             double* gpacked, double* g_wcat, void* stream);
int lgn_local_fwd_static_f64(int kind, int nodes, int C, int CO, const double* XT, const double* UT, const double* wcat,
                             const int* w0, double* wpacked, double* outT, double* s_copy, int q_s, void* stream);

/* ---- whole training step (utils/train.py:283-343 inner loop); fused maxdim = 2 or table-driven networks -------
 * One call enqueues encoder -> decoder -> get_real('sum') -> Chamfer -> full backward (~80 launches, no host
 * sync, all buffers caller-owned and static => capturable in a HIP graph).  Parameters of both networks
 * live in ONE flat buffer `params`; gradients are written into `grads` at the same offsets (the call zero-
 * fills `grads` first; parameters that cannot receive gradient keep an exact 0).
 * Offsets (in elements) are given per slot, in this order (L = n_levels, nlin = mlp_nlin):
 *   encoder: input_func_node (0,0),(1,1) | per level: a,b,c,linear.0.weight,.bias,linear.1.weight,.bias |
 *            per level: cat_mix (0,0),(1,1) | per level: linear.0.weight,.bias ... linear.{nlin-1} | mix_reps (0,0),(1,1)
 *   decoder: latent_to_graph (0,0),(1,1) | input_func_node (0,0),(1,1) | radial | cat_mix | mlp | mix_to_output (0,0),(1,1)
 * Pooling is 'min&max' (lgn/models/lgn_encoder.py:419-583); the decoder consumes 2*tau_v latent vectors.
 */
typedef struct lgn_net_desc {
  int B, N;
  int n_levels;            /* message-passing levels per network (<= 4) */
  int enc_channels[5];     /* n_levels + 1 entries */
  int dec_channels[5];
  int tau_s, tau_v;        /* encoder latent multiplicities before the min&max concatenation */
  int mlp_hidden_mul;      /* CGMLP hidden width = mlp_hidden_mul * 2C  (reference: mlp_width) */
  int mlp_nlin;            /* Linear layers per CGMLP (mlp_depth + 1) */
  int tau_v_in;            /* decoder: latent vectors it consumes; 0 = 2 * tau_v (the 'min&max' concatenation) */
  /* Table-driven networks (any level with maxdim = 3): set enc_tables[l] / dec_tables[l] for EVERY level of that network
   * (host structs holding device pointers, as for lgn_local_fwd_f64; all NULL = the fused maxdim = 2 kernels).  The level
   * features are then the packed tensors X_l [2][B][N][C_l][Q_l] of lgn_moments_fwd_f64 / lgn_local_fwd_f64 with
   * Q_l = *_Q[l] components per channel, the scalar irrep (0,0) at component *_qs[l] and the vector irrep (1,1) at
   * components *_qv[l] .. +3 (l = 0 .. n_levels).  The CatMix parameter slot (.., 0) of level l is the lowest offset of the
   * level's CatMix weights in the flat block; tables[l]->out_w0 are offsets from there (slot (.., 1) is ignored). */
  const lgn_local_tables* enc_tables[4];
  const lgn_local_tables* dec_tables[4];
  int enc_Q[5], enc_qs[5], enc_qv[5];
  int dec_Q[5], dec_qs[5], dec_qv[5];
  /* LGN_NET_* bits.  Switches that change the LAYOUT of buffers living across calls (the activations a forward leaves for
   * its backward) are part of the descriptor, fixed when the caller creates it -- not read from the environment per call, so
   * a forward and its backward can never disagree. */
  int flags;
  int activation;          /* LGN_ACT_* of every CGMLP of both networks (the reference builds them from one --activation) */
  int n_in_scalars;        /* encoder: input scalars per node, K (0 or 1: the mass alone).  K > 1 -- jet_features and / or
                              data['scalars'], lgn/models/lgn_encoder.py:372-411: the mass, then K - 1 values per node the caller
                              passes as in_scalars [B][N][K - 1] -- to the per-network calls and (ABI 17, maxdim = 2 networks) the
                              whole-step calls */
  int latent_pool;         /* encoder: how the latent channels are pooled over the particles (aggregate(), lgn/models/
                              lgn_encoder.py:419-496): 0 = 'min&max' (the reference default), else LGN_POOL(...).  With P output
                              blocks (one per pooling under '&', one in all under '+') the latent space is lat_s [2][B][P tau_s],
                              lat_v [2][B][P tau_v][4], and the decoder of a whole step takes Tin = P tau_v vectors */
  int dec_N;               /* whole-step call only: particles the decoder reconstructs when that differs from the encoder's node count N
                              (jet_features: the encoder works on N = particles + 1 nodes, lgn_encoder.py:372-411); 0 = N.  With
                              dec_N != N or n_in_scalars > 1 the step runs its four end stages as launches of their own (maxdim = 2
                              networks; table-driven networks refuse) */
} lgn_net_desc;
/* latent pooling code: n = 1..4 poolings o0..o3 (LGN_POOL_MIN / MAX / MEAN), avg = 0: concatenated ('a&b'), 1: averaged ('a+b').
 * min / max pick ONE particle per (plane, channel) -- by the value itself (min) / its square (max) for scalars, by the Minkowski
 * square of the Cartesian vector for vectors (get_min_features / get_max_features, lgn_encoder.py:538-583); mean = torch.mean over
 * the particle axis, padded particles included.  ('sum' returns an extra axis in the reference: per-operator path only.) */
#define LGN_POOL_MIN 0
#define LGN_POOL_MAX 1
#define LGN_POOL_MEAN 2
#define LGN_POOL_MIX 3        /* only as LGN_POOL(1, 0, LGN_POOL_MIX, 0, 0, 0): map_to_latent = 'mix' -- no pooling, the latent MixReps
                                 weights (encoder output slots) are [2][tau][N C] and act on all (particle, channel) pairs of a jet
                                 (lgn_encoder.py:226-232,313-319); one output block */
#define LGN_POOL(n, avg, o0, o1, o2, o3) ((n) | ((avg) << 3) | ((o0) << 4) | ((o1) << 6) | ((o2) << 8) | ((o3) << 10))
#define LGN_NET_NO_STATIC 1   /* table-driven levels: run-time-table kernels + node-major features (cross-check of the
                                 compile-time-table kernels; lgn/_native.py sets it from LGN_AMD_NO_STATIC at creation) */
/* kernel-selecting cross-check switches, frozen the same way (lgn/_native.py: net_flags; the partial-row counts, whether the loss
 * rides on the last decoder level all follow from them -- a forward, its backward
 * and the workspace sizing can never disagree): */
#define LGN_NET_DEC_PAIRWISE 2   /* LGN_AMD_DEC_PAIRWISE=1: decoder levels as O(N^2) pair sweeps instead of the separable form */
#define LGN_NET_LEVEL_V2 4       /* LGN_AMD_LEVEL_V2=1: three-kernel level backward also for N <= 40 */
#define LGN_NET_MLP_V1 8         /* LGN_AMD_MLP_V1=1: the CGMLP keeps the 12-wave kernels (csrc/mlp_mfma.hip) where the chain kernels
                                    (csrc/mlp_chain.hip: large batches, H = 6 x 2C <= 48) would run; cross-check */
/* (bit 32, and bit 8 before ABI 14: round 4's CGMLP-inside-the-level-kernels switches; measured slower in every regime, removed) */
#define LGN_NET_MOMENTS_V1 16    /* LGN_AMD_MOMENTS_V1=1: component-chunked moments kernels (with LGN_NET_NO_STATIC) */
#define LGN_NET_BWD_ORDERED 64   /* LGN_AMD_BWD_ORDERED=1: the encoder level backward runs its radial-gradient GEMM per ordered pair tile
                                    (the form before round 4's symmetric sweep; cross-check) */
#define LGN_NET_DEC_UNFUSED 256  /* LGN_AMD_DEC_UNFUSED=1: table-driven decoder levels as moments tensor + per-node kernels (the round-5
                                    sequence) instead of the fused separable form of csrc/generic_local_sep.hip; cross-check.  Changes
                                    the activation / scratch layouts: fixed in the descriptor like the others */
#define LGN_NET_MOMENTS_SPLIT 512 /* LGN_AMD_MOMENTS_SPLIT=1: the encoder's table-driven level backward runs its two pair sweeps as two
                                    kernels (moments_bwd_nodes2 + moments_bwd_G2: the round-5 form) instead of the merged one; cross-check */
#define LGN_NET_MLP_BWD1 1024    /* LGN_AMD_MLP_BWD1=1: the chain CGMLP backward as ONE role per wave (four waves per workgroup: chain, weight
                                    gradients and image staging on the same wave -- the round-5 kernel) instead of the two-role kernel
                                    (eight waves: four carry the chain, four the weight gradients and the staging); below 8 129 rows: one chain
                                    wave per 16-row workgroup instead of a layer split over three; cross-check */
#define LGN_NET_SPLIT_TAIL 128   /* LGN_AMD_SPLIT_TAIL=1: the tail of a step (deferred reductions, radial finalisation, L1 + Adam) as the
                                    three separate launches instead of csrc/step_tail.hip's one (cross-check; bit-identical) */

/* Plan-time fit queries: bytes of LDS the largest per-jet end stage needs (one workgroup per jet; the limit is LGN_LDS_LIMIT).
 * encoder: input-stage backward (K input scalars, C0 = first channel count) and the latent stage (CL = last channel count,
 * Ts / Tv latent multiplicities, pool = LGN_POOL code); decoder: its input stage (Tin latent vectors) and output / loss stage;
 * junction: the fused encoder-latent + decoder-input kernels of the whole-step call.  -1: bad pooling code.  A caller whose
 * shape does not fit takes the per-operator path instead (_fused_ok of lgn/models/encoder.py and decoder.py, NativeTrainStep of lgn/step.py). */
#define LGN_LDS_LIMIT (160 * 1024)
long long lgn_encoder_end_lds_bytes(int N, int C0, int K, int CL, int Ts, int Tv, int pool);
long long lgn_decoder_end_lds_bytes(int N, int C0, int Tin, int CL);
long long lgn_junction_lds_bytes(int N, int CL, int Ts, int Tv, int pool, int C0);

int lgn_step_param_slots(const lgn_net_desc* d, int decoder);
long long lgn_step_workspace_doubles(const lgn_net_desc* d);
/* p4 [B][N][4] real Cartesian encoder input (already multiplied by the encoder's `scale`, lgn_encoder.py:376; with jet_features its
 * last node is the jet); target [B][Nd][4] the UNscaled batch the reconstruction is compared with (utils/train.py:285-292; may alias
 * p4 when scale == 1 and Nd == N), Nd = d->dec_N or N; mask [B][N]; in_scalars [B][N][K - 1] or NULL (d->n_in_scalars <= 1);
 * recon [2][B][Nd][4]; loss_part [B].  workspace_doubles = capacity of `workspace`: the call
 * fails before enqueuing anything if the current configuration needs more (lgn_step_workspace_doubles).
 * (ABI 16: the side_stream argument of the forked gradient reductions is gone with that path -- measured slower in every regime.) */
int lgn_step_fwd_bwd_f64(const lgn_net_desc* d, const double* params, double* grads, long long n_params,
                         const int64_t* enc_off, const int64_t* dec_off, const double* p4, const double* target,
                         const uint8_t* mask, const double* in_scalars, double* workspace, long long workspace_doubles,
                         double* recon, double* loss_part, void* stream);
/* grads += l1_lambda*sign(params) (utils/train.py:484-487); loss_out[0..2] = total, chamfer, sum|w|; optional Adam
 * (torch.optim.Adam defaults; the step counter lives on the device so that graph replays stay correct).
 * loss_out must hold 3 + LGN_FINALIZE_SCRATCH doubles: the results, then scratch -- the per-workgroup |w| partials, the cached
 * bias-correction powers {t, beta1^t, beta2^t} of the next odd / even step (checked against the step counter before use: a
 * restored counter or changed betas just recompute them) and, in the last slot, the finished-workgroup counter of the single
 * launch.  The caller zero-fills the block ONCE, at allocation; every kernel that uses it (this call's, lgn_step_train_f64's fused
 * tail) leaves the counters AND the |w| partial slots at zero -- the fused tail reads "zero = not yet written in this launch", so
 * the two calls may be mixed on one block.  A launch that died
 * part-way (device fault) leaves it dirty: zero the block again before reusing it -- with a non-zero counter no workgroup
 * recognises itself as the last one, loss_out[0..2] stay stale and the step counter is not advanced. */
#define LGN_FINALIZE_SCRATCH 2048
int lgn_step_finalize_f64(double* params, double* grads, long long n_params, const double* loss_part, int n_loss,
                          double l1_lambda, double* adam_m, double* adam_v, long long* step_dev,
                          double lr, double beta1, double beta2, double eps, int do_adam, double* loss_out, void* stream);
/* The whole training step of ONE process in one call: lgn_step_fwd_bwd_f64 followed by lgn_step_finalize_f64 (same arguments, same
 * results: gradients incl. the L1 sub-gradient in `grads`, loss terms in loss_out[0..2], Adam applied when do_adam).  With nothing
 * to do between the gradients and the optimiser (no all-reduce) the tail of the step -- the deferred reductions of all partial
 * rows, the radial-gradient finalisation, L1 + Adam, the loss assembly: three dependent launches above -- is ONE launch
 * (csrc/step_tail.hip; bit-identical results; LGN_NET_SPLIT_TAIL in d->flags keeps the three launches).  The table-driven (maxdim 3) step
 * and steps that do not fit the fused form take the three launches by themselves.  loss_out as for lgn_step_finalize_f64; the
 * scratch slots -11 .. -8 of the block are the per-level counters of the fused launch (zero between calls, like the last slot).
 * Data-parallel training keeps the two calls above: the gradient all-reduce sits between them. */
int lgn_step_train_f64(const lgn_net_desc* d, double* params, double* grads, long long n_params, const int64_t* enc_off,
                       const int64_t* dec_off, const double* p4, const double* target, const uint8_t* mask, const double* in_scalars,
                       double* workspace, long long workspace_doubles, double* recon, double* loss_part, int n_loss, double l1_lambda,
                       double* adam_m, double* adam_v, long long* step_dev, double lr, double beta1, double beta2, double eps,
                       int do_adam, double* loss_out, void* stream);

/* ---- one network at a time, maxdim = 2: what LGNEncoder.forward / LGNDecoder.forward (lgn/models/lgn_encoder.py:255-336,
 * lgn_decoder.py:218-303) and autograd's backward of them become under the module API.  Same parameter-slot layout as
 * above, but `params` / `grads` / `off` refer to ONE network's flat parameter block.  *_fwd writes the activations the
 * backward needs into `act` (lgn_net_workspace_doubles(d, decoder, 0) doubles, owned by the caller between the two
 * calls); *_bwd zero-fills `grads`, then writes every parameter gradient (dead parameters keep an exact 0) and uses
 * `scratch` (lgn_net_workspace_doubles(d, decoder, 1) doubles).
 *   encoder: p4 [B][N][4] (already scaled), mask [B][N], in_scalars [B][N][K - 1] (NULL when d->n_in_scalars <= 1: the input
 *            MixReps (0,0) weight is [2][C][K], slot 0) -> lat_s [2][B][P tau_s], lat_v [2][B][P tau_v][4] Cartesian
 *            (P = 2 for the default 'min&max' pooling, see lgn_net_desc.latent_pool); g_lat_s may be NULL (no gradient on the latent scalars: the last level's scalar
 *            branch is then skipped, like autograd would).
 *   decoder: lat_v [2][B][Tin][4] (Tin = tau_v_in, or 2 tau_v when 0) -> recon [2][B][N][4] complex Cartesian;
 *            backward from g_recon [2][B][N][4] to g_lat_v (the latent scalars never reach the output). */
long long lgn_net_workspace_doubles(const lgn_net_desc* d, int decoder, int which);
int lgn_encoder_fwd_f64(const lgn_net_desc* d, const double* params, const int64_t* off, const double* p4, const uint8_t* mask,
                        const double* in_scalars, double* act, long long act_doubles, double* lat_s, double* lat_v, void* stream);
int lgn_encoder_bwd_f64(const lgn_net_desc* d, const double* params, double* grads, long long n_params, const int64_t* off,
                        const double* p4, const uint8_t* mask, const double* in_scalars, const double* act, long long act_doubles,
                        const double* g_lat_s, const double* g_lat_v, double* scratch, long long scratch_doubles, void* stream);
int lgn_decoder_fwd_f64(const lgn_net_desc* d, const double* params, const int64_t* off, const double* lat_v, double* act,
                        long long act_doubles, double* recon, void* stream);
int lgn_decoder_bwd_f64(const lgn_net_desc* d, const double* params, double* grads, long long n_params, const int64_t* off,
                        const double* lat_v, const double* act, long long act_doubles, const double* g_recon, double* g_lat_v,
                        double* scratch, long long scratch_doubles, void* stream);

/* ---- Chamfer loss on its own (module API: lgn/losses.py ChamferLoss, the drop-in of utils/losses/chamfer_loss/chamfer_loss.py:7-31;
 * the whole-step call has the loss inside the decoder's last kernel).  x [B][N][4], y [B][M][4] real 4-vectors, cdist = sum of
 * squared component differences (distance_sq.py:263-304, even p: no eps).
 *   loss_part[b] = (sum_i min_j d_ij + sum_j min_i d_ij) / 2  [+ sum_mu (sum_i x - sum_j y)_mu^2 / (4 B) with jet_features: the
 *                  nn.MSELoss() of the jet momenta, chamfer_loss.py:25-29];  the loss is the sum of loss_part over the batch
 *   gx [B][N][4], gy [B][M][4] = d loss / d x, d loss / d y (first minimum on ties, as torch.min; gy may be needed for a target that
 *                  requires grad -- always written) */
int lgn_chamfer_f64(int B, int N, int M, const double* x, const double* y, int jet_features, double* loss_part, double* gx, double* gy,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LGN_AMD_H */
