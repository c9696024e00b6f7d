// lgn-autoencoder_amd/csrc/net.hpp -- host launchers of net_kernels.hip (the O(N)-per-jet network ends).
#pragma once
#include "ops.hpp"
#include "../../include/lgn_amd.h"

namespace lgn {
// latent pooling code of include/lgn_amd.h (LGN_POOL): 0 stands for min&max
__host__ __device__ inline int pool_canon(int code) { return code ? code : LGN_POOL(2, 0, LGN_POOL_MIN, LGN_POOL_MAX, 0, 0); }
__host__ __device__ inline int pool_n(int code) { return pool_canon(code) & 7; }
__host__ __device__ inline int pool_avg(int code) { return (pool_canon(code) >> 3) & 1; }
__host__ __device__ inline int pool_op(int code, int i) { return (pool_canon(code) >> (4 + 2 * i)) & 3; }
__host__ __device__ inline int pool_blocks(int code) { return pool_avg(code) ? 1 : pool_n(code); }   // P: output blocks per channel
// 'mix': no pooling at all -- the latent MixReps acts on the N C (particle, channel) pairs of the jet (lgn_encoder.py:226-232,313-319)
__host__ __device__ inline bool pool_is_mix(int code) { return code == LGN_POOL(1, 0, LGN_POOL_MIX, 0, 0, 0); }
inline bool pool_valid(int code) {
  if (pool_is_mix(code)) return true;
  if (code < 0 || code >= (1 << 12) || pool_n(code) < 1 || pool_n(code) > 4) return false;
  for (int i = 0; i < 4; ++i)
    if (pool_op(code, i) > LGN_POOL_MEAN || (i >= pool_n(code) && pool_op(code, i) != 0)) return false;
  return true;
}
// input channels of the latent MixReps weights: C, or N C under 'mix'
__host__ __device__ inline int pool_mix_in(int code, int N, int C) { return pool_is_mix(code) ? N * C : C; }
// z1/z2 (optional): buffers zeroed by the same launch (the step folds its two memsets into this first kernel)
// K input scalars per node: the mass, then xs [B][N][K-1] (K = 1: xs unused); w0 = MixReps weight [2][C][K]
int enc_input_fwd(int B, int N, int C, int K, const double* p4, const double* xs, const double* w0, const double* w1, double* s, double* v,
                  hipStream_t, double* z1 = nullptr, size_t n1 = 0, double* z2 = nullptr, size_t n2 = 0);
// enc_latent_fwd + dec_input_fwd (resp. dec_input_bwd + enc_latent_bwd) of a jet in one launch
int junction_fwd(int B, int N, int CL, int Ts, int Tv, int pool, const double* s, const double* v, const double* wl0, const double* wl1,
                 double* lat_s, double* lat_v, int* idx, int C0, const double* wg1, const double* w0, const double* w1, double* pdec,
                 double* s0, double* v0, hipStream_t);
int junction_bwd(int B, int N, int C0, int Tin, const double* lat_v, const double* wg1, const double* w1, const double* pdec,
                 const double* g_p, const double* g_s0, const double* g_v0, double* g_lat_v, double* part_dec, int CL, int Ts, int Tv,
                 int pool, const double* s, const double* v, const double* wl0, const double* wl1, const double* g_lat_s, const int* idx,
                 double* g_s, double* g_v, double* part_enc, hipStream_t);
int enc_input_bwd(int B, int N, int C, int K, const double* p4, const double* xs, const double* g_s, const double* g_v,
                  double* part /*[B][(2K + 2) C]: dW00 re [C][K], im [C][K] | dW11 re[C], im[C]*/, hipStream_t);
int enc_latent_fwd(int B, int N, int C, int Ts, int Tv, int pool, const double* s, const double* v, const double* wl0, const double* wl1,
                   double* lat_s, double* lat_v, int* idx, hipStream_t);
int enc_latent_bwd(int B, int N, int C, int Ts, int Tv, int pool, const double* s, const double* v, const double* wl0, const double* wl1,
                   const double* g_lat_s, const double* g_lat_v, const int* idx, double* g_s, double* g_v,
                   double* part /*[B][2(Ts+Tv)C]*/, hipStream_t);
int dec_input_fwd(int B, int N, int C, int Tin, const double* lat_v, const double* wg1, const double* w0, const double* w1,
                  double* pdec, double* s0, double* v0, hipStream_t);
int dec_input_bwd(int B, int N, int C, int Tin, const double* lat_v, const double* wg1, const double* w1, const double* pdec,
                  const double* g_p, const double* g_s0, const double* g_v0, double* g_lat_v,
                  double* part /*[B][4C + 2 N Tin]*/, hipStream_t);
int dec_output_loss(int B, int N, int C, const double* v, const double* wo1, const double* target, double loss_scale, double* recon,
                    double* loss_part /*[B]*/, double* g_v, double* part /*[B][2C]*/, hipStream_t);
// Chamfer loss per jet and its gradients (module API: lgn/losses.py); loss_part [B], gx [B][N][4], gy [B][M][4]
int chamfer_fwd(int B, int N, int M, const double* x, const double* y, int jet_features, double* loss_part, double* gx, double* gy,
                hipStream_t);
// decoder output without the loss (module API): recon [2][B][N][4]; backward from g_recon, part [B][2C]
int dec_output_fwd(int B, int N, int C, const double* v, const double* wo1, double* recon, hipStream_t);
int dec_output_bwd(int B, int N, int C, const double* v, const double* wo1, const double* g_recon, double* g_v, double* part, hipStream_t);
// LDS bytes of the largest per-jet end stage (plan-time fit queries; net_kernels.hip)
size_t encoder_end_lds_bytes(int N, int C0, int K, int CL, int Ts, int Tv, int pool);
size_t decoder_end_lds_bytes(int N, int C0, int Tin, int CL);
size_t junction_lds_bytes(int N, int CL, int Ts, int Tv, int pool, int C0);
int finalize_step(double* w, double* g, long n, const double* loss_part, int nB, double lambda, double* m, double* v, long* step_dev,
                  double lr, double beta1, double beta2, double eps, int do_adam, double* loss_out, hipStream_t st);
// LocalArgs from the C-ABI table struct (shared by api.hip and step.hip)
inline int local_args(LocalArgs& a, int nodes, int C, int CO, int Q, int Qout, const lgn_local_tables* t) {
  LGN_CHECK_ARG(t && t->row_ptr && t->t_type && t->t_a && t->t_b && t->t_coef && t->out_dim && t->out_nblk && t->out_row0 &&
                    t->out_q0 && t->out_w0 && t->u_ptr && t->u_row && t->u_coef && t->x_ptr && t->x_row && t->x_other && t->x_coef,
                "local: incomplete tables");
  LGN_CHECK_ARG(t->n_terms > 0 && t->n_u >= 0 && t->n_x >= 0, "local: table lengths missing");
  a.nodes = nodes; a.C = C; a.CO = CO; a.Q = Q; a.Qout = Qout;
  a.n_terms = t->n_terms; a.n_u = t->n_u; a.n_x = t->n_x; a.n_units = t->n_units;
  a.t = LocalTables{t->n_rows, t->n_out, t->n_w, t->row_ptr, t->t_type, t->t_a, t->t_b, t->t_coef, t->out_dim, t->out_nblk,
                    t->out_row0, t->out_q0, t->out_w0, t->u_ptr, t->u_row, t->u_coef, t->x_ptr, t->x_row, t->x_other, t->x_coef};
  return 0;
}

}  // namespace lgn
