// lgn-autoencoder_amd/csrc/net.hpp -- host launchers of net_kernels.hip (the O(N)-per-jet network ends).
#pragma once
#include "ops.hpp"
#include "../../include/lgn_amd.h"

namespace lgn {
int enc_input_fwd(int B, int N, int C, const double* p4, const double* w0, const double* w1, double* s, double* v, hipStream_t);
int enc_input_bwd(int B, int N, int C, const double* p4, const double* g_s, const double* g_v, double* part /*[B][4C]*/, hipStream_t);
int enc_latent_fwd(int B, int N, int C, int Ts, int Tv, const double* s, const double* v, const double* wl0, const double* wl1,
                   double* lat_s, double* lat_v, int* idx, hipStream_t);
int enc_latent_bwd(int B, int N, int C, int Ts, int Tv, const double* s, const double* v, const double* wl0, const double* wl1,
                   const double* g_lat_s, const double* g_lat_v, const int* idx, double* g_s, double* g_v,
                   double* part /*[B][2(Ts+Tv)C]*/, hipStream_t);
int dec_input_fwd(int B, int N, int C, int Tin, const double* lat_v, const double* wg1, const double* w0, const double* w1,
                  double* pdec, double* s0, double* v0, hipStream_t);
int dec_input_bwd(int B, int N, int C, int Tin, const double* lat_v, const double* wg1, const double* w1, const double* pdec,
                  const double* g_p, const double* g_s0, const double* g_v0, double* g_lat_v,
                  double* part /*[B][4C + 2 N Tin]*/, hipStream_t);
int dec_output_loss(int B, int N, int C, const double* v, const double* wo1, const double* target, double loss_scale, double* recon,
                    double* loss_part /*[B]*/, double* g_v, double* part /*[B][2C]*/, hipStream_t);
int finalize_step(double* w, double* g, long n, const double* loss_part, int nB, double lambda, double* m, double* v, long* step_dev,
                  double lr, double beta1, double beta2, double eps, int do_adam, double* loss_out, hipStream_t st);
}  // namespace lgn
