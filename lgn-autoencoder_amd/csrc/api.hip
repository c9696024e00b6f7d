// lgn-autoencoder_amd/csrc/api.hip -- extern "C" entry points declared in include/lgn_amd.h
#include "net.hpp"

using namespace lgn;

extern "C" {

int lgn_level_fwd_f64(int B, int N, int C, int CO, int decoder, const double* s_in, const double* v_in, const double* p,
                      const uint8_t* mask, const double* ra, const double* rb, const double* rc, const double* w0,
                      const double* b0, const double* w1, const double* b1, const double* wm0, const double* wm1,
                      double* ag0, double* ag1, double* s_out, double* v_out, void* stream) {
  LGN_CHECK_ARG(s_in && v_in && p && b0 && b1 && wm0 && wm1 && ag0 && ag1 && s_out && v_out, "level_fwd: null pointer");
  LGN_CHECK_ARG(decoder || (mask && ra && rb && rc && w0 && w1), "level_fwd: encoder needs mask and radial parameters");
  LevelArgs<double> a{B, N, C, CO, s_in, v_in, p, mask, ra, rb, rc, w0, b0, w1, b1, wm0, wm1, ag0, ag1, s_out, v_out};
  a.flags = level_flags_from_env();
  return level_fwd_dispatch<double>(a, decoder, (hipStream_t)stream);
}

int lgn_level_bwd_partial_rows(int B, int N, int decoder, int* rows_mix, int* rows_rad) {
  LGN_CHECK_ARG(B > 0 && N > 0 && rows_mix && rows_rad, "level_bwd_partial_rows: bad arguments");
  level_bwd_partial_rows(B, N, decoder, level_flags_from_env(), rows_mix, rows_rad);
  return 0;
}

int lgn_level_rad_partial_len(int C, int decoder) { return rad_partial_size(C, decoder != 0); }
int lgn_level_jet_split(int B, int N) { return N <= 40 ? level_jet_split(B, N) : 1; }

int lgn_level_bwd_f64(int B, int N, int C, int CO, int decoder, const double* s_in, const double* v_in, const double* p,
                      const uint8_t* mask, const double* ra, const double* rb, const double* rc, const double* w0,
                      const double* b0, const double* w1, const double* b1, const double* wm0, const double* wm1,
                      const double* ag0, const double* ag1, const double* g_s_out, const double* g_v_out, double* g_ag,
                      double* g_s_in, double* g_v_in, double* g_p, double* part_mix, double* part_rad, void* stream) {
  LGN_CHECK_ARG(s_in && v_in && p && b0 && b1 && wm0 && wm1 && ag0 && ag1 && g_s_out && g_v_out && g_ag && g_s_in &&
                    g_v_in && part_mix && part_rad, "level_bwd: null pointer");
  LGN_CHECK_ARG(decoder ? (g_p != nullptr) : (mask && ra && rb && rc && w0 && w1), "level_bwd: missing decoder g_p / encoder radial parameters");
  LevelBwdArgs<double> a{B, N, C, CO, s_in, v_in, p, mask, ra, rb, rc, w0, b0, w1, b1, wm0, wm1, ag0, ag1,
                         g_s_out, g_v_out, g_ag, g_s_in, g_v_in, g_p, part_mix, part_rad};
  a.flags = level_flags_from_env();
  return level_bwd_dispatch<double>(a, decoder, (hipStream_t)stream);
}

int lgn_cg_product_fwd_f64(int R, int N, int C, int D1, int D2, int DO, int mode, int nnz, const int* row_ptr, const int* col,
                           const double* coef, const double* x1, const double* x2, double* out, void* stream) {
  return cg_product_fwd(R, N, C, D1, D2, DO, mode, nnz, row_ptr, col, coef, x1, x2, out, (hipStream_t)stream);
}
int lgn_cg_product_bwd_f64(int R, int N, int C, int D1, int D2, int DO, int mode, int nnz, const int* row_ptr, const int* col,
                           const double* coef, const double* x1, const double* x2, const double* g_out, double* g_x1, double* g_x2,
                           void* stream) {
  return cg_product_bwd(R, N, C, D1, D2, DO, mode, nnz, row_ptr, col, coef, x1, x2, g_out, g_x1, g_x2, (hipStream_t)stream);
}

int lgn_reduce_partials_f64(const double* part, int rows, int n, double* out, int accumulate, void* stream) {
  LGN_CHECK_ARG(part && out && rows > 0 && n >= 0, "reduce_partials: bad arguments");
  return reduce_partials<double>(part, rows, n, out, accumulate, (hipStream_t)stream);
}

int lgn_radial_finalize_f64(const double* tot, int C, const double* ra, const double* rb, const double* rc, const double* w0,
                            const double* w1, double* g_a, double* g_b, double* g_c, double* g_w0, double* g_b0,
                            double* g_w1, double* g_b1, void* stream) {
  LGN_CHECK_ARG(tot && ra && rb && rc && w0 && w1 && g_a && g_b && g_c && g_w0 && g_b0 && g_w1 && g_b1 && C >= 1 && C <= 8,
                "radial_finalize: bad arguments");
  return rad_finalize<double>(tot, C, ra, rb, rc, w0, w1, g_a, g_b, g_c, g_w0, g_b0, g_w1, g_b1, (hipStream_t)stream);
}

static int fill_mlp(MlpArgs<double>& a, int M, int C, int H, int nlin, int activation, const double* const* w, const double* const* b) {
  LGN_CHECK_ARG(nlin >= 1 && nlin <= MLP_MAX_LIN && w && b, "cgmlp: bad layer list (nlin=%d)", nlin);
  a.M = M; a.C = C; a.H = H; a.nlin = nlin; a.act = activation;
  for (int l = 0; l < MLP_MAX_LIN; ++l) {
    a.w[l] = l < nlin ? w[l] : nullptr;
    a.b[l] = l < nlin ? b[l] : nullptr;
    if (l < nlin) LGN_CHECK_ARG(w[l] && b[l], "cgmlp: null weight pointer for layer %d", l);
  }
  return 0;
}

int lgn_cgmlp_fwd_f64(int M, int C, int H, int nlin, int activation, const double* const* w, const double* const* b,
                      const double* s_in, double* s_out, void* stream) {
  MlpArgs<double> a{};
  if (int rc = fill_mlp(a, M, C, H, nlin, activation, w, b)) return rc;
  LGN_CHECK_ARG(s_in && s_out, "cgmlp_fwd: null pointer");
  a.s_in = s_in; a.s_out = s_out;
  a.flags = level_flags_from_env();
  return mlp_dispatch<double>(a, false, (hipStream_t)stream);
}

int lgn_cgmlp_partial_rows(int M, int H) { return mlp_partial_rows(M, H); }

int lgn_cgmlp_bwd_f64(int M, int C, int H, int nlin, int activation, const double* const* w, const double* const* b,
                      const double* s_in, const double* g_out, double* g_in, double* part, int psize, void* stream) {
  MlpArgs<double> a{};
  if (int rc = fill_mlp(a, M, C, H, nlin, activation, w, b)) return rc;
  LGN_CHECK_ARG(s_in && g_out && g_in && part, "cgmlp_bwd: null pointer");
  a.s_in = s_in; a.g_out = g_out; a.g_in = g_in; a.part = part; a.psize = psize;
  a.flags = level_flags_from_env();
  return mlp_dispatch<double>(a, true, (hipStream_t)stream);
}

int lgn_mixreps_fwd_f64(int rows, int Cin, int Cout, int d, const double* w, const double* x, double* y, void* stream) {
  LGN_CHECK_ARG(w && x && y, "mixreps_fwd: null pointer");
  MixArgs<double> a{rows, Cin, Cout, d, w, x, y, nullptr, nullptr, nullptr};
  return mix_fwd<double>(a, (hipStream_t)stream);
}

int lgn_mixreps_partial_rows(int rows) { return mix_partial_rows(rows); }

int lgn_chamfer_f64(int B, int N, int M, const double* x, const double* y, int jet_features, double* loss_part, double* gx, double* gy,
                    void* stream) {
  LGN_CHECK_ARG(B > 0 && N > 0 && M > 0, "chamfer: empty input (B=%d N=%d M=%d)", B, N, M);
  LGN_CHECK_ARG(x && y && loss_part && gx && gy, "chamfer: null pointer");
  return chamfer_fwd(B, N, M, x, y, jet_features, loss_part, gx, gy, (hipStream_t)stream);
}

int lgn_mixreps_bwd_f64(int rows, int Cin, int Cout, int d, const double* w, const double* x, const double* g_y, double* g_x,
                        double* part, void* stream) {
  LGN_CHECK_ARG(w && x && g_y && part, "mixreps_bwd: null pointer");
  MixArgs<double> a{rows, Cin, Cout, d, w, x, nullptr, g_y, g_x, part};
  return mix_bwd<double>(a, (hipStream_t)stream);
}

static GenArgs gen_args(int B, int N, int C, int Q, const double* X, const double* p, const uint8_t* mask, const double* ra,
                        const double* rb, const double* rc, const double* w0, const double* b0, const double* w1, const double* b1) {
  GenArgs a{};
  a.B = B; a.N = N; a.C = C; a.Q = Q; a.X = X; a.p = p; a.mask = mask;
  a.ra = ra; a.rb = rb; a.rc = rc; a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1;
  a.flags = level_flags_from_env();
  return a;
}

int lgn_moments_fwd_f64(int B, int N, int C, int Q, int decoder, const double* X, const double* p, const uint8_t* mask,
                        const double* ra, const double* rb, const double* rc, const double* w0, const double* b0,
                        const double* w1, const double* b1, double* U, void* stream) {
  LGN_CHECK_ARG(X && p && b0 && b1 && U, "moments_fwd: null pointer");
  LGN_CHECK_ARG(decoder || (mask && ra && rb && rc && w0 && w1), "moments_fwd: encoder needs mask and radial parameters");
  GenArgs a = gen_args(B, N, C, Q, X, p, mask, ra, rb, rc, w0, b0, w1, b1);
  a.U = U;
  return moments_dispatch(a, decoder, 0, (hipStream_t)stream);
}

long long lgn_moments_scratch_doubles(int B, int N, int C, int decoder) {
  return (decoder || N > 32) ? 0 : (long long)moments2_gbuf_doubles(B, N, C);
}

int lgn_moments_bwd_f64(int B, int N, int C, int Q, int decoder, const double* X, const double* p, const uint8_t* mask,
                        const double* ra, const double* rb, const double* rc, const double* w0, const double* b0,
                        const double* w1, const double* b1, const double* gU, double* gX, double* g_p, double* part_rad,
                        double* scratch, void* stream) {
  LGN_CHECK_ARG(X && p && b0 && b1 && gU && gX && part_rad, "moments_bwd: null pointer");
  LGN_CHECK_ARG(decoder ? (g_p != nullptr) : (mask && ra && rb && rc && w0 && w1), "moments_bwd: missing decoder g_p / encoder radial parameters");
  GenArgs a = gen_args(B, N, C, Q, X, p, mask, ra, rb, rc, w0, b0, w1, b1);
  a.gU = gU; a.gX = gX; a.g_p = g_p; a.part_rad = part_rad; a.gbuf = scratch;
  if (int rc2 = moments_dispatch(a, decoder, 1, (hipStream_t)stream)) return rc2;
  return moments_dispatch(a, decoder, 2, (hipStream_t)stream);
}

int lgn_local_fwd_f64(int nodes, int C, int CO, int Q, int Qout, const lgn_local_tables* t, const double* X, const double* U,
                      const double* wcat, double* out, void* stream) {
  LocalArgs a{};
  if (int rc = local_args(a, nodes, C, CO, Q, Qout, t)) return rc;
  LGN_CHECK_ARG(X && U && wcat && out, "local_fwd: null pointer");
  a.X = X; a.U = U; a.wcat = wcat; a.out = out;
  return local_fwd(a, (hipStream_t)stream);
}

int lgn_local_partial_rows(int nodes) { return local_partial_rows(nodes); }

int lgn_local_bwd_static_f64(int kind, int nodes, int C, int CO, const double* XT, const double* UT, const double* wcat, const int* w0,
                             double* wpacked, const double* goT, double* gUT, double* gXT, double* part, double* gpacked, double* g_wcat,
                             void* stream) {
  LGN_CHECK_ARG(XT && UT && wcat && w0 && wpacked && goT && gUT && gXT && part && gpacked && g_wcat, "local_bwd_static: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = local_bwd_static(kind, nodes, C, CO, XT, UT, wcat, w0, wpacked, goT, gUT, gXT, part, st)) return rc;
  const int np = (int)local_static_packed_doubles(kind, C, CO), tiles = (nodes + 63) / 64;
  if (int rc = reduce_partials<double>(part, tiles, np, gpacked, 0, st)) return rc;
  return local_static_unpack_grads(kind, C, CO, w0, gpacked, g_wcat, st);
}

long long lgn_local_static_packed_doubles(int kind, int C, int CO) { return (long long)local_static_packed_doubles(kind, C, CO); }

int lgn_local_fwd_static_f64(int kind, int nodes, int C, int CO, const double* XT, const double* UT, const double* wcat, const int* w0,
                             double* wpacked, double* outT, double* s_copy, int q_s, void* stream) {
  LGN_CHECK_ARG(XT && UT && wcat && w0 && wpacked && outT, "local_fwd_static: null pointer");
  return local_fwd_static(kind, nodes, C, CO, XT, UT, wcat, w0, wpacked, outT, s_copy, q_s, (hipStream_t)stream);
}

int lgn_local_bwd_f64(int nodes, int C, int CO, int Q, int Qout, const lgn_local_tables* t, const double* X, const double* U,
                      const double* wcat, const double* g_out, double* gU, double* gX, double* part, void* stream) {
  LocalArgs a{};
  if (int rc = local_args(a, nodes, C, CO, Q, Qout, t)) return rc;
  LGN_CHECK_ARG(X && U && wcat && g_out && gU && gX && part, "local_bwd: null pointer");
  a.X = X; a.U = U; a.wcat = wcat; a.g_out = g_out; a.gU = gU; a.gX = gX; a.part = part;
  return local_bwd(a, (hipStream_t)stream);
}

}  // extern "C"
