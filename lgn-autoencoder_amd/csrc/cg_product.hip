// lgn-autoencoder_amd/csrc/cg_product.hip -- the Clebsch-Gordan product of two irreps as an operator of its own (round 5).
//
// Reference: cg_product / complex_kron_product, lgn/cg_lib/cg_ops.py:135-218 and :221-297 -- per channel the Kronecker product
// x1[m1] x2[m2] of two irrep components, optionally summed over the neighbour index first (aggregate), then multiplied by the
// stacked Clebsch-Gordan matrix [sum_r dim r][d1 d2] of the pair.  Inside the networks this product never exists by itself: the
// level kernels fuse it with the edge network and CatMix (level_fwd2.hip, generic_moments2.hip + generic_local_static.hip).  This
// file serves lgn.cg_lib.cg_product / CGProduct of the module API (the reference exports them) -- a small table-driven kernel,
// thread = (row, channel), the matrix in CSR form over its non-zeros; not a hot path.
//   mode 0            x1 [2][R][C][D1], x2 [2][R][C][D2]                      out[r] = H (x1[r] (x) x2[r])
//   mode 1 aggregate  x1 [2][B][N][N][C][D1] (edge-like), x2 [2][B][N][C][D2]  out[b,i] = H sum_j x1[b,i,j] (x) x2[b,j]     R = B N
//   mode 2 aggregate  x1 [2][B][N][C][D1], x2 [2][B][N][N][C][D2]              out[b,i] = H sum_j x1[b,j] (x) x2[b,i,j]
// out [2][R][C][DO]; term t of output row o: coefficient coef[t], column col[t] = m1 * D2 + m2, rows delimited by row_ptr[DO + 1].
// The sums run in index order (j, then the terms of a row): deterministic.
#include "ops.hpp"

namespace lgn {
namespace {

struct CgArgs {
  int R, N, C, D1, D2, DO, mode;
  const int* __restrict__ row_ptr;      // [DO + 1]
  const int* __restrict__ col;          // [nnz]
  const double* __restrict__ coef;      // [nnz]
  const double* __restrict__ x1;
  const double* __restrict__ x2;
  double* out;
  const double* __restrict__ g_out;     // backward
  double* g_x1;
  double* g_x2;
};

// element index of (row r [, neighbour j], channel c, component m) in an operand and the plane stride of that operand
__device__ __forceinline__ size_t node_at(const CgArgs& a, int r, int c, int D, int m) { return ((size_t)r * a.C + c) * D + m; }
__device__ __forceinline__ size_t edge_at(const CgArgs& a, int r, int j, int c, int D, int m) { return (((size_t)r * a.N + j) * a.C + c) * D + m; }

__global__ __launch_bounds__(BLOCK) void cg_product_fwd_kernel(CgArgs a) {
  const size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (e >= (size_t)a.R * a.C) return;
  const int r = (int)(e / a.C), c = (int)(e % a.C);
  const size_t p1 = (size_t)a.R * (a.mode == 1 ? a.N : 1) * a.C * a.D1, p2 = (size_t)a.R * (a.mode == 2 ? a.N : 1) * a.C * a.D2;
  const size_t po = (size_t)a.R * a.C * a.DO;
  const int b0 = a.mode ? (r / a.N) * a.N : 0;                  // first row of this row's jet
  for (int o = 0; o < a.DO; ++o) {
    double accr = 0.0, acci = 0.0;
    const int nj = a.mode ? a.N : 1;
    for (int j = 0; j < nj; ++j) {
      for (int t = a.row_ptr[o]; t < a.row_ptr[o + 1]; ++t) {
        const int m1 = a.col[t] / a.D2, m2 = a.col[t] % a.D2;
        const size_t i1 = a.mode == 1 ? edge_at(a, r, j, c, a.D1, m1) : node_at(a, a.mode == 2 ? b0 + j : r, c, a.D1, m1);
        const size_t i2 = a.mode == 2 ? edge_at(a, r, j, c, a.D2, m2) : node_at(a, a.mode == 1 ? b0 + j : r, c, a.D2, m2);
        const double ar = a.x1[i1], ai = a.x1[p1 + i1], br = a.x2[i2], bi = a.x2[p2 + i2], cf = a.coef[t];
        accr += cf * (ar * br - ai * bi);
        acci += cf * (ar * bi + ai * br);
      }
    }
    const size_t io = node_at(a, r, c, a.DO, o);
    a.out[io] = accr;
    a.out[po + io] = acci;
  }
}

// backward, receiver side: thread (r, c) owns g of everything indexed by its own row -- the node-like operand of mode 0, the
// edge-like operand of modes 1 / 2 (rows [r][j]); with g = coef * g_out[o]:  g_x1 += g conj(x2),  g_x2 += g conj(x1)
__global__ __launch_bounds__(BLOCK) void cg_product_bwd_own_kernel(CgArgs a) {
  const size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (e >= (size_t)a.R * a.C) return;
  const int r = (int)(e / a.C), c = (int)(e % a.C);
  const size_t p1 = (size_t)a.R * (a.mode == 1 ? a.N : 1) * a.C * a.D1, p2 = (size_t)a.R * (a.mode == 2 ? a.N : 1) * a.C * a.D2;
  const size_t po = (size_t)a.R * a.C * a.DO;
  const int b0 = a.mode ? (r / a.N) * a.N : 0;
  const int nj = a.mode ? a.N : 1;
  for (int j = 0; j < nj; ++j)
    for (int o = 0; o < a.DO; ++o) {
      const size_t io = node_at(a, r, c, a.DO, o);
      const double gr0 = a.g_out[io], gi0 = a.g_out[po + io];
      for (int t = a.row_ptr[o]; t < a.row_ptr[o + 1]; ++t) {
        const int m1 = a.col[t] / a.D2, m2 = a.col[t] % a.D2;
        const size_t i1 = a.mode == 1 ? edge_at(a, r, j, c, a.D1, m1) : node_at(a, a.mode == 2 ? b0 + j : r, c, a.D1, m1);
        const size_t i2 = a.mode == 2 ? edge_at(a, r, j, c, a.D2, m2) : node_at(a, a.mode == 1 ? b0 + j : r, c, a.D2, m2);
        const double gr = a.coef[t] * gr0, gi = a.coef[t] * gi0;
        if (a.mode != 2 && a.g_x1) {                            // x1 is indexed by this thread's row: it owns g_x1 there
          const double br = a.x2[i2], bi = a.x2[p2 + i2];
          a.g_x1[i1] += gr * br + gi * bi;
          a.g_x1[p1 + i1] += gi * br - gr * bi;
        }
        if (a.mode != 1 && a.g_x2) {
          const double ar = a.x1[i1], ai = a.x1[p1 + i1];
          a.g_x2[i2] += gr * ar + gi * ai;
          a.g_x2[p2 + i2] += gi * ar - gr * ai;
        }
      }
    }
}
// backward, source side of the aggregate: thread (node s, c) collects what every receiver i of its jet sends to node s
__global__ __launch_bounds__(BLOCK) void cg_product_bwd_src_kernel(CgArgs a) {
  const size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (e >= (size_t)a.R * a.C) return;
  const int s = (int)(e / a.C), c = (int)(e % a.C);
  const size_t p1 = (size_t)a.R * (a.mode == 1 ? a.N : 1) * a.C * a.D1, p2 = (size_t)a.R * (a.mode == 2 ? a.N : 1) * a.C * a.D2;
  const size_t po = (size_t)a.R * a.C * a.DO;
  const int b0 = (s / a.N) * a.N, j = s - b0;
  for (int ii = 0; ii < a.N; ++ii) {
    const int i = b0 + ii;
    for (int o = 0; o < a.DO; ++o) {
      const size_t io = node_at(a, i, c, a.DO, o);
      const double gr0 = a.g_out[io], gi0 = a.g_out[po + io];
      for (int t = a.row_ptr[o]; t < a.row_ptr[o + 1]; ++t) {
        const int m1 = a.col[t] / a.D2, m2 = a.col[t] % a.D2;
        const double gr = a.coef[t] * gr0, gi = a.coef[t] * gi0;
        if (a.mode == 1) {                                      // node-like x2[s] <- edge-like x1[i][j = s]
          const size_t i1 = edge_at(a, i, j, c, a.D1, m1), i2 = node_at(a, s, c, a.D2, m2);
          const double ar = a.x1[i1], ai = a.x1[p1 + i1];
          a.g_x2[i2] += gr * ar + gi * ai;
          a.g_x2[p2 + i2] += gi * ar - gr * ai;
        } else {
          const size_t i1 = node_at(a, s, c, a.D1, m1), i2 = edge_at(a, i, j, c, a.D2, m2);
          const double br = a.x2[i2], bi = a.x2[p2 + i2];
          a.g_x1[i1] += gr * br + gi * bi;
          a.g_x1[p1 + i1] += gi * br - gr * bi;
        }
      }
    }
  }
}

// one thread per (row, channel): the product in size_t (edge-like batches reach 2^31 / C rows before they reach the memory limit)
unsigned grid_of(int R, int C) { return (unsigned)(((size_t)R * C + BLOCK - 1) / BLOCK); }

int check(const CgArgs& a, int nnz) {
  LGN_CHECK_ARG(a.R > 0 && a.C > 0 && a.D1 > 0 && a.D2 > 0 && a.DO > 0 && nnz >= 0, "cg_product: empty operand (R=%d C=%d D1=%d D2=%d DO=%d)",
                a.R, a.C, a.D1, a.D2, a.DO);
  LGN_CHECK_ARG(a.mode >= 0 && a.mode <= 2, "cg_product: mode %d (0 plain, 1 / 2 aggregate with the first / second operand edge-like)", a.mode);
  LGN_CHECK_ARG(a.mode == 0 || (a.N > 0 && a.R % a.N == 0), "cg_product: aggregate needs rows = B * N (R=%d N=%d)", a.R, a.N);
  LGN_CHECK_ARG(a.row_ptr && a.col && a.coef && a.x1 && a.x2, "cg_product: null pointer");
  LGN_CHECK_ARG(((size_t)a.R * a.C + BLOCK - 1) / BLOCK <= 0x7fffffffu, "cg_product: %d rows x %d channels exceed one launch", a.R, a.C);
  return 0;
}

}  // namespace

int cg_product_fwd(int R, int N, int C, int D1, int D2, int DO, int mode, int nnz, const int* row_ptr, const int* col, const double* coef,
                   const double* x1, const double* x2, double* out, hipStream_t st) {
  CgArgs a{R, N, C, D1, D2, DO, mode, row_ptr, col, coef, x1, x2, out, nullptr, nullptr, nullptr};
  if (int rc = check(a, nnz)) return rc;
  LGN_CHECK_ARG(out, "cg_product: null output");
  hipLaunchKernelGGL(cg_product_fwd_kernel, dim3(grid_of(R, C)), dim3(BLOCK), 0, st, a);
  LGN_CHECK_LAUNCH();
  return 0;
}
// g_x1 / g_x2 are ACCUMULATED into (the caller zero-fills them); either may be null -- that operand is data, its gradient is not
// computed (the launch that would only produce it is skipped)
int cg_product_bwd(int R, int N, int C, int D1, int D2, int DO, int mode, int nnz, const int* row_ptr, const int* col, const double* coef,
                   const double* x1, const double* x2, const double* g_out, double* g_x1, double* g_x2, hipStream_t st) {
  CgArgs a{R, N, C, D1, D2, DO, mode, row_ptr, col, coef, x1, x2, nullptr, g_out, g_x1, g_x2};
  if (int rc = check(a, nnz)) return rc;
  LGN_CHECK_ARG(g_out && (g_x1 || g_x2), "cg_product_bwd: null pointer (g_out, and at least one of g_x1 / g_x2)");
  const unsigned grid = grid_of(R, C);
  // receiver side: g_x1 (modes 0, 1) / g_x2 (modes 0, 2); source side of an aggregate: the other operand
  const bool own = (mode != 2 && g_x1) || (mode != 1 && g_x2), src = mode && (mode == 1 ? g_x2 : g_x1);
  if (own) {
    hipLaunchKernelGGL(cg_product_bwd_own_kernel, dim3(grid), dim3(BLOCK), 0, st, a);
    LGN_CHECK_LAUNCH();
  }
  if (src) {
    hipLaunchKernelGGL(cg_product_bwd_src_kernel, dim3(grid), dim3(BLOCK), 0, st, a);
    LGN_CHECK_LAUNCH();
  }
  return 0;
}

}  // namespace lgn
