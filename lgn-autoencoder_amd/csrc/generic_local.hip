// lgn-autoencoder_amd/csrc/generic_local.hip -- per-node part of a message-passing level for arbitrary irreps:
// Clebsch-Gordan contraction of the neighbour moments (aggregate) and of node (x) node (power), concatenation
// with the node features and the complex CatMix -- forward and backward.
//
// Reference: cg_product (lgn/cg_lib/cg_ops.py:177-218: CG matrix applied after the neighbour sum, outputs of all
// contributing irrep pairs concatenated on the channel axis), CatReps/CatMixReps (lgn/nn/g_nn.py:160-190,260-278).
// The CG tables are sparse (5 726 non-zeros of 38 416 at maxdim 3); the host (lgn/plan.py) flattens
// "which products feed which concatenated row" into CSR term lists, so the device code is table driven:
//   cat[row][c] = sum_terms coef * { U[c][q][k] | X[c][q] | X[c][q1] X[c][q2] }       row = (out irrep, block, m)
//   out[o][q0_l + m] = sum_{block, c} W_l[o][block*C + c] cat[row(l, block, m)][c]
#include <stdlib.h>

#include "ops.hpp"

namespace lgn {

namespace {
LGN_STAMP_DECL
constexpr int NODES_PER_WG = 8;
constexpr int MAXW = 16;      // CatMix weights accumulated per thread in the backward (n_w <= MAXW * BLOCK)

// node data -> LDS: Ul [C][Q][5][2] (as in global), Xl [C*Q][2] (re, im interleaved); coalesced reads, then the table
// walks below hit LDS instead of scattered global addresses
__device__ __forceinline__ void stage_node(const LocalArgs& a, int node, double* Ul, double* Xl) {
  const int CQ = a.C * a.Q;
  const size_t plane = (size_t)a.nodes * CQ;
  const double* u = a.U + (size_t)node * CQ * 10;
  for (int e = threadIdx.x; e < CQ * 10; e += BLOCK) Ul[e] = u[e];
  for (int e = threadIdx.x; e < CQ; e += BLOCK) {
    Xl[2 * e] = a.X[(size_t)node * CQ + e];
    Xl[2 * e + 1] = a.X[plane + (size_t)node * CQ + e];
  }
}

// The next node's data is fetched into registers while the current node is processed (the loads' latency would otherwise
// sit at the head of every node's chain of barrier-separated phases) and moved to LDS once the current node is done with it.
constexpr int PF_U = 7;               // C * Q * 10 <= PF_U * BLOCK, C * Q <= BLOCK, CO * Qout <= BLOCK (checked at launch)
struct NodePrefetch {
  double u[PF_U], x0, x1, g0, g1;
};
__device__ __forceinline__ void prefetch_node(const LocalArgs& a, int node, bool with_g, NodePrefetch& P) {
  const int CQ = a.C * a.Q, COQ = a.CO * a.Qout, tid = threadIdx.x;
  const double* u = a.U + (size_t)node * CQ * 10;
#pragma unroll
  for (int i = 0; i < PF_U; ++i) P.u[i] = tid + i * BLOCK < CQ * 10 ? u[tid + i * BLOCK] : 0.0;
  P.x0 = tid < CQ ? a.X[(size_t)node * CQ + tid] : 0.0;
  P.x1 = tid < CQ ? a.X[(size_t)a.nodes * CQ + (size_t)node * CQ + tid] : 0.0;
  if (with_g) {
    P.g0 = tid < COQ ? a.g_out[(size_t)node * COQ + tid] : 0.0;
    P.g1 = tid < COQ ? a.g_out[(size_t)a.nodes * COQ + (size_t)node * COQ + tid] : 0.0;
  }
}
__device__ __forceinline__ void commit_node(const LocalArgs& a, const NodePrefetch& P, double* Ul, double* Xl, double* go) {
  const int CQ = a.C * a.Q, COQ = a.CO * a.Qout, tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < PF_U; ++i)
    if (tid + i * BLOCK < CQ * 10) Ul[tid + i * BLOCK] = P.u[i];
  if (tid < CQ) { Xl[2 * tid] = P.x0;  Xl[2 * tid + 1] = P.x1; }
  if (go && tid < COQ) { go[2 * tid] = P.g0;  go[2 * tid + 1] = P.g1; }
}

// The walk tables (row bounds, packed term codes, CG coefficients) are staged in LDS once per workgroup: every cat element
// otherwise starts with a chain of three dependent global loads (row_ptr -> term -> operand).
struct WalkTables {
  const int* row_ptr;            // [n_rows + 1]
  const int* code;               // [n_terms]: type | a << 2 | b << 12
  const double* coef;            // [n_terms]
};
__device__ __forceinline__ WalkTables stage_tables(const LocalArgs& a, double* coef, int* ints) {
  int* row_ptr = ints;
  int* code = ints + a.t.n_rows + 1;
  for (int e = threadIdx.x; e < a.n_terms; e += BLOCK) {
    coef[e] = a.t.t_coef[e];
    code[e] = (a.t.t_type[e] & 3) | (a.t.t_a[e] << 2) | (a.t.t_b[e] << 12);     // (bit 2 of t_type = end-of-row flag of the v2 walk)
  }
  for (int e = threadIdx.x; e <= a.t.n_rows; e += BLOCK) row_ptr[e] = a.t.row_ptr[e];
  return WalkTables{row_ptr, code, coef};
}
__host__ __device__ inline size_t walk_table_bytes(int n_rows, int n_terms) {
  return sizeof(double) * (size_t)n_terms + sizeof(int) * (((size_t)n_rows + 1 + n_terms + 1) & ~size_t(1));
}

// cat[row][c] = sum_terms coef * value.  A row's terms are fetched four at a time (clamped index, zero coefficient past
// the end, loads of all three operand kinds issued unconditionally) so that the round trips of a walk overlap; the terms
// are still added in list order.
__device__ __forceinline__ void build_cat(const WalkTables& t, int n_rows, int C, int Q, const double* Ul, const double* Xl, double* cat) {
  for (int e = threadIdx.x; e < n_rows * C; e += BLOCK) {
    const int row = e / C, c = e - row * C;
    const int beg = t.row_ptr[row], end = t.row_ptr[row + 1];
    const double* ub = Ul + (c * Q) * 10;
    const double* xb = Xl + 2 * (c * Q);
    cx<double> acc = {0, 0};
    for (int k = beg; k < end; k += 4) {
      int ty[4], ia[4], ib[4];
      double cf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = min(k + j, end - 1);
        const int code = t.code[kk];
        ty[j] = code & 3;
        ia[j] = (code >> 2) & 1023;
        ib[j] = code >> 12;
        cf[j] = k + j < end ? t.coef[kk] : 0.0;
      }
      cx<double> u[4], x[4], y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double* up = ub + 2 * (ty[j] == 0 ? ia[j] : 0);       // type 0: U[a = q*5 + k']
        const double* xp = xb + 2 * (ty[j] == 0 ? 0 : ia[j]);       // type 1, 2: X[a]
        const double* yp = xb + 2 * (ty[j] == 2 ? ib[j] : 0);       // type 2: X[a] * X[b]
        u[j] = {up[0], up[1]};
        x[j] = {xp[0], xp[1]};
        y[j] = {yp[0], yp[1]};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const cx<double> xy = cmul(x[j], y[j]);
        cx<double> v = ty[j] == 0 ? u[j] : x[j];
        if (ty[j] == 2) v = xy;
        acc.r += cf[j] * v.r;
        acc.i += cf[j] * v.i;
      }
    }
    cat[2 * e] = acc.r;
    cat[2 * e + 1] = acc.i;
  }
}

__device__ __forceinline__ int irrep_of(const LocalTables& t, int q) {
  int l = 0;
  while (l + 1 < t.n_out && q >= t.out_q0[l + 1]) ++l;
  return l;
}
}  // namespace
LGN_STAMP_READER(lgn_debug_stamps_local)

__global__ __launch_bounds__(BLOCK) void local_fwd_kernel(LocalArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* cat = reinterpret_cast<double*>(smem_raw);          // n_rows * C * 2
  const int C = a.C, CO = a.CO, Qo = a.Qout;
  double* Ul = cat + (size_t)a.t.n_rows * C * 2;              // C * Q * 10
  double* Xl = Ul + (size_t)C * a.Q * 10;                     // C * Q * 2
  double* tcoef = Xl + (size_t)C * a.Q * 2;                   // walk tables
  const WalkTables T = stage_tables(a, tcoef, reinterpret_cast<int*>(tcoef + a.n_terms));
  const size_t plo = (size_t)a.nodes * CO * Qo;
  NodePrefetch P;
  if ((int)blockIdx.x * NODES_PER_WG < a.nodes) stage_node(a, blockIdx.x * NODES_PER_WG, Ul, Xl);
  __syncthreads();
  for (int nl = 0; nl < NODES_PER_WG; ++nl) {
    const int node = blockIdx.x * NODES_PER_WG + nl;
    if (node >= a.nodes) break;
    const bool more = nl + 1 < NODES_PER_WG && node + 1 < a.nodes;
    if (more) prefetch_node(a, node + 1, false, P);
    build_cat(T, a.t.n_rows, C, a.Q, Ul, Xl, cat);
    __syncthreads();
    if (more) commit_node(a, P, Ul, Xl, nullptr);          // Ul / Xl are free once the cat rows exist
    for (int e = threadIdx.x; e < CO * Qo; e += BLOCK) {
      const int o = e / Qo, q = e - o * Qo;
      const int l = irrep_of(a.t, q), m = q - a.t.out_q0[l], d = a.t.out_dim[l], nb = a.t.out_nblk[l];
      const int K = nb * C;
      const double* wr = a.wcat + a.t.out_w0[l] + (size_t)o * K;
      const double* wi = wr + (size_t)CO * K;
      cx<double> acc0 = {0, 0}, acc1 = {0, 0};               // two chains: the sum over (block, channel) is latency bound
      const int row0 = a.t.out_row0[l];
      for (int blk = 0; blk < nb; ++blk) {
        const double* cr = cat + (size_t)(row0 + blk * d + m) * C * 2;
        int c = 0;
        for (; c + 1 < C; c += 2) {
          cfma(acc0, cx<double>{wr[blk * C + c], wi[blk * C + c]}, cx<double>{cr[2 * c], cr[2 * c + 1]});
          cfma(acc1, cx<double>{wr[blk * C + c + 1], wi[blk * C + c + 1]}, cx<double>{cr[2 * c + 2], cr[2 * c + 3]});
        }
        if (c < C) cfma(acc0, cx<double>{wr[blk * C + c], wi[blk * C + c]}, cx<double>{cr[2 * c], cr[2 * c + 1]});
      }
      const size_t oe = ((size_t)node * CO + o) * Qo + q;
      a.out[oe] = acc0.r + acc1.r;
      a.out[plo + oe] = acc0.i + acc1.i;
      if (a.s_copy && q == a.q_s) {                        // pre-MLP scalars kept for the CGMLP backward
        a.s_copy[(size_t)node * CO + o] = acc0.r + acc1.r;
        a.s_copy[(size_t)a.nodes * CO + (size_t)node * CO + o] = acc0.i + acc1.i;
      }
    }
    __syncthreads();                                       // cat is rewritten by the next node; its Ul / Xl are in place
  }
}

__global__ __launch_bounds__(BLOCK) void local_bwd_kernel(LocalArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int C = a.C, CO = a.CO, Q = a.Q, Qo = a.Qout;
  double* cat = reinterpret_cast<double*>(smem_raw);          // n_rows * C * 2
  double* gcat = cat;                                         // the row gradients overwrite the rows once the weight gradient has used them
  double* go = cat + (size_t)a.t.n_rows * C * 2;              // CO * Qo * 2
  double* Ul = go + (size_t)CO * Qo * 2;                      // C * Q * 10
  double* Xl = Ul + (size_t)C * Q * 10;                       // C * Q * 2
  int* winfo = reinterpret_cast<int*>(Xl + (size_t)C * Q * 2);  // n_w packed (first cat row, q0, d, o, c) of every CatMix weight
  for (int w = threadIdx.x; w < a.t.n_w; w += BLOCK) {        // decoded once per workgroup, not once per node
    int l = 0, base = 0;
    while (l + 1 < a.t.n_out && w >= base + CO * a.t.out_nblk[l] * C) { base += CO * a.t.out_nblk[l] * C; ++l; }
    const int K = a.t.out_nblk[l] * C, rel = w - base, o = rel / K, kc = rel - o * K, blk = kc / C, c = kc - blk * C;
    const int d = a.t.out_dim[l];
    winfo[w] = (a.t.out_row0[l] + blk * d) | (a.t.out_q0[l] << 10) | (d << 16) | (o << 20) | (c << 24);
  }
  int* rinfo = winfo + a.t.n_w;                              // per cat row: q | K << 8, and the offset of W_l[0][blk * C]
  int* rwoff = rinfo + a.t.n_rows;
  for (int row = threadIdx.x; row < a.t.n_rows; row += BLOCK) {
    int l = 0;
    while (l + 1 < a.t.n_out && row >= a.t.out_row0[l + 1]) ++l;
    const int d = a.t.out_dim[l], rel = row - a.t.out_row0[l], blk = rel / d, m = rel - blk * d;
    rinfo[row] = (a.t.out_q0[l] + m) | ((a.t.out_nblk[l] * C) << 8);
    rwoff[row] = a.t.out_w0[l] + blk * C;
  }
  // the transposed U lists (which cat rows a moment feeds) are walked by 5 C Q items per node: LDS copies
  int* uptr = rwoff + a.t.n_rows;                            // 5 Q + 1
  int* urow = uptr + (5 * Q + 1);                            // n_u
  double* ucoef = reinterpret_cast<double*>(urow + a.n_u + ((5 * Q + 1 + a.n_u + a.t.n_w + 2 * a.t.n_rows) & 1));   // 8-byte aligned
  double* tcoef = ucoef + a.n_u;                              // walk tables of build_cat
  const WalkTables T = stage_tables(a, tcoef, reinterpret_cast<int*>(tcoef + a.n_terms));
  for (int e = threadIdx.x; e < 5 * Q + 1; e += BLOCK) uptr[e] = a.t.u_ptr[e];
  for (int e = threadIdx.x; e < a.n_u; e += BLOCK) { urow[e] = a.t.u_row[e];  ucoef[e] = a.t.u_coef[e]; }
  const size_t plx = (size_t)a.nodes * C * Q;
  cx<double> dw[MAXW];
#pragma unroll
  for (int k = 0; k < MAXW; ++k) dw[k] = {0, 0};
  NodePrefetch P;

  for (int nl = 0; nl < NODES_PER_WG; ++nl) {
    const int node = blockIdx.x * NODES_PER_WG + nl;
    if (node >= a.nodes) break;
    if (nl == 0) STAMP(0);
    const bool more = nl + 1 < NODES_PER_WG && node + 1 < a.nodes;
    if (nl == 0) {
      prefetch_node(a, node, true, P);
      commit_node(a, P, Ul, Xl, go);
      __syncthreads();
    }
    if (more) prefetch_node(a, node + 1, true, P);
    if (nl == 0) STAMP(1);
    build_cat(T, a.t.n_rows, C, Q, Ul, Xl, cat);
    __syncthreads();
    if (nl == 0) STAMP(2);
    // CatMix weight gradient, accumulated over this workgroup's nodes:  dW_l[o][k] += sum_m g_out[o][q0+m] conj(cat[row][c])
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
      const int w = threadIdx.x + k * BLOCK;
      if (w < a.t.n_w) {
        const int info = winfo[w];
        const int row = info & 1023, q0 = (info >> 10) & 63, d = (info >> 16) & 15, o = (info >> 20) & 15, c = info >> 24;
        for (int m = 0; m < d; ++m)
          cfmac(dw[k], cx<double>{go[2 * (o * Qo + q0 + m)], go[2 * (o * Qo + q0 + m) + 1]},
                cx<double>{cat[2 * ((row + m) * C + c)], cat[2 * ((row + m) * C + c) + 1]});
      }
    }
    if (nl == 0) STAMP(3);
    __syncthreads();                                       // every read of cat is done: its buffer now takes the row gradients
    // gradient of the concatenated rows
    for (int e = threadIdx.x; e < a.t.n_rows * C; e += BLOCK) {
      const int row = e / C, c = e - row * C;
      const int q = rinfo[row] & 255, K = rinfo[row] >> 8;
      cx<double> acc = {0, 0}, acc1 = {0, 0};
      const double* wbase = a.wcat + rwoff[row] + c;
      int o = 0;
      for (; o + 1 < CO; o += 2) {                       // two chains: the weights come from global memory (L1)
        const double* wr = wbase + (size_t)o * K;
        const double* wi = wr + (size_t)CO * K;
        cfmac(acc, cx<double>{go[2 * (o * Qo + q)], go[2 * (o * Qo + q) + 1]}, cx<double>{wr[0], wi[0]});
        cfmac(acc1, cx<double>{go[2 * ((o + 1) * Qo + q)], go[2 * ((o + 1) * Qo + q) + 1]}, cx<double>{wr[K], wi[K]});
      }
      if (o < CO) {
        const double* wr = wbase + (size_t)o * K;
        const double* wi = wr + (size_t)CO * K;
        cfmac(acc, cx<double>{go[2 * (o * Qo + q)], go[2 * (o * Qo + q) + 1]}, cx<double>{wr[0], wi[0]});
      }
      acc.r += acc1.r;
      acc.i += acc1.i;
      gcat[2 * e] = acc.r;
      gcat[2 * e + 1] = acc.i;
    }
    if (nl == 0) STAMP(4);
    __syncthreads();
    if (nl == 0) STAMP(5);
    // gradient of the moments
    for (int e = threadIdx.x; e < C * Q * 5; e += BLOCK) {
      const int c = e / (Q * 5), uq = e - c * Q * 5;
      cx<double> acc = {0, 0};
      const int ub0 = uptr[uq], ue0 = uptr[uq + 1];
      for (int t = ub0; t < ue0; t += 4) {               // four terms in flight, added in list order
        double cf[4];
        cx<double> gv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int tt = min(t + j, ue0 - 1);
          cf[j] = t + j < ue0 ? ucoef[tt] : 0.0;
          const double* gp = gcat + 2 * (urow[tt] * C + c);
          gv[j] = {gp[0], gp[1]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc.r += cf[j] * gv[j].r;
          acc.i += cf[j] * gv[j].i;
        }
      }
      double* gu = a.gU + (((size_t)node * C + c) * Q) * 10 + 2 * uq;
      gu[0] = acc.r;
      gu[1] = acc.i;
    }
    if (nl == 0) STAMP(6);
    // gradient of the node features (direct block + power terms); the N^2 backward adds the aggregate part later
    // a component's term list is long (every product it enters): 8 lanes share one (channel, component), each walks
    // every 8th group of four terms, and the partial sums meet by lane shuffles
    for (int e8 = threadIdx.x; e8 < ((C * Q * 8 + 63) & ~63); e8 += BLOCK) {
      const int e = e8 >> 3, sub = e8 & 7;
      const bool live = e < C * Q;
      const int c = live ? e / Q : 0, q = live ? e - c * Q : 0;
      cx<double> acc = {0, 0};
      const int xb0 = a.t.x_ptr[q], xe0 = live ? a.t.x_ptr[q + 1] : xb0;
      for (int t = xb0 + 4 * sub; t < xe0; t += 32) {     // four terms in flight
        double cf[4];
        int oth[4];
        cx<double> gv[4], xo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int tt = min(t + j, xe0 - 1);
          cf[j] = t + j < xe0 ? a.t.x_coef[tt] : 0.0;
          oth[j] = a.t.x_other[tt];
          const double* gp = gcat + 2 * (a.t.x_row[tt] * C + c);
          gv[j] = {gp[0], gp[1]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double* xp = Xl + 2 * (c * Q + (oth[j] >= 0 ? oth[j] : 0));
          xo[j] = {xp[0], xp[1]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          cx<double> g = {cf[j] * gv[j].r, cf[j] * gv[j].i};
          const cx<double> gx = cmulc(g, xo[j]);
          if (oth[j] >= 0) g = gx;
          acc.r += g.r;
          acc.i += g.i;
        }
      }
#pragma unroll
      for (int m = 1; m < 8; m <<= 1) {
        acc.r += shfl_xor(acc.r, m);
        acc.i += shfl_xor(acc.i, m);
      }
      if (live && sub == 0) {
        const size_t xe = ((size_t)node * C + c) * Q + q;
        a.gX[xe] = acc.r;
        a.gX[plx + xe] = acc.i;
      }
    }
    if (nl == 0) STAMP(7);
    __syncthreads();                                       // every read of Ul / Xl / go / cat / gcat of this node is done
    if (more) commit_node(a, P, Ul, Xl, go);
    __syncthreads();
    if (nl == 0) STAMP(8);
  }
  STAMP(9);
  // partial row of this workgroup, layout like wcat: per irrep [2][CO][K]
  double* part = a.part + (size_t)blockIdx.x * 2 * a.t.n_w;
#pragma unroll
  for (int k = 0; k < MAXW; ++k) {
    const int w = threadIdx.x + k * BLOCK;
    if (w < a.t.n_w) {
      int l = 0, base = 0;
      while (l + 1 < a.t.n_out && w >= base + CO * a.t.out_nblk[l] * C) { base += CO * a.t.out_nblk[l] * C; ++l; }
      const int sz = CO * a.t.out_nblk[l] * C, rel = w - base;
      part[a.t.out_w0[l] + rel] = dw[k].r;       // plane 0 of irrep l (same offsets as the weights in wcat)
      part[a.t.out_w0[l] + sz + rel] = dw[k].i;  // plane 1
    }
  }
}

int local_partial_rows(int nodes) { return cdiv(nodes, NODES_PER_WG); }

// ---- packed <-> separate layouts at the ends of a table-driven network ---------------------------------------------
// X [2][nodes][C][Q]  <->  s [2][nodes][C] (component q_s), v [2][nodes][C][4] (components q_v .. q_v+3)
__global__ __launch_bounds__(BLOCK) void gen_pack_kernel(size_t total, int Q, int q_s, int q_v, const double* __restrict__ s,
                                                        const double* __restrict__ v, double* __restrict__ X) {
  for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < 2 * total * Q; e += (size_t)gridDim.x * BLOCK) {
    const size_t nc = e / Q;                 // (plane, node, channel)
    const int q = (int)(e - nc * Q);
    double val = 0.0;                        // components other than (0,0) / (1,1) are zero-filled
    if (q == q_s) val = s[nc];
    else if (q >= q_v && q < q_v + 4) val = v[nc * 4 + (q - q_v)];
    X[e] = val;
  }
}
__global__ __launch_bounds__(BLOCK) void gen_unpack_kernel(size_t total, int Q, int q_s, int q_v, const double* __restrict__ X,
                                                          double* __restrict__ s, double* __restrict__ v) {
  for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < 2 * total * 5; e += (size_t)gridDim.x * BLOCK) {
    const size_t nc = e / 5;
    const int k = (int)(e - nc * 5);
    if (k == 4) s[nc] = X[nc * Q + q_s];
    else v[nc * 4 + k] = X[nc * Q + q_v + k];
  }
}
// tile-blocked variants: XT [tile][C][Q][2][64]
__global__ __launch_bounds__(BLOCK) void gen_pack_tb_kernel(int M, int C, int Q, int q_s, int q_v, const double* __restrict__ s,
                                                           const double* __restrict__ v, double* __restrict__ XT) {
  const size_t tiles = (M + 63) / 64, total = tiles * C * Q * 128, pl = (size_t)M * C;
  for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * BLOCK) {
    const int lane = e & 63, z = (e >> 6) & 1;
    const size_t r = e >> 7;                  // (tile, c, q)
    const int q = (int)(r % Q), c = (int)((r / Q) % C);
    const size_t n = (r / ((size_t)Q * C)) * 64 + lane;
    double val = 0.0;
    if (n < (size_t)M) {
      const size_t nc = z * pl + n * C + c;
      if (q == q_s) val = s[nc];
      else if (q >= q_v && q < q_v + 4) val = v[nc * 4 + (q - q_v)];
    }
    XT[e] = val;
  }
}
__global__ __launch_bounds__(BLOCK) void gen_unpack_tb_kernel(int M, int C, int Q, int q_s, int q_v, const double* __restrict__ XT,
                                                             double* __restrict__ s, double* __restrict__ v) {
  const size_t total = (size_t)2 * M * C * 5, pl = (size_t)M * C;
  for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * BLOCK) {
    const size_t nc = e / 5;                  // (plane, node, channel)
    const int k = (int)(e - nc * 5), z = (int)(nc / pl);
    const size_t r = nc - z * pl;
    const int n = (int)(r / C), c = (int)(r - (size_t)n * C);
    const int q = k == 4 ? q_s : q_v + k;
    const double val = XT[((size_t)((n >> 6) * C + c) * Q + q) * 128 + z * 64 + (n & 63)];
    if (k == 4) s[nc] = val;
    else v[nc * 4 + k] = val;
  }
}

static int glue_grid(size_t n) {
  size_t g = (n + BLOCK - 1) / BLOCK;
  return (int)(g < 4096 ? (g ? g : 1) : 4096);
}
int gen_pack(size_t nodes_x_C, int Q, int q_s, int q_v, const double* s, const double* v, double* X, hipStream_t st) {
  LGN_CHECK_ARG(Q >= 5 && q_s >= 0 && q_s < Q && q_v >= 0 && q_v + 4 <= Q && (q_s < q_v || q_s >= q_v + 4), "gen_pack: bad component offsets");
  hipLaunchKernelGGL(gen_pack_kernel, dim3(glue_grid(2 * nodes_x_C * Q)), dim3(BLOCK), 0, st, nodes_x_C, Q, q_s, q_v, s, v, X);
  LGN_CHECK_LAUNCH();
  return 0;
}
int gen_pack_tb(int M, int C, int Q, int q_s, int q_v, const double* s, const double* v, double* XT, hipStream_t st) {
  LGN_CHECK_ARG(Q >= 5 && q_s >= 0 && q_s < Q && q_v >= 0 && q_v + 4 <= Q && (q_s < q_v || q_s >= q_v + 4), "gen_pack_tb: bad component offsets");
  hipLaunchKernelGGL(gen_pack_tb_kernel, dim3(glue_grid((size_t)((M + 63) / 64) * C * Q * 128)), dim3(BLOCK), 0, st, M, C, Q, q_s, q_v, s, v, XT);
  LGN_CHECK_LAUNCH();
  return 0;
}
int gen_unpack_tb(int M, int C, int Q, int q_s, int q_v, const double* XT, double* s, double* v, hipStream_t st) {
  LGN_CHECK_ARG(Q >= 5 && q_s >= 0 && q_s < Q && q_v >= 0 && q_v + 4 <= Q && (q_s < q_v || q_s >= q_v + 4), "gen_unpack_tb: bad component offsets");
  hipLaunchKernelGGL(gen_unpack_tb_kernel, dim3(glue_grid((size_t)2 * M * C * 5)), dim3(BLOCK), 0, st, M, C, Q, q_s, q_v, XT, s, v);
  LGN_CHECK_LAUNCH();
  return 0;
}
int gen_unpack(size_t nodes_x_C, int Q, int q_s, int q_v, const double* X, double* s, double* v, hipStream_t st) {
  LGN_CHECK_ARG(Q >= 5 && q_s >= 0 && q_s < Q && q_v >= 0 && q_v + 4 <= Q && (q_s < q_v || q_s >= q_v + 4), "gen_unpack: bad component offsets");
  hipLaunchKernelGGL(gen_unpack_kernel, dim3(glue_grid(2 * nodes_x_C * 5)), dim3(BLOCK), 0, st, nodes_x_C, Q, q_s, q_v, X, s, v);
  LGN_CHECK_LAUNCH();
  return 0;
}

int local_fwd(const LocalArgs& a, hipStream_t st) {
  LGN_CHECK_ARG(a.nodes > 0 && a.C >= 1 && a.CO >= 1, "local_fwd: empty input");
  LGN_CHECK_ARG(a.C * a.Q <= BLOCK && a.C * a.Q * 10 <= PF_U * BLOCK && a.CO * a.Qout <= BLOCK, "local_fwd: C*Q=%d too large", a.C * a.Q);
  LGN_CHECK_ARG(a.Q * 5 < 1024 && a.n_terms > 0, "local_fwd: Q=%d exceeds the packed term code", a.Q);
  const size_t smem = sizeof(double) * ((size_t)a.t.n_rows * a.C * 2 + (size_t)a.C * a.Q * 12) + walk_table_bytes(a.t.n_rows, a.n_terms);
  LGN_CHECK_ARG(smem <= 160 * 1024, "local_fwd: %zu B of LDS needed", smem);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(local_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(local_fwd_kernel, dim3(local_partial_rows(a.nodes)), dim3(BLOCK), smem, st, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

int local_bwd(const LocalArgs& a, hipStream_t st) {
  LGN_CHECK_ARG(a.nodes > 0 && a.C >= 1 && a.CO >= 1, "local_bwd: empty input");
  LGN_CHECK_ARG(a.C * a.Q <= BLOCK && a.C * a.Q * 10 <= PF_U * BLOCK && a.CO * a.Qout <= BLOCK, "local_bwd: C*Q=%d too large", a.C * a.Q);
  LGN_CHECK_ARG(a.t.n_w <= MAXW * BLOCK, "local_bwd: %d CatMix weights exceed the per-workgroup accumulator budget", a.t.n_w);
  // widths of the bit-packed decode tables of the kernel (winfo: row < 1024, q0 < 64, d < 16, o < 16, c < 128; rinfo: q < 256)
  LGN_CHECK_ARG(a.t.n_rows < 1024 && a.Qout < 64 && a.CO <= 15 && a.C <= 127 && a.Q < 256,
                "local_bwd: n_rows=%d Qout=%d CO=%d exceed the packed table fields", a.t.n_rows, a.Qout, a.CO);
  const size_t smem = sizeof(double) * ((size_t)a.t.n_rows * a.C * 2 + (size_t)a.CO * a.Qout * 2 + (size_t)a.C * a.Q * 12) +
                      sizeof(int) * ((size_t)a.t.n_w + 2 * (size_t)a.t.n_rows + 5 * (size_t)a.Q + 2 + (size_t)a.n_u) +
                      sizeof(double) * (size_t)a.n_u + 8 + walk_table_bytes(a.t.n_rows, a.n_terms);
  LGN_CHECK_ARG(a.Q * 5 < 1024 && a.n_terms > 0, "local_bwd: Q=%d exceeds the packed term code", a.Q);
  LGN_CHECK_ARG(smem <= 160 * 1024, "local_bwd: %zu B of LDS needed", smem);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(local_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(local_bwd_kernel, dim3(local_partial_rows(a.nodes)), dim3(BLOCK), smem, st, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace lgn
