// lgn-autoencoder_amd/csrc/mlp.hip -- CGMLP (scalar-irrep MLP) forward / backward.
//
// Reference: CGMLP.forward, lgn/models/lgn_levels.py:191-227 -- the (0,0) part (2,B,N,C,1) is viewed as
// (B*N) rows of 2C real features (index 2c+z), passed through Linear(2C->W), NH-1 x Linear(W->W),
// Linear(W->2C) with LeakyReLU(0.01) after all but the last Linear, and written back.  No mask.
//
// Mapping: a workgroup owns 64 rows (lane = row); its 4 waves own 4 disjoint slices of the layer's
// output neurons (OPT per thread).  Activations ping-pong through LDS transposed ([neuron][row], so
// row-contiguous = conflict-free); weights are wave-uniform reads.  The backward recomputes the
// forward with the hidden activations kept in registers, then walks the layers in reverse:
//   g_in  = g_pre W          (same mapping as the forward)
//   dW    = g_pre^T h_in     (register-tiled (o,k) outer products over the 64 rows, LDS operands)
// Per-workgroup weight-gradient partials are reduced afterwards (deterministic, no atomics).
#include "ops.hpp"

namespace lgn {

constexpr int ROWS = 64;
constexpr int RPAD = 65;          // padded row stride for the (o,k) outer-product reads

__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

template <typename T>
__device__ __forceinline__ void load_rows(const T* s, int M, int C, int row0, T* buf /*[2C][ROWS]*/) {
  // feature k = 2c + z  <-  s[z][row][c]
  const int D = 2 * C;
  for (int e = threadIdx.x; e < D * ROWS; e += BLOCK) {
    int k = e / ROWS, r = e - k * ROWS;
    int row = row0 + r;
    buf[k * ROWS + r] = row < M ? s[(size_t)(k & 1) * M * C + (size_t)row * C + (k >> 1)] : T(0);
  }
}

// one dense layer for this thread's OPT outputs:  acc[t] = bias[o] + sum_k W[o][k] * in[k][lane]
template <typename T, int OPT>
__device__ __forceinline__ void dense(const T* __restrict__ W, const T* __restrict__ bias, int Hin, int Hout, int og,
                                      const T* in, int lane, T (&acc)[OPT]) {
  const int o0 = og * OPT;
#pragma unroll
  for (int t = 0; t < OPT; ++t) acc[t] = (o0 + t < Hout) ? bias[o0 + t] : T(0);
  if (o0 >= Hout) return;
  for (int k = 0; k < Hin; ++k) {
    const T x = in[k * ROWS + lane];
#pragma unroll
    for (int t = 0; t < OPT; ++t)
      if (o0 + t < Hout) acc[t] += W[(size_t)(o0 + t) * Hin + k] * x;
  }
}

template <typename T, int OPT, int NH>
__global__ __launch_bounds__(BLOCK) void mlp_fwd_kernel(MlpArgs<T> a) {
  const int tid = threadIdx.x, lane = tid & 63, og = uniform(tid >> 6);
  const int D = 2 * a.C, H = a.H, M = a.M;
  const int row0 = blockIdx.x * ROWS;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* cur = reinterpret_cast<T*>(smem_raw);
  T* nxt = cur + max(H, D) * ROWS;

  load_rows(a.s_in, M, a.C, row0, cur);
  __syncthreads();
#pragma unroll
  for (int l = 0; l <= NH; ++l) {
    const int Hin = l == 0 ? D : H, Hout = l == NH ? D : H;
    T acc[OPT];
    dense<T, OPT>(a.w[l], a.b[l], Hin, Hout, og, cur, lane, acc);
#pragma unroll
    for (int t = 0; t < OPT; ++t) {
      const int o = og * OPT + t;
      if (o < Hout) nxt[o * ROWS + lane] = l < NH ? leaky(acc[t]) : acc[t];
    }
    __syncthreads();
    T* tmp = cur; cur = nxt; nxt = tmp;
  }
  for (int e = tid; e < D * ROWS; e += BLOCK) {
    int k = e / ROWS, r = e - k * ROWS, row = row0 + r;
    if (row < M) a.s_out[(size_t)(k & 1) * M * a.C + (size_t)row * a.C + (k >> 1)] = cur[k * ROWS + r];
  }
}

template <typename T, int OPT, int NH>
__global__ __launch_bounds__(BLOCK) void mlp_bwd_kernel(MlpArgs<T> a) {
  constexpr int TS = (OPT + 3) / 4;            // (o,k) register tile edge for the weight gradient: 16*TS >= H
  const int tid = threadIdx.x, lane = tid & 63, og = uniform(tid >> 6);
  const int D = 2 * a.C, H = a.H, M = a.M;
  const int HM = max(H, D);
  const int row0 = blockIdx.x * ROWS;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* cur = reinterpret_cast<T*>(smem_raw);      // forward ping / backward "g_pre" operand  [HM][RPAD]
  T* nxt = cur + HM * RPAD;                     // forward pong / backward "h_in"  operand  [HM][RPAD]
  T* xin = nxt + HM * RPAD;                     // the MLP input rows [D][ROWS]
  T* part = a.part + (size_t)blockIdx.x * a.psize;

  // ---- recompute the forward, hidden activations stay in registers ----------------------------
  load_rows(a.s_in, M, a.C, row0, xin);
  __syncthreads();
  T h[NH][OPT];
  {
    const T* in = xin;
    T* out = cur;
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      const int Hin = l == 0 ? D : H;
      T acc[OPT];
      dense<T, OPT>(a.w[l], a.b[l], Hin, H, og, in, lane, acc);
#pragma unroll
      for (int t = 0; t < OPT; ++t) {
        h[l][t] = leaky(acc[t]);
        const int o = og * OPT + t;
        if (o < H) out[o * ROWS + lane] = h[l][t];
      }
      __syncthreads();
      in = out;
      out = (out == cur) ? nxt : cur;
    }
  }

  // ---- backward sweep ---------------------------------------------------------------------------
  // gpre[t]: gradient w.r.t. the pre-activation of this thread's outputs of the current layer
  T gpre[OPT];
  {
    // last Linear (no activation): outputs o < D owned by og*OPT + t
#pragma unroll
    for (int t = 0; t < OPT; ++t) {
      const int o = og * OPT + t, row = row0 + lane;
      gpre[t] = (o < D && row < M) ? a.g_out[(size_t)(o & 1) * M * a.C + (size_t)row * a.C + (o >> 1)] : T(0);
    }
  }
  size_t poff_end = a.psize;
#pragma unroll
  for (int l = NH; l >= 0; --l) {
    const int Hin = l == 0 ? D : H, Hout = l == NH ? D : H;
    poff_end -= (size_t)Hout * Hin + Hout;
    T* pW = part + poff_end;
    T* pB = pW + (size_t)Hout * Hin;
    __syncthreads();                     // previous layer's readers of cur/nxt are done
    // operands to LDS: g_pre[o][row] and the layer input h_in[k][row]
#pragma unroll
    for (int t = 0; t < OPT; ++t) {
      const int o = og * OPT + t;
      if (o < Hout) cur[o * RPAD + lane] = gpre[t];
      if (l > 0) {
        if (o < H) nxt[o * RPAD + lane] = h[l > 0 ? l - 1 : 0][t];
      }
    }
    if (l == 0) {
      for (int e = tid; e < D * ROWS; e += BLOCK) {
        int k = e / ROWS, r = e - k * ROWS;
        nxt[k * RPAD + r] = xin[e];
      }
    }
    __syncthreads();
    // (a) weight / bias gradient partials: thread -> TS x TS tile of (o, k)
    {
      const int nto = (Hout + TS - 1) / TS, ntk = (Hin + TS - 1) / TS;
      for (int tile = tid; tile < nto * ntk; tile += BLOCK) {
        const int ot = tile / ntk, kt = tile - ot * ntk;
        T acc[TS][TS];
#pragma unroll
        for (int x = 0; x < TS; ++x)
#pragma unroll
          for (int y = 0; y < TS; ++y) acc[x][y] = T(0);
        for (int r = 0; r < ROWS; ++r) {
          T gv[TS], hv[TS];
#pragma unroll
          for (int x = 0; x < TS; ++x) {
            gv[x] = (ot * TS + x < Hout) ? cur[(ot * TS + x) * RPAD + r] : T(0);
            hv[x] = (kt * TS + x < Hin) ? nxt[(kt * TS + x) * RPAD + r] : T(0);
          }
#pragma unroll
          for (int x = 0; x < TS; ++x)
#pragma unroll
            for (int y = 0; y < TS; ++y) acc[x][y] += gv[x] * hv[y];
        }
#pragma unroll
        for (int x = 0; x < TS; ++x)
#pragma unroll
          for (int y = 0; y < TS; ++y)
            if (ot * TS + x < Hout && kt * TS + y < Hin) pW[(size_t)(ot * TS + x) * Hin + kt * TS + y] = acc[x][y];
      }
      for (int o = tid; o < Hout; o += BLOCK) {
        T s = T(0);
        for (int r = 0; r < ROWS; ++r) s += cur[o * RPAD + r];
        pB[o] = s;
      }
    }
    // (b) gradient w.r.t. the layer input, for the k's this thread owns
    {
      const int k0 = og * OPT;
      T gin[OPT];
#pragma unroll
      for (int t = 0; t < OPT; ++t) gin[t] = T(0);
      if (k0 < Hin) {
        const T* W = a.w[l];
        for (int o = 0; o < Hout; ++o) {
          const T g = cur[o * RPAD + lane];
#pragma unroll
          for (int t = 0; t < OPT; ++t)
            if (k0 + t < Hin) gin[t] += W[(size_t)o * Hin + k0 + t] * g;
        }
      }
      if (l > 0) {
#pragma unroll
        for (int t = 0; t < OPT; ++t) gpre[t] = gin[t] * (h[l > 0 ? l - 1 : 0][t] > T(0) ? T(1) : T(0.01));
      } else {
#pragma unroll
        for (int t = 0; t < OPT; ++t) {
          const int k = k0 + t, row = row0 + lane;
          if (k < D && row < M) a.g_in[(size_t)(k & 1) * M * a.C + (size_t)row * a.C + (k >> 1)] = gin[t];
        }
      }
    }
  }
}

static size_t mlp_param_count(int C, int H, int nlin) {
  const int D = 2 * C;
  if (nlin == 1) return (size_t)D * D + D;
  return ((size_t)H * D + H) + (size_t)(nlin - 2) * ((size_t)H * H + H) + ((size_t)D * H + D);
}

template <typename T, int OPT>
static int launch_mlp(const MlpArgs<T>& a, bool backward, hipStream_t stream) {
  constexpr int NH = 6;
  const int HM = a.H > 2 * a.C ? a.H : 2 * a.C;
  const int nblk = cdiv(a.M, ROWS);
  if (!backward) {
    size_t smem = sizeof(T) * 2 * HM * ROWS;
    auto kern = mlp_fwd_kernel<T, OPT, NH>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(BLOCK), smem, stream, a);
  } else {
    size_t smem = sizeof(T) * (2 * HM * RPAD + 2 * a.C * ROWS);
    auto kern = mlp_bwd_kernel<T, OPT, NH>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(BLOCK), smem, stream, a);
  }
  LGN_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int mlp_dispatch(const MlpArgs<T>& a, bool backward, hipStream_t stream) {
  LGN_CHECK_ARG(a.M > 0 && a.C > 0, "cgmlp: empty input (M=%d C=%d)", a.M, a.C);
  LGN_CHECK_ARG(a.nlin == 7, "cgmlp: only mlp_depth=6 (7 Linear layers) is built, got %d", a.nlin);
  LGN_CHECK_ARG(a.H >= 2 * a.C && a.H <= 96, "cgmlp: hidden width %d unsupported (2C..96)", a.H);
  if (backward) LGN_CHECK_ARG((size_t)a.psize == mlp_param_count(a.C, a.H, a.nlin), "cgmlp: psize mismatch");
  {  // matrix-core path for H <= 48; the VALU kernels below cover wider MLPs
    int rc = mlp_mfma_dispatch(a, backward, stream);
    if (rc != -2) return rc;
    static const bool valu_only = [] { const char* e = getenv("LGN_AMD_MLP_VALU"); return e && e[0] == '1'; }();
    if (!valu_only) {
      rc = mlp_mfma_wide_dispatch(a, backward, stream);
      if (rc != -2) return rc;
    }
  }
  const int opt = cdiv(a.H, 4);
  if (opt <= 3) return launch_mlp<T, 3>(a, backward, stream);
  if (opt <= 6) return launch_mlp<T, 6>(a, backward, stream);
  if (opt <= 9) return launch_mlp<T, 9>(a, backward, stream);
  if (opt <= 12) return launch_mlp<T, 12>(a, backward, stream);
  if (opt <= 15) return launch_mlp<T, 15>(a, backward, stream);
  if (opt <= 18) return launch_mlp<T, 18>(a, backward, stream);
  return launch_mlp<T, 24>(a, backward, stream);
}

template int mlp_dispatch<double>(const MlpArgs<double>&, bool, hipStream_t);

}  // namespace lgn
