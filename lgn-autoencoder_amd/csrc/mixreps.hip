// lgn-autoencoder_amd/csrc/mixreps.hip -- MixReps: per-irrep complex channel mixing, forward / backward.
//
// Reference: MixReps.forward -> g_torch.mix -> mix_zweight_zvec (lgn/nn/g_nn.py:95-117,
// lgn/g_lib/g_torch.py:217-255, lgn/g_lib/cplx_lib.py:7-25):  y[..,o,m] = sum_i W[o,i] x[..,i,m]
// with complex W (2,C_out,C_in), planar complex x (2,rows,C_in,d), no conjugation.
// Used for input_func_node, the encoder's mix_reps, latent_to_graph and mix_to_output (the CatMix
// of the message-passing levels is fused into level_fwd / level_bwd instead).
#include "ops.hpp"

namespace lgn {

constexpr int MIX_RCH = 128;   // rows per weight-gradient partial

template <typename T>
__global__ void mix_fwd_kernel(MixArgs<T> a) {
  const size_t total = (size_t)a.rows * a.Cout * a.d;
  const size_t px = (size_t)a.rows * a.Cin * a.d, py = total;
  const int wi = a.Cout * a.Cin;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int m = e % a.d;
    const int o = (e / a.d) % a.Cout;
    const size_t row = e / ((size_t)a.d * a.Cout);
    cx<T> acc = {T(0), T(0)};
    for (int i = 0; i < a.Cin; ++i) {
      const size_t xe = (row * a.Cin + i) * a.d + m;
      cfma(acc, cx<T>{a.w[o * a.Cin + i], a.w[wi + o * a.Cin + i]}, cx<T>{a.x[xe], a.x[px + xe]});
    }
    a.y[e] = acc.r;
    a.y[py + e] = acc.i;
  }
}

// g_x[row][i][m] = sum_o g_y[row][o][m] conj(W[o][i])
template <typename T>
__global__ void mix_bwd_x_kernel(MixArgs<T> a) {
  const size_t total = (size_t)a.rows * a.Cin * a.d;
  const size_t px = total, py = (size_t)a.rows * a.Cout * a.d;
  const int wi = a.Cout * a.Cin;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int m = e % a.d;
    const int i = (e / a.d) % a.Cin;
    const size_t row = e / ((size_t)a.d * a.Cin);
    cx<T> acc = {T(0), T(0)};
    for (int o = 0; o < a.Cout; ++o) {
      const size_t ye = (row * a.Cout + o) * a.d + m;
      cfmac(acc, cx<T>{a.g_y[ye], a.g_y[py + ye]}, cx<T>{a.w[o * a.Cin + i], a.w[wi + o * a.Cin + i]});
    }
    a.g_x[e] = acc.r;
    a.g_x[px + e] = acc.i;
  }
}

// partial dW[o][i] over a chunk of rows:  sum_rows sum_m g_y[row][o][m] conj(x[row][i][m])
// Two regimes: many weights (latent_to_graph: C_out = N) -> one thread per weight, serial over the chunk;
// few weights (input / output mixing over B*N rows) -> all threads split the chunk's (row, m) products of one
// weight and combine them with wave shuffles + LDS (fixed order -> reproducible).
template <typename T>
__global__ __launch_bounds__(BLOCK) void mix_bwd_w_kernel(MixArgs<T> a) {
  __shared__ T red[2][BLOCK / 64];
  const int wi = a.Cout * a.Cin;
  const size_t px = (size_t)a.rows * a.Cin * a.d, py = (size_t)a.rows * a.Cout * a.d;
  const int r0 = blockIdx.x * MIX_RCH;
  const int r1 = min(a.rows, r0 + MIX_RCH);
  T* part = a.part + (size_t)blockIdx.x * 2 * wi;
  if (wi > 64) {
    for (int e = threadIdx.x; e < wi; e += blockDim.x) {
      const int o = e / a.Cin, i = e - o * a.Cin;
      cx<T> acc = {T(0), T(0)};
      for (int row = r0; row < r1; ++row)
        for (int m = 0; m < a.d; ++m) {
          const size_t ye = ((size_t)row * a.Cout + o) * a.d + m, xe = ((size_t)row * a.Cin + i) * a.d + m;
          cfmac(acc, cx<T>{a.g_y[ye], a.g_y[py + ye]}, cx<T>{a.x[xe], a.x[px + xe]});
        }
      part[e] = acc.r;
      part[wi + e] = acc.i;
    }
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nel = (r1 - r0) * a.d;
  for (int e = 0; e < wi; ++e) {
    const int o = e / a.Cin, i = e - o * a.Cin;
    cx<T> acc = {T(0), T(0)};
    for (int q = threadIdx.x; q < nel; q += BLOCK) {
      const int row = r0 + q / a.d, m = q % a.d;
      const size_t ye = ((size_t)row * a.Cout + o) * a.d + m, xe = ((size_t)row * a.Cin + i) * a.d + m;
      cfmac(acc, cx<T>{a.g_y[ye], a.g_y[py + ye]}, cx<T>{a.x[xe], a.x[px + xe]});
    }
    acc.r = group_sum<64>(acc.r);
    acc.i = group_sum<64>(acc.i);
    if (lane == 0) { red[0][wave] = acc.r; red[1][wave] = acc.i; }
    __syncthreads();
    if (threadIdx.x == 0) {
      part[e] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      part[wi + e] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
    __syncthreads();
  }
}

template <typename T>
int mix_fwd(const MixArgs<T>& a, hipStream_t stream) {
  LGN_CHECK_ARG(a.rows > 0 && a.Cin > 0 && a.Cout > 0 && (a.d == 1 || a.d == 4 || a.d == 3 || a.d == 9),
                "mixreps: bad shape rows=%d Cin=%d Cout=%d d=%d", a.rows, a.Cin, a.Cout, a.d);
  const size_t total = (size_t)a.rows * a.Cout * a.d;
  const int grid = (int)((total + BLOCK - 1) / BLOCK < 4096 ? (total + BLOCK - 1) / BLOCK : 4096);
  hipLaunchKernelGGL(mix_fwd_kernel<T>, dim3(grid), dim3(BLOCK), 0, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

int mix_partial_rows(int rows) { return cdiv(rows, MIX_RCH); }

template <typename T>
int mix_bwd(const MixArgs<T>& a, hipStream_t stream) {
  LGN_CHECK_ARG(a.rows > 0 && a.Cin > 0 && a.Cout > 0, "mixreps bwd: bad shape");
  if (a.g_x) {
    const size_t total = (size_t)a.rows * a.Cin * a.d;
    const int grid = (int)((total + BLOCK - 1) / BLOCK < 4096 ? (total + BLOCK - 1) / BLOCK : 4096);
    hipLaunchKernelGGL(mix_bwd_x_kernel<T>, dim3(grid), dim3(BLOCK), 0, stream, a);
    LGN_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(mix_bwd_w_kernel<T>, dim3(mix_partial_rows(a.rows)), dim3(BLOCK), 0, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

template int mix_fwd<double>(const MixArgs<double>&, hipStream_t);
template int mix_bwd<double>(const MixArgs<double>&, hipStream_t);

}  // namespace lgn
