// lgn-autoencoder_amd/csrc/level_fwd.hip -- fused message-passing level, forward, maxdim = 2.
//
// Replaces, for one level: RadPolyTrig.forward (lgn/nn/position_levels.py:118-209), the
// GScalar*GVec edge product (lgn/models/lgn_cg.py:167), CGProduct(aggregate) and CGProduct(power)
// (lgn/cg_lib/cg_ops.py:135-298) and CatMixReps (lgn/nn/g_nn.py:260-278).
//
// Grid: (B jets, ceil(N / IT) row tiles).  256 threads = IT rows x JS neighbour slices, IT*JS = 256.
// Thread (il, js) owns output row i = tile*IT + il and visits neighbours j = js, js+JS, ...; the JS
// partial sums of a row live in adjacent lanes and are combined with wave shuffles.  Node features
// of the whole jet, positions, mask and all parameters are staged once in LDS.
#include <stdlib.h>

#include "level_dev.hpp"

namespace lgn {

template <typename T, int C, int JS, bool DEC>
__global__ __launch_bounds__(BLOCK) void level_fwd_kernel(LevelArgs<T> a) {
  using L = Carve<C, DEC>;
  constexpr int IT = BLOCK / JS;
  constexpr int R = L::R;
  const int N = a.N, B = a.B, CO = a.CO;
  const int b = blockIdx.x, tile = blockIdx.y;
  const int tid = threadIdx.x;

  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  T* nd = sm;                                   // N * NS
  T* pj = nd + L::even(N * L::NS);              // N * PS
  T* rp = pj + L::even(N * L::PS);              // RAD_SIZE
  T* wm = rp + L::even(L::RAD_SIZE);            // 2 irreps * 2 planes * CO * 5C
  T* agt = wm + L::even(4 * CO * 5 * C);        // IT * 20C
  uint8_t* mk = reinterpret_cast<uint8_t*>(agt + IT * 20 * C);

  load_jet<T, C, DEC>(a.s_in, a.v_in, a.p, a.mask, B, N, b, nd, pj, mk);
  load_radial<T, C, DEC>(a.ra, a.rb, a.rc, a.w0, a.b0, a.w1, a.b1, rp);
  for (int e = tid; e < 2 * CO * 5 * C; e += BLOCK) {
    wm[e] = a.wm0[e];
    wm[2 * CO * 5 * C + e] = a.wm1[e];
  }
  __syncthreads();

  // ---------------- edge network + aggregation -------------------------------------------
  const int il = tid / JS, js = tid % JS;
  const int i = tile * IT + il;
  const bool row_ok = i < N;
  const int ii = row_ok ? i : 0;

  cx<T> A1[C][4], A2[C][4], A3[C], A4[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    A3[c] = {T(0), T(0)};
    A4[c] = {T(0), T(0)};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      A1[c][m] = {T(0), T(0)};
      A2[c][m] = {T(0), T(0)};
    }
  }
  T pi[L::PS];
#pragma unroll
  for (int m = 0; m < L::PS; ++m) pi[m] = pj[ii * L::PS + m];
  const bool mi = DEC ? false : (mk[ii] != 0);

  if (row_ok) {
    for (int j = js; j < N; j += JS) {
      PairGeom<T, DEC> g = pair_geom<T, DEC>(pi, pj + j * L::PS, mi, DEC ? false : (mk[j] != 0));
      T rad[R];
      radial_eval<T, C, DEC>(rp, g.nrm, g.on, rad);
      const T* nj = nd + j * L::NS;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        cx<T> R0 = {rad[2 * c], rad[2 * c + 1]};
        cx<T> R1 = {rad[2 * C + 2 * c], rad[2 * C + 2 * c + 1]};
        cx<T> e0 = {R0.r - R0.i, R0.r + R0.i};           // R0 * (1 + 1i)
        cx<T> sj = {nj[c * 10], nj[c * 10 + 1]};
        cx<T> vj[4], e1[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          vj[m] = {nj[c * 10 + 2 + m], nj[c * 10 + 6 + m]};
          e1[m] = cmul(R1, g.q[m]);
          cfma(A1[c][m], vj[m], e0);
          cfma(A2[c][m], sj, e1[m]);
        }
        cfma(A4[c], sj, e0);
        cx<T> t = bil2(vj, e1);
        A3[c].r += t.r;
        A3[c].i += t.i;
      }
    }
  }

  // combine the JS neighbour slices of each row (adjacent lanes)
#pragma unroll
  for (int c = 0; c < C; ++c) {
    A3[c].r = group_sum<JS>(A3[c].r) * T(0.5);
    A3[c].i = group_sum<JS>(A3[c].i) * T(0.5);
    A4[c].r = group_sum<JS>(A4[c].r);
    A4[c].i = group_sum<JS>(A4[c].i);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      A1[c][m].r = group_sum<JS>(A1[c][m].r);
      A1[c][m].i = group_sum<JS>(A1[c][m].i);
      A2[c][m].r = group_sum<JS>(A2[c][m].r);
      A2[c][m].i = group_sum<JS>(A2[c][m].i);
    }
  }

  if (row_ok && js == 0) {
    // LDS tile for the CatMix stage: x0[k][z] (k < 2C), then x1[k][m][z]
    T* t0 = agt + il * 20 * C;
    T* t1 = t0 + 4 * C;
    const size_t pl0 = (size_t)B * N * 2 * C;
    T* g0 = a.ag0 + ((size_t)b * N + i) * 2 * C;
    T* g1 = a.ag1 + ((size_t)b * N + i) * 2 * C * 4;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      t0[2 * c] = A3[c].r;  t0[2 * c + 1] = A3[c].i;
      t0[2 * (C + c)] = A4[c].r;  t0[2 * (C + c) + 1] = A4[c].i;
      g0[c] = A3[c].r;  g0[pl0 + c] = A3[c].i;
      g0[C + c] = A4[c].r;  g0[pl0 + C + c] = A4[c].i;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        t1[(c * 4 + m) * 2] = A1[c][m].r;  t1[(c * 4 + m) * 2 + 1] = A1[c][m].i;
        t1[((C + c) * 4 + m) * 2] = A2[c][m].r;  t1[((C + c) * 4 + m) * 2 + 1] = A2[c][m].i;
        g1[c * 4 + m] = A1[c][m].r;  g1[pl0 * 4 + c * 4 + m] = A1[c][m].i;
        g1[(C + c) * 4 + m] = A2[c][m].r;  g1[pl0 * 4 + (C + c) * 4 + m] = A2[c][m].i;
      }
    }
  }
  __syncthreads();

  // ---------------- power + CatMix: thread (row, output channel) ----------------------------
  // Cat order per irrep: [aggregate (2C), node (C), power (2C)]  (lgn_levels.py:120, g_torch.py:204-213)
  // power (0,0) = [<v,v>, s*s] ; power (1,1) = [v*s, s*v]           
  {
    const int o = tid & 7, rl = tid >> 3;
    const int r = tile * IT + rl;
    if (rl < IT && o < CO && r < N) {
      const T* t0 = agt + rl * 20 * C;
      const T* t1 = t0 + 4 * C;
      const T* ni = nd + r * L::NS;
      const int K = 5 * C;
      const T* w0r = wm + (0 * CO + o) * K;
      const T* w0i = wm + (1 * CO + o) * K;
      const T* w1r = wm + 2 * CO * K + (0 * CO + o) * K;
      const T* w1i = wm + 2 * CO * K + (1 * CO + o) * K;
      cx<T> os = {T(0), T(0)}, ov[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) ov[m] = {T(0), T(0)};
#pragma unroll
      for (int k = 0; k < 2 * C; ++k) {           // aggregate block
        cfma(os, cx<T>{w0r[k], w0i[k]}, cx<T>{t0[2 * k], t0[2 * k + 1]});
        cx<T> w = {w1r[k], w1i[k]};
#pragma unroll
        for (int m = 0; m < 4; ++m) cfma(ov[m], w, cx<T>{t1[(k * 4 + m) * 2], t1[(k * 4 + m) * 2 + 1]});
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {               // node block + power blocks
        cx<T> s = {ni[c * 10], ni[c * 10 + 1]};
        cx<T> v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = {ni[c * 10 + 2 + m], ni[c * 10 + 6 + m]};
        cx<T> vv = bil2(v, v);
        vv.r *= T(0.5);
        vv.i *= T(0.5);
        cx<T> ss = cmul(s, s);
        cfma(os, cx<T>{w0r[2 * C + c], w0i[2 * C + c]}, s);
        cfma(os, cx<T>{w0r[3 * C + c], w0i[3 * C + c]}, vv);
        cfma(os, cx<T>{w0r[4 * C + c], w0i[4 * C + c]}, ss);
        cx<T> wn = {w1r[2 * C + c], w1i[2 * C + c]};
        cx<T> wp = {w1r[3 * C + c] + w1r[4 * C + c], w1i[3 * C + c] + w1i[4 * C + c]};   // v*s appears twice
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          cfma(ov[m], wn, v[m]);
          cfma(ov[m], wp, cmul(v[m], s));
        }
      }
      const size_t plo = (size_t)B * N * CO;
      const size_t e = ((size_t)b * N + r) * CO + o;
      a.s_out[e] = os.r;
      a.s_out[plo + e] = os.i;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        a.v_out[e * 4 + m] = ov[m].r;
        a.v_out[plo * 4 + e * 4 + m] = ov[m].i;
      }
    }
  }
}

template <int C, bool DEC>
static size_t level_fwd_smem(int N, int CO, int IT, size_t tsize) {
  using L = Carve<C, DEC>;
  size_t n = L::even(N * L::NS) + L::even(N * L::PS) + L::even(L::RAD_SIZE) + L::even(4 * CO * 5 * C) + IT * 20 * C;
  return n * tsize + (size_t)N + 16;
}

template <typename T, int C, bool DEC>
static int launch_level_fwd(const LevelArgs<T>& a, hipStream_t stream) {
  constexpr int JS = 8;
  constexpr int IT = BLOCK / JS;
  size_t smem = level_fwd_smem<C, DEC>(a.N, a.CO, IT, sizeof(T));
  LGN_CHECK_ARG(smem <= 160 * 1024, "level_fwd: N=%d C=%d needs %zu B of LDS (> 160 KiB)", a.N, a.C, smem);
  auto kern = level_fwd_kernel<T, C, JS, DEC>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  dim3 grid(a.B, cdiv(a.N, IT));
  hipLaunchKernelGGL(kern, grid, dim3(BLOCK), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

int level_fwd2_dispatch(const LevelArgs<double>& a, int decoder, hipStream_t stream);   // level_fwd2.hip

static bool use_v1() {
  static const bool v = [] { const char* e = getenv("LGN_AMD_LEVEL_V1"); return e && e[0] == '1'; }();
  return v;
}

template <typename T>
int level_fwd_dispatch(const LevelArgs<T>& a, int decoder, hipStream_t stream) {
  if (!use_v1()) return level_fwd2_dispatch(a, decoder, stream);      // matrix-core version (default)
  LGN_CHECK_ARG(a.B > 0 && a.N > 0, "level_fwd: empty batch (B=%d N=%d)", a.B, a.N);
  LGN_CHECK_ARG(a.CO >= 1 && a.CO <= 8, "level_fwd: C_out=%d unsupported (1..8)", a.CO);
  LGN_CHECK_ARG(a.B <= 65535 * 32, "level_fwd: batch too large");
#define LGN_CASE(CC)                                                              \
  case CC:                                                                        \
    return decoder ? launch_level_fwd<T, CC, true>(a, stream) : launch_level_fwd<T, CC, false>(a, stream);
  switch (a.C) {
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
    default:
      set_error("level_fwd: C_in=%d unsupported (1..8)", a.C);
      return -1;
  }
#undef LGN_CASE
}

template int level_fwd_dispatch<double>(const LevelArgs<double>&, int, hipStream_t);

}  // namespace lgn
