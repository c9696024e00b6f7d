// lgn-autoencoder_amd/csrc/level_bwd_dec_sep.hip -- decoder level backward in separable form for jets that do not fit
// level_bwd3 (N > 40): runs after level_bwd_mix_kernel (which leaves g_ag in `a.g_ag` and the direct part of the node
// gradient in g_s_in / g_v_in) and replaces the two pair sweeps (level_bwd_nodes2<DEC> + level_bwd_rad_dec).
//
// The decoder's edge mask is identically zero, so its radial weights are the per-channel constants R0, R1 and every sum
// over pairs separates into jet-level sums (derivation and notation: level_fwd2.hip / level_bwd3.hip, SEP).  One workgroup
// per jet; the (node, channel) terms of the sums go through LDS in slabs of 32 nodes and are added in node order.
#include "level_dev.hpp"
#include "ops.hpp"

namespace lgn {

namespace {
constexpr int SLAB = 32;
template <int C> struct GAS {
  static constexpr int A3 = 0, A4 = 2 * C, A1 = 4 * C, A2 = 12 * C, SIZE = 20 * C;
};
}  // namespace

template <int C>
__global__ __launch_bounds__(BLOCK) void level_bwd_dec_sep_kernel(LevelBwdArgs<double> a) {
  using G = GAS<C>;
  const int N = a.N, B = a.B;
  const int b = blockIdx.x, tid = threadIdx.x;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* pj = reinterpret_cast<double*>(smem_raw);     // N * 8 centred canonical momenta (re[4], im[4])
  double* tr = pj + N * 8;                              // SLAB * C * 20 terms
  double* sm = tr + SLAB * C * 20;                      // 50 C jet-level sums (layout of level_bwd3.hip)
  const size_t pls = (size_t)B * N * C, plp = (size_t)B * N * 4;

  for (int e = tid; e < N * 4; e += BLOCK) {
    pj[(e >> 2) * 8 + (e & 3)] = a.p[(size_t)b * N * 4 + e];
    pj[(e >> 2) * 8 + 4 + (e & 3)] = a.p[plp + (size_t)b * N * 4 + e];
  }
  __syncthreads();
  if (tid < 64) {                                       // centre on the jet mean (only differences p_i - p_j enter):
    const int k = tid & 7, part = tid >> 3;             // lane = (node part, component), parts meet by shuffles
    double mean = 0.0;
    for (int n = part; n < N; n += 8) mean += pj[n * 8 + k];
    mean += shfl_xor(mean, 8);
    mean += shfl_xor(mean, 16);
    mean += shfl_xor(mean, 32);
    if (part == 0) sm[k] = mean / N;
  }
  __syncthreads();
  for (int e = tid; e < N * 8; e += BLOCK) pj[e] -= sm[e & 7];
  __syncthreads();

  // ---- jet-level sums: S 0 | VS[m] 2+2m | SP[m] 10+2m | VP 18 | SG4 20 | SG3 22 | SG1[m] 24+2m | SG2[m] 32+2m | GP2 40 | GP3[m] 42+2m
#pragma unroll
  for (int round = 0; round < 3; ++round) {
    const int nv = round == 2 ? 10 : 20;
    double total = 0.0;
    for (int n0 = 0; n0 < N; n0 += SLAB) {
      const int rows = min(SLAB, N - n0);
      for (int e = tid; e < rows * C; e += BLOCK) {
        const int rl = e / C, c = e - rl * C, n = n0 + rl;
        const size_t ge = ((size_t)b * N + n) * C + c;
        const double* pn = pj + n * 8;
        const double* gi = a.g_ag + ((size_t)b * N + n) * G::SIZE;
        double* t = tr + e * 20;
        cx<double> pc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) pc[m] = {pn[m], pn[4 + m]};
        if (round == 0) {
          const cx<double> sn = {a.s_in[ge], a.s_in[pls + ge]};
          cx<double> v[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) v[m] = {a.v_in[ge * 4 + m], a.v_in[pls * 4 + ge * 4 + m]};
          t[0] = sn.r;  t[1] = sn.i;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            t[2 + 2 * m] = v[m].r;  t[3 + 2 * m] = v[m].i;
            const cx<double> sp = cmul(sn, pc[m]);
            t[10 + 2 * m] = sp.r;  t[11 + 2 * m] = sp.i;
          }
          const cx<double> vp = bil2(v, pc);
          t[18] = vp.r;  t[19] = vp.i;
        } else if (round == 1) {
          t[0] = gi[G::A4 + 2 * c];        t[1] = gi[G::A4 + 2 * c + 1];
          t[2] = 0.5 * gi[G::A3 + 2 * c];  t[3] = 0.5 * gi[G::A3 + 2 * c + 1];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            t[4 + 2 * m] = gi[G::A1 + (c * 4 + m) * 2];   t[5 + 2 * m] = gi[G::A1 + (c * 4 + m) * 2 + 1];
            t[12 + 2 * m] = gi[G::A2 + (c * 4 + m) * 2];  t[13 + 2 * m] = gi[G::A2 + (c * 4 + m) * 2 + 1];
          }
        } else {
          const cx<double> g3 = {0.5 * gi[G::A3 + 2 * c], 0.5 * gi[G::A3 + 2 * c + 1]};
          cx<double> gp2 = {0, 0};
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            cfmac(gp2, cx<double>{gi[G::A2 + (c * 4 + m) * 2], gi[G::A2 + (c * 4 + m) * 2 + 1]}, pc[m]);
            const cx<double> gp3 = cmulc(g3, pc[m]);
            t[2 + 2 * m] = gp3.r;  t[3 + 2 * m] = gp3.i;
          }
          t[0] = gp2.r;  t[1] = gp2.i;
        }
      }
      __syncthreads();
      if (tid < nv * C) {
        const int c = tid / nv, k = tid - c * nv;
#pragma unroll 8
        for (int rl = 0; rl < rows; ++rl) total += tr[(rl * C + c) * 20 + k];
      }
      __syncthreads();
    }
    if (tid < nv * C) {
      const int c = tid / nv, k = tid - c * nv;
      sm[c * 50 + round * 20 + k] = total;
    }
  }
  __syncthreads();

  // ---- node gradient: the neighbour part is added to the direct part already in g_s_in / g_v_in ------------------------
  for (int e = tid; e < N * C; e += BLOCK) {
    const int n = e / C, c = e - n * C;
    const double* q = sm + c * 50;
    const double* pn = pj + n * 8;
    const cx<double> R0 = {a.b0[c], a.b0[c]}, R1 = {a.b1[c], a.b1[c]};
    const cx<double> e0 = {R0.r - R0.i, R0.r + R0.i};
    cx<double> pc[4], pt[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pc[m] = {pn[m], pn[4 + m]};
    metric_perm(pc, pt);
    const cx<double> SG4 = {q[20], q[21]}, SG3 = {q[22], q[23]}, GP2 = {q[40], q[41]};
    cx<double> GP3[4], GPT3[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) GP3[m] = {q[42 + 2 * m], q[43 + 2 * m]};
    metric_perm(GP3, GPT3);
    cx<double> u = GP2;                                  // Gs = conj(e0) SG4 + conj(R1) (GP2 - sum_m SG2[m] conj(p[m]))
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const cx<double> t = cmulc(cx<double>{q[32 + 2 * m], q[33 + 2 * m]}, pc[m]);
      u.r -= t.r;  u.i -= t.i;
    }
    cx<double> gs = cmulc(SG4, e0);
    cfmac(gs, u, R1);
    const size_t ge = ((size_t)b * N + n) * C + c;
    a.g_s_in[ge] += gs.r;
    a.g_s_in[pls + ge] += gs.i;
#pragma unroll
    for (int m = 0; m < 4; ++m) {                        // Gv[m] = conj(e0) SG1[m] + conj(R1) (GPT3[m] - SG3 conj(pt[m]))
      cx<double> w = GPT3[m];
      const cx<double> t = cmulc(SG3, pt[m]);
      w.r -= t.r;  w.i -= t.i;
      cx<double> gv = cmulc(cx<double>{q[24 + 2 * m], q[25 + 2 * m]}, e0);
      cfmac(gv, w, R1);
      a.g_v_in[ge * 4 + m] += gv.r;
      a.g_v_in[pls * 4 + ge * 4 + m] += gv.i;
    }
  }
  // ---- position gradient: d p_n[m] += sum_c conj(R1) [ gA2_n[m] conj(S) + gA3_n conj(VSt[m]) - SG2[m] conj(s_n) - SG3 conj(vt_n[m]) ]
  for (int e = tid; e < N * 4; e += BLOCK) {
    const int n = e >> 2, m = e & 3;
    const int mp = m == 1 ? 3 : (m == 3 ? 1 : m);        // metric_perm index; component 2 changes sign
    const double sg = m == 2 ? -1.0 : 1.0;
    const double* gi = a.g_ag + ((size_t)b * N + n) * G::SIZE;
    cx<double> acc = {0, 0};
    for (int c = 0; c < C; ++c) {
      const double* q = sm + c * 50;
      const size_t ge = ((size_t)b * N + n) * C + c;
      const cx<double> R1 = {a.b1[c], a.b1[c]};
      const cx<double> S = {q[0], q[1]}, SG3 = {q[22], q[23]};
      const cx<double> VSt = {sg * q[2 + 2 * mp], sg * q[3 + 2 * mp]};
      const cx<double> vt = {sg * a.v_in[ge * 4 + mp], sg * a.v_in[pls * 4 + ge * 4 + mp]};
      const cx<double> g2 = {gi[G::A2 + (c * 4 + m) * 2], gi[G::A2 + (c * 4 + m) * 2 + 1]};
      const cx<double> g3 = {0.5 * gi[G::A3 + 2 * c], 0.5 * gi[G::A3 + 2 * c + 1]};
      cx<double> t = cmulc(g2, S);
      cfmac(t, g3, VSt);
      const cx<double> t2 = cmulc(cx<double>{q[32 + 2 * m], q[33 + 2 * m]}, cx<double>{a.s_in[ge], a.s_in[pls + ge]});
      t.r -= t2.r;  t.i -= t2.i;
      const cx<double> t3 = cmulc(SG3, vt);
      t.r -= t3.r;  t.i -= t3.i;
      cfmac(acc, t, R1);
    }
    a.g_p[((size_t)b * N + n) * 4 + m] += acc.r;
    a.g_p[plp + ((size_t)b * N + n) * 4 + m] += acc.i;
  }
  // ---- bias gradients of this jet ---------------------------------------------------------------------------
  if (tid < 2 * C) {
    const int lin = tid / C, c = tid - lin * C;
    const double* q = sm + c * 50;
    const cx<double> S = {q[0], q[1]};
    double* part = a.part_rad + (size_t)b * rad_partial_size(C, true);
    if (lin == 0) {
      cx<double> E0 = cmulc(cx<double>{q[20], q[21]}, S);
#pragma unroll
      for (int m = 0; m < 4; ++m) cfmac(E0, cx<double>{q[24 + 2 * m], q[25 + 2 * m]}, cx<double>{q[2 + 2 * m], q[3 + 2 * m]});
      part[tid] = (E0.r + E0.i) + (E0.i - E0.r);          // R0 = b0 (1+i): d b0 = Re G_R0 + Im G_R0, G_R0 = G_e0 (1-i)
    } else {
      cx<double> VS[4], VSt[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) VS[m] = {q[2 + 2 * m], q[3 + 2 * m]};
      metric_perm(VS, VSt);
      cx<double> E1 = cmulc(cx<double>{q[40], q[41]}, S);
      cx<double> neg = cmulc(cx<double>{q[22], q[23]}, cx<double>{q[18], q[19]});
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        cfmac(neg, cx<double>{q[32 + 2 * m], q[33 + 2 * m]}, cx<double>{q[10 + 2 * m], q[11 + 2 * m]});
        cfmac(E1, cx<double>{q[42 + 2 * m], q[43 + 2 * m]}, VSt[m]);
      }
      part[tid] = (E1.r - neg.r) + (E1.i - neg.i);
    }
  }
}

template <int C>
static int launch_dec_sep(const LevelBwdArgs<double>& a, hipStream_t stream) {
  const size_t smem = sizeof(double) * ((size_t)a.N * 8 + SLAB * C * 20 + 50 * C);
  LGN_CHECK_ARG(smem <= 160 * 1024, "level_bwd (decoder): N=%d needs %zu B of LDS", a.N, smem);
  auto kern = level_bwd_dec_sep_kernel<C>;
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(BLOCK), smem, stream, a);
  LGN_CHECK_LAUNCH();
  return 0;
}

// neighbour part of the decoder level backward from jet-level sums; one radial partial row per jet
int level_bwd_dec_sep_dispatch(const LevelBwdArgs<double>& a, hipStream_t stream) {
#define LGN_CASE(CC) case CC: return launch_dec_sep<CC>(a, stream);
  switch (a.C) {
    LGN_CASE(1) LGN_CASE(2) LGN_CASE(3) LGN_CASE(4) LGN_CASE(5) LGN_CASE(6) LGN_CASE(7) LGN_CASE(8)
    default: set_error("level_bwd: C_in=%d unsupported (1..8)", a.C); return -1;
  }
#undef LGN_CASE
}

}  // namespace lgn
