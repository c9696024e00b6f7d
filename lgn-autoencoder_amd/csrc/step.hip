// lgn-autoencoder_amd/csrc/step.hip -- one training step of the LGN autoencoder as a single native call.
//
// Counterpart of the inner loop of the reference's utils/train.py:283-343 (encoder -> decoder ->
// get_real('sum') -> Chamfer -> backward) for the maxdim=2 path: ~80 kernel launches enqueued back to back
// on the caller's stream, no host synchronisation, every buffer caller-owned and static -> the call can be
// captured into a HIP graph (the Python harness does so) and replayed.  Parameter gradients are written
// straight into the caller's flat gradient buffer at the same offsets as the parameters.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "net.hpp"
#include "../../include/lgn_amd.h"

namespace lgn {
namespace {

struct Bump {                       // workspace carving (also used for the size query with base == nullptr)
  double* base;
  size_t off = 0;
  double* take(size_t n) {
    n = (n + 15) & ~size_t(15);     // 128-byte granules
    double* p = base ? base + off : nullptr;
    off += n;
    return p;
  }
};

struct NetBuf {                     // per-network activations kept for the backward pass
  double *s[5], *v[5], *smix[4], *ag0[4], *ag1[4];
  double* hsave[4];                 // hidden activations of the level's CGMLP, kept for its backward (null: recomputed)
};

struct Work {
  NetBuf enc, dec;
  double *lat_s, *lat_v, *pdec, *g_lat_s, *g_lat_v, *g_p;
  double *gs[2], *gv[2], *gsmix, *g_ag, *zeros_s;
  double *parts;               // bump region: every producer of partial rows gets its own slice (reduced at the end)
  size_t parts_size;
  double* tot[2][4];           // reduced radial sums per (network, level)
  int* idx;
  size_t total;
  double* tail_cnt;            // 4 counters of the fused reduction (step_tail.hip, reduce_only), inside the zero block
  double* zero0() const { return tail_cnt < zeros_s ? tail_cnt : zeros_s; }     // start of the zero block
  size_t zero_doubles;         // zeros_s | g_p | g_lat_s | tail_cnt are contiguous: one memset
};

// deferred reductions: producers register their column ranges, one or two launches at the end reduce everything
struct Deferred {
  std::vector<RedSeg<double>> segs;
  double* parts;
  size_t off = 0, cap = 0;
  // nullptr when the slice does not fit: callers test it (DQ_TAKE) BEFORE enqueuing the kernel that would write there --
  // the capacity is a hand-maintained mirror of the take sequence (carve*), a disagreement must not reach the device
  double* take(size_t n) {
    n = (n + 15) & ~size_t(15);
    if (off + n > cap) return nullptr;
    double* p = parts + off;
    off += n;
    return p;
  }
  void add(const double* part, int rows, int stride, int col0, int n, double* out) {
    if (n > 0) segs.push_back(RedSeg<double>{part, rows, stride, col0, n, out});
  }
  int flush(hipStream_t st) {
    for (size_t i = 0; i < segs.size(); i += RED_MAX_SEG) {
      RedJob<double> job{};
      for (size_t k = i; k < segs.size() && k < i + RED_MAX_SEG; ++k) job.seg[job.nseg++] = segs[k];
      if (int rc = reduce_segments<double>(job, st)) return rc;
    }
    segs.clear();
    return 0;
  }
};

// The end of a backward whose gradients go on to somebody else (the all-reduce of a data-parallel step, autograd under the module
// API): every deferred reduction and the radial finalisation as ONE launch (step_tail.hip, reduce_only) -- `counters`: 4 words of the
// caller's zero block -- or, when that form does not fit (or with LGN_NET_SPLIT_TAIL), as reduce_segments + rad_finalize_batch.
int finish_reductions(Deferred& dq, const RadFinJob& fin, double* grads, long long n_params, double* counters, int flags, hipStream_t st) {
  if (counters && !(flags & LGN_NET_SPLIT_TAIL)) {
    StepTailArgs ro{};
    ro.g = grads;
    ro.n = (long)n_params;
    ro.reduce_only = 1;
    ro.counters = reinterpret_cast<unsigned long long*>(counters);
    const int rc = step_tail(dq.segs, fin, ro, st);
    if (rc == 0) { dq.segs.clear(); return 0; }
    if (rc != -2) return rc;
  }
  if (int rc = dq.flush(st)) return rc;
  return rad_finalize_batch(fin, st);
}

// partial rows of level l's CGMLP weight gradients
size_t mlp_part_rows(const lgn_net_desc& d, bool dec, int l) {
  const int* ch = dec ? d.dec_channels : d.enc_channels;
  return (size_t)mlp_partial_rows(d.B * (dec && d.dec_N > 0 ? d.dec_N : d.N), d.mlp_hidden_mul * 2 * ch[l + 1]);
}

inline int in_K(const lgn_net_desc& d) { return d.n_in_scalars > 1 ? d.n_in_scalars : 1; }      // encoder input scalars per node

int mlp_psize(int C, int H, int nlin) {
  const int D = 2 * C;
  return nlin == 1 ? D * D + D : (H * D + H) + (nlin - 2) * (H * H + H) + (D * H + D);
}

// decoder node count of a whole step: its own (jet_features gives the encoder one node more than the decoder reconstructs), or N
inline int dec_nodes(const lgn_net_desc& d) { return d.dec_N > 0 ? d.dec_N : d.N; }
// the step takes the two-kernel junction and the riding input stage only when both networks work on the same nodes and the mass is
// the only input scalar; otherwise the four end stages are launches of their own
inline bool step_is_split(const lgn_net_desc& d) { return dec_nodes(d) != d.N || d.n_in_scalars > 1; }

Work carve(const lgn_net_desc& d, double* base) {
  Work w{};
  Bump b{base};
  const int Nd = dec_nodes(d), Nmax = d.N > Nd ? d.N : Nd;
  const size_t BNe = (size_t)d.B * d.N, BNd = (size_t)d.B * Nd, BN = (size_t)d.B * Nmax;
  const int L = d.n_levels;
  int cmax = 0;
  for (int l = 0; l <= L; ++l) cmax = cmax > d.enc_channels[l] ? cmax : d.enc_channels[l], cmax = cmax > d.dec_channels[l] ? cmax : d.dec_channels[l];
  auto net = [&](NetBuf& n, const int* ch, bool dec) {
    const size_t BN = dec ? BNd : BNe;
    for (int l = 0; l <= L; ++l) {
      n.s[l] = b.take(2 * BN * ch[l]);
      n.v[l] = b.take(8 * BN * ch[l]);
    }
    for (int l = 0; l < L; ++l) {
      n.smix[l] = b.take(2 * BN * ch[l + 1]);
      n.ag0[l] = b.take(4 * BN * ch[l]);
      n.ag1[l] = b.take(16 * BN * ch[l]);
      // the last level's scalars never reach the loss: its CGMLP has no backward
      // (a CGMLP that rides on the level kernels recomputes: nothing kept)
      const size_t hs = l + 1 < L && BN <= mlp_save_max_rows()
                            ? mlp_saved_doubles((int)BN, d.mlp_hidden_mul * 2 * ch[l + 1], d.mlp_nlin) : 0;
      n.hsave[l] = hs ? b.take(hs) : nullptr;
    }
  };
  net(w.enc, d.enc_channels, false);
  net(w.dec, d.dec_channels, true);
  const int Ts = d.tau_s, Tv = d.tau_v, PB = pool_blocks(d.latent_pool);
  w.lat_s = b.take((size_t)2 * d.B * PB * Ts);
  w.lat_v = b.take((size_t)2 * d.B * PB * Tv * 4);
  w.g_lat_v = b.take((size_t)2 * d.B * PB * Tv * 4);
  w.pdec = b.take(8 * BN);
  for (int q = 0; q < 2; ++q) {
    w.gs[q] = b.take(2 * BN * cmax);
    w.gv[q] = b.take(8 * BN * cmax);
  }
  w.gsmix = b.take(2 * BN * cmax);
  w.g_ag = b.take(20 * BN * cmax);
  {  // contiguous zero-initialised region
    const size_t z0 = b.off;
    w.zeros_s = b.take(2 * BN * cmax);
    w.g_p = b.take(8 * BN);
    w.g_lat_s = b.take((size_t)2 * d.B * PB * d.tau_s);
    w.tail_cnt = b.take(8);
    w.zero_doubles = b.off - z0;
  }
  w.idx = reinterpret_cast<int*>(b.take(((size_t)d.B * 2 * (Ts + Tv) * 2 + 1) / 2 + 8));
  // partial rows: every producer keeps its own slice until the deferred reduction at the end of the step
  size_t psum = 0;
  for (int dec = 0; dec < 2; ++dec) {
    const int* ch = dec ? d.dec_channels : d.enc_channels;
    for (int l = 0; l < L; ++l) {
      int rm, rr;
      level_bwd_partial_rows(d.B, dec ? Nd : d.N, dec, d.flags, &rm, &rr);
      const size_t nmix = (size_t)4 * ch[l + 1] * 5 * ch[l], nrad = rad_partial_size(ch[l], dec != 0);
      psum += ((rm * nmix + 15) & ~size_t(15)) + ((rr * nrad + 15) & ~size_t(15));
      psum += (mlp_part_rows(d, dec, l) * mlp_psize(ch[l + 1], d.mlp_hidden_mul * 2 * ch[l + 1], d.mlp_nlin) + 15) & ~size_t(15);
      w.tot[dec][l] = b.take(nrad + 16);
    }
  }
  psum += 4 * (((size_t)d.B * (4 * cmax + 2 * (size_t)Nmax * PB * Tv + 2 * (size_t)(Ts + Tv) * pool_mix_in(d.latent_pool, Nmax, cmax)) + 15) & ~size_t(15));
  psum += ((size_t)d.B * (2 * in_K(d) + 2) * cmax + 15) & ~size_t(15);                  // input stage with K scalars (split step)
  w.parts = b.take(psum);
  w.parts_size = psum;
  w.total = b.off;
  return w;
}

struct Slots {                      // canonical parameter slot order shared with lgn/step.py
  int L, nlin;
  int in0(bool dec) const { return dec ? 2 : 0; }
  int rad(bool dec, int l, int k) const { return in0(dec) + 2 + 7 * l + k; }
  int mix(bool dec, int l, int k) const { return in0(dec) + 2 + 7 * L + 2 * l + k; }
  int mlp(bool dec, int l, int k) const { return in0(dec) + 2 + 9 * L + 2 * nlin * l + k; }
  int out0(bool dec) const { return in0(dec) + 2 + 9 * L + 2 * nlin * L; }
  int count(bool dec) const { return out0(dec) + 2; }
};

#define HIPOK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { set_error("%s: %s", #e, hipGetErrorString(e_)); return (int)e_; } } while (0)
#define DQ_NEW(var, n) double* var = nullptr; DQ_TAKE(var, n)
#define DQ_TAKE(lhs, n)                                                                                           \
  lhs = dq.take(n);                                                                                             \
  LGN_CHECK_ARG((lhs) != nullptr, "partial-row workspace overflow: %zu + %zu doubles > capacity %zu (carve / take mismatch)", \
                dq.off, (size_t)(n), dq.cap)
#define LGN_TRY(expr)            \
  do {                           \
    int rc_ = (expr);            \
    if (rc_ != 0) return rc_;    \
  } while (0)

// The encoder's input stage (input_func_node on mass and canonical momenta) rides on the first level's kernel, which then
// is the first kernel of the call and also clears z1 / z2 (LevelArgs::in_w0).
struct InputStage {
  const double *w0, *w1;
  double* z1 = nullptr;
  size_t z1n = 0;
  double* z2 = nullptr;
  size_t z2n = 0;
};
// The decoder output + Chamfer loss (forward and backward) rides on the decoder's last level forward (LevelArgs::loss_wo1).
struct LossStage {
  const double *wo1, *target;
  double scale;
  double *recon, *loss_part, *g_v, *wpart;
};

// forward of one network's level stack; returns via buffers
int levels_fwd(const lgn_net_desc& d, bool dec, const int* ch, const double* P, const int64_t* off, NetBuf& n, const double* pos,
               const uint8_t* mask, hipStream_t st, const InputStage* in0 = nullptr, const LossStage* loss = nullptr) {
  const Slots S{d.n_levels, d.mlp_nlin};
  for (int l = 0; l < d.n_levels; ++l) {
    auto p = [&](int slot) { return P + off[slot]; };
    LevelArgs<double> a{d.B, d.N, ch[l], ch[l + 1], n.s[l], n.v[l], pos, mask,
                        p(S.rad(dec, l, 0)), p(S.rad(dec, l, 1)), p(S.rad(dec, l, 2)), p(S.rad(dec, l, 3)), p(S.rad(dec, l, 4)),
                        p(S.rad(dec, l, 5)), p(S.rad(dec, l, 6)), p(S.mix(dec, l, 0)), p(S.mix(dec, l, 1)),
                        n.ag0[l], n.ag1[l], n.smix[l], n.v[l + 1]};
    if (l == 0 && !dec && in0) {
      a.in_w0 = in0->w0; a.in_w1 = in0->w1; a.in_s = n.s[0]; a.in_v = n.v[0];
      a.z1 = in0->z1; a.z1n = in0->z1n; a.z2 = in0->z2; a.z2n = in0->z2n;
    }
    if (l + 1 == d.n_levels && dec && loss) {
      a.loss_wo1 = loss->wo1; a.loss_target = loss->target; a.loss_scale = loss->scale; a.loss_recon = loss->recon;
      a.loss_part = loss->loss_part; a.loss_gv = loss->g_v; a.loss_wpart = loss->wpart;
    }
    a.flags = d.flags;
    LGN_TRY(level_fwd_dispatch<double>(a, dec, st));
    MlpArgs<double> m{};
    m.M = d.B * d.N; m.C = ch[l + 1]; m.H = d.mlp_hidden_mul * 2 * ch[l + 1]; m.nlin = d.mlp_nlin; m.act = d.activation; m.flags = d.flags;
    for (int q = 0; q < d.mlp_nlin; ++q) { m.w[q] = p(S.mlp(dec, l, 2 * q)); m.b[q] = p(S.mlp(dec, l, 2 * q + 1)); }
    m.s_in = n.smix[l]; m.s_out = n.s[l + 1];
    m.h_saved = n.hsave[l]; m.h_rows = mlp_saved_rows(m.M);
    LGN_TRY(mlp_dispatch<double>(m, false, st));
  }
  return 0;
}

// backward of one network's level stack.  On entry gs[cur]/gv[cur] hold the gradient w.r.t. (s[L], v[L]);
// has_s_grad says whether gs is non-zero (false for both networks of the autoencoder: the last level's
// scalars never reach the loss).  On exit gs[cur]/gv[cur] hold the gradient w.r.t. level 0.
// in0_grads (encoder only, optional): gradient slots of input_func_node's two weights.  When the first level's backward is the
// one-kernel form, the input stage's backward rides on it (LevelBwdArgs::part_in0) and *in0_done is set; otherwise the caller
// launches enc_input_bwd.
int levels_bwd(const lgn_net_desc& d, bool dec, const int* ch, const double* P, double* G, const int64_t* off, const NetBuf& n,
               const double* pos, const uint8_t* mask, Work& w, Deferred& dq, RadFinJob& fin, int& cur, bool has_s_grad,
               hipStream_t st, double* const* in0_grads = nullptr, bool* in0_done = nullptr) {
  const Slots S{d.n_levels, d.mlp_nlin};
  const int BN = d.B * d.N;
  for (int l = d.n_levels - 1; l >= 0; --l) {
    auto p = [&](int slot) { return P + off[slot]; };
    auto g = [&](int slot) { return G + off[slot]; };
    const int C = ch[l], CO = ch[l + 1];
    const double* g_smix = w.zeros_s;
    if (has_s_grad) {
      MlpArgs<double> m{};
      m.M = BN; m.C = CO; m.H = d.mlp_hidden_mul * 2 * CO; m.nlin = d.mlp_nlin; m.act = d.activation; m.flags = d.flags;
      for (int q = 0; q < d.mlp_nlin; ++q) { m.w[q] = p(S.mlp(dec, l, 2 * q)); m.b[q] = p(S.mlp(dec, l, 2 * q + 1)); }
      m.s_in = n.smix[l]; m.g_out = w.gs[cur]; m.g_in = w.gsmix;
      m.h_saved = n.hsave[l];
      m.h_rows = mlp_saved_rows(BN);
      m.psize = mlp_psize(CO, m.H, m.nlin);
      DQ_TAKE(m.part, (size_t)mlp_partial_rows(BN, m.H) * m.psize);
      LGN_TRY(mlp_dispatch<double>(m, true, st));
      // the MLP's parameters are contiguous in the flat buffer in (W_0, b_0, W_1, ...) order (checked at plan time)
      dq.add(m.part, mlp_partial_rows(BN, m.H), m.psize, 0, m.psize, g(S.mlp(dec, l, 0)));
      g_smix = w.gsmix;
    }
    int rm, rr;
    level_bwd_partial_rows(d.B, d.N, dec, d.flags, &rm, &rr);
    const int nmix = 4 * CO * 5 * C, nrad = rad_partial_size(C, dec);
    DQ_NEW(part_mix, (size_t)rm * nmix);
    DQ_NEW(part_rad, (size_t)rr * nrad);
    const int nxt = cur ^ 1;
    LevelBwdArgs<double> a{d.B, d.N, C, CO, n.s[l], n.v[l], pos, mask,
                           p(S.rad(dec, l, 0)), p(S.rad(dec, l, 1)), p(S.rad(dec, l, 2)), p(S.rad(dec, l, 3)), p(S.rad(dec, l, 4)),
                           p(S.rad(dec, l, 5)), p(S.rad(dec, l, 6)), p(S.mix(dec, l, 0)), p(S.mix(dec, l, 1)), n.ag0[l], n.ag1[l],
                           g_smix, w.gv[cur], w.g_ag, w.gs[nxt], w.gv[nxt], dec ? w.g_p : nullptr, part_mix, part_rad};
    const bool carry_in0 = !dec && l == 0 && in0_grads && level_bwd_carries_input(d.N, d.flags);
    if (carry_in0) { DQ_TAKE(a.part_in0, (size_t)rm * 4 * C); }
    a.flags = d.flags;
    LGN_TRY(level_bwd_dispatch<double>(a, dec, st));
    if (carry_in0) {
      dq.add(a.part_in0, rm, 4 * C, 0, 2 * C, in0_grads[0]);
      dq.add(a.part_in0, rm, 4 * C, 2 * C, 2 * C, in0_grads[1]);
      *in0_done = true;
    }
    // deferred reductions: CatMix weights (partial row = [wm0 | wm1]) + radial sums
    dq.add(part_mix, rm, nmix, 0, nmix / 2, g(S.mix(dec, l, 0)));
    dq.add(part_mix, rm, nmix, nmix / 2, nmix / 2, g(S.mix(dec, l, 1)));
    if (dec) {   // only the Linear biases receive gradient (all edges are "masked")
      dq.add(part_rad, rr, nrad, 0, C, g(S.rad(dec, l, 4)));
      dq.add(part_rad, rr, nrad, C, C, g(S.rad(dec, l, 6)));
    } else {
      double* tot = w.tot[0][l];
      dq.add(part_rad, rr, nrad, 0, nrad, tot);
      fin.it[fin.n++] = RadFinJob::Item{tot, C, p(S.rad(dec, l, 0)), p(S.rad(dec, l, 1)), p(S.rad(dec, l, 2)), p(S.rad(dec, l, 3)),
                                        p(S.rad(dec, l, 5)), g(S.rad(dec, l, 0)), g(S.rad(dec, l, 1)), g(S.rad(dec, l, 2)),
                                        g(S.rad(dec, l, 3)), g(S.rad(dec, l, 4)), g(S.rad(dec, l, 5)), g(S.rad(dec, l, 6))};
    }
    cur = nxt;
    has_s_grad = true;       // the level input scalars do carry gradient
  }
  return 0;
}

int check_desc(const lgn_net_desc* d) {
  LGN_CHECK_ARG(d, "step: null descriptor");
  LGN_CHECK_ARG(d->B > 0 && d->N > 0, "step: empty batch (B=%d N=%d)", d->B, d->N);
  LGN_CHECK_ARG(d->n_levels >= 1 && d->n_levels <= 4, "step: n_levels=%d unsupported (1..4)", d->n_levels);
  LGN_CHECK_ARG(d->mlp_nlin >= 4 && d->mlp_nlin <= 7, "step: mlp_depth must be 3 .. 6 (4 .. 7 Linear layers), got %d layers", d->mlp_nlin);
  LGN_CHECK_ARG(d->tau_s >= 1 && d->tau_v >= 1 && d->tau_v_in >= 0, "step: latent multiplicities must be positive");
  LGN_CHECK_ARG(d->n_in_scalars >= 0 && d->n_in_scalars <= 8, "step: n_in_scalars=%d unsupported (0..8)", d->n_in_scalars);
  LGN_CHECK_ARG(pool_valid(d->latent_pool), "step: latent_pool=%d is not an LGN_POOL(...) code", d->latent_pool);
  for (int l = 0; l <= d->n_levels; ++l)
    LGN_CHECK_ARG(d->enc_channels[l] >= 1 && d->enc_channels[l] <= 8 && d->dec_channels[l] >= 1 && d->dec_channels[l] <= 8,
                  "step: channel counts must be in 1..8");
  return 0;
}

}  // namespace

}  // namespace lgn


namespace lgn {
namespace {
// Zero-fill of up to three ranges in ONE launch of our own.  NOT hipMemsetAsync: under stream capture (ROCm 7.2, torch 2.10) the
// memset node of a replayed graph fills its range with a stale 16-byte pattern -- two pointer-like words, i.e. denormals ~7e-310 --
// from the second replay on (tools/graph_memset_check.py).  As gradients of dead parameters and "zero" upstream gradients that
// went unnoticed numerically; as counters it does not.
struct ZeroJob { double* p[3]; size_t n[3]; };
__global__ __launch_bounds__(BLOCK) void zero_ranges_kernel(ZeroJob job) {
  const size_t stride = (size_t)gridDim.x * BLOCK;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < job.n[q]; i += stride) job.p[q][i] = 0.0;
}
int zero_ranges(double* a, size_t na, double* b, size_t nb, double* c, size_t nc, hipStream_t st) {
  const ZeroJob job{{a, b, c}, {a ? na : 0, b ? nb : 0, c ? nc : 0}};
  const size_t most = std::max(job.n[0], std::max(job.n[1], job.n[2]));
  if (!most) return 0;
  const int blocks = (int)std::min<size_t>((most + BLOCK - 1) / BLOCK, 1024);
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(blocks), dim3(BLOCK), 0, st, job);
  LGN_CHECK_LAUNCH();
  return 0;
}
// grads [n] and the zero block [nz]: one range when the caller laid them out back to back (lgn/ops.py does)
int zero_grads_and_block(double* grads, size_t n, double* zeros, size_t nz, hipStream_t st) {
  if (zeros >= grads + n && zeros <= grads + n + 16) return zero_ranges(grads, (size_t)((zeros + nz) - grads), nullptr, 0, nullptr, 0, st);
  return zero_ranges(grads, n, zeros, nz, nullptr, 0, st);
}

int check_mlp_contiguous(const lgn_net_desc& d, bool dec, const int64_t* off) {
  const Slots S{d.n_levels, d.mlp_nlin};
  const int* ch = dec ? d.dec_channels : d.enc_channels;
  for (int l = 0; l < d.n_levels; ++l) {
    const int D = 2 * ch[l + 1], H = d.mlp_hidden_mul * D;
    int64_t expect = off[S.mlp(dec, l, 0)];
    for (int q = 0; q < d.mlp_nlin; ++q) {
      const int hin = q == 0 ? D : H, hout = q == d.mlp_nlin - 1 ? D : H;
      LGN_CHECK_ARG(off[S.mlp(dec, l, 2 * q)] == expect, "MLP weights are not contiguous in the flat parameter buffer");
      expect += (int64_t)hin * hout;
      LGN_CHECK_ARG(off[S.mlp(dec, l, 2 * q + 1)] == expect, "MLP biases are not contiguous in the flat parameter buffer");
      expect += hout;
    }
  }
  return 0;
}

}  // namespace
}  // namespace lgn

// ---------------------------------------------------------------------------------------------------------
// table-driven networks (maxdim = 3): same structure as above on packed features X_l [2][B][N][C_l][Q_l]
//   level forward : moments (generic_moments.hip) -> sparse CG + CatMix (generic_local.hip) -> CGMLP in place on the
//                   scalar column; backward in reverse.  Reference: LGNCG.forward lgn/models/lgn_cg.py:164-172.
// ---------------------------------------------------------------------------------------------------------
namespace lgn {
namespace {

inline bool is_generic(const lgn_net_desc& d, bool dec) { return (dec ? d.dec_tables[0] : d.enc_tables[0]) != nullptr; }

struct GenGeom {                   // per-network view of the descriptor
  const int *ch, *Q, *qs, *qv;
  const lgn_local_tables* const* tab;
};
inline GenGeom geom(const lgn_net_desc& d, bool dec) {
  return dec ? GenGeom{d.dec_channels, d.dec_Q, d.dec_qs, d.dec_qv, d.dec_tables}
             : GenGeom{d.enc_channels, d.enc_Q, d.enc_qs, d.enc_qv, d.enc_tables};
}

// every level has compile-time tables and the jets fit the channel-outermost moments kernels: tile-blocked feature layouts
// [tile][C][Q][2][64] end to end, generic_local_static.hip for the per-node part
inline bool is_static(const lgn_net_desc& d, bool dec) {
  const GenGeom g = geom(d, dec);
  if (d.N > 32) return false;
  for (int l = 0; l < d.n_levels; ++l)
    if (!g.tab[l] || g.tab[l]->static_kind == 0) return false;
  return !(d.flags & LGN_NET_NO_STATIC);                // run-time-table kernels (cross-check): fixed at descriptor creation
}
// decoder levels on the static kernels keep their separable moments on chip (generic_local_sep.hip, round 6): no U / dU tensors,
// one launch per level backward.  LGN_NET_DEC_UNFUSED: the round-5 sequence (dec_sep_fwd/bwd_tb + local_fwd/bwd_static), cross-check
inline bool is_sep_fused(const lgn_net_desc& d, bool dec) {
  return dec && is_static(d, dec) && !(d.flags & (LGN_NET_DEC_PAIRWISE | LGN_NET_DEC_UNFUSED));
}
inline size_t tb_doubles(const lgn_net_desc& d, int C, int Qx) { return (size_t)(((size_t)d.B * d.N + 63) / 64) * C * Qx * 128; }

int check_generic(const lgn_net_desc& d, bool dec) {
  const GenGeom g = geom(d, dec);
  for (int l = 0; l < d.n_levels; ++l) {
    LGN_CHECK_ARG(g.tab[l], "table-driven network: level %d has no tables (every level needs one)", l);
    LGN_CHECK_ARG(g.tab[l]->n_w > 0 && g.tab[l]->n_rows > 0, "table-driven network: empty tables at level %d", l);
  }
  for (int l = 0; l <= d.n_levels; ++l)
    LGN_CHECK_ARG(g.Q[l] >= 5 && g.Q[l] <= 64 && g.qs[l] >= 0 && g.qs[l] < g.Q[l] && g.qv[l] >= 0 && g.qv[l] + 4 <= g.Q[l],
                  "table-driven network: bad component layout at level %d (Q=%d qs=%d qv=%d)", l, g.Q[l], g.qs[l], g.qv[l]);
  LGN_CHECK_ARG(g.Q[0] == 5, "table-driven network: the input level carries (1,1) and (0,0) only (Q=5), got %d", g.Q[0]);
  return 0;
}

struct GenAct {                     // written by the forward, read by the backward
  double *s0, *v0;                  // input-kernel outputs [2][BN][C0], [2][BN][C0][4]
  double *X[5], *U[4], *smix[4];
  double* wp[4];                    // static path: packed CatMix weights of each level (written by the forward, reused by the backward)
  double* tbl[4];                   // fused decoder levels: jet table of each level (dec_sep_tab), instead of U
  double* pc;                       // ... and the nodes' centred momenta [B N][8]
  double *sL, *vL;                  // (0,0) / (1,1) of the last level, unpacked for the end kernels
  double* pdec;
  int* idx;
  size_t total;
};
GenAct carve_gen_act(const lgn_net_desc& d, bool dec, double* base) {
  GenAct a{};
  Bump b{base};
  const GenGeom g = geom(d, dec);
  const size_t BN = (size_t)d.B * d.N;
  const int L = d.n_levels;
  a.s0 = b.take(2 * BN * g.ch[0]);
  a.v0 = b.take(8 * BN * g.ch[0]);
  const bool tb = is_static(d, dec);                  // whole tiles of 64 nodes
  const bool sep = is_sep_fused(d, dec);
  for (int l = 0; l <= L; ++l) a.X[l] = b.take(tb ? tb_doubles(d, g.ch[l], g.Q[l]) : 2 * BN * g.ch[l] * g.Q[l]);
  a.pc = sep ? b.take(8 * BN) : nullptr;
  for (int l = 0; l < L; ++l) {
    a.U[l] = sep ? nullptr : b.take(tb ? tb_doubles(d, g.ch[l], 5 * g.Q[l]) : 10 * BN * g.ch[l] * g.Q[l]);
    a.tbl[l] = sep ? b.take((size_t)d.B * g.ch[l] * g.Q[l] * SEP_TBL_STRIDE) : nullptr;
    a.smix[l] = b.take(2 * BN * g.ch[l + 1]);
    a.wp[l] = tb ? b.take(local_static_packed_doubles(g.tab[l]->static_kind, g.ch[l], g.ch[l + 1])) : nullptr;
  }
  a.sL = b.take(2 * BN * g.ch[L]);
  a.vL = b.take(8 * BN * g.ch[L]);
  if (dec) a.pdec = b.take(8 * BN);
  else a.idx = reinterpret_cast<int*>(b.take(((size_t)d.B * 2 * (d.tau_s + d.tau_v) * 2 + 1) / 2 + 8));
  a.total = b.off;
  return a;
}

struct GenScratch {
  double *zero0, *g_p, *g_lat_s;    // zero block (first): zeros for a missing scalar gradient | decoder d p | encoder g_lat_s stand-in
  size_t zero_doubles;
  double *gs, *gv;                  // unpacked gradients at the two ends
  double *gX[2], *gU;
  double* gbuf;                     // encoder: pair-gradient scratch of the channel-outermost radial backward (generic_moments2.hip)
  double* gpk[4];                   // static path: reduced packed CatMix weight gradients per level
  double* gpb[4];                   // fused decoder levels: per-channel d p of each level [C_l][B N][8]
  double* tot[4];
  double* parts;
  size_t parts_size, total;
};
GenScratch carve_gen_scratch(const lgn_net_desc& d, bool dec, double* base) {
  GenScratch s{};
  Bump b{base};
  const GenGeom g = geom(d, dec);
  const size_t BN = (size_t)d.B * d.N;
  const int L = d.n_levels;
  size_t cq = 0, cmax = 0;
  for (int l = 0; l <= L; ++l) {
    cq = cq > (size_t)g.ch[l] * g.Q[l] ? cq : (size_t)g.ch[l] * g.Q[l];
    cmax = cmax > (size_t)g.ch[l] ? cmax : (size_t)g.ch[l];
  }
  {
    const size_t z0 = b.off;
    s.zero0 = b.take(2 * BN * cmax);
    s.g_p = b.take(dec ? 8 * BN : 0);
    s.g_lat_s = b.take(dec ? 0 : (size_t)2 * d.B * pool_blocks(d.latent_pool) * d.tau_s);
    s.zero_doubles = b.off - z0;
  }
  s.gs = b.take(2 * BN * cmax);
  s.gv = b.take(8 * BN * cmax);
  const bool tb = is_static(d, dec), sep = is_sep_fused(d, dec);
  const size_t tiles = (BN + 63) / 64;
  s.gX[0] = b.take(tb ? tiles * cq * 128 : 2 * BN * cq);
  s.gX[1] = b.take(tb ? tiles * cq * 128 : 2 * BN * cq);
  s.gU = sep ? nullptr : b.take(tb ? tiles * cq * 640 : 10 * BN * cq);
  for (int l = 0; l < L; ++l) {
    s.gpk[l] = tb ? b.take(local_static_packed_doubles(g.tab[l]->static_kind, g.ch[l], g.ch[l + 1])) : nullptr;
    s.gpb[l] = sep ? b.take((size_t)g.ch[l] * BN * 8) : nullptr;
  }
  s.gbuf = (!dec && d.N <= 32) ? b.take(moments2_gbuf_doubles(d.B, d.N, (int)cmax)) : nullptr;
  size_t psum = 0;
  for (int l = 0; l < L; ++l) {
    const size_t nrad = rad_partial_size(g.ch[l], dec);
    const size_t lrows = tb ? (sep ? (size_t)local_sep_part_rows(d.B) : tiles) * local_static_packed_doubles(g.tab[l]->static_kind, g.ch[l], g.ch[l + 1])
                            : (size_t)local_partial_rows((int)BN) * 2 * g.tab[l]->n_w;
    psum += ((lrows + 15) & ~size_t(15)) + (((size_t)d.B * nrad + 15) & ~size_t(15));
    psum += ((size_t)mlp_partial_rows((int)BN, d.mlp_hidden_mul * 2 * g.ch[l + 1]) * mlp_psize(g.ch[l + 1], d.mlp_hidden_mul * 2 * g.ch[l + 1], d.mlp_nlin) + 15) & ~size_t(15);
    s.tot[l] = b.take(nrad + 16);
  }
  const int Tin = d.tau_v_in > 0 ? d.tau_v_in : pool_blocks(d.latent_pool) * d.tau_v;
  if (dec) psum += (((size_t)d.B * 2 * g.ch[L] + 15) & ~size_t(15)) + (((size_t)d.B * (4 * g.ch[0] + 2 * (size_t)d.N * Tin) + 15) & ~size_t(15));
  else psum += (((size_t)d.B * 2 * (d.tau_s + d.tau_v) * pool_mix_in(d.latent_pool, d.N, g.ch[L]) + 15) & ~size_t(15)) + (((size_t)d.B * (2 * in_K(d) + 2) * g.ch[0] + 15) & ~size_t(15));
  s.parts = b.take(psum);
  s.parts_size = psum;
  s.total = b.off;
  return s;
}

GenArgs gen_level_args(const lgn_net_desc& d, bool dec, int l, const double* P, const int64_t* off, const double* X, const double* pos,
                       const uint8_t* mask) {
  const Slots S{d.n_levels, d.mlp_nlin};
  const GenGeom g = geom(d, dec);
  GenArgs a{};
  a.B = d.B; a.N = d.N; a.C = g.ch[l]; a.Q = g.Q[l]; a.X = X; a.p = pos; a.mask = mask;
  a.flags = d.flags;
  a.ra = P + off[S.rad(dec, l, 0)]; a.rb = P + off[S.rad(dec, l, 1)]; a.rc = P + off[S.rad(dec, l, 2)];
  a.w0 = P + off[S.rad(dec, l, 3)]; a.b0 = P + off[S.rad(dec, l, 4)]; a.w1 = P + off[S.rad(dec, l, 5)]; a.b1 = P + off[S.rad(dec, l, 6)];
  return a;
}

// X[0] (packed input features) must be in place; fills X[1..L], U, smix
int gen_levels_fwd(const lgn_net_desc& d, bool dec, const double* P, const int64_t* off, GenAct& a, const double* pos,
                   const uint8_t* mask, hipStream_t st) {
  const Slots S{d.n_levels, d.mlp_nlin};
  const GenGeom g = geom(d, dec);
  const int BN = d.B * d.N;
  const bool tb = is_static(d, dec);
  if (tb) {       // the packed CatMix weight images of all levels, one launch (the backward reuses them)
    StaticPackJob jobs[8];
    for (int l = 0; l < d.n_levels; ++l) {
      jobs[l] = StaticPackJob{g.tab[l]->static_kind, g.ch[l], g.ch[l + 1], {0, 0, 0, 0, 0}, P + off[S.mix(dec, l, 0)], a.wp[l]};
      for (int k = 0; k < 5; ++k) jobs[l].w0[k] = g.tab[l]->h_out_w0[k];
    }
    LGN_TRY(local_static_pack_batch(jobs, d.n_levels, false, st));
  }
  const bool sep = is_sep_fused(d, dec);
  for (int l = 0; l < d.n_levels; ++l) {
    GenArgs m = gen_level_args(d, dec, l, P, off, a.X[l], pos, mask);
    m.U = a.U[l];
    m.tb = tb;
    if (sep) {      // jet table instead of the moments tensor; the per-node kernel forms U where a term reads it
      LGN_TRY(dec_sep_tab(m, a.tbl[l], a.pc, st));
      LGN_TRY(local_fwd_sep(g.tab[l]->static_kind, d.B, d.N, g.ch[l], g.ch[l + 1], a.X[l], a.tbl[l], a.pc, a.wp[l], a.X[l + 1], a.smix[l],
                            g.qs[l + 1], st));
    } else if (tb) {
      LGN_TRY(moments_dispatch(m, dec, 0, st));
      int w0[8];
      for (int k = 0; k < 5; ++k) w0[k] = g.tab[l]->h_out_w0[k];
      LGN_TRY(local_fwd_static(g.tab[l]->static_kind, BN, g.ch[l], g.ch[l + 1], a.X[l], a.U[l], P + off[S.mix(dec, l, 0)], w0, a.wp[l],
                               a.X[l + 1], a.smix[l], g.qs[l + 1], st, /*packed=*/true));
    } else {
      LGN_TRY(moments_dispatch(m, dec, 0, st));
      LocalArgs la{};
      LGN_TRY(local_args(la, BN, g.ch[l], g.ch[l + 1], g.Q[l], g.Q[l + 1], g.tab[l]));
      la.X = a.X[l]; la.U = a.U[l]; la.wcat = P + off[S.mix(dec, l, 0)]; la.out = a.X[l + 1];
      la.s_copy = a.smix[l]; la.q_s = g.qs[l + 1];
      LGN_TRY(local_fwd(la, st));
    }
    MlpArgs<double> mm{};
    mm.M = BN; mm.C = g.ch[l + 1]; mm.H = d.mlp_hidden_mul * 2 * g.ch[l + 1]; mm.nlin = d.mlp_nlin; mm.act = d.activation; mm.flags = d.flags;
    for (int q = 0; q < d.mlp_nlin; ++q) { mm.w[q] = P + off[S.mlp(dec, l, 2 * q)]; mm.b[q] = P + off[S.mlp(dec, l, 2 * q + 1)]; }
    mm.s_in = a.smix[l];
    if (tb) { mm.s_out = a.X[l + 1] + (size_t)g.qs[l + 1] * 128; mm.tbQ = g.Q[l + 1]; }
    else { mm.s_out = a.X[l + 1] + g.qs[l + 1]; mm.ld = g.Q[l + 1]; }
    LGN_TRY(mlp_dispatch<double>(mm, false, st));
  }
  return 0;
}

// packed CatMix weight gradients (static path) are unpacked into the flat gradient AFTER the deferred reductions ran
typedef StaticPackJob UnpackJob;      // src = reduced packed gradients, dst = the CatMix slot of the flat gradient
int run_unpack_jobs(const std::vector<UnpackJob>& post, hipStream_t st) {
  if (post.empty()) return 0;
  return local_static_pack_batch(post.data(), (int)post.size(), true, st);
}

// On entry sc.gX[cur] holds the gradient w.r.t. X[L] (scalar column = 0 when !has_s_grad); on exit sc.gX[cur] the one w.r.t. X[0].
int gen_levels_bwd(const lgn_net_desc& d, bool dec, const double* P, double* G, const int64_t* off, const GenAct& a, const double* pos,
                   const uint8_t* mask, GenScratch& sc, Deferred& dq, RadFinJob& fin, std::vector<UnpackJob>& post, int& cur,
                   bool has_s_grad, hipStream_t st) {
  const Slots S{d.n_levels, d.mlp_nlin};
  const GenGeom g = geom(d, dec);
  const int BN = d.B * d.N;
  const bool tb = is_static(d, dec);
  for (int l = d.n_levels - 1; l >= 0; --l) {
    const int C = g.ch[l], CO = g.ch[l + 1];
    if (has_s_grad) {     // CGMLP backward, in place on the scalar column of the gradient
      MlpArgs<double> m{};
      m.M = BN; m.C = CO; m.H = d.mlp_hidden_mul * 2 * CO; m.nlin = d.mlp_nlin; m.act = d.activation; m.flags = d.flags;
      for (int q = 0; q < d.mlp_nlin; ++q) { m.w[q] = P + off[S.mlp(dec, l, 2 * q)]; m.b[q] = P + off[S.mlp(dec, l, 2 * q + 1)]; }
      m.s_in = a.smix[l];
      if (tb) { m.g_out = sc.gX[cur] + (size_t)g.qs[l + 1] * 128; m.g_in = sc.gX[cur] + (size_t)g.qs[l + 1] * 128; m.tbQ = g.Q[l + 1]; }
      else { m.g_out = sc.gX[cur] + g.qs[l + 1]; m.g_in = sc.gX[cur] + g.qs[l + 1]; m.ld = g.Q[l + 1]; }
      m.psize = mlp_psize(CO, m.H, m.nlin);
      DQ_TAKE(m.part, (size_t)mlp_partial_rows(BN, m.H) * m.psize);
      LGN_TRY(mlp_dispatch<double>(m, true, st));
      dq.add(m.part, mlp_partial_rows(BN, m.H), m.psize, 0, m.psize, G + off[S.mlp(dec, l, 0)]);
    }
    const int nxt = cur ^ 1;
    if (is_sep_fused(d, dec)) {      // per-node part + separable moments of the level in ONE launch; d p per channel, reduced after the stack
      int w0[8];
      for (int k = 0; k < 5; ++k) w0[k] = g.tab[l]->h_out_w0[k];
      const int kind = g.tab[l]->static_kind, np = (int)local_static_packed_doubles(kind, C, CO), rows = local_sep_part_rows(d.B);
      const int nrad = rad_partial_size(C, true);
      DQ_NEW(part, (size_t)rows * np);
      DQ_NEW(part_rad, (size_t)d.B * nrad);
      // (partial rows in the layout of the CatMix parameters themselves, np apart: the reduction writes the gradient, nothing to unpack)
      LGN_TRY(local_bwd_sep(kind, d.B, d.N, C, CO, a.X[l], a.tbl[l], a.pc, P + off[S.rad(dec, l, 4)], P + off[S.rad(dec, l, 6)], a.wp[l], w0,
                            sc.gX[cur], sc.gX[nxt], part, sc.gpb[l], part_rad, st));
      dq.add(part, rows, np, 0, 2 * g.tab[l]->n_w, G + off[S.mix(dec, l, 0)]);
      dq.add(part_rad, d.B, nrad, 0, C, G + off[S.rad(dec, l, 4)]);
      dq.add(part_rad, d.B, nrad, C, C, G + off[S.rad(dec, l, 6)]);
      cur = nxt;
      has_s_grad = true;
      continue;
    }
    if (tb) {       // compile-time-table kernel; its packed partial rows are reduced with everything else, then unpacked (post)
      int w0[8];
      for (int k = 0; k < 5; ++k) w0[k] = g.tab[l]->h_out_w0[k];
      const int kind = g.tab[l]->static_kind, np = (int)local_static_packed_doubles(kind, C, CO), tiles = (BN + 63) / 64;
      DQ_NEW(part, (size_t)tiles * np);
      LGN_TRY(local_bwd_static(kind, BN, C, CO, a.X[l], a.U[l], P + off[S.mix(dec, l, 0)], w0, a.wp[l], sc.gX[cur], sc.gU, sc.gX[nxt], part, st,
                               /*packed=*/true, /*param_layout=*/true));
      dq.add(part, tiles, np, 0, 2 * g.tab[l]->n_w, G + off[S.mix(dec, l, 0)]);
    } else {
      LocalArgs la{};
      LGN_TRY(local_args(la, BN, C, CO, g.Q[l], g.Q[l + 1], g.tab[l]));
      const int rows = local_partial_rows(BN), nw2 = 2 * g.tab[l]->n_w;
      la.X = a.X[l]; la.U = a.U[l]; la.wcat = P + off[S.mix(dec, l, 0)]; la.g_out = sc.gX[cur];
      la.gU = sc.gU; la.gX = sc.gX[nxt]; DQ_TAKE(la.part, (size_t)rows * nw2);
      LGN_TRY(local_bwd(la, st));
      dq.add(la.part, rows, nw2, 0, nw2, G + off[S.mix(dec, l, 0)]);
    }
    GenArgs m = gen_level_args(d, dec, l, P, off, a.X[l], pos, mask);
    m.tb = tb;
    const int nrad = rad_partial_size(C, dec);
    m.gU = sc.gU; m.gX = sc.gX[nxt]; m.g_p = dec ? sc.g_p : nullptr; DQ_TAKE(m.part_rad, (size_t)d.B * nrad);
    m.gbuf = sc.gbuf;
    LGN_TRY(moments_dispatch(m, dec, 1, st));
    LGN_TRY(moments_dispatch(m, dec, 2, st));
    if (dec) {
      dq.add(m.part_rad, d.B, nrad, 0, C, G + off[S.rad(dec, l, 4)]);
      dq.add(m.part_rad, d.B, nrad, C, C, G + off[S.rad(dec, l, 6)]);
    } else {
      dq.add(m.part_rad, d.B, nrad, 0, nrad, sc.tot[l]);
      fin.it[fin.n++] = RadFinJob::Item{sc.tot[l], C, m.ra, m.rb, m.rc, m.w0, m.w1, G + off[S.rad(dec, l, 0)], G + off[S.rad(dec, l, 1)],
                                        G + off[S.rad(dec, l, 2)], G + off[S.rad(dec, l, 3)], G + off[S.rad(dec, l, 4)],
                                        G + off[S.rad(dec, l, 5)], G + off[S.rad(dec, l, 6)]};
    }
    cur = nxt;
    has_s_grad = true;
  }
  return 0;
}

// packed <-> separate (0,0) / (1,1) features at the ends of the level stack, in the layout the network runs on
int net_pack(const lgn_net_desc& d, bool dec, int l, const double* s, const double* v, double* X, hipStream_t st) {
  const GenGeom g = geom(d, dec);
  if (is_static(d, dec)) return gen_pack_tb(d.B * d.N, g.ch[l], g.Q[l], g.qs[l], g.qv[l], s, v, X, st);
  return gen_pack((size_t)d.B * d.N * g.ch[l], g.Q[l], g.qs[l], g.qv[l], s, v, X, st);
}
int net_unpack(const lgn_net_desc& d, bool dec, int l, const double* X, double* s, double* v, hipStream_t st) {
  const GenGeom g = geom(d, dec);
  if (is_static(d, dec)) return gen_unpack_tb(d.B * d.N, g.ch[l], g.Q[l], g.qs[l], g.qv[l], X, s, v, st);
  return gen_unpack((size_t)d.B * d.N * g.ch[l], g.Q[l], g.qs[l], g.qv[l], X, s, v, st);
}

// decoder-side operands of junction_bwd (whole step)
struct JunctionBwd {
  int C0, Tin;
  const double *lat_v, *wg1, *w1, *pdec, *g_p, *g_s0, *g_v0;
  double* part_dec;
};

// ---- one table-driven network, forward / backward (shared by the per-network API and the whole step) ----------------
// with_latent = false (whole step): the latent stage runs in the junction kernel together with the decoder's input stage
int gen_encoder_fwd(const lgn_net_desc& d, const double* P, const int64_t* off, const double* p4, const uint8_t* mask, GenAct& a,
                    double* lat_s, double* lat_v, hipStream_t st, const double* xs = nullptr, bool with_latent = true) {
  const Slots S{d.n_levels, d.mlp_nlin};
  const GenGeom g = geom(d, false);
  const int L = d.n_levels;
  LGN_TRY(enc_input_fwd(d.B, d.N, g.ch[0], in_K(d), p4, xs, P + off[0], P + off[1], a.s0, a.v0, st));
  LGN_TRY(net_pack(d, false, 0, a.s0, a.v0, a.X[0], st));
  LGN_TRY(gen_levels_fwd(d, false, P, off, a, p4, mask, st));
  LGN_TRY(net_unpack(d, false, L, a.X[L], a.sL, a.vL, st));
  if (with_latent)
    LGN_TRY(enc_latent_fwd(d.B, d.N, g.ch[L], d.tau_s, d.tau_v, d.latent_pool, a.sL, a.vL, P + off[S.out0(false)], P + off[S.out0(false) + 1],
                           lat_s, lat_v, a.idx, st));
  return 0;
}

// the caller has zero-filled G and sc's zero block
// hand_dq / hand_fin (whole step): the pending reductions and radial finalisations are handed to the caller instead of being run here
int gen_encoder_bwd(const lgn_net_desc& d, const double* P, double* G, const int64_t* off, const double* p4, const uint8_t* mask,
                    const GenAct& a, const double* g_lat_s, const double* g_lat_v, GenScratch& sc, hipStream_t st,
                    const double* xs = nullptr, Deferred* hand_dq = nullptr, RadFinJob* hand_fin = nullptr, const JunctionBwd* jb = nullptr) {
  const Slots S{d.n_levels, d.mlp_nlin};
  const GenGeom g = geom(d, false);
  const int L = d.n_levels, B = d.B, N = d.N, Ts = d.tau_s, Tv = d.tau_v;
  Deferred dq;
  dq.parts = sc.parts;
  dq.cap = sc.parts_size;
  RadFinJob fin{};
  int cur = 0;
  {
    const int CL = g.ch[L], KL = pool_mix_in(d.latent_pool, N, CL), rowe = 2 * (Ts + Tv) * KL;
    DQ_NEW(parte, (size_t)B * rowe);
    if (jb)       // whole step: the decoder's input stage backward and this latent stage backward of a jet in one launch
      LGN_TRY(junction_bwd(B, N, jb->C0, jb->Tin, jb->lat_v, jb->wg1, jb->w1, jb->pdec, jb->g_p, jb->g_s0, jb->g_v0, const_cast<double*>(g_lat_v),
                           jb->part_dec, CL, Ts, Tv, d.latent_pool, a.sL, a.vL, P + off[S.out0(false)], P + off[S.out0(false) + 1],
                           g_lat_s ? g_lat_s : sc.g_lat_s, a.idx, sc.gs, sc.gv, parte, st));
    else
      LGN_TRY(enc_latent_bwd(B, N, CL, Ts, Tv, d.latent_pool, a.sL, a.vL, P + off[S.out0(false)], P + off[S.out0(false) + 1],
                             g_lat_s ? g_lat_s : sc.g_lat_s, g_lat_v, a.idx, sc.gs, sc.gv, parte, st));
    dq.add(parte, B, rowe, 0, 2 * Ts * KL, G + off[S.out0(false)]);
    dq.add(parte, B, rowe, 2 * Ts * KL, 2 * Tv * KL, G + off[S.out0(false) + 1]);
    LGN_TRY(net_pack(d, false, L, g_lat_s ? sc.gs : sc.zero0, sc.gv, sc.gX[cur], st));
  }
  std::vector<UnpackJob> post;
  LGN_TRY(gen_levels_bwd(d, false, P, G, off, a, p4, mask, sc, dq, fin, post, cur, g_lat_s != nullptr, st));
  {
    const int C0 = g.ch[0];
    LGN_TRY(net_unpack(d, false, 0, sc.gX[cur], sc.gs, sc.gv, st));
    const int K = in_K(d), row = (2 * K + 2) * C0;
    DQ_NEW(part, (size_t)B * row);
    LGN_TRY(enc_input_bwd(B, N, C0, K, p4, xs, sc.gs, sc.gv, part, st));
    dq.add(part, B, row, 0, 2 * C0 * K, G + off[0]);
    dq.add(part, B, row, 2 * C0 * K, 2 * C0, G + off[1]);
  }
  LGN_CHECK_ARG(dq.off <= dq.cap, "encoder_bwd: partial-row workspace overflow (%zu > %zu)", dq.off, dq.cap);
  if (hand_dq) {
    LGN_CHECK_ARG(post.empty() && hand_fin && hand_fin->n + fin.n <= (int)(sizeof(fin.it) / sizeof(fin.it[0])), "encoder_bwd: hand-over");
    hand_dq->segs.insert(hand_dq->segs.end(), dq.segs.begin(), dq.segs.end());
    for (int i = 0; i < fin.n; ++i) hand_fin->it[hand_fin->n++] = fin.it[i];
    return 0;
  }
  LGN_TRY(dq.flush(st));
  LGN_TRY(run_unpack_jobs(post, st));
  LGN_TRY(rad_finalize_batch(fin, st));
  return 0;
}

// forward up to the unpacked last-level features (the caller applies dec_output_fwd or the fused output + loss kernel)
int gen_decoder_fwd(const lgn_net_desc& d, const double* P, const int64_t* off, const double* lat_v, GenAct& a, hipStream_t st,
                    bool with_input = true) {
  const GenGeom g = geom(d, true);
  const int L = d.n_levels, Tin = d.tau_v_in > 0 ? d.tau_v_in : pool_blocks(d.latent_pool) * d.tau_v;
  if (with_input) LGN_TRY(dec_input_fwd(d.B, d.N, g.ch[0], Tin, lat_v, P + off[1], P + off[2], P + off[3], a.pdec, a.s0, a.v0, st));
  LGN_TRY(net_pack(d, true, 0, a.s0, a.v0, a.X[0], st));
  LGN_TRY(gen_levels_fwd(d, true, P, off, a, a.pdec, nullptr, st));
  LGN_TRY(net_unpack(d, true, L, a.X[L], a.sL, a.vL, st));
  return 0;
}

// sc.gv holds the gradient w.r.t. the last level's (1,1) features (from dec_output_bwd / dec_output_loss); dq carries the
// caller's pending reductions and is flushed by the caller
// part_in (whole step): the input stage's backward is the caller's (junction kernel) -- its partial rows are allocated and their
// reductions registered here, *part_in tells the caller where they go
int gen_decoder_bwd(const lgn_net_desc& d, const double* P, double* G, const int64_t* off, const double* lat_v, const GenAct& a,
                    double* g_lat_v, GenScratch& sc, Deferred& dq, RadFinJob& fin, std::vector<UnpackJob>& post, hipStream_t st,
                    double** part_in = nullptr) {
  const GenGeom g = geom(d, true);
  const int L = d.n_levels, B = d.B, N = d.N, Tin = d.tau_v_in > 0 ? d.tau_v_in : pool_blocks(d.latent_pool) * d.tau_v;
  int cur = 0;
  LGN_TRY(net_pack(d, true, L, sc.zero0, sc.gv, sc.gX[cur], st));
  LGN_TRY(gen_levels_bwd(d, true, P, G, off, a, a.pdec, nullptr, sc, dq, fin, post, cur, /*has_s_grad=*/false, st));
  if (is_sep_fused(d, true)) {     // d p of the levels: per-channel parts -> sc.g_p (zero on entry), in level and channel order
    const double* gpb[4];
    int cl[4];
    for (int l = 0; l < L; ++l) { gpb[l] = sc.gpb[L - 1 - l]; cl[l] = g.ch[L - 1 - l]; }      // (the order the levels ran in)
    LGN_TRY(local_sep_gp_reduce(gpb, cl, L, B * N, sc.g_p, st));
  }
  const int C0 = g.ch[0], row = 4 * C0 + 2 * N * Tin;
  LGN_TRY(net_unpack(d, true, 0, sc.gX[cur], sc.gs, sc.gv, st));
  DQ_NEW(part, (size_t)B * row);
  if (part_in) *part_in = part;
  else LGN_TRY(dec_input_bwd(B, N, C0, Tin, lat_v, P + off[1], P + off[3], a.pdec, sc.g_p, sc.gs, sc.gv, g_lat_v, part, st));
  dq.add(part, B, row, 0, 2 * C0, G + off[2]);
  dq.add(part, B, row, 2 * C0, 2 * C0, G + off[3]);
  dq.add(part, B, row, 4 * C0, 2 * N * Tin, G + off[1]);
  return 0;
}

// ---- whole training step on table-driven networks -------------------------------------------------------------------
struct GenStep {
  GenAct ea, da;
  GenScratch es, ds;
  double *lat_s, *lat_v, *g_lat_v;
  size_t total;
};
GenStep carve_gen_step(const lgn_net_desc& d, double* base) {
  GenStep g{};
  auto at = [&](size_t off) { return base ? base + off : nullptr; };
  size_t off = 0;
  g.ea = carve_gen_act(d, false, at(off)); off += (g.ea.total + 15) & ~size_t(15);
  g.da = carve_gen_act(d, true, at(off)); off += (g.da.total + 15) & ~size_t(15);
  g.es = carve_gen_scratch(d, false, at(off)); off += (g.es.total + 15) & ~size_t(15);
  g.ds = carve_gen_scratch(d, true, at(off)); off += (g.ds.total + 15) & ~size_t(15);
  const int Tin = d.tau_v_in > 0 ? d.tau_v_in : pool_blocks(d.latent_pool) * d.tau_v;
  g.lat_s = at(off); off += ((size_t)2 * d.B * pool_blocks(d.latent_pool) * d.tau_s + 15) & ~size_t(15);
  g.lat_v = at(off); off += ((size_t)2 * d.B * Tin * 4 + 15) & ~size_t(15);
  g.g_lat_v = at(off); off += ((size_t)2 * d.B * Tin * 4 + 15) & ~size_t(15);
  g.total = off;
  return g;
}

int gen_step_fwd_bwd(const lgn_net_desc& d, const double* params, double* grads, long long n_params, const int64_t* enc_off,
                     const int64_t* dec_off, const double* p4, const double* target, const uint8_t* mask, double* workspace,
                     long long workspace_doubles, double* recon, double* loss_part, hipStream_t st, const StepTailArgs* tail) {
  LGN_CHECK_ARG(is_generic(d, false) && is_generic(d, true), "step: encoder and decoder must both be table-driven (or both fused)");
  if (int rc = check_generic(d, false)) return rc;
  if (int rc = check_generic(d, true)) return rc;
  LGN_CHECK_ARG(d.tau_v_in == 0 || d.tau_v_in == pool_blocks(d.latent_pool) * d.tau_v,
                "step: the decoder must consume the encoder's %d pooled latent vectors", pool_blocks(d.latent_pool) * d.tau_v);
  GenStep g = carve_gen_step(d, workspace);
  LGN_CHECK_ARG((long long)g.total <= workspace_doubles, "step: workspace holds %lld doubles, this configuration needs %zu",
                workspace_doubles, g.total);
  if (int rc = check_mlp_contiguous(d, false, enc_off)) return rc;
  if (int rc = check_mlp_contiguous(d, true, dec_off)) return rc;
  const Slots S{d.n_levels, d.mlp_nlin};
  const int L = d.n_levels, B = d.B, N = d.N, CL = d.dec_channels[L];
  LGN_TRY(zero_ranges(grads, (size_t)n_params, g.es.zero0, g.es.zero_doubles, g.ds.zero0, g.ds.zero_doubles, st));
  // encoder latent stage + decoder input stage of a jet: ONE launch (the junction kernels of the maxdim-2 step), forward and backward
  const int Tin = pool_blocks(d.latent_pool) * d.tau_v, CLe = d.enc_channels[d.n_levels], C0d = d.dec_channels[0];
  LGN_TRY(gen_encoder_fwd(d, params, enc_off, p4, mask, g.ea, g.lat_s, g.lat_v, st, nullptr, /*with_latent=*/false));
  LGN_TRY(junction_fwd(B, N, CLe, d.tau_s, d.tau_v, d.latent_pool, g.ea.sL, g.ea.vL, params + enc_off[S.out0(false)],
                       params + enc_off[S.out0(false) + 1], g.lat_s, g.lat_v, g.ea.idx, C0d, params + dec_off[1], params + dec_off[2],
                       params + dec_off[3], g.da.pdec, g.da.s0, g.da.v0, st));
  LGN_TRY(gen_decoder_fwd(d, params, dec_off, g.lat_v, g.da, st, /*with_input=*/false));
  Deferred dq;
  dq.parts = g.ds.parts;
  dq.cap = g.ds.parts_size;
  RadFinJob fin{};
  {
    DQ_NEW(part, (size_t)B * 2 * CL);
    LGN_TRY(dec_output_loss(B, N, CL, g.da.vL, params + dec_off[S.out0(true) + 1], target, 1.0, recon, loss_part, g.ds.gv, part, st));
    dq.add(part, B, 2 * CL, 0, 2 * CL, grads + dec_off[S.out0(true) + 1]);
  }
  std::vector<UnpackJob> post;
  double* part_in = nullptr;
  LGN_TRY(gen_decoder_bwd(d, params, grads, dec_off, g.lat_v, g.da, g.g_lat_v, g.ds, dq, fin, post, st, &part_in));
  LGN_CHECK_ARG(dq.off <= dq.cap, "step: partial-row workspace overflow (%zu > %zu)", dq.off, dq.cap);
  const JunctionBwd jb{C0d, Tin, g.lat_v, params + dec_off[1], params + dec_off[3], g.da.pdec, g.ds.g_p, g.ds.gs, g.ds.gv, part_in};
  // Round 6: the static levels leave their CatMix partial rows in PARAMETER layout (nothing to unpack after the reduction), so the
  // two networks' reductions wait for ONE launch at the end of the step -- with `tail` (single process) the fused tail of the
  // maxdim-2 step (step_tail.hip: reductions + radial finalisation + L1 + Adam + loss), else reduce_segments + rad_finalize_batch.
  // Run-time-table levels (LGN_NET_NO_STATIC) keep packed rows and the round-5 sequence.
  if (!post.empty() || !is_static(d, false) || !is_static(d, true)) {
    // (the decoder's reductions run now: its input stage's backward must have written its partial rows -- no junction kernel here)
    LGN_TRY(dec_input_bwd(B, N, C0d, Tin, g.lat_v, params + dec_off[1], params + dec_off[3], g.da.pdec, g.ds.g_p, g.ds.gs, g.ds.gv, g.g_lat_v,
                          part_in, st));
    LGN_TRY(dq.flush(st));
    LGN_TRY(run_unpack_jobs(post, st));
    // the decoder never reads the latent scalars: no gradient on them
    LGN_TRY(gen_encoder_bwd(d, params, grads, enc_off, p4, mask, g.ea, nullptr, g.g_lat_v, g.es, st));
    if (tail)
      LGN_TRY(finalize_step(tail->w, tail->g, tail->n, tail->loss_part, tail->nB, tail->lambda, tail->m, tail->v, tail->step_dev, tail->lr,
                            tail->beta1, tail->beta2, tail->eps, tail->do_adam, tail->loss_out, st));
    return 0;
  }
  LGN_TRY(gen_encoder_bwd(d, params, grads, enc_off, p4, mask, g.ea, nullptr, g.g_lat_v, g.es, st, nullptr, &dq, &fin, &jb));
  if (tail && !(d.flags & LGN_NET_SPLIT_TAIL)) {
    const int rc = step_tail(dq.segs, fin, *tail, st);
    if (rc == 0) return 0;
    if (rc != -2) return rc;
  }
  LGN_TRY(dq.flush(st));
  LGN_TRY(rad_finalize_batch(fin, st));
  if (tail)
    LGN_TRY(finalize_step(tail->w, tail->g, tail->n, tail->loss_part, tail->nB, tail->lambda, tail->m, tail->v, tail->step_dev, tail->lr,
                          tail->beta1, tail->beta2, tail->eps, tail->do_adam, tail->loss_out, st));
  return 0;
}

}  // namespace
}  // namespace lgn

// ---------------------------------------------------------------------------------------------------------
// one network at a time (module API: LGNEncoder.forward / LGNDecoder.forward and their autograd backward)
// ---------------------------------------------------------------------------------------------------------
namespace lgn {
namespace {

struct NetAct {                     // written by *_fwd, read by *_bwd
  NetBuf n;
  double* pdec;                     // decoder: complex canonical positions [2][B][N][4]
  int* idx;                         // encoder: pooling indices
  size_t total;
};
NetAct carve_act(const lgn_net_desc& d, bool dec, double* base) {
  NetAct a{};
  Bump b{base};
  const size_t BN = (size_t)d.B * d.N;
  const int* ch = dec ? d.dec_channels : d.enc_channels;
  for (int l = 0; l <= d.n_levels; ++l) {
    a.n.s[l] = b.take(2 * BN * ch[l]);
    a.n.v[l] = b.take(8 * BN * ch[l]);
  }
  for (int l = 0; l < d.n_levels; ++l) {
    a.n.smix[l] = b.take(2 * BN * ch[l + 1]);
    a.n.ag0[l] = b.take(4 * BN * ch[l]);
    a.n.ag1[l] = b.take(16 * BN * ch[l]);
    // (the caller's upstream gradient may reach every level)
    const size_t hs = BN <= mlp_save_max_rows()
                          ? mlp_saved_doubles((int)BN, d.mlp_hidden_mul * 2 * ch[l + 1], d.mlp_nlin) : 0;
    a.n.hsave[l] = hs ? b.take(hs) : nullptr;
  }
  if (dec) a.pdec = b.take(8 * BN);
  else a.idx = reinterpret_cast<int*>(b.take(((size_t)d.B * 2 * (d.tau_s + d.tau_v) * 2 + 1) / 2 + 8));
  a.total = b.off;
  return a;
}

struct NetScratch {                 // backward only
  Work w;                           // gs, gv, gsmix, g_ag, zeros_s, g_p, g_lat_s (zero block), tot, parts
  size_t total;
};
NetScratch carve_scratch(const lgn_net_desc& d, bool dec, double* base) {
  NetScratch s{};
  Work& w = s.w;
  Bump b{base};
  const size_t BN = (size_t)d.B * d.N;
  const int L = d.n_levels, Ts = d.tau_s, Tv = d.tau_v;
  const int* ch = dec ? d.dec_channels : d.enc_channels;
  int cmax = 0;
  for (int l = 0; l <= L; ++l) cmax = cmax > ch[l] ? cmax : ch[l];
  {  // zero-initialised block FIRST: a caller that places `grads` right in front of the scratch gets one memset for both
    const size_t z0 = b.off;
    w.tail_cnt = b.take(8);
    w.zeros_s = b.take(2 * BN * cmax);
    w.g_p = b.take(dec ? 8 * BN : 0);
    w.g_lat_s = b.take(dec ? 0 : (size_t)2 * d.B * pool_blocks(d.latent_pool) * Ts);
    w.zero_doubles = b.off - z0;
  }
  for (int q = 0; q < 2; ++q) {
    w.gs[q] = b.take(2 * BN * cmax);
    w.gv[q] = b.take(8 * BN * cmax);
  }
  w.gsmix = b.take(2 * BN * cmax);
  w.g_ag = b.take(20 * BN * cmax);
  size_t psum = 0;
  for (int l = 0; l < L; ++l) {
    int rm, rr;
    level_bwd_partial_rows(d.B, d.N, dec, d.flags, &rm, &rr);
    const size_t nmix = (size_t)4 * ch[l + 1] * 5 * ch[l], nrad = rad_partial_size(ch[l], dec);
    psum += ((rm * nmix + 15) & ~size_t(15)) + ((rr * nrad + 15) & ~size_t(15));
    psum += (mlp_part_rows(d, dec, l) * mlp_psize(ch[l + 1], d.mlp_hidden_mul * 2 * ch[l + 1], d.mlp_nlin) + 15) & ~size_t(15);
    w.tot[dec ? 1 : 0][l] = b.take(nrad + 16);
  }
  // input / output ends: decoder  B x (2 C_L) + B x (4 C_0 + 2 N Tin);  encoder  B x 2 (Ts + Tv) C_L + B x 4 C_0
  const int Tin = d.tau_v_in > 0 ? d.tau_v_in : pool_blocks(d.latent_pool) * Tv;
  if (dec) psum += (((size_t)d.B * 2 * ch[L] + 15) & ~size_t(15)) + (((size_t)d.B * (4 * ch[0] + 2 * (size_t)d.N * Tin) + 15) & ~size_t(15));
  else {
    int rm0, rr0;                                          // input-stage partial rows: one per workgroup of the first level's backward
    level_bwd_partial_rows(d.B, d.N, 0, d.flags, &rm0, &rr0);
    if (rm0 < d.B) rm0 = d.B;
    psum += (((size_t)d.B * 2 * (Ts + Tv) * pool_mix_in(d.latent_pool, d.N, ch[L]) + 15) & ~size_t(15)) + (((size_t)rm0 * (2 * in_K(d) + 2) * ch[0] + 15) & ~size_t(15));
  }
  w.parts = b.take(psum);
  w.parts_size = psum;
  s.total = b.off;
  return s;
}

}  // namespace
}  // namespace lgn

using namespace lgn;

extern "C" {

long long lgn_net_workspace_doubles(const lgn_net_desc* d, int decoder, int which) {
  if (check_desc(d)) return -1;
  if (is_generic(*d, decoder != 0)) {
    if (check_generic(*d, decoder != 0)) return -1;
    return which == 0 ? (long long)carve_gen_act(*d, decoder != 0, nullptr).total : (long long)carve_gen_scratch(*d, decoder != 0, nullptr).total;
  }
  return which == 0 ? (long long)carve_act(*d, decoder != 0, nullptr).total : (long long)carve_scratch(*d, decoder != 0, nullptr).total;
}

int lgn_encoder_fwd_f64(const lgn_net_desc* d, const double* params, const int64_t* off, const double* p4, const uint8_t* mask,
                        const double* in_scalars, double* act, long long act_doubles, double* lat_s, double* lat_v, void* stream) {
  if (int rc = check_desc(d)) return rc;
  LGN_CHECK_ARG(params && off && p4 && mask && act && lat_s && lat_v, "encoder_fwd: null pointer");
  LGN_CHECK_ARG(d->n_in_scalars <= 1 || in_scalars, "encoder_fwd: %d input scalars per node, but in_scalars is NULL", d->n_in_scalars);
  if (is_generic(*d, false)) {
    if (int rc = check_generic(*d, false)) return rc;
    GenAct ga = carve_gen_act(*d, false, act);
    LGN_CHECK_ARG((long long)ga.total <= act_doubles, "encoder_fwd: activation buffer holds %lld doubles, needs %zu", act_doubles, ga.total);
    return gen_encoder_fwd(*d, params, off, p4, mask, ga, lat_s, lat_v, (hipStream_t)stream, in_scalars);
  }
  NetAct a = carve_act(*d, false, act);
  LGN_CHECK_ARG((long long)a.total <= act_doubles, "encoder_fwd: activation buffer holds %lld doubles, needs %zu", act_doubles, a.total);
  hipStream_t st = (hipStream_t)stream;
  const Slots S{d->n_levels, d->mlp_nlin};
  const int L = d->n_levels;
  const int* ce = d->enc_channels;
  if (in_K(*d) > 1) {        // several input scalars: the input stage is its own launch
    LGN_TRY(enc_input_fwd(d->B, d->N, ce[0], in_K(*d), p4, in_scalars, params + off[0], params + off[1], a.n.s[0], a.n.v[0], st));
    LGN_TRY(levels_fwd(*d, false, ce, params, off, a.n, p4, mask, st));
  } else {
    const InputStage in0{params + off[0], params + off[1]};               // (rides on the first level's kernel)
    LGN_TRY(levels_fwd(*d, false, ce, params, off, a.n, p4, mask, st, &in0));
  }
  LGN_TRY(enc_latent_fwd(d->B, d->N, ce[L], d->tau_s, d->tau_v, d->latent_pool, a.n.s[L], a.n.v[L], params + off[S.out0(false)],
                         params + off[S.out0(false) + 1], lat_s, lat_v, a.idx, st));
  return 0;
}

int lgn_encoder_bwd_f64(const lgn_net_desc* d, const double* params, double* grads, long long n_params, const int64_t* off,
                        const double* p4, const uint8_t* mask, const double* in_scalars, const double* act, long long act_doubles,
                        const double* g_lat_s, const double* g_lat_v, double* scratch, long long scratch_doubles, void* stream) {
  if (int rc = check_desc(d)) return rc;
  LGN_CHECK_ARG(params && grads && off && p4 && mask && act && g_lat_v && scratch && n_params > 0, "encoder_bwd: null pointer");
  LGN_CHECK_ARG(d->n_in_scalars <= 1 || in_scalars, "encoder_bwd: %d input scalars per node, but in_scalars is NULL", d->n_in_scalars);
  if (is_generic(*d, false)) {
    if (int rc = check_generic(*d, false)) return rc;
    GenAct ga = carve_gen_act(*d, false, const_cast<double*>(act));
    GenScratch gs = carve_gen_scratch(*d, false, scratch);
    LGN_CHECK_ARG((long long)ga.total <= act_doubles && (long long)gs.total <= scratch_doubles,
                  "encoder_bwd: buffers hold %lld / %lld doubles, need %zu / %zu", act_doubles, scratch_doubles, ga.total, gs.total);
    if (int rc = check_mlp_contiguous(*d, false, off)) return rc;
    LGN_TRY(zero_grads_and_block(grads, (size_t)n_params, gs.zero0, gs.zero_doubles, (hipStream_t)stream));
    return gen_encoder_bwd(*d, params, grads, off, p4, mask, ga, g_lat_s, g_lat_v, gs, (hipStream_t)stream, in_scalars);
  }
  NetAct a = carve_act(*d, false, const_cast<double*>(act));
  NetScratch sc = carve_scratch(*d, false, scratch);
  LGN_CHECK_ARG((long long)a.total <= act_doubles && (long long)sc.total <= scratch_doubles,
                "encoder_bwd: buffers hold %lld / %lld doubles, need %zu / %zu", act_doubles, scratch_doubles, a.total, sc.total);
  if (int rc = check_mlp_contiguous(*d, false, off)) return rc;
  hipStream_t st = (hipStream_t)stream;
  Work& w = sc.w;
  const Slots S{d->n_levels, d->mlp_nlin};
  const int L = d->n_levels, B = d->B, N = d->N, Ts = d->tau_s, Tv = d->tau_v;
  const int* ce = d->enc_channels;
  LGN_TRY(zero_grads_and_block(grads, (size_t)n_params, w.zero0(), w.zero_doubles, st));
  Deferred dq;
  dq.parts = w.parts;
  dq.cap = w.parts_size;
  RadFinJob fin{};
  int cur = 0;
  {
    const int CL = ce[L], KL = pool_mix_in(d->latent_pool, N, CL), rowe = 2 * (Ts + Tv) * KL;
    DQ_NEW(parte, (size_t)B * rowe);
    LGN_TRY(enc_latent_bwd(B, N, CL, Ts, Tv, d->latent_pool, a.n.s[L], a.n.v[L], params + off[S.out0(false)], params + off[S.out0(false) + 1],
                           g_lat_s ? g_lat_s : w.g_lat_s, g_lat_v, a.idx, w.gs[cur], w.gv[cur], parte, st));
    dq.add(parte, B, rowe, 0, 2 * Ts * KL, grads + off[S.out0(false)]);
    dq.add(parte, B, rowe, 2 * Ts * KL, 2 * Tv * KL, grads + off[S.out0(false) + 1]);
  }
  // without an upstream gradient on the latent scalars the last level's scalars (and its CGMLP) receive none
  double* const in0_grads[2] = {grads + off[0], grads + off[1]};
  bool in0_done = false;
  const int K = in_K(*d);      // (several input scalars: the input stage's backward is its own launch, nothing rides)
  LGN_TRY(levels_bwd(*d, false, ce, params, grads, off, a.n, p4, mask, w, dq, fin, cur, /*has_s_grad=*/g_lat_s != nullptr, st,
                     K > 1 ? nullptr : in0_grads, &in0_done));
  if (!in0_done) {
    const int C0 = ce[0], row = (2 * K + 2) * C0;
    DQ_NEW(part, (size_t)B * row);
    LGN_TRY(enc_input_bwd(B, N, C0, K, p4, in_scalars, w.gs[cur], w.gv[cur], part, st));
    dq.add(part, B, row, 0, 2 * C0 * K, grads + off[0]);
    dq.add(part, B, row, 2 * C0 * K, 2 * C0, grads + off[1]);
  }
  LGN_CHECK_ARG(dq.off <= dq.cap, "encoder_bwd: partial-row workspace overflow (%zu > %zu)", dq.off, dq.cap);
  // (measured at cfg2, captured module step: the one-launch form is 3 us SLOWER for the encoder alone -- 0.5966 against 0.5937 ms)
  LGN_TRY(finish_reductions(dq, fin, grads, n_params, nullptr, d->flags, st));
  return 0;
}

int lgn_decoder_fwd_f64(const lgn_net_desc* d, const double* params, const int64_t* off, const double* lat_v, double* act,
                        long long act_doubles, double* recon, void* stream) {
  if (int rc = check_desc(d)) return rc;
  LGN_CHECK_ARG(params && off && lat_v && act && recon, "decoder_fwd: null pointer");
  if (is_generic(*d, true)) {
    if (int rc = check_generic(*d, true)) return rc;
    GenAct ga = carve_gen_act(*d, true, act);
    LGN_CHECK_ARG((long long)ga.total <= act_doubles, "decoder_fwd: activation buffer holds %lld doubles, needs %zu", act_doubles, ga.total);
    LGN_TRY(gen_decoder_fwd(*d, params, off, lat_v, ga, (hipStream_t)stream));
    const Slots Sg{d->n_levels, d->mlp_nlin};
    return dec_output_fwd(d->B, d->N, d->dec_channels[d->n_levels], ga.vL, params + off[Sg.out0(true) + 1], recon, (hipStream_t)stream);
  }
  NetAct a = carve_act(*d, true, act);
  LGN_CHECK_ARG((long long)a.total <= act_doubles, "decoder_fwd: activation buffer holds %lld doubles, needs %zu", act_doubles, a.total);
  hipStream_t st = (hipStream_t)stream;
  const Slots S{d->n_levels, d->mlp_nlin};
  const int L = d->n_levels, Tin = d->tau_v_in > 0 ? d->tau_v_in : pool_blocks(d->latent_pool) * d->tau_v;
  const int* cd = d->dec_channels;
  LGN_TRY(dec_input_fwd(d->B, d->N, cd[0], Tin, lat_v, params + off[1], params + off[2], params + off[3], a.pdec, a.n.s[0], a.n.v[0], st));
  LGN_TRY(levels_fwd(*d, true, cd, params, off, a.n, a.pdec, nullptr, st));
  LGN_TRY(dec_output_fwd(d->B, d->N, cd[L], a.n.v[L], params + off[S.out0(true) + 1], recon, st));
  return 0;
}

int lgn_decoder_bwd_f64(const lgn_net_desc* d, const double* params, double* grads, long long n_params, const int64_t* off,
                        const double* lat_v, const double* act, long long act_doubles, const double* g_recon, double* g_lat_v,
                        double* scratch, long long scratch_doubles, void* stream) {
  if (int rc = check_desc(d)) return rc;
  LGN_CHECK_ARG(params && grads && off && lat_v && act && g_recon && g_lat_v && scratch && n_params > 0, "decoder_bwd: null pointer");
  if (is_generic(*d, true)) {
    if (int rc = check_generic(*d, true)) return rc;
    hipStream_t gst = (hipStream_t)stream;
    GenAct ga = carve_gen_act(*d, true, const_cast<double*>(act));
    GenScratch gs = carve_gen_scratch(*d, true, scratch);
    LGN_CHECK_ARG((long long)ga.total <= act_doubles && (long long)gs.total <= scratch_doubles,
                  "decoder_bwd: buffers hold %lld / %lld doubles, need %zu / %zu", act_doubles, scratch_doubles, ga.total, gs.total);
    if (int rc = check_mlp_contiguous(*d, true, off)) return rc;
    LGN_TRY(zero_grads_and_block(grads, (size_t)n_params, gs.zero0, gs.zero_doubles, gst));
    const Slots Sg{d->n_levels, d->mlp_nlin};
    const int CL = d->dec_channels[d->n_levels];
    Deferred dq;
    dq.parts = gs.parts;
    dq.cap = gs.parts_size;
    RadFinJob fin{};
    DQ_NEW(part, (size_t)d->B * 2 * CL);
    LGN_TRY(dec_output_bwd(d->B, d->N, CL, ga.vL, params + off[Sg.out0(true) + 1], g_recon, gs.gv, part, gst));
    dq.add(part, d->B, 2 * CL, 0, 2 * CL, grads + off[Sg.out0(true) + 1]);
    std::vector<UnpackJob> post;
    LGN_TRY(gen_decoder_bwd(*d, params, grads, off, lat_v, ga, g_lat_v, gs, dq, fin, post, gst));
    LGN_CHECK_ARG(dq.off <= dq.cap, "decoder_bwd: partial-row workspace overflow (%zu > %zu)", dq.off, dq.cap);
    LGN_TRY(dq.flush(gst));
    return run_unpack_jobs(post, gst);
  }
  NetAct a = carve_act(*d, true, const_cast<double*>(act));
  NetScratch sc = carve_scratch(*d, true, scratch);
  LGN_CHECK_ARG((long long)a.total <= act_doubles && (long long)sc.total <= scratch_doubles,
                "decoder_bwd: buffers hold %lld / %lld doubles, need %zu / %zu", act_doubles, scratch_doubles, a.total, sc.total);
  if (int rc = check_mlp_contiguous(*d, true, off)) return rc;
  hipStream_t st = (hipStream_t)stream;
  Work& w = sc.w;
  const Slots S{d->n_levels, d->mlp_nlin};
  const int L = d->n_levels, B = d->B, N = d->N, Tin = d->tau_v_in > 0 ? d->tau_v_in : pool_blocks(d->latent_pool) * d->tau_v;
  const int* cd = d->dec_channels;
  LGN_TRY(zero_grads_and_block(grads, (size_t)n_params, w.zero0(), w.zero_doubles, st));
  Deferred dq;
  dq.parts = w.parts;
  dq.cap = w.parts_size;
  RadFinJob fin{};
  int cur = 0;
  {
    DQ_NEW(part, (size_t)B * 2 * cd[L]);
    LGN_TRY(dec_output_bwd(B, N, cd[L], a.n.v[L], params + off[S.out0(true) + 1], g_recon, w.gv[cur], part, st));
    dq.add(part, B, 2 * cd[L], 0, 2 * cd[L], grads + off[S.out0(true) + 1]);
  }
  LGN_TRY(levels_bwd(*d, true, cd, params, grads, off, a.n, a.pdec, nullptr, w, dq, fin, cur, /*has_s_grad=*/false, st));
  {
    const int C0 = cd[0], row = 4 * C0 + 2 * N * Tin;
    DQ_NEW(part, (size_t)B * row);
    LGN_TRY(dec_input_bwd(B, N, C0, Tin, lat_v, params + off[1], params + off[3], a.pdec, w.g_p, w.gs[cur], w.gv[cur], g_lat_v, part, st));
    dq.add(part, B, row, 0, 2 * C0, grads + off[2]);
    dq.add(part, B, row, 2 * C0, 2 * C0, grads + off[3]);
    dq.add(part, B, row, 4 * C0, 2 * N * Tin, grads + off[1]);
  }
  LGN_CHECK_ARG(dq.off <= dq.cap, "decoder_bwd: partial-row workspace overflow (%zu > %zu)", dq.off, dq.cap);
  LGN_TRY(dq.flush(st));
  LGN_TRY(rad_finalize_batch(fin, st));
  return 0;
}

}  // extern "C"

extern "C" {

int lgn_step_param_slots(const lgn_net_desc* d, int decoder) {
  if (check_desc(d)) return -1;
  return Slots{d->n_levels, d->mlp_nlin}.count(decoder != 0);
}

long long lgn_encoder_end_lds_bytes(int N, int C0, int K, int CL, int Ts, int Tv, int pool) {
  return pool_valid(pool) ? (long long)encoder_end_lds_bytes(N, C0, K < 1 ? 1 : K, CL, Ts, Tv, pool) : -1;
}
long long lgn_decoder_end_lds_bytes(int N, int C0, int Tin, int CL) { return (long long)decoder_end_lds_bytes(N, C0, Tin, CL); }
long long lgn_junction_lds_bytes(int N, int CL, int Ts, int Tv, int pool, int C0) {
  return pool_valid(pool) ? (long long)junction_lds_bytes(N, CL, Ts, Tv, pool, C0) : -1;
}

long long lgn_step_workspace_doubles(const lgn_net_desc* d) {
  if (check_desc(d)) return -1;
  if (is_generic(*d, false) || is_generic(*d, true)) {
    if (check_generic(*d, false) || check_generic(*d, true)) return -1;
    return (long long)carve_gen_step(*d, nullptr).total;
  }
  return (long long)carve(*d, nullptr).total;
}

}  // extern "C"
// The step of networks that do not share their nodes or take more input scalars than the mass (jet_features: the encoder works on
// N + 1 nodes, lgn/models/lgn_encoder.py:372-411; data['scalars']): the same level stacks, the four end stages as launches of their
// own (no junction kernels, no riding input stage) -- every one of them takes its network's node count.
static int step_fwd_bwd_split(const lgn_net_desc& d, Work& w, const double* params, double* grads, long long n_params, const int64_t* enc_off,
                              const int64_t* dec_off, const double* p4, const double* target, const uint8_t* mask, const double* in_scalars,
                              double* recon, double* loss_part, hipStream_t st, const StepTailArgs* tail) {
  if (int rc = check_mlp_contiguous(d, false, enc_off)) return rc;
  if (int rc = check_mlp_contiguous(d, true, dec_off)) return rc;
  lgn_net_desc de = d, dd = d;                   // per-network views of the descriptor: the level sequencers read B, N, flags, MLP shape
  dd.N = dec_nodes(d);
  const Slots S{d.n_levels, d.mlp_nlin};
  const int L = d.n_levels, B = d.B, Ne = d.N, Nd = dd.N, Ts = d.tau_s, Tv = d.tau_v, K = in_K(d);
  const int* ce = d.enc_channels;
  const int* cd = d.dec_channels;
  const int Tin = pool_blocks(d.latent_pool) * Tv;
  LGN_CHECK_ARG(d.tau_v_in == 0 || d.tau_v_in == Tin, "step: the decoder must consume the encoder's %d pooled latent vectors", Tin);
  Deferred dq;
  dq.parts = w.parts;
  dq.cap = w.parts_size;
  RadFinJob fin{};
  LGN_TRY(zero_ranges(grads, (size_t)n_params, w.zero0(), w.zero_doubles, nullptr, 0, st));
  // ---------------- forward ----------------
  LGN_TRY(enc_input_fwd(B, Ne, ce[0], K, p4, in_scalars, params + enc_off[0], params + enc_off[1], w.enc.s[0], w.enc.v[0], st));
  LGN_TRY(levels_fwd(de, false, ce, params, enc_off, w.enc, p4, mask, st));
  LGN_TRY(enc_latent_fwd(B, Ne, ce[L], Ts, Tv, d.latent_pool, w.enc.s[L], w.enc.v[L], params + enc_off[S.out0(false)],
                         params + enc_off[S.out0(false) + 1], w.lat_s, w.lat_v, w.idx, st));
  LGN_TRY(dec_input_fwd(B, Nd, cd[0], Tin, w.lat_v, params + dec_off[1], params + dec_off[2], params + dec_off[3], w.pdec, w.dec.s[0],
                        w.dec.v[0], st));
  int cur = 0;
  {
    DQ_NEW(part, (size_t)B * 2 * cd[L]);
    const LossStage ls{params + dec_off[S.out0(true) + 1], target, 1.0, recon, loss_part, w.gv[cur], part};
    const bool rides = level_fwd_carries_loss(Nd, d.flags);
    LGN_TRY(levels_fwd(dd, true, cd, params, dec_off, w.dec, w.pdec, nullptr, st, nullptr, rides ? &ls : nullptr));
    if (!rides) LGN_TRY(dec_output_loss(B, Nd, cd[L], w.dec.v[L], ls.wo1, target, 1.0, recon, loss_part, w.gv[cur], part, st));
    dq.add(part, B, 2 * cd[L], 0, 2 * cd[L], grads + dec_off[S.out0(true) + 1]);
  }
  // ---------------- backward ----------------
  LGN_TRY(levels_bwd(dd, true, cd, params, grads, dec_off, w.dec, w.pdec, nullptr, w, dq, fin, cur, /*has_s_grad=*/false, st));
  {
    const int C0 = cd[0], row = 4 * C0 + 2 * Nd * Tin;
    const int CL = ce[L], KL = pool_mix_in(d.latent_pool, Ne, CL), rowe = 2 * (Ts + Tv) * KL;
    DQ_NEW(part, (size_t)B * row);
    DQ_NEW(parte, (size_t)B * rowe);
    const int rd = cur, wr = cur ^ 1;
    cur = wr;
    LGN_TRY(dec_input_bwd(B, Nd, C0, Tin, w.lat_v, params + dec_off[1], params + dec_off[3], w.pdec, w.g_p, w.gs[rd], w.gv[rd], w.g_lat_v,
                          part, st));
    // the decoder never reads the latent scalars: their gradient is the zero block's g_lat_s
    LGN_TRY(enc_latent_bwd(B, Ne, CL, Ts, Tv, d.latent_pool, w.enc.s[L], w.enc.v[L], params + enc_off[S.out0(false)],
                           params + enc_off[S.out0(false) + 1], w.g_lat_s, w.g_lat_v, w.idx, w.gs[wr], w.gv[wr], parte, st));
    dq.add(part, B, row, 0, 2 * C0, grads + dec_off[2]);
    dq.add(part, B, row, 2 * C0, 2 * C0, grads + dec_off[3]);
    dq.add(part, B, row, 4 * C0, 2 * Nd * Tin, grads + dec_off[1]);
    dq.add(parte, B, rowe, 0, 2 * Ts * KL, grads + enc_off[S.out0(false)]);
    dq.add(parte, B, rowe, 2 * Ts * KL, 2 * Tv * KL, grads + enc_off[S.out0(false) + 1]);
  }
  LGN_TRY(levels_bwd(de, false, ce, params, grads, enc_off, w.enc, p4, mask, w, dq, fin, cur, /*has_s_grad=*/false, st));
  {
    const int C0 = ce[0], row = (2 * K + 2) * C0;
    DQ_NEW(part, (size_t)B * row);
    LGN_TRY(enc_input_bwd(B, Ne, C0, K, p4, in_scalars, w.gs[cur], w.gv[cur], part, st));
    dq.add(part, B, row, 0, 2 * C0 * K, grads + enc_off[0]);
    dq.add(part, B, row, 2 * C0 * K, 2 * C0, grads + enc_off[1]);
  }
  LGN_CHECK_ARG(dq.off <= dq.cap, "step: partial-row workspace overflow (%zu > %zu)", dq.off, dq.cap);
  if (tail && !(d.flags & LGN_NET_SPLIT_TAIL)) {
    const int rc = step_tail(dq.segs, fin, *tail, st);
    if (rc == 0) return 0;
    if (rc != -2) return rc;
  }
  LGN_TRY(finish_reductions(dq, fin, grads, n_params, tail ? nullptr : w.tail_cnt, d.flags, st));
  if (tail)
    LGN_TRY(finalize_step(tail->w, tail->g, tail->n, tail->loss_part, tail->nB, tail->lambda, tail->m, tail->v, tail->step_dev, tail->lr,
                          tail->beta1, tail->beta2, tail->eps, tail->do_adam, tail->loss_out, st));
  return 0;
}

// forward + backward of a training step; with `tail` (single process: nothing sits between the gradients and the optimiser) the
// deferred reductions, the radial finalisation, L1 + Adam and the loss assembly are ONE launch (step_tail.hip) instead of three
static int step_fwd_bwd(const lgn_net_desc* d, const double* params, double* grads, long long n_params, const int64_t* enc_off,
                        const int64_t* dec_off, const double* p4, const double* target, const uint8_t* mask, const double* in_scalars,
                        double* workspace, long long workspace_doubles, double* recon, double* loss_part, void* stream,
                        const StepTailArgs* tail) {
  if (int rc = check_desc(d)) return rc;
  LGN_CHECK_ARG(params && grads && enc_off && dec_off && p4 && target && mask && workspace && recon && loss_part && n_params > 0,
                "step_fwd_bwd: null pointer");
  LGN_CHECK_ARG(d->n_in_scalars <= 1 || in_scalars, "step_fwd_bwd: %d input scalars per node, but in_scalars is NULL", d->n_in_scalars);
  hipStream_t st = (hipStream_t)stream;
  if (is_generic(*d, false) || is_generic(*d, true)) {
    LGN_CHECK_ARG(!step_is_split(*d), "step_fwd_bwd: table-driven networks take the mass as the only input scalar and one node count "
                  "for both networks (n_in_scalars=%d, N=%d, dec_N=%d): use the per-network calls", d->n_in_scalars, d->N, d->dec_N);
    return gen_step_fwd_bwd(*d, params, grads, n_params, enc_off, dec_off, p4, target, mask, workspace, workspace_doubles, recon,
                            loss_part, st, tail);
  }
  Work w = carve(*d, workspace);
  // the layout depends on run-time switches (LGN_AMD_DEC_PAIRWISE / LGN_AMD_LEVEL_V2 change the partial-row counts):
  // refuse before anything is enqueued if the caller sized the workspace under different settings
  LGN_CHECK_ARG((long long)w.total <= workspace_doubles, "step: workspace holds %lld doubles, this configuration needs %zu",
                workspace_doubles, w.total);
  const Slots S{d->n_levels, d->mlp_nlin};
  const int L = d->n_levels, B = d->B, N = d->N, Ts = d->tau_s, Tv = d->tau_v;
  const int* ce = d->enc_channels;
  const int* cd = d->dec_channels;
  if (step_is_split(*d))
    return step_fwd_bwd_split(*d, w, params, grads, n_params, enc_off, dec_off, p4, target, mask, in_scalars, recon, loss_part, st, tail);
  // MLP parameter blocks must be contiguous (W_0, b_0, W_1, b_1, ...) for the single-reduce path
  for (int dec = 0; dec < 2; ++dec) {
    const int64_t* off = dec ? dec_off : enc_off;
    const int* ch = dec ? cd : ce;
    for (int l = 0; l < L; ++l) {
      const int D = 2 * ch[l + 1], H = d->mlp_hidden_mul * D;
      int64_t expect = off[S.mlp(dec, l, 0)];
      for (int q = 0; q < d->mlp_nlin; ++q) {
        const int hin = q == 0 ? D : H, hout = q == d->mlp_nlin - 1 ? D : H;
        LGN_CHECK_ARG(off[S.mlp(dec, l, 2 * q)] == expect, "step: MLP weights are not contiguous in the flat parameter buffer");
        expect += (int64_t)hin * hout;
        LGN_CHECK_ARG(off[S.mlp(dec, l, 2 * q + 1)] == expect, "step: MLP biases are not contiguous in the flat parameter buffer");
        expect += hout;
      }
    }
  }
  Deferred dq;
  dq.parts = w.parts;
  dq.cap = w.parts_size;
  RadFinJob fin{};

  // ---------------- forward ----------------
  // the first kernel also zeroes the gradient buffer (dead parameters keep an exact zero) and zeros_s | g_p | g_lat_s
  // (the encoder's input stage and the two clears ride on the first level's kernel: levels_fwd / InputStage)
  const InputStage in0{params + enc_off[0], params + enc_off[1], grads, (size_t)n_params, w.zeros_s, w.zero_doubles};
  LGN_TRY(levels_fwd(*d, false, ce, params, enc_off, w.enc, p4, mask, st, &in0));
  LGN_TRY(junction_fwd(B, N, ce[L], Ts, Tv, d->latent_pool, w.enc.s[L], w.enc.v[L], params + enc_off[S.out0(false)],
                       params + enc_off[S.out0(false) + 1], w.lat_s, w.lat_v, w.idx, cd[0], params + dec_off[1], params + dec_off[2],
                       params + dec_off[3], w.pdec, w.dec.s[0], w.dec.v[0], st));
  // ---------------- loss (and its backward), on the last decoder level's kernel when that is one workgroup per jet ----------
  int cur = 0;
  {
    DQ_NEW(part, (size_t)B * 2 * cd[L]);
    const LossStage ls{params + dec_off[S.out0(true) + 1], target, 1.0, recon, loss_part, w.gv[cur], part};
    const bool rides = level_fwd_carries_loss(N, d->flags);
    LGN_TRY(levels_fwd(*d, true, cd, params, dec_off, w.dec, w.pdec, nullptr, st, nullptr, rides ? &ls : nullptr));
    if (!rides)
      LGN_TRY(dec_output_loss(B, N, cd[L], w.dec.v[L], ls.wo1, target, 1.0, recon, loss_part, w.gv[cur], part, st));
    dq.add(part, B, 2 * cd[L], 0, 2 * cd[L], grads + dec_off[S.out0(true) + 1]);
  }

  // ---------------- backward ----------------
  LGN_TRY(levels_bwd(*d, true, cd, params, grads, dec_off, w.dec, w.pdec, nullptr, w, dq, fin, cur, /*has_s_grad=*/false, st));
  {
    const int C0 = cd[0], Tin = pool_blocks(d->latent_pool) * Tv, row = 4 * C0 + 2 * N * Tin;
    const int CL = ce[L], KL = pool_mix_in(d->latent_pool, N, CL), rowe = 2 * (Ts + Tv) * KL;
    DQ_NEW(part, (size_t)B * row);
    DQ_NEW(parte, (size_t)B * rowe);
    // reads the gradient w.r.t. the decoder's level-0 features, writes the one w.r.t. the encoder's last level into the
    // other buffer pair (jets do not occupy the same slices when the two channel counts differ)
    const int rd = cur, wr = cur ^ 1;
    cur = wr;
    LGN_TRY(junction_bwd(B, N, C0, Tin, w.lat_v, params + dec_off[1], params + dec_off[3], w.pdec, w.g_p, w.gs[rd], w.gv[rd], w.g_lat_v, part,
                         CL, Ts, Tv, d->latent_pool, w.enc.s[L], w.enc.v[L], params + enc_off[S.out0(false)], params + enc_off[S.out0(false) + 1],
                         w.g_lat_s, w.idx, w.gs[wr], w.gv[wr], parte, st));
    dq.add(part, B, row, 0, 2 * C0, grads + dec_off[2]);
    dq.add(part, B, row, 2 * C0, 2 * C0, grads + dec_off[3]);
    dq.add(part, B, row, 4 * C0, 2 * N * Tin, grads + dec_off[1]);
    dq.add(parte, B, rowe, 0, 2 * Ts * KL, grads + enc_off[S.out0(false)]);
    dq.add(parte, B, rowe, 2 * Ts * KL, 2 * Tv * KL, grads + enc_off[S.out0(false) + 1]);
  }
  double* const in0_grads[2] = {grads + enc_off[0], grads + enc_off[1]};
  bool in0_done = false;
  LGN_TRY(levels_bwd(*d, false, ce, params, grads, enc_off, w.enc, p4, mask, w, dq, fin, cur, /*has_s_grad=*/false, st, in0_grads, &in0_done));
  if (!in0_done) {      // (three-kernel level backward: N > 40, LGN_AMD_LEVEL_V2)
    const int C0 = ce[0];
    DQ_NEW(part, (size_t)B * 4 * C0);
    LGN_TRY(enc_input_bwd(B, N, C0, 1, p4, nullptr, w.gs[cur], w.gv[cur], part, st));
    dq.add(part, B, 4 * C0, 0, 2 * C0, grads + enc_off[0]);
    dq.add(part, B, 4 * C0, 2 * C0, 2 * C0, grads + enc_off[1]);
  }
  LGN_CHECK_ARG(dq.off <= dq.cap, "step: partial-row workspace overflow (%zu > %zu)", dq.off, dq.cap);
  if (tail && !(d->flags & LGN_NET_SPLIT_TAIL)) {
    const int rc = step_tail(dq.segs, fin, *tail, st);
    if (rc == 0) return 0;
    if (rc != -2) return rc;                   // -2: does not fit the fused form -> the separate launches below
  }
  // (data-parallel step: the all-reduce follows -- reductions + radial finalisation as one launch where that fits)
  LGN_TRY(finish_reductions(dq, fin, grads, n_params, tail ? nullptr : w.tail_cnt, d->flags, st));
  if (tail)
    LGN_TRY(finalize_step(tail->w, tail->g, tail->n, tail->loss_part, tail->nB, tail->lambda, tail->m, tail->v, tail->step_dev, tail->lr,
                          tail->beta1, tail->beta2, tail->eps, tail->do_adam, tail->loss_out, st));
  return 0;
}
extern "C" {

int lgn_step_fwd_bwd_f64(const lgn_net_desc* d, const double* params, double* grads, long long n_params, const int64_t* enc_off,
                         const int64_t* dec_off, const double* p4, const double* target, const uint8_t* mask, const double* in_scalars,
                         double* workspace, long long workspace_doubles, double* recon, double* loss_part, void* stream) {
  return step_fwd_bwd(d, params, grads, n_params, enc_off, dec_off, p4, target, mask, in_scalars, workspace, workspace_doubles, recon,
                      loss_part, stream, nullptr);
}

int lgn_step_train_f64(const lgn_net_desc* d, double* params, double* grads, long long n_params, const int64_t* enc_off,
                       const int64_t* dec_off, const double* p4, const double* target, const uint8_t* mask, const double* in_scalars,
                       double* workspace, long long workspace_doubles, double* recon, double* loss_part, int n_loss, double l1_lambda, double* adam_m,
                       double* adam_v, long long* step_dev, double lr, double beta1, double beta2, double eps, int do_adam,
                       double* loss_out, void* stream) {
  LGN_CHECK_ARG(loss_out && n_loss > 0, "step_train: null pointer");
  LGN_CHECK_ARG(!do_adam || (adam_m && adam_v && step_dev), "step_train: Adam state missing");
  const StepTailArgs tail{params, grads, (long)n_params, adam_m, adam_v, reinterpret_cast<long*>(step_dev), l1_lambda, lr, beta1, beta2,
                          eps, do_adam, loss_part, n_loss, loss_out, 0, nullptr};
  // LGN_NET_SPLIT_TAIL (frozen into the descriptor): the three separate launches (reduce_segments, rad_finalize_batch, l1_adam) -- the
  // A/B switch of the fused tail
  return step_fwd_bwd(d, params, grads, n_params, enc_off, dec_off, p4, target, mask, in_scalars, workspace, workspace_doubles, recon,
                      loss_part, stream, &tail);
}

int lgn_step_finalize_f64(double* params, double* grads, long long n_params, const double* loss_part, int n_loss, double l1_lambda,
                          double* adam_m, double* adam_v, long long* step_dev, double lr, double beta1, double beta2, double eps,
                          int do_adam, double* loss_out, void* stream) {
  LGN_CHECK_ARG(params && grads && loss_part && loss_out && n_params > 0 && n_loss > 0, "step_finalize: null pointer");
  LGN_CHECK_ARG(!do_adam || (adam_m && adam_v && step_dev), "step_finalize: Adam state missing");
  hipStream_t st = (hipStream_t)stream;
  LGN_TRY(finalize_step(params, grads, (long)n_params, loss_part, n_loss, l1_lambda, adam_m, adam_v, reinterpret_cast<long*>(step_dev),
                        lr, beta1, beta2, eps, do_adam, loss_out, st));
  return 0;
}

}  // extern "C"
