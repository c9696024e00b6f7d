// lgn-autoencoder_amd/csrc/step_tail.hip -- the tail of a single-process training step in ONE launch (round 5).
//
// What the step did in three dependent launches at their latency floor -- reduce_segments (level_bwd.hip: every partial row of
// the backward, ~107 MB at cfg2, HBM-bound), rad_finalize_batch (level_bwd.hip: radial-parameter gradients from the reduced sums,
// one workgroup per level) and l1_adam (net_kernels.hip: L1 sub-gradient, Adam, loss assembly) -- 20 + 5 + 8.5 us of a 540 us
// step, 16 + 5 + 8 us of the 347 us step at 64 jets.  Reference: utils/train.py:482-492 (loss = chamfer + lambda * l1 norm,
// backward, two Adam steps).
//
// One tile = 256 columns of one segment (RedSeg: rows of partial sums -> one output range) in four groups of 64, 16 row groups per
// column as in reduce_segments_kernel: same summation order, same bits (64-column tiles: 1 100 workgroups whose exits -- two
// dependent atomic round trips each -- took longer than the three launches they replace).  A column whose output is a PARAMETER gradient is finished on the spot:
// g = sum + lambda sign(w), Adam moments, weight update -- Adam is elementwise, nothing waits for other tiles.  Parameters no
// segment covers (dead ones: their gradient is the zero the step's first kernel wrote) come as segments without rows.  The
// radial parameters of an encoder level need the level's reduced sums `tot` as a whole (rad_finalize: contractions over the
// 4C rows): the tiles of that segment publish their sums with device-scope atomic exchanges and count up a per-level counter;
// the tile that completes the count finalises the level and runs L1 + Adam on its radial parameters -- `tot` segments are the
// FIRST tiles of the grid, so this happens under the bulk of the reduction.  The last workgroup of the grid assembles the loss
// from per-tile |w| sums in index order (as l1_adam does): deterministic, whichever workgroup that is.
// Cross-workgroup hand-off as in l1_adam_kernel: atomic exchange (awaited) -> counter, atomic loads on the reading side; no
// __threadfence() (a device-scope fence writes back and invalidates the XCD's whole L2 on this part).
#include "net.hpp"
#include "ops.hpp"
#include "tail_dev.hpp"
#include "../../include/lgn_amd.h"
#include <algorithm>
#include <vector>

namespace lgn {
namespace {

constexpr int TAIL_RG = 16;                  // row groups per column (== RED_RG of level_bwd.hip: the same summation order)
constexpr int TAIL_THREADS = 64 * TAIL_RG;
#ifndef LGN_TAIL_CP
#define LGN_TAIL_CP 4
#endif
constexpr int TAIL_CP = LGN_TAIL_CP;                   // groups of 64 columns per workgroup (one owner wave each)
constexpr int TAIL_COLS = 64 * TAIL_CP;
constexpr int TAIL_MAX_LEV = 4;              // == RadFinJob::it

struct TailDev {
  double *w, *g, *m, *v;
  long n;
  double lambda, lr, beta1, beta2, eps;
  long* step_dev;
  int do_adam;
  double* l1_part;                           // [ntiles + nlev] |w| sums: tiles, then levels
  double* powers;                            // {t, beta1^t, beta2^t} x 2 (l1_adam_kernel's cache)
  unsigned long long* done;                  // finished workgroups of this launch
  unsigned long long* lev_done;              // [TAIL_MAX_LEV] finished tiles of a level's `tot` segment
  const double* loss_part;
  int nB;
  double* loss_out;
  int ntiles, nlev;
  int lev_seg[TAIL_MAX_LEV];                 // index (in the job) of the level's `tot` segment
  int lev_tiles[TAIL_MAX_LEV];
  int reduce_only;                           // sums and radial finalisation only: no L1, no Adam, no loss (the data-parallel step)
  RadFinJob fin;
};

static_assert(sizeof(RedJob<double>) + sizeof(TailDev) <= 4096, "step_tail_kernel: both argument blocks travel by value (4 KB of kernel arguments)");

__device__ __forceinline__ void put_shared(double* p, double x) {        // visible to every XCD once the return value is there
  unsigned long long old = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(x),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) : : "memory");
}
// |w| sums travel as -x (sign bit set, -0.0 for zero): a slot whose bits are zero has not been written in this launch.  The writer
// does not wait for the store -- the reader (the last workgroup, which learns that it is the last from a DIFFERENT address) polls a
// zero slot until the store, already in flight, lands; it leaves the slots zero for the next launch.
__device__ __forceinline__ void post_sum(double* p, double x) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(-x), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double take_sum(double* p) {
  unsigned long long b = 0ull;
  for (int spin = 0; spin < (1 << 20); ++spin) {
    b = __hip_atomic_load(reinterpret_cast<unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!b) return __longlong_as_double(0x7ff8000000000000LL);      // gave up (a store that never landed): a NaN loss, not a wrong one
  return -__longlong_as_double((long long)b);
}
__device__ __forceinline__ double get_shared(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
// sum over the workgroup's TAIL_THREADS threads in a fixed order; valid on thread 0
__device__ __forceinline__ double tail_block_sum(double v, double* red) {
  v = group_sum<64>(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < TAIL_RG; ++q) s += red[q];
  __syncthreads();
  return s;
}
// L1 sub-gradient + Adam of one parameter (weight wi, moments mi / vi) whose loss gradient is `gsum` -- the arithmetic of
// l1_adam_kernel, operation for operation; returns |w| before the update
__device__ __forceinline__ double finish_param(const TailDev& t, long i, double gsum, double wi, double mi0, double vi0, double bc1,
                                               double bc2_sqrt) {
  const AdamOut o = l1_adam_one(wi, gsum, mi0, vi0, t.lambda, t.lr, t.beta1, t.beta2, t.eps, bc1, bc2_sqrt);
  t.g[i] = o.g;
  if (t.do_adam) {
    t.m[i] = o.m;
    t.v[i] = o.v;
    t.w[i] = o.w;
  }
  return fabs(wi);
}
__device__ __forceinline__ double finish_param(const TailDev& t, long i, double gsum, double bc1, double bc2_sqrt) {
  return finish_param(t, i, gsum, t.w[i], t.do_adam ? t.m[i] : 0.0, t.do_adam ? t.v[i] : 0.0, bc1, bc2_sqrt);
}

__global__ __launch_bounds__(TAIL_THREADS) void step_tail_kernel(RedJob<double> job, TailDev t) {
  __shared__ double red[TAIL_CP][TAIL_RG][64];
  __shared__ double sred[TAIL_RG];
  __shared__ double bc[4];
  __shared__ int flag;
  __shared__ double sh[2 * 32 * NB + 2 * 32 + 32 * NB];       // rad_finalize: T1 | T2 | S | dB | w (R <= 32 rows)
  __shared__ double rabc[3 * NB];                             // the level's bell parameters a | b | c
  const int tid = threadIdx.x;
  int k = 0;
  while (k + 1 < job.nseg && (int)blockIdx.x >= job.tile0[k + 1]) ++k;
  const RedSeg<double> sg = job.seg[k];
  int lev = -1;
#pragma unroll
  for (int q = 0; q < TAIL_MAX_LEV; ++q)
    if (q < t.nlev && t.lev_seg[q] == k) lev = q;
  const int cl = tid & 63, rg = tid >> 6;
  const int cp = lev >= 0 ? 1 : TAIL_CP;                      // a level's `tot` in 64-column tiles: done early, finalised under the bulk
  const int col0 = ((int)blockIdx.x - job.tile0[k]) * 64 * cp;
  const bool is_param = sg.out >= t.g && sg.out < t.g + t.n;
  // the owner of a column (wave `pass` of the first TAIL_CP waves, lane cl) asks for its parameter's state BEFORE the sums: one
  // memory round trip for everything this workgroup reads
  const int mycol = col0 + rg * 64 + cl;
  const bool owner = rg < cp && mycol < sg.n;
  const long pidx = (long)(sg.out - t.g) + mycol;
  double wi = 0.0, mi = 0.0, vi = 0.0, g0 = 0.0;
  if (owner && is_param && !t.reduce_only) {
    wi = t.w[pidx];
    if (t.do_adam) { mi = t.m[pidx]; vi = t.v[pidx]; }
  }
  if (owner && sg.rows <= 0) g0 = sg.out[mycol];              // no producer: what the step's first kernel left there (zero)
  // thread 0: step number and both cached power slots, requested NOW (as a chain behind the sums -- counter, then its slot -- they
  // held every workgroup at its first barrier for two more memory round trips)
  long ti_prev = 0;
  double pw[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (tid == 0 && t.do_adam) {
    ti_prev = *t.step_dev;
#pragma unroll
    for (int q = 0; q < 6; ++q) pw[q] = t.powers[q];
  }
  // ---- the column sums (reduce_segments_kernel's order), TAIL_CP groups of 64 columns per workgroup ----
  double acc[TAIL_CP];
#pragma unroll
  for (int pass = 0; pass < TAIL_CP; ++pass) {
    const int col = col0 + pass * 64 + cl;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (pass < cp && col < sg.n && sg.rows > 0) {
      const double* p = sg.part + sg.col0 + col;
      int r = rg;
      for (; r + 3 * TAIL_RG < sg.rows; r += 4 * TAIL_RG) {
        s0 += p[(size_t)r * sg.stride];
        s1 += p[(size_t)(r + TAIL_RG) * sg.stride];
        s2 += p[(size_t)(r + 2 * TAIL_RG) * sg.stride];
        s3 += p[(size_t)(r + 3 * TAIL_RG) * sg.stride];
      }
      for (; r < sg.rows; r += TAIL_RG) s0 += p[(size_t)r * sg.stride];
    }
    acc[pass] = (s0 + s1) + (s2 + s3);
  }
  if (tid == 0) {
    bc[0] = bc[1] = bc[2] = bc[3] = 1.0;
    if (t.do_adam) {
      const long ti = ti_prev + 1;
      const double tt = (double)ti;
      const int odd = (int)(ti & 1);
      const double sl0 = odd ? pw[3] : pw[0], sl1 = odd ? pw[4] : pw[1], sl2 = odd ? pw[5] : pw[2];
      double p1, p2;
      if (sl0 == tt) { p1 = sl1; p2 = sl2; }
      else { p1 = pow(t.beta1, tt); p2 = pow(t.beta2, tt); }
      bc[0] = 1.0 - p1;
      bc[1] = sqrt(1.0 - p2);
      bc[2] = p1;
      bc[3] = p2;
    }
    flag = 0;
  }
#pragma unroll
  for (int pass = 0; pass < TAIL_CP; ++pass) red[pass][rg][cl] = acc[pass];
  __syncthreads();
  const double bc1 = bc[0], bc2_sqrt = bc[1];
  double l1 = 0.0;
  if (owner) {
    double s = g0;
    if (sg.rows > 0) {
      s = 0.0;
#pragma unroll
      for (int q = 0; q < TAIL_RG; ++q) s += red[rg][q][cl];
    }
    if (is_param && t.reduce_only) t.g[pidx] = s;
    else if (is_param) l1 = finish_param(t, pidx, s, wi, mi, vi, bc1, bc2_sqrt);
    else if (lev >= 0) put_shared(sg.out + mycol, s);
    else sg.out[mycol] = s;
  }
  // ---- radial parameters of an encoder level: the tile that completes the level's `tot` finalises it ----
  if (lev >= 0) {
    __syncthreads();                           // the exchanges of the TAIL_CP owner waves have returned
    if (tid == 0) {
      const unsigned long long old = __hip_atomic_fetch_add(t.lev_done + lev, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag = old == (unsigned long long)t.lev_tiles[lev] - 1;
    }
    __syncthreads();
    if (flag) {
      const RadFinJob::Item it = t.fin.it[lev];
      const int C = it.C, R = 4 * C, F = 2 * C;
      double* T1 = sh;
      double* T2 = T1 + R * NB;
      double* S = T2 + R * NB;
      double* dB = S + R;
      double* w = dB + R;                      // [R][NB]: rows 0..F-1 = w0, F..R-1 = w1
      // one radial parameter per thread -- R NB Linear weights | R biases | 3 NB bell parameters (a, b, c) -- whose state is requested
      // together with the sums: one memory round trip for the whole finalisation (R NB + R + 3 NB <= 732 threads at C = 8)
      const int nW = R * NB;
      double* gp = nullptr;
      int r = 0, kk = 0, which = -1;           // which: -1 weight, -2 bias, 0 / 1 / 2 = a / b / c
      if (tid < nW) {
        r = tid / NB;  kk = tid - r * NB;
        const int lin = r / F, f = r - lin * F;
        gp = (lin ? it.g_w1 : it.g_w0) + f * NB + kk;
      } else if (tid < nW + R) {
        r = tid - nW;  which = -2;
        const int lin = r / F, f = r - lin * F;
        gp = (lin ? it.g_b1 : it.g_b0) + f;
      } else if (tid < nW + R + 3 * NB) {
        const int e = tid - nW - R;
        which = e / NB;  kk = e - which * NB;
        gp = (which == 0 ? it.g_a : which == 1 ? it.g_b : it.g_c) + kk;
      }
      const long pi = gp ? (long)(gp - t.g) : 0;
      double pw_ = 0.0, pm_ = 0.0, pv_ = 0.0;
      if (gp && !t.reduce_only) {
        pw_ = t.w[pi];
        if (t.do_adam) { pm_ = t.m[pi]; pv_ = t.v[pi]; }
      }
      for (int e = tid; e < 2 * R * NB + 2 * R; e += TAIL_THREADS) sh[e] = get_shared(it.tot + e);
      for (int e = tid; e < R * NB; e += TAIL_THREADS) w[e] = e < F * NB ? it.w0[e] : it.w1[e - F * NB];
      for (int e = tid; e < 3 * NB; e += TAIL_THREADS) rabc[e] = (e < NB ? it.ra : e < 2 * NB ? it.rb : it.rc)[e % NB];
      __syncthreads();                         // every read of the level's parameters is done before any of them is updated
      const double* ra = rabc;
      const double* rb = rabc + NB;
      const double* rc = rabc + 2 * NB;
      double l1r = 0.0;
      if (gp) {                                // (rad_finalize_batch_kernel's arithmetic)
        double gsum;
        if (which == -1) gsum = radfin_weight(rb[kk], T1[tid], ra[kk], S[r]);
        else if (which == -2) gsum = dB[r];
        else {
          const double* X = which == 0 ? S : which == 1 ? T1 + kk : T2 + kk;
          const double d = radfin_dot(w + kk, NB, X, which == 0 ? 1 : NB, R);
          gsum = which == 2 ? radfin_c(rb[kk], rc[kk], d) : d;
        }
        if (t.reduce_only) t.g[pi] = gsum;
        else l1r = finish_param(t, pi, gsum, pw_, pm_, pv_, bc1, bc2_sqrt);
      }
      // the level's counter back to zero for the next launch (nobody counts on it again in this one)
      if (tid == 0) __hip_atomic_store(t.lev_done + lev, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t.reduce_only) return;
      l1r = tail_block_sum(l1r, sred);
      if (tid == 0) post_sum(t.l1_part + t.ntiles + lev, l1r);
    }
  }
  if (t.reduce_only) return;
  // ---- |w| of this tile; the last workgroup assembles the loss ----
  l1 = tail_block_sum(l1, sred);
  if (tid == 0) {
    post_sum(t.l1_part + blockIdx.x, l1);
    flag = __hip_atomic_fetch_add(t.done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1;
  }
  __syncthreads();
  if (!flag) return;
  double a = 0.0, l = 0.0;
  for (int i = tid; i < t.ntiles + t.nlev; i += TAIL_THREADS) a += take_sum(t.l1_part + i);
  for (int i = tid; i < t.nB; i += TAIL_THREADS) l += t.loss_part[i];
  a = tail_block_sum(a, sred);
  l = tail_block_sum(l, sred);
  if (tid == 0) {
    t.loss_out[0] = l + t.lambda * a;
    t.loss_out[1] = l;
    t.loss_out[2] = a;
    if (t.do_adam) {
      const long tn = *t.step_dev + 2;
      double* slot = t.powers + 3 * (tn & 1);
      slot[1] = bc[2] * t.beta1;
      slot[2] = bc[3] * t.beta2;
      slot[0] = (double)tn;
      *t.step_dev += 1;
    }
    *t.done = 0ull;
  }
}

}  // namespace

// Runs the fused tail for the segments a step collected.  Returns 0 when launched, -2 when this step does not fit the fused form
// (too many segments / tiles for one launch, overlapping outputs): the caller then takes the three-launch route.
int step_tail(const std::vector<RedSeg<double>>& segs, const RadFinJob& fin, const StepTailArgs& ta, hipStream_t st) {
  if (fin.n > TAIL_MAX_LEV) return -2;
  struct Range { long lo, hi; };
  std::vector<Range> cover;
  std::vector<RedSeg<double>> order;
  RedJob<double> job{};
  TailDev t{};
  t.w = ta.w; t.g = ta.g; t.m = ta.m; t.v = ta.v; t.n = ta.n;
  t.lambda = ta.lambda; t.lr = ta.lr; t.beta1 = ta.beta1; t.beta2 = ta.beta2; t.eps = ta.eps;
  t.step_dev = ta.step_dev; t.do_adam = ta.do_adam;
  t.loss_part = ta.loss_part; t.nB = ta.nB; t.loss_out = ta.loss_out;
  t.nlev = fin.n;
  t.fin = fin;
  auto in_grads = [&](const double* p) { return p >= ta.g && p < ta.g + ta.n; };
  // the levels' `tot` segments first (their finalisation then runs under the rest of the grid)
  for (int q = 0; q < fin.n; ++q) {
    int found = -1;
    for (size_t i = 0; i < segs.size(); ++i)
      if (segs[i].out == fin.it[q].tot && segs[i].n > 0) found = (int)i;
    if (found < 0) return -2;
    t.lev_seg[q] = (int)order.size();
    t.lev_tiles[q] = cdiv(segs[found].n, 64);
    order.push_back(segs[found]);
    const RadFinJob::Item& it = fin.it[q];
    const int F = 2 * it.C;
    const std::pair<const double*, int> outs[7] = {{it.g_a, NB}, {it.g_b, NB}, {it.g_c, NB}, {it.g_w0, F * NB}, {it.g_b0, F},
                                                   {it.g_w1, F * NB}, {it.g_b1, F}};
    for (const auto& o : outs) {
      if (!in_grads(o.first)) return -2;
      cover.push_back(Range{(long)(o.first - ta.g), (long)(o.first - ta.g) + o.second});
    }
  }
  for (const auto& s : segs) {
    if (s.n <= 0) continue;
    bool is_tot = false;
    for (int q = 0; q < fin.n; ++q) is_tot = is_tot || s.out == fin.it[q].tot;
    if (is_tot) continue;
    if (in_grads(s.out)) cover.push_back(Range{(long)(s.out - ta.g), (long)(s.out - ta.g) + s.n});
    order.push_back(s);
  }
  std::sort(cover.begin(), cover.end(), [](const Range& a, const Range& b) { return a.lo < b.lo; });
  long at = 0;
  for (const auto& r : cover) {
    if (r.lo < at || r.hi > ta.n) return -2;                  // two producers of one gradient: not a case of this kernel
    if (r.lo > at && !ta.reduce_only) order.push_back(RedSeg<double>{nullptr, 0, 0, 0, (int)(r.lo - at), ta.g + at});
    at = r.hi;
  }
  if (at < ta.n && !ta.reduce_only) order.push_back(RedSeg<double>{nullptr, 0, 0, 0, (int)(ta.n - at), ta.g + at});
  if ((int)order.size() > RED_MAX_SEG) return -2;
  int tiles = 0;
  for (const auto& s : order) {
    job.seg[job.nseg] = s;
    job.tile0[job.nseg] = tiles;
    tiles += cdiv(s.n, job.nseg < fin.n ? 64 : TAIL_COLS);        // (the first fin.n segments are the levels' `tot`)
    ++job.nseg;
  }
  job.tile0[job.nseg] = tiles;
  t.ntiles = tiles;
  t.reduce_only = ta.reduce_only;
  if (ta.reduce_only) {
    if (!ta.counters) return -2;
    t.lev_done = ta.counters;
  } else {
    if (tiles + fin.n > LGN_FINALIZE_SCRATCH - 12) return -2;
    double* scratch = ta.loss_out + 3;
    t.l1_part = scratch;
    t.lev_done = reinterpret_cast<unsigned long long*>(scratch + LGN_FINALIZE_SCRATCH - 11);
    t.powers = scratch + LGN_FINALIZE_SCRATCH - 7;
    t.done = reinterpret_cast<unsigned long long*>(scratch + LGN_FINALIZE_SCRATCH - 1);
  }
  hipLaunchKernelGGL(step_tail_kernel, dim3(tiles), dim3(TAIL_THREADS), 0, st, job, t);
  LGN_CHECK_LAUNCH();
  return 0;
}

}  // namespace lgn
