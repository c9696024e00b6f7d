// lgn-autoencoder_amd/csrc/tail_dev.hpp -- the arithmetic of a step's tail with every multiply-add spelled out, shared by the
// kernels that run it as separate launches (rad_finalize_batch_kernel in level_bwd.hip, l1_adam_kernel in net_kernels.hip) and
// by the fused launch (step_tail.hip): which products the compiler contracts into fused multiply-adds may differ from one kernel to
// the next, and the two routes are tested to agree bit for bit.
#pragma once
#include "common.hpp"

namespace lgn {

// radial-parameter gradients from a level's reduced sums (RadPolyTrig: lgn/nn/position_levels.py:118-209; sums T1 | T2 | S | dB)
__device__ __forceinline__ double radfin_weight(double rb, double t1, double ra, double s) { return __builtin_fma(rb, t1, ra * s); }
// sum_r w[r ws] x[r xs], r ascending
__device__ __forceinline__ double radfin_dot(const double* w, int ws, const double* x, int xs, int R) {
  double d = 0.0;
  for (int r = 0; r < R; ++r) d = __builtin_fma(w[r * ws], x[r * xs], d);
  return d;
}
__device__ __forceinline__ double radfin_c(double rb, double rc, double dc) { return ((-2.0 * rb) * rc) * dc; }

// L1 sub-gradient + Adam of one parameter (torch.optim.Adam defaults; utils/train.py:484-492): gradient incl. the L1 term, new
// moments, new weight
struct AdamOut { double g, m, v, w; };
__device__ __forceinline__ AdamOut l1_adam_one(double w, double gsum, double m, double v, double lambda, double lr, double beta1,
                                               double beta2, double eps, double bc1, double bc2_sqrt) {
  AdamOut o;
  o.g = gsum + lambda * (double)((w > 0.0) - (w < 0.0));       // (lambda * +-1 is exact: no contraction can change this)
  o.m = __builtin_fma(o.g - m, 1.0 - beta1, m);
  o.v = __builtin_fma((1.0 - beta2) * o.g, o.g, v * beta2);
  const double denom = sqrt(o.v) / bc2_sqrt + eps;
  o.w = w - (lr / bc1) * (o.m / denom);
  return o;
}

}  // namespace lgn
