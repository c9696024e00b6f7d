// lgn-autoencoder_amd/csrc/mlp.hip -- CGMLP (scalar-irrep MLP) dispatch.
//
// Reference: CGMLP.forward, lgn/models/lgn_levels.py:191-227 -- the (0,0) part (2,B,N,C,1) is viewed as
// (B*N) rows of 2C real features (index 2c+z), passed through Linear(2C->W), 5 x Linear(W->W),
// Linear(W->2C) with the activation (LeakyReLU(0.01) by default; get_activation_fn, lgn/nn/generic_levels.py:119-135) after
// all but the last Linear, and written back.  No mask.
// Both kernels run on the fp64 matrix cores: mlp_mfma.hip (W <= 48) and mlp_mfma_wide.hip (48 < W <= 96).
#include "ops.hpp"
#include "../../include/lgn_amd.h"

namespace lgn {

static size_t mlp_param_count(int C, int H, int nlin) {
  const int D = 2 * C;
  if (nlin == 1) return (size_t)D * D + D;
  return ((size_t)H * D + H) + (size_t)(nlin - 2) * ((size_t)H * H + H) + ((size_t)D * H + D);
}

template <typename T>
int mlp_dispatch(const MlpArgs<T>& a, bool backward, hipStream_t stream) {
  LGN_CHECK_ARG(a.M > 0 && a.C > 0, "cgmlp: empty input (M=%d C=%d)", a.M, a.C);
  LGN_CHECK_ARG(a.nlin >= 4 && a.nlin <= 7, "cgmlp: mlp_depth 3 .. 6 (4 .. 7 Linear layers) are built, got %d layers", a.nlin);
  LGN_CHECK_ARG(a.C <= 8, "cgmlp: %d channels unsupported (1..8)", a.C);
  LGN_CHECK_ARG(a.act >= 0 && a.act < LGN_ACT_COUNT, "cgmlp: unknown activation id %d (LGN_ACT_*)", a.act);
  LGN_CHECK_ARG(a.H >= 2 * a.C && a.H <= 96, "cgmlp: hidden width %d unsupported (2C..96)", a.H);
  if (backward) LGN_CHECK_ARG((size_t)a.psize == mlp_param_count(a.C, a.H, a.nlin), "cgmlp: psize mismatch");
  int rc = mlp_chain_dispatch(a, backward, stream);
  if (rc == -2) rc = mlp_mfma_dispatch(a, backward, stream);
  if (rc == -2) rc = mlp_mfma_wide_dispatch(a, backward, stream);
  if (rc == -2) { set_error("cgmlp: shape C=%d H=%d not covered", a.C, a.H); return -1; }
  return rc;
}

template int mlp_dispatch<double>(const MlpArgs<double>&, bool, hipStream_t);

}  // namespace lgn
