// lgn-autoencoder_amd/csrc/api_common.cpp -- error channel + ABI version of liblgn_amd.so
#include <stdarg.h>
#include <stdlib.h>

#include "level.hpp"
#include "../../include/lgn_amd.h"

namespace lgn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
// The per-operator entry points (lgn_level_*, lgn_moments_*) have no descriptor: they read the debug switches here, ONCE per call,
// and hand the bits down.  Whole-network calls never come here: their bits are frozen in lgn_net_desc.flags (lgn/_native.py).
int level_flags_from_env() {
  auto on = [](const char* name) { const char* e = getenv(name); return e && e[0] == '1'; };
  return (on("LGN_AMD_DEC_PAIRWISE") ? LVL_DEC_PAIRWISE : 0) | (on("LGN_AMD_LEVEL_V2") ? LVL_LEVEL_V2 : 0) |
         (on("LGN_AMD_MOMENTS_V1") ? LVL_MOMENTS_V1 : 0) | (on("LGN_AMD_MLP_V1") ? LVL_MLP_V1 : 0) | (on("LGN_AMD_BWD_ORDERED") ? LVL_BWD_ORDERED : 0) |
         (on("LGN_AMD_MOMENTS_SPLIT") ? LVL_MOMENTS_SPLIT : 0) | (on("LGN_AMD_MLP_BWD1") ? LVL_MLP_BWD1 : 0);
}
static_assert(LVL_DEC_PAIRWISE == LGN_NET_DEC_PAIRWISE && LVL_LEVEL_V2 == LGN_NET_LEVEL_V2 && LVL_MOMENTS_V1 == LGN_NET_MOMENTS_V1 &&
                  LVL_MLP_V1 == LGN_NET_MLP_V1 && LVL_MLP_BWD1 == LGN_NET_MLP_BWD1 &&
                  LVL_BWD_ORDERED == LGN_NET_BWD_ORDERED && LVL_MOMENTS_SPLIT == LGN_NET_MOMENTS_SPLIT, "LVL_* and LGN_NET_* are the same bits");
}  // namespace lgn

extern "C" {
const char* lgn_last_error(void) { return lgn::g_err; }
int lgn_abi_version(void) { return LGN_AMD_ABI_VERSION; }
}
