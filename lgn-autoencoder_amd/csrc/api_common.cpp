// lgn-autoencoder_amd/csrc/api_common.cpp -- error channel + ABI version of liblgn_amd.so
#include <stdarg.h>

#include "common.hpp"
#include "../../include/lgn_amd.h"

namespace lgn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace lgn

extern "C" {
const char* lgn_last_error(void) { return lgn::g_err; }
int lgn_abi_version(void) { return LGN_AMD_ABI_VERSION; }
}
