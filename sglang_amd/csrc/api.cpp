// Error plumbing and identification entry points of libsglang_amd.so.
#include <cstdarg>
#include <cstdio>

#include "sglang_amd.h"

namespace sgl_amd {

static thread_local char g_last_error[512] = "";

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

}  // namespace sgl_amd

extern "C" {

const char* sgl_amd_last_error(void) { return sgl_amd::g_last_error; }
int sgl_amd_abi_version(void) { return SGL_AMD_ABI_VERSION; }
const char* sgl_amd_target_arch(void) { return "gfx950"; }

}  // extern "C"
