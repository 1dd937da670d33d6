// Error reporting and ABI version of libgptq_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/gptq_hip.h"

namespace gptq {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gptq

extern "C" int gptq_hip_abi_version(void) { return 1; }
extern "C" const char* gptq_last_error(void) { return gptq::g_err; }
