// Error reporting and ABI version of libgptq_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace gptq {
SideCtx* side_ctx() {
  static std::mutex mu;
  static SideCtx ctx[64];
  static bool ready[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!ready[dev]) {
    SideCtx c;
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c.main_done, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c.side_done, hipEventDisableTiming) != hipSuccess) return nullptr;
    ctx[dev] = c;
    ready[dev] = true;
  }
  return &ctx[dev];
}

int lookahead_mask() {
  static int mask = [] {
    const char* e = getenv("GPTQ_LOOKAHEAD");
    return e ? atoi(e) : 2;
  }();
  return mask;
}

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
