// Error reporting and ABI version of libgptq_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace gptq {
SideCtx* side_ctx(hipStream_t main) {
  // one helper stream per (device, caller stream): solves enqueued on different streams stay independent
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SideCtx> table;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(dev, main);
  auto it = table.find(key);
  if (it != table.end()) return &it->second;
  // HIP maps streams onto few hardware queues (4 by default) and streams sharing one serialize: by default only
  // the first caller stream per device gets a helper (GPTQ_SIDE_STREAMS raises that); the others run serially.
  static const int limit = tune_knob("GPTQ_SIDE_STREAMS", 1);
  int mine = 0;
  for (const auto& kv : table) mine += kv.first.first == dev;
  if (mine >= limit) return nullptr;
  SideCtx c;
  // Lowest priority: the helper stream carries the big rank-512 updates (thousands of workgroups) that run underneath
  // the caller's stream, whose kernels are small, serial and on the critical path.  Priority decides who gets a freed
  // slot first; it does not preempt, so those updates are launched as two 67 KB workgroups per compute unit: ONE
  // finishing workgroup then frees enough LDS for any of the caller's kernels (the diagonal factorization needs 66 KB).
  // (Tried: confining the helper stream to a subset of the compute units with a queue CU mask -- every solve got
  // 35-50 % slower, masked queues cost more than the contention they remove; capping the updates' grids at one or two
  // workgroups per compute unit with a tile loop -- no gain at two, slower at one.)
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
  if (hipStreamCreateWithPriority(&c.stream, hipStreamNonBlocking, least) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&c.main_done, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&c.side_done, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&c.prep_done, hipEventDisableTiming) != hipSuccess) return nullptr;
  return &table.emplace(key, c).first->second;
}

void* scratch_buffer(hipStream_t main, size_t bytes) {
  struct Buf { void* p = nullptr; size_t n = 0; };
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, Buf> table;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  Buf& b = table[std::make_pair(dev, main)];
  if (b.n >= bytes) return b.p;
  if (b.p) {
    // work enqueued earlier on this stream may still use the old block
    if (hipStreamSynchronize(main) != hipSuccess || hipFree(b.p) != hipSuccess) return nullptr;
    b = Buf{};
  }
  if (hipMalloc(&b.p, bytes) != hipSuccess) { b = Buf{}; return nullptr; }
  b.n = bytes;
  return b.p;
}

int lookahead_mask() {
  static int mask = [] {
    const char* e = getenv("GPTQ_LOOKAHEAD");
    return e ? atoi(e) : 3;
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

extern "C" int gptq_hip_abi_version(void) { return 2; }
extern "C" const char* gptq_last_error(void) { return gptq::g_err; }
