// Shared helpers for the gfx950 GPTQ kernels (internal; the public C ABI is include/gptq_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/gptq_hip.h"

namespace gptq {

// Last error text (thread-local), exposed through gptq_last_error().
void set_error(const char* fmt, ...);

#define GPTQ_CHECK_ARG(cond, ...)                                   \
  do {                                                              \
    if (!(cond)) {                                                  \
      ::gptq::set_error(__VA_ARGS__);                               \
      return GPTQ_ERR_INVALID;                                      \
    }                                                               \
  } while (0)

#define GPTQ_CHECK_LAUNCH(what)                                     \
  do {                                                              \
    hipError_t e_ = hipGetLastError();                              \
    if (e_ != hipSuccess) {                                         \
      ::gptq::set_error("%s: %s", what, hipGetErrorString(e_));     \
      return GPTQ_ERR_HIP;                                          \
    }                                                               \
  } while (0)

#define GPTQ_CHECK_HIP(expr)                                        \
  do {                                                              \
    hipError_t e_ = (expr);                                         \
    if (e_ != hipSuccess) {                                         \
      ::gptq::set_error("%s: %s", #expr, hipGetErrorString(e_));    \
      return GPTQ_ERR_HIP;                                          \
    }                                                               \
  } while (0)

// Per-(device, caller stream) helper stream + events for look-ahead overlap inside one library call (created once,
// reused; the caller's stream stays the only externally visible ordering point).
struct SideCtx {
  hipStream_t stream = nullptr;
  hipEvent_t main_done = nullptr;   // recorded on the caller's stream
  hipEvent_t side_done = nullptr;   // recorded on the side stream
  hipEvent_t prep_done = nullptr;   // recorded on the side stream: input preparation that runs beside the factorization
};
SideCtx* side_ctx(hipStream_t main);   // nullptr if it cannot be created (callers then run serially)
// Library-owned device scratch per (device, caller stream), grown on demand and kept (nullptr on failure).
void* scratch_buffer(hipStream_t main, size_t bytes);
int lookahead_mask();               // bit 0: factorization chain, bit 1: column loop (env GPTQ_LOOKAHEAD)

// Kernels of the caller's stream in the solve (the serial chain) raise their waves' issue priority over the helper
// stream's far-update GEMMs that share their SIMDs (s_setprio: 0 = default ... 3).
#if defined(__HIPCC__)
__device__ __forceinline__ void critical_path_priority() { __builtin_amdgcn_s_setprio(3); }
#endif

// Tuning knobs that were measured and settled (each is documented where it is used) are compile-time constants in the
// product library; only the diagnostic library (-DGPTQ_DIAG) still reads them from the environment, for re-measuring.
#ifdef GPTQ_DIAG
static inline int tune_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int tune_knob(const char*, int dflt) { return dflt; }
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-provided workspace.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
  size_t used() const { return align_up(off, 256); }
};

}  // namespace gptq
