// placeholder: second-generation fp32 MFMA GEMM variants under evaluation (tools/microbench/gemm_f32_bench.hip)
#pragma once
#include "gemm_f32.h"
namespace gptq {
constexpr int GEMM2_VARIANTS = 0;
inline const char* gemm2_name(int) { return ""; }
inline void gemm2_launch(int, float*, int, const float*, int, const float*, int, int, int, int, bool, int, hipStream_t) {}
}  // namespace gptq
