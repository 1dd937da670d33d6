// Per-CU fill-rate microbenchmark (round 1): how fast can ONE workgroup per CU pull a [rows x 256 B]
// panel stream out of L2 / HBM, as a function of (a) LDS-DMA vs register loads, (b) bytes contiguous
// per wave-instruction (row stride), (c) waves issuing.  Build: hipcc --offload-arch=gfx950 -O3 fill_rate.hip -o fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: global_load_lds 16 B/lane into a 64 KiB LDS ring, `waves` waves issue, vmcnt kept at depth
// MODE 1: global_load_dwordx4 into registers (8 in flight per lane), xor-reduced
template <int MODE>
__global__ __launch_bounds__(512) void fill(const char* __restrict__ base, long panel_stride, long row_stride,
                                            int rows, int iters, unsigned* sink, int panels) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* p = base + (long)(blockIdx.x % panels) * panel_stride;
  // a wave-instruction covers 4 rows x 256 B (16 lanes x 16 B per row)
  const int rl = lane >> 4, ch = lane & 15;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    for (int r0 = wave * 4; r0 < rows; r0 += nw * 4 * 8) {
      if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + u * nw * 4;
          if (r < rows) {
            const char* g = p + (long)(r + rl) * row_stride + ch * 16;
            char* d = lds + ((wave * 8 + u) & 63) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)d, 16, 0, 0);
          }
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + u * nw * 4;
          v[u] = make_uint4(0, 0, 0, 0);
          if (r < rows) v[u] = *reinterpret_cast<const uint4*>(p + (long)(r + rl) * row_stride + ch * 16);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[0] = acc + lds[threadIdx.x];
}

int main() {
  const int rows = 2048;                     // tokens
  const int CUs = 256;
  for (int C : {2048, 8192}) {
    const long row_stride = (long)C * 2;     // fp16 activations [rows][C]
    const size_t bytes = (size_t)rows * row_stride;
    char* buf; unsigned* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes)); CK(hipMalloc(&sink, 4));
    const int panels = C / 128;              // 256-byte wide column panels
    for (int mode = 0; mode < 2; ++mode)
      for (int threads : {256, 512}) {
        for (int share : {1, 8}) {           // share = 8: groups of 8 workgroups stream the same panel (L2 reuse)
          hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          const int iters = 4;
          const int np = share == 1 ? panels : (CUs / 8 < panels ? CUs / 8 : panels);
          auto launch = [&]() {
            if (mode == 0) fill<0><<<CUs, threads, 65536>>>(buf, 256, row_stride, rows, iters, sink, np);
            else fill<1><<<CUs, threads, 65536>>>(buf, 256, row_stride, rows, iters, sink, np);
          };
          CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
          CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
          launch(); CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0)); for (int k = 0; k < 5; ++k) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
          const double per_wg = (double)rows * 256 * iters;
          printf("C=%d mode=%s threads=%d distinct_panels=%d : %.1f us, %.1f GB/s per CU, %.2f TB/s chip\n", C,
                 mode == 0 ? "lds-dma" : "regs", threads, np, ms * 1e3, per_wg / (ms * 1e-3) / 1e9,
                 per_wg * CUs / (ms * 1e-3) / 1e12);
        }
      }
    CK(hipFree(buf)); CK(hipFree(sink));
  }
  return 0;
}
