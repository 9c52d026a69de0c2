// Dev experiment: LDS-DMA fill rate per CU for piece shapes (8 rows x 128 B at a row stride vs 1 KiB contiguous).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// 256 threads; every wave issues `iters` groups of 8 pieces into its 8-KiB LDS slice, waiting vmcnt(0) per group.
// Source: `span` bytes (power of two) shared by all workgroups; piece p of group g starts at a pseudo-random
// 1-KiB-aligned offset; lane l reads 16 B at  base + (l>>3)*row_stride + (l&7)*16.
__global__ __launch_bounds__(256, 2) void fill_rate(const char* src, unsigned span, int row_stride, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, span, 0x00020000);
  const int loff = (lane >> 3) * row_stride + (lane & 7) * 16;
  unsigned h = blockIdx.x * 2654435761u + wave * 40503u;
  char* dst = lds + wave * 8192;
  for (int g = 0; g < iters; ++g) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      h = h * 1664525u + 1013904223u;
      const unsigned base = ((h >> 8) * 1024u) & (span - 1) & ~(unsigned)(8 * row_stride - 1 > 1023 ? 0 : 0);
      unsigned off = base + loff;
      if (off + 16 > span) off -= span / 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + p * 1024), 16, off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0 && lds[0] == 123) sink[0] = 1;
}

extern "C" int run_fill_rate(const char* src, unsigned span, int row_stride, int iters, int grid, int* sink, void* stream) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_rate), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(fill_rate, dim3(grid), dim3(256), 65536, (hipStream_t)stream, src, span, row_stride, iters, sink);
  return (int)hipGetLastError();
}
