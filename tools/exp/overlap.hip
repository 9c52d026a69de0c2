// Dev experiment: do LDS-DMA fills overlap with MFMAs / LDS fragment reads issued by the same waves?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bit 0: issue 8 LDS-DMA pieces per iteration; bit 1: 24 MFMAs per iteration; bit 2: 16 ds_read_b128 feeding them
// bit 3: barrier per iteration (after the vmcnt wait)
template <int MODE>
__global__ __launch_bounds__(256, 2) void overlap(const char* src, unsigned span, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, span, 0x00020000);
  const int loff = (lane >> 3) * 3072 + (lane & 7) * 16;
  unsigned h = blockIdx.x * 2654435761u + wave * 40503u;
  char* dst = lds + wave * 8192;              // DMA target: 32 KiB
  const char* rd = lds + 32768 + lane * 16;   // fragment reads: another 32 KiB
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  h16x8 fa[4], fb[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { fa[i][e] = (_Float16)(lane * 0.01f + i); fb[i][e] = (_Float16)(e * 0.1f); }
  for (int g = 0; g < iters; ++g) {
    if (MODE & 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    if (MODE & 1) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        h = h * 1664525u + 1013904223u;
        unsigned off = (((h >> 8) * 1024u) & (span - 1)) + loff;
        if (off + 16 > span) off -= span / 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + p * 1024), 16, off, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (MODE & 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[i] = *reinterpret_cast<const h16x8*>(rd + (ks * 8 + i) * 1024);
          fb[i] = *reinterpret_cast<const h16x8*>(rd + (ks * 8 + 4 + i) * 1024);
        }
      }
      if (MODE & 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i & 1], fb[i >> 1], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i & 1], fb[2 + (i >> 1)], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[2 + (i & 1)], fb[i >> 1], acc[i], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(MODE & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[threadIdx.x] = s + lds[lane];
}

template <int MODE>
static int go(const char* src, unsigned span, int iters, int grid, float* out, hipStream_t st) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&overlap<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(overlap<MODE>, dim3(grid), dim3(256), 65536, st, src, span, iters, out);
  return (int)hipGetLastError();
}
extern "C" int run_overlap(int mode, const char* src, unsigned span, int iters, int grid, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 1: return go<1>(src, span, iters, grid, out, st);
    case 2: return go<2>(src, span, iters, grid, out, st);
    case 3: return go<3>(src, span, iters, grid, out, st);
    case 6: return go<6>(src, span, iters, grid, out, st);
    case 7: return go<7>(src, span, iters, grid, out, st);
    case 15: return go<15>(src, span, iters, grid, out, st);
    case 14: return go<14>(src, span, iters, grid, out, st);
    case 9: return go<9>(src, span, iters, grid, out, st);
    case 11: return go<11>(src, span, iters, grid, out, st);
  }
  return -1;
}
