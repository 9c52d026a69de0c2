// How fast does one SIMD retire v_mfma_f32_16x16x32_f16 when consecutive MFMAs accumulate into the SAME tile (the split
// product's three terms) versus different tiles?  One or two waves per SIMD, no memory traffic, operands in registers
// (random bits).  Prints shader cycles per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain tools/exp/mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: groups of 3 on one accumulator (24 accumulators), 1: round robin over 24, 2: 32x32x16 groups of 3 (6 acc)
__global__ __launch_bounds__(256, 2) void chain(const h16x8* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
  h16x8 a0 = src[threadIdx.x], a1 = src[threadIdx.x + 256], b[8];
  for (int j = 0; j < 8; ++j) b[j] = src[threadIdx.x + 512 + 256 * j];
  f32x4 acc[24];
  f32x16 big[6];
  for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) big[i][e] = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[i & 3], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[4 + (i & 3)], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[i & 3], acc[i], 0, 0, 0);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 24; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k == 2 ? a1 : a0, b[(k == 1 ? 4 : 0) + (i & 3)], acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          big[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[i & 3], big[i], 0, 0, 0);
          big[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[4 + (i & 3)], big[i], 0, 0, 0);
          big[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[i & 3], big[i], 0, 0, 0);
        }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 6; ++i) s += big[i][0] + big[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const h16x8* src, float* out, unsigned long long* cyc, int blocks, int mfma_per_iter, double weight) {
  const int iters = 2000;
  hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double per = (double)h[0] / ((double)iters * mfma_per_iter);
  const int wps = blocks / 256;   // waves per SIMD
  printf("%-44s %d wave(s)/SIMD: %6.2f cycles per MFMA per wave = %6.2f per SIMD (x%.0f: %5.1f %% of the pipe), %.3f GHz\n", name, wps, per,
         per / wps, weight, 100.0 * weight / (per / wps), (double)h[0] / (ms * 1e6));
}

int main() {
  h16x8* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 256 * 10 * sizeof(h16x8)); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8);
  _Float16* h = (_Float16*)malloc(256 * 10 * 16);
  for (int i = 0; i < 256 * 10 * 8; ++i) h[i] = (_Float16)((rand() / (double)RAND_MAX - 0.5) * 2.0);
  hipMemcpy(src, h, 256 * 10 * 16, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {
    run<0>("16x16x32, three in a row per accumulator", src, out, cyc, blocks, 72, 16.0);
    run<1>("16x16x32, round robin over 24 accumulators", src, out, cyc, blocks, 72, 16.0);
    run<2>("32x32x16, three in a row per accumulator", src, out, cyc, blocks, 36, 32.0);
  }
  return 0;
}
