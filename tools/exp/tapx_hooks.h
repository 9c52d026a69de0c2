// In-kernel time stamps of conv_gemm_tapx_kernel for tools/exp/tapx_stamps.py (hipcc ... -include tools/exp/tapx_hooks.h):
// per phase of a K step the s_memtime cycles of one mid-grid workgroup's waves, and for every workgroup the s_memrealtime
// (100 MHz) of entry / K-loop start / K-loop end / exit.  The stamp values go to buffers nothing else reads; no output value
// depends on them.  (The product build defines these hooks empty: video-flow-ml_amd/vfml/csrc/conv_gemm_tapx.hip.)
#pragma once
#define VFML_TAPX_HOOKS 1
#include <hip/hip_runtime.h>
__device__ unsigned long long vfml_tapx_stamps[64];
__device__ unsigned long long vfml_tapx_timeline[4096 * 4];
extern "C" int vfml_debug_tapx_timeline(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(vfml_tapx_timeline), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : 1;
}
extern "C" int vfml_debug_tapx_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(vfml_tapx_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1;
}
#define TAPX_STAMP_(x)                                     \
  do {                                                     \
    x = __builtin_readcyclecounter();                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
  } while (0)
#define TAPX_HOOK_ENTRY() const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime()
#define TAPX_HOOK_KLOOP_BEGIN()                                                                                   \
  unsigned long long hc[5] = {0, 0, 0, 0, 0}, sX = 0, sM = 0, sW = 0, sY = 0;                                      \
  const unsigned long long k0c = __builtin_readcyclecounter(), k0r = __builtin_amdgcn_s_memrealtime()
// optional scheduling experiments of round 2 (none moved the launch by more than 1 %): -DVFML_TAPX_ALTPRIO=n lets the two
// workgroups of a CU take turns at the higher issue priority every n steps
#ifdef VFML_TAPX_ALTPRIO
#define TAPX_HOOK_STEP_BEGIN(st)                                                                                   \
  do {                                                                                                              \
    if ((((st) / VFML_TAPX_ALTPRIO) ^ (blockIdx.x >> 8)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); \
    TAPX_STAMP_(hc[0]);                                                                                             \
  } while (0)
#else
#define TAPX_HOOK_STEP_BEGIN(st) TAPX_STAMP_(hc[0])
#endif
#define TAPX_HOOK_STAMP(k) TAPX_STAMP_(hc[k])
#define TAPX_HOOK_STEP_END()                                                                         \
  do {                                                                                               \
    TAPX_STAMP_(hc[4]);                                                                              \
    sX += hc[1] - hc[0]; sM += hc[2] - hc[1]; sW += hc[3] - hc[2]; sY += hc[4] - hc[3];              \
  } while (0)
#define TAPX_HOOK_KLOOP_END(nsteps)                                                                                       \
  const unsigned long long tl_kend = __builtin_amdgcn_s_memrealtime();                                                    \
  if (lane == 0 && blockIdx.x == gridDim.x / 2 + 3) {                                                                     \
    vfml_tapx_stamps[wave * 8 + 0] = sX; vfml_tapx_stamps[wave * 8 + 1] = sM; vfml_tapx_stamps[wave * 8 + 2] = sW;        \
    vfml_tapx_stamps[wave * 8 + 3] = sY; vfml_tapx_stamps[wave * 8 + 4] = (unsigned long long)(nsteps);                   \
    vfml_tapx_stamps[wave * 8 + 5] = __builtin_readcyclecounter() - k0c;                                                  \
    vfml_tapx_stamps[wave * 8 + 6] = __builtin_amdgcn_s_memrealtime() - k0r;                                              \
  }
#define TAPX_HOOK_EXIT()                                                          \
  if (t == 0 && blockIdx.x < 4096) {                                              \
    vfml_tapx_timeline[blockIdx.x * 4 + 0] = tl_entry;                            \
    vfml_tapx_timeline[blockIdx.x * 4 + 1] = k0r;                                 \
    vfml_tapx_timeline[blockIdx.x * 4 + 2] = tl_kend;                             \
    vfml_tapx_timeline[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();    \
  }
