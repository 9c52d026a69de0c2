// Shared helpers for the vfml HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../../include/vfml.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef _Float16 vfml_h16x2 __attribute__((ext_vector_type(2)));

// (a, b) -> hi = round-to-nearest f16 pair (v_cvt_pk_f16_f32), lo = f16(x - hi): the two halves of a split-row
// channel (VFML_FMT_S16).  Nearest, not toward zero: a consumer that drops the lo half of an operand
// (VFML_CONV_MFMA2 / _MFMA1) then sees an UNBIASED f16 rounding of it (2^-12 relative), and |lo| <= half an ulp of
// hi.  The empty asm keeps hipcc from redoing the two conversions element by element for the subtraction.
__device__ __forceinline__ void vfml_split2(float a, float b, vfml_h16x2& h, vfml_h16x2& l) {
  vfml_h16x2 hh = {(_Float16)a, (_Float16)b};
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned hp = __builtin_bit_cast(unsigned, hh);
  asm volatile("" : "+v"(hp));
  hh = __builtin_bit_cast(vfml_h16x2, hp);
#endif
  h = hh;
  const vfml_h16x2 ll = {(_Float16)(a - (float)hh[0]), (_Float16)(b - (float)hh[1])};
  l = ll;
}

// A double read from LDS whose FIRST reader is a 32-bit VALU move per half, not the 64-bit-operand consumer itself.
// Round 2's two-stream finding (profiles/r02_kernel_anatomy.md section 7): a packed-f32 op - a 64-bit register-pair operand
// read straight out of a ds_read result - saw stale lanes 48-63 beside another stream's MFMA kernels.  v_add_f64 takes its
// operand the same way, and the norm statistics' LDS folds run on the prefetch stream beside the iterations' MFMA kernels;
// no disturbance of them was ever observed, but the cause is not established, so no 64-bit-operand VALU op of this library
// is the first reader of an LDS result (tests/test_abi.py scans the code object for both forms).  Two v_mov_b32 per fold
// term: nothing, next to the LDS round trip it follows.
__device__ __forceinline__ double vfml_lds_f64(const double* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  unsigned lo, hi;
  // (not `volatile`: a pure function of its input - the compiler may keep many LDS reads in flight ahead of the moves; as
  // volatile statements a 256-term fold ran one LDS latency per term, 20 us instead of 6 for instnorm_final_kernel)
  asm("v_mov_b32 %0, %1" : "=v"(lo) : "v"(u.x));
  asm("v_mov_b32 %0, %1" : "=v"(hi) : "v"(u.y));
  return __hiloint2double((int)hi, (int)lo);
#else
  return *p;
#endif
}

void vfml_set_error(const char* fmt, ...);

#define VFML_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      vfml_set_error(__VA_ARGS__);         \
      return 1;                            \
    }                                      \
  } while (0)

static inline int vfml_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vfml_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

static inline bool vfml_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
