// Shared helpers for the vfml HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../../include/vfml.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
