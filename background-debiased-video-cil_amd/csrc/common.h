// Shared helpers for the bdvcil HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "bdvcil_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

void bdv_set_error(const char* fmt, ...);

#define BDV_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      bdv_set_error(__VA_ARGS__);     \
      return BDV_EINVAL;              \
    }                                 \
  } while (0)

#define BDV_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      bdv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

static inline bool bdv_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// wave-level and block-level sum (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
