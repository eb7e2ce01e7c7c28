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

// ---- MaxPool2d(3, 2, 1) backward as a gather (shared by maxpool_bwd and the fused stem BatchNorm backward) ----------
// The 2x2 input block (2i..2i+1, 2j..2j+1) of channels c4 is covered by exactly the windows (i,j), (i,j+1), (i+1,j),
// (i+1,j+1); a pixel takes a window's gradient when the stored arg-max code equals its position r*3+s inside that
// window.  Contributions are added in (r, s) scan order.  g[0..3] = pixels (2i,2j), (2i,2j+1), (2i+1,2j), (2i+1,2j+1).
__device__ __forceinline__ void pool_take(float4& g, const uchar4 t, const float4 d, unsigned char me) {
  if (t.x == me) g.x += d.x;
  if (t.y == me) g.y += d.y;
  if (t.z == me) g.z += d.z;
  if (t.w == me) g.w += d.w;
}

__device__ __forceinline__ void pool_bwd_gather2x2(const float4* __restrict__ dout, const uchar4* __restrict__ idx, int64_t o00,
                                                   int CV, int Wo, bool right, bool down, float4 (&g)[4]) {
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const uchar4 none = make_uchar4(255, 255, 255, 255);
  const uchar4 t00 = idx[o00];
  const float4 d00 = dout[o00];
  const uchar4 t01 = right ? idx[o00 + CV] : none;
  const float4 d01 = right ? dout[o00 + CV] : z4;
  const uchar4 t10 = down ? idx[o00 + Wo * CV] : none;
  const float4 d10 = down ? dout[o00 + Wo * CV] : z4;
  const uchar4 t11 = (down && right) ? idx[o00 + (Wo + 1) * CV] : none;
  const float4 d11 = (down && right) ? dout[o00 + (Wo + 1) * CV] : z4;
  g[0] = g[1] = g[2] = g[3] = z4;
  pool_take(g[0], t00, d00, 4);  // (2i, 2j): centre of window (i,j)
  pool_take(g[1], t01, d01, 3);  // (2i, 2j+1): (r=1,s=0) of (i,j+1), then (1,2) of (i,j)
  pool_take(g[1], t00, d00, 5);
  pool_take(g[2], t10, d10, 1);  // (2i+1, 2j): (0,1) of (i+1,j), then (2,1) of (i,j)
  pool_take(g[2], t00, d00, 7);
  pool_take(g[3], t11, d11, 0);  // (2i+1, 2j+1): (0,0) of (i+1,j+1), (0,2) of (i+1,j), (2,0) of (i,j+1), (2,2) of (i,j)
  pool_take(g[3], t10, d10, 2);
  pool_take(g[3], t01, d01, 6);
  pool_take(g[3], t00, d00, 8);
}
