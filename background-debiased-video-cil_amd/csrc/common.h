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

// ---- activation storage ------------------------------------------------------------------------------------------------
// ES = bytes per activation element: 4 = fp32 (default everywhere), 2 = bf16 storage (BDV_ACT_BF16: BASELINE config 5, the
// reduced-precision mode).  A bf16 value widens to fp32 exactly (16-bit shift); a store rounds to nearest-even
// (v_cvt_pk_bf16_f32).  Tensors are addressed in groups of 4 consecutive elements: 16-byte (fp32) or 8-byte (bf16) accesses.
typedef __bf16 bdv_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float bdv_f32x2_t __attribute__((ext_vector_type(2)));
typedef float bdv_f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned bdv_u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bdv_pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((bdv_f32x2_t){a, b}, bdv_bf16x2_t));
}
__device__ __forceinline__ float4 bdv_widen_bf16x4(unsigned lo, unsigned hi) {
  return make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                     __uint_as_float(hi & 0xffff0000u));
}
template <int ES, bool NT = false>
__device__ __forceinline__ float4 act_ld4(const void* __restrict__ base, int64_t i4) {
  if constexpr (ES == 4) {
    const float4* p = reinterpret_cast<const float4*>(base) + i4;
    if constexpr (NT) {
      const bdv_f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const bdv_f32x4_t*>(p));
      return make_float4(v.x, v.y, v.z, v.w);
    } else {
      return *p;
    }
  } else {
    const bdv_u32x2_t* p = reinterpret_cast<const bdv_u32x2_t*>(base) + i4;
    bdv_u32x2_t v;
    if constexpr (NT) v = __builtin_nontemporal_load(p);
    else v = *p;
    return bdv_widen_bf16x4(v.x, v.y);
  }
}
template <int ES>
__device__ __forceinline__ void act_st4(void* __restrict__ base, int64_t i4, const float4 v) {
  if constexpr (ES == 4) {
    reinterpret_cast<float4*>(base)[i4] = v;
  } else {
    reinterpret_cast<bdv_u32x2_t*>(base)[i4] = (bdv_u32x2_t){bdv_pack_bf16x2(v.x, v.y), bdv_pack_bf16x2(v.z, v.w)};
  }
}
// A thread's 16-byte unit of an elementwise pass: ACT_U<ES> groups of 4 consecutive elements (1 for fp32, 2 for bf16).
#ifndef BDV_F32_UNITS
#define BDV_F32_UNITS 1      // channel groups (float4) per lane and iteration in the fp32 elementwise passes.  2 was measured
                             // (tools/bench_bn.py, two libraries in one call): apply passes 6.49 -> 6.62 ms, backward 12.17 -> 13.11 ms
                             // over the R50 sizes, 590.9 -> 572.2 clips/s in the step; only the 822 MB layer-1 pass with a residual,
                             // which misses the Infinity Cache entirely, gained (613 -> 539 us)
#endif
template <int ES> struct ActU { static constexpr int value = ES == 2 ? 2 : BDV_F32_UNITS; };
template <int ES, bool NT = false>
__device__ __forceinline__ void act_ld16(const void* __restrict__ base, int64_t iu, float4 (&v)[ActU<ES>::value]) {
  if constexpr (ES == 4) {
#pragma unroll
    for (int u = 0; u < ActU<4>::value; ++u) v[u] = act_ld4<4, NT>(base, iu * ActU<4>::value + u);
  } else {
    typedef unsigned u32x4_t_ __attribute__((ext_vector_type(4)));
    const u32x4_t_* p = reinterpret_cast<const u32x4_t_*>(base) + iu;
    u32x4_t_ r;
    if constexpr (NT) r = __builtin_nontemporal_load(p);
    else r = *p;
    v[0] = bdv_widen_bf16x4(r.x, r.y);
    v[1] = bdv_widen_bf16x4(r.z, r.w);
  }
}
template <int ES>
__device__ __forceinline__ void act_st16(void* __restrict__ base, int64_t iu, const float4 (&v)[ActU<ES>::value]) {
  if constexpr (ES == 4) {
#pragma unroll
    for (int u = 0; u < ActU<4>::value; ++u) reinterpret_cast<float4*>(base)[iu * ActU<4>::value + u] = v[u];
  } else {
    typedef unsigned u32x4_t_ __attribute__((ext_vector_type(4)));
    reinterpret_cast<u32x4_t_*>(base)[iu] = (u32x4_t_){bdv_pack_bf16x2(v[0].x, v[0].y), bdv_pack_bf16x2(v[0].z, v[0].w),
                                                      bdv_pack_bf16x2(v[1].x, v[1].y), bdv_pack_bf16x2(v[1].z, v[1].w)};
  }
}
// host-side dispatch on bdv act_dtype: BDV_ACT_SWITCH(dt, ES, launch<ES>(...))
#define BDV_ACT_SWITCH(dt, ES, ...)  \
  do {                               \
    if ((dt) == BDV_ACT_BF16) {      \
      constexpr int ES = 2;          \
      __VA_ARGS__;                   \
    } else {                         \
      constexpr int ES = 4;          \
      __VA_ARGS__;                   \
    }                                \
  } while (0)
#define BDV_REQUIRE_ACT(dt, name) BDV_REQUIRE((dt) == BDV_ACT_F32 || (dt) == BDV_ACT_BF16, "%s: act_dtype %d (BDV_ACT_F32 = 0 | BDV_ACT_BF16 = 1)", name, (int)(dt))

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

template <int ES = 4>
__device__ __forceinline__ void pool_bwd_gather2x2(const void* __restrict__ dout, const uchar4* __restrict__ idx, int64_t o00,
                                                   int CV, int Wo, bool right, bool down, float4 (&g)[4]) {
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const uchar4 none = make_uchar4(255, 255, 255, 255);
  const uchar4 t00 = idx[o00];
  const float4 d00 = act_ld4<ES>(dout, o00);
  const uchar4 t01 = right ? idx[o00 + CV] : none;
  const float4 d01 = right ? act_ld4<ES>(dout, o00 + CV) : z4;
  const uchar4 t10 = down ? idx[o00 + Wo * CV] : none;
  const float4 d10 = down ? act_ld4<ES>(dout, o00 + Wo * CV) : z4;
  const uchar4 t11 = (down && right) ? idx[o00 + (Wo + 1) * CV] : none;
  const float4 d11 = (down && right) ? act_ld4<ES>(dout, o00 + (Wo + 1) * CV) : z4;
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
