// How many cycles one v_mfma_f32_32x32x16_bf16 (and 16x16x32) takes per SIMD with one, two and four waves per SIMD issuing it
// back to back (four independent accumulators, register operands, random bf16 bits).  Wall clock by HIP events, cycles by s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool M16>
__global__ void k(float* out, unsigned long long* st, int iters) {
  const int tid = threadIdx.x;
  u32x4 ua = {0x3f803f00u + tid * 7u, 0x3f013f20u ^ (tid * 131u & 0x007f007fu), 0x3f7f3f01u, 0x3f333f44u + tid};
  u32x4 ub = {0x3f113f22u + tid * 3u, 0x3f553f66u ^ (tid * 17u & 0x007f007fu), 0x3f773f08u, 0x3f093f0au + tid};
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  float s = 0.f;
  unsigned long long t0, t1;
  if constexpr (M16) {
    f32x4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int e = 0; e < 4; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  } else {
    f32x16 c0, c1, c2, c3;
    for (int e = 0; e < 16; ++e) c0[e] = c1[e] = c2[e] = c3[e] = 0.f;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  }
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) st[blockIdx.x] = t1 - t0;
}

template <bool M16>
static void run(int threads, int blocks, int iters) {
  float* out;
  unsigned long long* st;
  CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
  CHECK(hipMalloc(&st, blocks * 8));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<M16>, dim3(blocks), dim3(threads), 0, 0, out, st, iters);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<M16>, dim3(blocks), dim3(threads), 0, 0, out, st, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h;
  CHECK(hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost));
  const double waves = (double)blocks * threads / 64.0, mfma = waves * 4.0 * iters;
  const double flop = mfma * (M16 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2);
  const double wps = waves / 1024.0;     // waves per SIMD
  printf("  %s  %4d threads x %4d blocks (%.1f waves / SIMD): %7.1f memtime ticks per MFMA per wave, wall %7.3f ms -> %7.1f TFLOP/s dense bf16, %5.1f ns per MFMA per SIMD\n",
         M16 ? "16x16x32" : "32x32x16", threads, blocks, wps, (double)h / (4.0 * iters), ms, flop / (ms * 1e-3) / 1e12, ms * 1e6 / (4.0 * iters * wps));
  CHECK(hipFree(out));
  CHECK(hipFree(st));
}

int main() {
  const int iters = 200000;
  run<false>(256, 256, iters);
  run<false>(512, 256, iters);
  run<false>(1024, 256, iters);
  run<true>(256, 256, iters);
  run<true>(512, 256, iters);
  run<true>(1024, 256, iters);
  return 0;
}
