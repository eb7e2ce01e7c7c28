// Microbenchmark: LDS-fed fp32 MFMA loop as in conv_mfma.hip's mma_stage, no global traffic in the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 16;
template <int LDA, int LDB, int TM, int TN, bool PIN>
__device__ __forceinline__ void mma_stage(const float* __restrict__ As, const float* __restrict__ Bs, f32x16 (&acc)[TM][TN], int wm0, int wn0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* ap = As + h * LDA + wm0 + r;
  const float* bp = Bs + h * LDB + wn0 + r;
  float a[2][TM], b[2][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a[0][i] = ap[32 * i];
#pragma unroll
  for (int j = 0; j < TN; ++j) b[0][j] = bp[32 * j];
#pragma unroll
  for (int s = 0; s < BK / 2; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < BK / 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[nxt][i] = ap[2 * (s + 1) * LDA + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[nxt][j] = bp[2 * (s + 1) * LDB + 32 * j];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    if (PIN) {
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    }
  }
}
// MODE 0: LDS->MFMA only; MODE 1: + one __syncthreads per K-step; MODE 2: registers only (no LDS reads)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  constexpr int BM = 128, BN = 128, LDA = BM + 4, LDB = BN + 4;
  __shared__ float smem[2 * BK * (LDA + LDB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * BK * (LDA + LDB); i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wm0 = (wave / 2) * 64, wn0 = (wave % 2) * 64;
  for (int it = 0; it < iters; ++it) {
    const float* As = smem + (it & 1) * BK * (LDA + LDB);
    if (MODE == 2) {
      float a0 = As[lane], b0 = As[lane + 64];
#pragma unroll
      for (int s = 0; s < 8; ++s)
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[i][j], 0, 0, 0);
    } else {
      mma_stage<LDA, LDB, 2, 2, true>(As, As + BK * LDA, acc, wm0, wn0, lane);
      if (MODE == 1) __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * 256 + tid] = s;
}
// GEMM-like loop: per K-step 4 x 16-byte global loads per thread (register-staged), transposing LDS store,
// one barrier; VARIANT 0 = loads issued before the MFMA stage (as conv_mfma.hip), 1 = single LDS buffer + 2 barriers
template <int VARIANT>
__global__ __launch_bounds__(256) void kg(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters, int ld) {
  constexpr int BM = 128, BN = 128, LDA = BM + 4, LDB = BN + 4, STAGE = BK * (LDA + LDB);
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  extern __shared__ float dynpad[];
  if (iters < 0) out[0] = dynpad[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / 2) * 64, wn0 = (wave % 2) * 64;
  const int arow = tid >> 2, kg_ = tid & 3;
  const float* ap = A + (size_t)(blockIdx.x % 512) * 128 * ld + (size_t)arow * ld + 4 * kg_;
  const float* bp = B + (size_t)(blockIdx.x % 16) * 128 * ld + (size_t)arow * ld + 4 * kg_;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 ra[2], rb[2];
  auto load = [&](int kt) {
    ra[0] = *reinterpret_cast<const float4*>(ap + kt * 16); ra[1] = *reinterpret_cast<const float4*>(ap + kt * 16 + 64 * (size_t)ld);
    rb[0] = *reinterpret_cast<const float4*>(bp + kt * 16); rb[1] = *reinterpret_cast<const float4*>(bp + kt * 16 + 64 * (size_t)ld);
  };
  auto store = [&](int buf) {
    float* As = smem + buf * STAGE; float* Bs = As + BK * LDA;
    for (int p = 0; p < 2; ++p) {
      float* d = As + (4 * kg_) * LDA + arow + 64 * p; d[0] = ra[p].x; d[LDA] = ra[p].y; d[2 * LDA] = ra[p].z; d[3 * LDA] = ra[p].w;
      float* e = Bs + (4 * kg_) * LDB + arow + 64 * p; e[0] = rb[p].x; e[LDB] = rb[p].y; e[2 * LDB] = rb[p].z; e[3 * LDB] = rb[p].w;
    }
  };
  load(0); store(0); __syncthreads();
  for (int kt = 0; kt < iters; ++kt) {
    const int cur = kt & 1;
    load(kt + 1);
    const float* As = smem + cur * STAGE;
    mma_stage<LDA, LDB, 2, 2, true>(As, As + BK * LDA, acc, wm0, wn0, lane);
    store(cur ^ 1);
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * 256 + tid] = s;
}
void rung(const char* name, int blocks, int iters, float* A, float* B, float* d, int ld, int dyn = 0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kg<0>, dim3(blocks), dim3(256), dyn, 0, A, B, d, iters, ld); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kg<0>, dim3(blocks), dim3(256), dyn, 0, A, B, d, iters, ld);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = (double)blocks * iters * 128.0 * 128.0 * 16 * 2;
  printf("%-28s blocks %5d (%.1f/CU): %.3f ms  %.1f TFLOP/s\n", name, blocks, blocks / 256.0, ms, flops / ms / 1e9);
}
template <int MODE>
void run(const char* name, int blocks, int iters, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = (double)blocks * iters * 128.0 * 128.0 * 16 * 2;
  printf("%-28s blocks %5d (%.1f/CU): %.3f ms  %.1f TFLOP/s\n", name, blocks, blocks / 256.0, ms, flops / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 8192 * 256 * 4);
  for (int b : {256, 512, 768, 1024}) {
    run<2>("regs only", b, 400, d);
    run<0>("LDS->MFMA", b, 400, d);
    run<1>("LDS->MFMA + barrier", b, 400, d);
  }
  const int ld = 2304 + 16, iters = 144;
  float *A, *B; hipMalloc(&A, (size_t)512 * 128 * ld * 4); hipMalloc(&B, (size_t)16 * 128 * ld * 4);
  hipMemset(A, 0, (size_t)512 * 128 * ld * 4); hipMemset(B, 0, (size_t)16 * 128 * ld * 4);
  for (int b : {256, 512, 768, 1024, 2048}) rung("global->reg->LDS->MFMA", b, iters, A, B, d, ld);
  hipFuncSetAttribute((const void*)kg<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 120000);
  for (int b : {1024, 2048, 3072}) rung("  same, capped 3 blocks/CU", b, iters, A, B, d, ld, 16000);
  for (int b : {1024, 2048, 3072}) rung("  same, capped 2 blocks/CU", b, iters, A, B, d, ld, 40000);
  for (int b : {1024, 2048}) rung("  same, capped 1 block/CU", b, iters, A, B, d, ld, 60000);
  return 0;
}
