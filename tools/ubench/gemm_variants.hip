// Microbenchmark of main-loop structures for the fp32-MFMA implicit GEMM (plain GEMM addressing, random data).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BK, int LDA, int LDB, int TM, int TN>
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
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
  }
}

// BK: K-step; NBUF: LDS stages (1: store after a barrier, 2 barriers per step; 2: one barrier per step)
// WM x WN waves (256 or 512 threads), block tile BM x BN, A and B both K-contiguous (row stride ld).
template <int BM, int BN, int WM, int WN, int BK, int NBUF>
__global__ __launch_bounds__(WM * WN * 64) void kg(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters, int ld) {
  constexpr int NT = WM * WN * 64;
  constexpr int LPR = BK / 4;                    // lanes per row
  constexpr int RPP = NT / LPR;                  // rows per pass
  constexpr int AP = BM / RPP, BP = BN / RPP;
  constexpr int LDA = BM + ((LPR == 4) ? 4 : 1), LDB = BN + ((LPR == 4) ? 4 : 1);
  constexpr int STAGE = BK * (LDA + LDB);
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int arow = tid / LPR, kg_ = tid % LPR;
  const float* ap = A + (size_t)(blockIdx.x % 256) * BM * ld + (size_t)arow * ld + 4 * kg_;
  const float* bp = B + (size_t)(blockIdx.x % 8) * BN * ld + (size_t)arow * ld + 4 * kg_;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 ra[AP], rb[BP];
  auto load = [&](int kt) {
#pragma unroll
    for (int p = 0; p < AP; ++p) ra[p] = *reinterpret_cast<const float4*>(ap + kt * BK + (size_t)p * RPP * ld);
#pragma unroll
    for (int p = 0; p < BP; ++p) rb[p] = *reinterpret_cast<const float4*>(bp + kt * BK + (size_t)p * RPP * ld);
  };
  auto store = [&](int buf) {
    float* As = smem + buf * STAGE; float* Bs = As + BK * LDA;
#pragma unroll
    for (int p = 0; p < AP; ++p) { float* d = As + (4 * kg_) * LDA + arow + RPP * p; d[0] = ra[p].x; d[LDA] = ra[p].y; d[2 * LDA] = ra[p].z; d[3 * LDA] = ra[p].w; }
#pragma unroll
    for (int p = 0; p < BP; ++p) { float* e = Bs + (4 * kg_) * LDB + arow + RPP * p; e[0] = rb[p].x; e[LDB] = rb[p].y; e[2 * LDB] = rb[p].z; e[3 * LDB] = rb[p].w; }
  };
  if (NBUF == 2) {
    load(0); store(0); __syncthreads();
    for (int kt = 0; kt < iters; ++kt) {
      const int cur = kt & 1;
      load(kt + 1);
      const float* As = smem + cur * STAGE;
      mma_stage<BK, LDA, LDB, TM, TN>(As, As + BK * LDA, acc, wm0, wn0, lane);
      store(cur ^ 1);
      __syncthreads();
    }
  } else {
    load(0);
    for (int kt = 0; kt < iters; ++kt) {
      __syncthreads();
      store(0);
      __syncthreads();
      load(kt + 1);
      mma_stage<BK, LDA, LDB, TM, TN>(smem, smem + BK * LDA, acc, wm0, wn0, lane);
    }
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[(size_t)blockIdx.x * NT + tid] = s;
}

template <int BM, int BN, int WM, int WN, int BK, int NBUF>
void run(const char* name, int K, float* A, float* B, float* d, int ld) {
  constexpr int LPR = BK / 4;
  constexpr int LDA = BM + ((LPR == 4) ? 4 : 1), LDB = BN + ((LPR == 4) ? 4 : 1);
  const size_t lds = (size_t)NBUF * BK * (LDA + LDB) * 4;
  auto kern = kg<BM, BN, WM, WN, BK, NBUF>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, WM * WN * 64, lds);
  const int iters = K / BK;
  for (int mult : {2, 3, 4, 6, 12}) {
    const int blocks = 256 * mult;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WM * WN * 64), lds, 0, A, B, d, iters, ld); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(WM * WN * 64), lds, 0, A, B, d, iters, ld);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * K * BM * BN * 2.0;
    printf("%-34s lds %6zu occ(api) %d  blocks %5d: %.3f ms  %.1f TFLOP/s\n", name, lds, occ, blocks, ms, flops / ms / 1e9);
  }
}

int main() {
  const int K = 2304, ld = K + 64;
  float *A, *B, *d; 
  size_t na = (size_t)256 * 256 * ld, nb = (size_t)8 * 256 * ld;
  hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&d, (size_t)4096 * 512 * 4);
  float* h = (float*)malloc(na * 4);
  for (size_t i = 0; i < na; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(A, h, na * 4, hipMemcpyHostToDevice); hipMemcpy(B, h, nb * 4, hipMemcpyHostToDevice);
  run<128, 128, 2, 2, 64, 1>("128x128 BK64 1buf", K, A, B, d, ld);
  run<256, 128, 4, 2, 64, 1>("256x128 8w(64x64) BK64 1buf", K, A, B, d, ld);
  run<128, 128, 2, 2, 16, 2>("128x128 BK16 2buf (old)", K, A, B, d, ld);
  run<128, 128, 2, 2, 32, 1>("128x128 BK32 1buf", K, A, B, d, ld);
  run<128, 128, 2, 2, 32, 2>("128x128 BK32 2buf", K, A, B, d, ld);
  run<256, 128, 2, 2, 16, 2>("256x128 4w(128x64) BK16 2buf", K, A, B, d, ld);
  run<256, 128, 4, 2, 16, 2>("256x128 8w(64x64) BK16 2buf", K, A, B, d, ld);
  run<256, 128, 4, 2, 32, 1>("256x128 8w(64x64) BK32 1buf", K, A, B, d, ld);
  run<256, 256, 4, 2, 16, 2>("256x256 8w(64x128) BK16 2buf", K, A, B, d, ld);
  return 0;
}
