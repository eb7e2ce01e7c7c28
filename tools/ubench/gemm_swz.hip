// Microbenchmark: row-major XOR-swizzled LDS image (16-byte stores, ds_read_b128 fragments covering 4 K-pairs)
// against the k-major image with ds_read_b32 fragments used by conv_mfma.hip.  Plain GEMM addressing, both operands K-contiguous.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;

// image: row m holds its 8 16-byte chunks at position chunk ^ ((m >> 1) & 7)
template <int TM, int TN>
__device__ __forceinline__ void mma_stage_swz(const float* __restrict__ As, const float* __restrict__ Bs, f32x16 (&acc)[TM][TN], int wm0, int wn0, int lane) {
  const int i = lane & 31, h = lane >> 5;
  const int sw = (i >> 1) & 7;
  const float* ap = As + (wm0 + i) * BK;
  const float* bp = Bs + (wn0 + i) * BK;
  float4 a[2][TM], b[2][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) a[0][t] = *reinterpret_cast<const float4*>(ap + t * 32 * BK + ((h ^ sw) << 2));
#pragma unroll
  for (int t = 0; t < TN; ++t) b[0][t] = *reinterpret_cast<const float4*>(bp + t * 32 * BK + ((h ^ sw) << 2));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cur = j & 1, nxt = cur ^ 1;
    if (j + 1 < 4) {
      const int c = (2 * (j + 1) + h) ^ sw;
#pragma unroll
      for (int t = 0; t < TM; ++t) a[nxt][t] = *reinterpret_cast<const float4*>(ap + t * 32 * BK + (c << 2));
#pragma unroll
      for (int t = 0; t < TN; ++t) b[nxt][t] = *reinterpret_cast<const float4*>(bp + t * 32 * BK + (c << 2));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
          const float av = e == 0 ? a[cur][ti].x : e == 1 ? a[cur][ti].y : e == 2 ? a[cur][ti].z : a[cur][ti].w;
          const float bv = e == 0 ? b[cur][tj].x : e == 1 ? b[cur][tj].y : e == 2 ? b[cur][tj].z : b[cur][tj].w;
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ti][tj], 0, 0, 0);
        }
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
  }
}

template <int BM, int BN, int WM, int WN, int MINB>
__global__ __launch_bounds__(256, MINB) void kg(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters, int ld) {
  constexpr int AP = BM / 32, BP = BN / 32;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  __shared__ __attribute__((aligned(16))) float smem[BK * (BM + BN)];
  float* const As = smem;
  float* const Bs = smem + BK * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int arow = tid >> 3, kg_ = tid & 7;
  const float* ap = A + (size_t)(blockIdx.x % 256) * BM * ld + (size_t)arow * ld + 4 * kg_;
  const float* bp = B + (size_t)(blockIdx.x % 8) * BN * ld + (size_t)arow * ld + 4 * kg_;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 ra[AP], rb[BP];
  auto load = [&](int kt) {
#pragma unroll
    for (int p = 0; p < AP; ++p) ra[p] = *reinterpret_cast<const float4*>(ap + kt * BK + (size_t)p * 32 * ld);
#pragma unroll
    for (int p = 0; p < BP; ++p) rb[p] = *reinterpret_cast<const float4*>(bp + kt * BK + (size_t)p * 32 * ld);
  };
  auto store = [&]() {
    const int sw = (arow >> 1) & 7;   // rows arow + 32p share the swizzle
#pragma unroll
    for (int p = 0; p < AP; ++p) *reinterpret_cast<float4*>(As + (arow + 32 * p) * BK + ((kg_ ^ sw) << 2)) = ra[p];
#pragma unroll
    for (int p = 0; p < BP; ++p) *reinterpret_cast<float4*>(Bs + (arow + 32 * p) * BK + ((kg_ ^ sw) << 2)) = rb[p];
  };
  load(0);
  for (int kt = 0; kt < iters; ++kt) {
    __syncthreads();
    store();
    __syncthreads();
    load(kt + 1);
    mma_stage_swz<TM, TN>(As, Bs, acc, wm0, wn0, lane);
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int BM, int BN, int WM, int WN, int MINB>
void run(const char* name, int K, float* A, float* B, float* d, int ld) {
  auto kern = kg<BM, BN, WM, WN, MINB>;
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 256, 0);
  const int iters = K / BK;
  for (int mult : {3, 4, 6, 12}) {
    const int blocks = 256 * mult;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, d, iters, ld); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, d, iters, ld);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * K * BM * BN * 2.0;
    printf("%-34s occ(api) %d  blocks %5d: %.3f ms  %.1f TFLOP/s\n", name, occ, blocks, ms, flops / ms / 1e9);
  }
}

int main() {
  const int K = 2304, ld = K + 64;
  float *A, *B, *d;
  size_t na = (size_t)256 * 128 * ld + 4096, nb = (size_t)8 * 128 * ld + 4096;
  hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&d, (size_t)4096 * 512 * 4);
  float* h = (float*)malloc(na * 4);
  for (size_t i = 0; i < na; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(A, h, na * 4, hipMemcpyHostToDevice); hipMemcpy(B, h, nb * 4, hipMemcpyHostToDevice);
  run<128, 128, 2, 2, 3>("128x128 swz b128, 3 blocks/CU", K, A, B, d, ld);
  run<128, 128, 2, 2, 4>("128x128 swz b128, 4 blocks/CU", K, A, B, d, ld);
  run<128, 64, 2, 2, 3>("128x64 swz b128", K, A, B, d, ld);
  return 0;
}
