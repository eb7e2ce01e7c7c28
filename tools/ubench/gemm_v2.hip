// Microbenchmark of the current conv main loop (BK = 32, one LDS stage, XOR-swizzled unpadded k-major image, operand
// reads with immediate offsets) as a plain GEMM, for different block / wave tile shapes.  Both operands K-contiguous.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;

template <int LDA, int LDB, int TM, int TN, int MS, int NS>
__device__ __forceinline__ void mma_stage(const float* __restrict__ As, const float* __restrict__ Bs, f32x16 (&acc)[TM][TN], int wm0, int wn0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* apv[8];
  const float* bpv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) apv[c] = As + h * LDA + wm0 + (r ^ (4 * c));
#pragma unroll
  for (int c = 0; c < 8; ++c) bpv[c] = Bs + h * LDB + wn0 + (r ^ (4 * c));
  auto ap = [&](int s, int i) { return apv[(s >> 1) & 7][2 * s * LDA + MS * i]; };
  auto bp = [&](int s, int j) { return bpv[(s >> 1) & 7][2 * s * LDB + NS * j]; };
  float a[2][TM], b[2][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a[0][i] = ap(0, i);
#pragma unroll
  for (int j = 0; j < TN; ++j) b[0][j] = bp(0, j);
#pragma unroll
  for (int s = 0; s < BK / 2; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < BK / 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[nxt][i] = ap(s + 1, i);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[nxt][j] = bp(s + 1, j);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
  }
}

// Row-pair variant: tile i of a wave = rows 2r + i of its 64-row group, so both tiles' operands are one 8-byte read.
template <int LDA, int LDB>
__device__ __forceinline__ void mma_stage_pair(const float* __restrict__ As, const float* __restrict__ Bs, f32x16 (&acc)[2][2], int wm0, int wn0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* apv[8];
  const float* bpv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) apv[c] = As + h * LDA + wm0 + ((2 * r) ^ (4 * c));
#pragma unroll
  for (int c = 0; c < 8; ++c) bpv[c] = Bs + h * LDB + wn0 + ((2 * r) ^ (4 * c));
  auto ap = [&](int s) { return *reinterpret_cast<const float2*>(apv[(s >> 1) & 7] + 2 * s * LDA); };
  auto bp = [&](int s) { return *reinterpret_cast<const float2*>(bpv[(s >> 1) & 7] + 2 * s * LDB); };
  // fragments for two K pairs at a time: the two 8-byte reads of an operand share a base and merge into one
  // ds_read2st64_b64
  float2 a[2][2], b[2][2];
  a[0][0] = ap(0); a[0][1] = ap(1);
  b[0][0] = bp(0); b[0][1] = bp(1);
#pragma unroll
  for (int sp = 0; sp < BK / 4; ++sp) {
    const int cur = sp & 1, nxt = cur ^ 1;
    if (sp + 1 < BK / 4) {
      a[nxt][0] = ap(2 * sp + 2); a[nxt][1] = ap(2 * sp + 3);
      b[nxt][0] = bp(2 * sp + 2); b[nxt][1] = bp(2 * sp + 3);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][q].x, b[cur][q].x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][q].x, b[cur][q].y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][q].y, b[cur][q].x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][q].y, b[cur][q].y, acc[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  }
}

template <int LD, int PASSES>
__device__ __forceinline__ void store_transposed(float* __restrict__ dst, const float4 (&v)[PASSES], int tid) {
  const int row = tid >> 3, kg = tid & 7;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    float* d = dst + (4 * kg) * LD + ((row + 32 * p) ^ (4 * kg));
    d[0] = v[p].x; d[LD] = v[p].y; d[2 * LD] = v[p].z; d[3 * LD] = v[p].w;
  }
}

template <int BM, int BN, int WM, int WN, int MINB, bool PAIR = false>
__global__ __launch_bounds__(256, MINB) void kg(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters, int ld, int stagger = 0) {
  // desynchronise the blocks that share a CU: block b, b + 256, b + 512 start 0, 1, 2 x `stagger` x 64 cycles apart
  for (int z = 0; z < (int)((blockIdx.x / 256) % 3) * stagger; ++z) __builtin_amdgcn_s_sleep(1);
  constexpr int AP = BM / 32, BP = BN / 32;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  __shared__ __attribute__((aligned(16))) float smem[BK * (BM + BN)];
  float* const As = smem;
  float* const Bs = smem + BK * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;
  const int arow = tid >> 3, kg_ = tid & 7;
  const float* ap = A + (size_t)(blockIdx.x % 128) * BM * ld + (size_t)arow * ld + 4 * kg_;
  const float* bp = B + (size_t)(blockIdx.x % 4) * BN * ld + (size_t)arow * ld + 4 * kg_;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 ra[AP], rb[BP];
#pragma unroll
  for (int p = 0; p < AP; ++p) ra[p] = *reinterpret_cast<const float4*>(ap + (size_t)p * 32 * ld);
#pragma unroll
  for (int p = 0; p < BP; ++p) rb[p] = *reinterpret_cast<const float4*>(bp + (size_t)p * 32 * ld);
  for (int kt = 0; kt < iters; ++kt) {
    __syncthreads();
    store_transposed<BM, AP>(As, ra, tid);
    store_transposed<BN, BP>(Bs, rb, tid);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < AP; ++p) ra[p] = *reinterpret_cast<const float4*>(ap + (kt + 1) * BK + (size_t)p * 32 * ld);
#pragma unroll
    for (int p = 0; p < BP; ++p) rb[p] = *reinterpret_cast<const float4*>(bp + (kt + 1) * BK + (size_t)p * 32 * ld);
    if constexpr (PAIR) mma_stage_pair<BM, BN>(As, Bs, acc, (wave / WN) * 64, (wave % WN) * 64, lane);
    else mma_stage<BM, BN, TM, TN, 32 * WM, 32 * WN>(As, Bs, acc, wm0, wn0, lane);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int BM, int BN, int WM, int WN, int MINB, bool PAIR = false>
void run(const char* name, int K, float* A, float* B, float* d, int ld) {
  auto kern = kg<BM, BN, WM, WN, MINB, PAIR>;
  const int iters = K / BK;
  for (int mult : {1, 2, 3, 4, 6, 12}) {
    const int blocks = 256 * mult;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, d, iters, ld, 0); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, d, iters, ld, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * K * BM * BN * 2.0;
    printf("%-36s blocks %5d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
  }
}

template <int BM, int BN, int WM, int WN, int MINB>
void ksweep(const char* name, float* A, float* B, float* d, int ld) {
  auto kern = kg<BM, BN, WM, WN, MINB, false>;
  for (int K : {576, 1152, 2304, 4608, 9216}) {
    const int iters = K / BK, blocks = 768;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, d, iters, ld, 0); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, d, iters, ld, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * K * BM * BN * 2.0;
    printf("%-30s K %5d (768 blocks): %.3f ms  %.1f TFLOP/s\n", name, K, ms, flops / ms / 1e9);
  }
}

template <int BM, int BN, int WM, int WN, int MINB>
void stagger_sweep(const char* name, float* A, float* B, float* d, int ld) {
  auto kern = kg<BM, BN, WM, WN, MINB, false>;
  const int K = 2304, iters = K / BK;
  for (int blocks : {768, 1536})
    for (int st : {0, 20, 40, 73, 110, 150, 220}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, d, iters, ld, st); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, d, iters, ld, st);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      double flops = (double)blocks * K * BM * BN * 2.0;
      printf("%-24s blocks %4d stagger %3d: %.3f ms  %.1f TFLOP/s\n", name, blocks, st, ms, flops / ms / 1e9);
    }
}

int main() {
  const int K = 2304, ld = 9216 + 64;
  float *A, *B, *d;
  size_t na = (size_t)128 * 256 * ld + 4096, nb = (size_t)4 * 256 * ld + 4096;
  hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&d, (size_t)4096 * 512 * 4);
  float* h = (float*)malloc(na * 4);
  for (size_t i = 0; i < na; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(A, h, na * 4, hipMemcpyHostToDevice); hipMemcpy(B, h, nb * 4, hipMemcpyHostToDevice);
  ksweep<128, 128, 2, 2, 3>("128x128 K sweep", A, B, d, ld);
  stagger_sweep<128, 128, 2, 2, 3>("128x128 stagger", A, B, d, ld);
  run<128, 128, 2, 2, 3>("128x128 wave 64x64, 3 blocks/CU", K, A, B, d, ld);
  run<128, 128, 2, 2, 3, true>("128x128 row-pair b64 reads", K, A, B, d, ld);
  run<256, 128, 2, 2, 2>("256x128 wave 128x64, 2 blocks/CU", K, A, B, d, ld);
  run<128, 256, 2, 2, 2>("128x256 wave 64x128, 2 blocks/CU", K, A, B, d, ld);
  run<256, 128, 2, 2, 1>("256x128 wave 128x64, regs for 1", K, A, B, d, ld);
  return 0;
}
