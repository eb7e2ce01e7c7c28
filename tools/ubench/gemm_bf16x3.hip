// fp32 GEMM out of bf16 pieces: each operand is split once into three round-to-nearest bf16 planes
// (a = hi + mid + lo, 24 significand bits), and C = A * B^T is accumulated in fp32 from the six piece products of
// relative weight >= 2^-16 with v_mfma_f32_32x32x16_bf16 (NPROD = 6), or from three of them (NPROD = 3: hi*hi + hi*mid +
// mid*hi, error ~2^-16, for comparison only).  Reports the rate in fp32-equivalent TFLOP/s (2*M*N*K / time) and the error
// against an fp64 reference on sampled rows, next to a plain fp32 FMA-chain result of the same rows (what a
// v_mfma_f32_32x32x2_f32 kernel produces up to summation order).
//
// SPLITA variants take A as plain fp32 and split it in the loader (global fp32 -> registers -> three bf16 planes in LDS):
// what a conv kernel would do to keep fp32 activations in HBM; B (the weights) stays pre-split.
//
// Both operands are K-contiguous ([M][K] and [N][K]).  Block tile BM x BN, 4 waves (2 x 2), BK = 32 (two 16-deep MFMA
// steps), one LDS stage with the next tile's global loads in flight during the MFMAs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32;
constexpr int LDK = BK + 8;  // row pitch in bf16: 80 bytes, keeps the 16-byte fragment reads of 8 rows on distinct banks

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ u16 bf16_rn(float f) {  // round to nearest even; inputs are finite here
  unsigned u = __float_as_uint(f);
  return (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f32(u16 h) { return __uint_as_float((unsigned)h << 16); }

// planes[0 | 1 | 2][rows][K] = hi | mid | lo
__global__ void split3_kernel(const float* __restrict__ x, u16* __restrict__ planes, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[i];
  const u16 hi = bf16_rn(a);
  const float r1 = a - bf16_f32(hi);
  const u16 mid = bf16_rn(r1);
  const float r2 = r1 - bf16_f32(mid);
  planes[i] = hi;
  planes[n + i] = mid;
  planes[2 * n + i] = bf16_rn(r2);
}

template <int BM, int BN, int NPROD, int WM = 2, int WN = 2, bool SPLITA = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16x3_kernel(const float* __restrict__ A32, const u16* __restrict__ A, const u16* __restrict__ B, float* __restrict__ C,
                                                           int M, int N, int K) {
  constexpr int NT = 64 * WM * WN;                        // threads
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;     // 32x32 tiles per wave in each direction
  constexpr int NPL = NPROD == 6 ? 3 : 2;                 // planes needed
  constexpr int AV = BM * BK / 8 / NT, BV = BN * BK / 8 / NT;  // 16-byte vectors per thread per plane
  extern __shared__ __attribute__((aligned(16))) u16 smem[];  // A planes then B planes (up to 120 KB: dynamic)
  u16(*As)[BM * LDK] = reinterpret_cast<u16(*)[BM * LDK]>(smem);
  u16(*Bs)[BN * LDK] = reinterpret_cast<u16(*)[BN * LDK]>(smem + NPL * BM * LDK);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int nbn = N / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x - bm * nbn;
  const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;
  const int r = lane & 31, h = lane >> 5;

  constexpr int AV32 = BM * BK / 4 / NT;                  // float4 vectors per thread when A arrives as fp32
  u32x4 ra[NPL * AV], rb[NPL * BV];
  f32x4 ra32[AV32];
  auto gload = [&](int kt) __attribute__((always_inline)) {
    if (SPLITA) {
#pragma unroll
      for (int v = 0; v < AV32; ++v) {
        const int idx = tid + NT * v, row = idx >> 3, kq = idx & 7;
        ra32[v] = *reinterpret_cast<const f32x4*>(A32 + (size_t)(bm * BM + row) * K + kt * BK + 4 * kq);
      }
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      if (!SPLITA) {
#pragma unroll
        for (int v = 0; v < AV; ++v) {
          const int idx = tid + NT * v, row = idx >> 2, kv = idx & 3;
          ra[p * AV + v] = *reinterpret_cast<const u32x4*>(A + p * planeA + (size_t)(bm * BM + row) * K + kt * BK + 8 * kv);
        }
      }
#pragma unroll
      for (int v = 0; v < BV; ++v) {
        const int idx = tid + NT * v, row = idx >> 2, kv = idx & 3;
        rb[p * BV + v] = *reinterpret_cast<const u32x4*>(B + p * planeB + (size_t)(bn * BN + row) * K + kt * BK + 8 * kv);
      }
    }
  };
  auto sstore = [&]() __attribute__((always_inline)) {
    if (SPLITA) {  // hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid): v_cvt_pk_bf16_f32 and subtractions
#pragma unroll
      for (int v = 0; v < AV32; ++v) {
        const int idx = tid + NT * v, row = idx >> 3, kq = idx & 7;
        const f32x4 a = ra32[v];
        const bf16x4 hi = __builtin_convertvector(a, bf16x4);
        const f32x4 r1 = a - __builtin_convertvector(hi, f32x4);
        const bf16x4 mid = __builtin_convertvector(r1, bf16x4);
        *reinterpret_cast<bf16x4*>(&As[0][row * LDK + 4 * kq]) = hi;
        *reinterpret_cast<bf16x4*>(&As[1][row * LDK + 4 * kq]) = mid;
        if (NPL == 3) {
          const f32x4 r2 = r1 - __builtin_convertvector(mid, f32x4);
          *reinterpret_cast<bf16x4*>(&As[2][row * LDK + 4 * kq]) = __builtin_convertvector(r2, bf16x4);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      if (!SPLITA) {
#pragma unroll
        for (int v = 0; v < AV; ++v) {
          const int idx = tid + NT * v, row = idx >> 2, kv = idx & 3;
          *reinterpret_cast<u32x4*>(&As[p][row * LDK + 8 * kv]) = ra[p * AV + v];
        }
      }
#pragma unroll
      for (int v = 0; v < BV; ++v) {
        const int idx = tid + NT * v, row = idx >> 2, kv = idx & 3;
        *reinterpret_cast<u32x4*>(&Bs[p][row * LDK + 8 * kv]) = rb[p * BV + v];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / BK;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a[NPL][TM], b[NPL][TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[p][i] = *reinterpret_cast<const bf16x8*>(&As[p][(wm * (BM / WM) + 32 * i + r) * LDK + 16 * s + 8 * h]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[p][j] = *reinterpret_cast<const bf16x8*>(&Bs[p][(wn * (BN / WN) + 32 * j + r) * LDK + 16 * s + 8 * h]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // smallest terms first
          if (NPROD == 6) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
    }
  }
  // C/D layout of the 32x32 MFMAs: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = bm * BM + wm * (BM / WM) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int col = bn * BN + wn * (BN / WN) + 32 * j + r;
        C[(size_t)row * N + col] = acc[i][j][e];
      }
}

template <int BM, int BN, int NPROD, int WM = 2, int WN = 2, bool SPLITA = false>
static void run(const char* name, const float* dA32, const u16* dA, const u16* dB, float* dC, int M, int N, int K, const float* hA, const float* hB,
                const double* ref, const int* rows, int nrows) {
  if (M % BM || N % BN) return;  // conv-like shapes: only the tiles that divide
  const dim3 grid((M / BM) * (N / BN));
  const size_t lds = (size_t)(NPROD == 6 ? 3 : 2) * (BM + BN) * LDK * sizeof(u16);
  CHECK(hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<BM, BN, NPROD, WM, WN, SPLITA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, NPROD, WM, WN, SPLITA>), grid, dim3(64 * WM * WN), lds, 0, dA32, dA, dB, dC, M, N, K);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, NPROD, WM, WN, SPLITA>), grid, dim3(64 * WM * WN), lds, 0, dA32, dA, dB, dC, M, N, K);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  float* hC = (float*)malloc((size_t)nrows * N * sizeof(float));
  for (int q = 0; q < nrows; ++q) CHECK(hipMemcpy(hC + (size_t)q * N, dC + (size_t)rows[q] * N, N * sizeof(float), hipMemcpyDeviceToHost));
  double max_rel = 0, sum_rel = 0, scale = 0;
  for (size_t i = 0; i < (size_t)nrows * N; ++i) scale = fmax(scale, fabs(ref[i]));
  for (size_t i = 0; i < (size_t)nrows * N; ++i) {
    const double e = fabs((double)hC[i] - ref[i]) / scale;
    max_rel = fmax(max_rel, e);
    sum_rel += e;
  }
  printf("%-28s %dx%dx%d  %7.3f ms  %7.1f TFLOP/s (fp32-equivalent)  max err / max|C| %.3e  mean %.3e\n", name, M, N, K, ms,
         2.0 * M * N * K / ms / 1e9, max_rel, sum_rel / ((double)nrows * N));
  free(hC);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 2304;
  if (M % 128 || N % 128 || K % BK) {
    fprintf(stderr, "M, N multiples of 128 and K a multiple of 32\n");
    return 1;
  }
  float* hA = (float*)malloc((size_t)M * K * 4);
  float* hB = (float*)malloc((size_t)N * K * 4);
  srand(1);
  for (size_t i = 0; i < (size_t)M * K; ++i) hA[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (size_t i = 0; i < (size_t)N * K; ++i) hB[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  // reference rows in fp64, and the same rows as an fp32 FMA chain
  const int nrows = 8;
  int rows[nrows];
  for (int q = 0; q < nrows; ++q) rows[q] = (int)(((long long)q * 1237 + 5) % M);
  double* ref = (double*)malloc((size_t)nrows * N * sizeof(double));
  double fp32_max = 0, fp32_sum = 0, scale = 0;
  for (int q = 0; q < nrows; ++q)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      float f = 0.f;
      for (int k = 0; k < K; ++k) {
        s += (double)hA[(size_t)rows[q] * K + k] * (double)hB[(size_t)n * K + k];
        f = fmaf(hA[(size_t)rows[q] * K + k], hB[(size_t)n * K + k], f);
      }
      ref[(size_t)q * N + n] = s;
      scale = fmax(scale, fabs(s));
      fp32_max = fmax(fp32_max, fabs((double)f - s));
      fp32_sum += fabs((double)f - s);
    }
  printf("fp32 FMA chain over K = %d (sequential order): max err / max|C| %.3e  mean %.3e\n", K, fp32_max / scale,
         fp32_sum / scale / ((double)nrows * N));

  float *dA32, *dB32, *dC;
  u16 *dA, *dB;
  CHECK(hipMalloc(&dA32, (size_t)M * K * 4));
  CHECK(hipMalloc(&dB32, (size_t)N * K * 4));
  CHECK(hipMalloc(&dA, (size_t)M * K * 6));
  CHECK(hipMalloc(&dB, (size_t)N * K * 6));
  CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  CHECK(hipMemcpy(dA32, hA, (size_t)M * K * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB32, hB, (size_t)N * K * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)(((size_t)M * K + 255) / 256)), dim3(256), 0, 0, dA32, dA, (size_t)M * K);
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)(((size_t)N * K + 255) / 256)), dim3(256), 0, 0, dB32, dB, (size_t)N * K);
  CHECK(hipDeviceSynchronize());
  run<128, 128, 6>("bf16x3, 6 products, 128x128", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<256, 128, 6>("bf16x3, 6 products, 256x128", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<256, 256, 6>("bf16x3, 6 products, 256x256", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<256, 256, 6, 2, 4>("bf16x3, 6 prod, 256x256 8w 2x4", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<256, 256, 6, 4, 2>("bf16x3, 6 prod, 256x256 8w 4x2", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<256, 128, 6, 4, 2>("bf16x3, 6 prod, 256x128 8w 4x2", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<128, 128, 6, 2, 4>("bf16x3, 6 prod, 128x128 8w 2x4", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<128, 128, 6, 2, 2, true>("6 prod, 128x128, A split in loader", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<128, 128, 6, 2, 4, true>("6 prod, 128x128 8w, A split", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<256, 256, 6, 2, 4, true>("6 prod, 256x256 8w, A split", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<256, 128, 6, 4, 2, true>("6 prod, 256x128 8w, A split", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<128, 128, 3>("bf16x2, 3 products, 128x128", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  run<256, 256, 3>("bf16x2, 3 products, 256x256", dA32, dA, dB, dC, M, N, K, hA, hB, ref, rows, nrows);
  return 0;
}
