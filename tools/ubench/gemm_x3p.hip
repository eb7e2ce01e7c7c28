// Round-2 microbenchmark for the bf16-piece conv K loop: C[M][N] = A[M][K] * B[N][K]^T in fp32 from
// three bf16 pieces per operand (six v_mfma_f32_32x32x16_bf16 products), as a plain GEMM with the operand forms a conv
// kernel has:
//   A (activations)  fp32 in HBM, K-contiguous rows, split into hi / mid / lo in the loader (registers -> LDS);
//   B (weights)      either fp32 split in the loader too (BPRE = false: what the round-1 kernels do), or pre-split once per
//                    step into bf16 planes laid out [plane][K/32][N][32] (BPRE = true: a K-step's tile is one contiguous
//                    block, loaded with 16-byte loads and stored to LDS with ds_write_b128, no VALU).
// Knobs: block tile BM x BN, waves WM x WN, NBUF LDS stages (1: two barriers per K-step, 2: one), PF register prefetch depth
// (1 or 2 K-steps of global loads in flight).  LDS rows are unpadded 64-byte rows (32 bf16) with the 16-byte chunk index
// XOR-ed by (row >> 2) & 3: the ds_read_b128 fragment reads and the ds_write_b64 / b128 stores are all conflict-free.
//
//   tools/ubench/gemm_x3p.bin [M N K]      default: a list of TSM-R50 conv GEMM shapes
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
constexpr int BK = 32;

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__device__ __forceinline__ u16 bf16_rn(float f) {
  unsigned u = __float_as_uint(f);
  return (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f32(u16 h) { return __uint_as_float((unsigned)h << 16); }

// B planes, tiled: planes[p][kt][n][j] = piece p of B[n][kt*32 + j]
__global__ void split3_tiled_kernel(const float* __restrict__ b, u16* __restrict__ planes, int N, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * K;
  if (i >= total) return;
  const int n = (int)(i / K), k = (int)(i - (size_t)n * K);
  const float a = b[i];
  const u16 hi = bf16_rn(a);
  const float r1 = a - bf16_f32(hi);
  const u16 mid = bf16_rn(r1);
  const float r2 = r1 - bf16_f32(mid);
  const size_t o = ((size_t)(k >> 5) * N + n) * 32 + (k & 31);
  planes[o] = hi;
  planes[total + o] = mid;
  planes[2 * total + o] = bf16_rn(r2);
}

__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, j = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// byte offset of 16-byte chunk c of row `row` in a plane image of 64-byte rows
__device__ __forceinline__ int lds_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

template <int BM, int BN, int WM, int WN, int NBUF, int PF, bool BPRE, int ILV = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_x3p_kernel(const float* __restrict__ A, const float* __restrict__ B32,
                                                               const u16* __restrict__ Bp, float* __restrict__ C, int M, int N,
                                                               int K) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM * 8 / NT;             // float4 loads of A per thread per K-step (8 per 32-float row)
  constexpr int BP32 = BN * 8 / NT;           // the same for an fp32 B
  constexpr int BPP = BN * 4 / NT;            // 16-byte loads per plane per thread for a pre-split B (4 per 64-byte row)
  static_assert(AP >= 1 && BPP >= 1, "tile too small for the thread count");
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;       // bytes
  constexpr int STAGE = 3 * (PLANE_A + PLANE_B);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int nbn = N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const size_t totalB = (size_t)N * K;
  const int nk = K / BK;

  const int arow = tid >> 3, kg = tid & 7;    // A / fp32 B: row arow + (NT/8) p, floats 4 kg .. 4 kg + 3
  const int brow = tid >> 2, bc = tid & 3;    // pre-split B: row brow + (NT/4) q, chunk bc

  f32x4 ra[PF][AP];
  f32x4 rb32[PF][BPRE ? 1 : BP32];
  u32x4 rbp[PF][BPRE ? 3 * BPP : 1];

  auto gload = [&](int kt, auto set) __attribute__((always_inline)) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int p = 0; p < AP; ++p)
      ra[S][p] = *reinterpret_cast<const f32x4*>(A + (size_t)(bm * BM + arow + (NT / 8) * p) * K + kt * BK + 4 * kg);
    if (BPRE) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int q = 0; q < BPP; ++q)
          rbp[S][pl * BPP + q] = *reinterpret_cast<const u32x4*>(Bp + pl * totalB + ((size_t)kt * N + bn * BN + brow + (NT / 4) * q) * 32 + 8 * bc);
    } else {
#pragma unroll
      for (int p = 0; p < BP32; ++p)
        rb32[S][p] = *reinterpret_cast<const f32x4*>(B32 + (size_t)(bn * BN + arow + (NT / 8) * p) * K + kt * BK + 4 * kg);
    }
  };

  auto pk2 = [](float a, float b) __attribute__((always_inline)) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
  };
  auto split_store = [&](unsigned char* base, int plane_bytes, int row, const f32x4 v) __attribute__((always_inline)) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned h0 = pk2(v.x, v.y), h1 = pk2(v.z, v.w);
    const float rx = v.x - __uint_as_float(h0 << 16), ry = v.y - __uint_as_float(h0 & 0xffff0000u);
    const float rz = v.z - __uint_as_float(h1 << 16), rw = v.w - __uint_as_float(h1 & 0xffff0000u);
    const unsigned m0 = pk2(rx, ry), m1 = pk2(rz, rw);
    const float sx = rx - __uint_as_float(m0 << 16), sy = ry - __uint_as_float(m0 & 0xffff0000u);
    const float sz = rz - __uint_as_float(m1 << 16), sw = rw - __uint_as_float(m1 & 0xffff0000u);
    unsigned char* q = base + lds_off(row, kg >> 1) + 8 * (kg & 1);
    *reinterpret_cast<u32x2*>(q) = (u32x2){h0, h1};
    *reinterpret_cast<u32x2*>(q + plane_bytes) = (u32x2){m0, m1};
    *reinterpret_cast<u32x2*>(q + 2 * plane_bytes) = (u32x2){pk2(sx, sy), pk2(sz, sw)};
  };

  auto sstore = [&](int buf, auto set) __attribute__((always_inline)) {
    constexpr int S = decltype(set)::value;
    unsigned char* const As = smem + buf * STAGE;
    unsigned char* const Bs = As + 3 * PLANE_A;
#pragma unroll
    for (int p = 0; p < AP; ++p) split_store(As, PLANE_A, arow + (NT / 8) * p, ra[S][p]);
    if (BPRE) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int q = 0; q < BPP; ++q)
          *reinterpret_cast<u32x4*>(Bs + pl * PLANE_B + lds_off(brow + (NT / 4) * q, bc)) = rbp[S][pl * BPP + q];
    } else {
#pragma unroll
      for (int p = 0; p < BP32; ++p) split_store(Bs, PLANE_B, arow + (NT / 8) * p, rb32[S][p]);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto mma = [&](int buf) __attribute__((always_inline)) {
    const unsigned char* const As = smem + buf * STAGE;
    const unsigned char* const Bs = As + 3 * PLANE_A;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a[3][TM], b[3][TN];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[p][i] = *reinterpret_cast<const bf16x8*>(As + p * PLANE_A + lds_off(wm * (BM / WM) + 32 * i + r, 2 * s + h));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[p][j] = *reinterpret_cast<const bf16x8*>(Bs + p * PLANE_B + lds_off(wn * (BN / WN) + 32 * j + r, 2 * s + h));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, PF == 2 ? 1 : 0>;

  if (NBUF == 1) {
    // one LDS stage: barrier, registers -> LDS, barrier, next loads, MFMAs (the round-1 structure; PF == 2 keeps a second
    // K-step of loads in flight)
    gload(0, S0{});
    if (PF == 2 && nk > 1) gload(1, S1{});
    for (int kt = 0; kt < nk; kt += 2) {
      __syncthreads();
      sstore(0, S0{});
      __syncthreads();
      if (kt + PF < nk) gload(kt + PF, S0{});
      mma(0);
      if (kt + 1 < nk) {
        __syncthreads();
        sstore(0, S1{});
        __syncthreads();
        if (kt + 1 + PF < nk) gload(kt + 1 + PF, S1{});
        mma(0);
      }
    }
  } else {
    // two LDS stages, one barrier per K-step: while the MFMAs of step kt read stage kt & 1, the registers of step kt + 1 are
    // split and stored into the other stage
    gload(0, S0{});
    sstore(0, S0{});
    if (nk > 1) gload(1, S1{});
    if (PF == 2 && nk > 2) gload(2, S0{});   // PF == 2: S0 was consumed by the store above
    __syncthreads();
    int kt0 = 0;
    if constexpr (ILV > 0 && PF == 2) {
      // steady state without conditions: MFMAs of step kt, split + store of step kt + 1 and the loads of step kt + 3 are ONE basic
      // block, so the scheduler may interleave them (ILV == 2: pinned with sched_group_barrier)
      constexpr int NM = TM * TN * 12, NDW = AP * 3 + 3 * BPP, NVM = AP + 3 * BPP;
      constexpr int VPM = (AP * 22 + NM - 1) / NM;
      auto pin = [&]() __attribute__((always_inline)) {
        if constexpr (ILV == 2) {
#pragma unroll
          for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            if (g % (NM / NDW) == 0 && g / (NM / NDW) < NDW) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (g % (NM / NVM) == 1 && g / (NM / NVM) < NVM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
        }
      };
      for (; kt0 + 4 < nk; kt0 += 2) {
        mma(0);
        sstore(1, S1{});
        gload(kt0 + 3, S1{});
        pin();
        __syncthreads();
        mma(1);
        sstore(0, S0{});
        gload(kt0 + 4, S0{});
        pin();
        __syncthreads();
      }
    }
    for (int kt = kt0; kt < nk; kt += 2) {
      // even step: MFMAs on stage 0; registers of step kt + 1 sit in set S1
      if (PF == 1 && kt + 1 < nk && kt > 0) gload(kt + 1, S1{});
      mma(0);
      if (kt + 1 < nk) sstore(1, S1{});
      if (PF == 2 && kt + 3 < nk) gload(kt + 3, S1{});
      __syncthreads();
      if (kt + 1 >= nk) break;
      // odd step: MFMAs on stage 1; registers of step kt + 2 sit in set S0
      if (PF == 1 && kt + 2 < nk) gload(kt + 2, S0{});
      mma(1);
      if (kt + 2 < nk) sstore(0, S0{});
      if (PF == 2 && kt + 4 < nk) gload(kt + 4, S0{});
      __syncthreads();
    }
  }

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

struct Ctx {
  const float *dA, *dB32;
  const u16* dBp;
  float* dC;
  int M, N, K;
  const double* ref;
  const int* rows;
  int nrows;
};

template <int BM, int BN, int WM, int WN, int NBUF, int PF, bool BPRE, int ILV = 0>
static void run(const char* name, const Ctx& c) {
  if (c.M % BM || c.N % BN) return;
  if (NBUF == 2 && PF == 1 && false) return;
  const dim3 grid((c.M / BM) * (c.N / BN));
  const size_t lds = (size_t)NBUF * 3 * (BM + BN) * 64;
  if (lds > 160 * 1024) return;
  auto kern = gemm_x3p_kernel<BM, BN, WM, WN, NBUF, PF, BPRE, ILV>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipMemset(c.dC, 0, (size_t)c.M * c.N * 4));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, 0, c.dA, c.dB32, c.dBp, c.dC, c.M, c.N, c.K);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, 0, c.dA, c.dB32, c.dBp, c.dC, c.M, c.N, c.K);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  float* hC = (float*)malloc((size_t)c.nrows * c.N * sizeof(float));
  for (int q = 0; q < c.nrows; ++q) CHECK(hipMemcpy(hC + (size_t)q * c.N, c.dC + (size_t)c.rows[q] * c.N, c.N * sizeof(float), hipMemcpyDeviceToHost));
  double max_rel = 0, scale = 0;
  for (size_t i = 0; i < (size_t)c.nrows * c.N; ++i) scale = fmax(scale, fabs(c.ref[i]));
  for (size_t i = 0; i < (size_t)c.nrows * c.N; ++i) max_rel = fmax(max_rel, fabs((double)hC[i] - c.ref[i]) / scale);
  hipFuncAttributes fa;
  CHECK(hipFuncGetAttributes(&fa, (const void*)kern));
  printf("  %-44s %7.3f ms %7.1f TF  err %.2e  vgpr %d lds %zuK blocks %d\n", name, ms, 2.0 * c.M * c.N * c.K / ms / 1e9, max_rel, fa.numRegs,
         lds / 1024, grid.x);
  fflush(stdout);
  free(hC);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

static void bench_shape(int M, int N, int K) {
  printf("== M %d N %d K %d\n", M, N, K);
  float* hA = (float*)malloc((size_t)M * K * 4);
  float* hB = (float*)malloc((size_t)N * K * 4);
  srand(1);
  for (size_t i = 0; i < (size_t)M * K; ++i) hA[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (size_t i = 0; i < (size_t)N * K; ++i) hB[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  const int nrows = 4;
  int rows[nrows];
  for (int q = 0; q < nrows; ++q) rows[q] = (int)(((long long)q * 1237 + 5) % M);
  double* ref = (double*)malloc((size_t)nrows * N * sizeof(double));
  for (int q = 0; q < nrows; ++q)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)hA[(size_t)rows[q] * K + k] * (double)hB[(size_t)n * K + k];
      ref[(size_t)q * N + n] = s;
    }
  float *dA, *dB32, *dC;
  u16* dBp;
  CHECK(hipMalloc(&dA, (size_t)M * K * 4));
  CHECK(hipMalloc(&dB32, (size_t)N * K * 4));
  CHECK(hipMalloc(&dBp, (size_t)N * K * 6));
  CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  CHECK(hipMemcpy(dA, hA, (size_t)M * K * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB32, hB, (size_t)N * K * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(split3_tiled_kernel, dim3((unsigned)(((size_t)N * K + 255) / 256)), dim3(256), 0, 0, dB32, dBp, N, K);
  CHECK(hipDeviceSynchronize());
  Ctx c = {dA, dB32, dBp, dC, M, N, K, ref, rows, nrows};
  //   BM   BN  WM WN NBUF PF BPRE
  run<128, 128, 2, 2, 1, 1, false>("128x128 4w 1buf pf1 B split in loop (r1)", c);
  run<128, 128, 2, 2, 1, 1, true>("128x128 4w 1buf pf1 B pre-split", c);
  run<128, 128, 2, 2, 1, 2, true>("128x128 4w 1buf pf2 B pre-split", c);
  run<128, 128, 2, 2, 2, 1, true>("128x128 4w 2buf pf1 B pre-split", c);
  run<128, 128, 2, 2, 2, 2, true>("128x128 4w 2buf pf2 B pre-split", c);
  run<128, 256, 2, 4, 1, 1, true>("128x256 8w 1buf pf1 B pre-split", c);
  run<128, 256, 2, 4, 1, 2, true>("128x256 8w 1buf pf2 B pre-split", c);
  run<128, 256, 2, 4, 2, 1, true>("128x256 8w 2buf pf1 B pre-split", c);
  run<128, 256, 2, 4, 2, 2, true>("128x256 8w 2buf pf2 B pre-split", c);
  run<256, 128, 4, 2, 2, 1, true>("256x128 8w 2buf pf1 B pre-split", c);
  run<256, 128, 4, 2, 2, 2, true>("256x128 8w 2buf pf2 B pre-split", c);
  run<256, 256, 2, 4, 1, 1, true>("256x256 8w 1buf pf1 B pre-split", c);
  run<256, 256, 2, 4, 1, 2, true>("256x256 8w 1buf pf2 B pre-split", c);
  run<128, 256, 2, 4, 2, 2, false>("128x256 8w 2buf pf2 B split in loop", c);
  run<128, 256, 2, 4, 2, 2, true, 1>("128x256 8w 2buf pf2 peeled", c);
  run<128, 256, 2, 4, 2, 2, true, 2>("128x256 8w 2buf pf2 peeled + pinned interleave", c);
  run<256, 128, 4, 2, 2, 2, true, 1>("256x128 8w 2buf pf2 peeled", c);
  run<256, 128, 4, 2, 2, 2, true, 2>("256x128 8w 2buf pf2 peeled + pinned interleave", c);
  run<128, 128, 2, 2, 2, 2, true, 2>("128x128 4w 2buf pf2 peeled + pinned interleave", c);
  CHECK(hipFree(dA));
  CHECK(hipFree(dB32));
  CHECK(hipFree(dBp));
  CHECK(hipFree(dC));
  free(hA);
  free(hB);
  free(ref);
}

int main(int argc, char** argv) {
  if (argc > 3) {
    bench_shape(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]));
    return 0;
  }
  // GEMM views of TSM-R50 conv sites at N = 256 frames (M = pixels, N = Cout, K = taps * Cin)
  const int shapes[][3] = {
      {8192, 4096, 2304},   // large reference GEMM
      {50176, 256, 2304},   // layer3 3x3 256 -> 256
      {50176, 1024, 256},   // layer3 1x1 256 -> 1024
      {50176, 256, 1024},   // layer3 1x1 1024 -> 256
      {200704, 512, 128},   // layer2 1x1 128 -> 512
      {200704, 128, 1152},  // layer2 3x3 128 -> 128
      {802816, 256, 64},    // layer1 1x1 64 -> 256
      {12544, 2048, 512},   // layer4 1x1 512 -> 2048
      {12544, 512, 4608},   // layer4 3x3 512 -> 512
  };
  for (auto& s : shapes) bench_shape(s[0], s[1], s[2]);
  return 0;
}
