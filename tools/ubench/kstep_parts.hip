// Round-3 microbenchmark: what one 32-deep K-step of the 8-wave bf16-piece loop costs, built up part by part.
// Workgroup = 8 waves (2 x 4), block tile 128 x 256, wave tile 64 x 64 (48 v_mfma_f32_32x32x16_bf16 per wave and K-step: six piece
// products), one workgroup per CU on all 256 CUs, operands random bf16.  Parts (each level adds to the previous one):
//   0  MFMAs on register operands only
//   1  + the 24 ds_read_b128 fragment reads per wave and K-step from a static LDS image (64-byte rows, chunk XOR swizzle)
//   2  + one __syncthreads() per K-step (two LDS stages, alternating)
//   3  + the LDS stores of a K-step's operand tiles (A: 12 ds_write_b64 per thread of split pieces, B: 6 ds_write_b128) from registers
//   4  + the piece split of the activation tile (20 VALU per float4) feeding those stores
//   5  + the global loads of the next tiles (A fp32 16 KB, B planes 48 KB per K-step, L2-resident)
//   6  as 5, but the B planes go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPRs, no ds_write_b128): issued first in
//      the step, a counted s_waitcnt vmcnt(2) (the two activation loads stay in flight) and a raw s_barrier
//   7  as 6 with __syncthreads() (the compiler then drains every load at the barrier)
// Prints cycles per K-step (s_memtime over the loop until ALL waves of the workgroup are done, median over workgroups), the in-kernel
// clock and TFLOP/s-equivalent.  (Levels 0 and 1 have no barrier in the loop: the compiler hoists the fragment reads of level 1 out
// of it, and the two waves of a SIMD are served oldest first -- wave 0 alone sees 1537 cycles per K-step, its partner runs after it.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BM = 128, BN = 256, WM = 2, WN = 4, NT = 512;
constexpr int PA = BM * 64, PB = BN * 64, STAGE = 3 * (PA + PB);
__device__ __forceinline__ int lds_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ unsigned pk2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2)); }

template <int LEVEL>
__global__ __launch_bounds__(512) void kstep_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Bp, float* __restrict__ out,
                                                     unsigned long long* __restrict__ stamps, int nk, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  // fill both stages with pseudo-random bf16 bits (finite values)
  for (int i = tid; i < 2 * STAGE / 4; i += NT) {
    unsigned v = (unsigned)(i * 2654435761u) ^ (unsigned)(blockIdx.x * 40503u);
    v = (v & 0x807f807fu) | 0x3f003f00u;     // exponents near 0: values in [0.5, 2)
    reinterpret_cast<unsigned*>(smem)[i] = v;
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8 ra[3][2], rb[3][2];
  for (int p = 0; p < 3; ++p) for (int i = 0; i < 2; ++i) {
    ra[p][i] = *reinterpret_cast<const bf16x8*>(smem + p * PA + lds_off(wm * 64 + 32 * i + r, h));
    rb[p][i] = *reinterpret_cast<const bf16x8*>(smem + 3 * PA + p * PB + lds_off(wn * 64 + 32 * i + r, h));
  }
  const int arow = tid >> 3, kg = tid & 7, brow = tid >> 2, bc = tid & 3;
  f32x4 va[2];
  u32x4 vb[6];
  for (int p = 0; p < 2; ++p) va[p] = (f32x4){1.f + tid * 1e-3f, 0.5f, 0.25f + p, 2.f};
  for (int p = 0; p < 6; ++p) vb[p] = (u32x4){0x3f803f80u + tid, 0x3f003f00u, 0x3e803e80u, 0x3f803f00u};
  const size_t bplane = (size_t)BN * K;      // u16 elements per plane of this block's B slice: [K/32][BN][32]

  auto mma = [&](int buf) __attribute__((always_inline)) {
    const unsigned char* As = smem + buf * STAGE;
    const unsigned char* Bs = As + 3 * PA;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][2], b[3][2];
      if constexpr (LEVEL >= 1) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int i = 0; i < 2; ++i) a[p][i] = *reinterpret_cast<const bf16x8*>(As + p * PA + lds_off(wm * 64 + 32 * i + r, 2 * s + h));
#pragma unroll
          for (int j = 0; j < 2; ++j) b[p][j] = *reinterpret_cast<const bf16x8*>(Bs + p * PB + lds_off(wn * 64 + 32 * j + r, 2 * s + h));
        }
      } else {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < 2; ++i) { a[p][i] = ra[p][i]; b[p][i] = rb[p][i]; }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)(3 * bplane * 2), 0x00020000);
  auto dma_b = [&](int kt, int buf) __attribute__((always_inline)) {
    unsigned char* Bs = smem + buf * STAGE + 3 * PA;
    const int prow = lane >> 2, pslot = lane & 3;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int q = wave + 8 * j;            // 48 pieces of 16 rows: 3 planes x 16
      const int pl = q >> 4, rb16 = q & 15;
      const int row = rb16 * 16 + prow;
      const int c = pslot ^ ((row >> 2) & 3);
      const int src = (int)((pl * bplane + ((size_t)(kt % (K / 32)) * BN + row) * 32 + 8 * c) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr_t)(Bs + pl * PB + rb16 * 1024), 16, src, 0, 0, 0);
    }
  };
  auto gload = [&](int kt) __attribute__((always_inline)) {
    if constexpr (LEVEL >= 6) {
#pragma unroll
      for (int p = 0; p < 2; ++p) va[p] = *reinterpret_cast<const f32x4*>(A + (size_t)(arow + 64 * p) * K + (kt % (K / 32)) * 32 + 4 * kg);
    } else if constexpr (LEVEL >= 5) {
#pragma unroll
      for (int p = 0; p < 2; ++p) va[p] = *reinterpret_cast<const f32x4*>(A + (size_t)(arow + 64 * p) * K + (kt % (K / 32)) * 32 + 4 * kg);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          vb[pl * 2 + q] = *reinterpret_cast<const u32x4*>(Bp + pl * bplane + ((size_t)(kt % (K / 32)) * BN + brow + 128 * q) * 32 + 8 * bc);
    }
  };
  auto sstore = [&](int buf) __attribute__((always_inline)) {
    if constexpr (LEVEL >= 3) {
      unsigned char* As = smem + buf * STAGE;
      unsigned char* Bs = As + 3 * PA;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        unsigned char* q = As + lds_off(arow + 64 * p, kg >> 1) + 8 * (kg & 1);
        if constexpr (LEVEL >= 4) {
          const f32x4 v = va[p];
          const unsigned h0 = pk2(v.x, v.y), h1 = pk2(v.z, v.w);
          const float rx = v.x - __uint_as_float(h0 << 16), ry = v.y - __uint_as_float(h0 & 0xffff0000u);
          const float rz = v.z - __uint_as_float(h1 << 16), rw = v.w - __uint_as_float(h1 & 0xffff0000u);
          const unsigned m0 = pk2(rx, ry), m1 = pk2(rz, rw);
          const float sx = rx - __uint_as_float(m0 << 16), sy = ry - __uint_as_float(m0 & 0xffff0000u);
          const float sz = rz - __uint_as_float(m1 << 16), sw = rw - __uint_as_float(m1 & 0xffff0000u);
          *reinterpret_cast<u32x2*>(q) = (u32x2){h0, h1};
          *reinterpret_cast<u32x2*>(q + PA) = (u32x2){m0, m1};
          *reinterpret_cast<u32x2*>(q + 2 * PA) = (u32x2){pk2(sx, sy), pk2(sz, sw)};
        } else {
          const u32x2 c = (u32x2){__float_as_uint(va[p].x) & 0x3fff3fffu, __float_as_uint(va[p].y) & 0x3fff3fffu};
          *reinterpret_cast<u32x2*>(q) = c;
          *reinterpret_cast<u32x2*>(q + PA) = c;
          *reinterpret_cast<u32x2*>(q + 2 * PA) = c;
        }
      }
      if constexpr (LEVEL < 6) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4*>(Bs + pl * PB + lds_off(brow + 128 * q, bc)) = vb[pl * 2 + q] & (u32x4){0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu};
      }
    }
  };
  auto sync = [&]() __attribute__((always_inline)) {
    if constexpr (LEVEL == 6) {
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else if constexpr (LEVEL >= 2) {
      __syncthreads();
    }
  };

  gload(0);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int kt = 0; kt < nk; kt += 2) {
    if constexpr (LEVEL >= 6) dma_b(kt + 1, 1);
    mma(0);
    sstore(1);
    gload(kt + 1);
    sync();
    if constexpr (LEVEL >= 6) dma_b(kt + 2, 0);
    mma(1);
    sstore(0);
    gload(kt + 2);
    sync();
  }
  __syncthreads();      // the matrix pipe serves the older wave of a SIMD first: without this, wave 0's clock would only show ITS half
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = w1 - w0;
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[(size_t)blockIdx.x * NT + tid] = s + va[0].x + __uint_as_float(vb[0].x);
}

template <int LEVEL>
static void run(const char* name, const float* dA, const unsigned short* dB, float* dout, unsigned long long* dst, int nk, int K) {
  auto kern = kstep_kernel<LEVEL>;
  const int lds = 2 * STAGE;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NT), lds, 0, dA, dB, dout, dst, nk, K);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(512);
  CHECK(hipMemcpy(h.data(), dst, 512 * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, us;
  for (int b = 0; b < 256; ++b) { cyc.push_back((double)h[2 * b] / nk); us.push_back((double)h[2 * b + 1] * 0.01 / nk); }
  std::sort(cyc.begin(), cyc.end());
  std::sort(us.begin(), us.end());
  const double c = cyc[128], u = us[128];
  hipFuncAttributes fa;
  CHECK(hipFuncGetAttributes(&fa, (const void*)kern));
  // one K-step of one workgroup = 2 * 128 * 256 * 32 fp32-equivalent FLOP; 256 workgroups
  printf("  level %d %-46s %7.0f cycles / K-step  %6.3f us  clock %5.0f MHz  MFMA-pipe %4.2f  %6.1f TF  vgpr %d\n", LEVEL, name, c, u, c / u,
         3072.0 / c, 2.0 * 128 * 256 * 32 * 256 / (u * 1e-6) / 1e12, fa.numRegs);
  fflush(stdout);
}

int main() {
  const int K = 2304, nk = 512;
  float* dA;
  unsigned short* dB;
  float* dout;
  unsigned long long* dst;
  CHECK(hipMalloc(&dA, (size_t)BM * K * 4));
  CHECK(hipMalloc(&dB, (size_t)3 * BN * K * 2));
  CHECK(hipMalloc(&dout, (size_t)256 * NT * 4));
  CHECK(hipMalloc(&dst, 512 * 8));
  std::vector<float> hA((size_t)BM * K);
  std::vector<unsigned short> hB((size_t)3 * BN * K);
  srand(1);
  for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : hB) v = (unsigned short)(0x3c00 + (rand() & 0x3ff));
  CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  printf("128x256 tile, 8 waves, 256 workgroups, %d K-steps; ideal = 3072 cycles per K-step (96 MFMAs of 32 cycles per SIMD)\n", nk);
  run<0>("MFMAs on register operands", dA, dB, dout, dst, nk, K);
  run<1>("+ fragment reads (ds_read_b128)", dA, dB, dout, dst, nk, K);
  run<2>("+ one barrier per K-step", dA, dB, dout, dst, nk, K);
  run<3>("+ LDS stores of the operand tiles", dA, dB, dout, dst, nk, K);
  run<4>("+ piece split of the activation tile", dA, dB, dout, dst, nk, K);
  run<5>("+ global loads (L2-resident)", dA, dB, dout, dst, nk, K);
  run<6>("B planes by LDS-DMA, counted wait + raw barrier", dA, dB, dout, dst, nk, K);
  run<7>("B planes by LDS-DMA, __syncthreads()", dA, dB, dout, dst, nk, K);
  return 0;
}
