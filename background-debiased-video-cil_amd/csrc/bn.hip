// BatchNorm2d (train / eval) on NHWC [M][C] fp32, fused with ReLU and the residual add, plus the
// backward.  HBM-bound column reductions: partial sums per row-block (fp32), fixed-order final
// reduction in fp64 (deterministic, no atomics).
//
// Replaces UPSTREAM mmaction ConvModule.bn / .activate and the Bottleneck/BasicBlock
// `out + identity -> relu`.
#include "common.h"

namespace {

constexpr int MAX_RB_TIMES_C = 524288;  // partial-slab capacity in floats per quantity

struct BnGrid {
  int CVB;  // float4 column vectors per block
  int RL;   // row lanes per block (CVB * RL == 256)
  int CC;   // column chunks
  int RB;   // row blocks
  int rows_per_block;
};

BnGrid bn_grid(int64_t M, int C) {
  BnGrid b;
  const int cv = C / 4;
  b.CVB = cv < 64 ? cv : 64;
  b.RL = 256 / b.CVB;
  b.CC = (cv + b.CVB - 1) / b.CVB;
  int64_t rb = (M + 127) / 128;
  const int64_t cap = 2048 / b.CC > 0 ? 2048 / b.CC : 1;
  if (rb > cap) rb = cap;
  if (rb < 1) rb = 1;
  while (rb * C > MAX_RB_TIMES_C) --rb;
  b.RB = (int)rb;
  b.rows_per_block = (int)((M + rb - 1) / rb);
  return b;
}

bool bn_c_ok(int C) { return C >= 64 && (C % 256 == 0 || C == 64 || C == 128); }

// ---- forward statistics ------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ y, float* __restrict__ psum,
                                                          float* __restrict__ psq, int64_t M, int C, int CVB, int RL,
                                                          int rows_per_block) {
  __shared__ float4 sh[2][256];
  const int tid = threadIdx.x;
  const int cv = tid % CVB, rl = tid / CVB;
  const int c4 = blockIdx.y * CVB + cv;  // float4 column index
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  const int CV = C / 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  const float4* yp = reinterpret_cast<const float4*>(y);
  for (int64_t r = r0 + rl; r < r1; r += RL) {
    const float4 v = yp[r * CV + c4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
  }
  sh[0][tid] = s;
  sh[1][tid] = q;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < RL; ++k) {
      const float4 a = sh[0][k * CVB + cv], b = sh[1][k * CVB + cv];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
    }
    reinterpret_cast<float4*>(psum)[(int64_t)blockIdx.x * CV + c4] = s;
    reinterpret_cast<float4*>(psq)[(int64_t)blockIdx.x * CV + c4] = q;
  }
}

// Fixed-order column sum of two [RB][C] fp32 slabs in fp64: 4 channels x FIN_LANES row-lanes per block (1024 threads).
// (Round 3 tried 16 channels per block with 16-byte row accesses -- a quarter of the memory transactions, the same summation order:
// 16.1 / 12.9 us per launch against 12.3 / 10.0 us for this form in the same profile; the kernel is bound by its fp64 adds and the
// load latency chain, which four times fewer blocks concentrate on four times fewer CUs.  Reverted; profiles/r03_notes.md.)
// The loop is a chain of dependent-latency loads (up to 25 088 partial rows for the stem), so the rows are spread over 256
// lanes per channel with four independent chains each; the lane sums are combined in a fixed order (16 groups of 16).
constexpr int FIN_LANES = 256;

__device__ __forceinline__ void slab_colsum2(const float* __restrict__ a, const float* __restrict__ b, int RB, int C, int c,
                                             int rl, double (*sh)[FIN_LANES][5], double& sa, double& sb) {
  double x = 0.0, y = 0.0;
  if (c < C) {
    double x1 = 0.0, x2 = 0.0, x3 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
    int r = rl;
    for (; r + 3 * FIN_LANES < RB; r += 4 * FIN_LANES) {
      const float a0 = a[(int64_t)r * C + c], a1 = a[(int64_t)(r + FIN_LANES) * C + c];
      const float a2 = a[(int64_t)(r + 2 * FIN_LANES) * C + c], a3 = a[(int64_t)(r + 3 * FIN_LANES) * C + c];
      const float b0 = b[(int64_t)r * C + c], b1 = b[(int64_t)(r + FIN_LANES) * C + c];
      const float b2 = b[(int64_t)(r + 2 * FIN_LANES) * C + c], b3 = b[(int64_t)(r + 3 * FIN_LANES) * C + c];
      x += (double)a0; x1 += (double)a1; x2 += (double)a2; x3 += (double)a3;
      y += (double)b0; y1 += (double)b1; y2 += (double)b2; y3 += (double)b3;
    }
    for (; r < RB; r += FIN_LANES) {
      x += (double)a[(int64_t)r * C + c];
      y += (double)b[(int64_t)r * C + c];
    }
    x = (x + x1) + (x2 + x3);
    y = (y + y1) + (y2 + y3);
  }
  const int cl = threadIdx.x & 3;
  sh[0][rl][cl] = x;
  sh[1][rl][cl] = y;
  __syncthreads();
  double u = 0.0, v = 0.0;
  if (rl < 16) {  // lane rl sums lanes 16 rl .. 16 rl + 15, in order
    for (int k = 0; k < 16; ++k) {
      u += sh[0][16 * rl + k][cl];
      v += sh[1][16 * rl + k][cl];
    }
  }
  __syncthreads();
  if (rl < 16) {
    sh[0][rl][cl] = u;
    sh[1][rl][cl] = v;
  }
  __syncthreads();
  sa = 0.0;
  sb = 0.0;
  if (rl == 0) {
    for (int k = 0; k < 16; ++k) {
      sa += sh[0][k][cl];
      sb += sh[1][k][cl];
    }
  }
}

__global__ __launch_bounds__(4 * FIN_LANES) void bn_finalize_kernel(const float* __restrict__ psum, const float* __restrict__ psq, int RB,
                                                           int64_t M, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                           float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double sh[2][FIN_LANES][5];
  const int c = blockIdx.x * 4 + (threadIdx.x & 3), rl = threadIdx.x >> 2;
  double s, q;
  slab_colsum2(psum, psq, RB, C, c, rl, sh, s, q);
  if (rl != 0 || c >= C) return;
  const double mean = s / (double)M;
  double var = q / (double)M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  save_mean[c] = meanf;
  save_invstd[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - meanf * sc;
  if (running_mean != nullptr) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_params_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// Streaming reads of tensors that are not touched again before they have left every cache (the raw conv output and the incoming
// gradient in the apply passes) are issued non-temporal, so that the tensors the next kernels read (the pass's own output) keep
// their place in L2 / Infinity Cache: 54.39 -> 53.51 ms per training step in one process (tools/ab_step.py).
// (second box: plain 55.20, level 1 54.68, level 2 54.50).  BDVCIL_BN_NT: 0 = plain loads, 1 = y / dout of bn_apply and
// bn_bwd_apply, 2 (default) = also the residual of bn_apply (the block input: next read in the backward pass).
static int bn_nt_level() {  // read per call (a getenv is nothing beside a launch): tools/ab_step.py flips it between rounds
  const char* e = getenv("BDVCIL_BN_NT");
  return e != nullptr ? atoi(e) : 2;
}
static bool bn_nt_enabled() { return bn_nt_level() != 0; }

// ---- apply -------------------------------------------------------------------------------
// ReLU sign mask: bit e of mask[] = (out[e] > 0) for flat element index e (one uint32 per 32 channels).
// Backward kernels read this 1-bit-per-element mask instead of re-reading the fp32 activation.
__device__ __forceinline__ unsigned nibble_gt0(const float4& o) {
  return (o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u);
}
__device__ __forceinline__ unsigned mask_nibble(const uint32_t* __restrict__ mask, int64_t i4) {
  return (mask[i4 >> 3] >> (4 * (int)(i4 & 7))) & 0xFu;
}
// ReLU sign bits of 4 channels derived from the conv output: the forward's own expression (bn_apply_kernel / the PRE loaders)
__device__ __forceinline__ unsigned derive_nibble(const float4 v, const float4 sc, const float4 sh) {
  return (fmaf(v.x, sc.x, sh.x) > 0.f ? 1u : 0u) | (fmaf(v.y, sc.y, sh.y) > 0.f ? 2u : 0u) | (fmaf(v.z, sc.z, sh.z) > 0.f ? 4u : 0u) |
         (fmaf(v.w, sc.w, sh.w) > 0.f ? 8u : 0u);
}
__device__ __forceinline__ float4 apply_nibble(float4 g, unsigned nib) {
  g.x = (nib & 1u) ? g.x : 0.f;
  g.y = (nib & 2u) ? g.y : 0.f;
  g.z = (nib & 4u) ? g.z : 0.f;
  g.w = (nib & 8u) ? g.w : 0.f;
  return g;
}

// res_scale / res_shift (optional): the residual is itself a raw conv output that still needs its own BatchNorm
// (the downsample branch of a block): r = res * res_scale + res_shift is formed here instead of in a pass of its own.
// ES: bytes per element of y / res / out (common.h: 4 = fp32, 2 = bf16 storage)
template <bool RELU, bool RES, bool NT = false, bool NTR = false, int ES = 4>
__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ y, const float4* __restrict__ scale,
                                                        const float4* __restrict__ shift, const void* __restrict__ res,
                                                        const float4* __restrict__ res_scale, const float4* __restrict__ res_shift,
                                                        void* __restrict__ out, uint32_t* __restrict__ mask, int64_t n4, int CV) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // A thread's unit is 16 bytes = U groups of 4 channels (fp32: 1, bf16: 2).  n4 is a multiple of 8 (C % 32 == 0) and so is the
  // stride: the 8 / U lanes of one mask word stay together.
  constexpr int U = ActU<ES>::value;
  constexpr int LW = 8 / U;          // lanes per mask word
  for (int64_t iu = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; iu < n4 / U; iu += stride) {
    float4 v[U], r[U], o[U];
    act_ld16<ES, NT>(y, iu, v);
    if (RES) act_ld16<ES, NTR>(res, iu, r);
    unsigned bits = 0u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c4 = (int)((iu * U + u) % CV);
      const float4 sc = scale[c4], sh = shift[c4];
      o[u].x = fmaf(v[u].x, sc.x, sh.x);   // explicit: the conv loaders that apply this BatchNorm themselves (PRE) use the same expression
      o[u].y = fmaf(v[u].y, sc.y, sh.y);
      o[u].z = fmaf(v[u].z, sc.z, sh.z);
      o[u].w = fmaf(v[u].w, sc.w, sh.w);
      if (RES) {
        float4 q = r[u];
        if (res_scale != nullptr) {
          const float4 rs = res_scale[c4], rb = res_shift[c4];
          q.x = q.x * rs.x + rb.x; q.y = q.y * rs.y + rb.y; q.z = q.z * rs.z + rb.z; q.w = q.w * rs.w + rb.w;
        }
        o[u].x += q.x; o[u].y += q.y; o[u].z += q.z; o[u].w += q.w;
      }
      if (RELU) {
        bits |= nibble_gt0(o[u]) << (4 * u);
        o[u].x = fmaxf(o[u].x, 0.f); o[u].y = fmaxf(o[u].y, 0.f); o[u].z = fmaxf(o[u].z, 0.f); o[u].w = fmaxf(o[u].w, 0.f);
      }
    }
    if (RELU && mask != nullptr) {
      unsigned m = bits << (4 * U * (threadIdx.x & (LW - 1)));
      m |= __shfl_xor(m, 1, 64);
      m |= __shfl_xor(m, 2, 64);
      if (LW == 8) m |= __shfl_xor(m, 4, 64);
      if ((threadIdx.x & (LW - 1)) == 0) mask[iu / LW] = m;
    }
    act_st16<ES>(out, iu, o);
  }
}

// ---- backward ----------------------------------------------------------------------------
// RELU: 0 = none, 1 = the forward's 1-bit mask, 2 = the sign derived from y (rscale / rshift: the unit's mask was never written)
template <int RELU, int ES = 4>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const void* __restrict__ dout, const uint32_t* __restrict__ mask,
                                                              const void* __restrict__ y, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* __restrict__ p1,
                                                              float* __restrict__ p2, int64_t M, int C, int CVB, int RL,
                                                              int rows_per_block, const float* __restrict__ rscale,
                                                              const float* __restrict__ rshift) {
  __shared__ float4 sh[2][256];
  const int tid = threadIdx.x;
  const int cv = tid % CVB, rl = tid / CVB;
  const int c4 = blockIdx.y * CVB + cv;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  const int CV = C / 4;
  const float4 mu = reinterpret_cast<const float4*>(mean)[c4];
  const float4 is = reinterpret_cast<const float4*>(invstd)[c4];
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int64_t r = r0 + rl; r < r1; r += RL) {
    const int64_t i = r * CV + c4;
    float4 g = act_ld4<ES>(dout, i);
    const float4 v = act_ld4<ES>(y, i);
    if (RELU == 1) g = apply_nibble(g, mask_nibble(mask, i));
    if (RELU == 2) g = apply_nibble(g, derive_nibble(v, reinterpret_cast<const float4*>(rscale)[c4], reinterpret_cast<const float4*>(rshift)[c4]));
    s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
    s2.x += g.x * ((v.x - mu.x) * is.x);
    s2.y += g.y * ((v.y - mu.y) * is.y);
    s2.z += g.z * ((v.z - mu.z) * is.z);
    s2.w += g.w * ((v.w - mu.w) * is.w);
  }
  sh[0][tid] = s1;
  sh[1][tid] = s2;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < RL; ++k) {
      const float4 a = sh[0][k * CVB + cv], b = sh[1][k * CVB + cv];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
    }
    reinterpret_cast<float4*>(p1)[(int64_t)blockIdx.x * CV + c4] = s1;
    reinterpret_cast<float4*>(p2)[(int64_t)blockIdx.x * CV + c4] = s2;
  }
}

// coef[0][c] = gamma*invstd, coef[1][c] = sum(g)/M, coef[2][c] = sum(g*xhat)/M
__global__ __launch_bounds__(4 * FIN_LANES) void bn_bwd_finalize_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int RB,
                                                               int64_t M, int C, const float* __restrict__ gamma,
                                                               const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float beta_acc, float* __restrict__ coef) {
  __shared__ double sh[2][FIN_LANES][5];
  const int c = blockIdx.x * 4 + (threadIdx.x & 3), rl = threadIdx.x >> 2;
  double s1, s2;
  slab_colsum2(p1, p2, RB, C, c, rl, sh, s1, s2);
  if (rl != 0 || c >= C) return;
  if (dgamma != nullptr) dgamma[c] = (beta_acc != 0.f ? beta_acc * dgamma[c] : 0.f) + (float)s2;
  if (dbeta != nullptr) dbeta[c] = (beta_acc != 0.f ? beta_acc * dbeta[c] : 0.f) + (float)s1;
  const float a = gamma[c] * invstd[c];
  coef[c] = a;
  coef[C + c] = (float)(s1 / (double)M);
  coef[2 * C + c] = (float)(s2 / (double)M);
}

template <int RELU, bool NT = false, int ES = 4>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ dout, const uint32_t* __restrict__ mask,
                                                            const void* __restrict__ y, const float4* __restrict__ mean,
                                                            const float4* __restrict__ invstd, const float4* __restrict__ coef,
                                                            void* __restrict__ dy, int64_t n4, int CV,
                                                            const float4* __restrict__ rscale, const float4* __restrict__ rshift) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  constexpr int U = ActU<ES>::value;   // 16-byte units: U groups of 4 channels per thread (bn_apply_kernel)
  for (int64_t iu = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; iu < n4 / U; iu += stride) {
    float4 gg[U], vv[U], dd[U];
    act_ld16<ES, NT>(dout, iu, gg);
    act_ld16<ES, NT>(y, iu, vv);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = iu * U + u;
      const int c4 = (int)(i % CV);
      float4 g = gg[u];
      const float4 v = vv[u], mu = mean[c4], is = invstd[c4];
      if (RELU == 1) g = apply_nibble(g, mask_nibble(mask, i));
      if (RELU == 2) g = apply_nibble(g, derive_nibble(v, rscale[c4], rshift[c4]));
      const float4 a = coef[c4], b = coef[CV + c4], c = coef[2 * CV + c4];
      float4 d;
      d.x = a.x * (g.x - b.x - ((v.x - mu.x) * is.x) * c.x);
      d.y = a.y * (g.y - b.y - ((v.y - mu.y) * is.y) * c.y);
      d.z = a.z * (g.z - b.z - ((v.z - mu.z) * is.z) * c.z);
      d.w = a.w * (g.w - b.w - ((v.w - mu.w) * is.w) * c.w);
      dd[u] = d;
    }
    act_st16<ES>(dy, iu, dd);
  }
}

// ---- BatchNorm(+ReLU) backward behind a MaxPool2d(3, 2, 1) (the stem) --------------------------------------------
// The gradient entering the BN+ReLU is the max-pool backward of dpool; it is gathered on the fly (pool_bwd_gather2x2)
// in both passes instead of being materialised: thread = (2x2 input block, 4 channels), grid.y strides (frame, block
// row).  256 % CV == 0, so a thread keeps its 4 channels and the per-block reduction groups threads by tid % CV.
template <bool APPLY, int ESD = 4>
__global__ __launch_bounds__(256) void bn_bwd_pool_kernel(const void* __restrict__ dpool, const uchar4* __restrict__ idx,
                                                           const uint32_t* __restrict__ mask, const float4* __restrict__ y,
                                                           const float4* __restrict__ mean, const float4* __restrict__ invstd,
                                                           const float4* __restrict__ coef, float4* __restrict__ dy,
                                                           float* __restrict__ p1, float* __restrict__ p2, int N, int H, int W,
                                                           int CV, int Ho, int Wo) {
  __shared__ float4 sh[2][256];
  const int tid = threadIdx.x;
  const int c4 = tid % CV;
  const float4 mu = mean[c4], is = invstd[c4];
  float4 ca, cb, cc;
  if (APPLY) {
    ca = coef[c4];
    cb = coef[CV + c4];
    cc = coef[2 * CV + c4];
  }
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int row = blockIdx.y; row < N * Ho; row += gridDim.y) {
    const int n = row / Ho, i = row - n * Ho;
    const int64_t obase = (int64_t)n * Ho * Wo * CV;
    const int64_t ibase = (int64_t)n * H * W * CV;
    const bool down = i + 1 < Ho, h1ok = 2 * i + 1 < H;
    for (int q = blockIdx.x * blockDim.x + tid; q < Wo * CV; q += gridDim.x * blockDim.x) {
      const int j = q / CV;  // q % CV == c4
      const bool right = j + 1 < Wo, w1ok = 2 * j + 1 < W;
      float4 g[4];
      pool_bwd_gather2x2<ESD>(dpool, idx, obase + (i * Wo + j) * CV + c4, CV, Wo, right, down, g);
      const int64_t p00 = ibase + ((2 * i) * W + 2 * j) * CV + c4;
      const int64_t pix[4] = {p00, p00 + CV, p00 + (int64_t)W * CV, p00 + (int64_t)(W + 1) * CV};
      const bool ok[4] = {true, w1ok, h1ok, h1ok && w1ok};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!ok[k]) continue;
        const float4 gv = apply_nibble(g[k], mask_nibble(mask, pix[k]));
        const float4 v = y[pix[k]];
        const float4 xh = make_float4((v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w);
        if (APPLY) {
          float4 d;
          d.x = ca.x * (gv.x - cb.x - xh.x * cc.x);
          d.y = ca.y * (gv.y - cb.y - xh.y * cc.y);
          d.z = ca.z * (gv.z - cb.z - xh.z * cc.z);
          d.w = ca.w * (gv.w - cb.w - xh.w * cc.w);
          dy[pix[k]] = d;
        } else {
          s1.x += gv.x; s1.y += gv.y; s1.z += gv.z; s1.w += gv.w;
          s2.x += gv.x * xh.x; s2.y += gv.y * xh.y; s2.z += gv.z * xh.z; s2.w += gv.w * xh.w;
        }
      }
    }
  }
  if (!APPLY) {
    sh[0][tid] = s1;
    sh[1][tid] = s2;
    __syncthreads();
    if (tid < CV) {
      for (int k = 1; k < 256 / CV; ++k) {
        const float4 a = sh[0][k * CV + tid], b = sh[1][k * CV + tid];
        s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
        s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
      }
      const int64_t slab_row = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
      reinterpret_cast<float4*>(p1)[slab_row * CV + tid] = s1;
      reinterpret_cast<float4*>(p2)[slab_row * CV + tid] = s2;
    }
  }
}

template <int ES = 4>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const void* __restrict__ dout, const uint32_t* __restrict__ mask,
                                                        const void* __restrict__ add, void* __restrict__ g, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 d = apply_nibble(act_ld4<ES>(dout, i), mask_nibble(mask, i));
    if (add != nullptr) {
      const float4 a = act_ld4<ES>(add, i);
      d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
    }
    act_st4<ES>(g, i, d);
  }
}

template <int ES = 4>
__global__ __launch_bounds__(256) void add_kernel(const void* __restrict__ a, const void* __restrict__ b,
                                                   void* __restrict__ out, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 x = act_ld4<ES>(a, i), y = act_ld4<ES>(b, i);
    act_st4<ES>(out, i, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w));
  }
}

int ew_grid(int64_t n4) {
  int64_t b = (n4 + 255) / 256;
  // Blocks per launch of the grid-stride passes.  A cap of 4096 blocks (16 MB between a thread's successive accesses) cost the
  // passes over the large tensors a quarter of their bandwidth: the 822 MB layer-1 apply pass with a residual 548 us (4.5 TB/s)
  // capped at 4096, 454 us at 65536, 416 us (5.9 TB/s) with one unit per thread; all apply passes of a step 6.00 -> 5.05 ms,
  // backward 11.93 -> 11.21 ms (tools/bench_bn.py), 589.4 -> 601.9 clips/s in the step (two alternating pairs, one box).
  // (round 2's BDVCIL_EW_BLOCKS switch is gone: the A/B is recorded in profiles/r02_ab_streams.txt)
  if (b > (1 << 20)) b = 1 << 20;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" size_t bdv_bn_workspace_bytes(int64_t M, int C) {
  (void)M;
  // two partial slabs (<= MAX_RB_TIMES_C floats each) + 3*C coefficients
  return (size_t)(2 * MAX_RB_TIMES_C + 3 * (size_t)(C > 0 ? C : 0)) * sizeof(float);
}

extern "C" int bdv_bn_train_stats(const float* y, int64_t M, int C, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, float* save_mean,
                                  float* save_invstd, float* scale, float* shift, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  BDV_REQUIRE(y && gamma && beta && save_mean && save_invstd && scale && shift && workspace, "bdv_bn_train_stats: null pointer");
  BDV_REQUIRE(M > 0 && bn_c_ok(C), "bdv_bn_train_stats: unsupported M=%lld C=%d", (long long)M, C);
  BDV_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bdv_bn_train_stats: running stats must come in pairs");
  BDV_REQUIRE(bdv_aligned16(y) && bdv_aligned16(workspace), "bdv_bn_train_stats: alignment");
  if (workspace_bytes < bdv_bn_workspace_bytes(M, C)) {
    bdv_set_error("bdv_bn_train_stats: workspace too small");
    return BDV_EWORKSPACE;
  }
  const BnGrid b = bn_grid(M, C);
  float* psum = (float*)workspace;
  float* psq = psum + MAX_RB_TIMES_C;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_partial_kernel, dim3(b.RB, b.CC), dim3(256), 0, s, y, psum, psq, M, C, b.CVB, b.RL, b.rows_per_block);
  BDV_LAUNCH_CHECK("bdv_bn_train_stats(partial)");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(4 * FIN_LANES), 0, s, (const float*)psum, (const float*)psq, b.RB,
                     M, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift);
  BDV_LAUNCH_CHECK("bdv_bn_train_stats(finalize)");
  return BDV_OK;
}

extern "C" int bdv_bn_train_finalize(const float* partial, int rows, int64_t M, int C, const float* gamma, const float* beta,
                                     float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                                     float* save_invstd, float* scale, float* shift, void* stream) {
  BDV_REQUIRE(partial && gamma && beta && save_mean && save_invstd && scale && shift, "bdv_bn_train_finalize: null pointer");
  BDV_REQUIRE(rows > 0 && M > 0 && C > 0 && C % 4 == 0, "bdv_bn_train_finalize: bad shape");
  BDV_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bdv_bn_train_finalize: running stats must come in pairs");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(4 * FIN_LANES), 0, (hipStream_t)stream, partial,
                     partial + (size_t)rows * C, rows, M, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean,
                     save_invstd, scale, shift);
  BDV_LAUNCH_CHECK("bdv_bn_train_finalize");
  return BDV_OK;
}

extern "C" int bdv_bn_eval_params(int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, void* stream) {
  BDV_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bdv_bn_eval_params: bad argument");
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  BDV_LAUNCH_CHECK("bdv_bn_eval_params");
  return BDV_OK;
}

extern "C" int bdv_bn_apply(const void* y, const float* scale, const float* shift, const void* res, const float* res_scale,
                            const float* res_shift, void* out, uint32_t* relu_mask, int64_t M, int C, int relu, int act_dtype,
                            void* stream) {
  BDV_REQUIRE(y && scale && shift && out, "bdv_bn_apply: null pointer");
  BDV_REQUIRE_ACT(act_dtype, "bdv_bn_apply");
  BDV_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (res_scale == nullptr || res != nullptr),
              "bdv_bn_apply: res_scale and res_shift come together and need res");
  BDV_REQUIRE(bdv_aligned16(res_scale) && bdv_aligned16(res_shift), "bdv_bn_apply: alignment");
  BDV_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bdv_bn_apply: bad shape");
  BDV_REQUIRE(relu_mask == nullptr || (relu && C % 32 == 0), "bdv_bn_apply: relu_mask needs relu and C %% 32 == 0");
  BDV_REQUIRE(bdv_aligned16(y) && bdv_aligned16(out) && bdv_aligned16(scale) && bdv_aligned16(shift) &&
                  (res == nullptr || bdv_aligned16(res)), "bdv_bn_apply: alignment");
  BDV_REQUIRE(act_dtype == BDV_ACT_F32 || C % 8 == 0, "bdv_bn_apply: bf16 tensors need C %% 8 == 0 (16-byte units)");
  const int64_t n4 = M * C / 4;
  const int CV = C / 4;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(ew_grid(act_dtype == BDV_ACT_BF16 ? n4 / 2 : n4 / BDV_F32_UNITS)), blk(256);
  const float4 *sc = (const float4*)scale, *sh = (const float4*)shift;
  const float4 *rs = (const float4*)res_scale, *rb = (const float4*)res_shift;
#define BDV_APPLY(...) hipLaunchKernelGGL((bn_apply_kernel<__VA_ARGS__, ES>), grid, blk, 0, s, y, sc, sh, res, rs, rb, out, relu_mask, n4, CV)
  BDV_ACT_SWITCH(act_dtype, ES, {
    if (relu && res && bn_nt_level() >= 2) BDV_APPLY(true, true, true, true);
    else if (relu && res && bn_nt_enabled()) BDV_APPLY(true, true, true, false);
    else if (relu && bn_nt_enabled()) BDV_APPLY(true, false, true, false);
    else if (relu && res) BDV_APPLY(true, true, false, false);
    else if (relu) BDV_APPLY(true, false, false, false);
    else if (res) BDV_APPLY(false, true, false, false);
    else BDV_APPLY(false, false, false, false);
  });
#undef BDV_APPLY
  BDV_LAUNCH_CHECK("bdv_bn_apply");
  return BDV_OK;
}

extern "C" int bdv_bn_backward(const void* dout, const uint32_t* relu_mask, const void* y, const float* gamma,
                               const float* save_mean, const float* save_invstd, void* dy, float* dgamma,
                               float* dbeta, float beta_acc, int64_t M, int C, int relu, const float* stat_partial,
                               int stat_rows, const float* relu_scale, const float* relu_shift, void* workspace,
                               size_t workspace_bytes, int act_dtype, void* stream) {
  BDV_REQUIRE(dout && y && gamma && save_mean && save_invstd && dy && workspace, "bdv_bn_backward: null pointer");
  BDV_REQUIRE_ACT(act_dtype, "bdv_bn_backward");
  BDV_REQUIRE(stat_partial == nullptr || (stat_rows > 0 && bdv_aligned16(stat_partial)), "bdv_bn_backward: bad stat_partial");
  const bool derive = relu && relu_mask == nullptr && relu_scale != nullptr;
  BDV_REQUIRE(!relu || derive || (relu_mask && C % 32 == 0), "bdv_bn_backward: relu needs the forward ReLU mask (C %% 32 == 0) or relu_scale / relu_shift");
  BDV_REQUIRE(!derive || (relu_shift != nullptr && bdv_aligned16(relu_scale) && bdv_aligned16(relu_shift)), "bdv_bn_backward: relu_scale / relu_shift");
  BDV_REQUIRE(M > 0 && bn_c_ok(C), "bdv_bn_backward: unsupported M=%lld C=%d", (long long)M, C);
  BDV_REQUIRE(bdv_aligned16(dout) && bdv_aligned16(y) && bdv_aligned16(dy) && bdv_aligned16(workspace) &&
                  bdv_aligned16(save_mean) && bdv_aligned16(save_invstd), "bdv_bn_backward: alignment");
  if (workspace_bytes < bdv_bn_workspace_bytes(M, C)) {
    bdv_set_error("bdv_bn_backward: workspace too small");
    return BDV_EWORKSPACE;
  }
  const BnGrid b = bn_grid(M, C);
  float* p1 = (float*)workspace;
  float* p2 = p1 + MAX_RB_TIMES_C;
  float* coef = p2 + MAX_RB_TIMES_C;
  hipStream_t s = (hipStream_t)stream;
  const float *q1 = p1, *q2 = p2;
  int rows = b.RB;
  if (stat_partial != nullptr) {  // the producer of dout (a dgrad epilogue) already reduced per 128-row tile
    q1 = stat_partial;
    q2 = stat_partial + (size_t)stat_rows * C;
    rows = stat_rows;
  } else {
#define BDV_BWD_PARTIAL(RELU_)                                                                                                  \
  hipLaunchKernelGGL((bn_bwd_partial_kernel<RELU_, ES>), dim3(b.RB, b.CC), dim3(256), 0, s, dout, relu_mask, y, save_mean,     \
                     save_invstd, p1, p2, M, C, b.CVB, b.RL, b.rows_per_block, relu_scale, relu_shift)
    BDV_ACT_SWITCH(act_dtype, ES, {
      if (derive) BDV_BWD_PARTIAL(2);
      else if (relu) BDV_BWD_PARTIAL(1);
      else BDV_BWD_PARTIAL(0);
    });
#undef BDV_BWD_PARTIAL
    BDV_LAUNCH_CHECK("bdv_bn_backward(partial)");
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(4 * FIN_LANES), 0, s, q1, q2, rows, M, C, gamma, save_invstd, dgamma,
                     dbeta, beta_acc, coef);
  BDV_LAUNCH_CHECK("bdv_bn_backward(finalize)");
  BDV_REQUIRE(act_dtype == BDV_ACT_F32 || C % 8 == 0, "bdv_bn_backward: bf16 tensors need C %% 8 == 0 (16-byte units)");
  const int64_t n4 = M * C / 4;
  const dim3 grid(ew_grid(act_dtype == BDV_ACT_BF16 ? n4 / 2 : n4 / BDV_F32_UNITS)), blk(256);
  const float4 *rs4 = (const float4*)relu_scale, *rh4 = (const float4*)relu_shift;
#define BDV_BWD_APPLY(RELU_, NT_)                                                                                              \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<RELU_, NT_, ES>), grid, blk, 0, s, dout, relu_mask, y,                               \
                     (const float4*)save_mean, (const float4*)save_invstd, (const float4*)coef, dy, n4, C / 4, rs4, rh4)
  const bool nt = bn_nt_enabled();
  BDV_ACT_SWITCH(act_dtype, ES, {
    if (derive) { if (nt) BDV_BWD_APPLY(2, true); else BDV_BWD_APPLY(2, false); }
    else if (relu) { if (nt) BDV_BWD_APPLY(1, true); else BDV_BWD_APPLY(1, false); }
    else BDV_BWD_APPLY(0, false);
  });
#undef BDV_BWD_APPLY
  BDV_LAUNCH_CHECK("bdv_bn_backward(apply)");
  return BDV_OK;
}

extern "C" int bdv_bn_backward_maxpool(const void* dpool, const uint8_t* pool_idx, const uint32_t* relu_mask, const float* y,
                                       const float* gamma, const float* save_mean, const float* save_invstd, float* dy,
                                       float* dgamma, float* dbeta, float beta_acc, int N, int H, int W, int C, void* workspace,
                                       size_t workspace_bytes, int dpool_dtype, void* stream) {
  BDV_REQUIRE(dpool && pool_idx && relu_mask && y && gamma && save_mean && save_invstd && dy && workspace,
              "bdv_bn_backward_maxpool: null pointer");
  BDV_REQUIRE_ACT(dpool_dtype, "bdv_bn_backward_maxpool");
  BDV_REQUIRE(N > 0 && H > 1 && W > 1 && C >= 32 && C % 32 == 0 && 256 % (C / 4) == 0,
              "bdv_bn_backward_maxpool: unsupported shape (C must be 32, 64, 128, 256, 512 or 1024)");
  BDV_REQUIRE(bdv_aligned16(dpool) && bdv_aligned16(y) && bdv_aligned16(dy) && bdv_aligned16(workspace) &&
                  bdv_aligned16(save_mean) && bdv_aligned16(save_invstd) && (((uintptr_t)pool_idx) & 3) == 0,
              "bdv_bn_backward_maxpool: alignment");
  const int64_t M = (int64_t)N * H * W;
  BDV_REQUIRE(M * (C / 4) < (1ll << 31), "bdv_bn_backward_maxpool: tensor too large");
  if (workspace_bytes < bdv_bn_workspace_bytes(M, C)) {
    bdv_set_error("bdv_bn_backward_maxpool: workspace too small");
    return BDV_EWORKSPACE;
  }
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, CV = C / 4;
  const int gx = (Wo * CV + 255) / 256;
  int gy = N * Ho;
  const int cap = MAX_RB_TIMES_C / C / gx;  // slab rows = gx * gy
  if (gy > cap) gy = cap;
  if (gy > 4096) gy = 4096;
  BDV_REQUIRE(gy >= 1, "bdv_bn_backward_maxpool: row too wide for the partial slab");
  float* p1 = (float*)workspace;
  float* p2 = p1 + MAX_RB_TIMES_C;
  float* coef = p2 + MAX_RB_TIMES_C;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(gx, gy), blk(256);
  BDV_ACT_SWITCH(dpool_dtype, ES, hipLaunchKernelGGL((bn_bwd_pool_kernel<false, ES>), grid, blk, 0, s, dpool, (const uchar4*)pool_idx, relu_mask,
                     (const float4*)y, (const float4*)save_mean, (const float4*)save_invstd, (const float4*)nullptr,
                     (float4*)nullptr, p1, p2, N, H, W, CV, Ho, Wo));
  BDV_LAUNCH_CHECK("bdv_bn_backward_maxpool(partial)");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(4 * FIN_LANES), 0, s, (const float*)p1, (const float*)p2, gx * gy, M, C,
                     gamma, save_invstd, dgamma, dbeta, beta_acc, coef);
  BDV_LAUNCH_CHECK("bdv_bn_backward_maxpool(finalize)");
  BDV_ACT_SWITCH(dpool_dtype, ES, hipLaunchKernelGGL((bn_bwd_pool_kernel<true, ES>), grid, blk, 0, s, dpool, (const uchar4*)pool_idx, relu_mask,
                     (const float4*)y, (const float4*)save_mean, (const float4*)save_invstd, (const float4*)coef, (float4*)dy,
                     (float*)nullptr, (float*)nullptr, N, H, W, CV, Ho, Wo));
  BDV_LAUNCH_CHECK("bdv_bn_backward_maxpool(apply)");
  return BDV_OK;
}

extern "C" int bdv_relu_bwd(const void* dout, const uint32_t* relu_mask, const void* add, void* g, int64_t numel, int act_dtype,
                            void* stream) {
  BDV_REQUIRE(dout && relu_mask && g && numel > 0 && numel % 32 == 0, "bdv_relu_bwd: bad argument");
  BDV_REQUIRE_ACT(act_dtype, "bdv_relu_bwd");
  BDV_REQUIRE(bdv_aligned16(dout) && bdv_aligned16(g) && (!add || bdv_aligned16(add)), "bdv_relu_bwd: alignment");
  const int64_t n4 = numel / 4;
  BDV_ACT_SWITCH(act_dtype, ES, hipLaunchKernelGGL((relu_bwd_kernel<ES>), dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, dout,
                     relu_mask, add, g, n4));
  BDV_LAUNCH_CHECK("bdv_relu_bwd");
  return BDV_OK;
}

extern "C" int bdv_add(const void* a, const void* b, void* out, int64_t numel, int act_dtype, void* stream) {
  BDV_REQUIRE(a && b && out && numel > 0 && numel % 4 == 0, "bdv_add: bad argument");
  BDV_REQUIRE_ACT(act_dtype, "bdv_add");
  BDV_REQUIRE(bdv_aligned16(a) && bdv_aligned16(b) && bdv_aligned16(out), "bdv_add: alignment");
  const int64_t n4 = numel / 4;
  BDV_ACT_SWITCH(act_dtype, ES, hipLaunchKernelGGL((add_kernel<ES>), dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n4));
  BDV_LAUNCH_CHECK("bdv_add");
  return BDV_OK;
}
