// RandAugment on uint8 frames resident in HBM (the augmentation stage of the data path
// that feeds the fused background-mix front-end).
//
// Replaces libs/pipelines/rand_augment.py:17-160 (the fifteen operations of augment_list(), :163-220) as the reference
// applies them in RandAugment._rand_aug (:237-264): one operation and magnitude per clip, the same for every frame of
// the clip.  The reference runs them through Pillow one frame at a time in a DataLoader worker; here a whole batch of
// clips (B, T, H, W, 3) goes through one "slot" (= one of the n operations of every clip) in three launches.  Results
// are bit-identical to Pillow's C routines, whose arithmetic each device function restates:
//   ImageOps.autocontrast / equalize / solarize / posterize   per-channel 256-entry tables (double / integer arithmetic)
//   ImageEnhance.Color / Contrast / Brightness / Sharpness     Image.blend(degenerate, image, factor): single-precision
//                                                              in1 + alpha * (in2 - in1), truncated; L = ITU-R 601-2 in
//                                                              16.16 fixed point; SMOOTH = 3x3 float kernel / 13, edges kept
//   Image.transform(AFFINE, nearest) / Image.rotate            16.16 fixed-point source coordinates (affine_fixed), or the
//                                                              accumulated double coordinates of ImagingScaleAffine when
//                                                              the matrix has no cross terms (translations)
//   ImageDraw.rectangle                                        inclusive integer rectangle (CutoutAbs)
// HBM-bound byte work: no MFMA here.
#include "common.h"

// Pillow's C code is compiled without fused multiply-add: every product below must be rounded before it is added.
// Contraction is switched off for this translation unit (and by -ffp-contract=off in the Makefile); the arithmetic is
// written with plain operators because HIP's __fmul_rn / __dadd_rn are header inlines that carry the header's
// contraction setting with them and do get fused.
#pragma clang fp contract(off)

namespace {

enum AugOp : int {
  AUG_IDENTITY = 0,
  AUG_AUTOCONTRAST = 1,
  AUG_EQUALIZE = 2,
  AUG_SOLARIZE = 3,     // d[0] = threshold
  AUG_POSTERIZE = 4,    // i[1] = bits
  AUG_COLOR = 5,        // d[0] = factor
  AUG_CONTRAST = 6,     // d[0] = factor
  AUG_BRIGHTNESS = 7,   // d[0] = factor
  AUG_SHARPNESS = 8,    // d[0] = factor
  AUG_AFFINE_FIXED = 9, // i[1..6] = a0 a1 a2 a3 a4 a5 in 16.16 fixed point, i[7] = fill RGB
  AUG_AFFINE_SCALE = 10,// d[0..3] = a[0] a[2] a[4] a[5], i[7] = fill RGB
  AUG_CUTOUT = 11,      // i[1..4] = x0 y0 x1 y1 (inclusive), i[7] = fill RGB
  AUG_NUM_OPS = 12
};

constexpr int IP = 8, DP = 4;  // int / double parameters per clip

__device__ __forceinline__ bool needs_hist(int op) { return op == AUG_AUTOCONTRAST || op == AUG_EQUALIZE || op == AUG_CONTRAST; }
__device__ __forceinline__ bool is_lut(int op) { return op >= AUG_AUTOCONTRAST && op <= AUG_POSTERIZE; }

// ImagingConvert rgb2l: L24(rgb) >> 16 with rounding
__device__ __forceinline__ unsigned gray_of(unsigned r, unsigned g, unsigned b) {
  return (r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16;
}

// Blend.c, 0 <= alpha <= 1: (UINT8)((int)in1 + alpha * ((int)in2 - (int)in1)) in single precision, no contraction
__device__ __forceinline__ unsigned char blend_u8(int deg, int img, float alpha) {
  const float r = (float)deg + alpha * (float)(img - deg);
  return (unsigned char)(int)r;
}

// ---- pass 1: per-frame histograms of R, G, B and L for the clips whose operation needs them -------------------------
// grid (chunks, B*T); hist (B*T, 4, 256) zeroed by the launcher
__global__ __launch_bounds__(256) void aug_hist_kernel(const unsigned char* __restrict__ in, const int* __restrict__ op_i,
                                                        unsigned* __restrict__ hist, int T, int HW) {
  const int frame = blockIdx.y, clip = frame / T;
  if (!needs_hist(op_i[clip * IP])) return;
  __shared__ unsigned h[4 * 256];
  for (int i = threadIdx.x; i < 4 * 256; i += 256) h[i] = 0;
  __syncthreads();
  const unsigned char* src = in + (size_t)frame * HW * 3;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const unsigned r = src[3 * p], g = src[3 * p + 1], b = src[3 * p + 2];
    atomicAdd(&h[r], 1u);
    atomicAdd(&h[256 + g], 1u);
    atomicAdd(&h[512 + b], 1u);
    atomicAdd(&h[768 + gray_of(r, g, b)], 1u);
  }
  __syncthreads();
  unsigned* dst = hist + (size_t)frame * 1024;
  for (int i = threadIdx.x; i < 4 * 256; i += 256)
    if (h[i]) atomicAdd(&dst[i], h[i]);
}

// ---- pass 2: per-frame tables ------------------------------------------------------------------------------------------
// grid B*T, 256 threads.  lut (B*T, 3, 256) for the table operations; mean (B*T) for Contrast; tab (B, W + H) source
// column / row of every output column / row for AFFINE_SCALE (-1 = outside), written by the clip's first frame.
__global__ __launch_bounds__(256) void aug_table_kernel(const int* __restrict__ op_i, const double* __restrict__ op_d,
                                                         const unsigned* __restrict__ hist, unsigned char* __restrict__ lut,
                                                         int* __restrict__ mean, int* __restrict__ tab, int T, int H, int W) {
  const int frame = blockIdx.x, clip = frame / T, tid = threadIdx.x;
  const int op = op_i[clip * IP];
  const unsigned* h = hist + (size_t)frame * 1024;
  unsigned char* L = lut + (size_t)frame * 768;
  __shared__ int lo[3], hi[3];
  __shared__ long long eq_step[3];
  if (op == AUG_AUTOCONTRAST) {
    // ImageOps.autocontrast(cutoff=0): lo / hi = first / last occupied bin; identity when hi <= lo;
    // lut[ix] = clip(int(ix * scale + offset)), scale = 255.0 / (hi - lo), offset = -lo * scale (doubles, two roundings)
    if (tid < 3) {
      int l = 0, hgh = 255;
      while (l < 256 && h[tid * 256 + l] == 0) ++l;
      while (hgh >= 0 && h[tid * 256 + hgh] == 0) --hgh;
      lo[tid] = l;
      hi[tid] = hgh;
    }
    __syncthreads();
    for (int c = 0; c < 3; ++c) {
      int v = tid;
      if (hi[c] > lo[c]) {
        const double scale = 255.0 / (double)(hi[c] - lo[c]);
        const double offset = -(double)lo[c] * scale;
        const double t = (double)tid * scale + offset;
        v = (int)t;  // toward zero, as Python's int()
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
      }
      L[c * 256 + tid] = (unsigned char)v;
    }
  } else if (op == AUG_EQUALIZE) {
    // ImageOps.equalize: step = (sum of occupied bins - last occupied bin) // 255; identity when fewer than two bins are
    // occupied or step == 0; lut[i] = (step // 2 + sum_{j<i} h[j]) // step, clipped to 255 by Image.point
    if (tid < 3) {
      long long total = 0, last = 0;
      int occupied = 0;
      for (int i = 0; i < 256; ++i) {
        const unsigned v = h[tid * 256 + i];
        if (v) {
          ++occupied;
          total += v;
          last = v;
        }
      }
      const long long step = occupied <= 1 ? 0 : (total - last) / 255;
      eq_step[tid] = step;
      if (step) {
        long long n = step / 2;
        for (int i = 0; i < 256; ++i) {
          const long long q = n / step;
          L[tid * 256 + i] = (unsigned char)(q > 255 ? 255 : q);
          n += h[tid * 256 + i];
        }
      }
    }
    __syncthreads();
    for (int c = 0; c < 3; ++c)
      if (!eq_step[c]) L[c * 256 + tid] = (unsigned char)tid;
  } else if (op == AUG_SOLARIZE) {
    const double thr = op_d[clip * DP];
    const unsigned char v = (double)tid < thr ? (unsigned char)tid : (unsigned char)(255 - tid);
    L[tid] = L[256 + tid] = L[512 + tid] = v;
  } else if (op == AUG_POSTERIZE) {
    const int bits = op_i[clip * IP + 1];
    const unsigned char v = (unsigned char)(tid & ~((1 << (8 - bits)) - 1));
    L[tid] = L[256 + tid] = L[512 + tid] = v;
  } else if (op == AUG_CONTRAST) {
    // ImageEnhance.Contrast: int(ImageStat.Stat(L image).mean[0] + 0.5); the sum of j * h[j] is an exact integer in double
    if (tid == 0) {
      double s = 0.0;
      for (int j = 0; j < 256; ++j) s += (double)j * (double)h[768 + j];
      const double m = s / ((double)H * (double)W);
      mean[frame] = (int)(m + 0.5);
    }
  } else if (op == AUG_AFFINE_SCALE && frame == clip * T) {
    // ImagingScaleAffine: xo = a[2] + a[0] * 0.5, then xo += a[0] per column; COORD(v) = v < 0 ? -1 : (int)v.  The
    // accumulation is sequential in the reference and is kept so (W + H additions per clip).
    const double a0 = op_d[clip * DP], a2 = op_d[clip * DP + 1], a4 = op_d[clip * DP + 2], a5 = op_d[clip * DP + 3];
    int* t = tab + (size_t)clip * (W + H);
    if (tid == 0) {
      double xo = a2 + a0 * 0.5;
      for (int x = 0; x < W; ++x) {
        const int xi = xo < 0.0 ? -1 : (int)xo;
        t[x] = (xi >= 0 && xi < W) ? xi : -1;
        xo += a0;
      }
    } else if (tid == 64) {
      double yo = a5 + a4 * 0.5;
      for (int y = 0; y < H; ++y) {
        const int yi = yo < 0.0 ? -1 : (int)yo;
        t[W + y] = (yi >= 0 && yi < H) ? yi : -1;
        yo += a4;
      }
    }
  }
}

// ---- pass 3: one output pixel per thread -----------------------------------------------------------------------------------
__device__ __forceinline__ void put_rgb(unsigned char* dst, unsigned packed) {
  dst[0] = (unsigned char)(packed >> 16);
  dst[1] = (unsigned char)(packed >> 8);
  dst[2] = (unsigned char)packed;
}

// grid (ceil(H*W / 256), B*T)
__global__ __launch_bounds__(256) void aug_apply_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                         const int* __restrict__ op_i, const double* __restrict__ op_d,
                                                         const unsigned char* __restrict__ lut, const int* __restrict__ mean,
                                                         const int* __restrict__ tab, int T, int H, int W) {
  const int frame = blockIdx.y, clip = frame / T;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int* pi = op_i + clip * IP;
  const int op = pi[0];  // uniform per block
  const unsigned char* src = in + (size_t)frame * H * W * 3;
  unsigned char* dst = out + ((size_t)frame * H * W + p) * 3;
  const int y = p / W, x = p - y * W;
  if (op == AUG_IDENTITY) {
    dst[0] = src[3 * p];
    dst[1] = src[3 * p + 1];
    dst[2] = src[3 * p + 2];
  } else if (is_lut(op)) {
    const unsigned char* L = lut + (size_t)frame * 768;
    dst[0] = L[src[3 * p]];
    dst[1] = L[256 + src[3 * p + 1]];
    dst[2] = L[512 + src[3 * p + 2]];
  } else if (op >= AUG_COLOR && op <= AUG_SHARPNESS) {
    const float alpha = (float)op_d[clip * DP];  // _imaging.c parses a double and passes (float)alpha to ImagingBlend
    const int r = src[3 * p], g = src[3 * p + 1], b = src[3 * p + 2];
    int d0, d1, d2;
    if (op == AUG_COLOR) {
      d0 = d1 = d2 = (int)gray_of(r, g, b);
    } else if (op == AUG_CONTRAST) {
      d0 = d1 = d2 = mean[frame];
    } else if (op == AUG_BRIGHTNESS) {
      d0 = d1 = d2 = 0;
    } else {
      // ImageFilter.SMOOTH through ImagingFilter3x3: border pixels are copied; inside, per channel,
      // ss = 0.5; ss += row(y+1) . k[0:3]; ss += row(y) . k[3:6]; ss += row(y-1) . k[6:9], each row product summed
      // left to right in single precision, k = (1,1,1,1,5,1,1,1,1) / 13 as floats; clip8 truncates
      if (y == 0 || y == H - 1 || x == 0 || x == W - 1) {
        d0 = r;
        d1 = g;
        d2 = b;
      } else {
        const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
        int d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float ss = 0.5f;
#pragma unroll
          for (int dy = 1; dy >= -1; --dy) {
            const unsigned char* row = src + ((size_t)(y + dy) * W + x) * 3 + c;
            const float kc = dy == 0 ? k5 : k1;
            const float t = ((float)row[-3] * k1 + (float)row[0] * kc) + (float)row[3] * k1;
            ss += t;
          }
          d[c] = ss <= 0.0f ? 0 : (ss >= 255.0f ? 255 : (int)ss);
        }
        d0 = d[0];
        d1 = d[1];
        d2 = d[2];
      }
    }
    dst[0] = blend_u8(d0, r, alpha);
    dst[1] = blend_u8(d1, g, alpha);
    dst[2] = blend_u8(d2, b, alpha);
  } else if (op == AUG_AFFINE_FIXED) {
    // affine_fixed: xx = a2 + y * a1 + x * a0 (32-bit wrap-around as in C), source pixel (xx >> 16, yy >> 16)
    const unsigned a0 = pi[1], a1 = pi[2], a2 = pi[3], a3 = pi[4], a4 = pi[5], a5 = pi[6];
    const int xin = (int)(a2 + (unsigned)y * a1 + (unsigned)x * a0) >> 16;
    const int yin = (int)(a5 + (unsigned)y * a4 + (unsigned)x * a3) >> 16;
    if (xin >= 0 && xin < W && yin >= 0 && yin < H) {
      const unsigned char* s = src + ((size_t)yin * W + xin) * 3;
      dst[0] = s[0];
      dst[1] = s[1];
      dst[2] = s[2];
    } else {
      put_rgb(dst, (unsigned)pi[7]);
    }
  } else if (op == AUG_AFFINE_SCALE) {
    const int* t = tab + (size_t)clip * (W + H);
    const int xin = t[x], yin = t[W + y];
    if (xin >= 0 && yin >= 0) {
      const unsigned char* s = src + ((size_t)yin * W + xin) * 3;
      dst[0] = s[0];
      dst[1] = s[1];
      dst[2] = s[2];
    } else {
      put_rgb(dst, (unsigned)pi[7]);
    }
  } else {  // AUG_CUTOUT
    if (x >= pi[1] && x <= pi[3] && y >= pi[2] && y <= pi[4]) {
      put_rgb(dst, (unsigned)pi[7]);
    } else {
      dst[0] = src[3 * p];
      dst[1] = src[3 * p + 1];
      dst[2] = src[3 * p + 2];
    }
  }
}

size_t aug_ws_layout(int B, int T, int H, int W, size_t* lut_off, size_t* mean_off, size_t* tab_off) {
  const size_t frames = (size_t)B * T;
  size_t off = frames * 1024 * sizeof(unsigned);  // hist
  *lut_off = off;
  off += frames * 768;
  off = (off + 15) & ~(size_t)15;
  *mean_off = off;
  off += frames * sizeof(int);
  off = (off + 15) & ~(size_t)15;
  *tab_off = off;
  off += (size_t)B * (W + H) * sizeof(int);
  return (off + 255) & ~(size_t)255;
}

}  // namespace

extern "C" size_t bdv_randaug_workspace_bytes(int B, int T, int H, int W) {
  size_t a, b, c;
  return aug_ws_layout(B, T, H, W, &a, &b, &c);
}

extern "C" int bdv_randaug_apply(const uint8_t* in, uint8_t* out, const int32_t* op_i, const double* op_d, int B, int T,
                                 int H, int W, void* ws, size_t ws_bytes, void* stream) {
  BDV_REQUIRE(in && out && op_i && op_d, "bdv_randaug_apply: null pointer");
  BDV_REQUIRE(in != out, "bdv_randaug_apply: in-place is not supported (neighbourhood and geometric operations)");
  BDV_REQUIRE(B > 0 && T > 0 && H >= 3 && W >= 3, "bdv_randaug_apply: bad shape B=%d T=%d H=%d W=%d", B, T, H, W);
  BDV_REQUIRE((long long)B * T <= 65535, "bdv_randaug_apply: B*T = %lld frames exceed one grid dimension", (long long)B * T);
  BDV_REQUIRE((long long)H * W < (1ll << 30), "bdv_randaug_apply: frame too large");
  size_t lut_off, mean_off, tab_off;
  const size_t need = aug_ws_layout(B, T, H, W, &lut_off, &mean_off, &tab_off);
  BDV_REQUIRE(ws && ws_bytes >= need, "bdv_randaug_apply: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)ws;
  unsigned* hist = (unsigned*)base;
  const int frames = B * T, HW = H * W;
  hipError_t e = hipMemsetAsync(hist, 0, (size_t)frames * 1024 * sizeof(unsigned), s);
  if (e != hipSuccess) {
    bdv_set_error("bdv_randaug_apply: memset failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  int chunks = (HW + 256 * 16 - 1) / (256 * 16);
  if (chunks > 64) chunks = 64;
  hipLaunchKernelGGL(aug_hist_kernel, dim3(chunks, frames), dim3(256), 0, s, in, op_i, hist, T, HW);
  BDV_LAUNCH_CHECK("aug_hist_kernel");
  hipLaunchKernelGGL(aug_table_kernel, dim3(frames), dim3(256), 0, s, op_i, op_d, hist, (unsigned char*)(base + lut_off),
                     (int*)(base + mean_off), (int*)(base + tab_off), T, H, W);
  BDV_LAUNCH_CHECK("aug_table_kernel");
  hipLaunchKernelGGL(aug_apply_kernel, dim3((HW + 255) / 256, frames), dim3(256), 0, s, in, out, op_i, op_d,
                     (const unsigned char*)(base + lut_off), (const int*)(base + mean_off), (const int*)(base + tab_off), T, H, W);
  BDV_LAUNCH_CHECK("aug_apply_kernel");
  return BDV_OK;
}

// ---- Resize / MultiScaleCrop + Resize of the frame pipeline: OpenCV's INTER_LINEAR on 8-bit images -------------------------------
// UPSTREAM mmaction2 Resize -> mmcv.imresize(interpolation='bilinear') -> cv2.resize(..., INTER_LINEAR) (configs/ucf101/
// bgmix_plus_randAug/...py:127, :136; MultiScaleCrop's crop is a view, the Resize after it does the resampling).  cv2 is absent
// from this image: the arithmetic below restates OpenCV's published fixed-point algorithm (imgproc/resize.cpp: coefficients in 11
// fraction bits from float weights rounded half-to-even, a horizontal pass into 32-bit rows, a vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2; the exact-2x shrink is the 2x2 box average cv::resize switches to)
// -- PARITY UNPINNED, see oracle/resize_oracle.py.  The float / double operations are the same IEEE operations without contraction
// (this file is built with -ffp-contract=off).
namespace {

struct ResizeAxis {
  int s0, s1;     // source taps (already clamped into the box)
  int a0, a1;     // 11-bit weights
};

__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

// x axis: a tap left of / beyond the last column collapses onto the edge with weight 1 (cv::resize clamps fx there)
__device__ __forceinline__ ResizeAxis resize_axis_x(int d, double scale, int ssize) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= s;
  if (s < 0) {
    f = 0.f;
    s = 0;
  }
  if (s >= ssize - 1) {
    f = 0.f;
    s = ssize - 1;
  }
  ResizeAxis r;
  r.s0 = s;
  r.s1 = s + 1 < ssize ? s + 1 : ssize - 1;
  r.a0 = sat_short(__float2int_rn((1.f - f) * 2048.f));
  r.a1 = sat_short(__float2int_rn(f * 2048.f));
  return r;
}

// y axis: the weights keep the unclamped fraction, only the row indices are clipped (resizeGeneric_Invoker)
__device__ __forceinline__ ResizeAxis resize_axis_y(int d, double scale, int ssize) {
  float f = (float)((d + 0.5) * scale - 0.5);
  const int s = (int)floorf(f);
  f -= s;
  ResizeAxis r;
  r.s0 = s < 0 ? 0 : s >= ssize ? ssize - 1 : s;
  r.s1 = s + 1 < 0 ? 0 : s + 1 >= ssize ? ssize - 1 : s + 1;
  r.a0 = sat_short(__float2int_rn((1.f - f) * 2048.f));
  r.a1 = sat_short(__float2int_rn(f * 2048.f));
  return r;
}

struct ResizeBox {
  const uint8_t* base;   // first pixel of the box
  size_t pitch;
  int bw, bh, mode;      // mode 0: resample, 1: copy (same size), 2: 2x2 box mean (exact 2x shrink)
  double scale_x, scale_y;
};

// one output pixel of a box -> packed 0x00BBGGRR
__device__ __forceinline__ unsigned resize_pixel(const ResizeBox& b, int dx, int dy) {
  unsigned out = 0;
  if (b.mode == 1) {   // same size: cv::resize copies
#pragma unroll
    for (int c = 0; c < 3; ++c) out |= (unsigned)b.base[dy * b.pitch + dx * 3 + c] << (8 * c);
    return out;
  }
  if (b.mode == 2) {   // exact 2x shrink: INTER_LINEAR is replaced by the fast INTER_AREA (2x2 mean, rounded)
    const uint8_t* p = b.base + (size_t)(2 * dy) * b.pitch + (size_t)(2 * dx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) out |= (unsigned)((p[c] + p[3 + c] + p[b.pitch + c] + p[b.pitch + 3 + c] + 2) >> 2) << (8 * c);
    return out;
  }
  const ResizeAxis ax = resize_axis_x(dx, b.scale_x, b.bw), ay = resize_axis_y(dy, b.scale_y, b.bh);
  const uint8_t* r0 = b.base + (size_t)ay.s0 * b.pitch;
  const uint8_t* r1 = b.base + (size_t)ay.s1 * b.pitch;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = r0[ax.s0 * 3 + c] * ax.a0 + r0[ax.s1 * 3 + c] * ax.a1;
    const int h1 = r1[ax.s0 * 3 + c] * ax.a0 + r1[ax.s1 * 3 + c] * ax.a1;
    const int v = (((ay.a0 * (h0 >> 4)) >> 16) + ((ay.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
    out |= (unsigned)(v & 255) << (8 * c);
  }
  return out;
}

// grid (pixel groups of one frame, frame): four consecutive output pixels of a frame per thread -- one integer division and the
// box's two double divisions per THREAD (the first form did them, and a 64-bit division, per pixel and was bound by that arithmetic);
// aligned dword stores when a frame is a whole number of dwords.  boxes: per group of T frames (x0, y0, w, h) inside the source
// frame, or NULL = the whole frame.
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ boxes,
                                                                uint8_t* __restrict__ dst, int T, int Hs, int Ws, int Hd, int Wd) {
  const unsigned npix = (unsigned)Hd * (unsigned)Wd;
  const unsigned p0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
  if (p0 >= npix) return;
  const unsigned n = blockIdx.y;
  int x0 = 0, y0 = 0;
  ResizeBox b;
  b.bw = Ws;
  b.bh = Hs;
  if (boxes != nullptr) {
    const int32_t* q = boxes + 4 * (n / (unsigned)T);
    x0 = q[0];
    y0 = q[1];
    b.bw = q[2];
    b.bh = q[3];
  }
  b.pitch = (size_t)Ws * 3;
  b.base = src + ((size_t)n * Hs + y0) * b.pitch + (size_t)x0 * 3;
  b.mode = (b.bw == Wd && b.bh == Hd) ? 1 : (b.bw == 2 * Wd && b.bh == 2 * Hd) ? 2 : 0;
  b.scale_x = 1.0 / ((double)Wd / b.bw);
  b.scale_y = 1.0 / ((double)Hd / b.bh);
  int dy = (int)(p0 / (unsigned)Wd), dx = (int)(p0 - (unsigned)dy * (unsigned)Wd);
  unsigned px[4];
  const int cnt = npix - p0 < 4u ? (int)(npix - p0) : 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    px[k] = k < cnt ? resize_pixel(b, dx, dy) : 0u;
    if (++dx == Wd) {
      dx = 0;
      ++dy;
    }
  }
  uint8_t* out = dst + ((size_t)n * npix + p0) * 3;
  if (cnt == 4 && (npix & 3u) == 0u) {
    unsigned* o = reinterpret_cast<unsigned*>(out);
    o[0] = px[0] | (px[1] << 24);
    o[1] = (px[1] >> 8) | (px[2] << 16);
    o[2] = (px[2] >> 16) | (px[3] << 8);
  } else {
    for (int k = 0; k < cnt; ++k) {
      out[3 * k] = (uint8_t)px[k];
      out[3 * k + 1] = (uint8_t)(px[k] >> 8);
      out[3 * k + 2] = (uint8_t)(px[k] >> 16);
    }
  }
}

}  // namespace

extern "C" int bdv_resize_linear_u8(const uint8_t* src, int N, int Hs, int Ws, const int32_t* boxes, int frames_per_box,
                                    const int32_t* boxes_host, uint8_t* dst, int Hd, int Wd, void* stream) {
  BDV_REQUIRE(src && dst && src != dst, "bdv_resize_linear_u8: null pointer / in-place");
  BDV_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "bdv_resize_linear_u8: bad shape N=%d %dx%d -> %dx%d", N, Hs, Ws, Hd, Wd);
  BDV_REQUIRE((boxes == nullptr) == (boxes_host == nullptr), "bdv_resize_linear_u8: the boxes are needed on the device AND on the host (they are validated here)");
  if (boxes != nullptr) {
    BDV_REQUIRE(frames_per_box > 0 && N % frames_per_box == 0, "bdv_resize_linear_u8: %d frames are not whole groups of %d", N, frames_per_box);
    for (int i = 0; i < N / frames_per_box; ++i) {   // an out-of-range box would read outside the tensor: refuse on the host
      const int32_t* b = boxes_host + 4 * i;
      BDV_REQUIRE(b[2] > 0 && b[3] > 0 && b[0] >= 0 && b[1] >= 0 && b[0] + b[2] <= Ws && b[1] + b[3] <= Hs,
                  "bdv_resize_linear_u8: box %d = (x %d, y %d, w %d, h %d) leaves the %d x %d frame", i, b[0], b[1], b[2], b[3], Ws, Hs);
    }
  }
  BDV_REQUIRE(N <= 65535 && (long long)Hd * Wd < (1ll << 31) && (long long)Hs * Ws < (1ll << 29), "bdv_resize_linear_u8: at most 65535 frames per call; frame too large");
  BDV_REQUIRE((((uintptr_t)dst) & 3) == 0, "bdv_resize_linear_u8: dst must be 4-byte aligned");
  const unsigned groups = ((unsigned)Hd * (unsigned)Wd + 3u) / 4u;   // four pixels per thread
  hipLaunchKernelGGL(resize_linear_u8_kernel, dim3((groups + 255u) / 256u, (unsigned)N), dim3(256), 0, (hipStream_t)stream, src, boxes, dst,
                     boxes ? frames_per_box : 1, Hs, Ws, Hd, Wd);
  BDV_LAUNCH_CHECK("bdv_resize_linear_u8");
  return BDV_OK;
}
