// HBM-bound helpers around the conv stack: layout change at the boundary, max/avg pooling, and
// the fused background-mix + normalize front-end (uint8 in, NHWC4 fp32 out).
#include "common.h"

namespace {

// (N,3,H,W) -> (N,H,W,4); one thread per pixel: three coalesced plane reads, one 16-byte store.
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float4* __restrict__ out, int64_t npix,
                                                              int HW) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
    const int64_t n = i / HW;
    const int64_t p = i - n * HW;
    const float* b = x + n * 3 * HW + p;
    out[i] = make_float4(b[0], b[HW], b[2 * (int64_t)HW], 0.f);
  }
}

// MaxPool2d(3, 2, 1) NHWC; thread = (output pixel, 4 channels).  First maximum in (r,s) scan order wins.
// grid.y = (frame, output row): the only division left per thread is the small 32-bit one by the channel-vector count.
template <int ESO = 4>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float4* __restrict__ x, void* __restrict__ out,
                                                           uchar4* __restrict__ idx, int N, int H, int W, int CV, int Ho, int Wo) {
  for (int row = blockIdx.y; row < N * Ho; row += gridDim.y) {
    const int n = row / Ho, ho = row - n * Ho;
    const float4* xin = x + (int64_t)n * H * W * CV;
    const int64_t orow = ((int64_t)n * Ho + ho) * Wo * CV;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Wo * CV; i += gridDim.x * blockDim.x) {
      const int wo = i / CV, c4 = i - wo * CV;
      float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      uchar4 mi = make_uchar4(255, 255, 255, 255);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int hi = 2 * ho - 1 + r;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int wi = 2 * wo - 1 + s;
          if ((unsigned)wi >= (unsigned)W) continue;
          const float4 v = xin[(hi * W + wi) * CV + c4];
          const unsigned char t = (unsigned char)(r * 3 + s);
          if (v.x > m.x || mi.x == 255) { m.x = v.x; mi.x = t; }
          if (v.y > m.y || mi.y == 255) { m.y = v.y; mi.y = t; }
          if (v.z > m.z || mi.z == 255) { m.z = v.z; mi.z = t; }
          if (v.w > m.w || mi.w == 255) { m.w = v.w; mi.w = t; }
        }
      }
      act_st4<ESO>(out, orow + i, m);
      idx[orow + i] = mi;
    }
  }
}

// Stem tail in one pass: a = relu(y * scale + shift) (train-mode BatchNorm apply), MaxPool2d(3, 2, 1) of a, and the
// 1-bit ReLU mask of a -- the activation itself is never written (it is only needed pooled, and as a sign mask by the
// BatchNorm backward).  thread = (pooled pixel, 4 channels); the thread also owns the 2x2 input pixels
// (2ho..2ho+1, 2wo..2wo+1), which lie inside its window, for the mask: 8 neighbouring lanes hold the 32 channels of one
// pixel and merge their nibbles into the mask word.
template <int ESO = 4>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const float4* __restrict__ y, const float4* __restrict__ scale,
                                                                   const float4* __restrict__ shift, void* __restrict__ out,
                                                                   uchar4* __restrict__ idx, uint32_t* __restrict__ mask, int N,
                                                                   int H, int W, int CV, int Ho, int Wo) {
  for (int row = blockIdx.y; row < N * Ho; row += gridDim.y) {
    const int n = row / Ho, ho = row - n * Ho;
    const float4* yin = y + (int64_t)n * H * W * CV;
    const int64_t orow = ((int64_t)n * Ho + ho) * Wo * CV;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Wo * CV; i += gridDim.x * blockDim.x) {
      const int wo = i / CV, c4 = i - wo * CV;
      const float4 sc = scale[c4], sh = shift[c4];
      float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      uchar4 mi = make_uchar4(255, 255, 255, 255);
      unsigned nib[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int hi = 2 * ho - 1 + r;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int wi = 2 * wo - 1 + s;
          if ((unsigned)wi >= (unsigned)W) continue;
          float4 v = yin[(hi * W + wi) * CV + c4];
          v.x = fmaxf(v.x * sc.x + sh.x, 0.f);
          v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
          v.z = fmaxf(v.z * sc.z + sh.z, 0.f);
          v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
          const unsigned char t = (unsigned char)(r * 3 + s);
          if (v.x > m.x || mi.x == 255) { m.x = v.x; mi.x = t; }
          if (v.y > m.y || mi.y == 255) { m.y = v.y; mi.y = t; }
          if (v.z > m.z || mi.z == 255) { m.z = v.z; mi.z = t; }
          if (v.w > m.w || mi.w == 255) { m.w = v.w; mi.w = t; }
          if (r >= 1 && s >= 1)
            nib[r - 1][s - 1] = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
        }
      }
      act_st4<ESO>(out, orow + i, m);
      idx[orow + i] = mi;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          unsigned w32 = nib[a][b] << (4 * (c4 & 7));
          w32 |= __shfl_xor(w32, 1, 64);
          w32 |= __shfl_xor(w32, 2, 64);
          w32 |= __shfl_xor(w32, 4, 64);
          const int hi = 2 * ho + a, wi = 2 * wo + b;
          if ((c4 & 7) == 0 && hi < H && wi < W) mask[(((int64_t)n * H + hi) * W + wi) * (CV / 8) + (c4 >> 3)] = w32;
        }
    }
  }
}

// Backward as a gather without divergence (pool_bwd_gather2x2 in common.h): a thread owns the 2x2 input block
// (2i..2i+1, 2j..2j+1) x 4 channels and loads each of the four windows that cover it once.  grid.y = (frame, block row i).
template <int ESD = 4>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const void* __restrict__ dout, const uchar4* __restrict__ idx,
                                                           float4* __restrict__ dx, int N, int H, int W, int CV, int Ho, int Wo) {
  for (int row = blockIdx.y; row < N * Ho; row += gridDim.y) {
    const int n = row / Ho, i = row - n * Ho;
    const int64_t obase = (int64_t)n * Ho * Wo * CV;
    const int64_t ibase = (int64_t)n * H * W * CV;
    const bool down = i + 1 < Ho, h1ok = 2 * i + 1 < H;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Wo * CV; q += gridDim.x * blockDim.x) {
      const int j = q / CV, c4 = q - j * CV;
      const bool right = j + 1 < Wo, w1ok = 2 * j + 1 < W;
      float4 g[4];
      pool_bwd_gather2x2<ESD>(dout, idx, obase + (i * Wo + j) * CV + c4, CV, Wo, right, down, g);
      const int64_t p00 = ibase + ((2 * i) * W + 2 * j) * CV + c4;
      dx[p00] = g[0];
      if (w1ok) dx[p00 + CV] = g[1];
      if (h1ok) {
        dx[p00 + W * CV] = g[2];
        if (w1ok) dx[p00 + (W + 1) * CV] = g[3];
      }
    }
  }
}

// [N][HW][C] -> [N][C]
template <int ES = 4>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const void* __restrict__ x, float4* __restrict__ out, int N, int HW,
                                                           int CV) {
  const int64_t total = (int64_t)N * CV;
  const float inv = 1.f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % CV);
    const int64_t n = i / CV;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < HW; ++p) {
      const float4 v = act_ld4<ES>(x, (n * HW + p) * CV + c4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[i] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}

template <int ES = 4>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float4* __restrict__ dout, void* __restrict__ dx, int N, int HW,
                                                           int CV) {
  const int64_t total = (int64_t)N * HW * CV;
  const float inv = 1.f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % CV);
    const int64_t n = i / ((int64_t)HW * CV);
    const float4 d = dout[n * CV + c4];
    act_st4<ES>(dx, i, make_float4(d.x * inv, d.y * inv, d.z * inv, d.w * inv));
  }
}

struct NormParams {
  float mean[3], std[3], inv_std[3];
  float alpha;
};

// Fused front-end.  One thread = one pixel of one clip position (b, h, w): the background pixel is
// loaded and normalised once and reused for the T frames of the clip.
// BGF: the background arrives as fp32 pixel values in [0, 255] (the output of bg_resize_crop_kernel: torchvision's Resize on a
// float image is not rounded back to uint8) instead of uint8.
template <bool BGF>
__global__ __launch_bounds__(256) void bgmix_normalize_kernel(const uint8_t* __restrict__ frames, const void* __restrict__ bg,
                                                               const uint8_t* __restrict__ mix, NormParams np,
                                                               float4* __restrict__ out_nhwc4, float* __restrict__ out_nchw,
                                                               int B, int T, int HW) {
  // the reference blends with separate torch multiplies and adds: no fused multiply-add here (HIP's __fmul_rn / __fadd_rn
  // are header inlines the compiler is free to contract, so plain operators under this pragma)
#pragma clang fp contract(off)
  const int64_t total = (int64_t)B * HW;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int b = (int)(i / HW);
    const int p = (int)(i - (int64_t)b * HW);
    const bool do_mix = mix != nullptr && mix[b] != 0;
    float bgn[3] = {0.f, 0.f, 0.f};
    if (do_mix) {
      const int64_t o = ((int64_t)b * HW + p) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = BGF ? reinterpret_cast<const float*>(bg)[o + c] : (float)reinterpret_cast<const uint8_t*>(bg)[o + c];
        bgn[c] = (v - np.mean[c]) / np.std[c];
      }
    }
    const float one_m_alpha = 1.f - np.alpha;
    for (int t = 0; t < T; ++t) {
      const int64_t f = (int64_t)b * T + t;
      const uint8_t* q = frames + (f * HW + p) * 3;
      float v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float xn = ((float)q[c] - np.mean[c]) * np.inv_std[c];
        if (do_mix) xn = xn * one_m_alpha + bgn[c] * np.alpha;
        v[c] = xn;
      }
      if (out_nhwc4 != nullptr) out_nhwc4[f * HW + p] = make_float4(v[0], v[1], v[2], 0.f);
      if (out_nchw != nullptr) {
        float* o = out_nchw + f * 3 * HW + p;
        o[0] = v[0];
        o[HW] = v[1];
        o[2 * (int64_t)HW] = v[2];
      }
    }
  }
}

// Test-time front-end: fixed crops (+ horizontal flips) of every frame, normalised, in the order the crop transforms
// emit them (crop-major, then the T frames).  One thread = one output pixel of one (b, crop), looping over T.
struct CropTable {
  int n;
  int x[BDV_MAX_CROPS], y[BDV_MAX_CROPS], flip[BDV_MAX_CROPS];
};

__global__ __launch_bounds__(256) void crop_normalize_kernel(const uint8_t* __restrict__ frames, CropTable ct, NormParams np,
                                                              float4* __restrict__ out_nhwc4, float* __restrict__ out_nchw,
                                                              int B, int T, int H, int W, int ch, int cw) {
#pragma clang fp contract(off)
  const int chw = ch * cw;
  const int64_t total = (int64_t)B * ct.n * chw;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int p = (int)(i % chw);
    const int bc = (int)(i / chw);
    const int b = bc / ct.n, k = bc - b * ct.n;
    const int oy = p / cw, ox = p - oy * cw;
    const int sy = ct.y[k] + oy, sx = ct.x[k] + (ct.flip[k] ? cw - 1 - ox : ox);
    for (int t = 0; t < T; ++t) {
      const uint8_t* q = frames + ((((int64_t)b * T + t) * H + sy) * W + sx) * 3;
      float v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = ((float)q[c] - np.mean[c]) * np.inv_std[c];
      const int64_t f = ((int64_t)b * ct.n + k) * T + t;
      if (out_nhwc4 != nullptr) out_nhwc4[f * chw + p] = make_float4(v[0], v[1], v[2], 0.f);
      if (out_nchw != nullptr) {
        float* o = out_nchw + f * 3 * chw + p;
        o[0] = v[0];
        o[chw] = v[1];
        o[2 * (int64_t)chw] = v[2];
      }
    }
  }
}

int ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > (1 << 20)) b = 1 << 20;   // (one unit per thread: see ew_grid in bn.hip)
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

namespace {
// MaxPool3d((2,1,1), stride (2,1,1)) of UPSTREAM mmaction ResNet3d (pool2, between layer1 and layer2 of I3D): the larger of
// frames 2t and 2t + 1 of a clip, element by element.  sel: 1 bit per output element, set when the second frame won (ties
// go to the first frame, as torch's max-pool does); the backward routes the gradient by it and writes zeros elsewhere.
__global__ __launch_bounds__(256) void maxpool_t2_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ out,
                                                             uint32_t* __restrict__ sel, int64_t frame4, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {   // n4, stride: multiples of 8
    const int64_t f = i / frame4, r = i - f * frame4;
    const float4 a = x[(2 * f) * frame4 + r], b = x[(2 * f + 1) * frame4 + r];
    float4 o;
    unsigned m = 0;
    o.x = b.x > a.x ? (m |= 1u, b.x) : a.x;
    o.y = b.y > a.y ? (m |= 2u, b.y) : a.y;
    o.z = b.z > a.z ? (m |= 4u, b.z) : a.z;
    o.w = b.w > a.w ? (m |= 8u, b.w) : a.w;
    out[i] = o;
    m <<= 4 * (threadIdx.x & 7);
    m |= __shfl_xor(m, 1, 64);
    m |= __shfl_xor(m, 2, 64);
    m |= __shfl_xor(m, 4, 64);
    if ((threadIdx.x & 7) == 0) sel[i >> 3] = m;
  }
}

__global__ __launch_bounds__(256) void maxpool_t2_bwd_kernel(const float4* __restrict__ dout, const uint32_t* __restrict__ sel,
                                                             float4* __restrict__ dx, int64_t frame4, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t f = i / frame4, r = i - f * frame4;
    const unsigned nib = (sel[i >> 3] >> (4 * (int)(i & 7))) & 0xFu;
    const float4 g = dout[i];
    float4 a, b;
    a.x = (nib & 1u) ? 0.f : g.x; b.x = (nib & 1u) ? g.x : 0.f;
    a.y = (nib & 2u) ? 0.f : g.y; b.y = (nib & 2u) ? g.y : 0.f;
    a.z = (nib & 4u) ? 0.f : g.z; b.z = (nib & 4u) ? g.z : 0.f;
    a.w = (nib & 8u) ? 0.f : g.w; b.w = (nib & 8u) ? g.w : 0.f;
    dx[(2 * f) * frame4 + r] = a;
    dx[(2 * f + 1) * frame4 + r] = b;
  }
}
}  // namespace

extern "C" int bdv_maxpool_t2_fwd(const float* x, float* out, uint32_t* sel, int64_t frames_out, int64_t frame_elems, void* stream) {
  BDV_REQUIRE(x && out && sel && frames_out > 0 && frame_elems > 0 && frame_elems % 32 == 0, "bdv_maxpool_t2_fwd: bad argument");
  BDV_REQUIRE(bdv_aligned16(x) && bdv_aligned16(out), "bdv_maxpool_t2_fwd: alignment");
  const int64_t n4 = frames_out * frame_elems / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  hipLaunchKernelGGL(maxpool_t2_fwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)out, sel,
                     frame_elems / 4, n4);
  BDV_LAUNCH_CHECK("bdv_maxpool_t2_fwd");
  return BDV_OK;
}

extern "C" int bdv_maxpool_t2_bwd(const float* dout, const uint32_t* sel, float* dx, int64_t frames_out, int64_t frame_elems,
                                  void* stream) {
  BDV_REQUIRE(dout && dx && sel && frames_out > 0 && frame_elems > 0 && frame_elems % 32 == 0, "bdv_maxpool_t2_bwd: bad argument");
  BDV_REQUIRE(bdv_aligned16(dout) && bdv_aligned16(dx), "bdv_maxpool_t2_bwd: alignment");
  const int64_t n4 = frames_out * frame_elems / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  hipLaunchKernelGGL(maxpool_t2_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)dout, sel, (float4*)dx,
                     frame_elems / 4, n4);
  BDV_LAUNCH_CHECK("bdv_maxpool_t2_bwd");
  return BDV_OK;
}


namespace {
// Background pipeline of BackgroundMixDataset before Normalize (libs/loader/comix_loader.py:72-73): torchvision Resize(size) of
// the FLOAT image read by read_image(...).float() -- bilinear, align_corners=False, no antialias filter (identical to the
// antialiased form whenever the image is enlarged, which is the case for every dataset of the configs: 240- and 256-pixel frames
// to 256) -- followed by RandomCrop at (top[b], left[b]).  One thread = one output pixel; the arithmetic follows ATen's
// upsample_bilinear2d (source index scale * (dst + 0.5) - 0.5 clamped at 0, lambda in fp32, rows combined after columns).
__global__ __launch_bounds__(256) void bg_resize_crop_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ top,
                                                              const int32_t* __restrict__ left, float* __restrict__ out, int B, int Hs,
                                                              int Ws, int Hr, int Wr, int ch, int cw) {
#pragma clang fp contract(off)
  const int64_t total = (int64_t)B * ch * cw;
  const float sh = (float)Hs / (float)Hr, sw = (float)Ws / (float)Wr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((int64_t)ch * cw));
    const int r = (int)(i - (int64_t)b * ch * cw);
    const int y = r / cw + top[b], x = r % cw + left[b];
    float fy = sh * ((float)y + 0.5f) - 0.5f, fx = sw * ((float)x + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    float ly = fy - (float)y0, lx = fx - (float)x0;
    ly = fminf(fmaxf(ly, 0.f), 1.f);
    lx = fminf(fmaxf(lx, 0.f), 1.f);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const uint8_t* img = src + (int64_t)b * Hs * Ws * 3;
    const uint8_t *p00 = img + ((int64_t)y0 * Ws + x0) * 3, *p01 = img + ((int64_t)y0 * Ws + x1) * 3;
    const uint8_t *p10 = img + ((int64_t)y1 * Ws + x0) * 3, *p11 = img + ((int64_t)y1 * Ws + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t0 = (float)p00[c] * hx + (float)p01[c] * lx;
      const float t1 = (float)p10[c] * hx + (float)p11[c] * lx;
      out[i * 3 + c] = t0 * hy + t1 * ly;
    }
  }
}
}  // namespace

extern "C" int bdv_bg_resize_crop_u8(const uint8_t* src, int B, int Hs, int Ws, int Hr, int Wr, const int32_t* top,
                                     const int32_t* left, int crop_h, int crop_w, float* out, void* stream) {
  BDV_REQUIRE(src && top && left && out, "bdv_bg_resize_crop_u8: null pointer");
  BDV_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Hr > 0 && Wr > 0 && crop_h > 0 && crop_w > 0 && crop_h <= Hr && crop_w <= Wr,
              "bdv_bg_resize_crop_u8: bad shape (%dx%d -> %dx%d, crop %dx%d)", Hs, Ws, Hr, Wr, crop_h, crop_w);
  const int64_t total = (int64_t)B * crop_h * crop_w;
  hipLaunchKernelGGL(bg_resize_crop_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, top, left, out, B, Hs, Ws,
                     Hr, Wr, crop_h, crop_w);
  BDV_LAUNCH_CHECK("bdv_bg_resize_crop_u8");
  return BDV_OK;
}

extern "C" int bdv_nchw3_to_nhwc4(const float* x, float* out, int N, int H, int W, void* stream) {
  BDV_REQUIRE(x && out && N > 0 && H > 0 && W > 0, "bdv_nchw3_to_nhwc4: bad argument");
  BDV_REQUIRE(bdv_aligned16(out), "bdv_nchw3_to_nhwc4: alignment");
  const int64_t npix = (int64_t)N * H * W;
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(ew_grid(npix)), dim3(256), 0, (hipStream_t)stream, x, (float4*)out, npix, H * W);
  BDV_LAUNCH_CHECK("bdv_nchw3_to_nhwc4");
  return BDV_OK;
}

extern "C" int bdv_maxpool_fwd(const float* x, void* out, uint8_t* idx, int N, int H, int W, int C, int out_dtype, void* stream) {
  BDV_REQUIRE_ACT(out_dtype, "bdv_maxpool_fwd");
  BDV_REQUIRE(x && out && idx && N > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0, "bdv_maxpool_fwd: bad argument");
  BDV_REQUIRE(bdv_aligned16(x) && bdv_aligned16(out) && (((uintptr_t)idx) & 3) == 0, "bdv_maxpool_fwd: alignment");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  BDV_REQUIRE((int64_t)H * W * (C / 4) < (1ll << 31) && (int64_t)N * Ho < (1ll << 31), "bdv_maxpool_fwd: frame or batch too large");
  const int rowv = Wo * (C / 4);
  BDV_ACT_SWITCH(out_dtype, ES, hipLaunchKernelGGL((maxpool_fwd_kernel<ES>), dim3((rowv + 255) / 256, N * Ho < 65535 ? N * Ho : 65535), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                     out, (uchar4*)idx, N, H, W, C / 4, Ho, Wo));
  BDV_LAUNCH_CHECK("bdv_maxpool_fwd");
  return BDV_OK;
}

extern "C" int bdv_bn_relu_maxpool_fwd(const float* y, const float* scale, const float* shift, void* out, uint8_t* idx,
                                       uint32_t* relu_mask, int N, int H, int W, int C, int out_dtype, void* stream) {
  BDV_REQUIRE_ACT(out_dtype, "bdv_bn_relu_maxpool_fwd");
  BDV_REQUIRE(y && scale && shift && out && idx && relu_mask && N > 0 && H > 1 && W > 1 && C > 0 && C % 32 == 0,
              "bdv_bn_relu_maxpool_fwd: bad argument (C must be a multiple of 32)");
  BDV_REQUIRE(bdv_aligned16(y) && bdv_aligned16(scale) && bdv_aligned16(shift) && bdv_aligned16(out) &&
                  (((uintptr_t)idx) & 3) == 0 && (((uintptr_t)relu_mask) & 3) == 0, "bdv_bn_relu_maxpool_fwd: alignment");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  BDV_REQUIRE((int64_t)H * W * (C / 4) < (1ll << 31) && (int64_t)N * Ho < (1ll << 31), "bdv_bn_relu_maxpool_fwd: too large");
  const int rowv = Wo * (C / 4);
  BDV_ACT_SWITCH(out_dtype, ES, hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<ES>), dim3((rowv + 255) / 256, N * Ho < 65535 ? N * Ho : 65535), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)y, (const float4*)scale, (const float4*)shift, out, (uchar4*)idx,
                     relu_mask, N, H, W, C / 4, Ho, Wo));
  BDV_LAUNCH_CHECK("bdv_bn_relu_maxpool_fwd");
  return BDV_OK;
}

extern "C" int bdv_maxpool_bwd(const void* dout, const uint8_t* idx, float* dx, int N, int H, int W, int C, int dout_dtype, void* stream) {
  BDV_REQUIRE_ACT(dout_dtype, "bdv_maxpool_bwd");
  BDV_REQUIRE(dout && dx && idx && N > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0, "bdv_maxpool_bwd: bad argument");
  BDV_REQUIRE(bdv_aligned16(dout) && bdv_aligned16(dx) && (((uintptr_t)idx) & 3) == 0, "bdv_maxpool_bwd: alignment");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  BDV_REQUIRE((int64_t)Ho * Wo * (C / 4) < (1ll << 31) && (int64_t)N * H < (1ll << 31), "bdv_maxpool_bwd: frame or batch too large");
  const int rowv = Wo * (C / 4);
  BDV_ACT_SWITCH(dout_dtype, ES, hipLaunchKernelGGL((maxpool_bwd_kernel<ES>), dim3((rowv + 255) / 256, N * Ho < 65535 ? N * Ho : 65535), dim3(256), 0, (hipStream_t)stream, dout,
                     (const uchar4*)idx, (float4*)dx, N, H, W, C / 4, Ho, Wo));
  BDV_LAUNCH_CHECK("bdv_maxpool_bwd");
  return BDV_OK;
}

extern "C" int bdv_avgpool_fwd(const void* x, float* out, int N, int HW, int C, int act_dtype, void* stream) {
  BDV_REQUIRE_ACT(act_dtype, "bdv_avgpool_fwd");
  BDV_REQUIRE(x && out && N > 0 && HW > 0 && C > 0 && C % 4 == 0, "bdv_avgpool_fwd: bad argument");
  BDV_REQUIRE(bdv_aligned16(x) && bdv_aligned16(out), "bdv_avgpool_fwd: alignment");
  const int64_t total = (int64_t)N * (C / 4);
  BDV_ACT_SWITCH(act_dtype, ES, hipLaunchKernelGGL((avgpool_fwd_kernel<ES>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, (float4*)out, N,
                     HW, C / 4));
  BDV_LAUNCH_CHECK("bdv_avgpool_fwd");
  return BDV_OK;
}

extern "C" int bdv_avgpool_bwd(const float* dout, void* dx, int N, int HW, int C, int act_dtype, void* stream) {
  BDV_REQUIRE_ACT(act_dtype, "bdv_avgpool_bwd");
  BDV_REQUIRE(dout && dx && N > 0 && HW > 0 && C > 0 && C % 4 == 0, "bdv_avgpool_bwd: bad argument");
  BDV_REQUIRE(bdv_aligned16(dout) && bdv_aligned16(dx), "bdv_avgpool_bwd: alignment");
  const int64_t total = (int64_t)N * HW * (C / 4);
  BDV_ACT_SWITCH(act_dtype, ES, hipLaunchKernelGGL((avgpool_bwd_kernel<ES>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)dout, dx,
                     N, HW, C / 4));
  BDV_LAUNCH_CHECK("bdv_avgpool_bwd");
  return BDV_OK;
}

extern "C" int bdv_crop_normalize_u8(const uint8_t* frames, const int32_t* crops, int ncrops, int crop_h, int crop_w,
                                     const float mean[3], const float inv_std[3], float* out_nhwc4, float* out_nchw, int B,
                                     int T, int H, int W, void* stream) {
  BDV_REQUIRE(frames && crops && mean && inv_std, "bdv_crop_normalize_u8: null pointer");
  BDV_REQUIRE(out_nhwc4 || out_nchw, "bdv_crop_normalize_u8: no output requested");
  BDV_REQUIRE(ncrops > 0 && ncrops <= BDV_MAX_CROPS, "bdv_crop_normalize_u8: %d crops (1..%d supported)", ncrops, BDV_MAX_CROPS);
  BDV_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && crop_h > 0 && crop_w > 0 && crop_h <= H && crop_w <= W,
              "bdv_crop_normalize_u8: bad shape %dx%d crop of %dx%d", crop_h, crop_w, H, W);
  BDV_REQUIRE(out_nhwc4 == nullptr || bdv_aligned16(out_nhwc4), "bdv_crop_normalize_u8: alignment");
  CropTable ct;
  ct.n = ncrops;
  for (int k = 0; k < ncrops; ++k) {
    ct.x[k] = crops[3 * k];
    ct.y[k] = crops[3 * k + 1];
    ct.flip[k] = crops[3 * k + 2] != 0;
    BDV_REQUIRE(ct.x[k] >= 0 && ct.y[k] >= 0 && ct.x[k] + crop_w <= W && ct.y[k] + crop_h <= H,
                "bdv_crop_normalize_u8: crop %d at (%d, %d) leaves the %dx%d frame", k, ct.x[k], ct.y[k], H, W);
  }
  NormParams np;
  for (int c = 0; c < 3; ++c) {
    np.mean[c] = mean[c];
    np.std[c] = 0.f;
    np.inv_std[c] = inv_std[c];
  }
  np.alpha = 0.f;
  const int64_t total = (int64_t)B * ncrops * crop_h * crop_w;
  hipLaunchKernelGGL(crop_normalize_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, frames, ct, np,
                     (float4*)out_nhwc4, out_nchw, B, T, H, W, crop_h, crop_w);
  BDV_LAUNCH_CHECK("bdv_crop_normalize_u8");
  return BDV_OK;
}

extern "C" int bdv_bgmix_normalize_u8(const uint8_t* frames, const void* bg, int bg_f32, const uint8_t* mix, float alpha,
                                      const float mean[3], const float std[3], const float inv_std[3], float* out_nhwc4,
                                      float* out_nchw, int B, int T, int H, int W, void* stream) {
  BDV_REQUIRE(frames && mean && std && inv_std, "bdv_bgmix_normalize_u8: null pointer");
  BDV_REQUIRE((bg == nullptr) == (mix == nullptr), "bdv_bgmix_normalize_u8: bg and mix come together");
  BDV_REQUIRE(out_nhwc4 || out_nchw, "bdv_bgmix_normalize_u8: no output requested");
  BDV_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0, "bdv_bgmix_normalize_u8: bad shape");
  BDV_REQUIRE(out_nhwc4 == nullptr || bdv_aligned16(out_nhwc4), "bdv_bgmix_normalize_u8: alignment");
  NormParams np;
  for (int c = 0; c < 3; ++c) {
    np.mean[c] = mean[c];
    np.std[c] = std[c];
    np.inv_std[c] = inv_std[c];
  }
  np.alpha = alpha;
  const int64_t total = (int64_t)B * H * W;
  if (bg_f32)
    hipLaunchKernelGGL((bgmix_normalize_kernel<true>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, frames, bg, mix, np,
                       (float4*)out_nhwc4, out_nchw, B, T, H * W);
  else
    hipLaunchKernelGGL((bgmix_normalize_kernel<false>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, frames, bg, mix, np,
                       (float4*)out_nhwc4, out_nchw, B, T, H * W);
  BDV_LAUNCH_CHECK("bdv_bgmix_normalize_u8");
  return BDV_OK;
}
