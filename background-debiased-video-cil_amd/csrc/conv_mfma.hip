// Implicit-GEMM convolution (fprop / dgrad / wgrad) on the fp32-input MFMA
// v_mfma_f32_32x32x2_f32 for gfx950.  NHWC activations, [Cout][R][S][Cin] weights.
//
// Replaces: F.conv2d + autograd inside UPSTREAM mmaction ConvModule, and UPSTREAM
// TemporalShift.shift (fused into the activation-tile gather).
//
// Structure of every kernel: 256 threads = 4 waves; block tile BM x BN, K-step 16;
// LDS image of both operands is k-major ([k][m], row stride BM+4 floats) so one MFMA operand
// is one conflict-free ds_read_b32 (lane l reads [k = 2s + l/32][m = base + l%32]);
// register-staged global->LDS double buffering with one barrier per K-step.
#include "common.h"

namespace {

constexpr int BK = 16;

struct Geom {
  int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, T, fold;
  int M;     // GEMM rows: N*Ho*Wo (fprop / wgrad reduction length), N*H*W (dgrad)
  int Ktot;  // fprop: R*S*Cin ; dgrad: R*S*Cout ; wgrad: row length of dw = R*S*Cin
};

template <int LDA, int LDB, int TM, int TN>
__device__ __forceinline__ void mma_stage(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[TM][TN], int wm0, int wn0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* ap = As + h * LDA + wm0 + r;
  const float* bp = Bs + h * LDB + wn0 + r;
#pragma unroll
  for (int s = 0; s < BK / 2; ++s) {
    float a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = ap[2 * s * LDA + 32 * i];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = bp[2 * s * LDB + 32 * j];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// temporal-shift class of input channel c: +1 -> read frame t+1, -1 -> read frame t-1, 0 -> copy
__device__ __forceinline__ int shift_class(int c, int fold) {
  return (fold > 0) ? (c < fold ? 1 : (c < 2 * fold ? -1 : 0)) : 0;
}

// K-contiguous source tile (rows x 16 k) -> transposing store into the k-major LDS image.
template <int LD, int PASSES>
__device__ __forceinline__ void store_transposed(float* __restrict__ dst, const float4 (&v)[PASSES], int tid) {
  const int row = tid >> 2, kg = tid & 3;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    float* d = dst + (4 * kg) * LD + row + 64 * p;
    d[0] = v[p].x;
    d[LD] = v[p].y;
    d[2 * LD] = v[p].z;
    d[3 * LD] = v[p].w;
  }
}

// M-contiguous source tile (16 k rows x COLS) -> direct 16-byte stores.
template <int LD, int COLS, int PASSES>
__device__ __forceinline__ void store_direct(float* __restrict__ dst, const float4 (&v)[PASSES], int tid) {
  constexpr int V = COLS / 4;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int idx = tid + 256 * p;
    const int krow = idx / V, c4 = idx - krow * V;
    *reinterpret_cast<float4*>(dst + krow * LD + 4 * c4) = v[p];
  }
}

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// =========================================================================================
// fprop: y[m, co] = sum_{tap, ci} x_shift[n, ho*st + r - p, wo*st + s - p, ci] * w[co, tap, ci]
// =========================================================================================
template <int BM, int BN, int WM, int WN, bool VEC_TAP>
__global__ __launch_bounds__(256) void conv_fprop_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, Geom g, int MT, int NT) {
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 64, BP = BN / 64;
  constexpr int STAGE = BK * (LDA + LDB);
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); keep all the
  // Cout-tiles of one activation row-tile on one XCD so the A tile is fetched into one L2 only.
  const int id = blockIdx.x;
  const int xcd = id & 7, jj = id >> 3;
  const int mt = (jj / NT) * 8 + xcd, nt = jj % NT;
  if (mt >= MT) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int arow = tid >> 2, kg = tid & 3;
  const int HoWo = g.Ho * g.Wo;

  int a_n[AP], a_t[AP], a_hi0[AP], a_wi0[AP];
  bool a_ok[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + 64 * p;
    a_ok[p] = m < g.M;
    const int mm = a_ok[p] ? m : 0;
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    a_n[p] = n;
    a_t[p] = n % g.T;
    a_hi0[p] = ho * g.stride - g.pad;
    a_wi0[p] = wo * g.stride - g.pad;
  }
  const int RS = g.R * g.S;

  float4 ra[AP], rb[BP];
  auto load = [&](int kt) {
    int tap, c;
    if (VEC_TAP) {
      const int k = kt * BK + 4 * kg;
      tap = k / g.Cin;
      c = k - tap * g.Cin;
    } else {
      const int k0 = kt * BK;
      tap = k0 / g.Cin;
      c = k0 - tap * g.Cin + 4 * kg;
    }
    const int r = tap / g.S, s = tap - r * g.S;
    const bool kvalid = tap < RS;
    const int cls = shift_class(c, g.fold);
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int hi = a_hi0[p] + r, wi = a_wi0[p] + s;
      const bool v = a_ok[p] && kvalid && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W &&
                     (unsigned)(a_t[p] + cls) < (unsigned)g.T;
      const size_t off = ((size_t)((a_n[p] + cls) * g.H + hi) * g.W + wi) * g.Cin + c;
      ra[p] = v ? *reinterpret_cast<const float4*>(x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int kb = kt * BK + 4 * kg;
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int co = nt * BN + arow + 64 * p;
      rb[p] = (kb < g.Ktot) ? *reinterpret_cast<const float4*>(w + (size_t)co * g.Ktot + kb)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * LDA;
    store_transposed<LDA, AP>(As, ra, tid);
    store_transposed<LDB, BP>(Bs, rb, tid);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (g.Ktot + BK - 1) / BK;
  load(0);
  store(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load(kt + 1);
    const float* As = smem + cur * STAGE;
    mma_stage<LDA, LDB, TM, TN>(As, As + BK * LDA, acc, wm0, wn0, lane);
    if (kt + 1 < nk) store(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = nt * BN + wn0 + 32 * j + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mt * BM + wm0 + 32 * i + acc_row(e, lane);
        if (row < g.M) y[(size_t)row * g.Cout + col] = acc[i][j][e];
      }
    }
}

// =========================================================================================
// dgrad: dxs[m=(n,h,w), ci] = sum_{tap, co} dy[n, (h+p-r)/st, (w+p-s)/st, co] * w[co, tap, ci]
// Stride 2 is decomposed into the 4 input-pixel parity classes (blockIdx.y): a class only visits the
// taps whose (h+p-r) is even, so no MFMA work is spent on structural zeros.
// epilogue: temporal un-shift (scatter to frame t+/-1) + optional masked residual-gradient add.
// =========================================================================================
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          float* __restrict__ dx, const float* __restrict__ add_src,
                                                          const float* __restrict__ add_mask, Geom g, int NT) {
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 64, BP = BN / 64;
  constexpr int BV = BN / 4;
  constexpr int STAGE = BK * (LDA + LDB);
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  // parity class of the input pixel (stride 1: a single class)
  const int st = g.stride;
  const int ph = blockIdx.y / st, pw = blockIdx.y - ph * st;
  const int Hc = (g.H - ph + st - 1) / st, Wc = (g.W - pw + st - 1) / st;
  const int Mc = g.N * Hc * Wc;
  const int MT = (Mc + BM - 1) / BM;
  const int r0 = (ph + g.pad) % st, s0 = (pw + g.pad) % st;
  const int nr = r0 < g.R ? (g.R - r0 + st - 1) / st : 0;
  const int ns = s0 < g.S ? (g.S - s0 + st - 1) / st : 0;
  const int bh = (ph + g.pad - r0) / st, bw = (pw + g.pad - s0) / st;

  const int id = blockIdx.x;
  const int xcd = id & 7, jj = id >> 3;
  const int mt = (jj / NT) * 8 + xcd, nt = jj % NT;
  if (mt >= MT) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int arow = tid >> 2, kg = tid & 3;
  const int HcWc = Hc * Wc;
  const int HW = g.H * g.W;
  const int RS = g.R * g.S;

  int a_n[AP], a_h[AP], a_w[AP];
  bool a_ok[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + 64 * p;
    a_ok[p] = m < Mc;
    const int mm = a_ok[p] ? m : 0;
    const int n = mm / HcWc;
    const int rem = mm - n * HcWc;
    const int hc = rem / Wc;
    a_n[p] = n;
    a_h[p] = hc + bh;
    a_w[p] = rem - hc * Wc + bw;
  }

  float4 ra[AP], rb[BP];
  auto load = [&](int kt) {
    const int k0 = kt * BK;
    const int ct = k0 / g.Cout;             // tap index inside this class
    const int co0 = k0 - ct * g.Cout;
    const int ir = ct / ns, is = ct - ir * ns;
    const int tap = (r0 + ir * st) * g.S + (s0 + is * st);
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int ho = a_h[p] - ir, wo = a_w[p] - is;
      const bool v = a_ok[p] && (unsigned)ho < (unsigned)g.Ho && (unsigned)wo < (unsigned)g.Wo;
      const size_t off = ((size_t)(a_n[p] * g.Ho + ho) * g.Wo + wo) * g.Cout + co0 + 4 * kg;
      ra[p] = v ? *reinterpret_cast<const float4*>(dy + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int idx = tid + 256 * p;
      const int krow = idx / BV, c4 = idx - krow * BV;
      const size_t off = ((size_t)(co0 + krow) * RS + tap) * g.Cin + nt * BN + 4 * c4;
      rb[p] = *reinterpret_cast<const float4*>(w + off);
    }
  };
  auto store = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * LDA;
    store_transposed<LDA, AP>(As, ra, tid);
    store_direct<LDB, BN, BP>(Bs, rb, tid);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = nr * ns * g.Cout / BK;     // 0 for a class no filter tap reaches (e.g. 1x1 stride 2, odd pixels)
  if (nk > 0) {
    load(0);
    store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) load(kt + 1);
      const float* As = smem + cur * STAGE;
      mma_stage<LDA, LDB, TM, TN>(As, As + BK * LDA, acc, wm0, wn0, lane);
      if (kt + 1 < nk) store(cur ^ 1);
      __syncthreads();
    }
  }

  // Epilogue.  Forward read xs[frame n] = x[frame n + cls]; so the gradient of row m goes to
  // frame n + cls when that frame is inside the clip.  Rows whose target falls outside the clip
  // ("orphans") instead write the zero that the unreachable frame at the other clip end needs,
  // which makes the scatter a bijection over dx.
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int mrow = mt * BM + wm0 + 32 * i + acc_row(e, lane);
      if (mrow >= Mc) continue;
      int row, n;
      if (st == 1) {
        row = mrow;
        n = 0;
      } else {
        n = mrow / HcWc;
        const int rem = mrow - n * HcWc;
        const int hc = rem / Wc;
        row = (n * g.H + hc * st + ph) * g.W + (rem - hc * Wc) * st + pw;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = nt * BN + wn0 + 32 * j + (lane & 31);
        const int cls = shift_class(col, g.fold);
        float v = acc[i][j][e];
        int drow = row;
        if (cls != 0) {
          if (st == 1) n = row / HW;
          const int t = n % g.T;
          if ((unsigned)(t + cls) < (unsigned)g.T) {
            drow = row + cls * HW;
          } else {
            drow = row - cls * (g.T - 1) * HW;
            v = 0.f;
          }
        }
        const size_t o = (size_t)drow * g.Cin + col;
        if (add_src != nullptr) {
          float a = add_src[o];
          if (add_mask != nullptr && !(add_mask[o] > 0.f)) a = 0.f;
          v += a;
        }
        dx[o] = v;
      }
    }
}

// =========================================================================================
// wgrad: slab[split][co][tap*Cin + ci] = sum_{m in split} dy[m, co] * x_shift[pix(m, tap), ci]
// =========================================================================================
template <int BM, int BN, int WM, int WN, bool VEC_TAP>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ slab, Geom g, int MTw, int NTw,
                                                          int kt_per_split) {
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 64, BP = BN / 64;
  constexpr int AV = BM / 4, BV = BN / 4;
  constexpr int STAGE = BK * (LDA + LDB);
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int mt = blockIdx.x % MTw, nt = blockIdx.x / MTw;
  const int split = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int HoWo = g.Ho * g.Wo;
  const int RS = g.R * g.S;

  // column -> (tap, ci) for this thread's B loads (fixed over the K loop)
  int b_tap[BP], b_ci[BP], b_krow[BP];
  bool b_cok[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int idx = tid + 256 * p;
    const int krow = idx / BV, c4 = idx - krow * BV;
    b_krow[p] = krow;
    if (VEC_TAP) {
      const int ncol = nt * BN + 4 * c4;
      b_tap[p] = ncol / g.Cin;
      b_ci[p] = ncol - b_tap[p] * g.Cin;
      b_cok[p] = ncol < g.Ktot;
    } else {
      const int per_tap = g.Cin / BN;
      b_tap[p] = nt / per_tap;
      b_ci[p] = (nt - b_tap[p] * per_tap) * BN + 4 * c4;
      b_cok[p] = true;
    }
  }

  const int kt_begin = split * kt_per_split;
  const int nkt_all = (g.M + BK - 1) / BK;
  const int kt_end = min(kt_begin + kt_per_split, nkt_all);

  float4 ra[AP], rb[BP];
  auto load = [&](int kt) {
    const int m0 = kt * BK;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int idx = tid + 256 * p;
      const int krow = idx / AV, c4 = idx - krow * AV;
      const int m = m0 + krow;
      ra[p] = (m < g.M) ? *reinterpret_cast<const float4*>(dy + (size_t)m * g.Cout + mt * BM + 4 * c4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int m = m0 + b_krow[p];
      const int mm = m < g.M ? m : 0;
      const int n = mm / HoWo;
      const int rem = mm - n * HoWo;
      const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
      const int r = b_tap[p] / g.S, s = b_tap[p] - r * g.S;
      const int hi = ho * g.stride - g.pad + r, wi = wo * g.stride - g.pad + s;
      const int cls = shift_class(b_ci[p], g.fold);
      const int t = n % g.T;
      const bool v = m < g.M && b_cok[p] && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W &&
                     (unsigned)(t + cls) < (unsigned)g.T;
      const size_t off = ((size_t)((n + cls) * g.H + hi) * g.W + wi) * g.Cin + b_ci[p];
      rb[p] = v ? *reinterpret_cast<const float4*>(x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * LDA;
    store_direct<LDA, BM, AP>(As, ra, tid);
    store_direct<LDB, BN, BP>(Bs, rb, tid);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (kt_begin < kt_end) {
    load(kt_begin);
    store(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int cur = (kt - kt_begin) & 1;
      if (kt + 1 < kt_end) load(kt + 1);
      const float* As = smem + cur * STAGE;
      mma_stage<LDA, LDB, TM, TN>(As, As + BK * LDA, acc, wm0, wn0, lane);
      if (kt + 1 < kt_end) store(cur ^ 1);
      __syncthreads();
    }
  }

  float* out = slab + (size_t)split * g.Cout * g.Ktot;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int col;
    bool cok = true;
    if (VEC_TAP) {
      col = nt * BN + wn0 + 32 * j + (lane & 31);
      cok = col < g.Ktot;
    } else {
      const int per_tap = g.Cin / BN;
      const int tap = nt / per_tap;
      col = tap * g.Cin + (nt - tap * per_tap) * BN + wn0 + 32 * j + (lane & 31);
    }
    if (!cok) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mt * BM + wm0 + 32 * i + acc_row(e, lane);
        out[(size_t)row * g.Ktot + col] = acc[i][j][e];
      }
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, float beta, int splits,
                                    int64_t numel4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel4; i += stride) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 v = reinterpret_cast<const float4*>(slab)[(int64_t)k * numel4 + i];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    float4* d = reinterpret_cast<float4*>(dw) + i;
    if (beta != 0.f) {
      const float4 o = *d;
      s.x += beta * o.x;
      s.y += beta * o.y;
      s.z += beta * o.z;
      s.w += beta * o.w;
    }
    *d = s;
  }
}

int check_geom(const bdv_conv_geom* g, const char* who) {
  BDV_REQUIRE(g != nullptr, "%s: geom is NULL", who);
  BDV_REQUIRE(g->N > 0 && g->H > 0 && g->W > 0 && g->Ho > 0 && g->Wo > 0, "%s: non-positive extent", who);
  BDV_REQUIRE(g->Cin > 0 && g->Cin % 4 == 0, "%s: Cin=%d must be a positive multiple of 4", who, g->Cin);
  BDV_REQUIRE(g->Cout > 0 && g->Cout % 64 == 0, "%s: Cout=%d must be a multiple of 64", who, g->Cout);
  BDV_REQUIRE(g->R > 0 && g->S > 0 && g->pad >= 0, "%s: bad filter", who);
  BDV_REQUIRE(g->stride == 1 || g->stride == 2, "%s: stride %d unsupported", who, g->stride);
  BDV_REQUIRE(g->Ho == (g->H + 2 * g->pad - g->R) / g->stride + 1 && g->Wo == (g->W + 2 * g->pad - g->S) / g->stride + 1,
              "%s: Ho/Wo inconsistent with H/W/pad/stride", who);
  BDV_REQUIRE(g->Cin % BK == 0 || g->Cin == 4, "%s: Cin=%d must be a multiple of 16 (or exactly 4)", who, g->Cin);
  if (g->fold > 0) {
    BDV_REQUIRE(g->fold % 4 == 0 && 2 * g->fold <= g->Cin, "%s: fold=%d must be a multiple of 4 and <= Cin/2", who,
                g->fold);
    BDV_REQUIRE(g->T > 0 && g->N % g->T == 0, "%s: N=%d not a multiple of T=%d", who, g->N, g->T);
  }
  BDV_REQUIRE((int64_t)g->N * g->H * g->W * g->Cin < (1ll << 31) && (int64_t)g->N * g->Ho * g->Wo * g->Cout < (1ll << 31),
              "%s: tensor exceeds 2^31 elements", who);
  return BDV_OK;
}

Geom make_geom(const bdv_conv_geom* g) {
  Geom d;
  d.N = g->N; d.H = g->H; d.W = g->W; d.Cin = g->Cin; d.Ho = g->Ho; d.Wo = g->Wo; d.Cout = g->Cout;
  d.R = g->R; d.S = g->S; d.stride = g->stride; d.pad = g->pad;
  d.T = g->fold > 0 ? g->T : 1;
  d.fold = g->fold;
  d.M = 0; d.Ktot = 0;
  return d;
}

struct WgradPlan {
  bool small;  // 64x64 tiles
  bool vec_tap;
  int MTw, NTw, splits, kt_per_split;
};

WgradPlan plan_wgrad(const bdv_conv_geom* g) {
  WgradPlan p;
  p.vec_tap = (g->Cin % BK) != 0;
  p.small = p.vec_tap || (g->Cout % 128) != 0 || (g->Cin % 128) != 0;
  const int bm = p.small ? 64 : 128, bn = p.small ? 64 : 128;
  const int ktot = g->R * g->S * g->Cin;
  p.MTw = g->Cout / bm;
  p.NTw = p.vec_tap ? (ktot + bn - 1) / bn : g->R * g->S * (g->Cin / bn);
  const int64_t M = (int64_t)g->N * g->Ho * g->Wo;
  const int nkt = (int)((M + BK - 1) / BK);
  const int tiles = p.MTw * p.NTw;
  int splits = (1536 + tiles - 1) / tiles;
  if (splits > 512) splits = 512;
  int per = (nkt + splits - 1) / splits;
  if (per < 8) per = 8;
  p.kt_per_split = per;
  p.splits = (nkt + per - 1) / per;
  return p;
}

}  // namespace

extern "C" int bdv_conv_fprop(const float* x, const float* w, float* y, const bdv_conv_geom* gg, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_fprop")) return e;
  BDV_REQUIRE(x && w && y, "bdv_conv_fprop: null pointer");
  BDV_REQUIRE(bdv_aligned16(x) && bdv_aligned16(w) && bdv_aligned16(y), "bdv_conv_fprop: pointers must be 16-byte aligned");
  Geom g = make_geom(gg);
  g.M = g.N * g.Ho * g.Wo;
  g.Ktot = g.R * g.S * g.Cin;
  hipStream_t s = (hipStream_t)stream;
  const bool vec_tap = (g.Cin % BK) != 0;
  BDV_REQUIRE(!(vec_tap && g.fold > 0), "bdv_conv_fprop: shift needs Cin %% 16 == 0");
  if (g.Cout % 128 == 0) {
    constexpr int BM = 128, BN = 128;
    const int MT = (g.M + BM - 1) / BM, NT = g.Cout / BN;
    const int grid = ((MT + 7) / 8) * 8 * NT;
    if (vec_tap)
      hipLaunchKernelGGL((conv_fprop_kernel<BM, BN, 2, 2, true>), dim3(grid), dim3(256), 0, s, x, w, y, g, MT, NT);
    else
      hipLaunchKernelGGL((conv_fprop_kernel<BM, BN, 2, 2, false>), dim3(grid), dim3(256), 0, s, x, w, y, g, MT, NT);
  } else {
    constexpr int BM = 256, BN = 64;
    const int MT = (g.M + BM - 1) / BM, NT = g.Cout / BN;
    const int grid = ((MT + 7) / 8) * 8 * NT;
    if (vec_tap)
      hipLaunchKernelGGL((conv_fprop_kernel<BM, BN, 4, 1, true>), dim3(grid), dim3(256), 0, s, x, w, y, g, MT, NT);
    else
      hipLaunchKernelGGL((conv_fprop_kernel<BM, BN, 4, 1, false>), dim3(grid), dim3(256), 0, s, x, w, y, g, MT, NT);
  }
  BDV_LAUNCH_CHECK("bdv_conv_fprop");
  return BDV_OK;
}

extern "C" int bdv_conv_dgrad(const float* dy, const float* w, float* dx, const float* add_src,
                              const float* add_mask_src, const bdv_conv_geom* gg, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_dgrad")) return e;
  BDV_REQUIRE(dy && w && dx, "bdv_conv_dgrad: null pointer");
  BDV_REQUIRE(bdv_aligned16(dy) && bdv_aligned16(w) && bdv_aligned16(dx), "bdv_conv_dgrad: pointers must be 16-byte aligned");
  BDV_REQUIRE(gg->Cin % 64 == 0, "bdv_conv_dgrad: Cin=%d must be a multiple of 64", gg->Cin);
  BDV_REQUIRE(add_src != nullptr || add_mask_src == nullptr, "bdv_conv_dgrad: add_mask_src without add_src");
  Geom g = make_geom(gg);
  g.M = g.N * g.H * g.W;
  g.Ktot = g.R * g.S * g.Cout;
  hipStream_t s = (hipStream_t)stream;
  const int st = g.stride;
  const int Mc0 = g.N * ((g.H + st - 1) / st) * ((g.W + st - 1) / st);   // largest parity class
  if (g.Cin % 128 == 0) {
    constexpr int BM = 128, BN = 128;
    const int MT = (Mc0 + BM - 1) / BM, NT = g.Cin / BN;
    const dim3 grid(((MT + 7) / 8) * 8 * NT, st * st);
    hipLaunchKernelGGL((conv_dgrad_kernel<BM, BN, 2, 2>), grid, dim3(256), 0, s, dy, w, dx, add_src, add_mask_src, g, NT);
  } else {
    constexpr int BM = 256, BN = 64;
    const int MT = (Mc0 + BM - 1) / BM, NT = g.Cin / BN;
    const dim3 grid(((MT + 7) / 8) * 8 * NT, st * st);
    hipLaunchKernelGGL((conv_dgrad_kernel<BM, BN, 4, 1>), grid, dim3(256), 0, s, dy, w, dx, add_src, add_mask_src, g, NT);
  }
  BDV_LAUNCH_CHECK("bdv_conv_dgrad");
  return BDV_OK;
}

extern "C" size_t bdv_conv_wgrad_workspace_bytes(const bdv_conv_geom* g) {
  if (check_geom(g, "bdv_conv_wgrad_workspace_bytes")) return 0;
  const WgradPlan p = plan_wgrad(g);
  return (size_t)p.splits * g->Cout * g->R * g->S * g->Cin * sizeof(float);
}

extern "C" int bdv_conv_wgrad(const float* dy, const float* x, float* dw, float beta, const bdv_conv_geom* gg,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_wgrad")) return e;
  BDV_REQUIRE(dy && x && dw && workspace, "bdv_conv_wgrad: null pointer");
  BDV_REQUIRE(bdv_aligned16(dy) && bdv_aligned16(x) && bdv_aligned16(dw) && bdv_aligned16(workspace),
              "bdv_conv_wgrad: pointers must be 16-byte aligned");
  const WgradPlan p = plan_wgrad(gg);
  const size_t need = bdv_conv_wgrad_workspace_bytes(gg);
  if (workspace_bytes < need) {
    bdv_set_error("bdv_conv_wgrad: workspace %zu < required %zu bytes", workspace_bytes, need);
    return BDV_EWORKSPACE;
  }
  BDV_REQUIRE(!(p.vec_tap && gg->fold > 0), "bdv_conv_wgrad: shift needs Cin %% 16 == 0");
  Geom g = make_geom(gg);
  g.M = g.N * g.Ho * g.Wo;
  g.Ktot = g.R * g.S * g.Cin;
  hipStream_t s = (hipStream_t)stream;
  float* slab = (float*)workspace;
  const dim3 grid(p.MTw * p.NTw, p.splits);
  if (!p.small) {
    hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, false>), grid, dim3(256), 0, s, dy, x, slab, g, p.MTw, p.NTw,
                       p.kt_per_split);
  } else if (p.vec_tap) {
    hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, true>), grid, dim3(256), 0, s, dy, x, slab, g, p.MTw, p.NTw,
                       p.kt_per_split);
  } else {
    hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, false>), grid, dim3(256), 0, s, dy, x, slab, g, p.MTw, p.NTw,
                       p.kt_per_split);
  }
  BDV_LAUNCH_CHECK("bdv_conv_wgrad");
  const int64_t numel4 = (int64_t)g.Cout * g.Ktot / 4;
  int rb = (int)((numel4 + 255) / 256);
  if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, s, (const float*)slab, dw, beta, p.splits, numel4);
  BDV_LAUNCH_CHECK("bdv_conv_wgrad(reduce)");
  return BDV_OK;
}
