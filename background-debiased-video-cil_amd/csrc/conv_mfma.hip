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
#include <stdlib.h>
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
  // operand fragments are fetched one K-step ahead of the MFMAs that consume them
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
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    // pin the order: next step's LDS reads first, then this step's MFMAs (hipcc otherwise sinks the reads
    // to just before their use and exposes the LDS latency every K-step)
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
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

// ---- work decomposition -------------------------------------------------------------------
// A launch covers `dp_tiles` whole output tiles (one block each, XCD-aware order) followed by `rem_tiles` tiles
// whose K loop is cut into `split` slices (one block per slice, partial accumulators to a slab, summed in fixed
// order by a fix-up kernel).  The host planner picks (dp_tiles, split) so that the block count is a multiple of
// the number of co-resident blocks: on 256 CUs a tile count just above a multiple otherwise costs a whole extra
// round (measured: 784 tiles -> 82 TFLOP/s, 768 or 1024 tiles -> 113-120).
struct Work {
  int dp_tiles, rem_tiles, split;
};

// bijective remap so that blocks b, b+8, b+16, ... (same XCD under round-robin dispatch) get consecutive tiles
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, j = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

struct WorkItem {
  int tile, kb, ke, pslot;  // pslot < 0: whole tile, write the result; else partial-slab segment index
};

__device__ __forceinline__ WorkItem get_work(int b, const Work& wk, int nk) {
  WorkItem it;
  if (b < wk.dp_tiles) {
    it.tile = xcd_remap(b, wk.dp_tiles);
    it.kb = 0;
    it.ke = nk;
    it.pslot = -1;
  } else {
    const int rem = b - wk.dp_tiles;
    const int rt = rem / wk.split, sl = rem - rt * wk.split;
    it.tile = wk.dp_tiles + rt;
    it.kb = (int)(((long long)sl * nk) / wk.split);
    it.ke = (int)(((long long)(sl + 1) * nk) / wk.split);
    it.pslot = wk.split > 1 ? rem : -1;
  }
  return it;
}

template <int TM, int TN>
__device__ __forceinline__ void store_partial(float* __restrict__ slab, int pslot, const f32x16 (&acc)[TM][TN], int tid) {
  float* base = slab + (size_t)pslot * (TM * TN * 16) * 256 + tid;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) base[(size_t)((i * TN + j) * 16 + e) * 256] = acc[i][j][e];
}

__device__ __forceinline__ void fprop_store(float* __restrict__ y, const Geom& g, int row, int col, float v) {
  if (row < g.M) y[(size_t)row * g.Cout + col] = v;
}

// dgrad element store for stride 1: temporal un-shift scatter + optional masked residual add (see kernel comment)
__device__ __forceinline__ void dgrad_store(float* __restrict__ dx, const float* __restrict__ add_src,
                                            const uint32_t* __restrict__ add_mask, const Geom& g, int HW, int row, int n_hint,
                                            bool have_n, int col, float v) {
  const int cls = shift_class(col, g.fold);
  int drow = row;
  if (cls != 0) {
    const int n = have_n ? n_hint : row / HW;
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
    if (add_mask != nullptr && !((add_mask[o >> 5] >> (o & 31)) & 1u)) a = 0.f;
    v += a;
  }
  dx[o] = v;
}

// BatchNorm batch statistics fused into the fprop epilogue: per tile, the column sums of y and y^2 over the
// tile's valid rows -> bn_partial[0][mt][co], bn_partial[1][mt][co]; a fixed-order fp64 finalize sums the MT rows.
// cs/cq hold this lane's sums over its accumulator registers for each of its TN columns.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void tile_colstats(float* __restrict__ smem, float (&cs)[BN / WN / 32], float (&cq)[BN / WN / 32],
                                              float* __restrict__ bn_partial, int MT, int Cout, int mt, int nt, int tid) {
  constexpr int TN = BN / WN / 32;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn0 = (wave % WN) * (BN / WN);
#pragma unroll
  for (int j = 0; j < TN; ++j) {  // lanes l and l+32 hold the same column, different rows
    cs[j] += __shfl_xor(cs[j], 32, 64);
    cq[j] += __shfl_xor(cq[j], 32, 64);
  }
  __syncthreads();  // every wave is done with the operand stages in LDS
  if (lane < 32) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      smem[wm * BN + wn0 + 32 * j + lane] = cs[j];
      smem[(WM + wm) * BN + wn0 + 32 * j + lane] = cq[j];
    }
  }
  __syncthreads();
  if (tid < BN) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < WM; ++k) {
      a += smem[k * BN + tid];
      b += smem[(WM + k) * BN + tid];
    }
    bn_partial[(size_t)mt * Cout + nt * BN + tid] = a;
    bn_partial[((size_t)MT + mt) * Cout + nt * BN + tid] = b;
  }
}

// =========================================================================================
// fprop: y[m, co] = sum_{tap, ci} x_shift[n, ho*st + r - p, wo*st + s - p, ci] * w[co, tap, ci]
// =========================================================================================
template <int BM, int BN, int WM, int WN, bool VEC_TAP>
__global__ __launch_bounds__(256) void conv_fprop_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, Geom g, int NT, Work wk,
                                                          float* __restrict__ slab, float* __restrict__ bn_partial, int MT) {
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 64, BP = BN / 64;
  constexpr int STAGE = BK * (LDA + LDB);
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  // Tile order: Cout-tile fastest, so the blocks that share an activation row-tile are consecutive and
  // (through xcd_remap) land on one XCD / one L2.
  const int nk = (g.Ktot + BK - 1) / BK;
  const WorkItem it = get_work(blockIdx.x, wk, nk);
  const int mt = it.tile / NT, nt = it.tile - mt * NT;

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
      // tap-fastest K order: the R*S taps of one 16-channel chunk are consecutive K-steps, so the shifted
      // re-reads of the same activation rows hit L1/L2 instead of going back to the fabric R*S times.
      const int chunk = kt / RS;
      tap = kt - chunk * RS;
      c = chunk * BK + 4 * kg;
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
    const int kb = VEC_TAP ? kt * BK + 4 * kg : tap * g.Cin + c;
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

  load(it.kb);
  store(0);
  __syncthreads();
  for (int kt = it.kb; kt < it.ke; ++kt) {
    const int cur = (kt - it.kb) & 1;
    if (kt + 1 < it.ke) load(kt + 1);
    const float* As = smem + cur * STAGE;
    mma_stage<LDA, LDB, TM, TN>(As, As + BK * LDA, acc, wm0, wn0, lane);
    if (kt + 1 < it.ke) store(cur ^ 1);
    __syncthreads();
  }

  if (it.pslot >= 0) {
    store_partial<TM, TN>(slab, it.pslot, acc, tid);
    return;
  }
  float cs[TN], cq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = nt * BN + wn0 + 32 * j + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mt * BM + wm0 + 32 * i + acc_row(e, lane);
        const float v = acc[i][j][e];
        fprop_store(y, g, row, col, v);
        if (row < g.M) {
          cs[j] += v;
          cq[j] += v * v;
        }
      }
    }
  if (bn_partial != nullptr)
    tile_colstats<BM, BN, WM, WN>(smem, cs, cq, bn_partial, MT, g.Cout, mt, nt, tid);
}

// fix-up for the K-split remainder tiles: sum the `split` partial accumulators in slice order, then the same
// element store (and BN column statistics) as the main kernel.  grid = rem_tiles, 256 threads mapped like the
// main kernel.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_fprop_fixup_kernel(const float* __restrict__ slab, float* __restrict__ y, Geom g,
                                                                int NT, Work wk, float* __restrict__ bn_partial, int MT) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NACC = TM * TN * 16;
  __shared__ float smem[2 * WM * BN];
  const int rt = blockIdx.x, tile = wk.dp_tiles + rt;
  const int mt = tile / NT, nt = tile - mt * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  // all NACC loads of one slice are independent: keep them in flight together, slices summed in order
  float v[TM * TN * 16];
#pragma unroll
  for (int f = 0; f < NACC; ++f) v[f] = 0.f;
  for (int sl = 0; sl < wk.split; ++sl) {
    const float* src = slab + (size_t)(rt * wk.split + sl) * NACC * 256 + tid;
#pragma unroll
    for (int f = 0; f < NACC; ++f) v[f] += src[(size_t)f * 256];
  }
  float cs[TN], cq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float val = v[(i * TN + j) * 16 + e];
        const int row = mt * BM + wm0 + 32 * i + acc_row(e, lane);
        fprop_store(y, g, row, nt * BN + wn0 + 32 * j + (lane & 31), val);
        if (row < g.M) {
          cs[j] += val;
          cq[j] += val * val;
        }
      }
  if (bn_partial != nullptr) tile_colstats<BM, BN, WM, WN>(smem, cs, cq, bn_partial, MT, g.Cout, mt, nt, tid);
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
                                                          const uint32_t* __restrict__ add_mask, Geom g, int NT, Work wk,
                                                          float* __restrict__ slab) {
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

  const int nk = nr * ns * g.Cout / BK;     // 0 for a class no filter tap reaches (e.g. 1x1 stride 2, odd pixels)
  int mt, nt;
  WorkItem it;
  if (st == 1) {
    it = get_work(blockIdx.x, wk, nk);
    mt = it.tile / NT;
    nt = it.tile - mt * NT;
  } else {  // parity classes have different sizes: padded grid, no K split
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    mt = (jj / NT) * 8 + xcd;
    nt = jj % NT;
    if (mt >= MT) return;
    it.tile = 0; it.kb = 0; it.ke = nk; it.pslot = -1;
  }

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
    const int ntap = nr * ns;               // tap-fastest K order (see fprop)
    const int chunk = kt / ntap;
    const int ct = kt - chunk * ntap;       // tap index inside this class
    const int co0 = chunk * BK;
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

  if (it.ke > it.kb) {
    load(it.kb);
    store(0);
    __syncthreads();
    for (int kt = it.kb; kt < it.ke; ++kt) {
      const int cur = (kt - it.kb) & 1;
      if (kt + 1 < it.ke) load(kt + 1);
      const float* As = smem + cur * STAGE;
      mma_stage<LDA, LDB, TM, TN>(As, As + BK * LDA, acc, wm0, wn0, lane);
      if (kt + 1 < it.ke) store(cur ^ 1);
      __syncthreads();
    }
  }
  if (it.pslot >= 0) {
    store_partial<TM, TN>(slab, it.pslot, acc, tid);
    return;
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
      int row = mrow, n = 0;
      if (st != 1) {
        n = mrow / HcWc;
        const int rem = mrow - n * HcWc;
        const int hc = rem / Wc;
        row = (n * g.H + hc * st + ph) * g.W + (rem - hc * Wc) * st + pw;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
        dgrad_store(dx, add_src, add_mask, g, HW, row, n, st != 1, nt * BN + wn0 + 32 * j + (lane & 31), acc[i][j][e]);
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_dgrad_fixup_kernel(const float* __restrict__ slab, float* __restrict__ dx,
                                                                const float* __restrict__ add_src,
                                                                const uint32_t* __restrict__ add_mask, Geom g, int NT, Work wk) {
  constexpr int TN = BN / WN / 32;
  constexpr int NACC = (BM / WM / 32) * TN * 16;
  const int rt = blockIdx.x, tile = wk.dp_tiles + rt;
  const int mt = tile / NT, nt = tile - mt * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int HW = g.H * g.W;
#pragma unroll 4
  for (int q = 0; q < 16; ++q) {
    const int f = blockIdx.y * 16 + q;
    float v = 0.f;
    for (int sl = 0; sl < wk.split; ++sl) v += slab[((size_t)(rt * wk.split + sl) * NACC + f) * 256 + tid];
    const int e = f & 15, ij = f >> 4, i = ij / TN, j = ij - i * TN;
    const int row = mt * BM + wm0 + 32 * i + acc_row(e, lane);
    if (row < g.M) dgrad_store(dx, add_src, add_mask, g, HW, row, 0, false, nt * BN + wn0 + 32 * j + (lane & 31), v);
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

// ---- host-side planning ---------------------------------------------------------------------

bool debug_plan() {
  static const bool on = getenv("BDVCIL_DEBUG_PLAN") != nullptr;
  return on;
}

// number of co-resident 256-thread blocks of `kernel` on the whole device (occupancy API x CU count)
template <class KernelT>
int resident_blocks(KernelT kernel) {
  int dev = 0, cus = 256, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 768;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, 256, 0) != hipSuccess || per_cu <= 0) {
    (void)hipGetLastError();
    per_cu = 3;
  }
  return cus * per_cu;
}

// Pick the K-split of the remainder tiles.  A CU is MFMA-bound once it holds >= 3 blocks, so the launch time is
// set by the CU with the most work: whole rounds of `cus` tiles run data-parallel, the last partial round
// (r < cus tiles) is cut along K into `split` slices per tile so that its blocks spread over all CUs.
// Cost model in microseconds: one K-iteration of a 64-accumulator block = 0.85 us of one CU's MFMA pipes;
// a CU with 1 / 2 blocks reaches ~65 % / ~85 % of that rate; the fix-up moves 2 x seg_bytes per slice, mostly
// through the Infinity Cache (~8 TB/s), and is latency-bound per slice.
Work plan_work(int tiles, int nk, int W, size_t seg_bytes, size_t ws_bytes) {
  const int cus = 256;
  (void)W;
  Work wk;
  const int q = tiles / cus, r = tiles - q * cus;
  wk.dp_tiles = tiles;
  wk.rem_tiles = 0;
  wk.split = 1;
  if (r == 0 || nk < 8) return wk;
  auto eff = [&](int c) { return (q + c) >= 3 ? 1.0 : ((q + c) == 2 ? 0.85 : 0.65); };
  double best = 1e30;
  int best_s = 1;
  for (int sp = 1; sp <= 32 && nk / sp >= 4; ++sp) {
    const size_t need = (size_t)r * sp * seg_bytes;
    if (sp > 1 && need > ws_bytes) break;
    const int c = (int)(((long long)r * sp + cus - 1) / cus);
    double cost = c * ((double)nk / sp) * 0.85 / eff(c);
    // fix-up launch: each thread walks the slices serially (measured ~2.5 us per slice) + launch boundary
    if (sp > 1) cost += 10.0 + 2.5 * sp + (double)need * 2.0 / 8.0e6;
    if (cost < best * 0.95) {  // prefer fewer slices unless clearly better
      best = cost;
      best_s = sp;
    }
  }
  if (best_s > 1) {
    wk.dp_tiles = q * cus;
    wk.rem_tiles = r;
    wk.split = best_s;
  }
  return wk;
}

constexpr size_t kSegBytes128 = 64 * 256 * sizeof(float);  // 128x128 or 256x64 tile: 64 accumulator regs x 256 threads
constexpr size_t kMaxSplitWorkspace = 160ull << 20;

struct WgradPlan {
  bool small;  // 64x64 tiles
  bool vec_tap;
  int MTw, NTw, splits, kt_per_split;
};

WgradPlan plan_wgrad(const bdv_conv_geom* g, int W) {
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
  // blocks = tiles * splits should fill whole rounds of W co-resident blocks; more rounds = less slab traffic per
  // flop is not true here (slab bytes grow with splits), so take the smallest round count whose fill is >= 90 %.
  int best_s = 1;
  double best_cost = 1e30;
  const double dw_bytes = (double)g->Cout * ktot * 4.0;
  for (int k = 1; k <= 4; ++k) {
    int sp = (int)(((long long)k * W) / tiles);
    if (sp < 1) sp = 1;
    if (sp > 1024) sp = 1024;
    while (sp > 1 && (nkt + sp - 1) / sp < 8) --sp;
    const int per = (nkt + sp - 1) / sp;
    const double rounds = (double)(((long long)tiles * sp + W - 1) / W);
    const double t_iter = 0.85 * ((double)W / 256.0) * (p.small ? 0.3 : 1.0);
    const double cost = rounds * per * t_iter + 4.0 + sp * dw_bytes / 4.0e6;
    if (cost < best_cost) {
      best_cost = cost;
      best_s = sp;
    }
  }
  int per = (nkt + best_s - 1) / best_s;
  p.kt_per_split = per;
  p.splits = (nkt + per - 1) / per;
  return p;
}

int wgrad_resident(const bdv_conv_geom* g) {
  const bool vec_tap = (g->Cin % BK) != 0;
  const bool small = vec_tap || (g->Cout % 128) != 0 || (g->Cin % 128) != 0;
  static const int w_big = resident_blocks(conv_wgrad_kernel<128, 128, 2, 2, false>);
  static const int w_small = resident_blocks(conv_wgrad_kernel<64, 64, 2, 2, false>);
  return small ? w_small : w_big;
}

struct FdPlan {
  bool wide;  // 128x128 tile (else 256x64)
  int MT, NT, nk, W;
  Work wk;
};

FdPlan plan_fprop(const Geom& g, size_t ws_bytes) {
  FdPlan p;
  p.wide = (g.Cout % 128) == 0;
  const int bm = p.wide ? 128 : 256, bn = p.wide ? 128 : 64;
  p.MT = (g.M + bm - 1) / bm;
  p.NT = g.Cout / bn;
  p.nk = (g.Ktot + BK - 1) / BK;
  const bool vec_tap = (g.Cin % BK) != 0;
  static const int w0 = resident_blocks(conv_fprop_kernel<128, 128, 2, 2, false>);
  static const int w1 = resident_blocks(conv_fprop_kernel<256, 64, 4, 1, false>);
  static const int w2 = resident_blocks(conv_fprop_kernel<256, 64, 4, 1, true>);
  p.W = p.wide ? w0 : (vec_tap ? w2 : w1);
  p.wk = plan_work(p.MT * p.NT, p.nk, p.W, kSegBytes128, ws_bytes);
  return p;
}

FdPlan plan_dgrad(const Geom& g, size_t ws_bytes) {
  FdPlan p;
  p.wide = (g.Cin % 128) == 0;
  const int bm = p.wide ? 128 : 256, bn = p.wide ? 128 : 64;
  const int st = g.stride;
  const int Mc0 = g.N * ((g.H + st - 1) / st) * ((g.W + st - 1) / st);  // largest parity class
  p.MT = (Mc0 + bm - 1) / bm;
  p.NT = g.Cin / bn;
  p.nk = g.R * g.S * g.Cout / BK;
  static const int w0 = resident_blocks(conv_dgrad_kernel<128, 128, 2, 2>);
  static const int w1 = resident_blocks(conv_dgrad_kernel<256, 64, 4, 1>);
  p.W = p.wide ? w0 : w1;
  if (st == 1) {
    p.wk = plan_work(p.MT * p.NT, p.nk, p.W, kSegBytes128, ws_bytes);
  } else {
    p.wk.dp_tiles = ((p.MT + 7) / 8) * 8 * p.NT;  // padded grid per parity class
    p.wk.rem_tiles = 0;
    p.wk.split = 1;
  }
  return p;
}

}  // namespace

extern "C" size_t bdv_conv_workspace_bytes(const bdv_conv_geom* gg, int kind) {
  if (check_geom(gg, "bdv_conv_workspace_bytes")) return 0;
  if (kind == 2) {
    const WgradPlan p = plan_wgrad(gg, wgrad_resident(gg));
    return (size_t)p.splits * gg->Cout * gg->R * gg->S * gg->Cin * sizeof(float);
  }
  Geom g = make_geom(gg);
  FdPlan p;
  if (kind == 0) {
    g.M = g.N * g.Ho * g.Wo;
    g.Ktot = g.R * g.S * g.Cin;
    p = plan_fprop(g, kMaxSplitWorkspace);
  } else {
    if (gg->Cin % 64 != 0) return 0;
    g.M = g.N * g.H * g.W;
    g.Ktot = g.R * g.S * g.Cout;
    p = plan_dgrad(g, kMaxSplitWorkspace);
  }
  const size_t need = p.wk.split > 1 ? (size_t)p.wk.rem_tiles * p.wk.split * kSegBytes128 : 0;
  return need > 16 ? need : 16;
}

extern "C" int bdv_conv_fprop_stat_rows(const bdv_conv_geom* gg) {
  if (check_geom(gg, "bdv_conv_fprop_stat_rows")) return 0;
  const int bm = (gg->Cout % 128) == 0 ? 128 : 256;
  const int64_t M = (int64_t)gg->N * gg->Ho * gg->Wo;
  return (int)((M + bm - 1) / bm);
}

extern "C" int bdv_conv_fprop(const float* x, const float* w, float* y, const bdv_conv_geom* gg, float* bn_partial,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_fprop")) return e;
  BDV_REQUIRE(x && w && y, "bdv_conv_fprop: null pointer");
  BDV_REQUIRE(bdv_aligned16(x) && bdv_aligned16(w) && bdv_aligned16(y) && bdv_aligned16(workspace),
              "bdv_conv_fprop: pointers must be 16-byte aligned");
  Geom g = make_geom(gg);
  g.M = g.N * g.Ho * g.Wo;
  g.Ktot = g.R * g.S * g.Cin;
  hipStream_t s = (hipStream_t)stream;
  const bool vec_tap = (g.Cin % BK) != 0;
  BDV_REQUIRE(!(vec_tap && g.fold > 0), "bdv_conv_fprop: shift needs Cin %% 16 == 0");
  const FdPlan p = plan_fprop(g, workspace ? (workspace_bytes < kMaxSplitWorkspace ? workspace_bytes : kMaxSplitWorkspace) : 0);
  const int blocks = p.wk.dp_tiles + p.wk.rem_tiles * p.wk.split;
  float* slab = (float*)workspace;
  if (debug_plan())
    fprintf(stderr, "[bdv plan] fprop %dx%d Cin %d Cout %d k%d s%d: tiles %d nk %d W %d -> dp %d rem %d split %d\n", g.H, g.W,
            g.Cin, g.Cout, g.R, g.stride, p.MT * p.NT, p.nk, p.W, p.wk.dp_tiles, p.wk.rem_tiles, p.wk.split);
  if (p.wide) {
    if (vec_tap)
      hipLaunchKernelGGL((conv_fprop_kernel<128, 128, 2, 2, true>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, p.wk, slab, bn_partial, p.MT);
    else
      hipLaunchKernelGGL((conv_fprop_kernel<128, 128, 2, 2, false>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, p.wk, slab, bn_partial, p.MT);
  } else {
    if (vec_tap)
      hipLaunchKernelGGL((conv_fprop_kernel<256, 64, 4, 1, true>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, p.wk, slab, bn_partial, p.MT);
    else
      hipLaunchKernelGGL((conv_fprop_kernel<256, 64, 4, 1, false>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, p.wk, slab, bn_partial, p.MT);
  }
  BDV_LAUNCH_CHECK("bdv_conv_fprop");
  if (p.wk.split > 1) {
    const dim3 fg(p.wk.rem_tiles);
    if (p.wide)
      hipLaunchKernelGGL((conv_fprop_fixup_kernel<128, 128, 2, 2>), fg, dim3(256), 0, s, (const float*)slab, y, g, p.NT, p.wk,
                         bn_partial, p.MT);
    else
      hipLaunchKernelGGL((conv_fprop_fixup_kernel<256, 64, 4, 1>), fg, dim3(256), 0, s, (const float*)slab, y, g, p.NT, p.wk,
                         bn_partial, p.MT);
    BDV_LAUNCH_CHECK("bdv_conv_fprop(fixup)");
  }
  return BDV_OK;
}

extern "C" int bdv_conv_dgrad(const float* dy, const float* w, float* dx, const float* add_src,
                              const uint32_t* add_mask_src, const bdv_conv_geom* gg, void* workspace, size_t workspace_bytes,
                              void* stream) {
  if (int e = check_geom(gg, "bdv_conv_dgrad")) return e;
  BDV_REQUIRE(dy && w && dx, "bdv_conv_dgrad: null pointer");
  BDV_REQUIRE(bdv_aligned16(dy) && bdv_aligned16(w) && bdv_aligned16(dx) && bdv_aligned16(workspace),
              "bdv_conv_dgrad: pointers must be 16-byte aligned");
  BDV_REQUIRE(gg->Cin % 64 == 0, "bdv_conv_dgrad: Cin=%d must be a multiple of 64", gg->Cin);
  BDV_REQUIRE(add_src != nullptr || add_mask_src == nullptr, "bdv_conv_dgrad: add_mask_src without add_src");
  Geom g = make_geom(gg);
  g.M = g.N * g.H * g.W;
  g.Ktot = g.R * g.S * g.Cout;
  hipStream_t s = (hipStream_t)stream;
  const int st = g.stride;
  const FdPlan p = plan_dgrad(g, workspace ? (workspace_bytes < kMaxSplitWorkspace ? workspace_bytes : kMaxSplitWorkspace) : 0);
  const dim3 grid(p.wk.dp_tiles + p.wk.rem_tiles * p.wk.split, st * st);
  float* slab = (float*)workspace;
  if (debug_plan())
    fprintf(stderr, "[bdv plan] dgrad %dx%d Cin %d Cout %d k%d s%d: tiles %d nk %d W %d -> dp %d rem %d split %d\n", g.H, g.W,
            g.Cin, g.Cout, g.R, g.stride, p.MT * p.NT, p.nk, p.W, p.wk.dp_tiles, p.wk.rem_tiles, p.wk.split);
  if (p.wide)
    hipLaunchKernelGGL((conv_dgrad_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, dy, w, dx, add_src, add_mask_src, g, p.NT, p.wk, slab);
  else
    hipLaunchKernelGGL((conv_dgrad_kernel<256, 64, 4, 1>), grid, dim3(256), 0, s, dy, w, dx, add_src, add_mask_src, g, p.NT, p.wk, slab);
  BDV_LAUNCH_CHECK("bdv_conv_dgrad");
  if (p.wk.split > 1) {
    const dim3 fg(p.wk.rem_tiles, 4);
    if (p.wide)
      hipLaunchKernelGGL((conv_dgrad_fixup_kernel<128, 128, 2, 2>), fg, dim3(256), 0, s, (const float*)slab, dx, add_src, add_mask_src, g, p.NT, p.wk);
    else
      hipLaunchKernelGGL((conv_dgrad_fixup_kernel<256, 64, 4, 1>), fg, dim3(256), 0, s, (const float*)slab, dx, add_src, add_mask_src, g, p.NT, p.wk);
    BDV_LAUNCH_CHECK("bdv_conv_dgrad(fixup)");
  }
  return BDV_OK;
}

extern "C" int bdv_conv_wgrad(const float* dy, const float* x, float* dw, float beta, const bdv_conv_geom* gg,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_wgrad")) return e;
  BDV_REQUIRE(dy && x && dw && workspace, "bdv_conv_wgrad: null pointer");
  BDV_REQUIRE(bdv_aligned16(dy) && bdv_aligned16(x) && bdv_aligned16(dw) && bdv_aligned16(workspace),
              "bdv_conv_wgrad: pointers must be 16-byte aligned");
  const WgradPlan p = plan_wgrad(gg, wgrad_resident(gg));
  const size_t need = (size_t)p.splits * gg->Cout * gg->R * gg->S * gg->Cin * sizeof(float);
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
  if (debug_plan())
    fprintf(stderr, "[bdv plan] wgrad %dx%d Cin %d Cout %d k%d s%d: tiles %d W %d -> splits %d x %d k-iters (%s)\n", gg->H, gg->W,
            gg->Cin, gg->Cout, gg->R, gg->stride, p.MTw * p.NTw, wgrad_resident(gg), p.splits, p.kt_per_split,
            p.small ? "64x64" : "128x128");
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
