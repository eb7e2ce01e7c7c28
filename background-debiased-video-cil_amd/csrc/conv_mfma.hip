// Implicit-GEMM convolution (fprop / dgrad / wgrad) on the fp32-input MFMA
// v_mfma_f32_32x32x2_f32 for gfx950.  NHWC activations, [Cout][R][S][Cin] weights.
//
// Replaces: F.conv2d + autograd inside UPSTREAM mmaction ConvModule, and UPSTREAM
// TemporalShift.shift (fused into the activation-tile gather).
//
// Structure of every kernel (tools/ubench/gemm_v2.hip is this loop as a plain GEMM):
//   * 256 threads = 4 waves; block tile BM x BN; K-step 32; wave tile = 2x2 / 2x1 MFMA 32x32 tiles, the tiles of a
//     wave interleaved with those of the other wave (wave w: rows 32w..32w+31 of every 64-row group);
//   * ONE LDS stage: K-major image ([k][m]) of both operands in unpadded rows, so one MFMA operand is one
//     conflict-free ds_read_b32 (lane l reads [k = 2s + l/32][m = base + l%32]) and every read of a stage has an
//     immediate offset (ds_read2st64_b32); fragments are fetched one K-pair ahead of the MFMAs that use them;
//   * register-staged global loads, issued right after the stage is published and consumed after the MFMAs:
//     barrier, store registers -> LDS, barrier, issue next loads, 64 MFMAs;
//   * K-contiguous operands are read as full 128-byte lines (8 lanes x 16 B per row) and stored transposed with an
//     XOR swizzle of the column (store_transposed); M-contiguous operands are stored with 16-byte writes;
//   * halo / clip-end / ragged lanes use buffer loads with an out-of-range offset (hardware returns zeros).
// The kernels are bound by instruction issue (a 32x32x2 fp32 MFMA holds its SIMD for 64 cycles and every other
// instruction adds its issue cycles: utilisation ~ 64 / (64 + 4 * VALU-per-MFMA + 10)), hence the emphasis on
// address arithmetic that folds into immediates or is carried incrementally.
#include <stdlib.h>
#include "common.h"

// ---- in-kernel time stamps (diagnostic builds only: make stamps -> libbdvcil_hip_stamps.so, tools/stamp_tiles.py) --------------
// With -DBDV_STAMPS lane 0 of every workgroup of the plane kernels writes s_memtime at five points of its life (start, first
// K-step published, K loop done, epilogue issued, stores drained) plus s_memrealtime at both ends and its hardware id into a
// buffer of its own (bdv_debug_set_stamps); no output value depends on them.  Without the macro BDV_STAMP expands to nothing:
// the product library executes no stamp.
#ifdef BDV_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
__device__ int g_stamp_cap = 0;
#define BDV_STAMP(slot)                                                                                                          \
  do {                                                                                                                           \
    if (threadIdx.x == 0 && g_stamp_buf != nullptr && (int)(blockIdx.x + gridDim.x * blockIdx.y) < g_stamp_cap) {                 \
      unsigned long long* q_ = g_stamp_buf + (size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 8;                                  \
      if ((slot) == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
      q_[slot] = __builtin_amdgcn_s_memtime();                                                                                   \
      if ((slot) == 0) { q_[5] = __builtin_amdgcn_s_memrealtime(); q_[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } \
      if ((slot) == 4) q_[6] = __builtin_amdgcn_s_memrealtime();                                                                 \
    }                                                                                                                            \
  } while (0)
#else
#define BDV_STAMP(slot) do { } while (0)
#endif

namespace {

constexpr int BK = 32;

struct Geom {
  int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, pad_w, T, fold;  // pad: rows (H), pad_w: columns (W)
  int Rt, st_t;  // temporal filter taps / temporal stride of the I3D stem (Cin = 4 only; 1, 1 otherwise): output frame n reads the
                 // input frames n * st_t + dt - Rt / 2 of its clip, dt = 0 .. Rt - 1 (T = OUTPUT frames per clip, T * st_t input frames)
  float rcp_RS, rcp_S;
  int M;     // GEMM rows: N*Ho*Wo (fprop / wgrad reduction length), N*H*W (dgrad)
  int Ktot;  // fprop: R*S*Cin ; dgrad: R*S*Cout ; wgrad: row length of dw = R*S*Cin
  float rcp_HoWo, rcp_Wo;  // fast division helpers (dividends < 2^22)
  int stagger;             // two-stage plane kernels: SIMD partners run a K-step in opposite order (pl_pipeline2)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kOOB = (int)0x80000000;  // byte offset beyond any tensor (< 2^31 bytes checked on the host)

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t rsrc, int voff, float4 v) {
  u32x4 d;
  d.x = __float_as_uint(v.x);
  d.y = __float_as_uint(v.y);
  d.z = __float_as_uint(v.z);
  d.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, voff, 0, 0);
}

// Activation access of the plane kernels by element size ES (common.h): byte offsets are already scaled by ES; a thread's unit
// is 4 consecutive channels = 16 bytes of fp32 or 8 bytes of bf16 (widened exactly / rounded to nearest-even).
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
template <int ES>
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t rsrc, int voff) {
  if constexpr (ES == 4) {
    return buf_load16(rsrc, voff, 0);
  } else {
    const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0);
    return bdv_widen_bf16x4(v.x, v.y);
  }
}
template <int ES>
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t rsrc, int voff, float4 v) {
  if constexpr (ES == 4) {
    buf_store16(rsrc, voff, v);
  } else {
    __builtin_amdgcn_raw_buffer_store_b64((u32x2v){bdv_pack_bf16x2(v.x, v.y), bdv_pack_bf16x2(v.z, v.w)}, rsrc, voff, 0, 0);
  }
}

// q = m / d, r = m % d for 0 <= m < 2^22 with a float reciprocal and one correction step each way
__device__ __forceinline__ void fast_divmod(int m, int d, float rcp, int& q, int& r) {
  q = (int)((float)m * rcp);
  r = m - q * d;
  if (r < 0) {
    r += d;
    --q;
  }
  if (r >= d) {
    r -= d;
    ++q;
  }
}

// The TM (TN) MFMA tiles of a wave are MS (NS) = 32 * waves-per-dimension rows apart ("interleaved" wave tiling:
// wave w owns rows 32w..32w+31 of every 32*WM-row group).  With an unpadded 128-float LDS row a tile pair is then 64
// dwords apart and a K pair 256 dwords, which is what lets the compiler address all 32 operand reads of a stage as
// ds_read2st64_b32 with immediate offsets from one base register instead of one v_add per read.
// SWA / SWB: the operand image was written by store_transposed (K-contiguous source): unpadded 128-float rows with
// the column XOR-ed by 4 * ((k >> 2) & 7), which spreads the transposing 4-byte stores over all banks without a row
// pad.  The reader needs 8 base addresses per operand (one per swizzle value, registers) and again only immediates.
template <int LDA, int LDB, int TM, int TN, int MS, int NS, bool SWA, bool SWB>
__device__ __forceinline__ void mma_stage(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[TM][TN], int wm0, int wn0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* apv[SWA ? 8 : 1];
  const float* bpv[SWB ? 8 : 1];
#pragma unroll
  for (int c = 0; c < (SWA ? 8 : 1); ++c) apv[c] = As + h * LDA + wm0 + (r ^ (4 * c));
#pragma unroll
  for (int c = 0; c < (SWB ? 8 : 1); ++c) bpv[c] = Bs + h * LDB + wn0 + (r ^ (4 * c));
  // element [k = 2s + h][tile i]: swizzle value (k >> 2) & 7 = (s >> 1) & 7
  auto ap = [&](int s, int i) { return apv[SWA ? ((s >> 1) & 7) : 0][2 * s * LDA + MS * i]; };
  auto bp = [&](int s, int j) { return bpv[SWB ? ((s >> 1) & 7) : 0][2 * s * LDB + NS * j]; };
  // operand fragments are fetched one K-pair ahead of the MFMAs that consume them
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
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    // pin the order: next step's LDS reads first, then this step's MFMAs (hipcc otherwise sinks the reads
    // to just before their use and exposes the LDS latency every K-pair)
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
  }
}

// temporal-shift class of input channel c: +1 -> read frame t+1, -1 -> read frame t-1, 0 -> copy
__device__ __forceinline__ int shift_class(int c, int fold) {
  return (fold > 0) ? (c < fold ? 1 : (c < 2 * fold ? -1 : 0)) : 0;
}

// K-contiguous source tile (ROWS x 32 k, thread = (row tid/8, 16-byte group kg = tid%8)) -> transposing store into the
// k-major image with unpadded rows of LD floats; column ^ (4 * kg) (kg = k >> 2) makes the 32 lanes of a store
// (4 rows x 8 kg) hit 32 different banks.
template <int LD, int PASSES>
__device__ __forceinline__ void store_transposed(float* __restrict__ dst, const float4 (&v)[PASSES], int tid) {
  const int row = tid >> 3, kg = tid & 7;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    float* d = dst + (4 * kg) * LD + ((row + 32 * p) ^ (4 * kg));
    d[0] = v[p].x;
    d[LD] = v[p].y;
    d[2 * LD] = v[p].z;
    d[3 * LD] = v[p].w;
  }
}

// M-contiguous source tile (32 k rows x COLS) -> direct 16-byte stores
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
// order by a fix-up kernel).  The host planner picks (dp_tiles, split) so that the last, partial round of
// co-resident blocks consists of many short blocks (measured: 784 tiles -> 82 TFLOP/s, 768 -> 113).
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

template <int TM, int TN, int NT = 256>
__device__ __forceinline__ void store_partial(float* __restrict__ slab, int pslot, const f32x16 (&acc)[TM][TN], int tid) {
  float* base = slab + (size_t)pslot * (TM * TN * 16) * NT + tid;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) base[(size_t)((i * TN + j) * 16 + e) * NT] = acc[i][j][e];
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

__device__ __forceinline__ void fprop_store(float* __restrict__ y, const Geom& g, int row, int col, float v) {
  if (row < g.M) y[(size_t)row * g.Cout + col] = v;
}

// BatchNorm-backward statistics of the tensor a dgrad produces, taken in its epilogue (bdv_bn_stat_fuse): the dx a
// thread stores is the gradient w.r.t. the BN(+ReLU) output of the previous conv unit, so  sum(g)  and  sum(g * xhat)
// (g = dx * mask, xhat = (y - mean) * invstd) are accumulated per thread over the pieces it stores (a thread keeps the
// same 4 columns in every pass of the staged epilogue; with a temporal shift a piece lands in another frame, and y /
// mask are read at that destination: every dx element is still written exactly once), reduced over the tile in fixed
// order and written to partial[0 | 1][mt][Cin]; the separate statistics pass over dx, y and the mask disappears.
struct BnStat {
  const void* y;         // conv output of the unit whose BatchNorm-backward statistics are taken (element type = the kernel's ES)
  const uint32_t* mask;  // may be null (no ReLU, or the sign is derived from y: rscale)
  const float* mean;
  const float* invstd;
  float* partial;
  int MT;
  const float* rscale;   // optional: ReLU sign = y * rscale + rshift > 0 (units whose mask was never written)
  const float* rshift;
};

struct StatAcc {
  float4 s1, s2, mu, is;
  __device__ __forceinline__ void init(const BnStat& st, int col) {
    s1 = s2 = make_float4(0.f, 0.f, 0.f, 0.f);
    mu = *reinterpret_cast<const float4*>(st.mean + col);
    is = *reinterpret_cast<const float4*>(st.invstd + col);
  }
  // ReLU sign bits of 4 channels from the conv output itself (the forward's fused multiply-add); scale / shift are fetched where
  // they are used (L1-resident) rather than held in 8 more registers through the epilogue
  __device__ __forceinline__ static unsigned sign_bits(const BnStat& st, int col, const float4 yv) {
    const float4 rsc = *reinterpret_cast<const float4*>(st.rscale + col), rsh = *reinterpret_cast<const float4*>(st.rshift + col);
    return (fmaf(yv.x, rsc.x, rsh.x) > 0.f ? 1u : 0u) | (fmaf(yv.y, rsc.y, rsh.y) > 0.f ? 2u : 0u) |
           (fmaf(yv.z, rsc.z, rsh.z) > 0.f ? 4u : 0u) | (fmaf(yv.w, rsc.w, rsh.w) > 0.f ? 8u : 0u);
  }
  __device__ __forceinline__ void accumulate(const float4 v, const float4 yv) {  // v: masked gradient, yv: conv output
    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
    s2.x += v.x * ((yv.x - mu.x) * is.x);
    s2.y += v.y * ((yv.y - mu.y) * is.y);
    s2.z += v.z * ((yv.z - mu.z) * is.z);
    s2.w += v.w * ((yv.w - mu.w) * is.w);
  }
};

// tile reduction of the per-thread sums: thread = (row group tid / V, column vector tid % V), V = BN / 4
template <int BN, int NT = 256>
__device__ __forceinline__ void stat_flush(const StatAcc& a, const BnStat& st, float* __restrict__ smem, int Cin, int mt, int nt,
                                           int tid) {
  constexpr int V = BN / 4, G = NT / V;
  __syncthreads();  // every thread is done reading the staged tile
  const int grp = tid / V, c4 = tid - grp * V;
  *reinterpret_cast<float4*>(smem + grp * BN + 4 * c4) = a.s1;
  *reinterpret_cast<float4*>(smem + (G + grp) * BN + 4 * c4) = a.s2;
  __syncthreads();
  if (tid < BN) {
    float x = 0.f, y = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) {
      x += smem[k * BN + tid];
      y += smem[(G + k) * BN + tid];
    }
    st.partial[(size_t)mt * Cin + nt * BN + tid] = x;
    st.partial[((size_t)st.MT + mt) * Cin + nt * BN + tid] = y;
  }
}

// Epilogue staging: the accumulator layout (one column per lane, 16 rows in registers) would store 4 bytes per
// lane; going through the (now free) LDS stage turns the tile into row-major order so that every thread handles
// 16-byte pieces of a row: 4x fewer, 4x wider global instructions (the 1x1 convs with wide outputs are bound by
// this traffic, not by MFMA).  Two passes of WM*32 rows; emit(row_in_tile, col_in_tile, float4).
template <int BM, int BN, int WM, int WN, class F>
__device__ __forceinline__ void staged_epilogue(const f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* __restrict__ smem, int tid,
                                                F&& emit) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NT = 64 * WM * WN;
  constexpr int V = BN / 4;                     // float4 per row
  constexpr int PER = WM * 32 * V / NT;         // float4 per thread per pass
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn0 = (wave % WN) * 32;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();  // LDS is free: K loop / previous pass finished
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) smem[(wm * 32 + acc_row(e, lane)) * BN + wn0 + 32 * WN * j + (lane & 31)] = acc[i][j][e];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = tid + NT * q;
      const int lr = idx / V, c4 = idx - lr * V;
      const float4 v = *reinterpret_cast<const float4*>(smem + lr * BN + 4 * c4);
      emit(32 * WM * i + lr, 4 * c4, v);  // staged row lr = 32 * wave_m + r  ->  tile row 32*WM*i + lr
    }
  }
}

// dgrad epilogue: like staged_epilogue, but the per-piece work (residual / mask loads, statistics loads) is batched and
// branch-free.  Each thread first computes the destination offsets of its PER pieces of a pass and issues all their
// loads as buffer loads (pieces past the last row get an out-of-range offset: loads return zeros, stores are dropped),
// then combines and stores.  With one `if (row < M)` branch per piece the compiler had serialised the pieces, i.e.
// 16 dependent global-load latencies per tile: the conv1 sites (K loop of 2-16 steps, residual add) ran at half their
// HBM roofline.
// rowmap(mrow) -> dx pixel index of tile row mrow (identity for stride 1; parity-class map for stride 2).

// DERIVE: the ReLU sign of the statistics may come from the conv output itself (BnStat::rscale); compiled out of the fp32-MFMA
// kernels, whose 168-register budget has no room for it (the model only derives signs in the bf16-piece arithmetic).
// PIPE: two register sets, the loads of batch b + 1 issued before batch b is combined and stored (see below); only for kernels
// whose register budget has room beside the accumulators (64 accumulator registers, <= 2 waves per SIMD).
template <int BM, int BN, int WM, int WN, bool DERIVE = true, int ES = 4, bool PIPE = false, class RowMap>
__device__ __forceinline__ void dgrad_epilogue(const f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* __restrict__ smem, int tid,
                                               void* __restrict__ dx, const void* __restrict__ add_src,
                                               const uint32_t* __restrict__ add_mask, const Geom& g, int mt, int nt, int Mrows,
                                               const BnStat& stat, RowMap&& rowmap) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NT = 64 * WM * WN;
  constexpr int V = BN / 4;
  constexpr int PER = WM * 32 * V / NT;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn0 = (wave % WN) * 32;
  const int HW = g.H * g.W;
  const float rcp_HW = 1.0f / (float)HW;
  const int total_bytes = g.N * HW * g.Cin * ES;
  const int mask_bytes = g.N * HW * (g.Cin / 8);
  const __amdgpu_buffer_rsrc_t dxr = __builtin_amdgcn_make_buffer_rsrc((void*)dx, 0, total_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t adr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(add_src ? add_src : dx), 0, add_src ? total_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t amr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(add_mask ? (const void*)add_mask : (const void*)dx), 0, add_mask ? mask_bytes : 0, 0x00020000);
  const bool do_stat = stat.y != nullptr;
  const __amdgpu_buffer_rsrc_t syr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(do_stat ? stat.y : dx), 0, do_stat ? total_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t smr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(do_stat && stat.mask ? (const void*)stat.mask : (const void*)dx), 0, do_stat && stat.mask ? mask_bytes : 0, 0x00020000);
  // the thread's 4 columns are the same for every piece (256 % V == 0)
  const int col = nt * BN + 4 * (tid % V);
  const int cls = shift_class(col, g.fold);
  StatAcc sa;
  if (do_stat) sa.init(stat, col);
  // The pieces of a tile are handled in batches of HB (register budget: 2 float4 + 3 words per piece and set); a batch's global
  // loads (residual, its mask, the statistics operand and its mask) do not depend on the staged tile, so the loads of batch b + 1
  // are issued BEFORE batch b is combined and stored (two register sets): the tile pays one load latency instead of one per batch.
  // Round 2 issued load -> combine -> store batch by batch; the buffer stores of a batch and the loads of the next one may alias
  // (in-place add_src), so the compiler could not hoist them itself: in-kernel stamps put the epilogue of the conv1 sites at
  // 12 - 17 us beside a K loop of 18 us (tools/stamp_tiles.py).  No hazard is created: a batch reads and writes only its own
  // pieces, and every dx element is written exactly once.
  constexpr int HB = PER < 4 ? PER : 4;  // pieces in flight per thread and register set
  constexpr int NBATCH = PER / HB;       // batches per pass
  static_assert(PER % HB == 0, "pieces per pass must be whole batches");
  constexpr int NSETS = PIPE ? 2 : 1;
  float4 a[NSETS][HB], yv[NSETS][HB];
  uint32_t am[NSETS][HB], sm[NSETS][HB];
  int off[NSETS][HB];
  bool zero[NSETS][HB];
  auto issue = [&](int set, int i, int q0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      const int lr = (tid + NT * (q0 + u)) / V;
      const int mrow = mt * BM + 32 * WM * i + lr;
      const bool ok = mrow < Mrows;
      const int row = rowmap(ok ? mrow : 0);
      int drow = row;
      bool z = false;
      if (cls != 0) {  // temporal un-shift: frame n -> n + cls inside the clip, else the zero the far clip end needs
        int n, rem;
        fast_divmod(row, HW, rcp_HW, n, rem);
        const int t = n % g.T;
        const bool inside = (unsigned)(t + cls) < (unsigned)g.T;
        drow = inside ? row + cls * HW : row - cls * (g.T - 1) * HW;
        z = !inside;
      }
      const int o = drow * g.Cin + col;
      off[set][u] = ok ? o * ES : kOOB;
      zero[set][u] = z;
      a[set][u] = buf_ld4<ES>(adr, off[set][u]);  // zeros when there is no add_src
      am[set][u] = add_mask ? __builtin_amdgcn_raw_buffer_load_b32(amr, ok ? (o >> 5) << 2 : kOOB, 0, 0) : 0xffffffffu;
      if (do_stat) {
        yv[set][u] = buf_ld4<ES>(syr, off[set][u]);
        sm[set][u] = stat.mask ? __builtin_amdgcn_raw_buffer_load_b32(smr, ok ? (o >> 5) << 2 : kOOB, 0, 0) : 0xffffffffu;
      }
    }
  };
  auto finish = [&](int set, int q0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      const int lr = (tid + NT * (q0 + u)) / V;
      const float4 t4 = *reinterpret_cast<const float4*>(smem + lr * BN + 4 * (tid % V));
      const int sh = (off[set][u] / ES) & 31;  // bit position of the piece's first channel in its mask word
      const unsigned nib = (am[set][u] >> sh) & 0xFu;
      float4 r = zero[set][u] ? make_float4(0.f, 0.f, 0.f, 0.f) : t4;
      r.x += (nib & 1u) ? a[set][u].x : 0.f;
      r.y += (nib & 2u) ? a[set][u].y : 0.f;
      r.z += (nib & 4u) ? a[set][u].z : 0.f;
      r.w += (nib & 8u) ? a[set][u].w : 0.f;
      buf_st4<ES>(dxr, off[set][u], r);
      if (do_stat && off[set][u] != kOOB) {
        const unsigned sn = (DERIVE && stat.rscale != nullptr) ? StatAcc::sign_bits(stat, col, yv[set][u]) : (sm[set][u] >> sh) & 0xFu;
        float4 gq;
        gq.x = (sn & 1u) ? r.x : 0.f;
        gq.y = (sn & 2u) ? r.y : 0.f;
        gq.z = (sn & 4u) ? r.z : 0.f;
        gq.w = (sn & 8u) ? r.w : 0.f;
        sa.accumulate(gq, yv[set][u]);
      }
    }
  };
  if constexpr (PIPE) issue(0, 0, 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();  // LDS is free: K loop / previous pass finished
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) smem[(wm * 32 + acc_row(e, lane)) * BN + wn0 + 32 * WN * j + (lane & 31)] = acc[i][j][e];
    __syncthreads();
#pragma unroll
    for (int bq = 0; bq < NBATCH; ++bq) {
      if constexpr (PIPE) {
        const int b = i * NBATCH + bq;                     // compile-time after unrolling
        if (b + 1 < TM * NBATCH) issue((b + 1) & 1, (b + 1) / NBATCH, ((b + 1) % NBATCH) * HB);
        finish(b & 1, bq * HB);
      } else {                                             // round-2 order: load, combine, store, batch by batch
        issue(0, i, bq * HB);
        finish(0, bq * HB);
      }
    }
  }
  // stride 2: one block of stat.MT / 4 partial rows per parity class (blockIdx.y); stride 1: gridDim.y == 1
  if (do_stat) stat_flush<BN, NT>(sa, stat, smem, g.Cin, blockIdx.y * (stat.MT >> (gridDim.y == 4 ? 2 : 0)) + mt, nt, tid);
}

// BatchNorm batch statistics fused into the fprop epilogue: per tile, the column sums of y and y^2 over the
// tile's valid rows -> bn_partial[0][mt][co], bn_partial[1][mt][co]; a fixed-order fp64 finalize sums the MT rows.
// cs/cq hold this lane's sums over its accumulator registers for each of its TN columns.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void tile_colstats(float* __restrict__ smem, float (&cs)[BN / WN / 32], float (&cq)[BN / WN / 32],
                                              float* __restrict__ bn_partial, int MT, int Cout, int mt, int nt, int tid) {
  constexpr int TN = BN / WN / 32;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn0 = (wave % WN) * 32;
#pragma unroll
  for (int j = 0; j < TN; ++j) {  // lanes l and l+32 hold the same column, different rows
    cs[j] += __shfl_xor(cs[j], 32, 64);
    cq[j] += __shfl_xor(cq[j], 32, 64);
  }
  __syncthreads();  // every wave is done with the operand stage in LDS
  if (lane < 32) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      smem[wm * BN + wn0 + 32 * WN * j + lane] = cs[j];
      smem[(WM + wm) * BN + wn0 + 32 * WN * j + lane] = cq[j];
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

// What the fprop epilogue does besides storing y: BatchNorm batch statistics (training), or -- eval-mode BatchNorm
// folded into the conv -- out = relu?(acc * scale[c] + shift[c] (+ residual)), which removes the separate
// bn_apply pass of every inference / frozen-teacher forward.
struct FpropEpi {
  float* bn_partial;  // [2][MT][Cout] or null
  int MT;
  const float* scale;  // null: plain store
  const float* shift;
  const void* res;     // optional residual, same shape and element type as y
  int relu;
  // BatchNorm + ReLU of the PRODUCER applied to the activation operand in the loader (conv_fprop_pl_kernel<..., PRE>):
  // a = max(x * pre_scale[ci] + pre_shift[ci], 0)
  const float* pre_scale;
  const float* pre_shift;
};

template <int BM, int BN, int WM, int WN, int ES = 4>
__device__ __forceinline__ void fprop_affine_epilogue(const f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* __restrict__ smem,
                                                      void* __restrict__ y, const Geom& g, const FpropEpi& epi, int mt, int nt,
                                                      int tid) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NT = 64 * WM * WN;
  constexpr int V = BN / 4;
  constexpr int PER = WM * 32 * V / NT;
  constexpr int HB = PER < 4 ? PER : 4;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn0 = (wave % WN) * 32;
  const int total_bytes = g.M * g.Cout * ES;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, total_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(epi.res ? epi.res : (const void*)y), 0, epi.res ? total_bytes : 0, 0x00020000);
  const int col = nt * BN + 4 * (tid % V);  // the thread's 4 columns are the same for every piece (256 % V == 0)
  const float4 sc = *reinterpret_cast<const float4*>(epi.scale + col);
  const float4 sh = *reinterpret_cast<const float4*>(epi.shift + col);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) smem[(wm * 32 + acc_row(e, lane)) * BN + wn0 + 32 * WN * j + (lane & 31)] = acc[i][j][e];
    __syncthreads();
#pragma unroll
    for (int q0 = 0; q0 < PER; q0 += HB) {  // residual loads of HB pieces in flight (see dgrad_epilogue)
      float4 v[HB], r[HB];
      int off[HB];
#pragma unroll
      for (int u = 0; u < HB; ++u) {
        const int lr = (tid + NT * (q0 + u)) / V;
        const int row = mt * BM + 32 * WM * i + lr;
        off[u] = row < g.M ? (row * g.Cout + col) * ES : kOOB;
        v[u] = *reinterpret_cast<const float4*>(smem + lr * BN + 4 * (tid % V));
        r[u] = buf_ld4<ES>(rr, off[u]);  // zeros without a residual
      }
#pragma unroll
      for (int u = 0; u < HB; ++u) {
        float4 o;
        o.x = v[u].x * sc.x + sh.x + r[u].x;
        o.y = v[u].y * sc.y + sh.y + r[u].y;
        o.z = v[u].z * sc.z + sh.z + r[u].z;
        o.w = v[u].w * sc.w + sh.w + r[u].w;
        if (epi.relu) {
          o.x = fmaxf(o.x, 0.f);
          o.y = fmaxf(o.y, 0.f);
          o.z = fmaxf(o.z, 0.f);
          o.w = fmaxf(o.w, 0.f);
        }
        buf_st4<ES>(yr, off[u], o);
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int ES = 4>
__device__ __forceinline__ void fprop_epilogue(const f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* __restrict__ smem,
                                               void* __restrict__ y, const Geom& g, const FpropEpi& epi, int mt, int nt, int tid) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32;
  if (epi.scale != nullptr) {
    fprop_affine_epilogue<BM, BN, WM, WN, ES>(acc, smem, y, g, epi, mt, nt, tid);
    return;
  }
  float* const bn_partial = epi.bn_partial;
  const int MT = epi.MT;
  staged_epilogue<BM, BN, WM, WN>(acc, smem, tid, [&](int tr, int tc, float4 v) {
    const int row = mt * BM + tr;
    if (row < g.M) act_st4<ES>(y, ((size_t)row * g.Cout + nt * BN + tc) >> 2, v);
  });
  if (bn_partial != nullptr) {
    float cs[TN], cq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = mt * BM + wm0 + 32 * WM * i + acc_row(e, lane);
          const float v = row < g.M ? acc[i][j][e] : 0.f;
          cs[j] += v;
          cq[j] += v * v;
        }
    tile_colstats<BM, BN, WM, WN>(smem, cs, cq, bn_partial, MT, g.Cout, mt, nt, tid);
  }
}

// =========================================================================================
// fprop: y[m, co] = sum_{tap, ci} x_shift[n, ho*st + r - p, wo*st + s - p, ci] * w[co, tap, ci]
// K order is tap-fastest: the R*S taps of one 32-channel chunk are consecutive K-steps, so the shifted re-reads of
// the same activation rows hit L1/L2 instead of going back to the fabric R*S times.
// =========================================================================================
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 3) void conv_fprop_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, Geom g, int NT, Work wk,
                                                          float* __restrict__ slab, FpropEpi epi) {
  constexpr int LDA = BM, LDB = BN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  __shared__ __attribute__((aligned(16))) float smem[BK * (LDA + LDB)];
  float* const As = smem;
  float* const Bs = smem + BK * LDA;

  // Tile order: Cout-tile fastest, so the blocks that share an activation row-tile are consecutive and
  // (through xcd_remap) land on one XCD / one L2.
  const int nk = g.Ktot / BK;
  const WorkItem it = get_work(blockIdx.x, wk, nk);
  const int mt = it.tile / NT, nt = it.tile - mt * NT;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;  // tile i / j of the wave: + 32*WM*i / + 32*WN*j
  const int arow = tid >> 3, kg = tid & 7;
  const int HoWo = g.Ho * g.Wo;
  const int frame_bytes = g.H * g.W * g.Cin * 4;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.N * frame_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, g.Cout * g.Ktot * 4, 0x00020000);

  int a_base[AP], a_t[AP], a_hi0[AP], a_wi0[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + 32 * p;
    const bool ok = m < g.M;
    int n, rem, ho, wo;
    fast_divmod(ok ? m : 0, HoWo, g.rcp_HoWo, n, rem);
    fast_divmod(rem, g.Wo, g.rcp_Wo, ho, wo);
    a_t[p] = n % g.T;
    a_hi0[p] = ok ? ho * g.stride - g.pad : -(1 << 20);  // rows past M fail every bounds test
    a_wi0[p] = wo * g.stride - g.pad_w;
    a_base[p] = ((n * g.H + ho * g.stride - g.pad) * g.W + wo * g.stride - g.pad_w) * g.Cin * 4 + 16 * kg;
  }
  int b_base[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) b_base[p] = ((nt * BN + arow + 32 * p) * g.Ktot + 4 * kg) * 4;

  // K index state (uniform): K-step kt = chunk * R*S + r * S + s
  const int RS = g.R * g.S;
  int chunk = it.kb / RS, r, s;
  {
    const int tap = it.kb - chunk * RS;
    r = tap / g.S;
    s = tap - r * g.S;
  }

  float4 ra[AP], rb[BP];
  auto load = [&]() {
    const int cls = shift_class(chunk * BK + 4 * kg, g.fold);
    const int koff_a = ((r * g.W + s) * g.Cin + chunk * BK) * 4;
    const int koff_b = ((r * g.S + s) * g.Cin + chunk * BK) * 4;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool v = (unsigned)(a_hi0[p] + r) < (unsigned)g.H && (unsigned)(a_wi0[p] + s) < (unsigned)g.W &&
                     (unsigned)(a_t[p] + cls) < (unsigned)g.T;
      // invalid lanes: set the top bit -> beyond num_records -> the load returns zeros
      ra[p] = buf_load16(xr, (a_base[p] + koff_a + cls * frame_bytes) | (v ? 0 : kOOB), 0);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) rb[p] = buf_load16(wr, b_base[p], koff_b);
    // advance to the next K-step (branch-free)
    s += 1;
    const int ws_ = (s == g.S) ? 1 : 0;
    s = ws_ ? 0 : s;
    r += ws_;
    const int wr_ = (r == g.R) ? 1 : 0;
    r = wr_ ? 0 : r;
    chunk += wr_;
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  load();
  for (int kt = it.kb; kt < it.ke; ++kt) {
    __syncthreads();  // everyone is done reading the previous stage
    store_transposed<LDA, AP>(As, ra, tid);
    store_transposed<LDB, BP>(Bs, rb, tid);
    __syncthreads();
    if (kt + 1 < it.ke) load();  // in flight during the MFMAs
    mma_stage<LDA, LDB, TM, TN, 32 * WM, 32 * WN, true, true>(As, Bs, acc, wm0, wn0, lane);
  }

  if (it.pslot >= 0) {
    store_partial<TM, TN>(slab, it.pslot, acc, tid);
    return;
  }
  fprop_epilogue<BM, BN, WM, WN>(acc, smem, y, g, epi, mt, nt, tid);
}

// ---- experimental: fp32 products from bf16 pieces -------------------------------------------------
// Same implicit GEMM, loader, work planner and epilogue as conv_fprop_kernel; only the K loop differs: every fp32 operand
// value is split in registers into three round-to-nearest bf16 pieces (hi + mid + lo = 24 significand bits), the pieces go
// to LDS as k-contiguous rows, and each 16-deep step accumulates the six piece products of relative weight >= 2^-16 with
// v_mfma_f32_32x32x16_bf16 (smallest first).  The dropped products are below 2^-24 relative, the order of fp32 rounding.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int X3_LDK = BK + 8;  // bf16 per LDS row: 80 bytes, the 16-byte fragment reads of 16 lanes cover all banks once

// fp32 x 4 -> three bf16 pieces each, as packed pairs (x, y) and (z, w).  Written on pairs so that one v_cvt_pk_bf16_f32
// converts two values, the leading piece is turned back into fp32 with a shift / a mask, and the remainders are taken with
// v_pk_add_f32: 20 VALU per float4, where the element-wise __builtin_convertvector form compiles to 32.  The values are
// the same bit for bit (round-to-nearest-even conversions, exact subtractions).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
__device__ __forceinline__ u32x2_t bf16_hi_pairs(const float4 v) { return (u32x2_t){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}; }
__device__ __forceinline__ void split3_pairs(const float4 v, u32x2_t& hi, u32x2_t& mid, u32x2_t& lo) {
  const unsigned h0 = pack_bf16x2(v.x, v.y), h1 = pack_bf16x2(v.z, v.w);
  const float rx = v.x - __uint_as_float(h0 << 16), ry = v.y - __uint_as_float(h0 & 0xffff0000u);
  const float rz = v.z - __uint_as_float(h1 << 16), rw = v.w - __uint_as_float(h1 & 0xffff0000u);
  const unsigned m0 = pack_bf16x2(rx, ry), m1 = pack_bf16x2(rz, rw);
  const float sx = rx - __uint_as_float(m0 << 16), sy = ry - __uint_as_float(m0 & 0xffff0000u);
  const float sz = rz - __uint_as_float(m1 << 16), sw = rw - __uint_as_float(m1 & 0xffff0000u);
  hi = (u32x2_t){h0, h1};
  mid = (u32x2_t){m0, m1};
  lo = (u32x2_t){pack_bf16x2(sx, sy), pack_bf16x2(sz, sw)};
}

// thread (row = tid / 8 + 32 p, k = 4 (tid % 8) .. + 3) -> planes[0 | 1 | 2][row][k]
template <int ROWS, int P>
__device__ __forceinline__ void store_split3(unsigned short* __restrict__ dst, const float4 (&v)[P], int tid) {
  const int arow = tid >> 3, kg = tid & 7;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    u32x2_t hi, mid, lo;
    split3_pairs(v[p], hi, mid, lo);
    unsigned short* q = dst + (arow + 32 * p) * X3_LDK + 4 * kg;
    *reinterpret_cast<u32x2_t*>(q) = hi;
    *reinterpret_cast<u32x2_t*>(q + ROWS * X3_LDK) = mid;
    *reinterpret_cast<u32x2_t*>(q + 2 * ROWS * X3_LDK) = lo;
  }
}

// lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8 h + 0..7] and B[k][col r] of a 16-deep step
template <int TM, int TN, int MS, int NS, int ROWS_A, int ROWS_B>
__device__ __forceinline__ void mma_stage_x3(const unsigned short* __restrict__ As, const unsigned short* __restrict__ Bs,
                                             f32x16 (&acc)[TM][TN], int wm0, int wn0, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < BK / 16; ++s) {
    bf16x8_t a[3][TM], b[3][TN];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[p][i] = *reinterpret_cast<const bf16x8_t*>(As + (p * ROWS_A + wm0 + MS * i + r) * X3_LDK + 16 * s + 8 * h);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[p][j] = *reinterpret_cast<const bf16x8_t*>(Bs + (p * ROWS_B + wn0 + NS * j + r) * X3_LDK + 16 * s + 8 * h);
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
}

// BPRE: w points at the F planes of bdv_conv_split_weights instead of the fp32 weights: the weight tile of a K-step is then
// three contiguous blocks of 64-byte rows, loaded with 16-byte loads and stored as they are (no split VALU for that operand).
template <int BM, int BN, int WM, int WN, bool BPRE = false>
__global__ __launch_bounds__(256, 2) void conv_fprop_x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, Geom g, int NT, Work wk,
                                                          float* __restrict__ slab, FpropEpi epi) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  static_assert(3 * (BM + BN) * X3_LDK * 2 >= WM * 32 * BN * 4, "the epilogue stages a tile pass in the same LDS");
  __shared__ __attribute__((aligned(16))) unsigned short smem16[3 * (BM + BN) * X3_LDK];
  unsigned short* const As = smem16;
  unsigned short* const Bs = smem16 + 3 * BM * X3_LDK;
  float* const smem = reinterpret_cast<float*>(smem16);

  // Tile order: Cout-tile fastest, so the blocks that share an activation row-tile are consecutive and
  // (through xcd_remap) land on one XCD / one L2.
  const int nk = g.Ktot / BK;
  const WorkItem it = get_work(blockIdx.x, wk, nk);
  const int mt = it.tile / NT, nt = it.tile - mt * NT;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;  // tile i / j of the wave: + 32*WM*i / + 32*WN*j
  const int arow = tid >> 3, kg = tid & 7;
  const int HoWo = g.Ho * g.Wo;
  const int frame_bytes = g.H * g.W * g.Cin * 4;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.N * frame_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, BPRE ? 3 * g.Cout * g.Ktot * 2 : g.Cout * g.Ktot * 4, 0x00020000);
  constexpr int BQ = BN * 4 / 256;                    // BPRE: 16-byte loads per plane and thread
  const int brow = tid >> 2, bc = tid & 3;
  const int plane_bytes = g.Cout * g.Ktot * 2;
  int kt_w = it.kb;

  int a_base[AP], a_t[AP], a_hi0[AP], a_wi0[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + 32 * p;
    const bool ok = m < g.M;
    int n, rem, ho, wo;
    fast_divmod(ok ? m : 0, HoWo, g.rcp_HoWo, n, rem);
    fast_divmod(rem, g.Wo, g.rcp_Wo, ho, wo);
    a_t[p] = n % g.T;
    a_hi0[p] = ok ? ho * g.stride - g.pad : -(1 << 20);  // rows past M fail every bounds test
    a_wi0[p] = wo * g.stride - g.pad_w;
    a_base[p] = ((n * g.H + ho * g.stride - g.pad) * g.W + wo * g.stride - g.pad_w) * g.Cin * 4 + 16 * kg;
  }
  int b_base[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) b_base[p] = ((nt * BN + arow + 32 * p) * g.Ktot + 4 * kg) * 4;

  // K index state (uniform): K-step kt = chunk * R*S + r * S + s
  const int RS = g.R * g.S;
  int chunk = it.kb / RS, r, s;
  {
    const int tap = it.kb - chunk * RS;
    r = tap / g.S;
    s = tap - r * g.S;
  }

  float4 ra[AP], rb[BP];
  u32x4 rbp[BPRE ? 3 * BQ : 1];
  auto load = [&]() {
    const int cls = shift_class(chunk * BK + 4 * kg, g.fold);
    const int koff_a = ((r * g.W + s) * g.Cin + chunk * BK) * 4;
    const int koff_b = ((r * g.S + s) * g.Cin + chunk * BK) * 4;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool v = (unsigned)(a_hi0[p] + r) < (unsigned)g.H && (unsigned)(a_wi0[p] + s) < (unsigned)g.W &&
                     (unsigned)(a_t[p] + cls) < (unsigned)g.T;
      // invalid lanes: set the top bit -> beyond num_records -> the load returns zeros
      ra[p] = buf_load16(xr, (a_base[p] + koff_a + cls * frame_bytes) | (v ? 0 : kOOB), 0);
    }
    if (BPRE) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int q = 0; q < BQ; ++q)
          rbp[pl * BQ + q] = __builtin_amdgcn_raw_buffer_load_b128(wr, (nt * BN + brow + 64 * q) * 64 + 16 * bc + pl * plane_bytes, kt_w * g.Cout * 64, 0);
      kt_w += 1;
    } else {
#pragma unroll
      for (int p = 0; p < BP; ++p) rb[p] = buf_load16(wr, b_base[p], koff_b);
    }
    // advance to the next K-step (branch-free)
    s += 1;
    const int ws_ = (s == g.S) ? 1 : 0;
    s = ws_ ? 0 : s;
    r += ws_;
    const int wr_ = (r == g.R) ? 1 : 0;
    r = wr_ ? 0 : r;
    chunk += wr_;
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  load();
  for (int kt = it.kb; kt < it.ke; ++kt) {
    __syncthreads();  // everyone is done reading the previous stage
    store_split3<BM, AP>(As, ra, tid);
    if (BPRE) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int q = 0; q < BQ; ++q) *reinterpret_cast<u32x4*>(Bs + (pl * BN + brow + 64 * q) * X3_LDK + 8 * bc) = rbp[pl * BQ + q];
    } else {
      store_split3<BN, BP>(Bs, rb, tid);
    }
    __syncthreads();
    if (kt + 1 < it.ke) load();  // in flight during the MFMAs
    mma_stage_x3<TM, TN, 32 * WM, 32 * WN, BM, BN>(As, Bs, acc, wm0, wn0, lane);
  }

  if (it.pslot >= 0) {
    store_partial<TM, TN>(slab, it.pslot, acc, tid);
    return;
  }
  fprop_epilogue<BM, BN, WM, WN>(acc, smem, y, g, epi, mt, nt, tid);
}

// wgrad variant of the bf16-piece loop: both operands arrive contiguous along channels, not along the contraction (pixels).
// A thread loads rows k and k + 1 of its 4 channels, and the LDS image of a plane is [16 k-pairs][COLS] 32-bit words, a word
// holding the bf16 pieces of (k, k + 1) of one channel: written with 16-byte stores, and a lane collects its 8 consecutive k
// of one channel with four ds_read_b32 whose words already have the element order of the MFMA operand (32 lanes read 128
// contiguous bytes: conflict-free).  Four times the LDS read instructions of the row-major image of fprop / dgrad.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int X3_PAIR_ROWS = BK / 2;

// v[2q], v[2q+1] = rows (k, k+1) of k-pair  kp = tid / V + 8 q, channels 4 (tid % V) .. + 3
template <int COLS, int PASSES>
__device__ __forceinline__ void store_split3_pairs(unsigned* __restrict__ dst, const float4 (&v)[PASSES], int tid) {
  constexpr int V = COLS / 4, PLANE = X3_PAIR_ROWS * COLS;
  static_assert(PASSES == 4 && 256 / V == 8, "row assignment below: 8 row groups x 2 rows x 2 halves");
  const int c4 = tid % V, rg = tid / V;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const f32x4_t e = {v[2 * q].x, v[2 * q].y, v[2 * q].z, v[2 * q].w};
    const f32x4_t o = {v[2 * q + 1].x, v[2 * q + 1].y, v[2 * q + 1].z, v[2 * q + 1].w};
    const bf16x4_t eh = __builtin_convertvector(e, bf16x4_t), oh = __builtin_convertvector(o, bf16x4_t);
    const f32x4_t e1 = e - __builtin_convertvector(eh, f32x4_t), o1 = o - __builtin_convertvector(oh, f32x4_t);
    const bf16x4_t em = __builtin_convertvector(e1, bf16x4_t), om = __builtin_convertvector(o1, bf16x4_t);
    const f32x4_t e2 = e1 - __builtin_convertvector(em, f32x4_t), o2 = o1 - __builtin_convertvector(om, f32x4_t);
    const bf16x4_t el = __builtin_convertvector(e2, bf16x4_t), ol = __builtin_convertvector(o2, bf16x4_t);
    unsigned* w = dst + (rg + 8 * q) * COLS + 4 * c4;
    *reinterpret_cast<bf16x8_t*>(w) = __builtin_shufflevector(eh, oh, 0, 4, 1, 5, 2, 6, 3, 7);
    *reinterpret_cast<bf16x8_t*>(w + PLANE) = __builtin_shufflevector(em, om, 0, 4, 1, 5, 2, 6, 3, 7);
    *reinterpret_cast<bf16x8_t*>(w + 2 * PLANE) = __builtin_shufflevector(el, ol, 0, 4, 1, 5, 2, 6, 3, 7);
  }
}

template <int TM, int TN, int MS, int NS, int COLS_A, int COLS_B>
__device__ __forceinline__ void mma_stage_x3_pairs(const unsigned* __restrict__ As, const unsigned* __restrict__ Bs,
                                                   f32x16 (&acc)[TM][TN], int wm0, int wn0, int lane) {
  constexpr int PA = X3_PAIR_ROWS * COLS_A, PB = X3_PAIR_ROWS * COLS_B;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < BK / 16; ++s) {
    bf16x8_t a[3][TM], b[3][TN];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        u32x4_t f;
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = As[p * PA + (8 * s + 4 * h + e) * COLS_A + wm0 + MS * i + r];
        a[p][i] = __builtin_bit_cast(bf16x8_t, f);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        u32x4_t f;
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = Bs[p * PB + (8 * s + 4 * h + e) * COLS_B + wn0 + NS * j + r];
        b[p][j] = __builtin_bit_cast(bf16x8_t, f);
      }
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
}

// fprop for the stem (Cin = 4 on NHWC4): one filter tap per 16-byte load, K = R*S*4 padded to a multiple of 32
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 3) void conv_fprop_c4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, Geom g, int NT, FpropEpi epi) {
  constexpr int LDA = BM, LDB = BN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  __shared__ __attribute__((aligned(16))) float smem[BK * (LDA + LDB)];
  float* const As = smem;
  float* const Bs = smem + BK * LDA;
  const int tile = xcd_remap(blockIdx.x, epi.MT * NT);
  const int mt = tile / NT, nt = tile - mt * NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;  // tile i / j of the wave: + 32*WM*i / + 32*WN*j
  const int arow = tid >> 3, kg = tid & 7;
  const int HoWo = g.Ho * g.Wo;
  const int RS = g.R * g.S;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.N * g.H * g.W * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, g.Cout * g.Ktot * 4, 0x00020000);

  int a_base[AP], a_hi0[AP], a_wi0[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + 32 * p;
    const bool ok = m < g.M;
    const int mm = ok ? m : 0;  // up to 2^23 output pixels here: exact integer division
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    a_hi0[p] = ok ? ho * g.stride - g.pad : -(1 << 20);
    a_wi0[p] = wo * g.stride - g.pad_w;
    a_base[p] = ((n * g.H + ho * g.stride - g.pad) * g.W + wo * g.stride - g.pad_w) * 16;
  }
  float4 ra[AP], rb[BP];
  auto load = [&](int kt) {
    const int tap = kt * (BK / 4) + kg;  // this thread's filter tap
    const int r = tap / g.S, s = tap - r * g.S;
    const bool kv = tap < RS;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool v = kv && (unsigned)(a_hi0[p] + r) < (unsigned)g.H && (unsigned)(a_wi0[p] + s) < (unsigned)g.W;
      ra[p] = buf_load16(xr, (a_base[p] + (r * g.W + s) * 16) | (v ? 0 : kOOB), 0);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p)
      rb[p] = buf_load16(wr, (((nt * BN + arow + 32 * p) * g.Ktot + tap * 4) * 4) | (kv ? 0 : kOOB), 0);
  };
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  const int nk = (g.Ktot + BK - 1) / BK;
  load(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_transposed<LDA, AP>(As, ra, tid);
    store_transposed<LDB, BP>(Bs, rb, tid);
    __syncthreads();
    if (kt + 1 < nk) load(kt + 1);
    mma_stage<LDA, LDB, TM, TN, 32 * WM, 32 * WN, true, true>(As, Bs, acc, wm0, wn0, lane);
  }
  fprop_epilogue<BM, BN, WM, WN>(acc, smem, y, g, epi, mt, nt, tid);
}

// The stem in the bf16-piece arithmetic: loader of conv_fprop_c4_kernel, K loop of conv_fprop_x3_kernel (both operands are
// split in the loader; a thread's 16-byte load is one filter tap = 4 consecutive K-values, which is store_split3's mapping).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_fprop_c4_x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   float* __restrict__ y, Geom g, int NT, FpropEpi epi) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  static_assert(3 * (BM + BN) * X3_LDK * 2 >= WM * 32 * BN * 4, "the epilogue stages a tile pass in the same LDS");
  __shared__ __attribute__((aligned(16))) unsigned short smem16[3 * (BM + BN) * X3_LDK];
  unsigned short* const As = smem16;
  unsigned short* const Bs = smem16 + 3 * BM * X3_LDK;
  float* const smem = reinterpret_cast<float*>(smem16);
  const int tile = xcd_remap(blockIdx.x, epi.MT * NT);
  const int mt = tile / NT, nt = tile - mt * NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;
  const int arow = tid >> 3, kg = tid & 7;
  const int HoWo = g.Ho * g.Wo;
  const int RS = g.R * g.S, RST = RS * g.Rt;
  const int Tin = g.T * g.st_t, pad_t = g.Rt / 2;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.N * g.st_t * g.H * g.W * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, g.Cout * g.Ktot * 4, 0x00020000);

  int a_base[AP], a_hi0[AP], a_wi0[AP], a_t0[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + 32 * p;
    const bool ok = m < g.M;
    const int mm = ok ? m : 0;  // up to 2^23 output pixels here: exact integer division
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    a_hi0[p] = ok ? ho * g.stride - g.pad : -(1 << 20);
    a_wi0[p] = wo * g.stride - g.pad_w;
    a_t0[p] = (n % g.T) * g.st_t - pad_t;       // input frame (in its clip) that temporal tap 0 reads; T = 1 without temporal taps
    a_base[p] = (((n * g.st_t - pad_t) * g.H + ho * g.stride - g.pad) * g.W + wo * g.stride - g.pad_w) * 16;
  }
  float4 ra[AP], rb[BP];
  auto load = [&](int kt) {
    const int tap = kt * (BK / 4) + kg;  // this thread's filter tap = (dt * R + r) * S + s
    int dt, rs, r, s;
    fast_divmod(tap, RS, g.rcp_RS, dt, rs);
    fast_divmod(rs, g.S, g.rcp_S, r, s);
    const bool kv = tap < RST;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool v = kv && (unsigned)(a_hi0[p] + r) < (unsigned)g.H && (unsigned)(a_wi0[p] + s) < (unsigned)g.W &&
                     (unsigned)(a_t0[p] + dt) < (unsigned)Tin;
      ra[p] = buf_load16(xr, (a_base[p] + ((dt * g.H + r) * g.W + s) * 16) | (v ? 0 : kOOB), 0);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p)
      rb[p] = buf_load16(wr, (((nt * BN + arow + 32 * p) * g.Ktot + tap * 4) * 4) | (kv ? 0 : kOOB), 0);
  };
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  const int nk = (g.Ktot + BK - 1) / BK;
  load(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_split3<BM, AP>(As, ra, tid);
    store_split3<BN, BP>(Bs, rb, tid);
    __syncthreads();
    if (kt + 1 < nk) load(kt + 1);
    mma_stage_x3<TM, TN, 32 * WM, 32 * WN, BM, BN>(As, Bs, acc, wm0, wn0, lane);
  }
  fprop_epilogue<BM, BN, WM, WN>(acc, smem, y, g, epi, mt, nt, tid);
}

// fix-up for the K-split remainder tiles: sum the `split` partial accumulators in slice order, then the same
// element store (and BN column statistics) as the main kernel.  grid = rem_tiles, 256 threads mapped like the
// main kernel.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_fprop_fixup_kernel(const float* __restrict__ slab, float* __restrict__ y, Geom g,
                                                                int NT, Work wk, FpropEpi epi) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NACC = TM * TN * 16;
  constexpr int NTHR = 64 * WM * WN;
  __shared__ __attribute__((aligned(16))) float smem[WM * 32 * BN];
  const int rt = blockIdx.x, tile = wk.dp_tiles + rt;
  const int mt = tile / NT, nt = tile - mt * NT;
  const int tid = threadIdx.x;
  // all NACC loads of one slice are independent: keep them in flight together, slices summed in order
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  for (int sl = 0; sl < wk.split; ++sl) {
    const float* src = slab + (size_t)(rt * wk.split + sl) * NACC * NTHR + tid;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += src[(size_t)((i * TN + j) * 16 + e) * NTHR];
  }
  fprop_epilogue<BM, BN, WM, WN>(acc, smem, y, g, epi, mt, nt, tid);
}

// =========================================================================================
// dgrad: dxs[m=(n,h,w), ci] = sum_{tap, co} dy[n, (h+p-r)/st, (w+p-s)/st, co] * w[co, tap, ci]
// Stride 2 is decomposed into the 4 input-pixel parity classes (blockIdx.y): a class only visits the
// taps whose (h+p-r) is even, so no MFMA work is spent on structural zeros.
// epilogue: temporal un-shift (scatter to frame t+/-1) + optional masked residual-gradient add.
// =========================================================================================
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 3) void conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          float* __restrict__ dx, const float* __restrict__ add_src,
                                                          const uint32_t* __restrict__ add_mask, Geom g, int NT, Work wk,
                                                          float* __restrict__ slab, BnStat stat) {
  constexpr int LDA = BM, LDB = BN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  constexpr int BV = BN / 4;
  constexpr int AOFF = (BK * LDA + 3) & ~3;  // keep the B image 16-byte aligned
  constexpr int SMEM = (AOFF + BK * LDB) > (WM * 32 * BN) ? (AOFF + BK * LDB) : (WM * 32 * BN);
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* const As = smem;
  float* const Bs = smem + AOFF;

  // parity class of the input pixel (stride 1: a single class)
  const int st = g.stride;
  const int ph = blockIdx.y / st, pw = blockIdx.y - ph * st;
  const int Hc = (g.H - ph + st - 1) / st, Wc = (g.W - pw + st - 1) / st;
  const int Mc = g.N * Hc * Wc;
  const int MT = (Mc + BM - 1) / BM;
  const int r0 = (ph + g.pad) % st, s0 = (pw + g.pad_w) % st;
  const int nr = r0 < g.R ? (g.R - r0 + st - 1) / st : 0;
  const int ns = s0 < g.S ? (g.S - s0 + st - 1) / st : 0;
  const int bh = (ph + g.pad - r0) / st, bw = (pw + g.pad_w - s0) / st;
  const int ntap = nr * ns;
  const int nk = ntap * g.Cout / BK;  // 0 for a class no filter tap reaches (e.g. 1x1 stride 2, odd pixels)

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
    it.tile = 0;
    it.kb = 0;
    it.ke = nk;
    it.pslot = -1;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;  // tile i / j of the wave: + 32*WM*i / + 32*WN*j
  const int arow = tid >> 3, kg = tid & 7;
  const int HcWc = Hc * Wc;
  const int RS = g.R * g.S;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.N * g.Ho * g.Wo * g.Cout * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, g.Cout * RS * g.Cin * 4, 0x00020000);

  int a_base[AP], a_h[AP], a_w[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + 32 * p;
    const bool ok = m < Mc;
    const int mm = ok ? m : 0;
    const int n = mm / HcWc;
    const int rem = mm - n * HcWc;
    const int hc = rem / Wc, wc = rem - hc * Wc;
    a_h[p] = ok ? hc + bh : -(1 << 20);
    a_w[p] = wc + bw;
    a_base[p] = ((n * g.Ho + hc + bh) * g.Wo + wc + bw) * g.Cout * 4 + 16 * kg;
  }
  // weights: tile [32 co rows][BN ci] of tap `tap`: element ((co0 + krow) * RS + tap) * Cin + nt*BN + 4*c4
  int b_base[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int idx = tid + 256 * p;
    const int krow = idx / BV, c4 = idx - krow * BV;
    b_base[p] = (krow * RS * g.Cin + nt * BN + 4 * c4) * 4;
  }

  // K index state (uniform): kt = chunk * ntap + ir * ns + is  (tap-fastest)
  int chunk = ntap > 0 ? it.kb / ntap : 0, ir, is;
  {
    const int ct = ntap > 0 ? it.kb - chunk * ntap : 0;
    ir = ns > 0 ? ct / ns : 0;
    is = ct - ir * ns;
  }

  float4 ra[AP], rb[BP];
  auto load = [&]() {
    const int tap = (r0 + ir * st) * g.S + (s0 + is * st);
    const int koff_a = (chunk * BK - (ir * g.Wo + is) * g.Cout) * 4;
    const int koff_b = (chunk * BK * RS + tap) * g.Cin * 4;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool v = (unsigned)(a_h[p] - ir) < (unsigned)g.Ho && (unsigned)(a_w[p] - is) < (unsigned)g.Wo;
      ra[p] = buf_load16(yr, (a_base[p] + koff_a) | (v ? 0 : kOOB), 0);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) rb[p] = buf_load16(wr, b_base[p], koff_b);
    is += 1;
    const int w1 = (is == ns) ? 1 : 0;
    is = w1 ? 0 : is;
    ir += w1;
    const int w2 = (ir == nr) ? 1 : 0;
    ir = w2 ? 0 : ir;
    chunk += w2;
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  if (it.ke > it.kb) {
    load();
    for (int kt = it.kb; kt < it.ke; ++kt) {
      __syncthreads();
      store_transposed<LDA, AP>(As, ra, tid);
      store_direct<LDB, BN, BP>(Bs, rb, tid);
      __syncthreads();
      if (kt + 1 < it.ke) load();
      mma_stage<LDA, LDB, TM, TN, 32 * WM, 32 * WN, true, false>(As, Bs, acc, wm0, wn0, lane);
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
  dgrad_epilogue<BM, BN, WM, WN, BN == 64>(acc, smem, tid, dx, add_src, add_mask, g, mt, nt, Mc, stat, [&](int mrow) {
    if (st == 1) return mrow;
    const int n = mrow / HcWc;
    const int rem = mrow - n * HcWc;
    const int hc = rem / Wc;
    return (n * g.H + hc * st + ph) * g.W + (rem - hc * Wc) * st + pw;
  });
}

template <int BM, int BN, int WM, int WN, bool BPRE = false>
__global__ __launch_bounds__(256, 2) void conv_dgrad_x3_kernel(const float* __restrict__ dy, const float* __restrict__ wt,
                                                          float* __restrict__ dx, const float* __restrict__ add_src,
                                                          const uint32_t* __restrict__ add_mask, Geom g, int NT, Work wk,
                                                          float* __restrict__ slab, BnStat stat) {
  // bf16-piece variant of conv_dgrad_kernel (see conv_fprop_x3_kernel).  wt = the weights transposed per tap,
  // [R*S][Cin][Cout], so that the B operand is k-contiguous like dy; BPRE: wt = the D planes of bdv_conv_split_weights.
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  static_assert(3 * (BM + BN) * X3_LDK * 2 >= WM * 32 * BN * 4, "the epilogue stages a tile pass in the same LDS");
  __shared__ __attribute__((aligned(16))) unsigned short smem16[3 * (BM + BN) * X3_LDK];
  unsigned short* const As = smem16;
  unsigned short* const Bs = smem16 + 3 * BM * X3_LDK;
  float* const smem = reinterpret_cast<float*>(smem16);

  // parity class of the input pixel (stride 1: a single class)
  const int st = g.stride;
  const int ph = blockIdx.y / st, pw = blockIdx.y - ph * st;
  const int Hc = (g.H - ph + st - 1) / st, Wc = (g.W - pw + st - 1) / st;
  const int Mc = g.N * Hc * Wc;
  const int MT = (Mc + BM - 1) / BM;
  const int r0 = (ph + g.pad) % st, s0 = (pw + g.pad_w) % st;
  const int nr = r0 < g.R ? (g.R - r0 + st - 1) / st : 0;
  const int ns = s0 < g.S ? (g.S - s0 + st - 1) / st : 0;
  const int bh = (ph + g.pad - r0) / st, bw = (pw + g.pad_w - s0) / st;
  const int ntap = nr * ns;
  const int nk = ntap * g.Cout / BK;  // 0 for a class no filter tap reaches (e.g. 1x1 stride 2, odd pixels)

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
    it.tile = 0;
    it.kb = 0;
    it.ke = nk;
    it.pslot = -1;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;  // tile i / j of the wave: + 32*WM*i / + 32*WN*j
  const int arow = tid >> 3, kg = tid & 7;
  const int HcWc = Hc * Wc;
  const int RS = g.R * g.S;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.N * g.Ho * g.Wo * g.Cout * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wt, 0, BPRE ? 3 * g.Cout * RS * g.Cin * 2 : g.Cout * RS * g.Cin * 4, 0x00020000);
  constexpr int BQ = BN * 4 / 256;
  const int brow = tid >> 2, bc = tid & 3;
  const int plane_bytes = g.Cout * RS * g.Cin * 2;
  const int nchunk = g.Cout / BK;

  int a_base[AP], a_h[AP], a_w[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + 32 * p;
    const bool ok = m < Mc;
    const int mm = ok ? m : 0;
    const int n = mm / HcWc;
    const int rem = mm - n * HcWc;
    const int hc = rem / Wc, wc = rem - hc * Wc;
    a_h[p] = ok ? hc + bh : -(1 << 20);
    a_w[p] = wc + bw;
    a_base[p] = ((n * g.Ho + hc + bh) * g.Wo + wc + bw) * g.Cout * 4 + 16 * kg;
  }
  // transposed weights: row ci = nt*BN + arow + 32p of tap `tap`, 4 consecutive co: (tap * Cin + ci) * Cout + co0 + 4*kg
  int b_base[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) b_base[p] = ((nt * BN + arow + 32 * p) * g.Cout + 4 * kg) * 4;

  // K index state (uniform): kt = chunk * ntap + ir * ns + is  (tap-fastest)
  int chunk = ntap > 0 ? it.kb / ntap : 0, ir, is;
  {
    const int ct = ntap > 0 ? it.kb - chunk * ntap : 0;
    ir = ns > 0 ? ct / ns : 0;
    is = ct - ir * ns;
  }

  float4 ra[AP], rb[BP];
  u32x4 rbp[BPRE ? 3 * BQ : 1];
  auto load = [&]() {
    const int tap = (r0 + ir * st) * g.S + (s0 + is * st);
    const int koff_a = (chunk * BK - (ir * g.Wo + is) * g.Cout) * 4;
    const int koff_b = (tap * g.Cin * g.Cout + chunk * BK) * 4;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool v = (unsigned)(a_h[p] - ir) < (unsigned)g.Ho && (unsigned)(a_w[p] - is) < (unsigned)g.Wo;
      ra[p] = buf_load16(yr, (a_base[p] + koff_a) | (v ? 0 : kOOB), 0);
    }
    if (BPRE) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int q = 0; q < BQ; ++q)
          rbp[pl * BQ + q] = __builtin_amdgcn_raw_buffer_load_b128(wr, (nt * BN + brow + 64 * q) * 64 + 16 * bc + pl * plane_bytes,
                                                                   (tap * nchunk + chunk) * g.Cin * 64, 0);
    } else {
#pragma unroll
      for (int p = 0; p < BP; ++p) rb[p] = buf_load16(wr, b_base[p], koff_b);
    }
    is += 1;
    const int w1 = (is == ns) ? 1 : 0;
    is = w1 ? 0 : is;
    ir += w1;
    const int w2 = (ir == nr) ? 1 : 0;
    ir = w2 ? 0 : ir;
    chunk += w2;
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  if (it.ke > it.kb) {
    load();
    for (int kt = it.kb; kt < it.ke; ++kt) {
      __syncthreads();
      store_split3<BM, AP>(As, ra, tid);
      if (BPRE) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int q = 0; q < BQ; ++q) *reinterpret_cast<u32x4*>(Bs + (pl * BN + brow + 64 * q) * X3_LDK + 8 * bc) = rbp[pl * BQ + q];
      } else {
        store_split3<BN, BP>(Bs, rb, tid);
      }
      __syncthreads();
      if (kt + 1 < it.ke) load();
      mma_stage_x3<TM, TN, 32 * WM, 32 * WN, BM, BN>(As, Bs, acc, wm0, wn0, lane);
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
  dgrad_epilogue<BM, BN, WM, WN, true, 4, true>(acc, smem, tid, dx, add_src, add_mask, g, mt, nt, Mc, stat, [&](int mrow) {
    if (st == 1) return mrow;
    const int n = mrow / HcWc;
    const int rem = mrow - n * HcWc;
    const int hc = rem / Wc;
    return (n * g.H + hc * st + ph) * g.W + (rem - hc * Wc) * st + pw;
  });
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_dgrad_fixup_kernel(const float* __restrict__ slab, float* __restrict__ dx,
                                                                const float* __restrict__ add_src,
                                                                const uint32_t* __restrict__ add_mask, Geom g, int NT, Work wk,
                                                                BnStat stat) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NACC = TM * TN * 16;
  constexpr int NTHR = 64 * WM * WN;
  __shared__ __attribute__((aligned(16))) float smem[WM * 32 * BN];
  const int rt = blockIdx.x, tile = wk.dp_tiles + rt;
  const int mt = tile / NT, nt = tile - mt * NT;
  const int tid = threadIdx.x;
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  for (int sl = 0; sl < wk.split; ++sl) {
    const float* src = slab + (size_t)(rt * wk.split + sl) * NACC * NTHR + tid;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += src[(size_t)((i * TN + j) * 16 + e) * NTHR];
  }
  dgrad_epilogue<BM, BN, WM, WN>(acc, smem, tid, dx, add_src, add_mask, g, mt, nt, g.M, stat, [](int mrow) { return mrow; });
}

// =========================================================================================
// wgrad: slab[split][co][tap*Cin + ci] = sum_{m in split} dy[m, co] * x_shift[pix(m, tap), ci]
// Both operands are M-contiguous (16-byte direct LDS stores).  C4 = stem (Cin = 4, one tap per 16-byte load).
// =========================================================================================
template <int BM, int BN, int WM, int WN, bool C4, bool INCR>
__global__ __launch_bounds__(256, 3) void conv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ slab, Geom g, int MTw, int NTw,
                                                          int kt_per_split) {
  constexpr int LDA = BM, LDB = BN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  constexpr int AV = BM / 4, BV = BN / 4;
  __shared__ __attribute__((aligned(16))) float smem[BK * (LDA + LDB)];
  float* const As = smem;
  float* const Bs = smem + BK * LDA;

  // All tiles of one K slice read the same rows of dy and (up to the filter halo) of x.  Work index w = slice * tiles + tile,
  // handed out so that one XCD gets a contiguous range of w: the tiles of a slice then run at the same time behind the
  // same L2 and each operand row is fetched from the fabric about once instead of once per tile (measured before the
  // remap: 0.77 GB fetched per launch of the 128x128 kernel, 2-4 GB for the 64-channel layers and the stem).
  const int tiles = MTw * NTw;
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int split = wi / tiles;
  const int tile = wi - split * tiles;
  const int mt = tile % MTw, nt = tile / MTw;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;  // tile i / j of the wave: + 32*WM*i / + 32*WN*j
  const int HoWo = g.Ho * g.Wo;
  const int frame_bytes = g.H * g.W * g.Cin * 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.N * frame_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.M * g.Cout * 4, 0x00020000);

  // A: dy rows m0 + krow, columns mt*BM + 4*c4
  int a_off[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int idx = tid + 256 * p;
    const int krow = idx / AV, c4 = idx - krow * AV;
    a_off[p] = (krow * g.Cout + mt * BM + 4 * c4) * 4;
  }
  // B: column -> (tap, ci) for this thread's loads (fixed over the K loop)
  int b_krow[BP], b_off[BP], b_r[BP], b_s[BP], b_cls[BP];
  bool b_cok[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int idx = tid + 256 * p;
    const int krow = idx / BV, c4 = idx - krow * BV;
    b_krow[p] = krow;
    int tap, ci;
    if (C4) {
      const int ncol = nt * BN + 4 * c4;
      tap = ncol / g.Cin;
      ci = ncol - tap * g.Cin;
      b_cok[p] = ncol < g.Ktot;
    } else {
      const int per_tap = g.Cin / BN;
      tap = nt / per_tap;
      ci = (nt - tap * per_tap) * BN + 4 * c4;
      b_cok[p] = true;
    }
    b_r[p] = tap / g.S - g.pad;
    b_s[p] = tap % g.S - g.pad_w;
    b_cls[p] = shift_class(ci, g.fold);
    b_off[p] = ((b_r[p] * g.W + b_s[p]) * g.Cin + ci) * 4 + b_cls[p] * frame_bytes;
  }

  const int kt_begin = split * kt_per_split;
  const int nkt_all = (g.M + BK - 1) / BK;
  const int kt_end = min(kt_begin + kt_per_split, nkt_all);

  // INCR: the B rows of a thread advance by BK = 32 output pixels per K step, so the input position (hi, wi), the
  // frame-in-clip index and the byte offset of the pixel are carried from step to step with two conditional wraps
  // and additions only -- no division, modulo or integer multiply in the loop (the loader used to cost 325 VALU
  // instructions per K step against 145 in fprop, and VALU issue is what holds the MFMA pipe of these kernels
  // below 70 %).  The host selects INCR when a step spans less than one frame and at most Ho - 1 whole rows.
  const int st = g.stride;
  const int d_ho = BK / g.Wo, d_wo = BK - d_ho * g.Wo;
  const int wrap_w = g.Wo * st, wrap_h = g.Ho * st;
  const int px = g.Cin * 4;                                   // bytes per input pixel
  const int inc0 = (d_ho * st * g.W + d_wo * st) * px;        // plain advance
  const int inc1 = (st * g.W - wrap_w) * px;                  // extra when wo wraps into the next output row
  const int inc2 = (g.H * g.W - wrap_h * g.W) * px;           // extra when ho wraps into the next frame
  int s_hi[BP], s_wi[BP], s_t[BP], s_off[BP];
  if (INCR) {
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int m = split * kt_per_split * BK + b_krow[p];
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
      s_hi[p] = ho * st;
      s_wi[p] = wo * st;
      s_t[p] = n % g.T;
      s_off[p] = ((n * g.H + s_hi[p]) * g.W + s_wi[p]) * px + b_off[p];
    }
  }

  float4 ra[AP], rb[BP];
  auto load = [&](int kt) {
    const int m0 = kt * BK;
#pragma unroll
    for (int p = 0; p < AP; ++p)  // rows past M lie past num_records (the range check covers the vector offset): zeros
      ra[p] = buf_load16(yr, a_off[p] + m0 * g.Cout * 4, 0);
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int m = m0 + b_krow[p];
      const bool mok = m < g.M;
      int hi, wi, t, off;
      if (INCR) {
        hi = s_hi[p];
        wi = s_wi[p];
        t = s_t[p];
        off = s_off[p];
        const int w2 = wi + d_wo * st;
        const bool c1 = w2 >= wrap_w;
        s_wi[p] = c1 ? w2 - wrap_w : w2;
        const int h2 = hi + d_ho * st + (c1 ? st : 0);
        const bool c2 = h2 >= wrap_h;
        s_hi[p] = c2 ? h2 - wrap_h : h2;
        const int t2 = t + (c2 ? 1 : 0);
        s_t[p] = t2 == g.T ? 0 : t2;
        s_off[p] = off + inc0 + (c1 ? inc1 : 0) + (c2 ? inc2 : 0);
      } else {  // exact integer divisions (tiny feature maps)
        const int mm = mok ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        const int ho = rem / g.Wo;
        hi = ho * st;
        wi = (rem - ho * g.Wo) * st;
        t = n % g.T;
        off = ((n * g.H + hi) * g.W + wi) * px + b_off[p];
      }
      const bool v = mok && b_cok[p] && (unsigned)(hi + b_r[p]) < (unsigned)g.H && (unsigned)(wi + b_s[p]) < (unsigned)g.W &&
                     (unsigned)(t + b_cls[p]) < (unsigned)g.T;
      rb[p] = buf_load16(xr, off | (v ? 0 : kOOB), 0);
    }
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  if (kt_begin < kt_end) {
    load(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      __syncthreads();
      store_direct<LDA, BM, AP>(As, ra, tid);
      store_direct<LDB, BN, BP>(Bs, rb, tid);
      __syncthreads();
      if (kt + 1 < kt_end) load(kt + 1);
      mma_stage<LDA, LDB, TM, TN, 32 * WM, 32 * WN, false, false>(As, Bs, acc, wm0, wn0, lane);
    }
  }

  float* out = slab + (size_t)split * g.Cout * g.Ktot;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int col;
    bool cok = true;
    if (C4) {
      col = nt * BN + wn0 + 32 * WN * j + (lane & 31);
      cok = col < g.Ktot;
    } else {
      const int per_tap = g.Cin / BN;
      const int tap = nt / per_tap;
      col = tap * g.Cin + (nt - tap * per_tap) * BN + wn0 + 32 * WN * j + (lane & 31);
    }
    if (!cok) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mt * BM + wm0 + 32 * WM * i + acc_row(e, lane);
        out[(size_t)row * g.Ktot + col] = acc[i][j][e];
      }
  }
}

template <int BM, int BN, int WM, int WN, bool C4, bool INCR>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ slab, Geom g, int MTw, int NTw,
                                                          int kt_per_split) {
  // bf16-piece variant of conv_wgrad_kernel (see conv_fprop_x3_kernel / mma_stage_x3_pairs)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  constexpr int AV = BM / 4, BV = BN / 4;
  static_assert(BM == 128 && BN == 128, "row-pair assignment of the loader");
  __shared__ __attribute__((aligned(16))) unsigned smem32[3 * X3_PAIR_ROWS * (BM + BN)];
  unsigned* const As = smem32;
  unsigned* const Bs = smem32 + 3 * X3_PAIR_ROWS * BM;

  // All tiles of one K slice read the same rows of dy and (up to the filter halo) of x.  Work index w = slice * tiles + tile,
  // handed out so that one XCD gets a contiguous range of w: the tiles of a slice then run at the same time behind the
  // same L2 and each operand row is fetched from the fabric about once instead of once per tile (measured before the
  // remap: 0.77 GB fetched per launch of the 128x128 kernel, 2-4 GB for the 64-channel layers and the stem).
  const int tiles = MTw * NTw;
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int split = wi / tiles;
  const int tile = wi - split * tiles;
  const int mt = tile % MTw, nt = tile / MTw;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;  // tile i / j of the wave: + 32*WM*i / + 32*WN*j
  const int HoWo = g.Ho * g.Wo;
  const int frame_bytes = g.H * g.W * g.Cin * 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.N * frame_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.M * g.Cout * 4, 0x00020000);

  // A: dy rows m0 + krow, columns mt*BM + 4*c4
  int a_off[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int c4 = tid % AV;
    const int krow = 2 * (tid / AV) + (p & 1) + 16 * (p >> 1);  // rows (k, k+1) of a k-pair sit in one thread
    a_off[p] = (krow * g.Cout + mt * BM + 4 * c4) * 4;
  }
  // B: column -> (tap, ci) for this thread's loads (fixed over the K loop)
  int b_krow[BP], b_off[BP], b_r[BP], b_s[BP], b_cls[BP];
  bool b_cok[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int c4 = tid % BV;
    const int krow = 2 * (tid / BV) + (p & 1) + 16 * (p >> 1);
    b_krow[p] = krow;
    int tap, ci;
    if (C4) {
      const int ncol = nt * BN + 4 * c4;
      tap = ncol / g.Cin;
      ci = ncol - tap * g.Cin;
      b_cok[p] = ncol < g.Ktot;
    } else {
      const int per_tap = g.Cin / BN;
      tap = nt / per_tap;
      ci = (nt - tap * per_tap) * BN + 4 * c4;
      b_cok[p] = true;
    }
    b_r[p] = tap / g.S - g.pad;
    b_s[p] = tap % g.S - g.pad_w;
    b_cls[p] = shift_class(ci, g.fold);
    b_off[p] = ((b_r[p] * g.W + b_s[p]) * g.Cin + ci) * 4 + b_cls[p] * frame_bytes;
  }

  const int kt_begin = split * kt_per_split;
  const int nkt_all = (g.M + BK - 1) / BK;
  const int kt_end = min(kt_begin + kt_per_split, nkt_all);

  // INCR: the B rows of a thread advance by BK = 32 output pixels per K step, so the input position (hi, wi), the
  // frame-in-clip index and the byte offset of the pixel are carried from step to step with two conditional wraps
  // and additions only -- no division, modulo or integer multiply in the loop (the loader used to cost 325 VALU
  // instructions per K step against 145 in fprop, and VALU issue is what holds the MFMA pipe of these kernels
  // below 70 %).  The host selects INCR when a step spans less than one frame and at most Ho - 1 whole rows.
  const int st = g.stride;
  const int d_ho = BK / g.Wo, d_wo = BK - d_ho * g.Wo;
  const int wrap_w = g.Wo * st, wrap_h = g.Ho * st;
  const int px = g.Cin * 4;                                   // bytes per input pixel
  const int inc0 = (d_ho * st * g.W + d_wo * st) * px;        // plain advance
  const int inc1 = (st * g.W - wrap_w) * px;                  // extra when wo wraps into the next output row
  const int inc2 = (g.H * g.W - wrap_h * g.W) * px;           // extra when ho wraps into the next frame
  int s_hi[BP], s_wi[BP], s_t[BP], s_off[BP];
  if (INCR) {
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int m = split * kt_per_split * BK + b_krow[p];
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
      s_hi[p] = ho * st;
      s_wi[p] = wo * st;
      s_t[p] = n % g.T;
      s_off[p] = ((n * g.H + s_hi[p]) * g.W + s_wi[p]) * px + b_off[p];
    }
  }

  float4 ra[AP], rb[BP];
  auto load = [&](int kt) {
    const int m0 = kt * BK;
#pragma unroll
    for (int p = 0; p < AP; ++p)  // rows past M lie past num_records (the range check covers the vector offset): zeros
      ra[p] = buf_load16(yr, a_off[p] + m0 * g.Cout * 4, 0);
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int m = m0 + b_krow[p];
      const bool mok = m < g.M;
      int hi, wi, t, off;
      if (INCR) {
        hi = s_hi[p];
        wi = s_wi[p];
        t = s_t[p];
        off = s_off[p];
        const int w2 = wi + d_wo * st;
        const bool c1 = w2 >= wrap_w;
        s_wi[p] = c1 ? w2 - wrap_w : w2;
        const int h2 = hi + d_ho * st + (c1 ? st : 0);
        const bool c2 = h2 >= wrap_h;
        s_hi[p] = c2 ? h2 - wrap_h : h2;
        const int t2 = t + (c2 ? 1 : 0);
        s_t[p] = t2 == g.T ? 0 : t2;
        s_off[p] = off + inc0 + (c1 ? inc1 : 0) + (c2 ? inc2 : 0);
      } else {  // exact integer divisions (tiny feature maps)
        const int mm = mok ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        const int ho = rem / g.Wo;
        hi = ho * st;
        wi = (rem - ho * g.Wo) * st;
        t = n % g.T;
        off = ((n * g.H + hi) * g.W + wi) * px + b_off[p];
      }
      const bool v = mok && b_cok[p] && (unsigned)(hi + b_r[p]) < (unsigned)g.H && (unsigned)(wi + b_s[p]) < (unsigned)g.W &&
                     (unsigned)(t + b_cls[p]) < (unsigned)g.T;
      rb[p] = buf_load16(xr, off | (v ? 0 : kOOB), 0);
    }
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  if (kt_begin < kt_end) {
    load(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      __syncthreads();
      store_split3_pairs<BM, AP>(As, ra, tid);
      store_split3_pairs<BN, BP>(Bs, rb, tid);
      __syncthreads();
      if (kt + 1 < kt_end) load(kt + 1);
      mma_stage_x3_pairs<TM, TN, 32 * WM, 32 * WN, BM, BN>(As, Bs, acc, wm0, wn0, lane);
    }
  }

  float* out = slab + (size_t)split * g.Cout * g.Ktot;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int col;
    bool cok = true;
    if (C4) {
      col = nt * BN + wn0 + 32 * WN * j + (lane & 31);
      cok = col < g.Ktot;
    } else {
      const int per_tap = g.Cin / BN;
      const int tap = nt / per_tap;
      col = tap * g.Cin + (nt - tap * per_tap) * BN + wn0 + 32 * WN * j + (lane & 31);
    }
    if (!cok) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mt * BM + wm0 + 32 * WM * i + acc_row(e, lane);
        out[(size_t)row * g.Ktot + col] = acc[i][j][e];
      }
  }
}

// dw = beta*dw + sum over splits (fixed order).  64 float4 elements x 4 split-lanes per block.
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ slab, float* __restrict__ dw, float beta, int splits,
                                                   int64_t numel4, int block) {
  __shared__ float4 sh[4][64];
  const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int64_t i = (int64_t)block * 64 + el;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < numel4) {
    for (int k = sg; k < splits; k += 4) {
      const float4 v = reinterpret_cast<const float4*>(slab)[(int64_t)k * numel4 + i];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
  }
  sh[sg][el] = s;
  __syncthreads();
  if (sg == 0 && i < numel4) {
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 v = sh[k][el];
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

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, float beta,
                                                            int splits, int64_t numel4) {
  wgrad_reduce_block(slab, dw, beta, splits, numel4, blockIdx.x);
}

// The same reduction for up to BDV_MAX_REDUCE_ITEMS weight gradients in one launch (one stage's worth): block b works on
// the item whose block range contains it.  Per element the summation order is that of wgrad_reduce_kernel.
struct ReduceBatch {
  const float* slab[BDV_MAX_REDUCE_ITEMS];
  float* dw[BDV_MAX_REDUCE_ITEMS];
  int splits[BDV_MAX_REDUCE_ITEMS];
  int numel4[BDV_MAX_REDUCE_ITEMS];
  int first_block[BDV_MAX_REDUCE_ITEMS + 1];
  int n;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(ReduceBatch rb, float beta) {
  int k = 0;
  while (k + 1 < rb.n && (int)blockIdx.x >= rb.first_block[k + 1]) ++k;  // uniform per block
  wgrad_reduce_block(rb.slab[k], rb.dw[k], beta, rb.splits[k], rb.numel4[k], blockIdx.x - rb.first_block[k]);
}

// =========================================================================================
// Round-2 kernels for the bf16-piece arithmetic ("P" family): 8 waves per workgroup, one workgroup per CU, the weights
// pre-split once per step into bf16 planes (bdv_conv_split_weights), two LDS stages and one barrier per K-step.
//
// What changed against conv_fprop_x3_kernel / conv_dgrad_x3_kernel, and why (tools/ubench/gemm_x3p.hip, profiles/r02_*):
// those kernels sit at ~37 % of the bf16 MFMA time because every K-step pays, per 48 MFMAs of a wave, ~180 VALU instructions
// for splitting BOTH operand tiles into pieces plus 24 LDS stores, and the issue slots of a SIMD are full before its matrix
// pipe is.  Here
//   * the weight operand arrives as ready-made planes [piece][K-step][row][32] (a K-step's tile is one contiguous block):
//     16-byte loads, ds_write_b128, no VALU;
//   * the activation tile is split once per 256 (or 128) output columns instead of once per 128: VALU per MFMA halves again;
//   * LDS rows are 64 bytes (32 bf16), unpadded, with the 16-byte chunk index XOR-ed by (row >> 2) & 3: fragment reads
//     (ds_read_b128) and both kinds of stores are conflict-free, and two stages of a 128x256 tile fit in 144 KB;
//   * with two stages the registers of step k+1 are split and stored while other waves still read stage k: one barrier per
//     K-step, and the compiler interleaves the split with the MFMAs.
// Loader (halo / clip-end / ragged rows via out-of-range buffer offsets), work planner, K-split fix-ups and all epilogues are
// the ones of the kernels above.
// =========================================================================================

// byte offset of 16-byte chunk c (0..3) of row `row` in a plane image of 64-byte rows
__device__ __forceinline__ int pl_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

// fp32 x 4 (row `row`, floats 4 kg .. 4 kg + 3 of the 32-deep step) -> hi / mid / lo planes of the stage image
__device__ __forceinline__ void pl_split_store(unsigned char* __restrict__ base, int plane_bytes, int row, int kg, const float4 v) {
  u32x2_t hi, mid, lo;
  split3_pairs(v, hi, mid, lo);
  unsigned char* q = base + pl_off(row, kg >> 1) + 8 * (kg & 1);
  *reinterpret_cast<u32x2_t*>(q) = hi;
  *reinterpret_cast<u32x2_t*>(q + plane_bytes) = mid;
  *reinterpret_cast<u32x2_t*>(q + 2 * plane_bytes) = lo;
}

// 32-deep K-step on one stage: wave (wm, wn) owns rows 32 WM i + 32 wm + r and columns 32 WN j + 32 wn + c (interleaved
// tiling, as the epilogues expect).  fa[s] / fb[s] are the lane's byte offsets of its fragment of 16-deep step s in tile 0 of
// plane 0 (pl_frag_off); every other fragment of the stage is a compile-time constant away (the swizzle term only depends on
// (row >> 2) & 3, which multiples of 32 rows do not change), so the reads carry immediate offsets.
__device__ __forceinline__ int pl_frag_off(int row0, int s, int lane) {
  const int r = lane & 31, h = lane >> 5;
  return pl_off(row0 + r, 2 * s + h);
}

// NP = 3: the six piece products (fp32-level result).  NP = 1: only the leading bf16 piece of each operand, one MFMA product
// per step -- the reduced-precision arithmetic of BASELINE config 5 (bf16 operands, fp32 accumulate), see bdv_conv_fprop_pl.
// NP = 2: the two leading pieces of each operand (16 significand bits) and the three products hi*hi + hi*mid + mid*hi: dropped
// terms <= 2^-16 (mid*mid) + 2 * 2^-17 (the operands' remainders) relative per product -- between TF32 (2^-11) and fp32.
template <int BM, int BN, int WM, int WN, int NP = 3>
__device__ __forceinline__ void mma_stage_pl(const unsigned char* __restrict__ As, const unsigned char* __restrict__ Bs,
                                             f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const int (&fa)[2], const int (&fb)[2]) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int PA = BM * 64, PB = BN * 64;
#pragma unroll
  for (int s = 0; s < BK / 16; ++s) {
    bf16x8_t a[NP][TM], b[NP][TN];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[p][i] = *reinterpret_cast<const bf16x8_t*>(As + fa[s] + (p * PA + 32 * WM * i * 64));
#pragma unroll
      for (int j = 0; j < TN; ++j) b[p][j] = *reinterpret_cast<const bf16x8_t*>(Bs + fb[s] + (p * PB + 32 * WN * j * 64));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {  // smallest terms first
        if constexpr (NP == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
        }
        if constexpr (NP >= 2) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
      }
  }
}

// fp32 x 4 -> the three planes (NP = 1: the leading one) of a stage image at the thread's precomputed byte offset
template <int NP = 3>
__device__ __forceinline__ void pl_split_store_at(unsigned char* __restrict__ q, int plane_bytes, const float4 v) {
  if constexpr (NP == 1) {
    *reinterpret_cast<u32x2_t*>(q) = bf16_hi_pairs(v);
    return;
  }
  u32x2_t hi, mid, lo;
  split3_pairs(v, hi, mid, lo);
  *reinterpret_cast<u32x2_t*>(q) = hi;
  *reinterpret_cast<u32x2_t*>(q + plane_bytes) = mid;
  if constexpr (NP == 3) *reinterpret_cast<u32x2_t*>(q + 2 * plane_bytes) = lo;
}

// ---- weights -> bf16 planes ------------------------------------------------------------------
// F (fprop):  [piece][K-step = chunk * R*S + tap][co][32 ci of the chunk]      (chunk = ci / 32; tap-fastest, the K order of fprop)
// D (dgrad):  [piece][tap][co chunk][ci][32 co of the chunk]                   (the contraction index co contiguous per ci row)
// One block = one (32 co x 32 ci) tile of one tap; D is written through an LDS transpose so both stores are 64-byte rows.
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ F,
                                                             unsigned short* __restrict__ D, int Cout, int RS, int Cin) {
  __shared__ unsigned short tile[3][32][34];
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, tap = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t plane = (size_t)Cout * RS * Cin;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int co_l = ty + 8 * p;
    const float a = w[((size_t)(co0 + co_l) * RS + tap) * Cin + ci0 + tx];
    const __bf16 hi = (__bf16)a;
    const float r1 = a - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const __bf16 lo = (__bf16)(r1 - (float)mid);
    const unsigned short h16 = __builtin_bit_cast(unsigned short, hi), m16 = __builtin_bit_cast(unsigned short, mid),
                         l16 = __builtin_bit_cast(unsigned short, lo);
    if (F != nullptr) {
      const size_t o = (((size_t)(ci0 >> 5) * RS + tap) * Cout + co0 + co_l) * 32 + tx;
      F[o] = h16;
      F[plane + o] = m16;
      F[2 * plane + o] = l16;
    }
    tile[0][co_l][tx] = h16;
    tile[1][co_l][tx] = m16;
    tile[2][co_l][tx] = l16;
  }
  if (D == nullptr) return;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ci_l = ty + 8 * p;  // tx = co within the chunk
    const size_t o = (((size_t)tap * (Cout >> 5) + (co0 >> 5)) * Cin + ci0 + ci_l) * 32 + tx;
    D[o] = tile[0][tx][ci_l];
    D[plane + o] = tile[1][tx][ci_l];
    D[2 * plane + o] = tile[2][tx][ci_l];
  }
}

// Two LDS stages, two register sets, one barrier per K-step.  While the MFMAs of step i read stage i & 1, the registers of step
// i + 1 (loaded during step i - 1) are split and stored into the other stage and the loads of step i + 3 are issued.  The
// steady state is free of conditions, so a step's MFMAs, stores and loads are one basic block the scheduler can interleave
// (tools/ubench/gemm_x3p.hip: 196 -> 215 TFLOP/s on a large GEMM, 159 -> 175 on the 3x3 256-channel site against the form with
// one register set and `if (kt + 1 < ke)` around every store).  n = number of K-steps; load(set) loads the NEXT step in order.
using PlSet0 = std::integral_constant<int, 0>;
using PlSet1 = std::integral_constant<int, 1>;
// STAGGER (round 3): the two waves of a SIMD run the steady-state step in OPPOSITE order.  A step's MFMAs (48 - 96 per wave) and its
// vector work (the piece split of the next activation tile, the LDS stores, the loads' address arithmetic: ~170 instructions) were
// one after the other in every wave, and the barrier kept SIMD partners in the same phase -- the matrix pipe idled while both waves
// split, the vector issue idled while both multiplied: in-kernel stamps (tools/stamp_tiles.py) give 2.24 us per K-step of the
// 128x256 tile = 4370 cycles at 1.95 GHz for 3072 cycles of MFMA work, the rest being the ~1360 cycles the two waves' vector work
// takes to issue.  Waves >= NW/2 (the SIMD partners of waves 0 .. NW/2 - 1: a workgroup's waves go to the SIMDs in cyclic order) now
// do store + load first and the MFMAs last, so one wave's vector block runs beside its partner's MFMA block.  Any order inside a
// step is legal: the stage a step stores was read before the previous barrier, the stage it reads was stored before it.
template <typename LoadF, typename StoreF, typename MmaF>
__device__ __forceinline__ void pl_pipeline2(int n, LoadF&& load, StoreF&& store, MmaF&& mma, bool late = false) {
  load(PlSet0{});
  store(0, PlSet0{});
  if (n > 1) load(PlSet1{});
  if (n > 2) load(PlSet0{});
  __syncthreads();
  BDV_STAMP(1);
  int i = 0;
  if (late) {
    for (; i + 4 < n; i += 2) {
      store(1, PlSet1{});
      load(PlSet1{});
      mma(0);
      __syncthreads();
      store(0, PlSet0{});
      load(PlSet0{});
      mma(1);
      __syncthreads();
    }
  } else {
    for (; i + 4 < n; i += 2) {
      mma(0);
      store(1, PlSet1{});
      load(PlSet1{});
      __syncthreads();
      mma(1);
      store(0, PlSet0{});
      load(PlSet0{});
      __syncthreads();
    }
  }
  for (; i < n; i += 2) {
    mma(0);
    if (i + 1 < n) store(1, PlSet1{});
    if (i + 3 < n) load(PlSet1{});
    __syncthreads();
    if (i + 1 >= n) break;
    mma(1);
    if (i + 2 < n) store(0, PlSet0{});
    if (i + 4 < n) load(PlSet0{});
    __syncthreads();
  }
}

// Occupancy of the 8-wave plane kernels (second __launch_bounds__ argument = waves per SIMD).  NP = 3: one workgroup per CU (two
// waves per SIMD, 256 VGPRs): the K loop is MFMA-bound and the three-plane stages fill the LDS.  NP = 1 (single-product
// arithmetic, BASELINE config 5): a sixth of the MFMA work per K-step leaves the loop waiting for its global loads (1.2 us per
// K-step measured, 0.27 us of it MFMA time), and a one-plane stage is a third of the LDS -- two workgroups per CU (four waves
// per SIMD, 128 VGPRs) keep twice the loads in flight and let one workgroup's MFMA phase run under the other's wait.  Only where
// the tile fits 128 VGPRs: measured per training step at batch 64 (gpurun_out/np1_*.log), fprop 128x256 7.40 -> 6.39 ms (52 - 64
// bytes of scratch), fprop 256x64 1.97 -> 1.39, dgrad 256x64 3.64 -> 3.19; the others spill their accumulators (236 - 600 bytes
// per lane) and run 2 - 3 x slower (dgrad 128x256 13.3 -> 25.3 ms), so they keep one workgroup per CU.
// Four-wave workgroups (the 128x128 tile, round 3): two per CU (256 VGPRs each), so that one tile's prologue / epilogue runs beside
// the other's K loop.
constexpr int pl_waves_per_simd(int waves, int np, bool fits128) { return np == 1 && waves == 8 && fits128 ? 4 : waves == 4 ? 2 : waves / 4; }

// ---- fprop ------------------------------------------------------------------------------------
// ES: bytes per element of x / y / the residual (4 = fp32, 2 = bf16 storage: NP = 1 only, the loader's conversion is then exact)
template <int BM, int BN, int WM, int WN, int NBUF, int NP = 3, bool PRE = false, int ES = 4>
__global__ __launch_bounds__(64 * WM * WN, pl_waves_per_simd(WM * WN, NP, BN == 64 || (BM == 128 && BN == 256))) void conv_fprop_pl_kernel(const void* __restrict__ x,
                                                                                      const unsigned short* __restrict__ wp,
                                                                                      void* __restrict__ y, Geom g, int NT, Work wk,
                                                                                      float* __restrict__ slab, FpropEpi epi) {
  constexpr int NTHR = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  // a thread's unit of the activation tile is 16 bytes: 4 fp32 values (8 threads per 32-deep row) or 8 bf16 values (4 threads per
  // row; the bytes go to LDS as they are: the image of a bf16 row IS the tensor's row)
  constexpr int KG = ES == 2 ? 4 : 8;
  constexpr int AP = BM * KG / NTHR;    // 16-byte loads of the activation tile per thread and K-step
  constexpr int BPP = (BN * 4 + NTHR - 1) / NTHR;   // 16-byte loads per weight plane per thread and K-step
  constexpr int PA = BM * 64, PB = BN * 64, STAGE = NP * (PA + PB);   // NP planes per operand and stage
  constexpr int SMEM = NBUF * STAGE >= WM * 32 * BN * 4 ? NBUF * STAGE : WM * 32 * BN * 4;   // K-loop stages, then the epilogue's staging
  static_assert(AP >= 1 && SMEM <= 160 * 1024, "tile / epilogue staging do not fit");
  __shared__ __attribute__((aligned(16))) unsigned char smem_b[SMEM];
  float* const smem = reinterpret_cast<float*>(smem_b);

  BDV_STAMP(0);
  const int nk = g.Ktot / BK;
  const WorkItem it = get_work(blockIdx.x, wk, nk);
  const int mt = it.tile / NT, nt = it.tile - mt * NT;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int arow = tid / KG, kg = tid % KG;
  const int brow = tid >> 2, bc = tid & 3;
  const int HoWo = g.Ho * g.Wo;
  static_assert(ES == 4 || (NP == 1 && !PRE), "bf16 storage: single-product arithmetic only");
  const int frame_bytes = g.H * g.W * g.Cin * ES;
  const int plane_bytes = g.Cout * g.Ktot * 2;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.N * frame_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, 3 * plane_bytes, 0x00020000);

  int a_base[AP], a_t[AP], a_hi0[AP], a_wi0[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + (NTHR / KG) * p;
    const bool ok = m < g.M;
    int n, rem, ho, wo;
    fast_divmod(ok ? m : 0, HoWo, g.rcp_HoWo, n, rem);
    fast_divmod(rem, g.Wo, g.rcp_Wo, ho, wo);
    a_t[p] = n % g.T;
    a_hi0[p] = ok ? ho * g.stride - g.pad : -(1 << 20);  // rows past M fail every bounds test
    a_wi0[p] = wo * g.stride - g.pad_w;
    a_base[p] = ((n * g.H + ho * g.stride - g.pad) * g.W + wo * g.stride - g.pad_w) * g.Cin * ES + 16 * kg;
  }
  int b_base[BPP];
  const bool b_active = BN * 4 >= NTHR || tid < BN * 4;   // a 64-row weight tile is loaded by the first four waves only
#pragma unroll
  for (int q = 0; q < BPP; ++q) b_base[q] = b_active ? (nt * BN + brow + (NTHR / 4) * q) * 64 + 16 * bc : kOOB;

  // K index state (uniform): K-step kt = chunk * R*S + r * S + s; the weight planes are stored in this order
  const int RS = g.R * g.S;
  int chunk = it.kb / RS, r, s, kt_w = it.kb;
  {
    const int tap = it.kb - chunk * RS;
    r = tap / g.S;
    s = tap - r * g.S;
  }

  constexpr int NSET = NBUF;           // two stages: two register sets (pl_pipeline2)
  float4 ra[NSET][AP];
  u32x4 rb[NSET][NP * BPP];
  // PRE: the activation operand is the RAW output of the producing conv; its BatchNorm + ReLU (scale / shift of this thread's 4
  // K-channels) is applied between the load and the split.  Halo / clip-end / ragged lanes must stay zero: their validity travels
  // with the register set (pvalid).
  float4 psc[PRE ? NSET : 1], psh[PRE ? NSET : 1];
  unsigned pvalid[PRE ? NSET : 1];
  auto load = [&](auto set) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    if constexpr (PRE) {
      psc[SET] = *reinterpret_cast<const float4*>(epi.pre_scale + chunk * BK + 4 * kg);   // (PRE: fp32 tensors only, KG = 8)
      psh[SET] = *reinterpret_cast<const float4*>(epi.pre_shift + chunk * BK + 4 * kg);
      pvalid[SET] = 0u;
    }
    const int cls = shift_class(chunk * BK + (BK / KG) * kg, g.fold);
    const int koff_a = ((r * g.W + s) * g.Cin + chunk * BK) * ES;
    const int koff_b = kt_w * g.Cout * 64;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool v = (unsigned)(a_hi0[p] + r) < (unsigned)g.H && (unsigned)(a_wi0[p] + s) < (unsigned)g.W &&
                     (unsigned)(a_t[p] + cls) < (unsigned)g.T;
      ra[SET][p] = buf_load16(xr, (a_base[p] + koff_a + cls * frame_bytes) | (v ? 0 : kOOB), 0);   // ES = 2: eight bf16, as raw bits
      if constexpr (PRE) pvalid[SET] |= (v ? 1u : 0u) << p;
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
      for (int q = 0; q < BPP; ++q) rb[SET][pl * BPP + q] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_base[q] + pl * plane_bytes, koff_b, 0);
    kt_w += 1;
    s += 1;
    const int ws_ = (s == g.S) ? 1 : 0;
    s = ws_ ? 0 : s;
    r += ws_;
    const int wr_ = (r == g.R) ? 1 : 0;
    r = wr_ ? 0 : r;
    chunk += wr_;
  };
  // LDS offsets of this thread: stores (rows arow + 64 p / brow + 128 q keep (row >> 2) & 3) and fragment reads
  const int st_a = ES == 2 ? pl_off(arow, kg) : pl_off(arow, kg >> 1) + 8 * (kg & 1), st_b = pl_off(brow, bc);
  const int fa[2] = {pl_frag_off(32 * wm, 0, lane), pl_frag_off(32 * wm, 1, lane)};
  const int fb[2] = {pl_frag_off(32 * wn, 0, lane), pl_frag_off(32 * wn, 1, lane)};
  auto store = [&](int stage, auto set) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    unsigned char* const As = smem_b + stage * STAGE;
    unsigned char* const Bs = As + NP * PA;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      float4 v = ra[SET][p];
      if constexpr (PRE) {   // the same expression as bn_apply_kernel (fused multiply-add, then max): bit-identical activations
        const bool ok = (pvalid[SET] >> p) & 1u;
        v.x = ok ? fmaxf(fmaf(v.x, psc[SET].x, psh[SET].x), 0.f) : 0.f;
        v.y = ok ? fmaxf(fmaf(v.y, psc[SET].y, psh[SET].y), 0.f) : 0.f;
        v.z = ok ? fmaxf(fmaf(v.z, psc[SET].z, psh[SET].z), 0.f) : 0.f;
        v.w = ok ? fmaxf(fmaf(v.w, psc[SET].w, psh[SET].w), 0.f) : 0.f;
      }
      if constexpr (ES == 2) *reinterpret_cast<float4*>(As + st_a + (NTHR / KG) * p * 64) = v;
      else pl_split_store_at<NP>(As + st_a + (NTHR / KG) * p * 64, PA, v);
    }
    if (b_active) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int q = 0; q < BPP; ++q) *reinterpret_cast<u32x4*>(Bs + st_b + (pl * PB + (NTHR / 4) * q * 64)) = rb[SET][pl * BPP + q];
    }
  };
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  if constexpr (NBUF == 2) {
    pl_pipeline2(it.ke - it.kb, load, store, [&](int stage) __attribute__((always_inline)) {
      mma_stage_pl<BM, BN, WM, WN, NP>(smem_b + stage * STAGE, smem_b + stage * STAGE + NP * PA, acc, fa, fb);
    }, g.stagger != 0 && wave >= WM * WN / 2);
  } else {
    load(PlSet0{});
    for (int kt = it.kb; kt < it.ke; ++kt) {
      __syncthreads();
      store(0, PlSet0{});
      __syncthreads();
      if (kt == it.kb) BDV_STAMP(1);
      if (kt + 1 < it.ke) load(PlSet0{});
      mma_stage_pl<BM, BN, WM, WN, NP>(smem_b, smem_b + NP * PA, acc, fa, fb);
    }
  }
  BDV_STAMP(2);

  if (it.pslot >= 0) {
    store_partial<TM, TN, NTHR>(slab, it.pslot, acc, tid);
    return;
  }
  fprop_epilogue<BM, BN, WM, WN, ES>(acc, smem, y, g, epi, mt, nt, tid);
  BDV_STAMP(3);
  BDV_STAMP(4);
}

// ---- dgrad ------------------------------------------------------------------------------------
// dp = the D planes of bdv_conv_split_weights: B rows = input channels ci, contraction over (tap, co) with co contiguous.
template <int BM, int BN, int WM, int WN, int NBUF, int NP = 3, int ES = 4>
__global__ __launch_bounds__(64 * WM * WN, BM == 64 ? 4 : pl_waves_per_simd(WM * WN, NP, BN == 64)) void conv_dgrad_pl_kernel(const void* __restrict__ dy,
                                                                                      const unsigned short* __restrict__ dp,
                                                                                      void* __restrict__ dx,
                                                                                      const void* __restrict__ add_src,
                                                                                      const uint32_t* __restrict__ add_mask, Geom g,
                                                                                      int NT, Work wk, float* __restrict__ slab,
                                                                                      BnStat stat) {
  constexpr int NTHR = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int KG = ES == 2 ? 4 : 8;   // threads per 32-deep row of the activation tile: 16 bytes each (conv_fprop_pl_kernel)
  constexpr int AP = BM * KG / NTHR;
  constexpr int BPP = (BN * 4 + NTHR - 1) / NTHR;
  constexpr int PA = BM * 64, PB = BN * 64, STAGE = NP * (PA + PB);   // NP planes per operand and stage
  constexpr int SMEM = NBUF * STAGE >= WM * 32 * BN * 4 ? NBUF * STAGE : WM * 32 * BN * 4;   // K-loop stages, then the epilogue's staging
  static_assert(AP >= 1 && SMEM <= 160 * 1024, "tile / epilogue staging do not fit");
  __shared__ __attribute__((aligned(16))) unsigned char smem_b[SMEM];
  float* const smem = reinterpret_cast<float*>(smem_b);

  BDV_STAMP(0);
  // parity class of the input pixel (stride 1: a single class)
  const int st = g.stride;
  const int ph = blockIdx.y / st, pw = blockIdx.y - ph * st;
  const int Hc = (g.H - ph + st - 1) / st, Wc = (g.W - pw + st - 1) / st;
  const int Mc = g.N * Hc * Wc;
  const int MT = (Mc + BM - 1) / BM;
  const int r0 = (ph + g.pad) % st, s0 = (pw + g.pad_w) % st;
  const int nr = r0 < g.R ? (g.R - r0 + st - 1) / st : 0;
  const int ns = s0 < g.S ? (g.S - s0 + st - 1) / st : 0;
  const int bh = (ph + g.pad - r0) / st, bw = (pw + g.pad_w - s0) / st;
  const int ntap = nr * ns;
  const int nk = ntap * g.Cout / BK;  // 0 for a class no filter tap reaches (e.g. 1x1 stride 2, odd pixels)

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
    it.tile = 0;
    it.kb = 0;
    it.ke = nk;
    it.pslot = -1;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int arow = tid / KG, kg = tid % KG;
  const int brow = tid >> 2, bc = tid & 3;
  const int HcWc = Hc * Wc;
  const int RS = g.R * g.S;
  const int plane_bytes = g.Cout * RS * g.Cin * 2;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.N * g.Ho * g.Wo * g.Cout * ES, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)dp, 0, 3 * plane_bytes, 0x00020000);

  int a_base[AP], a_h[AP], a_w[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = mt * BM + arow + (NTHR / KG) * p;
    const bool ok = m < Mc;
    const int mm = ok ? m : 0;
    const int n = mm / HcWc;
    const int rem = mm - n * HcWc;
    const int hc = rem / Wc, wc = rem - hc * Wc;
    a_h[p] = ok ? hc + bh : -(1 << 20);
    a_w[p] = wc + bw;
    a_base[p] = ((n * g.Ho + hc + bh) * g.Wo + wc + bw) * g.Cout * ES + 16 * kg;
  }
  int b_base[BPP];
  const bool b_active = BN * 4 >= NTHR || tid < BN * 4;   // a 64-row weight tile is loaded by the first four waves only
#pragma unroll
  for (int q = 0; q < BPP; ++q) b_base[q] = b_active ? (nt * BN + brow + (NTHR / 4) * q) * 64 + 16 * bc : kOOB;

  // K index state (uniform): kt = chunk * ntap + ir * ns + is  (tap-fastest)
  int chunk = ntap > 0 ? it.kb / ntap : 0, ir, is;
  {
    const int ct = ntap > 0 ? it.kb - chunk * ntap : 0;
    ir = ns > 0 ? ct / ns : 0;
    is = ct - ir * ns;
  }
  const int nchunk = g.Cout / BK;

  constexpr int NSET = NBUF;
  float4 ra[NSET][AP];
  u32x4 rb[NSET][NP * BPP];
  auto load = [&](auto set) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const int tap = (r0 + ir * st) * g.S + (s0 + is * st);
    const int koff_a = (chunk * BK - (ir * g.Wo + is) * g.Cout) * ES;
    const int koff_b = (tap * nchunk + chunk) * g.Cin * 64;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool v = (unsigned)(a_h[p] - ir) < (unsigned)g.Ho && (unsigned)(a_w[p] - is) < (unsigned)g.Wo;
      ra[SET][p] = buf_load16(yr, (a_base[p] + koff_a) | (v ? 0 : kOOB), 0);   // ES = 2: eight bf16, as raw bits
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
      for (int q = 0; q < BPP; ++q) rb[SET][pl * BPP + q] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_base[q] + pl * plane_bytes, koff_b, 0);
    is += 1;
    const int w1 = (is == ns) ? 1 : 0;
    is = w1 ? 0 : is;
    ir += w1;
    const int w2 = (ir == nr) ? 1 : 0;
    ir = w2 ? 0 : ir;
    chunk += w2;
  };
  // LDS offsets of this thread: stores (rows arow + 64 p / brow + 128 q keep (row >> 2) & 3) and fragment reads
  const int st_a = ES == 2 ? pl_off(arow, kg) : pl_off(arow, kg >> 1) + 8 * (kg & 1), st_b = pl_off(brow, bc);
  const int fa[2] = {pl_frag_off(32 * wm, 0, lane), pl_frag_off(32 * wm, 1, lane)};
  const int fb[2] = {pl_frag_off(32 * wn, 0, lane), pl_frag_off(32 * wn, 1, lane)};
  auto store = [&](int stage, auto set) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    unsigned char* const As = smem_b + stage * STAGE;
    unsigned char* const Bs = As + NP * PA;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      if constexpr (ES == 2) *reinterpret_cast<float4*>(As + st_a + (NTHR / KG) * p * 64) = ra[SET][p];
      else pl_split_store_at<NP>(As + st_a + (NTHR / KG) * p * 64, PA, ra[SET][p]);
    }
    if (b_active) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int q = 0; q < BPP; ++q) *reinterpret_cast<u32x4*>(Bs + st_b + (pl * PB + (NTHR / 4) * q * 64)) = rb[SET][pl * BPP + q];
    }
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  if (it.ke > it.kb) {
    if constexpr (NBUF == 2) {
      pl_pipeline2(it.ke - it.kb, load, store, [&](int stage) __attribute__((always_inline)) {
        mma_stage_pl<BM, BN, WM, WN, NP>(smem_b + stage * STAGE, smem_b + stage * STAGE + NP * PA, acc, fa, fb);
      }, g.stagger != 0 && wave >= WM * WN / 2);
    } else {
      load(PlSet0{});
      for (int kt = it.kb; kt < it.ke; ++kt) {
        __syncthreads();
        store(0, PlSet0{});
        __syncthreads();
        if (kt == it.kb) BDV_STAMP(1);
        if (kt + 1 < it.ke) load(PlSet0{});
        mma_stage_pl<BM, BN, WM, WN, NP>(smem_b, smem_b + NP * PA, acc, fa, fb);
      }
    }
  }
  BDV_STAMP(2);
  if (it.pslot >= 0) {
    store_partial<TM, TN, NTHR>(slab, it.pslot, acc, tid);
    return;
  }
  constexpr bool EPI_PIPE = TM * TN <= 4 && NP >= 2 && BM != 64;   // 64 accumulator registers, two waves per SIMD
  dgrad_epilogue<BM, BN, WM, WN, true, ES, EPI_PIPE>(acc, smem, tid, dx, add_src, add_mask, g, mt, nt, Mc, stat, [&](int mrow) {
    if (st == 1) return mrow;
    const int n = mrow / HcWc;
    const int rem = mrow - n * HcWc;
    const int hc = rem / Wc;
    return (n * g.H + hc * st + ph) * g.W + (rem - hc * Wc) * st + pw;
  });
  BDV_STAMP(3);
  BDV_STAMP(4);
}

// ---- wgrad, "P" family ---------------------------------------------------------------------------
// dw tile BM (output channels) x BN (input channels of one filter tap), contraction over output pixels in steps of 32.
// Both operands arrive with the contraction index as the ROW index (a pixel's channels are contiguous), i.e. transposed
// with respect to what an MFMA operand fragment wants (8 consecutive k of one channel).  The planes are therefore kept in
// LDS as they arrive, [32 pixels][channels] bf16 with a pitch of 2 * channels + 64 bytes, and the fragments are fetched with
// ds_read_b64_tr_b16 (a 4 x 16 block read column-major per 16 lanes: two of them give a lane its 8 k-values).  With the
// 64-byte skew the four rows of a block fall on distinct banks: reads and the 8-byte piece stores are conflict-free.
// Eight waves (2 x 4), one workgroup per CU, one LDS stage.
typedef short s16x4_t __attribute__((ext_vector_type(4)));

template <int COLS, int NP = 3, int ROWS = 32>
__device__ __forceinline__ void pl_store_rows(unsigned char* __restrict__ base, int krow, int c4, const float4 v) {
  constexpr int PITCH = 2 * COLS + 64, PLANE = ROWS * PITCH;
  if constexpr (NP == 1) {
    *reinterpret_cast<u32x2_t*>(base + krow * PITCH + 8 * c4) = bf16_hi_pairs(v);
    return;
  }
  u32x2_t hi, mid, lo;
  split3_pairs(v, hi, mid, lo);
  unsigned char* q = base + krow * PITCH + 8 * c4;
  *reinterpret_cast<u32x2_t*>(q) = hi;
  *reinterpret_cast<u32x2_t*>(q + PLANE) = mid;
  if constexpr (NP == 3) *reinterpret_cast<u32x2_t*>(q + 2 * PLANE) = lo;
}

// 8 k-values (pixels 16 s + 8 h .. + 7) of channel `cb + (lane & 31)` of plane image `p`
template <int COLS>
__device__ __forceinline__ bf16x8_t pl_frag_tr(const unsigned char* __restrict__ plane, int cb, int s, int lane) {
  constexpr int PITCH = 2 * COLS + 64;
  const int h = lane >> 5, q = (lane & 15) >> 2;
  const unsigned char* a = plane + (16 * s + 8 * h + q) * PITCH + 2 * (cb + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
  const s16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
  const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * PITCH));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const s16x8_t f = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, f);
}

// MTAP: the BN columns of a tile span BN / Cin whole filter taps (64-channel layers: a 3x3 filter row = 192 columns).
// WM x WN waves; the 4-wave forms (64-wide tiles, 60-72 KB of LDS) run two workgroups per CU.
// KW = pixels per LDS stage.  32: one stage, two barriers per step (load -> barrier -> split + store -> barrier -> MFMAs).
// 16: two stages of 16 pixels in the same LDS and two register sets, pl_pipeline2: the split + store of the next 16 pixels and the
// loads of the ones after are issued among the MFMAs of the current stage, one barrier per 16 pixels -- used for the 128 x 256 and
// 256 x 128 tiles (0.349 -> 0.315 ms on the 256 -> 128 site, 0.181 -> 0.162 on 128 <-> 512); the 256 x 256 tile has no registers
// left for a second set.
template <int BM, int BN, int WM, int WN, bool INCR, int NP = 3, bool MTAP = false, int KW = 32, int ES = 4>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_wgrad_pl_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                                         float* __restrict__ slab, Geom g, int MTw, int NTw,
                                                                         int kt_per_split, const float* __restrict__ pre_scale,
                                                                         const float* __restrict__ pre_shift) {
  constexpr int NTHR = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  // a thread's unit of either operand is 16 bytes of one pixel's channels: 4 fp32 or 8 bf16 values (bf16: stored to LDS as they are)
  constexpr int GV = ES == 2 ? 8 : 4;
  static_assert(TM >= 1 && TN >= 1 && (32 * (BM / GV)) % NTHR == 0 && (32 * (BN / GV)) % NTHR == 0, "tile / thread mapping");
  static_assert(ES == 4 || NP == 1, "bf16 storage: single-product arithmetic (and Cin % 8 == 0: a 16-byte unit stays inside a filter tap)");
  constexpr int AV = BM / GV, BV = BN / GV;
  static_assert(KW == 32 || KW == 16, "pixels per stage");
  constexpr int NST = 32 / KW;          // LDS stages = register sets
  constexpr int AP = KW * AV / NTHR, BP = KW * BV / NTHR;
  static_assert(AP >= 1 && BP >= 1 && (KW * AV) % NTHR == 0 && (KW * BV) % NTHR == 0, "stage / thread mapping");
  constexpr int PITCH_A = 2 * BM + 64, PITCH_B = 2 * BN + 64;
  constexpr int PLANE_A = KW * PITCH_A, PLANE_B = KW * PITCH_B;
  constexpr int STAGE = 3 * (PLANE_A + PLANE_B);
  __shared__ __attribute__((aligned(16))) unsigned char smem_b[NST * STAGE];
  // pre_scale != null: x is the producer's RAW conv output; its BatchNorm + ReLU (per input channel) is applied between the load
  // and the split (the same fused multiply-add as bn_apply_kernel).  The BN columns' scale / shift sit in LDS; halo / ragged
  // lanes must stay zero, so their validity travels with the register set.
  __shared__ __attribute__((aligned(16))) float pre_lds[2][BN];
  const bool pre = pre_scale != nullptr;

  // slice-major, XCD-contiguous work order: all tiles of one K slice read the same rows of dy and x (see conv_wgrad_kernel)
  const int tiles = MTw * NTw;
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int split = wi / tiles;
  const int tile = wi - split * tiles;
  const int mt = tile % MTw, nt = tile / MTw;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int HoWo = g.Ho * g.Wo;
  const int frame_bytes = g.H * g.W * g.Cin * ES;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.N * g.st_t * frame_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.M * g.Cout * ES, 0x00020000);

  // A: dy rows m0 + krow, columns mt*BM + 4*c4
  int a_off[AP], a_krow[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int idx = tid + NTHR * p;
    a_krow[p] = idx / AV;
    a_off[p] = (a_krow[p] * g.Cout + mt * BM + GV * (idx % AV)) * ES;
  }
  // B: GEMM column nt * BN + j = (tap, ci) with ci fastest; rows = input pixels of the tap
  int b_krow[BP], b_off[BP], b_cls[BP], b_r[BP], b_s[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int idx = tid + NTHR * p;
    b_krow[p] = idx / BV;
    const int col = nt * BN + GV * (idx % BV);
    const int tap = MTAP ? col / g.Cin : (nt * BN) / g.Cin;   // one tap per tile unless MTAP; tap = (dt * R + r) * S + s
    const int ci = col - tap * g.Cin;
    const int dt = MTAP ? tap / (g.R * g.S) : 0, rs = tap - dt * g.R * g.S;
    b_r[p] = (!MTAP || tap < g.Rt * g.R * g.S) ? rs / g.S - g.pad : -(1 << 20);   // columns past the last tap (stem: 49 taps in 64) load zeros
    b_s[p] = rs % g.S - g.pad_w;
    // frame offset of the tap in input frames: the temporal shift's -1 / 0 / +1, or the stem's temporal tap dt - Rt / 2
    b_cls[p] = shift_class(ci, g.fold) + (MTAP ? dt - g.Rt / 2 : 0);
    b_off[p] = ((b_r[p] * g.W + b_s[p]) * g.Cin + ci) * ES + b_cls[p] * frame_bytes;
  }

  if (pre) {
    for (int j = tid; j < BN; j += NTHR) {
      const int col = nt * BN + j;
      const int tapj = MTAP ? col / g.Cin : (nt * BN) / g.Cin;
      const int cj = col - tapj * g.Cin;
      pre_lds[0][j] = pre_scale[cj];
      pre_lds[1][j] = pre_shift[cj];
    }
    __syncthreads();
  }
  // kt counts KW-pixel steps (kt_per_split is given in 32-pixel steps)
  const int kt_begin = split * kt_per_split * NST;
  const int nkt_all = (g.M + KW - 1) / KW;
  const int kt_end = min(kt_begin + kt_per_split * NST, nkt_all);

  // incremental pixel state per B row (see conv_wgrad_kernel): no division in the loop
  const int st = g.stride;
  const int d_ho = KW / g.Wo, d_wo = KW - d_ho * g.Wo;
  const int wrap_w = g.Wo * st, wrap_h = g.Ho * st;
  const int px = g.Cin * ES;
  const int inc0 = (d_ho * st * g.W + d_wo * st) * px;
  const int inc1 = (st * g.W - wrap_w) * px;
  const int inc2 = (g.st_t * g.H * g.W - wrap_h * g.W) * px;   // to the next output frame = st_t input frames on
  int s_hi[BP], s_wi[BP], s_t[BP], s_off[BP];
  if (INCR) {
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int m = kt_begin * KW + b_krow[p];
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
      s_hi[p] = ho * st;
      s_wi[p] = wo * st;
      s_t[p] = n % g.T;
      s_off[p] = ((n * g.st_t * g.H + s_hi[p]) * g.W + s_wi[p]) * px + b_off[p];
    }
  }

  float4 ra[NST][AP], rb[NST][BP];
  unsigned bvalid[NST];
  int kt_load = kt_begin;           // load(set) fetches the next step in order
  auto load = [&](auto set) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const int m0 = kt_load * KW;
    kt_load += 1;
    bvalid[SET] = 0u;
#pragma unroll
    for (int p = 0; p < AP; ++p)  // rows past M lie past num_records: zeros
      ra[SET][p] = buf_load16(yr, a_off[p] + m0 * g.Cout * ES, 0);
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int m = m0 + b_krow[p];
      const bool mok = m < g.M;
      int hi, wi_, t, off;
      if (INCR) {
        hi = s_hi[p];
        wi_ = s_wi[p];
        t = s_t[p];
        off = s_off[p];
        const int w2 = wi_ + d_wo * st;
        const bool c1 = w2 >= wrap_w;
        s_wi[p] = c1 ? w2 - wrap_w : w2;
        const int h2 = hi + d_ho * st + (c1 ? st : 0);
        const bool c2 = h2 >= wrap_h;
        s_hi[p] = c2 ? h2 - wrap_h : h2;
        const int t2 = t + (c2 ? 1 : 0);
        s_t[p] = t2 == g.T ? 0 : t2;
        s_off[p] = off + inc0 + (c1 ? inc1 : 0) + (c2 ? inc2 : 0);
      } else {
        const int mm = mok ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        const int ho = rem / g.Wo;
        hi = ho * st;
        wi_ = (rem - ho * g.Wo) * st;
        t = n % g.T;
        off = ((n * g.st_t * g.H + hi) * g.W + wi_) * px + b_off[p];
      }
      const bool v = mok && (unsigned)(hi + b_r[p]) < (unsigned)g.H && (unsigned)(wi_ + b_s[p]) < (unsigned)g.W &&
                     (unsigned)(t * g.st_t + b_cls[p]) < (unsigned)(g.T * g.st_t);
      rb[SET][p] = buf_load16(xr, off | (v ? 0 : kOOB), 0);
      bvalid[SET] |= (v ? 1u : 0u) << p;
    }
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  auto store = [&](int stage, auto set) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    unsigned char* const As = smem_b + stage * STAGE;
    unsigned char* const Bs = As + 3 * PLANE_A;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      if constexpr (ES == 2) *reinterpret_cast<float4*>(As + a_krow[p] * PITCH_A + 16 * ((tid + NTHR * p) % AV)) = ra[SET][p];
      else pl_store_rows<BM, NP, KW>(As, a_krow[p], (tid + NTHR * p) % AV, ra[SET][p]);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int c4 = (tid + NTHR * p) % BV;
      float4 v = rb[SET][p];
      if (pre) {
        const float4 sc = *reinterpret_cast<const float4*>(&pre_lds[0][4 * c4]), sh = *reinterpret_cast<const float4*>(&pre_lds[1][4 * c4]);
        const bool ok = (bvalid[SET] >> p) & 1u;
        v.x = ok ? fmaxf(fmaf(v.x, sc.x, sh.x), 0.f) : 0.f;
        v.y = ok ? fmaxf(fmaf(v.y, sc.y, sh.y), 0.f) : 0.f;
        v.z = ok ? fmaxf(fmaf(v.z, sc.z, sh.z), 0.f) : 0.f;
        v.w = ok ? fmaxf(fmaf(v.w, sc.w, sh.w), 0.f) : 0.f;
      }
      if constexpr (ES == 2) *reinterpret_cast<float4*>(Bs + b_krow[p] * PITCH_B + 16 * c4) = v;
      else pl_store_rows<BN, NP, KW>(Bs, b_krow[p], c4, v);
    }
  };
  auto mma = [&](int stage) __attribute__((always_inline)) {
    const unsigned char* const As = smem_b + stage * STAGE;
    const unsigned char* const Bs = As + 3 * PLANE_A;
#pragma unroll
    for (int s = 0; s < KW / 16; ++s) {
      // the column fragments of the 16-deep step stay in registers, the row fragments are fetched one 32-row tile at a time
      // (all of them at once are 72 VGPRs for a 256 x 256 tile: with two register sets of loads that spills)
      bf16x8_t b[NP][TN];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int j = 0; j < TN; ++j) b[p][j] = pl_frag_tr<BN>(Bs + p * PLANE_B, 32 * WN * j + 32 * wn, s, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        bf16x8_t a[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = pl_frag_tr<BM>(As + p * PLANE_A, 32 * WM * i + 32 * wm, s, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (NP == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0][j], acc[i][j], 0, 0, 0);
          }
          if constexpr (NP >= 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0][j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };
  if (kt_begin < kt_end) {
    if constexpr (NST == 2) {
      pl_pipeline2(kt_end - kt_begin, load, store, mma, g.stagger != 0 && (int)(threadIdx.x >> 6) >= WM * WN / 2);
    } else {
      load(PlSet0{});
      for (int kt = kt_begin; kt < kt_end; ++kt) {
        __syncthreads();
        store(0, PlSet0{});
        __syncthreads();
        if (kt + 1 < kt_end) load(PlSet0{});
        mma(0);
      }
    }
  }

  float* out = slab + (size_t)split * g.Cout * g.Ktot;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = nt * BN + 32 * WN * j + 32 * wn + (lane & 31);
    if (MTAP && col >= g.Ktot) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mt * BM + 32 * WM * i + 32 * wm + acc_row(e, lane);
        out[(size_t)row * g.Ktot + col] = acc[i][j][e];
      }
  }
}

// ---- host side --------------------------------------------------------------------------------

int check_geom(const bdv_conv_geom* g, const char* who) {
  BDV_REQUIRE(g != nullptr, "%s: geom is NULL", who);
  BDV_REQUIRE(g->N > 0 && g->H > 0 && g->W > 0 && g->Ho > 0 && g->Wo > 0, "%s: non-positive extent", who);
  BDV_REQUIRE(g->Cin > 0 && g->Cin % 4 == 0, "%s: Cin=%d must be a positive multiple of 4", who, g->Cin);
  BDV_REQUIRE(g->Cout > 0 && g->Cout % 64 == 0, "%s: Cout=%d must be a multiple of 64", who, g->Cout);
  BDV_REQUIRE(g->R > 0 && g->S > 0 && g->pad >= 0, "%s: bad filter", who);
  BDV_REQUIRE(g->stride == 1 || g->stride == 2, "%s: stride %d unsupported", who, g->stride);
  const int pad_w = g->pad_w < 0 ? g->pad : g->pad_w;
  BDV_REQUIRE(g->Ho == (g->H + 2 * g->pad - g->R) / g->stride + 1 && g->Wo == (g->W + 2 * pad_w - g->S) / g->stride + 1,
              "%s: Ho/Wo inconsistent with H/W/pad/stride", who);
  BDV_REQUIRE(g->Cin % BK == 0 || g->Cin == 4, "%s: Cin=%d must be a multiple of 32 (or exactly 4)", who, g->Cin);
  if (g->Rt > 1) {
    BDV_REQUIRE(g->Cin == 4 && g->fold == 0, "%s: temporal taps (Rt=%d) are implemented for the stem only (Cin = 4, no shift)", who, g->Rt);
    BDV_REQUIRE((g->st_t == 1 || g->st_t == 2) && g->T > 0 && g->N % g->T == 0, "%s: Rt=%d needs st_t 1|2 and N %% T == 0 (T = output frames per clip)", who, g->Rt);
    BDV_REQUIRE((int64_t)g->N * g->st_t * g->H * g->W * g->Cin < (1ll << 29), "%s: input exceeds 2^29 elements", who);
  }
  if (g->fold > 0) {
    BDV_REQUIRE(g->fold % 4 == 0 && 2 * g->fold <= g->Cin, "%s: fold=%d must be a multiple of 4 and <= Cin/2", who,
                g->fold);
    BDV_REQUIRE(g->T > 0 && g->N % g->T == 0, "%s: N=%d not a multiple of T=%d", who, g->N, g->T);
    BDV_REQUIRE(g->Cin % BK == 0, "%s: temporal shift needs Cin %% 32 == 0", who);
  }
  // byte offsets are 32-bit with the top bit reserved as the out-of-range marker
  BDV_REQUIRE((int64_t)g->N * g->H * g->W * g->Cin < (1ll << 29) && (int64_t)g->N * g->Ho * g->Wo * g->Cout < (1ll << 29),
              "%s: tensor exceeds 2^29 elements", who);
  BDV_REQUIRE((int64_t)g->Cout * g->R * g->S * g->Cin < (1ll << 29), "%s: weight tensor too large", who);
  // fast_divmod (float reciprocal) is exact below 2^22 pixels; the Cin = 4 kernels use integer division instead
  BDV_REQUIRE((int64_t)g->N * g->Ho * g->Wo < (1ll << 22) || g->Cin == 4, "%s: more than 2^22 output pixels", who);
  return BDV_OK;
}

// BDVCIL_STAGGER=0 turns the wave stagger of pl_pipeline2 off (A/B runs); read per call, so that tools can flip it between rounds
static bool pl_stagger_enabled() {
  const char* e = getenv("BDVCIL_STAGGER");
  return e == nullptr || atoi(e) != 0;
}

Geom make_geom(const bdv_conv_geom* g) {
  Geom d;
  d.stagger = pl_stagger_enabled() ? 1 : 0;
  d.N = g->N; d.H = g->H; d.W = g->W; d.Cin = g->Cin; d.Ho = g->Ho; d.Wo = g->Wo; d.Cout = g->Cout;
  d.R = g->R; d.S = g->S; d.stride = g->stride; d.pad = g->pad; d.pad_w = g->pad_w < 0 ? g->pad : g->pad_w;
  d.T = (g->fold > 0 || g->Rt > 1) ? g->T : 1;
  d.fold = g->fold;
  d.Rt = g->Rt > 1 ? g->Rt : 1;
  d.st_t = g->Rt > 1 ? g->st_t : 1;
  d.rcp_RS = 1.0f / (float)(g->R * g->S);
  d.rcp_S = 1.0f / (float)g->S;
  d.M = 0; d.Ktot = 0;
  d.rcp_HoWo = 1.0f / (float)(g->Ho * g->Wo);
  d.rcp_Wo = 1.0f / (float)g->Wo;
  return d;
}

bool debug_plan() {
  static const bool on = getenv("BDVCIL_DEBUG_PLAN") != nullptr;
  return on;
}

// Pick the K-split of the remainder tiles.
// Measured on MI355X (tools/ubench): 3 blocks of these kernels are resident per CU, i.e. W = 768 blocks run at a
// time and a CU is MFMA-bound with 2-3 of them; a launch whose block count is just above a multiple of 768 pays
// a whole extra, poorly filled round.  So whole rounds of W tiles run data-parallel and the remaining r < W tiles
// are cut along K into `split` slices each: the tail then consists of many short blocks that fill every CU.
// Cost model (us): one 32-deep K-iteration of a 64-accumulator block = `iter_us` of its CU's matrix pipes; a CU
// holding 1 / 2 / 3 blocks reaches ~60 / 92 / 100 % of that rate; the fix-up costs ~10 us + 2.5 us per slice + its
// bytes at ~8 TB/s.
Work plan_work(int tiles, int nk, double iter_us, size_t seg_bytes, size_t ws_bytes, int per_cu = 3) {
  const int cus = 256, W = per_cu * cus;
  Work wk;
  const int q = tiles / W, r = tiles - q * W;
  wk.dp_tiles = tiles;
  wk.rem_tiles = 0;
  wk.split = 1;
  if (r == 0 || nk < 4) return wk;
  auto tail_time = [&](int blocks, double iters) {  // time of `blocks` equal blocks of `iters` K-iterations
    const int full = blocks / W, rest = blocks - full * W;
    const int c = (rest + cus - 1) / cus;            // blocks per CU in the last, partial round
    const double e = c >= 3 ? 1.0 : (c == 2 ? 0.92 : 0.6);
    return iters * iter_us * (3.0 * full + (c > 0 ? c / e : 0.0));
  };
  double best = 1e30;
  int best_s = 1;
  for (int sp = 1; sp <= 24 && nk / sp >= 2; ++sp) {
    const size_t need = (size_t)r * sp * seg_bytes;
    if (sp > 1 && need > ws_bytes) break;
    double cost = tail_time(r * sp, (double)nk / sp);
    if (sp > 1) cost += 10.0 + 2.5 * sp + (double)need * 2.0 / 8.0e6;
    if (cost < best * 0.95) {  // prefer fewer slices unless clearly better
      best = cost;
      best_s = sp;
    }
  }
  if (best_s > 1) {
    wk.dp_tiles = q * W;
    wk.rem_tiles = r;
    wk.split = best_s;
  }
  return wk;
}


constexpr size_t kMaxSplitWorkspace = 160ull << 20;

struct FdPlan {
  bool wide;  // 128x128 tile (else 128x64)
  int MT, NT, nk;
  size_t seg_bytes;
  Work wk;
};

FdPlan plan_fprop(const Geom& g, size_t ws_bytes, bool x3 = false) {
  FdPlan p;
  p.wide = (g.Cout % 128) == 0;
  const int bn = p.wide ? 128 : 64;
  p.MT = (g.M + 127) / 128;
  p.NT = g.Cout / bn;
  p.nk = (g.Ktot + BK - 1) / BK;
  p.seg_bytes = (size_t)(p.wide ? 64 : 32) * 256 * sizeof(float);
  if (g.Cin % BK != 0) {  // stem kernel: no K split
    p.wk.dp_tiles = p.MT * p.NT;
    p.wk.rem_tiles = 0;
    p.wk.split = 1;
  } else {
    // the bf16-piece kernel: two blocks per CU (60 KB of LDS each), a K-iteration of 48 bf16 MFMAs instead of 64 fp32 ones
    p.wk = (x3 && p.wide) ? plan_work(p.MT * p.NT, p.nk, 0.9, p.seg_bytes, ws_bytes, 2)
                          : plan_work(p.MT * p.NT, p.nk, p.wide ? 1.7 : 0.85, p.seg_bytes, ws_bytes);
  }
  return p;
}

FdPlan plan_dgrad(const Geom& g, size_t ws_bytes, bool x3 = false) {
  FdPlan p;
  p.wide = (g.Cin % 128) == 0;
  const int bn = p.wide ? 128 : 64;
  const int st = g.stride;
  const int Mc0 = g.N * ((g.H + st - 1) / st) * ((g.W + st - 1) / st);  // largest parity class
  p.MT = (Mc0 + 127) / 128;
  p.NT = g.Cin / bn;
  p.nk = g.R * g.S * g.Cout / BK;
  p.seg_bytes = (size_t)(p.wide ? 64 : 32) * 256 * sizeof(float);
  if (st == 1) {
    p.wk = (x3 && p.wide) ? plan_work(p.MT * p.NT, p.nk, 0.9, p.seg_bytes, ws_bytes, 2)
                          : plan_work(p.MT * p.NT, p.nk, p.wide ? 1.7 : 0.85, p.seg_bytes, ws_bytes);
  } else {
    p.wk.dp_tiles = ((p.MT + 7) / 8) * 8 * p.NT;  // padded grid per parity class
    p.wk.rem_tiles = 0;
    p.wk.split = 1;
  }
  return p;
}

// ---- planner of the "P" kernels (8 waves, ONE workgroup per CU: a round is 256 blocks) ----------------------------------
// Tile configurations: 0 = 128 x 256 (two LDS stages), 1 = 256 x 128 (two stages), 2 = 256 x 256 (one stage, opt-in),
// 3 = 256 x 64 (two stages; the 64-channel layers).
struct PlCfg {
  int BM, BN, nbuf;
  double iter_us;  // one 32-deep K-step of a block, measured per-CU rate (tools/ubench/gemm_x3p.hip: ~200 / ~230 TFLOP/s chip-wide)
};
constexpr int kNumPlCfg = 6;
// 4 = 64 x 128, four waves, FOUR workgroups per CU: for dgrads whose epilogue streams as many bytes as their K loop takes MFMA time
// (a block's conv1), so that one tile's epilogue runs beside other tiles' K loops; dgrad only
// 5 = 128 x 128, four waves, TWO workgroups per CU, one LDS stage (48 KB): the tile of the round-1 conv_*_x3 kernels on the plane
// kernels' conflict-free LDS image (64-byte rows, 16-byte chunk XOR, immediate-offset fragment reads) and their weight planes
constexpr PlCfg kPlCfg[kNumPlCfg] = {{128, 256, 2, 2.7}, {256, 128, 2, 2.7}, {256, 256, 1, 4.7}, {256, 64, 2, 1.9}, {64, 128, 1, 0.7},
                                     {128, 128, 1, 1.6}};
constexpr int pl_cfg_wm(int cfg) { return cfg == 0 || cfg == 2 || cfg == 4 || cfg == 5 ? 2 : 4; }
constexpr int pl_cfg_wn(int cfg) { return cfg == 4 || cfg == 5 ? 2 : 8 / pl_cfg_wm(cfg); }
constexpr int pl_cfg_round(int cfg) { return cfg == 5 ? 512 : 256; }   // workgroups resident on the chip at once

struct PlPlan {
  int cfg;  // -1: shape not covered by the P kernels
  int MT, NT, nk;
  size_t seg_bytes;
  Work wk;
  double est_us;
};

// DP whole rounds of 256 blocks, the remainder tiles K-split `s` ways (fix-up kernel): cost in us incl. the fix-up's slab traffic
Work plan_work_pl(int tiles, int nk, double iter_us, size_t seg_bytes, size_t ws_bytes, double* est_us, int W = 256) {
  Work wk;
  const int q = tiles / W, r = tiles - q * W;
  wk.dp_tiles = tiles;
  wk.rem_tiles = 0;
  wk.split = 1;
  double best = (double)(q + (r > 0 ? 1 : 0)) * nk * iter_us;
  if (r > 0 && nk >= 4) {
    for (int sp = 2; sp <= 16 && nk / sp >= 2; ++sp) {
      const size_t need = (size_t)r * sp * seg_bytes;
      if (need > ws_bytes) break;
      const int rounds = (r * sp + W - 1) / W;
      const int per = (nk + sp - 1) / sp;
      const double cost = ((double)q * nk + (double)rounds * per) * iter_us + 8.0 + 2.0 * (double)need / 4.0e6;
      if (cost < best * 0.97) {
        best = cost;
        wk.dp_tiles = q * W;
        wk.rem_tiles = r;
        wk.split = sp;
      }
    }
  }
  *est_us = best;
  return wk;
}

// Tile override for tests and A/B runs: BDVCIL_PL_TILE (read once) or bdv_conv_debug_force_tile(); -1 = choose by the cost model.
int g_pl_tile_forced = -2;  // -2: not initialised
int pl_tile_override() {
  if (g_pl_tile_forced == -2) g_pl_tile_forced = getenv("BDVCIL_PL_TILE") ? atoi(getenv("BDVCIL_PL_TILE")) : -1;
  return g_pl_tile_forced;
}

// M = GEMM rows, ncols = GEMM columns (Cout for fprop, Cin for dgrad), nk = 32-deep K-steps; ksplit_ok: stride-1 launches only
// Which kernel family runs a site.  Rules read off tools/tune_conv.py (every choice timed per TSM-R50 site in one process, with
// the training step's fused epilogues: profiles/r02_tune_conv_fused.txt; plain calls: profiles/r02_tune_conv.txt); differences
// between neighbouring choices are a few per cent, the rules keep the clear ones:
//   * 64 output columns: 256x64 tiles for 3x3 filters (compute-bound: +16 %), the fp32-MFMA 64-wide kernels for 1x1 (HBM-bound);
//   * 128 output columns: 256x128 tiles when K >= 512 (fprop) / for stride-1 3x3 filters (dgrad), else the two-workgroups-per-CU
//     kernels (conv_*_x3_kernel): eight or fewer K-steps cannot hide an 8-wave workgroup's prologue and epilogue;
//   * multiples of 256 columns, fprop: 128x256 tiles (two stages, pl_pipeline2) for K >= 512 or K = 64, else the x3 kernels;
//   * multiples of 256 columns, dgrad: 256x256 tiles, except 128x256 for stride-2 3x3 filters (four parity classes with a quarter
//     of the taps each: more, smaller tiles) and for 512 columns with K >= 2048; the x3 kernels for short K with >= 1024 columns
//     and for a block's conv1 (1x1 with the temporal shift): in the training step its epilogue adds the identity-branch gradient
//     and takes the BatchNorm-backward statistics, i.e. it streams three tensors of the output's size and is HBM-bound; two
//     workgroups per CU overlap one tile's epilogue with another's K loop and win for >= 512 columns and for 256 columns with
//     K >= 128;
//   * BDVCIL_DGRAD_64X128=1 (off): 64x128 tiles with FOUR workgroups per CU for those conv1 dgrads and the 128-column conv3 dgrads.
//     Alone they are faster (0.541 against 0.612 ms on 256<-64 @56, 0.225 against 0.263 on 1024<-256 @14, ...: -0.76 ms per step
//     summed over the sites, profiles/r02_tune_conv_fused.txt); in the whole step, where the weight gradients of the layer above
//     run beside them on the second stream, three alternating pairs of runs gave 587.9 / 588.1 / 586.4 clips/s with the rule
//     against 589.9 / 590.9 / 588.1 without (profiles/r02_ab_streams.txt) -- kept as a tested tile, not chosen by default.
// Returns the tile configuration, or -1 for the kernels that take fp32 weights.  BDVCIL_PL_TILE / bdv_conv_debug_force_tile
// override the rules wherever the forced tile divides the column count.
int pl_pick(bool dgrad, int ncols, int nk, int taps, int stride, int pieces = 3, bool shifted = false) {
  const int forced = pl_tile_override();
  if (pieces == 1) {  // the single-product arithmetic only exists in the plane kernels
    // A sixth of the MFMA work per K-step: the loop waits for its loads, not for the matrix pipe.  For the data gradient (whose
    // epilogue streams the residual, the mask and the statistics operands on top) many small workgroups hide that better than one
    // large one: 64x128 tiles, four workgroups per CU, for stride 1 and multiples of 128 columns.  Training step at batch 64,
    // bf16 storage: dgrad 18.9 -> 15.6 ms, 1100 -> 1158 clips/s; fp32 tensors: level (gpurun_out/np1_*.log).  The same tile for
    // fprop (8.8 against 8.0 ms) and 128x128 weight-gradient tiles (7.4 against 7.7 ms) were measured and left out.
    const bool small_ok = dgrad && stride == 1;
    if (forced >= 0 && forced < kNumPlCfg && ncols % kPlCfg[forced].BN == 0 && (forced != 4 || small_ok)) return forced;
    static const bool np1_small = getenv("BDVCIL_NP1_SMALL_TILES") == nullptr || atoi(getenv("BDVCIL_NP1_SMALL_TILES")) != 0;
    if (np1_small && small_ok && ncols % 128 == 0) return 4;
    return ncols % 256 == 0 ? 0 : ncols % 128 == 0 ? 1 : ncols % 64 == 0 ? 3 : -1;
  }
  if (pieces == 2) {
    // Two pieces per operand, three products: only the plane kernels have the form, so the sites the three-piece rules leave to the
    // conv_*_x3 kernels take the 128x128 four-wave tile (two workgroups per CU, like them) here.
    if (forced >= 0 && forced < kNumPlCfg && ncols % kPlCfg[forced].BN == 0 && (forced != 4 || (dgrad && stride == 1))) return forced;
    const int c = pl_pick(dgrad, ncols, nk, taps, stride, 3, shifted);
    return c >= 0 ? c : ncols % 128 == 0 ? 5 : ncols % 64 == 0 ? 3 : -1;
  }
  if (forced == kNumPlCfg) return -1;  // "none": the kernels of the other family everywhere
  if (forced >= 0 && ncols % kPlCfg[forced].BN == 0 && (forced != 4 || (dgrad && stride == 1))) return forced;   // (three pieces: the 64x128 tile exists for dgrad only)
  if (ncols % 64 != 0) return -1;
  if (ncols % 128 != 0) return taps > 1 ? 3 : -1;
  if (!dgrad) {
    if (ncols % 256 != 0) return nk >= 16 ? 1 : -1;
    if (nk >= 16 || nk <= 2) return 0;
    return -1;
  }
  // (round 2 also had modes 2 / 3 -- the conv1 dgrads of layers 1-2 / layer 1 only: level within the run-to-run spread,
  // profiles/r02_ab_streams.txt; removed)
  static const bool small_tiles = getenv("BDVCIL_DGRAD_64X128") != nullptr && atoi(getenv("BDVCIL_DGRAD_64X128")) == 1;
  if (ncols % 256 != 0) return taps > 1 && stride == 1 ? 1 : (small_tiles && taps == 1 && stride == 1 && nk >= 16 ? 4 : -1);
  if (stride == 2 && taps > 1) return 0;
  if (small_tiles && shifted && taps == 1) return 4;
  if (nk <= 8 && ncols >= 1024) return -1;
  if (shifted && taps == 1 && (ncols >= 512 || nk >= 4)) return -1;
  if (ncols == 512 && nk >= 64) return 0;
  return 2;
}

// The tile configuration (hence the row count of the fused-statistics partials) does not depend on the caller's workspace;
// only the K-split of the remainder tiles does.
PlPlan plan_pl(int cfg, int M, int ncols, int nk, size_t ws_bytes, bool ksplit_ok) {
  PlPlan p;
  p.cfg = cfg;
  p.est_us = 0.0;
  if (cfg < 0) return p;
  const PlCfg& k = kPlCfg[cfg];
  p.MT = (M + k.BM - 1) / k.BM;
  p.NT = ncols / k.BN;
  p.nk = nk;
  p.seg_bytes = (size_t)k.BM * k.BN * sizeof(float);
  if (ksplit_ok && cfg != 4) {   // (the 64 x 128 tile runs four workgroups per CU: thousands of tiles, no remainder worth splitting)
    p.wk = plan_work_pl(p.MT * p.NT, nk, k.iter_us, p.seg_bytes, ws_bytes, &p.est_us, pl_cfg_round(cfg));
  } else {
    p.wk.dp_tiles = p.MT * p.NT;
    p.wk.rem_tiles = 0;
    p.wk.split = 1;
    p.est_us = (double)((p.MT * p.NT + 255) / 256) * nk * k.iter_us;
  }
  return p;
}

// the two-workgroups-per-CU bf16-piece kernels read their weight operand from the planes as well (round 2's BDVCIL_R1_PLANES=0,
// "split it in the K loop as in round 1", is gone: 14.56 -> 13.98 ms over all fprop sites was its A/B)
constexpr bool r1_planes_enabled() { return true; }

// the stem (Cin = 4) on the bf16-piece K loop (BDVCIL_C4_X3=0: on the fp32-MFMA kernel, as in round 1)
bool c4_x3_enabled() {
  static const bool on = getenv("BDVCIL_C4_X3") == nullptr || atoi(getenv("BDVCIL_C4_X3")) != 0;
  return on;
}

int pl_fprop_cfg(const bdv_conv_geom* g, int pieces = 3) {
  if (g->Cin % BK != 0) return -1;
  return pl_pick(false, g->Cout, g->R * g->S * g->Cin / BK, g->R * g->S, g->stride, pieces);
}
int pl_dgrad_cfg(const bdv_conv_geom* g, int pieces = 3) {
  if (g->Cout % BK != 0) return -1;
  return pl_pick(true, g->Cin, g->R * g->S * g->Cout / BK, g->R * g->S, g->stride, pieces, g->fold > 0);
}
bool pl_fprop_ok(const bdv_conv_geom* g, int pieces = 3) { return pl_fprop_cfg(g, pieces) >= 0; }
// plane-kernel tile of a site whose input BatchNorm the loader applies (pre_scale): the planner's choice, or the widest tile that
// divides the column count where the planner would have used a kernel that cannot do it
int pl_fprop_cfg_pre(const bdv_conv_geom* g) {
  if (g->Cin % BK != 0 || g->Cout % 64 != 0) return -1;
  const int cfg = pl_fprop_cfg(g, 3);
  if (cfg >= 0) return cfg;
  return g->Cout % 256 == 0 ? 0 : g->Cout % 128 == 0 ? 1 : 3;
}
bool pl_dgrad_ok(const bdv_conv_geom* g, int pieces = 3) { return pl_dgrad_cfg(g, pieces) >= 0; }


struct WgradPlan {
  bool small;  // 64x64 tiles
  bool c4;
  int MTw, NTw, splits, kt_per_split;
};

WgradPlan plan_wgrad(const bdv_conv_geom* g) {
  static const int W = getenv("BDVCIL_WGRAD_W") ? atoi(getenv("BDVCIL_WGRAD_W")) : 768;
  WgradPlan p;
  p.c4 = (g->Cin % BK) != 0;
  p.small = p.c4 || (g->Cout % 128) != 0 || (g->Cin % 128) != 0;
  const int bm = p.small ? 64 : 128, bn = p.small ? 64 : 128;
  const int ktot = g->R * g->S * g->Cin;
  p.MTw = g->Cout / bm;
  p.NTw = p.c4 ? (ktot + bn - 1) / bn : g->R * g->S * (g->Cin / bn);
  const int64_t M = (int64_t)g->N * g->Ho * g->Wo;
  const int nkt = (int)((M + BK - 1) / BK);
  const int tiles = p.MTw * p.NTw;
  // blocks = tiles * splits should fill whole rounds of co-resident blocks; slab bytes grow with splits
  int best_s = 1;
  double best_cost = 1e30;
  const double dw_bytes = (double)g->Cout * ktot * 4.0;
  const int Wk = p.small ? 2 * W : W;  // the 16-accumulator blocks fit twice as many per CU
  for (int k = 1; k <= 4; ++k) {
    int sp = (int)(((long long)k * Wk) / tiles);
    if (sp < 1) sp = 1;
    if (sp > 1024) sp = 1024;
    while (sp > 1 && (nkt + sp - 1) / sp < 4) --sp;
    const int per = (nkt + sp - 1) / sp;
    const double rounds = (double)(((long long)tiles * sp + Wk - 1) / Wk);
    const double t_iter = 1.7 * 3.0 * (p.small ? 0.25 : 1.0);
    const double cost = rounds * per * t_iter + 4.0 + sp * dw_bytes / 4.0e6;
    if (cost < best_cost) {
      best_cost = cost;
      best_s = sp;
    }
  }
  const int per = (nkt + best_s - 1) / best_s;
  p.kt_per_split = per;
  p.splits = (nkt + per - 1) / per;
  return p;
}

// Tile forms of conv_wgrad_pl_kernel: 0 = 128/256 x 128/256 (8 waves); the 64-channel layers: 1 = 64 x 192 (a filter row of
// three taps x 64 input channels), 2 = 64 x 64 (1x1), 3 = 256 x 64 (1x1), 4 = 64 x 256 (1x1), 5 = the stem (Cin = 4 on NHWC4:
// 64 x 256 = 64 taps of 4 channels, R*S <= 64, the columns past the last tap stay empty); -1 = not covered.
// BDVCIL_WGRAD_2STAGE=0: the one-stage weight-gradient loop for every tile (read per call: A/B runs flip it)
bool wgrad_two_stage() {
  const char* e = getenv("BDVCIL_WGRAD_2STAGE");
  return e == nullptr || atoi(e) != 0;
}

int pl_wgrad_form(const bdv_conv_geom* g) {
  static const bool small_on = getenv("BDVCIL_PL_WGRAD64") == nullptr || atoi(getenv("BDVCIL_PL_WGRAD64")) != 0;
  const int RS = g->R * g->S;
  if (g->Cin % 128 == 0 && g->Cout % 128 == 0) return 0;
  if (!small_on) return -1;
  if (g->Cout == 64 && g->Cin == 64) return RS % 3 == 0 ? 1 : RS == 1 ? 2 : -1;
  if (g->Cout % 256 == 0 && g->Cin == 64 && RS == 1) return 3;
  if (g->Cout == 64 && g->Cin % 256 == 0 && RS == 1) return 4;
  if (g->Cout == 64 && g->Cin == 4 && c4_x3_enabled()) return 5;   // any number of taps: tiles of 64 taps (Rt * R * S of them)
  // bf16 storage has no other kernel family to fall back to: the 64 x 64 tile (one filter tap per tile) covers what is left
  // (R18 / R34: 64 -> 128 channels)
  if (g->act_dtype == BDV_ACT_BF16 && g->Cout % 64 == 0 && g->Cin % 64 == 0) return 2;
  return -1;
}
bool pl_wgrad_ok(const bdv_conv_geom* g) { return pl_wgrad_form(g) >= 0; }

// wgrad of the P family: one workgroup per CU, tiles of 128 / 256 output channels x 128 / 256 input channels of a tap
struct WgradPlPlan {
  int form, BM, BN, MTw, NTw, splits, kt_per_split;
};

WgradPlPlan plan_wgrad_pl(const bdv_conv_geom* g) {
  WgradPlPlan p;
  p.form = pl_wgrad_form(g);
  static const int kBM[6] = {0, 64, 64, 256, 64, 64}, kBN[6] = {0, 192, 64, 64, 256, 256};
  p.BM = p.form == 0 ? (g->Cout % 256 == 0 ? 256 : 128) : kBM[p.form];
  p.BN = p.form == 0 ? (g->Cin % 256 == 0 ? 256 : 128) : kBN[p.form];
  p.MTw = g->Cout / p.BM;
  p.NTw = ((g->Rt > 1 ? g->Rt : 1) * g->R * g->S * g->Cin + p.BN - 1) / p.BN;
  const int64_t M = (int64_t)g->N * g->Ho * g->Wo;
  const int nkt = (int)((M + BK - 1) / BK);
  const int tiles = p.MTw * p.NTw;
  const int W = p.form == 0 ? 256 : 512;   // the 4-wave forms run two workgroups per CU
  // K-step of one workgroup, microseconds: MFMA-bound for the large tiles; the 64-channel layers are bound by streaming their
  // operands from HBM (a K-step reads 32 x (BM + BN) floats; 256 CUs share ~5 TB/s)
  const double t_iter = p.form == 0 ? 2.7 * (double)(p.BM * p.BN) / (128.0 * 256.0)
                                    : fmax(2.7 * (double)(p.BM * p.BN) / (128.0 * 256.0) * 2.0, 32.0 * (p.BM + p.BN) * 4.0 * 2.0 / 20.0e3);
  const double dw_bytes = (double)g->Cout * (g->Rt > 1 ? g->Rt : 1) * g->R * g->S * g->Cin * 4.0;
  int best_s = 1;
  double best_cost = 1e30;
  for (int k = 1; k <= 6; ++k) {  // k rounds of 256 blocks
    int sp = (int)(((long long)k * W) / tiles);
    if (sp < 1) sp = 1;
    if (sp > 1024) sp = 1024;
    while (sp > 1 && (nkt + sp - 1) / sp < 4) --sp;
    const int per = (nkt + sp - 1) / sp;
    const double rounds = (double)(((long long)tiles * sp + W - 1) / W);
    const double cost = rounds * per * t_iter + 4.0 + 2.0 * sp * dw_bytes / 4.0e6;
    if (cost < best_cost) {
      best_cost = cost;
      best_s = sp;
    }
  }
  const int per = (nkt + best_s - 1) / best_s;
  p.kt_per_split = per;
  p.splits = (nkt + per - 1) / per;
  return p;
}

}  // namespace

extern "C" size_t bdv_conv_workspace_bytes(const bdv_conv_geom* gg, int kind) {
  if (check_geom(gg, "bdv_conv_workspace_bytes")) return 0;
  if (kind == 2) {
    const WgradPlan p = plan_wgrad(gg);
    return (size_t)p.splits * gg->Cout * gg->R * gg->S * gg->Cin * sizeof(float);
  }
  Geom g = make_geom(gg);
  FdPlan p;
  if (kind == 0) {
    g.M = g.N * g.Ho * g.Wo;
    g.Ktot = g.Rt * g.R * g.S * g.Cin;
    p = plan_fprop(g, kMaxSplitWorkspace);
  } else {
    if (gg->Cin % 64 != 0) return 0;
    g.M = g.N * g.H * g.W;
    g.Ktot = g.R * g.S * g.Cout;
    p = plan_dgrad(g, kMaxSplitWorkspace);
  }
  size_t need = p.wk.split > 1 ? (size_t)p.wk.rem_tiles * p.wk.split * p.seg_bytes : 0;
  // the P kernels (bdv_conv_fprop_pl / bdv_conv_dgrad_pl) plan their own K-split
  for (int pieces = 1; pieces <= 3; ++pieces) {
    if (kind == 0 && (pl_fprop_ok(gg, pieces) || (pieces == 3 && pl_fprop_cfg_pre(gg) >= 0))) {
      const PlPlan q = plan_pl(pl_fprop_ok(gg, pieces) ? pl_fprop_cfg(gg, pieces) : pl_fprop_cfg_pre(gg), g.M, g.Cout, g.Ktot / BK, kMaxSplitWorkspace, true);
      const size_t n2 = q.wk.split > 1 ? (size_t)q.wk.rem_tiles * q.wk.split * q.seg_bytes : 0;
      need = n2 > need ? n2 : need;
    } else if (kind == 1 && pl_dgrad_ok(gg, pieces) && gg->stride == 1) {
      const PlPlan q = plan_pl(pl_dgrad_cfg(gg, pieces), g.M, g.Cin, g.Ktot / BK, kMaxSplitWorkspace, true);
      const size_t n2 = q.wk.split > 1 ? (size_t)q.wk.rem_tiles * q.wk.split * q.seg_bytes : 0;
      need = n2 > need ? n2 : need;
    }
  }
  return need > 16 ? need : 16;
}

extern "C" int bdv_conv_fprop_stat_rows(const bdv_conv_geom* gg) {
  if (check_geom(gg, "bdv_conv_fprop_stat_rows")) return 0;
  const int64_t M = (int64_t)gg->N * gg->Ho * gg->Wo;
  return (int)((M + 127) / 128);
}

namespace {
int conv_fprop_impl(const float* x, const float* w, float* y, const bdv_conv_geom* gg, float* bn_partial,
                    const bdv_conv_affine* affine, void* workspace, size_t workspace_bytes, void* stream, bool x3,
                    const void* planes_f = nullptr) {
  if (int e = check_geom(gg, "bdv_conv_fprop")) return e;
  BDV_REQUIRE(x && w && y, "bdv_conv_fprop: null pointer");
  FpropEpi epi = {bn_partial, 0, nullptr, nullptr, nullptr, 0};
  if (affine != nullptr) {
    BDV_REQUIRE(bn_partial == nullptr, "bdv_conv_fprop: batch statistics and the folded eval-mode BatchNorm exclude each other");
    BDV_REQUIRE(affine->scale && affine->shift, "bdv_conv_fprop: null pointer in bdv_conv_affine");
    BDV_REQUIRE(bdv_aligned16(affine->scale) && bdv_aligned16(affine->shift) && bdv_aligned16(affine->residual),
                "bdv_conv_fprop: bdv_conv_affine pointers must be 16-byte aligned");
    epi.scale = affine->scale;
    epi.shift = affine->shift;
    epi.res = affine->residual;
    epi.relu = affine->relu;
  }
  BDV_REQUIRE(bdv_aligned16(x) && bdv_aligned16(w) && bdv_aligned16(y) && bdv_aligned16(workspace),
              "bdv_conv_fprop: pointers must be 16-byte aligned");
  Geom g = make_geom(gg);
  g.M = g.N * g.Ho * g.Wo;
  g.Ktot = g.Rt * g.R * g.S * g.Cin;
  hipStream_t s = (hipStream_t)stream;
  const bool c4 = (g.Cin % BK) != 0;
  const FdPlan p = plan_fprop(g, workspace ? (workspace_bytes < kMaxSplitWorkspace ? workspace_bytes : kMaxSplitWorkspace) : 0, x3 && !c4);
  const int blocks = p.wk.dp_tiles + p.wk.rem_tiles * p.wk.split;
  epi.MT = p.MT;
  float* slab = (float*)workspace;
  if (debug_plan())
    fprintf(stderr, "[bdv plan] fprop %dx%d Cin %d Cout %d k%d s%d: tiles %d nk %d -> dp %d rem %d split %d\n", g.H, g.W,
            g.Cin, g.Cout, g.R, g.stride, p.MT * p.NT, p.nk, p.wk.dp_tiles, p.wk.rem_tiles, p.wk.split);
  BDV_REQUIRE(gg->Rt <= 1 || (c4 && x3 && c4_x3_enabled()), "bdv_conv_fprop: temporal taps (Rt=%d) need the bf16-piece stem kernel (bdv_conv_fprop_x3)", gg->Rt);
  if (c4 && x3 && c4_x3_enabled()) {
    if (p.wide)
      hipLaunchKernelGGL((conv_fprop_c4_x3_kernel<128, 128, 2, 2>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, epi);
    else
      hipLaunchKernelGGL((conv_fprop_c4_x3_kernel<128, 64, 2, 2>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, epi);
  } else if (c4) {
    if (p.wide)
      hipLaunchKernelGGL((conv_fprop_c4_kernel<128, 128, 2, 2>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, epi);
    else
      hipLaunchKernelGGL((conv_fprop_c4_kernel<128, 64, 2, 2>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, epi);
  } else if (p.wide && x3) {
    if (planes_f != nullptr)
      hipLaunchKernelGGL((conv_fprop_x3_kernel<128, 128, 2, 2, true>), dim3(blocks), dim3(256), 0, s, x, (const float*)planes_f, y, g, p.NT, p.wk, slab, epi);
    else
      hipLaunchKernelGGL((conv_fprop_x3_kernel<128, 128, 2, 2>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, p.wk, slab, epi);
  } else if (p.wide) {
    hipLaunchKernelGGL((conv_fprop_kernel<128, 128, 2, 2>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, p.wk, slab, epi);
  } else {
    hipLaunchKernelGGL((conv_fprop_kernel<128, 64, 2, 2>), dim3(blocks), dim3(256), 0, s, x, w, y, g, p.NT, p.wk, slab, epi);
  }
  BDV_LAUNCH_CHECK("bdv_conv_fprop");
  if (p.wk.split > 1) {
    const dim3 fg(p.wk.rem_tiles);
    if (p.wide)
      hipLaunchKernelGGL((conv_fprop_fixup_kernel<128, 128, 2, 2>), fg, dim3(256), 0, s, (const float*)slab, y, g, p.NT, p.wk, epi);
    else
      hipLaunchKernelGGL((conv_fprop_fixup_kernel<128, 64, 2, 2>), fg, dim3(256), 0, s, (const float*)slab, y, g, p.NT, p.wk, epi);
    BDV_LAUNCH_CHECK("bdv_conv_fprop(fixup)");
  }
  return BDV_OK;
}
}  // namespace

extern "C" int bdv_conv_fprop(const float* x, const float* w, float* y, const bdv_conv_geom* gg, float* bn_partial,
                              const bdv_conv_affine* affine, void* workspace, size_t workspace_bytes, void* stream) {
  return conv_fprop_impl(x, w, y, gg, bn_partial, affine, workspace, workspace_bytes, stream, false);
}

extern "C" int bdv_conv_fprop_x3(const float* x, const float* w, float* y, const bdv_conv_geom* gg, float* bn_partial,
                                 const bdv_conv_affine* affine, void* workspace, size_t workspace_bytes, void* stream) {
  return conv_fprop_impl(x, w, y, gg, bn_partial, affine, workspace, workspace_bytes, stream, true);
}

extern "C" int bdv_conv_dgrad_stat_rows(const bdv_conv_geom* gg) {
  if (check_geom(gg, "bdv_conv_dgrad_stat_rows")) return 0;
  const int st = gg->stride;
  if (st == 1) return (int)(((int64_t)gg->N * gg->H * gg->W + 127) / 128);
  const int64_t Mc0 = (int64_t)gg->N * ((gg->H + st - 1) / st) * ((gg->W + st - 1) / st);  // largest parity class
  return (int)(st * st * ((Mc0 + 127) / 128));
}

namespace {
// w_t != nullptr: the bf16-piece kernel on the per-tap transposed weights (wide tiles only), else the fp32-MFMA kernels
int conv_dgrad_impl(const float* dy, const float* w, const float* w_t, float* dx, const float* add_src,
                    const uint32_t* add_mask_src, const bdv_conv_geom* gg, const bdv_bn_stat_fuse* bn_stat, void* workspace,
                    size_t workspace_bytes, void* stream, const void* planes_d = nullptr) {
  if (int e = check_geom(gg, "bdv_conv_dgrad")) return e;
  BnStat stat = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  if (bn_stat != nullptr) {
    BDV_REQUIRE(gg->stride == 1 || (gg->R >= gg->stride && gg->S >= gg->stride),
                "bdv_conv_dgrad: fused BatchNorm statistics with stride 2 need a filter that reaches every input pixel (R, S >= 2)");
    BDV_REQUIRE(bn_stat->y && bn_stat->mean && bn_stat->invstd && bn_stat->partial, "bdv_conv_dgrad: null pointer in bdv_bn_stat_fuse");
    BDV_REQUIRE(bdv_aligned16(bn_stat->y) && bdv_aligned16(bn_stat->mean) && bdv_aligned16(bn_stat->invstd) &&
                    bdv_aligned16(bn_stat->partial), "bdv_conv_dgrad: bdv_bn_stat_fuse pointers must be 16-byte aligned");
    BDV_REQUIRE(bn_stat->relu_mask == nullptr || gg->Cin % 32 == 0, "bdv_conv_dgrad: a ReLU mask needs Cin %% 32 == 0");
    stat.y = bn_stat->y;
    stat.mask = bn_stat->relu_mask;
    stat.mean = bn_stat->mean;
    stat.invstd = bn_stat->invstd;
    stat.partial = bn_stat->partial;
    stat.rscale = bn_stat->relu_mask == nullptr ? bn_stat->relu_scale : nullptr;
    stat.rshift = bn_stat->relu_shift;
    stat.MT = bdv_conv_dgrad_stat_rows(gg);
    BDV_REQUIRE(stat.rscale == nullptr || w_t != nullptr || planes_d != nullptr || gg->Cin % 128 != 0,
                "bdv_conv_dgrad: a ReLU sign derived from y (relu_scale) is not compiled into the 128-wide fp32-MFMA kernel");
    if (gg->stride != 1)  // parity classes of odd-sized inputs have fewer row tiles than the largest: their rows stay zero
      (void)hipMemsetAsync(bn_stat->partial, 0, (size_t)2 * stat.MT * gg->Cin * sizeof(float), (hipStream_t)stream);
  }
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
  const bool x3 = (w_t != nullptr || planes_d != nullptr) && (g.Cin % 128) == 0;
  BDV_REQUIRE(w_t == nullptr || bdv_aligned16(w_t), "bdv_conv_dgrad_x3: w_t must be 16-byte aligned");
  const FdPlan p = plan_dgrad(g, workspace ? (workspace_bytes < kMaxSplitWorkspace ? workspace_bytes : kMaxSplitWorkspace) : 0, x3);
  const dim3 grid(p.wk.dp_tiles + p.wk.rem_tiles * p.wk.split, st * st);
  float* slab = (float*)workspace;
  if (debug_plan())
    fprintf(stderr, "[bdv plan] dgrad %dx%d Cin %d Cout %d k%d s%d: tiles %d nk %d -> dp %d rem %d split %d\n", g.H, g.W,
            g.Cin, g.Cout, g.R, g.stride, p.MT * p.NT, p.nk, p.wk.dp_tiles, p.wk.rem_tiles, p.wk.split);
  if (p.wide && x3)
    if (planes_d != nullptr)
      hipLaunchKernelGGL((conv_dgrad_x3_kernel<128, 128, 2, 2, true>), grid, dim3(256), 0, s, dy, (const float*)planes_d, dx, add_src, add_mask_src, g, p.NT, p.wk, slab, stat);
    else
      hipLaunchKernelGGL((conv_dgrad_x3_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, dy, w_t, dx, add_src, add_mask_src, g, p.NT, p.wk, slab, stat);
  else if (p.wide)
    hipLaunchKernelGGL((conv_dgrad_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, dy, w, dx, add_src, add_mask_src, g, p.NT, p.wk, slab, stat);
  else
    hipLaunchKernelGGL((conv_dgrad_kernel<128, 64, 2, 2>), grid, dim3(256), 0, s, dy, w, dx, add_src, add_mask_src, g, p.NT, p.wk, slab, stat);
  BDV_LAUNCH_CHECK("bdv_conv_dgrad");
  if (p.wk.split > 1) {
    const dim3 fg(p.wk.rem_tiles);
    if (p.wide)
      hipLaunchKernelGGL((conv_dgrad_fixup_kernel<128, 128, 2, 2>), fg, dim3(256), 0, s, (const float*)slab, dx, add_src, add_mask_src, g, p.NT, p.wk, stat);
    else
      hipLaunchKernelGGL((conv_dgrad_fixup_kernel<128, 64, 2, 2>), fg, dim3(256), 0, s, (const float*)slab, dx, add_src, add_mask_src, g, p.NT, p.wk, stat);
    BDV_LAUNCH_CHECK("bdv_conv_dgrad(fixup)");
  }
  return BDV_OK;
}
}  // namespace

extern "C" int bdv_conv_dgrad(const float* dy, const float* w, float* dx, const float* add_src,
                              const uint32_t* add_mask_src, const bdv_conv_geom* gg, const bdv_bn_stat_fuse* bn_stat,
                              void* workspace, size_t workspace_bytes, void* stream) {
  return conv_dgrad_impl(dy, w, nullptr, dx, add_src, add_mask_src, gg, bn_stat, workspace, workspace_bytes, stream);
}

extern "C" int bdv_conv_dgrad_x3(const float* dy, const float* w, const float* w_t, float* dx, const float* add_src,
                                 const uint32_t* add_mask_src, const bdv_conv_geom* gg, const bdv_bn_stat_fuse* bn_stat,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  BDV_REQUIRE(w_t != nullptr, "bdv_conv_dgrad_x3: w_t is null");
  return conv_dgrad_impl(dy, w, w_t, dx, add_src, add_mask_src, gg, bn_stat, workspace, workspace_bytes, stream);
}

// ---- weights as bf16 planes + the P kernels --------------------------------------------------------------------------
extern "C" size_t bdv_conv_weight_planes_bytes(const bdv_conv_geom* gg) {
  if (check_geom(gg, "bdv_conv_weight_planes_bytes")) return 0;
  return (size_t)3 * gg->Cout * gg->R * gg->S * gg->Cin * sizeof(unsigned short);
}

#ifdef BDV_STAMPS
// diagnostic builds only (not declared in include/bdvcil_hip.h, not in the product library): where the kernels write their stamps
extern "C" int bdv_debug_set_stamps(void* buf, int cap_workgroups) {
  unsigned long long* p = (unsigned long long*)buf;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_cap), &cap_workgroups, sizeof(int)) != hipSuccess) return -1;
  return 0;
}
#endif

extern "C" int bdv_conv_debug_force_tile(int cfg) {
  BDV_REQUIRE(cfg >= -1 && cfg <= kNumPlCfg, "bdv_conv_debug_force_tile: cfg %d (-1 automatic, 0 = 128x256, 1 = 256x128, 2 = 256x256, 3 = 256x64, 4 = 64x128 (dgrad), 5 = 128x128 (four waves), 6 = none)", cfg);
  g_pl_tile_forced = cfg;
  return BDV_OK;
}

// Name of the main kernel a call will launch (as rocprofv3 prints it, without the anonymous namespace): for profiles and
// bench.py's per-kernel accounting.  kind 0 = fprop, 1 = dgrad, 2 = wgrad; arith 0 = fp32 MFMA entry points, 1 = the default
// bf16-piece entry points (bdv_conv_fprop_pl / bdv_conv_dgrad_pl / bdv_conv_wgrad_partial_pl), 2 = the same with pieces = 1,
// 3 = with pieces = 2.
extern "C" int bdv_conv_kernel_name(const bdv_conv_geom* gg, int kind, int arith, char* out, size_t n) {
  if (int e = check_geom(gg, "bdv_conv_kernel_name")) return e;
  BDV_REQUIRE(out && n > 0 && kind >= 0 && kind <= 2, "bdv_conv_kernel_name: bad argument");
  const bool c4 = gg->Cin % BK != 0;
  const int pieces = arith == 2 ? 1 : arith == 3 ? 2 : 3;
  if (kind == 0) {
    const int cfg = arith ? pl_fprop_cfg(gg, pieces) : -1;
    if (cfg >= 0) {
      snprintf(out, n, "conv_fprop_pl_kernel<%d, %d, %d, %d, %d, %d>", kPlCfg[cfg].BM, kPlCfg[cfg].BN, pl_cfg_wm(cfg), pl_cfg_wn(cfg), kPlCfg[cfg].nbuf, pieces);
    } else if (c4) {
      snprintf(out, n, "conv_fprop_c4_%skernel<128, %d, 2, 2>", arith && c4_x3_enabled() ? "x3_" : "", gg->Cout % 128 == 0 ? 128 : 64);
    } else if (gg->Cout % 128 == 0 && arith) {
      snprintf(out, n, "conv_fprop_x3_kernel<128, 128, 2, 2, %s>", r1_planes_enabled() ? "true" : "false");
    } else {
      snprintf(out, n, "conv_fprop_kernel<128, %d, 2, 2>", gg->Cout % 128 == 0 ? 128 : 64);
    }
  } else if (kind == 1) {
    const int cfg = arith ? pl_dgrad_cfg(gg, pieces) : -1;
    if (cfg >= 0) {
      snprintf(out, n, "conv_dgrad_pl_kernel<%d, %d, %d, %d, %d, %d>", kPlCfg[cfg].BM, kPlCfg[cfg].BN, pl_cfg_wm(cfg), pl_cfg_wn(cfg), kPlCfg[cfg].nbuf, pieces);
    } else if (gg->Cin % 128 == 0 && arith) {
      snprintf(out, n, "conv_dgrad_x3_kernel<128, 128, 2, 2, %s>", r1_planes_enabled() ? "true" : "false");
    } else {
      snprintf(out, n, "conv_dgrad_kernel<128, %d, 2, 2>", gg->Cin % 128 == 0 ? 128 : 64);
    }
  } else {
    if (arith && pl_wgrad_ok(gg)) {
      const WgradPlPlan p = plan_wgrad_pl(gg);
      snprintf(out, n, "conv_wgrad_pl_kernel<%d, %d", p.BM, p.BN);
    } else {
      const WgradPlan p = plan_wgrad(gg);
      snprintf(out, n, "conv_wgrad_kernel<%d, %d, 2, 2", p.small ? 64 : 128, p.small ? 64 : 128);
    }
  }
  return BDV_OK;
}

extern "C" int bdv_conv_uses_planes(const bdv_conv_geom* gg, int kind, int pieces) {
  if (check_geom(gg, "bdv_conv_uses_planes")) return 0;
  if (kind == 0) return pl_fprop_ok(gg, pieces) || (pieces >= 2 && r1_planes_enabled() && gg->Cin % BK == 0 && gg->Cout % 128 == 0) ? 1 : 0;
  if (kind == 1) return pl_dgrad_ok(gg, pieces) || (pieces >= 2 && r1_planes_enabled() && gg->Cout % BK == 0 && gg->Cin % 128 == 0) ? 1 : 0;
  return 0;
}

extern "C" int bdv_conv_fprop_pl_stat_rows(const bdv_conv_geom* gg, int pieces) {
  if (check_geom(gg, "bdv_conv_fprop_pl_stat_rows")) return 0;
  if (!pl_fprop_ok(gg, pieces)) return bdv_conv_fprop_stat_rows(gg);
  const PlPlan p = plan_pl(pl_fprop_cfg(gg, pieces), gg->N * gg->Ho * gg->Wo, gg->Cout, gg->R * gg->S * gg->Cin / BK, kMaxSplitWorkspace, true);
  return p.cfg >= 0 ? p.MT : 0;
}

extern "C" int bdv_conv_fprop_pre_stat_rows(const bdv_conv_geom* gg) {
  if (check_geom(gg, "bdv_conv_fprop_pre_stat_rows")) return 0;
  const int cfg = pl_fprop_cfg_pre(gg);
  if (cfg < 0) return 0;
  const PlPlan p = plan_pl(cfg, gg->N * gg->Ho * gg->Wo, gg->Cout, gg->R * gg->S * gg->Cin / BK, kMaxSplitWorkspace, true);
  return p.MT;
}

extern "C" int bdv_conv_dgrad_pl_stat_rows(const bdv_conv_geom* gg, int pieces) {
  if (check_geom(gg, "bdv_conv_dgrad_pl_stat_rows")) return 0;
  if (!pl_dgrad_ok(gg, pieces)) return bdv_conv_dgrad_stat_rows(gg);
  const int st = gg->stride;
  const int Mc0 = gg->N * ((gg->H + st - 1) / st) * ((gg->W + st - 1) / st);
  const PlPlan p = plan_pl(pl_dgrad_cfg(gg, pieces), st == 1 ? gg->N * gg->H * gg->W : Mc0, gg->Cin, gg->R * gg->S * gg->Cout / BK,
                           kMaxSplitWorkspace, st == 1);
  return p.cfg >= 0 ? st * st * p.MT : 0;
}

extern "C" int bdv_conv_split_weights(const float* w, const bdv_conv_geom* gg, void* planes_fprop, void* planes_dgrad, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_split_weights")) return e;
  BDV_REQUIRE(w && (planes_fprop || planes_dgrad), "bdv_conv_split_weights: null pointer");
  BDV_REQUIRE(gg->Cin % 32 == 0 && gg->Cout % 32 == 0, "bdv_conv_split_weights: Cin=%d and Cout=%d must be multiples of 32", gg->Cin, gg->Cout);
  BDV_REQUIRE(bdv_aligned16(w) && bdv_aligned16(planes_fprop) && bdv_aligned16(planes_dgrad), "bdv_conv_split_weights: alignment");
  const dim3 grid(gg->Cin / 32, gg->Cout / 32, gg->R * gg->S);
  hipLaunchKernelGGL(split_weights_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)planes_fprop,
                     (unsigned short*)planes_dgrad, gg->Cout, gg->R * gg->S, gg->Cin);
  BDV_LAUNCH_CHECK("bdv_conv_split_weights");
  return BDV_OK;
}

extern "C" int bdv_conv_fprop_pre_ok(const bdv_conv_geom* gg) {
  if (check_geom(gg, "bdv_conv_fprop_pre_ok")) return 0;
  return pl_fprop_cfg_pre(gg) >= 0 && gg->fold == 0 ? 1 : 0;
}

extern "C" int bdv_conv_fprop_pl(const void* x, const float* w, const void* planes_fprop, void* y, const bdv_conv_geom* gg,
                                 float* bn_partial, const bdv_conv_affine* affine, void* workspace, size_t workspace_bytes,
                                 int pieces, const float* pre_scale, const float* pre_shift, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_fprop_pl")) return e;
  BDV_REQUIRE(pieces >= 1 && pieces <= 3, "bdv_conv_fprop_pl: pieces = %d (3, 2 or 1)", pieces);
  const bool pre = pre_scale != nullptr;
  const bool h16 = gg->act_dtype == BDV_ACT_BF16;
  BDV_REQUIRE_ACT(gg->act_dtype, "bdv_conv_fprop_pl");
  if (h16)   // bf16 storage: the single-product plane kernels are the only kernels that read / write it
    BDV_REQUIRE(pieces == 1 && !pre && planes_fprop != nullptr && pl_fprop_ok(gg, 1),
                "bdv_conv_fprop_pl: bf16 activation storage needs pieces = 1, the weight planes, Cin %% 32 == 0 and Cout %% 64 == 0 (Cin=%d Cout=%d)", gg->Cin, gg->Cout);
  if (pre) {
    BDV_REQUIRE(pre_shift != nullptr && planes_fprop != nullptr && pieces == 3 && gg->fold == 0 && pl_fprop_cfg_pre(gg) >= 0,
                "bdv_conv_fprop_pl: a producer BatchNorm in the loader needs the plane kernels (pieces 3, no temporal shift, Cin %% 32 == 0)");
    BDV_REQUIRE(bdv_aligned16(pre_scale) && bdv_aligned16(pre_shift), "bdv_conv_fprop_pl: pre_scale / pre_shift must be 16-byte aligned");
  }
  if (!pre && (planes_fprop == nullptr || !pl_fprop_ok(gg, pieces)))  // sites of the two-workgroups-per-CU kernels (weights from the planes too)
    return conv_fprop_impl((const float*)x, w, (float*)y, gg, bn_partial, affine, workspace, workspace_bytes, stream, true,
                           gg->Cin % BK == 0 && r1_planes_enabled() ? planes_fprop : nullptr);
  BDV_REQUIRE(x && y, "bdv_conv_fprop_pl: null pointer");
  FpropEpi epi = {bn_partial, 0, nullptr, nullptr, nullptr, 0, pre_scale, pre_shift};
  if (affine != nullptr) {
    BDV_REQUIRE(bn_partial == nullptr, "bdv_conv_fprop_pl: batch statistics and the folded eval-mode BatchNorm exclude each other");
    BDV_REQUIRE(affine->scale && affine->shift, "bdv_conv_fprop_pl: null pointer in bdv_conv_affine");
    BDV_REQUIRE(bdv_aligned16(affine->scale) && bdv_aligned16(affine->shift) && bdv_aligned16(affine->residual),
                "bdv_conv_fprop_pl: bdv_conv_affine pointers must be 16-byte aligned");
    epi.scale = affine->scale;
    epi.shift = affine->shift;
    epi.res = affine->residual;
    epi.relu = affine->relu;
  }
  BDV_REQUIRE(bdv_aligned16(x) && bdv_aligned16(planes_fprop) && bdv_aligned16(y) && bdv_aligned16(workspace),
              "bdv_conv_fprop_pl: pointers must be 16-byte aligned");
  Geom g = make_geom(gg);
  g.M = g.N * g.Ho * g.Wo;
  g.Ktot = g.Rt * g.R * g.S * g.Cin;
  BDV_REQUIRE((int64_t)3 * g.Cout * g.Ktot * 2 < (1ll << 31), "bdv_conv_fprop_pl: weight planes exceed 2^31 bytes");
  hipStream_t s = (hipStream_t)stream;
  const PlPlan p = plan_pl(pre ? pl_fprop_cfg_pre(gg) : pl_fprop_cfg(gg, pieces), g.M, g.Cout, g.Ktot / BK, workspace ? (workspace_bytes < kMaxSplitWorkspace ? workspace_bytes : kMaxSplitWorkspace) : 0, !h16);  // (no K split with bf16 storage: the fix-up kernels are fp32)
  BDV_REQUIRE(p.cfg >= 0, "bdv_conv_fprop_pl: no tile configuration for Cout=%d", g.Cout);
  // bn_partial has one row per row tile of THIS kernel: bdv_conv_fprop_pl_stat_rows(g)
  const int blocks = p.wk.dp_tiles + p.wk.rem_tiles * p.wk.split;
  epi.MT = p.MT;
  float* slab = (float*)workspace;
  const unsigned short* wp = (const unsigned short*)planes_fprop;
  if (debug_plan())
    fprintf(stderr, "[bdv plan] fprop_pl %dx%d Cin %d Cout %d k%d s%d: cfg %d tiles %d nk %d -> dp %d rem %d split %d (est %.0f us)\n", g.H,
            g.W, g.Cin, g.Cout, g.R, g.stride, p.cfg, p.MT * p.NT, p.nk, p.wk.dp_tiles, p.wk.rem_tiles, p.wk.split, p.est_us);
#define BDV_FPROP_PL(BM_, BN_, WM_, WN_, NB_)                                                                                         \
  do {                                                                                                                                \
    if (h16)                                                                                                                          \
      hipLaunchKernelGGL((conv_fprop_pl_kernel<BM_, BN_, WM_, WN_, NB_, 1, false, 2>), dim3(blocks), dim3(64 * WM_ * WN_), 0, s, x, wp, y, \
                         g, p.NT, p.wk, slab, epi);                                                                                   \
    else if (pre)                                                                                                                     \
      hipLaunchKernelGGL((conv_fprop_pl_kernel<BM_, BN_, WM_, WN_, NB_, 3, true>), dim3(blocks), dim3(64 * WM_ * WN_), 0, s, x, wp, y,  \
                         g, p.NT, p.wk, slab, epi);                                                                                   \
    else if (pieces == 1)                                                                                                             \
      hipLaunchKernelGGL((conv_fprop_pl_kernel<BM_, BN_, WM_, WN_, NB_, 1>), dim3(blocks), dim3(64 * WM_ * WN_), 0, s, x, wp, y, g,    \
                         p.NT, p.wk, slab, epi);                                                                                      \
    else if (pieces == 2)                                                                                                             \
      hipLaunchKernelGGL((conv_fprop_pl_kernel<BM_, BN_, WM_, WN_, NB_, 2>), dim3(blocks), dim3(64 * WM_ * WN_), 0, s, x, wp, y, g,    \
                         p.NT, p.wk, slab, epi);                                                                                      \
    else                                                                                                                              \
      hipLaunchKernelGGL((conv_fprop_pl_kernel<BM_, BN_, WM_, WN_, NB_, 3>), dim3(blocks), dim3(64 * WM_ * WN_), 0, s, x, wp, y, g,    \
                         p.NT, p.wk, slab, epi);                                                                                      \
    BDV_LAUNCH_CHECK("bdv_conv_fprop_pl");                                                                                            \
    if (p.wk.split > 1) {                                                                                                             \
      hipLaunchKernelGGL((conv_fprop_fixup_kernel<BM_, BN_, WM_, WN_>), dim3(p.wk.rem_tiles), dim3(64 * WM_ * WN_), 0, s,             \
                         (const float*)slab, (float*)y, g, p.NT, p.wk, epi);                                                                  \
      BDV_LAUNCH_CHECK("bdv_conv_fprop_pl(fixup)");                                                                                   \
    }                                                                                                                                 \
  } while (0)
  if (p.cfg == 0) BDV_FPROP_PL(128, 256, 2, 4, 2);
  else if (p.cfg == 1) BDV_FPROP_PL(256, 128, 4, 2, 2);
  else if (p.cfg == 2) BDV_FPROP_PL(256, 256, 2, 4, 1);
  else if (p.cfg == 5) BDV_FPROP_PL(128, 128, 2, 2, 1);
  else BDV_FPROP_PL(256, 64, 4, 2, 2);
#undef BDV_FPROP_PL
  return BDV_OK;
}

extern "C" int bdv_conv_dgrad_pl(const void* dy, const float* w, const void* planes_dgrad, void* dx, const void* add_src,
                                 const uint32_t* add_mask_src, const bdv_conv_geom* gg, const bdv_bn_stat_fuse* bn_stat,
                                 void* workspace, size_t workspace_bytes, int pieces, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_dgrad_pl")) return e;
  BDV_REQUIRE(pieces >= 1 && pieces <= 3, "bdv_conv_dgrad_pl: pieces = %d (3, 2 or 1)", pieces);
  const bool h16 = gg->act_dtype == BDV_ACT_BF16;
  BDV_REQUIRE_ACT(gg->act_dtype, "bdv_conv_dgrad_pl");
  if (h16)
    BDV_REQUIRE(pieces == 1 && planes_dgrad != nullptr && pl_dgrad_ok(gg, 1),
                "bdv_conv_dgrad_pl: bf16 activation storage needs pieces = 1, the weight planes, Cout %% 32 == 0 and Cin %% 64 == 0 (Cin=%d Cout=%d)", gg->Cin, gg->Cout);
  if (planes_dgrad == nullptr || !pl_dgrad_ok(gg, pieces))
    return conv_dgrad_impl((const float*)dy, w, nullptr, (float*)dx, (const float*)add_src, add_mask_src, gg, bn_stat, workspace, workspace_bytes, stream,
                           gg->Cout % BK == 0 && r1_planes_enabled() ? planes_dgrad : nullptr);
  BnStat stat = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  if (bn_stat != nullptr) {
    BDV_REQUIRE(gg->stride == 1 || (gg->R >= gg->stride && gg->S >= gg->stride),
                "bdv_conv_dgrad_pl: fused BatchNorm statistics with stride 2 need a filter that reaches every input pixel (R, S >= 2)");
    BDV_REQUIRE(bn_stat->y && bn_stat->mean && bn_stat->invstd && bn_stat->partial, "bdv_conv_dgrad_pl: null pointer in bdv_bn_stat_fuse");
    BDV_REQUIRE(bdv_aligned16(bn_stat->y) && bdv_aligned16(bn_stat->mean) && bdv_aligned16(bn_stat->invstd) &&
                    bdv_aligned16(bn_stat->partial), "bdv_conv_dgrad_pl: bdv_bn_stat_fuse pointers must be 16-byte aligned");
    BDV_REQUIRE(bn_stat->relu_mask == nullptr || gg->Cin % 32 == 0, "bdv_conv_dgrad_pl: a ReLU mask needs Cin %% 32 == 0");
    stat.y = bn_stat->y;
    stat.mask = bn_stat->relu_mask;
    stat.mean = bn_stat->mean;
    stat.invstd = bn_stat->invstd;
    stat.partial = bn_stat->partial;
    stat.rscale = bn_stat->relu_mask == nullptr ? bn_stat->relu_scale : nullptr;
    stat.rshift = bn_stat->relu_shift;
  }
  BDV_REQUIRE(dy && dx, "bdv_conv_dgrad_pl: null pointer");
  BDV_REQUIRE(bdv_aligned16(dy) && bdv_aligned16(planes_dgrad) && bdv_aligned16(dx) && bdv_aligned16(workspace),
              "bdv_conv_dgrad_pl: pointers must be 16-byte aligned");
  BDV_REQUIRE(add_src != nullptr || add_mask_src == nullptr, "bdv_conv_dgrad_pl: add_mask_src without add_src");
  Geom g = make_geom(gg);
  g.M = g.N * g.H * g.W;
  g.Ktot = g.R * g.S * g.Cout;
  BDV_REQUIRE((int64_t)3 * g.Cin * g.Ktot * 2 < (1ll << 31), "bdv_conv_dgrad_pl: weight planes exceed 2^31 bytes");
  hipStream_t s = (hipStream_t)stream;
  const int st = g.stride;
  const int Mc0 = g.N * ((g.H + st - 1) / st) * ((g.W + st - 1) / st);  // largest parity class
  PlPlan p = plan_pl(pl_dgrad_cfg(gg, pieces), st == 1 ? g.M : Mc0, g.Cin, g.Ktot / BK,
                     workspace ? (workspace_bytes < kMaxSplitWorkspace ? workspace_bytes : kMaxSplitWorkspace) : 0, st == 1 && !h16);
  BDV_REQUIRE(p.cfg >= 0, "bdv_conv_dgrad_pl: no tile configuration for Cin=%d", g.Cin);
  if (st != 1) p.wk.dp_tiles = ((p.MT + 7) / 8) * 8 * p.NT;  // padded grid per parity class
  // the statistics partial has one row per row tile of THIS kernel (stride 2: per parity class and row tile)
  stat.MT = st * st * p.MT;
  if (st != 1 && stat.y != nullptr)
    (void)hipMemsetAsync(stat.partial, 0, (size_t)2 * stat.MT * g.Cin * sizeof(float), s);
  const int work_items = p.wk.dp_tiles + p.wk.rem_tiles * p.wk.split;
  const dim3 grid(work_items, st * st);
  float* slab = (float*)workspace;
  const unsigned short* dp = (const unsigned short*)planes_dgrad;
  if (debug_plan())
    fprintf(stderr, "[bdv plan] dgrad_pl %dx%d Cin %d Cout %d k%d s%d: cfg %d tiles %d nk %d -> dp %d rem %d split %d (est %.0f us)\n", g.H,
            g.W, g.Cin, g.Cout, g.R, g.stride, p.cfg, p.MT * p.NT, p.nk, p.wk.dp_tiles, p.wk.rem_tiles, p.wk.split, p.est_us);
#define BDV_DGRAD_PL(BM_, BN_, WM_, WN_, NB_)                                                                                          \
  do {                                                                                                                                 \
    if (h16)                                                                                                                           \
      hipLaunchKernelGGL((conv_dgrad_pl_kernel<BM_, BN_, WM_, WN_, NB_, 1, 2>), grid, dim3(64 * WM_ * WN_), 0, s, dy, dp, dx, add_src,  \
                         add_mask_src, g, p.NT, p.wk, slab, stat);                                                                     \
    else if (pieces == 1)                                                                                                              \
      hipLaunchKernelGGL((conv_dgrad_pl_kernel<BM_, BN_, WM_, WN_, NB_, 1>), grid, dim3(64 * WM_ * WN_), 0, s, dy, dp, dx, add_src,     \
                         add_mask_src, g, p.NT, p.wk, slab, stat);                                                                     \
    else if (pieces == 2)                                                                                                              \
      hipLaunchKernelGGL((conv_dgrad_pl_kernel<BM_, BN_, WM_, WN_, NB_, 2>), grid, dim3(64 * WM_ * WN_), 0, s, dy, dp, dx, add_src,     \
                         add_mask_src, g, p.NT, p.wk, slab, stat);                                                                     \
    else                                                                                                                               \
      hipLaunchKernelGGL((conv_dgrad_pl_kernel<BM_, BN_, WM_, WN_, NB_, 3>), grid, dim3(64 * WM_ * WN_), 0, s, dy, dp, dx, add_src,     \
                         add_mask_src, g, p.NT, p.wk, slab, stat);                                                                     \
    BDV_LAUNCH_CHECK("bdv_conv_dgrad_pl");                                                                                             \
    if (p.wk.split > 1) {                                                                                                              \
      hipLaunchKernelGGL((conv_dgrad_fixup_kernel<BM_, BN_, WM_, WN_>), dim3(p.wk.rem_tiles), dim3(64 * WM_ * WN_), 0, s,              \
                         (const float*)slab, (float*)dx, (const float*)add_src, add_mask_src, g, p.NT, p.wk, stat);                                          \
      BDV_LAUNCH_CHECK("bdv_conv_dgrad_pl(fixup)");                                                                                    \
    }                                                                                                                                  \
  } while (0)
  if (p.cfg == 0) BDV_DGRAD_PL(128, 256, 2, 4, 2);
  else if (p.cfg == 1) BDV_DGRAD_PL(256, 128, 4, 2, 2);
  else if (p.cfg == 2) BDV_DGRAD_PL(256, 256, 2, 4, 1);
  else if (p.cfg == 4) BDV_DGRAD_PL(64, 128, 2, 2, 1);
  else if (p.cfg == 5) BDV_DGRAD_PL(128, 128, 2, 2, 1);
  else BDV_DGRAD_PL(256, 64, 4, 2, 2);
#undef BDV_DGRAD_PL
  return BDV_OK;
}

namespace {

// main kernel of a weight gradient: split-K partial products into `slab` (plan_wgrad(gg).splits slices of dw's size)
int wgrad_partial(const char* who, const float* dy, const float* x, const bdv_conv_geom* gg, void* workspace, size_t workspace_bytes,
                  hipStream_t s, int* splits_out, bool x3 = false) {
  BDV_REQUIRE(gg == nullptr || gg->Rt <= 1, "%s: temporal taps (Rt=%d) need the bf16-piece plane kernel (bdv_conv_wgrad_partial_pl, BDVCIL_C4_X3)", who, gg->Rt);
  const WgradPlan p = plan_wgrad(gg);
  const size_t need = (size_t)p.splits * gg->Cout * gg->R * gg->S * gg->Cin * sizeof(float);
  if (workspace_bytes < need) {
    bdv_set_error("%s: workspace %zu < required %zu bytes", who, workspace_bytes, need);
    return BDV_EWORKSPACE;
  }
  Geom g = make_geom(gg);
  g.M = g.N * g.Ho * g.Wo;
  g.Ktot = g.Rt * g.R * g.S * g.Cin;
  float* slab = (float*)workspace;
  const dim3 grid(p.MTw * p.NTw * p.splits);
  if (debug_plan())
    fprintf(stderr, "[bdv plan] wgrad %dx%d Cin %d Cout %d k%d s%d: tiles %d -> splits %d x %d k-iters (%s)\n", gg->H, gg->W,
            gg->Cin, gg->Cout, gg->R, gg->stride, p.MTw * p.NTw, p.splits, p.kt_per_split, p.small ? "64x64" : "128x128");
  const bool incr = g.Ho * g.Wo > BK && BK / g.Wo + 1 <= g.Ho;
#define BDV_WGRAD(BMN, C4F)                                                                                                    \
  do {                                                                                                                         \
    if (incr)                                                                                                                  \
      hipLaunchKernelGGL((conv_wgrad_kernel<BMN, BMN, 2, 2, C4F, true>), grid, dim3(256), 0, s, dy, x, slab, g, p.MTw, p.NTw,   \
                         p.kt_per_split);                                                                                      \
    else                                                                                                                       \
      hipLaunchKernelGGL((conv_wgrad_kernel<BMN, BMN, 2, 2, C4F, false>), grid, dim3(256), 0, s, dy, x, slab, g, p.MTw, p.NTw,  \
                         p.kt_per_split);                                                                                      \
  } while (0)
  if (!p.small && x3) {  // experimental bf16-piece K loop (128x128 tiles only)
    if (incr)
      hipLaunchKernelGGL((conv_wgrad_x3_kernel<128, 128, 2, 2, false, true>), grid, dim3(256), 0, s, dy, x, slab, g, p.MTw, p.NTw,
                         p.kt_per_split);
    else
      hipLaunchKernelGGL((conv_wgrad_x3_kernel<128, 128, 2, 2, false, false>), grid, dim3(256), 0, s, dy, x, slab, g, p.MTw, p.NTw,
                         p.kt_per_split);
  } else if (!p.small) {
    BDV_WGRAD(128, false);
  } else if (p.c4) {
    BDV_WGRAD(64, true);
  } else {
    BDV_WGRAD(64, false);
  }
#undef BDV_WGRAD
  BDV_LAUNCH_CHECK(who);
  *splits_out = p.splits;
  return BDV_OK;
}

}  // namespace

extern "C" int bdv_conv_wgrad(const float* dy, const float* x, float* dw, float beta, const bdv_conv_geom* gg,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_wgrad")) return e;
  BDV_REQUIRE(dy && x && dw && workspace, "bdv_conv_wgrad: null pointer");
  BDV_REQUIRE(bdv_aligned16(dy) && bdv_aligned16(x) && bdv_aligned16(dw) && bdv_aligned16(workspace),
              "bdv_conv_wgrad: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int splits = 0;
  if (int e = wgrad_partial("bdv_conv_wgrad", dy, x, gg, workspace, workspace_bytes, s, &splits)) return e;
  const int64_t numel4 = (int64_t)gg->Cout * gg->R * gg->S * gg->Cin / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)((numel4 + 63) / 64)), dim3(256), 0, s, (const float*)workspace, dw, beta, splits,
                     numel4);
  BDV_LAUNCH_CHECK("bdv_conv_wgrad(reduce)");
  return BDV_OK;
}

extern "C" int bdv_conv_wgrad_splits(const bdv_conv_geom* gg) {
  if (check_geom(gg, "bdv_conv_wgrad_splits")) return 0;
  return plan_wgrad(gg).splits;
}

extern "C" int bdv_conv_wgrad_partial(const float* dy, const float* x, const bdv_conv_geom* gg, void* slab, size_t slab_bytes,
                                      void* stream) {
  if (int e = check_geom(gg, "bdv_conv_wgrad_partial")) return e;
  BDV_REQUIRE(dy && x && slab, "bdv_conv_wgrad_partial: null pointer");
  BDV_REQUIRE(bdv_aligned16(dy) && bdv_aligned16(x) && bdv_aligned16(slab), "bdv_conv_wgrad_partial: pointers must be 16-byte aligned");
  int splits = 0;
  return wgrad_partial("bdv_conv_wgrad_partial", dy, x, gg, slab, slab_bytes, (hipStream_t)stream, &splits);
}

extern "C" int bdv_conv_wgrad_partial_x3(const float* dy, const float* x, const bdv_conv_geom* gg, void* slab, size_t slab_bytes,
                                         void* stream) {
  if (int e = check_geom(gg, "bdv_conv_wgrad_partial_x3")) return e;
  BDV_REQUIRE(dy && x && slab, "bdv_conv_wgrad_partial_x3: null pointer");
  BDV_REQUIRE(bdv_aligned16(dy) && bdv_aligned16(x) && bdv_aligned16(slab), "bdv_conv_wgrad_partial_x3: pointers must be 16-byte aligned");
  int splits = 0;
  return wgrad_partial("bdv_conv_wgrad_partial_x3", dy, x, gg, slab, slab_bytes, (hipStream_t)stream, &splits, true);
}

extern "C" int bdv_conv_wgrad_pl_splits(const bdv_conv_geom* gg) {
  if (check_geom(gg, "bdv_conv_wgrad_pl_splits")) return 0;
  return pl_wgrad_ok(gg) ? plan_wgrad_pl(gg).splits : plan_wgrad(gg).splits;
}

extern "C" int bdv_conv_wgrad_pre_ok(const bdv_conv_geom* gg) {
  if (check_geom(gg, "bdv_conv_wgrad_pre_ok")) return 0;
  return pl_wgrad_ok(gg) && gg->Cin % BK == 0 && gg->fold == 0 ? 1 : 0;
}

extern "C" int bdv_conv_wgrad_partial_pl(const void* dy, const void* x, const bdv_conv_geom* gg, void* slab, size_t slab_bytes,
                                         int pieces, const float* pre_scale, const float* pre_shift, void* stream) {
  if (int e = check_geom(gg, "bdv_conv_wgrad_partial_pl")) return e;
  BDV_REQUIRE(pre_scale == nullptr || (pre_shift != nullptr && pl_wgrad_ok(gg) && gg->Cin % BK == 0 && gg->fold == 0),
              "bdv_conv_wgrad_partial_pl: a producer BatchNorm in the loader needs the plane kernel, Cin %% 32 == 0 and no temporal shift");
  BDV_REQUIRE(pieces >= 1 && pieces <= 3, "bdv_conv_wgrad_partial_pl: pieces = %d (3, 2 or 1)", pieces);
  BDV_REQUIRE(dy && x && slab, "bdv_conv_wgrad_partial_pl: null pointer");
  BDV_REQUIRE(bdv_aligned16(dy) && bdv_aligned16(x) && bdv_aligned16(slab), "bdv_conv_wgrad_partial_pl: pointers must be 16-byte aligned");
  const bool h16 = gg->act_dtype == BDV_ACT_BF16;
  BDV_REQUIRE_ACT(gg->act_dtype, "bdv_conv_wgrad_partial_pl");
  if (h16)
    BDV_REQUIRE(pieces == 1 && pre_scale == nullptr && pl_wgrad_ok(gg) && gg->Cin % BK == 0,
                "bdv_conv_wgrad_partial_pl: bf16 activation storage needs pieces = 1 and the plane kernel (Cin %% 32 == 0; Cin=%d Cout=%d)", gg->Cin, gg->Cout);
  if (!pl_wgrad_ok(gg)) {
    int splits = 0;
    return wgrad_partial("bdv_conv_wgrad_partial_pl", (const float*)dy, (const float*)x, gg, slab, slab_bytes, (hipStream_t)stream, &splits, pieces >= 2);
  }
  const WgradPlPlan p = plan_wgrad_pl(gg);
  const size_t need = (size_t)p.splits * gg->Cout * (gg->Rt > 1 ? gg->Rt : 1) * gg->R * gg->S * gg->Cin * sizeof(float);
  if (slab_bytes < need) {
    bdv_set_error("bdv_conv_wgrad_partial_pl: slab %zu < required %zu bytes", slab_bytes, need);
    return BDV_EWORKSPACE;
  }
  Geom g = make_geom(gg);
  g.M = g.N * g.Ho * g.Wo;
  g.Ktot = g.Rt * g.R * g.S * g.Cin;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(p.MTw * p.NTw * p.splits);
  if (debug_plan())
    fprintf(stderr, "[bdv plan] wgrad_pl %dx%d Cin %d Cout %d k%d s%d: %dx%d tiles %d -> splits %d x %d k-iters\n", gg->H, gg->W, gg->Cin,
            gg->Cout, gg->R, gg->stride, p.BM, p.BN, p.MTw * p.NTw, p.splits, p.kt_per_split);
  const bool incr = g.Ho * g.Wo > BK && BK / g.Wo + 1 <= g.Ho;
#define BDV_WGRAD_PL2(BM_, BN_, WM_, WN_, INCR_, NP_, MTAP_, KW_, ES_)                                                                \
  hipLaunchKernelGGL((conv_wgrad_pl_kernel<BM_, BN_, WM_, WN_, INCR_, NP_, MTAP_, KW_, ES_>), grid, dim3(64 * WM_ * WN_), 0, s, dy, x, \
                     (float*)slab, g, p.MTw, p.NTw, p.kt_per_split, pre_scale, pre_shift)
#define BDV_WGRAD_PL(BM_, BN_, WM_, WN_, MTAP_, KW_)                                                                             \
  do {                                                                                                                           \
    if (h16 && incr) BDV_WGRAD_PL2(BM_, BN_, WM_, WN_, true, 1, MTAP_, (KW_ == 16 ? 32 : KW_), 2);                               \
    else if (h16) BDV_WGRAD_PL2(BM_, BN_, WM_, WN_, false, 1, MTAP_, (KW_ == 16 ? 32 : KW_), 2);                                 \
    else if (incr && pieces == 3) BDV_WGRAD_PL2(BM_, BN_, WM_, WN_, true, 3, MTAP_, KW_, 4);                                     \
    else if (pieces == 3) BDV_WGRAD_PL2(BM_, BN_, WM_, WN_, false, 3, MTAP_, KW_, 4);                                            \
    else if (incr && pieces == 2) BDV_WGRAD_PL2(BM_, BN_, WM_, WN_, true, 2, MTAP_, KW_, 4);                                     \
    else if (pieces == 2) BDV_WGRAD_PL2(BM_, BN_, WM_, WN_, false, 2, MTAP_, KW_, 4);                                            \
    else if (incr) BDV_WGRAD_PL2(BM_, BN_, WM_, WN_, true, 1, MTAP_, KW_, 4);                                                    \
    else BDV_WGRAD_PL2(BM_, BN_, WM_, WN_, false, 1, MTAP_, KW_, 4);                                                             \
  } while (0)
  const bool two_stage = wgrad_two_stage() && !h16;   // (bf16 tensors: a 16-pixel stage is half a 16-byte unit per thread)
  if (p.form == 1) BDV_WGRAD_PL(64, 192, 2, 2, true, 32);
  else if (p.form == 2) BDV_WGRAD_PL(64, 64, 2, 2, false, 32);
  else if (p.form == 3) BDV_WGRAD_PL(256, 64, 4, 1, false, 32);
  else if (p.form == 4) BDV_WGRAD_PL(64, 256, 1, 4, false, 32);
  else if (p.form == 5) BDV_WGRAD_PL(64, 256, 1, 4, true, 32);
  else if (p.BM == 256 && p.BN == 256) BDV_WGRAD_PL(256, 256, 2, 4, false, 32);   // two register sets would spill here (128 accumulators)
  else if (p.BM == 256 && two_stage) BDV_WGRAD_PL(256, 128, 2, 4, false, 16);
  else if (p.BM == 256) BDV_WGRAD_PL(256, 128, 2, 4, false, 32);
  else if (p.BN == 256 && two_stage) BDV_WGRAD_PL(128, 256, 2, 4, false, 16);
  else if (p.BN == 256) BDV_WGRAD_PL(128, 256, 2, 4, false, 32);
  else BDV_WGRAD_PL(128, 128, 2, 4, false, 32);
#undef BDV_WGRAD_PL
#undef BDV_WGRAD_PL2
  BDV_LAUNCH_CHECK("bdv_conv_wgrad_partial_pl");
  return BDV_OK;
}

extern "C" int bdv_wgrad_reduce_batched(const float* const* slabs, float* const* dws, const int* splits, const int64_t* numels,
                                        int n, float beta, void* stream) {
  BDV_REQUIRE(slabs && dws && splits && numels, "bdv_wgrad_reduce_batched: null pointer");
  BDV_REQUIRE(n > 0 && n <= BDV_MAX_REDUCE_ITEMS, "bdv_wgrad_reduce_batched: %d items (1..%d supported)", n, BDV_MAX_REDUCE_ITEMS);
  ReduceBatch rb;
  rb.n = n;
  int blocks = 0;
  for (int k = 0; k < n; ++k) {
    BDV_REQUIRE(slabs[k] && dws[k] && bdv_aligned16(slabs[k]) && bdv_aligned16(dws[k]), "bdv_wgrad_reduce_batched: item %d: bad pointer", k);
    BDV_REQUIRE(splits[k] > 0 && numels[k] > 0 && numels[k] % 4 == 0 && numels[k] / 4 < (1ll << 31),
                "bdv_wgrad_reduce_batched: item %d: bad size", k);
    rb.slab[k] = slabs[k];
    rb.dw[k] = dws[k];
    rb.splits[k] = splits[k];
    rb.numel4[k] = (int)(numels[k] / 4);
    rb.first_block[k] = blocks;
    blocks += (rb.numel4[k] + 63) / 64;
  }
  rb.first_block[n] = blocks;
  hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rb, beta);
  BDV_LAUNCH_CHECK("bdv_wgrad_reduce_batched");
  return BDV_OK;
}
