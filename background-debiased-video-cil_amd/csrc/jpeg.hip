// Baseline-JPEG decode for the frame pipeline: what the reference gets from
// RawFrameDecode -> mmcv.imfrombytes -> cv2.imdecode, i.e. libjpeg(-turbo) with its defaults -- ISLOW inverse DCT, "fancy" chroma
// upsampling, fixed-point YCbCr -> RGB -- bit for bit (oracle/jpeg_oracle.py is the CPU restatement, pinned by Pillow's libjpeg-turbo).
//
// Two stages, the split every hybrid GPU decoder uses: the entropy-coded segment is a serial bit stream (one Huffman symbol says
// where the next begins), so it is decoded on the HOST (bdv_jpeg_parse / bdv_jpeg_entropy_decode, thread-safe: one image per
// calling thread) into quantised coefficient blocks; everything after that is independent per block / per pixel and runs on the
// GPU for a whole batch of equal-geometry images in two launches (bdv_jpeg_reconstruct_u8): dequantise + 8x8 inverse DCT -> component
// planes, then upsample + colour conversion -> interleaved RGB uint8, the (B, H, W, 3) layout RandAugment and the front-ends take.
// Both kernels are byte / integer work bound by HBM traffic (2 bytes of coefficients in, 1 byte out per sample; then 1.5 - 3 bytes
// in, 3 out per pixel): no MFMA; coalesced 16-byte loads staged through LDS for the block kernel, aligned dword stores for the pixel
// kernel.
#include <string.h>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "common.h"

namespace {

// zigzag position -> natural (row-major) index
const unsigned char kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
  bool defined = false;
  int maxcode[18];
  int mincode[17];
  int valptr[17];
  unsigned char sym[256];
  unsigned short look[512];   // the next 9 bits of the stream -> (code length << 8) | symbol for codes of <= 9 bits, else 0
  void build(const unsigned char* counts, const unsigned char* symbols, int n) {
    memset(look, 0, sizeof(look));
    memcpy(sym, symbols, n);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valptr[len] = k;
      mincode[len] = code;
      for (int i = 0; i < counts[len - 1]; ++i, ++code, ++k)
        if (len <= 9 && k < 256 && code < (1 << len))
          for (int fill = 0; fill < (1 << (9 - len)); ++fill) look[((code << (9 - len)) | fill) & 511] = (unsigned short)((len << 8) | sym[k]);
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 1 << 30;
    defined = true;
  }
};

struct ScanComp {
  int ci, td, ta;
};

// Markers of one stream.  The Huffman tables in force at each SOS are copied into the scan (a stream may redefine them between scans).
struct Scan {
  int ncomp;
  ScanComp c[3];
  size_t pos;
  int restart_interval;
  HuffTable dc[4], ac[4];
};

struct Parsed {
  int width = 0, height = 0, ncomp = 0;
  int id[3], h[3], v[3], tq[3];
  unsigned short qt[4][64];
  bool qt_defined[4] = {false, false, false, false};
  int nscans = 0;
  Scan scans[4];
};

// thread-local parse state reused by bdv_jpeg_entropy_decode (a Scan holds 8 tables: ~10 KB, too large for a small stack frame per call chain)
thread_local Parsed g_parsed;

int parse_stream(const unsigned char* d, size_t n, Parsed& P, const char* who) {
  P = Parsed();
  BDV_REQUIRE(n >= 4 && d[0] == 0xFF && d[1] == 0xD8, "%s: not a JPEG stream (no SOI marker)", who);
  HuffTable dc[4], ac[4];
  int restart_interval = 0;
  size_t p = 2;
  while (p + 1 < n) {
    if (d[p] != 0xFF) {
      ++p;
      continue;
    }
    const int m = d[p + 1];
    p += 2;
    if (m == 0xD9) break;
    if (m == 0x01 || m == 0xFF || (m >= 0xD0 && m <= 0xD7)) {
      if (m == 0xFF) --p;  // fill byte: the next 0xFF starts the marker
      continue;
    }
    BDV_REQUIRE(p + 2 <= n, "%s: truncated marker segment", who);
    const size_t L = ((size_t)d[p] << 8) | d[p + 1];
    BDV_REQUIRE(L >= 2 && p + L <= n, "%s: marker 0x%02x: bad segment length", who, m);
    const unsigned char* seg = d + p + 2;
    const size_t sl = L - 2;
    if (m == 0xDB) {
      size_t q = 0;
      while (q < sl) {
        const int pq = seg[q] >> 4, tq = seg[q] & 15;
        BDV_REQUIRE(tq < 4 && q + 1 + (pq ? 128 : 64) <= sl, "%s: bad DQT segment", who);
        for (int i = 0; i < 64; ++i)
          P.qt[tq][kNatural[i]] = pq ? (unsigned short)((seg[q + 1 + 2 * i] << 8) | seg[q + 2 + 2 * i]) : seg[q + 1 + i];
        P.qt_defined[tq] = true;
        q += 1 + (pq ? 128 : 64);
      }
    } else if (m == 0xC0 || m == 0xC1) {
      BDV_REQUIRE(sl >= 6 && seg[0] == 8, "%s: only 8-bit samples are supported", who);
      P.height = (seg[1] << 8) | seg[2];
      P.width = (seg[3] << 8) | seg[4];
      P.ncomp = seg[5];
      BDV_REQUIRE(P.ncomp == 1 || P.ncomp == 3, "%s: %d components (grey and YCbCr only)", who, P.ncomp);
      BDV_REQUIRE(sl >= (size_t)(6 + 3 * P.ncomp) && P.width > 0 && P.height > 0, "%s: bad frame header", who);
      for (int i = 0; i < P.ncomp; ++i) {
        P.id[i] = seg[6 + 3 * i];
        P.h[i] = seg[7 + 3 * i] >> 4;
        P.v[i] = seg[7 + 3 * i] & 15;
        P.tq[i] = seg[8 + 3 * i];
        BDV_REQUIRE(P.tq[i] < 4, "%s: bad quantisation table selector", who);
      }
    } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xC7) || (m >= 0xC9 && m <= 0xCB) || (m >= 0xCD && m <= 0xCF)) {
      BDV_REQUIRE(false, "%s: SOF marker 0x%02x: only baseline / extended-sequential Huffman streams are supported (progressive, lossless and arithmetic coding are not)", who, m);
    } else if (m == 0xC4) {
      size_t q = 0;
      while (q < sl) {
        BDV_REQUIRE(q + 17 <= sl, "%s: bad DHT segment", who);
        const int tc = seg[q] >> 4, th = seg[q] & 15;
        int cnt = 0;
        for (int i = 0; i < 16; ++i) cnt += seg[q + 1 + i];
        BDV_REQUIRE(tc < 2 && th < 4 && cnt <= 256 && q + 17 + cnt <= sl, "%s: bad DHT segment", who);
        (tc ? ac : dc)[th].build(seg + q + 1, seg + q + 17, cnt);
        q += 17 + cnt;
      }
    } else if (m == 0xDD) {
      BDV_REQUIRE(sl >= 2, "%s: bad DRI segment", who);
      restart_interval = (seg[0] << 8) | seg[1];
    } else if (m == 0xDA) {
      BDV_REQUIRE(P.ncomp > 0, "%s: scan before the frame header", who);
      BDV_REQUIRE(P.nscans < 4, "%s: more than four scans", who);
      Scan& S = P.scans[P.nscans];
      S.ncomp = seg[0];
      BDV_REQUIRE(S.ncomp >= 1 && S.ncomp <= P.ncomp && sl >= (size_t)(4 + 2 * S.ncomp), "%s: bad scan header", who);
      for (int i = 0; i < S.ncomp; ++i) {
        int ci = -1;
        for (int k = 0; k < P.ncomp; ++k)
          if (P.id[k] == seg[1 + 2 * i]) ci = k;
        BDV_REQUIRE(ci >= 0, "%s: scan names an unknown component", who);
        S.c[i] = {ci, seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15};
        BDV_REQUIRE(S.c[i].td < 4 && S.c[i].ta < 4 && dc[S.c[i].td].defined && ac[S.c[i].ta].defined, "%s: scan uses an undefined Huffman table", who);
      }
      BDV_REQUIRE(seg[1 + 2 * S.ncomp] == 0 && seg[2 + 2 * S.ncomp] == 63 && seg[3 + 2 * S.ncomp] == 0, "%s: not a sequential scan", who);
      S.pos = p + L;
      S.restart_interval = restart_interval;
      for (int k = 0; k < 4; ++k) {
        S.dc[k] = dc[k];
        S.ac[k] = ac[k];
      }
      ++P.nscans;
      p += L;
      while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] != 0 && !(d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7))) ++p;
      continue;
    }
    p += L;
  }
  BDV_REQUIRE(P.ncomp > 0 && P.nscans > 0, "%s: no frame header / no scan", who);
  for (int i = 0; i < P.ncomp; ++i) BDV_REQUIRE(P.qt_defined[P.tq[i]], "%s: component %d uses an undefined quantisation table", who, i);
  if (P.ncomp == 3)
    BDV_REQUIRE(P.h[1] == 1 && P.v[1] == 1 && P.h[2] == 1 && P.v[2] == 1 &&
                    ((P.h[0] == 1 && P.v[0] == 1) || (P.h[0] == 2 && P.v[0] == 1) || (P.h[0] == 2 && P.v[0] == 2)),
                "%s: sampling factors %dx%d,%dx%d,%dx%d: only 4:4:4, 4:2:2 and 4:2:0 are supported", who, P.h[0], P.v[0], P.h[1], P.v[1], P.h[2], P.v[2]);
  else
    P.h[0] = P.v[0] = 1;   // a single component is never subsampled, whatever the header says (jdinput.c)
  return BDV_OK;
}

void fill_info(const Parsed& P, bdv_jpeg_info* info) {
  memset(info, 0, sizeof(*info));
  info->width = P.width;
  info->height = P.height;
  info->ncomp = P.ncomp;
  int hmax = 1, vmax = 1;
  for (int i = 0; i < P.ncomp; ++i) {
    hmax = P.h[i] > hmax ? P.h[i] : hmax;
    vmax = P.v[i] > vmax ? P.v[i] : vmax;
  }
  const int mcux = (P.width + 8 * hmax - 1) / (8 * hmax), mcuy = (P.height + 8 * vmax - 1) / (8 * vmax);
  long long off = 0;
  for (int i = 0; i < P.ncomp; ++i) {
    info->h[i] = P.h[i];
    info->v[i] = P.v[i];
    info->blocks_w[i] = mcux * P.h[i];
    info->blocks_h[i] = mcuy * P.v[i];
    info->down_w[i] = (P.width * P.h[i] + hmax - 1) / hmax;
    info->down_h[i] = (P.height * P.v[i] + vmax - 1) / vmax;
    info->coef_offset[i] = off;
    off += (long long)info->blocks_w[i] * info->blocks_h[i] * 64;
    memcpy(info->qt[i], P.qt[P.tq[i]], sizeof(info->qt[i]));
  }
  info->coef_count = off;
}

// MSB-first reader of an entropy-coded segment: 0xFF00 -> 0xFF; at a marker it feeds zeros and stays there
struct BitReader {
  const unsigned char* d;
  size_t n, p;
  uint64_t acc = 0;
  int have = 0;
  void fill() {
    while (have <= 48) {
      unsigned b = 0;
      if (p < n) {
        b = d[p];
        if (b == 0xFF) {
          const unsigned nxt = p + 1 < n ? d[p + 1] : 0xD9;
          if (nxt == 0)
            p += 2;
          else
            b = 0;
        } else {
          ++p;
        }
      }
      acc = (acc << 8) | b;
      have += 8;
    }
  }
  // one symbol needs at most 16 code bits + 15 value bits: top the window up once per symbol, then peek / skip without checks
  inline void ensure() {
    if (have < 32) fill();
  }
  inline unsigned peek(int k) const { return (unsigned)(acc >> (have - k)) & ((1u << k) - 1); }
  inline void skip(int k) { have -= k; }
  inline int get(int k) {   // after ensure(): k <= 16
    if (k == 0) return 0;
    have -= k;
    return (int)((acc >> have) & ((1u << k) - 1));
  }
  void restart() {
    acc = 0;
    have = 0;
    while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) ++p;
    p += 2;
  }
};

inline int huff_decode(BitReader& br, const HuffTable& t) {
  br.ensure();
  const unsigned e = t.look[br.peek(9)];
  if (e) {
    br.skip(e >> 8);
    return e & 255;
  }
  int len = 10;
  int code = (int)br.peek(10);
  while (code > t.maxcode[len]) {
    if (++len > 16) {
      br.skip(16);
      return 0;   // not a code of this table (corrupt data): decode on, as the IJG decoder does after its warning
    }
    code = (int)br.peek(len);
  }
  br.skip(len);
  return t.sym[(t.valptr[len] + code - t.mincode[len]) & 255];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

inline void decode_block(BitReader& br, const HuffTable& dct, const HuffTable& act, int& pred, short* blk) {
  int s = huff_decode(br, dct);
  if (s > 16) s = 16;   // (corrupt data: no valid DC category is that large; keep the bit window's contract)
  if (s) pred += extend(br.get(s), s);
  blk[0] = (short)pred;
  for (int k = 1; k < 64;) {
    const int rs = huff_decode(br, act), r = rs >> 4;
    s = rs & 15;
    if (s) {
      k += r;
      if (k > 63) break;
      blk[kNatural[k]] = (short)extend(br.get(s), s);
      ++k;
    } else if (r == 15) {
      k += 16;
    } else {
      break;
    }
  }
}

// ---- device: dequantise + ISLOW inverse DCT (the published IJG algorithm: 13-bit constants, 2 extra bits after pass 1) ----------
struct JpegGeom {
  int W, H, ncomp;
  int bw[3], bh[3], dw[3], dh[3], hs[3], vs[3];   // hs / vs: upsampling factors of the component (hmax / h, vmax / v)
  long long coef_off[3], coef_count;
  long long plane_off[3], plane_bytes;           // byte offsets of the component planes inside one image's workspace
  int blocks_total, block_first[4];
};

__device__ __forceinline__ void idct_1d(int (&v)[8], int shift) {
  int z2 = v[2], z3 = v[6];
  int z1 = (z2 + z3) * 4433;
  int tmp2 = z1 + z3 * (-15137);
  int tmp3 = z1 + z2 * 6270;
  z2 = v[0];
  z3 = v[4];
  int tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = v[7];
  tmp1 = v[5];
  tmp2 = v[3];
  tmp3 = v[1];
  z1 = tmp0 + tmp3;
  z2 = tmp1 + tmp2;
  z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * 9633;
  tmp0 *= 2446;
  tmp1 *= 16819;
  tmp2 *= 25172;
  tmp3 *= 12299;
  z1 *= -7373;
  z2 *= -20995;
  z3 = z3 * (-16069) + z5;
  z4 = z4 * (-3196) + z5;
  tmp0 += z1 + z3;
  tmp1 += z2 + z4;
  tmp2 += z2 + z3;
  tmp3 += z1 + z4;
  const int r = 1 << (shift - 1);
  v[0] = (tmp10 + tmp3 + r) >> shift;
  v[7] = (tmp10 - tmp3 + r) >> shift;
  v[1] = (tmp11 + tmp2 + r) >> shift;
  v[6] = (tmp11 - tmp2 + r) >> shift;
  v[2] = (tmp12 + tmp1 + r) >> shift;
  v[5] = (tmp12 - tmp1 + r) >> shift;
  v[3] = (tmp13 + tmp0 + r) >> shift;
  v[4] = (tmp13 - tmp0 + r) >> shift;
}

// the decoder's post-IDCT range-limit table, indexed with (x & 1023): clamp(x + 128, 0, 255) for |x| < 512
__device__ __forceinline__ unsigned range_limit_idct(int x) {
  const int i = x & 1023;
  return i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896;
}

typedef short s16x8 __attribute__((ext_vector_type(8)));

// One thread per 8x8 block, 256 blocks per workgroup.  The coefficients of consecutive blocks are contiguous (128 bytes each, also
// across components and images), so the workgroup's 32 KB are fetched with fully coalesced 16-byte loads and handed to their threads
// through LDS: block b's chunk r sits at b * 144 + r * 16 -- the 16-byte pad per block makes the per-thread 16-byte reads of 16
// neighbouring lanes fall on all 64 banks once.  A thread then holds its block in registers for both passes and writes 8 x 8 bytes;
// neighbouring threads are neighbouring blocks of a block row, i.e. their row stores are contiguous.
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const short* __restrict__ coefs, const unsigned short* __restrict__ qts,
                                                         unsigned char* __restrict__ planes, JpegGeom g, int B) {
  __shared__ __attribute__((aligned(16))) unsigned char stage[256 * 144];
  const unsigned nblk = (unsigned)g.blocks_total * (unsigned)B;     // < 2^31 (checked on the host): 32-bit index arithmetic throughout
  const unsigned first = blockIdx.x * 256u;
  const int tid = threadIdx.x;
  const uint4* src4 = reinterpret_cast<const uint4*>(coefs) + (size_t)first * 8;   // 8 chunks of 16 bytes per block
  const unsigned chunks_left = (nblk - first < 256u ? nblk - first : 256u) * 8u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = i * 256 + tid;
    if ((unsigned)c < chunks_left) *reinterpret_cast<uint4*>(stage + (c >> 3) * 144 + (c & 7) * 16) = src4[c];
  }
  __syncthreads();
  const unsigned t = first + tid;
  if (t >= nblk) return;
  const int img = (int)(t / (unsigned)g.blocks_total), bi = (int)(t - (unsigned)img * (unsigned)g.blocks_total);
  const int c = bi >= g.block_first[2] ? 2 : bi >= g.block_first[1] ? 1 : 0;
  const int b = bi - g.block_first[c];
  const int by = b / g.bw[c], bx = b - by * g.bw[c];
  const unsigned short* qt = qts + ((size_t)img * 3 + c) * 64;
  int m[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const s16x8 cv = *reinterpret_cast<const s16x8*>(stage + tid * 144 + r * 16);
    const s16x8 qv = *reinterpret_cast<const s16x8*>(qt + 8 * r);
#pragma unroll
    for (int k = 0; k < 8; ++k) m[r][k] = (int)cv[k] * (int)(unsigned short)qv[k];
  }
  // pass 1: columns
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int col[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) col[r] = m[r][k];
    idct_1d(col, 13 - 2);
#pragma unroll
    for (int r = 0; r < 8; ++r) m[r][k] = col[r];
  }
  // pass 2: rows -> samples
  unsigned char* dst = planes + (size_t)img * g.plane_bytes + g.plane_off[c] + ((size_t)by * 8 * g.bw[c] + bx) * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    idct_1d(m[r], 13 + 2 + 3);
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lo |= range_limit_idct(m[r][k]) << (8 * k);
      hi |= range_limit_idct(m[r][k + 4]) << (8 * k);
    }
    *reinterpret_cast<uint2*>(dst + (size_t)r * 8 * g.bw[c]) = make_uint2(lo, hi);
  }
}

// chroma sample at luma position (x, y): the decoder's "fancy" triangle filters (3/4 nearer + 1/4 further sample, alternating
// rounding), over the REAL samples only (the padding of the block grid never takes part); plain replication for <= 2 columns
__device__ __forceinline__ int chroma_at(const unsigned char* __restrict__ p, int pitch, int dw, int dh, int hs, int vs, int x, int y) {
  if (hs == 1 && vs == 1) return p[(size_t)y * pitch + x];
  const int c = x >> 1;
  if (dw <= 2) return p[(size_t)(vs == 2 ? y >> 1 : y) * pitch + c];
  if (vs == 1) {   // h2v1
    const int s = p[(size_t)y * pitch + c];
    if (x & 1) return c == dw - 1 ? s : (3 * s + p[(size_t)y * pitch + c + 1] + 2) >> 2;
    return c == 0 ? s : (3 * s + p[(size_t)y * pitch + c - 1] + 1) >> 2;
  }
  // h2v2: column sums of this row and its nearer neighbour (the edge rows see themselves)
  const int i = y >> 1;
  int nb = (y & 1) ? i + 1 : i - 1;
  nb = nb < 0 ? 0 : nb >= dh ? dh - 1 : nb;
  const unsigned char* r0 = p + (size_t)i * pitch;
  const unsigned char* r1 = p + (size_t)nb * pitch;
  const int cs = 3 * r0[c] + r1[c];
  if (x & 1) return c == dw - 1 ? (4 * cs + 7) >> 4 : (3 * cs + 3 * r0[c + 1] + r1[c + 1] + 7) >> 4;
  return c == 0 ? (4 * cs + 8) >> 4 : (3 * cs + 3 * r0[c - 1] + r1[c - 1] + 8) >> 4;
}

__device__ __forceinline__ unsigned clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

// Fixed-point YCbCr -> RGB with 16 fraction bits and the decoder's rounding; one pixel -> packed 0x00BBGGRR.
__device__ __forceinline__ unsigned jpeg_pixel(const unsigned char* __restrict__ base, const JpegGeom& g, int x, int y) {
  const int Y = base[g.plane_off[0] + (size_t)y * 8 * g.bw[0] + x];
  if (g.ncomp == 1) return (unsigned)Y * 0x010101u;
  const int cb = chroma_at(base + g.plane_off[1], 8 * g.bw[1], g.dw[1], g.dh[1], g.hs[1], g.vs[1], x, y) - 128;
  const int cr = chroma_at(base + g.plane_off[2], 8 * g.bw[2], g.dw[2], g.dh[2], g.hs[2], g.vs[2], x, y) - 128;
  const unsigned r = clamp255(Y + ((91881 * cr + 32768) >> 16));
  const unsigned gg = clamp255(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
  const unsigned b = clamp255(Y + ((116130 * cb + 32768) >> 16));
  return r | (gg << 8) | (b << 16);
}

// grid (pixel groups of one image, image): four consecutive pixels of an image per thread (a group may straddle a row end), ONE
// integer division per thread; when an image is a whole number of dwords (W * H % 4 == 0) the 12 bytes leave as three aligned dword
// stores -- the first form of this kernel spent its time on a 64-bit division per pixel and on byte stores.
__global__ __launch_bounds__(256) void jpeg_color_kernel(const unsigned char* __restrict__ planes, unsigned char* __restrict__ rgb, JpegGeom g) {
  const unsigned npix = (unsigned)g.W * (unsigned)g.H;
  const unsigned p0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
  if (p0 >= npix) return;
  const unsigned img = blockIdx.y;
  const unsigned char* base = planes + (size_t)img * g.plane_bytes;
  unsigned char* out = rgb + ((size_t)img * npix + p0) * 3;
  int y = (int)(p0 / (unsigned)g.W), x = (int)(p0 - (unsigned)y * (unsigned)g.W);
  unsigned px[4];
  const int n = npix - p0 < 4u ? (int)(npix - p0) : 4;
  if (g.ncomp == 3 && g.hs[1] == 2 && g.vs[1] == 2 && g.dw[1] > 2 && x + 3 < g.W) {
    // 4:2:0, the four pixels in one row: they need the chroma column sums (3 * this row + the nearer row) of at most five
    // neighbouring columns, loaded ONCE for the group; with the column index clamped the edge forms of the filter
    // ((4 cs + 8) >> 4, (4 cs + 7) >> 4) are the interior forms with cs[c - 1] = cs[c] resp. cs[c + 1] = cs[c]
    const int dw = g.dw[1], dh = g.dh[1], i = y >> 1, c_lo = x >> 1;
    int nb = (y & 1) ? i + 1 : i - 1;
    nb = nb < 0 ? 0 : nb >= dh ? dh - 1 : nb;
    int cs[2][5];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const unsigned char* cp = base + g.plane_off[1 + pl];
      const int pitch = 8 * g.bw[1 + pl];
      const unsigned char* r0 = cp + (size_t)i * pitch;
      const unsigned char* r1 = cp + (size_t)nb * pitch;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        int c = c_lo - 1 + j;
        c = c < 0 ? 0 : c >= dw ? dw - 1 : c;
        cs[pl][j] = 3 * r0[c] + r1[c];
      }
    }
    const unsigned char* yp = base + g.plane_off[0] + (size_t)y * 8 * g.bw[0] + x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xi = x + k, j = (xi >> 1) - c_lo + 1;      // 1 .. 3
      int cc[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        cc[pl] = ((xi & 1) ? (3 * cs[pl][j] + cs[pl][j + 1] + 7) >> 4 : (3 * cs[pl][j] + cs[pl][j - 1] + 8) >> 4) - 128;
      const int Y = yp[k], cb = cc[0], cr = cc[1];
      const unsigned r = clamp255(Y + ((91881 * cr + 32768) >> 16));
      const unsigned gg = clamp255(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
      const unsigned b = clamp255(Y + ((116130 * cb + 32768) >> 16));
      px[k] = r | (gg << 8) | (b << 16);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      px[k] = k < n ? jpeg_pixel(base, g, x, y) : 0u;
      if (++x == g.W) {
        x = 0;
        ++y;
      }
    }
  }
  if (n == 4 && (npix & 3u) == 0u) {
    unsigned* o = reinterpret_cast<unsigned*>(out);
    o[0] = px[0] | (px[1] << 24);
    o[1] = (px[1] >> 8) | (px[2] << 16);
    o[2] = (px[2] >> 16) | (px[3] << 8);
  } else {
    for (int k = 0; k < n; ++k) {
      out[3 * k] = (unsigned char)px[k];
      out[3 * k + 1] = (unsigned char)(px[k] >> 8);
      out[3 * k + 2] = (unsigned char)(px[k] >> 16);
    }
  }
}

int make_jpeg_geom(const bdv_jpeg_info* info, JpegGeom& g, const char* who) {
  BDV_REQUIRE(info != nullptr, "%s: null geometry", who);
  BDV_REQUIRE((info->ncomp == 1 || info->ncomp == 3) && info->width > 0 && info->height > 0 && info->width <= 65535 && info->height <= 65535,
              "%s: bad geometry (%d components, %d x %d)", who, info->ncomp, info->width, info->height);
  memset(&g, 0, sizeof(g));
  g.W = info->width;
  g.H = info->height;
  g.ncomp = info->ncomp;
  int hmax = 1, vmax = 1;
  for (int c = 0; c < g.ncomp; ++c) {
    BDV_REQUIRE(info->h[c] >= 1 && info->h[c] <= 2 && info->v[c] >= 1 && info->v[c] <= 2, "%s: sampling factor of component %d", who, c);
    hmax = info->h[c] > hmax ? info->h[c] : hmax;
    vmax = info->v[c] > vmax ? info->v[c] : vmax;
  }
  long long coff = 0, poff = 0;
  int nblk = 0;
  for (int c = 0; c < g.ncomp; ++c) {
    const int mcux = (g.W + 8 * hmax - 1) / (8 * hmax), mcuy = (g.H + 8 * vmax - 1) / (8 * vmax);
    // the caller's struct must be what bdv_jpeg_parse produces for these sizes: the kernels index with it
    BDV_REQUIRE(info->blocks_w[c] == mcux * info->h[c] && info->blocks_h[c] == mcuy * info->v[c] &&
                    info->down_w[c] == (g.W * info->h[c] + hmax - 1) / hmax && info->down_h[c] == (g.H * info->v[c] + vmax - 1) / vmax &&
                    info->coef_offset[c] == coff,
                "%s: geometry fields of component %d are inconsistent with the image size", who, c);
    g.bw[c] = info->blocks_w[c];
    g.bh[c] = info->blocks_h[c];
    g.dw[c] = info->down_w[c];
    g.dh[c] = info->down_h[c];
    g.hs[c] = hmax / info->h[c];
    g.vs[c] = vmax / info->v[c];
    BDV_REQUIRE(c == 0 ? (g.hs[c] == 1 && g.vs[c] == 1) : ((g.hs[c] == 1 && g.vs[c] == 1) || (g.hs[c] == 2 && g.vs[c] <= 2)),
                "%s: component %d: only 4:4:4, 4:2:2 and 4:2:0 layouts", who, c);
    g.coef_off[c] = coff;
    g.plane_off[c] = poff;
    g.block_first[c] = nblk;
    coff += (long long)g.bw[c] * g.bh[c] * 64;
    poff += (long long)g.bw[c] * g.bh[c] * 64;
    nblk += g.bw[c] * g.bh[c];
  }
  for (int c = g.ncomp; c < 4; ++c) g.block_first[c] = 1 << 30;
  BDV_REQUIRE(info->coef_count == coff, "%s: coef_count is inconsistent with the block grids", who);
  g.coef_count = coff;
  g.plane_bytes = poff;
  g.blocks_total = nblk;
  return BDV_OK;
}

}  // namespace

extern "C" int bdv_jpeg_parse(const unsigned char* data, size_t n, bdv_jpeg_info* info) {
  BDV_REQUIRE(data != nullptr && info != nullptr, "bdv_jpeg_parse: null pointer");
  Parsed& P = g_parsed;
  if (int e = parse_stream(data, n, P, "bdv_jpeg_parse")) return e;
  fill_info(P, info);
  return BDV_OK;
}

extern "C" int bdv_jpeg_entropy_decode(const unsigned char* data, size_t n, const bdv_jpeg_info* info, short* coefs) {
  BDV_REQUIRE(data != nullptr && info != nullptr && coefs != nullptr, "bdv_jpeg_entropy_decode: null pointer");
  Parsed& P = g_parsed;
  if (int e = parse_stream(data, n, P, "bdv_jpeg_entropy_decode")) return e;
  bdv_jpeg_info mine;
  fill_info(P, &mine);
  // everything but the quantisation tables must be what the caller sized its buffers for
  BDV_REQUIRE(mine.width == info->width && mine.height == info->height && mine.ncomp == info->ncomp && mine.coef_count == info->coef_count &&
                  memcmp(mine.h, info->h, sizeof(mine.h)) == 0 && memcmp(mine.v, info->v, sizeof(mine.v)) == 0,
              "bdv_jpeg_entropy_decode: the stream (%d x %d, %d components) does not have the geometry of `info` (%d x %d, %d)", mine.width,
              mine.height, mine.ncomp, info->width, info->height, info->ncomp);
  memset(coefs, 0, (size_t)mine.coef_count * sizeof(short));
  for (int si = 0; si < P.nscans; ++si) {
    const Scan& S = P.scans[si];
    BitReader br{data, n, S.pos};
    int pred[3] = {0, 0, 0};
    int ux_n, uy_n;
    if (S.ncomp > 1) {
      ux_n = mine.blocks_w[0] / mine.h[0];
      uy_n = mine.blocks_h[0] / mine.v[0];
    } else {   // a one-component scan runs over the component's own block grid, without the MCU padding
      ux_n = (mine.down_w[S.c[0].ci] + 7) / 8;
      uy_n = (mine.down_h[S.c[0].ci] + 7) / 8;
    }
    int unit = 0;
    for (int uy = 0; uy < uy_n; ++uy)
      for (int ux = 0; ux < ux_n; ++ux, ++unit) {
        if (S.restart_interval && unit && unit % S.restart_interval == 0) {
          br.restart();
          pred[0] = pred[1] = pred[2] = 0;
        }
        for (int k = 0; k < S.ncomp; ++k) {
          const int ci = S.c[k].ci;
          const int nh = S.ncomp > 1 ? mine.h[ci] : 1, nv = S.ncomp > 1 ? mine.v[ci] : 1;
          for (int by = 0; by < nv; ++by)
            for (int bx = 0; bx < nh; ++bx) {
              const int X = ux * nh + bx, Y = uy * nv + by;
              short* blk = coefs + mine.coef_offset[ci] + ((size_t)Y * mine.blocks_w[ci] + X) * 64;
              decode_block(br, S.dc[S.c[k].td], S.ac[S.c[k].ta], pred[ci], blk);
            }
        }
      }
  }
  return BDV_OK;
}

extern "C" int bdv_jpeg_entropy_decode_batch(const unsigned char* const* data, const size_t* sizes, int n, const bdv_jpeg_info* info,
                                             short* coefs, unsigned short* qts, int threads) {
  BDV_REQUIRE(data && sizes && info && coefs && qts && n > 0, "bdv_jpeg_entropy_decode_batch: null pointer / empty batch");
  BDV_REQUIRE(threads >= 1 && threads <= 256, "bdv_jpeg_entropy_decode_batch: %d threads (1..256)", threads);
  std::atomic<int> next(0), failed(0);
  std::mutex mu;
  std::string first_error;
  int first_code = BDV_OK;
  auto worker = [&]() {
    for (int i = next.fetch_add(1); i < n && !failed.load(); i = next.fetch_add(1)) {
      short* dst = coefs + (size_t)i * info->coef_count;
      int e = data[i] ? bdv_jpeg_entropy_decode(data[i], sizes[i], info, dst) : BDV_EINVAL;
      if (e == BDV_OK) {   // the tables of THIS stream (bdv_jpeg_entropy_decode has just parsed it on this thread)
        bdv_jpeg_info mine;
        fill_info(g_parsed, &mine);
        memcpy(qts + (size_t)i * 3 * 64, mine.qt, sizeof(mine.qt));
      } else {
        std::lock_guard<std::mutex> lock(mu);
        if (!failed.exchange(1)) {
          first_code = e;
          first_error = "image " + std::to_string(i) + ": " + (data[i] ? bdv_last_error() : "null stream");
        }
      }
    }
  };
  const int nt = threads < n ? threads : n;
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  if (failed.load()) {
    bdv_set_error("bdv_jpeg_entropy_decode_batch: %s", first_error.c_str());
    return first_code;
  }
  return BDV_OK;
}

extern "C" size_t bdv_jpeg_workspace_bytes(const bdv_jpeg_info* info, int B) {
  JpegGeom g;
  if (B <= 0 || make_jpeg_geom(info, g, "bdv_jpeg_workspace_bytes") != BDV_OK) return 0;
  return (size_t)g.plane_bytes * B;
}

extern "C" int bdv_jpeg_reconstruct_u8(const short* coefs, const unsigned short* qts, const bdv_jpeg_info* info, int B, void* workspace,
                                       size_t workspace_bytes, unsigned char* rgb, void* stream) {
  JpegGeom g;
  if (int e = make_jpeg_geom(info, g, "bdv_jpeg_reconstruct_u8")) return e;
  BDV_REQUIRE(coefs && qts && workspace && rgb && B > 0, "bdv_jpeg_reconstruct_u8: null pointer / empty batch");
  BDV_REQUIRE((((uintptr_t)rgb) & 3) == 0, "bdv_jpeg_reconstruct_u8: rgb must be 4-byte aligned");
  BDV_REQUIRE(bdv_aligned16(coefs) && bdv_aligned16(qts) && bdv_aligned16(workspace), "bdv_jpeg_reconstruct_u8: coefs, qts and workspace must be 16-byte aligned");
  BDV_REQUIRE(workspace_bytes >= (size_t)g.plane_bytes * B, "bdv_jpeg_reconstruct_u8: workspace %zu < required %zu bytes", workspace_bytes, (size_t)g.plane_bytes * B);
  BDV_REQUIRE((long long)g.blocks_total * B < (1ll << 31) && B <= 65535 && (long long)g.W * g.H < (1ll << 31), "bdv_jpeg_reconstruct_u8: batch too large for one launch (at most 65535 images)");
  hipStream_t s = (hipStream_t)stream;
  const int nb = (int)(((long long)g.blocks_total * B + 255) / 256);
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3(nb), dim3(256), 0, s, coefs, qts, (unsigned char*)workspace, g, B);
  BDV_LAUNCH_CHECK("bdv_jpeg_reconstruct_u8(idct)");
  const unsigned groups = ((unsigned)g.W * (unsigned)g.H + 3u) / 4u;   // four pixels per thread
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((groups + 255u) / 256u, (unsigned)B), dim3(256), 0, s, (const unsigned char*)workspace, rgb, g);
  BDV_LAUNCH_CHECK("bdv_jpeg_reconstruct_u8(color)");
  return BDV_OK;
}
