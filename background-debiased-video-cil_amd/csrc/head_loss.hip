// Classifier heads (LSC cosine classifier, IncrementalNet linear), segment consensus, dropout,
// fused losses (LSCLoss, soft-target CE), top-k accuracy and the feature-distillation MSE.
// All tensors here are tiny next to the conv stack ((B*T, D) x (K, D)); kernels favour
// determinism (fixed-order reductions, no atomics) over peak rate.
#include "common.h"

namespace {

constexpr float COS_EPS = 1e-8f;  // F.cosine_similarity default eps

// ---------------------------------------------------------------------------------------
// dot[n][j] = <x_n, w_j>  (+ optional norms).  A block stages RB rows of x in LDS so that every w row it streams from
// L2 is used RB times (the one-row version re-read all of w per row: 212 MB of L2 traffic for 256 x 2048 x 101).
// MODE 0: linear (out = dot + bias); MODE 1: LSC (cos + proxy reduction).  Per (n, j) the summation order is the same
// for every RB.
// ---------------------------------------------------------------------------------------
constexpr int HEAD_RB = 4;

template <int MODE>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        float* __restrict__ xnorm, float* __restrict__ wnorm,
                                                        float* __restrict__ cosbuf, int N, int D, int K, int P) {
  constexpr int RB = HEAD_RB;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;               // RB x D
  float* cs = sm + RB * D;      // RB x K*P
  __shared__ float red[RB][4];
  const int n0 = blockIdx.x * RB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KP = K * P;
  const int rows = min(RB, N - n0);
  float ss[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    ss[r] = 0.f;
    if (r < rows)
      for (int d = tid; d < D; d += 256) {
        const float v = x[(size_t)(n0 + r) * D + d];
        xs[r * D + d] = v;
        ss[r] += v * v;
      }
    else
      for (int d = tid; d < D; d += 256) xs[r * D + d] = 0.f;
    ss[r] = wave_sum(ss[r]);
    if (lane == 0) red[r][wave] = ss[r];
  }
  __syncthreads();
  float nx[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    nx[r] = 1.f;
    if (MODE == 1) {
      nx[r] = fmaxf(sqrtf(red[r][0] + red[r][1] + red[r][2] + red[r][3]), COS_EPS);
      if (tid == 0 && r < rows) xnorm[n0 + r] = nx[r];
    }
  }
  for (int j = wave; j < KP; j += 4) {
    const float* wr = w + (size_t)j * D;
    float dot[RB], wq = 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r) dot[r] = 0.f;
#pragma unroll 8   // eight weight loads in flight per lane: the loop is a chain of L2 latencies otherwise (0.29 ms for 256 x 2048 x 101)
    for (int d = lane; d < D; d += 64) {
      const float wv = wr[d];
#pragma unroll
      for (int r = 0; r < RB; ++r) dot[r] += xs[r * D + d] * wv;
      if (MODE == 1) wq += wv * wv;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) dot[r] = wave_sum(dot[r]);
    if (MODE == 1) {
      wq = wave_sum(wq);
      const float nw = fmaxf(sqrtf(wq), COS_EPS);
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
          if (r < rows) {
            const float c = dot[r] / (nx[r] * nw);
            cs[r * KP + j] = c;
            cosbuf[(size_t)(n0 + r) * KP + j] = c;
          }
        if (n0 == 0) wnorm[j] = nw;
      }
    } else if (lane == 0) {
#pragma unroll
      for (int r = 0; r < RB; ++r)
        if (r < rows) out[(size_t)(n0 + r) * K + j] = dot[r] + (bias != nullptr ? bias[j] : 0.f);
    }
  }
  if (MODE == 1) {
    __syncthreads();
    for (int q = tid; q < rows * K; q += 256) {
      const int r = q / K, k = q - r * K;
      const float* c = cs + r * KP + k * P;
      float mx = c[0];
      for (int p = 1; p < P; ++p) mx = fmaxf(mx, c[p]);
      float den = 0.f, num = 0.f;
      for (int p = 0; p < P; ++p) {
        const float e = expf(c[p] - mx);
        den += e;
        num += e * c[p];
      }
      out[(size_t)(n0 + r) * K + k] = num / den;
    }
  }
}

// LSC backward, part A (grid N): dcos[n,:] from dsim; dx[n,:] = sum_j G[n,j] w_j - a_n x_n
__global__ __launch_bounds__(256) void lsc_bwd_dx_kernel(const float* __restrict__ dsim, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ xnorm,
                                                          const float* __restrict__ wnorm, const float* __restrict__ cosbuf,
                                                          float* __restrict__ dx, float* __restrict__ dcos_ws, int N, int D, int K,
                                                          int P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* G = sm;  // K*P
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KP = K * P;
  const float nx = xnorm[n];
  float part = 0.f;
  for (int k = tid; k < K; k += 256) {
    const float* c = cosbuf + (size_t)n * KP + k * P;
    float mx = c[0];
    for (int p = 1; p < P; ++p) mx = fmaxf(mx, c[p]);
    float den = 0.f, num = 0.f;
    for (int p = 0; p < P; ++p) {
      const float e = expf(c[p] - mx);
      den += e;
      num += e * c[p];
    }
    const float sbar = num / den;
    const float g = dsim[(size_t)n * K + k];
    for (int p = 0; p < P; ++p) {
      const float a = expf(c[p] - mx) / den;
      const float dc = g * a * (1.f + c[p] - sbar);
      dcos_ws[(size_t)n * KP + k * P + p] = dc;
      G[k * P + p] = dc / (nx * wnorm[k * P + p]);
      part += dc * c[p];
    }
  }
  part = wave_sum(part);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  const float a_n = (red[0] + red[1] + red[2] + red[3]) / (nx * nx);
  for (int d = tid; d < D; d += 256) {
    float acc = 0.f;
    for (int j = 0; j < KP; ++j) acc += G[j] * w[(size_t)j * D + d];
    dx[(size_t)n * D + d] = acc - a_n * x[(size_t)n * D + d];
  }
}

// LSC backward, part B (grid K*P): dw[j,:] = beta*dw + sum_n G[n,j] x_n - b_j w_j
__global__ __launch_bounds__(256) void lsc_bwd_dw_kernel(const float* __restrict__ dcos, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ xnorm,
                                                          const float* __restrict__ wnorm, const float* __restrict__ cosbuf,
                                                          float* __restrict__ dw, float beta, int N, int D, int KP) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* G = sm;  // N
  __shared__ float red[4];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float nw = wnorm[j];
  float part = 0.f;
  for (int n = tid; n < N; n += 256) {
    const float dc = dcos[(size_t)n * KP + j];
    G[n] = dc / (xnorm[n] * nw);
    part += dc * cosbuf[(size_t)n * KP + j];
  }
  part = wave_sum(part);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  const float b_j = (red[0] + red[1] + red[2] + red[3]) / (nw * nw);
  for (int d = tid; d < D; d += 256) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += G[n] * x[(size_t)n * D + d];
    float v = acc - b_j * w[(size_t)j * D + d];
    if (beta != 0.f) v += beta * dw[(size_t)j * D + d];
    dw[(size_t)j * D + d] = v;
  }
}

// linear backward: dx[n,:] = sum_k dout[n,k] w_k   (grid N)
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                             float* __restrict__ dx, int N, int D, int K) {
  constexpr int RB = HEAD_RB;  // rows per block: each w element loaded once per RB rows
  extern __shared__ __attribute__((aligned(16))) float sm[];  // RB x K
  const int n0 = blockIdx.x * RB, tid = threadIdx.x;
  const int rows = min(RB, N - n0);
  for (int q = tid; q < RB * K; q += 256) {
    const int r = q / K, k = q - r * K;
    sm[q] = r < rows ? dout[(size_t)(n0 + r) * K + k] : 0.f;
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    float acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; ++k) {
      const float wv = w[(size_t)k * D + d];
#pragma unroll
      for (int r = 0; r < RB; ++r) acc[r] += sm[r * K + k] * wv;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
      if (r < rows) dx[(size_t)(n0 + r) * D + d] = acc[r];
  }
}

__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                             float* __restrict__ dw, float* __restrict__ db, float beta, int N, int D,
                                                             int K) {
  constexpr int KB = 8;  // classes per block: each x element loaded once per KB classes; grid = (ceil(K/KB), ceil(D/256))
  extern __shared__ __attribute__((aligned(16))) float sm[];  // N x KB
  const int k0 = blockIdx.x * KB, tid = threadIdx.x, d = blockIdx.y * 256 + tid;
  const int ks = min(KB, K - k0);
  for (int q = tid; q < N * KB; q += 256) {
    const int n = q / KB, kk = q - n * KB;
    sm[q] = kk < ks ? dout[(size_t)n * K + k0 + kk] : 0.f;
  }
  __syncthreads();
  if (d < D) {
    float acc[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) acc[kk] = 0.f;
    for (int n = 0; n < N; ++n) {
      const float xv = x[(size_t)n * D + d];
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) acc[kk] += sm[n * KB + kk] * xv;
    }
#pragma unroll
    for (int kk = 0; kk < KB; ++kk)
      if (kk < ks) {
        float v = acc[kk];
        if (beta != 0.f) v += beta * dw[(size_t)(k0 + kk) * D + d];
        dw[(size_t)(k0 + kk) * D + d] = v;
      }
  }
  if (db != nullptr && blockIdx.y == 0 && tid < ks) {
    float t = 0.f;
    for (int n = 0; n < N; ++n) t += sm[n * KB + tid];
    db[k0 + tid] = (beta != 0.f ? beta * db[k0 + tid] : 0.f) + t;
  }
}

__global__ void consensus_fwd_kernel(const float* __restrict__ s, float* __restrict__ out, int B, int T, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K, k = i - b * K;
  float acc = 0.f;
  for (int t = 0; t < T; ++t) acc += s[((size_t)b * T + t) * K + k];
  out[i] = acc / (float)T;
}

__global__ void consensus_bwd_kernel(const float* __restrict__ dout, float* __restrict__ ds, int B, int T, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T * K) return;
  const int k = i % K, b = i / (T * K);
  ds[i] = dout[(size_t)b * K + k] / (float)T;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t numel, float p, float scale,
                               uint64_t seed) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const uint64_t h = splitmix64(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);  // [0,1)
    out[i] = u >= p ? x[i] * scale : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// LSCLoss (lsc_loss.py:36-56): one block, waves stride over rows, fixed-order final reduce.
// ---------------------------------------------------------------------------------------
// class_weights (lsc_loss.py:50-51, optional): the row's term num - log(den) is scaled by class_weights[target] BEFORE the negation
// and the hinge, i.e. l_i = max(w_i * (log den - num), 0); a weight of 1 leaves every value bit for bit as without weights.
__global__ __launch_bounds__(256) void lsc_loss_kernel(const float* __restrict__ sim, const int64_t* __restrict__ targets,
                                                        const float* __restrict__ eta_p, float margin, int hinge,
                                                        const float* __restrict__ class_weights,
                                                        float* __restrict__ loss, float* __restrict__ dsim, float* __restrict__ deta,
                                                        int B, int K) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float eta = eta_p[0];
  const float invB = 1.f / (float)B;
  float loss_acc = 0.f, deta_acc = 0.f;
  for (int b = wave; b < B; b += 4) {
    const float* row = sim + (size_t)b * K;
    const int y = (int)targets[b];
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
      const float s = eta * (row[k] - margin);
      if (s > mx) { mx = s; am = k; }
    }
    // wave argmax with first-index tie break
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64);
      const int oa = __shfl_xor(am, o, 64);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    float esum = 0.f;
    for (int k = lane; k < K; k += 64)
      if (k != y) esum += expf(eta * (row[k] - margin) - mx);
    esum = wave_sum(esum);
    const float den = 1.f + esum;  // exp(0) of the zeroed positive slot (Appendix C.1)
    const float num = eta * (row[y] - margin) - mx;
    const float wgt = class_weights != nullptr ? class_weights[y] : 1.f;
    const float l = wgt * (logf(den) - num);
    const bool active = !hinge || l >= 0.f;
    if (lane == 0) loss_acc += active ? l : 0.f;
    float dpart = 0.f;
    for (int k = lane; k < K; k += 64) {
      float g = 0.f;  // dl/ds_k
      if (active) {
        g = (k == y) ? -1.f : expf(eta * (row[k] - margin) - mx) / den;
        if (k == am) g += 1.f / den;  // gradient through the row-max subtraction
        g *= wgt;
      }
      dsim[(size_t)b * K + k] = g * eta * invB;
      dpart += g * (row[k] - margin);
    }
    dpart = wave_sum(dpart);
    if (lane == 0) deta_acc += dpart;
  }
  if (lane == 0) {
    red[0][wave] = loss_acc;
    red[1][wave] = deta_acc;
  }
  __syncthreads();
  if (tid == 0) {
    loss[0] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * invB;
    deta[0] = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * invB;
  }
}

// soft-target CE (icarl.py:123-125) / plain CE when soft == nullptr
__global__ __launch_bounds__(256) void softce_kernel(const float* __restrict__ score, const float* __restrict__ soft,
                                                      const int64_t* __restrict__ labels, float* __restrict__ loss,
                                                      float* __restrict__ dscore, int B, int K) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float invB = 1.f / (float)B;
  float loss_acc = 0.f;
  for (int b = wave; b < B; b += 4) {
    const float* row = score + (size_t)b * K;
    const int y = soft ? -1 : (int)labels[b];
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, row[k]);
    mx = wave_max(mx);
    float es = 0.f, tsum = 0.f, ts = 0.f;
    for (int k = lane; k < K; k += 64) {
      es += expf(row[k] - mx);
      const float t = soft ? soft[(size_t)b * K + k] : (k == y ? 1.f : 0.f);
      tsum += t;
      ts += t * row[k];
    }
    es = wave_sum(es);
    tsum = wave_sum(tsum);
    ts = wave_sum(ts);
    const float lse = mx + logf(es);
    if (lane == 0) loss_acc += lse * tsum - ts;
    for (int k = lane; k < K; k += 64) {
      const float t = soft ? soft[(size_t)b * K + k] : (k == y ? 1.f : 0.f);
      dscore[(size_t)b * K + k] = (expf(row[k] - lse) * tsum - t) * invB;
    }
  }
  if (lane == 0) red[wave] = loss_acc;
  __syncthreads();
  if (tid == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) * invB;
}

// one wave per row: targets = onehot or softmax(prev logits) for old-class samples
__global__ __launch_bounds__(256) void icarl_targets_kernel(const int64_t* __restrict__ labels, const float* __restrict__ prev,
                                                             int prevK, const float* __restrict__ base, float* __restrict__ tgt,
                                                             int B, int K) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int y = (int)labels[b];
  if (prev != nullptr && y < prevK) {
    const float* row = prev + (size_t)b * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, row[k]);
    mx = wave_max(mx);
    float es = 0.f;
    for (int k = lane; k < K; k += 64) es += expf(row[k] - mx);
    es = wave_sum(es);
    for (int k = lane; k < K; k += 64) tgt[(size_t)b * K + k] = expf(row[k] - mx) / es;
  } else {
    for (int k = lane; k < K; k += 64) tgt[(size_t)b * K + k] = base != nullptr ? base[(size_t)b * K + k] : ((k == y) ? 1.f : 0.f);
  }
}

// ActorCutMix smooth labels (libs/losses/acm_smooth_ce.py:18-28): y = onehot(label) * lam + (1 - lam) * onehot(bg),
// lam = 1 - (1 - foreground_ratio)^alpha, background label -1 counted as class 0.  One wave per row.
__global__ __launch_bounds__(256) void acm_targets_kernel(const int64_t* __restrict__ labels, const int64_t* __restrict__ bg,
                                                           const float* __restrict__ fg_ratio, float alpha,
                                                           float* __restrict__ tgt, int B, int K) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int y = (int)labels[b];
  int yb = (int)bg[b];
  if (yb == -1) yb = 0;
  const float lam = 1.f - powf(1.f - fg_ratio[b], alpha);
  for (int k = lane; k < K; k += 64) tgt[(size_t)b * K + k] = (k == y ? 1.f : 0.f) * lam + (1.f - lam) * (k == yb ? 1.f : 0.f);
}

// average_clip: out[b,:] = mean_i softmax(s[b*n+i,:])  (or plain mean); one wave per b
__global__ __launch_bounds__(256) void softmax_mean_kernel(const float* __restrict__ s, float* __restrict__ out, int B, int n, int K,
                                                            int apply_softmax) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  for (int k = lane; k < K; k += 64) out[(size_t)b * K + k] = 0.f;
  for (int i = 0; i < n; ++i) {
    const float* row = s + ((size_t)b * n + i) * K;
    float mx = 0.f, es = 1.f;
    if (apply_softmax) {
      mx = -INFINITY;
      for (int k = lane; k < K; k += 64) mx = fmaxf(mx, row[k]);
      mx = wave_max(mx);
      es = 0.f;
      for (int k = lane; k < K; k += 64) es += expf(row[k] - mx);
      es = wave_sum(es);
    }
    for (int k = lane; k < K; k += 64) {
      const float v = apply_softmax ? expf(row[k] - mx) / es : row[k];
      out[(size_t)b * K + k] += v;
    }
  }
  const float inv = 1.f / (float)n;
  for (int k = lane; k < K; k += 64) out[(size_t)b * K + k] *= inv;
}

__global__ __launch_bounds__(256) void topk_acc_kernel(const float* __restrict__ score, const int64_t* __restrict__ labels,
                                                        float* __restrict__ acc, int B, int K) {
  __shared__ int red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int h1 = 0, h5 = 0;
  for (int b = wave; b < B; b += 4) {
    const float* row = score + (size_t)b * K;
    const float ys = row[(int)labels[b]];
    int greater = 0;
    for (int k = lane; k < K; k += 64) greater += row[k] > ys ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) greater += __shfl_xor(greater, o, 64);
    h1 += greater < 1;
    h5 += greater < 5;
  }
  if (lane == 0) {
    red[0][wave] = h1;
    red[1][wave] = h5;
  }
  __syncthreads();
  if (tid == 0) {
    acc[0] = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)B;
    acc[1] = (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)B;
  }
}

// ---- feature-distillation MSE ---------------------------------------------------------------
constexpr int RED_BLOCKS = 1024;

template <int ES = 4>
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const void* __restrict__ a, const void* __restrict__ b,
                                                              float* __restrict__ partial, int64_t n4) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = act_ld4<ES>(a, i), y = act_ld4<ES>(b, i);
    const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
    s += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void mse_finalize_kernel(const float* __restrict__ partial, int nb, double inv_numel, float* __restrict__ out) {
  const int lane = threadIdx.x;  // one wave
  double s = 0.0;
  for (int i = lane; i < nb; i += 64) s += (double)partial[i];
  s = wave_sum_d(s);
  if (lane == 0) out[0] = (float)(s * inv_numel);
}

template <int ES = 4>
__global__ __launch_bounds__(256) void kd_mse_bwd_kernel(const void* __restrict__ a, const void* __restrict__ b,
                                                          const float* __restrict__ gdev, float ghost, void* __restrict__ da,
                                                          int64_t n4) {
  const float c = ghost * (gdev != nullptr ? gdev[0] : 1.f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = act_ld4<ES>(a, i), y = act_ld4<ES>(b, i);
    act_st4<ES>(da, i, make_float4(c * (x.x - y.x), c * (x.y - y.y), c * (x.z - y.z), c * (x.w - y.w)));
  }
}

}  // namespace

#define HL_STREAM ((hipStream_t)stream)

extern "C" int bdv_lsc_fwd(const float* x, const float* w, float* sim, float* xnorm, float* wnorm, float* cosbuf, int N, int D,
                           int K, int P, void* stream) {
  BDV_REQUIRE(x && w && sim && xnorm && wnorm && cosbuf && N > 0 && D > 0 && K > 0 && P > 0, "bdv_lsc_fwd: bad argument");
  const size_t lds = (size_t)HEAD_RB * (D + K * P) * sizeof(float);
  BDV_REQUIRE(lds <= 60000, "bdv_lsc_fwd: D + K*P too large for LDS");
  hipLaunchKernelGGL((head_fwd_kernel<1>), dim3((N + HEAD_RB - 1) / HEAD_RB), dim3(256), lds, HL_STREAM, x, w, (const float*)nullptr, sim, xnorm, wnorm,
                     cosbuf, N, D, K, P);
  BDV_LAUNCH_CHECK("bdv_lsc_fwd");
  return BDV_OK;
}

extern "C" int bdv_lsc_bwd(const float* dsim, const float* x, const float* w, const float* xnorm, const float* wnorm,
                           const float* cosbuf, float* dx, float* dw, float beta_w, float* dcos_ws, int N, int D, int K, int P,
                           void* stream) {
  BDV_REQUIRE(dsim && x && w && xnorm && wnorm && cosbuf && dcos_ws && N > 0 && D > 0 && K > 0 && P > 0,
              "bdv_lsc_bwd: bad argument");
  BDV_REQUIRE((size_t)K * P * 4 <= 60000 && (size_t)N * 4 <= 60000, "bdv_lsc_bwd: N or K*P too large for LDS");
  BDV_REQUIRE(dx != nullptr, "bdv_lsc_bwd: dx is required (dcos is produced by the dx pass)");
  hipLaunchKernelGGL(lsc_bwd_dx_kernel, dim3(N), dim3(256), (size_t)K * P * 4, HL_STREAM, dsim, x, w, xnorm, wnorm, cosbuf, dx,
                     dcos_ws, N, D, K, P);
  BDV_LAUNCH_CHECK("bdv_lsc_bwd(dx)");
  if (dw != nullptr) {
    hipLaunchKernelGGL(lsc_bwd_dw_kernel, dim3(K * P), dim3(256), (size_t)N * 4, HL_STREAM, (const float*)dcos_ws, x, w, xnorm, wnorm,
                       cosbuf, dw, beta_w, N, D, K * P);
    BDV_LAUNCH_CHECK("bdv_lsc_bwd(dw)");
  }
  return BDV_OK;
}

extern "C" int bdv_linear_fwd(const float* x, const float* w, const float* b, float* out, int N, int D, int K, void* stream) {
  BDV_REQUIRE(x && w && out && N > 0 && D > 0 && K > 0, "bdv_linear_fwd: bad argument");
  const size_t lds = (size_t)HEAD_RB * (D + K) * sizeof(float);
  BDV_REQUIRE(lds <= 60000, "bdv_linear_fwd: D + K too large for LDS");
  hipLaunchKernelGGL((head_fwd_kernel<0>), dim3((N + HEAD_RB - 1) / HEAD_RB), dim3(256), lds, HL_STREAM, x, w, b, out, (float*)nullptr, (float*)nullptr,
                     (float*)nullptr, N, D, K, 1);
  BDV_LAUNCH_CHECK("bdv_linear_fwd");
  return BDV_OK;
}

extern "C" int bdv_linear_bwd(const float* dout, const float* x, const float* w, float* dx, float* dw, float* db, float beta_w,
                              int N, int D, int K, void* stream) {
  BDV_REQUIRE(dout && x && w && N > 0 && D > 0 && K > 0, "bdv_linear_bwd: bad argument");
  BDV_REQUIRE((size_t)K * HEAD_RB * 4 <= 60000 && (size_t)N * 8 * 4 <= 60000, "bdv_linear_bwd: N or K too large for LDS");
  if (dx != nullptr) {
    hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3((N + HEAD_RB - 1) / HEAD_RB), dim3(256), (size_t)K * HEAD_RB * 4, HL_STREAM, dout, w,
                       dx, N, D, K);
    BDV_LAUNCH_CHECK("bdv_linear_bwd(dx)");
  }
  if (dw != nullptr) {
    hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3((K + 7) / 8, (D + 255) / 256), dim3(256), (size_t)N * 8 * 4, HL_STREAM, dout, x, dw,
                       db, beta_w, N, D, K);
    BDV_LAUNCH_CHECK("bdv_linear_bwd(dw)");
  }
  return BDV_OK;
}

extern "C" int bdv_consensus_fwd(const float* s, float* out, int B, int T, int K, void* stream) {
  BDV_REQUIRE(s && out && B > 0 && T > 0 && K > 0, "bdv_consensus_fwd: bad argument");
  hipLaunchKernelGGL(consensus_fwd_kernel, dim3((B * K + 255) / 256), dim3(256), 0, HL_STREAM, s, out, B, T, K);
  BDV_LAUNCH_CHECK("bdv_consensus_fwd");
  return BDV_OK;
}

extern "C" int bdv_consensus_bwd(const float* dout, float* ds, int B, int T, int K, void* stream) {
  BDV_REQUIRE(dout && ds && B > 0 && T > 0 && K > 0, "bdv_consensus_bwd: bad argument");
  hipLaunchKernelGGL(consensus_bwd_kernel, dim3((B * T * K + 255) / 256), dim3(256), 0, HL_STREAM, dout, ds, B, T, K);
  BDV_LAUNCH_CHECK("bdv_consensus_bwd");
  return BDV_OK;
}

extern "C" int bdv_dropout(const float* x, float* out, int64_t numel, float p, uint64_t seed, void* stream) {
  BDV_REQUIRE(x && out && numel > 0 && p >= 0.f && p < 1.f, "bdv_dropout: bad argument");
  int64_t blocks = (numel + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_kernel, dim3((int)blocks), dim3(256), 0, HL_STREAM, x, out, numel, p, 1.f / (1.f - p), seed);
  BDV_LAUNCH_CHECK("bdv_dropout");
  return BDV_OK;
}

extern "C" int bdv_lsc_loss(const float* sim, const int64_t* targets, const float* eta, float margin, int hinge,
                            const float* class_weights, float* loss,
                            float* dsim, float* deta, int B, int K, void* stream) {
  BDV_REQUIRE(sim && targets && eta && loss && dsim && deta && B > 0 && K > 0, "bdv_lsc_loss: bad argument");
  hipLaunchKernelGGL(lsc_loss_kernel, dim3(1), dim3(256), 0, HL_STREAM, sim, targets, eta, margin, hinge, class_weights, loss, dsim, deta, B, K);
  BDV_LAUNCH_CHECK("bdv_lsc_loss");
  return BDV_OK;
}

extern "C" int bdv_softce_loss(const float* score, const float* soft_targets, const int64_t* labels, float* loss, float* dscore,
                               int B, int K, void* stream) {
  BDV_REQUIRE(score && loss && dscore && B > 0 && K > 0, "bdv_softce_loss: bad argument");
  BDV_REQUIRE((soft_targets != nullptr) != (labels != nullptr), "bdv_softce_loss: give exactly one of soft_targets / labels");
  hipLaunchKernelGGL(softce_kernel, dim3(1), dim3(256), 0, HL_STREAM, score, soft_targets, labels, loss, dscore, B, K);
  BDV_LAUNCH_CHECK("bdv_softce_loss");
  return BDV_OK;
}

extern "C" int bdv_icarl_targets(const int64_t* labels, const float* prev_logits, int prev_K, const float* base_targets,
                                 float* targets, int B, int K, void* stream) {
  BDV_REQUIRE(labels && targets && B > 0 && K > 0, "bdv_icarl_targets: bad argument");
  BDV_REQUIRE(base_targets != targets, "bdv_icarl_targets: base_targets must not alias targets");
  hipLaunchKernelGGL(icarl_targets_kernel, dim3((B + 3) / 4), dim3(256), 0, HL_STREAM, labels, prev_logits, prev_K, base_targets,
                     targets, B, K);
  BDV_LAUNCH_CHECK("bdv_icarl_targets");
  return BDV_OK;
}

extern "C" int bdv_acm_targets(const int64_t* labels, const int64_t* background_labels, const float* foreground_ratio, float alpha,
                               float* targets, int B, int K, void* stream) {
  BDV_REQUIRE(labels && background_labels && foreground_ratio && targets && B > 0 && K > 0, "bdv_acm_targets: bad argument");
  hipLaunchKernelGGL(acm_targets_kernel, dim3((B + 3) / 4), dim3(256), 0, HL_STREAM, labels, background_labels, foreground_ratio,
                     alpha, targets, B, K);
  BDV_LAUNCH_CHECK("bdv_acm_targets");
  return BDV_OK;
}

extern "C" int bdv_softmax_mean(const float* s, float* out, int B, int n, int K, int apply_softmax, void* stream) {
  BDV_REQUIRE(s && out && B > 0 && n > 0 && K > 0, "bdv_softmax_mean: bad argument");
  hipLaunchKernelGGL(softmax_mean_kernel, dim3((B + 3) / 4), dim3(256), 0, HL_STREAM, s, out, B, n, K, apply_softmax);
  BDV_LAUNCH_CHECK("bdv_softmax_mean");
  return BDV_OK;
}

extern "C" int bdv_topk_acc(const float* score, const int64_t* labels, float* acc, int B, int K, void* stream) {
  BDV_REQUIRE(score && labels && acc && B > 0 && K > 0, "bdv_topk_acc: bad argument");
  hipLaunchKernelGGL(topk_acc_kernel, dim3(1), dim3(256), 0, HL_STREAM, score, labels, acc, B, K);
  BDV_LAUNCH_CHECK("bdv_topk_acc");
  return BDV_OK;
}

extern "C" size_t bdv_reduce_workspace_bytes(void) { return RED_BLOCKS * sizeof(float); }

extern "C" int bdv_kd_mse_fwd(const void* cur, const void* prev, float* mse, int64_t numel, void* workspace,
                              size_t workspace_bytes, int act_dtype, void* stream) {
  BDV_REQUIRE_ACT(act_dtype, "bdv_kd_mse_fwd");
  BDV_REQUIRE(cur && prev && mse && workspace && numel > 0 && numel % 4 == 0, "bdv_kd_mse_fwd: bad argument");
  BDV_REQUIRE(bdv_aligned16(cur) && bdv_aligned16(prev), "bdv_kd_mse_fwd: alignment");
  if (workspace_bytes < bdv_reduce_workspace_bytes()) {
    bdv_set_error("bdv_kd_mse_fwd: workspace too small");
    return BDV_EWORKSPACE;
  }
  const int64_t n4 = numel / 4;
  int nb = (int)((n4 + 255) / 256);
  if (nb > RED_BLOCKS) nb = RED_BLOCKS;
  BDV_ACT_SWITCH(act_dtype, ES, hipLaunchKernelGGL((sqdiff_partial_kernel<ES>), dim3(nb), dim3(256), 0, HL_STREAM, cur, prev,
                     (float*)workspace, n4));
  BDV_LAUNCH_CHECK("bdv_kd_mse_fwd(partial)");
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(64), 0, HL_STREAM, (const float*)workspace, nb, 1.0 / (double)numel, mse);
  BDV_LAUNCH_CHECK("bdv_kd_mse_fwd(finalize)");
  return BDV_OK;
}

extern "C" int bdv_kd_mse_bwd(const void* cur, const void* prev, const float* gscale_dev, float gscale_host, void* dcur,
                              int64_t numel, int act_dtype, void* stream) {
  BDV_REQUIRE_ACT(act_dtype, "bdv_kd_mse_bwd");
  BDV_REQUIRE(cur && prev && dcur && numel > 0 && numel % 4 == 0, "bdv_kd_mse_bwd: bad argument");
  BDV_REQUIRE(bdv_aligned16(cur) && bdv_aligned16(prev) && bdv_aligned16(dcur), "bdv_kd_mse_bwd: alignment");
  const int64_t n4 = numel / 4;
  int64_t nb = (n4 + 255) / 256;
  if (nb > (1 << 20)) nb = 1 << 20;
  BDV_ACT_SWITCH(act_dtype, ES, hipLaunchKernelGGL((kd_mse_bwd_kernel<ES>), dim3((int)nb), dim3(256), 0, HL_STREAM, cur, prev, gscale_dev,
                     gscale_host * 2.f / (float)numel, dcur, n4));
  BDV_LAUNCH_CHECK("bdv_kd_mse_bwd");
  return BDV_OK;
}
