// Representation path after the backbone: clip representations from the pooled
// features, nearest-mean-of-exemplars (NME) cosine classifier, per-class means and iCaRL herding selection.
//
// Replaces (file:line under the reference tree):
//   libs/cil/cil.py:501-506,:564-571   _extract_repr + predict_step: consensus mean over segments, F.normalize, crop mean
//   libs/cil/cil.py:945-960            NME: cosine similarity to the class means, mean over crops, argmax
//   libs/cil/cil.py:1079-1083          class means of the exemplar representations
//   libs/cil/memory_selection.py:70-92,:150-164   Herding.construct_exemplar inner loop + calc_mean_features
// Tensors are tiny next to the conv stack; kernels favour determinism (fixed-order reductions, first-index ties).
#include "common.h"

namespace {

constexpr float NORMALIZE_EPS = 1e-12f;  // F.normalize default eps
constexpr float COS_EPS = 1e-8f;         // F.cosine_similarity default eps
constexpr float PDIST_EPS = 1e-6f;       // torch.pairwise_distance default eps (added to the difference)

__device__ __forceinline__ float block_sum(float v, float* red) {
  // 256 threads = 4 waves; every thread gets the total (fixed order)
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// grid = B.  feat (B*crops*T, D) -> repr (B*crops, D) = normalize(mean_t feat), mean_crops (B, D) = mean_c repr
__global__ __launch_bounds__(256) void repr_kernel(const float* __restrict__ feat, float* __restrict__ repr,
                                                    float* __restrict__ mean_crops, int crops, int T, int D) {
  extern __shared__ float sm[];  // D: consensus of the current crop; D: running crop sum
  float* cur = sm;
  float* acc = sm + D;
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int d = tid; d < D; d += 256) acc[d] = 0.f;
  for (int c = 0; c < crops; ++c) {
    const float* src = feat + (size_t)(b * crops + c) * T * D;
    float ss = 0.f;
    for (int d = tid; d < D; d += 256) {
      float s = 0.f;
      for (int t = 0; t < T; ++t) s += src[(size_t)t * D + d];
      s /= (float)T;
      cur[d] = s;
      ss += s * s;
    }
    const float nrm = fmaxf(sqrtf(block_sum(ss, red)), NORMALIZE_EPS);
    for (int d = tid; d < D; d += 256) {
      const float v = cur[d] / nrm;
      repr[(size_t)(b * crops + c) * D + d] = v;
      acc[d] += v;
    }
  }
  for (int d = tid; d < D; d += 256) mean_crops[(size_t)b * D + d] = acc[d] / (float)crops;
}

// grid = rows.  out = x / max(||x||, eps)
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, int D, float eps) {
  __shared__ float red[4];
  const float* r = x + (size_t)blockIdx.x * D;
  float ss = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) ss += r[d] * r[d];
  const float nrm = fmaxf(sqrtf(block_sum(ss, red)), eps);
  for (int d = threadIdx.x; d < D; d += 256) out[(size_t)blockIdx.x * D + d] = r[d] / nrm;
}

// grid = S.  sim[s,k] = mean_c <x_{s,c}/max(|x|,eps), wn_k>, pred[s] = first argmax_k
__global__ __launch_bounds__(256) void nme_kernel(const float* __restrict__ repr, const float* __restrict__ wn,
                                                   float* __restrict__ sim, int64_t* __restrict__ pred, int crops, int D, int K) {
  extern __shared__ float sm[];  // D normalised row, K accumulators
  float* xs = sm;
  float* acc = sm + D;
  __shared__ float red[4];
  __shared__ float bestv[4];
  __shared__ int besti[4];
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < K; k += 256) acc[k] = 0.f;
  for (int c = 0; c < crops; ++c) {
    const float* x = repr + (size_t)(s * crops + c) * D;
    float ss = 0.f;
    for (int d = tid; d < D; d += 256) ss += x[d] * x[d];
    const float nx = fmaxf(sqrtf(block_sum(ss, red)), COS_EPS);
    for (int d = tid; d < D; d += 256) xs[d] = x[d] / nx;
    __syncthreads();
    for (int k = wave; k < K; k += 4) {
      const float* w = wn + (size_t)k * D;
      float dot = 0.f;
      for (int d = lane; d < D; d += 64) dot += xs[d] * w[d];
      dot = wave_sum(dot);
      if (lane == 0) acc[k] += dot;  // k is owned by one wave: fixed crop order
    }
    __syncthreads();
  }
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int k = tid; k < K; k += 256) {
    const float v = acc[k] / (float)crops;
    sim[(size_t)s * K + k] = v;
    if (v > bv) {
      bv = v;
      bi = k;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if (lane == 0) {
    bestv[wave] = bv;
    besti[wave] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) {
        bv = bestv[w];
        bi = besti[w];
      }
    pred[s] = bi == 0x7fffffff ? 0 : bi;  // all-NaN row: torch.argmax returns the NaN position; 0 is the documented deviation
  }
}

// grid = (K, ceil(D/256)).  out[k,d] = mean over rows with label k (row order); empty class -> NaN like torch.mean
__global__ __launch_bounds__(256) void class_means_kernel(const float* __restrict__ repr, const int64_t* __restrict__ labels,
                                                           float* __restrict__ out, int n, int D) {
  const int k = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
  if (d >= D) return;
  float s = 0.f;
  int cnt = 0;
  for (int i = 0; i < n; ++i)
    if (labels[i] == k) {
      s += repr[(size_t)i * D + d];
      ++cnt;
    }
  out[(size_t)k * D + d] = s / (float)cnt;
}

// One block = the whole greedy herding loop of one class.
// nf: (n, D) scratch (normalised features), alive: n flags, LDS: class mean, its cosine-normalised copy, moving mean.
__global__ __launch_bounds__(256) void herding_kernel(const float* __restrict__ feat, float* __restrict__ nf, int* __restrict__ alive,
                                                       float* __restrict__ class_mean, int64_t* __restrict__ out_idx,
                                                       float* __restrict__ out_dist, int n, int D, int m, int cosine) {
  extern __shared__ float sm[];
  float* cm = sm;           // class mean as the reference returns it
  float* cmn = sm + D;      // cosine: cm / max(|cm|, 1e-8)
  float* mov = sm + 2 * D;  // moving exemplar mean
  __shared__ float red[4];
  __shared__ float bestv[4];
  __shared__ int besti[4];
  __shared__ int chosen_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // normalised features (memory_selection.py:156-159)
  for (int i = wave; i < n; i += 4) {
    const float* x = feat + (size_t)i * D;
    float nrm = 1.f;
    if (cosine) {
      float ss = 0.f;
      for (int d = lane; d < D; d += 64) ss += x[d] * x[d];
      nrm = fmaxf(sqrtf(wave_sum(ss)), NORMALIZE_EPS);
    }
    for (int d = lane; d < D; d += 64) nf[(size_t)i * D + d] = cosine ? x[d] / nrm : x[d];
    if (lane == 0) alive[i] = 1;
  }
  // class mean over the raw features (:161), normalised when cosine (:162-163)
  float ss = 0.f;
  for (int d = tid; d < D; d += 256) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += feat[(size_t)i * D + d];
    s /= (float)n;
    cm[d] = s;
    mov[d] = 0.f;
    ss += s * s;
  }
  const float tot = block_sum(ss, red);
  if (cosine) {
    const float nrm = fmaxf(sqrtf(tot), NORMALIZE_EPS);
    float ss2 = 0.f;
    for (int d = tid; d < D; d += 256) {
      const float v = cm[d] / nrm;
      cm[d] = v;
      ss2 += v * v;
    }
    const float n2 = fmaxf(sqrtf(block_sum(ss2, red)), COS_EPS);
    for (int d = tid; d < D; d += 256) cmn[d] = cm[d] / n2;
  }
  for (int d = tid; d < D; d += 256) class_mean[d] = cm[d];
  __syncthreads();

  for (int it = 1; it <= m; ++it) {
    const float fa = (float)(it - 1), fb = (float)it;
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int i = wave; i < n; i += 4) {
      if (!alive[i]) continue;
      const float* r = nf + (size_t)i * D;
      float dist;
      if (cosine) {  // 1 - sum((tmp / max(|tmp|, eps)) * cmn)   (:77)
        float s2 = 0.f;
        for (int d = lane; d < D; d += 64) {
          const float t = mov[d] * fa / fb + r[d] / fb;
          s2 += t * t;
        }
        const float nt = fmaxf(sqrtf(wave_sum(s2)), COS_EPS);
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) {
          const float t = mov[d] * fa / fb + r[d] / fb;
          dot += (t / nt) * cmn[d];
        }
        dist = 1.f - wave_sum(dot);
      } else {  // || tmp - mean + 1e-6 ||_2   (:79)
        float s2 = 0.f;
        for (int d = lane; d < D; d += 64) {
          const float t = mov[d] * fa / fb + r[d] / fb;
          const float df = t - cm[d] + PDIST_EPS;
          s2 += df * df;
        }
        dist = sqrtf(wave_sum(s2));
      }
      if (dist < bv) {  // rows ascend within a wave: the first minimum is kept
        bv = dist;
        bi = i;
      }
    }
    if (lane == 0) {
      bestv[wave] = bv;
      besti[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      float v = bestv[0];
      int ix = besti[0];
      for (int w = 1; w < 4; ++w)
        if (bestv[w] < v || (bestv[w] == v && besti[w] < ix)) {
          v = bestv[w];
          ix = besti[w];
        }
      if (ix == 0x7fffffff) {  // every remaining distance is NaN: take the first remaining row (torch.argmin of NaNs)
        for (int i = 0; i < n; ++i)
          if (alive[i]) {
            ix = i;
            break;
          }
        v = NAN;
      }
      chosen_s = ix;
      out_idx[it - 1] = ix;
      out_dist[it - 1] = v;
      alive[ix] = 0;
    }
    __syncthreads();
    const float* r = nf + (size_t)chosen_s * D;
    for (int d = tid; d < D; d += 256) mov[d] = mov[d] * fa / fb + r[d] / fb;  // (:83)
    __syncthreads();
  }
}

}  // namespace

#define RP_STREAM ((hipStream_t)stream)

extern "C" int bdv_repr_from_features(const float* feat, float* repr, float* mean_crops, int B, int crops, int T, int D,
                                      void* stream) {
  BDV_REQUIRE(feat && repr && mean_crops && B > 0 && crops > 0 && T > 0 && D > 0, "bdv_repr_from_features: bad argument");
  BDV_REQUIRE(D <= 4096, "bdv_repr_from_features: D=%d exceeds 4096", D);
  hipLaunchKernelGGL(repr_kernel, dim3(B), dim3(256), 2 * D * sizeof(float), RP_STREAM, feat, repr, mean_crops, crops, T, D);
  BDV_LAUNCH_CHECK("bdv_repr_from_features");
  return BDV_OK;
}

extern "C" size_t bdv_nme_workspace_bytes(int K, int D) { return (size_t)(K > 0 ? K : 0) * (D > 0 ? D : 0) * sizeof(float); }

extern "C" int bdv_nme_classify(const float* repr, const float* class_means, float* similarity, int64_t* pred, int S,
                                int crops, int D, int K, void* workspace, size_t workspace_bytes, void* stream) {
  BDV_REQUIRE(repr && class_means && similarity && pred && workspace && S > 0 && crops > 0 && D > 0 && K > 0,
              "bdv_nme_classify: bad argument");
  BDV_REQUIRE((size_t)(D + K) * sizeof(float) <= 64 * 1024, "bdv_nme_classify: D + K = %d exceeds the 64 KB LDS budget", D + K);
  if (workspace_bytes < bdv_nme_workspace_bytes(K, D)) {
    bdv_set_error("bdv_nme_classify: workspace %zu < required %zu bytes", workspace_bytes, bdv_nme_workspace_bytes(K, D));
    return BDV_EWORKSPACE;
  }
  float* wn = (float*)workspace;
  hipLaunchKernelGGL(row_normalize_kernel, dim3(K), dim3(256), 0, RP_STREAM, class_means, wn, D, COS_EPS);
  BDV_LAUNCH_CHECK("bdv_nme_classify(normalize)");
  hipLaunchKernelGGL(nme_kernel, dim3(S), dim3(256), (size_t)(D + K) * sizeof(float), RP_STREAM, repr, (const float*)wn, similarity,
                     pred, crops, D, K);
  BDV_LAUNCH_CHECK("bdv_nme_classify");
  return BDV_OK;
}

extern "C" int bdv_class_means(const float* repr, const int64_t* labels, float* means, int n, int D, int K, void* stream) {
  BDV_REQUIRE(repr && labels && means && n > 0 && D > 0 && K > 0, "bdv_class_means: bad argument");
  hipLaunchKernelGGL(class_means_kernel, dim3(K, (D + 255) / 256), dim3(256), 0, RP_STREAM, repr, labels, means, n, D);
  BDV_LAUNCH_CHECK("bdv_class_means");
  return BDV_OK;
}

extern "C" size_t bdv_herding_workspace_bytes(int n, int D) {
  if (n <= 0 || D <= 0) return 0;
  return (size_t)n * D * sizeof(float) + (((size_t)n * sizeof(int) + 15) & ~(size_t)15);
}

extern "C" int bdv_herding_select(const float* features, int n, int D, int num_exemplars, int cosine_distance,
                                  float* class_mean, int64_t* indices, float* dist, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  BDV_REQUIRE(features && class_mean && indices && dist && workspace && n > 0 && D > 0, "bdv_herding_select: bad argument");
  BDV_REQUIRE(num_exemplars >= 0 && num_exemplars <= n,
              "bdv_herding_select: %d exemplars requested from %d samples (the reference's argmin fails on the empty remainder)",
              num_exemplars, n);
  BDV_REQUIRE((size_t)3 * D * sizeof(float) <= 96 * 1024, "bdv_herding_select: D=%d exceeds the LDS budget", D);
  if (workspace_bytes < bdv_herding_workspace_bytes(n, D)) {
    bdv_set_error("bdv_herding_select: workspace %zu < required %zu bytes", workspace_bytes, bdv_herding_workspace_bytes(n, D));
    return BDV_EWORKSPACE;
  }
  float* nf = (float*)workspace;
  int* alive = (int*)((char*)workspace + (size_t)n * D * sizeof(float));
  auto kern = herding_kernel;
  const size_t lds = (size_t)3 * D * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      bdv_set_error("bdv_herding_select: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, RP_STREAM, features, nf, alive, class_mean, indices, dist, n, D, num_exemplars,
                     cosine_distance);
  BDV_LAUNCH_CHECK("bdv_herding_select");
  return BDV_OK;
}
