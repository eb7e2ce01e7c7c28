// Multi-tensor global-norm clip + SGD(momentum, weight decay).  One launch per operation over
// all parameter tensors (device-side pointer tables), deterministic fixed-order norm reduction.
//
// Replaces torch.optim.SGD.step (built at libs/cil/cil.py:467 with the groups of
// libs/models/cil_heads/tsm.py:273-303) and PL's clip_grad_norm_ (libs/cil/cil.py:743).
#include "common.h"

namespace {

constexpr int CHUNKS = 32;  // blocks per tensor

__global__ __launch_bounds__(256) void multi_sqnorm_partial_kernel(const float* const* __restrict__ grads,
                                                                    const int64_t* __restrict__ numels,
                                                                    float* __restrict__ partial) {
  __shared__ float red[4];
  const int t = blockIdx.x, cy = blockIdx.y, tid = threadIdx.x;
  const float* g = grads[t];
  const int64_t n = numels[t];
  float s = 0.f;
  for (int64_t i = (int64_t)cy * 256 + tid; i < n; i += (int64_t)CHUNKS * 256) {
    const float v = g[i];
    s += v * v;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) partial[(int64_t)t * CHUNKS + cy] = red[0] + red[1] + red[2] + red[3];
}

__global__ void sqnorm_finalize_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  const int lane = threadIdx.x;  // one wave
  double s = 0.0;
  for (int i = lane; i < n; i += 64) s += (double)partial[i];
  s = wave_sum_d(s);
  if (lane == 0) out[0] = (float)s;
}

__global__ void clip_coef_kernel(const float* __restrict__ sqnorm, float grad_scale, float max_norm, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float c = 1.f;
  if (max_norm > 0.f) {
    const float total = sqrtf(sqnorm[0]) * grad_scale;
    c = fminf(max_norm / (total + 1e-6f), 1.f);
  }
  coef[0] = c;
}

__global__ __launch_bounds__(256) void multi_sgd_kernel(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                         float* const* __restrict__ bufs, const int64_t* __restrict__ numels,
                                                         const float* __restrict__ lrs, const float* __restrict__ wds,
                                                         float momentum, float grad_scale, const float* __restrict__ clip_coef) {
  const int t = blockIdx.x, cy = blockIdx.y, tid = threadIdx.x;
  float* p = params[t];
  const float* g = grads[t];
  float* b = bufs[t];
  const int64_t n = numels[t];
  const float lr = lrs[t], wd = wds[t];
  const float gs = grad_scale * (clip_coef != nullptr ? clip_coef[0] : 1.f);
  for (int64_t i = (int64_t)cy * 256 + tid; i < n; i += (int64_t)CHUNKS * 256) {
    const float pv = p[i];
    const float gv = g[i] * gs + wd * pv;
    const float bv = momentum * b[i] + gv;
    b[i] = bv;
    p[i] = pv - lr * bv;
  }
}

}  // namespace

extern "C" int bdv_multi_sqnorm(const float* const* grads, const int64_t* numels, int ntensors, float* out_sqnorm, void* workspace,
                                size_t workspace_bytes, void* stream) {
  BDV_REQUIRE(grads && numels && out_sqnorm && workspace && ntensors > 0, "bdv_multi_sqnorm: bad argument");
  if (workspace_bytes < (size_t)ntensors * CHUNKS * sizeof(float)) {
    bdv_set_error("bdv_multi_sqnorm: workspace too small (need %zu bytes)", (size_t)ntensors * CHUNKS * sizeof(float));
    return BDV_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(multi_sqnorm_partial_kernel, dim3(ntensors, CHUNKS), dim3(256), 0, s, grads, numels, (float*)workspace);
  BDV_LAUNCH_CHECK("bdv_multi_sqnorm(partial)");
  hipLaunchKernelGGL(sqnorm_finalize_kernel, dim3(1), dim3(64), 0, s, (const float*)workspace, ntensors * CHUNKS, out_sqnorm);
  BDV_LAUNCH_CHECK("bdv_multi_sqnorm(finalize)");
  return BDV_OK;
}

extern "C" int bdv_clip_coef(const float* sqnorm, float grad_scale, float max_norm, float* clip_coef, void* stream) {
  BDV_REQUIRE(sqnorm && clip_coef, "bdv_clip_coef: null pointer");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sqnorm, grad_scale, max_norm, clip_coef);
  BDV_LAUNCH_CHECK("bdv_clip_coef");
  return BDV_OK;
}

extern "C" int bdv_multi_sgd(float* const* params, const float* const* grads, float* const* bufs, const int64_t* numels,
                             const float* lrs, const float* wds, int ntensors, float momentum, float grad_scale,
                             const float* clip_coef, void* stream) {
  BDV_REQUIRE(params && grads && bufs && numels && lrs && wds && ntensors > 0, "bdv_multi_sgd: bad argument");
  hipLaunchKernelGGL(multi_sgd_kernel, dim3(ntensors, CHUNKS), dim3(256), 0, (hipStream_t)stream, params, grads, bufs, numels, lrs,
                     wds, momentum, grad_scale, clip_coef);
  BDV_LAUNCH_CHECK("bdv_multi_sgd");
  return BDV_OK;
}
