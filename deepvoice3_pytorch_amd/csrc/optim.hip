// Optimiser tail of the reference train step (train.py:755-759): global-norm gradient clip
// (torch.nn.utils.clip_grad_norm_) + Adam, over ONE flat fp32 parameter arena (all trainable
// parameters are views into it; so are the gradients, which is also what the RCCL all-reduce
// buckets slice).  HBM-bound: 4 reads + 3 writes of 4 B per parameter.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n,
                                                             float* __restrict__ partial) {
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  const int64_t n4 = n & ~(int64_t)3;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n4; i += stride) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n4)) {
    const float v = g[n4 + threadIdx.x];
    s += v * v;
  }
  __shared__ float red[4];
  s = dv3_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = sqrt(sum partial)   (out[1] = sum, for cross-rank reduction of squared norms)
__global__ __launch_bounds__(256) void sqnorm_finish_kernel(const float* __restrict__ partial,
                                                            int n_partial, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
  __shared__ float red[4];
  s = dv3_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3];
    out[0] = sqrtf(t);
    out[1] = t;
  }
}

// hyper = {lr, 1 - beta1^t, sqrt(1 - beta2^t)} on the device (graph-replay friendly).
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p,
                                                        const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        int64_t n, const float* __restrict__ grad_norm,
                                                        float clip, const float* __restrict__ hyper,
                                                        float beta1, float beta2, float eps,
                                                        float weight_decay, float grad_prescale) {
  float coef = grad_prescale;
  if (clip > 0.f && grad_norm) {
    // clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    // grad_norm is the norm of the (summed) arena; the averaged gradient's norm is prescale * that
    const float c = clip / (grad_norm[0] * grad_prescale + 1e-6f);
    coef *= fminf(c, 1.0f);
  }
  const float lr = hyper[0], bc1 = hyper[1], bc2s = hyper[2];
  const float step_size = lr / bc1;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float gi = g[i] * coef;
    const float pi = p[i];
    if (weight_decay != 0.f) gi += weight_decay * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

}  // namespace

extern "C" int dv3_grad_sqnorm_f32(const float* g, int64_t n, float* partial, int32_t n_partial,
                                   float* out2, void* stream) {
  DV3_REQUIRE(g && partial && out2 && n > 0 && n_partial > 0, "grad_sqnorm: bad args");
  DV3_REQUIRE(((uintptr_t)g & 15) == 0, "grad_sqnorm: gradient arena must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int nb = (int)dv3_cdiv64(n, 256 * 4 * 8);
  if (nb < 1) nb = 1;
  if (nb > n_partial) nb = n_partial;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, partial);
  hipLaunchKernelGGL(sqnorm_finish_kernel, dim3(1), dim3(256), 0, st, partial, nb, out2);
  return dv3_check_launch("grad_sqnorm_f32");
}

extern "C" int dv3_clip_adam_f32(float* p, const float* g, float* m, float* v, int64_t n,
                                 const float* grad_norm, float clip, const float* hyper, float beta1,
                                 float beta2, float eps, float weight_decay, float grad_prescale,
                                 void* stream) {
  DV3_REQUIRE(p && g && m && v && hyper && n > 0, "clip_adam: bad args");
  int64_t nb = dv3_cdiv64(n, 256 * 4);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m,
                     v, n, grad_norm, clip, hyper, beta1, beta2, eps, weight_decay, grad_prescale);
  return dv3_check_launch("clip_adam_f32");
}
