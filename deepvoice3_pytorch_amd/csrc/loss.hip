// Fused loss value + gradient kernels for the reference train step
// (train.py:261-291 sequence_mask/MaskedL1Loss, :537-582 logit/masked_mean/spec_loss,
//  :585-601 guided_attention(s), :614,:714 BCELoss, :733-740 attention loss).
// All are HBM-bound single passes: read prediction + target once, write the gradient once,
// block partial sums -> deterministic second-stage reduce (no float atomics).
#include "common.h"

namespace {

constexpr int kLossBlock = 256;

__device__ __forceinline__ void block_reduce4(float v[4], float* out /* [4] per block */) {
  __shared__ float red[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = dv3_wave_sum(v[k]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[w][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) out[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] +
                                          red[2][threadIdx.x] + red[3][threadIdx.x];
}

// binary divergence of one element and its derivative (train.py:537-556):
//   L = logit(y_hat) = log(y_hat + eps) - log(1 - y_hat + eps),  z = -y L + log1p(exp(L)),  dz/dy_hat = (sigmoid(L) - y) L'
// With a = y_hat + eps, b = 1 - y_hat + eps:  exp(L) = a / b,  log1p(exp(L)) = log(a + b) - log(b),
// sigmoid(L) = a / (a + b): two logarithms and two reciprocals instead of four transcendental calls -- the loss kernels
// were VALU-bound on them (round 2: 145 us per launch whatever the access pattern).  a + b = 1 + 2 eps.
// FAST (the tiled kernel, default; dv3_debug_set(57, 0) = libm): the three logarithms by v_log_f32 (__logf: 1 ulp of
// log2 x, i.e. ~1e-7 absolute here) -- with logf the tiled kernel was still bound by its vector work (~100 instructions per
// element; 103 us for the 64 x 804 x 513 linear-spectrogram loss, round 6).  The gradient takes no logarithm.
template <bool FAST = false>
__device__ __forceinline__ void spec_bd(float yh, float y, float& z, float& dz) {
  const float eps = 1e-8f;
  const float a = yh + eps, b = 1.f - yh + eps;
  constexpr float LN2 = 0.69314718055994530942f;
  const float la = FAST ? __builtin_amdgcn_logf(a) * LN2 : logf(a), lb = FAST ? __builtin_amdgcn_logf(b) * LN2 : logf(b);
  const float ab = a + b;
  z = -y * (la - lb) + ((FAST ? __builtin_amdgcn_logf(ab) * LN2 : logf(ab)) - lb);
  const float ra = __builtin_amdgcn_rcpf(a), rb = __builtin_amdgcn_rcpf(b);
  dz = (a * __builtin_amdgcn_rcpf(ab) - y) * (ra + rb);
}

// frames that take part (ABI 42, dv3_spec_loss_desc.t_valid): the batch's own maximum when the tensors are padded further
__device__ __forceinline__ int spec_t_valid(const dv3_spec_loss_desc& p) {
  if (!p.t_valid) return p.T;
  const int tv = p.t_valid[0];
  return tv < p.T ? (tv > p.r ? tv : p.r + 1) : p.T;
}

// mask_sum as train.py:286-290 computes it: sum over the expanded (B, T-r, D) mask
__device__ __forceinline__ float spec_mask_sum(const dv3_spec_loss_desc& p) {
  float ms = 0.f;
  const int Tr = spec_t_valid(p) - p.r;
  for (int b = 0; b < p.B; ++b) {
    int l = p.lengths[b] - p.r;
    l = l < 0 ? 0 : (l > Tr ? Tr : l);
    ms += (float)l;
  }
  return ms * (float)p.D;
}

__global__ __launch_bounds__(kLossBlock) void spec_loss_kernel(const dv3_spec_loss_desc p) {
  const int Tr = p.T - p.r, D = p.D;
  const int Trv = spec_t_valid(p) - p.r;       // == Tr unless the batch is padded beyond its own maximum
  const int64_t n = (int64_t)p.B * Tr * D;
  const bool use_mask = p.w_masked > 0.f && p.lengths;
  const float msum = use_mask ? spec_mask_sum(p) : 1.f;
  const float inv_n = 1.0f / (float)((int64_t)p.B * Trv * D);
  const float wm = use_mask ? p.w_masked : 0.f;
  const float c_all = (1.f - wm) * inv_n, c_msk = use_mask ? wm / msum : 0.f;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // l1, l1 masked, z, z masked
  const int64_t stride = (int64_t)gridDim.x * kLossBlock;
  // iterate with the faster-varying axis of y_hat innermost so its accesses coalesce
  const bool t_fast = p.yh_ts < p.yh_ds;
  for (int64_t i = (int64_t)blockIdx.x * kLossBlock + threadIdx.x; i < n; i += stride) {
    int dd, t, b;
    if (t_fast) {
      t = (int)(i % Tr);
      const int64_t bd = i / Tr;
      dd = (int)(bd % D);
      b = (int)(bd / D);
    } else {
      dd = (int)(i % D);
      const int64_t bt = i / D;
      t = (int)(bt % Tr);
      b = (int)(bt / Tr);
    }
    const int64_t ih = b * p.yh_bs + t * p.yh_ts + dd * p.yh_ds;          // y_hat[:, :-r]
    const int64_t iy = b * p.y_bs + (int64_t)(t + p.r) * p.y_ts + dd * p.y_ds;  // y[:, r:]
    if (t >= Trv) {
      if (p.dyh) p.dyh[ih] = 0.f;
      continue;
    }
    const float yh = p.y_hat[ih], y = p.y[iy];
    const float m = (use_mask && (t + p.r) < p.lengths[b]) ? 1.f : 0.f;
    const float diff = yh - y;
    const float ad = fabsf(diff);
    acc[0] += ad;
    acc[1] += m * ad;
    float dz = 0.f;
    if (p.w_bd > 0.f) {
      float z;
      spec_bd(yh, y, z, dz);
      acc[2] += z;
      acc[3] += m * z;
    }
    if (p.dyh) {
      const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
      const float coef = c_all + c_msk * m;
      p.dyh[ih] = p.gscale * coef * ((1.f - p.w_bd) * sgn + p.w_bd * dz);
    }
  }
  block_reduce4(acc, p.scratch + (int64_t)blockIdx.x * 4);
}

// The same loss when the prediction is time-fastest (the model's (B, T, D) outputs are transposed views of its BCT
// tensors) and the target is bin-fastest (collate_fn's (B, T, D) arrays): either thread order leaves one of the two
// tensors read with a D- or T-element stride (round 2: ~1 TB/s, 300 us for the linear-spectrogram loss).  A workgroup
// takes 64 frames x 64 bins at a time: the target tile is read bin-fastest into LDS, then every thread works
// frame-fastest -- prediction read, gradient write and the LDS reads (row stride 65) are all unit-stride.  Persistent
// grid (tiles are walked with a grid stride) so that the block partial sums fit the caller's scratch.
template <bool FAST>
__global__ __launch_bounds__(kLossBlock) void spec_loss_tiled_kernel(const dv3_spec_loss_desc p, int t_tiles, int d_tiles,
                                                                     int n_tiles) {
  __shared__ float ys[64 * 65];
  const int Tr = p.T - p.r, D = p.D;
  const int Trv = spec_t_valid(p) - p.r;
  const bool use_mask = p.w_masked > 0.f && p.lengths;
  const float msum = use_mask ? spec_mask_sum(p) : 1.f;
  const float inv_n = 1.0f / (float)((int64_t)p.B * Trv * D);
  const float wm = use_mask ? p.w_masked : 0.f;
  const float c_all = (1.f - wm) * inv_n, c_msk = use_mask ? wm / msum : 0.f;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int tid = threadIdx.x;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int dt = tile % d_tiles, tt_ = (tile / d_tiles) % t_tiles, b = tile / (d_tiles * t_tiles);
    const int t0 = tt_ * 64, d0 = dt * 64;
    __syncthreads();                                   // the previous tile's LDS reads are done
#pragma unroll 4
    for (int q = tid; q < 64 * 64; q += kLossBlock) {  // target tile, bin-fastest: y[b][t0 + tl + r][d0 + dl]
      const int tl = q >> 6, dl = q & 63;
      const int t = t0 + tl, dd = d0 + dl;
      ys[tl * 65 + dl] = (t < Tr && dd < D) ? p.y[b * p.y_bs + (int64_t)(t + p.r) * p.y_ts + dd] : 0.f;
    }
    __syncthreads();
    const int len_b = use_mask ? p.lengths[b] : 0;
#pragma unroll 4
    for (int q = tid; q < 64 * 64; q += kLossBlock) {  // frame-fastest
      const int dl = q >> 6, tl = q & 63;
      const int t = t0 + tl, dd = d0 + dl;
      if (t >= Tr || dd >= D) continue;
      const int64_t ih = b * p.yh_bs + t + (int64_t)dd * p.yh_ds;
      if (t >= Trv) {
        if (p.dyh) p.dyh[ih] = 0.f;
        continue;
      }
      const float yh = p.y_hat[ih], y = ys[tl * 65 + dl];
      const float m = (use_mask && (t + p.r) < len_b) ? 1.f : 0.f;
      const float diff = yh - y;
      const float ad = fabsf(diff);
      acc[0] += ad;
      acc[1] += m * ad;
      float dz = 0.f;
      if (p.w_bd > 0.f) {
        float z;
        spec_bd<FAST>(yh, y, z, dz);
        acc[2] += z;
        acc[3] += m * z;
      }
      if (p.dyh) {
        const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
        const float coef = c_all + c_msk * m;
        p.dyh[ih] = p.gscale * coef * ((1.f - p.w_bd) * sgn + p.w_bd * dz);
      }
    }
  }
  block_reduce4(acc, p.scratch + (int64_t)blockIdx.x * 4);
}

// the last r frames of y_hat take no part in the loss: their gradient is zero
__global__ __launch_bounds__(256) void spec_loss_tail_zero_kernel(const dv3_spec_loss_desc p) {
  const int64_t n = (int64_t)p.B * p.r * p.D;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int dd = (int)(i % p.D);
  const int64_t bt = i / p.D;
  const int t = p.T - p.r + (int)(bt % p.r);
  const int b = (int)(bt / p.r);
  p.dyh[b * p.yh_bs + t * p.yh_ts + dd * p.yh_ds] = 0.f;
}

__global__ __launch_bounds__(256) void spec_loss_finish_kernel(const dv3_spec_loss_desc p,
                                                               int n_blocks) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n_blocks; i += 256)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += p.scratch[(int64_t)i * 4 + k];
  __shared__ float fin[4];
  block_reduce4(acc, fin);
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool use_mask = p.w_masked > 0.f && p.lengths;
    const float msum = use_mask ? spec_mask_sum(p) : 1.f;
    const float n = (float)((int64_t)p.B * (spec_t_valid(p) - p.r) * p.D);
    const float wm = use_mask ? p.w_masked : 0.f;
    const float l1 = wm * (use_mask ? fin[1] / msum : 0.f) + (1.f - wm) * fin[0] / n;
    const float bd = p.w_bd > 0.f ? wm * (use_mask ? fin[3] / msum : 0.f) + (1.f - wm) * fin[2] / n : 0.f;
    p.out4[0] = l1;
    p.out4[1] = bd;
    p.out4[2] = (1.f - p.w_bd) * l1 + p.w_bd * bd;
    p.out4[3] = msum;
  }
}

// ---- guided attention ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void guided_attn_kernel(const float* __restrict__ attn,
                                                          const int32_t* __restrict__ in_len,
                                                          const int32_t* __restrict__ out_len,
                                                          float* __restrict__ dattn,
                                                          float* __restrict__ scratch, int L, int B,
                                                          int Tq, int Tk, float g, float gscale,
                                                          const int32_t* __restrict__ tq_valid,
                                                          const int32_t* __restrict__ tk_valid) {
  const int64_t per = (int64_t)B * Tq * Tk;
  // the mean's element count: the tensor's, or (ABI 42) that of the batch's own maxima
  const int64_t n = tq_valid ? (int64_t)L * B * min(tq_valid[0], Tq) * min(tk_valid[0], Tk) : per * L;
  const float inv_n = 1.0f / (float)n;
  const double inv2g2 = 1.0 / (2.0 * (double)g * (double)g);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += stride) {
    const int nk = (int)(i % Tk);
    const int64_t bt = i / Tk;
    const int t = (int)(bt % Tq), b = (int)(bt / Tq);
    const int N = in_len[b], T = out_len[b];
    float w = 0.f;
    if (nk < N && t < T) {
      // train.py:585-591 evaluates this in float64 and stores float32
      const double dlt = (double)nk / (double)N - (double)t / (double)T;
      w = (float)(1.0 - exp(-dlt * dlt * inv2g2));
    }
    for (int l = 0; l < L; ++l) {
      acc[0] += attn[(int64_t)l * per + i] * w;
      if (dattn) dattn[(int64_t)l * per + i] = gscale * w * inv_n;
    }
  }
  block_reduce4(acc, scratch + (int64_t)blockIdx.x * 4);
}

// out1 = (sum of the block partial sums) * scale; with va (and vb) the scale is 1 / (count * min(va[0], cap_a) [* min(vb[0], cap_b)])
__global__ __launch_bounds__(256) void sum_finish_kernel(const float* __restrict__ scratch,
                                                         int n_blocks, float scale,
                                                         float* __restrict__ out1,
                                                         const int32_t* __restrict__ va = nullptr, int cap_a = 0,
                                                         const int32_t* __restrict__ vb = nullptr, int cap_b = 0,
                                                         int64_t count = 0) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n_blocks; i += 256) acc[0] += scratch[(int64_t)i * 4];
  __shared__ float fin[4];
  block_reduce4(acc, fin);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (va) {
      int64_t n = count * min(va[0], cap_a);
      if (vb) n *= min(vb[0], cap_b);
      scale = 1.0f / (float)n;
    }
    out1[0] = fin[0] * scale;
  }
}

// ---- BCE (nn.BCELoss, mean; log clamped at -100 as torch does) ------------------------------
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ p,
                                                  const float* __restrict__ t,
                                                  float* __restrict__ dp, float* __restrict__ scratch,
                                                  int64_t n, float gscale, int T = 0,
                                                  const int32_t* __restrict__ t_valid = nullptr) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // ABI 42: [rows][T] with only the first t_valid[0] columns taking part
  const int Tv = t_valid ? min(t_valid[0], T) : T;
  const float inv_n = 1.0f / (float)(t_valid ? (n / T) * Tv : n);
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if (t_valid && (int)(i % T) >= Tv) {
      if (dp) dp[i] = 0.f;
      continue;
    }
    const float x = p[i], y = t[i];
    const float lx = fmaxf(logf(x), -100.f), l1x = fmaxf(logf(1.f - x), -100.f);
    acc[0] += -(y * lx + (1.f - y) * l1x);
    if (dp) dp[i] = gscale * inv_n * (x - y) / fmaxf((1.f - x) * x, 1e-12f);
  }
  block_reduce4(acc, scratch + (int64_t)blockIdx.x * 4);
}

inline int loss_blocks(int64_t n) {
  int64_t b = dv3_cdiv64(n, (int64_t)kLossBlock * 4);
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (int)b;
}

}  // namespace

int g_loss_fast_log = 1;   // dv3_debug_set(57, v): the tiled spectrogram loss takes its logarithms by v_log_f32 (0 = logf)

extern "C" int dv3_spec_loss_scratch_floats(int32_t B, int32_t T, int32_t D) {
  // block partial sums of either form: the flat grid, or one block per 64 x 64 tile (at most 1024)
  int64_t tiles = (int64_t)B * dv3_cdiv(T, 64) * dv3_cdiv(D, 64);
  if (tiles > 1024) tiles = 1024;
  const int64_t flat = loss_blocks((int64_t)B * T * D);
  return (int)(4 * (tiles > flat ? tiles : flat) + 16);
}

extern "C" int dv3_spec_loss_f32(const dv3_spec_loss_desc* d, void* stream) {
  DV3_REQUIRE(d && d->y_hat && d->y && d->out4 && d->scratch, "spec_loss: null pointer");
  DV3_REQUIRE(d->B > 0 && d->D > 0 && d->r >= 0 && d->T > d->r, "spec_loss: bad dims");
  DV3_REQUIRE(d->w_masked <= 0.f || d->lengths, "spec_loss: masked weight needs lengths");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)d->B * (d->T - d->r) * d->D;
  int nb = loss_blocks(n);
  if (d->yh_ts == 1 && d->y_ds == 1 && d->yh_ds > 1 && d->y_ts > 1 && (int64_t)d->B * d->yh_bs < (1ll << 40)) {
    // time-fastest prediction against a bin-fastest target: the tiled form
    const int t_tiles = dv3_cdiv(d->T - d->r, 64), d_tiles = dv3_cdiv(d->D, 64);
    const int64_t nt = (int64_t)d->B * t_tiles * d_tiles;
    nb = (int)(nt < 1024 ? nt : 1024);
    if (g_loss_fast_log) hipLaunchKernelGGL(spec_loss_tiled_kernel<true>, dim3(nb), dim3(kLossBlock), 0, st, *d, t_tiles, d_tiles, (int)nt);
    else hipLaunchKernelGGL(spec_loss_tiled_kernel<false>, dim3(nb), dim3(kLossBlock), 0, st, *d, t_tiles, d_tiles, (int)nt);
  } else {
    hipLaunchKernelGGL(spec_loss_kernel, dim3(nb), dim3(kLossBlock), 0, st, *d);
  }
  if (d->dyh && d->r > 0) {
    const int64_t nt = (int64_t)d->B * d->r * d->D;
    hipLaunchKernelGGL(spec_loss_tail_zero_kernel, dim3((unsigned)dv3_cdiv64(nt, 256)), dim3(256), 0,
                       st, *d);
  }
  hipLaunchKernelGGL(spec_loss_finish_kernel, dim3(1), dim3(256), 0, st, *d, nb);
  return dv3_check_launch("spec_loss_f32");
}

extern "C" int dv3_guided_attn_loss_f32(const float* attn, const int32_t* in_len,
                                        const int32_t* out_len, float* dattn, float* out1,
                                        float* scratch, int32_t L, int32_t B, int32_t Tq, int32_t Tk,
                                        float g, float gscale, void* stream) {
  DV3_REQUIRE(attn && in_len && out_len && out1 && scratch, "guided_attn: null pointer");
  DV3_REQUIRE(L > 0 && B > 0 && Tq > 0 && Tk > 0 && g > 0.f, "guided_attn: bad dims");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = (int64_t)B * Tq * Tk;
  const int nb = loss_blocks(per);
  hipLaunchKernelGGL(guided_attn_kernel, dim3(nb), dim3(256), 0, st, attn, in_len, out_len, dattn,
                     scratch, L, B, Tq, Tk, g, gscale, (const int32_t*)nullptr, (const int32_t*)nullptr);
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, scratch, nb,
                     1.0f / (float)(per * L), out1, (const int32_t*)nullptr, 0, (const int32_t*)nullptr, 0, (int64_t)0);
  return dv3_check_launch("guided_attn_loss_f32");
}

extern "C" int dv3_guided_attn_loss_valid_f32(const float* attn, const int32_t* in_len, const int32_t* out_len,
                                              float* dattn, float* out1, float* scratch, int32_t L, int32_t B,
                                              int32_t Tq, int32_t Tk, float g, float gscale,
                                              const int32_t* tq_valid, const int32_t* tk_valid, void* stream) {
  DV3_REQUIRE(attn && in_len && out_len && out1 && scratch && tq_valid && tk_valid, "guided_attn_valid: null pointer");
  DV3_REQUIRE(L > 0 && B > 0 && Tq > 0 && Tk > 0 && g > 0.f, "guided_attn_valid: bad dims");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = (int64_t)B * Tq * Tk;
  const int nb = loss_blocks(per);
  hipLaunchKernelGGL(guided_attn_kernel, dim3(nb), dim3(256), 0, st, attn, in_len, out_len, dattn,
                     scratch, L, B, Tq, Tk, g, gscale, tq_valid, tk_valid);
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, scratch, nb, 0.f, out1, tq_valid, (int)Tq, tk_valid,
                     (int)Tk, (int64_t)L * B);
  return dv3_check_launch("guided_attn_loss_valid_f32");
}

extern "C" int dv3_bce_loss_f32(const float* p, const float* t, float* dp, float* out1,
                                float* scratch, int64_t n, float gscale, void* stream) {
  DV3_REQUIRE(p && t && out1 && scratch && n > 0, "bce_loss: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int nb = loss_blocks(n);
  hipLaunchKernelGGL(bce_kernel, dim3(nb), dim3(256), 0, st, p, t, dp, scratch, n, gscale, 0, (const int32_t*)nullptr);
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, scratch, nb, 1.0f / (float)n, out1,
                     (const int32_t*)nullptr, 0, (const int32_t*)nullptr, 0, (int64_t)0);
  return dv3_check_launch("bce_loss_f32");
}

extern "C" int dv3_bce_loss_valid_f32(const float* p, const float* t, float* dp, float* out1, float* scratch,
                                      int64_t rows, int32_t T, const int32_t* t_valid, float gscale, void* stream) {
  DV3_REQUIRE(p && t && out1 && scratch && t_valid && rows > 0 && T > 0, "bce_loss_valid: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = rows * T;
  const int nb = loss_blocks(n);
  hipLaunchKernelGGL(bce_kernel, dim3(nb), dim3(256), 0, st, p, t, dp, scratch, n, gscale, (int)T, t_valid);
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, scratch, nb, 0.f, out1, t_valid, (int)T,
                     (const int32_t*)nullptr, 0, rows);
  return dv3_check_launch("bce_loss_valid_f32");
}
