// Attention softmax for deepvoice3_pytorch/deepvoice3.py:145-165 (AttentionLayer.forward):
// padding mask (-inf), monotonic window mask, softmax over keys, dropout.  The score and
// context contractions around it run on the MFMA tap-GEMM kernels (conv_gemm / wgrad_gemm).
// One wave per (b, tq) row, lanes along keys.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void attn_softmax_kernel(const dv3_softmax_desc p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)p.B * p.Tq) return;
  const int b = (int)(row / p.Tq);
  const int Tk = p.Tk;
  float* s = p.s + row * Tk;
  int lo = 0, hi = Tk;
  if (p.key_len) hi = min(hi, p.key_len[b]);
  if (p.last_attended) {
    // deepvoice3.py:150-156: keep [last - back, last + ahead)
    const int la = p.last_attended[0];
    const int wlo = la - p.win_back, whi = la + p.win_ahead;
    if (wlo > 0) lo = max(lo, wlo);
    if (whi < Tk) hi = min(hi, whi);
  }
  float mx = -INFINITY;
  for (int n = lo + lane; n < hi; n += 64) mx = fmaxf(mx, s[n]);
  mx = dv3_wave_max(mx);
  float sum = 0.f;
  for (int n = lo + lane; n < hi; n += 64) sum += expf(s[n] - mx);
  sum = dv3_wave_sum(sum);
  const float inv = 1.0f / sum;
  float* pd = p.pd ? p.pd + row * Tk : nullptr;
  for (int n = lane; n < Tk; n += 64) {
    float v = 0.f;
    if (n >= lo && n < hi) v = expf(s[n] - mx) * inv;
    s[n] = v;
    if (pd) {
      float d = v;
      if (p.mask) d = dv3_keep(p.mask, row, p.mask_rs, n) ? v * p.drop_scale : 0.f;
      pd[n] = d * p.pd_scale;
    }
  }
}

// ds = p * (dp - sum_k dp*p),  dp = dpd*keep*scale + dp_direct
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const dv3_softmax_bwd_desc p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)p.B * p.Tq) return;
  const int Tk = p.Tk;
  const float* pr = p.p + row * Tk;
  const float* dpd = p.dpd ? p.dpd + row * Tk : nullptr;
  const float* dpx = p.dp_direct ? p.dp_direct + row * Tk : nullptr;
  float* ds = p.ds + row * Tk;
  float dot = 0.f;
  for (int n = lane; n < Tk; n += 64) {
    float d = 0.f;
    if (dpd) {
      d = dpd[n] * p.drop_scale;   // drop_scale carries 1/(1-p) AND the forward's pd_scale
      if (p.mask && !dv3_keep(p.mask, row, p.mask_rs, n)) d = 0.f;
    }
    if (dpx) d += dpx[n];
    dot += d * pr[n];
  }
  dot = dv3_wave_sum(dot);
  for (int n = lane; n < Tk; n += 64) {
    float d = 0.f;
    if (dpd) {
      d = dpd[n] * p.drop_scale;   // drop_scale carries 1/(1-p) AND the forward's pd_scale
      if (p.mask && !dv3_keep(p.mask, row, p.mask_rs, n)) d = 0.f;
    }
    if (dpx) d += dpx[n];
    ds[n] = pr[n] * (d - dot);
  }
}

// argmax over keys of row (b = 0, last query) -> last_attended (deepvoice3.py:445: the
// reference takes batch item 0 only)
__global__ __launch_bounds__(64) void attn_argmax_kernel(const float* __restrict__ p, int Tk,
                                                         int32_t* __restrict__ out) {
  const int lane = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int n = lane; n < Tk; n += 64) {
    const float v = p[n];
    if (v > best) { best = v; bi = n; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[0] = bi;
}

}  // namespace

extern "C" int dv3_attn_softmax_f32(const dv3_softmax_desc* d, void* stream) {
  DV3_REQUIRE(d && d->s && d->B > 0 && d->Tq > 0 && d->Tk > 0, "attn_softmax: bad args");
  if (d->mask) DV3_REQUIRE(d->pd && d->mask_rs * 32 >= d->Tk, "attn_softmax: mask needs pd and a wide enough row stride");
  const int64_t rows = (int64_t)d->B * d->Tq;
  hipLaunchKernelGGL(attn_softmax_kernel, dim3((unsigned)dv3_cdiv64(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, *d);
  return dv3_check_launch("attn_softmax_f32");
}

extern "C" int dv3_attn_softmax_bwd_f32(const dv3_softmax_bwd_desc* d, void* stream) {
  DV3_REQUIRE(d && d->p && d->ds && (d->dpd || d->dp_direct), "attn_softmax_bwd: bad args");
  DV3_REQUIRE(d->B > 0 && d->Tq > 0 && d->Tk > 0, "attn_softmax_bwd: bad dims");
  const int64_t rows = (int64_t)d->B * d->Tq;
  hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((unsigned)dv3_cdiv64(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, *d);
  return dv3_check_launch("attn_softmax_bwd_f32");
}

extern "C" int dv3_attn_argmax_i32(const float* p_row, int32_t Tk, int32_t* out, void* stream) {
  DV3_REQUIRE(p_row && out && Tk > 0, "attn_argmax: bad args");
  hipLaunchKernelGGL(attn_argmax_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p_row, Tk, out);
  return dv3_check_launch("attn_argmax_i32");
}
