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
  const float pd_scale = p.pd_scale_dev ? p.pd_scale * p.pd_scale_dev[0] : p.pd_scale;
  float* pd = p.pd ? p.pd + row * Tk : nullptr;
  for (int n = lane; n < Tk; n += 64) {
    float v = 0.f;
    if (n >= lo && n < hi) v = expf(s[n] - mx) * inv;
    s[n] = v;
    if (pd) {
      float d = v;
      if (p.mask) d = dv3_keep(p.mask, row, p.mask_rs, n) ? v * p.drop_scale : 0.f;
      pd[n] = d * pd_scale;
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
  const float drop_scale = p.scale_dev ? p.drop_scale * p.scale_dev[0] : p.drop_scale;
  float dot = 0.f;
  for (int n = lane; n < Tk; n += 64) {
    float d = 0.f;
    if (dpd) {
      d = dpd[n] * drop_scale;   // drop_scale carries 1/(1-p) AND the forward's pd_scale
      if (p.mask && !dv3_keep(p.mask, row, p.mask_rs, n)) d = 0.f;
    }
    if (dpx) d += dpx[n];
    dot += d * pr[n];
  }
  dot = dv3_wave_sum(dot);
  for (int n = lane; n < Tk; n += 64) {
    float d = 0.f;
    if (dpd) {
      d = dpd[n] * drop_scale;   // drop_scale carries 1/(1-p) AND the forward's pd_scale
      if (p.mask && !dv3_keep(p.mask, row, p.mask_rs, n)) d = 0.f;
    }
    if (dpx) d += dpx[n];
    ds[n] = pr[n] * (d - dot);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused attention forward (AttentionLayer.forward's core, deepvoice3.py:143-171): scores = q^T k on the fp32 matrix
// cores -> padding mask -> softmax -> dropout * sqrt(Tk) -> context = v pd^T on the fp32 matrix cores, ONE launch per
// layer instead of five (scores GEMM, softmax, dropout bits aside, context GEMM, and no transposes).  Exact fp32 MFMA
// (v_mfma_f32_32x32x2_f32): attention is < 0.5 % of the step's FLOPs and feeds the 1e-4 parity of the outputs.
// One workgroup = 32 queries of one batch item, 4 waves:
//   phase 1  wave w computes the 32 x 32 score tiles of key tiles w, w+4, ...; both operands are read straight from
//            the BCT tensors in MFMA fragment order (lane = query / key, 32 lanes = 128 contiguous bytes), K = channels;
//            the scores land in LDS [32][Tk + 1]
//   phase 2  8 threads per query row: max, exp, sum; P (the returned alignment) and pd = P * keep * scale go to HBM
//            once, pd also stays in LDS
//   phase 3  wave w computes the 32 x 32 context tiles of channel tiles w, w+4, ...: A = v^T rows ([Tk][E], lanes =
//            channels, contiguous), B = pd from LDS, K = keys; ctx[b][e][t] stored with lanes along t
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void attn_fwd_kernel(const dv3_attn_fwd_desc p) {
  extern __shared__ float sc[];            // [32][ld], ld = Tk + 1 (odd: conflict-free column reads)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * 32;
  const int E = p.E, Tq = p.Tq, Tk = p.Tk;
  const int ld = Tk + 1;
  const int nkt = (Tk + 31) / 32;
  const float* __restrict__ qb = p.q + (int64_t)b * E * Tq;
  const float* __restrict__ kb = p.k + (int64_t)b * E * Tk;
  const int tq = min(t0 + l31, Tq - 1);     // clamped: rows beyond Tq are computed and dropped
  // ---- phase 1: scores ----
  for (int kt = wave; kt < nkt; kt += 4) {
    const int n = min(kt * 32 + l31, Tk - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // operands of 16 MFMAs (32 channels) are fetched as one batch, the next batch before the current MFMAs: the
    // one-load-pair-per-MFMA form left every exact-fp32 MFMA (64 cycles) waiting for two L2 round trips (round 2:
    // 149 us per launch for 2 GFLOP)
    auto fetch1 = [&](int e0, float (&a)[16], float (&bv)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = e0 + 2 * i + lhi;
        const int ee = min(e, E - 1);
        const float av = qb[(int64_t)ee * Tq + tq];
        bv[i] = kb[(int64_t)ee * Tk + n];
        a[i] = e < E ? av : 0.f;
      }
    };
    float a0[16], b0[16], a1[16], b1[16];
    fetch1(0, a0, b0);
    for (int e0 = 0; e0 < E; e0 += 64) {
      if (e0 + 32 < E) fetch1(e0 + 32, a1, b1);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc, 0, 0, 0);
      if (e0 + 32 >= E) break;
      if (e0 + 64 < E) fetch1(e0 + 64, a0, b0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc, 0, 0, 0);
    }
    // C layout: col = lane & 31 (key), row = (r & 3) + 8 * (r >> 2) + 4 * lhi (query)
    const int col = kt * 32 + l31;
    if (col < Tk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * lhi) * ld + col] = acc[r];
    }
  }
  __syncthreads();
  // ---- phase 2: softmax over keys, 8 threads per query row ----
  {
    const int row = tid >> 3, sub = tid & 7;
    const int t = t0 + row;
    int hi = Tk;
    if (p.key_len) hi = min(hi, p.key_len[b]);
    float* s = sc + row * ld;
    float mx = -INFINITY;
    for (int n = sub; n < hi; n += 8) mx = fmaxf(mx, s[n]);
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
    for (int n = sub; n < hi; n += 8) sum += expf(s[n] - mx);
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float inv = 1.0f / sum;
    const float pd_scale = p.pd_scale_dev ? p.pd_scale * p.pd_scale_dev[0] : p.pd_scale;
    const int64_t grow = (int64_t)b * Tq + t;
    for (int n = sub; n < Tk; n += 8) {
      float v = 0.f;
      if (n < hi) v = expf(s[n] - mx) * inv;
      float d = v;
      if (p.mask && t < Tq) d = dv3_keep(p.mask, grow, p.mask_rs, n) ? v * p.drop_scale : 0.f;
      d *= pd_scale;
      s[n] = d;
      if (t < Tq) {
        p.P[grow * Tk + n] = v;
        p.pd[grow * Tk + n] = d;
      }
    }
  }
  __syncthreads();
  // ---- phase 3: context[e][t] = sum_n vT[n][e] * pd[t][n] ----
  const float* __restrict__ vtb = p.vT + (int64_t)b * Tk * E;
  for (int et = wave; et * 32 < E; et += 4) {
    const int e = min(et * 32 + l31, E - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto fetch3 = [&](int n0, float (&a)[16], float (&bv)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + 2 * i + lhi;
        const int nn = min(n, Tk - 1);
        const float av = vtb[(int64_t)nn * E + e];
        bv[i] = sc[l31 * ld + nn];
        a[i] = n < Tk ? av : 0.f;
      }
    };
    float a0[16], b0[16], a1[16], b1[16];
    fetch3(0, a0, b0);
    for (int n0 = 0; n0 < Tk; n0 += 64) {
      if (n0 + 32 < Tk) fetch3(n0 + 32, a1, b1);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc, 0, 0, 0);
      if (n0 + 32 >= Tk) break;
      if (n0 + 64 < Tk) fetch3(n0 + 64, a0, b0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc, 0, 0, 0);
    }
    // C: col = lane & 31 (query), rows = channels
    const int t = t0 + l31;
    if (t < Tq) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ec = et * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (ec < E) p.ctx[((int64_t)b * E + ec) * Tq + t] = acc[r];
      }
    }
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

extern "C" int dv3_attn_fwd_f32(const dv3_attn_fwd_desc* d, void* stream) {
  DV3_REQUIRE(d && d->q && d->k && d->vT && d->ctx && d->P && d->pd, "attn_fwd: null pointer");
  DV3_REQUIRE(d->B > 0 && d->E > 0 && d->Tq > 0 && d->Tk > 0 && d->B <= 65535, "attn_fwd: bad dims");
  if (d->mask) DV3_REQUIRE(d->mask_rs * 32 >= d->Tk, "attn_fwd: mask row stride too small");
  const size_t lds = (size_t)32 * (d->Tk + 1) * sizeof(float);
  DV3_REQUIRE(lds <= 64 * 1024, "attn_fwd: Tk = %d too long for the fused kernel (use the unfused path)", d->Tk);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(dv3_cdiv(d->Tq, 32), d->B), dim3(256), lds, (hipStream_t)stream, *d);
  return dv3_check_launch("attn_fwd");
}

extern "C" int dv3_attn_argmax_i32(const float* p_row, int32_t Tk, int32_t* out, void* stream) {
  DV3_REQUIRE(p_row && out && Tk > 0, "attn_argmax: bad args");
  hipLaunchKernelGGL(attn_argmax_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p_row, Tk, out);
  return dv3_check_launch("attn_argmax_i32");
}
