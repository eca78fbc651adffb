// Audio inverse on the GPU: what synthesis.py does on the host after the model,
// audio.inv_spectrogram (audio.py:37-43): _denormalize -> _db_to_amp (audio.py:84-93) -> magnitude
// ** power -> phase reconstruction -> istft -> inv_preemphasis (audio.py:26-28).
//
// The reference delegates phase reconstruction to the third-party `lws` package (not vendored,
// "parity unpinned"); the north-star asks for Griffin-Lim as FFT + reduction kernels.
// These kernels implement Griffin-Lim with torch.stft / torch.istft conventions (periodic Hann
// window of n_fft = 1024, hop 256, center=True with reflect padding, onesided 513 bins, istft
// normalised by the overlap-added squared window) so the CPU oracle (oracle/audio_oracle.py)
// can be an independent restatement on torch-CPU FFTs.
//
// One workgroup (256 threads) transforms one 1024-point frame in LDS: Stockham auto-sort radix-4,
// five passes, one butterfly per thread per pass, twiddles from sincospif.  Frames are
// independent, so the grid is B*T workgroups; the only cross-frame step is the overlap-add, a
// gather over <= 4 frames per sample (deterministic, no atomics).
#include "common.h"

namespace {

constexpr int NFFT = 1024;
constexpr int NBIN = NFFT / 2 + 1;
constexpr float PI_F = 3.14159265358979323846f;

struct cplx {
  float x, y;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }

__device__ __forceinline__ float hann(int n) { return 0.5f - 0.5f * cospif(2.0f * (float)n / (float)NFFT); }

// W[j] = exp(+2 pi i j / 1024), filled once per workgroup (4 sincospif per thread instead of 3 per butterfly and
// pass: the transcendental calls were most of a frame's instructions).  hann(n) = 0.5 - 0.5 * Re W[n].
__device__ __forceinline__ void fill_twiddles(cplx* W, int tid) {
  for (int j = tid; j < NFFT; j += 256) {
    float s, c;
    sincospif(2.0f * (float)j / (float)NFFT, &s, &c);
    W[j] = cplx{c, s};
  }
}
__device__ __forceinline__ float hann_t(const cplx* W, int n) { return 0.5f - 0.5f * W[n].x; }

// In-LDS 1024-point complex FFT (SIGN = -1 forward, +1 inverse, unnormalised).  `a` holds the
// input in natural order; the result ends in `b` (5 passes: a->b->a->b->a->b).  256 threads.
template <int SIGN>
__device__ __forceinline__ void fft1024(cplx* a, cplx* b, int tid, const cplx* W = nullptr) {
  cplx* src = a;
  cplx* dst = b;
#pragma unroll
  for (int ns = 1; ns < NFFT; ns *= 4) {
    __syncthreads();
    const int k = tid & (ns - 1);
    const float ang = (float)SIGN * 2.0f * (float)k / (float)(ns * 4);  // in units of pi
    cplx v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = src[tid + r * (NFFT / 4)];
      if (r) {
        if (W) {       // exp(SIGN * 2 pi i k r / (4 ns)) = W[k r 256/ns] (conjugated for SIGN < 0); k r / (4 ns) < 3/4
          const cplx w = W[k * r * (NFFT / 4 / ns)];
          v[r] = cmul(v[r], cplx{w.x, SIGN < 0 ? -w.y : w.y});
        } else {
          float s, c;
          sincospif(ang * (float)r, &s, &c);
          v[r] = cmul(v[r], cplx{c, s});
        }
      }
    }
    // radix-4 DFT, natural order: X[q] = sum_m v[m] * w^(q*m), w = exp(SIGN * i*pi/2)
    const cplx s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
    const cplx s13 = cadd(v[1], v[3]), d13 = csub(v[1], v[3]);
    const cplx jd13 = (SIGN < 0) ? cplx{d13.y, -d13.x} : cplx{-d13.y, d13.x};  // w * d13
    const int j0 = ((tid - k) << 2) + k;  // (tid / ns) * ns * 4 + k
    dst[j0] = cadd(s02, s13);
    dst[j0 + ns] = cadd(d02, jd13);
    dst[j0 + 2 * ns] = csub(s02, s13);
    dst[j0 + 3 * ns] = csub(d02, jd13);
    cplx* t = src; src = dst; dst = t;
  }
  __syncthreads();
}

// mag[b][t][k] = (10^((clip(x,0,1)*(-min_db) + min_db + ref_db) / 20)) ^ power      audio.py:39-41,84-93
__global__ void gl_prepare_kernel(const float* __restrict__ lin, float* __restrict__ mag, int64_t n,
                                  float min_db, float ref_db, float power) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float x = fminf(fmaxf(lin[i], 0.f), 1.f);
    const float db = x * (-min_db) + min_db + ref_db;
    mag[i] = exp2f(db * 0.05f * power * 3.32192809488736234787f);  // 10^(db/20*power)
  }
}

// frames[b][t][n] = hann[n] * irfft(mag[b][t][:] * phasor[b][t][:])[n]   (phasor NULL: zero phase)
// LWS: the conventions of lws.lws(1024, hop) (audio.py:54-55; oracle/audio_oracle.py: lws_windows): `sw` = the
// perfect-reconstruction synthesis window (the overlap-add normaliser is folded into it)
template <bool LWS>
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ mag,
                                                           const float* __restrict__ phasor,
                                                           float* __restrict__ frames, const float* __restrict__ sw) {
  __shared__ cplx A[NFFT], Bf[NFFT];
  const int tid = threadIdx.x;
  const int64_t fr = blockIdx.x;
  const float* m = mag + fr * NBIN;
  const float* ph = phasor ? phasor + fr * NBIN * 2 : nullptr;
  for (int k = tid; k <= NFFT / 2; k += 256) {
    cplx z{m[k], 0.f};
    if (ph) z = cplx{m[k] * ph[2 * k], m[k] * ph[2 * k + 1]};
    if (k == 0 || k == NFFT / 2) z.y = 0.f;  // c2r ignores the imaginary part of DC / Nyquist
    A[k] = z;
    if (k > 0 && k < NFFT / 2) A[NFFT - k] = cplx{z.x, -z.y};
  }
  fft1024<+1>(A, Bf, tid);
  float* out = frames + fr * NFFT;
  for (int n = tid; n < NFFT; n += 256) out[n] = Bf[n].x * (1.0f / NFFT) * (LWS ? sw[n] : hann(n));
}

// y[b][i] = sum_t frames[b][t][p - t*hop] / sum_t hann^2[p - t*hop],  p = i + NFFT/2, i < hop*(T-1)
// LWS: p = i + (NFFT - hop) (the zero padding lws strips), no division (the synthesis window carries the normaliser)
template <bool LWS>
__global__ void ola_kernel(const float* __restrict__ frames, float* __restrict__ y, int T, int hop,
                           int L) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const int p = i + (LWS ? NFFT - hop : NFFT / 2);
  int t_hi = p / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  int t_lo = (p - NFFT + hop) / hop;  // smallest t with p - t*hop <= NFFT-1  (ceil((p-NFFT+1)/hop))
  if (p - NFFT + 1 <= 0) t_lo = 0;
  float acc = 0.f, wsum = 0.f;
  const float* fb = frames + (int64_t)b * T * NFFT;
  // interior samples of the hop = N/4 periodic-Hann framing meet four frames whose squared windows sum to exactly 3/2
  // (sum_j sin^4(x + j pi/4) = 3/2): no transcendental per tap there; the edges keep the general form
  const bool interior = LWS || (hop * 4 == NFFT && t_hi - t_lo == 3 && p - t_lo * hop < NFFT && p - t_hi * hop >= 0);
  for (int t = t_lo; t <= t_hi; ++t) {
    const int n = p - t * hop;
    if (n < 0 || n >= NFFT) continue;
    acc += fb[(int64_t)t * NFFT + n];
    if (!interior) {
      const float w = hann(n);
      wsum += w * w;
    }
  }
  y[(int64_t)b * L + i] = LWS ? acc : (interior ? acc * (2.0f / 3.0f) : acc / wsum);
}

// phasor[b][t][k] = Z / max(|Z|, 1e-8), Z = rfft(hann * reflect_pad(y[b])[t*hop : t*hop + NFFT])[k]
// (spec, optional: Z itself, for tests / spectral convergence)
// LWS: Z = rfft(aw * zero_pad(y[b], NFFT - hop)[t*hop : t*hop + NFFT]) -- lws.lws(1024, hop).stft (sqrt-Hann analysis window
// `aw`, zeros instead of reflection)
template <bool LWS>
__global__ __launch_bounds__(256) void stft_phase_kernel(const float* __restrict__ y,
                                                         float* __restrict__ phasor,
                                                         float* __restrict__ spec,
                                                         float* __restrict__ mag_bct, int T, int hop, int L,
                                                         const float* __restrict__ aw) {
  __shared__ cplx A[NFFT], Bf[NFFT];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / T, t = blockIdx.x - b * T;
  const float* yb = y + (int64_t)b * L;
  for (int n = tid; n < NFFT; n += 256) {
    if constexpr (LWS) {
      const int i = t * hop + n - (NFFT - hop);
      A[n] = cplx{(i >= 0 && i < L) ? yb[i] * aw[n] : 0.f, 0.f};
    } else {
      int i = t * hop + n - NFFT / 2;  // index into the un-padded signal
      if (i < 0) i = -i;
      if (i >= L) i = 2 * (L - 1) - i;
      A[n] = cplx{yb[i] * hann(n), 0.f};
    }
  }
  fft1024<-1>(A, Bf, tid);
  const int64_t fr = blockIdx.x;
  for (int k = tid; k <= NFFT / 2; k += 256) {
    const cplx z = Bf[k];
    if (spec) {
      spec[(fr * NBIN + k) * 2] = z.x;
      spec[(fr * NBIN + k) * 2 + 1] = z.y;
    }
    if (mag_bct) mag_bct[((int64_t)b * NBIN + k) * T + t] = sqrtf(z.x * z.x + z.y * z.y);
    if (phasor) {
      const float inv = 1.0f / fmaxf(sqrtf(z.x * z.x + z.y * z.y), 1e-8f);
      phasor[(fr * NBIN + k) * 2] = z.x * inv;
      phasor[(fr * NBIN + k) * 2 + 1] = z.y * inv;
    }
  }
}

// One Griffin-Lim projection per frame without leaving LDS: STFT of the current signal estimate -> unit phase ->
// times the target magnitude -> inverse FFT -> synthesis window.  Equals stft_phase_kernel followed by
// istft_frames_kernel (same arithmetic, same order) minus the phasor round trip through HBM (8 KB per frame).
__global__ __launch_bounds__(256) void gl_project_kernel(const float* __restrict__ y, const float* __restrict__ mag,
                                                         float* __restrict__ frames, int T, int hop, int L) {
  __shared__ cplx A[NFFT], Bf[NFFT], W[NFFT];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / T, t = blockIdx.x - b * T;
  const float* yb = y + (int64_t)b * L;
  fill_twiddles(W, tid);
  __syncthreads();
  for (int n = tid; n < NFFT; n += 256) {
    int i = t * hop + n - NFFT / 2;
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    A[n] = cplx{yb[i] * hann_t(W, n), 0.f};
  }
  fft1024<-1>(A, Bf, tid, W);
  const int64_t fr = blockIdx.x;
  const float* m = mag + fr * NBIN;
  for (int k = tid; k <= NFFT / 2; k += 256) {
    const cplx z = Bf[k];
    const float inv = 1.0f / fmaxf(sqrtf(z.x * z.x + z.y * z.y), 1e-8f);
    const float px = z.x * inv, py = z.y * inv;       // the unit phasor stft_phase_kernel would store
    cplx w{m[k] * px, m[k] * py};
    if (k == 0 || k == NFFT / 2) w.y = 0.f;
    A[k] = w;
    if (k > 0 && k < NFFT / 2) A[NFFT - k] = cplx{w.x, -w.y};
  }
  fft1024<+1>(A, Bf, tid, W);
  float* out = frames + fr * NFFT;
  for (int n = tid; n < NFFT; n += 256) out[n] = Bf[n].x * (1.0f / NFFT) * hann_t(W, n);
}

// The same projection for TWO frames per workgroup: both input frames are real, so they ride one complex FFT as
// z = x1 + i x2 (X1[k] = (Z[k] + conj Z[N-k]) / 2, X2[k] = (Z[k] - conj Z[N-k]) / 2i), and the two Hermitian target
// spectra ride one inverse FFT as V = Y1 + i Y2 (y1 = Re v, y2 = Im v): half the FFT passes -- the LDS traffic that
// bounds this kernel -- per frame.  Frames (2q, 2q+1) of one batch item; an odd last frame pairs with nothing.
template <bool LWS>
__global__ __launch_bounds__(256) void gl_project2_kernel(const float* __restrict__ y, const float* __restrict__ mag,
                                                          float* __restrict__ frames, int T, int hop, int L, int TP,
                                                          const float* __restrict__ aw, const float* __restrict__ sw) {
  __shared__ cplx A[NFFT], Bf[NFFT], W[NFFT];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / TP, t1 = 2 * (blockIdx.x - b * TP);
  const bool two = t1 + 1 < T;
  const float* yb = y + (int64_t)b * L;
  fill_twiddles(W, tid);
  __syncthreads();
  for (int n = tid; n < NFFT; n += 256) {
    if constexpr (LWS) {      // lws framing: zeros outside the signal, sqrt-Hann analysis window from the table
      const int i1 = t1 * hop + n - (NFFT - hop), i2 = i1 + hop;
      const float h = aw[n];
      A[n] = cplx{(i1 >= 0 && i1 < L) ? yb[i1] * h : 0.f, (two && i2 >= 0 && i2 < L) ? yb[i2] * h : 0.f};
    } else {
      int i1 = t1 * hop + n - NFFT / 2, i2 = i1 + hop;
      if (i1 < 0) i1 = -i1;
      if (i1 >= L) i1 = 2 * (L - 1) - i1;
      if (i2 < 0) i2 = -i2;
      if (i2 >= L) i2 = 2 * (L - 1) - i2;
      const float h = hann_t(W, n);
      A[n] = cplx{yb[i1] * h, two ? yb[i2] * h : 0.f};
    }
  }
  fft1024<-1>(A, Bf, tid, W);
  const int64_t fr = (int64_t)b * T + t1;
  const float* m1 = mag + fr * NBIN;
  const float* m2 = m1 + (two ? NBIN : 0);
  for (int k = tid; k <= NFFT / 2; k += 256) {
    const cplx zk = Bf[k], zn = Bf[(NFFT - k) & (NFFT - 1)];
    // X1 = (zk + conj zn) / 2, X2 = (zk - conj zn) / (2i) = ((zk.y + zn.y) - i (zk.x - zn.x)) / 2
    const cplx x1{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
    const cplx x2{0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x)};
    const float inv1 = 1.0f / fmaxf(sqrtf(x1.x * x1.x + x1.y * x1.y), 1e-8f);
    const float inv2 = 1.0f / fmaxf(sqrtf(x2.x * x2.x + x2.y * x2.y), 1e-8f);
    cplx w1{m1[k] * (x1.x * inv1), m1[k] * (x1.y * inv1)};
    cplx w2{m2[k] * (x2.x * inv2), m2[k] * (x2.y * inv2)};
    if (k == 0 || k == NFFT / 2) w1.y = w2.y = 0.f;
    if (!two) w2 = cplx{0.f, 0.f};
    // V[k] = w1 + i w2 ; V[N-k] = conj(w1) + i conj(w2)
    A[k] = cplx{w1.x - w2.y, w1.y + w2.x};
    if (k > 0 && k < NFFT / 2) A[NFFT - k] = cplx{w1.x + w2.y, -w1.y + w2.x};
  }
  fft1024<+1>(A, Bf, tid, W);
  float* out = frames + fr * NFFT;
  for (int n = tid; n < NFFT; n += 256) {
    const float h = (LWS ? sw[n] : hann_t(W, n)) * (1.0f / NFFT);
    out[n] = Bf[n].x * h;
    if (two) out[NFFT + n] = Bf[n].y * h;
  }
}

// y[n] = x[n] + coef * y[n-1] per row (scipy.signal.lfilter([1], [1, -coef]); audio.py:26-28), in
// place.  One workgroup per row: 256 contiguous segments scanned locally, carries chained in LDS.
__global__ __launch_bounds__(256) void deemphasis_kernel(float* __restrict__ y, int L, float coef) {
  __shared__ float seg_end[256];
  __shared__ float carry[256];
  const int tid = threadIdx.x;
  float* row = y + (int64_t)blockIdx.x * L;
  const int len = (L + 255) / 256;
  const int lo = min(tid * len, L), hi = min(lo + len, L);
  float v = 0.f;
  for (int i = lo; i < hi; ++i) {
    v = row[i] + coef * v;
    row[i] = v;
  }
  seg_end[tid] = v;
  __syncthreads();
  if (tid == 0) {
    float c = 0.f;  // value of y just before segment s
    for (int s = 0; s < 256; ++s) {
      carry[s] = c;
      const int n = min((s + 1) * len, L) - min(s * len, L);
      c = seg_end[s] + powf(coef, (float)n) * c;
    }
  }
  __syncthreads();
  const float c = carry[tid];
  if (c != 0.f) {
    float g = coef;
    for (int i = lo; i < hi; ++i) {
      row[i] += g * c;
      g *= coef;
    }
  }
}

// y[n] = x[n] - coef * x[n-1], y[0] = x[0]   (nnmnkwii.preprocessing.preemphasis, audio.py:21-23)
__global__ void preemphasis_kernel(const float* __restrict__ x, float* __restrict__ y, int L, float coef) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const float* xr = x + (int64_t)b * L;
  y[(int64_t)b * L + i] = i ? xr[i] - coef * xr[i - 1] : xr[0];
}

// out = clip((20*log10(max(min_level, x)) - ref_db - min_db) / -min_db, 0, 1)   audio.py:34-35,79-89
__global__ void amp_to_db_norm_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                      float min_db, float ref_db) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float min_level = exp2f(min_db * 0.05f * 3.32192809488736234787f);
  for (; i < n; i += stride) {
    const float db = 20.0f * log10f(fmaxf(min_level, x[i])) - ref_db;
    out[i] = fminf(fmaxf((db - min_db) / (-min_db), 0.f), 1.f);
  }
}

}  // namespace

extern "C" int dv3_preemphasis_f32(const float* x, float* y, int32_t B, int32_t L, float coef,
                                   void* stream) {
  DV3_REQUIRE(x && y && B > 0 && L > 0, "preemphasis: bad arguments");
  hipLaunchKernelGGL(preemphasis_kernel, dim3(dv3_cdiv(L, 256), B), dim3(256), 0, (hipStream_t)stream, x, y,
                     L, coef);
  return dv3_check_launch("preemphasis");
}

extern "C" int dv3_amp_to_db_norm_f32(const float* x, float* out, int64_t n, float min_level_db,
                                      float ref_level_db, void* stream) {
  DV3_REQUIRE(x && out && n > 0, "amp_to_db_norm: bad arguments");
  const int blocks = (int)(dv3_cdiv64(n, 256) < 4096 ? dv3_cdiv64(n, 256) : 4096);
  hipLaunchKernelGGL(amp_to_db_norm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, n,
                     min_level_db, ref_level_db);
  return dv3_check_launch("amp_to_db_norm");
}

extern "C" int dv3_gl_prepare_f32(const float* lin, float* mag, int64_t n, float min_level_db,
                                  float ref_level_db, float power, void* stream) {
  DV3_REQUIRE(lin && mag && n > 0, "gl_prepare: bad arguments");
  const int blocks = (int)(dv3_cdiv64(n, 256) < 4096 ? dv3_cdiv64(n, 256) : 4096);
  hipLaunchKernelGGL(gl_prepare_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, lin, mag, n,
                     min_level_db, ref_level_db, power);
  return dv3_check_launch("gl_prepare");
}

extern "C" int dv3_istft_frames_f32(const float* mag, const float* phasor, float* frames, int32_t B,
                                    int32_t T, void* stream) {
  DV3_REQUIRE(mag && frames && B > 0 && T > 0, "istft_frames: bad arguments");
  hipLaunchKernelGGL(istft_frames_kernel<false>, dim3((unsigned)((int64_t)B * T)), dim3(256), 0,
                     (hipStream_t)stream, mag, phasor, frames, (const float*)nullptr);
  return dv3_check_launch("istft_frames");
}
extern "C" int dv3_lws_istft_frames_f32(const float* mag, const float* phasor, const float* swin, float* frames, int32_t B,
                                        int32_t T, void* stream) {
  DV3_REQUIRE(mag && frames && swin && B > 0 && T > 0, "lws_istft_frames: bad arguments");
  hipLaunchKernelGGL(istft_frames_kernel<true>, dim3((unsigned)((int64_t)B * T)), dim3(256), 0,
                     (hipStream_t)stream, mag, phasor, frames, swin);
  return dv3_check_launch("lws_istft_frames");
}

extern "C" int dv3_overlap_add_f32(const float* frames, float* y, int32_t B, int32_t T, int32_t hop,
                                   void* stream) {
  DV3_REQUIRE(frames && y && B > 0 && T > 1 && hop > 0 && hop <= 1024, "overlap_add: bad arguments");
  const int L = hop * (T - 1);
  hipLaunchKernelGGL(ola_kernel<false>, dim3(dv3_cdiv(L, 256), B), dim3(256), 0, (hipStream_t)stream, frames, y,
                     T, hop, L);
  return dv3_check_launch("overlap_add");
}

// lws framing: T frames cover (T + 1) * hop - 1024 samples (a signal padded with 1024 - hop zeros on both sides)
static inline int lws_len(int T, int hop) { return (T + 1) * hop - NFFT; }
extern "C" int dv3_lws_overlap_add_f32(const float* frames, float* y, int32_t B, int32_t T, int32_t hop, void* stream) {
  DV3_REQUIRE(frames && y && B > 0 && T > 1 && hop > 0 && hop <= 1024 && lws_len(T, hop) > 0, "lws_overlap_add: bad arguments");
  const int L = lws_len(T, hop);
  hipLaunchKernelGGL(ola_kernel<true>, dim3(dv3_cdiv(L, 256), B), dim3(256), 0, (hipStream_t)stream, frames, y, T, hop, L);
  return dv3_check_launch("lws_overlap_add");
}
extern "C" int dv3_lws_stft_f32(const float* y, const float* awin, float* phasor, float* spec, float* mag_bct, int32_t B,
                                int32_t T, int32_t hop, int32_t L, void* stream) {
  DV3_REQUIRE(y && awin && (phasor || spec || mag_bct) && B > 0 && T > 1 && hop > 0 && hop <= 1024 && L > 0,
              "lws_stft: bad arguments");
  DV3_REQUIRE(L <= lws_len(T, hop) && L > lws_len(T - 1, hop), "lws_stft: %d frames do not frame %d samples at hop %d", T, L, hop);
  hipLaunchKernelGGL(stft_phase_kernel<true>, dim3((unsigned)((int64_t)B * T)), dim3(256), 0, (hipStream_t)stream, y, phasor,
                     spec, mag_bct, T, hop, L, awin);
  return dv3_check_launch("lws_stft");
}
extern "C" int dv3_lws_gl_project_f32(const float* y, const float* mag, const float* awin, const float* swin, float* frames,
                                      int32_t B, int32_t T, int32_t hop, void* stream) {
  DV3_REQUIRE(y && mag && frames && awin && swin && B > 0 && T > 1 && hop > 0 && hop <= 1024 && lws_len(T, hop) > 0,
              "lws_gl_project: bad arguments");
  const int TP = (T + 1) / 2;
  hipLaunchKernelGGL(gl_project2_kernel<true>, dim3((unsigned)((int64_t)B * TP)), dim3(256), 0, (hipStream_t)stream, y, mag,
                     frames, T, hop, lws_len(T, hop), TP, awin, swin);
  return dv3_check_launch("lws_gl_project");
}

extern "C" int dv3_stft_phase_f32(const float* y, float* phasor, float* spec, float* mag_bct, int32_t B,
                                  int32_t T, int32_t hop, void* stream) {
  DV3_REQUIRE(y && (phasor || spec || mag_bct) && B > 0 && T > 1 && hop > 0, "stft_phase: bad arguments");
  const int L = hop * (T - 1);
  DV3_REQUIRE(L > 512, "stft_phase: signal shorter than the reflect padding");
  hipLaunchKernelGGL(stft_phase_kernel<false>, dim3((unsigned)((int64_t)B * T)), dim3(256), 0,
                     (hipStream_t)stream, y, phasor, spec, mag_bct, T, hop, L, (const float*)nullptr);
  return dv3_check_launch("stft_phase");
}

extern "C" int dv3_gl_project_f32(const float* y, const float* mag, float* frames, int32_t B, int32_t T, int32_t hop,
                                  void* stream) {
  DV3_REQUIRE(y && mag && frames && B > 0 && T > 1 && hop > 0, "gl_project: bad arguments");
  const int L = hop * (T - 1);
  DV3_REQUIRE(L > 512, "gl_project: signal shorter than the reflect padding");
  const int TP = (T + 1) / 2;     // two real frames per complex FFT
  hipLaunchKernelGGL(gl_project2_kernel<false>, dim3((unsigned)((int64_t)B * TP)), dim3(256), 0, (hipStream_t)stream, y, mag,
                     frames, T, hop, L, TP, (const float*)nullptr, (const float*)nullptr);
  return dv3_check_launch("gl_project");
}

// The same filter in parallel over the row.  |coef| < 1, so the response to a sample dies off geometrically: a chunk of
// DEEMPH_CH outputs is exact to fp32 rounding when its recursion starts DEEMPH_W samples earlier from zero state (what is
// dropped is bounded by coef^W / (1 - coef) * max|x|: 1e-12 * max|x| at the presets' 0.97) -- no carry crosses a
// workgroup.  One workgroup = one chunk of one row: every thread scans 16 contiguous samples in registers (four 16-byte
// loads), the 256 segment carries are chained through LDS, outputs leave as 16-byte stores.  64 rows x 206 k samples:
// 4288 workgroups instead of the 64 of the serial form (938 us, round 2).
constexpr int DEEMPH_W = 1024, DEEMPH_CH = 3072, DEEMPH_E = (DEEMPH_W + DEEMPH_CH) / 256;
static_assert(DEEMPH_E == 16, "sixteen samples per thread");
__global__ __launch_bounds__(256) void deemphasis_chunk_kernel(const float* __restrict__ x, float* __restrict__ y, int L,
                                                               float coef) {
  __shared__ float seg_end[256];
  __shared__ float carry[256];
  const int tid = threadIdx.x;
  const float* xr = x + (int64_t)blockIdx.y * L;
  float* yr = y + (int64_t)blockIdx.y * L;
  const int out0 = blockIdx.x * DEEMPH_CH;                    // first output of the chunk
  const int i0 = out0 - DEEMPH_W + tid * DEEMPH_E;            // this thread's first sample (may lie before the row)
  float v[DEEMPH_E];
  const bool vec = i0 >= 0 && i0 + DEEMPH_E <= L && ((((uintptr_t)(xr + i0)) & 15) == 0);
  if (vec) {
#pragma unroll
    for (int q = 0; q < DEEMPH_E / 4; ++q) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(xr + i0 + 4 * q);
      v[4 * q] = t4[0]; v[4 * q + 1] = t4[1]; v[4 * q + 2] = t4[2]; v[4 * q + 3] = t4[3];
    }
  } else {
#pragma unroll
    for (int j = 0; j < DEEMPH_E; ++j) v[j] = (i0 + j >= 0 && i0 + j < L) ? xr[i0 + j] : 0.f;
  }
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < DEEMPH_E; ++j) {
    acc = v[j] + coef * acc;
    v[j] = acc;
  }
  seg_end[tid] = acc;
  __syncthreads();
  if (tid == 0) {
    float cE = coef;
#pragma unroll
    for (int j = 1; j < DEEMPH_E; ++j) cE *= coef;            // coef^16
    float c = 0.f;                                            // filter state just before segment s
    for (int s = 0; s < 256; ++s) {
      carry[s] = c;
      c = seg_end[s] + cE * c;
    }
  }
  __syncthreads();
  if (i0 + DEEMPH_E <= out0 || i0 >= L) return;               // warm-up segments and segments past the row write nothing
  const float c = carry[tid];
  float g = coef;
#pragma unroll
  for (int j = 0; j < DEEMPH_E; ++j) {
    v[j] += g * c;
    g *= coef;
  }
  if (vec && i0 >= out0 && ((((uintptr_t)(yr + i0)) & 15) == 0)) {
#pragma unroll
    for (int q = 0; q < DEEMPH_E / 4; ++q)
      *reinterpret_cast<f32x4*>(yr + i0 + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  } else {
#pragma unroll
    for (int j = 0; j < DEEMPH_E; ++j)
      if (i0 + j >= out0 && i0 + j < L) yr[i0 + j] = v[j];
  }
}

extern "C" int dv3_deemphasis_f32(const float* x, float* y, int32_t B, int32_t L, float coef, void* stream) {
  DV3_REQUIRE(x && y && B > 0 && L > 0, "deemphasis: bad arguments");
  // the chunked form needs the warm-up to swallow the filter's memory; a coefficient too close to 1 (or >= 1) runs
  // the serial-per-row form, which is in place
  const float a = fabsf(coef);
  const bool chunked = a < 1.f && (a == 0.f || DEEMPH_W * logf(a) <= logf(1e-9f * (1.f - a))) && x != y;
  if (chunked) {
    hipLaunchKernelGGL(deemphasis_chunk_kernel, dim3(dv3_cdiv(L, DEEMPH_CH), B), dim3(256), 0, (hipStream_t)stream, x, y, L, coef);
    return dv3_check_launch("deemphasis");
  }
  if (x != y) {
    hipError_t e = hipMemcpyAsync(y, x, (size_t)B * L * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) {
      dv3_set_error("deemphasis: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
  }
  hipLaunchKernelGGL(deemphasis_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, y, L, coef);
  return dv3_check_launch("deemphasis");
}
