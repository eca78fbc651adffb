// Autoregressive decode step kernels (reference: Decoder.incremental_forward, deepvoice3_pytorch/deepvoice3.py:397-473;
// conv.Conv1d.incremental_forward, conv.py:17-46; AttentionLayer.forward with last_attended, deepvoice3.py:132-176).
//
// A decoder step at synthesis time is ~20 dependent, tiny problems (64 columns): the time is kernel-boundary latency,
// not arithmetic.  Two fused kernels replace the ~100 launches of the module-by-module path:
//   conv_step   one incremental conv layer: ring-buffer update (no window shifting: a device step counter t picks the
//               slot t mod L), the k-tap GEMV over the window, and the whole layer tail -- bias, speaker bias,
//               GLU / highway gate, ReLU / sigmoid, up to two sqrt(.5) residuals, the step's position encoding -- plus
//               the optional sigmoid copy and the write into the stacked per-step output, in ONE launch;
//   attn_step   one attention read: scores over the monotonic window [last-1, last+3) (or all keys), softmax,
//               context, the argmax of batch item 0 for the next step's window, the stacked alignment.
// Both read the step counter from device memory, so a captured hipGraph of one step replays for every step.
// Arithmetic: plain fp32 FMA chains (exact fp32; a 64-column problem is not matrix-core work).
#include "common.h"
#include <math.h>

namespace {

constexpr int NB = 4;   // batch items per workgroup

// developer stamps (dv3_decode_program.reserved & 64): s_memtime of workgroup 0 / thread 0 at the phase boundaries of
// every entry of one step; read back with dv3_debug_read(3, ...)
constexpr int DEC_STAMPS = 8, DEC_STAMP_ENTRIES = 40;
__device__ unsigned long long g_dec_stamps[DEC_STAMP_ENTRIES * DEC_STAMPS];
__device__ __forceinline__ void dec_stamp(unsigned long long* st, int k) {
  if (st && threadIdx.x == 0) st[k] = __builtin_readcyclecounter();
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Workgroup = MT output channels x NB batch items.  What bounds a layer at 64 columns is neither bandwidth nor FLOPs
// but the INSTRUCTIONS a wave issues between its few memory round trips (measured with the phase stamps below: halving
// the weight bytes changed nothing, every restructuring of the loads changed nothing, ~3000 instructions per wave did),
// so the tile is written for a short, branch-free instruction stream:
//   * weights in step-tile order (dv3_conv_step_pack_f32): [row block][window element, zero-padded to a multiple of
//     64][16 rows], gated layers [16 `a` rows | 16 gate rows] -- one dense block per workgroup, 16-byte requests at
//     compile-time strides, no bounds checks (padding rows are zeros, the LDS window is zero-filled to match);
//   * the 256 threads are 4 row-quads x (gated: a | gate half x 32, plain: 64) slices of the K axis; slice s owns
//     window elements s, s + slices, ... (consecutive LDS rows per wave: conflict-free);
//   * a thread's first KPF weights are requested BEFORE anything else -- by the persistent program even before the
//     barrier that ends the previous layer, since weights depend on no activation -- so that the window staging,
//     the tail's operands and the weights are in flight together;
//   * every predicate is wave-uniform (clamped indices instead of per-lane conditions).
constexpr int MT = 16;
constexpr int KPF = 24;
constexpr int KPAD_Q = 64;                                             // window elements are padded to a multiple of this
constexpr int RED_A = 64 * MT * 2 * NB, RED_B = 16 * MT * 2 * NB;     // floats of the two reduction stages
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool GATED>
struct StepLane {            // this thread's place in a tile
  static constexpr int NKS = GATED ? 32 : 64;       // K slices
  static constexpr int REC = GATED ? 2 * MT : MT;   // floats per window element of a row block
  int mq, ks, half;
  __device__ __forceinline__ StepLane() {
    const int tid = threadIdx.x;
    mq = tid & 3;
    half = GATED ? (tid >> 2) & 1 : 0;
    ks = GATED ? tid >> 3 : tid >> 2;
  }
};

__device__ __forceinline__ int step_kpad(const dv3_conv_step_desc& p) { return (p.J * p.Cin + KPAD_Q - 1) / KPAD_Q * KPAD_Q; }

template <bool GATED>
__device__ __forceinline__ void conv_step_prefetch_t(const dv3_conv_step_desc& p, const int mblk, f32x4 (&w)[KPF]) {
  typedef StepLane<GATED> LN;
  const LN ln;
  const int kpad = step_kpad(p);
  const int nu = kpad / LN::NKS;
  const float* __restrict__ A = p.a + ((int64_t)mblk * kpad + ln.ks) * LN::REC + ln.half * MT + 4 * ln.mq;
#pragma unroll
  for (int u = 0; u < KPF; ++u)
    if (u < nu) w[u] = *reinterpret_cast<const f32x4*>(A + u * (LN::NKS * LN::REC));
}
__device__ __forceinline__ void conv_step_prefetch(const dv3_conv_step_desc& p, const int mblk, f32x4 (&w)[KPF]) {
  if (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY) conv_step_prefetch_t<true>(p, mblk, w);
  else conv_step_prefetch_t<false>(p, mblk, w);
}

// one (output-channel block, batch block) tile of a layer at step t; all 256 threads of the workgroup take part.
// loaded: w already holds conv_step_prefetch(p, mblk) (else the tile requests it first).
template <bool GATED>
__device__ __forceinline__ void conv_step_tile_t(const dv3_conv_step_desc& p, const int t, const int mblk, const int b0,
                                                 float* lds, f32x4 (&w)[KPF], const bool loaded, unsigned long long* st) {
  typedef StepLane<GATED> LN;
  dec_stamp(st, 0);
  if (!loaded) conv_step_prefetch_t<GATED>(p, mblk, w);
  const int tid = threadIdx.x;
  const LN ln;
  const int mq = ln.mq, ks = ln.ks;
  const int Mrows = GATED ? p.Cg : p.M;
  const int Cin = p.Cin, J = p.J, Ktot = J * Cin, B = p.B;
  const int kpad = step_kpad(p);
  const int nu = kpad / LN::NKS;
  const int L = p.L;
  const int slot = L > 0 ? t % L : 0;
  const float* __restrict__ xin = p.x + (int64_t)t * p.x_ts;
  float* Xs = lds;                                  // [kpad][NB]
  float* redA = lds + (size_t)kpad * NB;            // [slice][MT][2*NB]
  float* redB = redA + RED_A;                       // [16][MT][2*NB]

  // the tail's operands, requested now: wave 0's lane (row, item) finishes output (row, b0 + item); lanes past the
  // layer / the batch read a clamped element and store nothing
  const int frow = tid & (MT - 1), fi = (tid >> 4) & (NB - 1);
  const int m = mblk * MT + frow, fb = b0 + fi;
  const bool fok = m < Mrows && fb < B;
  const int mc = min(m, Mrows - 1), fbc = min(fb, B - 1);
  float t_ba = 0.f, t_bg = 0.f, t_spk = 0.f, t_xr = 0.f, t_r = 0.f, t_r2 = 0.f, t_pa = 0.f;
  if (tid < MT * NB) {
    if (p.bias) {
      t_ba = p.bias[mc];
      if (GATED) t_bg = p.bias[p.Cg + mc];
    }
    if (GATED && p.spk) t_spk = p.spk[(int64_t)fbc * p.spk_bs + mc];
    if (GATED && (p.mode == DV3_EPI_HIGHWAY || p.residual)) t_xr = xin[(int64_t)fbc * p.x_bs + mc];
    if (!GATED && p.r) t_r = p.r[(int64_t)fbc * p.r_bs + mc];
    if (p.r2) t_r2 = p.r2[(int64_t)fbc * p.r2_bs + mc];
    if (p.post_add) t_pa = p.post_add[(int64_t)t * p.post_add_ts + (int64_t)fbc * p.post_add_bs + mc];
  }

  // ---- stage the window: tap J-1 is the new frame, tap j the frame (J-1-j)*dil steps back ----
  f32x4* Xs4w = reinterpret_cast<f32x4*>(Xs);
  const bool ring_owner = mblk == 0 && p.ring;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (tid < kpad - Ktot) Xs4w[Ktot + tid] = zero;           // the padding rows meet zero weights: keep them finite
  for (int c = tid; c < Cin; c += 256) {
    for (int j0 = 0; j0 < J; j0 += 4) {
      f32x4 v[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = j0 + jj;
        if (j < J) {                                          // uniform
          const float* src;
          int64_t bs;
          if (j == J - 1) {
            src = xin + c;
            bs = p.x_bs;
          } else {
            int sl = (slot - (J - 1 - j) * p.dil) % L;
            if (sl < 0) sl += L;
            src = p.ring + (int64_t)sl * B * Cin + c;
            bs = Cin;
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) v[jj][nb] = src[(int64_t)min(b0 + nb, B - 1) * bs];    // clamped: rows past the
        }                                                                                       // batch are never stored
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = j0 + jj;
        if (j < J) {
          Xs4w[j * Cin + c] = v[jj];
          // the new frame enters the ring (slot-major [L][B][Cin]); no tap reads this slot ((J-1)*dil < L)
          if (ring_owner && j == J - 1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              if (b0 + nb < B) p.ring[((int64_t)slot * B + b0 + nb) * Cin + c] = v[jj][nb];
          }
        }
      }
    }
  }
  dec_stamp(st, 1);
  __syncthreads();
  dec_stamp(st, 2);

  // ---- GEMV: 4 rows (of the `a` or of the gate half) x NB batch items per thread over its K slice; accumulators as
  // pairs over the batch index: v_pk_fma_f32 does two of the tile's FMAs per lane and issue slot ----
  f32x2 acc[4][NB / 2];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < NB / 2; ++i) acc[r][i] = f32x2{0.f, 0.f};
  const f32x4* Xs4 = reinterpret_cast<const f32x4*>(Xs) + ks;
  auto fma_block = [&](const f32x4& wv, const f32x4& xv) __attribute__((always_inline)) {
    const f32x2 x01 = {xv[0], xv[1]}, x23 = {xv[2], xv[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x2 w2 = {wv[r], wv[r]};
      acc[r][0] = __builtin_elementwise_fma(w2, x01, acc[r][0]);
      acc[r][1] = __builtin_elementwise_fma(w2, x23, acc[r][1]);
    }
  };
#pragma unroll
  for (int u = 0; u < KPF; ++u)
    if (u < nu) fma_block(w[u], Xs4[u * LN::NKS]);
  if (nu > KPF) {                                   // windows longer than the prefetch (k * Cin > 768)
    constexpr int U = 8;
    const float* __restrict__ A = p.a + ((int64_t)mblk * kpad + ks) * LN::REC + ln.half * MT + 4 * mq;
    for (int ub = KPF; ub < nu; ub += U) {
      f32x4 rw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) rw[u] = *reinterpret_cast<const f32x4*>(A + min(ub + u, nu - 1) * (LN::NKS * LN::REC));
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (ub + u < nu) fma_block(rw[u], Xs4[(ub + u) * LN::NKS]);
    }
  }
  dec_stamp(st, 3);
  // ---- reduce the slices in a fixed order: [slice][row][a x NB | g x NB] -> 16 groups -> one lane per output ----
  // (a slice's 4-row block is 8 chunks of 16 bytes [row][a | g]; chunk c is stored at c ^ (slice & 7), which spreads
  //  the 64 lanes of a wave over all 8 chunk positions of the LDS banks for the writes here and the reads below)
  f32x4* redA4 = reinterpret_cast<f32x4*>(redA);
  {
    const int blk = (ks * MT + 4 * mq) * 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 q = {acc[r][0][0], acc[r][0][1], acc[r][1][0], acc[r][1][1]};
      redA4[blk + ((2 * r + ln.half) ^ (ks & 7))] = q;
    }
  }
  __syncthreads();
  {
    constexpr int PER = LN::NKS / 16;
    const int row = tid & (MT - 1), part = tid >> 4;        // 16 parts x PER slices
    f32x4 va[PER], vg[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int sl = part * PER + q;
      const int blk = (sl * MT + (row & ~3)) * 2;
      va[q] = redA4[blk + ((2 * (row & 3)) ^ (sl & 7))];
      vg[q] = GATED ? redA4[blk + ((2 * (row & 3) + 1) ^ (sl & 7))] : zero;
    }
    f32x4 sa = zero, sg = zero;
#pragma unroll
    for (int q = 0; q < PER; ++q) { sa += va[q]; sg += vg[q]; }
    f32x4* wr = reinterpret_cast<f32x4*>(redB + ((size_t)part * MT + row) * (2 * NB));
    wr[0] = sa;
    wr[1] = sg;
  }
  __syncthreads();
  dec_stamp(st, 4);
  if (tid < MT * NB) {
    // one lane per (row, batch item): its 16 partial sums, then the layer tail
    float pa[16], pg[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float* o = redB + ((size_t)q * MT + frow) * (2 * NB);
      pa[q] = o[fi];
      pg[q] = GATED ? o[NB + fi] : 0.f;
    }
    float sa = 0.f, sg = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { sa += pa[q]; sg += pg[q]; }
    const float rs2 = 0.70710678118654752440f;
    float y;
    if (GATED) {
      float a = sa + t_ba;
      const float g = sg + t_bg;
      if (p.spk) a += t_spk;
      const float sgm = sigmoidf_(g);
      if (p.mode == DV3_EPI_GLU) y = p.residual ? (a * sgm + t_xr) * rs2 : a * sgm;
      else y = sgm * a + (1.0f - sgm) * t_xr;
      if (p.r2) y = (y + t_r2) * rs2;
    } else {
      float v = sa + t_ba;
      if (p.mode == DV3_EPI_RELU) v = fmaxf(v, 0.f);
      else if (p.mode == DV3_EPI_SIGMOID) v = sigmoidf_(v);
      else if (p.mode == DV3_EPI_SOFTSIGN) v = v / (1.0f + fabsf(v));
      if (p.r) v = (v + t_r) * rs2;
      if (p.r2) v = (v + t_r2) * rs2;
      y = v;
    }
    if (fok) {
      if (p.y_pre) p.y_pre[(int64_t)fb * p.y_pre_bs + m] = y;
      if (p.post_add) y += t_pa;
      p.y[(int64_t)fb * p.y_bs + m] = y;
      float o = y;
      if (p.y_act) {
        o = sigmoidf_(y);
        p.y_act[(int64_t)fb * p.y_act_bs + m] = o;
      }
      if (p.out_seq) p.out_seq[(int64_t)t * p.out_seq_ts + (int64_t)fb * p.out_seq_bs + m] = o;
    }
  }
  dec_stamp(st, 5);
}

__device__ __forceinline__ void conv_step_tile(const dv3_conv_step_desc& p, const int t, const int mblk, const int b0,
                                               float* lds, f32x4 (&w)[KPF], const bool loaded,
                                               unsigned long long* st = nullptr) {
  if (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY) conv_step_tile_t<true>(p, t, mblk, b0, lds, w, loaded, st);
  else conv_step_tile_t<false>(p, t, mblk, b0, lds, w, loaded, st);
}

__global__ __launch_bounds__(256) void conv_step_kernel(const dv3_conv_step_desc p) {
  extern __shared__ float lds[];
  f32x4 w[KPF];
  conv_step_tile(p, p.t ? p.t[0] : p.t_value, blockIdx.x, blockIdx.y * NB, lds, w, false);
}

// one attention read of batch item b at step t; all 256 threads of the workgroup take part
__device__ __forceinline__ void attn_step_item(const dv3_attn_step_desc& p, const int t, const int b, float* lds,
                                               float* red, int* redi) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int E = p.E, Tk = p.Tk;
  float* q = lds;
  float* sc = lds + E;
  int lo = 0, hi = Tk;
  if (p.last_attended) {       // deepvoice3.py:150-156
    const int la = p.last_attended[t & 1];
    const int back = la - p.win_back, ahead = la + p.win_ahead;
    if (back > 0) lo = back;
    if (ahead < Tk) hi = ahead;
  }
  for (int e = tid; e < E; e += 256) q[e] = p.q[(int64_t)b * p.q_bs + e];
  __syncthreads();
  // scores: a wave takes 4 consecutive keys at a time, its lanes split the E axis (16 independent loads per lane in
  // flight, one round trip for a 4-key monotonic window); with (B, Tk, E) keys a row is 4 contiguous 256-byte reads
  const int es = p.kv_tke ? 1 : Tk, ns = p.kv_tke ? E : 1;
  const float* __restrict__ kb = p.k + (int64_t)b * E * Tk;
  for (int n0 = lo + 4 * wave; n0 < hi; n0 += 16) {
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int e = lane; e < E; e += 64) {
      const float qe = q[e];
#pragma unroll
      for (int i = 0; i < 4; ++i) part[i] = fmaf(qe, kb[(int64_t)min(n0 + i, hi - 1) * ns + (int64_t)e * es], part[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s = dv3_wave_sum(part[i]);
      if (lane == 0 && n0 + i < hi) sc[n0 + i] = s;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int n = tid; n < Tk; n += 256)
    if (n >= lo && n < hi) mx = fmaxf(mx, sc[n]);
  mx = dv3_wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int n = tid; n < Tk; n += 256) {
    const float e = (n >= lo && n < hi) ? expf(sc[n] - mx) : 0.f;
    sc[n] = e;
    sum += e;
  }
  sum = dv3_wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  __syncthreads();
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int n = tid; n < Tk; n += 256) {
    const float pr = sc[n] * inv;
    sc[n] = pr;
    if (p.attn) p.attn[(int64_t)b * Tk + n] = pr;
    if (p.attn_seq) p.attn_seq[(int64_t)t * p.attn_seq_ts + (int64_t)b * Tk + n] = pr;
    if (pr > best) { best = pr; bi = n; }
  }
  __syncthreads();
  // context (deepvoice3.py:167-171): sum_n p[n] v[e][n] * (Tk * sqrt(1/Tk))
  const float scale = (float)Tk * sqrtf(1.0f / (float)Tk);
  const float* __restrict__ vb = p.v + (int64_t)b * E * Tk;
  for (int e = tid; e < E; e += 256) {
    float c = 0.f;
    int n = lo;
    for (; n + 4 <= hi; n += 4) {         // loads first: one round trip per 4 keys
      float vv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) vv[i] = vb[(int64_t)(n + i) * ns + (int64_t)e * es];
#pragma unroll
      for (int i = 0; i < 4; ++i) c = fmaf(sc[n + i], vv[i], c);
    }
    for (; n < hi; ++n) c = fmaf(sc[n], vb[(int64_t)n * ns + (int64_t)e * es], c);
    p.ctx[(int64_t)b * p.ctx_bs + e] = c * scale;
  }
  // next step's window: argmax of batch item 0 (deepvoice3.py:445), first maximum
  if (b == 0 && p.last_attended) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { red[wave] = best; redi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (red[w] > best || (red[w] == best && redi[w] < bi)) { best = red[w]; bi = redi[w]; }
      p.last_attended[(t + 1) & 1] = bi;
    }
  }
}

__global__ __launch_bounds__(256) void attn_step_kernel(const dv3_attn_step_desc p) {
  extern __shared__ float lds[];   // q [E] | scores / probabilities [Tk]
  __shared__ float red[4];
  __shared__ int redi[4];
  attn_step_item(p, p.t ? p.t[0] : p.t_value, blockIdx.x, lds, red, redi);
}

// fwd_pack [Ktot][lda] -> step-tile order, window elements zero-padded to kpad (include/dv3hip.h: dv3_conv_step_pack_f32)
__global__ __launch_bounds__(256) void conv_step_pack_kernel(const float* __restrict__ src, const int lda, const int a_half,
                                                             const int Ktot, const int kpad, const int rows, const int gated,
                                                             float* __restrict__ out, const int64_t n) {
  const int rec = gated ? 2 * MT : MT;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int e = (int)(idx % rec);
    const int64_t q = idx / rec;
    const int kk = (int)(q % kpad), blk = (int)(q / kpad);
    const int half = e / MT, r = blk * MT + (e % MT);
    out[idx] = (r < rows && kk < Ktot) ? src[(int64_t)kk * lda + (half ? a_half : 0) + r] : 0.f;
  }
}

// ---- the whole decoder loop as one persistent launch (include/dv3hip.h: dv3_decode_program_run) ----
constexpr int SYNC_STRIDE = 32;             // ints between two barrier counters (128 bytes: one counter per line)
constexpr unsigned SPIN_LIMIT = 1u << 21;   // x ~1 us per probe: a barrier that takes seconds is a lost workgroup

// Monotonic-counter barrier among `n` workgroups: arrival k of a workgroup waits until the counter reaches k * n.
// Release / acquire at agent scope: the activations a layer hands to the next one cross CUs (and XCD L2s).  The spin
// itself is a relaxed device-scope load (sc1: served past the XCD's L2); the ONE acquire fence after it drops the
// stale lines -- an acquire load per spin would invalidate the L2 the other workgroups are streaming weights through.
__device__ __forceinline__ bool wg_barrier(int32_t* ctr, const unsigned n, unsigned& target, const int flags) {
  __shared__ int ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    target += n;
    if (!(flags & 1) && !(flags & 8)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int good = 1;
    while ((unsigned)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > SPIN_LIMIT) { good = 0; break; }
    }
    // one wave's acquire serves the workgroup: the invalidate acts on the CU's L1 and on the XCD's L2, not on the wave
    if (!(flags & 1) && !(flags & 16)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok = good;
  }
  __syncthreads();
  if (!(flags & 1) && (flags & 16)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok != 0;
}

__global__ __launch_bounds__(256) void decode_program_kernel(const dv3_decode_program prog, const int groups, const int P) {
  extern __shared__ float lds[];
  __shared__ float red[4];
  __shared__ int redi[4];
  __shared__ int stop_s;
  // blockIdx -> (batch group, member): consecutive ids sit on different XCDs (id % 8), so a group takes ids of one
  // residue class when the group count allows it -- its barrier counter and activations stay in one L2
  int g, i;
  if ((P & 7) == 0) {
    // members i = xcd (mod 8): an XCD works on the SAME output-channel blocks for every batch group, so each weight
    // row enters exactly one L2 (the launch-per-layer grids have this property through their (mblk, bblk) order)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = P >> 3;
    i = xcd + 8 * (slot % per);
    g = slot / per;
  } else {
    g = blockIdx.x / P;
    i = blockIdx.x % P;
  }
  int32_t* all_ctr = prog.sync;
  int32_t* grp_ctr = prog.sync + SYNC_STRIDE * (1 + g);
  unsigned all_target = 0, grp_target = 0;
  const int b0 = g * NB;
  int steps = 0;
  bool alive = true;
  f32x4 w[KPF];
  bool loaded = false;
  auto tiles_of = [](const dv3_conv_step_desc& p) {
    const bool gated = p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY;
    return ((gated ? p.Cg : p.M) + MT - 1) / MT;
  };
  if (prog.entries[0].kind == 0 && i < tiles_of(prog.entries[0].conv)) {
    conv_step_prefetch(prog.entries[0].conv, i, w);
    loaded = true;
  }
  for (int s = 0; s < prog.n_steps && alive; ++s) {
    const int t = prog.t0 + s;
    for (int e = 0; e < prog.n_entries && alive; ++e) {
      const dv3_decode_entry& en = prog.entries[e];
      if (en.kind == 0) {
        const dv3_conv_step_desc& p = en.conv;
        const int tiles = tiles_of(p);
        for (int mblk = i; mblk < tiles; mblk += P) {
          unsigned long long* st = ((prog.reserved & 64) && blockIdx.x == 0 && s == 8 && e < DEC_STAMP_ENTRIES)
                                       ? g_dec_stamps + e * DEC_STAMPS : nullptr;
          conv_step_tile(p, t, mblk, b0, lds, w, loaded && mblk == i, st);
          __syncthreads();
        }
      } else {
        const int b = b0 + i;
        if (i < NB && b < prog.B) attn_step_item(en.attn, t, b, lds, red, redi);
      }
      // the next layer's weights (this workgroup's first tile of it) are requested before the barrier is waited on
      {
        const dv3_decode_entry& nx = prog.entries[e + 1 < prog.n_entries ? e + 1 : 0];
        loaded = false;
        if (!(prog.reserved & 32) && nx.kind == 0 && i < tiles_of(nx.conv)) {
          conv_step_prefetch(nx.conv, i, w);
          loaded = true;
        }
      }
      if ((prog.reserved & 64) && blockIdx.x == 0 && s == 8 && e < DEC_STAMP_ENTRIES) dec_stamp(g_dec_stamps + e * DEC_STAMPS, 6);
      alive = (prog.reserved & 2) ? true : wg_barrier(grp_ctr, (unsigned)P, grp_target, prog.reserved);
      if ((prog.reserved & 64) && blockIdx.x == 0 && s == 8 && e < DEC_STAMP_ENTRIES) dec_stamp(g_dec_stamps + e * DEC_STAMPS, 7);
    }
    if (!alive) break;
    alive = (prog.reserved & 4) ? true : wg_barrier(all_ctr, (unsigned)(groups * P), all_target, prog.reserved);
    if (!alive) break;
    steps = s + 1;
    // the reference's stop rule (deepvoice3.py:463-470), evaluated identically by every workgroup
    const int done_steps = t + 1;
    if (threadIdx.x == 0) {
      int stop = 0;
      if (prog.done_seq && done_steps > prog.min_steps) {
        stop = 1;
        for (int b = 0; b < prog.B; ++b)
          if (!(prog.done_seq[(int64_t)t * prog.done_ts + b] > 0.5f)) { stop = 0; break; }
      }
      if (!stop && prog.done_seq && done_steps > prog.max_steps) stop = 1;
      stop_s = stop;
    }
    __syncthreads();
    const int stop = stop_s;
    __syncthreads();
    if (stop) break;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) prog.steps_out[0] = alive ? steps : -1;
}

}  // namespace

static int conv_step_check(const dv3_conv_step_desc* d, bool need_t, size_t* lds_out);
static int attn_step_check(const dv3_attn_step_desc* d, bool need_t, size_t* lds_out);

extern "C" int dv3_conv_step_f32(const dv3_conv_step_desc* d, void* stream) {
  size_t lds = 0;
  const int rc = conv_step_check(d, true, &lds);
  if (rc != DV3_OK) return rc;
  const bool gated = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  const int rows = gated ? d->Cg : d->M;
  hipLaunchKernelGGL(conv_step_kernel, dim3(dv3_cdiv(rows, MT), dv3_cdiv(d->B, NB)), dim3(256), lds, (hipStream_t)stream, *d);
  return dv3_check_launch("conv_step");
}

static int conv_step_check(const dv3_conv_step_desc* d, bool need_t, size_t* lds_out) {
  DV3_REQUIRE(d && d->x && d->a && d->y, "conv_step: null pointer");
  DV3_REQUIRE(d->B > 0 && d->Cin > 0 && d->M > 0 && d->J >= 1 && d->dil >= 1, "conv_step: bad dims");
  const bool gated = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  if (gated) {
    DV3_REQUIRE(d->M == 2 * d->Cg, "conv_step: bad gated layout");
    DV3_REQUIRE(!(d->mode == DV3_EPI_HIGHWAY || d->residual) || d->Cin == d->Cg, "conv_step: residual needs Cin == Cout");
  } else {
    DV3_REQUIRE(d->mode >= DV3_EPI_LINEAR && d->mode <= DV3_EPI_SOFTSIGN && d->mode != DV3_EPI_DGRAD, "conv_step: bad mode");
  }
  if (d->J > 1) DV3_REQUIRE(d->ring && (d->t || !need_t) && d->L >= (d->J - 1) * d->dil + 1,
                            "conv_step: k > 1 needs ring, t and L >= (k-1)*d+1");
  if (d->post_add || d->out_seq || d->x_ts) DV3_REQUIRE(d->t || !need_t, "conv_step: post_add / out_seq / x_ts need the step counter");
  const size_t lds = (size_t)dv3_conv_step_lds_bytes(d->J, d->Cin);
  DV3_REQUIRE(((uintptr_t)d->a & 15) == 0, "conv_step: the step-tile weight image must be 16-byte aligned");
  DV3_REQUIRE(lds <= DV3_CONV_STEP_LDS_MAX, "conv_step: window too large for LDS (%zu bytes)", lds);
  *lds_out = lds;
  return DV3_OK;
}

extern "C" int dv3_conv_step_lds_bytes(int32_t J, int32_t Cin) {
  if (J <= 0 || Cin <= 0 || (int64_t)J * Cin > (1 << 24)) return 1 << 30;
  return (int)(((size_t)dv3_cdiv(J * Cin, KPAD_Q) * KPAD_Q * NB + RED_A + RED_B) * sizeof(float));
}

extern "C" int dv3_conv_step_pack_floats(int32_t Ktot, int32_t M, int32_t Cg) {
  if (Ktot <= 0 || M <= 0 || Cg < 0) return 0;
  const int rows = Cg > 0 ? Cg : M;
  return dv3_cdiv(rows, MT) * (dv3_cdiv(Ktot, KPAD_Q) * KPAD_Q) * (Cg > 0 ? 2 * MT : MT);
}

extern "C" int dv3_conv_step_pack_f32(const float* fwd_pack, int32_t lda, int32_t a_half, int32_t Ktot, int32_t M,
                                      int32_t Cg, float* out, void* stream) {
  DV3_REQUIRE(fwd_pack && out && Ktot > 0 && M > 0 && Cg >= 0, "conv_step_pack: bad arguments");
  DV3_REQUIRE(Cg == 0 ? lda >= M : (M == 2 * Cg && a_half >= Cg && lda >= a_half + Cg), "conv_step_pack: bad fwd_pack layout");
  const int64_t n = dv3_conv_step_pack_floats(Ktot, M, Cg);
  const int rows = Cg > 0 ? Cg : M;
  hipLaunchKernelGGL(conv_step_pack_kernel, dim3((unsigned)(dv3_cdiv64(n, 256) < 2048 ? dv3_cdiv64(n, 256) : 2048)), dim3(256), 0,
                     (hipStream_t)stream, fwd_pack, lda, a_half, Ktot, dv3_cdiv(Ktot, KPAD_Q) * KPAD_Q, rows, Cg > 0 ? 1 : 0, out, n);
  return dv3_check_launch("conv_step_pack");
}

int dv3_decode_read_stamps(void* dst, int64_t bytes) {
  if (bytes <= 0 || bytes > (int64_t)sizeof(g_dec_stamps)) return DV3_EINVAL;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dec_stamps), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? DV3_OK : DV3_ELAUNCH;
}

static int decode_groups(int B) { return dv3_cdiv(B, NB); }
static int decode_wg_per_group(int B, int want) {
  const int groups = decode_groups(B);
  int P = want > 0 ? want : 16;
  while (P > 1 && groups * P > 256) P >>= 1;     // one workgroup per CU: the grid has to be co-resident
  return P;
}

extern "C" int dv3_decode_program_sync_ints(int32_t B) { return B > 0 ? SYNC_STRIDE * (1 + decode_groups(B)) : 0; }

// every entry is validated like the single-launch entry points, on the host copy the caller uploaded
extern "C" int dv3_decode_program_run(const dv3_decode_program* d, void* stream) {
  DV3_REQUIRE(d && d->entries && d->entries_host && d->sync && d->steps_out, "decode_program: null pointer");
  DV3_REQUIRE(d->n_entries > 0 && d->B > 0 && d->n_steps > 0 && d->t0 >= 0, "decode_program: bad dims");
  size_t lds_max = 0;
  for (int e = 0; e < d->n_entries; ++e) {
    const dv3_decode_entry& en = d->entries_host[e];
    size_t l = 0;
    DV3_REQUIRE(en.kind == 0 || en.kind == 1, "decode_program: entry %d has kind %d", e, en.kind);
    const int rc = en.kind == 0 ? conv_step_check(&en.conv, false, &l) : attn_step_check(&en.attn, false, &l);
    if (rc != DV3_OK) return rc;
    DV3_REQUIRE((en.kind == 0 ? en.conv.B : en.attn.B) == d->B, "decode_program: entry %d has another batch size", e);
    if (l > lds_max) lds_max = l;
  }
  const int groups = decode_groups(d->B);
  DV3_REQUIRE(groups <= 256, "decode_program: batch %d needs more than 256 co-resident workgroups", d->B);
  const int P = decode_wg_per_group(d->B, d->wg_per_group);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d->sync, 0, sizeof(int32_t) * SYNC_STRIDE * (1 + groups), st);
  if (e != hipSuccess) {
    dv3_set_error("decode_program: %s", hipGetErrorString(e));
    return DV3_ELAUNCH;
  }
  const size_t lds = lds_max;
  hipLaunchKernelGGL(decode_program_kernel, dim3(groups * P), dim3(256), lds, st, *d, groups, P);
  return dv3_check_launch("decode_program");
}

extern "C" int dv3_decode_program_launch(const dv3_decode_program* d, void* stream) {
  DV3_REQUIRE(d && d->entries_host, "decode_program_launch: null pointer");
  DV3_REQUIRE(d->n_entries > 0 && d->B > 0 && d->n_steps > 0 && d->t0 >= 0, "decode_program_launch: bad dims");
  hipStream_t st = (hipStream_t)stream;
  // validate once; per launch only the step index changes
  size_t lds[256];
  DV3_REQUIRE(d->n_entries <= 256, "decode_program_launch: more than 256 entries");
  for (int e = 0; e < d->n_entries; ++e) {
    const dv3_decode_entry& en = d->entries_host[e];
    DV3_REQUIRE(en.kind == 0 || en.kind == 1, "decode_program_launch: entry %d has kind %d", e, en.kind);
    const int rc = en.kind == 0 ? conv_step_check(&en.conv, false, &lds[e]) : attn_step_check(&en.attn, false, &lds[e]);
    if (rc != DV3_OK) return rc;
  }
  for (int s = 0; s < d->n_steps; ++s) {
    const int t = d->t0 + s;
    for (int e = 0; e < d->n_entries; ++e) {
      const dv3_decode_entry& en = d->entries_host[e];
      if (en.kind == 0) {
        dv3_conv_step_desc c = en.conv;
        c.t = nullptr;
        c.t_value = t;
        const bool gated = c.mode == DV3_EPI_GLU || c.mode == DV3_EPI_HIGHWAY;
        hipLaunchKernelGGL(conv_step_kernel, dim3(dv3_cdiv(gated ? c.Cg : c.M, MT), dv3_cdiv(c.B, NB)), dim3(256), lds[e], st, c);
      } else {
        dv3_attn_step_desc a = en.attn;
        a.t = nullptr;
        a.t_value = t;
        hipLaunchKernelGGL(attn_step_kernel, dim3(a.B), dim3(256), lds[e], st, a);
      }
    }
  }
  return dv3_check_launch("decode_program_launch");
}

static int attn_step_check(const dv3_attn_step_desc* d, bool need_t, size_t* lds_out) {
  DV3_REQUIRE(d && d->q && d->k && d->v && d->ctx, "attn_step: null pointer");
  DV3_REQUIRE(d->B > 0 && d->E > 0 && d->Tk > 0, "attn_step: bad dims");
  if (d->last_attended || d->attn_seq) DV3_REQUIRE(d->t || !need_t, "attn_step: the window / stacked output need the step counter");
  const size_t lds = ((size_t)d->E + d->Tk) * sizeof(float);
  DV3_REQUIRE(lds <= 64 * 1024, "attn_step: E + Tk too large for LDS");
  *lds_out = lds;
  return DV3_OK;
}

extern "C" int dv3_attn_step_f32(const dv3_attn_step_desc* d, void* stream) {
  size_t lds = 0;
  const int rc = attn_step_check(d, true, &lds);
  if (rc != DV3_OK) return rc;
  hipLaunchKernelGGL(attn_step_kernel, dim3(d->B), dim3(256), lds, (hipStream_t)stream, *d);
  return dv3_check_launch("attn_step");
}
