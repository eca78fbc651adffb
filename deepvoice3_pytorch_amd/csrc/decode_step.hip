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

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Workgroup = MT output channels x NB batch items; its 256 threads are MT channel lanes x KS slices of the K axis
// (k-tap window x input channels), so a whole layer's weights are streamed by 16 x (B / NB) workgroups with 16 loads in
// flight per thread: the layer is a latency problem (1.5 MB of weights, 64 columns), not a bandwidth or FLOP one.
constexpr int MT = 16, KS = 16;
// one (output-channel block, batch block) tile of a layer at step t; all 256 threads of the workgroup take part
__device__ __forceinline__ void conv_step_tile(const dv3_conv_step_desc& p, const int t, const int mblk, const int b0,
                                               float* lds) {
  const int tid = threadIdx.x, ml = tid & (MT - 1), ks = tid / MT;
  const bool gated = p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY;
  const int Mrows = gated ? p.Cg : p.M;
  const int Cin = p.Cin, J = p.J, Ktot = J * Cin, B = p.B;
  const int L = p.L;
  const int slot = L > 0 ? t % L : 0;
  const float* __restrict__ xin = p.x + (int64_t)t * p.x_ts;
  float* Xs = lds;                          // [Ktot][NB]
  float* red = lds + (size_t)Ktot * NB;     // [KS][MT][2*NB]

  // ---- stage the window: tap J-1 is the new frame, tap j the frame (J-1-j)*dil steps back ----
  for (int idx = tid; idx < Ktot * NB; idx += 256) {
    const int nb = idx / Ktot, kk = idx - nb * Ktot;      // consecutive threads: consecutive channels (coalesced)
    const int j = kk / Cin, c = kk - j * Cin;
    const int b = b0 + nb;
    float v = 0.f;
    if (b < B) {
      if (j == J - 1) {
        v = xin[(int64_t)b * p.x_bs + c];
      } else {
        int s = slot - (J - 1 - j) * p.dil;
        s %= L;
        if (s < 0) s += L;
        v = p.ring[((int64_t)s * B + b) * Cin + c];
      }
    }
    Xs[kk * NB + nb] = v;
  }
  // the new frame enters the ring (slot-major [L][B][Cin]); the taps above never read this slot ((J-1)*dil < L)
  if (mblk == 0 && p.ring) {
    for (int idx = tid; idx < NB * Cin; idx += 256) {
      const int nb = idx / Cin, c = idx - nb * Cin;
      const int b = b0 + nb;
      if (b < B) p.ring[((int64_t)slot * B + b) * Cin + c] = xin[(int64_t)b * p.x_bs + c];
    }
  }
  __syncthreads();

  // ---- GEMV ----
  const int m = mblk * MT + ml;
  const bool mok = m < Mrows;
  const float* __restrict__ Aa = p.a + (mok ? m : 0);
  const float* __restrict__ Ag = p.a + p.a_half + (mok ? m : 0);
  float acc_a[NB], acc_g[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc_a[i] = acc_g[i] = 0.f;
  const int kq = (Ktot + KS - 1) / KS;
  const int k0 = ks * kq, k1 = min(Ktot, k0 + kq);
  const f32x4* Xs4 = reinterpret_cast<const f32x4*>(Xs);
  if (gated) {
#pragma unroll 8
    for (int kk = k0; kk < k1; ++kk) {
      const float wa = Aa[(int64_t)kk * p.lda];
      const float wg = Ag[(int64_t)kk * p.lda];
      const f32x4 xv = Xs4[kk];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        acc_a[i] = fmaf(wa, xv[i], acc_a[i]);
        acc_g[i] = fmaf(wg, xv[i], acc_g[i]);
      }
    }
  } else {
#pragma unroll 8
    for (int kk = k0; kk < k1; ++kk) {
      const float wa = Aa[(int64_t)kk * p.lda];
      const f32x4 xv = Xs4[kk];
#pragma unroll
      for (int i = 0; i < NB; ++i) acc_a[i] = fmaf(wa, xv[i], acc_a[i]);
    }
  }
  float* my = red + ((size_t)ks * MT + ml) * (2 * NB);
#pragma unroll
  for (int i = 0; i < NB; ++i) { my[i] = acc_a[i]; my[NB + i] = acc_g[i]; }
  __syncthreads();
  if (ks == 0 && mok) {
#pragma unroll
    for (int i = 0; i < NB; ++i) { acc_a[i] = 0.f; acc_g[i] = 0.f; }
    for (int q = 0; q < KS; ++q) {          // fixed order: deterministic
      const float* o = red + ((size_t)q * MT + ml) * (2 * NB);
#pragma unroll
      for (int i = 0; i < NB; ++i) { acc_a[i] += o[i]; acc_g[i] += o[NB + i]; }
    }

    // ---- the layer tail ----
    const float rs2 = 0.70710678118654752440f;
    const float ba = p.bias ? p.bias[m] : 0.f;
    const float bg = (gated && p.bias) ? p.bias[p.Cg + m] : 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int b = b0 + i;
      if (b >= B) continue;
      float y;
      if (gated) {
        float a = acc_a[i] + ba;
        const float g = acc_g[i] + bg;
        if (p.spk) a += p.spk[(int64_t)b * p.spk_bs + m];
        const float s = sigmoidf_(g);
        const float xr = (p.mode == DV3_EPI_HIGHWAY || p.residual) ? xin[(int64_t)b * p.x_bs + m] : 0.f;
        if (p.mode == DV3_EPI_GLU) y = p.residual ? (a * s + xr) * rs2 : a * s;
        else y = s * a + (1.0f - s) * xr;
        if (p.r2) y = (y + p.r2[(int64_t)b * p.r2_bs + m]) * rs2;
      } else {
        float v = acc_a[i] + ba;
        if (p.mode == DV3_EPI_RELU) v = fmaxf(v, 0.f);
        else if (p.mode == DV3_EPI_SIGMOID) v = sigmoidf_(v);
        else if (p.mode == DV3_EPI_SOFTSIGN) v = v / (1.0f + fabsf(v));
        if (p.r) v = (v + p.r[(int64_t)b * p.r_bs + m]) * rs2;
        if (p.r2) v = (v + p.r2[(int64_t)b * p.r2_bs + m]) * rs2;
        y = v;
      }
      if (p.y_pre) p.y_pre[(int64_t)b * p.y_pre_bs + m] = y;
      if (p.post_add) y += p.post_add[(int64_t)t * p.post_add_ts + (int64_t)b * p.post_add_bs + m];
      p.y[(int64_t)b * p.y_bs + m] = y;
      float o = y;
      if (p.y_act) {
        o = sigmoidf_(y);
        p.y_act[(int64_t)b * p.y_act_bs + m] = o;
      }
      if (p.out_seq) p.out_seq[(int64_t)t * p.out_seq_ts + (int64_t)b * p.out_seq_bs + m] = o;
    }
  }
}

__global__ __launch_bounds__(256) void conv_step_kernel(const dv3_conv_step_desc p) {
  extern __shared__ float lds[];
  conv_step_tile(p, p.t ? p.t[0] : 0, blockIdx.x, blockIdx.y * NB, lds);
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
  const float* __restrict__ kb = p.k + (int64_t)b * E * Tk;
  float mx = -INFINITY;
  for (int n = tid; n < Tk; n += 256) {
    float s = -INFINITY;
    if (n >= lo && n < hi) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int e = 0;
      for (; e + 4 <= E; e += 4) {
        a0 = fmaf(q[e], kb[(int64_t)e * Tk + n], a0);
        a1 = fmaf(q[e + 1], kb[(int64_t)(e + 1) * Tk + n], a1);
        a2 = fmaf(q[e + 2], kb[(int64_t)(e + 2) * Tk + n], a2);
        a3 = fmaf(q[e + 3], kb[(int64_t)(e + 3) * Tk + n], a3);
      }
      for (; e < E; ++e) a0 = fmaf(q[e], kb[(int64_t)e * Tk + n], a0);
      s = (a0 + a1) + (a2 + a3);
    }
    sc[n] = s;
    mx = fmaxf(mx, s);
  }
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
    for (int n = lo; n < hi; ++n) c = fmaf(sc[n], vb[(int64_t)e * Tk + n], c);
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
  attn_step_item(p, p.t ? p.t[0] : 0, blockIdx.x, lds, red, redi);
}

// ---- the whole decoder loop as one persistent launch (include/dv3hip.h: dv3_decode_program_run) ----
constexpr int SYNC_STRIDE = 32;             // ints between two barrier counters (128 bytes: one counter per line)
constexpr unsigned SPIN_LIMIT = 1u << 21;   // x ~1 us per probe: a barrier that takes seconds is a lost workgroup

// Monotonic-counter barrier among `n` workgroups: arrival k of a workgroup waits until the counter reaches k * n.
// Release / acquire at agent scope: the activations a layer hands to the next one cross CUs (and XCD L2s).  The spin
// itself is a relaxed device-scope load (sc1: served past the XCD's L2); the ONE acquire fence after it drops the
// stale lines -- an acquire load per spin would invalidate the L2 the other workgroups are streaming weights through.
__device__ __forceinline__ bool wg_barrier(int32_t* ctr, const unsigned n, unsigned& target) {
  __shared__ int ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    target += n;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int good = 1;
    while ((unsigned)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > SPIN_LIMIT) { good = 0; break; }
    }
    ok = good;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
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
  if ((groups & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    g = xcd + 8 * (slot / P);
    i = slot % P;
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
  for (int s = 0; s < prog.n_steps && alive; ++s) {
    const int t = prog.t0 + s;
    for (int e = 0; e < prog.n_entries && alive; ++e) {
      const dv3_decode_entry& en = prog.entries[e];
      if (en.kind == 0) {
        const dv3_conv_step_desc& p = en.conv;
        const bool gated = p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY;
        const int tiles = ((gated ? p.Cg : p.M) + MT - 1) / MT;
        for (int mblk = i; mblk < tiles; mblk += P) {
          conv_step_tile(p, t, mblk, b0, lds);
          __syncthreads();
        }
      } else {
        const int b = b0 + i;
        if (i < NB && b < prog.B) attn_step_item(en.attn, t, b, lds, red, redi);
      }
      alive = wg_barrier(grp_ctr, (unsigned)P, grp_target);
    }
    if (!alive) break;
    alive = wg_barrier(all_ctr, (unsigned)(groups * P), all_target);
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
    DV3_REQUIRE(d->M == 2 * d->Cg && d->a_half >= d->Cg && d->lda >= d->a_half + d->Cg, "conv_step: bad gated layout");
    DV3_REQUIRE(!(d->mode == DV3_EPI_HIGHWAY || d->residual) || d->Cin == d->Cg, "conv_step: residual needs Cin == Cout");
  } else {
    DV3_REQUIRE(d->lda >= d->M && d->mode >= DV3_EPI_LINEAR && d->mode <= DV3_EPI_SOFTSIGN && d->mode != DV3_EPI_DGRAD,
                "conv_step: bad mode / lda");
  }
  if (d->J > 1) DV3_REQUIRE(d->ring && (d->t || !need_t) && d->L >= (d->J - 1) * d->dil + 1,
                            "conv_step: k > 1 needs ring, t and L >= (k-1)*d+1");
  if (d->post_add || d->out_seq || d->x_ts) DV3_REQUIRE(d->t || !need_t, "conv_step: post_add / out_seq / x_ts need the step counter");
  const size_t lds = ((size_t)d->J * d->Cin * NB + (size_t)KS * MT * 2 * NB) * sizeof(float);
  DV3_REQUIRE(lds <= 64 * 1024, "conv_step: window too large for LDS (%zu bytes)", lds);
  *lds_out = lds;
  return DV3_OK;
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
