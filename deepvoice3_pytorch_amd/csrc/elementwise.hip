// HBM-bound helpers around the tap-GEMMs: gate/activation backward, dropout keep-bits,
// layout changes, embedding gather / dense grad, sinusoidal position encodings.
// Reference call sites are cited per kernel.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// Gate backward.  One wave per (b, channel) row, lanes along time (coalesced); row sums of the
// pre-activation gradients are written per (b, row) for the deterministic bias / speaker-bias
// reduction.  Autograd of modules.py:157-164 (GLU) and :224-226 (highway).
// VEC4: 16 bytes per lane and access for ANY T -- the tensors' bases are 16-byte aligned and every operand's row starts
// at the same element phase (row * T) & 3 (checked by the launcher: C % 4 == 0 in the gated modes), so a row is
// [head: up to 3 elements][quads at 16-byte boundaries][tail: up to 3 elements]; the head and tail elements are one
// 4-byte access of the first lanes.  T % 4 == 0 has no head or tail and sums in the order it always did.
// ------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const dv3_gate_bwd_desc p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // b*C + ch
  const int C = p.C, T = p.T;
  if (row >= (int64_t)p.B * C) return;
  const int b = (int)(row / C), ch = (int)(row % C);
  const float* dy = p.dy + row * T;
  float sa = 0.f, sg = 0.f;
  if (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY) {
    const float* a = p.ab_or_y + ((int64_t)b * 2 * C + ch) * T;
    const float* g = a + (int64_t)C * T;
    // bf16 pre-gate save: the same element offsets in a 2-byte tensor
    const uint16_t* a16 = reinterpret_cast<const uint16_t*>(p.ab_or_y) + ((int64_t)b * 2 * C + ch) * T;
    const uint16_t* g16 = a16 + (int64_t)C * T;
    const bool ab16 = p.ab_bf16 != 0;
    float* da = p.dab + ((int64_t)b * 2 * C + ch) * T;
    float* dg = da + (int64_t)C * T;
    const float k = (p.mode == DV3_EPI_GLU && p.residual) ? 0.70710678118654752440f : 1.0f;
    const float* x = p.x ? p.x + row * T : nullptr;
    float* dres = p.dres ? p.dres + row * T : nullptr;
    const bool glu = p.mode == DV3_EPI_GLU;
    // one element: (dy, a, g, x) -> (da, dg, dres)
    // (common.h: the same function runs inside the input-gradient tails that take this kernel's place, round 6)
    auto elem = [&](float dyv, float av, float gv, float xv, float& va, float& vg, float& vr) {
      dv3_gate_deriv(dyv * k, av, gv, xv, glu, va, vg, vr);
    };
    if (VEC4) {
      const int head = min(T, (int)((4 - ((row * T) & 3)) & 3));     // elements before the row's first 16-byte boundary
      const int nq = (T - head) >> 2;
      const int n_edge = T - 4 * nq;                                  // head + tail elements: at most 6
      if (lane < n_edge) {
        const int t = lane < head ? lane : 4 * nq + lane;             // tail element j sits at head + 4 nq + j
        float va, vg, vr;
        const float at = ab16 ? __uint_as_float((uint32_t)a16[t] << 16) : a[t];
        const float gt = ab16 ? __uint_as_float((uint32_t)g16[t] << 16) : g[t];
        elem(dy[t], at, gt, glu ? 0.f : x[t], va, vg, vr);
        if (dres) dres[t] = vr;
        if (p.dab_pair) {
          reinterpret_cast<uint32_t*>(da)[t] = dv3_pair_word(va);
          reinterpret_cast<uint32_t*>(dg)[t] = dv3_pair_word(vg);
        } else {
          da[t] = va;
          dg[t] = vg;
        }
        sa += va;
        sg += vg;
      }
      const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy + head);
      const f32x4* a4 = reinterpret_cast<const f32x4*>(a + head);
      const f32x4* g4 = reinterpret_cast<const f32x4*>(g + head);
      const f32x4* x4 = reinterpret_cast<const f32x4*>(x + head);
      f32x4* da4 = reinterpret_cast<f32x4*>(da + head);
      f32x4* dg4 = reinterpret_cast<f32x4*>(dg + head);
      f32x4* dres4 = reinterpret_cast<f32x4*>(dres + head);
      for (int q = lane; q < nq; q += 64) {
        const f32x4 dv = dy4[q];
        f32x4 av, gv;
        if (ab16) {         // 4 bf16 = 8 bytes per lane and operand
          const uint2 ua = reinterpret_cast<const uint2*>(a16 + head)[q], ug = reinterpret_cast<const uint2*>(g16 + head)[q];
          av = f32x4{__uint_as_float(ua.x << 16), __uint_as_float(ua.x & 0xffff0000u), __uint_as_float(ua.y << 16), __uint_as_float(ua.y & 0xffff0000u)};
          gv = f32x4{__uint_as_float(ug.x << 16), __uint_as_float(ug.x & 0xffff0000u), __uint_as_float(ug.y << 16), __uint_as_float(ug.y & 0xffff0000u)};
        } else {
          av = a4[q];
          gv = g4[q];
        }
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
        if (!glu) xv = x4[q];
        f32x4 oa, og, orr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float va, vg, vr;
          elem(dv[e], av[e], gv[e], xv[e], va, vg, vr);
          oa[e] = va; og[e] = vg; orr[e] = vr;
          sa += va;
          sg += vg;
        }
        if (p.dab_pair) {      // (uniform) the pre-gate gradient as the bf16 hi / lo pair its consumers would build from it
          typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
          reinterpret_cast<u32x4_*>(da + head)[q] = u32x4_{dv3_pair_word(oa[0]), dv3_pair_word(oa[1]), dv3_pair_word(oa[2]), dv3_pair_word(oa[3])};
          reinterpret_cast<u32x4_*>(dg + head)[q] = u32x4_{dv3_pair_word(og[0]), dv3_pair_word(og[1]), dv3_pair_word(og[2]), dv3_pair_word(og[3])};
        } else {
          da4[q] = oa;
          dg4[q] = og;
        }
        if (dres) dres4[q] = orr;
      }
    } else {
      for (int t = lane; t < T; t += 64) {
        float va, vg, vr;
        const float at = ab16 ? __uint_as_float((uint32_t)a16[t] << 16) : a[t];
        const float gt = ab16 ? __uint_as_float((uint32_t)g16[t] << 16) : g[t];
        elem(dy[t], at, gt, glu ? 0.f : x[t], va, vg, vr);
        if (dres) dres[t] = vr;
        if (p.dab_pair) {
          reinterpret_cast<uint32_t*>(da)[t] = dv3_pair_word(va);
          reinterpret_cast<uint32_t*>(dg)[t] = dv3_pair_word(vg);
        } else {
          da[t] = va;
          dg[t] = vg;
        }
        sa += va;
        sg += vg;
      }
    }
    sa = dv3_wave_sum(sa);
    sg = dv3_wave_sum(sg);
    if (lane == 0 && p.bias_part) {
      p.bias_part[(int64_t)b * 2 * C + ch] = sa;
      p.bias_part[(int64_t)b * 2 * C + C + ch] = sg;
    }
  } else {
    const float* y = p.ab_or_y ? p.ab_or_y + row * T : nullptr;
    float* dpre = p.dab ? p.dab + row * T : nullptr;
    const int mode = p.mode;
    const float alpha = p.alpha;
    auto act = [&](float dyv, float yv) {
      float d = dyv * alpha;
      if (mode == DV3_EPI_RELU) d = yv > 0.f ? d : 0.f;
      else if (mode == DV3_EPI_SIGMOID) d = d * yv * (1.0f - yv);
      else if (mode == DV3_EPI_SOFTSIGN) { const float q = 1.0f - fabsf(yv); d = d * q * q; }
      return d;
    };
    if (VEC4) {
      const int head = min(T, (int)((4 - ((row * T) & 3)) & 3));
      const int nq = (T - head) >> 2;
      const int n_edge = T - 4 * nq;
      if (lane < n_edge) {
        const int t = lane < head ? lane : 4 * nq + lane;
        const float d = act(dy[t], y ? y[t] : 0.f);
        if (dpre) dpre[t] = d;
        sa += d;
      }
      const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy + head);
      const f32x4* y4 = reinterpret_cast<const f32x4*>(y + head);
      f32x4* dpre4 = reinterpret_cast<f32x4*>(dpre + head);
      for (int q = lane; q < nq; q += 64) {
        const f32x4 dv = dy4[q];
        f32x4 yv = {0.f, 0.f, 0.f, 0.f};
        if (y) yv = y4[q];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = act(dv[e], yv[e]);
          sa += o[e];
        }
        if (dpre) dpre4[q] = o;
      }
    } else {
      for (int t = lane; t < T; t += 64) {
        const float d = act(dy[t], y ? y[t] : 0.f);
        if (dpre) dpre[t] = d;
        sa += d;
      }
    }
    sa = dv3_wave_sum(sa);
    if (lane == 0 && p.bias_part) p.bias_part[row] = sa;
  }
}

// ------------------------------------------------------------------------------------------
// Gate backward on channel-blocked bf16 tensors (include/dv3hip.h "c8": bf16 [B][C8][T][8]).  One workgroup per
// (b, 8-channel group), threads along time: every access is a whole 16-byte unit; the eight per-channel row sums
// (x2 for the gate half) are reduced across the workgroup for the deterministic bias reduction.
// ------------------------------------------------------------------------------------------
// FAST (default, dv3_debug_set(56, 0) = the libm forms of rounds 3-6): the sigmoid by v_exp_f32 + v_rcp_f32 as in the
// forward tails and the fp32 kernel above -- with expf and a true division the kernel was bound by its vector work
// (~45 instructions per element against 10 bytes: 55 % of the HBM rate stand-alone); the results are bf16 anyway.
typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
template <bool FAST>
__global__ __launch_bounds__(256) void gate_bwd_c8_kernel(const dv3_gate_bwd_desc p) {
  __shared__ float part[16][256];
  const int C = p.C, T = p.T, G = (C + 7) >> 3;     // channels >= C of the last group are zero in every c8 tensor
  const int g = blockIdx.x % G, b = blockIdx.x / G;
  const int tid = threadIdx.x;
  const bool gated = p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY;
  const int c8y = (C + 31) / 32 * 4, c8ab = (2 * C + 31) / 32 * 4;
  const gb_bf16x8* dy = reinterpret_cast<const gb_bf16x8*>(p.dy) + ((int64_t)b * c8y + g) * T;
  float sa[8], sg[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sa[e] = sg[e] = 0.f;
  if (gated) {
    const gb_bf16x8* au = reinterpret_cast<const gb_bf16x8*>(p.ab_or_y) + ((int64_t)b * c8ab + g) * T;
    const gb_bf16x8* gu = au + (int64_t)G * T;
    gb_bf16x8* dau = reinterpret_cast<gb_bf16x8*>(p.dab) + ((int64_t)b * c8ab + g) * T;
    gb_bf16x8* dgu = dau + (int64_t)G * T;
    const gb_bf16x8* xu = p.x ? reinterpret_cast<const gb_bf16x8*>(p.x) + ((int64_t)b * c8y + g) * T : nullptr;
    gb_bf16x8* dru = p.dres ? reinterpret_cast<gb_bf16x8*>(p.dres) + ((int64_t)b * c8y + g) * T : nullptr;
    const bool glu = p.mode == DV3_EPI_GLU;
    const float k = (glu && p.residual) ? 0.70710678118654752440f : 1.0f;
    for (int t = tid; t < T; t += 256) {
      const gb_bf16x8 dv = dy[t], av = au[t], gv = gu[t];
      gb_bf16x8 xv = dv;
      if (!glu) xv = xu[t];
      gb_bf16x8 oa, og, orr;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)dv[e] * k;
        const float sgm = FAST ? __builtin_amdgcn_rcpf(1.0f + __expf(-(float)gv[e])) : 1.0f / (1.0f + expf(-(float)gv[e]));
        const float va = d * sgm;
        float vg, vr;
        if (glu) {
          vg = d * (float)av[e] * sgm * (1.0f - sgm);
          vr = d;
        } else {
          vg = d * ((float)av[e] - (float)xv[e]) * sgm * (1.0f - sgm);
          vr = d * (1.0f - sgm);
        }
        oa[e] = (__bf16)va; og[e] = (__bf16)vg; orr[e] = (__bf16)vr;
        sa[e] += va;
        sg[e] += vg;
      }
      dau[t] = oa;
      dgu[t] = og;
      if (dru) dru[t] = orr;
    }
  } else {
    const gb_bf16x8* yu = p.ab_or_y ? reinterpret_cast<const gb_bf16x8*>(p.ab_or_y) + ((int64_t)b * c8y + g) * T : nullptr;
    gb_bf16x8* du = p.dab ? reinterpret_cast<gb_bf16x8*>(p.dab) + ((int64_t)b * c8y + g) * T : nullptr;
    for (int t = tid; t < T; t += 256) {
      const gb_bf16x8 dv = dy[t];
      gb_bf16x8 yv = dv, od;
      if (yu) yv = yu[t];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = (float)dv[e] * p.alpha;
        const float y = (float)yv[e];
        if (p.mode == DV3_EPI_RELU) d = y > 0.f ? d : 0.f;
        else if (p.mode == DV3_EPI_SIGMOID) d = d * y * (1.0f - y);
        else if (p.mode == DV3_EPI_SOFTSIGN) { const float q = 1.0f - fabsf(y); d = d * q * q; }
        od[e] = (__bf16)d;
        sa[e] += d;
      }
      if (du) du[t] = od;
    }
  }
  if (!p.bias_part) return;
  // 16 row sums over 256 threads: partials through LDS ([value][thread], conflict free), thread (value, slice) adds 16 of
  // them in a fixed order, a 16-lane butterfly finishes (row-local shuffles; a 64-lane butterfly per value and wave
  // was most of this kernel's time at T = 200)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    part[e][tid] = sa[e];
    part[8 + e][tid] = sg[e];
  }
  __syncthreads();
  const int v = tid >> 4, sl = tid & 15;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += part[v][sl + 16 * i];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
  if (sl == 0) {
    const int ch = g * 8 + (v & 7);
    if (gated) p.bias_part[(int64_t)b * 2 * C + (v >= 8 ? C : 0) + ch] = acc;
    else if (v < 8 && ch < C) p.bias_part[(int64_t)b * C + ch] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 keep-bit generator (replaces F.dropout's bernoulli_: modules.py:147,210).
// One thread per 32-bit mask word = 4 Philox calls x 8 sixteen-bit uniforms.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ uint32_t philox_keep_word(int64_t w, uint32_t thr, uint64_t seed, uint64_t site) {
  uint32_t word = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint64_t ci = (uint64_t)w * 4 + q;
    uint32_t o[4];
    philox4x32_10((uint32_t)ci, (uint32_t)(ci >> 32), (uint32_t)site, (uint32_t)(site >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t lo = o[e] & 0xFFFFu, hi = o[e] >> 16;
      word |= (uint32_t)(lo >= thr) << (8 * q + 2 * e);
      word |= (uint32_t)(hi >= thr) << (8 * q + 2 * e + 1);
    }
  }
  return word;
}

__global__ __launch_bounds__(256) void dropout_bits_kernel(uint32_t* __restrict__ bits,
                                                           int64_t n_words, uint32_t thr,
                                                           uint64_t seed, uint64_t site,
                                                           const uint64_t* __restrict__ dev_off) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  if (dev_off) seed += dev_off[0];
  bits[w] = philox_keep_word(w, thr, seed, site);
}

// The same keep decisions written straight as keep-BYTES of the channel-blocked layout ([B][C8][T], bit e of byte
// (b, g, t) = channel 8g+e; include/dv3hip.h "c8") -- what dv3_dropout_bits followed by dv3_mask_bits_to_c8 produce,
// in one launch.  One thread per (b, group, 32-frame word): eight Philox words, an 8 x 32 bit transpose, 32 bytes.
__device__ __forceinline__ void dropout_keep_c8_block(uint8_t* __restrict__ out, int B, int C, int T, int rs,
                                                      int c8p, uint32_t thr, uint64_t seed, uint64_t site,
                                                      const uint64_t* __restrict__ dev_off,
                                                      uint32_t* __restrict__ bits_out, uint32_t block,
                                                      uint32_t (*words)[9]) {
  // a workgroup = 32 entries (b, group, 32-frame word) x 8 channels: one Philox word per thread, the 8 x 32 bit
  // transpose through LDS, then thread (entry, q) writes bytes 4q .. 4q+3 of its entry (a wave covers 8 entries =
  // 256 contiguous bytes when T is a multiple of 32)
  const uint32_t n_ent = (uint32_t)B * (uint32_t)c8p * (uint32_t)rs;             // < 2^31 (host-checked)
  const uint32_t ent = block * 32u + (threadIdx.x >> 3);                           // ((b * c8p) + g) * rs + wi
  const uint32_t e = threadIdx.x & 7;
  if (dev_off) seed += dev_off[0];
  uint32_t wi = 0, bg = 0;
  if (ent < n_ent) {
    wi = ent % (uint32_t)rs;
    bg = ent / (uint32_t)rs;
    const uint32_t g = bg % (uint32_t)c8p, b = bg / (uint32_t)c8p;
    const uint32_t ch = g * 8u + e;
    const int64_t wrow = ((int64_t)b * C + ch) * rs + wi;
    const uint32_t w = ch < (uint32_t)C ? philox_keep_word(wrow, thr, seed, site) : 0u;
    words[threadIdx.x >> 3][e] = w;
    if (bits_out && ch < (uint32_t)C) bits_out[wrow] = w;      // the same decisions in the keep-bit form [B*C][rs]
  }
  __syncthreads();
  if (ent >= n_ent || !out) return;        // (out == NULL: the keep-bit form alone, a site of dv3_dropout_keep_c8_multi)
  const uint32_t q = e;                    // bytes 4q .. 4q+3
  const uint32_t* w = words[threadIdx.x >> 3];
  const uint32_t t0 = wi * 32u + 4u * q;
  uint8_t* o = out + (size_t)bg * T + t0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (t0 + i >= (uint32_t)T) break;
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) m |= ((w[c] >> (4 * q + i)) & 1u) << c;
    o[i] = (uint8_t)m;
  }
}
__global__ __launch_bounds__(256) void dropout_keep_c8_kernel(uint8_t* __restrict__ out, int B, int C, int T, int rs,
                                                              int c8p, uint32_t thr, uint64_t seed, uint64_t site,
                                                              const uint64_t* __restrict__ dev_off,
                                                              uint32_t* __restrict__ bits_out) {
  __shared__ uint32_t words[32][9];
  dropout_keep_c8_block(out, B, C, T, rs, c8p, thr, seed, site, dev_off, bits_out, blockIdx.x, words);
}
// several dropout sites in ONE launch (round 6: a step's ~25-35 mask launches, 6 us each on the forward's single queue):
// block -> (site, block of the site); the table is a kernel argument.  Per site the decisions of the single-site kernel.
struct DropoutMultiArgs {
  dv3_dropout_site s[DV3_DROPOUT_MULTI_MAX];
  uint32_t first_block[DV3_DROPOUT_MULTI_MAX + 1];
  uint32_t thr[DV3_DROPOUT_MULTI_MAX];
  int32_t n;
  uint64_t seed;
  const uint64_t* dev_off;
};
__global__ __launch_bounds__(256) void dropout_keep_c8_multi_kernel(const DropoutMultiArgs a) {
  __shared__ uint32_t words[32][9];
  const uint32_t blk = blockIdx.x;
  int l = 0;
#pragma unroll 8
  for (int k = 1; k < DV3_DROPOUT_MULTI_MAX; ++k)
    if (k < a.n && a.first_block[k] <= blk) l = k;
  const dv3_dropout_site& e = a.s[l];
  dropout_keep_c8_block(e.keep, e.B, e.C, e.T, (e.T + 31) / 32, (e.C + 31) / 32 * 4, a.thr[l], a.seed, e.site, a.dev_off, e.bits,
                        blk - a.first_block[l], words);
}

// ------------------------------------------------------------------------------------------
// y[b][c][r] = alpha * x[b][r][c] (+ add[b][c][r]) -- BTC <-> BCT (deepvoice3.py:85,92,318,324,
// 352-356: the reference's .transpose(1,2)[.contiguous()] calls).  32x32 LDS tile.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x,
                                                        float* __restrict__ y,
                                                        const float* __restrict__ add, int R, int C,
                                                        float alpha) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xb = x + (int64_t)b * R * C;
  float* yb = y + (int64_t)b * R * C;
  const float* ab = add ? add + (int64_t)b * R * C : nullptr;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? xb[(int64_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < C && r < R) {
      float v = alpha * tile[tx][ty + 8 * k];
      if (ab) v += ab[(int64_t)c * R + r];
      yb[(int64_t)c * R + r] = v;
    }
  }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a,
                                                    const float* __restrict__ b,
                                                    float* __restrict__ out, int64_t n, float alpha) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) out[i] = alpha * (a[i] + (b ? b[i] : 0.f));
}

__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x,
                                                            const uint32_t* __restrict__ bits, int rs,
                                                            float scale, float* __restrict__ out,
                                                            int64_t rows, int T) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  for (int t = lane; t < T; t += 64)
    out[row * T + t] = dv3_keep(bits, row, rs, t) ? x[row * T + t] * scale : 0.f;
}

// dy [B][O][2T] -> out [B][2*O][T] with out[b][j*O+o][t] = dy[b][o][2t+j]
// (operand of the ConvTranspose1d k2 s2 backward GEMMs; deepvoice3.py:519-520,527-528)
__global__ __launch_bounds__(256) void deinterleave2_kernel(const float* __restrict__ dy,
                                                            float* __restrict__ out, int O, int T,
                                                            int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) {  // i over [B][O][2T]
    const int tt = (int)(i % (2 * T));
    const int64_t bo = i / (2 * T);
    const int o = (int)(bo % O);
    const int64_t b = bo / O;
    const int j = tt & 1, t = tt >> 1;
    out[((b * 2 + j) * O + o) * (int64_t)T + t] = dy[i];
  }
}

// the same with wide accesses: a thread takes 4 consecutive samples of a row (one 16-byte load) and writes 2 even + 2
// odd ones (two 8-byte stores); T % 2 == 0 (rows of 2T floats stay 16-byte aligned, rows of T floats 8-byte aligned)
// and 16-byte aligned bases.  i over [B*O][T/2]
typedef float dl_f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void deinterleave2_vec_kernel(const float* __restrict__ dy, float* __restrict__ out,
                                                                int O, int T, int64_t n2) {
  const int T2 = T >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % T2);
    const int64_t bo = i / T2;
    const int o = (int)(bo % O);
    const int64_t b = bo / O;
    const f32x4 v = *reinterpret_cast<const f32x4*>(dy + bo * 2 * (int64_t)T + 4 * (int64_t)q);
    const dl_f32x2 ev = {v[0], v[2]}, od = {v[1], v[3]};
    *reinterpret_cast<dl_f32x2*>(out + ((b * 2 + 0) * O + o) * (int64_t)T + 2 * q) = ev;
    *reinterpret_cast<dl_f32x2*>(out + ((b * 2 + 1) * O + o) * (int64_t)T + 2 * q) = od;
  }
}

// ------------------------------------------------------------------------------------------
// Embedding gather straight into BCT (+dropout): deepvoice3.py:74-75, nyanko.py:63.
// block = 32 time steps x 32 channels through an LDS tile so both sides are coalesced.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_bct_kernel(const int64_t* __restrict__ idx,
                                                            const float* __restrict__ w,
                                                            float* __restrict__ out,
                                                            const uint32_t* __restrict__ mask,
                                                            int mask_rs, float dscale, int T, int C,
                                                            int n_vocab) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = t0 + ty + 8 * k, c = c0 + tx;
    float v = 0.f;
    if (t < T && c < C) {
      int64_t id = idx[(int64_t)b * T + t];
      id = id < 0 ? 0 : (id >= n_vocab ? n_vocab - 1 : id);
      v = w[id * C + c];
    }
    tile[ty + 8 * k][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, t = t0 + tx;
    if (c < C && t < T) {
      float v = tile[tx][ty + 8 * k];
      if (mask) v = dv3_keep(mask, (int64_t)b * C + c, mask_rs, t) ? v * dscale : 0.f;
      out[((int64_t)b * C + c) * T + t] = v;
    }
  }
}

// Incremental-conv window (conv.py:34-46: `input_buffer[:, :-1] = input_buffer[:, 1:]; buffer[:, -1] = x`):
// buf [rows][L] shifts left by one frame and takes x[row] as its newest frame.  One wave per row: all
// lanes load (l+1) before any lane stores (l), so the in-place shift is race free; static addresses keep
// the decode step hipGraph-capturable.
__global__ __launch_bounds__(256) void shift_append_kernel(float* __restrict__ buf,
                                                           const float* __restrict__ x, int64_t rows,
                                                           int L, int64_t x_stride) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* b = buf + row * L;
  const float newest = x[row * x_stride];
  for (int l0 = 0; l0 < L; l0 += 64) {       // L <= 64 for every reference layer; loop kept for generality
    const int l = l0 + lane;
    float v = 0.f;
    if (l < L) v = (l + 1 < L) ? b[l + 1] : newest;
    __builtin_amdgcn_s_waitcnt(0);           // every load of this pass has landed
    __builtin_amdgcn_wave_barrier();
    if (l < L) b[l] = v;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

// Dense embedding gradient, deterministic: one block per vocabulary id.  The (b,t) positions that
// hold the id are compacted IN ORDER into LDS (each thread scans a contiguous slice, counts, an
// exclusive scan gives its write offset), then every thread owns one channel and sums dout over
// the list:  dW[v][c] = sum_{(b,t): idx==v} dout[b][c][t]*keep*scale.  padding_idx row = 0.
// The index array is processed in ranges of EMB_RANGE positions so the list always fits.
constexpr int EMB_RANGE = 4096;
__global__ __launch_bounds__(256) void embedding_bct_bwd_kernel(const int64_t* __restrict__ idx,
                                                                const float* __restrict__ dout,
                                                                float* __restrict__ dw,
                                                                const uint32_t* __restrict__ mask,
                                                                int mask_rs, float dscale, int B, int T,
                                                                int C, int padding_idx) {
  __shared__ int list[EMB_RANGE];
  __shared__ int cnt[256];
  __shared__ int total;
  const int v = blockIdx.x;
  const int tid = threadIdx.x;
  const int n = B * T;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};   // channels tid, tid+256, ... (C <= 1024)
  if (v != padding_idx) {
    for (int r0 = 0; r0 < n; r0 += EMB_RANGE) {
      const int rn = min(EMB_RANGE, n - r0);
      const int len = (rn + 255) / 256;
      const int lo = min(tid * len, rn), hi = min(lo + len, rn);
      int c = 0;
      for (int i = lo; i < hi; ++i) c += (idx[r0 + i] == v);
      cnt[tid] = c;
      __syncthreads();
      if (tid == 0) {
        int run = 0;
        for (int k = 0; k < 256; ++k) {
          const int t = cnt[k];
          cnt[k] = run;
          run += t;
        }
        total = run;
      }
      __syncthreads();
      int o = cnt[tid];
      for (int i = lo; i < hi; ++i)
        if (idx[r0 + i] == v) list[o++] = r0 + i;
      __syncthreads();
      const int m = total;
      for (int q = 0; q < m; ++q) {
        const int bt = list[q];
        const int b = bt / T, t = bt - b * T;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ch = tid + k * 256;
          if (ch < C) {
            float g = dout[((int64_t)b * C + ch) * T + t];
            if (mask) g = dv3_keep(mask, (int64_t)b * C + ch, mask_rs, t) ? g * dscale : 0.f;
            acc[k] += g;
          }
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ch = tid + k * 256;
    if (ch < C) dw[(int64_t)v * C + ch] = acc[k];
  }
}

// ------------------------------------------------------------------------------------------
// SinusoidalEncoding.forward (modules.py:30-64): gather the raw-angle table at `pos`, scale
// by the (per-batch) rate w, sin on even / cos on odd channels, row 0 (padding) untouched.
//   out[b][c][t] = base[b][c][t] + enc    (base may be NULL)      -- BCT output
// table: [n_pos][C] fp32 (the module's .weight, in the reference's state_dict).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sincos_pos_kernel(const int64_t* __restrict__ pos,
                                                         const float* __restrict__ table,
                                                         const float* __restrict__ w, int w_per_batch,
                                                         const float* __restrict__ base,
                                                         float* __restrict__ out, int T, int C,
                                                         int n_pos, int apply_sincos) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float rate = w ? w[w_per_batch ? b : 0] : 1.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = t0 + ty + 8 * k, c = c0 + tx;
    float v = 0.f;
    if (t < T && c < C) {
      int64_t p = pos[(int64_t)b * T + t];
      p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
      const float ang = rate * table[p * C + c];
      if (!apply_sincos || p == 0) v = ang;
      else v = (c & 1) ? cosf(ang) : sinf(ang);
    }
    tile[ty + 8 * k][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, t = t0 + tx;
    if (c < C && t < T) {
      const int64_t o = ((int64_t)b * C + c) * T + t;
      out[o] = tile[tx][ty + 8 * k] + (base ? base[o] : 0.f);
    }
  }
}

// dw[b] = sum_{c,t} dout[b][c][t] * a * (c even ? cos(w a) : -sin(w a)), a = table[pos][c], pos != 0
__global__ __launch_bounds__(256) void sincos_pos_bwd_kernel(const int64_t* __restrict__ pos,
                                                             const float* __restrict__ table,
                                                             const float* __restrict__ w, int w_per_batch,
                                                             const float* __restrict__ dout,
                                                             float* __restrict__ dw, int T, int C,
                                                             int n_pos) {
  const int b = blockIdx.x;
  const float rate = w[w_per_batch ? b : 0];
  float s = 0.f;
  for (int idx = threadIdx.x; idx < C * T; idx += 256) {
    const int c = idx / T, t = idx % T;
    int64_t p = pos[(int64_t)b * T + t];
    p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
    const float a = table[p * C + c];
    const float d = dout[((int64_t)b * C + c) * T + t];
    float de;
    if (p == 0) de = a;
    else de = (c & 1) ? -sinf(rate * a) * a : cosf(rate * a) * a;
    s += d * de;
  }
  __shared__ float red[4];
  s = dv3_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dw[b] = red[0] + red[1] + red[2] + red[3];
}

// two-stage form: chunk `blockIdx.y` of batch item `blockIdx.x` (a contiguous range of the flat (c, t) index)
__global__ __launch_bounds__(256) void sincos_pos_bwd_part_kernel(const int64_t* __restrict__ pos, const float* __restrict__ table,
                                                                  const float* __restrict__ w, int w_per_batch,
                                                                  const float* __restrict__ dout, float* __restrict__ partial,
                                                                  int T, int C, int n_pos) {
  const int b = blockIdx.x, ch = blockIdx.y, nch = gridDim.y;
  const float rate = w[w_per_batch ? b : 0];
  const int total = C * T, per = (total + nch - 1) / nch;
  const int lo = ch * per, hi = min(total, lo + per);
  float s = 0.f;
  for (int idx = lo + threadIdx.x; idx < hi; idx += 256) {
    const int c = idx / T, t = idx - c * T;
    int64_t p = pos[(int64_t)b * T + t];
    p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
    const float a = table[p * C + c];
    const float d = dout[((int64_t)b * C + c) * T + t];
    float de;
    if (p == 0) de = a;
    else de = (c & 1) ? -sinf(rate * a) * a : cosf(rate * a) * a;
    s += d * de;
  }
  __shared__ float red[4];
  s = dv3_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)b * nch + ch] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void sincos_pos_bwd_finish_kernel(const float* __restrict__ partial, int nch, float* __restrict__ dw, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int k = 0; k < nch; ++k) s += partial[(int64_t)b * nch + k];
  dw[b] = s;
}

// dtable[p][c] = sum over the (b, t) with pos[b][t] == p of dout[b][c][t] * d enc / d table   (p >= 1; row 0 is the
// padding row, F.embedding(padding_idx=0) gives it no gradient: modules.py:45-64 with trainable position tables).
// One workgroup per table row, lanes over channels, a fixed (b, t) scan order: deterministic, no atomics.
__global__ __launch_bounds__(256) void sincos_pos_table_bwd_kernel(const int64_t* __restrict__ pos,
                                                                   const float* __restrict__ table,
                                                                   const float* __restrict__ w, int w_per_batch,
                                                                   const float* __restrict__ dout,
                                                                   float* __restrict__ dtable, int B, int T, int C,
                                                                   int n_pos, int apply_sincos) {
  const int p = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    if (p > 0) {
      const float a = table[(int64_t)p * C + c];
      for (int b = 0; b < B; ++b) {
        const float rate = w ? w[w_per_batch ? b : 0] : 1.0f;
        float de = rate;
        if (apply_sincos) de = (c & 1) ? -sinf(rate * a) * rate : cosf(rate * a) * rate;
        for (int t = 0; t < T; ++t) {
          int64_t q = pos[(int64_t)b * T + t];
          q = q < 0 ? 0 : (q >= n_pos ? n_pos - 1 : q);
          if (q == p) acc += dout[((int64_t)b * C + c) * T + t] * de;
        }
      }
    }
    dtable[(int64_t)p * C + c] = acc;
  }
}

// Ragged -> padded rows of 32-bit words (the padding half of train.collate_fn, train.py:293-360, on the
// device): one workgroup per output row.  HBM-bound byte mover: each output word is written once, each
// source word read at most once, both coalesced along the row.
__global__ __launch_bounds__(256) void ragged_pad_rows_kernel(const uint32_t* __restrict__ src,
                                                              const int32_t* __restrict__ row_off,
                                                              uint32_t* __restrict__ out, int T_out, int D,
                                                              int lead, int t_stride) {
  const int t = blockIdx.x, b = blockIdx.y;
  const int r0 = row_off[b], n = row_off[b + 1] - r0;
  const int s = t * t_stride - lead;
  const bool inside = s >= 0 && s < n;                      // uniform per workgroup
  const uint32_t* in = src + (int64_t)(r0 + (inside ? s : 0)) * D;
  uint32_t* o = out + ((int64_t)b * T_out + t) * D;
  for (int d = threadIdx.x; d < D; d += 256) o[d] = inside ? in[d] : 0u;
}

}  // namespace

extern "C" int dv3_sincos_pos_bwd_f32(const int64_t* pos, const float* table, const float* w,
                                      int32_t w_per_batch, const float* dout, float* dw, int32_t B,
                                      int32_t T, int32_t C, int32_t n_pos, void* stream) {
  DV3_REQUIRE(pos && table && w && dout && dw && B > 0 && T > 0 && C > 0 && n_pos > 0, "sincos_pos_bwd: bad args");
  hipLaunchKernelGGL(sincos_pos_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pos, table, w,
                     w_per_batch, dout, dw, T, C, n_pos);
  return dv3_check_launch("sincos_pos_bwd_f32");
}

extern "C" int dv3_sincos_pos_bwd2_f32(const int64_t* pos, const float* table, const float* w, int32_t w_per_batch,
                                       const float* dout, float* partial, int32_t n_chunks, float* dw, int32_t B, int32_t T,
                                       int32_t C, int32_t n_pos, void* stream) {
  DV3_REQUIRE(pos && table && w && dout && dw && partial && B > 0 && T > 0 && C > 0 && n_pos > 0, "sincos_pos_bwd2: bad args");
  DV3_REQUIRE(n_chunks >= 1 && n_chunks <= 1024, "sincos_pos_bwd2: 1..1024 chunks");
  hipLaunchKernelGGL(sincos_pos_bwd_part_kernel, dim3(B, n_chunks), dim3(256), 0, (hipStream_t)stream, pos, table, w,
                     w_per_batch, dout, partial, T, C, n_pos);
  hipLaunchKernelGGL(sincos_pos_bwd_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial, n_chunks, dw, B);
  return dv3_check_launch("sincos_pos_bwd2_f32");
}

extern "C" int dv3_sincos_pos_table_bwd_f32(const int64_t* pos, const float* table, const float* w, int32_t w_per_batch,
                                            const float* dout, float* dtable, int32_t B, int32_t T, int32_t C,
                                            int32_t n_pos, int32_t apply_sincos, void* stream) {
  DV3_REQUIRE(pos && table && dout && dtable && B > 0 && T > 0 && C > 0 && n_pos > 0, "sincos_pos_table_bwd: bad args");
  hipLaunchKernelGGL(sincos_pos_table_bwd_kernel, dim3(n_pos), dim3(256), 0, (hipStream_t)stream, pos, table, w,
                     w_per_batch, dout, dtable, B, T, C, n_pos, apply_sincos);
  return dv3_check_launch("sincos_pos_table_bwd_f32");
}

int g_gate_c8_fast = 1;   // dv3_debug_set(56, v): gate_bwd_c8's sigmoid by v_exp_f32 + v_rcp_f32 (0 = expf and a division)
int g_gate_vec = 1;   // dv3_debug_set(55, v): 0 = 16-byte accesses only for gated layers with T % 4 == 0 (rounds 3-6a), the rest 4-byte

extern "C" int dv3_gate_bwd_f32(const dv3_gate_bwd_desc* d, void* stream) {
  DV3_REQUIRE(d && d->dy, "gate_bwd: null dy");
  DV3_REQUIRE(d->B > 0 && d->C > 0 && d->T > 0, "gate_bwd: bad dims");
  const bool gated = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  if (gated) {
    DV3_REQUIRE(d->ab_or_y && d->dab, "gate_bwd: gated mode needs ab and dab");
    DV3_REQUIRE(d->mode != DV3_EPI_HIGHWAY || d->x, "gate_bwd: highway needs x");
  } else {
    DV3_REQUIRE(d->mode == DV3_EPI_LINEAR || d->ab_or_y, "gate_bwd: activation mode needs y");
    DV3_REQUIRE(d->mode != DV3_EPI_DGRAD, "gate_bwd: bad mode");
  }
  const int64_t rows = (int64_t)d->B * d->C;
  if (d->c8) {
    const uintptr_t pc = (uintptr_t)d->dy | (uintptr_t)d->ab_or_y | (uintptr_t)d->dab | (uintptr_t)d->x | (uintptr_t)d->dres;
    DV3_REQUIRE((!gated || (d->C & 7) == 0) && (pc & 15) == 0 && !d->ab_bf16,
                "gate_bwd: c8 tensors need 16-byte alignment and, in the gated modes, C % 8 == 0");
    const int64_t blocks = (int64_t)d->B * ((d->C + 7) / 8);
    DV3_REQUIRE(blocks < (1ll << 31), "gate_bwd: grid too large");
    if (g_gate_c8_fast) hipLaunchKernelGGL(gate_bwd_c8_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(gate_bwd_c8_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d);
    return dv3_check_launch("gate_bwd_c8");
  }
  // 16 bytes per lane whenever the bases are 16-byte aligned and the operands' rows share their element phase
  const uintptr_t ptrs = (uintptr_t)d->dy | (uintptr_t)d->ab_or_y | (uintptr_t)d->dab | (uintptr_t)d->x | (uintptr_t)d->dres;
  DV3_REQUIRE(!d->ab_bf16 || gated, "gate_bwd: ab_bf16 is for the gated modes");
  DV3_REQUIRE(!d->dab_pair || (gated && !d->c8), "gate_bwd: pair words are written by the gated modes on fp32 tensors");
  if ((ptrs & 15) == 0 && (g_gate_vec ? (!gated || (d->C & 3) == 0) : (gated && (d->T & 3) == 0)))
    hipLaunchKernelGGL(gate_bwd_kernel<true>, dim3((unsigned)dv3_cdiv64(rows, 4)), dim3(256), 0,
                       (hipStream_t)stream, *d);
  else
    hipLaunchKernelGGL(gate_bwd_kernel<false>, dim3((unsigned)dv3_cdiv64(rows, 4)), dim3(256), 0,
                       (hipStream_t)stream, *d);
  return dv3_check_launch("gate_bwd_f32");
}

extern "C" int dv3_dropout_bits(uint32_t* bits, int64_t n_words, float p, uint64_t seed,
                                uint64_t site, const uint64_t* dev_seed_offset, void* stream) {
  DV3_REQUIRE(bits && n_words > 0, "dropout_bits: bad args");
  DV3_REQUIRE(p >= 0.f && p < 1.f, "dropout_bits: p out of range");
  uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
  hipLaunchKernelGGL(dropout_bits_kernel, dim3((unsigned)dv3_cdiv64(n_words, 256)), dim3(256), 0,
                     (hipStream_t)stream, bits, n_words, thr, seed, site, dev_seed_offset);
  return dv3_check_launch("dropout_bits");
}

static int dropout_keep_launch(uint8_t* out, uint32_t* bits, int32_t B, int32_t C, int32_t T, float p, uint64_t seed,
                               uint64_t site, const uint64_t* dev_seed_offset, void* stream);
extern "C" int dv3_dropout_keep_c8(uint8_t* out, int32_t B, int32_t C, int32_t T, float p, uint64_t seed, uint64_t site,
                                   const uint64_t* dev_seed_offset, void* stream) {
  return dropout_keep_launch(out, nullptr, B, C, T, p, seed, site, dev_seed_offset, stream);
}
extern "C" int dv3_dropout_bits_keep(uint32_t* bits, uint8_t* keep, int32_t B, int32_t C, int32_t T, float p, uint64_t seed,
                                     uint64_t site, const uint64_t* dev_seed_offset, void* stream) {
  DV3_REQUIRE(bits, "dropout_bits_keep: null pointer");
  return dropout_keep_launch(keep, bits, B, C, T, p, seed, site, dev_seed_offset, stream);
}
static int dropout_keep_launch(uint8_t* out, uint32_t* bits, int32_t B, int32_t C, int32_t T, float p, uint64_t seed,
                               uint64_t site, const uint64_t* dev_seed_offset, void* stream) {
  DV3_REQUIRE(out && B > 0 && C > 0 && T > 0, "dropout_keep_c8: bad args");
  DV3_REQUIRE(p >= 0.f && p < 1.f, "dropout_keep_c8: p out of range");
  const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
  const int rs = (T + 31) / 32, c8p = (C + 31) / 32 * 4;
  const int64_t n = (int64_t)B * c8p * rs;
  DV3_REQUIRE((int64_t)B * c8p * T < (1ll << 31), "dropout_keep_c8: mask exceeds the 2 GB the kernel can address");
  hipLaunchKernelGGL(dropout_keep_c8_kernel, dim3((unsigned)dv3_cdiv64(n, 32)), dim3(256), 0, (hipStream_t)stream, out,
                     B, C, T, rs, c8p, thr, seed, site, dev_seed_offset, bits);
  return dv3_check_launch("dropout_keep_c8");
}

extern "C" int dv3_dropout_keep_c8_multi(const dv3_dropout_site* sites, int32_t n, uint64_t seed,
                                         const uint64_t* dev_seed_offset, void* stream) {
  DV3_REQUIRE(sites && n > 0 && n <= DV3_DROPOUT_MULTI_MAX, "dropout_keep_c8_multi: 1..%d sites", DV3_DROPOUT_MULTI_MAX);
  DropoutMultiArgs a;
  int64_t total = 0;
  for (int l = 0; l < n; ++l) {
    const dv3_dropout_site& e = sites[l];
    DV3_REQUIRE((e.keep || e.bits) && e.B > 0 && e.C > 0 && e.T > 0, "dropout_keep_c8_multi: bad site %d", l);
    DV3_REQUIRE(e.p >= 0.f && e.p < 1.f, "dropout_keep_c8_multi: p out of range (site %d)", l);
    const int rs = (e.T + 31) / 32, c8p = (e.C + 31) / 32 * 4;
    DV3_REQUIRE((int64_t)e.B * c8p * e.T < (1ll << 31), "dropout_keep_c8_multi: mask %d exceeds the 2 GB the kernel can address", l);
    a.s[l] = e;
    a.thr[l] = (uint32_t)(e.p * 65536.0f + 0.5f);
    a.first_block[l] = (uint32_t)total;
    total += dv3_cdiv64((int64_t)e.B * c8p * rs, 32);
  }
  DV3_REQUIRE(total < (1ll << 31), "dropout_keep_c8_multi: grid too large");
  for (int l = n; l <= DV3_DROPOUT_MULTI_MAX; ++l) a.first_block[l] = (uint32_t)total;
  a.n = n; a.seed = seed; a.dev_off = dev_seed_offset;
  hipLaunchKernelGGL(dropout_keep_c8_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
  return dv3_check_launch("dropout_keep_c8_multi");
}

extern "C" int dv3_dropout_apply_f32(const float* x, const uint32_t* bits, int32_t bits_rs,
                                     float scale, float* out, int64_t rows, int32_t T, void* stream) {
  DV3_REQUIRE(x && bits && out && rows > 0 && T > 0 && bits_rs * 32 >= T, "dropout_apply: bad args");
  hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)dv3_cdiv64(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, x, bits, bits_rs, scale, out, rows, T);
  return dv3_check_launch("dropout_apply_f32");
}

extern "C" int dv3_transpose_f32(const float* x, float* y, const float* add, int32_t B, int32_t R,
                                 int32_t C, float alpha, void* stream) {
  DV3_REQUIRE(x && y && B > 0 && R > 0 && C > 0, "transpose: bad args");
  hipLaunchKernelGGL(transpose_kernel, dim3(dv3_cdiv(C, 32), dv3_cdiv(R, 32), B), dim3(256), 0,
                     (hipStream_t)stream, x, y, add, R, C, alpha);
  return dv3_check_launch("transpose_f32");
}

extern "C" int dv3_axpby_f32(const float* a, const float* b, float* out, int64_t n, float alpha,
                             void* stream) {
  DV3_REQUIRE(a && out && n > 0, "axpby: bad args");
  int64_t blocks = dv3_cdiv64(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b,
                     out, n, alpha);
  return dv3_check_launch("axpby_f32");
}

__global__ void sum_scalars_kernel(const float* a, const float* b, const float* c, const float* d, float* out) {
  float s = a[0] + b[0];
  if (c) s += c[0];
  if (d) s += d[0];
  out[0] = s;
}
extern "C" int dv3_sum_scalars_f32(const float* a, const float* b, const float* c, const float* d, float* out,
                                   void* stream) {
  DV3_REQUIRE(a && b && out, "sum_scalars: bad args");
  hipLaunchKernelGGL(sum_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, b, c, d, out);
  return dv3_check_launch("sum_scalars_f32");
}
// A fill KERNEL, not hipMemsetAsync (round 6): captured into a hipGraph the runtime's memset node replayed with a fill
// pattern whose odd 32-bit words were garbage in some processes -- the gradient arena "zeroed" to {0, c, 0, c, ...} and
// every parameter gradient off by c on half of its elements (found when the step's first launches moved:
// profiles/r06_memset_node.txt).  Bytes up to the first / after the last 16-byte boundary singly, the rest as 16-byte stores.
namespace {
__global__ __launch_bounds__(256) void fill_bytes_kernel(unsigned char* __restrict__ p, uint32_t word, int64_t head, int64_t n16,
                                                         int64_t tail) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ v = {word, word, word, word};
  u32x4_* const q = reinterpret_cast<u32x4_*>(p + head);
  for (int64_t k = i; k < n16; k += stride) q[k] = v;
  if (i < head) p[i] = (unsigned char)word;
  if (i < tail) p[head + n16 * 16 + i] = (unsigned char)word;
}
}  // namespace
namespace {
__global__ __launch_bounds__(256) void fill_rows_kernel(unsigned char* __restrict__ p, uint32_t word, int64_t rows, int64_t row16,
                                                        int64_t stride16) {
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ v = {word, word, word, word};
  u32x4_* const q = reinterpret_cast<u32x4_*>(p);
  const int64_t n = rows * row16;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const int64_t r = k / row16, c = k - r * row16;
    q[r * stride16 + c] = v;
  }
}
}  // namespace
extern "C" int dv3_memset_rows_b8(void* p, int32_t value, int64_t rows, int64_t row_bytes, int64_t row_stride_bytes, void* stream) {
  DV3_REQUIRE(p && rows >= 0 && row_bytes >= 0 && row_stride_bytes >= row_bytes, "memset_rows: bad args");
  DV3_REQUIRE((((uintptr_t)p | (uintptr_t)row_bytes | (uintptr_t)row_stride_bytes) & 15) == 0, "memset_rows: 16-byte units");
  if (rows == 0 || row_bytes == 0) return DV3_OK;
  const uint32_t word = ((uint32_t)value & 0xffu) * 0x01010101u;
  int64_t blocks = (rows * (row_bytes >> 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)p, word, rows,
                     row_bytes >> 4, row_stride_bytes >> 4);
  return dv3_check_launch("memset_rows_b8");
}
extern "C" int dv3_memset_b8(void* p, int32_t value, int64_t bytes, void* stream) {
  DV3_REQUIRE(p && bytes >= 0, "memset: bad args");
  if (bytes == 0) return DV3_OK;
  const uint32_t b = (uint32_t)value & 0xffu, word = b * 0x01010101u;
  int64_t head = (int64_t)((16 - ((uintptr_t)p & 15)) & 15);
  if (head > bytes) head = bytes;
  const int64_t n16 = (bytes - head) >> 4, tail = bytes - head - n16 * 16;
  int64_t blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)p, word, head,
                     n16, tail);
  return dv3_check_launch("memset_b8");
}

extern "C" int dv3_deinterleave2_f32(const float* dy, float* out, int32_t B, int32_t O, int32_t T,
                                     void* stream) {
  DV3_REQUIRE(dy && out && B > 0 && O > 0 && T > 0, "deinterleave2: bad args");
  const int64_t n = (int64_t)B * O * 2 * T;
  if ((T & 1) == 0 && (((uintptr_t)dy | (uintptr_t)out) & 15) == 0) {
    const int64_t n2 = n / 4;
    int64_t blocks = dv3_cdiv64(n2, 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(deinterleave2_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, out, O, T, n2);
    return dv3_check_launch("deinterleave2_f32");
  }
  int64_t blocks = dv3_cdiv64(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(deinterleave2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     dy, out, O, T, n);
  return dv3_check_launch("deinterleave2_f32");
}

extern "C" int dv3_embedding_bct_f32(const int64_t* idx, const float* w, float* out,
                                     const uint32_t* mask, int32_t mask_rs, float drop_scale,
                                     int32_t B, int32_t T, int32_t C, int32_t n_vocab, void* stream) {
  DV3_REQUIRE(idx && w && out && B > 0 && T > 0 && C > 0 && n_vocab > 0, "embedding: bad args");
  hipLaunchKernelGGL(embedding_bct_kernel, dim3(dv3_cdiv(C, 32), dv3_cdiv(T, 32), B), dim3(256), 0,
                     (hipStream_t)stream, idx, w, out, mask, mask_rs, drop_scale, T, C, n_vocab);
  return dv3_check_launch("embedding_bct_f32");
}

extern "C" int dv3_embedding_bct_bwd_f32(const int64_t* idx, const float* dout, float* dw,
                                         const uint32_t* mask, int32_t mask_rs, float drop_scale,
                                         int32_t B, int32_t T, int32_t C, int32_t n_vocab,
                                         int32_t padding_idx, void* stream) {
  DV3_REQUIRE(idx && dout && dw && B > 0 && T > 0 && C > 0 && n_vocab > 0, "embedding_bwd: bad args");
  DV3_REQUIRE(C <= 1024, "embedding_bwd: C > 1024 not supported");
  hipLaunchKernelGGL(embedding_bct_bwd_kernel, dim3(n_vocab), dim3(256), 0,
                     (hipStream_t)stream, idx, dout, dw, mask, mask_rs, drop_scale, B, T, C,
                     padding_idx);
  return dv3_check_launch("embedding_bct_bwd_f32");
}

extern "C" int dv3_ragged_pad_rows_b32(const uint32_t* src, const int32_t* row_off, uint32_t* out, int32_t B,
                                       int32_t T_out, int32_t D, int32_t lead, int32_t t_stride, void* stream) {
  DV3_REQUIRE(src && row_off && out, "ragged_pad_rows: null pointer");
  DV3_REQUIRE(B > 0 && T_out > 0 && D > 0 && t_stride > 0 && lead >= 0 && B <= 65535, "ragged_pad_rows: bad arguments");
  hipLaunchKernelGGL(ragged_pad_rows_kernel, dim3((unsigned)T_out, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     src, row_off, out, T_out, D, lead, t_stride);
  return dv3_check_launch("ragged_pad_rows_b32");
}

extern "C" int dv3_shift_append_f32(float* buf, const float* x, int64_t rows, int32_t L, int64_t x_stride,
                                    void* stream) {
  DV3_REQUIRE(buf && x && rows > 0 && L > 0, "shift_append: bad args");
  hipLaunchKernelGGL(shift_append_kernel, dim3((unsigned)dv3_cdiv64(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, buf, x, rows, L, x_stride);
  return dv3_check_launch("shift_append_f32");
}

extern "C" int dv3_sincos_pos_bct_f32(const int64_t* pos, const float* table, const float* w,
                                      int32_t w_per_batch, const float* base, float* out, int32_t B,
                                      int32_t T, int32_t C, int32_t n_pos, int32_t apply_sincos,
                                      void* stream) {
  DV3_REQUIRE(pos && table && out && B > 0 && T > 0 && C > 0 && n_pos > 0, "sincos_pos: bad args");
  hipLaunchKernelGGL(sincos_pos_kernel, dim3(dv3_cdiv(C, 32), dv3_cdiv(T, 32), B), dim3(256), 0,
                     (hipStream_t)stream, pos, table, w, w_per_batch, base, out, T, C, n_pos,
                     apply_sincos);
  return dv3_check_launch("sincos_pos_bct_f32");
}

// ------------------------------------------------------------------------------------------------------------------
// Valid-length steps (ABI 42, include/dv3hip.h: dv3_zero_tail_b32).  A batch padded beyond its own longest item -- to a
// lattice shape, so that a captured step can be replayed for it -- must still compute what the reference computes on the
// batch padded to its own maximum: there a non-causal convolution reads nn.Conv1d's zero padding beyond the last frame
// (modules.py:139-143), here it would read the previous layer's output at the surplus frames (bias, ReLU, ...).  So the
// surplus columns of every activation of the non-causal stacks (and of their gradients in backward) are set to zero; the
// batch's own maximum is a DEVICE scalar (a replayed graph has no host in the loop), the host promises only an upper
// bound for the number of surplus columns.  rows * max_tail * words threads, consecutive threads on consecutive words.
// ------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void zero_tail_kernel(uint32_t* __restrict__ x, int64_t rows, int T, int words,
                                                        const int32_t* __restrict__ t_valid, int mult, int max_tail) {
  const int64_t per_row = (int64_t)max_tail * words;
  const int64_t n = rows * per_row;
  int64_t tv64 = (int64_t)t_valid[0] * mult;
  const int tv = tv64 < 0 ? 0 : (tv64 > T ? T : (int)tv64);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / per_row;
    const int64_t k = i - row * per_row;            // word inside the row's last max_tail columns
    const int col = T - max_tail + (int)(k / words);
    if (col >= tv && col >= 0) x[(row * T + col) * words + (k % words)] = 0u;
  }
}
}  // namespace
extern "C" int dv3_zero_tail_b32(void* x, int64_t rows, int32_t T, int32_t words, const int32_t* t_valid, int32_t mult,
                                 int32_t max_tail, void* stream) {
  DV3_REQUIRE(x && t_valid && rows >= 0 && T > 0 && words > 0 && mult > 0 && max_tail >= 0, "zero_tail: bad args");
  if (max_tail > T) max_tail = T;
  if (rows == 0 || max_tail == 0) return DV3_OK;
  int64_t blocks = (rows * max_tail * words + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(zero_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t*)x, rows, (int)T,
                     (int)words, t_valid, (int)mult, (int)max_tail);
  return dv3_check_launch("zero_tail_b32");
}
