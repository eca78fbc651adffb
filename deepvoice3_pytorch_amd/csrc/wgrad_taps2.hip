// Split-bf16 weight-gradient GEMM, all taps per workgroup, operands fetched TWO steps ahead (autograd of F.conv1d
// w.r.t. its weight; reference call sites deepvoice3_pytorch/modules.py:153,216 through loss.backward(), train.py:755):
//
//   out[s][j][m][c] = sum_{(b,t) in slab s} g[b][m][t] * xd[b][c][t + j*dil - padL]
//
// Same tile, LDS images, arithmetic and accumulation order as wgrad_taps_kernel (wgrad_gemm_bf16x3.hip) -- the slabs
// agree bit for bit.  That kernel fetches step s+1's operands right before the MFMAs of step s and converts them right
// after: both operands stream from HBM, and one MFMA phase (~1 us) does not cover the fetch.  Here
//   * two register sets: the set converted in step s was fetched in step s-2 and is refilled for step s+2 at once;
//   * every load is unconditional and of one form (two 16-byte loads per unit from an offset clamped into the tensor; a
//     unit that would start outside it -- the first / last rows only -- is re-aligned in registers), so the compiler
//     counts its vmcnt waits exactly instead of draining the queue;
//   * all eight waves stay in phase (one barrier per step): the two waves of a SIMD hide each other's LDS latency and
//     barrier skew; a ping-pong schedule of the same kernel measured slower (profiles/r03_wgrad_pingpong_experiment.md).
#include "common.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

namespace {

struct WgradT2Args {
  dv3_wgrad_desc d;
  int m_tiles, c_tiles;
};

constexpr int BKT = 32, KB = 4, PAD = 2;
constexpr int JT = 3, BM = 128, BN = 128, NT = 512;
constexpr int LDM = BM + PAD, LDN = BN + PAD;
constexpr int GBUF = 2 * KB * LDM, XTAP = 2 * KB * LDN, BUF = GBUF + JT * XTAP;

__device__ __forceinline__ void wt2_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const f32x2 f = {v[i], v[i + 1]};
    const bf16x2 h = __builtin_convertvector(f, bf16x2);
    const f32x2 r = f - __builtin_convertvector(h, f32x2);
    const bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi[i] = h[0]; hi[i + 1] = h[1];
    lo[i] = l[0]; lo[i + 1] = l[1];
  }
}
template <typename T>
__device__ __forceinline__ T wt2_ldg(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
// bit e set <=> 0 <= t + e < len
__device__ __forceinline__ uint32_t wt2_valid8(int t, int len) {
  const int elo = max(0, -t), ehi = min(8, len - t);
  return ehi > elo ? (((1u << ehi) - 1u) & ~((1u << elo) - 1u)) : 0u;
}
__device__ __forceinline__ void wt2_mask8(float (&v)[8], uint32_t m) {
#pragma unroll
  for (int e = 0; e < 8; ++e)
    v[e] = __uint_as_float(__float_as_uint(v[e]) & (uint32_t)__builtin_amdgcn_sbfe((int)m, e, 1));
}
// 8 consecutive floats at element offset `off`: ALWAYS two 16-byte loads from the offset clamped into the tensor
// (`total` elements); returns the clamp distance (off - clamped), non-zero only for units at the tensor's two ends.
__device__ __forceinline__ int wt2_load8(const float* __restrict__ base, int off, int total, float (&v)[8]) {
  const int offc = min(max(off, 0), total - 8);
  const f32x4u a = wt2_ldg<f32x4u>(base, (uint32_t)offc * 4u);
  const f32x4u b = wt2_ldg<f32x4u>(base, (uint32_t)offc * 4u + 16u);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  return off - offc;
}
// element e of the unit is element e + sh of what was loaded (sh != 0: tensor ends, rare); missing elements become 0
// (they lie outside the tensor, where the validity mask is 0 anyway)
__device__ __forceinline__ void wt2_realign(float (&v)[8], int sh) {
  // opaque inside the (rare) branch: otherwise the 64 lane masks (e + sh == q) are hoisted in front of the branch and
  // computed for every unit of every step (measured: 366 us per launch instead of 203)
  asm volatile("" : "+v"(sh));
  float w[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float x = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) x = (e + sh == q) ? v[q] : x;
    w[e] = x;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = w[e];
}

// GP (round 6): g holds PAIR WORDS (include/dv3hip.h: dv3_wgrad_desc.g_pair) -- its unit is staged with eight v_perm_b32
// instead of the fp32 -> bf16-pair conversion (the validity masks and the re-alignment act on whole words either way)
//
// DIL (round 6, "one window for the three taps"): 0 = the form above (per tap: two 16-byte loads of the 8 shifted
// elements, their keep-bits, one fp32 -> pair conversion -- 24 conversions per x unit, three quarters of the vector work
// of a step, which is what keeps this kernel at a third of its roof: all eight waves convert at the same time and the
// matrix pipe waits).  1 / 3 = the dilation is this compile-time constant (every layer of the three presets whose grid
// matters: the converter's and the decoder's d = 1 and 3 layers): the three taps of a unit read the overlapping windows
// [t - padL + j d, + 8), so ONE window of 8 + 2 d elements is fetched (three / four 16-byte loads instead of six),
// masked (one pair of keep-bit words instead of three) and converted ONCE (10 / 14 conversions instead of 24), and the
// three shifted units are cut out of the converted window: dwords as they are for an even element shift, one
// v_alignbit_b32 per dword for an odd one.  The same pair per element as the per-tap form: bit-identical slabs.
template <int N>
__device__ __forceinline__ void wt2_realign_n(float (&v)[N], int sh) {
  asm volatile("" : "+v"(sh));        // (see wt2_realign: keep the select chains inside the rare branch)
  float w[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    float x = 0.f;
#pragma unroll
    for (int q = 0; q < N; ++q) x = (e + sh == q) ? v[q] : x;
    w[e] = x;
  }
#pragma unroll
  for (int e = 0; e < N; ++e) v[e] = w[e];
}
// bit e set <=> 0 <= t + e < len, e < n
__device__ __forceinline__ uint32_t wt2_valid_n(int t, int len, int n) {
  const int elo = max(0, -t), ehi = min(n, len - t);
  return ehi > elo ? (((1u << ehi) - 1u) & ~((1u << elo) - 1u)) : 0u;
}
// IL (window forms): the staging of the NEXT step's tile -- masking, conversion, cutting, LDS stores: ~100 vector
// instructions that touch nothing the current step's MFMAs read -- is issued BETWEEN those MFMAs
// (__builtin_amdgcn_sched_group_barrier: a few vector instructions behind each matrix instruction) instead of after them.
// All eight waves run in phase here, so "after" means every wave converts while the matrix pipes of the CU idle; the
// matrix pipe takes one instruction per 32 cycles from a wave, the conversion rides in the issue slots it leaves.  What
// makes it possible: the window form's register set (186-202 against the per-tap form's 214-230; round 5's attempt needed
// ~246 and spilled).  The rare paths that branch (re-alignment at the tensor's two ends, the validity mask of a g unit
// in a row tail) run BEFORE the interleaved region -- a branch would cut the scheduling region -- and the x window's
// validity / keep mask is applied unconditionally (it is two instructions per element).  Same values: bit-identical slabs.
template <bool MASK, int ABL = 0, bool GP = false, int DIL = 0, bool IL = false>
__global__ __launch_bounds__(NT) void wgrad_taps2_kernel(const WgradT2Args args) {
  static_assert(DIL == 0 || DIL == 1 || DIL == 3, "window form: d = 1 or 3");
  static_assert(!IL || (DIL > 0 && ABL == 0), "interleaved staging: the window forms");
  constexpr int WLEN = 8 + 2 * DIL;                 // elements of the window of one unit
  constexpr int NW4 = (WLEN + 3) / 4;               // 16-byte loads per window
  constexpr int WL = DIL > 0 ? 4 * NW4 : 1;
  const dv3_wgrad_desc& p = args.d;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_t2[];
  bf16x8* const smem = reinterpret_cast<bf16x8*>(smem_raw_t2);       // [2 buffers][G hi, G lo | 3 x (X hi, X lo)]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, lhi = lane >> 5;

  int pid = dv3_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = pid % args.m_tiles; pid /= args.m_tiles;
  const int ct = pid % args.c_tiles;
  const int s = pid / args.c_tiles;
  const int m0 = mt * BM, c0 = ct * BN;
  const int T = p.T, Tin = p.Tin, M = p.M, Cin = p.Cin;

  const int urow = tid >> 2, uk8 = tid & 3;
  // the two register sets hold RAW fetched data only (32 + 6 registers each); validity masks, clamp distances and the
  // dropout word alignment are recomputed from the step index when the set is converted
  float rg[2][8], rx[2][DIL > 0 ? 1 : JT][DIL > 0 ? 1 : 8];
  float rw[2][WL];                                  // window form: the raw window of each register set
  uint32_t mlo[MASK ? 2 : 1][DIL > 0 ? 1 : JT], mhi[MASK ? 2 : 1][DIL > 0 ? 1 : JT];
  const int n_tc = (T + BKT - 1) / BKT;
  int nsteps, step0 = 0;
  {
    const int total = p.B * n_tc, q = (total + p.n_slabs - 1) / p.n_slabs;
    step0 = s * q;
    nsteps = max(0, min(q, total - step0));
  }
  const int g_total = (p.B - 1) * (int)p.g_bs + (M - 1) * (int)p.g_rs + T;
  const int x_total = (p.B - 1) * (int)p.x_bs + (Cin - 1) * (int)p.x_rs + Tin;
  const int wl = (Tin + 31) / 32 - 1;
  const int gm = m0 + urow, xc = c0 + urow;
  const bool grow_ok = gm < M, xrow_ok = xc < Cin;
  const int grow_off = (grow_ok ? gm : M - 1) * (int)p.g_rs + uk8 * 8;
  const int xcc = xrow_ok ? xc : Cin - 1;
  const int xrow_off = xcc * (int)p.x_rs + uk8 * 8;
  const int xm_off = xcc * p.xmask_rs;

  auto load_step = [&](int step, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    const int gs = step0 + min(step, nsteps - 1);          // past the end: re-fetch the last step (never used)
    const int b = gs / n_tc, tc = gs - b * n_tc;
    const int t0 = tc * BKT;
    const int gb = b * (int)p.g_bs + t0, xb = b * (int)p.x_bs + t0;
    (void)wt2_load8(p.g, gb + grow_off, g_total, rg[S]);
    if constexpr (DIL > 0) {
      const int woff = xb + xrow_off - p.padL;
      const int offc = min(max(woff, 0), x_total - WL);
#pragma unroll
      for (int q = 0; q < NW4; ++q) {
        const f32x4u a = wt2_ldg<f32x4u>(p.x, (uint32_t)(offc + 4 * q) * 4u);
        rw[S][4 * q] = a[0]; rw[S][4 * q + 1] = a[1]; rw[S][4 * q + 2] = a[2]; rw[S][4 * q + 3] = a[3];
      }
      if constexpr (MASK) {
        const int tx = t0 + uk8 * 8 - p.padL;
        const uint32_t mo = (uint32_t)(b * Cin * p.xmask_rs + xm_off);
        const int w0 = min(max(tx, 0) >> 5, wl);
        mlo[S][0] = wt2_ldg<uint32_t>(p.xmask, (mo + (uint32_t)w0) * 4u);
        mhi[S][0] = wt2_ldg<uint32_t>(p.xmask, (mo + (uint32_t)min(w0 + 1, wl)) * 4u);
      }
    } else
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int shift = j * p.dil - p.padL;
      (void)wt2_load8(p.x, xb + xrow_off + shift, x_total, rx[S][j]);
      if constexpr (MASK) {
        const int tx = t0 + uk8 * 8 + shift;
        const uint32_t mo = (uint32_t)(b * Cin * p.xmask_rs + xm_off);
        const int w0 = min(max(tx, 0) >> 5, wl);
        mlo[S][j] = wt2_ldg<uint32_t>(p.xmask, (mo + (uint32_t)w0) * 4u);
        mhi[S][j] = wt2_ldg<uint32_t>(p.xmask, (mo + (uint32_t)min(w0 + 1, wl)) * 4u);
      }
    }
  };
  auto write_step = [&](int step, int buf, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    bf16x8* dst = smem + buf * BUF;
    const int gs = step0 + min(step, nsteps - 1);
    const int b = gs / n_tc, tc = gs - b * n_tc;
    const int t0 = tc * BKT;
    const int gb = b * (int)p.g_bs + t0, xb = b * (int)p.x_bs + t0;
    {
      const int off = gb + grow_off;
      const int sh = off - min(max(off, 0), g_total - 8);
      uint32_t vm = 0xffu;
      if (t0 + BKT > T) vm = wt2_valid8(t0 + uk8 * 8, T);
      if (!grow_ok) vm = 0u;
      if (__any(sh != 0)) wt2_realign(rg[S], sh);
      if (__any(vm != 0xffu)) wt2_mask8(rg[S], vm);
      bf16x8 hi, lo;
      if constexpr (GP) {
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(rg[S][e]);
        dv3_pair_units(w, hi, lo);
      } else wt2_split8(rg[S], hi, lo);
      const int o = uk8 * LDM + urow;
      dst[o] = hi;
      dst[KB * LDM + o] = lo;
    }
    if constexpr (DIL > 0) {
      const int woff = xb + xrow_off - p.padL;
      const int sh = woff - min(max(woff, 0), x_total - WL);
      const int tx = t0 + uk8 * 8 - p.padL;                          // time of the window's element 0
      constexpr uint32_t FULL = (1u << WLEN) - 1u;
      uint32_t xm = FULL;
      const bool x_edge = t0 - p.padL < 0 || t0 + BKT - p.padL + 2 * DIL > Tin;   // (uniform)
      if (x_edge) xm = wt2_valid_n(tx, Tin, WLEN);
      if (!xrow_ok) xm = 0u;
      if constexpr (MASK) {
        uint32_t bits = __builtin_amdgcn_alignbit(mhi[S][0], mlo[S][0], (uint32_t)(max(tx, 0) & 31));
        if (x_edge && tx < 0) bits = (-tx < 32) ? bits << (-tx) : 0u;
        xm &= bits;
      }
      if (__any(sh != 0)) wt2_realign_n<WL>(rw[S], sh);
      if (MASK || __any(xm != FULL)) {
#pragma unroll
        for (int e = 0; e < WLEN; ++e)
          rw[S][e] = __uint_as_float(__float_as_uint(rw[S][e]) & (uint32_t)__builtin_amdgcn_sbfe((int)xm, e, 1));
      }
      // the window as bf16 pairs, two elements per dword and plane
      uint32_t hw[WLEN / 2], lw[WLEN / 2];
#pragma unroll
      for (int k = 0; k < WLEN / 2; ++k) {
        const f32x2 f = {rw[S][2 * k], rw[S][2 * k + 1]};
        const bf16x2 h = __builtin_convertvector(f, bf16x2);
        const f32x2 r = f - __builtin_convertvector(h, f32x2);
        const bf16x2 l = __builtin_convertvector(r, bf16x2);
        hw[k] = __builtin_bit_cast(uint32_t, h);
        lw[k] = __builtin_bit_cast(uint32_t, l);
      }
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int e0 = j * DIL;                                      // element of the window the tap's unit starts at
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        u32x4_ h4, l4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if ((e0 & 1) == 0) {
            h4[i] = hw[e0 / 2 + i];
            l4[i] = lw[e0 / 2 + i];
          } else {
            h4[i] = __builtin_amdgcn_alignbit(hw[(e0 + 1) / 2 + i], hw[(e0 - 1) / 2 + i], 16u);
            l4[i] = __builtin_amdgcn_alignbit(lw[(e0 + 1) / 2 + i], lw[(e0 - 1) / 2 + i], 16u);
          }
        }
        bf16x8* dx = dst + GBUF + j * XTAP;
        const int o = uk8 * LDN + urow;
        dx[o] = __builtin_bit_cast(bf16x8, h4);
        dx[KB * LDN + o] = __builtin_bit_cast(bf16x8, l4);
      }
    } else
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      bf16x8* dx = dst + GBUF + j * XTAP;
      const int shift = j * p.dil - p.padL;
      const int off = xb + xrow_off + shift;
      const int sh = off - min(max(off, 0), x_total - 8);
      const int tx = t0 + uk8 * 8 + shift;
      const bool x_edge = t0 + shift < 0 || t0 + BKT + shift > Tin;
      uint32_t xm = 0xffu;
      if (x_edge) xm = wt2_valid8(tx, Tin);
      if (!xrow_ok) xm = 0u;
      if constexpr (MASK) {
        uint32_t bits = __builtin_amdgcn_alignbit(mhi[S][j], mlo[S][j], (uint32_t)(max(tx, 0) & 31));
        if (x_edge && tx < 0) bits = (-tx < 32) ? bits << (-tx) : 0u;
        xm &= bits;
      }
      if (__any(sh != 0)) wt2_realign(rx[S][j], sh);
      if (MASK || __any(xm != 0xffu)) wt2_mask8(rx[S][j], xm);
      bf16x8 hi, lo;
      wt2_split8(rx[S][j], hi, lo);
      const int o = uk8 * LDN + urow;
      dx[o] = hi;
      dx[KB * LDN + o] = lo;
    }
  };

  f32x16 acc[JT][2];
#pragma unroll
  for (int j = 0; j < JT; ++j)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][a][r] = 0.f;

  auto mfma_step = [&](int cur) __attribute__((always_inline)) {
    const bf16x8* GsH = smem + cur * BUF;
    const bf16x8* GsL = GsH + KB * LDM;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1 && ABL != 6) __builtin_amdgcn_sched_barrier(0);   // one k16 block's ten fragments live at a time
      const int k8 = 2 * ks + lhi;
      const int ai = k8 * LDM + wm * 64 + l31;
      const bf16x8 ah0 = GsH[ai], ah1 = GsH[ai + 32];
      const bf16x8 al0 = GsL[ai], al1 = GsL[ai + 32];
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const bf16x8* XsH = GsH + GBUF + j * XTAP;
        const int xi = k8 * LDN + wc * 32 + l31;
        const bf16x8 bh = XsH[xi];
        const bf16x8 bl = XsH[KB * LDN + xi];
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh, acc[j][1], 0, 0, 0);
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl, acc[j][1], 0, 0, 0);
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[j][1], 0, 0, 0);
      }
    }
  };
  // ---- IL: the step's pieces as separate lambdas ----
  uint32_t xrow_bits = xrow_ok ? 0xffffffffu : 0u;
  asm volatile("" : "+v"(xrow_bits));          // opaque: keeps the compiler from turning the AND back into a branch on xrow_ok
  auto step_coords = [&](int step, int& t0, int& gb, int& xb) __attribute__((always_inline)) {
    const int gs = step0 + min(step, nsteps - 1);
    const int b = gs / n_tc, tc = gs - b * n_tc;
    t0 = tc * BKT;
    gb = b * (int)p.g_bs + t0;
    xb = b * (int)p.x_bs + t0;
  };
  // rare, branching fix-ups of register set S (the tile of `step`): before the interleaved region
  auto fix_step = [&](int step, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    int t0, gb, xb;
    step_coords(step, t0, gb, xb);
    {
      const int off = gb + grow_off;
      const int sh = off - min(max(off, 0), g_total - 8);
      uint32_t vm = 0xffu;
      if (t0 + BKT > T) vm = wt2_valid8(t0 + uk8 * 8, T);
      if (!grow_ok) vm = 0u;
      if (__any(sh != 0)) wt2_realign(rg[S], sh);
      if (__any(vm != 0xffu)) wt2_mask8(rg[S], vm);
    }
    if constexpr (DIL > 0) {
      const int woff = xb + xrow_off - p.padL;
      const int sh = woff - min(max(woff, 0), x_total - WL);
      if (__any(sh != 0)) wt2_realign_n<WL>(rw[S], sh);
    }
  };
  // vector part 1: the g unit's pair, the x window masked and converted (hw / lw: the window as bf16 pairs)
  auto conv_part1 = [&](int step, auto set_c, bf16x8& ghi, bf16x8& glo, uint32_t (&hw)[WLEN / 2], uint32_t (&lw)[WLEN / 2]) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    int t0, gb, xb;
    step_coords(step, t0, gb, xb);
    if constexpr (GP) {
      uint32_t w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(rg[S][e]);
      dv3_pair_units(w, ghi, glo);
    } else wt2_split8(rg[S], ghi, glo);
    const int tx = t0 + uk8 * 8 - p.padL;
    uint32_t xm = wt2_valid_n(tx, Tin, WLEN) & xrow_bits;      // (no control flow: a branch would cut the scheduling region)
    if constexpr (MASK) {
      uint32_t bits = __builtin_amdgcn_alignbit(mhi[S][0], mlo[S][0], (uint32_t)(max(tx, 0) & 31));
      const uint32_t shifted = (-tx < 32) ? bits << ((-tx) & 31) : 0u;
      bits = tx < 0 ? shifted : bits;
      xm &= bits;
    }
#pragma unroll
    for (int k = 0; k < WLEN / 2; ++k) {
      const float a0 = __uint_as_float(__float_as_uint(rw[S][2 * k]) & (uint32_t)__builtin_amdgcn_sbfe((int)xm, 2 * k, 1));
      const float a1 = __uint_as_float(__float_as_uint(rw[S][2 * k + 1]) & (uint32_t)__builtin_amdgcn_sbfe((int)xm, 2 * k + 1, 1));
      const f32x2 f = {a0, a1};
      const bf16x2 h = __builtin_convertvector(f, bf16x2);
      const f32x2 r = f - __builtin_convertvector(h, f32x2);
      const bf16x2 l = __builtin_convertvector(r, bf16x2);
      hw[k] = __builtin_bit_cast(uint32_t, h);
      lw[k] = __builtin_bit_cast(uint32_t, l);
    }
  };
  // vector part 2 + the eight LDS stores
  auto conv_part2 = [&](int buf, const bf16x8& ghi, const bf16x8& glo, const uint32_t (&hw)[WLEN / 2], const uint32_t (&lw)[WLEN / 2]) __attribute__((always_inline)) {
    bf16x8* dst = smem + buf * BUF;
    {
      const int o = uk8 * LDM + urow;
      dst[o] = ghi;
      dst[KB * LDM + o] = glo;
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      constexpr int dummy = 0; (void)dummy;
      const int e0 = j * DIL;
      typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
      u32x4_ h4, l4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((e0 & 1) == 0) {
          h4[i] = hw[e0 / 2 + i];
          l4[i] = lw[e0 / 2 + i];
        } else {
          h4[i] = __builtin_amdgcn_alignbit(hw[(e0 + 1) / 2 + i], hw[(e0 - 1) / 2 + i], 16u);
          l4[i] = __builtin_amdgcn_alignbit(lw[(e0 + 1) / 2 + i], lw[(e0 - 1) / 2 + i], 16u);
        }
      }
      bf16x8* dx = dst + GBUF + j * XTAP;
      const int o = uk8 * LDN + urow;
      dx[o] = __builtin_bit_cast(bf16x8, h4);
      dx[KB * LDN + o] = __builtin_bit_cast(bf16x8, l4);
    }
  };
  // one k16 block of the current tile: ten fragment reads, eighteen MFMAs
  auto mfma_ks = [&](int cur, auto ks_c) __attribute__((always_inline)) {
    constexpr int ks = decltype(ks_c)::value;
    const bf16x8* GsH = smem + cur * BUF;
    const bf16x8* GsL = GsH + KB * LDM;
    const int k8 = 2 * ks + lhi;
    const int ai = k8 * LDM + wm * 64 + l31;
    const bf16x8 ah0 = GsH[ai], ah1 = GsH[ai + 32];
    const bf16x8 al0 = GsL[ai], al1 = GsL[ai + 32];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const bf16x8* XsH = GsH + GBUF + j * XTAP;
      const int xi = k8 * LDN + wc * 32 + l31;
      const bf16x8 bh = XsH[xi];
      const bf16x8 bl = XsH[KB * LDN + xi];
      acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh, acc[j][0], 0, 0, 0);
      acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh, acc[j][1], 0, 0, 0);
      acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl, acc[j][0], 0, 0, 0);
      acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl, acc[j][1], 0, 0, 0);
      acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[j][0], 0, 0, 0);
      acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[j][1], 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  if constexpr (IL) {
    if (nsteps > 0) {
      load_step(0, S0{});
      load_step(1, S1{});
      write_step(0, 0, S0{});
      load_step(2, S0{});
      __syncthreads();
      auto step = [&](int st, auto set_c) __attribute__((always_inline)) {
        fix_step(st + 1, set_c);
        bf16x8 ghi, glo;
        uint32_t hw[WLEN / 2], lw[WLEN / 2];
        __builtin_amdgcn_sched_barrier(0);
        mfma_ks(st & 1, std::integral_constant<int, 0>{});
        conv_part1(st + 1, set_c, ghi, glo, hw, lw);
        // (the address arithmetic of the fragment reads and the reads themselves head the pipeline: left unconstrained,
        //  the solver hands those vector instructions to the first MFMA's group, finds no valid order and drops the lot)
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int m = 0; m < 18; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // five VALU
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_ks(st & 1, std::integral_constant<int, 1>{});
        conv_part2((st + 1) & 1, ghi, glo, hw, lw);
#pragma unroll
        for (int m = 0; m < 10; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // one DS write
        }
        __builtin_amdgcn_sched_barrier(0);
        load_step(st + 3, set_c);
        __syncthreads();
      };
      for (int st = 0; st < nsteps; st += 2) {
        step(st, S1{});
        if (st + 1 < nsteps) step(st + 1, S0{});
      }
    }
  } else
  if (nsteps > 0) {
    // ---- prologue: tile 0 -> buffer 0 (all waves); set 1 <- step 1, set 0 <- step 2 ----
    load_step(0, S0{});
    load_step(1, S1{});
    write_step(0, 0, S0{});
    load_step(2, S0{});
    __syncthreads();

    // steps in pairs so the register-set index is static: step st + 1 is staged from set (st + 1) & 1
    auto step = [&](int st, auto set_c) __attribute__((always_inline)) {
      if (ABL != 1) mfma_step(st & 1);
      if (ABL != 2) {
        write_step(st + 1, (st + 1) & 1, set_c);      // past the end: a re-fetched tile into the buffer nobody reads
        load_step(st + 3, set_c);
      }
      __syncthreads();
    };
    for (int st = 0; st < nsteps; st += 2) {
      step(st, S1{});
      if (st + 1 < nsteps) step(st + 1, S0{});
    }
  }

  const float oscale = MASK ? p.drop_scale : 1.0f;
  const int c = c0 + wc * 32 + l31;
  if (c < Cin) {
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      float* __restrict__ ob = p.out + (int64_t)s * p.out_ss + (int64_t)j * M * p.ldo;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (m < M) ob[(int64_t)m * p.ldo + c] = acc[j][mi][r] * oscale;
        }
    }
  }
}

template <bool MASK, int ABL = 0, bool GP = false, int DIL = 0, bool IL = false>
int launch_wgrad_taps2(const WgradT2Args& a, int64_t nb, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * BUF * 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_taps2_kernel<MASK, ABL, GP, DIL, IL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      dv3_set_error("wgrad_taps2: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_taps2_kernel<MASK, ABL, GP, DIL, IL>), dim3((unsigned)nb), dim3(NT), lds, st, a);
  return dv3_check_launch("wgrad_taps2");
}

}  // namespace

int g_wgrad_t2_il = 1;       // dv3_debug_set(48, v): the window forms stage the next tile between the MFMAs (0 = after them)
int g_wgrad_t2_window = 1;   // dv3_debug_set(47, v): the one-window-for-three-taps form of d = 1 / 3 launches (0 = the per-tap form)
int g_wgrad_t2_abl = 0;   // dv3_debug_set(16, v): timing-only ablations (1 no MFMAs, 2 no staging, 6 k16 blocks not pinned apart)

// three taps, three-term split, K split over contiguous ranges (called by dv3_wgrad_gemm_bf16x3_dispatch)
int dv3_wgrad_taps2_dispatch(const dv3_wgrad_desc* d, hipStream_t st) {
  WgradT2Args a;
  a.d = *d;
  a.m_tiles = dv3_cdiv(d->M, BM);
  a.c_tiles = dv3_cdiv(d->Cin, BN);
  const int64_t nb = (int64_t)a.m_tiles * a.c_tiles * d->n_slabs;
  DV3_REQUIRE(nb < (1ll << 31), "wgrad_gemm: grid too large");
  g_dv3_last_wgrad = 3000 + 40;
#ifdef DV3_EXPERIMENTS
  if (g_wgrad_t2_abl && !d->xmask) {
    switch (g_wgrad_t2_abl) {
      case 1: return launch_wgrad_taps2<false, 1>(a, nb, st);
      case 2: return launch_wgrad_taps2<false, 2>(a, nb, st);
      case 6: return launch_wgrad_taps2<false, 6>(a, nb, st);
    }
  }
#endif
  // window form (kernel template DIL): d = 1 or 3, a tensor long enough for the window's loads
  const int win = (g_wgrad_t2_window && (d->dil == 1 || d->dil == 3) &&
                   (int64_t)(d->B - 1) * d->x_bs + (int64_t)(d->Cin - 1) * d->x_rs + d->Tin >= 16) ? d->dil : 0;
  const bool il = win && g_wgrad_t2_il;
  g_dv3_last_wgrad += (d->g_pair ? 1 : 0) + (win ? 2 : 0) + (il ? 4 : 0);   // ...41 pair-word g, 42 window form, 43 both; +4 interleaved staging
#define DV3_T2(M, G) \
  (win == 1 ? (il ? launch_wgrad_taps2<M, 0, G, 1, true>(a, nb, st) : launch_wgrad_taps2<M, 0, G, 1>(a, nb, st)) \
   : win == 3 ? (il ? launch_wgrad_taps2<M, 0, G, 3, true>(a, nb, st) : launch_wgrad_taps2<M, 0, G, 3>(a, nb, st)) \
   : launch_wgrad_taps2<M, 0, G, 0>(a, nb, st))
  if (d->g_pair) return d->xmask ? DV3_T2(true, true) : DV3_T2(false, true);
  return d->xmask ? DV3_T2(true, false) : DV3_T2(false, false);
#undef DV3_T2
}
