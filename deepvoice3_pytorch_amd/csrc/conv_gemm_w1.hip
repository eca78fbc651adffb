// Split-operand tap-GEMM, ONE WAVE PER SIMD (experimental, opt-in: dv3_debug_set(12, 1) or tile_hint 30).
// Same operation, operands, arithmetic and accumulation order as conv_gemm_bf16x3.hip (the dilated 1-D convolution
// + fused Conv1dGLU / HighwayConv1d / DGRAD tail; reference semantics deepvoice3_pytorch/modules.py:145-164,
// 205-226), so its results are bit-identical to that kernel's; only the execution structure differs.
//
// Why (profiles/r02x_cu_phase_ubench.md): with two waves per SIMD taking turns (ping-pong), the LOAD phase next to a
// computing partner takes 1.8-1.9k cycles per 768 MFMA cycles -- the matrix pipe is 40-56 % busy however the phases
// are arranged -- while a wave hides its OWN fragment reads completely when they are issued between its MFMAs one
// phase ahead (48 MFMAs + 16 ds_read_b128 = 1546 cycles; a barrier per 48 MFMAs costs 3 %).  So:
//   * 4 waves per workgroup, one workgroup per CU, 512 registers per wave: a 128 x 128 per-wave tile (MI = 2 row
//     pairs x NI = 4 column sub-tiles; 256 accumulator registers in AccVGPRs).  A k16 block is 48 MFMAs fed by 16
//     fragment reads -- half the LDS reads per MFMA of the 64 x 64 tiles, the energy lever of DESIGN.md section 3.2.
//   * fragments are double-buffered in registers: while the 48 MFMAs of one k16 block issue, the 16 reads of the
//     next block (across step boundaries) are in flight.
//   * block tile 256 rows (128 `a` + 128 gate) x 256 columns; LDS: weight panel [2 buffers][hi|lo][4 k8][256]
//     (64 KB), activation tile [2 buffers][hi|lo][4 k8][256 + halo] (80 KB).
//   * one K step = (32-channel chunk, tap); its first half also stores the next step's weight panel and a third of the
//     next chunk's activation tile (fetched one full step earlier: ~3k cycles of latency cover) and issues the
//     following fetches; ONE barrier per step, between the halves:
//        half 0 of step s: store A(s+1), X(c+1) part j | fetch A(s+2), X part of the next step | read frags (s, q=1)
//                          | MFMA (s, q=0) | barrier
//        half 1 of step s: read frags (s+1, q=0) | MFMA (s, q=1)
//     A(s+1) overwrites A(s-1), last read before the barrier of step s-1; it is first read in half 1 of step s, after
//     this step's barrier.  X(c+1) overwrites X(c-1) (dead since chunk c began) and is complete before the barrier of
//     the chunk's last step, i.e. before the first prefetch that reads it.
// Three taps only (the models' kernel size); other shapes keep the two-waves-per-SIMD kernels.
#include "conv_common.h"
#include <math.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int KB = 4;          // k8 blocks per 32-channel chunk
constexpr int BKC = 32;
constexpr int HALO_MAX = 64;

__device__ __forceinline__ void w1_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
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
__device__ __forceinline__ void w1_split8_f16(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  f16x8 h8, l8;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const f32x2 f = {__builtin_amdgcn_fmed3f(v[i], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[i + 1], -65504.f, 65504.f)};
    const f16x2 h = __builtin_convertvector(f, f16x2);
    const f32x2 r = f - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    h8[i] = h[0]; h8[i + 1] = h[1];
    l8[i] = l[0]; l8[i + 1] = l[1];
  }
  hi = __builtin_bit_cast(bf16x8, h8);
  lo = __builtin_bit_cast(bf16x8, l8);
}
template <bool F16>
__device__ __forceinline__ f32x16 w1_mma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ T w1_ldg(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

constexpr int W1_MI = 2, W1_NI = 4;
struct W1Frags {
  bf16x8 ah[W1_MI][2], al[W1_MI][2], bh[W1_NI], bl[W1_NI];
};

// ABL: timing-only ablations (dv3_debug_set(13, v); results are wrong): 1 no epilogue, 2 no global fetches in the
// steady state, 3 no conversions / LDS stores, 4 no fragment reads, 5 MFMAs only (2 + 3 + 4), 6 no barrier
template <bool MASK, bool F16, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv_w1_kernel(const ConvArgs args) {
  constexpr int WN = 2, MI = W1_MI, NI = W1_NI, JT = 3;
  constexpr int BM = 2 * MI * 64, BMH = 2 * MI * 32, BN = WN * NI * 32, NT = 256;
  constexpr int AU = KB * BM / NT;                                  // 4 weight units per plane per thread per step
  constexpr int XI = (KB * (BN + HALO_MAX) + NT - 1) / NT;          // 5 activation items per thread per chunk
  constexpr int XG = (XI + JT - 1) / JT;                            // 2 of them staged per step
  const dv3_conv_desc& p = args.d;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w1[];
  const int dil = p.dil;
  const int BNH = BN + (JT - 1) * dil;
  bf16x8* const As = reinterpret_cast<bf16x8*>(smem_w1);            // [2][hi|lo][KB][BM]
  bf16x8* const Xs = As + 2 * 2 * KB * BM;                          // [2][hi|lo][KB][BNH]
  const int xbuf = 2 * KB * BNH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int pid = dv3_xcd_remap(blockIdx.x, args.n_blocks);
  const int mt = pid % args.m_tiles;
  const int n0 = (pid / args.m_tiles) * BN;

  const bool gated = (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY);
  int h0b, h1b;
  if (gated) { h0b = mt * BMH; h1b = p.a_half + mt * BMH; }
  else { h0b = mt * BM; h1b = mt * BM + BMH; }

  const int Cin = p.Cin, T = p.Tout, lda = p.lda;
  const int Ntot = p.B * T;
  const int k8_total = args.kp >> 3;
  const bf16x8* __restrict__ Wh = reinterpret_cast<const bf16x8*>(p.a_split);
  const int64_t wplane = (int64_t)JT * k8_total * lda;
  const uint32_t* __restrict__ xmask = p.xmask;
  const float xscale = F16 ? (float)(1 << DV3_F16_ACT_SHIFT) : 1.0f;
  const float dscale = p.drop_scale * xscale;
  const int nchunks = (Cin + BKC - 1) / BKC;
  const int nsteps = nchunks * JT;

  // ---- per-lane column validity per tap (the conv's zero padding at sequence edges) ----
  uint32_t vbits = 0;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
    const int bc = n / T, tc = n - bc * T;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int ts = tc + j * dil - p.padL;
      if (n < Ntot && ts >= 0 && ts < T) vbits |= 1u << (j * NI + ni);
    }
  }

  // ---- staging constants ----
  // Activation tile [KB][BNH] items: thread t stages column t of each of the KB channel blocks (items 0..KB-1: one
  // offset + a uniform stride) and, for the halo columns BN .. BNH-1 (at most 64 x KB = NT items), one more item:
  // channel block t / 64, column BN + t % 64.  Two offsets, two mask offsets and two bit positions per thread
  // instead of five each (register budget: 128 fragment + 48-64 staging registers are live across a step).
  static_assert(XI == KB + 1 && HALO_MAX * KB == NT, "item mapping below");
  const int n_halo = BNH - BN;                   // 2 * dil
  const uint32_t x_rsb = (uint32_t)p.x_rs * 4u, m_rsb = (uint32_t)p.xmask_rs * 4u;
  uint32_t xo0, xo1, mo0 = 0, mo1 = 0, sh0 = 0, sh1 = 0;
  const int hk8 = tid >> 6, hq = BN + (tid & 63);      // the halo item of this thread
  const bool halo_ok = (tid & 63) < n_halo;
  {
    auto col = [&](int q, uint32_t& xo, uint32_t& mo, uint32_t& sh) {
      const int f = n0 - p.padL + q;
      int bf = 0, tf = 0;
      if (f >= 0 && f < Ntot) {
        bf = f / T;
        tf = f - bf * T;
      }
      xo = ((uint32_t)bf * (uint32_t)p.x_bs + (uint32_t)tf) * 4u;
      mo = ((uint32_t)(bf * Cin) * (uint32_t)p.xmask_rs + (uint32_t)(tf >> 5)) * 4u;
      sh = (uint32_t)(tf & 31);
    };
    col(tid, xo0, mo0, sh0);
    col(halo_ok ? hq : 0, xo1, mo1, sh1);
    xo1 += (uint32_t)hk8 * 8u * x_rsb;
    mo1 += (uint32_t)hk8 * 8u * m_rsb;
  }
  // weight panel: unit u of this thread is row k8 = u, column tid of the block's BM = NT columns
  static_assert(BM == NT, "one panel column per thread");
  uint32_t aoff0;
  {
    const bool hi_half = tid >= BMH;
    const int gcol = (hi_half ? h1b : h0b) + (tid - (hi_half ? BMH : 0));
    aoff0 = (uint32_t)(gcol < lda ? gcol : 0) * 16u;
  }
  const uint32_t a_rs = (uint32_t)lda * 16u;

  bf16x8 ra[2][AU];
  float rx[XG][8];
  uint32_t rm[MASK ? XG : 1][8];

  auto load_A = [&](int c, int j) {
    const bf16x8* src = Wh + (int64_t)(j * k8_total + c * KB) * lda;
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      ra[0][u] = w1_ldg<bf16x8>(src + (int64_t)u * lda, aoff0);
      ra[1][u] = w1_ldg<bf16x8>(src + wplane + (int64_t)u * lda, aoff0);
    }
  };
  auto write_A = [&](int buf) {
    bf16x8* dst = As + buf * (2 * KB * BM);
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      dst[tid + u * NT] = ra[0][u];
      dst[KB * BM + tid + u * NT] = ra[1][u];
    }
  };
  // part G of chunk c: items [G*XG, G*XG + XG) -- items 0..KB-1 are (channel block i, column tid), item KB the halo one
  auto load_X = [&](int c, auto gc) {
    constexpr int G = decltype(gc)::value;
    // Cin % 32 == 0 (dispatch): every chunk is whole -- uniform row bases + loop-invariant per-thread offsets
    const char* xb = reinterpret_cast<const char*>(p.x) + (int64_t)c * BKC * x_rsb;
    const char* mb = reinterpret_cast<const char*>(xmask) + (int64_t)c * BKC * m_rsb;
    // eight uniform row bases (SGPR pairs) + one 32-bit per-thread offset per item: more distinct bases would not fit
    // the scalar registers and come back as spilled 64-bit vector addresses
    uint32_t xo[XG], mo[XG];
#pragma unroll
    for (int s = 0; s < XG; ++s) {
      const int i = G * XG + s;
      xo[s] = i < KB ? xo0 + (uint32_t)(i * 8) * x_rsb : xo1;
      mo[s] = i < KB ? mo0 + (uint32_t)(i * 8) * m_rsb : mo1;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int s = 0; s < XG; ++s) {
        const int i = G * XG + s;
        if (i >= XI) continue;
        rx[s][e] = w1_ldg<float>(xb + (int64_t)e * x_rsb, xo[s]);
        if (MASK) rm[s][e] = w1_ldg<uint32_t>(mb + (int64_t)e * m_rsb, mo[s]);
      }
    }
  };
  auto write_X = [&](int buf, auto gc) {
    constexpr int G = decltype(gc)::value;
    bf16x8* dst = Xs + buf * xbuf;
#pragma unroll
    for (int s = 0; s < XG; ++s) {
      const int i = G * XG + s;
      if (i >= XI) continue;
      const bool live = i < KB || halo_ok;
      const int idx = i < KB ? i * BNH + tid : hk8 * BNH + hq;
      const uint32_t sh = i < KB ? sh0 : sh1;
      if (live) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = rx[s][e];
          if (MASK) v[e] *= ((rm[s][e] >> sh) & 1u) ? dscale : 0.f;
          else if (F16) v[e] *= xscale;
        }
        bf16x8 hi, lo;
        if constexpr (F16) w1_split8_f16(v, hi, lo); else w1_split8(v, hi, lo);
        dst[idx] = hi;
        dst[KB * BNH + idx] = lo;
      }
    }
  };

  const int a_off = wm * (MI * 32) + l31;
  const int x_off = wn * (NI * 32) + l31;
  auto read_frags = [&](W1Frags& f, int abuf, int xb, int j, int q) {
    const bf16x8* AsH = As + abuf * (2 * KB * BM);
    const bf16x8* AsL = AsH + KB * BM;
    const bf16x8* XsH = Xs + xb * xbuf;
    const bf16x8* XsL = XsH + KB * BNH;
    const int k8 = 2 * q + lhi;
    const int ai = k8 * BM + a_off;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      f.ah[mi][0] = AsH[ai + mi * 32];
      f.ah[mi][1] = AsH[ai + mi * 32 + BMH];
      f.al[mi][0] = AsL[ai + mi * 32];
      f.al[mi][1] = AsL[ai + mi * 32 + BMH];
    }
    const int xi = k8 * BNH + x_off + j * dil;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      f.bh[ni] = XsH[xi + ni * 32];
      f.bl[ni] = XsL[xi + ni * 32];
    }
  };
  // the conv's zero padding at sequence edges: columns whose tap-j input lies outside their own batch item.  Always
  // applied (32 selects per k16 block, in the MFMAs' shadow): a wave-uniform test would split the scheduling region.
  auto fix_frags = [&](W1Frags& f, int j) {
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const bool ok = (vbits >> (j * NI + ni)) & 1u;
      f.bh[ni] = ok ? f.bh[ni] : zero8;
      f.bl[ni] = ok ? f.bl[ni] : zero8;
    }
  };
  // two accumulator arrays (one per row pair): a single 1 KB private array is left in scratch memory by the compiler
  static_assert(MI == 2, "accA / accB below");
  f32x16 accA[2][NI], accB[2][NI];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[h][ni][r] = accB[h][ni][r] = 0.f;
  // per accumulator: lo*hi, hi*lo, hi*hi (the order of the other split kernels); issued term-major, so an accumulator
  // is revisited 16 MFMAs later
  auto mfma_half = [&](const W1Frags& f) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        accA[h][ni] = w1_mma<F16>(f.al[0][h], f.bh[ni], accA[h][ni]);
        accB[h][ni] = w1_mma<F16>(f.al[1][h], f.bh[ni], accB[h][ni]);
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        accA[h][ni] = w1_mma<F16>(f.ah[0][h], f.bl[ni], accA[h][ni]);
        accB[h][ni] = w1_mma<F16>(f.ah[1][h], f.bl[ni], accB[h][ni]);
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        accA[h][ni] = w1_mma<F16>(f.ah[0][h], f.bh[ni], accA[h][ni]);
        accB[h][ni] = w1_mma<F16>(f.ah[1][h], f.bh[ni], accB[h][ni]);
      }
  };

  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, 1>;
  using G2 = std::integral_constant<int, 2>;

  // ---- prologue: A(0), the whole of X(0); then A(1) and part 0 of X(1) into the staging registers ----
  load_A(0, 0);
  write_A(0);
  load_X(0, G0{}); write_X(0, G0{});
  load_X(0, G1{}); write_X(0, G1{});
  load_X(0, G2{}); write_X(0, G2{});
  if (nsteps > 1) load_A(0, 1);
  if (nchunks > 1) load_X(1, G0{});
  __syncthreads();
  W1Frags F0, F1;
  read_frags(F0, 0, 0, 0, 0);

  // one step; the tap is a compile-time constant so the activation part to store / fetch is static.  STEADY: every
  // guard below is known true (all but the last two chunks) -- a step is then two straight-line blocks.
  auto step = [&](int s, int c, auto jc, auto steady_c) {
    constexpr int j = decltype(jc)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    constexpr int jn = (j + 1) % JT;
    const int cn = c + (j == JT - 1 ? 1 : 0);        // chunk of step s + 1
    // the 256 accumulator registers live in AccVGPRs
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        asm volatile("" : "+a"(accA[h][ni]));
        asm volatile("" : "+a"(accB[h][ni]));
      }
    // ---------------- half 0 ----------------
    constexpr bool kStore = !(STEADY && (ABL == 3 || ABL == 5)), kFetch = !(STEADY && (ABL == 2 || ABL == 5));
    constexpr bool kRead = !(STEADY && (ABL == 4 || ABL == 5));
    if (kStore && (STEADY || s + 1 < nsteps)) write_A((s + 1) & 1);
    if (kStore && (STEADY || c + 1 < nchunks)) write_X((c + 1) & 1, std::integral_constant<int, j>{});
    __builtin_amdgcn_sched_barrier(0);               // refill the staging registers only after they were stored
    if (kFetch && (STEADY || s + 2 < nsteps)) {      // step s + 2 = (c2, j2)
      constexpr int j2 = (j + 2) % JT;
      const int c2 = c + (j + 2) / JT;
      load_A(c2, j2);
    }
    if (kFetch && (STEADY || cn + 1 < nchunks)) load_X(cn + 1, std::integral_constant<int, jn>{});   // stored by step s + 1
    if (kRead) read_frags(F1, s & 1, c & 1, j, 1);
    __builtin_amdgcn_sched_barrier(0);               // the reads are issued BEFORE the MFMAs they overlap with
    fix_frags(F0, j);
    mfma_half(F0);
    if (!(STEADY && (ABL == 6 || ABL == 5))) __syncthreads();
    // ---------------- half 1 ----------------
    if (kRead && (STEADY || s + 1 < nsteps)) read_frags(F0, (s + 1) & 1, cn & 1, jn, 0);
    __builtin_amdgcn_sched_barrier(0);
    fix_frags(F1, j);
    mfma_half(F1);
  };
  using TrueT = std::integral_constant<bool, true>;
  using FalseT = std::integral_constant<bool, false>;
  int c = 0;
  for (; c + 2 < nchunks; ++c) {
    step(c * JT, c, G0{}, TrueT{});
    step(c * JT + 1, c, G1{}, TrueT{});
    step(c * JT + 2, c, G2{}, TrueT{});
  }
  for (; c < nchunks; ++c) {
    step(c * JT, c, G0{}, FalseT{});
    step(c * JT + 1, c, G1{}, FalseT{});
    step(c * JT + 2, c, G2{}, FalseT{});
  }

  if constexpr (F16) {
    constexpr float kInv = 1.0f / (float)(1 << (DV3_F16_WEIGHT_SHIFT + DV3_F16_ACT_SHIFT));
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[h][ni][r] *= kInv; accB[h][ni][r] *= kInv; }
  }
  int bcol[NI], tcol[NI];
  bool okc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
    okc[ni] = n < Ntot;
    bcol[ni] = n / T;
    tcol[ni] = n - bcol[ni] * T;
  }
  if (ABL != 1 || accA[0][0][0] + accB[1][NI - 1][7] == 1.2345e30f) {
    // the shared epilogue on 64-row x 64-column pieces (its two-column-sub-tile instantiation, the one the other split
    // kernels use): the four-sub-tile form keeps 64 residual loads + 32 bias values per lane live and spills around
    // every load batch (measured: 75 of 255 us per launch)
    auto piece = [&](f32x16 (&acc)[2][NI], int row0, int nh) {
      f32x16 t[2][2];
      int bc[2], tc[2];
      bool ok[2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) t[h][q] = acc[h][nh * 2 + q];
#pragma unroll
      for (int q = 0; q < 2; ++q) { bc[q] = bcol[nh * 2 + q]; tc[q] = tcol[nh * 2 + q]; ok[q] = okc[nh * 2 + q]; }
      conv_epilogue<BM, BMH, 2, 0, false>(p, t, gated, mt, row0, lhi, bc, tc, ok);
    };
    piece(accA, wm * (MI * 32), 0);
    piece(accA, wm * (MI * 32), 1);
    piece(accB, wm * (MI * 32) + 32, 0);
    piece(accB, wm * (MI * 32) + 32, 1);
  }
}

template <int ABL>
int launch_w1_abl(const ConvArgs& a, size_t lds, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)conv_w1_kernel<false, true, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((conv_w1_kernel<false, true, ABL>), dim3(a.n_blocks), dim3(256), lds, st, a);
  return dv3_check_launch("conv_gemm_w1(abl)");
}

template <bool MASK, bool F16>
int launch_w1(const ConvArgs& a, size_t lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_w1_kernel<MASK, F16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) {
      dv3_set_error("conv_gemm_w1: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_w1_kernel<MASK, F16>), dim3(a.n_blocks), dim3(256), lds, st, a);
  return dv3_check_launch("conv_gemm_w1");
}

}  // namespace

int g_w1_abl = 0;   // dv3_debug_set(13, v)
// called by dv3_conv_gemm_bf16x3_dispatch (conv_gemm_bf16x3.hip) when the one-wave-per-SIMD kernel is asked for;
// returns 1 when the shape is not eligible (the caller goes on to the other tiles), else a DV3_* code
int dv3_conv_gemm_w1_dispatch(const dv3_conv_desc* d, hipStream_t st) {
  const bool gated = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  if (d->J != 3 || d->split_terms == 1 || (d->J - 1) * d->dil > HALO_MAX || (d->Cin & 31)) return 1;
  if (d->io_bf16 || d->x_planes) return 1;
  const int BM = 256, BMH = 128, BN = 256;
  const int BNH = BN + 2 * d->dil;
  const size_t lds = (size_t)(2 * 2 * KB * BM + 2 * 2 * KB * BNH) * 16;
  if (lds > 160 * 1024) return 1;
  ConvArgs a;
  a.d = *d;
  a.a_scalar = 0;
  a.kp = (d->Cin + 31) / 32 * 32;
  a.m_tiles = gated ? dv3_cdiv(d->Cg, BMH) : dv3_cdiv(d->M, BM);
  a.n_tiles = (int)dv3_cdiv64((int64_t)d->B * d->Tout, BN);
  const int64_t nb = (int64_t)a.m_tiles * a.n_tiles;
  DV3_REQUIRE(nb < (1ll << 31), "conv_gemm: grid too large");
  a.n_blocks = (int)nb;
  a.stagger = 0;
  g_dv3_last_conv = (d->split_terms == DV3_SPLIT_F16X3 ? 5000 : 3000) + 300;
  if (g_w1_abl && d->split_terms == DV3_SPLIT_F16X3 && !d->xmask) {
    switch (g_w1_abl) {
      case 1: return launch_w1_abl<1>(a, lds, st);
      case 2: return launch_w1_abl<2>(a, lds, st);
      case 3: return launch_w1_abl<3>(a, lds, st);
      case 4: return launch_w1_abl<4>(a, lds, st);
      case 5: return launch_w1_abl<5>(a, lds, st);
      case 6: return launch_w1_abl<6>(a, lds, st);
    }
  }
  if (d->split_terms == DV3_SPLIT_F16X3) return d->xmask ? launch_w1<true, true>(a, lds, st) : launch_w1<false, true>(a, lds, st);
  return d->xmask ? launch_w1<true, false>(a, lds, st) : launch_w1<false, false>(a, lds, st);
}
