// Persistent "planes" tap-GEMM: the dilated 1-D convolution + fused Conv1dGLU / HighwayConv1d / DGRAD tail of
// conv_gemm_bf16x3.hip (reference semantics: deepvoice3_pytorch/modules.py:145-164, 205-226, and its autograd),
// for the case where BOTH operands arrive already split into 16-bit operand planes:
//   weights      dv3_split_pack_bf16 / dv3_weight_norm_split_pack_bf16 image  [plane][j][k8][m][8]
//   activations  dv3_split_planes_f32 layout (or a producing epilogue)        [plane][b][c8][t][8]
// so staging is plain 16-byte copies (no conversion, no dropout work: the producer applied the keep-bits).
//
// What differs from the ping-pong kernel, and why:
//   * persistent workgroups: the grid is sized to the chip (two 4-wave workgroups per CU for the 128-column
//     tiles, one 8-wave workgroup for 128x256) and every workgroup walks tiles  t = r*G + remap(w).  The first
//     operand fetches of the NEXT tile are issued before the CURRENT tile's epilogue, so the tail's HBM traffic
//     (residual reads, y / pre-gate stores) overlaps them instead of a cold prologue after a fresh dispatch.
//   * the second co-resident workgroup of a CU starts half a tile late (args.stagger), so one workgroup's tail
//     runs beside the other's main loop instead of both idling the matrix pipes together.
//   * fragments are double-buffered in REGISTERS at half-step (k16) granularity: while the 12 MFMAs of one k16
//     block issue, the 8 fragment reads of the next block are in flight, across step boundaries.  LDS tiles are
//     double-buffered (weights per step, activations per chunk); with every LDS write placed in the second half
//     of a step one barrier per step (in its middle) orders all reads and writes:
//        step s, first half : read frags (s, q=1) | MFMA (s, q=0)                          | barrier
//        step s, second half: write A(s+2), X(chunk of s+2 if it starts there) ; fetch A(s+3), X(chunk of s+3 ...)
//                             read frags (s+1, q=0) | MFMA (s, q=1)
//     A(s) is read in the second half of step s-1 and the first half of step s; A(s+2) overwrites it after the
//     barrier of step s; it is first read in the second half of step s+1, after that step's barrier.
//   * same accumulator layout and accumulation order as the other tap-GEMM kernels: the shared epilogue
//     (conv_common.h) is used verbatim.
#include "conv_common.h"
#include <math.h>
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

int dv3_conv_c8pp_dispatch(const dv3_conv_desc* d, hipStream_t st);   // conv_c8pp.hip

namespace {

constexpr int KB = 4;          // k8 blocks per 32-channel chunk
constexpr int HALO_MAX = 64;   // (J-1)*dil supported by the register staging (model max: 2*27)

template <bool F16>
__device__ __forceinline__ f32x16 mma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ T ldg_off(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// c8 input with dropout: zero the dropped channels of one unit (bit e of the keep-byte = channel e)
__device__ __forceinline__ bf16x8 keep8(const bf16x8& v, uint32_t m) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 d = __builtin_bit_cast(u32x4, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)m, 2 * i, 1) & 0xffffu;
    const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)m, 2 * i + 1, 1) << 16;
    d[i] &= (lo | hi);
  }
  return __builtin_bit_cast(bf16x8, d);
}

template <int NI>
struct Frags {
  bf16x8 ah[2], al[2], bh[NI], bl[NI];
};

// ABL: timing-only ablations (dv3_debug_set(6, v); results are wrong): 1 no LDS stores in the steady state,
// 2 no fragment reads in the steady state, 3 no MFMAs, 4 no epilogue, 5 no global fetches in the steady state,
// 6 no barriers in the steady state
// compile-time loop over the taps of one chunk
template <int JT, typename F>
__device__ __forceinline__ void steady_chunk(int, F&& f) {
  f(std::integral_constant<int, 0>{});
  if constexpr (JT > 1) f(std::integral_constant<int, 1>{});
  if constexpr (JT > 2) f(std::integral_constant<int, 2>{});
  static_assert(JT <= 3, "unrolled tap counts: 1..3");
}
// Issue-order request for the straight-line block just emitted: NM MFMAs with the block's NW LDS stores, NV
// global fetches and NR fragment reads spread one (or a few) per MFMA gap, stores first (their data has been
// in registers for a step), then the fetches that refill those registers, then the reads.
template <int NM, int NW, int NV, int NR>
__device__ __forceinline__ void interleave() {
  constexpr int NO = NW + NV + NR;
  constexpr int PER = NM > 0 ? (NO + NM - 1) / NM : NO;   // non-MFMA issues per gap
  int emitted = 0;
  (void)emitted;
#define DV3_SGB_SLOT(K)                                                                          \
  if constexpr ((K) < NM) {                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
    constexpr int lo = (K) * PER, hi = ((K) + 1) * PER < NO ? ((K) + 1) * PER : NO;              \
    constexpr int w = (hi > lo ? ((hi < NW ? hi : NW) - (lo < NW ? lo : NW)) : 0);               \
    constexpr int v = (hi > lo ? ((hi < NW + NV ? hi : NW + NV) - (lo < NW + NV ? lo : NW + NV)) - w : 0); \
    constexpr int r = (hi > lo ? (hi - lo) - w - v : 0);                                         \
    if constexpr (w > 0) __builtin_amdgcn_sched_group_barrier(0x200, w, 0);                      \
    if constexpr (v > 0) __builtin_amdgcn_sched_group_barrier(0x020, v, 0);                      \
    if constexpr (r > 0) __builtin_amdgcn_sched_group_barrier(0x100, r, 0);                      \
  }
  DV3_SGB_SLOT(0) DV3_SGB_SLOT(1) DV3_SGB_SLOT(2) DV3_SGB_SLOT(3) DV3_SGB_SLOT(4) DV3_SGB_SLOT(5)
  DV3_SGB_SLOT(6) DV3_SGB_SLOT(7) DV3_SGB_SLOT(8) DV3_SGB_SLOT(9) DV3_SGB_SLOT(10) DV3_SGB_SLOT(11)
#undef DV3_SGB_SLOT
}

template <int WM, int WN, int NI, int TERMS, bool F16, int ABL = 0, int JT = 0>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv_planes_kernel(const ConvArgs args) {
  static_assert(!F16 || TERMS == 3, "the fp16 form is the three-term split");
  constexpr int BM = WM * 64, BMH = WM * 32, BN = WN * NI * 32, NT = WM * WN * 64;
  constexpr int PL = TERMS == 3 ? 2 : 1;                    // operand planes in use
  constexpr int AU = KB * BM / NT;                          // A units per plane per thread per step
  constexpr int XI = (KB * (BN + HALO_MAX) + NT - 1) / NT;  // X units per plane per thread per chunk
  static_assert(KB * BM % NT == 0, "A panel must split evenly");
  const dv3_conv_desc& p = args.d;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int J = p.J, dil = p.dil;
  const int BNH = BN + (J - 1) * dil;
  bf16x8* const As = reinterpret_cast<bf16x8*>(smem_raw);   // [2 buffers][PL][KB][BM]
  // [2 buffers][PL][XPS]: a plane holds [KB][BNH] units and is padded to XPS = XI * NT, so that every thread
  // stores all of its XI staged units unconditionally (no exec-masked branch in the steady-state block)
  constexpr int XPS = XI * NT;
  bf16x8* const Xs = As + 2 * PL * KB * BM;
  constexpr int xbuf = PL * XPS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  const bool gated = (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY);
  const int T = p.Tout, lda = p.lda, B = p.B;
  const int Ntot = B * T;
  const int k8_total = args.kp >> 3;          // == p.x_c8p
  const int nchunks = args.kp >> 5;
  const int nsteps = nchunks * J;
  const bf16x8* __restrict__ Wh = reinterpret_cast<const bf16x8*>(p.a_split);
  const int64_t wplane = (int64_t)J * k8_total * lda;       // 16-byte units per weight plane
  const bf16x8* __restrict__ XP = reinterpret_cast<const bf16x8*>(p.x_planes);
  const int64_t xplane = (int64_t)B * k8_total * T;         // 16-byte units per activation plane
  const int n_items = KB * BNH;
  const int a_off = wm * 32 + l31;
  const int x_off = wn * (NI * 32) + l31;

  // the second co-resident workgroup of each CU starts late (blocks are dealt round-robin over the CUs:
  // block b and block b + #CUs share a CU -- a placement used for speed only)
  if (args.stagger > 0 && ((blockIdx.x / args.a_scalar) & 1)) {
    for (int i = 0; i < args.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  uint32_t xoff[XI];   // byte offset of unit (b, k8, t) inside a plane, chunk 0
  uint32_t aoff[AU];   // byte offset of this thread's weight units inside a (tap, chunk) panel row
  bf16x8 ra[PL][AU], rx[PL][XI];
  // bf16 storage (single-term kernel): the c8 input may carry dropout keep-bytes, one per unit (uniform pointer)
  const uint8_t* __restrict__ const xkeep = TERMS == 1 ? p.xmask_c8 : nullptr;
  uint32_t rk[TERMS == 1 ? XI : 1];

  auto tile_offsets = [&](int mt, int n0) {
    int h0b, h1b;
    if (gated) { h0b = mt * BMH; h1b = p.a_half + mt * BMH; }
    else { h0b = mt * BM; h1b = mt * BM + BMH; }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int idx = tid + i * NT;
      const int k8 = idx / BNH, q = idx - k8 * BNH;
      const int f = n0 - p.padL + q;
      int bf = 0, tf = 0;
      if (idx < n_items && f >= 0 && f < Ntot) {
        bf = f / T;
        tf = f - bf * T;
      }
      const int k8c = k8 < KB ? k8 : 0;
      xoff[i] = (((uint32_t)bf * (uint32_t)k8_total + (uint32_t)k8c) * (uint32_t)T + (uint32_t)tf) * 16u;
    }
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const int idx = tid + u * NT;  // k8 * BM + col
      const int col = idx % BM, k8 = idx / BM;
      const bool hi_half = col >= BMH;
      const int gcol = (hi_half ? h1b : h0b) + (col - (hi_half ? BMH : 0));
      aoff[u] = (uint32_t)(k8 * lda + (gcol < lda ? gcol : 0)) * 16u;
    }
  };
  auto load_A = [&](int c, int j) {
    const bf16x8* src = Wh + (int64_t)(j * k8_total + c * KB) * lda;  // uniform
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int u = 0; u < AU; ++u) ra[pl][u] = ldg_off<bf16x8>(src + pl * wplane, aoff[u]);
  };
  auto write_A = [&](int buf) {
    bf16x8* dst = As + buf * (PL * KB * BM);
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int u = 0; u < AU; ++u) dst[pl * KB * BM + tid + u * NT] = ra[pl][u];
  };
  auto load_X = [&](int c) {
    const bf16x8* src = XP + (int64_t)c * KB * T;                     // uniform: chunk c = 4 k8 blocks further
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int i = 0; i < XI; ++i) rx[pl][i] = ldg_off<bf16x8>(src + pl * xplane, xoff[i]);
    if (TERMS == 1 && xkeep) {
      const uint8_t* mk = xkeep + (int64_t)c * KB * T;
#pragma unroll
      for (int i = 0; i < XI; ++i) rk[TERMS == 1 ? i : 0] = ldg_off<uint8_t>(mk, xoff[i] >> 4);
    }
  };
  auto write_X = [&](int buf) {
    bf16x8* dst = Xs + buf * xbuf;
    if (TERMS == 1 && xkeep) {
#pragma unroll
      for (int i = 0; i < XI; ++i) rx[0][i] = keep8(rx[0][i], rk[TERMS == 1 ? i : 0]);
    }
#pragma unroll
    for (int i = 0; i < XI; ++i)
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) dst[pl * XPS + tid + i * NT] = rx[pl][i];
  };

  uint32_t vbits = 0, need = 0;
  auto read_frags = [&](Frags<NI>& f, int abuf, int c, int j, int q) {
    const bf16x8* AsH = As + abuf * (PL * KB * BM);
    const bf16x8* XsH = Xs + (c & 1) * xbuf;
    const int k8 = 2 * q + lhi;
    const int ai = k8 * BM + a_off;
    f.ah[0] = AsH[ai];
    f.ah[1] = AsH[ai + BMH];
    if (TERMS == 3) {
      f.al[0] = AsH[KB * BM + ai];
      f.al[1] = AsH[KB * BM + ai + BMH];
    }
    const int xi = k8 * BNH + x_off + j * dil;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      f.bh[ni] = XsH[xi + ni * 32];
      if (TERMS == 3) f.bl[ni] = XsH[XPS + xi + ni * 32];
    }
  };
  // the conv's zero padding at sequence edges: columns whose tap-j input lies outside their own batch item
  auto fix_frags = [&](Frags<NI>& f, int j) {
    if ((need >> j) & 1u) {
      const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const bool ok = (vbits >> (j * NI + ni)) & 1u;
        f.bh[ni] = ok ? f.bh[ni] : zero8;
        if (TERMS == 3) f.bl[ni] = ok ? f.bl[ni] : zero8;
      }
    }
  };
  // the same without the wave-uniform test (straight-line steady state: 2 selects per MFMA, in its shadow)
  auto fix_frags_nb = [&](Frags<NI>& f, int j) {
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const bool ok = (vbits >> (j * NI + ni)) & 1u;
      f.bh[ni] = ok ? f.bh[ni] : zero8;
      if (TERMS == 3) f.bl[ni] = ok ? f.bl[ni] : zero8;
    }
  };
  f32x16 acc[2][NI];
  auto mfma = [&](const Frags<NI>& f) {
    if (TERMS == 3) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        acc[0][ni] = mma16<F16>(f.al[0], f.bh[ni], acc[0][ni]);
        acc[1][ni] = mma16<F16>(f.al[1], f.bh[ni], acc[1][ni]);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        acc[0][ni] = mma16<F16>(f.ah[0], f.bl[ni], acc[0][ni]);
        acc[1][ni] = mma16<F16>(f.ah[1], f.bl[ni], acc[1][ni]);
      }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      acc[0][ni] = mma16<F16>(f.ah[0], f.bh[ni], acc[0][ni]);
      acc[1][ni] = mma16<F16>(f.ah[1], f.bh[ni], acc[1][ni]);
    }
  };

  const int G = gridDim.x;
  int tile = dv3_xcd_remap(blockIdx.x, G);
  if (tile >= args.n_blocks) return;
  {
    const int mt = tile % args.m_tiles, nt = tile / args.m_tiles;
    tile_offsets(mt, nt * BN);
    load_A(0, 0);
    load_X(0);
  }
  Frags<NI> F0, F1;
  while (true) {
    const int mt = tile % args.m_tiles;
    const int n0 = (tile / args.m_tiles) * BN;
    // ---- this lane's output columns: per-tap validity of the shifted read ----
    vbits = 0;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
      const int bc = n / T, tc = n - bc * T;
      for (int j = 0; j < J; ++j) {
        const int ts = tc + j * dil - p.padL;
        if (n < Ntot && ts >= 0 && ts < T) vbits |= 1u << (j * NI + ni);
      }
    }
    need = 0;
    for (int j = 0; j < J; ++j) {
      const uint32_t all = ((1u << NI) - 1u) << (j * NI);
      if (!__all((vbits & all) == all)) need |= 1u << j;
    }
    need = __builtin_amdgcn_readfirstlane(need);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][ni][r] = 0.f;

    // cursors: W = step s+2 (its panel is written this iteration), L = step s+3 (fetched), R = step s+1
    // (fragments read), C = step s (computed); each a (chunk, tap) pair.  Fragments are edge-fixed right
    // before their MFMAs (tap jC for both k16 blocks of step s).
    auto generic_steps = [&](int s_begin, int s_end) {
      int cC, jC, cR, jR, cW, jW, cL, jL;
      {
        const int sc = s_begin < 0 ? 0 : s_begin, sr = s_begin + 1 < 0 ? 0 : s_begin + 1;
        cC = sc / J; jC = sc - cC * J;
        cR = sr / J; jR = sr - cR * J;
        cW = (s_begin + 2) / J; jW = (s_begin + 2) - cW * J;
        cL = (s_begin + 3) / J; jL = (s_begin + 3) - cL * J;
      }
      for (int s = s_begin; s < s_end; ++s) {
        if (ABL == 9) {   // keep the accumulators in the AccVGPR half of the register file
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+a"(acc[h][ni]));
        }
        if (s >= 0) {
          // ---------------- first half: k16 block 0 of step s ----------------
          if (ABL != 2 && ABL != 8) read_frags(F1, s & 1, cC, jC, 1);
          fix_frags(F0, jC);
          if (ABL != 3) mfma(F0);
          if (ABL != 6 && ABL != 8) __syncthreads();
        }
        // ---------------- second half ----------------
        if (s + 2 < nsteps && ((ABL != 1 && ABL != 8) || s < 0)) {
          write_A((s + 2) & 1);
          if (jW == 0) write_X(cW & 1);
        }
        if (s + 3 < nsteps && ((ABL != 5 && ABL != 8) || s < 0)) {
          load_A(cL, jL);
          if (jL == 0) load_X(cL);
        }
        const bool rd = s + 1 >= 0 && s + 1 < nsteps;
        if (rd && ((ABL != 2 && ABL != 8) || s < 1)) read_frags(F0, (s + 1) & 1, cR, jR, 0);
        if (s >= 0) {
          fix_frags(F1, jC);
          if (ABL != 3) mfma(F1);
        }
        if (s < 0) __syncthreads();     // warm-up iterations have no first half
        cC = cR; jC = jR;
        cR = cW; jR = jW;
        cW = cL; jW = jL;
        if (++jL == J) { jL = 0; ++cL; }
      }
    };
    // Steady state for the tap counts the models use (JT = 3 or 1): the J steps of a chunk unrolled, every
    // guard of the generic loop known true, taps static -- one straight-line block per half step, so the
    // fragment reads, panel stores and fetches can be spread over the MFMA gaps (sched_group_barrier) instead
    // of being issued as a burst in front of them (eight waves bursting 8 reads each stall every wave's MFMA
    // issue behind its own LDS issue: measured +25 us per launch at the north-star shape).
    int s_done = -2;
    if constexpr (JT > 0 && ABL == 0) {
      const int ci_lo = JT == 1 ? 2 : 1, ci_hi = nchunks - 2;       // inclusive
      if (J == JT && ci_hi >= ci_lo) {
        generic_steps(-2, ci_lo * JT - 2);
        for (int ci = ci_lo; ci <= ci_hi; ++ci) {
          steady_chunk<JT>(ci, [&](auto jjc) {
            constexpr int jj = decltype(jjc)::value;
            constexpr int jL = (jj + 1) % JT;
            constexpr int jC = (jj + 2 * JT - 2) % JT;
            constexpr int jR = (jj + JT - 1) % JT;
            const int sp = ci * JT + jj;                        // the W step; s = sp - 2
            const int cC = ci - (jj < 2 ? (JT == 1 ? 2 : 1) : 0);
            const int cR = ci - (jj < 1 ? 1 : 0);
            const int cL = ci + (jj + 1 == JT ? 1 : 0);
            // ---- first half ----
            read_frags(F1, sp & 1, cC, jC, 1);
            fix_frags_nb(F0, jC);
            mfma(F0);
            interleave<TERMS * 2 * NI, 0, 0, (2 + NI) * PL>();
            __syncthreads();
            // ---- second half ----
            write_A(sp & 1);
            if constexpr (jj == 0) write_X(ci & 1);
            load_A(cL, jL);
            if constexpr (jL == 0) load_X(cL);
            read_frags(F0, (sp - 1) & 1, cR, jR, 0);
            fix_frags_nb(F1, jC);
            mfma(F1);
            interleave<TERMS * 2 * NI, AU * PL + (jj == 0 ? XI * PL : 0), AU * PL + (jL == 0 ? XI * PL : 0),
                       (2 + NI) * PL>();
          });
        }
        s_done = (ci_hi + 1) * JT - 2;
      }
    }
    generic_steps(s_done, nsteps);

    // ---- next tile's first fetches, then this tile's epilogue ----
    const int next = tile + G;
    const bool has_next = next < args.n_blocks;
    if (has_next) {
      tile_offsets(next % args.m_tiles, (next / args.m_tiles) * BN);
      load_A(0, 0);
      load_X(0);
    }
    if constexpr (F16) {   // the accumulators carry 2^(weight shift + activation shift) x the result
      constexpr float kInv = 1.0f / (float)(1 << (DV3_F16_WEIGHT_SHIFT + DV3_F16_ACT_SHIFT));
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[h][ni][r] *= kInv;
    }
    {
      int bcol[NI], tcol[NI];
      bool okc[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
        okc[ni] = n < Ntot;
        bcol[ni] = n / T;
        tcol[ni] = n - bcol[ni] * T;
      }
      if (TERMS == 1 && xkeep) {      // x * keep / (1-p): the 1/(1-p) of a masked c8 input, exact on the accumulators
        const float ds = p.drop_scale;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][ni][r] *= ds;
      }
      if ((ABL != 4 && ABL != 8) || acc[0][0][0] + acc[1][NI - 1][7] == 1.2345e30f) {
        if (TERMS == 1 && (p.io_bf16 & DV3_IO_OUT_C8))
          conv_epilogue_c8<BM, BMH, NI>(p, acc, gated, mt, wm * 32, lhi, bcol, tcol, okc);
        else
          conv_epilogue<BM, BMH, NI, 0, TERMS == 1>(p, acc, gated, mt, wm * 32, lhi, bcol, tcol, okc);
      }
    }
    if (!has_next) break;
    tile = next;
  }
}

// ---- activation planes from an fp32 BCT tensor (+ dropout keep-bits) ----
__device__ __forceinline__ void split8_bf16(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
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
// one thread per 16-byte unit (b, c8, t); t fastest: the 8 channel reads are row-coalesced, the two unit stores
// are 16-byte coalesced
__global__ __launch_bounds__(256) void split_planes_kernel(const dv3_planes_desc p, int c8p, uint32_t* range_ctr) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c8 = blockIdx.y, b = blockIdx.z;
  if (t >= p.T) return;
  const float scale = p.scale * (p.dtype == DV3_SPLIT_DTYPE_F16 ? (float)(1 << DV3_F16_ACT_SHIFT) : 1.0f);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = c8 * 8 + e;
    float x = 0.f;
    if (ch < p.C) {
      x = p.x[(int64_t)b * p.x_bs + (int64_t)ch * p.x_rs + t];
      bool keep = true;
      if (p.mask) keep = (p.mask[((int64_t)b * p.C + ch) * p.mask_rs + (t >> 5)] >> (t & 31)) & 1u;
      x = keep ? x * scale : 0.f;
    }
    v[e] = x;
  }
  bf16x8 hi, lo;
  if (p.dtype == DV3_SPLIT_DTYPE_F16) dv3_note_range(range_ctr, dv3_split8_f16(v, hi, lo)); else split8_bf16(v, hi, lo);
  bf16x8* out = reinterpret_cast<bf16x8*>(p.out);
  const int64_t u = ((int64_t)b * c8p + c8) * p.T + t;
  out[u] = hi;
  out[(int64_t)p.B * c8p * p.T + u] = lo;
}

// ---- fp32 (B,C,T) <-> c8 (bf16 [B][C8][T][8]) and keep-bits -> keep-bytes: one thread per unit (b, group, t), t fastest
__global__ __launch_bounds__(256) void to_c8_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_rs,
                                                    bf16x8* __restrict__ out, int C, int T, int c8p) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  bf16x8 u;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const int ch = g * 8 + e;
    const f32x2 f = {ch < C ? x[(int64_t)b * x_bs + (int64_t)ch * x_rs + t] : 0.f,
                     ch + 1 < C ? x[(int64_t)b * x_bs + (int64_t)(ch + 1) * x_rs + t] : 0.f};
    const bf16x2 h = __builtin_convertvector(f, bf16x2);
    u[e] = h[0]; u[e + 1] = h[1];
  }
  out[((int64_t)b * c8p + g) * T + t] = u;
}
__global__ __launch_bounds__(256) void from_c8_kernel(const bf16x8* __restrict__ x, float* __restrict__ out,
                                                      int64_t out_bs, int64_t out_rs, int C, int T, int c8p) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const bf16x8 u = x[((int64_t)b * c8p + g) * T + t];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = g * 8 + e;
    if (ch < C) out[(int64_t)b * out_bs + (int64_t)ch * out_rs + t] = (float)u[e];
  }
}
__global__ __launch_bounds__(256) void mask_bits_to_c8_kernel(const uint32_t* __restrict__ bits, int rs,
                                                              uint8_t* __restrict__ out, int C, int T, int c8p) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  uint32_t m = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = g * 8 + e;
    if (ch < C) m |= ((bits[((int64_t)b * C + ch) * rs + (t >> 5)] >> (t & 31)) & 1u) << e;
  }
  out[((int64_t)b * c8p + g) * T + t] = (uint8_t)m;
}

int g_planes_tile = 0;      // dv3_debug_set(4, v)
int g_planes_mid_thr = 1;   // dv3_debug_set(8, v): 128x128 tiles once they number v/2 x the CUs, else 128x64 (measured:
                            // 1 -> nyanko bf16 step 13.12 ms, 2 -> 13.25, 4 -> 13.90; scripts/tile_thr_ab.py)
int g_planes_stagger = -1;  // dv3_debug_set(5, v)
int g_planes_abl = 0;       // dv3_debug_set(6, v)

#ifdef DV3_EXPERIMENTS
template <int WM, int WN, int NI, int ABL>
int launch_planes_abl(const ConvArgs& a, size_t lds, int grid, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)conv_planes_kernel<WM, WN, NI, 3, true, ABL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((conv_planes_kernel<WM, WN, NI, 3, true, ABL>), dim3(grid), dim3(WM * WN * 64), lds, st, a);
  return dv3_check_launch("conv_planes(abl)");
}

template <int WM, int WN, int NI, int ABL>
int launch_planes_abl1(const ConvArgs& a, size_t lds, int grid, hipStream_t st) {    // single-term bf16 (c8) ablations
  (void)hipFuncSetAttribute((const void*)conv_planes_kernel<WM, WN, NI, 1, false, ABL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((conv_planes_kernel<WM, WN, NI, 1, false, ABL>), dim3(grid), dim3(WM * WN * 64), lds, st, a);
  return dv3_check_launch("conv_planes(abl)");
}
#endif

template <int WM, int WN, int NI, int TERMS, bool F16, int JT>
int launch_planes_j(const ConvArgs& a, size_t lds, int grid, hipStream_t st) {
  static bool attr_set = false;  // raise the dynamic-LDS cap once per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_planes_kernel<WM, WN, NI, TERMS, F16, 0, JT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      dv3_set_error("conv_planes: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_planes_kernel<WM, WN, NI, TERMS, F16, 0, JT>), dim3(grid), dim3(WM * WN * 64), lds, st, a);
  return dv3_check_launch("conv_planes");
}
// dv3_debug_set(7, v): the unrolled steady state with the issue-order requests: 0 off, 1 on, -1 (default) = on for the
// single-term kernel only.  Three-term kernels: measured slower (reads land late behind 24 MFMAs).  Single-term: its
// 8 MFMAs per step do not cover the generic loop's scalar bookkeeping -- 3-8 % faster unrolled, bit-identical
// (scripts/planes_steady_c8.py: 80.4 -> 74.8 us eval, 92.3 -> 87.7 us training forward at the north-star shape).
int g_planes_steady = -1;
template <int WM, int WN, int NI, int TERMS, bool F16>
int launch_planes_t(const ConvArgs& a, size_t lds, int grid, hipStream_t st) {
  const bool steady = g_planes_steady < 0 ? TERMS == 1 : g_planes_steady != 0;
  if (steady && a.d.J == 3) return launch_planes_j<WM, WN, NI, TERMS, F16, 3>(a, lds, grid, st);
  if (steady && a.d.J == 1) return launch_planes_j<WM, WN, NI, TERMS, F16, 1>(a, lds, grid, st);
  return launch_planes_j<WM, WN, NI, TERMS, F16, 0>(a, lds, grid, st);
}
template <int WM, int WN, int NI>
int launch_planes(const ConvArgs& a, size_t lds, int grid, hipStream_t st) {
#ifdef DV3_EXPERIMENTS
  if (g_planes_abl && a.d.split_terms == DV3_SPLIT_F16X3 && NI == 2) {
    switch (g_planes_abl) {
      case 1: return launch_planes_abl<WM, WN, NI, 1>(a, lds, grid, st);
      case 2: return launch_planes_abl<WM, WN, NI, 2>(a, lds, grid, st);
      case 3: return launch_planes_abl<WM, WN, NI, 3>(a, lds, grid, st);
      case 4: return launch_planes_abl<WM, WN, NI, 4>(a, lds, grid, st);
      case 5: return launch_planes_abl<WM, WN, NI, 5>(a, lds, grid, st);
      case 6: return launch_planes_abl<WM, WN, NI, 6>(a, lds, grid, st);
      case 7: return launch_planes_abl<WM, WN, NI, 7>(a, lds, grid, st);
      case 8: return launch_planes_abl<WM, WN, NI, 8>(a, lds, grid, st);
      case 9: return launch_planes_abl<WM, WN, NI, 9>(a, lds, grid, st);
    }
  }
  if (g_planes_abl && a.d.split_terms == 1 && NI == 2) {
    switch (g_planes_abl) {
      case 1: return launch_planes_abl1<WM, WN, NI, 1>(a, lds, grid, st);
      case 2: return launch_planes_abl1<WM, WN, NI, 2>(a, lds, grid, st);
      case 3: return launch_planes_abl1<WM, WN, NI, 3>(a, lds, grid, st);
      case 4: return launch_planes_abl1<WM, WN, NI, 4>(a, lds, grid, st);
      case 5: return launch_planes_abl1<WM, WN, NI, 5>(a, lds, grid, st);
      case 6: return launch_planes_abl1<WM, WN, NI, 6>(a, lds, grid, st);
      case 8: return launch_planes_abl1<WM, WN, NI, 8>(a, lds, grid, st);
    }
  }
#endif
  if (a.d.split_terms == DV3_SPLIT_F16X3) return launch_planes_t<WM, WN, NI, 3, true>(a, lds, grid, st);
  if (a.d.split_terms == 1) return launch_planes_t<WM, WN, NI, 1, false>(a, lds, grid, st);
  return launch_planes_t<WM, WN, NI, 3, false>(a, lds, grid, st);
}

int dv3_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;
  }
  return n;
}

}  // namespace

extern "C" int dv3_split_planes_f32(const dv3_planes_desc* d, void* stream) {
  DV3_REQUIRE(d && d->x && d->out, "split_planes: null pointer");
  DV3_REQUIRE(d->B > 0 && d->C > 0 && d->T > 0, "split_planes: bad dims");
  DV3_REQUIRE(d->dtype == DV3_SPLIT_DTYPE_BF16 || d->dtype == DV3_SPLIT_DTYPE_F16, "split_planes: bad dtype");
  DV3_REQUIRE(((uintptr_t)d->out & 15) == 0, "split_planes: out must be 16-byte aligned");
  if (d->mask) DV3_REQUIRE(d->mask_rs * 32 >= d->T, "split_planes: mask row stride too small");
  const int c8p = (d->C + 31) / 32 * 4;
  DV3_REQUIRE(c8p <= 65535 && d->B <= 65535, "split_planes: grid too large");
  hipLaunchKernelGGL(split_planes_kernel, dim3(dv3_cdiv(d->T, 256), c8p, d->B), dim3(256), 0, (hipStream_t)stream, *d, c8p, dv3_range_ctr());
  return dv3_check_launch("split_planes");
}

extern "C" int dv3_to_c8_f32(const float* x, int64_t x_bs, int64_t x_rs, uint16_t* out, int32_t B, int32_t C,
                             int32_t T, void* stream) {
  DV3_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "to_c8: bad arguments");
  DV3_REQUIRE(((uintptr_t)out & 15) == 0, "to_c8: out must be 16-byte aligned");
  const int c8p = (C + 31) / 32 * 4;
  DV3_REQUIRE(c8p <= 65535 && B <= 65535, "to_c8: grid too large");
  hipLaunchKernelGGL(to_c8_kernel, dim3(dv3_cdiv(T, 256), c8p, B), dim3(256), 0, (hipStream_t)stream, x, x_bs, x_rs,
                     reinterpret_cast<bf16x8*>(out), C, T, c8p);
  return dv3_check_launch("to_c8");
}
extern "C" int dv3_from_c8_f32(const uint16_t* x, float* out, int64_t out_bs, int64_t out_rs, int32_t B, int32_t C,
                               int32_t T, void* stream) {
  return dv3_from_c8_head_f32(x, (C + 31) / 32 * 4, out, out_bs, out_rs, B, C, T, stream);
}
extern "C" int dv3_from_c8_head_f32(const uint16_t* x, int32_t c8p, float* out, int64_t out_bs, int64_t out_rs,
                                    int32_t B, int32_t C, int32_t T, void* stream) {
  DV3_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "from_c8: bad arguments");
  DV3_REQUIRE(((uintptr_t)x & 15) == 0, "from_c8: x must be 16-byte aligned");
  DV3_REQUIRE(c8p * 8 >= C, "from_c8: the tensor holds fewer than C channels");
  DV3_REQUIRE(c8p <= 65535 && B <= 65535, "from_c8: grid too large");
  hipLaunchKernelGGL(from_c8_kernel, dim3(dv3_cdiv(T, 256), dv3_cdiv(C, 8), B), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const bf16x8*>(x), out, out_bs, out_rs, C, T, c8p);
  return dv3_check_launch("from_c8");
}
extern "C" int dv3_mask_bits_to_c8(const uint32_t* bits, int32_t bits_rs, uint8_t* out, int32_t B, int32_t C,
                                   int32_t T, void* stream) {
  DV3_REQUIRE(bits && out && B > 0 && C > 0 && T > 0, "mask_bits_to_c8: bad arguments");
  DV3_REQUIRE(bits_rs * 32 >= T, "mask_bits_to_c8: mask row stride too small");
  const int c8p = (C + 31) / 32 * 4;
  DV3_REQUIRE(c8p <= 65535 && B <= 65535, "mask_bits_to_c8: grid too large");
  hipLaunchKernelGGL(mask_bits_to_c8_kernel, dim3(dv3_cdiv(T, 256), c8p, B), dim3(256), 0, (hipStream_t)stream, bits,
                     bits_rs, out, C, T, c8p);
  return dv3_check_launch("mask_bits_to_c8");
}

// called by dv3_conv_gemm_f32 (conv_gemm.hip) when d->x_planes != NULL; returns 1 when the shape is not
// eligible (caller falls back to the other kernels when it can), else a DV3_* code.
int dv3_conv_planes_dispatch(const dv3_conv_desc* d, hipStream_t st) {
  const bool gated = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  if (!d->a_split || d->a_bs != 0 || (d->lda & 3) || d->Tin != d->Tout) return 1;
  if ((d->J - 1) * d->dil > HALO_MAX || d->J * 2 > 32) return 1;
  const int kp = (d->Cin + 31) / 32 * 32;
  if (d->x_c8p != kp / 8) return 1;
  if (((uintptr_t)d->x_planes & 15) != 0) return 1;
  if ((int64_t)d->B * d->x_c8p * d->Tout >= (1ll << 28)) return 1;          // 32-bit byte offsets per plane
  if ((int64_t)d->J * (kp / 8) * d->lda >= (1ll << 27)) return 1;
  const int64_t ntot = (int64_t)d->B * d->Tout;
  if (ntot >= (1ll << 30)) return 1;
  // single-term bf16 on c8 input: the 256 x 256 k32 ping-pong kernel (conv_c8pp.hip) where its grid fills the chip
  // (tile_hint 40 forces it, dv3_debug_set(19, v) moves the threshold)
  if (d->split_terms == 1 && (d->tile_hint == 40 || (d->tile_hint == 0 && g_planes_tile == 0))) {
    const int rc = dv3_conv_c8pp_dispatch(d, st);
    if (rc != 1) return rc;
    if (d->tile_hint == 40) return 1;
  }
  // tile: 1 = 128x128 (4 waves, two workgroups per CU), 2 = 128x64 (4 waves), 9 = 128x256 (8 waves, one per CU)
  int id = g_planes_tile;
  if (id != 1 && id != 2 && id != 9) {
    const int64_t mt = gated ? dv3_cdiv(d->Cg, 64) : dv3_cdiv(d->M, 128);
    // measured at the north-star shape (profiles/r02c_planes_kernel_ablation.md): the 8-wave 128x256 tile is the
    // fastest where it fills the chip; two co-resident 128x128 workgroups otherwise; 128x64 for small problems
    const int64_t cols128 = dv3_cdiv64(ntot, 128);
    id = (mt * dv3_cdiv64(ntot, 256) >= (int64_t)dv3_num_cus()) ? 9
         : (2 * mt * cols128 >= g_planes_mid_thr * (int64_t)dv3_num_cus()) ? 1 : 2;
  }
  const int BM = 128, BMH = 64, BN = id == 1 ? 128 : id == 2 ? 64 : 256;
  const int NT = id == 9 ? 512 : 256;
  const int PL = d->split_terms == 1 ? 1 : 2;
  const int BNH = BN + (d->J - 1) * d->dil;
  const int XI = (KB * (BN + HALO_MAX) + NT - 1) / NT;
  const size_t lds = (size_t)(2 * PL * KB * BM + 2 * PL * XI * NT) * 16;   // X planes padded to XI * NT units
  if (lds > 160 * 1024) return 1;
  ConvArgs a;
  a.d = *d;
  a.kp = kp;
  a.m_tiles = gated ? dv3_cdiv(d->Cg, BMH) : dv3_cdiv(d->M, BM);
  a.n_tiles = (int)dv3_cdiv64(ntot, BN);
  const int64_t nb = (int64_t)a.m_tiles * a.n_tiles;
  DV3_REQUIRE(nb < (1ll << 31), "conv_planes: grid too large");
  a.n_blocks = (int)nb;
  const int cus = dv3_num_cus();
  a.a_scalar = cus;                       // (re-used field) number of CUs, for the stagger parity
  const int per_cu = (NT == 256 && 2 * lds <= 160 * 1024) ? 2 : 1;
  const int grid = (int)(nb < (int64_t)cus * per_cu ? nb : (int64_t)cus * per_cu);
  const int nsteps = (kp / 32) * d->J;
  // half a tile's main loop: nsteps x 24 MFMAs x 32 cycles x 2 waves per SIMD / 2, in s_sleep(127) units (~8.1k cycles)
  a.stagger = 0;
  if (per_cu == 2 && grid > cus) a.stagger = g_planes_stagger >= 0 ? g_planes_stagger : (nsteps * 768 + 4000) / 8128;
  g_dv3_last_conv = (d->split_terms == DV3_SPLIT_F16X3 ? 6000 : d->split_terms == 1 ? 8000 : 7000) + id * 10;
  switch (id) {
    case 1: return launch_planes<2, 2, 2>(a, lds, grid, st);
    case 2: return launch_planes<2, 2, 1>(a, lds, grid, st);
    case 9: return launch_planes<2, 4, 2>(a, lds, grid, st);
  }
  return 1;
}

int dv3_planes_debug_set(int what, int value) {
  if (what == 4) g_planes_tile = value;
  if (what == 5) g_planes_stagger = value;
  if (what == 6) g_planes_abl = value;
  if (what == 7) g_planes_steady = value;
  if (what == 8) g_planes_mid_thr = value;
  return DV3_OK;
}
