// Single-term bf16 tap-GEMM on channel-blocked bf16 ("c8") activations, 256 x 256 workgroup tile, 8 waves of 128 x 64,
// ping-pong at k32 granularity -- the forward / input-gradient kernel of the bf16 configurations (BASELINE configs 3/4:
// nyanko_ljspeech, deepvoice3_vctk; reference semantics deepvoice3_pytorch/modules.py:145-164, 205-226 and their
// autograd; layer shapes nyanko.py:28-58,364-399, deepvoice3.py:39-67,213-264).
//
// Same contract, operand images, accumulation order and fused tails as the single-term instantiation of
// conv_planes_kernel (conv_planes.hip), so its results are bit-identical to that kernel's.  What changes:
//
//   * a wave owns 128 rows (two 32-row sub-tiles of the `a` half + the matching gate rows) x 64 columns: 8 accumulator
//     blocks; per k16 block 6 operand fragments feed 8 MFMAs (the 64 x 64 wave tile of the 128 x 256 kernel reads 4
//     fragments per 4 MFMAs).  Round 3's PMC pass of that kernel: matrix pipe busy 24 % of the launch, half the wave
//     cycles parked in s_waitcnt -- at one MFMA per fragment the LDS pipe (256 B/clk) is as busy as the matrix pipe.
//   * the two waves of a SIMD alternate LOAD / COMPUTE phases of one (32-channel chunk, tap) step: a LOAD phase stores
//     the thread's share of the NEXT step's weight panel and of the NEXT chunk's activation tile (fetched one step / one
//     chunk earlier: plain 16-byte copies, c8 IS the operand layout), re-issues those fetches for the step / chunk
//     after, and reads the 12 fragments of its step; the COMPUTE phase is 16 MFMAs (512 matrix-pipe cycles) on registers
//     only.  While one wave of a SIMD computes, its partner loads.
//   * every global load is unconditional and issued in one fixed order per step (the tail re-fetches the last panel /
//     chunk into buffers nobody reads), so each s_waitcnt vmcnt is exact (conv_gemm_pp2.hip measured why).
//
// LDS: weight panels [2][4 k8][256 rows] = 32 KB, activation tiles [2][XI * 512 units] <= 48 KB.
// Dropout: keep-BYTES of the c8 input (dv3_conv_desc.xmask_c8), applied to the staged units; 1/(1-p) on the accumulators.
#include "conv_common.h"
#include <math.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KB = 4, HALO_MAX = 64;
constexpr int BM = 256, BMH = 128, MI = 2, NI = 2;
// NW = 8 (rounds 4-5): one 8-wave workgroup per CU, 256 x 256 tile, the two waves of a SIMD half a step apart (ping-pong).
// NW = 4 (round 5): a 4-wave workgroup on a 256 x 128 tile (2 x 2 waves of the same 128 x 64 wave tile), 57 KB of LDS, TWO
// of them per CU: independent workgroups, so the prologue and the chip-wide tail burst of one run under the main loop of
// the other (a third of the 8-wave launch is prologue + tail); inside a workgroup the four waves run the
// LOAD / COMPUTE phases together (one barrier per step), the SIMD partner is a wave of the other workgroup.
template <int NW> struct C8ppGeo {
  static constexpr int NT = NW * 64, WN = NW / 2, BN = WN * NI * 32;
  static constexpr int AU = KB * BM / NT;                   // weight-panel units per thread per step (2 / 4)
};

template <typename T>
__device__ __forceinline__ T c8pp_ldg(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
// zero the dropped channels of one unit (bit e of the keep-byte = channel e)
__device__ __forceinline__ bf16x8 c8pp_keep8(const bf16x8& v, uint32_t m) {
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

// ABL (make EXP=1 only; dv3_debug_set(21, v)): timing-only ablations, results are wrong: 1 no MFMAs, 2 no staging (no
// global fetches, no LDS stores) in the loop, 3 no tail, 5 no fragment reads in the loop, 6 no barriers in the loop
// RF (round 5): the twelve fragment reads of a LOAD phase are issued FIRST and the staging (panel / tile stores, the
// refetches) runs while they land -- the phase is then max(reads, staging) long instead of their sum (the reads of four
// waves take 200-350 cycles of the 512 a partner's 16 MFMAs last).  Same instructions, same results.
// RL (round 5): the residual of a Conv1dGLU / HighwayConv1d layer IS its input (modules.py:139,163,224-226: y = f(conv(x))
// + x): the 128 `a` channels of this tile are four of the 32-channel chunks the main loop stages anyway.  Their units
// (raw, before the keep-bytes) are stored a second time into a 64 KB strip [16 channel groups][256 columns] behind the
// tile buffers, and the tail reads the residual from there instead of fetching 33.5 MB (north star) from memory inside
// the chip-wide tail burst.  Taken when the descriptor's residual tensor is the input tensor itself.
template <int JT, bool MASK, int ABL = 0, bool RF = false, bool RL = false, int NW = 8>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv_c8pp_kernel(const ConvArgs args) {
  constexpr int NT = C8ppGeo<NW>::NT, WN = C8ppGeo<NW>::WN, BN = C8ppGeo<NW>::BN, AU = C8ppGeo<NW>::AU;
  constexpr bool PP = NW == 8;               // ping-pong halves
  constexpr int XI = (KB * (BN + (JT > 1 ? HALO_MAX : 0)) + NT - 1) / NT;   // activation units per thread per chunk
  constexpr int XPS = XI * NT;                                              // units per tile buffer (padded: no store is predicated)
  static_assert(JT == 1 || JT == 3, "tap counts of the models' layers");
  static_assert(JT == 1 || XI == JT, "three-tap layers: one activation item per tap phase");
  const dv3_conv_desc& p = args.d;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int dil = p.dil;
  const int BNH = BN + (JT - 1) * dil;
  bf16x8* const As = reinterpret_cast<bf16x8*>(smem_raw);   // [2 buffers][KB][BM]
  bf16x8* const Xs = As + 2 * KB * BM;                      // [2 buffers][XPS]  ([KB][BNH] + padding)
  bf16x8* const Rs = Xs + 2 * XPS;                          // RL: [BMH / 8 channel groups][BN] residual units

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int pid = dv3_xcd_remap(blockIdx.x, args.n_blocks);
  const int mt = pid % args.m_tiles;
  const int nt = pid / args.m_tiles;
  const int n0 = nt * BN;

  const bool gated = (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY);
  int h0b, h1b;
  if (gated) {
    h0b = mt * BMH; h1b = p.a_half + mt * BMH;
  } else {
    h0b = mt * BM; h1b = mt * BM + BMH;
  }

  const int T = p.Tout, lda = p.lda, B = p.B;
  const int Ntot = B * T;
  const int k8_total = args.kp >> 3;                        // == p.x_c8p
  const int nchunks = args.kp >> 5;
  const bf16x8* __restrict__ Wh = reinterpret_cast<const bf16x8*>(p.a_split);
  const bf16x8* __restrict__ XP = reinterpret_cast<const bf16x8*>(p.x_planes);
  const uint8_t* __restrict__ const xkeep = p.xmask_c8;
  const int n_items = KB * BNH;

  // ---- this lane's output columns: per-tap validity of the shifted read (the conv's zero padding at sequence edges) ----
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
  uint32_t need = 0;
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    const uint32_t all = ((1u << NI) - 1u) << (j * NI);
    if (!__all((vbits & all) == all)) need |= 1u << j;
  }
  need = __builtin_amdgcn_readfirstlane(need);

  // ---- this thread's staging units, fixed over the K loop ----
  uint32_t xoff[XI];                // byte offset of unit (b, k8, t) inside the c8 tensor, chunk 0
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
  uint32_t aoff[AU];                // byte offset of this thread's panel units inside a (tap, chunk) panel row block
#pragma unroll
  for (int u = 0; u < AU; ++u) {
    const int idx = tid + u * NT;   // k8 * BM + col
    const int col = idx % BM, k8 = idx / BM;
    const bool hi_half = col >= BMH;
    const int gcol = (hi_half ? h1b : h0b) + (col - (hi_half ? BMH : 0));
    aoff[u] = (uint32_t)(k8 * lda + (gcol < lda ? gcol : 0)) * 16u;
  }

  bf16x8 ra[AU], rx[XI];
  uint32_t rk[MASK ? XI : 1];

  auto load_A = [&](int chunk, int j) {
    const bf16x8* src = Wh + (int64_t)(j * k8_total + chunk * KB) * lda;   // uniform
#pragma unroll
    for (int u = 0; u < AU; ++u) ra[u] = c8pp_ldg<bf16x8>(src, aoff[u]);
  };
  auto write_A = [&](int buf) {
    bf16x8* dst = As + buf * (KB * BM);
#pragma unroll
    for (int u = 0; u < AU; ++u) dst[tid + u * NT] = ra[u];
  };
  auto load_X_item = [&](int chunk, auto ic) {
    constexpr int i = decltype(ic)::value;
    const bf16x8* src = XP + (int64_t)chunk * KB * T;                       // uniform: chunk c = 4 k8 blocks further
    rx[i] = c8pp_ldg<bf16x8>(src, xoff[i]);
    if constexpr (MASK) rk[i] = (uint32_t)c8pp_ldg<uint8_t>(xkeep + (int64_t)chunk * KB * T, xoff[i] >> 4);
  };
  // `chunk`: the 32-channel chunk the item belongs to (RL: chunks [mt * 4, mt * 4 + 4) hold this tile's `a` channels)
  auto write_X_item = [&](int buf, auto ic, int chunk) {
    constexpr int i = decltype(ic)::value;
    bf16x8 v = rx[i];
    if constexpr (RL) {
      const int cr = chunk - mt * (BMH / 32);                        // uniform
      if (cr >= 0 && cr < BMH / 32) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));                                  // (recomputed here, four times per tile: no register held for it)
        const int idx = t_ + i * NT;
        const int k8 = idx / BNH, col = idx - k8 * BNH - p.padL;      // unit (k8, q) of the haloed tile; output column q - padL
        if (idx < n_items && col >= 0 && col < BN) Rs[(cr * KB + k8) * BN + col] = v;
      }
    }
    if constexpr (MASK) v = c8pp_keep8(v, rk[i]);
    Xs[buf * XPS + tid + i * NT] = v;
  };
  using U0 = std::integral_constant<int, 0>;
  using U1 = std::integral_constant<int, 1>;
  using U2 = std::integral_constant<int, 2>;
  auto load_X_all = [&](int chunk) {
    load_X_item(chunk, U0{});
    load_X_item(chunk, U1{});
    if constexpr (XI == 3) load_X_item(chunk, U2{});
  };
  auto write_X_all = [&](int buf, int chunk) {
    write_X_item(buf, U0{}, chunk);
    write_X_item(buf, U1{}, chunk);
    if constexpr (XI == 3) write_X_item(buf, U2{}, chunk);
  };

  f32x16 acc[MI][2][NI];   // [row sub-tile][a rows | gate rows][column sub-tile]
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][h][ni][r] = 0.f;

  const int a_off = wm * (MI * 32) + l31;
  const int x_off = wn * (NI * 32) + l31;

  // ---- prologue: step 0's panel and chunk 0's tile into buffer 0; then the fetches that run a step / a chunk ahead ----
  load_A(0, 0);
  load_X_all(0);
  write_A(0);
  write_X_all(0, 0);
  __syncthreads();
  {
    // step 1 = (chunk 0, tap 1) for three-tap layers, (chunk 1, tap 0) for 1 x 1 layers; past the end: re-fetch step 0
    int c1 = JT == 1 ? 1 : 0, j1 = JT == 1 ? 0 : 1;
    if (c1 >= nchunks) { c1 = 0; j1 = 0; }
    load_A(c1, j1);
    load_X_all(min(1, nchunks - 1));
  }

  // ---- ping-pong main loop: waves w and w + 4 share a SIMD and run the same phase sequence one phase apart ----
  //   interval:   I0        I1        I2        I3
  //   waves 0-3:  L(0)      C(0)      L(1)      C(1) ...
  //   waves 4-7:  -         L(0)      C(0)      L(1) ...
  // LDS hazards: the panel of step s+1 is stored during the L phases of step s (intervals 2s, 2s+1) into the buffer last
  // read in the L phases of step s-1 (intervals 2s-2, 2s-1) and first read in L(s+1) (interval 2s+2); the tile of chunk
  // c+1 during the L phases of chunk c into the buffer last read in chunk c-1.  Every interval ends with a barrier.
  const int late = PP ? (wave >> 2) : 0;
  if (late && ABL != 6) __syncthreads();
  if constexpr (!PP) {
    // the two workgroups of a CU start together: the odd one of a pair waits a little so that they do not run the
    // same phase (args.stagger units of 64 cycles; which blocks share a CU is the dispatcher's business -- adjacent
    // ones by observation)
    if (args.stagger > 0 && (blockIdx.x & args.stagger_mask) != 0)
      for (int i = 0; i < args.stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }
  for (int c = 0; c < nchunks; ++c) {
    const bf16x8* XsC = Xs + (c & 1) * XPS;
    const int cx = min(c + 2, nchunks - 1);          // the chunk fetched during this one (the tail re-fetches the last)
    const bool last_chunk = c + 1 == nchunks;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int cur = (c * JT + j) & 1;
      const bf16x8* AsC = As + cur * (KB * BM);
      const bool fix = (need >> j) & 1u;
      // ---------------- LOAD ----------------
      bf16x8 fa[2][MI][2], fb[2][NI];
      auto read_frags = [&]() {
      if (ABL == 5) {                       // fragments from registers that are live anyway: no LDS reads
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) fa[ks][mi][0] = fa[ks][mi][1] = ra[ks];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) fb[ks][ni] = rx[ni];
        }
      } else
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k8 = 2 * ks + lhi;
        const int ai = k8 * BM + a_off;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          fa[ks][mi][0] = AsC[ai + mi * 32];
          fa[ks][mi][1] = AsC[ai + mi * 32 + BMH];
        }
        const int xi = k8 * BNH + x_off + j * dil;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ks][ni] = XsC[xi + ni * 32];
      }
      };
      auto stage = [&]() {
      if (ABL != 2) {
        // the next step's panel: store (fetched in this wave's previous LOAD phase), then fetch the panel after
        int j2 = j + 2, c2 = c;
        if (JT == 1) { j2 = 0; c2 = c + 2; }
        else if (j2 >= JT) { j2 -= JT; c2 = c + 1; }
        if (c2 >= nchunks) { c2 = c; j2 = j; }            // past the end: re-fetch the current panel
        write_A(cur ^ 1);
        load_A(c2, j2);
        // the next chunk's tile: one item per tap phase (three-tap layers) or all of it (1 x 1 layers)
        if constexpr (JT == 1) {
          write_X_all((c + 1) & 1, c + 1);
          load_X_all(cx);
        } else {
          if (j == 0) { write_X_item((c + 1) & 1, U0{}, c + 1); load_X_item(cx, U0{}); }
          if (j == 1) { write_X_item((c + 1) & 1, U1{}, c + 1); load_X_item(cx, U1{}); }
          if (j == 2) { write_X_item((c + 1) & 1, U2{}, c + 1); load_X_item(cx, U2{}); }
        }
      }
      };
      if constexpr (RF) {
        read_frags();
        __builtin_amdgcn_sched_barrier(0);
        stage();
        __builtin_amdgcn_sched_barrier(0);
      } else {
        stage();
        __builtin_amdgcn_sched_barrier(0);
        read_frags();
      }
      if (fix) {
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const bool ok = (vbits >> (j * NI + ni)) & 1u;
            fb[ks][ni] = ok ? fb[ks][ni] : zero8;
          }
      }
      if (ABL != 6) __syncthreads();
      // the MFMAs are register-only: without the fences the compiler sinks them below the second barrier into the next
      // LOAD phase and the ping-pong degenerates into the in-phase loop (conv_gemm_pp2.hip)
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- COMPUTE: 16 MFMAs of one (chunk, tap) step ----------------
      if (ABL == 1) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(fa[ks][mi][0]), "v"(fa[ks][mi][1]));
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(fb[ks][ni]));
        }
      } else
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mi][0], fb[ks][ni], acc[mi][0][ni], 0, 0, 0);
            acc[mi][1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mi][1], fb[ks][ni], acc[mi][1][ni], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
      // (4-wave form: the barrier after the LOAD phase orders everything: the next LOAD's stores go to buffers whose last
      //  reads are behind it, its reads to buffers whose stores are behind it)
      if (PP && ABL != 6 && (!(last_chunk && j == JT - 1) || !late)) __syncthreads();
    }
  }
  if (ABL == 3 && acc[0][0][0][0] + acc[1][1][1][7] != 1.2345e30f) return;

  // ---- fused tail (conv_common.h), one 32-row sub-tile at a time ----
  int n0e = __builtin_amdgcn_readfirstlane(n0);
  asm volatile("" : "+s"(n0e));
  int bcol[NI], tcol[NI];
  bool okc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0e + wn * (NI * 32) + ni * 32 + l31;
    okc[ni] = n < Ntot;
    bcol[ni] = n / T;
    tcol[ni] = n - bcol[ni] * T;
  }
  if constexpr (MASK) {      // x * keep / (1-p): the 1/(1-p) of a masked c8 input, exact on the accumulators
    const float ds = p.drop_scale;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][h][ni][r] *= ds;
  }
  if (p.io_bf16 & DV3_IO_OUT_C8) {
    // RL: the residual units of this lane's columns sit in the strip (every store to it is behind a barrier this wave passed)
    const unsigned char* rl = RL ? reinterpret_cast<const unsigned char*>(Rs) + (size_t)(wn * (NI * 32) + l31) * 16 : nullptr;
    conv_epilogue_c8<BM, BMH, NI>(p, acc[0], gated, mt, wm * (MI * 32), lhi, bcol, tcol, okc, rl, BN);
    conv_epilogue_c8<BM, BMH, NI>(p, acc[1], gated, mt, wm * (MI * 32) + 32, lhi, bcol, tcol, okc, rl, BN);
  } else {
    conv_epilogue<BM, BMH, NI, 0, true>(p, acc[0], gated, mt, wm * (MI * 32), lhi, bcol, tcol, okc);
    conv_epilogue<BM, BMH, NI, 0, true>(p, acc[1], gated, mt, wm * (MI * 32) + 32, lhi, bcol, tcol, okc);
  }
}

template <int JT, bool MASK, int ABL = 0, bool RF = false, bool RL = false, int NW = 8>
int launch_c8pp(const ConvArgs& a, size_t lds, hipStream_t st) {
  constexpr int NT = C8ppGeo<NW>::NT, BN = C8ppGeo<NW>::BN;
  if (RL) lds += (size_t)(BMH / 8) * BN * 16;      // the residual strip
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_c8pp_kernel<JT, MASK, ABL, RF, RL, NW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      dv3_set_error("conv_c8pp: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_c8pp_kernel<JT, MASK, ABL, RF, RL, NW>), dim3(a.n_blocks), dim3(NT), lds, st, a);
  return dv3_check_launch("conv_c8pp");
}

}  // namespace

// dv3_debug_set(30, v): fragment reads first in a LOAD phase (RF instantiations; 0 = the round-4 order).  Default since
// round 5: bit-identical, 0.87-0.93 of the staging-first time over the presets' shapes in one process
// (scripts/r5_ship_check.py, profiles/r05_c8pp_reads_first.txt: north star eval 63.9 -> 59.3 us, masked training forward
// 77.4 -> 72.0 us, input gradient 58.5 -> 54.3 us; C = 512, T = 800: 201.5 -> 183.0 / 241.0 -> 209.0 / 186.0 -> 164.2 us)
int g_c8pp_rf = 1;
int g_c8pp_nw4 = 2;            // dv3_debug_set(34, v): two 4-wave workgroups per CU on 256 x 128 tiles (NW = 4): 0 never, 1 always, 2 by the rule in the dispatcher
int g_c8pp_stagger = 0, g_c8pp_stagger_mask = 1;   // dv3_debug_set(35 / 36, v): start delay (x 64 cycles) of the blocks with (blockIdx & mask) != 0
int g_c8pp_rl = 0;             // dv3_debug_set(32, v), experiment build: the residual of a gated layer read from LDS (RL instantiations)
int g_c8pp_abl = 0;            // dv3_debug_set(21, v): timing-only ablations (EXP build)
int g_c8pp_min_tiles = 128;   // dv3_debug_set(19, v): the 256 x 256 c8 kernel serves eligible shapes whose grid has at
                              // least v tiles (0 = never; 1 = always)

// Called by dv3_conv_planes_dispatch (conv_planes.hip) for single-term bf16 layers on c8 input.  Returns 1 when the
// shape is not eligible (the caller continues with the 128-row planes kernel), else a DV3_* code.
int dv3_conv_c8pp_dispatch(const dv3_conv_desc* d, hipStream_t st) {
  if (g_c8pp_min_tiles <= 0 || d->split_terms != 1 || !d->a_split || !d->x_planes) return 1;
  if ((d->J != 1 && d->J != 3) || (d->J - 1) * d->dil > HALO_MAX) return 1;
  if (d->a_bs != 0 || (d->lda & 3) || d->Tin != d->Tout) return 1;
  const bool gated = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  const int kp = (d->Cin + 31) / 32 * 32;
  if (d->x_c8p != kp / 8) return 1;
  const int64_t ntot = (int64_t)d->B * d->Tout;
  const int64_t m_tiles = gated ? dv3_cdiv(d->Cg, BMH) : dv3_cdiv(d->M, BM);
  constexpr int BN = C8ppGeo<8>::BN, NT = C8ppGeo<8>::NT;
  const int64_t nb = m_tiles * dv3_cdiv64(ntot, BN);
  // Which form (round 5, profiles/r05_c8pp_two_workgroups_per_cu.txt; one process, graph-timed, B = 64):
  //   * 8 waves on 256 x 256 (NW = 8): eval forward and input gradients of grids that fill the chip (>= 256 tiles: the
  //     4-wave form is 5-14 % slower there: in-phase SIMD partners, twice the weight-panel traffic per column);
  //   * two 4-wave workgroups per CU on 256 x 128 (NW = 4): the masked training forward with its pre-gate save (three
  //     times the tail stores: -4 ... -8 % at every size, -14 % at 100 tiles) and input gradients of grids that do NOT
  //     fill the chip with 256 x 256 tiles (64 ... 191 of them: -21 ... -26 %; the 8-wave form leaves CUs idle there);
  //   * below that the 128-row planes kernel (return 1).
  const bool is_dgrad = d->mode == DV3_EPI_DGRAD;
  const bool masked_fwd = d->xmask_c8 != nullptr && !is_dgrad;
  bool use_nw4 = g_c8pp_nw4 == 1 || d->tile_hint == 41;
  if (g_c8pp_nw4 == 2 && d->tile_hint != 40 && d->tile_hint != 41) {   // (a forcing hint survives the size rule: ADVICE r5)
    if (masked_fwd) use_nw4 = nb >= 100;
    else if (is_dgrad) use_nw4 = nb >= 64 && nb < 192;   // (201 tiles: 8-wave 49 us, 4-wave 55: profiles/r05_conv_census_nyanko_bf16_c8.txt)
  }
  if (d->tile_hint != 40 && !use_nw4 && nb < g_c8pp_min_tiles) return 1;
  const int XI = (KB * (BN + (d->J > 1 ? HALO_MAX : 0)) + NT - 1) / NT;
  const size_t lds = (size_t)(2 * KB * BM + 2 * XI * NT) * 16;
  ConvArgs a;
  a.d = *d;
  a.a_scalar = 0;
  a.kp = kp;
  a.m_tiles = (int)m_tiles;
  a.n_tiles = (int)dv3_cdiv64(ntot, BN);
  DV3_REQUIRE(nb < (1ll << 31), "conv_c8pp: grid too large");
  a.n_blocks = (int)nb;
  g_dv3_last_conv = 9000 + 100 + 1;     // single-term c8, 256 x 256 tile, ping-pong
  const bool mask = d->xmask_c8 != nullptr;
  if (use_nw4) {
    // two 4-wave workgroups per CU on 256 x 128 tiles (NW = 4)
    constexpr int BN4 = C8ppGeo<4>::BN, NT4 = C8ppGeo<4>::NT;
    const int XI4 = (KB * (BN4 + (d->J > 1 ? HALO_MAX : 0)) + NT4 - 1) / NT4;
    const size_t lds4 = (size_t)(2 * KB * BM + 2 * XI4 * NT4) * 16;
    a.n_tiles = (int)dv3_cdiv64(ntot, BN4);
    const int64_t nb4 = m_tiles * a.n_tiles;
    DV3_REQUIRE(nb4 < (1ll << 31), "conv_c8pp: grid too large");
    a.n_blocks = (int)nb4;
    a.stagger = g_c8pp_stagger;
    a.stagger_mask = g_c8pp_stagger_mask;
    g_dv3_last_conv = 9000 + 110 + 1;   // ... 256 x 128 tile, two workgroups per CU
    if (d->J == 3) return mask ? launch_c8pp<3, true, 0, true, false, 4>(a, lds4, st) : launch_c8pp<3, false, 0, true, false, 4>(a, lds4, st);
    return mask ? launch_c8pp<1, true, 0, true, false, 4>(a, lds4, st) : launch_c8pp<1, false, 0, true, false, 4>(a, lds4, st);
  }
#ifdef DV3_EXPERIMENTS
  if (g_c8pp_abl && !mask && d->J == 3) {
    switch (g_c8pp_abl) {
      case 1: return launch_c8pp<3, false, 1>(a, lds, st);
      case 2: return launch_c8pp<3, false, 2>(a, lds, st);
      case 3: return launch_c8pp<3, false, 3>(a, lds, st);
      case 5: return launch_c8pp<3, false, 5>(a, lds, st);
      case 6: return launch_c8pp<3, false, 6>(a, lds, st);
    }
  }
#endif
#ifdef DV3_EXPERIMENTS
  // residual from LDS (RL): MEASURED AND RETIRED (round 5, profiles/r05_c8pp_residual_from_lds.txt): bit-identical, but 4-10 %
  // SLOWER over the presets' shapes (north star eval 61.2 -> 65.6 us) -- the tail is not waiting for the residual fetch,
  // and the second store of the staged units plus the strip reads cost more than the 33 MB they keep out of the burst.
  const bool has_res = gated && d->r && (d->mode == DV3_EPI_HIGHWAY || d->residual);
  const bool rl = g_c8pp_rl && g_c8pp_rf && has_res && (const void*)d->r == (const void*)d->x_planes && d->Cg == d->Cin && (d->io_bf16 & DV3_IO_OUT_C8) &&
                  lds + (size_t)(BMH / 8) * BN * 16 <= 160 * 1024;
  if (rl) {
    if (d->J == 3) return mask ? launch_c8pp<3, true, 0, true, true>(a, lds, st) : launch_c8pp<3, false, 0, true, true>(a, lds, st);
    return mask ? launch_c8pp<1, true, 0, true, true>(a, lds, st) : launch_c8pp<1, false, 0, true, true>(a, lds, st);
  }
#endif
  if (g_c8pp_rf) {
    if (d->J == 3) return mask ? launch_c8pp<3, true, 0, true>(a, lds, st) : launch_c8pp<3, false, 0, true>(a, lds, st);
    return mask ? launch_c8pp<1, true, 0, true>(a, lds, st) : launch_c8pp<1, false, 0, true>(a, lds, st);
  }
  if (d->J == 3) return mask ? launch_c8pp<3, true>(a, lds, st) : launch_c8pp<3, false>(a, lds, st);
  return mask ? launch_c8pp<1, true>(a, lds, st) : launch_c8pp<1, false>(a, lds, st);
}

int dv3_c8pp_debug_set(int what, int value) {
  if (what == 19) g_c8pp_min_tiles = value;
  if (what == 21) g_c8pp_abl = value;
  if (what == 30) g_c8pp_rf = value;
  if (what == 32) g_c8pp_rl = value;
  if (what == 34) g_c8pp_nw4 = value;
  if (what == 35) g_c8pp_stagger = value;
  if (what == 36) g_c8pp_stagger_mask = value;
  return DV3_OK;
}
