// Split-operand tap-GEMM, 256 x 256 workgroup tile, 8 waves of 128 x 64, ping-pong at k16 granularity.
//
// Same contract, operand images, accumulation order and fused tail as conv_gemm_bf16x3.hip (reference semantics:
// deepvoice3_pytorch/modules.py:145-164, 205-226; the input gradient of the same layers), so its results are
// bit-identical to that kernel's.  What changes is the work per wave and the phase structure:
//
//   * a wave owns 128 rows (two 32-row sub-tiles of the `a` half + the matching gate rows) x 64 columns: 8 accumulator
//     blocks, 12 operand fragments per k16 block feed 24 MFMAs (the 64 x 64 wave tile of the 128 x 256 kernel reads 8
//     fragments per 12 MFMAs): a third fewer LDS fragment reads, half the weight-panel bytes and half the
//     activation-tile conversions per MFMA.
//   * the two waves of a SIMD alternate LOAD / COMPUTE phases of ONE k16 block (24 MFMAs = ~800 cycles): a LOAD phase
//     reads its 12 fragments and does a fixed share of the staging -- one of the thread's two weight-panel units of the
//     NEXT step (store + refetch for the step after), and the activation-tile items assigned to that phase of the chunk
//     (convert + store for the next chunk, refetch for the one after) -- so every LOAD phase carries the same ~800 cycles
//     of work as the COMPUTE phase it runs beside (round-2 stamps of the 128 x 256 kernel: LOAD 820 / 1700 cycles
//     against COMPUTE 800, profiles/r01e_pingpong_phase_stamps.md).
//
// LDS: [2 buffers] x {A hi, A lo}[4 k8][256 rows] = 64 KB, [2 buffers] x {X hi, X lo}[4 k8][256 + halo] <= 80 KB.
// Dropout (MASK): keep-BYTES [B][C8][T] (bit e of byte (b, g, t) = channel 8g+e; dv3_conv_desc.xmask_c8): one byte
// load per staged item instead of eight keep-bit words, which is what lets the staging registers fit beside the
// 128 accumulator registers.
#include "conv_common.h"
#include <math.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BKC = 32, KB = 4, HALO_MAX = 64;
constexpr int BM = 256, BMH = 128, BN = 256, NT = 512, MI = 2, NI = 2, WN = 4;
constexpr int JT = 3;                                       // taps (the models' kernel size); other sizes: conv_gemm_bf16x3.hip
constexpr int AU = KB * BM / NT;                            // weight-panel units per plane per thread per step (2)
constexpr int XI = (KB * (BN + HALO_MAX) + NT - 1) / NT;    // activation items per thread per chunk (3)
static_assert(AU == 2, "one panel unit per k16 phase");

__device__ __forceinline__ void pp2_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
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
template <bool F16>
__device__ __forceinline__ f32x16 pp2_mma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool OPAQUE>
__device__ __forceinline__ const ConvArgs* dv3_opaque_args(const ConvArgs* a) {
  if constexpr (OPAQUE) {
    int off = 0;
    asm volatile("" : "+s"(off));
    return reinterpret_cast<const ConvArgs*>(reinterpret_cast<const char*>(a) + off);
  } else {
    return a;
  }
}
template <typename T>
__device__ __forceinline__ T pp2_ldg(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// ABL 4 (dv3_debug_set(13, 4)): phase timestamps of ONE workgroup, [wave][slot][0 = s_memrealtime (100 MHz),
// 1 = s_memtime (shader clock)]; read back with dv3_debug_read(2, ...)
constexpr int PP2_STAMPS = 320;
__device__ unsigned long long g_pp2_stamps[8 * PP2_STAMPS * 2];

// SK (stream-K form, see the note above dv3_conv_gemm_pp2_dispatch): the grid is one workgroup per CU and a workgroup
// walks a contiguous range of (tile, 32-channel chunk) units -- at most one leading segment that ends a tile another
// workgroup began (its accumulators go to the workspace) and then segments that begin a tile (the workgroup that
// holds a tile's chunk 0 adds the other workgroups' parts and runs the fused tail).
// ORD (bit flags; round 5: what a LOAD phase spends its time on, profiles/r05_pp2_load_phase.txt):
//   1  the twelve fragment reads are issued FIRST in a LOAD phase and the staging (panel unit store / fetch, activation
//      item conversion + store, activation fetches) runs while they land; only the edge fix and the barrier wait for them
//   2  lean staging: the two weight-panel unit offsets live in registers (two VGPRs instead of ~15 VALU per phase) and the
//      fp16 pair of an in-range item is built without the clamps (the range test the kernel already makes picks the path)
//  32  the Conv1dGLU / highway tail moves 16 bytes per access: a 4 x 4 transpose inside each quad of lanes (DPP) turns the
//      accumulator's one-frame-per-lane layout into four consecutive frames of one row per lane (conv_common.h)
//  64  the activation items are converted and stored in COMPUTE phases, between the MFMAs, instead of in LOAD phases
//   4  the tile's residual rows are touched (one 4-byte load per 128-byte line) during the last chunk of the main loop, so
//      that the tail reads them from L2 while the chip-wide tail burst only writes
//  PW (round 6): the activation tensor holds PAIR WORDS (include/dv3hip.h: dv3_conv_desc.x_pair) -- the bf16 hi / lo pair of
//      every element, built once by the kernel that produced the tensor: an item is staged with eight v_perm_b32 instead
//      of the fp32 -> pair conversion (bf16-pair instantiations without dropout: the input gradient of a gated layer)
//  FG (round 6): an input-gradient launch whose tail also runs the gate backward of the layer that PRODUCED this layer's
//      input (dv3_conv_desc.pg; conv_common.h: conv_epilogue_dgrad_gate / conv_epilogue_wide_block_gate).  A separate
//      instantiation that contains that tail and no other: as a run-time branch of the shared tail it cost every
//      instantiation of this kernel its spill-free register allocation.
template <bool MASK, bool F16, int ABL = 0, bool SK = false, int ORD = 0, bool PW = false, bool FG = false>
__global__ __launch_bounds__(NT) void conv_gemm_pp2_kernel(const ConvArgs args) {
  static_assert(!PW || (!MASK && !F16 && !SK), "pair words: the unmasked bf16-pair tile-per-workgroup form");
  static_assert(!FG || (!MASK && !F16 && !SK), "fused gate backward: the unmasked bf16-pair tile-per-workgroup form");
  const dv3_conv_desc& p = args.d;
  int n_stamp = 0;
  auto stamp = [&]() {
    if constexpr (ABL == 4) {
      if (blockIdx.x == gridDim.x / 2 + 3 && n_stamp < PP2_STAMPS) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) {
          g_pp2_stamps[((threadIdx.x >> 6) * PP2_STAMPS + n_stamp) * 2] = t0;
          g_pp2_stamps[((threadIdx.x >> 6) * PP2_STAMPS + n_stamp) * 2 + 1] = t1;
        }
      }
      ++n_stamp;
    }
  };
  stamp();                               // slot 0: kernel entry
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int J = JT;
  bf16x8* const As = reinterpret_cast<bf16x8*>(smem_raw);   // [2 buffers][hi | lo][KB][BM]
  bf16x8* const Xs = As + 2 * 2 * KB * BM;                  // [2 buffers][hi | lo][KB][BNH]

  int pid_ = dv3_xcd_remap(blockIdx.x, args.n_blocks);
  if constexpr (SK) {
    // Round 5: the hand-over below makes the workgroup that BEGINS a cut tile wait for the workgroups that
    // hold the rest of it -- the ones with the next HIGHER index, whose piece of that tile is their FIRST segment.  With
    // indices in dispatch order that is a wait for workgroups dispatched LATER: fine while the whole grid is resident, a
    // hang when it is not (a CU mask, CU-holding kernels of another stream).  The index used from here on therefore runs
    // AGAINST the dispatch order inside an XCD's group of workgroups (dv3_xcd_remap gives XCD x the contiguous range
    // [x q, (x + 1) q), slot s = blockIdx / 8): slot s takes index q - 1 - s.  "The next higher index" is then the
    // workgroup dispatched 8 blocks EARLIER, and what is waited for is the first thing it does, which depends on
    // nothing.  (The groups own whole numbers of tiles -- see the dispatcher -- so a wait never leaves the group.)
    const int q = 1 << args.sk_qshift;
    pid_ = (pid_ & ~(q - 1)) + (q - 1 - (pid_ & (q - 1)));
  }
  const int pid = pid_;
  const int nchunks = p.Cin / BKC;       // whole chunks only (dispatcher)
  // stream-K: this workgroup's unit range [seg_u, seg_end); unit = tile * nchunks + chunk.  A group (pid >> sk_qshift)
  // owns a whole number of tiles and deals its units out evenly (divisions on the host: a uniform division here leaves
  // its reciprocal -- and a zero -- in vector registers).
  int seg_u = 0, seg_end = 0, sk_base = 0, sk_rem = 0, sk_g0 = 0;
  if constexpr (SK) {
    const int grp = pid >> args.sk_qshift, slot = pid & ((1 << args.sk_qshift) - 1);
    const bool big = grp < args.sk_tr;      // the first sk_tr groups hold one tile more
    sk_base = big ? args.sk_base : args.sk_base2;
    sk_rem = big ? args.sk_rem : args.sk_rem2;
    sk_g0 = (grp * args.sk_tg + min(grp, args.sk_tr)) << args.sk_shift;      // the group's first unit
    seg_u = sk_g0 + slot * sk_base + min(slot, sk_rem);
    seg_end = seg_u + sk_base + (slot < sk_rem ? 1 : 0);
  }
  // weight panels by LDS-DMA (dma_A_unit): the experiment of round 3 (+2 % on the tile-per-workgroup kernel, retired
  // there) -- and the form the stream-K variants use: it frees the eight staging registers of the panel unit, which is
  // what keeps their main loop free of scratch reloads
  constexpr bool DMA_A = ABL == 11;
  constexpr bool WIDE_GLU = (ORD & 32) != 0;   // 16-byte gated tail (quad transpose in registers)
  int tid_ = threadIdx.x;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  do {   // one pass per segment (a single pass without SK)
  // The descriptor is read through a pointer made opaque once per segment (and once more before the tail): its ~60
  // scalar fields are invariant, and hoisted out of the segment loop they would all be live -- spilled -- across the
  // main loop (144 scalar spills and 20 vector reloads per chunk in the first build of this form).
  const ConvArgs& A = *dv3_opaque_args<SK>(&args);
  const dv3_conv_desc& p = A.d;
  const int dil = p.dil;
  const int BNH = BN + (J - 1) * dil;
  const int xbuf = 2 * KB * BNH;
  // ... and the thread index likewise: everything derived from it (staging offsets, validity bits: ~50 registers) is
  // then recomputed per segment instead of being hoisted and held across the main loop
  // (no vector register may carry it from one segment to the next: anything live through the tail -- the part with the
  // highest register pressure -- is spilled there and then RELOADED AT EVERY USE in the main loop, each reload behind a
  // vmcnt(0): the wave index is kept as a scalar, the lane index is produced by a volatile v_mbcnt pair)
  if constexpr (SK) {
    int lane_;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
    tid_ = wave_s * 64 + lane_;
  }
  const int tid = tid_;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;
  int tile_ = pid, c0_ = 0, c1_ = nchunks;
  if constexpr (SK) {
    tile_ = seg_u >> A.sk_shift;            // chunks per tile is a power of two in this form (dispatcher)
    c0_ = seg_u - tile_ * nchunks;
    c1_ = min(nchunks, c0_ + (seg_end - seg_u));
  }
  const int tile = tile_, c0 = SK ? c0_ : 0, c1 = SK ? c1_ : nchunks;
  // (stream-K form: row tiles per column tile is a power of two too -- every uniform division costs a reciprocal and a
  // zero in vector registers that the allocator then carries, spilled, through the main loop)
  const int mt = SK ? (tile & (A.m_tiles - 1)) : tile % A.m_tiles;
  const int nt = SK ? (tile >> A.sk_mshift) : tile / A.m_tiles;
  const int n0 = nt * BN;

  const bool gated = (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY);
  int h0b, h1b;
  if (gated) {
    h0b = mt * BMH; h1b = p.a_half + mt * BMH;
  } else {
    h0b = mt * BM; h1b = mt * BM + BMH;
  }

  const int Cin = p.Cin, T = p.Tout, lda = p.lda, B = p.B;
  const int Ntot = B * T;
  const int k8_total = A.kp >> 3;
  const bf16x8* __restrict__ Wh = reinterpret_cast<const bf16x8*>(p.a_split);
  const int64_t plane = (int64_t)J * k8_total * lda;   // 16-byte units per plane
  const float xscale = F16 ? (float)(1 << DV3_F16_ACT_SHIFT) : 1.0f;
  const float dscale = p.drop_scale * xscale;

  // ---- this lane's output columns: per-tap validity of the shifted read (as conv_gemm_bf16x3.hip) ----
  uint32_t vbits = 0;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
    const int bc = n / T, tc = n - bc * T;
    for (int j = 0; j < J; ++j) {
      const int ts = tc + j * dil - p.padL;
      if (n < Ntot && ts >= 0 && ts < T) vbits |= 1u << (j * NI + ni);
    }
  }
  uint32_t need = 0;
  for (int j = 0; j < J; ++j) {
    const uint32_t all = ((1u << NI) - 1u) << (j * NI);
    if (!__all((vbits & all) == all)) need |= 1u << j;
  }
  need = __builtin_amdgcn_readfirstlane(need);

  // ---- this thread's activation staging items: flat column -> (batch, time), fixed over chunks ----
  uint32_t xoff[XI];                // byte offset of (b, k8*8, t) from p.x
  const int n_items = KB * BNH;
  const uint32_t x_rsb = (uint32_t)p.x_rs * 4u;
  const uint32_t c8p = (uint32_t)((Cin + 31) / 32 * 4);
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
    const int xk8 = k8 < KB ? k8 * 8 : 0;
    xoff[i] = ((uint32_t)bf * (uint32_t)p.x_bs + (uint32_t)tf) * 4u + (uint32_t)xk8 * x_rsb;
  }
  // weight panel: per-unit column offset inside a (tap, k8) row of the split image; recomputed at each use from an
  // opaque copy of the thread index (a handful of VALU) instead of living in registers across the loop
  constexpr bool LEAN = (ORD & 2) != 0 && !SK;
  uint32_t aoff_r[2] = {0u, 0u};
  auto aoff_of = [&](int u) -> uint32_t {
    if constexpr (LEAN) return aoff_r[u];
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    const int idx = t_ + u * NT;  // k8 * BM + col
    const int col = idx % BM, k8 = idx / BM;
    const bool hi_half = col >= BMH;
    const int gcol = (hi_half ? h1b : h0b) + (col - (hi_half ? BMH : 0));
    return (uint32_t)(k8 * lda + (gcol < lda ? gcol : 0)) * 16u;
  };

  if constexpr (LEAN) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * NT;
      const int col = idx % BM, k8 = idx / BM;
      const bool hi_half = col >= BMH;
      const int gcol = (hi_half ? h1b : h0b) + (col - (hi_half ? BMH : 0));
      aoff_r[u] = (uint32_t)(k8 * lda + (gcol < lda ? gcol : 0)) * 16u;
    }
  }

  // ---- register staging ----
  bf16x8 ra[2];            // ONE weight-panel unit (hi, lo) in flight: fetched in one LOAD phase, stored in the next
  float rx[XI][8];
  uint32_t rk[MASK ? XI : 1];

  // Every load below is UNCONDITIONAL and every k16 phase of a chunk issues the same loads in the same order (the
  // tail re-fetches the last panel / chunk and stores into buffers nobody reads any more): the compiler then counts
  // its s_waitcnt vmcnt(N) exactly and a LOAD phase only ever waits for loads issued at least two LOAD phases (four
  // barrier intervals) earlier.  With loads under `if`s it falls back to the smallest N over all paths, which made
  // every phase wait for the activation fetches issued just before it (round-3 stamps: 1200 cycles of a 1500-cycle
  // LOAD phase; profiles/r03_pp2_phase_stamps.md).
  auto load_A_unit = [&](int chunk, int j, auto uc) {
    constexpr int u = decltype(uc)::value;
    const bf16x8* srch = Wh + (int64_t)(j * k8_total + chunk * KB) * lda;  // uniform
    const uint32_t ao = aoff_of(u);
    ra[0] = pp2_ldg<bf16x8>(srch, ao);
    ra[1] = pp2_ldg<bf16x8>(srch + plane, ao);
  };
  auto write_A_unit = [&](int buf, auto uc) {
    constexpr int u = decltype(uc)::value;
    bf16x8* dst = As + buf * (2 * KB * BM);
    dst[tid + u * NT] = ra[0];
    dst[KB * BM + tid + u * NT] = ra[1];
  };
  // EXPERIMENT (ABL == 11, dv3_debug_set(13, 11); compiled and inspected, NOT yet run on hardware -- round 4's first
  // measurement): the same unit straight from global memory into its LDS slot with global_load_lds_dwordx4 -- no
  // register round trip, no ds_write.  The panel image As[buf][plane][k8][BM] is indexed tid + u * NT, i.e. the 64
  // lanes of a wave own 64 CONSECUTIVE 16-byte units: exactly the layout the LDS-DMA instruction writes (M0 = the
  // wave's base, lane i lands at base + 16 i).  What the ISA of this first form shows (24 global_load_lds_dwordx4, 24
  // fewer ds_write_b128 and 7 fewer registers than the shipped loop): the workgroup-scope fence inside __syncthreads()
  // makes the compiler wait vmcnt(0) before the barrier that ends the issuing phase -- the whole memory latency once per
  // step.  The form to measure next therefore replaces that one barrier by a raw s_barrier with a COUNTED wait
  // (the 20 activation loads issued after the DMA may stay in flight: vmcnt is in order), as the uniform load schedule
  // already allows for the register path.
  auto dma_A_unit = [&](int buf, int chunk, int j, auto uc) {
    constexpr int u = decltype(uc)::value;
    const bf16x8* srch = Wh + (int64_t)(j * k8_total + chunk * KB) * lda;  // uniform
    const uint32_t ao = aoff_of(u);
    bf16x8* dst = As + buf * (2 * KB * BM) + u * NT + wave * 64;           // uniform per wave
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const char*>(srch) + ao), (lptr)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const char*>(srch + plane) + ao), (lptr)(dst + KB * BM), 16, 0, 0);
  };
  // half an item (four of its eight channel rows): one uniform base per chunk + a 32-bit per-thread offset
  auto load_X_half = [&](int chunk, auto ic, auto hc) {
    constexpr int i = decltype(ic)::value, h = decltype(hc)::value;
    const char* xb = reinterpret_cast<const char*>(p.x) + (int64_t)(chunk * BKC) * x_rsb;
    uint32_t rs = x_rsb;
    asm volatile("" : "+s"(rs));      // opaque per call site: the 24 row offsets are recomputed (one SALU + one VALU per
                                      // load), not hoisted out of the loop into 24 live registers
#pragma unroll
    for (int e = 4 * h; e < 4 * h + 4; ++e) {
      uint32_t er = (uint32_t)e * rs;
      if constexpr (SK) {
        // row 0 too as base + 32-bit register offset: as a folded `xoff + 0` the compiler addresses it with a 64-bit
        // register pair that it keeps (in this form: spills, and reloads behind a vmcnt(0)) across the loop
        if (e == 0) { er = 0; asm volatile("" : "+s"(er)); }
      }
      rx[i][e] = pp2_ldg<float>(xb, xoff[i] + er);
    }
    if constexpr (MASK && h == 1) {
      // keep-byte (b, chunk * 4 + k8, t) of this item: its offset is recomputed here (two integer divisions per item and
      // chunk) rather than held in a register across the loop
      int t_ = tid;
      asm volatile("" : "+v"(t_));
      const int idx = t_ + i * NT;
      const int k8 = idx / BNH, q = idx - k8 * BNH;
      const int f = n0 - p.padL + q;
      int bf = 0, tf = 0;
      if (idx < n_items && f >= 0 && f < Ntot) {
        bf = f / T;
        tf = f - bf * T;
      }
      const uint32_t mo = ((uint32_t)bf * c8p + (uint32_t)(k8 < KB ? k8 : 0)) * (uint32_t)T + (uint32_t)tf;
      rk[i] = (uint32_t)pp2_ldg<uint8_t>(p.xmask_c8 + (int64_t)chunk * 4 * T, mo);
    }
  };
  auto write_X_item = [&](int buf, auto ic) {
    constexpr int i = decltype(ic)::value;
    bf16x8* dst = Xs + buf * xbuf;
    const int idx = tid + i * NT;
    float v[8];
    uint32_t keep = MASK ? rk[i] : 0u;
    // stream-K form: hide that `keep` is a zero-extended byte -- the compiler tests bit 7 as a signed-byte compare against a
    // ZERO REGISTER, takes the zero from a 64-bit pair that lives (spilled) across the segment loop, and reloads the pair
    // here, in the main loop, behind a vmcnt(0)
    if constexpr (MASK && SK) asm("" : "+v"(keep));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = rx[i][e];
      if constexpr (MASK) v[e] *= ((keep >> e) & 1u) ? dscale : 0.f;
      else if (F16) v[e] *= xscale;
    }
    bf16x8 hi, lo;
    if constexpr (PW) {
      uint32_t w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(rx[i][e]);
      dv3_pair_units(w, hi, lo);
    } else if constexpr (F16 && LEAN) {
      // the range test first (it is made anyway); a wave whose eight values all sit inside the fp16 range -- every wave of
      // a healthy run -- builds the pair without the clamps (identity there) and takes the residual as one fused
      // multiply-add per element: a - hi is exact in fp32 either way, so the pair is bit-identical to dv3_split8_f16's
      const float m0 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));
      const float m1 = fmaxf(fmaxf(fabsf(v[3]), fabsf(v[4])), fabsf(v[5]));
      const float m = fmaxf(fmaxf(m0, m1), fmaxf(fabsf(v[6]), fabsf(v[7])));
      const bool bad = !(m <= 65504.f);
      if (__builtin_expect(__any(bad), 0)) {
        dv3_note_range(A.range_ctr, dv3_split8_f16(v, hi, lo));
      } else {
        f16x8 h8, l8;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 f = {v[e], v[e + 1]};
          typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
          const f16x2_ h = __builtin_convertvector(f, f16x2_);
          h8[e] = h[0]; h8[e + 1] = h[1];
          if constexpr ((ORD & 8) != 0) {
            // lo = fp16_rn(a - hi) as ONE mixed-precision fma per element (hi read as the fp16 half it is, a as fp32; the
            // fp32 result a - hi is exact, so the single rounding is the one the convert / subtract / convert chain makes)
            uint32_t l2 = 0u;
            const uint32_t h2 = __builtin_bit_cast(uint32_t, h);
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(l2) : "v"(h2), "v"(v[e]));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l2) : "v"(h2), "v"(v[e + 1]));
            const f16x2_ l = __builtin_bit_cast(f16x2_, l2);
            l8[e] = l[0]; l8[e + 1] = l[1];
          } else {
            l8[e] = (_Float16)(v[e] - (float)h[0]);
            l8[e + 1] = (_Float16)(v[e + 1] - (float)h[1]);
          }
        }
        hi = __builtin_bit_cast(bf16x8, h8);
        lo = __builtin_bit_cast(bf16x8, l8);
      }
    } else if constexpr (F16) dv3_note_range(A.range_ctr, dv3_split8_f16(v, hi, lo)); else pp2_split8(v, hi, lo);
    if (idx < n_items) {
      dst[idx] = hi;
      dst[KB * BNH + idx] = lo;
    }
  };
  // ORD & 64: the same conversion + store as straight-line code (it runs between the MFMAs of a COMPUTE phase: a branch
  // would cut the scheduling region).  The range events of the wave are counted in a scalar (popcount of the ballot) and
  // added to the sticky counter once, after the main loop; lanes without an item store to two spare units behind the
  // activation buffers instead of being masked off.
  uint32_t sbad = 0;
  auto write_X_item_nb = [&](int buf, auto ic) {
    constexpr int i = decltype(ic)::value;
    bf16x8* dst = Xs + buf * xbuf;
    const int idx = tid + i * NT;
    float v[8];
    const uint32_t keep = MASK ? rk[i] : 0u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = rx[i][e];
      if constexpr (MASK) {
        v[e] *= ((keep >> e) & 1u) ? dscale : 0.f;
        // the product is ROUNDED before the pair is taken (write_X_item's arithmetic): without this the compiler contracts
        // x * scale - hi into one fma here, i.e. takes the residual of the unrounded product (1 / (1 - p) is no power of two)
        asm("" : "+v"(v[e]));
      } else if (F16) v[e] *= xscale;
    }
    bf16x8 hi, lo;
    if constexpr (PW) {
      uint32_t w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(rx[i][e]);
      dv3_pair_units(w, hi, lo);
    } else if constexpr (F16) {
      const bool bad = dv3_split8_f16(v, hi, lo);
      sbad += (uint32_t)__builtin_popcountll(__ballot(bad));
    } else {
      pp2_split8(v, hi, lo);
    }
    const int spare = 2 * xbuf - buf * xbuf;          // unit index (from dst) of the two spare units
    const bool live = i < 2 || idx < n_items;          // items 0 and 1 always exist (KB * BNH >= 2 * NT)
    dst[live ? idx : spare] = hi;
    dst[live ? KB * BNH + idx : spare + 1] = lo;
  };
  using U0 = std::integral_constant<int, 0>;
  using U1 = std::integral_constant<int, 1>;
  using U2 = std::integral_constant<int, 2>;
  constexpr bool CVC = (ORD & 64) != 0 && !SK;          // activation items converted in COMPUTE phases
  static_assert(XI == 3 && JT == 3, "three activation items per thread, one per tap's pair of k16 phases");

  f32x16 acc[MI][2][NI];   // [row sub-tile][a rows | gate rows][column sub-tile]
  float zero_ = 0.f;
  if constexpr (SK) asm volatile("" : "+v"(zero_));   // per segment, opaque: a constant zero block would be hoisted out of
                                                      // the segment loop and held (spilled) across the main loop
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][h][ni][r] = zero_;

  const int a_off = wm * (MI * 32) + l31;
  const int x_off = wn * (NI * 32) + l31;

  // ---- prologue: step 0's panel and chunk 0's tile into buffer 0; then the fetches that run a step / a chunk ahead,
  //      issued in the order the loop issues them ----
  load_A_unit(c0, 0, U0{});
  write_A_unit(0, U0{});
  load_A_unit(c0, 0, U1{});
  load_X_half(c0, U0{}, U0{}); load_X_half(c0, U0{}, U1{});
  load_X_half(c0, U1{}, U0{}); load_X_half(c0, U1{}, U1{});
  load_X_half(c0, U2{}, U0{}); load_X_half(c0, U2{}, U1{});
  write_A_unit(0, U1{});
  write_X_item(0, U0{});
  write_X_item(0, U1{});
  write_X_item(0, U2{});
  __syncthreads();
  {
    const int cp = min(c0 + 1, c1 - 1);
    if (!DMA_A) load_A_unit(c0, 1, U0{});          // unit 0 of step 1's panel: stored by the first LOAD phase
    load_X_half(cp, U0{}, U0{});
    load_X_half(cp, U0{}, U1{});
    load_X_half(cp, U1{}, U0{}); load_X_half(cp, U1{}, U1{});
    load_X_half(cp, U2{}, U0{});
    if constexpr (!CVC) load_X_half(cp, U2{}, U1{});   // (CVC: fetched by the first LOAD phase of the loop)
  }

  // ---- ping-pong main loop: waves w and w+4 share a SIMD and run the same phase sequence one phase apart ----
  //   interval:   I0        I1        I2        I3
  //   waves 0-3:  L(0)      C(0)      L(1)      C(1) ...
  //   waves 4-7:  -         L(0)      C(0)      L(1) ...
  // Phase q = 2 * tap + s of a chunk (s = k16 block of the 32-channel chunk).  LDS hazards: the panel of step t+1 is
  // stored during the four L phases of step t (two per half) into the buffer last read in the L phases of step t-1 and
  // first read in L of step t+1; the tile of chunk c+1 during the L phases of chunk c into the buffer last read in
  // chunk c-1.  Every interval ends with a workgroup barrier.
  // ORD & 4: touch the residual rows of this tile during the last chunk (one 4-byte LDS-DMA load per 128-byte line and
  // lane into a 2 KB scratch strip behind the operand buffers -- nothing returns to a register, nothing reads the strip):
  // thread -> (row = tid / 4 of the 128 `a` rows, column segment tid % 4 of 64 columns); two loads per thread
  const bool pf_on = (ORD & 4) != 0 && !SK && gated && p.r != nullptr;
  const int late = wave >> 2;
  stamp();                               // slot 1: prologue done
  if (late) __syncthreads();
  for (int c = c0; c < c1; ++c) {
    const int cr = c - c0;                           // buffer parities count from the segment's first chunk
    const bf16x8* XsH = Xs + (cr & 1) * xbuf;
    const bf16x8* XsL = XsH + KB * BNH;
    const int cx = min(c + 2, c1 - 1);               // the chunk fetched during this one (the tail re-fetches the last)
    const bool last_chunk = c + 1 == c1;
#pragma unroll
    for (int q = 0; q < 2 * JT; ++q) {
      constexpr int dummy = 0; (void)dummy;
      const int j = q >> 1, s = q & 1;
      const int cur = (cr * JT + j) & 1;
      const bf16x8* AsH = As + cur * (2 * KB * BM);
      const bf16x8* AsL = AsH + KB * BM;
      const bool fix = (need >> j) & 1u;
      stamp();                           // 2 + 5*phase: LOAD begins
      // ---------------- LOAD ----------------
      // Order inside the phase: staging first, fragment reads last (pinned with sched_barrier) -- the twelve fragments
      // (48 registers) are dead until then, which keeps the conversion temporaries of the staging inside the
      // 256-register budget next to the 128 accumulator registers.
      bf16x8 ah[MI][2], al[MI][2], bh[NI], bl[NI];
      auto read_frags = [&](bool do_a, bool do_b) {
        const int k8 = 2 * s + lhi;
        if (do_a) {
          const int ai = k8 * BM + a_off;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            ah[mi][0] = AsH[ai + mi * 32];
            ah[mi][1] = AsH[ai + mi * 32 + BMH];
            al[mi][0] = AsL[ai + mi * 32];
            al[mi][1] = AsL[ai + mi * 32 + BMH];
          }
        }
        if (do_b) {
          const int xi = k8 * BNH + x_off + j * dil;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            bh[ni] = XsH[xi + ni * 32];
            bl[ni] = XsL[xi + ni * 32];
          }
        }
      };
      auto fix_frags = [&]() {
        const int k8 = 2 * s + lhi; (void)k8;
        if (fix) {
          const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const bool ok = (vbits >> (j * NI + ni)) & 1u;
            bh[ni] = ok ? bh[ni] : zero8;
            bl[ni] = ok ? bl[ni] : zero8;
          }
        }
      };
      // ORD & 16 (with 1): only the eight weight fragments go first, the four activation fragments follow the staging (16
      // registers less are live beside the conversion: what the masked instantiation needs to stay free of scratch reloads)
      if constexpr ((ORD & 1) != 0) {
        read_frags(true, (ORD & 16) == 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ABL != 2) {
        // panel unit s of the next step: store (it was fetched in this wave's previous LOAD phase), then fetch the
        // unit the NEXT LOAD phase stores: one unit (8 registers) in flight, two barrier intervals to land (an L2 hit)
        int jn = j + 1, cn = c;
        if (jn >= JT) { jn = 0; cn = c + 1; }
        int j2 = jn + 1, c2 = cn;
        if (j2 >= JT) { j2 = 0; c2 = cn + 1; }
        if (cn >= c1) { cn = c; jn = j; }               // past the end: re-fetch the current panel
        if (c2 >= c1) { c2 = c; j2 = j; }
        if constexpr (DMA_A) {
          // both units of the NEXT step's panel by LDS-DMA in the step's first phase: their buffer (cur ^ 1) was last
          // read in the previous step and is first read two LOAD phases from now (the buffer of the step after is
          // still being read during this one, so nothing can be sent there yet)
          if (s == 0) {
            dma_A_unit(cur ^ 1, cn, jn, U0{});
            dma_A_unit(cur ^ 1, cn, jn, U1{});
          }
          (void)c2; (void)j2;
        } else if (s == 0) {
          if (ABL != 9) write_A_unit(cur ^ 1, U0{});
          if (ABL != 7) load_A_unit(cn, jn, U1{});       // unit 1 of the next step's panel
        } else {
          if (ABL != 9) write_A_unit(cur ^ 1, U1{});
          if (ABL != 7) load_A_unit(c2, j2, U0{});       // unit 0 of the panel after
        }
        // activation item j of the next chunk: convert + store in the tap's first phase, fetch its halves for the
        // chunk after in the tap's two phases
        constexpr bool WX = ABL != 8, LX = ABL != 6;      // timing-only ablations: no conversion + store / no fetch
        if constexpr (CVC) {
          // conversion + store of item i in the COMPUTE phase 2 i (below); its halves for the chunk after come in the two
          // LOAD phases that follow it (the second half of item 2 in the next chunk's first phase: cx1 = that chunk's successor)
          const int cx1 = min(c + 1, c1 - 1);
          if (q == 0) load_X_half(cx1, U2{}, U1{});
          if (q == 1) load_X_half(cx, U0{}, U0{});
          if (q == 2) load_X_half(cx, U0{}, U1{});
          if (q == 3) load_X_half(cx, U1{}, U0{});
          if (q == 4) load_X_half(cx, U1{}, U1{});
          if (q == 5) load_X_half(cx, U2{}, U0{});
        } else {
        if (q == 0) { if (WX) write_X_item((cr + 1) & 1, U0{}); if (LX) load_X_half(cx, U0{}, U0{}); }
        if (q == 1) { if (LX) load_X_half(cx, U0{}, U1{}); }
        if (q == 2) { if (WX) write_X_item((cr + 1) & 1, U1{}); if (LX) load_X_half(cx, U1{}, U0{}); }
        if (q == 3) { if (LX) load_X_half(cx, U1{}, U1{}); }
        if (q == 4) { if (WX) write_X_item((cr + 1) & 1, U2{}); if (LX) load_X_half(cx, U2{}, U0{}); }
        if (q == 5) { if (LX) load_X_half(cx, U2{}, U1{}); }
        }
        if constexpr ((ORD & 4) != 0 && !SK) {
          if ((q == 1 || q == 3) && last_chunk && pf_on) {
            typedef const __attribute__((address_space(1))) void* gptr;
            typedef __attribute__((address_space(3))) void* lptr;
            unsigned char* strip = smem_raw + (size_t)(2 * 2 * KB * BM + 2 * xbuf) * 16 + wave * 256;
            // (the offset is computed here, twice per tile, from an opaque copy of the thread index: nothing lives in a
            // register across the loop for it)
            int t_ = tid;
            asm volatile("" : "+v"(t_));
            const int row = t_ >> 2, col = (t_ & 3) * 64 + (q >> 1) * 32;
            const int ch = min(mt * BMH + row, p.Cg - 1);
            const int n = min(n0 + col, Ntot - 1);
            const int bb = n / T, tt = n - bb * T;
            const uint32_t po = ((uint32_t)bb * (uint32_t)p.r_bs + (uint32_t)ch * (uint32_t)p.r_rs + (uint32_t)tt) * 4u;
            __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const char*>(p.r) + po), (lptr)strip, 4, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp();                           // +1: staging issued
      if constexpr ((ORD & 1) == 0) read_frags(true, true);
      else if constexpr ((ORD & 16) != 0) read_frags(false, true);
      fix_frags();
      if constexpr (ABL == 4) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();                         // +2: fragments landed
      }
      __syncthreads();
      // The MFMAs are register-only instructions: without the two scheduling fences the compiler sinks them below the
      // second barrier into the next LOAD phase -- legal, but then both waves of a SIMD issue their MFMAs in the same
      // barrier interval and stage in the same interval, i.e. the ping-pong degenerates into the in-phase loop.
      if (ABL != 5) __builtin_amdgcn_sched_barrier(0);
      stamp();                           // +3: COMPUTE begins
      // ---------------- COMPUTE: 24 MFMAs of one k16 block ----------------
      if (ABL != 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][0][ni] = pp2_mma<F16>(al[mi][0], bh[ni], acc[mi][0][ni]);
            acc[mi][1][ni] = pp2_mma<F16>(al[mi][1], bh[ni], acc[mi][1][ni]);
          }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][0][ni] = pp2_mma<F16>(ah[mi][0], bl[ni], acc[mi][0][ni]);
            acc[mi][1][ni] = pp2_mma<F16>(ah[mi][1], bl[ni], acc[mi][1][ni]);
          }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][0][ni] = pp2_mma<F16>(ah[mi][0], bh[ni], acc[mi][0][ni]);
            acc[mi][1][ni] = pp2_mma<F16>(ah[mi][1], bh[ni], acc[mi][1][ni]);
          }
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(ah[mi][0]), "v"(ah[mi][1]), "v"(al[mi][0]), "v"(al[mi][1]));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(bh[ni]), "v"(bl[ni]));
      }
      if constexpr (CVC) {
        // the conversion + store of activation item q / 2 for the next chunk's tile, spread between the MFMAs: two vector
        // ALU instructions behind each of the first 22, the two stores behind the last two (the matrix pipe takes one
        // instruction per 32 cycles from this wave: the conversion rides in the issue slots it leaves)
        if (q == 0) write_X_item_nb((cr + 1) & 1, U0{});
        if (q == 2) write_X_item_nb((cr + 1) & 1, U1{});
        if (q == 4) write_X_item_nb((cr + 1) & 1, U2{});
        if ((q & 1) == 0) {
#pragma unroll
          for (int m = 0; m < 22; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // three VALU
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
      stamp();                           // +4: MFMAs issued
      if (ABL != 5) __builtin_amdgcn_sched_barrier(0);
      if (!(last_chunk && q == 2 * JT - 1) || !late) __syncthreads();
    }
  }
  stamp();                               // main loop left
  if constexpr (CVC && F16) {
    if (sbad != 0 && lane == 0 && A.range_ctr) atomicAdd(A.range_ctr, sbad);
  }

  bool run_tail = true;
  if constexpr (SK) {
    constexpr int ACC = MI * 2 * NI * 16;            // accumulator registers per thread (128)
#ifdef DV3_EXPERIMENTS
    const int sk_abl = A.sk_abl;                     // timing-only ablations (dv3_debug_set(26, bits)): experiment build only
#else
    constexpr int sk_abl = 0;
#endif
    float* const ws = A.sk_ws;
    int* const flags = A.sk_flags;
    if (c0 != 0) {
      // a tile another workgroup began: hand the accumulators over.  Image [32 groups of 4 registers][512 threads][4]:
      // one 16-byte store / load per thread and group, consecutive threads consecutive (layout-agnostic)
      char* dst = reinterpret_cast<char*>(ws) + ((size_t)pid * (ACC * NT) + (size_t)tid * 4) * 4;
      if (!(sk_abl & 1))
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              typedef float f32x4_ __attribute__((ext_vector_type(4)));
              const f32x4_ v4 = {acc[mi][h][ni][r4 * 4], acc[mi][h][ni][r4 * 4 + 1], acc[mi][h][ni][r4 * 4 + 2], acc[mi][h][ni][r4 * 4 + 3]};
              const char* d4 = dst + (size_t)((((mi * 2 + h) * NI + ni) * 4 + r4) * NT) * 16;
              asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d4), "v"(v4) : "memory");
            }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // The consumer may sit on another XCD (its own L2).  The image goes out as system-scope stores (sc0 sc1: written
      // through to the memory side) and only their completion is awaited before the flag -- an agent-scope RELEASE fence instead
      // would write back the whole L2 of this XCD (buffer_wbl2), once per wave, and the matching ACQUIRE would invalidate
      // the consumer's: the first build of this form lost 30-70 us per launch to exactly that.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + pid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      run_tail = false;
    } else if (c1 < nchunks) {
      // this workgroup began the tile: add the parts of the workgroups that follow it (their FIRST segments, finished
      // long before this one -- the last of this workgroup's range), in workgroup order (deterministic sum)
      const int qm = (1 << A.sk_qshift) - 1;
      for (int w2 = pid + 1; (w2 & qm) != 0; ++w2) {          // (the group's last workgroup ends on a tile boundary)
        const int s2 = sk_g0 + (w2 & qm) * sk_base + min(w2 & qm, sk_rem);
        if (s2 >= (tile + 1) * nchunks) break;
        if (tid == 0 && !(sk_abl & 4))
          while (__hip_atomic_load(flags + w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the image comes in by LDS-DMA (global_load_lds_dwordx4, sc0 sc1: read at the memory side), 16 loads per thread in
        // flight and no staging registers, in two halves of 128 KB (every LDS read of the main loop is behind the barrier
        // above); each lane reads back the 16 bytes the DMA put at its own slot
        const char* src = reinterpret_cast<const char*>(ws) + ((size_t)w2 * (ACC * NT) + (size_t)tid * 4) * 4;
        typedef const __attribute__((address_space(1))) void* gptr_;
        typedef __attribute__((address_space(3))) void* lptr_;
        if (!(sk_abl & 2))
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int jq = 0; jq < 16; ++jq) {
            const int qg = half * 16 + jq;
            __builtin_amdgcn_global_load_lds((gptr_)(src + (size_t)(qg * NT) * 16), (lptr_)(smem_raw + (jq * NT + wave * 64) * 16), 16, 0, 17);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int jq = 0; jq < 16; ++jq) {
            const int qg = half * 16 + jq;
            typedef float f32x4_ __attribute__((ext_vector_type(4)));
            const f32x4_ t4 = *reinterpret_cast<const f32x4_*>(smem_raw + (jq * NT + tid) * 16);
            const int blk = qg >> 2, r4 = qg & 3;
            const int mi = blk / (2 * NI), h = (blk / NI) & 1, ni = blk % NI;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mi][h][ni][r4 * 4 + e] += t4[e];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the second half lands in the same slots
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + w2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
    }
  }

  // ---- fused tail (conv_common.h), one 32-row sub-tile at a time ----
  // Inside the segment loop of the stream-K form the tail reads the descriptor through an opaque pointer: its ~40
  // scalar fields are loop invariant, and hoisted out of the segment loop they would be live (and spilled) across the
  // main loop.
  const ConvArgs& AT = *dv3_opaque_args<SK>(&args);
  const dv3_conv_desc& pt = AT.d;
  if (run_tail && (ABL != 3 || acc[0][0][0][0] + acc[1][1][1][7] == 1.2345e30f)) {
    if constexpr (F16) {   // the accumulators carry 2^(weight shift + activation shift) x the result
      constexpr float kInv = 1.0f / (float)(1 << (DV3_F16_WEIGHT_SHIFT + DV3_F16_ACT_SHIFT));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][h][ni][r] *= kInv;
    }
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
    if constexpr (FG) {
      const int nw0 = n0e + wn * (NI * 32);
      if (dv3_wide_gate_ok(pt, AT.wide)) {
        float* wl = reinterpret_cast<float*>(smem_raw) + wave * (DV3_WIDE_LDS / 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int row0 = wm * (MI * 32) + mi * 32;
          conv_epilogue_wide_block_gate<BM, BMH>(pt, acc[mi][0], mt, row0, 0, lane, nw0, Ntot, wl);
          conv_epilogue_wide_block_gate<BM, BMH>(pt, acc[mi][1], mt, row0, 1, lane, nw0, Ntot, wl);
        }
      } else {
        conv_epilogue_dgrad_gate<BM, BMH, NI>(pt, acc[0], mt, wm * (MI * 32), lhi, l31, bcol, tcol, okc, nw0 >> 5);
        conv_epilogue_dgrad_gate<BM, BMH, NI>(pt, acc[1], mt, wm * (MI * 32) + 32, lhi, l31, bcol, tcol, okc, nw0 >> 5);
      }
    } else
    if (ABL != 10 && dv3_wide_epilogue_ok(pt, AT.wide)) {
      // 16-byte epilogue through LDS (conv_common.h): every LDS read of the main loop is behind the last barrier this
      // wave passed, so the whole allocation is free; each wave transposes in its own 8.5 KB
      float* wl = reinterpret_cast<float*>(smem_raw) + wave * (DV3_WIDE_LDS / 4);
      const int nw0 = n0e + wn * (NI * 32);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row0 = wm * (MI * 32) + mi * 32;
        conv_epilogue_wide_block<BM, BMH>(pt, acc[mi][0], mt, row0, 0, lane, nw0, Ntot, wl);
        conv_epilogue_wide_block<BM, BMH>(pt, acc[mi][1], mt, row0, 1, lane, nw0, Ntot, wl);
      }
    } else if (WIDE_GLU && dv3_wide_glu_ok(pt)) {
      // 16-byte Conv1dGLU / highway tail through a quad transpose in registers (conv_common.h)
      constexpr int TA = ABL == 12 ? 12 : ABL == 13 ? 13 : 0;
      const int nw0 = n0e + wn * (NI * 32);
      conv_epilogue_glu_wide<BMH, NI, TA>(pt, acc[0], mt, wm * (MI * 32), lane, nw0, Ntot);
      conv_epilogue_glu_wide<BMH, NI, TA>(pt, acc[1], mt, wm * (MI * 32) + 32, lane, nw0, Ntot);
    } else {
      constexpr int TA = ABL == 12 ? 7 : ABL == 13 ? 8 : 0;     // experiment build: 12 = no residual load, 13 = no stores
      // round 6: a wave whose 64 rows and 64 columns all lie inside the tensor takes the straight-line gated tail
      // (conv_common.h: conv_epilogue_glu_interior; every test below is wave-uniform)
      // (not in the stream-K instantiations: their tail sits inside the segment loop and any addition to it costs the
      //  main loop its registers -- 70 -> 2 500 spills with this one)
      bool interior = false;
      if constexpr (!SK)
        interior = TA == 0 && AT.fast_tail && gated && !pt.spk && pt.store_mode == DV3_STORE_BCT &&
                   mt * BMH + wm * (MI * 32) + MI * 32 <= pt.Cg && n0e + wn * (NI * 32) + NI * 32 <= Ntot;
      if (!SK && interior) {
        conv_epilogue_glu_interior<BMH, NI>(pt, acc[0], mt, wm * (MI * 32), lhi, bcol, tcol);
        conv_epilogue_glu_interior<BMH, NI>(pt, acc[1], mt, wm * (MI * 32) + 32, lhi, bcol, tcol);
      } else {
        conv_epilogue<BM, BMH, NI, TA, false>(pt, acc[0], gated, mt, wm * (MI * 32), lhi, bcol, tcol, okc);
        conv_epilogue<BM, BMH, NI, TA, false>(pt, acc[1], gated, mt, wm * (MI * 32) + 32, lhi, bcol, tcol, okc);
      }
    }
  }
  stamp();                               // tail stores issued
  if constexpr (SK) {
    seg_u += c1 - c0;
    __syncthreads();                     // the next segment's prologue overwrites the LDS the wide tail used
  }
  } while (SK && seg_u < seg_end);
}

#ifndef DV3_PP2_ISA_ONLY   // (developer: compile one instantiation for ISA inspection, scripts/pp2_isa.sh)
template <bool MASK, bool F16, int ABL = 0, bool SK = false, int ORD = 0, bool PW = false, bool FG = false>
int launch_pp2(const ConvArgs& a, size_t lds, hipStream_t st) {
  if ((ORD & 4) != 0 && lds + 8 * 256 > 160 * 1024) return launch_pp2<MASK, F16, ABL, SK, (ORD & ~4), PW, FG>(a, lds, st);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_pp2_kernel<MASK, F16, ABL, SK, ORD, PW, FG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      dv3_set_error("conv_gemm_pp2: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  if ((ORD & 4) != 0) lds += 8 * 256;   // the prefetch strip
  else if ((ORD & 64) != 0) lds += 32;   // two spare units behind the activation buffers
  hipLaunchKernelGGL((conv_gemm_pp2_kernel<MASK, F16, ABL, SK, ORD, PW, FG>), dim3(a.n_blocks), dim3(NT), lds, st, a);
  return dv3_check_launch("conv_gemm_pp2");
}

#endif  // DV3_PP2_ISA_ONLY
}  // namespace

#ifndef DV3_PP2_ISA_ONLY
int dv3_pp2_read_stamps(void* dst, int64_t bytes) {
  if (bytes <= 0 || bytes > (int64_t)sizeof(unsigned long long) * 8 * PP2_STAMPS * 2) return DV3_EINVAL;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pp2_stamps), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? DV3_OK : DV3_ELAUNCH;
}
extern int g_x3_wide;
int g_pp2_ord = 0;   // experiment build: dv3_debug_set(29, v) with v outside {0, 17, 81}: further ORD variants of the fp16-pair kernel
// dv3_debug_set(29 / 31, v): ORD of the unmasked / masked instantiations (0, 17, 81).  Round 5 default 81 (weight
// fragments first, activation items converted between the MFMAs): bit-identical to ORD 0, measured in one process at the
// north-star shape (scripts/r5_ship_check.py, profiles/r05_pp2_load_phase.txt): fp16-pair forward 152.3 -> 148.8 us, masked
// training forward 169.8 -> 166.9 us, bf16-pair input gradient 134.3 -> 132.3 us.  The masked bf16-pair instantiation
// (legacy bf16x3 mode) keeps ORD 0: there the compiler contracts the mask multiply differently in the two forms.
int g_pp2_ord_u = 81, g_pp2_ord_m = 81;
int g_pp2_fast_tail = 1;   // dv3_debug_set(50, v): interior sub-tiles of a gated launch take the straight-line tail (0 = the guarded tail everywhere)
int g_pp2_abl = 0;   // dv3_debug_set(13, v): timing-only ablations of the unmasked kernel (1 no MFMAs, 2 no staging, 3 no tail)

// Stream-K (round 4).  A 256 x 256 tile grid rarely divides the 256 CUs: the encoder layers of the benchmark step are 152
// tiles (41 % of the chip idle for the whole launch), the 512-channel converter layers 808 (3.16 rounds: the last one a
// sixth full), the decoder's 102.  With a caller-provided workspace (dv3_conv_desc.sk_ws) the launch is instead ONE
// workgroup per CU, each walking an equal share of the (tile, 32-channel chunk) units; a tile cut between workgroups
// is summed by the one that holds its first chunk.  Taken when the share (+ g_pp2_sk_overhead chunks for the extra
// prologue, the hand-over and the wait) is below g_pp2_sk_gain % of the tile-per-workgroup schedule's chunks per CU.
// Results differ from the tile-per-workgroup launch by the fp32 summation order of the cut tiles only (deterministic:
// the cut is a function of the shape).
int g_pp2_sk = 1;            // dv3_debug_set(22, v): 0 never, 1 by the rule above (forward launches), 2 whenever a workspace is given,
                             // 3 by the rule in both directions
int g_pp2_sk_overhead = 2;   // dv3_debug_set(23, v)
int g_pp2_sk_gain = 80;      // dv3_debug_set(24, v)
int g_pp2_sk_abl = 0;        // dv3_debug_set(26, v): timing-only ablations (1 no hand-over stores, 2 no hand-over loads, 4 no wait)
static int pp2_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}
int64_t dv3_pp2_sk_ws_bytes() { return (int64_t)pp2_cu_count() * (128 * NT * 4 + 64); }
extern "C" int dv3_conv_streamk_ws_bytes(void) { return (int)dv3_pp2_sk_ws_bytes(); }

// Shapes this kernel takes (called by dv3_conv_gemm_bf16x3_dispatch): three-term split operands, fp32 (B, C, T)
// activations, dropout as keep-bytes.  Returns 1 when not eligible.
int dv3_conv_gemm_pp2_dispatch(const dv3_conv_desc* d, hipStream_t st) {
  if (d->split_terms == 1 || !d->a_split) return 1;
  if (d->xmask && !d->xmask_c8) return 1;            // keep-bits only: the 128 x 256 kernel stages those
  if (d->J != JT || (d->J - 1) * d->dil > HALO_MAX || (d->Cin & 31)) return 1;
  if (d->a_bs != 0 || (d->lda & 3) || d->Tin != d->Tout) return 1;
  const bool gated = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  const bool f16 = d->split_terms == DV3_SPLIT_F16X3;
  const int BNH = BN + (d->J - 1) * d->dil;
  const size_t lds = (size_t)(2 * 2 * KB * BM + 2 * 2 * KB * BNH) * 16;
  if (lds > 160 * 1024) return 1;
  ConvArgs a;
  a.d = *d;
  a.a_scalar = 0;
  a.wide = g_x3_wide;
  a.fast_tail = g_pp2_fast_tail;
  a.range_ctr = f16 ? dv3_range_ctr() : nullptr;
  a.kp = (d->Cin + 31) / 32 * 32;
  a.m_tiles = gated ? dv3_cdiv(d->Cg, BMH) : dv3_cdiv(d->M, BM);
  a.n_tiles = (int)dv3_cdiv64((int64_t)d->B * d->Tout, BN);
  const int64_t nb = (int64_t)a.m_tiles * a.n_tiles;
  DV3_REQUIRE(nb < (1ll << 31), "conv_gemm: grid too large");
  a.n_blocks = (int)nb;
  g_dv3_last_conv = (f16 ? 5000 : 3000) + 100 + 1;     // tile id 10, ping-pong
  const bool mask = d->xmask_c8 != nullptr;
  if (d->x_pair || d->pg) {
    // round 6: pair-word input and / or the producer's gate backward in the tail: the bf16-pair kernel without dropout,
    // tile per workgroup, the default LOAD-phase structure (ORD 81)
    if (mask || f16) return 1;
    g_dv3_last_conv += (d->x_pair ? 5 : 0) + (d->pg ? 10 : 0);   // ...106 pair-word staging, 111 fused gate backward, 116 both
    if (d->pg) return d->x_pair ? launch_pp2<false, false, 0, false, 81, true, true>(a, lds, st)
                                : launch_pp2<false, false, 0, false, 81, false, true>(a, lds, st);
    return launch_pp2<false, false, 0, false, 81, true>(a, lds, st);
  }
  {
    const int P = pp2_cu_count(), S = d->Cin / BKC;
    const int64_t units = nb * S;
    const int64_t dp = dv3_cdiv64(nb, P) * S, sk = dv3_cdiv64(units, P) + g_pp2_sk_overhead;
    const bool ws_ok = d->sk_ws && d->sk_ws_bytes >= dv3_pp2_sk_ws_bytes() && units < (1ll << 30);
    // Forward launches only by default (g_pp2_sk 1): the input-gradient launches of a training step run beside the
    // weight-gradient stream, whose workgroups take the CUs a short grid leaves idle -- there the one-workgroup-per-CU
    // form gains nothing for the step (measured: forward alone -1.7 %, whole step +-0.1 % with both directions in this
    // form, scripts/r4_sk_step_ab.py).  3 = both directions.
    const bool dir_ok = d->mode != DV3_EPI_DGRAD || g_pp2_sk >= 2;
    // per-XCD unit ranges: P / 8 workgroups (a power of two) per group, every group at least one tile
    const int q = P / 8;
    const bool grp_ok = (P % 8) == 0 && q > 0 && (q & (q - 1)) == 0 && nb >= 8;
    if (ws_ok && dir_ok && grp_ok && g_pp2_sk && (S & (S - 1)) == 0 && (a.m_tiles & (a.m_tiles - 1)) == 0 && units >= 2 * P && (g_pp2_sk == 2 || sk * 100 < dp * g_pp2_sk_gain)) {
      a.sk_abl = g_pp2_sk_abl;
      a.sk_units = (int)units;
      a.sk_tg = (int)(nb / 8);
      a.sk_tr = (int)(nb % 8);
      a.sk_qshift = __builtin_ctz((unsigned)q);
      const int64_t ub = (int64_t)(a.sk_tg + 1) * S, us = (int64_t)a.sk_tg * S;     // units of a group with / without the extra tile
      a.sk_base = (int)(ub / q);
      a.sk_rem = (int)(ub % q);
      a.sk_base2 = (int)(us / q);
      a.sk_rem2 = (int)(us % q);
      a.sk_shift = __builtin_ctz((unsigned)S);
      a.sk_mshift = __builtin_ctz((unsigned)a.m_tiles);
      a.n_blocks = P;
      a.sk_flags = reinterpret_cast<int*>(d->sk_ws);
      a.sk_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(d->sk_ws) + (size_t)P * 64);
      g_dv3_last_conv += 1;                            // ...102: stream-K form
      const size_t lds_sk = lds;
      if (f16) return mask ? launch_pp2<true, true, 0, true>(a, lds_sk, st) : launch_pp2<false, true, 0, true>(a, lds_sk, st);
      return mask ? launch_pp2<true, false, 0, true>(a, lds_sk, st) : launch_pp2<false, false, 0, true>(a, lds_sk, st);
    }
  }
  {
    // LOAD-phase structure (ORD, see the kernel): measured variants, selectable at run time (dv3_debug_set(29, v) for the
    // unmasked instantiations, (31, v) for the masked ones); anything else needs the experiment build
    const int o = mask ? (f16 ? g_pp2_ord_m : 0) : g_pp2_ord_u;
#define DV3_PP2_ORD_SHIP(oo) \
    if (o == oo) { \
      if (f16) return mask ? launch_pp2<true, true, 0, false, oo>(a, lds, st) : launch_pp2<false, true, 0, false, oo>(a, lds, st); \
      return mask ? launch_pp2<true, false, 0, false, oo>(a, lds, st) : launch_pp2<false, false, 0, false, oo>(a, lds, st); \
    }
    DV3_PP2_ORD_SHIP(17) DV3_PP2_ORD_SHIP(81)
#undef DV3_PP2_ORD_SHIP
  }
#ifdef DV3_EXPERIMENTS
  if (g_pp2_ord && f16) {
    switch (g_pp2_ord) {
#define DV3_PP2_ORD_CASE(o) case o: return mask ? launch_pp2<true, true, 0, false, o>(a, lds, st) : launch_pp2<false, true, 0, false, o>(a, lds, st);
#define DV3_PP2_ORD_CASE_U(o) case o: if (!mask) return launch_pp2<false, true, 0, false, o>(a, lds, st); break;
      DV3_PP2_ORD_CASE_U(1) DV3_PP2_ORD_CASE_U(2) DV3_PP2_ORD_CASE_U(3) DV3_PP2_ORD_CASE_U(4) DV3_PP2_ORD_CASE_U(5) DV3_PP2_ORD_CASE_U(7)
      DV3_PP2_ORD_CASE_U(10) DV3_PP2_ORD_CASE_U(11) DV3_PP2_ORD_CASE_U(15)
      DV3_PP2_ORD_CASE(19) DV3_PP2_ORD_CASE(27) DV3_PP2_ORD_CASE(31) DV3_PP2_ORD_CASE(32) DV3_PP2_ORD_CASE(49) DV3_PP2_ORD_CASE(64) DV3_PP2_ORD_CASE(65)
#undef DV3_PP2_ORD_CASE_U
#undef DV3_PP2_ORD_CASE
    }
  }
  if (g_pp2_abl && !mask && f16) {
    switch (g_pp2_abl) {
      case 1: return launch_pp2<false, true, 1>(a, lds, st);
      case 2: return launch_pp2<false, true, 2>(a, lds, st);
      case 3: return launch_pp2<false, true, 3>(a, lds, st);
      case 4: return launch_pp2<false, true, 4>(a, lds, st);
      case 5: return launch_pp2<false, true, 5>(a, lds, st);
      case 6: return launch_pp2<false, true, 6>(a, lds, st);
      case 7: return launch_pp2<false, true, 7>(a, lds, st);
      case 8: return launch_pp2<false, true, 8>(a, lds, st);
      case 9: return launch_pp2<false, true, 9>(a, lds, st);
      case 10: return launch_pp2<false, true, 10>(a, lds, st);
      case 11: return launch_pp2<false, true, 11>(a, lds, st);   // EXPERIMENT: weight panels by LDS-DMA (see dma_A_unit)
      case 12: return launch_pp2<false, true, 12>(a, lds, st);   // tail without the residual load
      case 13: return launch_pp2<false, true, 13>(a, lds, st);   // tail without the stores
      case 14: return launch_pp2<false, true, 12, false, 32>(a, lds, st);   // ... the same two of the 16-byte tail
      case 15: return launch_pp2<false, true, 13, false, 32>(a, lds, st);
    }
  }
#endif
  if (f16) return mask ? launch_pp2<true, true>(a, lds, st) : launch_pp2<false, true>(a, lds, st);
  return mask ? launch_pp2<true, false>(a, lds, st) : launch_pp2<false, false>(a, lds, st);
}
#else
namespace {
template __global__ void conv_gemm_pp2_kernel<DV3_PP2_ISA_MASK, true, 0, false, DV3_PP2_ISA_ORD>(const ConvArgs);
}
#endif  // DV3_PP2_ISA_ONLY
