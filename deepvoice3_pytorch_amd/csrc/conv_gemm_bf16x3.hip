// Split-bf16 ("bf16x3") tap-GEMM: the same im2col-free dilated 1-D convolution + fused
// Conv1dGLU / HighwayConv1d tail as conv_gemm.hip (reference semantics:
// deepvoice3_pytorch/modules.py:145-164, 205-226), but the contraction runs on the bf16 matrix
// cores.  Every fp32 operand v is written as hi + lo, hi = bf16_rn(v), lo = bf16_rn(v - hi), and
//     acc += A_lo*B_hi + A_hi*B_lo + A_hi*B_hi        (fp32 accumulate, v_mfma_f32_32x32x16_bf16)
// Three bf16 MFMAs per product block cost 3/16 of the fp32-MFMA cycles for ~2^-18 relative
// operand error (include/dv3hip.h, "Split-bf16").
//
// Structure (DESIGN.md "conv_gemm_bf16x3"):
//   * GEMM view: M = output channels, N = B*T flattened (a column tile may span several batch
//     items, so short sequences -- T = 150 text, 200 decoder steps -- fill 128-column tiles), K =
//     (input-channel chunk of 32) x (tap).  One K step = one (chunk, tap) pair.
//   * weights arrive pre-split from dv3_split_pack_bf16 as [plane][j][k8][m][8]: the (tap, k8)
//     panel of the block's BM rows is one contiguous run of 16-byte units, staged with straight
//     coalesced 16-byte copies into the identical LDS image, double buffered per step.
//   * activations stay fp32 BCT in HBM.  Per chunk a block stages a haloed [32 ch][BN+(J-1)*dil]
//     tile over the FLAT (b,t) axis: each thread reads 8 channels of one column (8 row-coalesced
//     loads), applies the dropout keep-bit, splits into hi/lo and writes two 16-byte units: LDS
//     image [plane][k8][column][8 ch], double buffered per chunk; the J taps read it at shifted
//     columns (no im2col, one HBM read).  Where a shifted read leaves the column's own batch item
//     (the conv's zero padding) the fragment is zeroed in registers -- only waves whose 32..64
//     columns touch a sequence edge for that tap take that path (wave-uniform test).
//   * next step's weight panel is fetched into registers BEFORE the step's MFMAs and written to the
//     other LDS buffer after them; the next chunk's activation tile (HBM latency) is fetched J
//     steps ahead.  One barrier per step.
//   * an MFMA fragment (lane = one row/column, 8 consecutive k) is ONE ds_read_b128 whose 32
//     lanes of a half-wave cover 512 contiguous bytes: conflict free for both operands.
//   * wave tile 64(M: the `a` rows + their gate rows) x NI*32(N), same accumulator layout as the
//     fp32 kernel, so the epilogue (conv_common.h) is shared verbatim.
#include "conv_common.h"
#include <math.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BKC = 32;        // channels per K chunk
constexpr int KB = 4;          // k8 blocks per chunk
constexpr int HALO_MAX = 64;   // (J-1)*dil supported by the register staging (model max: 2*27)

// split 8 floats into hi / lo bf16x8
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
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

// one 32x32x16 MFMA on raw 16-byte operand units, bf16 or fp16
template <bool F16>
__device__ __forceinline__ f32x16 mma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// uniform base + zero-extended 32-bit byte offset: selects the SGPR-base global_load form (one
// VALU add per load instead of a 64-bit address build)
template <typename T>
__device__ __forceinline__ T ldg_off(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// ABL 10 (dv3_debug_set(1, 10), 128x256 ping-pong tile): per-wave phase timestamps of ONE workgroup,
// [wave][slot][0 = s_memrealtime (100 MHz), 1 = s_memtime]; read back with dv3_debug_read(1, ...)
constexpr int STAMP_SLOTS = 192;
__device__ unsigned long long g_x3_stamps[8 * STAMP_SLOTS * 2];

// DPJ (round 5, "deep prefetch"): 0 = the loops below; 1 / 3 = the tap count is this compile-time constant and the
// global fetches run DA steps (weight panels) / DX chunks (activation tiles) ahead through register rings (see the
// DPJ main loop).  For grids that leave a CU with one or two workgroups -- the decoder's T = 200 layers, every layer at
// the preset's own batch 16 -- a step of the in-phase loop costs one exposed L2 / HBM round trip (its panel is fetched
// at the top of the step that stores it); with the rings the round trip is paid once per tile.
//
// KS (round 5, "k-split"): 1 = one group of WM x WN waves; 2 = TWO such groups in one workgroup, each with its own LDS
// buffers, staging the first / second half of the input-channel chunks of the SAME output tile; the second group's
// accumulators go through LDS to the first, which adds them (first half + second half: a fixed order, a function of the
// shape) and runs the tail.  For grids that leave a CU one workgroup (profiles/r05_conv_census_dv3lj_b16.txt: every
// layer at the preset's batch 16) a lone wave per SIMD serialises fragment reads -> MFMAs -> stores -> barrier; the
// second group is the second wave per SIMD that overlaps them, on a k-range half as long.
// FG (round 6): an input-gradient launch whose tail also runs the gate backward of the layer that PRODUCED this layer's
// input (dv3_conv_desc.pg; conv_common.h) -- separate instantiations (bf16 pair, no dropout) that contain that tail only.
template <int WM, int WN, int NI, bool MASK, int ABL = 0, int TERMS = 3, int MI = 1, bool PP = false, bool F16 = false, int DPJ = 0, int KS = 1, bool FG = false>
__global__ __launch_bounds__(WM* WN * 64 * KS) void conv_gemm_bf16x3_kernel(const ConvArgs args) {
  static_assert(!FG || (!MASK && !F16 && TERMS == 3 && ABL == 0 && DPJ == 0), "fused gate backward: the unmasked bf16-pair forms");
  static_assert(!F16 || TERMS == 3, "the fp16 form is the three-term split");
  static_assert(KS == 1 || (KS == 2 && !PP && DPJ == 0 && ABL == 0), "k-split: two groups on the in-phase loop");
  static_assert(DPJ == 0 || (DPJ == 1 || DPJ == 3), "deep prefetch: 1 or 3 taps");
  static_assert(DPJ == 0 || (!PP && MI == 1 && ABL == 0 && TERMS == 3), "deep prefetch: the in-phase three-term tiles");
  // rings: weight panels in flight (steps) / activation tiles in flight (chunks).  One step of the 1-tap form issues
  // 4 + 8 loads (+ 8 keep-bit words when masked): the masked ring is one step shorter to stay under the 63 loads vmcnt counts
  constexpr int DA = DPJ == 0 ? 1 : (DPJ == 1 ? (MASK ? 3 : 4) : 3);
  constexpr int DX = DPJ == 0 ? 1 : (DPJ == 1 ? (MASK ? 3 : 4) : 2);
  static_assert(!PP || (WM * WN == 8 && MI == 1 && (ABL == 0 || ABL >= 10)), "ping-pong: 8 waves, one row sub-tile per wave");
  constexpr int BM = WM * MI * 64, BMH = WM * MI * 32, BN = WN * NI * 32;
  constexpr int NT = WM * WN * 64;
  constexpr int AU = KB * BM / NT;                          // A units per plane per thread per step
  constexpr int XI = (KB * (BN + (DPJ == 1 ? 0 : HALO_MAX)) + NT - 1) / NT;  // X items per thread per chunk
  static_assert(KB * BM % NT == 0, "A panel must split evenly");
  const dv3_conv_desc& p = args.d;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int J = DPJ ? DPJ : p.J, dil = p.dil;   // (the dispatcher launches a DPJ form only for that tap count)
  const int BNH = BN + (J - 1) * dil;
  const int xbuf = 2 * KB * BNH;  // units per X buffer (hi + lo)
  // k-split: group index, and thread / wave index INSIDE the group (all staging and tile indexing below is per group)
  const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int grp = KS > 1 ? wave_all / (WM * WN) : 0;
  const int grp_units = 2 * 2 * KB * BM + 2 * xbuf;   // 16-byte units of one group's buffers
  // [2 buffers] x { A hi [KB][BM], A lo [KB][BM] } then [2 buffers] x { X hi [KB][BNH], X lo }
  bf16x8* const As = reinterpret_cast<bf16x8*>(smem_raw) + grp * grp_units;
  bf16x8* const Xs = As + 2 * 2 * KB * BM;

  const int tid = (int)threadIdx.x - grp * NT;
  const int lane = tid & 63;
  const int wave = wave_all - grp * (WM * WN);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int pid = dv3_xcd_remap(blockIdx.x, args.n_blocks);
  const int mt = pid % args.m_tiles;
  const int nt = pid / args.m_tiles;
  const int n0 = nt * BN;

  const bool gated = (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY);
  int h0b, h1b, lim0, lim1;
  if (gated) {
    h0b = mt * BMH; h1b = p.a_half + mt * BMH; lim0 = p.a_half; lim1 = p.lda;
  } else {
    h0b = mt * BM; h1b = mt * BM + BMH; lim0 = lim1 = p.lda;
  }

  const int Cin = p.Cin, T = p.Tout, lda = p.lda, B = p.B;
  const int Ntot = B * T;
  const int k8_total = args.kp >> 3;
  const bf16x8* __restrict__ Wh = reinterpret_cast<const bf16x8*>(p.a_split);
  const int64_t plane = (int64_t)J * k8_total * lda;  // 16-byte units per plane
  const uint32_t* __restrict__ xmask = p.xmask;
  // fp16 form: the activation scale 2^DV3_F16_ACT_SHIFT rides on the dropout scale (or is applied alone)
  const float xscale = F16 ? (float)(1 << DV3_F16_ACT_SHIFT) : 1.0f;
  const float dscale = p.drop_scale * xscale;
  const bool xpw = !MASK && !F16 && TERMS == 3 && p.x_pair != 0;   // round 6: x holds pair words (include/dv3hip.h)

  // ---- this lane's output columns: (batch, time) and per-tap validity of the shifted read ----
  uint32_t vbits = 0;  // bit j*NI+ni: the tap-j input of column ni lies inside its batch item
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
    const int bc = n / T, tc = n - bc * T;
    for (int j = 0; j < J; ++j) {
      const int ts = tc + j * dil - p.padL;
      if (n < Ntot && ts >= 0 && ts < T) vbits |= 1u << (j * NI + ni);
    }
  }
  uint32_t need = 0;  // wave-uniform: taps for which some lane of this wave must zero its fragment
  for (int j = 0; j < J; ++j) {
    const uint32_t all = ((1u << NI) - 1u) << (j * NI);
    if (!__all((vbits & all) == all)) need |= 1u << j;
  }
  need = __builtin_amdgcn_readfirstlane(need);

  // ---- this thread's X staging items: flat column -> (batch, time), fixed over chunks ----
  // Nothing staged needs zeroing: columns outside the tensor or outside an output column's own
  // batch item are zeroed per fragment (vbits), channels >= Cin (clamped reads) meet zero weight
  // rows, and weight rows beyond the tile's valid range only feed output rows the epilogue drops.
  uint32_t xoff[XI];                // byte offset of (b, k8*8, t) from p.x  (< 2^32, host-checked)
  uint32_t xmo[MASK ? XI : 1];      // byte offset of word (b*Cin + k8*8, t>>5) in xmask
  int xsh[MASK ? XI : 1];           // bit position t & 31
  int xk8[XI];
  const int n_items = KB * BNH;
  // bf16 storage (single-term kernels): x may be a bf16 tensor -- half the bytes, widened exactly while staging
  const bool xbf = (TERMS == 1) && (p.io_bf16 & DV3_IO_IN_BF16);
  const uint32_t xsz = xbf ? 2u : 4u;
  const uint32_t x_rsb = (uint32_t)p.x_rs * xsz, m_rsb = (uint32_t)p.xmask_rs * 4u;
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
    xk8[i] = k8 < KB ? k8 * 8 : 0;
    xoff[i] = ((uint32_t)bf * (uint32_t)p.x_bs + (uint32_t)tf) * xsz + (uint32_t)xk8[i] * x_rsb;
    if (MASK) {
      xmo[i] = ((uint32_t)(bf * Cin) * (uint32_t)p.xmask_rs + (uint32_t)(tf >> 5)) * 4u + (uint32_t)xk8[i] * m_rsb;
      xsh[i] = tf & 31;
    }
  }
  // weight panel: per-unit column offset inside a (tap, k8) row of the split image
  uint32_t aoff[AU];   // bytes
#pragma unroll
  for (int u = 0; u < AU; ++u) {
    const int idx = tid + u * NT;  // k8 * BM + col
    const int col = idx % BM, k8 = idx / BM;
    const bool hi_half = col >= BMH;
    const int gcol = (hi_half ? h1b : h0b) + (col - (hi_half ? BMH : 0));
    aoff[u] = (uint32_t)(k8 * lda + (gcol < lda ? gcol : 0)) * 16u;
  }

  // ---- register staging ----
  bf16x8 ra[DA][2][AU];
  float rx[DX][XI][8];
  uint32_t rm[MASK ? DX : 1][MASK ? XI : 1][8];

  const int nchunks_all = (Cin + BKC - 1) / BKC;
  const int nch_first = KS > 1 ? (nchunks_all + 1) / 2 : nchunks_all;      // k-split: the first group's share
  const int c_base = grp * nch_first;                                      // this group's first chunk
  const int nchunks = grp == 0 ? nch_first : nchunks_all - nch_first;      // ... and its number of chunks (local indices below)

  auto load_A = [&](int chunk_l, int j, int slot = 0) {
    const int chunk = chunk_l + c_base;
    const bf16x8* srch = Wh + (int64_t)(j * k8_total + chunk * KB) * lda;  // uniform
    const bf16x8* srcl = srch + plane;
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      ra[slot][0][u] = ldg_off<bf16x8>(srch, aoff[u]);
      if (TERMS == 3) ra[slot][1][u] = ldg_off<bf16x8>(srcl, aoff[u]);
    }
  };
  auto write_A = [&](int buf, int slot = 0) {
    bf16x8* dst = As + buf * (2 * KB * BM);
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      dst[tid + u * NT] = ra[slot][0][u];
      if (TERMS == 3) dst[KB * BM + tid + u * NT] = ra[slot][1][u];
    }
  };
  auto load_X = [&](int chunk_l, int slot = 0) {
    const int c0 = (chunk_l + c_base) * BKC;
    if (c0 + BKC <= Cin) {
      // whole chunk in range (uniform): 8 uniform row bases + one loop-invariant per-thread offset
      // per item -> every load is the SGPR-base form with no address arithmetic
      const char* xb = reinterpret_cast<const char*>(p.x) + (int64_t)c0 * x_rsb;
      const char* mb = reinterpret_cast<const char*>(xmask) + (int64_t)c0 * m_rsb;
      if (TERMS == 1 && xbf) {      // bf16 tensor: one 2-byte load per element, widened exactly (uniform branch, hoisted)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int i = 0; i < XI; ++i) {
            rx[slot][i][e] = __uint_as_float((uint32_t)ldg_off<uint16_t>(xb + (int64_t)e * x_rsb, xoff[i]) << 16);
            if (MASK) rm[slot][i][e] = ldg_off<uint32_t>(mb + (int64_t)e * m_rsb, xmo[i]);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int i = 0; i < XI; ++i) {
            rx[slot][i][e] = ldg_off<float>(xb + (int64_t)e * x_rsb, xoff[i]);
            if (MASK) rm[slot][i][e] = ldg_off<uint32_t>(mb + (int64_t)e * m_rsb, xmo[i]);
          }
        }
      }
    } else {
      // last, partial chunk: channels >= Cin are clamped to the last real row (they meet zero
      // weight rows)
#pragma unroll
      for (int i = 0; i < XI; ++i) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // row relative to this item's k8 block (mod 2^32: may be "negative" when Cin < k8*8)
          const uint32_t dc = (uint32_t)(min(c0 + xk8[i] + e, Cin - 1) - xk8[i]);
          rx[slot][i][e] = xbf ? __uint_as_float((uint32_t)ldg_off<uint16_t>(p.x, xoff[i] + dc * x_rsb) << 16)
                         : ldg_off<float>(p.x, xoff[i] + dc * x_rsb);
          if (MASK) rm[slot][i][e] = ldg_off<uint32_t>(xmask, xmo[MASK ? i : 0] + dc * m_rsb);
        }
      }
    }
  };
  auto write_X = [&](int buf, int slot = 0) {
    bf16x8* dst = Xs + buf * xbuf;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int idx = tid + i * NT;
      if (idx < n_items) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = rx[slot][i][e];
          if (MASK) v[e] *= ((rm[slot][i][e] >> xsh[i]) & 1u) ? dscale : 0.f;
          else if (F16) v[e] *= xscale;
        }
        bf16x8 hi, lo;
        if constexpr (F16) dv3_note_range(args.range_ctr, dv3_split8_f16(v, hi, lo));
        else if constexpr (!MASK && TERMS == 3) {
          if (xpw) {      // pair words (dv3_conv_desc.x_pair, wave-uniform): the pair is already there
            uint32_t w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(rx[slot][i][e]);
            dv3_pair_units(w, hi, lo);
          } else split8(v, hi, lo);
        } else split8(v, hi, lo);
        dst[idx] = hi;
        if (TERMS == 3) dst[KB * BNH + idx] = lo;
      }
    }
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

  const int nsteps = (ABL == 5 || ABL == 9) ? 0 : nchunks * J;
  const int a_off = wm * (MI * 32) + l31;
  const int x_off = wn * (NI * 32) + l31;

  if constexpr (DPJ == 0) {
    load_A(0, 0);
    load_X(0);
    write_A(0);
    write_X(0);
    __syncthreads();
  }

  int n_stamp = 0;
  auto stamp = [&]() {
    if constexpr (ABL == 10) {
      if (blockIdx.x == gridDim.x / 2 && n_stamp < STAMP_SLOTS) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), t1 = __builtin_readcyclecounter();
        if (lane == 0) {
          g_x3_stamps[(wave * STAMP_SLOTS + n_stamp) * 2] = t0;
          g_x3_stamps[(wave * STAMP_SLOTS + n_stamp) * 2 + 1] = t1;
        }
      }
      ++n_stamp;
    }
  };
  int c = 0, j = 0;
  if constexpr (PP) {
    // ---- ping-pong main loop (8-wave tiles, one workgroup per CU) ----
    // Waves w and w+4 of a workgroup share a SIMD (MI355X_MICROARCH.md, "Two waves per SIMD"): its
    // matrix pipe serves one MFMA stream at a time, so the two run the SAME step sequence half a
    // step apart -- while one issues its 24 back-to-back MFMAs from registers (COMPUTE) the other
    // fetches its 16 fragments of the next step from LDS, zeroes sequence-edge columns and stores
    // its share of the following tiles (LOAD).  One barrier per phase; the late half enters the
    // loop one barrier later and skips the last one, so both execute 2*nsteps barriers.
    //   interval:   I0      I1      I2      I3
    //   waves 0-3:  L(0)    C(0)    L(1)    C(1) ...
    //   waves 4-7:  -       L(0)    C(0)    L(1) ...
    // LDS hazards: the tile of step s+1 is stored during L(s) by each half (intervals 2s, 2s+1)
    // into the buffer last read for step s-1 (intervals 2s-2, 2s-1) and first read in interval
    // 2s+2; the activation tile of chunk c+1 likewise during the L of chunk c's last tap.  Global
    // fetches run a full step ahead of their store: right after L(s) has stored the panel of step
    // s+1 it fetches that of step s+2 into the same registers, and the activation tile of chunk
    // c+2 right after chunk c+1's was stored.
    const int late = wave >> 2;
    load_A((1 < nsteps && J == 1) ? 1 : 0, (1 < nsteps && J > 1) ? 1 : 0);
    if (1 < nchunks) load_X(1);
    stamp();                       // slot 0: prologue done
    // wave priority scheme (measurement knob, dv3_debug_set(14, v)): 1 = LOAD phases at priority 3, 2 = COMPUTE
    // phases at priority 3, 3 / 4 = the late / early half at static priority 1
    const int prio = args.prio;
    if (prio == 3 && late) __builtin_amdgcn_s_setprio(1);
    if (prio == 4 && !late) __builtin_amdgcn_s_setprio(1);
    if (late) __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
      const int cur = step & 1;
      int jn = j + 1, cn = c;
      if (jn == J) { jn = 0; cn = c + 1; }
      const bool has_next = step + 1 < nsteps;
      const bool new_chunk = has_next && jn == 0;
      stamp();                     // slot 1 + 6*step: LOAD begins
      if (prio == 1) __builtin_amdgcn_s_setprio(3);
      // ---------------- LOAD ----------------
      bf16x8 ah[2][2], al[2][2], bh[2][NI], bl[2][NI];
      {
        const bf16x8* AsH = As + cur * (2 * KB * BM);
        const bf16x8* AsL = AsH + KB * BM;
        const bf16x8* XsH = Xs + (c & 1) * xbuf;
        const bf16x8* XsL = XsH + KB * BNH;
        const bool fix = (need >> j) & 1u;
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int k8 = 2 * s + lhi;
          const int ai = k8 * BM + a_off;
          ah[s][0] = AsH[ai];
          ah[s][1] = AsH[ai + BMH];
          al[s][0] = (TERMS == 3) ? AsL[ai] : ah[s][0];
          al[s][1] = (TERMS == 3) ? AsL[ai + BMH] : ah[s][1];
          const int xi = k8 * BNH + x_off + j * dil;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            bh[s][ni] = XsH[xi + ni * 32];
            bl[s][ni] = (TERMS == 3) ? XsL[xi + ni * 32] : bh[s][ni];
          }
        }
        if (fix) {
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const bool ok = (vbits >> (j * NI + ni)) & 1u;
              bh[s][ni] = ok ? bh[s][ni] : zero8;
              if (TERMS == 3) bl[s][ni] = ok ? bl[s][ni] : zero8;
            }
        }
      }
      // stores of the next tiles, then the fetches that refill their registers: every vmcnt wait
      // of this phase precedes its fetches, so it only covers loads issued a full step ago, and
      // the COMPUTE phase is MFMAs only.  The panel store / fetch pair is unconditional (the last
      // steps re-fetch the current panel and store into the buffer nobody reads any more): a
      // conditional pair makes the compiler guard the fetch's registers with a vmcnt(0) that
      // would then sit behind the activation fetch issued just above.
      stamp();                     // fragments in registers
      write_A(cur ^ 1);
      stamp();                     // panel stored
      if (new_chunk) {
        write_X((c + 1) & 1);
        if (cn + 1 < nchunks) load_X(cn + 1);
      }
      {
        int j2 = jn + 1, c2 = cn;
        if (j2 == J) { j2 = 0; c2 = cn + 1; }
        const bool more = step + 2 < nsteps;
        load_A(more ? c2 : c, more ? j2 : j);
      }
      stamp();                     // LOAD issued (the SMEM read waits for the LDS queue)
      __syncthreads();
      stamp();                     // COMPUTE begins
      if (prio == 1) __builtin_amdgcn_s_setprio(0);
      if (prio == 2) __builtin_amdgcn_s_setprio(3);
      // ---------------- COMPUTE ----------------
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (TERMS == 3) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[0][0][ni] = mma16<F16>(al[s][0], bh[s][ni], acc[0][0][ni]);
            acc[0][1][ni] = mma16<F16>(al[s][1], bh[s][ni], acc[0][1][ni]);
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[0][0][ni] = mma16<F16>(ah[s][0], bl[s][ni], acc[0][0][ni]);
            acc[0][1][ni] = mma16<F16>(ah[s][1], bl[s][ni], acc[0][1][ni]);
          }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[0][0][ni] = mma16<F16>(ah[s][0], bh[s][ni], acc[0][0][ni]);
          acc[0][1][ni] = mma16<F16>(ah[s][1], bh[s][ni], acc[0][1][ni]);
        }
      }
      stamp();                     // MFMAs issued
      if (prio == 2) __builtin_amdgcn_s_setprio(0);
      if (has_next || !late) __syncthreads();
      j = jn;
      c = cn;
    }
    stamp();                       // slot 1 + 6*nsteps: main loop left
  } else if constexpr (DPJ > 0) {
    // ---- deep-prefetch main loop (in-phase tiles, compile-time tap count) ----
    // Step s = (chunk s / J, tap s % J).  At its top the panel of step s + DA goes into ring slot s % DA (the panel that
    // slot held was stored at the end of step s - 1) and, at a chunk's first tap, the activation tile of chunk c + DX into
    // slot c % DX; at its end the panel of step s + 1 (fetched DA - 1 steps ago) and, at a chunk's last tap, the tile of
    // chunk c + 1 (fetched (DX - 1) chunks ago) are stored into the other LDS buffer.  U steps are unrolled so that every
    // ring index is a compile-time constant (U a multiple of DA and of DX * J; U and U / J even: the LDS buffer parities too).
    // Same fragment images, same MFMA order as the loop below: bit-identical results.
    //
    // The loop starts U steps BEFORE step 0: the virtual steps run the same fetches and stores (indices clamped into
    // the tile, so they re-fetch step 0's panel at worst) and stores, and skip only the fragment reads + MFMAs; so do the
    // up to U - 1 steps behind the last one (the loop leaves at a round boundary only).  That fills the rings in consumption order with the loop's
    // own instruction stream -- no separate prologue whose differently-ordered pending fetches the compiler would have
    // to merge into the loop header's wait counts (a first version with a peeled prologue drained the rings to 11
    // outstanding loads at the top of every U-th step) -- and every fetch and store is unconditional, so each
    // s_waitcnt vmcnt(N) is exact.
    constexpr int U = (DPJ == 1 && DA == 4) ? 4 : 6;
    static_assert(U % DA == 0 && (U / DPJ) % DX == 0 && U % DPJ == 0 && U % 2 == 0 && (U / DPJ) % 2 == 0,
                  "ring slots and LDS buffer parities must be compile-time");
    const int last_step = nsteps - 1, last_chunk = nchunks - 1;
#pragma unroll
    for (int a = 0; a < DA; ++a)
#pragma unroll
      for (int u = 0; u < AU; ++u) ra[a][0][u] = ra[a][1][u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int x = 0; x < DX; ++x)
#pragma unroll
      for (int i = 0; i < XI; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          rx[x][i][e] = 0.f;
          if (MASK) rm[x][i][e] = 0u;
        }
    int s_first = -U;
    asm volatile("" : "+s"(s_first));   // opaque: keeps the compiler from peeling the virtual round into a prologue again
    for (int s0 = s_first; s0 < nsteps; s0 += U) {
#pragma unroll
      for (int dd = 0; dd < U; ++dd) {
        const int s = s0 + dd;
        const int jj = dd % DPJ;                 // tap
        const int cx = dd / DPJ;                 // chunk, relative to s0 / J (a multiple of DX)
        {
          const int sa = min(max(s + DA, 0), last_step);
          load_A(sa / DPJ, sa % DPJ, dd % DA);
        }
        if (jj == 0) load_X(min(max(s0 / DPJ + cx + DX, 0), last_chunk), cx % DX);
        if (s >= 0 && s < nsteps) {   // (no early exit from the round: the compiler funnels every loop exit through one
                                      //  block with an edge back to the header, whose wait counts then assume the worst exit)
          const bf16x8* AsH = As + (dd & 1) * (2 * KB * BM);
          const bf16x8* AsL = AsH + KB * BM;
          const bf16x8* XsH = Xs + (cx & 1) * xbuf;
          const bf16x8* XsL = XsH + KB * BNH;
          const bool fix = (need >> jj) & 1u;
          const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int k8 = 2 * s2 + lhi;
            const int ai = k8 * BM + a_off;
            const bf16x8 ah0 = AsH[ai], ah1 = AsH[ai + BMH], al0 = AsL[ai], al1 = AsL[ai + BMH];
            bf16x8 bh[NI], bl[NI];
            const int xi = k8 * BNH + x_off + jj * dil;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              bh[ni] = XsH[xi + ni * 32];
              bl[ni] = XsL[xi + ni * 32];
            }
            if (fix) {
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) {
                const bool ok = (vbits >> (jj * NI + ni)) & 1u;
                bh[ni] = ok ? bh[ni] : zero8;
                bl[ni] = ok ? bl[ni] : zero8;
              }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[0][0][ni] = mma16<F16>(al0, bh[ni], acc[0][0][ni]);
              acc[0][1][ni] = mma16<F16>(al1, bh[ni], acc[0][1][ni]);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[0][0][ni] = mma16<F16>(ah0, bl[ni], acc[0][0][ni]);
              acc[0][1][ni] = mma16<F16>(ah1, bl[ni], acc[0][1][ni]);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[0][0][ni] = mma16<F16>(ah0, bh[ni], acc[0][0][ni]);
              acc[0][1][ni] = mma16<F16>(ah1, bh[ni], acc[0][1][ni]);
            }
          }
        }
        // stores for step s + 1, unconditional like the fetches (a conditional store leaves "maybe still pending" marks
        // on its ring slot at the join, which reach the loop header).  The last step stores into buffers nobody reads any
        // more; the virtual steps before -1 store ring slots that are not fetched yet: zeros (see the initialisation
        // above the loop -- the fp16 range guard must not see register garbage), into buffers overwritten before use.
        write_A((dd & 1) ^ 1, (dd + 1) % DA);
        if (jj == DPJ - 1) write_X((cx & 1) ^ 1, (cx + 1) % DX);
        __syncthreads();
      }
    }
  } else
  for (int step = 0; step < (KS > 1 ? nch_first * J : nsteps); ++step) {
    if (KS > 1 && step >= nsteps) {   // the second group of an odd chunk count: one chunk fewer, same number of barriers
      __syncthreads();
      continue;
    }
    const int cur = step & 1;
    int jn = j + 1, cn = c;
    if (jn == J) { jn = 0; cn = c + 1; }
    const bool has_next = step + 1 < nsteps;
    const bool new_chunk = has_next && jn == 0;
    if (ABL != 1 && ABL != 2 && ABL != 3) {
      if (has_next) load_A(cn, jn);
      // the activation tile streams from HBM (a weight panel is an L2 hit): fetch chunk c+1 at the
      // FIRST tap of chunk c, J steps ahead of the write that needs it
      if (j == 0 && c + 1 < nchunks) load_X(c + 1);
    }

    // ---------------- MFMA: two k16 steps of tap j ----------------
    {
      const bf16x8* AsH = As + cur * (2 * KB * BM);
      const bf16x8* AsL = AsH + KB * BM;
      const bf16x8* XsH = Xs + (c & 1) * xbuf;
      const bf16x8* XsL = XsH + KB * BNH;
      const bool fix = (need >> j) & 1u;
      const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int k8 = 2 * s + lhi;
        const int ai = k8 * BM + a_off;
        bf16x8 ah[MI][2], al[MI][2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          ah[mi][0] = AsH[ai + mi * 32];
          ah[mi][1] = AsH[ai + mi * 32 + BMH];
          al[mi][0] = ah[mi][0];
          al[mi][1] = ah[mi][1];
          if (TERMS == 3) { al[mi][0] = AsL[ai + mi * 32]; al[mi][1] = AsL[ai + mi * 32 + BMH]; }
        }
        bf16x8 bh[NI], bl[NI];
        const int xi = k8 * BNH + x_off + j * dil;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bh[ni] = XsH[xi + ni * 32];
          bl[ni] = (TERMS == 3) ? XsL[xi + ni * 32] : bh[ni];
        }
        if (fix) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const bool ok = (vbits >> (j * NI + ni)) & 1u;
            bh[ni] = ok ? bh[ni] : zero8;
            if (TERMS == 3) bl[ni] = ok ? bl[ni] : zero8;
          }
        }
        if (ABL == 4) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(ah[mi][0]), "v"(ah[mi][1]), "v"(al[mi][0]), "v"(al[mi][1]));
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(bh[ni]), "v"(bl[ni]));
          continue;
        }
        // small terms first; each accumulator is touched once per pass (no back-to-back RAW)
        if (TERMS == 3) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[mi][0][ni] = mma16<F16>(al[mi][0], bh[ni], acc[mi][0][ni]);
              acc[mi][1][ni] = mma16<F16>(al[mi][1], bh[ni], acc[mi][1][ni]);
            }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[mi][0][ni] = mma16<F16>(ah[mi][0], bl[ni], acc[mi][0][ni]);
              acc[mi][1][ni] = mma16<F16>(ah[mi][1], bl[ni], acc[mi][1][ni]);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][0][ni] = mma16<F16>(ah[mi][0], bh[ni], acc[mi][0][ni]);
            acc[mi][1][ni] = mma16<F16>(ah[mi][1], bh[ni], acc[mi][1][ni]);
          }
      }
    }

    if (ABL != 2 && ABL != 3) {
      if (has_next) write_A(cur ^ 1);
      if (new_chunk) write_X((c + 1) & 1);
    }
    if (ABL != 3) __syncthreads();
    j = jn;
    c = cn;
  }

  if constexpr (KS > 1) {
    // second group -> LDS (its own buffers: the main loop's last barrier is behind every read of them) -> first group
    float* red = reinterpret_cast<float*>(reinterpret_cast<bf16x8*>(smem_raw) + grp_units);
    constexpr int NR = MI * 2 * NI * 16;
    static_assert((size_t)WM * WN * NR * 64 * 4 <= (size_t)(2 * 2 * KB * BM) * 16, "the accumulator image must fit one group's panels");
    float* mine = red + (wave * NR) * 64 + lane;
    if (grp == 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(((mi * 2 + h) * NI + ni) * 16 + r) * 64] = acc[mi][h][ni][r];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][h][ni][r] += mine[(((mi * 2 + h) * NI + ni) * 16 + r) * 64];
  }
  // ABL 6: skip the epilogue but keep the accumulators live
  if ((ABL != 6 && ABL != 9) || acc[0][0][0][0] + acc[MI - 1][1][0][0] + acc[0][0][NI - 1][5] + acc[MI - 1][1][NI - 1][7] == 1.2345e30f) {
    static_assert(MI == 1 || MI == 2, "row sub-tiles per wave");
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
    // (batch, time) of this lane's output columns, recomputed from an opaque copy of the tile origin
    // instead of being carried through the main loop in registers
    int n0e = n0;
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
      bool wide_done = false;
      if constexpr (NI == 2) {
        if (dv3_wide_gate_ok(p, args.wide) && (size_t)(WM * WN) * DV3_WIDE_LDS <= (size_t)(2 * 2 * KB * BM + 2 * xbuf) * 16) {
          float* wl = reinterpret_cast<float*>(smem_raw) + wave * (DV3_WIDE_LDS / 4);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int row0 = wm * (MI * 32) + mi * 32;
            conv_epilogue_wide_block_gate<BM, BMH>(p, acc[mi][0], mt, row0, 0, lane, nw0, Ntot, wl);
            conv_epilogue_wide_block_gate<BM, BMH>(p, acc[mi][1], mt, row0, 1, lane, nw0, Ntot, wl);
          }
          wide_done = true;
        }
      }
      if (!wide_done) {
        conv_epilogue_dgrad_gate<BM, BMH, NI>(p, acc[0], mt, wm * (MI * 32), lhi, l31, bcol, tcol, okc, nw0 >> 5);
        if (MI == 2) conv_epilogue_dgrad_gate<BM, BMH, NI>(p, acc[MI - 1], mt, wm * (MI * 32) + 32, lhi, l31, bcol, tcol, okc, nw0 >> 5);
      }
    } else
    if (TERMS == 1 && (p.io_bf16 & DV3_IO_OUT_C8)) {   // bf16 storage: channel-blocked y / ab / residuals
      conv_epilogue_c8<BM, BMH, NI>(p, acc[0], gated, mt, wm * (MI * 32), lhi, bcol, tcol, okc);
      if (MI == 2) conv_epilogue_c8<BM, BMH, NI>(p, acc[MI - 1], gated, mt, wm * (MI * 32) + 32, lhi, bcol, tcol, okc);
    } else if (NI == 2 && TERMS == 3 && ABL == 0 && dv3_wide_epilogue_ok(p, args.wide) &&
               (size_t)(WM * WN) * DV3_WIDE_LDS <= (size_t)(2 * 2 * KB * BM + 2 * xbuf) * 16) {
      // 16-byte epilogue through LDS (conv_common.h): the main loop's last barrier is behind every LDS read
      if constexpr (NI == 2) {
        float* wl = reinterpret_cast<float*>(smem_raw) + wave * (DV3_WIDE_LDS / 4);
        const int nw0 = n0e + wn * (NI * 32);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int row0 = wm * (MI * 32) + mi * 32;
          conv_epilogue_wide_block<BM, BMH>(p, acc[mi][0], mt, row0, 0, lane, nw0, Ntot, wl);
          conv_epilogue_wide_block<BM, BMH>(p, acc[mi][1], mt, row0, 1, lane, nw0, Ntot, wl);
        }
      }
    } else {
      conv_epilogue<BM, BMH, NI, ABL, TERMS == 1>(p, acc[0], gated, mt, wm * (MI * 32), lhi, bcol, tcol, okc);
      if (MI == 2)
        conv_epilogue<BM, BMH, NI, ABL, TERMS == 1>(p, acc[MI - 1], gated, mt, wm * (MI * 32) + 32, lhi, bcol, tcol, okc);
    }
  }
  stamp();                         // last slot: epilogue stores issued
}

// packed fp32 [J][K][lda] -> split image [plane][j][k8][m][8]
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ src,
                                                         bf16x8* __restrict__ dst, int J, int K,
                                                         int lda, int k8_total, int dtype, uint32_t* range_ctr) {
  const int64_t n = (int64_t)J * k8_total * lda;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int m = (int)(idx % lda);
  const int64_t row = idx / lda;  // j*k8_total + k8
  const int j = (int)(row / k8_total), k8 = (int)(row % k8_total);
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k8 * 8 + i;
    v[i] = (k < K) ? src[((int64_t)j * K + k) * lda + m] : 0.f;
    if (dtype == DV3_SPLIT_DTYPE_F16) v[i] *= (float)(1 << DV3_F16_WEIGHT_SHIFT);
  }
  bf16x8 hi, lo;
  if (dtype == DV3_SPLIT_DTYPE_F16) dv3_note_range(range_ctr, dv3_split8_f16(v, hi, lo)); else split8(v, hi, lo);
  dst[idx] = hi;
  dst[n + idx] = lo;
}

#ifdef DV3_X3_ISA_ONLY   // scripts/x3_isa.sh: one instantiation only, for reading its ISA
#ifndef DV3_X3_ISA_KS
#define DV3_X3_ISA_KS 1
#endif
template __global__ void conv_gemm_bf16x3_kernel<2, 2, 1, DV3_X3_ISA_MASK, 0, 3, 1, false, DV3_X3_ISA_F16, DV3_X3_ISA_DPJ, DV3_X3_ISA_KS>(const ConvArgs);
}  // namespace
#else
template <int WM, int WN, int NI, bool MASK, int TERMS, int MI = 1, bool PP = false, bool F16 = false, int DPJ = 0, int KS = 1, bool FG = false>
int launch_x3_m(const ConvArgs& a, size_t lds, hipStream_t st) {
  static bool attr_set = false;  // raise the dynamic-LDS cap once per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_bf16x3_kernel<WM, WN, NI, MASK, 0, TERMS, MI, PP, F16, DPJ, KS, FG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      dv3_set_error("conv_gemm_bf16x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  dim3 grid(a.n_blocks), block(WM * WN * 64 * KS);
  hipLaunchKernelGGL((conv_gemm_bf16x3_kernel<WM, WN, NI, MASK, 0, TERMS, MI, PP, F16, DPJ, KS, FG>), grid, block, lds * KS, st, a);
  return dv3_check_launch("conv_gemm_bf16x3");
}
int g_x3_ablate = 0;   // debug: dv3_debug_set(); ablation variants of the 128x128 unmasked tile
#ifdef DV3_EXPERIMENTS   // timing-only ablations / phase stamps: `make EXP=1` (not in the shipped library)
template <int ABL>
int launch_x3_abl(const ConvArgs& a, size_t lds, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)conv_gemm_bf16x3_kernel<2, 2, 2, false, ABL>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((conv_gemm_bf16x3_kernel<2, 2, 2, false, ABL>), dim3(a.n_blocks), dim3(256), lds, st, a);
  return dv3_check_launch("conv_gemm_bf16x3(abl)");
}
#endif
// main loop of the 8-wave tiles, dv3_debug_set(3, v): 0 = in-phase, 1 = ping-pong (default)
int g_x3_pingpong = 1;
template <int WM, int WN, int NI, int MI, bool PP>
int launch_x3_big_pp(const ConvArgs& a, size_t lds, hipStream_t st) {
  if (a.d.split_terms == DV3_SPLIT_F16X3)
    return a.d.xmask ? launch_x3_m<WM, WN, NI, true, 3, MI, PP, true>(a, lds, st) : launch_x3_m<WM, WN, NI, false, 3, MI, PP, true>(a, lds, st);
  if (a.d.split_terms == 1)
    return a.d.xmask ? launch_x3_m<WM, WN, NI, true, 1, MI, PP>(a, lds, st) : launch_x3_m<WM, WN, NI, false, 1, MI, PP>(a, lds, st);
  return a.d.xmask ? launch_x3_m<WM, WN, NI, true, 3, MI, PP>(a, lds, st) : launch_x3_m<WM, WN, NI, false, 3, MI, PP>(a, lds, st);
}
template <int WM, int WN, int NI, int MI>
int launch_x3_big(const ConvArgs& a, size_t lds, hipStream_t st) {
#ifdef DV3_EXPERIMENTS
  if constexpr (WM == 2 && WN == 4 && MI == 1) {
    if (g_x3_ablate == 10 && g_x3_pingpong && !a.d.xmask && (a.d.split_terms == 0 || a.d.split_terms == 3)) {
      (void)hipFuncSetAttribute((const void*)conv_gemm_bf16x3_kernel<2, 4, 2, false, 10, 3, 1, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((conv_gemm_bf16x3_kernel<2, 4, 2, false, 10, 3, 1, true>), dim3(a.n_blocks), dim3(512), lds, st, a);
      return dv3_check_launch("conv_gemm_bf16x3(stamps)");
    }
  }
#endif
  if constexpr (WM * WN == 8 && MI == 1) {
    if (g_x3_pingpong) return launch_x3_big_pp<WM, WN, NI, MI, true>(a, lds, st);
  }
  return launch_x3_big_pp<WM, WN, NI, MI, false>(a, lds, st);
}
template <int WM, int WN, int NI>
int launch_x3(const ConvArgs& a, size_t lds, hipStream_t st) {
#ifdef DV3_EXPERIMENTS
  if (g_x3_ablate && WM == 2 && WN == 2 && NI == 2 && !a.d.xmask && (a.d.split_terms == 0 || a.d.split_terms == 3)) {
    switch (g_x3_ablate) {
      case 1: return launch_x3_abl<1>(a, lds, st);
      case 2: return launch_x3_abl<2>(a, lds, st);
      case 3: return launch_x3_abl<3>(a, lds, st);
      case 4: return launch_x3_abl<4>(a, lds, st);
      case 5: return launch_x3_abl<5>(a, lds, st);
      case 6: return launch_x3_abl<6>(a, lds, st);
      case 7: return launch_x3_abl<7>(a, lds, st);
      case 8: return launch_x3_abl<8>(a, lds, st);
      case 9: return launch_x3_abl<9>(a, lds, st);
    }
  }
#endif
  if constexpr (WM == 2 && WN == 2 && NI == 1) {
    // k-split form of the 128 x 64 tile (kernel template KS = 2): chosen by the dispatcher (a.ks)
    if (a.ks == 2 && a.d.split_terms != 1) {
      const bool f16 = a.d.split_terms == DV3_SPLIT_F16X3, m = a.d.xmask != nullptr;
      if (f16) return m ? launch_x3_m<2, 2, 1, true, 3, 1, false, true, 0, 2>(a, lds, st) : launch_x3_m<2, 2, 1, false, 3, 1, false, true, 0, 2>(a, lds, st);
      return m ? launch_x3_m<2, 2, 1, true, 3, 1, false, false, 0, 2>(a, lds, st) : launch_x3_m<2, 2, 1, false, 3, 1, false, false, 0, 2>(a, lds, st);
    }
  }
#ifdef DV3_EXPERIMENTS   // measured and retired (profiles/r05_deep_prefetch_rings.txt): experiment build only
  if constexpr (WM == 2 && WN == 2 && NI == 1) {
    // deep-prefetch form of the 128 x 64 tile (kernel template DPJ): chosen by the dispatcher (a.dp = 1 or 3 taps)
    if (a.dp == 1 || a.dp == 3) {
      const bool f16 = a.d.split_terms == DV3_SPLIT_F16X3, m = a.d.xmask != nullptr;
      if (a.dp == 1) {
        if (f16) return m ? launch_x3_m<2, 2, 1, true, 3, 1, false, true, 1>(a, lds, st) : launch_x3_m<2, 2, 1, false, 3, 1, false, true, 1>(a, lds, st);
        return m ? launch_x3_m<2, 2, 1, true, 3, 1, false, false, 1>(a, lds, st) : launch_x3_m<2, 2, 1, false, 3, 1, false, false, 1>(a, lds, st);
      }
      if (f16) return m ? launch_x3_m<2, 2, 1, true, 3, 1, false, true, 3>(a, lds, st) : launch_x3_m<2, 2, 1, false, 3, 1, false, true, 3>(a, lds, st);
      return m ? launch_x3_m<2, 2, 1, true, 3, 1, false, false, 3>(a, lds, st) : launch_x3_m<2, 2, 1, false, 3, 1, false, false, 3>(a, lds, st);
    }
  }
#endif
  if (a.d.split_terms == DV3_SPLIT_F16X3)
    return a.d.xmask ? launch_x3_m<WM, WN, NI, true, 3, 1, false, true>(a, lds, st) : launch_x3_m<WM, WN, NI, false, 3, 1, false, true>(a, lds, st);
  if (a.d.split_terms == 1)
    return a.d.xmask ? launch_x3_m<WM, WN, NI, true, 1>(a, lds, st) : launch_x3_m<WM, WN, NI, false, 1>(a, lds, st);
  return a.d.xmask ? launch_x3_m<WM, WN, NI, true, 3>(a, lds, st) : launch_x3_m<WM, WN, NI, false, 3>(a, lds, st);
}

int g_x3_ks = 1;        // dv3_debug_set(44, v): k-split form of the 128 x 64 tile: 0 never, 1 by the rule in the dispatcher, 2 wherever eligible
int g_x3_ks_max_blocks = 256, g_x3_ks_min_steps = 8;   // dv3_debug_set(45 / 46, v): the rule's bounds
int g_x3_dp = 0;        // dv3_debug_set(43, v), experiment build: deep-prefetch form of the 128 x 64 tile: 0 never, 1 small grids, 2 always
int g_x3_rel8 = 86;     // dv3_debug_set(42, v): relative cost (percent) of the 256 x 128 ping-pong tile in the picker below.  Rounds 2-4: 93
                        // (north-star sweep).  Round 5's census of a real step (profiles/r05_conv_census_dv3lj_b64.txt) has it ahead
                        // of the 128 x 256 tile stand-alone wherever the two tie on rounds (the encoder's input gradients, the
                        // 1 x 1 layers at T = 804); whole steps at 86: 15.56 -> 15.45 ms at B = 64, 7.75 -> 7.71 at B = 16
                        // (profiles/r05_rel8_step_ab.txt)
int g_x3_j1_flat = 1;   // dv3_debug_set(27, v)
int g_x3_rel2 = 112;   // dv3_debug_set(9, v): relative cost (percent) of the 128x64 tile in the picker below
// bf16x3 tile choice: padded work over the FLAT column axis, weight-panel traffic penalised
// (a block re-reads its A panel every K step, so narrow column tiles starve the matrix pipe).
const TileCfg* pick_tile_x3(const dv3_conv_desc* d, bool gated, int want_tile) {
  const TileCfg* best = nullptr;
  double best_cost = 0;
  const int64_t ntot = (int64_t)d->B * d->Tout;
  for (const TileCfg& c : kCfgs) {
    if (want_tile && c.id != want_tile) continue;
    if (d->pg && !(c.id == 1 || c.id == 2 || c.id == 8 || c.id == 9)) continue;   // the tiles with a fused-gate-backward tail
    const int BM = c.wm * c.mi * 64, BMH = c.wm * c.mi * 32, BN = c.wn * c.ni * 32;
    const int64_t mt = gated ? dv3_cdiv(d->Cg, BMH) : dv3_cdiv(d->M, BM);
    const int64_t ntl = dv3_cdiv64(ntot, BN);
    const double slots = c.id > 6 ? 256.0 : 512.0;   // the 256-wide tiles: one workgroup per CU
    // time ~ (rounds of the chip's 2 x 256 resident workgroups) x (tile work) x (operand traffic
    // per MFMA ~ 1/BN for the weight panel + 1/BM for the activation tile).  A last round with at
    // most one workgroup per CU runs those workgroups unshared, i.e. faster.
    const double blocks = (double)mt * ntl;
    const double full = floor(blocks / slots), rem = blocks - slots * full;
    const double rounds = (full + (rem == 0 ? 0.0 : (c.id <= 6 && rem <= 256 ? 0.6 : 1.0))) * (slots / 512.0);
    // measured time per unit of tile work relative to the 128x128 tile (north-star shape, d = 1..27):
    // the 2-wave 64-row tiles and the 32-column tiles stage far more per MFMA
    // 8-wave 256-wide tiles (one workgroup per CU): less weight-panel traffic per MFMA; the
    // 128-row-per-wave tile (7, one wave per SIMD) measured slower everywhere: opt-in only (hint 27)
    static const double kRel[10] = {0, 1.0, 1.12, 2.0, 2.2, 1.8, 2.0, 9.9, 0.93, 0.87};
    double rel = c.id == 2 ? g_x3_rel2 * 0.01 : c.id == 8 ? g_x3_rel8 * 0.01 : kRel[c.id];
    // 1 x 1 convolutions / Linear layers with K <= 512 (8-16 k32 steps per tile): prologue and tail dominate and the wide
    // tiles' staging advantage is gone -- measured per unit of modeled work the 128 x 64, 128 x 128, 256 x 128 and
    // 128 x 256 tiles cost the same (scripts/small_gemm_tiles.py: with the J = 3 weights the picker took the 128 x 256
    // tile for 64 x 804 columns x 512 rows at 135 us where the 128 x 64 tile runs 107), so only the rounds decide
    if (g_x3_j1_flat && d->J == 1 && d->Cin <= 512 && (c.id == 1 || c.id == 2 || c.id == 8 || c.id == 9)) rel = 1.0;
    const double cost = rounds * BM * BN * rel;
    if (!best || cost < best_cost) {
      best = &c;
      best_cost = cost;
    }
  }
  return best;
}

}  // namespace

int g_x3_pp2 = 128;  // dv3_debug_set(12, v): the 256 x 256 k16 ping-pong kernel (conv_gemm_pp2.hip) serves eligible shapes
                     // whose grid has at least v tiles (0 = never).  Measured at B=64 over the presets' conv shapes
                     // (scripts/pp2_sweep.py, profiles/r03_pp2_sweep.txt): 0.73-0.88 of the 128 x 256 / 128 x 64 kernels'
                     // time from 152 tiles up, 1.3-1.9 x at 50-100 tiles (half the chip idle).
extern int g_pp2_ord, g_pp2_ord_u, g_pp2_ord_m, g_pp2_abl, g_pp2_sk, g_pp2_sk_overhead, g_pp2_sk_gain, g_pp2_sk_abl, g_pp2_fast_tail;
int g_x3_pp2_sk_units = 8;   // measured (scripts/pp2_sk_check.py): at 9.5 units per CU (the encoder layers: 152 / 76 tiles) the
                             // stream-K form beats the 128-wide kernels by 6-8 %, at 6.3 (101 tiles x 16) it loses to them
int dv3_conv_gemm_pp2_dispatch(const dv3_conv_desc* d, hipStream_t st);   // conv_gemm_pp2.hip
int g_x3_wide = 1;  // dv3_debug_set(18, v): 16-byte epilogue through LDS (conv_common.h): 0 off, 1 DGRAD, 
int g_x3_prio = 0; // dv3_debug_set(14, v): wave priority scheme of the ping-pong main loop

// called by dv3_conv_gemm_f32 (conv_gemm.hip) when d->a_split != NULL; returns 1 when the shape
// is not eligible (caller falls back to the exact kernel), else a DV3_* code.
int dv3_conv_gemm_bf16x3_dispatch(const dv3_conv_desc* d, hipStream_t st) {
  const bool gated = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  if (d->a_bs != 0 || (d->lda & 3) || d->Tin != d->Tout) return 1;
  if ((d->J - 1) * d->dil > HALO_MAX || d->J * 2 > 32) return 1;
  // 32-bit byte offsets inside the kernel
  if ((int64_t)d->B * d->Tout >= (1ll << 30) || (int64_t)d->B * d->x_bs >= (1ll << 30)) return 1;
  if (d->xmask && (int64_t)d->B * d->Cin * d->xmask_rs >= (1ll << 30)) return 1;
  if ((int64_t)d->J * ((d->Cin + 31) / 32 * 4) * d->lda >= (1ll << 27)) return 1;
  // 256 x 256 tile, k16 ping-pong (conv_gemm_pp2.hip): tile_hint 30 forces it, dv3_debug_set(12, 1) prefers it
  const bool gated0 = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  const int64_t pp2_tiles = (int64_t)(gated0 ? dv3_cdiv(d->Cg, 128) : dv3_cdiv(d->M, 256)) * dv3_cdiv64((int64_t)d->B * d->Tout, 256);
  // with a stream-K workspace the 256 x 256 kernel no longer needs a grid that fills the chip: enough (tile, chunk)
  // units for every CU instead (g_x3_pp2_sk_units per CU, dv3_debug_set(25, v))
  const bool sk_fill = d->sk_ws && g_pp2_sk && (d->mode != DV3_EPI_DGRAD || g_pp2_sk >= 2) && g_x3_pp2 > 0 && pp2_tiles * (d->Cin / 32) >= (int64_t)g_x3_pp2_sk_units * 256;
  if (d->tile_hint == 30 || (g_x3_pp2 > 0 && d->tile_hint == 0 && (pp2_tiles >= g_x3_pp2 || sk_fill))) {
    const int rc = dv3_conv_gemm_pp2_dispatch(d, st);
    if (rc != 1) return rc;
    if (d->tile_hint == 30) return 1;
  }
  // the flat column axis needs batch-strided tensors only through (b, t) addressing: fine for all
  const TileCfg* best = pick_tile_x3(d, gated, d->tile_hint > 20 ? d->tile_hint - 20 : 0);
  if (!best) return 1;
  // k-split form (kernel template KS = 2, the 128 x 64 tile, three-term splits; 2 x (48 KB + halo) of LDS): grids of at most
  // one 128 x 64 tile per CU, k-ranges of at least 8 steps (profiles/r05_k_split.txt: 0.76-0.94 x there, 1.15-2 x the
  // time from 300 tiles up, nothing below 8 steps).  Where the picker went to a smaller tile for such a grid (64 x 64,
  // 128 x 32: more, lonelier workgroups) the k-split 128 x 64 tile is the faster way to more waves per CU.
  bool ks_ok = false;
  if (g_x3_ks && d->split_terms != 1 && (d->tile_hint == 0 || d->tile_hint == 22)) {
    const int64_t nb2 = (int64_t)(gated ? dv3_cdiv(d->Cg, 64) : dv3_cdiv(d->M, 128)) * dv3_cdiv64((int64_t)d->B * d->Tout, 64);
    const int nch = (d->Cin + 31) / 32;
    // (8 ... 15 k-steps: only grids of at most 160 tiles -- at B = 64 the 201-tile M = 80 layers measured 0.93-0.95 x
    //  stand-alone and +0.3 % on the step, where they share the chip with the weight-gradient stream)
    const int ksteps = nch * d->J;
    ks_ok = nch >= 2 && (g_x3_ks == 2 || (ksteps >= g_x3_ks_min_steps && nb2 <= (ksteps >= 16 ? g_x3_ks_max_blocks : g_x3_ks_max_blocks * 5 / 8)));
    if (ks_ok && d->tile_hint == 0 && (best->id == 4 || best->id == 6)) best = &kCfgs[1];
  }
  const int BM = best->wm * best->mi * 64, BMH = best->wm * best->mi * 32, BN = best->wn * best->ni * 32;
  const int BNH = BN + (d->J - 1) * d->dil;
  const size_t lds = (size_t)(2 * 2 * KB * BM + 2 * 2 * KB * BNH) * 16;
  if (lds > 160 * 1024) return 1;
  ConvArgs a;
  a.d = *d;
  a.a_scalar = 0;
  a.wide = g_x3_wide;
  a.prio = g_x3_prio;
  a.range_ctr = d->split_terms == DV3_SPLIT_F16X3 ? dv3_range_ctr() : nullptr;
  a.kp = (d->Cin + 31) / 32 * 32;
  a.m_tiles = gated ? dv3_cdiv(d->Cg, BMH) : dv3_cdiv(d->M, BM);
  a.n_tiles = (int)dv3_cdiv64((int64_t)d->B * d->Tout, BN);
  const int64_t nb = (int64_t)a.m_tiles * a.n_tiles;
  DV3_REQUIRE(nb < (1ll << 31), "conv_gemm: grid too large");
  a.n_blocks = (int)nb;
  a.ks = (ks_ok && best->id == 2 && 2 * lds <= 160 * 1024) ? 2 : 0;
  // deep-prefetch form (the 128 x 64 tile, 1 or 3 taps, three-term splits): experiment build only, off by default
  a.dp = 0;
#ifdef DV3_EXPERIMENTS
  if (best->id == 2 && d->split_terms != 1 && (d->J == 1 || d->J == 3) && g_x3_dp) {
    if (g_x3_dp == 2 || nb <= 2 * 256) a.dp = d->J;
  }
#endif
  g_dv3_last_conv = (d->split_terms == 1 ? 4000 : d->split_terms == DV3_SPLIT_F16X3 ? 5000 : 3000) + best->id * 10 +
                    ((best->id >= 8 && g_x3_pingpong) ? 1 : 0) + (a.dp ? 5 : 0) + (a.ks == 2 ? 2 : 0);
  if (d->pg) {
    // round 6: the fused gate backward lives in its own instantiations (bf16 pair, no dropout) of four tiles
    if (d->xmask || d->split_terms == DV3_SPLIT_F16X3 || d->split_terms == 1) return 1;
    g_dv3_last_conv += 600;            // 36xx: fused gate backward in the tail
    switch (best->id) {
      case 1: return launch_x3_m<2, 2, 2, false, 3, 1, false, false, 0, 1, true>(a, lds, st);
      case 2: return a.ks == 2 ? launch_x3_m<2, 2, 1, false, 3, 1, false, false, 0, 2, true>(a, lds, st)
                               : launch_x3_m<2, 2, 1, false, 3, 1, false, false, 0, 1, true>(a, lds, st);
      case 8: return launch_x3_m<4, 2, 2, false, 3, 1, true, false, 0, 1, true>(a, lds, st);
      case 9: return launch_x3_m<2, 4, 2, false, 3, 1, true, false, 0, 1, true>(a, lds, st);
    }
    return 1;
  }
  switch (best->id) {
    case 1: return launch_x3<2, 2, 2>(a, lds, st);
    case 2: return launch_x3<2, 2, 1>(a, lds, st);
    case 3: return launch_x3<4, 1, 1>(a, lds, st);
    case 4: return launch_x3<2, 1, 1>(a, lds, st);
    case 5: return launch_x3<1, 2, 2>(a, lds, st);
    case 6: return launch_x3<1, 2, 1>(a, lds, st);
    case 7: return launch_x3_big<2, 2, 2, 2>(a, lds, st);
    case 8: return launch_x3_big<4, 2, 2, 1>(a, lds, st);
    case 9: return launch_x3_big<2, 4, 2, 1>(a, lds, st);
  }
  return 1;
}

extern int g_wgrad_tile, g_wgrad_prio, g_wgrad_t2_abl, g_wgrad_taps2_default, g_wgrad_t2_window, g_wgrad_t2_il;   // wgrad_gemm_bf16x3.hip, wgrad_taps2.hip
extern int g_spk_prefetch;                                                                                 // speaker_bias.hip
extern int g_gate_c8_fast;                                                                                 // elementwise.hip
extern int g_loss_fast_log;                                                                                // loss.hip
extern int g_gate_vec;                                                                                     // elementwise.hip
extern int g_wn_bwd_vec4;                                                                                  // weight_norm.hip
extern int g_wgrad_c8_pf2, g_wgrad_c8_il, g_wgrad_c8_tr, g_spk_abl;                                                       // wgrad_c8.hip
int dv3_planes_debug_set(int what, int value);   // conv_planes.hip
int dv3_c8pp_debug_set(int what, int value);     // conv_c8pp.hip
int dv3_conv_census_set(int on);                 // conv_gemm.hip
extern "C" int dv3_debug_set(int what, int value) {
#ifndef DV3_EXPERIMENTS
  // the timing-only ablation / stamp instantiations are compiled with `make EXP=1` only: say so instead of silently
  // timing the production kernel
  DV3_REQUIRE(!(value != 0 && (what == 1 || what == 6 || what == 13 || what == 16 || what == 21 || what == 26 || what == 28 || what == 32 || what == 43)),
              "debug_set(%d, %d): ablation variants are not in this build (make EXP=1)", what, value);
#endif
  if (what >= 4 && what <= 8) return dv3_planes_debug_set(what, value);
  if (what == 40) return dv3_conv_census_set(value);
  if (what == 19 || what == 21 || what == 30 || what == 32 || (what >= 34 && what <= 36)) return dv3_c8pp_debug_set(what, value);
  if (what == 20) g_wgrad_c8_pf2 = value;
  if (what == 49) g_wgrad_c8_il = value;
  if (what == 9) g_x3_rel2 = value;
  if (what == 12) g_x3_pp2 = value;
  if (what == 13) g_pp2_abl = value;
  if (what == 29 || what == 31) {
    const bool ship = value == 0 || value == 17 || value == 81;
#ifndef DV3_EXPERIMENTS
    DV3_REQUIRE(ship, "debug_set(%d, %d): this LOAD-phase variant is not in this build (make EXP=1)", what, value);
#endif
    if (what == 29) { g_pp2_ord_u = ship ? value : 0; g_pp2_ord = ship ? 0 : value; }
    else g_pp2_ord_m = ship ? value : 0;
  }
  if (what == 50) g_pp2_fast_tail = value;
  if (what == 51) g_wn_bwd_vec4 = value;
  if (what == 52) g_wgrad_c8_tr = value;
  if (what == 54) g_spk_prefetch = value;
  if (what == 55) g_gate_vec = value;
  if (what == 56) g_gate_c8_fast = value;
  if (what == 57) g_loss_fast_log = value;
  if (what == 22) g_pp2_sk = value;
  if (what == 23) g_pp2_sk_overhead = value;
  if (what == 24) g_pp2_sk_gain = value;
  if (what == 25) g_x3_pp2_sk_units = value;
  if (what == 26) g_pp2_sk_abl = value;
  if (what == 27) g_x3_j1_flat = value;
  if (what == 43) g_x3_dp = value;
  if (what == 42) g_x3_rel8 = value;
  if (what == 44) g_x3_ks = value;
  if (what == 45) g_x3_ks_max_blocks = value;
  if (what == 46) g_x3_ks_min_steps = value;
  if (what == 28) g_spk_abl = value;
  if (what == 14) g_x3_prio = value;
  if (what == 18) g_x3_wide = value;
  if (what == 15) g_wgrad_prio = value;
  if (what == 16) g_wgrad_t2_abl = value;
  if (what == 17) g_wgrad_taps2_default = value;
  if (what == 47) g_wgrad_t2_window = value;
  if (what == 48) g_wgrad_t2_il = value;
  if (what == 1) g_x3_ablate = value;
  if (what == 2) g_wgrad_tile = value;
  if (what == 3) g_x3_pingpong = value;
  return DV3_OK;
}

int dv3_pp2_read_stamps(void* dst, int64_t bytes);   // conv_gemm_pp2.hip
int dv3_decode_read_stamps(void* dst, int64_t bytes);  // decode_step.hip
int dv3_conv_census_read(int what, void* dst, int64_t bytes);   // conv_gemm.hip
extern "C" int dv3_debug_read(int what, void* dst, int64_t bytes) {
  if (what == 40 || what == 41) return dv3_conv_census_read(what, dst, bytes);
  if (what == 2 && dst) return dv3_pp2_read_stamps(dst, bytes);
  if (what == 3 && dst) return dv3_decode_read_stamps(dst, bytes);
  DV3_REQUIRE(what == 1 && dst && bytes > 0 && bytes <= (int64_t)sizeof(unsigned long long) * 8 * STAMP_SLOTS * 2,
              "debug_read: bad arguments");
  hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_x3_stamps), (size_t)bytes, 0, hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    dv3_set_error("debug_read: %s", hipGetErrorString(e));
    return DV3_ELAUNCH;
  }
  return DV3_OK;
}

extern "C" int dv3_split_pack_bf16(const float* packed, uint16_t* out, int32_t J, int32_t K,
                                   int32_t lda, int32_t dtype, void* stream) {
  DV3_REQUIRE(dtype == DV3_SPLIT_DTYPE_BF16 || dtype == DV3_SPLIT_DTYPE_F16, "split_pack: bad dtype");
  DV3_REQUIRE(packed && out, "split_pack: null pointer");
  DV3_REQUIRE(J >= 1 && K >= 1 && lda >= 1 && ((uintptr_t)out & 15) == 0, "split_pack: bad arguments");
  const int k8_total = (K + 31) / 32 * 4;
  const int64_t n = (int64_t)J * k8_total * lda;
  hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)dv3_cdiv64(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, packed, reinterpret_cast<bf16x8*>(out), J, K, lda, k8_total, (int)dtype,
                     dv3_range_ctr());
  return dv3_check_launch("split_pack_bf16");
}
#endif   // DV3_X3_ISA_ONLY
