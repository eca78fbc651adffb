// Weight-gradient GEMM of the bf16-storage path (BASELINE configs 3/4): both operands are channel-blocked bf16
// tensors  [B][C8][T][8]  (include/dv3hip.h "c8"), single-term bf16 MFMA, fp32 accumulate.  Autograd of F.conv1d
// w.r.t. its weight (reference call sites deepvoice3_pytorch/modules.py:153,216 through loss.backward()):
//
//   out[s][j][m][c] = sum_{(b, chunk) in slab s} sum_t g[b][m][t] * xd[b][c][t + j*dil - padL]
//
// Same GEMM view, tile (128 gradient rows x 128 input channels, 8 waves, all JT taps per workgroup), LDS image
// [k8][row + pad][8 time steps], slab partition and accumulation order as wgrad_taps_kernel (wgrad_gemm_bf16x3.hip).
// What differs is staging: the MFMA K axis is TIME, while a c8 unit holds 8 CHANNELS of one frame, so a thread takes
// an 8-channel x 4-frame block (four 16-byte loads, any tap shift is unit-aligned), transposes it in registers
// (16 v_perm_b32) and writes eight 8-byte half units.  Nothing is converted: the operands are already bf16.  The
// dropout keep-bytes of x (one byte per unit) expand to 16-byte AND masks through a 256-entry table in LDS.
#include "common.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace {

struct WgradC8Args {
  dv3_wgrad_desc d;
  int m_tiles, c_tiles;
};

constexpr int BKT = 32;  // time steps per K step
constexpr int KB = 4;    // k8 blocks per K step
constexpr int PAD = 2;   // units of padding per k8 block

template <typename T>
__device__ __forceinline__ T ldg_off(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// PF2 (round 4, default): operands fetched TWO steps ahead into two register sets, as wgrad_taps2.hip does for the split
// kernels (both operands stream from HBM; one step of MFMAs -- ~0.4 us -- does not cover a fetch issued right before
// it).  Every load is unconditional (frames clamped into the row), validity is recomputed from the step index when the
// set is transposed, the tail re-fetches the last step.  Same tile, LDS image and accumulation order: bit-identical
// slabs.  One workgroup per CU either way (the grid is sized to the chip), so the kernel takes the 256-register budget.
// IL (round 6, three-tap PF2 form): the staging of the next step's tile (zero padding, keep masks, the 8 x 4 transposes,
// eight LDS stores: ~70 vector instructions per thread that touch nothing the current step's twelve MFMAs read) is issued
// BETWEEN those MFMAs (__builtin_amdgcn_sched_group_barrier) instead of after them -- all eight waves run in phase, so
// "after" leaves the matrix pipes idle while every wave transposes (wgrad_taps2.hip, IL).  Same values: bit-identical.
template <int JT, bool MASK, bool PF2, bool IL = false>
__global__ __launch_bounds__(512) void wgrad_c8_kernel(const WgradC8Args args) {
  static_assert(!IL || (PF2 && JT == 3), "interleaved staging: the three-tap two-steps-ahead form (every thread stages)");
  constexpr int BM = 128, BN = 128;
  constexpr int LDM = BM + PAD, LDN = BN + PAD;
  constexpr int GBUF = KB * LDM, XTAP = KB * LDN, BUF = GBUF + JT * XTAP;   // 16-byte units per buffer
  constexpr int NSTG = (1 + JT) * 128;                                      // staging threads
  const dv3_wgrad_desc& p = args.d;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_c8[];
  bf16x8* const smem = reinterpret_cast<bf16x8*>(smem_raw_c8);              // [2 buffers][G | JT x X]
  u32x4* const lut = reinterpret_cast<u32x4*>(smem + 2 * BUF);              // keep-byte -> 8 x 16-bit lane masks

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, lhi = lane >> 5;

  int pid = dv3_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = pid % args.m_tiles; pid /= args.m_tiles;
  const int ct = pid % args.c_tiles;
  const int s = pid / args.c_tiles;
  const int m0 = mt * BM, c0 = ct * BN;
  const int T = p.T, M = p.M, Cin = p.Cin;      // Tin == T (same-length layers)
  const int c8g = (M + 31) / 32 * 4, c8x = (Cin + 31) / 32 * 4;

  if (MASK && tid < 256) {
    u32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      e[i] = (((uint32_t)tid >> (2 * i)) & 1u ? 0xffffu : 0u) | (((uint32_t)tid >> (2 * i + 1)) & 1u ? 0xffff0000u : 0u);
    lut[tid] = e;
  }

  // this thread's staging block: panel 0 = g, 1.. = the x taps; an 8-channel group x 4 frames (half a k8 block)
  const int panel = tid >> 7;
  const int blk = (tid & 127) >> 1, hsel = tid & 1;
  const int grp = blk >> 2, k8 = blk & 3;
  const bool stager = tid < NSTG;
  const bool is_g = panel == 0;
  const int shift = is_g ? 0 : (panel - 1) * p.dil - p.padL;
  const int grow = (is_g ? m0 : c0) / 8 + grp;                  // channel group inside the tensor
  const bool grp_ok = stager && grow * 8 < (is_g ? M : Cin);
  const int c8t = is_g ? c8g : c8x;
  const char* const src = reinterpret_cast<const char*>(is_g ? (const void*)p.g : (const void*)p.x);
  const int tq = k8 * 8 + hsel * 4 + shift;                     // first frame of the block relative to the chunk start

  const int n_tc = (T + BKT - 1) / BKT;
  int nsteps, step0;
  {
    const int total = p.B * n_tc, q = (total + p.n_slabs - 1) / p.n_slabs;
    step0 = s * q;
    nsteps = max(0, min(q, total - step0));
  }

  constexpr int NSET = PF2 ? 2 : 1;
  u32x4 ru[NSET][4];
  uint32_t rkeep[MASK ? NSET : 1][4];
  // `step` may run past the end (PF2's look-ahead): it is clamped to the last step, whose re-fetched data is never used
  auto load_step = [&](int step, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    if (!stager) return;
    const int gs = step0 + min(step, nsteps - 1);
    const int b = gs / n_tc, tc = gs - b * n_tc;
    const int t0 = tc * BKT + tq;
    const uint32_t ubase = (uint32_t)((b * c8t + (grp_ok ? grow : 0)) * T);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t u = ubase + (uint32_t)min(max(t0 + i, 0), T - 1);
      ru[S][i] = ldg_off<u32x4>(src, u * 16u);
      if constexpr (MASK) rkeep[S][i] = is_g ? 0xffu : (uint32_t)ldg_off<uint8_t>(p.xmask_c8, u);
    }
  };
  auto write_step = [&](int step, int buf, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    if constexpr (NSTG < 512) {      // (JT == 3: every thread stages -- and a branch would cut IL's scheduling region)
      if (!stager) return;
    }
    const int gs = step0 + min(step, nsteps - 1);
    const int t0 = (gs % n_tc) * BKT + tq;
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + i;
      if (!(grp_ok && t >= 0 && t < T)) ru[S][i] = zero;       // the conv's zero padding / rows beyond the tensor
      if constexpr (MASK) ru[S][i] &= lut[rkeep[S][i]];
    }
    // 8 channels x 4 frames -> per channel e: frames (0,1) and (2,3) packed into two dwords
    bf16x8* dst = smem + buf * BUF + (is_g ? 0 : GBUF + (panel - 1) * XTAP) + k8 * (is_g ? LDM : LDN) + grp * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int q = e >> 1;
      const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
      u32x2 o;
      o[0] = __builtin_amdgcn_perm(ru[S][1][q], ru[S][0][q], sel);
      o[1] = __builtin_amdgcn_perm(ru[S][3][q], ru[S][2][q], sel);
      reinterpret_cast<u32x2*>(dst + e)[hsel] = o;
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, PF2 ? 1 : 0>;

  f32x16 acc[JT][2];
#pragma unroll
  for (int j = 0; j < JT; ++j)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][a][r] = 0.f;

  auto mfma_step = [&](int cur) {
    const bf16x8* Gs = smem + cur * BUF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = 2 * ks + lhi;
      const int ai = kk * LDM + wm * 64 + l31;
      const bf16x8 a0 = Gs[ai], a1 = Gs[ai + 32];
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const bf16x8 bh = Gs[GBUF + j * XTAP + kk * LDN + wc * 32 + l31];
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[j][1], 0, 0, 0);
      }
    }
  };

  __syncthreads();          // the keep-byte table
  if constexpr (PF2) {
    if (nsteps > 0) {
      // prologue: step 0 -> buffer 0; set 1 <- step 1, set 0 <- step 2
      load_step(0, S0{});
      load_step(1, S1{});
      write_step(0, 0, S0{});
      load_step(2, S0{});
      __syncthreads();
      // steps in pairs so the register-set index is static: step st + 1 is staged from set (st + 1) & 1
      // IL: the step as six fenced segments -- [two MFMAs of one (k16 block, tap)] [the fragment reads of the next
      // segment] [a sixth of the next tile's staging] -- in source order (sched_barrier(0) between segments: the
      // compiler's own pipeline solver did not take this kernel's staging apart).  Per accumulator the k16 blocks are
      // added in the order of mfma_step: bit-identical.
      auto stage_unit = [&](int t0, auto set_c, auto ic) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value, i = decltype(ic)::value;
        const u32x4 zero = {0u, 0u, 0u, 0u};
        const int t = t0 + i;
        if (!(grp_ok && t >= 0 && t < T)) ru[S][i] = zero;
        if constexpr (MASK) ru[S][i] &= lut[rkeep[S][i]];
      };
      auto store_half = [&](int buf, auto set_c, auto hc) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value, h = decltype(hc)::value;
        bf16x8* dst = smem + buf * BUF + (is_g ? 0 : GBUF + (panel - 1) * XTAP) + k8 * (is_g ? LDM : LDN) + grp * 8;
#pragma unroll
        for (int e = 4 * h; e < 4 * h + 4; ++e) {
          const int q = e >> 1;
          const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
          u32x2 o;
          o[0] = __builtin_amdgcn_perm(ru[S][1][q], ru[S][0][q], sel);
          o[1] = __builtin_amdgcn_perm(ru[S][3][q], ru[S][2][q], sel);
          reinterpret_cast<u32x2*>(dst + e)[hsel] = o;
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      auto step = [&](int st, auto set_c) __attribute__((always_inline)) {
        if constexpr (!IL) {
          mfma_step(st & 1);
          write_step(st + 1, (st + 1) & 1, set_c);     // past the end: a re-fetched tile into the buffer nobody reads
        } else {
          const bf16x8* Gs = smem + (st & 1) * BUF;
          const int gs = step0 + min(st + 1, nsteps - 1);
          const int t0 = (gs % n_tc) * BKT + tq;
          const int nbuf = (st + 1) & 1;
          const int ai0 = lhi * LDM + wm * 64 + l31, ai1 = (2 + lhi) * LDM + wm * 64 + l31;
          const int xi0 = GBUF + lhi * LDN + wc * 32 + l31, xi1 = GBUF + (2 + lhi) * LDN + wc * 32 + l31;
          __builtin_amdgcn_sched_barrier(0);
          bf16x8 a0 = Gs[ai0], a1 = Gs[ai0 + 32], b = Gs[xi0];
          __builtin_amdgcn_sched_barrier(0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[0][1], 0, 0, 0);
          bf16x8 b1 = Gs[xi0 + XTAP];
          stage_unit(t0, set_c, I0{});
          __builtin_amdgcn_sched_barrier(0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
          b = Gs[xi0 + 2 * XTAP];
          stage_unit(t0, set_c, I1{});
          __builtin_amdgcn_sched_barrier(0);
          acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[2][0], 0, 0, 0);
          acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[2][1], 0, 0, 0);
          a0 = Gs[ai1]; a1 = Gs[ai1 + 32]; b1 = Gs[xi1];
          stage_unit(t0, set_c, I2{});
          __builtin_amdgcn_sched_barrier(0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[0][1], 0, 0, 0);
          b = Gs[xi1 + XTAP];
          stage_unit(t0, set_c, I3{});
          __builtin_amdgcn_sched_barrier(0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1][1], 0, 0, 0);
          b1 = Gs[xi1 + 2 * XTAP];
          store_half(nbuf, set_c, I0{});
          __builtin_amdgcn_sched_barrier(0);
          acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[2][0], 0, 0, 0);
          acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[2][1], 0, 0, 0);
          store_half(nbuf, set_c, I1{});
          __builtin_amdgcn_sched_barrier(0);
        }
        load_step(st + 3, set_c);
        __syncthreads();
      };
      for (int st = 0; st < nsteps; st += 2) {
        step(st, S1{});
        if (st + 1 < nsteps) step(st + 1, S0{});
      }
    }
  } else {
    if (nsteps > 0) {
      load_step(0, S0{});
      write_step(0, 0, S0{});
    }
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
      if (step + 1 < nsteps) load_step(step + 1, S0{});
      mfma_step(step & 1);
      if (step + 1 < nsteps) write_step(step + 1, (step + 1) & 1, S0{});
      __syncthreads();
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

template <int JT, bool MASK, bool PF2, bool IL = false>
int launch_c8(const WgradC8Args& a, int64_t nb, hipStream_t st) {
  constexpr int LDM = 128 + PAD;
  constexpr size_t lds = (size_t)2 * (KB * LDM + JT * KB * LDM) * 16 + 256 * 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_c8_kernel<JT, MASK, PF2, IL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      dv3_set_error("wgrad_c8: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_c8_kernel<JT, MASK, PF2, IL>), dim3((unsigned)nb), dim3(512), lds, st, a);
  return dv3_check_launch("wgrad_c8");
}

// TR (round 6): the same GEMM with the operand tiles left in the tensors' own unit form -- LDS image [channel group][frame]
// of 16-byte units (8 channels of one frame), written as loaded -- and the MFMA fragments (8 consecutive FRAMES of one
// channel per lane) cut from it by the hardware's transposing read, ds_read_b64_tr_b16: per 16-lane group a
// [4 frames][16 channels] block, lane i supplying the 8 bytes of frame i / 4, channels 4 (i % 4) .. + 3, and receiving
// channel i's four frames.  What that removes from the loop: the 8 x 4 register transposes (16 v_perm_b32 per thread and
// step), the three shifted copies of the x tile (a tap is a frame offset of the read address: x is staged ONCE, as a
// window of 32 + (J - 1) dil frames), half of the LDS write volume.  Same tile, slab partition, K order and lane
// positions of every frame inside the MFMAs as wgrad_c8_kernel: bit-identical slabs.
// Bank rule of the read (MI355X_MICROARCH.md: two passes of 32 lanes, bank = (addr / 4) % 64): a pass covers four
// consecutive channel groups x 64 contiguous bytes, so the group stride must be 64 or 192 (mod 256) bytes: frames per
// group padded to 4 or 12 (mod 16) -- 36 for g, `twx` for the window.
struct WgradC8TrArgs {
  dv3_wgrad_desc d;
  int m_tiles, c_tiles;
  int wx, twx;       // frames of the x window; its padded row length in the LDS image
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 lds_tr(const unsigned char* base, int byte_off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + byte_off));
}
__device__ __forceinline__ bf16x8 lds_tr8(const unsigned char* base, int byte_off) {   // frames 0..3 | 4..7 of this lane's channel
  const s16x4 lo = lds_tr(base, byte_off), hi = lds_tr(base, byte_off + 64);
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int JT, bool MASK, int NXU, bool IL = false, int ABL = 0, bool PAIR = false>
__global__ __launch_bounds__(512) void wgrad_c8_tr_kernel(const WgradC8TrArgs args) {
  static_assert(!IL || JT == 3, "interleaved staging: the three-tap form");
  static_assert(NXU >= 1 && NXU <= 3, "x units a thread stages: 16 * window <= 512 NXU");
  constexpr int BM = 128, BN = 128, TWG = 36;
  constexpr int GU = 16 * TWG;                 // units of the g image
  const dv3_wgrad_desc& p = args.d;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_tr[];
  const int twx = args.twx, wx = args.wx;
  const int BUFU = GU + 16 * twx;
  // PAIR: four buffers and ONE barrier per two K steps -- step st computes from buffer st % 4 while step st + 2 is written
  // (the barrier-to-barrier stretch holds 24 MFMAs per wave instead of 12; same order of everything that is added)
  constexpr int DST = PAIR ? 2 : 1, BMSK = 2 * DST - 1;
  u32x4* const smem = reinterpret_cast<u32x4*>(smem_raw_tr);                // [2 DST buffers][g | x window]
  u32x4* const lut = smem + 2 * DST * BUFU;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, lhi = lane >> 5;

  int pid = dv3_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = pid % args.m_tiles; pid /= args.m_tiles;
  const int ct = pid % args.c_tiles;
  const int s = pid / args.c_tiles;
  const int m0 = mt * BM, c0 = ct * BN;
  const int T = p.T, M = p.M, Cin = p.Cin;
  const int c8g = (M + 31) / 32 * 4, c8x = (Cin + 31) / 32 * 4;

  if (MASK && tid < 256) {
    u32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      e[i] = (((uint32_t)tid >> (2 * i)) & 1u ? 0xffffu : 0u) | (((uint32_t)tid >> (2 * i + 1)) & 1u ? 0xffff0000u : 0u);
    lut[tid] = e;
  }

  // ---- this thread's staging units, fixed over the K loop: one of g, up to NXU of the x window ----
  const int gcb = tid >> 5, gtt = tid & 31;
  const bool g_ok = (m0 / 8 + gcb) * 8 < M;
  const int g_row = m0 / 8 + (g_ok ? gcb : 0);
  const int g_dst = gcb * TWG + gtt;
  bool x_in[NXU], x_ok[NXU];
  int x_row[NXU], x_tt[NXU], x_dst[NXU];
#pragma unroll
  for (int n = 0; n < NXU; ++n) {
    const int u = tid + 512 * n;
    const int cb = u / wx;
    x_in[n] = cb < 16;
    x_tt[n] = u - cb * wx - p.padL;
    x_ok[n] = x_in[n] && (c0 / 8 + cb) * 8 < Cin;
    x_row[n] = c0 / 8 + (x_ok[n] ? cb : 0);
    x_dst[n] = GU + cb * twx + (u - cb * wx);
  }

  const int n_tc = (T + BKT - 1) / BKT;
  int nsteps, step0;
  {
    const int total = p.B * n_tc, q = (total + p.n_slabs - 1) / p.n_slabs;
    step0 = s * q;
    nsteps = max(0, min(q, total - step0));
  }

  u32x4 rg[2], rx[2][NXU];
  uint32_t rk[2][NXU];
  // `step` may run past the end (the look-ahead): clamped to the last step, whose re-fetched data is never used
  auto load_step = [&](int step, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    const int gs = step0 + min(step, nsteps - 1);
    const int b = gs / n_tc, tc = gs - b * n_tc;
    const int t0 = tc * BKT;
    rg[S] = ldg_off<u32x4>(p.g, ((uint32_t)((b * c8g + g_row) * T) + (uint32_t)min(t0 + gtt, T - 1)) * 16u);
#pragma unroll
    for (int n = 0; n < NXU; ++n) {
      const uint32_t u = (uint32_t)((b * c8x + x_row[n]) * T) + (uint32_t)min(max(t0 + x_tt[n], 0), T - 1);
      // every load unconditional (a thread without an n-th unit re-reads a clamped address): with loads under a
      // branch the compiler cannot count what is in flight and drains the queue (vmcnt(0)) at every barrier
      rx[S][n] = ldg_off<u32x4>(p.x, u * 16u);
      if constexpr (MASK) rk[S][n] = (uint32_t)ldg_off<uint8_t>(p.xmask_c8, u);
    }
  };
  auto write_step = [&](int step, int buf, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    const int gs = step0 + min(step, nsteps - 1);
    const int t0 = (gs % n_tc) * BKT;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    u32x4* const dst = smem + buf * BUFU;
    dst[g_dst] = (g_ok && t0 + gtt < T) ? rg[S] : zero;            // the conv's zero padding / rows beyond the tensor
#pragma unroll
    for (int n = 0; n < NXU; ++n) {
      const int t = t0 + x_tt[n];
      u32x4 v = (x_ok[n] && t >= 0 && t < T) ? rx[S][n] : zero;
      if constexpr (MASK) v &= lut[rk[S][n]];
      if (x_in[n]) dst[x_dst[n]] = v;     // (units beyond the first 512: whole waves skip)
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x16 acc[JT][2];
#pragma unroll
  for (int j = 0; j < JT; ++j)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][a][r] = 0.f;

  // ---- fragment addresses (bytes inside a buffer): 16-lane group gi -> channels 16 gi .. + 15 of the wave's 32;
  //      lane i of it supplies frame i / 4, channel quarter i % 4 ----
  const int li = lane & 15, gi = (lane >> 4) & 1, fr = li >> 2, qt = li & 3;
  const int a_off = ((wm * 8 + gi * 2 + (qt >> 1)) * TWG + 8 * lhi + fr) * 16 + (qt & 1) * 8;
  int b_off[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j)
    b_off[j] = (GU + (wc * 4 + gi * 2 + (qt >> 1)) * twx + 8 * lhi + fr + j * p.dil) * 16 + (qt & 1) * 8;

  auto mfma_step = [&](int cur) __attribute__((always_inline)) {
    const unsigned char* const Bs = smem_raw_tr + (size_t)cur * BUFU * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 a0 = lds_tr8(Bs + a_off, ks * 256), a1 = lds_tr8(Bs + a_off, ks * 256 + 4 * TWG * 16);
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const bf16x8 bh = lds_tr8(Bs + b_off[j], ks * 256);
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[j][1], 0, 0, 0);
      }
    }
  };

  __syncthreads();          // the keep-byte table
  if (nsteps > 0) {
    load_step(0, S0{});
    load_step(1, S1{});
    write_step(0, 0, S0{});
    if constexpr (PAIR) write_step(1, 1, S1{});
    load_step(2, S0{});
    if constexpr (PAIR) load_step(3, S1{});
    __syncthreads();
    auto step = [&](int st, auto set_c) __attribute__((always_inline)) {
      if constexpr (!IL) {
        if constexpr (ABL != 1) mfma_step(st & BMSK);
        if constexpr (ABL != 2) write_step(st + DST, (st + DST) & BMSK, set_c);     // past the end: a re-fetched tile into a buffer nobody reads
        if constexpr (ABL != 3) load_step(st + 2 + DST, set_c);
      } else {
        // the step as six fenced segments, [two MFMAs of one (k16 block, tap)] [the fragment reads of the next segment]
        // [a piece of the next tile's staging], as wgrad_c8_kernel's IL form; per accumulator the k16 blocks are added in
        // mfma_step's order: bit-identical
        constexpr int S = decltype(set_c)::value;
        const unsigned char* const Bs = smem_raw_tr + (size_t)(st & BMSK) * BUFU * 16;
        u32x4* const dst = smem + ((st + DST) & BMSK) * BUFU;
        const int gs = step0 + min(st + DST, nsteps - 1);
        const int t0 = (gs % n_tc) * BKT;
        const u32x4 zero = {0u, 0u, 0u, 0u};
        auto stage_x = [&](auto nc, const u32x4 keep) __attribute__((always_inline)) {
          constexpr int n = decltype(nc)::value;
          const int t = t0 + x_tt[n];
          u32x4 v = (x_ok[n] && t >= 0 && t < T) ? rx[S][n] : zero;
          if constexpr (MASK) v &= keep;
          dst[x_dst[n]] = v;
        };
        using N0 = std::integral_constant<int, 0>;
        using N1 = std::integral_constant<int, 1>;
        using N2 = std::integral_constant<int, 2>;
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 a0 = lds_tr8(Bs + a_off, 0), a1 = lds_tr8(Bs + a_off, 4 * TWG * 16), b = lds_tr8(Bs + b_off[0], 0);
        u32x4 k0 = zero;
        if constexpr (MASK) k0 = lut[rk[S][0]];
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[0][1], 0, 0, 0);
        bf16x8 b1 = lds_tr8(Bs + b_off[1], 0);
        dst[g_dst] = (g_ok && t0 + gtt < T) ? rg[S] : zero;
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        b = lds_tr8(Bs + b_off[2], 0);
        stage_x(N0{}, k0);
        __builtin_amdgcn_sched_barrier(0);
        acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[2][1], 0, 0, 0);
        a0 = lds_tr8(Bs + a_off, 256); a1 = lds_tr8(Bs + a_off, 256 + 4 * TWG * 16); b1 = lds_tr8(Bs + b_off[0], 256);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[0][1], 0, 0, 0);
        b = lds_tr8(Bs + b_off[1], 256);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1][1], 0, 0, 0);
        b1 = lds_tr8(Bs + b_off[2], 256);
        __builtin_amdgcn_sched_barrier(0);
        acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[2][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // the window's units beyond the first 512 (32 of them at dilation 1): whole waves skip
        if constexpr (NXU > 1) { if (x_in[1]) stage_x(N1{}, MASK ? lut[rk[S][1]] : zero); }
        if constexpr (NXU > 2) { if (x_in[2]) stage_x(N2{}, MASK ? lut[rk[S][2]] : zero); }
        load_step(st + 2 + DST, set_c);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!PAIR || (st & 1)) __syncthreads();
    };
    // (the set a step stages from holds the step DST ahead: the odd set at even steps without PAIR, the even one with it)
    for (int st = 0; st < nsteps; st += 2) {
      if constexpr (PAIR) step(st, S0{}); else step(st, S1{});
      if (st + 1 < nsteps) { if constexpr (PAIR) step(st + 1, S1{}); else step(st + 1, S0{}); }
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

template <int JT, bool MASK, int NXU, bool IL = false, int ABL = 0, bool PAIR = false>
int launch_c8_tr(const WgradC8TrArgs& a, int64_t nb, hipStream_t st) {
  const size_t lds = (size_t)(PAIR ? 4 : 2) * (16 * 36 + 16 * a.twx) * 16 + 256 * 16;   // the buffers, keep table
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_c8_tr_kernel<JT, MASK, NXU, IL, ABL, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      dv3_set_error("wgrad_c8: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_c8_tr_kernel<JT, MASK, NXU, IL, ABL, PAIR>), dim3((unsigned)nb), dim3(512), lds, st, a);
  return dv3_check_launch("wgrad_c8_tr");
}

}  // namespace

// dv3_debug_set(52, v): operand fragments by ds_read_b64_tr_b16 from the untransposed tile -- 0 = the register-transposing
// forms, 1 = plain, 2 = staging between the MFMAs, 3 = one barrier per two K steps, 4 = both (default; one tap: as 3),
// 11..13 = timing ablations (no MFMAs / no LDS writes / no global loads)
int g_wgrad_c8_tr = 4;
int g_wgrad_c8_il = 1;    // dv3_debug_set(49, v): the three-tap form stages the next tile between the MFMAs (0 = after them)
int g_wgrad_c8_pf2 = 1;   // dv3_debug_set(20, v): 1 = operands fetched two steps ahead (default), 0 = the round-2 one-step form

// called by dv3_wgrad_gemm_f32 (wgrad_gemm.hip) when d->c8 is set
int dv3_wgrad_c8_dispatch(const dv3_wgrad_desc* d, hipStream_t st) {
  DV3_REQUIRE(d->J == 1 || d->J == 3, "wgrad_gemm: the c8 form serves 1 and 3 taps (J=%d)", d->J);
  DV3_REQUIRE(d->T == d->Tin && d->k_split && d->split_bf16 == 2, "wgrad_gemm: the c8 form is the single-term bf16, "
              "same-length, contiguous-K-split kernel");
  DV3_REQUIRE((((uintptr_t)d->g | (uintptr_t)d->x) & 15) == 0, "wgrad_gemm: c8 tensors need 16-byte alignment");
  DV3_REQUIRE(!d->xmask, "wgrad_gemm: the c8 form takes keep-bytes (xmask_c8), not keep-bits");
  const int64_t c8g = (d->M + 31) / 32 * 4, c8x = (d->Cin + 31) / 32 * 4;
  DV3_REQUIRE((int64_t)d->B * c8g * d->T < (1ll << 28) && (int64_t)d->B * c8x * d->T < (1ll << 28),
              "wgrad_gemm: a c8 tensor exceeds the 4 GB the kernel can address");
  WgradC8Args a;
  a.d = *d;
  a.m_tiles = dv3_cdiv(d->M, 128);
  a.c_tiles = dv3_cdiv(d->Cin, 128);
  const int64_t nb = (int64_t)a.m_tiles * a.c_tiles * d->n_slabs;
  DV3_REQUIRE(nb < (1ll << 31), "wgrad_gemm: grid too large");
  const int wx = 32 + (d->J - 1) * d->dil;
  if (g_wgrad_c8_tr && wx <= 96 && d->padL >= 0 && d->padL <= (d->J - 1) * d->dil) {
    WgradC8TrArgs t;
    t.d = *d; t.m_tiles = a.m_tiles; t.c_tiles = a.c_tiles; t.wx = wx;
    for (t.twx = wx; (t.twx & 15) != 4 && (t.twx & 15) != 12; ++t.twx) {}
    g_dv3_last_wgrad = 5100 + d->J;       // 5101 / 5103: fragments by the transposing LDS read
    const bool wide = 16 * wx > 1024;     // a third x unit per thread (dilation 27)
    if (d->J == 3 && d->xmask_c8 && !wide && g_wgrad_c8_tr >= 11 && g_wgrad_c8_tr <= 13) {     // ablations (timing only)
      if (g_wgrad_c8_tr == 11) return launch_c8_tr<3, true, 2, false, 1>(t, nb, st);
      if (g_wgrad_c8_tr == 12) return launch_c8_tr<3, true, 2, false, 2>(t, nb, st);
      return launch_c8_tr<3, true, 2, false, 3>(t, nb, st);
    }
    if (d->J == 3 && (g_wgrad_c8_tr == 3 || g_wgrad_c8_tr == 4)) {
      const bool il = g_wgrad_c8_tr == 4;
      g_dv3_last_wgrad += il ? 240 : 200;             // 5303 / 5343: one barrier per two K steps
      if (wide) {
        if (il) return d->xmask_c8 ? launch_c8_tr<3, true, 3, true, 0, true>(t, nb, st) : launch_c8_tr<3, false, 3, true, 0, true>(t, nb, st);
        return d->xmask_c8 ? launch_c8_tr<3, true, 3, false, 0, true>(t, nb, st) : launch_c8_tr<3, false, 3, false, 0, true>(t, nb, st);
      }
      if (il) return d->xmask_c8 ? launch_c8_tr<3, true, 2, true, 0, true>(t, nb, st) : launch_c8_tr<3, false, 2, true, 0, true>(t, nb, st);
      return d->xmask_c8 ? launch_c8_tr<3, true, 2, false, 0, true>(t, nb, st) : launch_c8_tr<3, false, 2, false, 0, true>(t, nb, st);
    }
    if (d->J == 1 && (g_wgrad_c8_tr == 3 || g_wgrad_c8_tr == 4)) {
      g_dv3_last_wgrad += 200;
      return d->xmask_c8 ? launch_c8_tr<1, true, 1, false, 0, true>(t, nb, st) : launch_c8_tr<1, false, 1, false, 0, true>(t, nb, st);
    }
    if (d->J == 3 && g_wgrad_c8_tr == 2) {
      g_dv3_last_wgrad += 40;             // 5143: staging between the MFMAs
      if (wide) return d->xmask_c8 ? launch_c8_tr<3, true, 3, true>(t, nb, st) : launch_c8_tr<3, false, 3, true>(t, nb, st);
      return d->xmask_c8 ? launch_c8_tr<3, true, 2, true>(t, nb, st) : launch_c8_tr<3, false, 2, true>(t, nb, st);
    }
    if (d->J == 3 && wide) return d->xmask_c8 ? launch_c8_tr<3, true, 3>(t, nb, st) : launch_c8_tr<3, false, 3>(t, nb, st);
    if (d->J == 3) return d->xmask_c8 ? launch_c8_tr<3, true, 2>(t, nb, st) : launch_c8_tr<3, false, 2>(t, nb, st);
    return d->xmask_c8 ? launch_c8_tr<1, true, 1>(t, nb, st) : launch_c8_tr<1, false, 1>(t, nb, st);
  }
  if (g_wgrad_c8_pf2) {
    g_dv3_last_wgrad = 5000 + 20 + d->J;
    if (d->J == 3 && g_wgrad_c8_il) {
      g_dv3_last_wgrad += 40;            // 5063: staging between the MFMAs
      return d->xmask_c8 ? launch_c8<3, true, true, true>(a, nb, st) : launch_c8<3, false, true, true>(a, nb, st);
    }
    if (d->J == 3) return d->xmask_c8 ? launch_c8<3, true, true>(a, nb, st) : launch_c8<3, false, true>(a, nb, st);
    return d->xmask_c8 ? launch_c8<1, true, true>(a, nb, st) : launch_c8<1, false, true>(a, nb, st);
  }
  g_dv3_last_wgrad = 5000 + d->J;
  if (d->J == 3) return d->xmask_c8 ? launch_c8<3, true, false>(a, nb, st) : launch_c8<3, false, false>(a, nb, st);
  return d->xmask_c8 ? launch_c8<1, true, false>(a, nb, st) : launch_c8<1, false, false>(a, nb, st);
}
