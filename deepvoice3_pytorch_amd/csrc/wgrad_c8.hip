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

}  // namespace

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
