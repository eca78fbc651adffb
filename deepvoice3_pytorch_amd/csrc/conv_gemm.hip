// dv3_conv_gemm_f32: im2col-free dilated 1-D convolution as a tap-GEMM on the gfx950 fp32
// matrix cores, with the Conv1dGLU / HighwayConv1d tail fused into the epilogue.
//
// Reference semantics: deepvoice3_pytorch/modules.py:145-164 (Conv1dGLU._forward),
// :205-226 (HighwayConv1d._forward), conv.py:7-16 (nn.Conv1d), plus the 1x1 convs /
// Linear / ConvTranspose1d / torch.bmm call sites listed in include/dv3hip.h.
//
// Design (see DESIGN.md "conv_gemm"):
//   * one workgroup = WM x WN waves; each wave owns a 64(M) x NI*32(N) accumulator block:
//     two 32-row M sub-tiles (the `a` rows and their gate rows, so GLU is lane-local) times
//     NI 32-column time sub-tiles, v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
//   * K loop over input-channel chunks of BKC; per chunk the packed weight tile
//     [J][BKC][BM] and ONE haloed input tile [BKC][BN+(J-1)*dil] are staged in LDS; the J
//     taps read the same input tile at shifted columns (no im2col, input read once per chunk).
//   * dropout is applied while staging x (keep-bits from dv3_dropout_bits), bias / speaker
//     bias / gate / residual / sqrt(.5) / activation in the epilogue.
//   * fragment reads are ds_read_b32 with lanes along the contiguous axis: conflict-free,
//     and at the fp32 MFMA rate (64 cycles per instruction) LDS bandwidth is <15% used.
#include "conv_common.h"
#include <string.h>

namespace {

template <int WM, int WN, int NI, int BKC>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_f32_kernel(const ConvArgs args) {
  constexpr int BM = WM * 64;   // rows staged per chunk (two halves of BMH)
  constexpr int BMH = WM * 32;  // rows per half
  constexpr int BN = WN * NI * 32;
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  const dv3_conv_desc& p = args.d;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int J = p.J, dil = p.dil;
  const int BNH = BN + (J - 1) * dil;
  float* As = smem;                 // [J][BKC][BM]
  float* Xs = smem + J * BKC * BM;  // [BKC][BNH]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  // ---- tile coordinates (XCD-aware: m fastest so blocks sharing an x tile share an L2) ----
  const int pid = dv3_xcd_remap(blockIdx.x, args.n_blocks);
  const int mt = pid % args.m_tiles;
  const int nt = (pid / args.m_tiles) % args.n_tiles;
  const int b = pid / (args.m_tiles * args.n_tiles);
  const int n0 = nt * BN;

  const bool gated = (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY);
  // column bases of the two row-halves inside a packed weight row, and their valid limits
  int h0b, h1b, lim0, lim1;
  if (gated) {
    h0b = mt * BMH;
    h1b = p.a_half + mt * BMH;
    lim0 = p.a_half;  // a_half = Cg rounded up to 4, zero padded
    lim1 = p.lda;
  } else {
    h0b = mt * BM;
    h1b = mt * BM + BMH;
    lim0 = lim1 = p.lda;
  }

  const float* __restrict__ Ag = p.a + (int64_t)b * p.a_bs;
  const float* __restrict__ Xg = p.x + (int64_t)b * p.x_bs;
  const uint32_t* __restrict__ xmask = p.xmask;
  const float dscale = p.drop_scale;

  f32x16 acc[2][NI];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][ni][r] = 0.f;

  const int Cin = p.Cin, Tin = p.Tin, lda = p.lda;
  const int tg0 = n0 - p.padL;  // global time of staged column 0

  for (int c0 = 0; c0 < Cin; c0 += BKC) {
    // ---------------- stage packed weights: J*BKC rows x BM floats ----------------
    if (args.a_scalar) {
      for (int idx = tid; idx < J * BKC * BM; idx += NT) {
        const int col = idx % BM;
        const int row = idx / BM;
        const int j = row / BKC, kc = row % BKC;
        const int c = c0 + kc;
        const bool hi = col >= BMH;
        const int gcol = (hi ? h1b : h0b) + (col - (hi ? BMH : 0));
        const int lim = hi ? lim1 : lim0;
        float v = 0.f;
        if (c < Cin && gcol < lim) v = Ag[((int64_t)j * Cin + c) * lda + gcol];
        As[row * BM + col] = v;
      }
    } else
    for (int idx = tid; idx < J * BKC * (BM / 4); idx += NT) {
      const int c4 = idx % (BM / 4);
      const int row = idx / (BM / 4);  // j*BKC + kc
      const int j = row / BKC, kc = row % BKC;
      const int c = c0 + kc;
      const int col = c4 * 4;
      const bool hi = col >= BMH;
      const int gcol = (hi ? h1b : h0b) + (col - (hi ? BMH : 0));
      const int lim = hi ? lim1 : lim0;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < Cin && gcol < lim)
        v = *reinterpret_cast<const f32x4*>(Ag + ((int64_t)j * Cin + c) * lda + gcol);
      *reinterpret_cast<f32x4*>(As + row * BM + col) = v;
    }
    // ---------------- stage haloed input tile: BKC rows x BNH floats ----------------
    for (int kc = wave; kc < BKC; kc += NW) {
      const int c = c0 + kc;
      const float* __restrict__ xrow = Xg + (int64_t)c * p.x_rs;
      const int64_t mrow = ((int64_t)b * Cin + c) * p.xmask_rs;
      for (int q = lane; q < BNH; q += 64) {
        const int tg = tg0 + q;
        float v = 0.f;
        if (c < Cin && tg >= 0 && tg < Tin) {
          v = xrow[tg];
          if (xmask) {
            const uint32_t w = xmask[mrow + (tg >> 5)];
            v = ((w >> (tg & 31)) & 1u) ? v * dscale : 0.f;
          }
        }
        Xs[kc * BNH + q] = v;
      }
    }
    __syncthreads();

    // ---------------- MFMA over taps x channel pairs ----------------
    const float* a_base = As + wm * 32 + l31 + lhi * BM;
    const float* x_base = Xs + wn * (NI * 32) + l31 + lhi * BNH;
    for (int j = 0; j < J; ++j) {
      const float* aj = a_base + j * (BKC * BM);
      const float* xj = x_base + j * dil;
#pragma unroll
      for (int kk = 0; kk < BKC; kk += 2) {
        const float a0 = aj[kk * BM];
        const float a1 = aj[kk * BM + BMH];
        float bv[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni] = xj[kk * BNH + ni * 32];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[ni], acc[0][ni], 0, 0, 0);
          acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[ni], acc[1][ni], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  int bcol[NI], tcol[NI];
  bool okc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    bcol[ni] = b;
    tcol[ni] = n0 + wn * (NI * 32) + ni * 32 + l31;
    okc[ni] = tcol[ni] < p.Tout;
  }
  conv_epilogue<BM, BMH, NI>(p, acc, gated, mt, wm * 32, lhi, bcol, tcol, okc);
}

// ------------------------------------------------------------------------------------------
// Streaming variant: no LDS, no barriers.  At the fp32 MFMA rate (64 cycles per 32x32x2
// instruction) a wave needs only ~4 B/clk of operands, so each wave fetches its own A/B
// fragments coalesced from L1/L2 (128-B row segments per half-wave) through a PF-deep software
// prefetch ring and never synchronises with its neighbours.  rocprofv3 on the LDS-staged kernel
// showed the matrix pipe 52% busy with waves parked 34% of their life at s_waitcnt/s_barrier
// (profiles/r01_conv_gemm_f32_pmc.md); this removes every such wait.
// K order: channel pairs outer, taps inner, so the J shifted re-reads of an x row hit L1.
// ------------------------------------------------------------------------------------------
template <int WM, int WN, int NI, bool MASK>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_f32_stream_kernel(const ConvArgs args) {
  constexpr int BM = WM * 64, BMH = WM * 32, BN = WN * NI * 32;
  constexpr int PF = 4;  // k-steps in flight
  const dv3_conv_desc& p = args.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int pid = dv3_xcd_remap(blockIdx.x, args.n_blocks);
  const int mt = pid % args.m_tiles;
  const int nt = (pid / args.m_tiles) % args.n_tiles;
  const int b = pid / (args.m_tiles * args.n_tiles);
  const int n0 = nt * BN;

  const bool gated = (p.mode == DV3_EPI_GLU || p.mode == DV3_EPI_HIGHWAY);
  int h0b, h1b, lim0, lim1;
  if (gated) {
    h0b = mt * BMH; h1b = p.a_half + mt * BMH; lim0 = p.a_half; lim1 = p.lda;
  } else {
    h0b = mt * BM; h1b = mt * BM + BMH; lim0 = lim1 = p.lda;
  }
  const int J = p.J, dil = p.dil, Cin = p.Cin, Tin = p.Tin, lda = p.lda;
  const int colA0 = h0b + wm * 32 + l31, colA1 = h1b + wm * 32 + l31;
  const bool okA0 = colA0 < lim0, okA1 = colA1 < lim1;
  const float* __restrict__ Xg = p.x + (int64_t)b * p.x_bs;
  const uint32_t* __restrict__ xmask = p.xmask;
  const float dscale = p.drop_scale;
  int tgb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) tgb[ni] = n0 + wn * (NI * 32) + ni * 32 + l31 - p.padL;

  f32x16 acc[2][NI];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][ni][r] = 0.f;

  const int nks = ((Cin + 1) >> 1) * J;
  float ra0[PF], ra1[PF], rb[PF][NI];
  int lc = lhi, lj = 0;  // (channel, tap) of the next k-step to LOAD (this lane's half)

  // Loads are UNCONDITIONAL on clamped (always valid) addresses; the ring holds the RAW values
  // and the validity predicate is recomputed and applied (bitwise AND with an optimiser-opaque
  // mask) right before the MFMA that consumes them.  A predicated load, or a select on the loaded
  // value at load time, makes hipcc branch around the load and drain vmcnt(0) -- or wait for the
  // data a few instructions after issuing it -- which serialises the prefetch ring
  // (cdna_hip_programming.md, ".s-level traps" (c)).
  const int colA0c = okA0 ? colA0 : 0, colA1c = okA1 ? colA1 : 0;
  const float* __restrict__ Ab = p.a + (int64_t)b * p.a_bs;
  auto load_step = [&](float& a0, float& a1, float (&bv)[NI]) {
    const int cc = min(lc, Cin - 1);
    const int arow = (lj * Cin + cc) * lda;          // packed weights are < 2^31 elements
    a0 = Ab[arow + colA0c];
    a1 = Ab[arow + colA1c];
    const float* __restrict__ xrow = Xg + (int64_t)cc * p.x_rs;
    const int sh = lj * dil;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int tgc = min(max(tgb[ni] + sh, 0), Tin - 1);
      float v = xrow[tgc];
      if (MASK) {
        const uint32_t w = xmask[((int64_t)b * Cin + cc) * p.xmask_rs + (tgc >> 5)];
        uint32_t mk = (uint32_t)(-(int)((w >> (tgc & 31)) & 1u));
        v = __uint_as_float(__float_as_uint(v * dscale) & mk);
      }
      bv[ni] = v;
    }
    if (++lj == J) { lj = 0; lc += 2; }
  };
  int uc = lhi, uj = 0;  // (channel, tap) of the next k-step to USE
  auto use_step = [&](float ra0_, float ra1_, const float (&rbv)[NI]) {
    const bool live = uc < Cin;   // steps past nks have uc >= Cin as well
    uint32_t m0 = (uint32_t)(-(int)(live && okA0)), m1 = (uint32_t)(-(int)(live && okA1));
    asm volatile("" : "+v"(m0), "+v"(m1));
    const float a0 = __uint_as_float(__float_as_uint(ra0_) & m0);
    const float a1 = __uint_as_float(__float_as_uint(ra1_) & m1);
    const int sh = uj * dil;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int tg = tgb[ni] + sh;
      uint32_t mb = (uint32_t)(-(int)(live && tg >= 0 && tg < Tin));
      asm volatile("" : "+v"(mb));
      const float bvv = __uint_as_float(__float_as_uint(rbv[ni]) & mb);
      acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bvv, acc[0][ni], 0, 0, 0);
      acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bvv, acc[1][ni], 0, 0, 0);
    }
    if (++uj == J) { uj = 0; uc += 2; }
  };

#pragma unroll
  for (int s = 0; s < PF; ++s) load_step(ra0[s], ra1[s], rb[s]);

  for (int ks = 0; ks < nks; ks += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      use_step(ra0[s], ra1[s], rb[s]);
      load_step(ra0[s], ra1[s], rb[s]);   // refill this ring slot for step ks + s + PF
    }
  }
  int bcol[NI], tcol[NI];
  bool okc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    bcol[ni] = b;
    tcol[ni] = n0 + wn * (NI * 32) + ni * 32 + l31;
    okc[ni] = tcol[ni] < p.Tout;
  }
  conv_epilogue<BM, BMH, NI>(p, acc, gated, mt, wm * 32, lhi, bcol, tcol, okc);
}

template <int WM, int WN, int NI>
int launch_stream(const ConvArgs& a, hipStream_t st) {
  dim3 grid(a.n_blocks), block(WM * WN * 64);
  if (a.d.xmask) {
    hipLaunchKernelGGL((conv_gemm_f32_stream_kernel<WM, WN, NI, true>), grid, block, 0, st, a);
  } else {
    hipLaunchKernelGGL((conv_gemm_f32_stream_kernel<WM, WN, NI, false>), grid, block, 0, st, a);
  }
  return dv3_check_launch("conv_gemm_f32(stream)");
}

template <int WM, int WN, int NI>
int launch_cfg(const ConvArgs& a, int bkc, size_t lds, hipStream_t st) {
  dim3 grid(a.n_blocks), block(WM * WN * 64);
  if (bkc == 16) {
    hipLaunchKernelGGL((conv_gemm_f32_kernel<WM, WN, NI, 16>), grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL((conv_gemm_f32_kernel<WM, WN, NI, 8>), grid, block, lds, st, a);
  }
  return dv3_check_launch("conv_gemm_f32");
}

}  // namespace

int dv3_conv_gemm_bf16x3_dispatch(const dv3_conv_desc* d, hipStream_t st);  // conv_gemm_bf16x3.hip
int dv3_conv_planes_dispatch(const dv3_conv_desc* d, hipStream_t st);       // conv_planes.hip

static int conv_gemm_dispatch(const dv3_conv_desc* d, void* stream);
// Launch census (dv3_debug_set(40, 1) starts, (40, 0) stops, dv3_debug_get(40) = count, dv3_debug_read(40 / 41, ...) =
// the descriptors / the kernel variant that served each): every tap-GEMM descriptor of a step, so that a script can
// re-issue each launch of a REAL training step stand-alone and time it (scripts/r5_conv_census.py)
namespace {
constexpr int CENSUS_CAP = 2048;
dv3_conv_desc g_census_desc[CENSUS_CAP];
int g_census_variant[CENSUS_CAP];
int g_census_n = 0;
bool g_census_on = false;
}  // namespace
int dv3_conv_census_set(int on) {
  if (on) g_census_n = 0;
  g_census_on = on != 0;
  return DV3_OK;
}
int dv3_conv_census_count() { return g_census_n; }
int dv3_conv_census_read(int what, void* dst, int64_t bytes) {
  const int64_t have = what == 40 ? (int64_t)g_census_n * (int64_t)sizeof(dv3_conv_desc) : (int64_t)g_census_n * (int64_t)sizeof(int);
  DV3_REQUIRE(dst && bytes >= 0 && bytes <= have, "debug_read: the census holds %lld bytes", (long long)have);
  memcpy(dst, what == 40 ? (const void*)g_census_desc : (const void*)g_census_variant, (size_t)bytes);
  return DV3_OK;
}
extern "C" int dv3_conv_gemm_f32(const dv3_conv_desc* d, void* stream) {
  const int rc = conv_gemm_dispatch(d, stream);
  if (g_census_on && d && rc == DV3_OK && g_census_n < CENSUS_CAP) {
    g_census_desc[g_census_n] = *d;
    g_census_variant[g_census_n++] = g_dv3_last_conv;
  }
  return rc;
}
static int conv_gemm_dispatch(const dv3_conv_desc* d, void* stream) {
  DV3_REQUIRE(d && (d->x || d->x_planes) && (d->a || d->a_split) && d->y, "conv_gemm: null pointer");
  DV3_REQUIRE(d->B > 0 && d->Cin > 0 && d->Tin > 0 && d->M > 0 && d->Tout > 0, "conv_gemm: bad dims");
  DV3_REQUIRE(d->J >= 1 && d->J <= 16 && d->dil >= 1, "conv_gemm: bad taps J=%d dil=%d", d->J, d->dil);
  const bool a_scalar = (d->lda & 3) || (d->a_half & 3) || (d->a_bs & 3) || ((uintptr_t)d->a & 15);
  const bool gated = d->mode == DV3_EPI_GLU || d->mode == DV3_EPI_HIGHWAY;
  if (gated) {
    DV3_REQUIRE(d->M == 2 * d->Cg, "conv_gemm: gated mode needs M == 2*Cg");
    DV3_REQUIRE(d->a_half >= d->Cg && d->lda >= d->a_half + d->Cg, "conv_gemm: bad a_half/lda");
    DV3_REQUIRE(d->mode == DV3_EPI_GLU ? (!d->residual || d->r) : (d->r != nullptr),
                "conv_gemm: residual/highway input missing");
    DV3_REQUIRE(d->store_mode == DV3_STORE_BCT, "conv_gemm: gated mode stores BCT only");
  } else {
    DV3_REQUIRE(d->lda >= d->M, "conv_gemm: lda < M");
    DV3_REQUIRE(d->mode >= DV3_EPI_LINEAR && d->mode <= DV3_EPI_SOFTSIGN, "conv_gemm: bad mode");
    if (d->store_mode == DV3_STORE_INTERLEAVE2) DV3_REQUIRE((d->M & 1) == 0, "interleave2 needs even M");
  }
  if (d->xmask) DV3_REQUIRE(d->xmask_rs * 32 >= d->Tin, "conv_gemm: xmask row stride too small");
  if (d->ymask) DV3_REQUIRE(d->ymask_rs * 32 >= d->Tout, "conv_gemm: ymask row stride too small");

  DV3_REQUIRE(dv3_conv_fits32(d), "conv_gemm: a tensor exceeds the 4 GB the epilogue can address");
  DV3_REQUIRE(d->io_bf16 == 0 || (d->a_split && d->split_terms == 1 && (d->tile_hint == 0 || d->tile_hint > 20)),
              "conv_gemm: bf16 activation storage is served by the single-term bf16 kernels only");
  if (d->io_bf16 & DV3_IO_OUT_C8) {   // channel-blocked bf16 outputs / residuals (include/dv3hip.h)
    const int Cout = gated ? d->Cg : d->M;
    DV3_REQUIRE(d->store_mode == DV3_STORE_BCT && (!gated || (d->Cg & 7) == 0),
                "conv_gemm: c8 storage needs the plain store and, for gated layers, Cg % 8 == 0");
    DV3_REQUIRE(!(d->io_bf16 & (DV3_IO_IN_BF16 | DV3_IO_OUT_BF16 | DV3_IO_AB_BF16)) && !d->ymask,
                "conv_gemm: c8 storage excludes the BCT bf16 flags and the bit-mask form of ymask");
    DV3_REQUIRE((((uintptr_t)d->y | (uintptr_t)d->r | (uintptr_t)d->r2 | (uintptr_t)d->ab) & 15) == 0,
                "conv_gemm: c8 tensors must be 16-byte aligned");
    const int64_t c8y = (Cout + 31) / 32 * 4, c8ab = (d->M + 31) / 32 * 4;
    DV3_REQUIRE((int64_t)d->B * c8y * d->Tout * 16 < (1ll << 32) && (!d->ab || (int64_t)d->B * c8ab * d->Tout * 16 < (1ll << 32)),
                "conv_gemm: a c8 tensor exceeds the 4 GB the epilogue can address");
  } else {
    DV3_REQUIRE(!d->ymask_c8, "conv_gemm: ymask_c8 belongs to a c8 DGRAD output");
  }
  DV3_REQUIRE(!d->ymask_c8 || d->mode == DV3_EPI_DGRAD, "conv_gemm: ymask_c8 is a DGRAD input");
  // keep-bytes mask a c8 input (x_planes, split_terms == 1), or accompany the keep-bits of an fp32 input (the
  // 256 x 256 split kernel stages the byte form, the others the bit form: both must describe the same decisions)
  DV3_REQUIRE(!d->xmask_c8 || d->mode != DV3_EPI_DGRAD, "conv_gemm: xmask_c8 masks the input of a forward layer");
  DV3_REQUIRE(!d->xmask_c8 || (d->x_planes && d->split_terms == 1) || (!d->x_planes && d->xmask && d->a_split),
              "conv_gemm: xmask_c8 needs a c8 input, or an fp32 input with its keep-bits and a split weight image");
  // round 6: the producer's gate backward in the input-gradient tail / pair-word input (include/dv3hip.h)
  if (d->pg) {
    DV3_REQUIRE(d->mode == DV3_EPI_DGRAD && d->store_mode == DV3_STORE_BCT && d->io_bf16 == 0 && !d->x_planes,
                "conv_gemm: pg rides on an fp32 (B, C, T) input-gradient launch");
    DV3_REQUIRE(d->a_split && d->split_terms != 1 && (d->tile_hint == 0 || d->tile_hint > 20),
                "conv_gemm: pg is served by the three-term split kernels");
    DV3_REQUIRE(d->pg_mode == DV3_EPI_GLU || d->pg_mode == DV3_EPI_HIGHWAY, "conv_gemm: pg_mode must be GLU or HIGHWAY");
    DV3_REQUIRE(d->dpg, "conv_gemm: pg without dpg");
    DV3_REQUIRE(d->pg_mode != DV3_EPI_HIGHWAY || (d->pg_x && d->dpg_res), "conv_gemm: a highway producer needs pg_x and dpg_res");
    DV3_REQUIRE((int64_t)d->B * 2 * d->M * d->Tout < (1ll << 30) &&
                (!d->pg_x || (int64_t)d->B * d->pg_x_bs + (int64_t)d->M * d->pg_x_rs < (1ll << 30)),
                "conv_gemm: the producer's pre-gate pair exceeds the 4 GB the epilogue can address");
  }
  if (d->x_pair)
    DV3_REQUIRE(d->a_split && (d->split_terms == 0 || d->split_terms == 3) && !d->xmask && !d->xmask_c8 && !d->x_planes &&
                d->io_bf16 == 0 && (d->tile_hint == 0 || d->tile_hint > 20),
                "conv_gemm: pair-word input is read by the bf16-pair split kernels, without dropout");
  // both operands pre-split: the persistent planes kernel.  No silent fallback: the planes carry the dropout
  // mask of the consuming layer, which the other kernels would have to be handed separately.
  if (d->x_planes) {
    DV3_REQUIRE(!d->xmask, "conv_gemm: x_planes already carry the dropout mask (xmask must be NULL)");
    const int rc = dv3_conv_planes_dispatch(d, (hipStream_t)stream);
    DV3_REQUIRE(rc != 1, "conv_gemm: shape not eligible for the planes kernel (needs a_split, Tin == Tout, "
                         "(J-1)*dil <= 64, x_c8p == round_up(Cin,32)/8)");
    return rc;
  }
  // split-bf16 operands given: the bf16x3 kernel (tile_hint 0 = auto, 21..26 = forced tile);
  // hints 1..16 keep the exact fp32 kernels for A/B runs; ineligible shapes fall through
  if (d->a_split && (d->tile_hint == 0 || d->tile_hint > 20)) {
    const int rc = dv3_conv_gemm_bf16x3_dispatch(d, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  DV3_REQUIRE(!d->pg && !d->x_pair, "conv_gemm: shape not eligible for the split kernels, which alone run the fused gate "
                                     "backward / read pair words (Tin == Tout, (J-1)*dil <= 64, no per-batch operand)");
  DV3_REQUIRE(d->io_bf16 == 0, "conv_gemm: shape not eligible for the split kernels, which alone take bf16 activations "
                               "(Tin == Tout, (J-1)*dil <= 64, no per-batch operand)");
  DV3_REQUIRE(d->tile_hint <= 20, "conv_gemm: tile_hint %d needs split-bf16 operands", d->tile_hint);
  DV3_REQUIRE(d->a, "conv_gemm: shape not eligible for the split-bf16 kernel and no fp32 operand image given");
  const int rows_half = gated ? d->Cg : 0;
  // tile_hint: 0 auto (streaming kernel), 1..6 streaming kernel with that tile, 11..16 the
  // LDS-staged kernel with tile (hint-10) -- kept for A/B measurements
  const bool use_lds = d->tile_hint > 10;
  const int want_tile = use_lds ? d->tile_hint - 10 : d->tile_hint;
  const TileCfg* best = dv3_pick_tile(d, gated, want_tile);
  DV3_REQUIRE(best, "conv_gemm: unknown tile_hint %d", d->tile_hint);

  ConvArgs a;
  a.d = *d;
  a.a_scalar = a_scalar ? 1 : 0;
  a.kp = 0;
  const int BM = best->wm * 64, BMH = best->wm * 32, BN = best->wn * best->ni * 32;
  a.m_tiles = gated ? dv3_cdiv(rows_half, BMH) : dv3_cdiv(d->M, BM);
  a.n_tiles = dv3_cdiv(d->Tout, BN);
  const int64_t nb = (int64_t)a.m_tiles * a.n_tiles * d->B;
  DV3_REQUIRE(nb < (1ll << 31), "conv_gemm: grid too large");
  a.n_blocks = (int)nb;

  const int BNH = BN + (d->J - 1) * d->dil;
  int bkc = 16;
  size_t lds = (size_t)(d->J * bkc * BM + bkc * BNH) * 4;
  if (lds > 64 * 1024) {
    bkc = 8;
    lds = (size_t)(d->J * bkc * BM + bkc * BNH) * 4;
  }
  if (use_lds) DV3_REQUIRE(lds <= 64 * 1024, "conv_gemm: LDS tile %zu B too large (J=%d dil=%d)", lds, d->J, d->dil);

  hipStream_t st = (hipStream_t)stream;
  g_dv3_last_conv = (use_lds ? 2000 : 1000) + best->id * 10;
  if (!use_lds) {
    switch (best->id) {
      case 1: return launch_stream<2, 2, 2>(a, st);
      case 2: return launch_stream<2, 2, 1>(a, st);
      case 3: return launch_stream<4, 1, 1>(a, st);
      case 4: return launch_stream<2, 1, 1>(a, st);
      case 5: return launch_stream<1, 2, 2>(a, st);
      case 6: return launch_stream<1, 2, 1>(a, st);
    }
  }
  switch (best->id) {
    case 1: return launch_cfg<2, 2, 2>(a, bkc, lds, st);
    case 2: return launch_cfg<2, 2, 1>(a, bkc, lds, st);
    case 3: return launch_cfg<4, 1, 1>(a, bkc, lds, st);
    case 4: return launch_cfg<2, 1, 1>(a, bkc, lds, st);
    case 5: return launch_cfg<1, 2, 2>(a, bkc, lds, st);
    case 6: return launch_cfg<1, 2, 1>(a, bkc, lds, st);
  }
  dv3_set_error("conv_gemm: unreachable");
  return DV3_EINVAL;
}
