// Shared pieces of the tap-GEMM kernels (conv_gemm.hip: exact fp32 MFMA; conv_gemm_bf16x3.hip:
// split-bf16 MFMA): launch arguments, the fused epilogue and the tile table.
#pragma once
#include "common.h"

namespace {

struct ConvArgs {
  dv3_conv_desc d;
  int m_tiles, n_tiles, n_blocks;
  int a_scalar;  // packed operand not 16-byte aligned (per-batch A = an activation): scalar staging
  int kp;        // bf16x3: K extent of the split weight image (Cin rounded up to 32)
  int stagger;   // planes kernel: s_sleep(127) repeats before the second co-resident workgroup starts
};

template <typename T>
__device__ __forceinline__ T dv3_ld(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void dv3_st(void* base, uint32_t byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// Shared epilogue: acc[h][ni] is the 32x32 fp32 tile of row-half h, column sub-tile ni; row0 = the
// slice's first row inside its half of the block tile (wave row offset; + mi*32 for MI > 1).
// C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// The lane's output column of sub-tile ni is (batch bcol[ni], time tcol[ni]), live when okc[ni]
// (per column, so a tile may span several batch items: the bf16x3 kernel flattens (b,t)).
// All addressing is uniform base + 32-bit byte offset (the host checks every tensor < 4 GB):
// the straightforward 64-bit form made this fully unrolled tail ~10k instructions -- more than
// the instruction cache, i.e. a fixed ~40 us of fetch stalls per launch at the north-star shape.
// ABL: measurement variants (dv3_debug_set): 7 = no residual load, 8 = no store.
// bf16 storage (IOB instantiations = the single-term bf16 kernels): activations may be bf16 tensors; loads widen
// exactly, stores round to nearest even.  The flags are wave-uniform.
__device__ __forceinline__ float dv3_ld_act(const void* base, uint32_t byte_off, bool bf) {
  if (bf) return __uint_as_float((uint32_t)dv3_ld<uint16_t>(base, byte_off) << 16);
  return dv3_ld<float>(base, byte_off);
}
__device__ __forceinline__ void dv3_st_act(void* base, uint32_t byte_off, float v, bool bf) {
  if (bf) {
    const __bf16 h = (__bf16)v;
    *reinterpret_cast<__bf16*>(reinterpret_cast<char*>(base) + byte_off) = h;
  } else {
    dv3_st(base, byte_off, v);
  }
}

template <int BM, int BMH, int NI, int ABL = 0, bool IOB = false>
__device__ __forceinline__ void conv_epilogue(const dv3_conv_desc& p, f32x16 (&acc)[2][NI], bool gated,
                                              int mt, int row0, int lhi, const int (&bcol)[NI],
                                              const int (&tcol)[NI], const bool (&okc)[NI]) {
  const float dscale = p.drop_scale;
  const uint32_t Tout = (uint32_t)p.Tout, M = (uint32_t)p.M, Cg = (uint32_t)p.Cg;
  const float rs2 = 0.70710678118654752440f;
  const bool inb = IOB && (p.io_bf16 & DV3_IO_IN_BF16), outb = IOB && (p.io_bf16 & DV3_IO_OUT_BF16);
  const bool abb16 = outb || (IOB && (p.io_bf16 & DV3_IO_AB_BF16));
  const uint32_t isz = inb ? 2u : 4u, osz = outb ? 2u : 4u;      // element sizes of the r / r2 and y tensors
  const uint32_t absz = abb16 ? 2u : 4u;                          // ... and of the saved pre-gate pair
  const uint32_t y_rs = (uint32_t)p.y_rs * osz, r_rs = (uint32_t)p.r_rs * isz, r2_rs = (uint32_t)p.r2_rs * isz;
  const bool il2 = p.store_mode == DV3_STORE_INTERLEAVE2;
  uint32_t yb[NI], rb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const uint32_t b = (uint32_t)bcol[ni], t = (uint32_t)tcol[ni];
    yb[ni] = (b * (uint32_t)p.y_bs + (il2 ? 2u * t : t)) * osz;
    rb[ni] = (b * (uint32_t)p.r_bs + t) * isz;
  }
  // Loads are issued as one batch per 32x(NI*32) half (clamped to valid addresses so they need no
  // predicate) and only then consumed: a load -> use -> store chain per element would serialise
  // 32..64 HBM latencies per wave (measured: 60 us of a 190 us launch at the north-star shape).
  if (gated) {
    const bool glu = p.mode == DV3_EPI_GLU;
    const bool has_r = !glu || p.residual;
    const float oscale = (glu && p.residual) ? rs2 : 1.0f;
    const uint32_t spk_rs = (uint32_t)p.spk_rs * 4u;
    uint32_t abb[NI], sb[NI], rbc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const uint32_t b = (uint32_t)bcol[ni], t = (uint32_t)tcol[ni];
      abb[ni] = (b * M * Tout + t) * absz;
      sb[ni] = (b * (uint32_t)p.spk_bs + t * (uint32_t)p.spk_ts) * 4u;
      rbc[ni] = okc[ni] ? rb[ni] : 0u;
    }
    uint32_t chv[16];
    float xr[16][NI], ba[16], bg[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      chv[r] = (uint32_t)(mt * BMH + row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi);
      const uint32_t chc = chv[r] < Cg ? chv[r] : Cg - 1;
      ba[r] = bg[r] = 0.f;
      if (p.bias) {
        ba[r] = p.bias[chc];
        bg[r] = p.bias[Cg + chc];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        xr[r][ni] = (has_r && ABL != 7) ? dv3_ld_act(p.r, rbc[ni] + chc * r_rs, inb) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t ch = chv[r];
      if (ch >= Cg) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (!okc[ni]) continue;
        float a = acc[0][ni][r] + ba[r];
        const float g = acc[1][ni][r] + bg[r];
        if (p.spk) a += dv3_ld<float>(p.spk, sb[ni] + ch * spk_rs);
        if (p.ab) {
          const uint32_t o = abb[ni] + ch * Tout * absz;
          dv3_st_act(p.ab, o, a, abb16);
          dv3_st_act(p.ab, o + Cg * Tout * absz, g, abb16);
        }
        const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
        // GLU: (a*s [+ x]) * (sqrt(.5) | 1)      HIGHWAY: s*a + (1-s)*x
        const float y = glu ? (a * s + xr[r][ni]) * oscale : s * a + (1.0f - s) * xr[r][ni];
        if (ABL != 8 || y == 1.2345e30f) dv3_st_act(p.y, yb[ni] + ch * y_rs, y, outb);
      }
    }
    return;
  }
  if (p.mode == DV3_EPI_DGRAD) {
    const uint32_t ym_rs = (uint32_t)p.ymask_rs * 4u;
    const float rsc = p.r_scale != 0.f ? p.r_scale : 1.0f;
    uint32_t ymb[NI], rbc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      ymb[ni] = okc[ni] ? ((uint32_t)bcol[ni] * M * (uint32_t)p.ymask_rs + ((uint32_t)tcol[ni] >> 5)) * 4u : 0u;
      rbc[ni] = okc[ni] ? rb[ni] : 0u;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t mv[16], wv[16][NI];
      float rv[16][NI];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mv[r] = (uint32_t)(mt * BM + h * BMH + row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi);
        const uint32_t mc = mv[r] < M ? mv[r] : M - 1;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          wv[r][ni] = p.ymask ? dv3_ld<uint32_t>(p.ymask, ymb[ni] + mc * ym_rs) : 0xffffffffu;
          rv[r][ni] = p.r ? dv3_ld_act(p.r, rbc[ni] + mc * r_rs, inb) : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (mv[r] >= M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (!okc[ni]) continue;
          float v = acc[h][ni][r];
          if (p.ymask) v = ((wv[r][ni] >> (tcol[ni] & 31)) & 1u) ? v * dscale : 0.f;
          dv3_st_act(p.y, yb[ni] + mv[r] * y_rs, v + rsc * rv[r][ni], outb);
        }
      }
    }
    return;
  }
  // LINEAR / RELU / SIGMOID / SOFTSIGN (+ up to two fused residuals, + interleaved store)
  uint32_t r2b[NI], rbc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    r2b[ni] = okc[ni] ? ((uint32_t)bcol[ni] * (uint32_t)p.r2_bs + (uint32_t)tcol[ni]) * isz : 0u;
    rbc[ni] = okc[ni] ? rb[ni] : 0u;
  }
  const uint32_t Mo = M >> 1;
  const int mode = p.mode;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t mv[16];
    float rv[16][NI], r2v[16][NI], bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      mv[r] = (uint32_t)(mt * BM + h * BMH + row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi);
      const uint32_t mc = mv[r] < M ? mv[r] : M - 1;
      const uint32_t mo = (il2 && mc >= Mo) ? mc - Mo : mc;
      bv[r] = p.bias ? p.bias[mo] : 0.f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        rv[r][ni] = p.r ? dv3_ld_act(p.r, rbc[ni] + mc * r_rs, inb) : 0.f;
        r2v[r][ni] = p.r2 ? dv3_ld_act(p.r2, r2b[ni] + mc * r2_rs, inb) : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t m = mv[r];
      if (m >= M) continue;
      const uint32_t mo = (il2 && m >= Mo) ? m - Mo : m;   // ConvTranspose: row m -> channel m % Mo
      const uint32_t odd = (il2 && m >= Mo) ? 4u : 0u;      // ... and output column 2t + m / Mo
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (!okc[ni]) continue;
        float v = acc[h][ni][r] + bv[r];
        if (mode == DV3_EPI_RELU) v = fmaxf(v, 0.f);
        else if (mode == DV3_EPI_SIGMOID) v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        else if (mode == DV3_EPI_SOFTSIGN) v = v * __builtin_amdgcn_rcpf(1.0f + fabsf(v));
        if (p.r) v = (v + rv[r][ni]) * rs2;
        if (p.r2) v = (v + r2v[r][ni]) * rs2;
        dv3_st_act(p.y, yb[ni] + mo * y_rs + (odd ? osz : 0u), v, outb);
      }
    }
  }
}

// the epilogue addresses every tensor with 32-bit byte offsets
static inline bool dv3_conv_fits32(const dv3_conv_desc* d) {
  const int64_t lim = (1ll << 30);  // elements (4-byte)
  const int64_t B = d->B;
  const int64_t To = d->store_mode == DV3_STORE_INTERLEAVE2 ? 2ll * d->Tout : d->Tout;
  if (B * d->y_bs + To >= lim || (int64_t)d->M * d->y_rs >= lim) return false;
  if (d->r && (B * d->r_bs + d->Tout >= lim || (int64_t)d->M * d->r_rs >= lim)) return false;
  if (d->r2 && (B * d->r2_bs + d->Tout >= lim || (int64_t)d->M * d->r2_rs >= lim)) return false;
  if (d->ab && B * d->M * d->Tout >= lim) return false;
  if (d->spk && (B * d->spk_bs + (int64_t)d->Cg * d->spk_rs + (int64_t)d->Tout * d->spk_ts >= lim)) return false;
  if (d->ymask && B * d->M * d->ymask_rs >= lim) return false;
  return true;
}

struct TileCfg {
  int id, wm, wn, ni, mi;   // mi: 32-row sub-tiles per half per wave (bf16x3 kernel; 1 elsewhere)
};
// id is what dv3_conv_desc.tile_hint selects.
static const TileCfg kCfgs[] = {
    {1, 2, 2, 2, 1},  // 128 x 128
    {2, 2, 2, 1, 1},  // 128 x 64
    {3, 4, 1, 1, 1},  // 256 x 32
    {4, 2, 1, 1, 1},  // 128 x 32
    {5, 1, 2, 2, 1},  // 64 x 128
    {6, 1, 2, 1, 1},  // 64 x 64
    {7, 2, 2, 2, 2},  // 256 x 128, 128 x 64 per wave (bf16x3 kernel only)
    {8, 4, 2, 2, 1},  // 256 x 128, 8 waves of 64 x 64 (bf16x3 kernel only)
    {9, 2, 4, 2, 1},  // 128 x 256, 8 waves of 64 x 64 (bf16x3 kernel only)
};

// pick a tile config: minimise padded work with a mild small-tile penalty
static inline const TileCfg* dv3_pick_tile(const dv3_conv_desc* d, bool gated, int want_tile) {
  const int rows_half = gated ? d->Cg : 0;
  const TileCfg* best = nullptr;
  double best_cost = 0;
  for (const TileCfg& c : kCfgs) {
    if (want_tile && c.id != want_tile) continue;
    const int BM = c.wm * 64, BMH = c.wm * 32, BN = c.wn * c.ni * 32;
    const int mt = gated ? dv3_cdiv(rows_half, BMH) : dv3_cdiv(d->M, BM);
    const int ntl = dv3_cdiv(d->Tout, BN);
    double work = (double)mt * BM * (double)ntl * BN;
    double pen = 1.0;
    if (BN == 64) pen *= 1.04;
    if (BN == 32) pen *= 1.10;
    if (BM == 64) pen *= 1.06;
    // too few blocks to fill 256 CUs: prefer finer tiles
    const double blocks = (double)mt * ntl * d->B;
    if (blocks < 512) pen *= 1.0 + 0.25 * (512 - blocks) / 512;
    const double cost = work * pen;
    if (!best || cost < best_cost) {
      best = &c;
      best_cost = cost;
    }
  }
  return best;
}

}  // namespace
