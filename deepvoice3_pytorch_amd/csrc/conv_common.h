// Shared pieces of the tap-GEMM kernels (conv_gemm.hip: exact fp32 MFMA; conv_gemm_bf16x3.hip:
// split-bf16 MFMA): launch arguments, the fused epilogue and the tile table.
#pragma once
#include "common.h"
#include <type_traits>

namespace {

struct ConvArgs {
  dv3_conv_desc d;
  int m_tiles, n_tiles, n_blocks;
  int a_scalar;  // packed operand not 16-byte aligned (per-batch A = an activation): scalar staging
  int kp;        // bf16x3: K extent of the split weight image (Cin rounded up to 32)
  int stagger;   // planes kernel: s_sleep(127) repeats before the second co-resident workgroup starts; conv_c8pp (NW = 4): s_sleep(1) repeats
  int stagger_mask = 1;   // conv_c8pp (NW = 4): blocks with (blockIdx & mask) != 0 are the delayed ones
  int wide = 1;                    // 16-byte input-gradient epilogue through LDS (0 = off)
  uint32_t* range_ctr = nullptr;   // f16x3: sticky fp16-range event counter (common.h), NULL = do not count
  int prio = 0;  // ping-pong tap-GEMM: wave priority scheme (dv3_debug_set(14, v); 0 = none)
  int fast_tail = 0;   // 256 x 256 kernel: interior sub-tiles of a gated launch take conv_epilogue_glu_interior (dv3_debug_set(50, v))
  int ks = 0;    // 128 x 64 split tile: 2 = the k-split form (two wave groups per workgroup, halves of the chunk range)
  int dp = 0;    // 128 x 64 split tile: deep-prefetch form with this compile-time tap count (1 or 3; 0 = the in-phase loop)
  // stream-K form of the 256 x 256 kernels: n_blocks = workgroups (one per CU), sk_units = tiles x chunks
  int sk_units = 0, sk_base = 0, sk_rem = 0, sk_shift = 0, sk_mshift = 0, sk_abl = 0;   // units; per-workgroup share and remainder (groups with the extra tile); log2(chunks per tile)
  int sk_base2 = 0, sk_rem2 = 0, sk_tg = 0, sk_tr = 0, sk_qshift = 0;                   // ... of the groups without it; tiles per XCD group, groups with one more; log2(workgroups per group)
  float* sk_ws = nullptr;      // [n_blocks][128 accumulator registers][512 threads]
  int* sk_flags = nullptr;     // [n_blocks], zero between launches
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

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the PRODUCER's gate backward inside the input-gradient tail (include/dv3hip.h: dv3_conv_desc.pg ...).
// The value an input-gradient launch writes, y = dL/d(out of the layer that produced this layer's input), is all the
// gate backward of that producer needs besides its saved pre-gate pair: the tail computes the pre-gate gradient on the
// tile it holds (dv3_gate_deriv, common.h: the function the stand-alone kernel runs), writes it as fp32 or as the PAIR
// WORDS the two gradient GEMMs of the producer stage without conversion, and leaves the bias partial sums per 32-column
// block [2M][n_part].  y is written as always.
// ---------------------------------------------------------------------------------------------------------------
// N values per lane, L lanes (consecutive lane indices, j = index inside the group): halving butterfly -- after log2(L)
// steps lane j holds, in v[0 .. N/L), the group's sums of the values j * (N/L) + [0, N/L).  N - N/L exchanges instead of
// N * log2(L); the order of the additions is a function of (N, L) only.
template <int N, int D>
__device__ __forceinline__ void dv3_reduce_scatter_step(float* v, int j) {
  if constexpr (D >= 1) {
    constexpr int n2 = N / 2;
    const bool up = (j & D) != 0;
#pragma unroll
    for (int i = 0; i < n2; ++i) {
      const float keep = up ? v[i + n2] : v[i];
      const float send = up ? v[i] : v[i + n2];
      v[i] = keep + __shfl_xor(send, D, 64);
    }
    dv3_reduce_scatter_step<n2, D / 2>(v, j);
  }
}
template <int N, int L>
__device__ __forceinline__ void dv3_reduce_scatter(float (&v)[N], int j) {
  static_assert(N % L == 0 && (L & (L - 1)) == 0, "reduce_scatter: N a multiple of L, L a power of two");
  dv3_reduce_scatter_step<N, L / 2>(v, j);
}
__device__ __forceinline__ void dv3_st_u32(void* base, uint32_t byte_off, uint32_t v) {
  *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// accumulator-order form (one frame per lane and column sub-tile): any Tout.  nblk0 = index of the wave's first
// 32-column block of the flat (b, t) axis.  The kernels compile these tails into SEPARATE instantiations (template
// parameter FG): as a run-time branch of the shared epilogue they cost every instantiation its spill-free tail.
template <int BM, int BMH, int NI>
__device__ __forceinline__ void conv_epilogue_dgrad_gate(const dv3_conv_desc& p, f32x16 (&acc)[2][NI], int mt, int row0,
                                                         int lhi, int l31, const int (&bcol)[NI], const int (&tcol)[NI],
                                                         const bool (&okc)[NI], int nblk0) {
  const float dscale = p.drop_scale;
  const uint32_t Tout = (uint32_t)p.Tout, M = (uint32_t)p.M;
  const uint32_t y_rs = (uint32_t)p.y_rs * 4u, r_rs = (uint32_t)p.r_rs * 4u, ym_rs = (uint32_t)p.ymask_rs * 4u;
  const uint32_t px_rs = (uint32_t)p.pg_x_rs * 4u, g_rs = Tout * 4u;
  const float rsc = p.r_scale != 0.f ? p.r_scale : 1.0f;
  const bool glu = p.pg_mode == DV3_EPI_GLU;
  const float k = (glu && p.pg_residual) ? 0.70710678118654752440f : 1.0f;
  const bool pair = p.pg_pair != 0;
  const int n_part = (int)(((int64_t)p.B * p.Tout + 31) >> 5);
  uint32_t yb[NI], rbc[NI], ymb[NI], pgb[NI], pxb[NI], rsb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const uint32_t b = okc[ni] ? (uint32_t)bcol[ni] : 0u, t = okc[ni] ? (uint32_t)tcol[ni] : 0u;
    yb[ni] = (b * (uint32_t)p.y_bs + t) * 4u;
    rbc[ni] = (b * (uint32_t)p.r_bs + t) * 4u;
    ymb[ni] = (b * M * (uint32_t)p.ymask_rs + (t >> 5)) * 4u;
    pgb[ni] = (b * 2u * M * Tout + t) * 4u;           // the pre-gate pair and its gradient: [B][2M][Tout]
    pxb[ni] = (b * (uint32_t)p.pg_x_bs + t) * 4u;
    rsb[ni] = (b * M * Tout + t) * 4u;                // dpg_res: [B][M][Tout]
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t rowb = (uint32_t)(mt * BM + h * BMH + row0 + 4 * lhi);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float v[32];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {       // four rows per batch of loads
        uint32_t wv[4];
        float rv[4], pa[4], pgt[4], px[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t m = rowb + (uint32_t)(e + 8 * rg);
          const uint32_t mc = m < M ? m : M - 1;
          wv[e] = p.ymask ? dv3_ld<uint32_t>(p.ymask, ymb[ni] + mc * ym_rs) : 0xffffffffu;
          rv[e] = p.r ? dv3_ld<float>(p.r, rbc[ni] + mc * r_rs) : 0.f;
          pa[e] = dv3_ld<float>(p.pg, pgb[ni] + mc * g_rs);
          pgt[e] = dv3_ld<float>(p.pg, pgb[ni] + (M + mc) * g_rs);
          px[e] = glu ? 0.f : dv3_ld<float>(p.pg_x, pxb[ni] + mc * px_rs);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rg + e;
          const uint32_t m = rowb + (uint32_t)(e + 8 * rg);
          float o = acc[h][ni][r];
          if (p.ymask) o = ((wv[e] >> (tcol[ni] & 31)) & 1u) ? o * dscale : 0.f;
          o = o + rsc * rv[e];
          float va = 0.f, vg = 0.f, vr = 0.f;
          if (okc[ni] && m < M) {
            dv3_st(p.y, yb[ni] + m * y_rs, o);
            dv3_gate_deriv(o * k, pa[e], pgt[e], px[e], glu, va, vg, vr);
            const uint32_t oa = pgb[ni] + m * g_rs, og = pgb[ni] + (M + m) * g_rs;
            if (pair) {
              dv3_st_u32(p.dpg, oa, dv3_pair_word(va));
              dv3_st_u32(p.dpg, og, dv3_pair_word(vg));
            } else {
              dv3_st(p.dpg, oa, va);
              dv3_st(p.dpg, og, vg);
            }
            if (!glu && p.dpg_res) dv3_st(p.dpg_res, rsb[ni] + m * g_rs, vr);
          }
          v[r] = va;
          v[16 + r] = vg;
        }
      }
      // lane l31 ends with the block sum of value l31: the `a` row (l31 & 15) of this half for l31 < 16, else its gate row
      dv3_reduce_scatter<32, 32>(v, l31);
      const int r = l31 & 15;
      const uint32_t m = rowb + (uint32_t)((r & 3) + 8 * (r >> 2));
      const int blk = nblk0 + ni;
      if (p.pg_part && m < M && blk < n_part)
        p.pg_part[(size_t)((l31 >> 4) ? M + m : m) * (size_t)n_part + (size_t)blk] = v[0];
    }
  }
}

// The gate's output (modules.py:162-164 GLU: (a * sigmoid(g) [+ x]) * sqrt(.5 | 1); :224-226 highway: s a + (1 - s) x) as ONE
// explicitly contracted expression shared by every fp32 tail of this header: left to the compiler, the guarded tail and
// the straight-line tail of round 6 contracted `a * s + x` differently (a one-ulp difference between the interior and
// the edge sub-tiles of one launch, and between kernels that are tested to agree bit for bit).
__device__ __forceinline__ float dv3_gate_out(float a, float s, float x, float oscale, bool glu) {
#pragma clang fp contract(off)
  if (glu) return __builtin_fmaf(a, s, x) * oscale;
  const float t = (1.0f - s) * x;
  return __builtin_fmaf(s, a, t);
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
        const float y = dv3_gate_out(a, s, xr[r][ni], oscale, glu);
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

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the Conv1dGLU / HighwayConv1d tail of an INTERIOR 32-row sub-tile -- every row below Cg, every column inside
// the tensor, fp32 tensors, no speaker bias: what all but the edge sub-tiles of a launch are.  conv_epilogue() above
// guards each of its 16 x NI elements with per-lane tests (row < Cg, column valid) and re-tests the launch-uniform
// switches (pre-gate save, speaker bias, GLU / highway) per element: ~950 branches in the 256 x 256 kernel's tails
// (13 of the tail's 26 us are neither loads nor stores).  Here the caller makes the tests ONCE per wave
// (they are wave-uniform) and this function is straight-line code, specialised on the two launch-uniform switches.
// Same operations per element in the same order as conv_epilogue(): bit-identical.
// ---------------------------------------------------------------------------------------------------------------
template <int BMH, int NI, bool AB, bool GLU>
__device__ __forceinline__ void conv_epilogue_glu_interior_t(const dv3_conv_desc& p, f32x16 (&acc)[2][NI], int mt, int row0, int lhi,
                                                             const int (&bcol)[NI], const int (&tcol)[NI]) {
  const uint32_t Tout = (uint32_t)p.Tout, M = (uint32_t)p.M, Cg = (uint32_t)p.Cg;
  const bool has_r = !GLU || p.residual;
  const float oscale = (GLU && p.residual) ? 0.70710678118654752440f : 1.0f;
  const uint32_t y_rs = (uint32_t)p.y_rs * 4u, r_rs = (uint32_t)p.r_rs * 4u, g_rs = Tout * 4u;
  uint32_t yb[NI], rb[NI], abb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const uint32_t b = (uint32_t)bcol[ni], t = (uint32_t)tcol[ni];
    yb[ni] = (b * (uint32_t)p.y_bs + t) * 4u;
    rb[ni] = (b * (uint32_t)p.r_bs + t) * 4u;
    abb[ni] = (b * M * Tout + t) * 4u;
  }
  const uint32_t ch0 = (uint32_t)(mt * BMH + row0 + 4 * lhi);
  float xr[16][NI], ba[16], bg[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t ch = ch0 + (uint32_t)((r & 3) + 8 * (r >> 2));
    ba[r] = bg[r] = 0.f;
    if (p.bias) {
      ba[r] = p.bias[ch];
      bg[r] = p.bias[Cg + ch];
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) xr[r][ni] = has_r ? dv3_ld<float>(p.r, rb[ni] + ch * r_rs) : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t ch = ch0 + (uint32_t)((r & 3) + 8 * (r >> 2));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const float a = acc[0][ni][r] + ba[r];
      const float g = acc[1][ni][r] + bg[r];
      if constexpr (AB) {
        const uint32_t o = abb[ni] + ch * g_rs;
        dv3_st(p.ab, o, a);
        dv3_st(p.ab, o + Cg * g_rs, g);
      }
      const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
      const float y = dv3_gate_out(a, s, xr[r][ni], oscale, GLU);
      dv3_st(p.y, yb[ni] + ch * y_rs, y);
    }
  }
}
template <int BMH, int NI>
__device__ __forceinline__ void conv_epilogue_glu_interior(const dv3_conv_desc& p, f32x16 (&acc)[2][NI], int mt, int row0, int lhi,
                                                           const int (&bcol)[NI], const int (&tcol)[NI]) {
  const bool glu = p.mode == DV3_EPI_GLU;
  if (p.ab) {
    if (glu) conv_epilogue_glu_interior_t<BMH, NI, true, true>(p, acc, mt, row0, lhi, bcol, tcol);
    else conv_epilogue_glu_interior_t<BMH, NI, true, false>(p, acc, mt, row0, lhi, bcol, tcol);
  } else {
    if (glu) conv_epilogue_glu_interior_t<BMH, NI, false, true>(p, acc, mt, row0, lhi, bcol, tcol);
    else conv_epilogue_glu_interior_t<BMH, NI, false, false>(p, acc, mt, row0, lhi, bcol, tcol);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wide form of the fp32 (B, C, T) epilogue of the INPUT GRADIENT (keep-mask, dropout scale, skip-path addend).  The MFMA
// accumulator gives a lane ONE column (frame) and 16 rows, so the narrow epilogue above moves every element with its
// own 4-byte access and loads one keep-mask word per element.  Here each 32 x 64 block is transposed through LDS (free
// once the main loop has left: every wave owns DV3_WIDE_LDS bytes): values go in with ds_write_b32 in accumulator
// order and come back with ds_read_b128 as FOUR CONSECUTIVE FRAMES of one row per lane, so the addend load, the
// keep-mask word (one per four outputs) and the store are 16-byte accesses: 4 x fewer global instructions.  Needs
// T % 4 == 0 (a group of four frames must not straddle two batch items of the flattened (b, t) axis) and 16-byte
// aligned tensors.  Same operations per element as the narrow form.  Measured at the north-star shape, same run:
// DGRAD 145.7 -> 136.9 us (256 x 256 kernel), 161.7 -> 153.0 us (128 x 256 kernel).  The Conv1dGLU forward tail was
// built the same way and measured SLOWER (eval 155 -> 166 us, with the pre-gate save 173 -> 211 us): that tail is one
// chip-wide HBM burst (all 256 workgroups reach it together: 134 MB in ~25 us = 5.4 TB/s), not an instruction-issue
// problem, and the extra LDS round trips only lengthen it (profiles/r03_pp2_check_ablations.txt); removed.
// ---------------------------------------------------------------------------------------------------------------
constexpr int DV3_WIDE_LD = 68;                                   // floats per staged row (64 + 4: rows land on shifted banks)
constexpr int DV3_WIDE_LDS = 32 * DV3_WIDE_LD * 4;                // bytes per wave

__device__ __forceinline__ bool dv3_wide_epilogue_ok(const dv3_conv_desc& p, int enable) {
  if (enable == 0 || p.mode != DV3_EPI_DGRAD) return false;
  if (p.store_mode != DV3_STORE_BCT || (p.Tout & 3) || (p.io_bf16 != 0)) return false;
  auto al = [](const void* q, int64_t rs, int64_t bs) { return q == nullptr || ((((uintptr_t)q) & 15) == 0 && (rs & 3) == 0 && (bs & 3) == 0); };
  if (!al(p.y, p.y_rs, p.y_bs) || !al(p.r, p.r_rs, p.r_bs)) return false;
  if (p.ymask && p.ymask_rs * 32 < p.Tout) return false;
  return true;
}
// ... and of the FG tails (the fused gate backward moves the producer's tensors 16 bytes at a time too)
__device__ __forceinline__ bool dv3_wide_gate_ok(const dv3_conv_desc& p, int enable) {
  if (!dv3_wide_epilogue_ok(p, enable)) return false;
  if ((((uintptr_t)p.pg | (uintptr_t)p.dpg | (uintptr_t)p.dpg_res) & 15) != 0) return false;
  if (p.pg_mode == DV3_EPI_HIGHWAY && ((((uintptr_t)p.pg_x) & 15) != 0 || (p.pg_x_rs & 3) != 0 || (p.pg_x_bs & 3) != 0)) return false;
  return true;
}

// The same block with the producer's gate backward (dv3_conv_desc.pg, see conv_epilogue_dgrad_gate): a lane holds four
// consecutive frames of eight rows, so the pre-gate pair is read and its gradient written 16 bytes at a time, and the
// 32-column bias partial sums are a sum over four frames + a butterfly over the eight lanes of a block.
template <int BM, int BMH>
__device__ __forceinline__ void conv_epilogue_wide_block_gate(const dv3_conv_desc& p, const f32x16 (&acc0)[2], int mt, int row0,
                                                              int half, int lane, int nw0, int Ntot, float* __restrict__ lds) {
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  const int l31 = lane & 31, lhi = lane >> 5;
  const uint32_t T = (uint32_t)p.Tout, M = (uint32_t)p.M;
  const int rr = lane >> 4, c4 = lane & 15;
  const int n4 = nw0 + c4 * 4;
  const bool okw = n4 < Ntot;
  const uint32_t b4 = okw ? (uint32_t)n4 / T : 0u, t4 = okw ? (uint32_t)n4 - b4 * T : 0u;
  const uint32_t rowb = (uint32_t)(mt * BM + half * BMH + row0);
  const float dscale = p.drop_scale;
  const float rsc = p.r_scale != 0.f ? p.r_scale : 1.0f;
  const bool glu = p.pg_mode == DV3_EPI_GLU;
  const float k = (glu && p.pg_residual) ? 0.70710678118654752440f : 1.0f;
  const bool pair = p.pg_pair != 0;
  const int n_part = (Ntot + 31) >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {       // accumulator order -> LDS [row][column]
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) lds[row * DV3_WIDE_LD + ni * 32 + l31] = acc0[ni][r];
  }
  float sv[16];
#pragma unroll
  for (int ig = 0; ig < 2; ++ig) {     // two batches of four rows: bounds the loads in flight beside the accumulators
    f32x4 rv[4], pa[4], pgt[4], px[4];
    uint32_t mw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t m = rowb + (uint32_t)(rr + 4 * (ig * 4 + q));
      const uint32_t mc = m < M ? m : M - 1;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      rv[q] = (p.r && okw) ? *reinterpret_cast<const f32x4*>(p.r + (size_t)b4 * p.r_bs + (size_t)mc * p.r_rs + t4) : z4;
      mw[q] = (p.ymask && okw) ? p.ymask[((size_t)b4 * M + mc) * (size_t)p.ymask_rs + (t4 >> 5)] : 0xffffffffu;
      pa[q] = okw ? *reinterpret_cast<const f32x4*>(p.pg + ((size_t)b4 * 2u * M + mc) * T + t4) : z4;
      pgt[q] = okw ? *reinterpret_cast<const f32x4*>(p.pg + ((size_t)b4 * 2u * M + M + mc) * T + t4) : z4;
      px[q] = (!glu && okw) ? *reinterpret_cast<const f32x4*>(p.pg_x + (size_t)b4 * p.pg_x_bs + (size_t)mc * p.pg_x_rs + t4) : z4;
    }
    if (ig == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ig * 4 + q;
      const uint32_t m = rowb + (uint32_t)(rr + 4 * i);
      const f32x4 w = *reinterpret_cast<const f32x4*>(lds + (rr + 4 * i) * DV3_WIDE_LD + c4 * 4);
      const bool live = okw && m < M;
      f32x4 o, da = {0.f, 0.f, 0.f, 0.f}, dg = {0.f, 0.f, 0.f, 0.f}, dr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = w[e];
        if (p.ymask) a = ((mw[q] >> ((t4 & 31) + e)) & 1u) ? a * dscale : 0.f;
        o[e] = a + rsc * rv[q][e];
        float va, vg, vr;
        dv3_gate_deriv(o[e] * k, pa[q][e], pgt[q][e], px[q][e], glu, va, vg, vr);
        da[e] = live ? va : 0.f;
        dg[e] = live ? vg : 0.f;
        dr[e] = vr;
      }
      if (live) {
        *reinterpret_cast<f32x4*>(p.y + (size_t)b4 * p.y_bs + (size_t)m * p.y_rs + t4) = o;
        float* const oa = p.dpg + ((size_t)b4 * 2u * M + m) * T + t4;
        float* const og = oa + (size_t)M * T;
        if (pair) {
          *reinterpret_cast<u32x4_*>(oa) = u32x4_{dv3_pair_word(da[0]), dv3_pair_word(da[1]), dv3_pair_word(da[2]), dv3_pair_word(da[3])};
          *reinterpret_cast<u32x4_*>(og) = u32x4_{dv3_pair_word(dg[0]), dv3_pair_word(dg[1]), dv3_pair_word(dg[2]), dv3_pair_word(dg[3])};
        } else {
          *reinterpret_cast<f32x4*>(oa) = da;
          *reinterpret_cast<f32x4*>(og) = dg;
        }
        if (!glu && p.dpg_res) *reinterpret_cast<f32x4*>(p.dpg_res + ((size_t)b4 * M + m) * T + t4) = dr;
      }
      sv[2 * i] = (da[0] + da[1]) + (da[2] + da[3]);
      sv[2 * i + 1] = (dg[0] + dg[1]) + (dg[2] + dg[3]);
    }
  }
  // eight lanes (c4 & 7) share a 32-column block: lane j ends with the `a` and gate sums of row rr + 4 j
  dv3_reduce_scatter<16, 8>(sv, c4 & 7);
  {
    const uint32_t m = rowb + (uint32_t)(rr + 4 * (c4 & 7));
    const int blk = (nw0 >> 5) + (c4 >> 3);
    if (p.pg_part && m < M && blk < n_part) {
      p.pg_part[(size_t)m * (size_t)n_part + (size_t)blk] = sv[0];
      p.pg_part[(size_t)(M + m) * (size_t)n_part + (size_t)blk] = sv[1];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done before the next block overwrites the rows
}

// One 32-row block of a wave (rows row0 + [0, 32) of half `half` of the tile), 64 columns starting at flat column nw0:
// y = keep ? acc * dscale : 0, + r_scale * r
template <int BM, int BMH>
__device__ __forceinline__ void conv_epilogue_wide_block(const dv3_conv_desc& p, const f32x16 (&acc0)[2], int mt, int row0,
                                                         int half, int lane, int nw0, int Ntot, float* __restrict__ lds) {
  const int l31 = lane & 31, lhi = lane >> 5;
  const uint32_t T = (uint32_t)p.Tout, M = (uint32_t)p.M;
  // wide-side coordinates of this lane: row rr + 4 * i of the block (i = 0..7), frames n4 .. n4 + 3
  const int rr = lane >> 4, c4 = lane & 15;
  const int n4 = nw0 + c4 * 4;
  const bool okw = n4 < Ntot;
  const uint32_t b4 = okw ? (uint32_t)n4 / T : 0u, t4 = okw ? (uint32_t)n4 - b4 * T : 0u;
  const uint32_t rowb = (uint32_t)(mt * BM + half * BMH + row0);
  const float dscale = p.drop_scale;
  const float rsc = p.r_scale != 0.f ? p.r_scale : 1.0f;
  f32x4 rv[8];
  uint32_t mw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {        // addend rows and keep words in the wide layout, issued before the LDS round trip
    const uint32_t m = rowb + (uint32_t)(rr + 4 * i);
    const uint32_t mc = m < M ? m : M - 1;
    rv[i] = (p.r && okw) ? *reinterpret_cast<const f32x4*>(p.r + (size_t)b4 * p.r_bs + (size_t)mc * p.r_rs + t4)
                         : f32x4{0.f, 0.f, 0.f, 0.f};
    mw[i] = (p.ymask && okw) ? p.ymask[((size_t)b4 * M + mc) * (size_t)p.ymask_rs + (t4 >> 5)] : 0xffffffffu;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {       // accumulator order -> LDS [row][column]
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) lds[row * DV3_WIDE_LD + ni * 32 + l31] = acc0[ni][r];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t m = rowb + (uint32_t)(rr + 4 * i);
    const f32x4 w = *reinterpret_cast<const f32x4*>(lds + (rr + 4 * i) * DV3_WIDE_LD + c4 * 4);
    if (!okw || m >= M) continue;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = w[e];
      if (p.ymask) a = ((mw[i] >> ((t4 & 31) + e)) & 1u) ? a * dscale : 0.f;
      o[e] = a + rsc * rv[i][e];
    }
    *reinterpret_cast<f32x4*>(p.y + (size_t)b4 * p.y_bs + (size_t)m * p.y_rs + t4) = o;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done before the next block overwrites the rows
}

// ---------------------------------------------------------------------------------------------------------------
// Wide form of the fp32 (B, C, T) Conv1dGLU / HighwayConv1d tail WITHOUT an LDS round trip (round 5).  The accumulator
// gives a lane one frame and 16 rows; a 4 x 4 transpose inside each QUAD of lanes (two butterfly steps of DPP quad
// permutes: lane j <-> j ^ 1 on register pairs (0,1) (2,3), then j <-> j ^ 2 on (0,2) (1,3); 8-16 vector ALU per block,
// no memory) leaves lane j of quad q with ONE row (8k + 4 lhi + j of its group k) and FOUR CONSECUTIVE FRAMES 4q .. 4q+3,
// so the residual load, the pre-gate save and the output store are 16-byte accesses (a wave instruction covers 8 rows
// x 128 contiguous bytes).  The pre-activations a = acc + bias and g = acc + bias are transposed, everything after
// them (save, sigmoid, fused multiply-add with the residual, scale) runs on the transposed values with the same
// operations per element as conv_epilogue(): bit-identical results.  Needs Tout % 4 == 0 (a quad's four frames stay
// inside one batch item of the flattened (b, t) axis), 16-byte aligned tensors, no speaker-bias tensor.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dv3_quad_swap1(float v) {   // value of lane j ^ 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float dv3_quad_swap2(float v) {   // value of lane j ^ 2
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
}
// in place: v[i] of lane j  <-  v[j] of lane i   (i, j = 0..3 inside the quad)
__device__ __forceinline__ void dv3_quad_transpose4(float (&v)[4], bool odd1, bool odd2) {
  // step 1: 2 x 2 blocks between lanes j, j ^ 1 on (v0, v1) and (v2, v3)
  {
    const float s0 = dv3_quad_swap1(odd1 ? v[0] : v[1]);
    const float s1 = dv3_quad_swap1(odd1 ? v[2] : v[3]);
    if (odd1) { v[0] = s0; v[2] = s1; } else { v[1] = s0; v[3] = s1; }
  }
  // step 2: between lanes j, j ^ 2 on (v0, v2) and (v1, v3)
  {
    const float s0 = dv3_quad_swap2(odd2 ? v[0] : v[2]);
    const float s1 = dv3_quad_swap2(odd2 ? v[1] : v[3]);
    if (odd2) { v[0] = s0; v[1] = s1; } else { v[2] = s0; v[3] = s1; }
  }
}

__device__ __forceinline__ bool dv3_wide_glu_ok(const dv3_conv_desc& p) {
  if (p.mode != DV3_EPI_GLU && p.mode != DV3_EPI_HIGHWAY) return false;
  if (p.store_mode != DV3_STORE_BCT || (p.Tout & 3) || p.io_bf16 != 0 || p.spk) return false;
  auto al = [](const void* q, int64_t rs, int64_t bs) { return q == nullptr || ((((uintptr_t)q) & 15) == 0 && (rs & 3) == 0 && (bs & 3) == 0); };
  if (!al(p.y, p.y_rs, p.y_bs) || !al(p.r, p.r_rs, p.r_bs)) return false;
  if (p.ab && ((((uintptr_t)p.ab) & 15) != 0)) return false;
  return true;
}

// One 32-row sub-tile of a wave: acc[0][ni] = `a` rows, acc[1][ni] = gate rows (as conv_epilogue); n0w = flat index of
// the wave's first column (the lane's quad q = (lane & 31) >> 2 owns frames n0w + ni * 32 + 4 q .. + 3).
// ABL (experiment build): 12 = no residual load, 13 = no stores
template <int BMH, int NI, int ABL = 0>
__device__ __forceinline__ void conv_epilogue_glu_wide(const dv3_conv_desc& p, f32x16 (&acc)[2][NI], int mt, int row0,
                                                       int lane, int n0w, int Ntot) {
  const int l31 = lane & 31, lhi = lane >> 5;
  const int j = l31 & 3, q = l31 >> 2;
  const bool odd1 = j & 1, odd2 = j & 2;
  const uint32_t T = (uint32_t)p.Tout, M = (uint32_t)p.M, Cg = (uint32_t)p.Cg;
  const bool glu = p.mode == DV3_EPI_GLU;
  const bool has_r = !glu || p.residual;
  const float oscale = (glu && p.residual) ? 0.70710678118654752440f : 1.0f;
  // wide-side coordinates: row 8 k + 4 lhi + j of the sub-tile (k = 0..3), frames n4 .. n4 + 3 of column sub-tile ni
  uint32_t yo[NI], ro[NI], ao[NI];
  bool okw[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n4 = n0w + ni * 32 + q * 4;
    okw[ni] = n4 < Ntot;
    const uint32_t b4 = okw[ni] ? (uint32_t)n4 / T : 0u, t4 = okw[ni] ? (uint32_t)n4 - b4 * T : 0u;
    yo[ni] = (b4 * (uint32_t)p.y_bs + t4) * 4u;
    ro[ni] = (b4 * (uint32_t)p.r_bs + t4) * 4u;
    ao[ni] = (b4 * M * T + t4) * 4u;
  }
  const uint32_t y_rs = (uint32_t)p.y_rs * 4u, r_rs = (uint32_t)p.r_rs * 4u;
  uint32_t chw[4];
  f32x4 xr[4][NI];
#pragma unroll
  for (int k = 0; k < 4; ++k) {        // residual rows in the wide layout, all requested before anything waits
    chw[k] = (uint32_t)(mt * BMH + row0 + 8 * k + 4 * lhi + j);
    const uint32_t chc = chw[k] < Cg ? chw[k] : Cg - 1;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      xr[k][ni] = (has_r && ABL != 12) ? dv3_ld<f32x4>(p.r, ro[ni] + chc * r_rs) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // bias of the rows this lane holds BEFORE the transpose (accumulator order: row 8 k + 4 lhi + i in register 4 k + i)
    float ba[4], bg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t chr = (uint32_t)(mt * BMH + row0 + 8 * k + 4 * lhi + i);
      const uint32_t chc = chr < Cg ? chr : Cg - 1;
      ba[i] = p.bias ? p.bias[chc] : 0.f;
      bg[i] = p.bias ? p.bias[Cg + chc] : 0.f;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float a[4], g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = acc[0][ni][4 * k + i] + ba[i];
        g[i] = acc[1][ni][4 * k + i] + bg[i];
      }
      dv3_quad_transpose4(a, odd1, odd2);
      dv3_quad_transpose4(g, odd1, odd2);
      const uint32_t ch = chw[k];
      if (ch >= Cg || !okw[ni]) continue;
      if (p.ab) {
        const uint32_t o = ao[ni] + ch * T * 4u;
        if (ABL != 13) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.ab) + o) = f32x4{a[0], a[1], a[2], a[3]};
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.ab) + o + Cg * T * 4u) = f32x4{g[0], g[1], g[2], g[3]};
        }
      }
      f32x4 y4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-g[e]));
        const float x = xr[k][ni][e];
        y4[e] = dv3_gate_out(a[e], s, x, oscale, glu);
      }
      if (ABL != 13 || y4[0] == 1.2345e30f) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.y) + yo[ni] + ch * y_rs) = y4;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Epilogue of the bf16-storage path: y (and the saved pre-gate pair, the residual inputs, the DGRAD addend and its
// keep-mask) live in the channel-blocked "c8" layout  bf16 [B][C8][T][8],  C8 = round_up(C,32)/8  (include/dv3hip.h).
// The 32x32 accumulator tile gives every lane four CONSECUTIVE channels (r&3) of four 8-channel groups (r>>2) at
// one frame, i.e. one 8-byte half of a 16-byte unit: 4 x NI eight-byte loads / stores per half tile instead of
// 16 x NI scalar ones, and a wave's store covers 32 whole units = 512 contiguous bytes.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dv3_unpack4(const uint2 u, float (&v)[4]) {
  v[0] = __uint_as_float(u.x << 16);
  v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16);
  v[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ uint2 dv3_pack4(const float (&v)[4]) {
  typedef __bf16 dv3_bf16x2 __attribute__((ext_vector_type(2)));
  typedef float dv3_f32x2 __attribute__((ext_vector_type(2)));
  const dv3_f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const dv3_bf16x2 ha = __builtin_convertvector(a, dv3_bf16x2), hb = __builtin_convertvector(b, dv3_bf16x2);
  uint2 o;
  o.x = __builtin_bit_cast(uint32_t, ha);
  o.y = __builtin_bit_cast(uint32_t, hb);
  return o;
}
__device__ __forceinline__ void dv3_st8(void* base, uint32_t byte_off, const uint2 v) {
  *reinterpret_cast<uint2*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// rl (gated forms, optional): the residual's c8 units of this lane's first column in LDS, [channel group of the tile's
// `a` rows][rl_cols columns] (conv_c8pp.hip, RL); column sub-tile ni is 32 units further.
template <int BM, int BMH, int NI>
__device__ __forceinline__ void conv_epilogue_c8(const dv3_conv_desc& p, f32x16 (&acc)[2][NI], bool gated,
                                                 int mt, int row0, int lhi, const int (&bcol)[NI],
                                                 const int (&tcol)[NI], const bool (&okc)[NI],
                                                 const unsigned char* rl = nullptr, int rl_cols = 0) {
  const uint32_t T = (uint32_t)p.Tout, M = (uint32_t)p.M, Cg = (uint32_t)p.Cg;
  const uint32_t Cout = gated ? Cg : M;
  const uint32_t c8y = (Cout + 31u) / 32u * 4u;        // 8-channel groups per batch item of y / r / r2
  const uint32_t gsz = T * 16u;                        // bytes between consecutive channel groups
  const uint32_t half = lhi ? 8u : 0u;
  const float rs2 = 0.70710678118654752440f;
  uint32_t ub[NI], ubc[NI];                            // byte offset of this lane's half of unit (b, group 0, t)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    ub[ni] = ((uint32_t)bcol[ni] * c8y * T + (uint32_t)tcol[ni]) * 16u + half;
    ubc[ni] = okc[ni] ? ub[ni] : half;                 // loads of dead columns read unit 0
  }
  if (gated) {
    const bool glu = p.mode == DV3_EPI_GLU;
    const bool has_r = !glu || p.residual;
    const float oscale = (glu && p.residual) ? rs2 : 1.0f;
    const uint32_t c8ab = (M + 31u) / 32u * 4u, gate8 = Cg >> 3;
    const uint32_t spk_rs = (uint32_t)p.spk_rs * 4u;
    uint32_t uab[NI], sb[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      uab[ni] = ((uint32_t)bcol[ni] * c8ab * T + (uint32_t)tcol[ni]) * 16u + half;
      sb[ni] = ((uint32_t)bcol[ni] * (uint32_t)p.spk_bs + (uint32_t)tcol[ni] * (uint32_t)p.spk_ts) * 4u;
    }
    uint2 xr[4][NI];
    uint32_t g8v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g8v[k] = (uint32_t)(mt * BMH + row0) / 8u + (uint32_t)k;
      const uint32_t g8c = g8v[k] * 8u < Cg ? g8v[k] : 0u;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (rl) xr[k][ni] = *reinterpret_cast<const uint2*>(rl + ((size_t)((row0 >> 3) + k) * rl_cols + ni * 32) * 16 + half);
        else xr[k][ni] = has_r ? dv3_ld<uint2>(p.r, ubc[ni] + g8c * gsz) : uint2{0u, 0u};
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t g8 = g8v[k];
      if (g8 * 8u >= Cg) continue;
      const uint32_t ch0 = g8 * 8u + (lhi ? 4u : 0u);
      float ba[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ba[e] = p.bias[ch0 + e];
          bg[e] = p.bias[Cg + ch0 + e];
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (!okc[ni]) continue;
        float xv[4], a[4], g[4], yv[4];
        dv3_unpack4(xr[k][ni], xv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = acc[0][ni][4 * k + e] + ba[e];
          g[e] = acc[1][ni][4 * k + e] + bg[e];
          if (p.spk) a[e] += dv3_ld<float>(p.spk, sb[ni] + (ch0 + e) * spk_rs);
        }
        if (p.ab) {
          dv3_st8(p.ab, uab[ni] + g8 * gsz, dv3_pack4(a));
          dv3_st8(p.ab, uab[ni] + (gate8 + g8) * gsz, dv3_pack4(g));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-g[e]));
          yv[e] = glu ? (a[e] * s + xv[e]) * oscale : s * a[e] + (1.0f - s) * xv[e];
        }
        dv3_st8(p.y, ub[ni] + g8 * gsz, dv3_pack4(yv));
      }
    }
    return;
  }
  if (p.mode == DV3_EPI_DGRAD) {
    const float dscale = p.drop_scale;
    const float rsc = p.r_scale != 0.f ? p.r_scale : 1.0f;
    const uint8_t* const ym = p.ymask_c8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint2 rv[4][NI];
      uint32_t kb[4][NI], g8v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g8v[k] = (uint32_t)(mt * BM + h * BMH + row0) / 8u + (uint32_t)k;
        const uint32_t g8c = g8v[k] * 8u < M ? g8v[k] : 0u;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          rv[k][ni] = p.r ? dv3_ld<uint2>(p.r, ubc[ni] + g8c * gsz) : uint2{0u, 0u};
          kb[k][ni] = ym ? (uint32_t)dv3_ld<uint8_t>(ym, (ubc[ni] + g8c * gsz) >> 4) : 0xffu;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (g8v[k] * 8u >= M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (!okc[ni]) continue;
          float r4[4], v[4];
          dv3_unpack4(rv[k][ni], r4);
          const uint32_t bits = kb[k][ni] >> (lhi ? 4 : 0);
          const uint32_t ch0 = g8v[k] * 8u + (lhi ? 4u : 0u);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = acc[h][ni][4 * k + e];
            if (ym) a = ((bits >> e) & 1u) ? a * dscale : 0.f;
            v[e] = ch0 + e < M ? a + rsc * r4[e] : 0.f;      // channels >= M of the last group stay zero
          }
          dv3_st8(p.y, ub[ni] + g8v[k] * gsz, dv3_pack4(v));
        }
      }
    }
    return;
  }
  // LINEAR / RELU / SIGMOID / SOFTSIGN (+ up to two fused residuals)
  const int mode = p.mode;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint2 rv[4][NI], r2v[4][NI];
    uint32_t g8v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g8v[k] = (uint32_t)(mt * BM + h * BMH + row0) / 8u + (uint32_t)k;
      const uint32_t g8c = g8v[k] * 8u < M ? g8v[k] : 0u;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        rv[k][ni] = p.r ? dv3_ld<uint2>(p.r, ubc[ni] + g8c * gsz) : uint2{0u, 0u};
        r2v[k][ni] = p.r2 ? dv3_ld<uint2>(p.r2, ubc[ni] + g8c * gsz) : uint2{0u, 0u};
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (g8v[k] * 8u >= M) continue;
      const uint32_t ch0 = g8v[k] * 8u + (lhi ? 4u : 0u);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = p.bias[ch0 + e < M ? ch0 + e : M - 1u];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (!okc[ni]) continue;
        float r4[4], q4[4], v[4];
        dv3_unpack4(rv[k][ni], r4);
        dv3_unpack4(r2v[k][ni], q4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = acc[h][ni][4 * k + e] + bv[e];
          if (mode == DV3_EPI_RELU) a = fmaxf(a, 0.f);
          else if (mode == DV3_EPI_SIGMOID) a = __builtin_amdgcn_rcpf(1.0f + __expf(-a));
          else if (mode == DV3_EPI_SOFTSIGN) a = a * __builtin_amdgcn_rcpf(1.0f + fabsf(a));
          if (p.r) a = (a + r4[e]) * rs2;
          if (p.r2) a = (a + q4[e]) * rs2;
          v[e] = ch0 + e < M ? a : 0.f;                       // channels >= M of the last group stay zero
        }
        dv3_st8(p.y, ub[ni] + g8v[k] * gsz, dv3_pack4(v));
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
