// Split-bf16 ("bf16x3") weight-gradient GEMM (autograd of F.conv1d w.r.t. its weight; reference
// call sites deepvoice3_pytorch/modules.py:153,216 through loss.backward(), train.py:755):
//
//   out[s][j][m][c] = sum_{b == s (mod S)} sum_t g[b][m][t] * xd[b][c][t + j*dil - padL]
//
// Same arithmetic as conv_gemm_bf16x3.hip: operands split on the fly into hi + lo bf16,
// A_lo*B_hi + A_hi*B_lo + A_hi*B_hi accumulated in fp32 on v_mfma_f32_32x32x16_bf16.
//
// GEMM view: M = gradient rows m, N = input channels c, K = (batch item, time), one tap j per
// block (grid = m-tiles x c-tiles x J x slabs; the J blocks of a tile share the g panel through
// L2).  Both operands are time-contiguous in HBM, which is the MFMA K axis: a thread stages one
// "unit" = 8 consecutive time steps of one row (32 contiguous bytes, two dwordx4 loads), splits
// it and writes two 16-byte units.  LDS image [plane][k8][row + pad][8]: a fragment read is 32
// consecutive rows of one k8 block = 512 contiguous bytes (conflict free); the k8 stride is padded
// by 2 units so the staging writes (4 lanes = the 4 k8 blocks of a row) hit distinct banks.
// The tap shift only moves the START of the x unit (unaligned 32-byte read from L1/L2), so no
// shifted LDS addressing is needed.  Double-buffered LDS; operands are prefetched into registers
// TWO K steps ahead (two register sets: both operands stream from HBM), one barrier per step.
#include "common.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

namespace {

struct WgradArgs {
  dv3_wgrad_desc d;
  int m_tiles, c_tiles;
};

constexpr int BKT = 32;  // time steps per K step
constexpr int KB = 4;    // k8 blocks per K step
constexpr int PAD = 2;   // units of padding per k8 block

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

// 8 consecutive floats row[t..t+8), zero outside [0, len); `row` may be any valid pointer when
// nothing is in range.
__device__ __forceinline__ void load8(const float* __restrict__ row, int t, int len, float (&v)[8]) {
  if (t >= 0 && t + 8 <= len) {
    const f32x4u a = *reinterpret_cast<const f32x4u*>(row + t);
    const f32x4u b = *reinterpret_cast<const f32x4u*>(row + t + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int tt = t + e;
      v[e] = (tt >= 0 && tt < len) ? row[tt] : 0.f;
    }
  }
}

template <int WM, int WN, bool MASK, int TERMS>
__global__ __launch_bounds__(WM* WN * 64, 2) void wgrad_gemm_bf16x3_kernel(const WgradArgs args) {
  constexpr int BM = WM * 64, BN = WN * 64, NT = WM * WN * 64;
  constexpr int LDM = BM + PAD, LDN = BN + PAD;  // units per k8 block
  constexpr int GU = BM * KB / NT, XU = BN * KB / NT;
  static_assert(BM * KB % NT == 0 && BN * KB % NT == 0, "tiles must split evenly");
  const dv3_wgrad_desc& p = args.d;

  // [2 buffers] x { G hi [KB][LDM], G lo, X hi [KB][LDN], X lo }
  __shared__ __attribute__((aligned(16))) bf16x8 smem[2 * (2 * KB * LDM + 2 * KB * LDN)];
  constexpr int BUF = 2 * KB * LDM + 2 * KB * LDN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  int pid = blockIdx.x;
  const int mt = pid % args.m_tiles; pid /= args.m_tiles;
  const int ct = pid % args.c_tiles; pid /= args.c_tiles;
  const int j = pid % p.J;
  const int s = pid / p.J;
  const int m0 = mt * BM, c0 = ct * BN;
  const int shift = j * p.dil - p.padL;
  const int T = p.T, Tin = p.Tin, M = p.M, Cin = p.Cin;

  // this thread's staging units: (row, k8) -- 4 consecutive lanes cover one row's 32 time steps
  int grow[GU], gk8[GU], xrow[XU], xk8[XU];
#pragma unroll
  for (int u = 0; u < GU; ++u) {
    const int idx = tid + u * NT;
    grow[u] = idx >> 2;
    gk8[u] = idx & 3;
  }
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int idx = tid + u * NT;
    xrow[u] = idx >> 2;
    xk8[u] = idx & 3;
  }

  // two register sets: the operands of step k live in set k&1 from two steps before their use
  // (both operands stream from HBM; one step of MFMAs does not cover that latency)
  float rg[2][GU][8], rx[2][XU][8];
  uint32_t rmask[2][MASK ? XU : 1];   // 8 keep-bits of the unit
  const int n_tc = (T + BKT - 1) / BKT;          // time chunks per batch item
  const int n_b = (p.B - s + p.n_slabs - 1) / p.n_slabs;
  const int nsteps = n_b * n_tc;

  auto load_step = [&](int step, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    const int bi = step / n_tc, tc = step - bi * n_tc;
    const int b = s + bi * p.n_slabs;
    const int t0 = tc * BKT;
    const float* __restrict__ gb = p.g + (int64_t)b * p.g_bs;
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int m = m0 + grow[u];
      const int mc = m < M ? m : M - 1;
      load8(gb + (int64_t)mc * p.g_rs, t0 + gk8[u] * 8, (m < M) ? T : 0, rg[S][u]);
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int c = c0 + xrow[u];
      const int cc = c < Cin ? c : Cin - 1;
      const int t = t0 + xk8[u] * 8;          // g-time of the unit; x is read at t + shift
      // g is zero for g-times >= T, so x only needs its own [0, Tin) clipping
      const int tx = t + shift;
      load8(xb + (int64_t)cc * p.x_rs, tx, (c < Cin) ? Tin : 0, rx[S][u]);
      if (MASK) {
        const uint32_t* __restrict__ mr = p.xmask + ((int64_t)b * Cin + cc) * p.xmask_rs;
        const int txc = max(tx, 0);
        const int w0 = txc >> 5;
        const int wl = (Tin + 31) / 32 - 1;
        const uint64_t lo = mr[min(w0, wl)], hi = mr[min(w0 + 1, wl)];
        uint32_t bits = (uint32_t)(((hi << 32) | lo) >> (txc & 31));
        if (tx < 0) bits = (-tx < 32) ? bits << (-tx) : 0u;   // bit e of `bits` <-> element e of the unit
        rmask[S][u] = bits;
      }
    }
  };
  auto write_step = [&](int buf, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    bf16x8* dst = smem + buf * BUF;
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      bf16x8 hi, lo;
      split8(rg[S][u], hi, lo);
      const int o = gk8[u] * LDM + grow[u];
      dst[o] = hi;
      if (TERMS == 3) dst[KB * LDM + o] = lo;
    }
    bf16x8* dx = dst + 2 * KB * LDM;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = rx[S][u][e];
        if (MASK) v[e] *= ((rmask[S][u] >> e) & 1u) ? p.drop_scale : 0.f;
      }
      bf16x8 hi, lo;
      split8(v, hi, lo);
      const int o = xk8[u] * LDN + xrow[u];
      dx[o] = hi;
      if (TERMS == 3) dx[KB * LDN + o] = lo;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][bq][r] = 0.f;

  auto mfma_step = [&](int cur) {
    const bf16x8* GsH = smem + cur * BUF;
    const bf16x8* GsL = GsH + KB * LDM;
    const bf16x8* XsH = GsH + 2 * KB * LDM;
    const bf16x8* XsL = XsH + KB * LDN;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int k8 = 2 * ks + lhi;
      const int ai = k8 * LDM + wm * 64 + l31;
      const int xi = k8 * LDN + wn * 64 + l31;
      const bf16x8 ah0 = GsH[ai], ah1 = GsH[ai + 32];
      const bf16x8 bh0 = XsH[xi], bh1 = XsH[xi + 32];
      if (TERMS == 1) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh1, acc[1][1], 0, 0, 0);
        continue;
      }
      const bf16x8 al0 = GsL[ai], al1 = GsL[ai + 32];
      const bf16x8 bl0 = XsL[xi], bl1 = XsL[xi + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh1, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl1, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh1, acc[1][1], 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  if (nsteps > 0) {
    load_step(0, S0{});
    write_step(0, S0{});
  }
  if (nsteps > 1) load_step(1, S1{});
  __syncthreads();

  // steps in pairs so the register-set index is static
  for (int step = 0; step < nsteps; step += 2) {
    // even step: LDS buffer 0 holds step; set 1 holds step+1; set 0 is free -> fetch step+2
    if (step + 2 < nsteps) load_step(step + 2, S0{});
    mfma_step(0);
    if (step + 1 < nsteps) write_step(1, S1{});
    __syncthreads();
    if (step + 1 >= nsteps) break;
    // odd step: buffer 1 holds step+1; set 0 holds step+2; set 1 is free -> fetch step+3
    if (step + 3 < nsteps) load_step(step + 3, S1{});
    mfma_step(1);
    if (step + 2 < nsteps) write_step(0, S0{});
    __syncthreads();
  }

  float* __restrict__ ob = p.out + (int64_t)s * p.out_ss + (int64_t)j * M * p.ldo;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int c = c0 + wn * 64 + ni * 32 + l31;
      if (c >= Cin) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m < M) ob[(int64_t)m * p.ldo + c] = acc[mi][ni][r];
      }
    }
}

}  // namespace

// called by dv3_wgrad_gemm_f32 (wgrad_gemm.hip) when d->split_bf16 is set
int dv3_wgrad_gemm_bf16x3_dispatch(const dv3_wgrad_desc* d, hipStream_t st) {
  WgradArgs a;
  a.d = *d;
  a.m_tiles = dv3_cdiv(d->M, 128);
  a.c_tiles = dv3_cdiv(d->Cin, 128);
  const int64_t nb = (int64_t)a.m_tiles * a.c_tiles * d->J * d->n_slabs;
  DV3_REQUIRE(nb < (1ll << 31), "wgrad_gemm: grid too large");
  if (d->split_bf16 == 2) {   // single-term bf16 (hi planes only)
    if (d->xmask) {
      hipLaunchKernelGGL((wgrad_gemm_bf16x3_kernel<2, 2, true, 1>), dim3((unsigned)nb), dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL((wgrad_gemm_bf16x3_kernel<2, 2, false, 1>), dim3((unsigned)nb), dim3(256), 0, st, a);
    }
  } else if (d->xmask) {
    hipLaunchKernelGGL((wgrad_gemm_bf16x3_kernel<2, 2, true, 3>), dim3((unsigned)nb), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((wgrad_gemm_bf16x3_kernel<2, 2, false, 3>), dim3((unsigned)nb), dim3(256), 0, st, a);
  }
  return dv3_check_launch("wgrad_gemm_bf16x3");
}
