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
  int prio = 0;   // all-taps kernel: wave priority scheme (dv3_debug_set(15, v))
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

template <typename T>
__device__ __forceinline__ T ldg_off(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
// 8 consecutive floats at element offset `off` of a tensor with `total` elements, RAW (the
// caller masks elements outside its row: a unit that merely crosses a row end reads the
// neighbouring row's finite values, which are then zeroed).  Uniform base + 32-bit offsets; the
// whole wave takes the two-dwordx4 path unless one of its units touches the tensor's two ends.
__device__ __forceinline__ void load8_raw(const float* __restrict__ base, int off, int total,
                                          float (&v)[8]) {
  const bool inside = (unsigned)off <= (unsigned)(total - 8);
  if (__all(inside)) {
    const f32x4u a = ldg_off<f32x4u>(base, (uint32_t)off * 4u);
    const f32x4u b = ldg_off<f32x4u>(base, (uint32_t)off * 4u + 16u);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ldg_off<float>(base, (uint32_t)min(max(off + e, 0), total - 1) * 4u);
  }
}
// bit e set <=> 0 <= t + e < len
__device__ __forceinline__ uint32_t valid8(int t, int len) {
  const int elo = max(0, -t), ehi = min(8, len - t);
  return ehi > elo ? (((1u << ehi) - 1u) & ~((1u << elo) - 1u)) : 0u;
}
// v[e] = bit e of m ? v[e] : 0
__device__ __forceinline__ void mask8(float (&v)[8], uint32_t m) {
#pragma unroll
  for (int e = 0; e < 8; ++e)
    v[e] = __uint_as_float(__float_as_uint(v[e]) & (uint32_t)__builtin_amdgcn_sbfe((int)m, e, 1));
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

  // the m-tile x c-tile x tap blocks of one K slab read the same slices of g and x: give each XCD a contiguous
  // run of logical block ids so they meet in ONE L2 (dispatch order spread them over all eight: 586 MB fetched
  // per launch against 201 MB algorithmic at the north-star shape, profiles/r02_hbm_traffic.json)
  int pid = dv3_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = pid % args.m_tiles; pid /= args.m_tiles;
  const int ct = pid % args.c_tiles; pid /= args.c_tiles;
  const int j = pid % p.J;
  const int s = pid / p.J;
  const int m0 = mt * BM, c0 = ct * BN;
  const int shift = j * p.dil - p.padL;
  const int T = p.T, Tin = p.Tin, M = p.M, Cin = p.Cin;

  const bool gpw = TERMS == 3 && p.g_pair != 0;
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
  uint32_t gval[2][GU], xval[2][XU];   // per-unit element masks: validity (& dropout keep-bits for x)
  const int n_tc = (T + BKT - 1) / BKT;          // time chunks per batch item
  // K partition: k_split 0 = slab s owns batch items s, s+S, ...; 1 = slab s owns the s-th contiguous
  // range of the flattened (batch item, chunk) sequence (any S: the grid can be sized to the chip)
  int nsteps, step0 = 0;
  if (p.k_split) {
    const int total = p.B * n_tc, q = (total + p.n_slabs - 1) / p.n_slabs;
    step0 = s * q;
    nsteps = max(0, min(q, total - step0));
  } else {
    nsteps = ((p.B - s + p.n_slabs - 1) / p.n_slabs) * n_tc;
  }
  const int g_total = (p.B - 1) * (int)p.g_bs + (M - 1) * (int)p.g_rs + T;      // < 2^31 (host-checked)
  const int x_total = (p.B - 1) * (int)p.x_bs + (Cin - 1) * (int)p.x_rs + Tin;
  const int wl = (Tin + 31) / 32 - 1;
  // per-unit row offsets (elements) and mask-row offsets (words), fixed over the K loop
  int grow_off[GU], xrow_off[XU], xm_off[MASK ? XU : 1];
  bool grow_ok[GU], xrow_ok[XU];
#pragma unroll
  for (int u = 0; u < GU; ++u) {
    const int m = m0 + grow[u];
    grow_ok[u] = m < M;
    grow_off[u] = (m < M ? m : M - 1) * (int)p.g_rs + gk8[u] * 8;
  }
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int c = c0 + xrow[u];
    xrow_ok[u] = c < Cin;
    const int cc = c < Cin ? c : Cin - 1;
    xrow_off[u] = cc * (int)p.x_rs + xk8[u] * 8 + shift;
    if (MASK) xm_off[u] = cc * p.xmask_rs;
  }

  auto load_step = [&](int step, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    const int gs = step0 + step;
    const int bi = gs / n_tc, tc = gs - bi * n_tc;
    const int b = p.k_split ? bi : s + bi * p.n_slabs;
    const int t0 = tc * BKT;
    const int gb = b * (int)p.g_bs + t0, xb = b * (int)p.x_bs + t0;
    const bool g_edge = t0 + BKT > T;                      // uniform: the chunk holds the row tail
    const bool x_edge = t0 + shift < 0 || t0 + BKT + shift > Tin;
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      load8_raw(p.g, gb + grow_off[u], g_total, rg[S][u]);
      uint32_t vm = 0xffu;
      if (g_edge) vm = valid8(t0 + gk8[u] * 8, T);
      gval[S][u] = grow_ok[u] ? vm : 0u;
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      load8_raw(p.x, xb + xrow_off[u], x_total, rx[S][u]);
      const int tx = t0 + xk8[u] * 8 + shift;   // g is zero for g-times >= T: x needs only its own clipping
      uint32_t vm = 0xffu;
      if (x_edge) vm = valid8(tx, Tin);
      if (!xrow_ok[u]) vm = 0u;
      if (MASK) {
        const uint32_t mo = (uint32_t)(b * Cin * p.xmask_rs + xm_off[u]);
        const int txc = max(tx, 0);
        const int w0 = min(txc >> 5, wl);
        const uint32_t lo = ldg_off<uint32_t>(p.xmask, (mo + (uint32_t)w0) * 4u);
        const uint32_t hi = ldg_off<uint32_t>(p.xmask, (mo + (uint32_t)min(w0 + 1, wl)) * 4u);
        uint32_t bits = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(txc & 31));
        if (x_edge && tx < 0) bits = (-tx < 32) ? bits << (-tx) : 0u;   // bit e <-> element e of the unit
        vm &= bits;
      }
      xval[S][u] = vm;
    }
  };
  auto write_step = [&](int buf, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    bf16x8* dst = smem + buf * BUF;
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      if (__any((gval[S][u] & 0xffu) != 0xffu)) mask8(rg[S][u], gval[S][u]);   // rare: row tails
      bf16x8 hi, lo;
      if (TERMS == 3 && gpw) {     // pair words (dv3_wgrad_desc.g_pair, uniform): the pair is already there
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(rg[S][u][e]);
        dv3_pair_units(w, hi, lo);
      } else split8(rg[S][u], hi, lo);
      const int o = gk8[u] * LDM + grow[u];
      dst[o] = hi;
      if (TERMS == 3) dst[KB * LDM + o] = lo;
    }
    bf16x8* dx = dst + 2 * KB * LDM;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      // dropout: keep-bits select here, the 1/(1-p) scale is applied once to the accumulators
      if (MASK || __any((xval[S][u] & 0xffu) != 0xffu)) mask8(rx[S][u], xval[S][u]);
      bf16x8 hi, lo;
      split8(rx[S][u], hi, lo);
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
  const float oscale = MASK ? p.drop_scale : 1.0f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int c = c0 + wn * 64 + ni * 32 + l31;
      if (c >= Cin) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m < M) ob[(int64_t)m * p.ldo + c] = acc[mi][ni][r] * oscale;
      }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// All taps in one workgroup (JT = 3, the models' kernel size): the per-tap form above runs J workgroups per tile that
// each fetch, split and store the SAME g panel; here the g panel is staged once per K step and serves the three
// shifted x panels.  8 waves: wave (wm, wc) owns 64 gradient rows x 32 input channels for all three taps, so its four g
// fragments feed 18 MFMAs per k16 block (10 fragment reads per 18 MFMAs instead of 8 per 12) and a workgroup stages
// 2048 units per 288 MFMAs instead of 3 x 1024.  Same arithmetic and accumulation order per output element (one tap's
// products in K order), same slab output, so the two forms agree bit for bit.
template <bool MASK, int TERMS>
__global__ __launch_bounds__(512, 2) void wgrad_taps_kernel(const WgradArgs args) {
  constexpr int JT = 3, BM = 128, BN = 128, NT = 512;
  constexpr int LDM = BM + PAD, LDN = BN + PAD;
  constexpr int GBUF = 2 * KB * LDM, XTAP = 2 * KB * LDN, BUF = GBUF + JT * XTAP;
  const dv3_wgrad_desc& p = args.d;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_t[];
  bf16x8* const smem = reinterpret_cast<bf16x8*>(smem_raw_t);       // [2 buffers][G hi, G lo | 3 x (X hi, X lo)]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, lhi = lane >> 5;

  int pid = dv3_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = pid % args.m_tiles; pid /= args.m_tiles;
  const int ct = pid % args.c_tiles;
  const int s = pid / args.c_tiles;
  const int m0 = mt * BM, c0 = ct * BN;
  const int T = p.T, Tin = p.Tin, M = p.M, Cin = p.Cin;

  // this thread's staging unit: (row, k8) -- the same for the g panel and the three x panels
  const int urow = tid >> 2, uk8 = tid & 3;
  float rg[1][8], rx[1][JT][8];      // one register set: the next step's operands are fetched before this step's MFMAs
  uint32_t gval[1], xval[1][JT];
  const int n_tc = (T + BKT - 1) / BKT;
  int nsteps, step0 = 0;
  {
    const int total = p.B * n_tc, q = (total + p.n_slabs - 1) / p.n_slabs;
    step0 = s * q;
    nsteps = max(0, min(q, total - step0));
  }
  const int g_total = (p.B - 1) * (int)p.g_bs + (M - 1) * (int)p.g_rs + T;
  const int x_total = (p.B - 1) * (int)p.x_bs + (Cin - 1) * (int)p.x_rs + Tin;
  const int wl = (Tin + 31) / 32 - 1;
  const int gm = m0 + urow, xc = c0 + urow;
  const bool grow_ok = gm < M, xrow_ok = xc < Cin;
  const int grow_off = (grow_ok ? gm : M - 1) * (int)p.g_rs + uk8 * 8;
  const int xcc = xrow_ok ? xc : Cin - 1;
  const int xrow_off = xcc * (int)p.x_rs + uk8 * 8;
  const int xm_off = xcc * p.xmask_rs;

  auto load_step = [&](int step, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    const int gs = step0 + step;
    const int b = gs / n_tc, tc = gs - b * n_tc;
    const int t0 = tc * BKT;
    const int gb = b * (int)p.g_bs + t0, xb = b * (int)p.x_bs + t0;
    load8_raw(p.g, gb + grow_off, g_total, rg[S]);
    uint32_t vm = 0xffu;
    if (t0 + BKT > T) vm = valid8(t0 + uk8 * 8, T);
    gval[S] = grow_ok ? vm : 0u;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int shift = j * p.dil - p.padL;
      const bool x_edge = t0 + shift < 0 || t0 + BKT + shift > Tin;
      load8_raw(p.x, xb + xrow_off + shift, x_total, rx[S][j]);
      const int tx = t0 + uk8 * 8 + shift;
      uint32_t xm = 0xffu;
      if (x_edge) xm = valid8(tx, Tin);
      if (!xrow_ok) xm = 0u;
      if (MASK) {
        const uint32_t mo = (uint32_t)(b * Cin * p.xmask_rs + xm_off);
        const int txc = max(tx, 0);
        const int w0 = min(txc >> 5, wl);
        const uint32_t lo = ldg_off<uint32_t>(p.xmask, (mo + (uint32_t)w0) * 4u);
        const uint32_t hi = ldg_off<uint32_t>(p.xmask, (mo + (uint32_t)min(w0 + 1, wl)) * 4u);
        uint32_t bits = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(txc & 31));
        if (x_edge && tx < 0) bits = (-tx < 32) ? bits << (-tx) : 0u;
        xm &= bits;
      }
      xval[S][j] = xm;
    }
  };
  auto write_step = [&](int buf, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    bf16x8* dst = smem + buf * BUF;
    {
      if (__any((gval[S] & 0xffu) != 0xffu)) mask8(rg[S], gval[S]);
      bf16x8 hi, lo;
      if (TERMS == 3 && p.g_pair != 0) {
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(rg[S][e]);
        dv3_pair_units(w, hi, lo);
      } else split8(rg[S], hi, lo);
      const int o = uk8 * LDM + urow;
      dst[o] = hi;
      if (TERMS == 3) dst[KB * LDM + o] = lo;
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      bf16x8* dx = dst + GBUF + j * XTAP;
      if (MASK || __any((xval[S][j] & 0xffu) != 0xffu)) mask8(rx[S][j], xval[S][j]);
      bf16x8 hi, lo;
      split8(rx[S][j], hi, lo);
      const int o = uk8 * LDN + urow;
      dx[o] = hi;
      if (TERMS == 3) dx[KB * LDN + o] = lo;
    }
  };

  f32x16 acc[JT][2];
#pragma unroll
  for (int j = 0; j < JT; ++j)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][a][r] = 0.f;

  auto mfma_step = [&](int cur) {
    const bf16x8* GsH = smem + cur * BUF;
    const bf16x8* GsL = GsH + KB * LDM;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int k8 = 2 * ks + lhi;
      const int ai = k8 * LDM + wm * 64 + l31;
      const bf16x8 ah0 = GsH[ai], ah1 = GsH[ai + 32];
      bf16x8 al0 = ah0, al1 = ah1;
      if (TERMS == 3) { al0 = GsL[ai]; al1 = GsL[ai + 32]; }
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const bf16x8* XsH = GsH + GBUF + j * XTAP;
        const int xi = k8 * LDN + wc * 32 + l31;
        const bf16x8 bh = XsH[xi];
        if (TERMS == 1) {
          acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[j][0], 0, 0, 0);
          acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[j][1], 0, 0, 0);
          continue;
        }
        const bf16x8 bl = XsH[KB * LDN + xi];
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh, acc[j][1], 0, 0, 0);
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl, acc[j][1], 0, 0, 0);
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[j][1], 0, 0, 0);
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;

  if (nsteps > 0) {
    load_step(0, S0{});
    write_step(0, S0{});
  }
  __syncthreads();
  const int prio = args.prio;   // measurement knob: 1 = staging at priority 3, 2 = MFMAs at priority 3
  for (int step = 0; step < nsteps; ++step) {
    if (prio == 1) __builtin_amdgcn_s_setprio(3);
    if (step + 1 < nsteps) load_step(step + 1, S0{});
    if (prio == 1) __builtin_amdgcn_s_setprio(0);
    if (prio == 2) __builtin_amdgcn_s_setprio(3);
    mfma_step(step & 1);
    if (prio == 2) __builtin_amdgcn_s_setprio(0);
    if (prio == 1) __builtin_amdgcn_s_setprio(3);
    if (step + 1 < nsteps) write_step((step + 1) & 1, S0{});
    if (prio == 1) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
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

}  // namespace

int g_wgrad_taps2_default = 1;   // the two-steps-ahead form of the all-taps kernel (wgrad_taps2.hip; masked 205 -> 173 us, unmasked
                                 // 164 -> 150 us at the north-star shape, bit-identical slabs); dv3_debug_set(17, 0) = the one-step form
int dv3_wgrad_taps2_dispatch(const dv3_wgrad_desc* d, hipStream_t st);
int g_wgrad_taps_default = 1;   // the all-taps form measured 7-14 % faster at every model shape (scripts/wgrad_ab.py)
int g_wgrad_prio = 0;   // debug (dv3_debug_set(15, v))
int g_wgrad_tile = 0;   // debug (dv3_debug_set(2, v)): 0 auto, 1 force 128x128 per tap, 2 force 256x128 per tap, 3 force all-taps

template <bool MASK, int TERMS>
static int launch_wgrad_taps_t(const WgradArgs& a, int64_t nb, hipStream_t st) {
  constexpr int LDM = 128 + PAD;
  constexpr size_t lds = (size_t)2 * (2 * KB * LDM + 3 * 2 * KB * LDM) * 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_taps_kernel<MASK, TERMS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      dv3_set_error("wgrad_taps: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_taps_kernel<MASK, TERMS>), dim3((unsigned)nb), dim3(512), lds, st, a);
  return dv3_check_launch("wgrad_taps");
}
static int launch_wgrad_taps(const WgradArgs& a, int64_t nb, hipStream_t st) {
  const dv3_wgrad_desc* d = &a.d;
  if (d->split_bf16 == 2)
    return d->xmask ? launch_wgrad_taps_t<true, 1>(a, nb, st) : launch_wgrad_taps_t<false, 1>(a, nb, st);
  return d->xmask ? launch_wgrad_taps_t<true, 3>(a, nb, st) : launch_wgrad_taps_t<false, 3>(a, nb, st);
}

template <int WM, int WN>
static int launch_wgrad_x3(const WgradArgs& a, int64_t nb, hipStream_t st) {
  const dv3_wgrad_desc* d = &a.d;
  dim3 grid((unsigned)nb), block(WM * WN * 64);
  if (d->split_bf16 == 2) {   // single-term bf16 (hi planes only)
    if (d->xmask) {
      hipLaunchKernelGGL((wgrad_gemm_bf16x3_kernel<WM, WN, true, 1>), grid, block, 0, st, a);
    } else {
      hipLaunchKernelGGL((wgrad_gemm_bf16x3_kernel<WM, WN, false, 1>), grid, block, 0, st, a);
    }
  } else if (d->xmask) {
    hipLaunchKernelGGL((wgrad_gemm_bf16x3_kernel<WM, WN, true, 3>), grid, block, 0, st, a);
  } else {
    hipLaunchKernelGGL((wgrad_gemm_bf16x3_kernel<WM, WN, false, 3>), grid, block, 0, st, a);
  }
  return dv3_check_launch("wgrad_gemm_bf16x3");
}

// called by dv3_wgrad_gemm_f32 (wgrad_gemm.hip) when d->split_bf16 is set
int dv3_wgrad_gemm_bf16x3_dispatch(const dv3_wgrad_desc* d, hipStream_t st) {
  // 32-bit element offsets inside the kernel
  if ((int64_t)d->B * d->g_bs + (int64_t)d->M * d->g_rs >= (1ll << 30) ||
      (int64_t)d->B * d->x_bs + (int64_t)d->Cin * d->x_rs >= (1ll << 30) ||
      (d->xmask && (int64_t)d->B * d->Cin * d->xmask_rs >= (1ll << 30)))
    return 1;   // caller falls back to the exact kernel
  WgradArgs a;
  a.d = *d;
  a.prio = g_wgrad_prio;
  if (d->J == 3 && d->k_split && d->split_bf16 != 2 && (g_wgrad_tile == 4 || (g_wgrad_tile == 0 && g_wgrad_taps2_default)) &&
      (int64_t)(d->B - 1) * d->g_bs + (int64_t)(d->M - 1) * d->g_rs + d->T >= 8 &&
      (int64_t)(d->B - 1) * d->x_bs + (int64_t)(d->Cin - 1) * d->x_rs + d->Tin >= 8)
    return dv3_wgrad_taps2_dispatch(d, st);
  // three taps, K split over contiguous ranges: one 8-wave workgroup per (m-tile, c-tile, slab) serves all taps
  if (d->J == 3 && d->k_split && (g_wgrad_tile == 3 || (g_wgrad_tile == 0 && g_wgrad_taps_default))) {
    a.m_tiles = dv3_cdiv(d->M, 128);
    a.c_tiles = dv3_cdiv(d->Cin, 128);
    const int64_t nb = (int64_t)a.m_tiles * a.c_tiles * d->n_slabs;
    DV3_REQUIRE(nb < (1ll << 31), "wgrad_gemm: grid too large");
    g_dv3_last_wgrad = (d->split_bf16 == 2 ? 4000 : 3000) + 30;
    return launch_wgrad_taps(a, nb, st);
  }
  // 128 x 128 (4 waves, two workgroups per CU).  The 256 x 128 8-wave variant (one workgroup per CU)
  // measured within run-to-run noise of it (+-5 %, scripts/x3_check.py): kept behind the debug knob.
  const bool big = g_wgrad_tile == 2;
  const int bm = big ? 256 : 128;
  a.m_tiles = dv3_cdiv(d->M, bm);
  a.c_tiles = dv3_cdiv(d->Cin, 128);
  const int64_t nb = (int64_t)a.m_tiles * a.c_tiles * d->J * d->n_slabs;
  DV3_REQUIRE(nb < (1ll << 31), "wgrad_gemm: grid too large");
  g_dv3_last_wgrad = (d->split_bf16 == 2 ? 4000 : 3000) + (big ? 20 : 10);
  return big ? launch_wgrad_x3<4, 2>(a, nb, st) : launch_wgrad_x3<2, 2>(a, nb, st);
}
