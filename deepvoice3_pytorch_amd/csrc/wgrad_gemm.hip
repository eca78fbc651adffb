// dv3_wgrad_gemm_f32: weight-gradient GEMM of the dilated conv (autograd of F.conv1d w.r.t.
// its weight, reference call sites deepvoice3_pytorch/modules.py:153,216), also used batched
// (n_slabs == B, J == 1) for the attention context product torch.bmm(p, values)
// (deepvoice3.py:167) and its gradients.
//
//   out[s][j][m][c] = sum_{b == s (mod S)} sum_t g[b][m][t] * xd[b][c][t + j*dil - padL]
//
// Both operands are time-contiguous ("NT" GEMM, K = time).  Tiles are staged in LDS in their
// global orientation with a +1 pad ([rows][BKT+1]) so the MFMA fragment reads (32 lanes = 32
// rows, one column) are bank-conflict free; v_mfma_f32_32x32x2_f32, 2x2 sub-tiles per wave.
// Split-K over the batch: each slab s is written separately (deterministic), and summed by
// dv3_weight_norm_bwd_f32.
#include "common.h"

namespace {

struct WgradArgs {
  dv3_wgrad_desc d;
  int m_tiles, c_tiles;
};

template <int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void wgrad_gemm_f32_kernel(const WgradArgs args) {
  constexpr int BM = WM * 64, BN = WN * 64, BKT = 32, LD = BKT + 1;
  constexpr int NT = WM * WN * 64;
  const dv3_wgrad_desc& p = args.d;
  __shared__ float Gs[BM * LD];
  __shared__ float Xs[BN * LD];

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

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][bq][r] = 0.f;

  const int T = p.T, Tin = p.Tin, M = p.M, Cin = p.Cin;
  const int tcol = tid & 31, trow = tid >> 5;
  constexpr int RSTEP = NT / 32;

  for (int b = s; b < p.B; b += p.n_slabs) {
    const float* __restrict__ gb = p.g + (int64_t)b * p.g_bs;
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
    for (int t0 = 0; t0 < T; t0 += BKT) {
      const int t = t0 + tcol;
      const int tg = t + shift;
      const bool tin = tg >= 0 && tg < Tin && t < T;
#pragma unroll 4
      for (int row = trow; row < BM; row += RSTEP) {
        const int m = m0 + row;
        float v = 0.f;
        if (m < M && t < T) v = gb[(int64_t)m * p.g_rs + t];
        Gs[row * LD + tcol] = v;
      }
#pragma unroll 4
      for (int row = trow; row < BN; row += RSTEP) {
        const int c = c0 + row;
        float v = 0.f;
        if (c < Cin && tin) {
          v = xb[(int64_t)c * p.x_rs + tg];
          if (p.xmask) {
            const uint32_t w = p.xmask[((int64_t)b * Cin + c) * p.xmask_rs + (tg >> 5)];
            v = ((w >> (tg & 31)) & 1u) ? v * p.drop_scale : 0.f;
          }
        }
        Xs[row * LD + tcol] = v;
      }
      __syncthreads();
      const float* ga = Gs + (wm * 64 + l31) * LD + lhi;
      const float* xa = Xs + (wn * 64 + l31) * LD + lhi;
#pragma unroll
      for (int kk = 0; kk < BKT; kk += 2) {
        const float a0 = ga[kk], a1 = ga[32 * LD + kk];
        const float b0 = xa[kk], b1 = xa[32 * LD + kk];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      __syncthreads();
    }
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

int dv3_wgrad_gemm_bf16x3_dispatch(const dv3_wgrad_desc* d, hipStream_t st);  // wgrad_gemm_bf16x3.hip
int dv3_wgrad_c8_dispatch(const dv3_wgrad_desc* d, hipStream_t st);           // wgrad_c8.hip

extern "C" int dv3_wgrad_gemm_f32(const dv3_wgrad_desc* d, void* stream) {
  DV3_REQUIRE(d && d->g && d->x && d->out, "wgrad_gemm: null pointer");
  DV3_REQUIRE(d->B > 0 && d->M > 0 && d->Cin > 0 && d->T > 0 && d->Tin > 0, "wgrad_gemm: bad dims");
  DV3_REQUIRE(d->J >= 1 && d->dil >= 1 && d->n_slabs >= 1 && (d->k_split || d->n_slabs <= d->B),
              "wgrad_gemm: bad J/dil/slabs");
  DV3_REQUIRE(d->ldo >= d->Cin, "wgrad_gemm: ldo < Cin");
  if (d->xmask) DV3_REQUIRE(d->xmask_rs * 32 >= d->Tin, "wgrad_gemm: xmask row stride too small");
  hipStream_t st = (hipStream_t)stream;
  if (d->c8) return dv3_wgrad_c8_dispatch(d, st);
  DV3_REQUIRE(!d->xmask_c8, "wgrad_gemm: xmask_c8 belongs to the c8 form");
  WgradArgs a;
  a.d = *d;
  const bool small = (d->M <= 64 && d->Cin <= 64);
  DV3_REQUIRE(!d->k_split || (d->split_bf16 && !small), "wgrad_gemm: k_split needs the split-bf16 kernel");
  DV3_REQUIRE(!d->g_pair || (d->split_bf16 == 1 && !small), "wgrad_gemm: pair-word g needs the three-term split kernels (M or Cin > 64)");
  if (d->split_bf16 && !small) {
    const int rc = dv3_wgrad_gemm_bf16x3_dispatch(d, st);
    if (rc != 1) return rc;
  }
  DV3_REQUIRE(!d->g_pair, "wgrad_gemm: shape not eligible for the split kernels, which alone read pair words");
  g_dv3_last_wgrad = 1000 + (small ? 0 : 10);
  if (small) {
    a.m_tiles = dv3_cdiv(d->M, 64);
    a.c_tiles = dv3_cdiv(d->Cin, 64);
    const int64_t nb = (int64_t)a.m_tiles * a.c_tiles * d->J * d->n_slabs;
    hipLaunchKernelGGL((wgrad_gemm_f32_kernel<1, 1>), dim3((unsigned)nb), dim3(64), 0, st, a);
  } else {
    a.m_tiles = dv3_cdiv(d->M, 128);
    a.c_tiles = dv3_cdiv(d->Cin, 128);
    const int64_t nb = (int64_t)a.m_tiles * a.c_tiles * d->J * d->n_slabs;
    DV3_REQUIRE(nb < (1ll << 31), "wgrad_gemm: grid too large");
    hipLaunchKernelGGL((wgrad_gemm_f32_kernel<2, 2>), dim3((unsigned)nb), dim3(256), 0, st, a);
  }
  return dv3_check_launch("wgrad_gemm_f32");
}
