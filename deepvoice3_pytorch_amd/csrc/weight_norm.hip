// Weight normalisation (nn.utils.weight_norm as applied by the reference factories,
// deepvoice3_pytorch/modules.py:80-109) fused with packing into the operand layouts of the
// tap-GEMM kernels, and its backward from the wgrad slabs.
//
//   w = g * v / ||v||   (norm over every dim but 0)
//   Conv1d / Linear      v [O][I][J]  norm per output channel o
//   ConvTranspose1d      v [I][O][J]  norm per INPUT channel i (dim 0 of its weight)
#include "common.h"

int g_wn_bwd_vec4 = 1;     // dv3_debug_set(51, v): 0 = the 4-byte gather everywhere (A/B, bit-identity tests)

namespace {

// scale[r] = 1/||v[r]||  (1 when g == NULL: plain weight).  One block per row.
__device__ __forceinline__ void wn_inv_norm_row(const float* __restrict__ v, const float* __restrict__ g,
                                                float* __restrict__ scale, int len, int r, float* red) {
  if (!g) {
    if (threadIdx.x == 0) scale[r] = 1.0f;
    return;
  }
  const float* row = v + (int64_t)r * len;
  float s = 0.f;
  for (int i = threadIdx.x; i < len; i += 256) {
    const float x = row[i];
    s += x * x;
  }
  s = dv3_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) scale[r] = 1.0f / sqrtf(red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void wn_inv_norm_kernel(const float* __restrict__ v,
                                                          const float* __restrict__ g,
                                                          float* __restrict__ scale, int len) {
  __shared__ float red[4];
  wn_inv_norm_row(v, g, scale, len, blockIdx.x, red);
}

// Forward pack, Conv1d/Linear: fwd[j][i][col(o)] = g[o]*scale[o]*v[o][i][j]
// block: 32 output channels x 32 input channels, all taps.  LDS tile [32 o][32*J + 1].
__global__ __launch_bounds__(256) void wn_pack_fwd_kernel(const dv3_wn_desc p) {
  extern __shared__ float tile[];  // [32][32*J+1]
  const int O = p.O, I = p.I, J = p.J;
  const int o0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  const int W = 32 * J, LD = W + 1;
  // load: rows o, contiguous (i,j) span of 32*J floats starting at i0*J
  for (int idx = threadIdx.x; idx < 32 * W; idx += 256) {
    const int ol = idx / W, q = idx % W;
    const int o = o0 + ol, i = i0 + q / J;
    float val = 0.f;
    if (o < O && i < I) {
      const float sc = p.g ? p.g[o] * p.scale[o] : 1.0f;
      val = sc * p.v[((int64_t)o * I + i0) * J + q];
    }
    tile[ol * LD + q] = val;
  }
  __syncthreads();
  // store: for each (j, i): 32 consecutive o
  for (int idx = threadIdx.x; idx < 32 * W; idx += 256) {
    const int ol = idx & 31, q = idx >> 5;  // q = il*J + j
    const int il = q / J, j = q % J;
    const int o = o0 + ol, i = i0 + il;
    if (o < O && i < I) {
      int col = o;
      if (p.glu_cg > 0 && o >= p.glu_cg) col = p.a_half + (o - p.glu_cg);
      p.fwd_pack[((int64_t)j * I + i) * p.lda + col] = tile[ol * LD + q];
    }
  }
}

// Backward (DGRAD operand) pack, Conv1d/Linear: bwd[J-1-j][o][i] = g[o]*scale[o]*v[o][i][j]
// one block per o; zero-fills the pad columns [I, ldb).
__global__ __launch_bounds__(256) void wn_pack_bwd_kernel(const dv3_wn_desc p) {
  const int o = blockIdx.x, I = p.I, J = p.J;
  const float sc = p.g ? p.g[o] * p.scale[o] : 1.0f;
  const float* row = p.v + (int64_t)o * I * J;
  for (int idx = threadIdx.x; idx < p.ldb * J; idx += 256) {
    const int j = idx / p.ldb, i = idx % p.ldb;
    const float val = (i < I) ? sc * row[i * J + j] : 0.f;
    p.bwd_pack[((int64_t)(J - 1 - j) * p.O + o) * p.ldb + i] = val;
  }
}

// ConvTranspose1d (v [I][O][J]): one block per input channel i.
//   fwd[0][i][j*O + o] = g[i]*scale[i]*v[i][o][j]            (K = I rows, M' = J*O cols)
//   bwd[0][j*O + o][i] = same value                          (K' = J*O rows, m = i cols)
__global__ __launch_bounds__(256) void wn_pack_transposed_kernel(const dv3_wn_desc p) {
  const int i = blockIdx.x, O = p.O, J = p.J;
  const float sc = p.g ? p.g[i] * p.scale[i] : 1.0f;
  const float* row = p.v + (int64_t)i * O * J;
  for (int idx = threadIdx.x; idx < p.lda; idx += 256) {
    float val = 0.f;
    if (idx < J * O) {
      const int j = idx / O, o = idx % O;
      val = sc * row[o * J + j];
      if (p.bwd_pack) p.bwd_pack[(int64_t)idx * p.ldb + i] = val;
    }
    p.fwd_pack[(int64_t)i * p.lda + idx] = val;
  }
}

// Fused weight norm + split-bf16 packing of BOTH tap-GEMM operands (forward and input-gradient)
// straight from v, g: what wn_pack_fwd + wn_pack_bwd + 2x split_pack produce, in one launch and
// without the fp32 images.  Block = 32 output channels x 32 input channels x all taps, staged
// through an LDS tile [32 o][32*J + 1] of scaled weights.
//   fwd image [plane][j][k8 over i][col(o)][8 i]     bwd image [plane][J-1-j][k8 over o][i][8 o]
typedef __bf16 wn_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void wn_split8(const float (&v)[8], wn_bf16x8& hi, wn_bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}
// scaled fp16 split of the forward image (include/dv3hip.h "f16x3"): a = v * 2^8; true when a left the fp16 range
__device__ __forceinline__ bool wn_split8_f16(const float (&v)[8], wn_bf16x8& hi, wn_bf16x8& lo) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = v[i] * (float)(1 << DV3_F16_WEIGHT_SHIFT);
  return dv3_split8_f16(a, hi, lo);
}
__device__ __forceinline__ void wn_split_both_block(const dv3_wn_desc& p, wn_bf16x8* __restrict__ fs,
                                                    wn_bf16x8* __restrict__ bs, int bx, int by, float* tile,
                                                    uint32_t* range_ctr) {
  const int O = p.O, I = p.I, J = p.J;
  const int o0 = bx * 32, i0 = by * 32;
  const int W = 32 * J, LD = W + 1;
  for (int idx = threadIdx.x; idx < 32 * W; idx += 256) {
    const int ol = idx / W, q = idx % W;
    const int o = o0 + ol, i = i0 + q / J;
    float val = 0.f;
    if (o < O && i < I) {
      const float sc = p.g ? p.g[o] * p.scale[o] : 1.0f;
      val = sc * p.v[((int64_t)o * I + i0) * J + q];
    }
    tile[ol * LD + q] = val;
  }
  __syncthreads();
  const int k8f = (I + 31) / 32 * 4, k8b = (O + 31) / 32 * 4;   // k8 blocks of the two images
  const int64_t plane_f = (int64_t)J * k8f * p.lda, plane_b = (int64_t)J * k8b * p.ldb;
  // forward units: (j, q in 0..3, ol): 8 consecutive i of output channel o
  for (int u = threadIdx.x; u < J * 4 * 32; u += 256) {
    const int ol = u & 31, q = (u >> 5) & 3, j = u >> 7;
    const int o = o0 + ol;
    if (o >= O) continue;
    int col = o;
    if (p.glu_cg > 0 && o >= p.glu_cg) col = p.a_half + (o - p.glu_cg);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[ol * LD + (q * 8 + e) * J + j];
    wn_bf16x8 hi, lo;
    bool bad = false;
    if (p.fwd_dtype == DV3_SPLIT_DTYPE_F16) bad = wn_split8_f16(v, hi, lo); else wn_split8(v, hi, lo);
    if (bad && range_ctr) atomicAdd(range_ctr, 1u);     // divergent loop: no wave-wide ballot here
    const int64_t g = ((int64_t)j * k8f + (i0 >> 3) + q) * p.lda + col;
    fs[g] = hi;
    fs[plane_f + g] = lo;
  }
  if (!bs) return;
  // input-gradient units: (j, q, il): 8 consecutive o of input channel i, taps reversed
  for (int u = threadIdx.x; u < J * 4 * 32; u += 256) {
    const int il = u & 31, q = (u >> 5) & 3, j = u >> 7;
    const int i = i0 + il;
    if (i >= I) continue;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(q * 8 + e) * LD + il * J + j];
    wn_bf16x8 hi, lo;
    wn_split8(v, hi, lo);
    const int64_t g = ((int64_t)(J - 1 - j) * k8b + (o0 >> 3) + q) * p.ldb + i;
    bs[g] = hi;
    bs[plane_b + g] = lo;
  }
}

__global__ __launch_bounds__(256) void wn_split_both_kernel(const dv3_wn_desc p, wn_bf16x8* __restrict__ fs,
                                                            wn_bf16x8* __restrict__ bs, uint32_t* range_ctr) {
  extern __shared__ float tile[];  // [32][32*J+1]
  wn_split_both_block(p, fs, bs, blockIdx.x, blockIdx.y, tile, range_ctr);
}

// ---- every weight-normed Conv1d / Linear layer of a model in TWO launches (the per-layer form costs two small
// launches per layer and step: 84 launches, ~0.6 ms of a 18 ms step).  `tab` lives in device memory, is built once
// (parameters, scales and images are views of fixed buffers) and lists the layers back to back; block b of the
// grid serves layer l = the last one with first_block[l] <= b.
__device__ __forceinline__ int wn_find_layer(const int32_t* __restrict__ first, int n, int b) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__global__ __launch_bounds__(256) void wn_inv_norm_multi_kernel(const dv3_wn_multi_entry* __restrict__ tab,
                                                                const int32_t* __restrict__ first_row, int n) {
  __shared__ float red[4];
  const int l = wn_find_layer(first_row, n, blockIdx.x);
  const dv3_wn_desc& d = tab[l].d;
  wn_inv_norm_row(d.v, d.g, d.scale, d.I * d.J, blockIdx.x - first_row[l], red);
}
__global__ __launch_bounds__(256) void wn_split_both_multi_kernel(const dv3_wn_multi_entry* __restrict__ tab,
                                                                  const int32_t* __restrict__ first_block, int n,
                                                                  uint32_t* range_ctr) {
  extern __shared__ float tile[];
  const int l = wn_find_layer(first_block, n, blockIdx.x);
  const dv3_wn_multi_entry e = tab[l];
  const int b = blockIdx.x - first_block[l];
  const int nbx = (e.d.O + 31) / 32;
  wn_split_both_block(e.d, reinterpret_cast<wn_bf16x8*>(e.fwd_split), reinterpret_cast<wn_bf16x8*>(e.bwd_split),
                      b % nbx, b / nbx, tile, range_ctr);
}

__global__ void zero_kernel(float* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = 0.f;
}

// ---- backward ------------------------------------------------------------------------
// one block per normalised row r (o, or i when transposed).  dW row gathered from the slabs
// into LDS, dot with v, then dv / dg.
__device__ __forceinline__ void wn_bwd_row(const dv3_wn_bwd_desc& p, const int r, const int nrows, float* dw, float* red,
                                           const int vec4_ok) {
  const int O = p.O, I = p.I, J = p.J;
  const int len = p.transposed ? O * J : I * J;
  const float* vrow = p.v + (int64_t)r * len;
  float dot = 0.f;
  // gather: walk the slabs in THEIR order (for a Conv1d row: J runs of I contiguous floats), several slab
  // loads in flight per thread; the row lands in LDS in the parameter's (i, j) order
  // 16-byte form (round 6): a thread sums four consecutive i of every slab, so a 768 / 1536-long row is one trip with all
  // its slab loads in flight instead of three / six dependent trips of 4-byte loads.  Per element the order of the
  // additions is the scalar loop's (partial k goes to accumulator k % 8; the same tree at the end): the same bits.
  const bool vec4 = !p.transposed && (I & 3) == 0 && (p.ldo & 3) == 0 && (p.slab_ss & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(p.slabs) & 15) == 0) && vec4_ok;
  if (vec4) {
    const int I4 = I >> 2;
    for (int q = threadIdx.x; q < J * I4; q += 256) {
      const int j = q / I4, i4 = q - j * I4;
      const f32x4* __restrict__ sp = reinterpret_cast<const f32x4*>(p.slabs + ((int64_t)j * O + r) * p.ldo) + i4;
      const int64_t ss4 = p.slab_ss >> 2;
      f32x4 s[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      int k = 0;
      for (; k + 8 <= p.n_slabs; k += 8) {
        f32x4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = sp[(int64_t)(k + u) * ss4];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += a[u];
      }
      for (; k + 4 <= p.n_slabs; k += 4) {
        f32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = sp[(int64_t)(k + u) * ss4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += a[u];
      }
      for (; k < p.n_slabs; ++k) s[0] += sp[(int64_t)k * ss4];
      const f32x4 tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
#pragma unroll
      for (int e = 0; e < 4; ++e) dw[(i4 * 4 + e) * J + j] = tot[e];
    }
  } else
  for (int q = threadIdx.x; q < len; q += 256) {
    int64_t off;
    int idx;
    if (!p.transposed) {
      const int j = q / I, i = q - j * I;
      off = ((int64_t)j * O + r) * p.ldo + i;  // slab[j][o=r][i]: consecutive threads, consecutive i
      idx = i * J + j;
    } else {
      const int o = q / J, j = q % J;
      off = ((int64_t)j * O + o) * p.ldo + r;  // slab[0][j*O+o][i=r]
      idx = q;
    }
    // eight partial sums in flight per element (round 4: with the [J][M][S][ldo] slab layout the S partial rows of a
    // weight row are contiguous, so deeper queues pay; with slabs 1.5 MB apart they measured slower)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    const float* __restrict__ sp = p.slabs + off;
    int k = 0;
    for (; k + 8 <= p.n_slabs; k += 8) {
      const float a0 = sp[(int64_t)k * p.slab_ss], a1 = sp[(int64_t)(k + 1) * p.slab_ss];
      const float a2 = sp[(int64_t)(k + 2) * p.slab_ss], a3 = sp[(int64_t)(k + 3) * p.slab_ss];
      const float a4 = sp[(int64_t)(k + 4) * p.slab_ss], a5 = sp[(int64_t)(k + 5) * p.slab_ss];
      const float a6 = sp[(int64_t)(k + 6) * p.slab_ss], a7 = sp[(int64_t)(k + 7) * p.slab_ss];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3; s4 += a4; s5 += a5; s6 += a6; s7 += a7;
    }
    for (; k + 4 <= p.n_slabs; k += 4) {
      s0 += sp[(int64_t)k * p.slab_ss];
      s1 += sp[(int64_t)(k + 1) * p.slab_ss];
      s2 += sp[(int64_t)(k + 2) * p.slab_ss];
      s3 += sp[(int64_t)(k + 3) * p.slab_ss];
    }
    for (; k < p.n_slabs; ++k) s0 += sp[(int64_t)k * p.slab_ss];
    dw[idx] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < len; idx += 256) dot += dw[idx] * vrow[idx];
  dot = dv3_wave_sum(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  dot = red[0] + red[1] + red[2] + red[3];
  // bias gradient (sum of the per-batch partials) rides along: block k reduces bias channel k
  if (p.bias_part && p.dbias) {
    for (int o = r; o < O; o += nrows) {
      float bs = 0.f;
      if (p.bias_part_t) {   // [O][n_part] (the input-gradient tail's 32-column partial sums): contiguous per channel
        const float* __restrict__ bp = p.bias_part + (int64_t)o * p.n_part;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        int k = threadIdx.x;
        for (; k + 768 < p.n_part; k += 1024) { b0 += bp[k]; b1 += bp[k + 256]; b2 += bp[k + 512]; b3 += bp[k + 768]; }
        for (; k < p.n_part; k += 256) b0 += bp[k];
        bs = (b0 + b1) + (b2 + b3);
      } else
      for (int k = threadIdx.x; k < p.n_part; k += 256) bs += p.bias_part[(int64_t)k * O + o];
      bs = dv3_wave_sum(bs);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bs;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float tot = red[0] + red[1] + red[2] + red[3];
        p.dbias[o] = p.accumulate ? p.dbias[o] + tot : tot;
      }
      __syncthreads();
    }
  }
  float* dvrow = p.dv + (int64_t)r * len;
  if (p.g) {
    const float sc = p.scale[r], gg = p.g[r];
    const float dg = dot * sc;
    if (threadIdx.x == 0) p.dg[r] = p.accumulate ? p.dg[r] + dg : dg;
    const float c1 = gg * sc, c2 = gg * sc * sc * dg;
    for (int idx = threadIdx.x; idx < len; idx += 256) {
      const float val = c1 * dw[idx] - c2 * vrow[idx];
      dvrow[idx] = p.accumulate ? dvrow[idx] + val : val;
    }
  } else {
    for (int idx = threadIdx.x; idx < len; idx += 256) dvrow[idx] = p.accumulate ? dvrow[idx] + dw[idx] : dw[idx];
  }
}

__global__ __launch_bounds__(256) void wn_bwd_kernel(const dv3_wn_bwd_desc p, const int vec4_ok) {
  extern __shared__ float dw[];  // [len]
  __shared__ float red[4];
  wn_bwd_row(p, blockIdx.x, gridDim.x, dw, red, vec4_ok);
}

// several layers in one launch: block -> (layer, row); the descriptors are kernel arguments (include/dv3hip.h)
struct WnBwdMultiArgs {
  dv3_wn_bwd_desc d[DV3_WN_BWD_MULTI_MAX];
  int32_t first_row[DV3_WN_BWD_MULTI_MAX + 1];
  int32_t n;
  int32_t vec4_ok;
};
__global__ __launch_bounds__(256) void wn_bwd_multi_kernel(const WnBwdMultiArgs a) {
  extern __shared__ float dw[];
  __shared__ float red[4];
  const int blk = blockIdx.x;
  int l = 0;
#pragma unroll
  for (int k = 1; k < DV3_WN_BWD_MULTI_MAX; ++k)
    if (k < a.n && a.first_row[k] <= blk) l = k;
  // (uniform per workgroup: the descriptor is read from the argument segment with scalar loads)
  wn_bwd_row(a.d[l], blk - a.first_row[l], a.first_row[l + 1] - a.first_row[l], dw, red, a.vec4_ok);
}

// dbias[o] = sum_p part[p][o]
__global__ void bias_reduce_kernel(const float* __restrict__ part, int n_part, int n,
                                   float* __restrict__ out) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  float s = 0.f;
  for (int k = 0; k < n_part; ++k) s += part[(int64_t)k * n + o];
  out[o] = s;
}

}  // namespace

extern "C" int dv3_weight_norm_pack_f32(const dv3_wn_desc* d, void* stream) {
  DV3_REQUIRE(d && d->v && d->scale && d->fwd_pack, "wn_pack: null pointer");
  DV3_REQUIRE(d->O > 0 && d->I > 0 && d->J > 0, "wn_pack: bad dims");
  DV3_REQUIRE((d->lda & 3) == 0, "wn_pack: lda must be a multiple of 4");
  hipStream_t st = (hipStream_t)stream;
  const dv3_wn_desc p = *d;
  if (!d->transposed) {
    DV3_REQUIRE(d->glu_cg == 0 || (2 * d->glu_cg == d->O && d->a_half >= d->glu_cg &&
                                   d->lda >= d->a_half + d->glu_cg),
                "wn_pack: bad GLU layout");
    DV3_REQUIRE(d->glu_cg > 0 || d->lda >= d->O, "wn_pack: lda < O");
    hipLaunchKernelGGL(wn_inv_norm_kernel, dim3(d->O), dim3(256), 0, st, d->v, d->g, d->scale,
                       d->I * d->J);
    // pads must be zero: clear the packed buffer when it has pad columns
    const bool has_pad = d->glu_cg > 0 ? (d->a_half != d->glu_cg || d->lda != 2 * d->glu_cg)
                                       : (d->lda != d->O);
    if (has_pad) {
      const int64_t n = (int64_t)d->J * d->I * d->lda;
      hipLaunchKernelGGL(zero_kernel, dim3((unsigned)dv3_cdiv64(n, 256 * 8)), dim3(256), 0, st,
                         d->fwd_pack, n);
    }
    const size_t lds = (size_t)32 * (32 * d->J + 1) * 4;
    hipLaunchKernelGGL(wn_pack_fwd_kernel, dim3(dv3_cdiv(d->O, 32), dv3_cdiv(d->I, 32)), dim3(256),
                       lds, st, p);
    if (d->bwd_pack) {
      DV3_REQUIRE((d->ldb & 3) == 0 && d->ldb >= d->I, "wn_pack: bad ldb");
      hipLaunchKernelGGL(wn_pack_bwd_kernel, dim3(d->O), dim3(256), 0, st, p);
    }
  } else {
    DV3_REQUIRE(d->lda >= d->J * d->O, "wn_pack(T): lda < J*O");
    if (d->bwd_pack) DV3_REQUIRE((d->ldb & 3) == 0 && d->ldb >= d->I, "wn_pack(T): bad ldb");
    hipLaunchKernelGGL(wn_inv_norm_kernel, dim3(d->I), dim3(256), 0, st, d->v, d->g, d->scale,
                       d->O * d->J);
    if (d->bwd_pack && d->ldb != d->I) {
      const int64_t n = (int64_t)d->J * d->O * d->ldb;
      hipLaunchKernelGGL(zero_kernel, dim3((unsigned)dv3_cdiv64(n, 256 * 8)), dim3(256), 0, st,
                         d->bwd_pack, n);
    }
    hipLaunchKernelGGL(wn_pack_transposed_kernel, dim3(d->I), dim3(256), 0, st, p);
  }
  return dv3_check_launch("weight_norm_pack_f32");
}

extern "C" int dv3_weight_norm_split_pack_bf16(const dv3_wn_desc* d, uint16_t* fwd_split,
                                               uint16_t* bwd_split, void* stream) {
  DV3_REQUIRE(d && d->v && d->scale && fwd_split, "wn_split_pack: null pointer");
  DV3_REQUIRE(d->O > 0 && d->I > 0 && d->J > 0 && !d->transposed, "wn_split_pack: bad dims / transposed layer");
  DV3_REQUIRE((d->lda & 3) == 0 && (!bwd_split || ((d->ldb & 3) == 0 && d->ldb >= d->I)), "wn_split_pack: bad lda/ldb");
  DV3_REQUIRE(d->glu_cg == 0 || (2 * d->glu_cg == d->O && d->a_half >= d->glu_cg &&
                                 d->lda >= d->a_half + d->glu_cg), "wn_split_pack: bad GLU layout");
  DV3_REQUIRE(d->glu_cg > 0 || d->lda >= d->O, "wn_split_pack: lda < O");
  DV3_REQUIRE(d->fwd_dtype == DV3_SPLIT_DTYPE_BF16 || d->fwd_dtype == DV3_SPLIT_DTYPE_F16, "wn_split_pack: bad fwd_dtype");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wn_inv_norm_kernel, dim3(d->O), dim3(256), 0, st, d->v, d->g, d->scale, d->I * d->J);
  const size_t lds = (size_t)32 * (32 * d->J + 1) * 4;
  DV3_REQUIRE(lds <= 64 * 1024, "wn_split_pack: too many taps");
  hipLaunchKernelGGL(wn_split_both_kernel, dim3(dv3_cdiv(d->O, 32), dv3_cdiv(d->I, 32)), dim3(256), lds, st,
                     *d, reinterpret_cast<wn_bf16x8*>(fwd_split), reinterpret_cast<wn_bf16x8*>(bwd_split), dv3_range_ctr());
  return dv3_check_launch("weight_norm_split_pack_bf16");
}

extern "C" int dv3_weight_norm_split_pack_multi(const dv3_wn_multi_entry* table_dev, const int32_t* first_row_dev,
                                               const int32_t* first_block_dev, int32_t n_layers, int32_t total_rows,
                                               int32_t total_blocks, int32_t max_taps, void* stream) {
  DV3_REQUIRE(table_dev && first_row_dev && first_block_dev, "wn_split_pack_multi: null pointer");
  DV3_REQUIRE(n_layers > 0 && total_rows > 0 && total_blocks > 0 && max_taps > 0, "wn_split_pack_multi: bad counts");
  const size_t lds = (size_t)32 * (32 * max_taps + 1) * 4;
  DV3_REQUIRE(lds <= 64 * 1024, "wn_split_pack_multi: too many taps");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wn_inv_norm_multi_kernel, dim3(total_rows), dim3(256), 0, st, table_dev, first_row_dev, n_layers);
  hipLaunchKernelGGL(wn_split_both_multi_kernel, dim3(total_blocks), dim3(256), lds, st, table_dev, first_block_dev, n_layers,
                     dv3_range_ctr());
  return dv3_check_launch("weight_norm_split_pack_multi");
}

static int wn_bwd_check(const dv3_wn_bwd_desc* d, int* rows, size_t* lds) {
  DV3_REQUIRE(d && d->slabs && d->v && d->dv, "wn_bwd: null pointer");
  DV3_REQUIRE(!d->g || (d->scale && d->dg), "wn_bwd: g given without scale/dg");
  DV3_REQUIRE(d->O > 0 && d->I > 0 && d->J > 0 && d->n_slabs > 0, "wn_bwd: bad dims");
  *rows = d->transposed ? d->I : d->O;
  const int len = d->transposed ? d->O * d->J : d->I * d->J;
  *lds = (size_t)len * 4;
  DV3_REQUIRE(*lds <= 64 * 1024, "wn_bwd: row too long (%d)", len);
  return DV3_OK;
}

extern "C" int dv3_weight_norm_bwd_f32(const dv3_wn_bwd_desc* d, void* stream) {
  int rows = 0;
  size_t lds = 0;
  const int rc = wn_bwd_check(d, &rows, &lds);
  if (rc != DV3_OK) return rc;
  hipLaunchKernelGGL(wn_bwd_kernel, dim3(rows), dim3(256), lds, (hipStream_t)stream, *d, g_wn_bwd_vec4);
  return dv3_check_launch("weight_norm_bwd_f32");
}

extern "C" int dv3_weight_norm_bwd_multi(const dv3_wn_bwd_desc* descs, int32_t n, void* stream) {
  DV3_REQUIRE(descs && n > 0 && n <= DV3_WN_BWD_MULTI_MAX, "wn_bwd_multi: 1..%d descriptors", DV3_WN_BWD_MULTI_MAX);
  WnBwdMultiArgs a;
  a.vec4_ok = g_wn_bwd_vec4;
  size_t lds_max = 0;
  int total = 0;
  for (int l = 0; l < n; ++l) {
    int rows = 0;
    size_t lds = 0;
    const int rc = wn_bwd_check(&descs[l], &rows, &lds);
    if (rc != DV3_OK) return rc;
    for (int k = 0; k < l; ++k)
      DV3_REQUIRE(descs[k].dv != descs[l].dv, "wn_bwd_multi: entries %d and %d write the same gradient", k, l);
    if (lds > lds_max) lds_max = lds;
    a.d[l] = descs[l];
    a.first_row[l] = total;
    total += rows;
  }
  for (int l = n; l <= DV3_WN_BWD_MULTI_MAX; ++l) a.first_row[l] = total;
  for (int l = n; l < DV3_WN_BWD_MULTI_MAX; ++l) a.d[l] = descs[0];
  a.n = n;
  hipLaunchKernelGGL(wn_bwd_multi_kernel, dim3((unsigned)total), dim3(256), lds_max, (hipStream_t)stream, a);
  return dv3_check_launch("weight_norm_bwd_multi");
}
