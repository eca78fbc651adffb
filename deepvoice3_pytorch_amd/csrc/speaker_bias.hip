// Per-frame speaker biases of a block of Conv1dGLU layers in one launch, and their backward in three.
//
// Reference (modules.py:158-162, deepvoice3.py:78-81,292-294): every Conv1dGLU of a multi-speaker model adds
// softsign(speaker_proj(speaker_embed_btc)) to the `a` half of its pre-activation, where speaker_embed_btc is the
// (B, 16) speaker embedding expanded over time and -- in training -- dropped PER FRAME, once per block (encoder /
// decoder / converter), so the bias really is a (B, C, T) tensor there.  As one weight-normed Linear(16 -> C) per layer
// on the tap-GEMM path that is, per layer, a K = 16 GEMM, a softsign backward, an M = 16 input-gradient GEMM, a
// K = B*T weight-gradient GEMM, a weight-norm backward and (bf16 storage) a c8 -> fp32 conversion: six launches of
// 15-30 us that are all launch / latency bound (deepvoice3_vctk: ~150 launches and 3.6 of 16.6 ms of kernel time per
// step, profiles/r04c_vctk_kernel_stats.csv).  K = 16 is not matrix-core work: here the layers of a block are ONE
// launch of plain fp32 FMAs bounded by the HBM traffic of the bias tensors themselves.
//
//   forward   out_l[b, c, t] = softsign(b_l[c] + sum_e W_l[c, e] * emb[b, e, t]),   W_l = g_l * v_l / ||v_l||   (rows)
//   backward  gp = dout_l * (1 - |out_l|)^2;  db_l = sum_{b,t} gp;  dW_l = sum_{b,t} gp (x) emb;
//             d emb = sum_l W_l^T gp;  (dv_l, dg_l) = weight-norm backward of dW_l          -- deterministic sums
#include "common.h"

namespace {

constexpr int EM = 16;        // speaker_embed_dim of the presets (hparams.py: speaker_embed_dim = 16); smaller E is zero-padded
constexpr int NT = 256;       // threads = frames per workgroup
constexpr int CB = 32;        // rows per backward tile

struct SpkArgs {
  dv3_spk_desc d;
  dv3_spk_layer layer[DV3_SPK_MAX_LAYERS];
  int row0[DV3_SPK_MAX_LAYERS + 1];   // first row of layer l in the concatenation of all layers' rows
};

// W rows [c0, c0 + n) of a layer, normalised, zero-padded to EM columns, into LDS as [n][EM]
__device__ __forceinline__ void stage_w(const dv3_spk_layer& L, int E, int c0, int n, float* Ws, int tid) {
  for (int i = tid; i < n; i += NT) {
    const int c = c0 + i;
    float v[EM];
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < EM; ++e) {
      v[e] = e < E ? L.v[(int64_t)c * E + e] : 0.f;
      ss += v[e] * v[e];
    }
    const float s = L.g ? L.g[c] / sqrtf(ss) : 1.0f;
#pragma unroll
    for (int e = 0; e < EM; ++e) Ws[i * EM + e] = v[e] * s;
  }
}

__global__ __launch_bounds__(NT) void spk_fwd_kernel(const SpkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const dv3_spk_desc& d = a.d;
  const dv3_spk_layer& L = a.layer[blockIdx.z];
  const int tid = threadIdx.x, b = blockIdx.y, t = blockIdx.x * NT + tid;
  const int C = L.C, T = d.T;
  float* Ws = smem;             // [C][EM]
  float* Bs = smem + C * EM;    // [C]
  stage_w(L, d.E, 0, C, Ws, tid);
  for (int c = tid; c < C; c += NT) Bs[c] = L.bias ? L.bias[c] : 0.f;
  __syncthreads();
  float ev[EM];
#pragma unroll
  for (int e = 0; e < EM; ++e) ev[e] = (e < d.E && t < T) ? d.e[(int64_t)b * d.e_bs + (int64_t)e * d.e_rs + t] : 0.f;
  if (t >= T) return;
  float* out = L.out + (int64_t)b * C * T + t;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const f32x4* w4 = reinterpret_cast<const f32x4*>(Ws + c * EM);
    float p[EM / 4];                     // four independent chains of four FMAs
#pragma unroll
    for (int q = 0; q < EM / 4; ++q) {
      const f32x4 w = w4[q];
      p[q] = w[0] * ev[4 * q] + w[1] * ev[4 * q + 1] + w[2] * ev[4 * q + 2] + w[3] * ev[4 * q + 3];
    }
    const float acc = Bs[c] + ((p[0] + p[1]) + (p[2] + p[3]));
    out[(int64_t)c * T] = acc * __builtin_amdgcn_rcpf(1.0f + fabsf(acc));      // v_rcp_f32: 1 ulp
  }
}

// backward, pass 1: one workgroup per (frame chunk of 256, batch item, layer); tiles of CB = 32 rows x 256 frames of gp go
// through LDS and feed two small exact-fp32 matrix products on the matrix cores (v_mfma_f32_32x32x2_f32):
//   dW  [32 rows][16 columns + a column of ones = db]  = gp [32 x 256] . emb^T [256 x 17]   wave w: frames 64 w .. 64 w + 63,
//                                                                                            the four partial tiles summed through LDS
//   d emb [16][256 frames] += W^T [16 x 32 rows] . gp [32 x 256]     wave w: its own 64 frames, accumulators live across the tiles
// (The first form did both with vector FMAs fed from LDS: 1 MB of LDS reads per tile and workgroup, 11 us per tile.)
// LDS: gp rows 264 floats apart, the embedding as [frame][17] (conflict-free for the access patterns below); the four
// partial dW tiles go through the gp tile's own storage once it has been consumed: 53 KB, three workgroups per CU.
constexpr int GPL = NT + 8, EVL = EM + 1;
template <int ABL>
__global__ __launch_bounds__(NT) void spk_bwd_kernel(const SpkArgs a, float* __restrict__ part, float* __restrict__ de_l, const int g_prefetch) {
  __shared__ __attribute__((aligned(16))) float Ws[CB * EM];
  __shared__ __attribute__((aligned(16))) float gp[CB * GPL];
  __shared__ __attribute__((aligned(16))) float evs[NT * EVL];
  const dv3_spk_desc& d = a.d;
  const int l = blockIdx.z;
  const dv3_spk_layer& L = a.layer[l];
  const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * NT, t = t0 + tid;
  const int C = L.C, T = d.T, nT = gridDim.x;
  const int n_blocks = d.B * nT, blk = b * nT + blockIdx.x;
#pragma unroll
  for (int e = 0; e < EM; ++e) evs[tid * EVL + e] = (e < d.E && t < T) ? d.e[(int64_t)b * d.e_bs + (int64_t)e * d.e_rs + t] : 0.f;
  evs[tid * EVL + EM] = t < T ? 1.0f : 0.f;            // the column of ones: dW's column 16 is db
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  f32x16 dacc[2];                                      // d emb of this wave's 64 frames: [frame block][C/D layout, rows = e]
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[jb][r] = 0.f;
  // The tile's loads are issued one tile AHEAD (round 6): right after tile k has gone from the registers into LDS, so that
  // they are in flight while tile k's two matrix products and its partial-sum pass run -- the pass used to wait out a
  // full HBM round trip per tile (1.4 TB/s in the step).  Same values, same order: bit-identical gradients.
  constexpr int RW = CB / (NT / 64);
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  f32x4 yv[RW], dv_[RW];
  u16x8 raw[CB / 8];
  const int f0 = 4 * lane;
  auto load_tile = [&](int c0) __attribute__((always_inline)) {
    const int n = min(CB, C - c0);
    const bool c8 = L.dout_c8p != 0;
    // wave w loads rows w, w + 4, ...: a lane takes four consecutive frames of a row as ONE 16-byte load per tensor (rows
    // are only 4-byte aligned: T is arbitrary), so the sixteen loads of a thread are all in flight at once -- as 64
    // four-byte loads the compiler issued them in small batches, one memory round trip each
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int i = wave + r * (NT / 64), c = c0 + i;
      const float* po = L.out + ((int64_t)b * C + c) * T + t0 + f0;
      const float* pd = L.dout + (int64_t)b * L.dout_bs + (int64_t)c * L.dout_rs + t0 + f0;
      if (i < n && ABL != 3 && t0 + f0 + 3 < T) {
        yv[r] = *reinterpret_cast<const f32x4u*>(po);
        if (!c8) dv_[r] = *reinterpret_cast<const f32x4u*>(pd);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool ok = i < n && ABL != 3 && t0 + f0 + k < T;
          yv[r][k] = ok ? po[k] : 1.0f;
          if (!c8) dv_[r][k] = ok ? pd[k] : 0.f;
        }
      }
      if (c8) dv_[r] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};      // the tile first holds (1 - |out|)^2 alone
    }
    // c8 gradient (a bf16 tensor [B][c8p][T][8]): this thread's frame of the tile's four channel groups, 16 bytes each
    if (c8) {
      const unsigned short* base = reinterpret_cast<const unsigned short*>(L.dout);
#pragma unroll
      for (int g = 0; g < CB / 8; ++g) {
        const bool ok = c0 + 8 * g < C && t < T && ABL != 3;
        raw[g] = ok ? *reinterpret_cast<const u16x8*>(base + (((int64_t)b * L.dout_c8p + (c0 >> 3) + g) * T + t) * 8)
                    : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  };
  load_tile(0);
  for (int c0 = 0; c0 < C; c0 += CB) {
    const int n = min(CB, C - c0);
    if (!g_prefetch && c0 > 0) load_tile(c0);
    __syncthreads();                         // the previous tile's readers (gp, Ws, red) are done
    stage_w(L, d.E, c0, n, Ws, tid);
    for (int i = n * EM + tid; i < CB * EM; i += NT) Ws[i] = 0.f;       // rows past the layer's last: no contribution
    const bool c8 = L.dout_c8p != 0;
    {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int i = wave + r * (NT / 64);
        f32x4 g4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float sg = 1.0f - fabsf(yv[r][k]);
          g4[k] = dv_[r][k] * (sg * sg);        // same association as the c8 path below (factor first)
        }
        *reinterpret_cast<f32x4*>(gp + i * GPL + f0) = g4;
      }
      if (c8) {
        __syncthreads();                     // the factor tile is complete
#pragma unroll
        for (int g = 0; g < CB / 8; ++g)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = 8 * g + j;
            const float dv1 = __builtin_bit_cast(float, (unsigned)raw[g][j] << 16);
            gp[i * GPL + tid] = i < n ? gp[i * GPL + tid] * dv1 : 0.f;     // only this thread touches (row, its frame)
          }
      }
    }
    if (g_prefetch && c0 + CB < C) load_tile(c0 + CB);        // in flight during this tile's products
    __syncthreads();
    // d emb: D[e][frame] += sum_rows W[row][e] * gp[row][frame];  A[i = e][k = row], B[k = row][j = frame]
    if (ABL != 2) {
#pragma unroll
      for (int k0 = 0; k0 < CB; k0 += 2) {
        const float av = l31 < EM ? Ws[(k0 + lhi) * EM + l31] : 0.f;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          const float bv = gp[(k0 + lhi) * GPL + wave * 64 + jb * 32 + l31];
          dacc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, dacc[jb], 0, 0, 0);
        }
      }
    }
    // dW (+ db): P[row][col] = sum over this wave's 64 frames of gp[row][frame] * emb[frame][col];  A[i = row][k = frame],
    // B[k = frame][j = col], col 16 = ones
    if (ABL != 1) {
      f32x16 wacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
#pragma unroll 8
      for (int k0 = 0; k0 < 64; k0 += 2) {
        const int f = wave * 64 + k0 + lhi;
        const float av = gp[l31 * GPL + f];
        const float bv = l31 <= EM ? evs[f * EVL + l31] : 0.f;
        wacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, wacc, 0, 0, 0);
      }
      __syncthreads();                       // every wave has read the gp tile for both products
      float* red = gp;                       // [wave][16 registers][64 lanes] = 16 KB of the tile's 33 KB
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = wacc[r];
      __syncthreads();
      // C/D layout: register r of lane (l31, lhi) is P[row = (r & 3) + 8 (r >> 2) + 4 lhi][col = l31]; 32 rows x 17 columns
      // = 544 sums of the four waves' tiles, in wave order
      for (int idx = tid; idx < CB * (EM + 1); idx += NT) {
        const int row = idx / (EM + 1), col = idx - row * (EM + 1);
        const int r = (row & 3) + 4 * (row >> 3), hi = (row >> 2) & 1;
        const float* q = red + r * 64 + hi * 32 + col;
        const float v = ((q[0] + q[16 * 64]) + q[2 * 16 * 64]) + q[3 * 16 * 64];
        if (row < n) part[(((int64_t)a.row0[l] + c0 + row) * n_blocks + blk) * (EM + 1) + col] = v;
      }
    }
  }
  // d emb out of the C/D layout: rows = e (registers 0..3 and 4..7 of each half: e = (r & 3) + 8 (r >> 2) + 4 lhi < 16)
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int tt = t0 + wave * 64 + jb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int e = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (tt < T && e < d.E) de_l[(((int64_t)l * d.B + b) * d.E + e) * T + tt] = dacc[jb][r];
    }
  }
}

// pass 2: four rows of the concatenated layers per workgroup, 64 threads per row (4 block groups x 16 columns): sum the
// row's partials (contiguous: [block][EM + 1]) in a fixed order, then the weight-norm backward of the row
// (dg = dW . v / ||v||, dv = g / ||v|| (dW - v (dW . v) / ||v||^2)), += into the gradients
__global__ __launch_bounds__(256) void spk_finish_kernel(const SpkArgs a, const float* __restrict__ part, int n_blocks) {
  const dv3_spk_desc& d = a.d;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.row0[d.n_layers]) return;
  const int e = lane & 15, grp = lane >> 4;
  int l = 0;
  while (row >= a.row0[l + 1]) ++l;
  const dv3_spk_layer& L = a.layer[l];
  const int c = row - a.row0[l];
  const float* p = part + (int64_t)row * n_blocks * (EM + 1);
  float dw = 0.f, db = 0.f;
  for (int k = grp; k < n_blocks; k += 4) {
    dw += p[(int64_t)k * (EM + 1) + e];
    if (e == 0) db += p[(int64_t)k * (EM + 1) + EM];
  }
  dw += __shfl_xor(dw, 16, 64);
  dw += __shfl_xor(dw, 32, 64);
  db += __shfl_xor(db, 16, 64);
  db += __shfl_xor(db, 32, 64);
  const float v = e < d.E ? L.v[(int64_t)c * d.E + e] : 0.f;
  float vv = v * v, dwv = dw * v;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    vv += __shfl_xor(vv, off, 16);
    dwv += __shfl_xor(dwv, off, 16);
  }
  if (grp != 0) return;
  if (e < d.E) {
    if (L.g) {
      const float rn = 1.0f / sqrtf(vv), gg = L.g[c];
      L.dv[(int64_t)c * d.E + e] += gg * rn * (dw - v * dwv / vv);
      if (e == 0) L.dg[c] += dwv * rn;
    } else {
      L.dv[(int64_t)c * d.E + e] += dw;
    }
  }
  if (e == 0 && L.dbias) L.dbias[c] += db;
}

// pass 3: d emb = sum over the layers (fixed order)
__global__ __launch_bounds__(256) void spk_de_sum_kernel(const float* __restrict__ de_l, float* __restrict__ de, int64_t n, int n_layers) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int l = 0; l < n_layers; ++l) s += de_l[(int64_t)l * n + i];
  de[i] = s;
}

int fill_args(SpkArgs& a, const dv3_spk_desc* d, const dv3_spk_layer* layers, bool bwd) {
  DV3_REQUIRE(d && layers, "speaker_bias: NULL descriptor");
  DV3_REQUIRE(d->n_layers >= 1 && d->n_layers <= DV3_SPK_MAX_LAYERS, "speaker_bias: 1..%d layers per call", DV3_SPK_MAX_LAYERS);
  DV3_REQUIRE(d->E >= 1 && d->E <= EM, "speaker_bias: speaker_embed_dim 1..%d", EM);
  DV3_REQUIRE(d->B >= 1 && d->T >= 1 && d->B <= 65535 && d->e, "speaker_bias: bad shape");
  a.d = *d;
  a.row0[0] = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    const dv3_spk_layer& L = layers[l];
    DV3_REQUIRE(L.C >= 1 && L.C <= 2048 && L.v && L.out, "speaker_bias: layer %d: bad arguments", l);
    if (bwd) DV3_REQUIRE(L.dout && L.dv && (L.dg || !L.g), "speaker_bias: layer %d: backward needs dout, dv (, dg)", l);
    if (bwd && L.dout_c8p) DV3_REQUIRE(L.dout_c8p * 8 >= L.C, "speaker_bias: layer %d: c8 gradient with fewer channels than C", l);
    a.layer[l] = L;
    a.row0[l + 1] = a.row0[l] + L.C;
  }
  return DV3_OK;
}

}  // namespace

int g_spk_prefetch = 1;   // dv3_debug_set(54, v): the backward's tile loads one tile ahead (0 = at the top of the tile)
int g_spk_abl = 0;   // dv3_debug_set(28, v): timing-only ablations of the backward (1 no dW, 2 no d emb, 3 no loads); make EXP=1

extern "C" int dv3_speaker_bias_fwd_f32(const dv3_spk_desc* d, const dv3_spk_layer* layers, void* stream) {
  SpkArgs a;
  const int rc = fill_args(a, d, layers, false);
  if (rc != DV3_OK) return rc;
  int cmax = 0;
  for (int l = 0; l < d->n_layers; ++l) cmax = cmax > layers[l].C ? cmax : layers[l].C;
  const size_t lds = (size_t)cmax * (EM + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)spk_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      dv3_set_error("speaker_bias_fwd: hipFuncSetAttribute failed");
      return DV3_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(spk_fwd_kernel, dim3((d->T + NT - 1) / NT, d->B, d->n_layers), dim3(NT), lds, (hipStream_t)stream, a);
  return dv3_check_launch("speaker_bias_fwd");
}

extern "C" int dv3_speaker_bias_bwd_scratch_floats(const dv3_spk_desc* d, const dv3_spk_layer* layers) {
  if (!d || !layers || d->n_layers < 1 || d->n_layers > DV3_SPK_MAX_LAYERS) return -1;
  int64_t rows = 0;
  for (int l = 0; l < d->n_layers; ++l) rows += layers[l].C;
  const int64_t nT = (d->T + NT - 1) / NT;
  const int64_t n = rows * d->B * nT * (EM + 1) + (int64_t)d->n_layers * d->B * d->E * d->T;
  return n < (1ll << 31) ? (int)n : -1;
}

extern "C" int dv3_speaker_bias_bwd_f32(const dv3_spk_desc* d, const dv3_spk_layer* layers, void* stream) {
  SpkArgs a;
  const int rc = fill_args(a, d, layers, true);
  if (rc != DV3_OK) return rc;
  const int need = dv3_speaker_bias_bwd_scratch_floats(d, layers);
  DV3_REQUIRE(need > 0 && d->scratch && d->scratch_floats >= need && d->de, "speaker_bias_bwd: workspace of %d floats and de needed", need);
  const int nT = (d->T + NT - 1) / NT;
  const int64_t rows = a.row0[d->n_layers];
  float* part = d->scratch;
  float* de_l = d->scratch + rows * d->B * nT * (EM + 1);
  hipStream_t st = (hipStream_t)stream;
#ifdef DV3_EXPERIMENTS
  switch (g_spk_abl) {
    case 1: hipLaunchKernelGGL(spk_bwd_kernel<1>, dim3(nT, d->B, d->n_layers), dim3(NT), 0, st, a, part, de_l, g_spk_prefetch); break;
    case 2: hipLaunchKernelGGL(spk_bwd_kernel<2>, dim3(nT, d->B, d->n_layers), dim3(NT), 0, st, a, part, de_l, g_spk_prefetch); break;
    case 3: hipLaunchKernelGGL(spk_bwd_kernel<3>, dim3(nT, d->B, d->n_layers), dim3(NT), 0, st, a, part, de_l, g_spk_prefetch); break;
    default: hipLaunchKernelGGL(spk_bwd_kernel<0>, dim3(nT, d->B, d->n_layers), dim3(NT), 0, st, a, part, de_l, g_spk_prefetch);
  }
#else
  hipLaunchKernelGGL(spk_bwd_kernel<0>, dim3(nT, d->B, d->n_layers), dim3(NT), 0, st, a, part, de_l, g_spk_prefetch);
#endif
  int rc2 = dv3_check_launch("speaker_bias_bwd");
  if (rc2 != DV3_OK) return rc2;
  hipLaunchKernelGGL(spk_finish_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a, (const float*)part, d->B * nT);
  rc2 = dv3_check_launch("speaker_bias_bwd(finish)");
  if (rc2 != DV3_OK) return rc2;
  const int64_t n = (int64_t)d->B * d->E * d->T;
  hipLaunchKernelGGL(spk_de_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)de_l, d->de, n, d->n_layers);
  return dv3_check_launch("speaker_bias_bwd(de)");
}
