// Shared device/host helpers for libdv3hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dv3hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DV3_WAVE 64

// ---- host-side error plumbing -------------------------------------------------------
void dv3_set_error(const char* fmt, ...);

#define DV3_REQUIRE(cond, ...)              \
  do {                                      \
    if (!(cond)) {                          \
      dv3_set_error(__VA_ARGS__);           \
      return DV3_EINVAL;                    \
    }                                       \
  } while (0)

static inline int dv3_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dv3_set_error("%s: %s", what, hipGetErrorString(e));
    return DV3_ELAUNCH;
  }
  return DV3_OK;
}

// dv3_debug_get(10 / 11): which kernel variant the last tap-GEMM / wgrad call launched (api.hip)
extern int g_dv3_last_conv, g_dv3_last_wgrad;

static inline int dv3_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t dv3_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- fp16 range guard of the f16x3 mode ----------------------------------------------
// Device address of the library's sticky range-event counter (api.hip: a __device__ word, no allocation): every kernel
// that builds scaled fp16 operand pairs adds the number of 16-byte units in which some |v * 2^s| left the fp16 range.
uint32_t* dv3_range_ctr();

// ---- device helpers -----------------------------------------------------------------
typedef __bf16 dv3_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dv3_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dv3_f16x2 __attribute__((ext_vector_type(2)));
typedef float dv3_f32x2 __attribute__((ext_vector_type(2)));

// Scaled fp16 pair of 8 values a[i] = v * 2^s (include/dv3hip.h, "f16x3"): hi = fp16_rn(clamp(a, +-65504)),
// lo = fp16_rn(a - hi).  The residual is taken from the UNCLAMPED a: inside the range nothing changes; up to
// 2 x 65504 the pair still carries a to fp16 precision; beyond that lo becomes Inf, and a NaN / Inf input gives a
// NaN / Inf lo -- the GEMM output turns non-finite exactly as the fp32 reference's would, instead of training on
// saturated values.  Returns true when some |a| exceeded 65504 (the caller counts it: dv3_note_range).
__device__ __forceinline__ bool dv3_split8_f16(const float (&a)[8], dv3_bf16x8& hi, dv3_bf16x8& lo) {
  dv3_f16x8 h8, l8;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const dv3_f32x2 f = {a[i], a[i + 1]};
    const dv3_f32x2 c = {__builtin_amdgcn_fmed3f(a[i], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(a[i + 1], -65504.f, 65504.f)};
    const dv3_f16x2 h = __builtin_convertvector(c, dv3_f16x2);
    const dv3_f32x2 r = f - __builtin_convertvector(h, dv3_f32x2);
    const dv3_f16x2 l = __builtin_convertvector(r, dv3_f16x2);
    h8[i] = h[0]; h8[i + 1] = h[1];
    l8[i] = l[0]; l8[i + 1] = l[1];
  }
  hi = __builtin_bit_cast(dv3_bf16x8, h8);
  lo = __builtin_bit_cast(dv3_bf16x8, l8);
  const float m0 = fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fabsf(a[2]));
  const float m1 = fmaxf(fmaxf(fabsf(a[3]), fabsf(a[4])), fabsf(a[5]));
  const float m = fmaxf(fmaxf(m0, m1), fmaxf(fabsf(a[6]), fabsf(a[7])));
  return !(m <= 65504.f);
}
// count the units of this wave whose scaled value left the fp16 range (rare path: one ballot per call)
__device__ __forceinline__ void dv3_note_range(uint32_t* ctr, bool bad) {
  if (__builtin_expect(__any(bad), 0)) {
    if (ctr && bad) atomicAdd(ctr, 1u);
  }
}

__device__ __forceinline__ float dv3_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

// One element of the gate backward (autograd of modules.py:157-164 GLU, :224-226 highway): dy already scaled by the
// layer's output factor (d = dy * sqrt(.5) for a residual GLU).  ONE definition for the stand-alone kernel
// (elementwise.hip) and for the input-gradient tails that run it for their producer (conv_common.h), with the
// contraction of its products switched off, so that the two routes agree bit for bit.  The sigmoid is the forward
// tail's (v_exp_f32 + v_rcp_f32).
__device__ __forceinline__ void dv3_gate_deriv(float d, float a, float g, float x, bool glu, float& va, float& vg, float& vr) {
#pragma clang fp contract(off)
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
  const float t = s * (1.0f - s);
  va = d * s;
  if (glu) {
    vg = (d * a) * t;
    vr = d;
  } else {
    vg = (d * (a - x)) * t;
    vr = d * (1.0f - s);
  }
}

// PAIR WORD of v (include/dv3hip.h, dv3_conv_desc.pg_pair): (bf16_rn(v) << 16) | bf16_rn(v - bf16_rn(v)) -- the hi / lo
// operands split8() of the gradient GEMMs builds from v while staging, built once where v is produced.
__device__ __forceinline__ uint32_t dv3_pair_word(float v) {
  const __bf16 h = (__bf16)v;
  const float r = v - (float)h;
  const __bf16 l = (__bf16)r;
  return ((uint32_t)__builtin_bit_cast(uint16_t, h) << 16) | (uint32_t)__builtin_bit_cast(uint16_t, l);
}
__device__ __forceinline__ float dv3_pair_value(uint32_t w) {   // hi + lo as fp32 (2^-17-class)
  return __uint_as_float(w & 0xffff0000u) + __uint_as_float(w << 16);
}
// eight pair words -> the hi and lo 16-byte operand units (element e of a unit = word e): two v_perm_b32 per word pair
__device__ __forceinline__ void dv3_pair_units(const uint32_t (&w)[8], dv3_bf16x8& hi, dv3_bf16x8& lo) {
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  u32x4_ h4, l4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h4[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x07060302u);
    l4[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x05040100u);
  }
  hi = __builtin_bit_cast(dv3_bf16x8, h4);
  lo = __builtin_bit_cast(dv3_bf16x8, l4);
}

// Bijective XCD-aware remap of a 1-D block id (MI355X: block b runs on XCD b % 8).  Gives each
// XCD a contiguous chunk of the logical tile order so tiles sharing an operand panel hit the
// same private L2.
__device__ __forceinline__ int dv3_xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = orig & 7, slot = orig >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

__device__ __forceinline__ float dv3_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float dv3_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// keep-bit lookup: bits laid out [row][row_stride words], bit t&31 of word t>>5
__device__ __forceinline__ bool dv3_keep(const uint32_t* __restrict__ bits, int64_t row,
                                         int rs, int t) {
  return (bits[row * rs + (t >> 5)] >> (t & 31)) & 1u;
}
