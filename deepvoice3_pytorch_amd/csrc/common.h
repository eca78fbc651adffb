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

// ---- device helpers -----------------------------------------------------------------
__device__ __forceinline__ float dv3_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

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
