// Measurement stand-in for an RCCL ring all-reduce on ONE GPU (round 6; SURVEY section 8e: one gradient all-reduce per
// step over xGMI -- no multi-GPU node was ever available to a build round, and a world-size-1 collective moves nothing and
// launches no ring kernel).  What a ring all-reduce of S bytes does to the GPU it runs on, as far as the training step
// beside it can tell:
//   * `channels` persistent workgroups of `threads` threads occupy CU slots (registers, wave slots) for its duration;
//   * they stream 2 (n - 1) / n x S bytes out of and into HBM (reduce-scatter + all-gather: every step reads a chunk and
//     writes a chunk);
//   * the duration is set by the links, not by the GPU: 2 (n - 1) / n x S / busbw.
// This kernel does exactly that and nothing else: workgroup w copies its share of `bytes` (16 bytes per thread and
// access, from `src` -- the gradient bucket, read only -- into `scratch`) in chunks, and after every chunk spins on the
// 100 MHz wall clock until its progress matches `bytes_per_us`.  It reduces nothing: the numbers of the step are the
// single-GPU step's.  Not part of the product path (dist.RingStandin, bench.py configs.ddp_standin, scripts/r6_*).
#include "common.h"

namespace {

__global__ __launch_bounds__(512) void ring_standin_kernel(const char* __restrict__ src, int64_t src_bytes, char* __restrict__ scratch,
                                                           int64_t scratch_bytes, int64_t bytes, float bytes_per_us) {
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  const int G = gridDim.x, w = blockIdx.x, NT = blockDim.x;
  const int64_t chunk = (int64_t)NT * 16 * 8;                    // bytes per workgroup and chunk: 8 accesses per thread
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
  int64_t done = 0;                                              // bytes this workgroup has moved
  for (int64_t off = (int64_t)w * chunk; off < bytes; off += (int64_t)G * chunk) {
    u32x4_ v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t o = (off + ((int64_t)i * NT + threadIdx.x) * 16) % (src_bytes - 15);
      v[i] = *reinterpret_cast<const u32x4_*>(src + (o & ~(int64_t)15));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t o = (off + ((int64_t)i * NT + threadIdx.x) * 16) % (scratch_bytes - 15);
      *reinterpret_cast<u32x4_*>(scratch + (o & ~(int64_t)15)) = v[i];
    }
    done += chunk;
    // the links set the pace: all G workgroups together move bytes_per_us
    const float due_us = (float)done * (float)G / bytes_per_us;
    while ((float)(__builtin_amdgcn_s_memrealtime() - t0) * 0.01f < due_us) __builtin_amdgcn_s_sleep(32);
  }
}

}  // namespace

extern "C" int dv3_ring_standin(const void* src, int64_t src_bytes, void* scratch, int64_t scratch_bytes, int64_t bytes,
                                int32_t channels, int32_t threads, float bytes_per_us, void* stream) {
  DV3_REQUIRE(src && scratch && src_bytes >= 64 && scratch_bytes >= 64 && bytes > 0, "ring_standin: bad buffers");
  DV3_REQUIRE(channels >= 1 && channels <= 256 && (threads == 256 || threads == 512), "ring_standin: 1..256 channels of 256 or 512 threads");
  DV3_REQUIRE(bytes_per_us > 0.f && (((uintptr_t)src | (uintptr_t)scratch) & 15) == 0, "ring_standin: bad rate / alignment");
  hipLaunchKernelGGL(ring_standin_kernel, dim3(channels), dim3(threads), 0, (hipStream_t)stream, (const char*)src, src_bytes,
                     (char*)scratch, scratch_bytes, bytes, bytes_per_us);
  return dv3_check_launch("ring_standin");
}
