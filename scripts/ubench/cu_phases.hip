// Primitive costs inside one 512-thread workgroup per CU on MI355X (gfx950): what a LOAD / COMPUTE phase of the
// bf16x3 tap-GEMM is made of, measured in isolation and next to a partner wave on the same SIMD.
// Developer micro-benchmark, not part of the product or its tests.
//   hipcc --offload-arch=gfx950 -O3 -o cu_phases cu_phases.hip && ./cu_phases
// Each test (template <role of waves 0-3, role of waves 4-7, barrier per iteration>: the timed loops are
// straight-line) runs ITER iterations per wave on 256 workgroups x 8 waves; waves 0-3 ("early", one per SIMD)
// and waves 4-7 ("late", their SIMD partners) may do different things.  s_memtime around the loop, per-iteration
// cycles of wave 0 and wave 4 of workgroup 0 are reported (plus the max over a few workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ITER = 256;
enum Role { IDLE = 0, MFMA24, LDSR16, GLD2, DSW2, VALU96, BARRIER, STAMP, PHASE_L, PHASE_C, PIPE_R, PIPE_ALL, PIPE_M, PIPE_MA, PIPE_RA, SCHED_R,
            GLD8S, GLDS2, LDSR16B, MFMA24G, GLD8NW };

struct Args {
  const f32x4* gsrc;          // >= 256 * 16 KB, L2-resident after the first touch
  float* sink;                // 256 * 512 floats
  unsigned long long* cyc;    // [blocks][8]
};

__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int RE, int RL, int SYNC, int PE = 0, int PL = 0>
__global__ __launch_bounds__(512) void k(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // explicit LDS address space: a volatile generic pointer would compile to flat_load / flat_store
  typedef __attribute__((address_space(3))) volatile bf16x8 lds_unit;
  lds_unit* lds = (lds_unit*)smem_raw;   // 96 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int late = wave >> 2;

  // operands / accumulators
  bf16x8 fa[4], fb[4];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) {
    for (int e = 0; e < 8; ++e) { fa[i][e] = (__bf16)(0.001f * (lane + i + e)); fb[i][e] = (__bf16)(0.002f * (lane - i + e)); }
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  }
  for (int i = tid; i < 96 * 1024 / 16; i += 512) lds[i] = fa[i & 3];
  __syncthreads();
  f32x4 gacc = {0.f, 0.f, 0.f, 0.f};
  float vacc = (float)lane;
  unsigned long long t0 = 0, t1 = 0, st = 0;

  auto mfma24 = [&]() {
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(i + q) & 3], fb[i], acc[i], 0, 0, 0);
  };
  auto ldsr16 = [&](int it) {
    // 16 fragment reads: lanes 0-31 / 32-63 read 512 contiguous bytes each (the kernel's conflict-free pattern)
    const int base = ((it & 1) * 2048 + wave * 256 + (lane >> 5) * 128 + (lane & 31));
    bf16x8 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = lds[base + (r * 64) % 1536 + (r >> 2) * 16];
    wait_lgkm0();
#pragma unroll
    for (int r = 0; r < 16; r += 4) fa[r >> 2] = v[r];           // consume (keeps the reads)
  };
  auto gld2 = [&](int it) {
    // two 16-byte loads per lane inside the workgroup's own 16 KB (L2 hits after the first pass)
    const f32x4* g0 = a.gsrc + (size_t)blockIdx.x * 1024;
    const f32x4 x0 = g0[(tid + it * 64) & 1023];
    const f32x4 x1 = g0[(tid + 512 + it * 64) & 1023];
    wait_vm0();
    gacc += x0 + x1;
  };
  // round 3: eight dword loads through ONE uniform base + 32-bit offsets (the tap-GEMM's activation fetch form)
  float g8acc = 0.f;
  auto gld8s = [&](int it, bool wait) {
    const float* g0 = reinterpret_cast<const float*>(a.gsrc + (size_t)blockIdx.x * 1024);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = g0[((tid + it * 64) & 511) + e * 512];
    if (wait) wait_vm0();
#pragma unroll
    for (int e = 0; e < 8; ++e) g8acc += v[e];
  };
  // two LDS-DMA pieces (global_load_lds_dwordx4: 1 KB per wave-instruction) + vmcnt(0)
  auto glds2 = [&](int it) {
    const f32x4* g0 = a.gsrc + (size_t)blockIdx.x * 1024;
    __attribute__((address_space(3))) void* l0 = (__attribute__((address_space(3))) void*)(smem_raw + 65536 + wave * 2048);
    __builtin_amdgcn_global_load_lds((const void*)(g0 + ((tid + it * 64) & 511)), l0, 16, 0, 0);
    __attribute__((address_space(3))) void* l1 = (__attribute__((address_space(3))) void*)(smem_raw + 65536 + wave * 2048 + 1024);
    __builtin_amdgcn_global_load_lds((const void*)(g0 + ((tid + 512 + it * 64) & 1023)), l1, 16, 0, 0);
    wait_vm0();
  };
  // 16 fragment reads issued back to back (plain loads: one wait at the end), the kernel's real LOAD pattern
  typedef __attribute__((address_space(3))) bf16x8 lds_plain0;
  auto ldsr16b = [&](int it) {
    lds_plain0* lp = (lds_plain0*)smem_raw;
    const int base = ((it & 1) * 2048 + wave * 256 + (lane >> 5) * 128 + (lane & 31));
    bf16x8 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = lp[base + (r * 64) % 1536 + (r >> 2) * 16];
    wait_lgkm0();
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(v[r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) fa[r] = v[4 * r];
  };
  // 24 MFMAs with one idle issue slot (s_nop 7 = 8 cycles) behind each
  auto mfma24g = [&]() {
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(i + q) & 3], fb[i], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7");
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  auto dsw2 = [&](int it) {
    lds[4096 + (it & 1) * 1024 + tid] = fa[0];
    lds[4096 + 2048 + (it & 1) * 1024 + tid] = fb[0];
    wait_lgkm0();
  };
  auto valu96 = [&]() {
#pragma unroll
    for (int q = 0; q < 96; ++q) vacc = vacc * 1.0001f + 0.5f;
  };
  // Intra-wave software pipeline (the shape a one-wave-per-SIMD tap-GEMM needs): 24 MFMAs on the CURRENT fragment
  // set while the 16 ds_read_b128 of the NEXT set (and, ALL: 2 ds_write_b128 + 2 global_load_dwordx4) are issued in
  // the gaps between them; the only wait is at the end of the phase.  Two phases per call so the two register sets
  // swap roles without copies.
  bf16x8 ga[4], gb[4];
  for (int i = 0; i < 4; ++i) { ga[i] = fa[i]; gb[i] = fb[i]; }
  // (raw instructions: the compiler's scheduler clusters volatile LDS loads in front of the MFMAs whatever
  // sched_group_barrier asks for, so the phase is written as inline asm in exactly the intended issue order)
  auto pipe_phase = [&](bf16x8 (&ca)[4], bf16x8 (&cb)[4], bf16x8 (&na)[4], bf16x8 (&nb)[4], int it, auto all_c,
                        auto reads_c, auto agpr_c) {
    constexpr bool ALL = decltype(all_c)::value, READS = decltype(reads_c)::value, AGPR = decltype(agpr_c)::value;
    const unsigned base = (unsigned)((it & 1) * 2048 + wave * 256 + (lane >> 5) * 128 + (lane & 31)) * 16u;
    const unsigned wbase = (unsigned)(4096 + (it & 1) * 1024 + tid) * 16u;
    const f32x4* gp = a.gsrc + (size_t)blockIdx.x * 1024 + ((tid + it * 64) & 511);
    bf16x8 v[16];
    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
#define MF(I, Q)                                                                                                          \
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[I]) : "v"(ca[((I) + (Q)) & 3]), "v"(cb[I])); \
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[I]) : "v"(ca[((I) + (Q)) & 3]), "v"(cb[I]));
#define RD(R)                                                                                                              \
    if constexpr (READS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[R]) : "v"(base), "n"((((R) * 64) % 1536 + ((R) >> 2) * 16) * 16)); \
    else v[R] = ca[(R) & 3];
    // 24 MFMAs; reads R0..R15 behind MFMAs 0..15, then (ALL) two panel stores and two global loads
    MF(0, 0) RD(0) MF(1, 0) RD(1) MF(2, 0) RD(2) MF(3, 0) RD(3)
    MF(0, 1) RD(4) MF(1, 1) RD(5) MF(2, 1) RD(6) MF(3, 1) RD(7)
    MF(0, 2) RD(8) MF(1, 2) RD(9) MF(2, 2) RD(10) MF(3, 2) RD(11)
    MF(0, 3) RD(12) MF(1, 3) RD(13) MF(2, 3) RD(14) MF(3, 3) RD(15)
    MF(0, 4)
    if constexpr (ALL) asm volatile("ds_write_b128 %0, %1" : : "v"(wbase), "v"(ca[0]) : "memory");
    MF(1, 4)
    if constexpr (ALL) asm volatile("ds_write_b128 %0, %1 offset:32768" : : "v"(wbase), "v"(cb[0]) : "memory");
    MF(2, 4)
    if constexpr (ALL) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x0) : "v"(gp) : "memory");
    MF(3, 4)
    if constexpr (ALL) asm volatile("global_load_dwordx4 %0, %1, off offset:4080" : "=v"(x1) : "v"(gp) : "memory");
    MF(0, 5) MF(1, 5) MF(2, 5) MF(3, 5)
#undef MF
#undef RD
    if constexpr (ALL) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (ALL) gacc += x0 + x1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { na[i] = v[4 * i]; nb[i] = v[4 * i + 1]; }
  };
  // The same pipeline with compiler-scheduled instructions: builtin MFMAs, plain (non-volatile) LDS loads of the NEXT
  // phase's eight fragments -- every load feeds an MFMA of the next phase, so none is dead -- and sched_group_barrier
  // requests "one ds_read behind each of the first eight MFMAs".  Two phases per call (sets swap roles).
  typedef __attribute__((address_space(3))) bf16x8 lds_plain;
  lds_plain* ldp = (lds_plain*)smem_raw;
  auto sched_phase = [&](bf16x8 (&ca)[4], bf16x8 (&cb)[4], bf16x8 (&na)[4], bf16x8 (&nb)[4], int it) {
    const int base = ((it & 1) * 2048 + wave * 256 + (lane >> 5) * 128 + (lane & 31));
#pragma unroll
    for (int i = 0; i < 4; ++i) { na[i] = ldp[base + i * 64]; nb[i] = ldp[base + 512 + i * 64 + 16]; }
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[(i + q) & 3], cb[i], acc[i], 0, 0, 0);
#define SG(K) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); if ((K) < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    SG(0) SG(1) SG(2) SG(3) SG(4) SG(5) SG(6) SG(7) SG(8) SG(9) SG(10) SG(11)
    SG(12) SG(13) SG(14) SG(15) SG(16) SG(17) SG(18) SG(19) SG(20) SG(21) SG(22) SG(23)
#undef SG
  };
  auto body = [&](auto role_c, int it) {          // straight-line per role: no dispatch inside the timed loop
    constexpr int ROLE = decltype(role_c)::value;
    if constexpr (ROLE == MFMA24 || ROLE == PHASE_C) mfma24();
    if constexpr (ROLE == LDSR16) ldsr16(it);
    if constexpr (ROLE == GLD2) gld2(it);
    if constexpr (ROLE == DSW2) dsw2(it);
    if constexpr (ROLE == GLD8S) gld8s(it, true);
    if constexpr (ROLE == GLD8NW) gld8s(it, false);
    if constexpr (ROLE == GLDS2) glds2(it);
    if constexpr (ROLE == LDSR16B) ldsr16b(it);
    if constexpr (ROLE == MFMA24G) mfma24g();
    if constexpr (ROLE == VALU96) valu96();
    if constexpr (ROLE == STAMP) st += __builtin_readcyclecounter() & 1;
    if constexpr (ROLE == PHASE_L) { ldsr16(it); dsw2(it); gld2(it); }   // a LOAD phase minus the activation tile
    if constexpr (ROLE == SCHED_R) {                                      // one iteration = TWO phases (48 MFMAs)
      sched_phase(fa, fb, ga, gb, it);
      sched_phase(ga, gb, fa, fb, it + 1);
    }
    if constexpr (ROLE >= PIPE_R && ROLE != SCHED_R) {                    // one iteration = ONE phase (24 MFMAs)
      using all_t = std::integral_constant<bool, ROLE == PIPE_ALL>;
      using rd_t = std::integral_constant<bool, ROLE == PIPE_R || ROLE == PIPE_ALL || ROLE == PIPE_RA>;
      using ag_t = std::integral_constant<bool, ROLE == PIPE_MA || ROLE == PIPE_RA>;
      if (it & 1) pipe_phase(ga, gb, fa, fb, it, all_t{}, rd_t{}, ag_t{});
      else pipe_phase(fa, fb, ga, gb, it, all_t{}, rd_t{}, ag_t{});
    }
    if constexpr (SYNC) { wait_lgkm0(); __builtin_amdgcn_s_barrier(); }
  };

  __syncthreads();
  // static wave priorities (round 3): PE for waves 0-3, PL for their SIMD partners
  if (late) { if constexpr (PL) __builtin_amdgcn_s_setprio(PL); } else { if constexpr (PE) __builtin_amdgcn_s_setprio(PE); }
  t0 = __builtin_readcyclecounter();
  if (late) {
    for (int it = 0; it < ITER; ++it) body(std::integral_constant<int, RL>{}, it);
  } else {
    for (int it = 0; it < ITER; ++it) body(std::integral_constant<int, RE>{}, it);
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) a.cyc[blockIdx.x * 8 + wave] = t1 - t0;
  float s = g8acc + vacc + gacc[0] + gacc[1] + gacc[2] + gacc[3] + (float)(st & 1);
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7] + (float)fa[i][0] + (float)ga[i][1] + (float)gb[i][2];
  a.sink[blockIdx.x * 512 + tid] = s;
}

static const char* kName[] = {"idle", "mfma24", "ldsr16", "gld2", "dsw2", "valu96", "barrier", "stamp", "phaseL", "phaseC", "pipeR", "pipeAll", "asmM(v)", "asmM(a)", "pipeR(a)", "schedR",
                              "gld8s", "glds2", "ldsr16b", "mfma24g", "gld8nw"};

template <int RE, int RL, int SYNC, int PE = 0, int PL = 0>
static void run(const Args& a, std::vector<unsigned long long>& h, int blocks) {
  (void)hipFuncSetAttribute((const void*)k<RE, RL, SYNC, PE, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<RE, RL, SYNC, PE, PL>), dim3(blocks), dim3(512), 96 * 1024, 0, a);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(2); }
  (void)hipMemcpy(h.data(), a.cyc, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long me = 0, ml = 0;
  for (int b = 0; b < blocks; ++b) { if (h[b * 8] > me) me = h[b * 8]; if (h[b * 8 + 4] > ml) ml = h[b * 8 + 4]; }
  printf("%-8s %-8s %d p%d%d | %9.1f %9.1f | %9.1f %9.1f\n", kName[RE], kName[RL], SYNC, PE, PL, h[0] / (double)ITER,
         h[4] / (double)ITER, me / (double)ITER, ml / (double)ITER);
}

int main() {
  const int blocks = 256;
  f32x4* gsrc; float* sink; unsigned long long* cyc;
  if (hipMalloc(&gsrc, (size_t)blocks * 16384) != hipSuccess) { printf("no GPU\n"); return 1; }
  (void)hipMemset(gsrc, 0, (size_t)blocks * 16384);
  (void)hipMalloc(&sink, (size_t)blocks * 512 * 4);
  (void)hipMalloc(&cyc, (size_t)blocks * 8 * 8);
  Args a{gsrc, sink, cyc};
  std::vector<unsigned long long> h(blocks * 8);
  (void)hipFuncSetAttribute((const void*)k<MFMA24, MFMA24, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (int warm = 0; warm < 400; ++warm)          // ~50 ms of matrix load: let the clock governor settle
    hipLaunchKernelGGL((k<MFMA24, MFMA24, 0>), dim3(blocks), dim3(512), 96 * 1024, 0, a);
  (void)hipDeviceSynchronize();
  printf("%-8s %-8s sync | cycles/iter: early(w0)  late(w4) | max over workgroups: early late\n", "early", "late");
  run<IDLE, IDLE, 0>(a, h, blocks);               // loop overhead
  run<STAMP, IDLE, 0>(a, h, blocks);
  run<BARRIER, BARRIER, 1>(a, h, blocks);
  run<MFMA24, IDLE, 0>(a, h, blocks);
  run<MFMA24, MFMA24, 0>(a, h, blocks);
  run<MFMA24, MFMA24, 1>(a, h, blocks);           // in-phase: both waves of a SIMD compute between barriers
  run<LDSR16, IDLE, 0>(a, h, blocks);
  run<LDSR16, LDSR16, 0>(a, h, blocks);
  run<LDSR16, MFMA24, 0>(a, h, blocks);
  run<GLD2, IDLE, 0>(a, h, blocks);
  run<GLD2, MFMA24, 0>(a, h, blocks);
  run<DSW2, IDLE, 0>(a, h, blocks);
  run<DSW2, MFMA24, 0>(a, h, blocks);
  run<VALU96, IDLE, 0>(a, h, blocks);
  run<VALU96, MFMA24, 0>(a, h, blocks);
  run<PHASE_L, IDLE, 0>(a, h, blocks);
  run<PHASE_L, PHASE_C, 0>(a, h, blocks);
  run<PHASE_L, PHASE_C, 1>(a, h, blocks);         // ping-pong: LOAD beside COMPUTE, one barrier per phase
  // round 3: the same pairs with a static wave priority for the memory-issuing (early) or the computing (late) waves
  run<GLD2, MFMA24, 0, 3, 0>(a, h, blocks);
  run<GLD2, MFMA24, 0, 0, 3>(a, h, blocks);
  run<VALU96, MFMA24, 0, 3, 0>(a, h, blocks);
  run<LDSR16, MFMA24, 0, 3, 0>(a, h, blocks);
  run<PHASE_L, PHASE_C, 0, 3, 0>(a, h, blocks);
  run<PHASE_L, PHASE_C, 1, 3, 0>(a, h, blocks);
  run<PHASE_L, PHASE_C, 1, 1, 0>(a, h, blocks);
  run<PHASE_L, PHASE_C, 1, 0, 3>(a, h, blocks);
  // round 3: what a memory-issuing wave pays beside a computing SIMD partner, by instruction kind
  run<LDSR16B, IDLE, 0>(a, h, blocks);            // 16 ds_read_b128 back to back, ONE wait (the kernels' real pattern)
  run<LDSR16B, LDSR16B, 0>(a, h, blocks);
  run<LDSR16B, MFMA24, 0>(a, h, blocks);
  run<GLD8S, IDLE, 0>(a, h, blocks);
  run<GLD8S, MFMA24, 0>(a, h, blocks);
  run<GLD8NW, IDLE, 0>(a, h, blocks);             // the same loads consumed late (the wait sits behind the adds' use)
  run<GLD8NW, MFMA24, 0>(a, h, blocks);
  run<GLDS2, IDLE, 0>(a, h, blocks);
  run<GLDS2, MFMA24, 0>(a, h, blocks);
  run<GLD2, MFMA24G, 0>(a, h, blocks);            // the MFMA partner leaves an idle slot behind each MFMA
  run<GLD8S, MFMA24G, 0>(a, h, blocks);
  run<MFMA24G, IDLE, 0>(a, h, blocks);
  // intra-wave pipelines: one wave per SIMD, two waves per SIMD, with a barrier per phase
  run<SCHED_R, IDLE, 0>(a, h, blocks);            // 48 MFMAs + 16 ds_read_b128 per iteration, compiler-scheduled
  run<SCHED_R, SCHED_R, 0>(a, h, blocks);
  run<SCHED_R, IDLE, 1>(a, h, blocks);
  run<PIPE_M, IDLE, 0>(a, h, blocks);             // the 24 MFMAs as raw instructions, VGPR / AccVGPR accumulators
  run<PIPE_MA, IDLE, 0>(a, h, blocks);
  run<PIPE_RA, IDLE, 0>(a, h, blocks);
  run<PIPE_RA, PIPE_RA, 0>(a, h, blocks);
  run<PIPE_R, IDLE, 0>(a, h, blocks);
  run<PIPE_R, PIPE_R, 0>(a, h, blocks);
  run<PIPE_ALL, IDLE, 0>(a, h, blocks);
  run<PIPE_ALL, PIPE_ALL, 0>(a, h, blocks);
  run<PIPE_ALL, IDLE, 1>(a, h, blocks);
  run<PIPE_ALL, PIPE_ALL, 1>(a, h, blocks);
  return 0;
}
