// Error plumbing + library identity for libdv3hip.so.
#include "common.h"
#include <atomic>
#include <mutex>
#include <vector>
#include <string.h>

static thread_local char g_err[512] = "";

void dv3_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int g_dv3_last_conv = 0, g_dv3_last_wgrad = 0;

// sticky fp16-range event counter of the f16x3 mode (include/dv3hip.h: dv3_f16_range_events)
__device__ uint32_t g_dv3_range_events;
uint32_t* dv3_range_ctr() {
  // resolved once (backward runs on autograd's thread: call_once, not an unsynchronised static); the __device__ word is
  // zero-initialised by the loader, so nothing is issued on any stream here (a first call may land inside a capture)
  static uint32_t* ptr = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_dv3_range_events)) == hipSuccess) ptr = (uint32_t*)p;
  });
  return ptr;
}
extern "C" int dv3_f16_range_events(int32_t* dst, int32_t reset, void* stream) {
  uint32_t* ctr = dv3_range_ctr();
  DV3_REQUIRE(ctr != nullptr, "f16_range_events: no device");
  hipStream_t st = (hipStream_t)stream;
  if (dst) {
    hipError_t e = hipMemcpyAsync(dst, ctr, sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
      dv3_set_error("f16_range_events: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
  }
  if (reset) {
    hipError_t e = hipMemsetAsync(ctr, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) {
      dv3_set_error("f16_range_events: %s", hipGetErrorString(e));
      return DV3_ELAUNCH;
    }
  }
  return DV3_OK;
}
extern "C" int dv3_stream_fork(void* from, void* to) {
  // a ring of events created once (backward runs on autograd's device thread, join() on the caller's: thread-safe)
  static hipEvent_t ring[256];
  static std::once_flag made;
  static std::atomic<unsigned> next{0};
  static hipError_t create_err = hipSuccess;
  std::call_once(made, [] {
    for (int i = 0; i < 256 && create_err == hipSuccess; ++i) create_err = hipEventCreateWithFlags(&ring[i], hipEventDisableTiming);
  });
  if (create_err != hipSuccess) {
    dv3_set_error("stream_fork: hipEventCreate: %s", hipGetErrorString(create_err));
    return DV3_ELAUNCH;
  }
  const unsigned i = next.fetch_add(1, std::memory_order_relaxed) % 256u;
  hipError_t e = hipEventRecord(ring[i], (hipStream_t)from);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)to, ring[i], 0);
  if (e != hipSuccess) {
    dv3_set_error("stream_fork: %s", hipGetErrorString(e));
    return DV3_ELAUNCH;
  }
  return DV3_OK;
}
// ---- the weight-gradient branch of a captured step as its own hipGraphs (include/dv3hip.h) ----
extern "C" int dv3_graph_side_begin(void* side_stream) {
  DV3_REQUIRE(side_stream, "graph_side_begin: the side stream must be a real (non-default) stream");
  hipError_t e = hipStreamBeginCapture((hipStream_t)side_stream, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) {
    dv3_set_error("graph_side_begin: %s", hipGetErrorString(e));
    return DV3_ELAUNCH;
  }
  return DV3_OK;
}
extern "C" int dv3_graph_side_end(void* side_stream, void** exec_out, int32_t* n_nodes_out) {
  DV3_REQUIRE(side_stream && exec_out, "graph_side_end: null argument");
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)side_stream, &graph);
  if (e != hipSuccess || !graph) {
    dv3_set_error("graph_side_end: hipStreamEndCapture: %s", hipGetErrorString(e));
    return DV3_ELAUNCH;
  }
  size_t n = 0;
  (void)hipGraphGetNodes(graph, nullptr, &n);
  if (n_nodes_out) *n_nodes_out = (int32_t)n;
  *exec_out = nullptr;
  if (n == 0) {
    (void)hipGraphDestroy(graph);
    return DV3_OK;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    dv3_set_error("graph_side_end: hipGraphInstantiate: %s", hipGetErrorString(e));
    return DV3_ELAUNCH;
  }
  *exec_out = (void*)exec;
  return DV3_OK;
}
// ---- ABI 43: fork points of a captured backward ordered by a device flag (include/dv3hip.h) ----
namespace {
__global__ void flag_signal_kernel(unsigned long long* flag, unsigned long long* epoch, int j, int bump) {
  unsigned long long e = *epoch;
  if (bump) {
    e += 1;
    *epoch = e;
  }
  __hip_atomic_store(flag, e * 4096ull + (unsigned long long)j, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void flag_wait_kernel(const unsigned long long* flag, unsigned long long* epoch, int j, int bump, unsigned* err,
                                 long long timeout_ticks) {
  unsigned long long e = *epoch;
  if (bump) {
    e += 1;
    *epoch = e;
  }
  const unsigned long long want = e * 4096ull + (unsigned long long)j;
  const long long t0 = (long long)wall_clock64();     // the 100 MHz constant clock
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(16);
    if ((long long)wall_clock64() - t0 > timeout_ticks) {
      atomicAdd(err, 1u);
      break;
    }
  }
}
}  // namespace
extern "C" int dv3_flag_signal(uint64_t* flag, uint64_t* epoch, int32_t j, int32_t bump, void* stream) {
  DV3_REQUIRE(flag && epoch && j > 0 && j < 4096, "flag_signal: bad arguments");
  hipLaunchKernelGGL(flag_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)flag,
                     (unsigned long long*)epoch, (int)j, (int)bump);
  return dv3_check_launch("flag_signal");
}
extern "C" int dv3_flag_wait(const uint64_t* flag, uint64_t* epoch, int32_t j, int32_t bump, uint32_t* err,
                             int32_t timeout_ms, void* stream) {
  DV3_REQUIRE(flag && epoch && err && j > 0 && j < 4096 && timeout_ms > 0, "flag_wait: bad arguments");
  hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned long long*)flag,
                     (unsigned long long*)epoch, (int)j, (int)bump, (unsigned*)err, (long long)timeout_ms * 100000ll);
  return dv3_check_launch("flag_wait");
}
extern "C" int dv3_graph_launch(void* exec, void* stream) {
  DV3_REQUIRE(exec, "graph_launch: null graph");
  hipError_t e = hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream);
  if (e != hipSuccess) {
    dv3_set_error("graph_launch: %s", hipGetErrorString(e));
    return DV3_ELAUNCH;
  }
  return DV3_OK;
}
extern "C" int dv3_graph_destroy(void* exec) {
  if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec);
  return DV3_OK;
}

int dv3_conv_census_count();   // conv_gemm.hip
extern "C" int dv3_debug_get(int what) {
  if (what == 10) return g_dv3_last_conv;
  if (what == 40) return dv3_conv_census_count();
  if (what == 11) return g_dv3_last_wgrad;
  return 0;
}

extern "C" const char* dv3_last_error(void) { return g_err; }
extern "C" int dv3_abi_version(void) { return DV3_ABI_VERSION; }

extern "C" int dv3_device_info(int dev, char* name, int name_len, int* n_cu) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    dv3_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
    return DV3_ELAUNCH;
  }
  if (name && name_len > 0) {
    strncpy(name, prop.gcnArchName, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  return DV3_OK;
}

// sizeof of every descriptor struct, so the ctypes mirror can be checked at load time
extern "C" int dv3_sizeof(const char* name) {
#define DV3_SZ(T) if (!strcmp(name, #T)) return (int)sizeof(T)
  DV3_SZ(dv3_conv_desc);
  DV3_SZ(dv3_wgrad_desc);
  DV3_SZ(dv3_wn_desc);
  DV3_SZ(dv3_wn_bwd_desc);
  DV3_SZ(dv3_gate_bwd_desc);
  DV3_SZ(dv3_softmax_desc);
  DV3_SZ(dv3_softmax_bwd_desc);
  DV3_SZ(dv3_spec_loss_desc);
  DV3_SZ(dv3_planes_desc);
  DV3_SZ(dv3_wn_multi_entry);
  DV3_SZ(dv3_conv_step_desc);
  DV3_SZ(dv3_attn_step_desc);
  DV3_SZ(dv3_decode_entry);
  DV3_SZ(dv3_decode_program);
  DV3_SZ(dv3_attn_fwd_desc);
  DV3_SZ(dv3_spk_layer);
  DV3_SZ(dv3_spk_desc);
  DV3_SZ(dv3_dropout_site);
#undef DV3_SZ
  return -1;
}
