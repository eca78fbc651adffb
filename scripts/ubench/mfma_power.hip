// MFMA-only micro-benchmark with ZERO against RANDOM operands: is the matrix pipe's rate on MI355X set by the power
// limit?  (the tap-GEMM launches run at 1.9-2.1 GHz instead of 2.4; round 2's verdict asked for
// this measurement.)  256 workgroups x 8 waves issue back-to-back v_mfma_f32_32x32x16_f16 on register operands for
// ~0.3 s per case; the host samples the shader clock (pp_dpm_sclk) and the socket power (hwmon power1_average) from sysfs
// meanwhile.  Zero operands toggle no multiplier bits: if the rate / clock differ between the two cases at identical
// instruction streams, the limit is electrical, not issue.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ITER = 4096;          // x 16 MFMAs per wave per launch

__global__ __launch_bounds__(512) void mfma_only(const f16x8* __restrict__ src, float* __restrict__ sink) {
  const int tid = threadIdx.x;
  f16x8 a[4], b[4];
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(tid * 8 + i) & 4095];
    b[i] = src[(tid * 8 + 4 + i) & 4095];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + q) & 3], b[i], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[blockIdx.x * 512 + tid] = s;      // keep the chain alive
}

static std::string find_sysfs(const char* leaf) {   // first card that has the file
  for (int c = 0; c < 16; ++c) {
    std::string p = "/sys/class/drm/card" + std::to_string(c) + "/device/" + leaf;
    if (FILE* f = fopen(p.c_str(), "r")) { fclose(f); return p; }
  }
  return "";
}
static std::string find_power() {
  for (int c = 0; c < 16; ++c) {
    std::string base = "/sys/class/drm/card" + std::to_string(c) + "/device/hwmon";
    if (DIR* d = opendir(base.c_str())) {
      while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "hwmon", 5)) continue;
        std::string p = base + "/" + e->d_name + "/power1_average";
        if (FILE* f = fopen(p.c_str(), "r")) { fclose(f); closedir(d); return p; }
      }
      closedir(d);
    }
  }
  return "";
}
static double read_sclk(const std::string& p) {     // the line marked '*'
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return 0;
  char line[128];
  double mhz = 0;
  while (fgets(line, sizeof line, f))
    if (strchr(line, '*')) { const char* c = strchr(line, ':'); if (c) mhz = atof(c + 1); }
  fclose(f);
  return mhz;
}
static double read_power(const std::string& p) {
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return 0;
  double uw = 0;
  if (fscanf(f, "%lf", &uw) != 1) uw = 0;
  fclose(f);
  return uw * 1e-6;
}

int main() {
  const int NWG = 256;
  f16x8* src;
  float* sink;
  hipMalloc(&src, 4096 * sizeof(f16x8));
  hipMalloc(&sink, NWG * 512 * sizeof(float));
  const std::string sclk = find_sysfs("pp_dpm_sclk"), pwr = find_power();
  for (int pass = 0; pass < 2; ++pass)
    for (int random = 0; random < 2; ++random) {
      std::vector<_Float16> h(4096 * 8);
      srand(1);
      for (auto& v : h) v = random ? (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f) : (_Float16)0.f;
      hipMemcpy(src, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
      hipLaunchKernelGGL(mfma_only, dim3(NWG), dim3(512), 0, 0, src, sink);
      hipDeviceSynchronize();
      std::atomic<bool> stop{false};
      double s_clk = 0, s_pw = 0;
      int n = 0;
      std::thread sampler([&] {
        while (!stop.load()) {
          s_clk += read_sclk(sclk);
          s_pw += read_power(pwr);
          ++n;
          std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
      });
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      const int launches = 40;
      hipEventRecord(e0, 0);
      for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(mfma_only, dim3(NWG), dim3(512), 0, 0, src, sink);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      stop.store(true);
      sampler.join();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double mfmas = (double)launches * NWG * 8 * ITER * 16;
      const double tf = mfmas * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
      // 4 SIMDs x 256 CUs, 8 passes x 4 cycles = 32 cycles per MFMA: the clock the rate implies if the pipe never idles
      const double implied_ghz = mfmas * 32.0 / (ms * 1e-3) / (256.0 * 4.0) / 1e9;
      if (pass == 1)
        printf("%-7s operands: %7.1f ms  %7.1f TFLOP/s (dense f16 MFMA)  implied pipe clock %.3f GHz | sampled sclk %.0f MHz, "
               "power %.0f W (%d samples)\n", random ? "random" : "zero", ms, tf, implied_ghz, n ? s_clk / n : 0.0,
               n ? s_pw / n : 0.0, n);
    }
  return 0;
}
