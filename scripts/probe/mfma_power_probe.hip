// What does the socket sustain at its power cap?  Register-resident back-to-back bf16 MFMAs on every SIMD of every CU (no LDS, no memory), for ~1.5 s
// per variant: v_mfma_f32_16x16x32_bf16 (the instruction of every GEMM in this library) and v_mfma_f32_32x32x16_bf16 (same FLOP/cycle, half the
// operand-register reads per FLOP), one or two waves per SIMD.  Prints sustained TF/s, the shader clock seen by the kernel (s_memtime / s_memrealtime)
// and, when readable, the hwmon socket power.  Operands are random (toggle rate matters for power).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mfma_power_probe scripts/probe/mfma_power_probe.hip && /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <glob.h>
#include <unistd.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512) void burn(const bf16x8* __restrict__ src, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(lane + 64 * i) & 1023]; b[i] = src[(lane * 7 + 64 * i + 13) & 1023]; }
  const long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  }
  const long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = (long long)(r1 - r0); }
}

static std::string hwmon_dir() {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, 64, 0) != hipSuccess) return "";
  for (char* c = bdf; *c; ++c) *c = tolower(*c);
  std::string pat = std::string("/sys/bus/pci/devices/") + bdf + "/hwmon/hwmon*";
  glob_t g; std::string r;
  if (glob(pat.c_str(), 0, nullptr, &g) == 0 && g.gl_pathc) r = g.gl_pathv[0];
  globfree(&g);
  return r;
}
static double read_num(const std::string& f) { FILE* fp = fopen(f.c_str(), "r"); if (!fp) return -1; double v = -1; if (fscanf(fp, "%lf", &v) != 1) v = -1; fclose(fp); return v; }

template <int SHAPE>
static void run(int waves_per_simd, const bf16x8* src, float* out, long long* cyc, const std::string& hw) {
  const int threads = 256 * waves_per_simd, grid = 256;
  const int iters = 20000;                                        // 16 (8) MFMAs of 16 (32) cycles per iteration: ~5 M cycles ~ 2.5 ms per launch
  const double flop_per_launch = (double)grid * (threads / 64) * iters * (SHAPE == 16 ? 16 * 16384.0 : 8 * 32768.0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double pw_sum = 0; int pw_n = 0; float last_ms = 0; double clk = 0;
  for (int rep = 0; rep < 12; ++rep) {                            // 12 x 50 launches ~ 1.5 s; the numbers of the last block are reported
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(burn<SHAPE>, dim3(grid), dim3(threads), 0, 0, src, out, cyc, iters);
    hipEventRecord(e1);
    while (hipEventQuery(e1) != hipSuccess) {
      if (!hw.empty() && rep >= 6) { const double w = read_num(hw + "/power1_input"); if (w > 0) { pw_sum += w / 1e6; ++pw_n; } }
      usleep(5000);
    }
    hipEventElapsedTime(&last_ms, e0, e1);
    std::vector<long long> h(grid * 2);
    hipMemcpy(h.data(), cyc, grid * 16, hipMemcpyDeviceToHost);
    long long c = 0, r = 0; for (int i = 0; i < grid; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
    clk = (double)c / r / 10.0;
  }
  printf("%dx%dx%d bf16, %d wave(s) per SIMD: %.0f TF/s sustained, shader clock %.2f GHz, socket %.0f W (cap %.0f W)\n", SHAPE, SHAPE, SHAPE == 16 ? 32 : 16,
         waves_per_simd, flop_per_launch * 50 / (last_ms * 1e-3) / 1e12, clk, pw_n ? pw_sum / pw_n : -1.0, hw.empty() ? -1.0 : read_num(hw + "/power1_cap") / 1e6);
}

int main() {
  std::vector<unsigned short> h(1024 * 8);
  srand(3);
  for (auto& v : h) { float f = (rand() % 2001 - 1000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8* src; float* out; long long* cyc;
  hipMalloc(&src, h.size() * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 16);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const std::string hw = hwmon_dir();
  run<16>(1, src, out, cyc, hw);
  run<32>(1, src, out, cyc, hw);
  run<16>(2, src, out, cyc, hw);
  run<32>(2, src, out, cyc, hw);
  return 0;
}
