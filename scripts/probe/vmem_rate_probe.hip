// Per-CU operand-fetch ceilings, measured with nothing else running in the workgroup: eight waves stream the k32 pieces of a 224 x 256 tile
// (same addresses and order as the GEMM K loop: 1 KiB per wave-instruction = 16 rows x 64 B), four pieces per wave and step, four steps in flight,
//   MODE 0: global_load_lds_dwordx4 (LDS-DMA into a ring of slots)      MODE 1: global_load_dwordx4 into registers (discarded)
//   MODE 2: global_load_dwordx4 + ds_write_b128 of the previous step's registers (the register-staged fill)
// Prints bytes per clock and CU.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/vmem_rate_probe scripts/probe/vmem_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void stream(const char* __restrict__ A, const char* __restrict__ W, int M, int N, int K, int steps, long long* cyc, unsigned* sink, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 224, BN = 256, SLOT = 32 * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tiles = N / BN, tm = blockIdx.x / n_tiles, tn = blockIdx.x % n_tiles;
  const int prow = lane >> 2, lc = lane & 3;
  const char* src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int g = wave + 8 * q;                                   // 32 pieces per step (30 real ones + 2 repeats of W rows)
    if (g < 14) src[q] = A + ((long)(tm * BM + g * 16 + prow) * ld + lc * 8) * 2;
    else src[q] = W + ((long)(tn * BN + ((g - 14) % 16) * 16 + prow) * ld + lc * 8) * 2;
  }
  const int nk = K / 32;
  u32x4 r[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) r[s][q] = u32x4{0, 0, 0, 0};
  const unsigned lds_w = (unsigned)(size_t)smem + wave * 4096 + lane * 16;
  unsigned acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  int k = 0;
  for (int s = 0; s < steps; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else      // MODE 6: barrier, one step less in flight
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");            // the set about to be overwritten has landed (three steps stay in flight)
      if (MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 8 || (MODE == 7 && u == 0) || (MODE == 9 && (u & 1) == 0)) __builtin_amdgcn_s_barrier();   // MODE 7: a barrier every 4 steps, MODE 9: every 2
      if (MODE == 8) { if (wave & 1) __builtin_amdgcn_s_sleep(4); if (wave & 2) __builtin_amdgcn_s_sleep(8); if (wave & 4) __builtin_amdgcn_s_sleep(16); }   // MODE 8: waves leave the barrier ~64 cycles apart    // MODE 3: + one workgroup barrier per step (what a shared LDS ring needs)
      if (MODE == 4) __builtin_amdgcn_s_sleep(8);                  // MODE 4: + ~500 idle cycles per step (room for the MFMAs of a step)
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (MODE == 2) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lds_w + u * SLOT), "v"(r[u][q]), "n"(0) : "memory");
          else asm volatile("" ::"v"(r[u][q]));
          acc ^= 1;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const char* p = src[q] + (long)(MODE == 5 ? (k + wave * 4 + q * 9) % nk : k) * 64;   // MODE 5: barrier, but every piece of a step at another k
        if (MODE == 0 || MODE >= 3)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                           (__attribute__((address_space(3))) void*)(smem + u * SLOT + (wave + 8 * q) * 1024), 16, 0, 0);
        else
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[u][q]) : "v"(p) : "memory");
      }
      k = k + 1 == nk ? 0 : k + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0) cyc[4096 + blockIdx.x * 8 + wave] = __builtin_readcyclecounter() - t0;   // this wave alone
  __builtin_amdgcn_s_barrier();                                    // the workgroup's time = its slowest wave's
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc ^= r[s][q][0];
  const long long t1 = __builtin_readcyclecounter();
  if (acc == 0x12345) sink[0] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* A, const char* W, int M, int N, int K, long long* cyc, unsigned* sink, const char* name, int ld = 1024) {
  auto kern = stream<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const int grid = (M / 224) * (N / 256), steps = 1024;
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 128 * 1024, 0, A, W, M, N, K, steps, cyc, sink, ld);
  hipDeviceSynchronize();
  std::vector<long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  long long sum = 0; for (auto v : h) sum += v;
  const double c = (double)sum / grid;
  printf("%-60s %.1f B/clk/CU (%.0f cycles per 61 440-byte K-step)\n", name, 32.0 * 1024 * steps / c, c / steps * 2 * 30 / 32);
  if (MODE == 0 || MODE == 1) {
    std::vector<long long> w(grid * 8);
    hipMemcpy(w.data(), cyc + 4096, grid * 64, hipMemcpyDeviceToHost);
    printf("    per wave (mean cycles per step over the workgroups):");
    for (int k = 0; k < 8; ++k) { double a = 0; for (int b = 0; b < grid; ++b) a += w[b * 8 + k]; printf(" %.0f", a / grid / steps); }
    printf("\n");
  }
  fflush(stdout);
}

int main() {
  const int M = 3584, N = 4096, K = 1024;
  char *A, *W; long long* cyc; unsigned* sink;
  hipMalloc(&A, (size_t)M * (K + 512) * 2); hipMalloc(&W, (size_t)N * (K + 512) * 2); hipMalloc(&cyc, 8192 * 8); hipMalloc(&sink, 64);
  hipMemset(A, 1, (size_t)M * K * 2); hipMemset(W, 2, (size_t)N * K * 2);
  run<0>(A, W, M, N, K, cyc, sink, "global_load_lds_dwordx4 (LDS-DMA)");
  run<1>(A, W, M, N, K, cyc, sink, "global_load_dwordx4 -> registers");
  run<2>(A, W, M, N, K, cyc, sink, "global_load_dwordx4 -> registers -> ds_write_b128");
  run<3>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier per k32 step");
  run<4>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier + s_sleep 8 per k32 step");
  run<5>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier, pieces of a step at different k");
  run<6>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier, three steps in flight instead of four");
  run<9>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier every 2 steps");
  run<7>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier every 4 steps");
  run<8>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier per step, waves staggered by s_sleep");
  run<3>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier, row pitch 2048 + 64 B", 1024 + 32);
  run<3>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier, row pitch 2048 + 128 B", 1024 + 64);
  run<3>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier, row pitch 2048 + 256 B", 1024 + 128);
  run<3>(A, W, M, N, K, cyc, sink, "LDS-DMA + s_barrier, row pitch 2048 + 512 B", 1024 + 256);
  run<0>(A, W, M, N, K, cyc, sink, "LDS-DMA free-running, row pitch 2048 + 128 B", 1024 + 64);
  return 0;
}
