// Probe: the 224 x 256 x 64 K-step of the expert GEMMs as a PLAIN eight-wave loop - one s_barrier per K-step, no ping-pong hand-over -
// with the operand ring split by operand: activations 2 slots x 28 KiB + weights 3 slots x 32 KiB = 152 KiB of LDS, k64-granular (128-byte rows:
// a global_load_lds_dwordx4 = 8 full cache lines; the k32-granular five-slot ring of w8_kloop_probe.hip moved half lines and stalled at 25 B/clk/CU).
// The weights of K-step kt+2 and the activations of kt+1 are requested at the top of K-step kt; one counted wait (vmcnt(4): the newest weight tile stays
// in flight) per K-step.  Question: cycles per K-step against the ping-pong kernel's 2 420 and the MFMA issue time of 1 792.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/w3 scripts/probe/w3_kloop_probe.hip && /tmp/w3
//   W3_ABL=1: no MFMAs (DMA + reads + barriers), 2: no DMA inside the loop, 3: no fragment reads / MFMAs (DMA + barriers only)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 224, BN = 256, BK = 64, TM = 112, TN = 64, FM = 7, FN = 4;
constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;       // 28 KiB, 32 KiB
constexpr int W_BASE = 2 * A_BYTES, LDS_TOTAL = 2 * A_BYTES + 3 * W_BYTES;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ void lds_read128(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int STRIDE, int CNT, int I = 0> __device__ __forceinline__ void lds_read_seq(bf16x8* dst, uint32_t addr) {
  if constexpr (I < CNT) { lds_read128<I * STRIDE>(dst[I], addr); lds_read_seq<STRIDE, CNT, I + 1>(dst, addr); }
}

template <int ABL>
__global__ __launch_bounds__(512, 2) void w3_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, float* __restrict__ C, int M, int N, int K,
                                                    long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int n_tiles = N / BN;
  const int mt = blockIdx.x / n_tiles, nt = blockIdx.x % n_tiles;
  const int r8 = lane >> 3, lchunk = (lane & 7) ^ r8;
  // A: 28 pieces of 8 rows: waves 0-5 own pieces 4w .. 4w+3, wave 6 pieces 24-27, wave 7 none.  W: 32 pieces, 4 per wave.
  const int na = wave < 7 ? 4 : 0;
  const uint16_t* a_src[4];
  const uint16_t* b_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tr = (wave * 4 + q) * 8 + r8;
    a_src[q] = A + (long)min(mt * BM + tr, M - 1) * K + lchunk * 8;
    b_src[q] = W + (long)(nt * BN + tr) * K + lchunk * 8;
  }
  auto stage_a = [&](char* base, int kt) {
    if (na) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + kt * BK),
                                         (__attribute__((address_space(3))) void*)(base + (wave * 4 + q) * 1024), 16, 0, 0);
    }
  };
  auto stage_w = [&](char* base, int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(base + (wave * 4 + q) * 1024), 16, 0, 0);
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
  const int c0 = (fq ^ sw) * 16, c1 = ((fq + 4) ^ sw) * 16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t a_base = lds0 + (wm * TM + fr) * 128, b_base = lds0 + W_BASE + (wn * TN + fr) * 128;
  bf16x8 fa0[FM], fb0[FN], fa1[FM], fb1[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) { fa0[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; fa1[i] = fa0[i]; }
#pragma unroll
  for (int j = 0; j < FN; ++j) { fb0[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; fb1[j] = fb0[j]; }
  auto mma = [&](const bf16x8(&fa)[FM], const bf16x8(&fb)[FN]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  const int nk = K / BK;
  stage_a(smem, 0);
  stage_w(smem + W_BASE, 0);
  stage_w(smem + W_BASE + W_BYTES, 1);
  int ws = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) wait_vmcnt<4>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if constexpr (ABL != 2) {
      if (kt + 1 < nk) stage_a(smem + ((kt + 1) & 1) * A_BYTES, kt + 1);
      if (kt + 2 < nk) stage_w(smem + W_BASE + (ws == 0 ? 2 : ws - 1) * W_BYTES, kt + 2);
    }
    const uint32_t ao = (kt & 1) * A_BYTES, wo = ws * W_BYTES;
    if constexpr (ABL != 3) {
      lds_read_seq<2048, FM>(fa0, a_base + ao + c0); lds_read_seq<2048, FN>(fb0, b_base + wo + c0);
      lds_read_seq<2048, FM>(fa1, a_base + ao + c1); lds_read_seq<2048, FN>(fb1, b_base + wo + c1);
      wait_lgkmcnt<FM + FN>();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ABL != 1) mma(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ABL != 1) mma(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
    }
    ws = ws == 2 ? 0 : ws + 1;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0 && cyc) cyc[blockIdx.x] = t1 - t0;
  if constexpr (ABL == 1 || ABL == 3) {   // keep the fragment registers alive
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) s += (float)fa0[i][0] + (float)fa1[i][0];
#pragma unroll
    for (int j = 0; j < FN; ++j) s += (float)fb0[j][0] + (float)fb1[j][0];
    acc[0][0][0] += s;
  }
  // swapped operands: lane owns row i*16 + fr, columns j*16 + fq*4 .. +3
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = mt * BM + wm * TM + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = nt * BN + wn * TN + j * 16 + fq * 4;
      if (row < M) *reinterpret_cast<float4*>(C + (long)row * N + col) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int ABL>
static void run(const char* name, const uint16_t* dA, const uint16_t* dW, float* dC, long long* dcyc, int M, int N) {
  auto kern = w3_kernel<ABL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  const int grid = (M / BM) * (N / BN);
  double us[2], cy[2];
  int idx = 0;
  for (int K : {1024, 2048}) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_TOTAL, 0, dA, dW, dC, M, N, K, dcyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_TOTAL, 0, dA, dW, dC, M, N, K, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    us[idx] = ms * 1e3 / reps;
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), dcyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    cy[idx] = s / grid;
    ++idx;
  }
  printf("%-28s K=1024 %7.2f us (loop %8.0f clk = %6.0f / K-step)   K=2048 %7.2f us (loop %8.0f clk = %6.0f / K-step)   slope %6.0f clk / K-step, %5.1f B/clk/CU\n", name,
         us[0], cy[0], cy[0] / 16, us[1], cy[1], cy[1] / 32, (cy[1] - cy[0]) / 16, 61440.0 / ((cy[1] - cy[0]) / 16));
}

int main() {
  const int M = 3584, N = 8192, Kmax = 2048;
  std::vector<uint16_t> hA((size_t)M * Kmax), hW((size_t)N * Kmax);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hA) v = f2bf(rnd());
  for (auto& v : hW) v = f2bf(rnd());
  uint16_t *dA, *dW; float* dC; long long* dcyc;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 4); hipMalloc(&dcyc, 4096 * 8);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  // correctness of the full loop at K = 1024 (row stride K: use the first 1024 columns of a K = 1024 layout -> re-upload compactly)
  {
    const int K = 1024;
    std::vector<uint16_t> a2((size_t)M * K), w2((size_t)N * K);
    for (int r = 0; r < M; ++r) memcpy(&a2[(size_t)r * K], &hA[(size_t)r * Kmax], K * 2);
    for (int r = 0; r < N; ++r) memcpy(&w2[(size_t)r * K], &hW[(size_t)r * Kmax], K * 2);
    hipMemcpy(dA, a2.data(), a2.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, w2.data(), w2.size() * 2, hipMemcpyHostToDevice);
    auto kern = w3_kernel<0>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipLaunchKernelGGL(kern, dim3((M / BM) * (N / BN)), dim3(512), LDS_TOTAL, 0, dA, dW, dC, M, N, K, dcyc);
    hipDeviceSynchronize();
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int t = 0; t < 4000; ++t) {
      const int r = (t * 7919) % M, c = (t * 104729) % N;
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(a2[(size_t)r * K + k]) * bf2f(w2[(size_t)c * K + k]);
      worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c]));
    }
    printf("check (4000 samples of C, K = 1024): max abs err %.3e %s\n", worst, worst < 2e-3 ? "OK" : "FAIL");
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  }
  // timing: operands laid out with row stride = K of the run (K = 2048 uses the full buffers; K = 1024 the first half as a [rows][1024] matrix)
  run<0>("full loop", dA, dW, dC, dcyc, M, N);
  run<1>("no MFMA (DMA+reads+bar)", dA, dW, dC, dcyc, M, N);
  run<2>("no DMA in loop", dA, dW, dC, dcyc, M, N);
  run<3>("DMA + barriers only", dA, dW, dC, dcyc, M, N);
  return 0;
}
