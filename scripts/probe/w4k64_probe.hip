// K-loop probe: FOUR waves (one per SIMD, 512 registers, 112 x 128 per wave = 7 x 8 accumulator fragments in AGPRs) on a k64-GRANULAR operand ring
// split by operand - activations 2 slots x 28 KiB + weights 3 slots x 32 KiB = 152 KiB, 128-byte rows (one global_load_lds_dwordx4 = 8 full cache
// lines; the k32-granular five-slot ring of w4_kloop_probe.hip moves half lines: 25 B/clk/CU = 2 373 cycles per K-step, which is what bound it) - and
// ONE barrier per K-step, placed in the MIDDLE of the step: after the wave has read the step's last fragments, before the MFMAs that consume them.
//   step kt:  [56 MFMAs on (kt, k32 half 0) | reads of (kt, half 1)]  ->  lgkmcnt(0), vmcnt(8), s_barrier  ->
//             [56 MFMAs on (kt, half 1) | reads of (kt+1, half 0) | DMA of A(kt+2) -> the A slot step kt just vacated, W(kt+3) -> the W slot of step kt]
// The barrier says "every wave has read all of step kt" (slots free) and "step kt+1 has landed for every wave" (counted wait: the 8 newest
// instructions, W(kt+2), stay in flight).  DMA alone on this ring: 1 734 cycles per K-step (w3_kloop_probe.hip); MFMA issue alone: 1 792.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/w4k64 scripts/probe/w4k64_probe.hip && /tmp/w4k64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 224, BN = 256;
constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, W_BASE = 2 * A_BYTES, LDS_TOTAL = 2 * A_BYTES + 3 * W_BYTES;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ void lds_read128(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void mfma_acc(f32x4& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}

// ABL: 0 = full, 1 = no MFMA, 2 = no DMA in the loop
template <int ABL>
__global__ __launch_bounds__(256, 1) void kloop(const __bf16* __restrict__ A, const __bf16* __restrict__ W, float* __restrict__ C, int M, int N, int K,
                                               int reps, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int n_tiles = N / BN;
  const int tm = blockIdx.x / n_tiles, tn = blockIdx.x % n_tiles;
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t a_rd[2], w_rd[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int c = ((fq + 4 * kh) ^ (fr & 7)) * 16;
    a_rd[kh] = lds0 + (wr * 112 + fr) * 128 + c;
    w_rd[kh] = lds0 + W_BASE + (wc * 128 + fr) * 128 + c;
  }
  // DMA: a piece = 8 rows x 128 B; lane i -> row i>>3, physical chunk i&7 = logical chunk (i&7)^(i>>3).  A: 28 pieces, 7 per wave; W: 32, 8 per wave.
  const int r8 = lane >> 3, lc = (lane & 7) ^ r8;
  uint32_t a_off[7], w_off[8];
#pragma unroll
  for (int q = 0; q < 7; ++q) a_off[q] = (uint32_t)(((long)min(tm * BM + (wave * 7 + q) * 8 + r8, M - 1) * K + lc * 8) * 2);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int lrow = (wave * 8 + q) * 8 + r8;                    // LDS row of the W tile -> weight row (a lane ends up with 8 consecutive columns)
    const int blk = lrow >> 5, rho = lrow & 31;
    const int col = blk * 32 + ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);
    w_off[q] = (uint32_t)(((long)(tn * BN + col) * K + lc * 8) * 2);
  }
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Wb = reinterpret_cast<const char*>(W);
  const int nk = K / 64;
  const int S = nk * reps;
  auto dma_a = [&](auto q_, int slot, int kstep) {
    constexpr int q = decltype(q_)::value;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ab + a_off[q] + (long)kstep * 128),
                                     (__attribute__((address_space(3))) void*)(smem + slot * A_BYTES + (wave * 7 + q) * 1024), 16, 0, 0);
  };
  auto dma_w = [&](auto q_, int slot, int kstep) {
    constexpr int q = decltype(q_)::value;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wb + w_off[q] + (long)kstep * 128),
                                     (__attribute__((address_space(3))) void*)(smem + W_BASE + slot * W_BYTES + (wave * 8 + q) * 1024), 16, 0, 0);
  };
  f32x4 acc[7][8];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 Af[2][7], Wf[2][8];

  // prologue: A(0), W(0), A(1), W(1), W(2) requested (in that order); step 0 landed; (0, half 0) in register set 0
  static_for<7>([&](auto q) { dma_a(q, 0, 0); });
  static_for<8>([&](auto q) { dma_w(q, 0, 0); });
  static_for<7>([&](auto q) { dma_a(q, 1, 1 % nk); });
  static_for<8>([&](auto q) { dma_w(q, 1, 1 % nk); });
  static_for<8>([&](auto q) { dma_w(q, 2, 2 % nk); });
  wait_vmcnt<23>();                                              // A(0), W(0) landed
  __builtin_amdgcn_s_barrier();
  static_for<7>([&](auto i) { lds_read128<decltype(i)::value * 2048>(Af[0][decltype(i)::value], a_rd[0]); });
  static_for<8>([&](auto j) { lds_read128<decltype(j)::value * 2048>(Wf[0][decltype(j)::value], w_rd[0]); });
  wait_lgkmcnt<0>();
  SB();

  int a_slot = 0, w_slot = 0;                                    // slots of step s
  int k_a = 2 % nk, k_w = 3 % nk;                                // next K-steps to request (A(s+2), W(s+3))
  const long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int s = 0; s < S; ++s) {
    // ---- half 0: 56 MFMAs on register set 0, the 15 fragments of (s, half 1) read underneath
    const uint32_t ar1 = a_rd[1] + a_slot * A_BYTES, wr1 = w_rd[1] + w_slot * W_BYTES;
    SB();
    static_for<56>([&](auto m_) {
      constexpr int m = decltype(m_)::value, i = m / 8, j = m % 8;
      if constexpr (ABL != 1) mfma_acc(acc[i][j], Wf[0][j], Af[0][i]);
      if constexpr (m % 2 == 1 && m / 2 < 15) {
        constexpr int r = m / 2;
        if constexpr (r < 7) lds_read128<r * 2048>(Af[1][r], ar1);
        else lds_read128<(r - 7) * 2048>(Wf[1][r - 7], wr1);
      }
    });
    SB();
    wait_lgkmcnt<0>();
    wait_vmcnt<8>();                                             // A(s+1), W(s+1) landed; W(s+2) may still be in flight
    __builtin_amdgcn_s_barrier();                                // every wave has read all of step s; step s+1 is visible
    SB();
    // ---- half 1: 56 MFMAs on register set 1; reads of (s+1, half 0); DMA A(s+2) -> A slot of step s, W(s+3) -> W slot of step s
    const int na_slot = a_slot ^ 1, nw_slot = w_slot == 2 ? 0 : w_slot + 1;
    const uint32_t ar0 = a_rd[0] + na_slot * A_BYTES, wr0 = w_rd[0] + nw_slot * W_BYTES;
    const int my_a = a_slot, my_w = w_slot, ka = k_a, kw = k_w;
    static_for<56>([&](auto m_) {
      constexpr int m = decltype(m_)::value, i = m / 8, j = m % 8;
      if constexpr (ABL != 1) mfma_acc(acc[i][j], Wf[1][j], Af[1][i]);
      if constexpr (ABL != 2 && m % 3 == 0 && m / 3 < 15) {      // the step's 15 DMA instructions spread over its MFMAs, activations first
        constexpr int d = m / 3;
        SB();
        if constexpr (d < 7) dma_a(std::integral_constant<int, d>{}, my_a, ka);
        else dma_w(std::integral_constant<int, d - 7>{}, my_w, kw);
        SB();
      }
      if constexpr (m % 2 == 1 && m / 2 < 15) {
        constexpr int r = m / 2;
        if constexpr (r < 7) lds_read128<r * 2048>(Af[0][r], ar0);
        else lds_read128<(r - 7) * 2048>(Wf[0][r - 7], wr0);
      }
    });
    SB();
    wait_lgkmcnt<0>();
    SB();
    a_slot = na_slot; w_slot = nw_slot;
    k_a = k_a + 1 == nk ? 0 : k_a + 1;
    k_w = k_w + 1 == nk ? 0 : k_w + 1;
  }
  const long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  wait_vmcnt<0>();
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  if (reps == 1) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int row = tm * BM + wr * 112 + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = tn * BN + wc * 128 + (j >> 1) * 32 + fq * 8 + (j & 1) * 4;
        if (row < M) *reinterpret_cast<f32x4*>(C + (long)row * N + col) = acc[i][j];
      }
    }
  } else {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][3];
    C[(long)blockIdx.x * 256 + tid] = sum;
  }
  if (tid == 0) { cyc[blockIdx.x] = t1 - t0; cyc[2048 + blockIdx.x] = (long long)(r1 - r0); }
}

__global__ void naive(const __bf16* A, const __bf16* W, float* C, int M, int N, int K) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= M || col >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(long)row * K + k] * (float)W[(long)col * K + k];
  C[(long)row * N + col] = s;
}

template <int ABL>
static void run(const char* name, const __bf16* A, const __bf16* W, float* C, float* Cref, long long* cyc, int M, int N, int K, int reps) {
  auto kern = kloop<ABL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  const int grid = (M / BM) * (N / BN);
  double worst = 0; size_t bad = 0;
  if (ABL == 0) {
    hipMemset(C, 0, (size_t)M * N * 4);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_TOTAL, 0, A, W, C, M, N, K, 1, cyc);
    hipDeviceSynchronize();
    std::vector<float> h((size_t)M * N), r((size_t)M * N);
    hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), Cref, r.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); ++i) {
      const double d = std::fabs((double)h[i] - r[i]);
      if (d > worst) worst = d;
      if (d > 2e-2 + 1e-3 * std::fabs(r[i])) ++bad;
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_TOTAL, 0, A, W, C, M, N, K, reps, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<long long> hc(grid), hr(grid);
  hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), cyc + 2048, grid * 8, hipMemcpyDeviceToHost);
  long long sum = 0, mx = 0, rsum = 0; for (auto v : hc) { sum += v; if (v > mx) mx = v; }
  for (auto v : hr) rsum += v;
  const double steps = (double)(K / 64) * reps;
  printf("%-22s max |diff| %.4f, %zu outside tolerance; %.0f cycles per K-step (mean over workgroups; slowest %.0f) = %.1f B/clk/CU; shader clock %.2f GHz; launch %.1f us for %d passes\n",
         name, worst, bad, (double)sum / grid / steps, (double)mx / steps, 61440.0 / ((double)sum / grid / steps), (double)sum / rsum / 10.0, best * 1e3, reps);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int M = 3584, N = 4096, K = 1024;
  std::vector<__bf16> hA((size_t)M * K), hW((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = (__bf16)((rand() % 2001 - 1000) / 1000.0f);
  for (auto& v : hW) v = (__bf16)((rand() % 2001 - 1000) / 8000.0f);
  __bf16 *A, *W; float *C, *Cref; long long* cyc;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&Cref, (size_t)M * N * 4); hipMalloc(&cyc, 4096 * 8);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(naive, dim3(N / 64, M / 4), dim3(256), 0, 0, A, W, Cref, M, N, K);
  hipDeviceSynchronize();
  run<0>("full loop", A, W, C, Cref, cyc, M, N, K, 8);
  run<0>("full loop (32 passes)", A, W, C, Cref, cyc, M, N, K, 32);
  run<2>("no DMA in the loop", A, W, C, Cref, cyc, M, N, K, 8);
  return 0;
}
