// K-loop probe, EIGHT free-running waves (two per SIMD, 256 registers each): the k32-slot ring of w4_kloop_probe.hip (LDS-DMA into NS slots of
// 224 + 256 rows x 64 B, one barrier per k32 step, fragments of step s+1 read under the MFMAs of step s) with the 2 x 4 wave grid of
// gemm_bf16_pp.hip (wave tile 112 x 64: 7 x 4 accumulator fragments, two k32 fragment sets = 88 registers).  The point: a wave stalled in the
// issue of an LDS-DMA piece (~60 cycles) or at the barrier leaves the matrix pipe to its SIMD partner - no explicit hand-over barriers.
// Prints cycles per K-step (MFMA issue alone = 1792; four-phase ping-pong loop ~2420; four-wave loop 2390).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/w8_kloop_probe scripts/probe/w8_kloop_probe.hip && /tmp/w8_kloop_probe
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
constexpr int A_BYTES = BM * 64, SLOT = (BM + BN) * 64, NPIECE = (BM + BN) / 16;

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
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ int swz(int x) { return (0x78 >> (2 * x)) & 3; }

template <int NS, int PRIO, int ABL = 0, int AUX = 0>          // AUX = cache-policy bits of the LDS-DMA instructions (1 sc0, 2 nt, 16 sc1); ABL (timing ablations, results garbage): 1 = no MFMAs, 2 = no fragment reads, 4 = no DMA
__global__ __launch_bounds__(512, 1) void kloop(const __bf16* __restrict__ A, const __bf16* __restrict__ W, float* __restrict__ C, int M, int N, int K,
                                               int reps, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int n_tiles = N / BN;
  const int tm = blockIdx.x / n_tiles, tn = blockIdx.x % n_tiles;
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t a_rd = lds0 + (wr * 112 + fr) * 64 + ((fq ^ swz(fr >> 2)) * 16);
  const uint32_t w_rd = lds0 + A_BYTES + (wc * 64 + fr) * 64 + ((fq ^ swz(fr >> 2)) * 16);
  const int prow = lane >> 2, lc = (lane & 3) ^ swz(lane >> 4);
  // pieces g = wave + 8 q (q < 4): g < 14 -> A rows 16 g.., g < 30 -> W rows 16 (g - 14)..; waves 6 and 7 have three
  const int npc = wave < 6 ? 4 : 3;
  const char* src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int g = min(wave + 8 * q, NPIECE - 1);
    if (g < BM / 16) src[q] = reinterpret_cast<const char*>(A) + ((long)min(tm * BM + g * 16 + prow, M - 1) * K + lc * 8) * 2;
    else src[q] = reinterpret_cast<const char*>(W) + ((long)(tn * BN + (g - BM / 16) * 16 + prow) * K + lc * 8) * 2;
  }
  const int nk = K / 32, S = nk * reps;
  auto stage_piece = [&](auto q_, int slot, int kstep) {
    constexpr int q = decltype(q_)::value;
    if (q < 3 || npc == 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + (long)kstep * 64),
                                       (__attribute__((address_space(3))) void*)(smem + slot * SLOT + (wave + 8 * q) * 1024), 16, 0, AUX);
  };
  auto wait_steps = [&](auto steps_) {
    constexpr int st = decltype(steps_)::value;
    if (npc == 4) wait_vmcnt<st * 4>(); else wait_vmcnt<st * 3>();
  };
  f32x4 acc[7][4];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 Af[2][7], Wf[2][4];
  if constexpr (ABL != 0) {                                      // (ablations leave fragment registers unwritten)
#pragma unroll
    for (int i = 0; i < 7; ++i) { Af[0][i] = bf16x8{}; Af[1][i] = bf16x8{}; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { Wf[0][j] = bf16x8{}; Wf[1][j] = bf16x8{}; }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) static_for<4>([&](auto q) { stage_piece(q, s, s % nk); });
  wait_steps(std::integral_constant<int, NS - 1>{});
  __builtin_amdgcn_s_barrier();
  static_for<7>([&](auto i) { lds_read128<decltype(i)::value * 1024>(Af[0][decltype(i)::value], a_rd); });
  static_for<4>([&](auto j) { lds_read128<decltype(j)::value * 1024>(Wf[0][decltype(j)::value], w_rd); });
  wait_lgkmcnt<0>();
  SB();
  int rd_slot = 1 % NS, st_slot = 0, st_k = NS % nk;
  const long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  auto step = [&](auto par_) {
    constexpr int cur = decltype(par_)::value, nxt = cur ^ 1;
    wait_steps(std::integral_constant<int, NS - 2>{});
    __builtin_amdgcn_s_barrier();
    SB();
    const uint32_t ar = a_rd + rd_slot * SLOT, wrd = w_rd + rd_slot * SLOT;
    const int my_slot = st_slot, my_k = st_k;
    st_slot = st_slot + 1 == NS ? 0 : st_slot + 1;
    st_k = st_k + 1 == nk ? 0 : st_k + 1;
    SB();
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    static_for<28>([&](auto m_) {
      constexpr int m = decltype(m_)::value, i = m / 4, j = m % 4;
      if constexpr (!(ABL & 1)) mfma_acc(acc[i][j], Wf[cur][j], Af[cur][i]);
      if constexpr (m % 2 == 0 && m / 2 < 4 && !(ABL & 4)) { SB(); stage_piece(std::integral_constant<int, m / 2>{}, my_slot, my_k); SB(); }
      constexpr int r = m < 8 ? (m % 2 == 1 ? m / 2 : -1) : m - 4;
      if constexpr (r >= 0 && r < 11 && !(ABL & 2)) {
        if constexpr (r < 7) lds_read128<r * 1024>(Af[nxt][r], ar);
        else lds_read128<(r - 7) * 1024>(Wf[nxt][r - 7], wrd);
      }
    });
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    SB();
    wait_lgkmcnt<0>();
    SB();
    if (ABL & 1) {                                                 // without MFMAs the fragment registers are dead: keep them allocated until the reads
#pragma unroll                                                     // have landed (the compiler does not know the asm reads are asynchronous)
      for (int i = 0; i < 7; ++i) asm volatile("" ::"v"(Af[nxt][i]));
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(Wf[nxt][j]));
    }
    rd_slot = rd_slot + 1 == NS ? 0 : rd_slot + 1;
  };
  constexpr std::integral_constant<int, 0> P0{};
  constexpr std::integral_constant<int, 1> P1{};
#pragma unroll 1
  for (int s = 0; s < S; s += 2) { step(P0); step(P1); }
  const long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  wait_vmcnt<0>();
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  if (reps == 1) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int row = tm * BM + wr * 112 + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = tn * BN + wc * 64 + j * 16 + fq * 4;
        if (row < M) *reinterpret_cast<f32x4*>(C + (long)row * N + col) = acc[i][j];
      }
    }
  } else {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
    C[(long)blockIdx.x * 512 + tid] = sum;
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

template <int NS, int PRIO, int ABL = 0, int AUX = 0>
static void run(const __bf16* A, const __bf16* W, float* C, float* Cref, long long* cyc, int M, int N, int K) {
  auto kern = kloop<NS, PRIO, ABL, AUX>;
  const int lds = NS * SLOT;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = (M / BM) * (N / BN);
  hipMemset(C, 0, (size_t)M * N * 4);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, A, W, C, M, N, K, 1, cyc);
  hipDeviceSynchronize();
  std::vector<float> h((size_t)M * N), r((size_t)M * N);
  hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), Cref, r.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0; size_t bad = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    const double d = std::fabs((double)h[i] - r[i]);
    if (d > worst) worst = d;
    if (d > 2e-2 + 1e-3 * std::fabs(r[i])) ++bad;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int blk = 0; blk < 6; ++blk) {                             // ~1 s of back-to-back launches: the sustained clock
    hipEventRecord(e0);
    for (int it = 0; it < 1000; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, A, W, C, M, N, K, 8, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> hc(grid), hr(grid);
  hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), cyc + 2048, grid * 8, hipMemcpyDeviceToHost);
  long long sum = 0, rsum = 0; for (auto v : hc) sum += v; for (auto v : hr) rsum += v;
  printf("8 waves, NS=%d, setprio=%d, ablation %d, aux %d: max |diff| %.4f, %zu outside tolerance; %.0f cycles per K-step; sustained: shader clock %.2f GHz, %.0f TF/s\n", NS, PRIO, ABL, AUX, worst, bad,
         (double)sum / grid / 128, (double)sum / rsum / 10.0, 2.0 * M * N * K * 8 * 1000 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
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
  run<5, 0>(A, W, C, Cref, cyc, M, N, K);
  run<5, 0, 1>(A, W, C, Cref, cyc, M, N, K);
  run<5, 0, 3>(A, W, C, Cref, cyc, M, N, K);
  run<5, 0, 4>(A, W, C, Cref, cyc, M, N, K);
  run<5, 0, 6>(A, W, C, Cref, cyc, M, N, K);
  run<5, 0, 3, 1>(A, W, C, Cref, cyc, M, N, K);
  run<5, 0, 3, 2>(A, W, C, Cref, cyc, M, N, K);
  run<5, 0, 3, 3>(A, W, C, Cref, cyc, M, N, K);
  run<5, 0, 3, 16>(A, W, C, Cref, cyc, M, N, K);
  run<5, 0, 0, 2>(A, W, C, Cref, cyc, M, N, K);
  return 0;
}
