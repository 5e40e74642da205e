// K-loop probe for a FOUR-wave formulation of the 224x256 ping-pong GEMM (see LABNOTES.md section 8): one wave per SIMD with the
// 512-register budget, each wave owning 112 rows x 128 columns (7 x 8 accumulator fragments), operands streamed by LDS-DMA into a ring of NS
// k32-granular slots (224 + 256 rows x 64 B = 30 KiB each), fragments of k32-step s+1 read under the MFMAs of step s, one barrier per k32 step
// (BPK = 2) or per K-step (BPK = 1).  Real operands, real DMA, result checked against a naive kernel.  Prints cycles per K-step (64 k):
// MFMA issue alone = 1792; the 8-wave four-phase loop of gemm_bf16_pp.hip measures ~2420.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/w4_kloop_probe scripts/probe/w4_kloop_probe.hip && /tmp/w4_kloop_probe
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
constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64, SLOT = A_BYTES + W_BYTES;

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

// accumulators pinned to AGPRs and updated in place; asm volatile keeps the hand-written MFMA / ds_read / DMA interleave
__device__ __forceinline__ void mfma_acc(f32x4& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ int swz(int x) { return (0x78 >> (2 * x)) & 3; }      // f = (0, 2, 3, 1): conflict-free ds_read_b128 on 64-byte rows

template <int NS, int BPK>
__global__ __launch_bounds__(256, 1) void kloop(const __bf16* __restrict__ A, const __bf16* __restrict__ W, float* __restrict__ C, int M, int N, int K,
                                               int reps, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int n_tiles = N / BN;
  // reps < 0: every workgroup works on tile (0, 0) - all operand traffic hits in L2 (separates the miss path from the L2 -> LDS path)
  const bool same = reps < 0;
  if (same) reps = -reps;
  const int tm = same ? 0 : blockIdx.x / n_tiles, tn = same ? 0 : blockIdx.x % n_tiles;
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t a_rd = lds0 + (wr * 112 + fr) * 64 + ((fq ^ swz(fr >> 2)) * 16);
  const uint32_t w_rd = lds0 + A_BYTES + (wc * 128 + fr) * 64 + ((fq ^ swz(fr >> 2)) * 16);
  // DMA sources: a piece = 16 rows x 64 B; lane i -> row i>>2, physical chunk i&3 = logical chunk (i&3) ^ swz(i>>4)
  const int prow = lane >> 2, lc = (lane & 3) ^ swz(lane >> 4);
  const int na = wave < 2 ? 4 : 3;                               // A pieces of this wave (14 in all)
  const int a_first = wave < 2 ? wave * 4 : 8 + (wave - 2) * 3;
  uint32_t a_off[4], w_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int arow = min(tm * BM + (a_first + min(q, na - 1)) * 16 + prow, M - 1);
    a_off[q] = (uint32_t)(((long)arow * K + lc * 8) * 2);
    const int rho_l = (wave * 4 + q) * 16 + prow;                // LDS row of the W tile
    const int blk = rho_l >> 5, rho = rho_l & 31;
    const int col = blk * 32 + ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);
    w_off[q] = (uint32_t)(((long)(tn * BN + col) * K + lc * 8) * 2);
  }
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Wb = reinterpret_cast<const char*>(W);
  const int nk = K / 32;                                          // k32 steps per pass
  const int S = nk * reps;

  // piece q of this wave's share of one k32 step (0-3: W, 4-6: A, 7: the fourth A piece of waves 0 and 1) -> LDS slot `slot`
  auto stage_piece = [&](auto q_, int slot, int kstep) {
    constexpr int q = decltype(q_)::value;
    const long kb = (long)kstep * 64;
    char* base = smem + slot * SLOT;
    if constexpr (q < 4) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wb + w_off[q] + kb),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave * 4 + q) * 1024), 16, 0, 0);
    } else {
      if (q < 7 || na == 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ab + a_off[q - 4] + kb),
                                         (__attribute__((address_space(3))) void*)(base + (a_first + q - 4) * 1024), 16, 0, 0);
    }
  };
  auto stage = [&](int slot, int kstep) { static_for<8>([&](auto q) { stage_piece(q, slot, kstep); }); };
  // counted wait that leaves `steps` whole k32 steps of this wave's DMA in flight
  auto wait_steps = [&](auto steps_) {
    constexpr int st = decltype(steps_)::value;
    if (na == 4) wait_vmcnt<st * 8>(); else wait_vmcnt<st * 7>();
  };

  f32x4 acc[7][8];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 Af[2][7], Wf[2][8];

  // prologue: steps 0 .. NS-1 requested; step 0 landed and in register set 0
  constexpr int PRE = BPK == 2 ? NS : NS - 1;                      // (one barrier per K-step: the first loop barrier fills slot NS-1 and slot 0)
#pragma unroll
  for (int s = 0; s < PRE; ++s) stage(s, s % nk);
  wait_steps(std::integral_constant<int, PRE - 1>{});
  __builtin_amdgcn_s_barrier();
  static_for<7>([&](auto i) { lds_read128<decltype(i)::value * 1024>(Af[0][decltype(i)::value], a_rd); });
  static_for<8>([&](auto j) { lds_read128<decltype(j)::value * 1024>(Wf[0][decltype(j)::value], w_rd); });
  wait_lgkmcnt<0>();
  SB();

  int rd_slot = 1 % NS;                                           // slot of step s+1
  int st_slot = 0;                                                // slot step s occupied: re-filled with step s+NS
  int st_k = PRE % nk;
  const long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  // one k32 step: [wait + barrier] -> DMA of step s+NS and fragment reads of step s+1 interleaved with the 56 MFMAs of step s
  auto step = [&](auto par_, bool sync, auto inflight_) {
    constexpr int cur = decltype(par_)::value, nxt = cur ^ 1;
    if (sync) {
      wait_steps(inflight_);
      __builtin_amdgcn_s_barrier();
    }
    SB();
    const uint32_t ar = a_rd + rd_slot * SLOT, wrd = w_rd + rd_slot * SLOT;
    const int my_slot = st_slot, my_k = st_k;
    if constexpr (BPK == 2) {
      st_slot = st_slot + 1 == NS ? 0 : st_slot + 1;
      st_k = st_k + 1 == nk ? 0 : st_k + 1;
    } else {
      if (sync) {                                                 // both freed slots re-filled at once: steps s+4 -> slot of s-1, s+5 -> slot of s
        int prev = st_slot == 0 ? NS - 1 : st_slot - 1;
        stage(prev, st_k);
        st_k = st_k + 1 == nk ? 0 : st_k + 1;
        stage(st_slot, st_k);
        st_k = st_k + 1 == nk ? 0 : st_k + 1;
      }
      st_slot = st_slot + 1 == NS ? 0 : st_slot + 1;
    }
    SB();
    static_for<56>([&](auto m_) {
      constexpr int m = decltype(m_)::value, i = m / 8, j = m % 8;
      mfma_acc(acc[i][j], Wf[cur][j], Af[cur][i]);
      if constexpr (BPK == 2 && m % 7 == 0) {                     // the step's DMA spread over its MFMAs: a burst right after the barrier fills the
        SB();                                                     // vector-memory queue and stalls the issuing wave (measured: +500 cycles per k32 step)
        stage_piece(std::integral_constant<int, m / 7>{}, my_slot, my_k);
        SB();
      }
      if constexpr (m % 2 == 1 && m / 2 < 15) {
        constexpr int r = m / 2;
        if constexpr (r < 7) lds_read128<r * 1024>(Af[nxt][r], ar);
        else lds_read128<(r - 7) * 1024>(Wf[nxt][r - 7], wrd);
      }
    });
    SB();
    wait_lgkmcnt<0>();
    SB();
    rd_slot = rd_slot + 1 == NS ? 0 : rd_slot + 1;
  };
  constexpr std::integral_constant<int, 0> P0{};
  constexpr std::integral_constant<int, 1> P1{};
#pragma unroll 1
  for (int s = 0; s < S; s += 2) {
    if constexpr (BPK == 2) {
      step(P0, true, std::integral_constant<int, NS - 2>{});
      step(P1, true, std::integral_constant<int, NS - 2>{});
    } else {
      step(P0, true, std::integral_constant<int, 1>{});
      step(P1, false, std::integral_constant<int, 0>{});
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  wait_vmcnt<0>();
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");     // the last MFMAs retire before the accumulators are read (inline asm: no hazard tracking)
  // (the loop's last step accumulated the fragments of step S, a wrapped re-read of step 0, never: step S's fragments are only READ, not used)
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

template <int NS, int BPK>
static void run(const __bf16* A, const __bf16* W, float* C, float* Cref, long long* cyc, int M, int N, int K, int reps) {
  auto kern = kloop<NS, BPK>;
  const int lds = NS * SLOT;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = (M / BM) * (N / BN);
  hipMemset(C, 0, (size_t)M * N * 4);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, W, C, M, N, K, 1, cyc);
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
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, W, C, M, N, K, reps, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<long long> hc(grid), hr(grid);
  hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), cyc + 2048, grid * 8, hipMemcpyDeviceToHost);
  long long sum = 0, mx = 0, rsum = 0; for (auto v : hc) { sum += v; if (v > mx) mx = v; }
  for (auto v : hr) rsum += v;
  printf("  K loop: %.1f us by the 100-MHz real-time counter -> shader clock %.2f GHz\n", (double)rsum / grid / 100.0, (double)sum / rsum / 10.0);
  const double steps = (double)(K / 64) * reps;
  printf("NS=%d barriers/K-step=%d: max |diff| %.4f, %zu outside tolerance; %.0f cycles per K-step (mean over workgroups; slowest %.0f); launch %.1f us for %d passes\n",
         NS, BPK, worst, bad, (double)sum / grid / steps, (double)mx / steps, best * 1e3, reps);
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
  if (getenv("W4_SAMETILE")) {                                   // L2-resident operands: what does the L2 -> LDS DMA path deliver per CU?
    auto kern = kloop<5, 2>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * SLOT);
    const int grid = (M / BM) * (N / BN);
    std::vector<long long> hc(grid);
    std::vector<long long> hr(grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {                         // ~1.5 s of back-to-back launches per mode: the clock the socket sustains with / without L2 misses
      float ms = 0;
      for (int blk = 0; blk < 8; ++blk) {
        hipEventRecord(e0);
        for (int it = 0; it < 1000; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 5 * SLOT, 0, A, W, C, M, N, K, mode ? -8 : 8, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
      hipMemcpy(hr.data(), cyc + 2048, grid * 8, hipMemcpyDeviceToHost);
      long long sum = 0, rsum = 0; for (auto v : hc) sum += v; for (auto v : hr) rsum += v;
      printf("%s: %.0f cycles per K-step = %.1f B/clk/CU of LDS-DMA; sustained: shader clock %.2f GHz, %.0f TF/s\n",
             mode ? "every workgroup on tile (0,0) [L2 hits]" : "256 distinct tiles", (double)sum / grid / 128, (BM + BN) * 128.0 / ((double)sum / grid / 128),
             (double)sum / rsum / 10.0, 2.0 * M * N * K * 8 * 1000 / (ms * 1e-3) / 1e12);
    }
    return 0;
  }
  if (getenv("W4_SUSTAIN")) {                                    // shader clock under sustained load: back-to-back launches for ~2 s
    auto kern = kloop<5, 2>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * SLOT);
    const int grid = (M / BM) * (N / BN);
    std::vector<long long> hc(grid), hr(grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it <= 10000; ++it) {
      const bool sample = it == 0 || it == 10 || it == 100 || it == 1000 || it == 3000 || it == 10000;
      if (sample) hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 5 * SLOT, 0, A, W, C, M, N, K, 8, cyc);
      if (sample) {
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), cyc + 2048, grid * 8, hipMemcpyDeviceToHost);
        long long sum = 0, rsum = 0; for (auto v : hc) sum += v; for (auto v : hr) rsum += v;
        printf("launch %5d: %.1f us, %.0f cycles per K-step, shader clock %.2f GHz, %.0f TF/s\n", it, ms * 1e3, (double)sum / grid / 128, (double)sum / rsum / 10.0,
               2.0 * M * N * K * 8 / (ms * 1e-3) / 1e12);
      }
    }
    return 0;
  }
  run<5, 2>(A, W, C, Cref, cyc, M, N, K, 8);
  run<5, 2>(A, W, C, Cref, cyc, M, N, K, 32);
  run<4, 2>(A, W, C, Cref, cyc, M, N, K, 8);
  run<5, 1>(A, W, C, Cref, cyc, M, N, K, 8);
  return 0;
}
