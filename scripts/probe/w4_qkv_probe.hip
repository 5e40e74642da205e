// Whole-kernel probe: the four-wave k32-slot-ring loop of w4_kloop_probe.hip on a 224 x (32 * FNW) tile, one tile per workgroup, bias + bf16
// epilogue - the QKV projection [1792 x 1024] x [3072 x 1024]^T as 8 x 32 = 256 tiles of 224 x 96 (FNW = 3), one per CU.  Timed like
// scripts/qkv_tile_probe.py (back-to-back launches cycling 12 weight matrices) and checked against a naive kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/w4_qkv_probe scripts/probe/w4_qkv_probe.hip && /tmp/w4_qkv_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

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
__device__ __forceinline__ int swz(int x) { return (0x78 >> (2 * x)) & 3; }
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u); ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}

template <int NS, int FNW>
__global__ __launch_bounds__(256, 1) void w4_gemm(const __bf16* __restrict__ A, const __bf16* __restrict__ W, const float* __restrict__ bias,
                                                 __bf16* __restrict__ C, int M, int N, int K, long long* stamps) {
  const long long t_0 = __builtin_readcyclecounter();
  constexpr int BM = 224, BN = 32 * FNW, A_BYTES = BM * 64, SLOT = (BM + BN) * 64;
  constexpr int NPIECE = (BM + BN) / 16, PPW = (NPIECE + 3) / 4;   // 1-KiB DMA pieces per k32 step; per wave (round-robin)
  static_assert(NPIECE % 4 == 0, "uniform DMA count per wave");
  constexpr int NM = 7 * FNW, NR = 7 + FNW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int n_tiles = N / BN;
  const int tm = blockIdx.x / n_tiles, tn = blockIdx.x % n_tiles;
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t a_rd = lds0 + (wr * 112 + fr) * 64 + ((fq ^ swz(fr >> 2)) * 16);
  const uint32_t w_rd = lds0 + A_BYTES + (wc * 16 * FNW + fr) * 64 + ((fq ^ swz(fr >> 2)) * 16);
  const int prow = lane >> 2, lc = (lane & 3) ^ swz(lane >> 4);
  // this wave's pieces g = wave + 4 q: g < 14 -> A rows 16 g.., else W rows 16 (g - 14)..; rows past M re-read the last row (never stored)
  const char* src[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int g = wave + 4 * q;
    if (g < BM / 16) src[q] = reinterpret_cast<const char*>(A) + ((long)min(tm * BM + g * 16 + prow, M - 1) * K + lc * 8) * 2;
    else src[q] = reinterpret_cast<const char*>(W) + ((long)(tn * BN + (g - BM / 16) * 16 + prow) * K + lc * 8) * 2;
  }
  const int nk = K / 32;
  auto stage_piece = [&](auto q_, int slot, int kstep) {
    constexpr int q = decltype(q_)::value;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + (long)kstep * 64),
                                     (__attribute__((address_space(3))) void*)(smem + slot * SLOT + (wave + 4 * q) * 1024), 16, 0, 0);
  };
  f32x4 acc[7][FNW];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < FNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 Af[2][7], Wf[2][FNW];
#pragma unroll
  for (int s = 0; s < NS; ++s) static_for<PPW>([&](auto q) { stage_piece(q, s, s); });
  // bias of this lane's columns while the first tiles are on their way
  float4 bq[FNW];
#pragma unroll
  for (int j = 0; j < FNW; ++j) bq[j] = *reinterpret_cast<const float4*>(bias + tn * BN + wc * 16 * FNW + j * 16 + fq * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (prologue only: the bias loads sit behind the DMA in the counter)
  __builtin_amdgcn_s_barrier();
  static_for<7>([&](auto i) { lds_read128<decltype(i)::value * 1024>(Af[0][decltype(i)::value], a_rd); });
  static_for<FNW>([&](auto j) { lds_read128<decltype(j)::value * 1024>(Wf[0][decltype(j)::value], w_rd); });
  wait_lgkmcnt<0>();
  SB();
  const long long t_1 = __builtin_readcyclecounter();
  int rd_slot = 1 % NS, st_slot = 0, st_k = NS;
  auto step = [&](auto par_, bool more) {
    constexpr int cur = decltype(par_)::value, nxt = cur ^ 1;
    wait_vmcnt<(NS - 2) * PPW>();
    __builtin_amdgcn_s_barrier();
    SB();
    const uint32_t ar = a_rd + rd_slot * SLOT, wrd = w_rd + rd_slot * SLOT;
    const int my_slot = st_slot, my_k = min(st_k, nk - 1);           // past the end: re-read the last step (valid memory, never consumed)
    st_slot = st_slot + 1 == NS ? 0 : st_slot + 1;
    ++st_k;
    SB();
    static_for<NM>([&](auto m_) {
      constexpr int m = decltype(m_)::value, i = m / FNW, j = m % FNW;
      mfma_acc(acc[i][j], Wf[cur][j], Af[cur][i]);
      if constexpr (m % 2 == 0 && m / 2 < PPW) { SB(); stage_piece(std::integral_constant<int, m / 2>{}, my_slot, my_k); SB(); }
      constexpr int r = m < 2 * PPW ? (m % 2 == 1 ? m / 2 : -1) : m - PPW;
      if constexpr (r >= 0 && r < NR) {
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
  for (int s = 0; s < nk; s += 2) { step(P0, true); step(P1, true); }
  const long long t_2 = __builtin_readcyclecounter();
  wait_vmcnt<0>();
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int row = tm * BM + wr * 112 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < FNW; ++j) {
      const int col = tn * BN + wc * 16 * FNW + j * 16 + fq * 4;
      f32x4 v = acc[i][j];
      v[0] += bq[j].x; v[1] += bq[j].y; v[2] += bq[j].z; v[3] += bq[j].w;
      if (row < M) *reinterpret_cast<uint2*>(C + (long)row * N + col) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
  if (stamps && tid == 0) {
    wait_vmcnt<0>();
    const long long t_3 = __builtin_readcyclecounter();
    stamps[blockIdx.x * 4 + 0] = t_1 - t_0; stamps[blockIdx.x * 4 + 1] = t_2 - t_1; stamps[blockIdx.x * 4 + 2] = t_3 - t_2;
  }
}

__global__ void naive(const __bf16* A, const __bf16* W, const float* bias, float* C, int M, int N, int K) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= M || col >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(long)row * K + k] * (float)W[(long)col * K + k];
  C[(long)row * N + col] = s + bias[col];
}

int main() {
  const int M = 1792, N = 3072, K = 1024, NL = 12;
  constexpr int NS = 5, FNW = 3;
  std::vector<__bf16> hA((size_t)M * K), hW((size_t)N * K);
  std::vector<float> hb(N);
  srand(1);
  for (auto& v : hA) v = (__bf16)((rand() % 2001 - 1000) / 1000.0f);
  for (auto& v : hW) v = (__bf16)((rand() % 2001 - 1000) / 8000.0f);
  for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.0f;
  __bf16 *A, *W[NL], *C; float *Cref, *b;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&Cref, (size_t)M * N * 4); hipMalloc(&b, N * 4);
  for (int l = 0; l < NL; ++l) { hipMalloc(&W[l], hW.size() * 2); hipMemcpy(W[l], hW.data(), hW.size() * 2, hipMemcpyHostToDevice); }
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(naive, dim3(N / 64, M / 4), dim3(256), 0, 0, A, W[0], b, Cref, M, N, K);
  auto kern = w4_gemm<NS, FNW>;
  const int lds = NS * (224 + 32 * FNW) * 64;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = ((M + 223) / 224) * (N / (32 * FNW));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, W[0], b, C, M, N, K, nullptr);
  hipDeviceSynchronize();
  std::vector<__bf16> h((size_t)M * N); std::vector<float> r((size_t)M * N);
  hipMemcpy(h.data(), C, h.size() * 2, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), Cref, r.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0; size_t bad = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    const double d = std::fabs((double)(float)h[i] - r[i]);
    if (d > worst) worst = d;
    if (d > 3e-2 + 1e-2 * std::fabs(r[i])) ++bad;
  }
  printf("224x%d tiles, %d workgroups, %d KiB LDS: max |diff| %.4f, %zu outside tolerance\n", 32 * FNW, grid, lds / 1024, worst, bad);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int round = 0; round < 5; ++round) {
    for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, W[l], b, C, M, N, K, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 48; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, W[i % NL], b, C, M, N, K, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("round %d: %.2f us per launch (%.0f TF/s)\n", round, ms * 1e3 / 48, 2.0 * M * N * K / (ms * 1e-3 / 48) / 1e12);
  }
  long long* st; hipMalloc(&st, 256 * 4 * 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, W[5], b, C, M, N, K, st);
  hipDeviceSynchronize();
  std::vector<long long> hs(256 * 4);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  double a0 = 0, a1 = 0, a2 = 0;
  for (int i = 0; i < grid; ++i) { a0 += hs[i * 4]; a1 += hs[i * 4 + 1]; a2 += hs[i * 4 + 2]; }
  printf("cycles (mean over workgroups): prologue %.0f, K loop %.0f (%.0f per K-step), epilogue %.0f\n", a0 / grid, a1 / grid, a1 / grid / (K / 64), a2 / grid);
  return 0;
}
