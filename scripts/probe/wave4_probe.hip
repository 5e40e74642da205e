// Micro-benchmark for a 4-wave (one wave per SIMD, 512-register budget) formulation of the 224x256x64 K-step of gemm_bf16_pp.hip:
// each wave owns 112 rows x 128 columns (7 x 8 accumulator fragments = 224 registers), reads its 7 + 8 operand fragments per k32 from LDS and
// issues 56 MFMAs per k32; operands stay resident in LDS (no DMA: this measures the MFMA / LDS-read / barrier skeleton only), ONE barrier per
// K-step.  Prints cycles per K-step; the 8-wave ping-pong loop measures ~2420 (1792 = MFMA issue only).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wave4_probe scripts/probe/wave4_probe.hip && /tmp/wave4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int BARRIERS>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int ksteps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // A: 224 rows x 128 B, W: 256 rows x 128 B (one K-step, reused)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  for (int i = threadIdx.x; i < (224 + 256) * 128 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x4 acc[7][8];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* ab = smem + (wr * 112 + fr) * 128 + ((fq ^ (fr & 7)) * 16);
  const char* wb = smem + 224 * 128 + (wc * 128 + fr) * 128 + ((fq ^ (fr & 7)) * 16);
  bf16x8 A[2][7], W[2][8];
  auto rd = [&](int set, int kh) {
#pragma unroll
    for (int i = 0; i < 7; ++i) A[set][i] = *reinterpret_cast<const bf16x8*>(ab + i * 2048 + (kh ? 64 : 0));
#pragma unroll
    for (int j = 0; j < 8; ++j) W[set][j] = *reinterpret_cast<const bf16x8*>(wb + j * 2048 + (kh ? 64 : 0));
  };
  rd(0, 0);
  const long long t0 = __builtin_readcyclecounter();
  for (int kt = 0; kt < ksteps; ++kt) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      rd((kh + 1) & 1, (kh + 1) & 1);                           // next k32's fragments, in flight under this k32's MFMAs
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[kh & 1][j], A[kh & 1][i], acc[i][j], 0, 0, 0);
    }
    if (BARRIERS == 1) __syncthreads();
    if (BARRIERS == 2) __builtin_amdgcn_s_barrier();            // raw barrier: no counter wait in front of it
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int lds = (224 + 256) * 128;
  const int ks = 64;
  for (int b = 0; b < 3; ++b) {
    for (int rep = 0; rep < 3; ++rep) {
      if (b == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), lds, 0, out, cyc, ks);
      else if (b == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), lds, 0, out, cyc, ks);
      else hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), lds, 0, out, cyc, ks);
      hipDeviceSynchronize();
    }
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    long long sum = 0; for (auto v : h) sum += v;
    printf("4 waves x (112x128) per CU, %s: %.0f cycles per K-step (64 k) - MFMA issue alone = 1792\n", b == 1 ? "__syncthreads per K-step" : b == 2 ? "raw s_barrier per K-step" : "no barrier", (double)sum / 256 / ks);
  }
  return 0;
}
