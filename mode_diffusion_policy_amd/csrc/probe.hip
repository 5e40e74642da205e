// Measurement aid, not on the denoising path: register-resident back-to-back v_mfma_f32_16x16x32_bf16 on every SIMD of the device.
// bench.py times a burst of these launches to report what the socket SUSTAINS at its power cap next to the datasheet peak its fractions are
// quoted against (LABNOTES.md section 8: 2.06 PF/s at 2.1 GHz and 1314 W on the boxes of round 2, not 2.5 PF/s).
#include "mode_common.h"

namespace mode {

__device__ __forceinline__ uint32_t burn_hash(uint32_t x) {       // lowbias32
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(512) void mfma_burn_kernel(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // bf16 values uniform in [-1, 1) - signs, exponents and mantissas all toggle (operand activity matters for power); the accumulators are reset
    // every 64 iterations so the sums stay finite over any iteration count
    uint32_t wa[4], wb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t h = burn_hash(seed[0] + (uint32_t)(lane * 131 + i * 17 + j)), g = burn_hash(h + 0x9e3779b9U);
      auto bf = [](uint32_t r) { return __float_as_uint((float)(r & 0xffff) * (1.0f / 32768.0f) - 1.0f) >> 16; };
      wa[j] = bf(h) | (bf(h >> 16) << 16);
      wb[j] = bf(g) | (bf(g >> 16) << 16);
    }
    __builtin_memcpy(&a[i], wa, 16); __builtin_memcpy(&b[i], wb, 16);
  }
  f32x4 acc[16];
  float s = 0.f;
  for (int it = 0; it < iters; ++it) {
    if ((it & 63) == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { if (it) s += acc[i][0] + acc[i][3]; acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  if (out) out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace mode

using namespace mode;

extern "C" int mode_probe_mfma_burn(const uint32_t* seed, float* out, int workgroups, int iters, double* flop_per_launch, void* stream) {
  if (!seed || workgroups <= 0 || iters <= 0) return MODE_ERR_BAD_ARG;
  hipLaunchKernelGGL(mfma_burn_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, seed, out, iters);
  MODE_LAUNCH_CHECK();
  if (flop_per_launch) *flop_per_launch = (double)workgroups * 8.0 * iters * 16.0 * (2.0 * 16 * 16 * 32);
  return MODE_OK;
}
