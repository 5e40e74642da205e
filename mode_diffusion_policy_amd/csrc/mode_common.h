// Device-side helpers shared by the gfx950 kernels of libmode_hip.so.  CDNA4 only: wave64, MFMA, LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "mode_hip.h"

namespace mode {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kWave = 64;

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -----------------------------------------------------------
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> packed bf16x2 (round-to-nearest-even): one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2_t v;
  v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

// Independent 32-bit dropout stream per (step seed, stream id).  The keep-masks hash (element index XOR stream seed); with consecutive integers
// as per-layer seeds the mask of layer l+1 would be the mask of layer l with its element indices XORed by a small constant (a permutation of
// the same stream).  Stream id = 2*layer (attention dropout) / 2*layer + 1 (expert dropout).  oracle/mode_oracle.py restates this bit for bit.
__host__ __device__ inline uint32_t mode_stream_seed(uint32_t seed, uint32_t stream) {
  uint32_t x = seed ^ (0x9e3779b9u * (stream + 1u));
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;   // lowbias32
  return x;
}

// ---- counter-based dropout (expert MLP): lowbias32 (Wellons) of (element index XOR stream seed); oracle/mode_oracle.py restates it bit for bit
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// keep-mask: element `idx` of stream `seed` survives with probability 1-p (thresh = p * 2^32)
__device__ __forceinline__ bool drop_keep(uint32_t seed, uint64_t idx, uint32_t thresh) {
  return hash_u32(hash_u32((uint32_t)idx ^ seed) + (uint32_t)(idx >> 32) * 0x9e3779b9U) >= thresh;
}

// ---- wave64 reductions via cross-lane shuffles ----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// g * sigmoid(g) with the hardware reciprocal (v_rcp_f32, 1 ulp): `__fdividef` compiles to the full IEEE division sequence on gfx950
// (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup, serialised through VCC) - measured 5.5k cycles for the 56 outputs per lane of a
// 224x256 SwiGLU tile epilogue against ~2k with this form.  Every kernel of the library shares this function (forward = backward = all tile shapes).
__device__ __forceinline__ float silu_f(float g) { return g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)); }
// One SwiGLU output from the value / gate accumulators: the fused-ln_2 row scale and the bias add are ONE explicit fma each.  Left to the
// compiler, `acc * rs + b` is contracted or not depending on the surrounding code, and two tile shapes of the same GEMM then round differently
// (seen: 505 of 14.7 M outputs one bf16 ulp apart) - every forward GEMM kernel must produce bit-identical results (batch-slice consistency).
__device__ __forceinline__ float swiglu_f(float acc_v, float acc_g, float rs, float bv, float bg) {
  return __builtin_fmaf(acc_v, rs, bv) * silu_f(__builtin_fmaf(acc_g, rs, bg));
}
// Sum of squares of a lane's four consecutive columns and the 16-chunk tree of a 64-column group (fused ln_2 producer, MODE_EPI_RESIDUAL_NORM): ONE definition
// with explicit fmas for every kernel that publishes these partial sums - the consumers' row norms, and with them the bf16 outputs, must not depend on which
// kernel a batch size selects (batch-slice consistency).  The tree is the xor-8 / 4 / 2 / 1 butterfly of 16 lanes holding chunks 0..15, read off lane 0.
__device__ __forceinline__ float ss4_f(float x, float y, float z, float w) {
  return __builtin_fmaf(w, w, __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
}
__device__ __forceinline__ float ss16_tree(const float* s) {          // s[j] = chunk j's ss4_f
  const float t0 = s[0] + s[8], t1 = s[1] + s[9], t2 = s[2] + s[10], t3 = s[3] + s[11], t4 = s[4] + s[12], t5 = s[5] + s[13], t6 = s[6] + s[14], t7 = s[7] + s[15];
  return ((t0 + t4) + (t2 + t6)) + ((t1 + t5) + (t3 + t7));
}
// One AdamW element update (torch's single-tensor order: decay the weight, m += (g - m)(1 - b1), v = v b2 + (1 - b2) g g, w -= step_size m / (sqrt(v) / sqrt(bc2) + eps)).
// ONE definition with every product / sum spelled out (explicit fmas, explicitly un-fused multiplies): the streaming optimizer pass (train_ops.hip:
// adamw_kernel) and the weight-gradient GEMM's fused epilogue (gemm_bf16_tr.hip, EPI = 2) must round alike - left to the compiler, `v * b2 + (1 - b2) * g * g`
// was contracted differently in the two kernels (seen: ~0.5 % of the second moments one ulp apart after the second step).
__device__ __forceinline__ void adamw_update_f(float& w, float& m, float& v, float gr, float decay, float b1, float b2, float eps, float step_size, float inv_bc2_sqrt) {
  const float wd = __fmul_rn(w, decay);
  m = __builtin_fmaf(__fsub_rn(gr, m), __fsub_rn(1.f, b1), m);
  v = __builtin_fmaf(v, b2, __fmul_rn(__fmul_rn(__fsub_rn(1.f, b2), gr), gr));
  const float denom = __builtin_fmaf(__fsqrt_rn(v), inv_bc2_sqrt, eps);
  w = __builtin_fmaf(-step_size, __fdiv_rn(m, denom), wd);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// XCD-aware, bijective remap of a linear workgroup id: hardware places block b on XCD b % 8 (observed, speed only), so give
// each XCD one contiguous chunk of the logical tile space and neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

// Kernel argument block shared by the forward bf16 GEMM kernels (gemm_bf16.hip, gemm_bf16_p256.hip)
struct GemmParams {
  const uint16_t* A; long lda;
  const uint16_t* W; long ldw; long w_estride;
  const float* bias; long bias_estride;
  const float* resid; long ldr;
  void* C; long ldc;
  const int* a_rows; const int* offsets; int E;
  int M, N, K, m_tiles, n_tiles;
  int split_k; long split_stride;     // split-K: blockIdx.y = K-slice, output slab = C + slice*split_stride elements
  const int* koffs; long c_gstride;   // K-groups: blockIdx.z = group, K range [koffs[z], koffs[z+1])
  int group_m;                        // m-tiles per rasterisation group (0 = kernel default)
  int identity_rows;                  // MODE_GEMM_IDENTITY_ROWS: a_rows[offsets[e] + i] == i, the gather may be computed instead of loaded
  // fused ln_2 (MODE_EPI_RESIDUAL_NORM producer / MODE_EPI_SWIGLU consumer): see include/mode_hip.h
  uint16_t* C2; long ldc2; const float* gain; float* ss_out;   // producer: bf16((acc+resid)*gain[n]) and per-64-column row sums of squares
  const float* ss_in; int ss_n; float ss_eps;                   // consumer: acc rows scaled by 1/max(sqrt(sum ss_in[row][0..ss_n)) * K^-1/2, eps)
};

// Sum of the per-64-column partial sums of squares of one row (fused ln_2), ascending j.  The first 16 (D <= 1024) are fetched by independent
// (index-clamped) loads so that one memory round trip covers them; producer: gemm_bf16.hip MODE_EPI_RESIDUAL_NORM.
__device__ __forceinline__ float sum_row_partials(const float* __restrict__ sp, int n) {
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = sp[min(j, n - 1)];      // clamped index + select below: branch-free, all loads in flight together
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += j < n ? v[j] : 0.f;
  for (int j = 16; j < n; ++j) s += sp[j];
  return s;
}

// Wave-cooperative form for rows that carry MORE than 16 partials (the small-batch chain: the weight-streaming c_proj publishes one partial per
// 16 columns, D/16 = 64 per row): lane j fetches partial j (+64, ...), butterfly sum — every lane of the wave returns the same value, and every
// consumer of such rows (up-projection prologue, combine, head) uses this one function so that they all see the same norm.  n <= 16 keeps the
// ascending-order sum above (the tiled chain's bit pattern).  All 64 lanes must call it together.
__device__ __forceinline__ float sum_row_partials_wave(const float* __restrict__ sp, int n, int lane) {
  if (n <= 16) return sum_row_partials(sp, n);
  float s = 0.f;
  for (int j = lane; j < n; j += 64) s += sp[j];
  return wave_sum(s);
}

// Rows up to which the bf16 GEMM uses the weight-streaming kernel of gemm_bf16_skinny.hip ("gemm_skinny_rows" option; 0 = off)
extern int g_gemm_skinny_rows;
extern int g_gemm_cfg;

#define MODE_LAUNCH_CHECK()                                  \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return (int)e__;                  \
  } while (0)

// ---- host: raise a kernel's dynamic-LDS limit once PER DEVICE ---------------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: a process that drives several GPUs (a threaded replica
// server, a test that hops devices) must set it on each of them, or the first > 64-KiB-LDS launch on the second device fails.  One instance per
// kernel instantiation (a function-local `static LdsLimitOnce`); a bit per device ordinal, set after the attribute call succeeded.  Thread-safe:
// two racing threads at worst both make the (idempotent) call.  Devices >= kMaxDevices are refused (MODE_ERR_UNSUPPORTED) rather than aliased.
constexpr int kMaxDevices = 64;
struct LdsLimitOnce {
  std::atomic<uint64_t> done{0};
  int ensure(const void* kern, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return MODE_ERR_UNSUPPORTED;
    const uint64_t bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return MODE_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return MODE_OK;
  }
};

}  // namespace mode
