// Shared by the transpose-read backward GEMMs (gemm_bf16_tr.hip: ring kernels; gemm_bf16_trws.hip: the wave-specialised weight-gradient + AdamW kernel):
// the kernel argument block, the LDS reads the compiler must not track, the column-group swizzle of a [k][n] tile.
#pragma once
#include "mode_common.h"

namespace mode {

struct TrParams {
  const uint16_t* A; long lda;
  const uint16_t* W; long ldw; long w_estride;
  void* C; long ldc;
  const int* offsets; int E;              // data gradient, grouped rows (MoE): expert e owns rows [offsets[e], offsets[e+1])
  const int* koffs; long c_gstride;       // weight gradient: blockIdx.z = group, K rows [koffs[z], koffs[z+1])
  const int* w_rows;                      // weight gradient: gather of W's K rows (dispatch permutation)
  int tap_cols; long tap_stride;          // w_rows in taps: output column n belongs to tap n / tap_cols, which reads W's columns n % tap_cols through w_rows + tap * tap_stride
  int M, N, K, m_tiles, n_tiles;
  int split_k; long split_stride;         // data gradient only: blockIdx.y = K-slice, partial sums to C + slice*split_stride (the consumer adds the slabs)
  // EPI = 1, data gradient dH = dY W2 fused with the SwishGLU (+ dropout) backward and the bias-gradient partial sums (train_ops.hip: swiglu_bwd_bias): C is not written
  const uint16_t* P; uint16_t* dP;        // pre-activations / their gradients [M, 2 N] (value | gate)
  uint32_t seed, thresh; float inv_keep;  // expert-dropout stream of this layer
  float* bsum;                            // [m-tile][2 N] column sums of the bf16-rounded dP over the tile's rows (a tile lies inside one expert's segment)
  int* tile_offs;                         // [E + 1] out: first m-tile of every expert (written by workgroup 0: the segment table of the bias-gradient column sum)
  // EPI = 2, weight gradient with the AdamW update in its epilogue (ModeAdamWFuse, include/mode_hip.h): C is not written; the pointers below are
  // already offset to this GEMM's output tensor (group z at + z * c_gstride elements, row pitch ldc)
  float* ad_p; float* ad_m; float* ad_v; uint16_t* ad_lp; float* ad_ema; float* ad_gsq;
  float ad_decay, ad_b1, ad_b2, ad_eps, ad_step_size, ad_inv_bc2_sqrt, ad_gscale, ad_ema_rate;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// LDS reads the compiler does not track (it would otherwise put `s_waitcnt vmcnt(0)` — i.e. the just-issued LDS-DMA of the NEXT tile —
// in front of every compiler-visible LDS read): hand-counted lgkmcnt waits + sched_barriers, as in the forward kernel.
template <int OFF>
__device__ __forceinline__ void lds_tr64(s16x4& dst, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_b128(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int N>
__device__ __forceinline__ void tr_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// f(k): 32-byte column-group swizzle of a [64 k][COLS] tile (COLS = 128: 8 groups per 256-B row; COLS = 64: 4 groups per 128-B row, odd
// rows already sit on the other half of the banks)
template <int COLS>
__device__ __forceinline__ int kn_swz(int row) {
  if constexpr (COLS == 128) return (row & 3) | (((row >> 3) & 1) << 2);
  else return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
}

}  // namespace mode
