// Implicit-GEMM k x k convolution on channels_last bf16 activations for gfx950 - forward and data gradient as ONE product over the filter taps:
//
//   C[m, n] = sum over taps t, channels c of  A[idx[t][m], c] * W_t[c, n]        (idx[t][m] = -1: the tap falls outside the image, a zero row)
//
//   forward        A = X [pixels, Cin], W = the channels_last weight [Cout][taps][Cin] read K-contiguously (k = t * Cin + c), C = Y [out pixels, Cout]
//   data gradient  A = dY [out pixels, Cout], W = the SAME memory read as [k = (t, cout)][n = cin] (row stride taps * Cin, tap offset t * Cin:
//                  MODE_GEMM_W_KN), C = dX [pixels, Cin]; idx[t][m] = the output pixel whose tap t read input pixel m
//
// i.e. mode_gemm with its row gather `a_rows` IN TAPS (ModeGemmDesc.a_tap_cols / a_rows_tap_stride, ABI 10): the A tile of a K-step belongs to one tap
// (a_tap_cols % 64 == 0) and is DMA'd from that tap's rows.  The reference gets these products from cuDNN / MIOpen through F.conv2d
// (mode/models/perceptual_encoders/pretrained_resnets.py:29, resnets.py:96); MIOpen's implicit-GEMM kernels run the ResNet-50 3 x 3 shapes at 150-300 TF/s
// plus zero / cast helper launches around their split-K variants (scripts/conv3x3_miopen_probe.py).
//
// Kernel: the 128 x BN ring tile of the library's other GEMMs - 4 wave64 as 2 x 2, BK = 64, two-slot LDS ring filled by `global_load_lds_dwordx4` (XOR-swizzled
// lane-linear images), v_mfma_f32_16x16x32_bf16 with swapped operands, output tile through LDS for 16-byte coalesced stores; the gather indices of K-step kt+2
// are requested while K-step kt+1 is staged (one vmcnt(0) per K-step covers both, two workgroups per CU hide each other's round trips).
#include "mode_common.h"
#include <type_traits>

namespace mode {

__device__ __attribute__((aligned(256))) uint16_t g_conv_zero_row[128];      // 256 B of zeros: DMA source of out-of-image taps

struct ConvGemmParams {
  const uint16_t* A; long lda;
  const int* idx; long idx_tstride;        // [taps][M]
  const uint16_t* W; long ldw;
  uint16_t* C; long ldc;
  int M, N, taps, tap_k, m_tiles, n_tiles;
};

namespace cg {
typedef short s16x4 __attribute__((ext_vector_type(4)));
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
// f(k): 32-byte column-group swizzle of a [64 k][COLS] tile (the image of gemm_bf16_tr.hip)
template <int COLS>
__device__ __forceinline__ int kn_swz(int row) {
  if constexpr (COLS == 128) return (row & 3) | (((row >> 3) & 1) << 2);
  else return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
}
}  // namespace cg

template <bool W_KN, int BN>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvGemmParams p) {
  using namespace cg;
  constexpr int BM = 128, BKT = 64, TM = 64, TN = BN / 2, FM = 4, FN = TN / 16;
  constexpr int A_BYTES = BM * BKT * 2, W_BYTES = BN * BKT * 2, STAGE_BYTES = A_BYTES + W_BYTES, NS = 2;
  constexpr int W_ROW = BN * 2;                                       // W_KN image: bytes per k-row
  constexpr int RPP = 1024 / W_ROW, NPW = (BKT / RPP) / 4, CHW = BN / 8; // W_KN: k-rows per 1-KiB DMA piece, pieces per wave, 16-B chunks per row
  constexpr int PB = BN / 32;                                         // forward: 8-row pieces of the [BN rows][64 k] weight tile per wave
  constexpr int CROW = BN * 2, CPR = CROW / 16, CSWZ = (CPR < 16 ? CPR : 16) - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nblk = p.m_tiles * p.n_tiles;
  const int sb = xcd_remap(blockIdx.x, nblk);
  const int mt = sb / p.n_tiles, nt = sb - mt * p.n_tiles;             // n fastest: the workgroups that gather the same A rows are neighbours on one XCD
  const int row0 = mt * BM, row_end = min(p.M, row0 + BM);
  const int n0 = nt * BN;
  const int nk = p.taps * p.tap_k / BKT;

  // ---- DMA sources
  const int r8 = lane >> 3, lchunk = (lane & 7) ^ r8;                  // [rows][64 k] images: 8-row pieces, chunk ^ (row & 7)
  int arow_m[4];                                                      // output rows behind this lane's four A pieces (clamped: rows past M are never stored)
#pragma unroll
  for (int q = 0; q < 4; ++q) arow_m[q] = min(row0 + (wave * 4 + q) * 8 + r8, row_end - 1);
  [[maybe_unused]] const uint16_t* b_src[PB];                          // forward: K-contiguous weight rows
  [[maybe_unused]] int kn_col_w[NPW > 0 ? NPW : 1];                    // W_KN: logical column of this lane's chunk, per piece
  const int krw = lane / CHW, pcw = lane % CHW;
  if constexpr (W_KN) {
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const int c = pcw ^ (kn_swz<BN>((wave * NPW + q) * RPP + krw) << 1);
      kn_col_w[q] = min(n0 + c * 8, p.N - 8);
    }
  } else {
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int n = min(n0 + (wave * PB + q) * 8 + r8, p.N - 1);
      b_src[q] = p.W + (long)n * p.ldw + lchunk * 8;
    }
  }
  int aidx[4] = {0, 0, 0, 0};                                          // gathered source rows of the NEXT tile to stage
  auto load_idx = [&](int kt) {
    if (kt < nk) {
      const int tap = kt * BKT / p.tap_k;
      const int* t = p.idx + (long)tap * p.idx_tstride;
#pragma unroll
      for (int q = 0; q < 4; ++q) aidx[q] = t[arow_m[q]];
    }
  };
  auto stage = [&](int slot, int kt) {
    char* base = smem + slot * STAGE_BYTES;
    const int k0 = kt * BKT;
    const int tap = k0 / p.tap_k, ck = k0 - tap * p.tap_k;            // tap of this K-step, first channel inside it
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int P = wave * 4 + q;
      const uint16_t* src = p.A + (long)aidx[q] * p.lda + ck + lchunk * 8;
      src = aidx[q] < 0 ? g_conv_zero_row + (lane & 7) * 8 : src;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + P * 1024), 16, 0, 0);
    }
    if constexpr (W_KN) {
#pragma unroll
      for (int q = 0; q < NPW; ++q) {
        const int P = wave * NPW + q;
        const long r = ck + P * RPP + krw;                             // weight row (output channel) of this k
        const uint16_t* src = p.W + r * p.ldw + (long)tap * p.N + kn_col_w[q];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(base + A_BYTES + P * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int P = wave * PB + q;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + k0),
                                         (__attribute__((address_space(3))) void*)(base + A_BYTES + P * 1024), 16, 0, 0);
      }
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int sw = fr & 7;
  const uint32_t a_off = (wm * TM + fr) * 128;
  [[maybe_unused]] uint32_t tr_w[FN];
  [[maybe_unused]] uint32_t b_off = 0;
  if constexpr (W_KN) {
    const int fsw_a = (fr >> 2) | ((fq & 1) << 2);
    const int fsw_w = BN == 128 ? fsw_a : (((fr >> 3) & 1) | ((fq & 1) << 1));
#pragma unroll
    for (int t = 0; t < FN; ++t)
      tr_w[t] = lds0 + A_BYTES + (fq * 8 + (fr >> 2)) * W_ROW + (fr & 1) * 8 + ((((((wn * FN + t) ^ fsw_w) << 1) | ((fr >> 1) & 1))) << 4);
  } else {
    b_off = A_BYTES + (wn * TN + fr) * 128;
  }

  load_idx(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stage(0, 0);
  load_idx(1);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // tile kt landed (this wave's pieces), the indices of tile kt+1 arrived
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) {
      stage(slot ^ 1, kt + 1);
      load_idx(kt + 2);
    }
    const uint32_t so = slot * STAGE_BYTES;
    bf16x8 fa[2][FM];
    [[maybe_unused]] bf16x8 fb[2][FN];
    [[maybe_unused]] s16x4 wlo[2][FN], whi[2][FN];
    auto read_half = [&](auto KH) {
      constexpr int kh = decltype(KH)::value;
      const uint32_t ab = lds0 + a_off + so + (((fq + kh * 4) ^ sw) * 16);
      lds_b128<0>(fa[kh][0], ab); lds_b128<2048>(fa[kh][1], ab); lds_b128<4096>(fa[kh][2], ab); lds_b128<6144>(fa[kh][3], ab);
      if constexpr (W_KN) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          lds_tr64<kh * 32 * W_ROW>(wlo[kh][j], tr_w[j] + so);
          lds_tr64<kh * 32 * W_ROW + 4 * W_ROW>(whi[kh][j], tr_w[j] + so);
        }
      } else {
        const uint32_t bb = lds0 + b_off + so + (((fq + kh * 4) ^ sw) * 16);
        lds_b128<0>(fb[kh][0], bb);
        if constexpr (FN > 1) lds_b128<2048>(fb[kh][1], bb);
        if constexpr (FN > 2) { lds_b128<4096>(fb[kh][2], bb); lds_b128<6144>(fb[kh][3], bb); }
      }
    };
    auto mma_half = [&](auto KH) {
      constexpr int kh = decltype(KH)::value;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          bf16x8 w;
          if constexpr (W_KN) w = join8(wlo[kh][j], whi[kh][j]);
          else w = fb[kh][j];
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, fa[kh][i], acc[i][j], 0, 0, 0);   // swapped: D[n][m]
        }
    };
    read_half(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_half(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    mma_half(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma_half(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    slot ^= 1;
  }

  // ---- epilogue: accumulators -> swizzled LDS tile -> coalesced 16-byte stores
  const int rows_valid = row_end - row0;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int rl = wm * TM + i * 16 + fr;
    char* crow = smem + rl * CROW;
    const int rsw = rl & CSWZ;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int bb = (wn * TN + j * 16 + fq * 4) * 2;
      char* dst = crow + ((((bb >> 4) ^ rsw) << 4) | (bb & 15));
      const f32x4 v = acc[i][j];
      *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int c = tid; c < BM * CPR; c += 256) {
    const int rl = c / CPR, ch = c % CPR;
    const int n = n0 + ch * 8;
    if (rl >= rows_valid || n >= p.N) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(smem + rl * CROW + ((ch ^ (rl & CSWZ)) << 4));
    *reinterpret_cast<uint4*>(p.C + (long)(row0 + rl) * p.ldc + n) = v;
  }
}

template <bool W_KN, int BN>
static int conv_launch(ConvGemmParams p, hipStream_t s) {
  p.m_tiles = (p.M + 127) / 128;
  p.n_tiles = (p.N + BN - 1) / BN;
  constexpr size_t lds = 2 * (128 * 64 * 2 + (size_t)BN * 64 * 2);
  auto kern = conv_gemm_kernel<W_KN, BN>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.m_tiles * p.n_tiles), dim3(256), lds, s, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

// mode_gemm with a_rows in taps (ModeGemmDesc.a_tap_cols > 0): validated here
int gemm_bf16_conv_launch(const ModeGemmDesc* d, hipStream_t s) {
  const bool w_kn = (d->flags & MODE_GEMM_W_KN) != 0;
  if (d->dtype != MODE_BF16 || d->out_dtype != MODE_BF16 || d->epilogue != MODE_EPI_NONE || (d->flags & ~MODE_GEMM_W_KN)) return MODE_ERR_UNSUPPORTED;
  if (!d->a_rows || d->a_tap_cols <= 0 || d->a_tap_cols % 64 || d->K <= 0 || d->K % d->a_tap_cols) return MODE_ERR_BAD_ARG;
  if (d->expert_offsets || d->k_group_offsets || d->w_rows || d->split_k > 1 || d->bias || d->resid) return MODE_ERR_UNSUPPORTED;
  if (d->N % 8 || d->N < 8 || d->lda % 8 || d->ldw % 8 || d->ldc % 8 || (((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->C) & 15)) return MODE_ERR_UNSUPPORTED;
  if (d->M <= 0) return MODE_OK;
  ConvGemmParams p;
  p.A = (const uint16_t*)d->A; p.lda = d->lda; p.idx = d->a_rows; p.idx_tstride = d->a_rows_tap_stride;
  p.W = (const uint16_t*)d->W; p.ldw = d->ldw; p.C = (uint16_t*)d->C; p.ldc = d->ldc;
  p.M = d->M; p.N = d->N; p.tap_k = d->a_tap_cols; p.taps = d->K / d->a_tap_cols; p.m_tiles = p.n_tiles = 0;
  const long t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
  const bool wide = d->N % 128 == 0 && t128 >= 384;              // enough 128-wide tiles for ~1.5 workgroups per CU; otherwise twice the workgroups
  if (w_kn) return wide ? conv_launch<true, 128>(p, s) : conv_launch<true, 64>(p, s);
  return wide ? conv_launch<false, 128>(p, s) : conv_launch<false, 64>(p, s);
}

}  // namespace mode
