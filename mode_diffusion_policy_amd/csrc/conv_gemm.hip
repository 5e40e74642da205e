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
  // fused inference epilogue (FUSE): y = post_film(relu(pre_film(batch_norm_eval(acc)) + residual)), the expression of csrc/encoder_ops.hip's fused pass
  const float* bn_mean; const float* bn_var; const float* bn_w; const float* bn_b; float bn_eps;
  const uint16_t* res; long ldr; int relu;
  const float* pre_g; const float* pre_b; const float* post_g; const float* post_b; int rows_per_sample;
  // training forward: per-m-tile column sums / sums of squares of the output AS STORED (bf16-rounded) - the partial statistics of the BatchNorm that follows
  float* stat_sum; float* stat_sq;         // [m_tiles][N]
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

// NS = 2: one vmcnt(0) per K-step, two workgroups per CU hide each other's round trips (the training shapes: thousands of tiles).  NS = 3: two K-steps in flight
// under a COUNTED wait - for grids that do not even fill the part once (the rollout's batch sizes), where a K-step would otherwise cost a full memory round trip
// (stage-4 3 x 3 at B = 32: 72 K-steps x ~2 us).  Index loads are issued BEFORE the DMA of the same iteration so that the counted wait (which leaves the newest
// K-step's DMA instructions in flight; loads retire in order) covers them.
template <bool W_KN, int BN, bool FUSE = false, int NS = 2>
__global__ __launch_bounds__(256, NS == 2 ? 2 : 1) void conv_gemm_kernel(const ConvGemmParams p) {
  using namespace cg;
  constexpr int BM = 128, BKT = 64, TM = 64, TN = BN / 2, FM = 4, FN = TN / 16;
  constexpr int A_BYTES = BM * BKT * 2, W_BYTES = BN * BKT * 2, STAGE_BYTES = A_BYTES + W_BYTES;
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
  auto load_idx = [&](int kt, int (&ix)[4]) {
    if (kt < nk) {
      if (p.idx) {
        const int tap = kt * BKT / p.tap_k;
        const int* t = p.idx + (long)tap * p.idx_tstride;
        // hand-issued (the compiler must not know these loads are outstanding: it would put `s_waitcnt vmcnt(0)` - i.e. the DMA of the K-steps in flight -
        // in front of their first use; the loop's own waits below cover them and pin the uses behind the wait)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int* a = t + arow_m[q];
          asm volatile("global_load_dword %0, %1, off" : "=v"(ix[q]) : "v"(a) : "memory");
        }
      } else {                                                       // no table: one tap, output row m reads input row m (1 x 1 / stride 1)
#pragma unroll
        for (int q = 0; q < 4; ++q) ix[q] = arow_m[q];
      }
    }
  };
  auto stage = [&](int slot, int kt, const int (&aidx)[4]) {
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

  constexpr int LOADS = 4 + (W_KN ? NPW : PB);                        // DMA instructions per wave and K-step
  int slot = 0;
  int ia[4] = {0, 0, 0, 0}, ib[4] = {0, 0, 0, 0};
#define CG_WAIT_IDX(N, ix) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ix[0]), "+v"(ix[1]), "+v"(ix[2]), "+v"(ix[3]) : "n"(N) : "memory")
  load_idx(0, ia);
  CG_WAIT_IDX(0, ia);
  stage(0, 0, ia);
  load_idx(1, ia);
  if constexpr (NS == 3) {
    CG_WAIT_IDX(0, ia);
#pragma unroll
    for (int q = 0; q < 4; ++q) ib[q] = ia[q];
    load_idx(2, ia);                                                   // requested BEFORE the DMA of K-step 1: the loop's counted wait covers it
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 1) stage(1, 1, ib);
  }
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (NS == 2) {
      CG_WAIT_IDX(0, ia);                                              // tile kt landed (this wave's pieces), the indices of tile kt+1 arrived
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < nk) {
        stage(slot ^ 1, kt + 1, ia);
        load_idx(kt + 2, ia);
      }
    } else {
      // in flight, oldest first: DMA kt | indices kt+2 | DMA kt+1.  Needed now: tile kt and the indices of kt+2 -> the newest K-step's DMA stays in flight
      if (kt + 1 < nk) CG_WAIT_IDX(LOADS, ia);
      else CG_WAIT_IDX(0, ia);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) ib[q] = ia[q];                         // indices of K-step kt+2 (arrived)
      load_idx(kt + 3, ia);                                            // requested BEFORE this iteration's DMA
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 2 < nk) stage(slot == 0 ? 2 : slot - 1, kt + 2, ib);
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
    if constexpr (NS == 2) slot ^= 1; else slot = (slot + 1 == 3) ? 0 : slot + 1;
  }

#undef CG_WAIT_IDX
  // ---- fused inference epilogue, in registers: a lane owns columns n .. n+3 of row m per (i, j)
  if constexpr (FUSE) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nc = min(n0 + wn * TN + j * 16 + fq * 4, p.N - 4);                 // (columns past N are never stored)
      float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bn_mean) {                                                              // eval-mode BatchNorm folded: the expressions of chan_params / bn_prepare_kernel
        const float4 mu = *reinterpret_cast<const float4*>(p.bn_mean + nc), va = *reinterpret_cast<const float4*>(p.bn_var + nc);
        const float mu_[4] = {mu.x, mu.y, mu.z, mu.w}, va_[4] = {va.x, va.y, va.z, va.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[r] = 1.0f / sqrtf(va_[r] + p.bn_eps);
        if (p.bn_w) { const float4 w4 = *reinterpret_cast<const float4*>(p.bn_w + nc); sc[0] *= w4.x; sc[1] *= w4.y; sc[2] *= w4.z; sc[3] *= w4.w; }
        if (p.bn_b) { const float4 b4 = *reinterpret_cast<const float4*>(p.bn_b + nc); sh[0] = b4.x; sh[1] = b4.y; sh[2] = b4.z; sh[3] = b4.w; }
#pragma unroll
        for (int r = 0; r < 4; ++r) sh[r] -= mu_[r] * sc[r];
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int mc = min(row0 + wm * TM + i * 16 + fr, row_end - 1);
        const long fb = (long)(mc / p.rows_per_sample) * p.N + nc;                    // FiLM parameters are per (sample, channel)
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = __builtin_fmaf(acc[i][j][r], sc[r], sh[r]);
        if (p.pre_g) {
          const float4 g4 = *reinterpret_cast<const float4*>(p.pre_g + fb), b4 = *reinterpret_cast<const float4*>(p.pre_b + fb);
          t[0] = __builtin_fmaf(g4.x, t[0], b4.x); t[1] = __builtin_fmaf(g4.y, t[1], b4.y); t[2] = __builtin_fmaf(g4.z, t[2], b4.z); t[3] = __builtin_fmaf(g4.w, t[3], b4.w);
        }
        if (p.res) {
          const uint2 rr = *reinterpret_cast<const uint2*>(p.res + (long)mc * p.ldr + nc);
          t[0] += bf16_bits_to_f32(rr.x & 0xffff); t[1] += bf16_bits_to_f32(rr.x >> 16); t[2] += bf16_bits_to_f32(rr.y & 0xffff); t[3] += bf16_bits_to_f32(rr.y >> 16);
        }
        if (p.relu) { t[0] = fmaxf(t[0], 0.f); t[1] = fmaxf(t[1], 0.f); t[2] = fmaxf(t[2], 0.f); t[3] = fmaxf(t[3], 0.f); }
        if (p.post_g) {
          const float4 g4 = *reinterpret_cast<const float4*>(p.post_g + fb), b4 = *reinterpret_cast<const float4*>(p.post_b + fb);
          t[0] = __builtin_fmaf(1.f + g4.x, t[0], b4.x); t[1] = __builtin_fmaf(1.f + g4.y, t[1], b4.y); t[2] = __builtin_fmaf(1.f + g4.z, t[2], b4.z);
          t[3] = __builtin_fmaf(1.f + g4.w, t[3], b4.w);
        }
        acc[i][j] = f32x4{t[0], t[1], t[2], t[3]};
      }
    }
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
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  const bool stats = p.stat_sum != nullptr;
  for (int c = tid; c < BM * CPR; c += 256) {
    const int rl = c / CPR, ch = c % CPR;                              // ch = tid % CPR for every pass: a thread walks rows of ONE 8-column chunk
    const int n = n0 + ch * 8;
    if (rl >= rows_valid || n >= p.N) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(smem + rl * CROW + ((ch ^ (rl & CSWZ)) << 4));
    *reinterpret_cast<uint4*>(p.C + (long)(row0 + rl) * p.ldc + n) = v;
    if (stats) {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16_bits_to_f32(w[j] & 0xffff), b = bf16_bits_to_f32(w[j] >> 16);
        ssum[2 * j] += a; ssum[2 * j + 1] += b;
        ssq[2 * j] = __builtin_fmaf(a, a, ssq[2 * j]); ssq[2 * j + 1] = __builtin_fmaf(b, b, ssq[2 * j + 1]);
      }
    }
  }
  if (stats) {                                                         // (uniform) the chunk's partners: lanes CPR apart in the wave, then the four waves through LDS
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (CPR == 8) { ssum[j] += __shfl_xor(ssum[j], 8, 64); ssq[j] += __shfl_xor(ssq[j], 8, 64); }
      ssum[j] += __shfl_xor(ssum[j], 16, 64); ssum[j] += __shfl_xor(ssum[j], 32, 64);
      ssq[j] += __shfl_xor(ssq[j], 16, 64); ssq[j] += __shfl_xor(ssq[j], 32, 64);
    }
    __syncthreads();                                                   // the output tile has been read by everyone
    float* red = reinterpret_cast<float*>(smem);                        // [4 waves][CPR][16]
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { red[(wave * CPR + lane) * 16 + j] = ssum[j]; red[(wave * CPR + lane) * 16 + 8 + j] = ssq[j]; }
    }
    __syncthreads();
    if (tid < CPR && n0 + tid * 8 < p.N) {
      float* o1 = p.stat_sum + (long)mt * p.N + n0 + tid * 8;
      float* o2 = p.stat_sq + (long)mt * p.N + n0 + tid * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o1[j] = ((red[(0 * CPR + tid) * 16 + j] + red[(1 * CPR + tid) * 16 + j]) + red[(2 * CPR + tid) * 16 + j]) + red[(3 * CPR + tid) * 16 + j];
        o2[j] = ((red[(0 * CPR + tid) * 16 + 8 + j] + red[(1 * CPR + tid) * 16 + 8 + j]) + red[(2 * CPR + tid) * 16 + 8 + j]) + red[(3 * CPR + tid) * 16 + 8 + j];
      }
    }
  }
}

int pp_num_cus();   // gemm_bf16_pp.hip

template <bool W_KN, int BN, bool FUSE, int NS>
static int conv_launch_ns(ConvGemmParams p, hipStream_t s) {
  constexpr size_t lds = NS * (128 * 64 * 2 + (size_t)BN * 64 * 2);
  auto kern = conv_gemm_kernel<W_KN, BN, FUSE, NS>;
  static LdsLimitOnce lds_once;
  if (lds > 48 * 1024) {
    const int rc = lds_once.ensure(reinterpret_cast<const void*>(kern), (int)lds);
    if (rc != MODE_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(p.m_tiles * p.n_tiles), dim3(256), lds, s, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

int g_conv_ns = 0;   // "conv_ns" option: 0 = auto (three-slot ring when the grid has fewer than two workgroups per CU), 2 / 3 = forced

template <bool W_KN, int BN, bool FUSE = false>
static int conv_launch(ConvGemmParams p, hipStream_t s) {
  p.m_tiles = (p.M + 127) / 128;
  p.n_tiles = (p.N + BN - 1) / BN;
  const bool deep = g_conv_ns ? g_conv_ns == 3 : (long)p.m_tiles * p.n_tiles < 2L * pp_num_cus();
  return deep ? conv_launch_ns<W_KN, BN, FUSE, 3>(p, s) : conv_launch_ns<W_KN, BN, FUSE, 2>(p, s);
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
  p.bn_mean = p.bn_var = p.bn_w = p.bn_b = nullptr; p.bn_eps = 0.f; p.res = nullptr; p.ldr = 0; p.relu = 0;
  p.pre_g = p.pre_b = p.post_g = p.post_b = nullptr; p.rows_per_sample = 1; p.stat_sum = p.stat_sq = nullptr;
  const long t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
  const bool wide = d->N % 128 == 0 && t128 >= 384;              // enough 128-wide tiles for ~1.5 workgroups per CU; otherwise twice the workgroups
  if (w_kn) return wide ? conv_launch<true, 128>(p, s) : conv_launch<true, 64>(p, s);
  return wide ? conv_launch<false, 128>(p, s) : conv_launch<false, 64>(p, s);
}

}  // namespace mode

// mode_conv_bn_act_fwd - inference: convolution (GEMM over the filter taps) + eval-mode BatchNorm + FiLM + residual + ReLU as ONE launch
extern "C" int mode_conv_bn_act_fwd(const ModeConvBnDesc* d, void* stream) {
  using namespace mode;
  if (!d || !d->x || !d->w || !d->y || d->M < 0 || d->Cin <= 0 || d->Cout <= 0 || d->taps <= 0) return MODE_ERR_BAD_ARG;
  if (d->Cin % 64 || d->Cout % 8 || d->ldx % 8 || d->ldw % 8 || d->ldy % 8 || (((uintptr_t)d->x | (uintptr_t)d->w | (uintptr_t)d->y) & 15)) return MODE_ERR_UNSUPPORTED;
  if (!d->idx && d->taps != 1) return MODE_ERR_BAD_ARG;
  if ((d->bn_mean != nullptr) != (d->bn_var != nullptr) || (d->pre_gamma != nullptr) != (d->pre_beta != nullptr) || (d->post_gamma != nullptr) != (d->post_beta != nullptr))
    return MODE_ERR_BAD_ARG;
  if (d->residual && (d->ldr % 4 || ((uintptr_t)d->residual & 7))) return MODE_ERR_UNSUPPORTED;
  if ((d->pre_gamma || d->post_gamma) && d->rows_per_sample <= 0) return MODE_ERR_BAD_ARG;
  if (((uintptr_t)d->bn_mean | (uintptr_t)d->bn_var | (uintptr_t)d->bn_weight | (uintptr_t)d->bn_bias | (uintptr_t)d->pre_gamma | (uintptr_t)d->pre_beta |
       (uintptr_t)d->post_gamma | (uintptr_t)d->post_beta) & 15) return MODE_ERR_UNSUPPORTED;
  if (d->M == 0) return MODE_OK;
  ConvGemmParams p;
  p.A = (const uint16_t*)d->x; p.lda = d->ldx; p.idx = d->idx; p.idx_tstride = d->idx_tap_stride;
  p.W = (const uint16_t*)d->w; p.ldw = d->ldw; p.C = (uint16_t*)d->y; p.ldc = d->ldy;
  p.M = d->M; p.N = d->Cout; p.tap_k = d->Cin; p.taps = d->taps; p.m_tiles = p.n_tiles = 0;
  p.bn_mean = d->bn_mean; p.bn_var = d->bn_var; p.bn_w = d->bn_weight; p.bn_b = d->bn_bias; p.bn_eps = d->bn_eps;
  p.res = (const uint16_t*)d->residual; p.ldr = d->ldr; p.relu = d->relu;
  p.pre_g = d->pre_gamma; p.pre_b = d->pre_beta; p.post_g = d->post_gamma; p.post_b = d->post_beta; p.rows_per_sample = d->rows_per_sample > 0 ? d->rows_per_sample : 1;
  if ((d->stat_sum != nullptr) != (d->stat_sq != nullptr) || (((uintptr_t)d->stat_sum | (uintptr_t)d->stat_sq) & 15)) return MODE_ERR_BAD_ARG;
  p.stat_sum = d->stat_sum; p.stat_sq = d->stat_sq;
  const long t128 = (long)((d->M + 127) / 128) * ((d->Cout + 127) / 128);
  const bool wide = d->Cout % 128 == 0 && t128 >= 384;
  return wide ? conv_launch<false, 128, true>(p, (hipStream_t)stream) : conv_launch<false, 64, true>(p, (hipStream_t)stream);
}
