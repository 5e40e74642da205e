// Skinny-M bf16 GEMM for gfx950: C[M,N] = epilogue(A[M,K] @ W[N,K]^T) when M is a handful of rows (B <= 2 environments = 14..28 tokens,
// the reference's own rollout is B = 1: mode_agent.py:630).  There the tiled kernel of gemm_bf16.hip launches 16..128 workgroups that each
// walk K serially with two 24-KiB tiles in flight — 0.4-0.7 TB/s of weight traffic.  This kernel is a weight STREAMER:
//   * one workgroup per 16 output columns (16 value + 16 gate rows of W1 for SwiGLU) and expert; its waves cut K into NW slices, so a
//     QKV projection is 192 workgroups x 4 waves and the expert up-projection 512 x E — enough independent 1-KiB loads in flight (U per
//     wave) to cover the HBM latency;
//   * W and A fragments go global -> VGPR straight in the MFMA operand layout (lane = row l&15, 8 consecutive k at (l>>4)*8): every weight
//     element is used once per 16 rows, so staging it in LDS buys nothing; A (<= 64 rows) is re-read from L2;
//   * v_mfma_f32_16x16x32_bf16 with the operands swapped like the tiled kernel (D[n][m]: a lane owns 4 consecutive columns of one row);
//   * the NW partial accumulators meet in LDS in a fixed order (deterministic), wave 0's lanes.. all waves share the epilogue work;
//   * rows are processed in blocks of MT*16 (MT <= 4): larger row counts loop over blocks and re-read the 16 x K weight slab from L2.
//   * KST > 0 (the chain's shapes: 256 reduction columns per wave): ALL of the wave's W and A fragments are requested before the first MFMA — one
//     memory round trip per kernel instead of one per pair of k-steps (the up-projection had four dependent ones: 13.4 us for 33.5 MB); the W
//     loads are non-temporal (every weight element is read by exactly one workgroup, once).
// Epilogues: NONE (+ split-K slabs), BIAS, BIAS_GELU, RESIDUAL, SWIGLU; grouped (expert_offsets) and gathered (a_rows) like the tiled kernel.
// Fused ln_2 of the small-batch chain (MODE_GEMM_SMALL_ROWS): RESIDUAL_NORM publishes one sum of squares per row and 16-column group
// ([M, N/16], this workgroup's columns), SWIGLU scales its accumulator rows by 1/norm from any number of partials per row
// (sum_row_partials_wave: the same function in the combine / head kernels).
#include "mode_common.h"

namespace mode {

__device__ __forceinline__ bf16x8 load_w_nt(const uint16_t* p) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return __builtin_bit_cast(bf16x8, v);
}

// Inverse RMSNorm factors of the row block [mb, mb + MT*16) into rsn[] (fused ln_2 consumer): wave w takes rows w, w + NW, ...  Two phases so
// that the caller can put each one's loads into a round trip it already pays: (1) the token ids of the wave's rows (beside the A-row gather
// indices), (2) the partial sums (beside the A fragments) and the reductions.  One uniform branch around each batch of loads - a per-row
// `a_rows ? a_rows[s] : s` compiles into a branch + load + s_waitcnt vmcnt(0) per row (eight serial round trips: 3 us of the B = 1
// up-projection).  Same arithmetic as sum_row_partials_wave (mode_common.h).
template <int RPW, int NW>
__device__ __forceinline__ void row_norm_tokens(const GemmParams& p, int mb, int row_end, int wave, int (&tok)[RPW]) {
  if (p.a_rows) {
#pragma unroll
    for (int q = 0; q < RPW; ++q) tok[q] = p.a_rows[min(mb + wave + q * NW, row_end - 1)];
  } else {
#pragma unroll
    for (int q = 0; q < RPW; ++q) tok[q] = min(mb + wave + q * NW, row_end - 1);
  }
}
// phase 2a: request the partial sums (one per lane and row when 16 < n <= 64; the ascending-order sums of sum_row_partials when n <= 16)
template <int RPW>
__device__ __forceinline__ void row_norm_load(const GemmParams& p, const int (&tok)[RPW], int lane, float (&sv)[RPW]) {
  if (p.ss_n > 16) {
    if (p.ss_n <= 64) {                                              // one partial per lane: all rows' loads in flight together
#pragma unroll
      for (int q = 0; q < RPW; ++q) sv[q] = p.ss_in[(long)tok[q] * p.ss_n + min(lane, p.ss_n - 1)];
    } else {
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        float s = 0.f;
        for (int j = lane; j < p.ss_n; j += 64) s += p.ss_in[(long)tok[q] * p.ss_n + j];
        sv[q] = s;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < RPW; ++q) sv[q] = sum_row_partials(p.ss_in + (long)tok[q] * p.ss_n, p.ss_n);
  }
}
// phase 2b: reduce and publish 1 / max(sqrt(sum) * K^-1/2, eps)
template <int RPW, int NW>
__device__ __forceinline__ void row_norm_reduce(const GemmParams& p, float (&sv)[RPW], int wave, int lane, float* rsn) {
  if (p.ss_n > 16) {
    if (p.ss_n <= 64) {
#pragma unroll
      for (int q = 0; q < RPW; ++q) sv[q] = 0.f + (lane < p.ss_n ? sv[q] : 0.f);
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) sv[q] = wave_sum(sv[q]);
  }
  const float rk = rsqrtf((float)p.K);
#pragma unroll
  for (int q = 0; q < RPW; ++q)
    if (lane == 0) rsn[wave + q * NW] = __frcp_rn(fmaxf(__fsqrt_rn(sv[q]) * rk, p.ss_eps));
}
template <int RPW, int NW>
__device__ __forceinline__ void row_norm_finish(const GemmParams& p, const int (&tok)[RPW], int wave, int lane, float* rsn) {
  float sv[RPW];
  row_norm_load<RPW>(p, tok, lane, sv);
  row_norm_reduce<RPW, NW>(p, sv, wave, lane, rsn);
}

template <int MT, int EPI, bool OUT_BF16, int NW, int KST>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_skinny_kernel(const GemmParams p) {
  constexpr int FNW = (EPI == MODE_EPI_SWIGLU) ? 2 : 1;            // W fragments per wave: value (+ gate)
  constexpr int U = (MT * FNW >= 4) ? 2 : 4;                         // k32 steps whose loads are in flight together (generic K)
  static_assert(MT <= NW && (MT * 16) % NW == 0, "one epilogue row fragment per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = reinterpret_cast<f32x4*>(smem);                      // [NW][MT][FNW][64]
  float* rsn = reinterpret_cast<float*>(smem + (size_t)NW * MT * FNW * 64 * 16);   // [MT*16] inverse row norms (fused ln_2 consumer)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int expert = blockIdx.y;
  const int n0 = blockIdx.x * 16;
  const int n = n0 + fq * 4;                                         // epilogue: a lane owns columns n .. n+3 of one row

  int row0 = 0, row_end = p.M;
  if (p.offsets) { row0 = p.offsets[expert]; row_end = p.offsets[expert + 1]; }
  if (row_end <= row0) return;
  const uint16_t* W = p.W + (long)expert * p.w_estride;
  const float* bias = p.bias ? p.bias + (long)expert * p.bias_estride : nullptr;

  // this wave's K range: slice blockIdx.z of split_k, cut into NW wave slices (multiples of 32)
  const int kspl = p.K / p.split_k, kw = kspl / NW;
  const int kbeg = blockIdx.z * kspl + wave * kw;
  const uint16_t* wrow[FNW];
  wrow[0] = W + (long)min(n0 + fr, p.N - 1) * p.ldw + kbeg + fq * 8;
  if constexpr (FNW == 2) wrow[1] = W + ((long)min(n0 + fr, p.N - 1) + p.N) * p.ldw + kbeg + fq * 8;
  char* Cout = reinterpret_cast<char*>(p.C) + (long)blockIdx.z * p.split_stride * (OUT_BF16 ? 2 : 4);

  for (int mb = row0; mb < row_end; mb += MT * 16) {
    f32x4 acc[MT][FNW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < FNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // epilogue operands of this wave's row fragment (wave i < MT finishes fragment i): requested ahead of the MFMAs, not behind the reduction
    const int ml = mb + wave * 16 + fr;
    const bool ok = wave < MT && ml < row_end && n < p.N;
    [[maybe_unused]] float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = make_float4(0.f, 0.f, 0.f, 0.f);

    if constexpr (KST > 0) {
      // ---- the wave's whole operand set in flight at once: W first (the HBM stream, non-temporal), then the L2-resident A rows
      bf16x8 wf[KST][FNW], af[KST][MT];
#pragma unroll
      for (int u = 0; u < KST; ++u)
#pragma unroll
        for (int j = 0; j < FNW; ++j) wf[u][j] = load_w_nt(wrow[j] + 32 * u);
      // round trip 1: gather indices of the A rows + token ids of the rows whose norms this wave computes (one uniform branch, not one per row)
      int arow_i[MT];
      [[maybe_unused]] int tok[MT * 16 / NW];
      if (p.a_rows) {
#pragma unroll
        for (int i = 0; i < MT; ++i) arow_i[i] = p.a_rows[min(mb + i * 16 + fr, row_end - 1)];   // rows past the segment re-read a valid row (never stored)
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) arow_i[i] = min(mb + i * 16 + fr, row_end - 1);
      }
      if constexpr (EPI == MODE_EPI_SWIGLU) {
        if (p.ss_in) row_norm_tokens<MT * 16 / NW, NW>(p, mb, row_end, wave, tok);
      }
      // epilogue operands: clamped addresses, unconditional loads (a branch around them would end in an s_waitcnt vmcnt(0) on everything above)
      {
        const int nc = min(n, p.N - 4);
        const long mlc = min(ml, row_end - 1);
        if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) e0 = *reinterpret_cast<const float4*>(bias + nc);
        if constexpr (EPI == MODE_EPI_SWIGLU) { e0 = *reinterpret_cast<const float4*>(bias + nc); e1 = *reinterpret_cast<const float4*>(bias + p.N + nc); }
        if constexpr (EPI == MODE_EPI_RESIDUAL || EPI == MODE_EPI_RESIDUAL_NORM) e0 = *reinterpret_cast<const float4*>(p.resid + mlc * p.ldr + nc);
        if constexpr (EPI == MODE_EPI_RESIDUAL_NORM) e1 = *reinterpret_cast<const float4*>(p.gain + nc);
      }
      // round trip 2: the A fragments (L2) and the rows' partial sums of squares
#pragma unroll
      for (int u = 0; u < KST; ++u)
#pragma unroll
        for (int i = 0; i < MT; ++i) af[u][i] = *reinterpret_cast<const bf16x8*>(p.A + (long)arow_i[i] * p.lda + kbeg + fq * 8 + 32 * u);
      if constexpr (EPI == MODE_EPI_SWIGLU) {
        if (p.ss_in) row_norm_finish<MT * 16 / NW, NW>(p, tok, wave, lane, rsn);
      }
#pragma unroll
      for (int u = 0; u < KST; ++u)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < FNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][j], af[u][i], acc[i][j], 0, 0, 0);
    } else {
      const uint16_t* arow[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int s = min(mb + i * 16 + fr, row_end - 1);
        arow[i] = p.A + (p.a_rows ? (long)p.a_rows[s] : (long)s) * p.lda + kbeg + fq * 8;
      }
      if constexpr (EPI == MODE_EPI_SWIGLU) {
        if (p.ss_in) {
          int tok[MT * 16 / NW];
          row_norm_tokens<MT * 16 / NW, NW>(p, mb, row_end, wave, tok);
          row_norm_finish<MT * 16 / NW, NW>(p, tok, wave, lane, rsn);
        }
      }
      for (int k = 0; k < kw; k += 32 * U) {
        bf16x8 wf[U][FNW], af[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ko = min(k + 32 * u, kw - 32);                   // tail steps re-read the last block and are skipped below
#pragma unroll
          for (int j = 0; j < FNW; ++j) wf[u][j] = *reinterpret_cast<const bf16x8*>(wrow[j] + ko);
#pragma unroll
          for (int i = 0; i < MT; ++i) af[u][i] = *reinterpret_cast<const bf16x8*>(arow[i] + ko);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (k + 32 * u < kw) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < FNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][j], af[u][i], acc[i][j], 0, 0, 0);
          }
        }
      }
      if (ok) {
        if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) e0 = *reinterpret_cast<const float4*>(bias + n);
        if constexpr (EPI == MODE_EPI_SWIGLU) { e0 = *reinterpret_cast<const float4*>(bias + n); e1 = *reinterpret_cast<const float4*>(bias + p.N + n); }
        if constexpr (EPI == MODE_EPI_RESIDUAL || EPI == MODE_EPI_RESIDUAL_NORM) e0 = *reinterpret_cast<const float4*>(p.resid + (long)ml * p.ldr + n);
        if constexpr (EPI == MODE_EPI_RESIDUAL_NORM) e1 = *reinterpret_cast<const float4*>(p.gain + n);
      }
    }

    // ---- K-slices of the NW waves meet in LDS, summed in wave order
    if (mb != row0) __syncthreads();                                 // previous block's readers are done
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < FNW; ++j) red[((wave * MT + i) * FNW + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    if (wave < MT) {                                                 // wave i finishes row fragment i
      f32x4 v[FNW];
#pragma unroll
      for (int j = 0; j < FNW; ++j) {
        v[j] = red[(wave * FNW + j) * 64 + lane];
        for (int w = 1; w < NW; ++w) v[j] += red[((w * MT + wave) * FNW + j) * 64 + lane];
      }
      if constexpr (EPI == MODE_EPI_RESIDUAL_NORM) {
        // x = acc + resid -> C (fp32); bf16(x * gain) -> C2; sum of x^2 over this workgroup's 16 columns -> ss_out[row][n0/16]: the four lanes
        // fq = 0..3 of a row hold 4 columns each (xor-shuffles 16, 32: fixed order); every lane takes part, out-of-range ones add zeros
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
          o[0] = v[0][0] + e0.x; o[1] = v[0][1] + e0.y; o[2] = v[0][2] + e0.z; o[3] = v[0][3] + e0.w;
          *reinterpret_cast<float4*>(Cout + ((long)ml * p.ldc + n) * 4) = make_float4(o[0], o[1], o[2], o[3]);
          *reinterpret_cast<uint2*>(p.C2 + (long)ml * p.ldc2 + n) = make_uint2(pack_bf16x2(o[0] * e1.x, o[1] * e1.y), pack_bf16x2(o[2] * e1.z, o[3] * e1.w));
        }
        float ss = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
        ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
        if (ok && fq == 0) p.ss_out[(long)ml * (p.N / 16) + blockIdx.x] = ss;
      } else if (ok) {
        f32x4 o = v[0];
        if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) {
          o[0] += e0.x; o[1] += e0.y; o[2] += e0.z; o[3] += e0.w;
          if constexpr (EPI == MODE_EPI_BIAS_GELU) { o[0] = gelu_erf_f(o[0]); o[1] = gelu_erf_f(o[1]); o[2] = gelu_erf_f(o[2]); o[3] = gelu_erf_f(o[3]); }
        } else if constexpr (EPI == MODE_EPI_SWIGLU) {
          if (p.ss_in) {
            const float rs = rsn[wave * 16 + fr];
            o[0] = swiglu_f(v[0][0], v[1][0], rs, e0.x, e1.x); o[1] = swiglu_f(v[0][1], v[1][1], rs, e0.y, e1.y);
            o[2] = swiglu_f(v[0][2], v[1][2], rs, e0.z, e1.z); o[3] = swiglu_f(v[0][3], v[1][3], rs, e0.w, e1.w);
          } else {
            o[0] = (v[0][0] + e0.x) * silu_f(v[1][0] + e1.x); o[1] = (v[0][1] + e0.y) * silu_f(v[1][1] + e1.y);
            o[2] = (v[0][2] + e0.z) * silu_f(v[1][2] + e1.z); o[3] = (v[0][3] + e0.w) * silu_f(v[1][3] + e1.w);
          }
        } else if constexpr (EPI == MODE_EPI_RESIDUAL) {
          o[0] += e0.x; o[1] += e0.y; o[2] += e0.z; o[3] += e0.w;
        }
        if constexpr (OUT_BF16) *reinterpret_cast<uint2*>(Cout + ((long)ml * p.ldc + n) * 2) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        else *reinterpret_cast<float4*>(Cout + ((long)ml * p.ldc + n) * 4) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// ---- The chain's shape: a K-slice of exactly 1024 columns per workgroup (QKV / c_proj / up-projection: K = 1024; down-projection: 4096 in four
// slices), at most 32 rows per segment.  What bounded the kernel above there was not HBM but the number of L1 requests: a lane's 16 bytes of an
// MFMA fragment lie 2-8 KB from its neighbours', so every load instruction is 64 separate cache-line requests, and the A rows - re-read by every
// workgroup - were two thirds of them (down-projection: 5.6 k requests per workgroup, 2.9 TB/s of weights).  Here the workgroup's A block goes
// through LDS: copied by LDS-DMA (a wave instruction = 1 KB of one row = 8 cache lines, no registers) into a row-major image with 16 bytes
// of padding per row (fragment reads conflict-free), read back as MFMA fragments.  Order of issue: index loads -> A DMA + W loads (nothing waits
// between them) -> one wait -> fragments -> MFMAs.
template <int MT, int EPI, bool OUT_BF16>
__global__ __launch_bounds__(256) void gemm_bf16_stream_kernel(const GemmParams p) {
  constexpr int NW = 4, KS = 8, ROWS = MT * 16, AROW = 2048 + 16, NIT = ROWS / 2;   // NIT 16-byte chunks of A per thread
  constexpr int FNW = (EPI == MODE_EPI_SWIGLU) ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* aimg = smem;                                                // [ROWS][AROW]; the reduction buffer re-uses it once the fragments are read
  f32x4* red = reinterpret_cast<f32x4*>(smem);                      // [NW][MT][FNW][64]
  float* rsn = reinterpret_cast<float*>(smem + (size_t)ROWS * AROW);   // [ROWS] inverse row norms (fused ln_2 consumer)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int expert = blockIdx.y;
  const int n0 = blockIdx.x * 16;
  const int n = n0 + fq * 4;

  int row0 = 0, row_end = p.M;
  if (p.offsets) { row0 = p.offsets[expert]; row_end = p.offsets[expert + 1]; }
  if (row_end <= row0) return;
  const uint16_t* W = p.W + (long)expert * p.w_estride;
  const float* bias = p.bias ? p.bias + (long)expert * p.bias_estride : nullptr;
  const int kslice = blockIdx.z * 1024;
  const int kbeg = kslice + wave * 256;
  const uint16_t* wrow[FNW];
  wrow[0] = W + (long)min(n0 + fr, p.N - 1) * p.ldw + kbeg + fq * 8;
  if constexpr (FNW == 2) wrow[1] = W + ((long)min(n0 + fr, p.N - 1) + p.N) * p.ldw + kbeg + fq * 8;
  char* Cout = reinterpret_cast<char*>(p.C) + (long)blockIdx.z * p.split_stride * (OUT_BF16 ? 2 : 4);

  for (int mb = row0; mb < row_end; mb += ROWS) {
    if (mb != row0) __syncthreads();                                 // previous block's epilogue readers are done with the shared buffer
    const int ml = mb + wave * 16 + fr;
    const bool ok = wave < MT && ml < row_end && n < p.N;
    // ---- round trip 1: source rows of this wave's A half-rows, token ids of the rows whose norms it computes (uniform branches around the batches)
    int src[NIT];                                                    // half-row j*4 + wave -> row (j*4 + wave) >> 1
    [[maybe_unused]] int tok[ROWS / NW];
    if (p.a_rows && p.identity_rows) {                               // promised a_rows[row0 + i] == i: no index round trip (MODE_GEMM_IDENTITY_ROWS)
#pragma unroll
      for (int j = 0; j < NIT; ++j) src[j] = min(mb + ((j * 4 + wave) >> 1), row_end - 1) - row0;
      if constexpr (EPI == MODE_EPI_SWIGLU) {
#pragma unroll
        for (int q = 0; q < ROWS / NW; ++q) tok[q] = min(mb + wave + q * NW, row_end - 1) - row0;
      }
    } else if (p.a_rows) {                                           // (both batches inside one branch: one wait for all of them)
#pragma unroll
      for (int j = 0; j < NIT; ++j) src[j] = p.a_rows[min(mb + ((j * 4 + wave) >> 1), row_end - 1)];   // rows past the segment re-read a valid row (never stored)
      if constexpr (EPI == MODE_EPI_SWIGLU) {
#pragma unroll
        for (int q = 0; q < ROWS / NW; ++q) tok[q] = p.a_rows[min(mb + wave + q * NW, row_end - 1)];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NIT; ++j) src[j] = min(mb + ((j * 4 + wave) >> 1), row_end - 1);
      if constexpr (EPI == MODE_EPI_SWIGLU) {
#pragma unroll
        for (int q = 0; q < ROWS / NW; ++q) tok[q] = min(mb + wave + q * NW, row_end - 1);
      }
    }
    // ---- round trip 2, all of it requested before anything is consumed: the A block by LDS-DMA (a wave instruction copies 1 KB of one row,
    // coalesced, straight into the padded row-major image: no registers, nothing to wait for before the weight loads go out), the epilogue
    // operands, the partial sums of squares, and the weight stream (non-temporal: read once, by this workgroup only)
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int hr = j * 4 + wave;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.A + (long)src[j] * p.lda + kslice + (hr & 1) * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(aimg + (hr >> 1) * AROW + (hr & 1) * 1024), 16, 0, 0);
    }
    [[maybe_unused]] float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      const int nc = min(n, p.N - 4);
      const long mlc = min(ml, row_end - 1);
      if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) e0 = *reinterpret_cast<const float4*>(bias + nc);
      if constexpr (EPI == MODE_EPI_SWIGLU) { e0 = *reinterpret_cast<const float4*>(bias + nc); e1 = *reinterpret_cast<const float4*>(bias + p.N + nc); }
      if constexpr (EPI == MODE_EPI_RESIDUAL || EPI == MODE_EPI_RESIDUAL_NORM) e0 = *reinterpret_cast<const float4*>(p.resid + mlc * p.ldr + nc);
      if constexpr (EPI == MODE_EPI_RESIDUAL_NORM) e1 = *reinterpret_cast<const float4*>(p.gain + nc);
    }
    [[maybe_unused]] float sv[ROWS / NW];
    if constexpr (EPI == MODE_EPI_SWIGLU) {
      if (p.ss_in) row_norm_load<ROWS / NW>(p, tok, lane, sv);
    }
    bf16x8 wf[KS][FNW];
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
      for (int j = 0; j < FNW; ++j) wf[u][j] = load_w_nt(wrow[j] + 32 * u);
    if constexpr (EPI == MODE_EPI_SWIGLU) {
      if (p.ss_in) row_norm_reduce<ROWS / NW, NW>(p, sv, wave, lane, rsn);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the DMA'd A block has landed (the compiler does not track it)
    __syncthreads();
    bf16x8 af[KS][MT];
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i) af[u][i] = *reinterpret_cast<const bf16x8*>(aimg + (i * 16 + fr) * AROW + (wave * 256 + 32 * u + fq * 8) * 2);
    f32x4 acc[MT][FNW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < FNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < FNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][j], af[u][i], acc[i][j], 0, 0, 0);

    // ---- K-slices of the four waves meet in LDS (over the A image: every wave has its fragments), summed in wave order
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < FNW; ++j) red[((wave * MT + i) * FNW + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    if (wave < MT) {                                                 // wave i finishes row fragment i
      f32x4 v[FNW];
#pragma unroll
      for (int j = 0; j < FNW; ++j) {
        v[j] = red[(wave * FNW + j) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) v[j] += red[((w * MT + wave) * FNW + j) * 64 + lane];
      }
      if constexpr (EPI == MODE_EPI_RESIDUAL_NORM) {
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
          o[0] = v[0][0] + e0.x; o[1] = v[0][1] + e0.y; o[2] = v[0][2] + e0.z; o[3] = v[0][3] + e0.w;
          *reinterpret_cast<float4*>(Cout + ((long)ml * p.ldc + n) * 4) = make_float4(o[0], o[1], o[2], o[3]);
          *reinterpret_cast<uint2*>(p.C2 + (long)ml * p.ldc2 + n) = make_uint2(pack_bf16x2(o[0] * e1.x, o[1] * e1.y), pack_bf16x2(o[2] * e1.z, o[3] * e1.w));
        }
        float ss = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
        ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
        if (ok && fq == 0) p.ss_out[(long)ml * (p.N / 16) + blockIdx.x] = ss;
      } else if (ok) {
        f32x4 o = v[0];
        if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) {
          o[0] += e0.x; o[1] += e0.y; o[2] += e0.z; o[3] += e0.w;
          if constexpr (EPI == MODE_EPI_BIAS_GELU) { o[0] = gelu_erf_f(o[0]); o[1] = gelu_erf_f(o[1]); o[2] = gelu_erf_f(o[2]); o[3] = gelu_erf_f(o[3]); }
        } else if constexpr (EPI == MODE_EPI_SWIGLU) {
          if (p.ss_in) {
            const float rs = rsn[wave * 16 + fr];
            o[0] = swiglu_f(v[0][0], v[1][0], rs, e0.x, e1.x); o[1] = swiglu_f(v[0][1], v[1][1], rs, e0.y, e1.y);
            o[2] = swiglu_f(v[0][2], v[1][2], rs, e0.z, e1.z); o[3] = swiglu_f(v[0][3], v[1][3], rs, e0.w, e1.w);
          } else {
            o[0] = (v[0][0] + e0.x) * silu_f(v[1][0] + e1.x); o[1] = (v[0][1] + e0.y) * silu_f(v[1][1] + e1.y);
            o[2] = (v[0][2] + e0.z) * silu_f(v[1][2] + e1.z); o[3] = (v[0][3] + e0.w) * silu_f(v[1][3] + e1.w);
          }
        } else if constexpr (EPI == MODE_EPI_RESIDUAL) {
          o[0] += e0.x; o[1] += e0.y; o[2] += e0.z; o[3] += e0.w;
        }
        if constexpr (OUT_BF16) *reinterpret_cast<uint2*>(Cout + ((long)ml * p.ldc + n) * 2) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        else *reinterpret_cast<float4*>(Cout + ((long)ml * p.ldc + n) * 4) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// ---- Medium M (33..128 rows: 3..9 environments), K = 1024, ungrouped: the QKV projection of small rollout batches.  The ring kernels
// spend ~800 cycles per 64-column K-step there whatever the ring depth (s_barrier turnaround + exposed LDS read latency around 128 cycles of
// MFMA: [448 x 1024] x [3072 x 1024] takes 11.6 us, a [112 x 1024] one 9.7 us).  This kernel has NO K loop over memory: a workgroup owns
// 32 rows x 64 columns; each of its four waves keeps the whole K = 1024 of its 16 weight rows in registers (32 fragments, requested in one
// burst) and the workgroup's A block sits in LDS as in gemm_bf16_stream_kernel (LDS-DMA, padded rows); then 32 x 2 MFMAs per wave straight
// through - the same k-ordered fp32 chain as every other forward kernel, so the results are bit-identical to the ring kernels (the batch-slice
// test compares B = 8 with the same samples inside B = 128).  W is re-read by the M/32 row blocks through L2 (default cache policy), which is
// why the kernel stops paying from ~200 rows on ("gemm_mid_rows" option, default 128).
template <int EPI, bool OUT_BF16>
__global__ __launch_bounds__(256) void gemm_bf16_mid_kernel(const GemmParams p) {
  constexpr int ROWS = 32, AROW = 2048 + 16, KS = 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* aimg = smem;                                                // [32][AROW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * 64 + wave * 16;
  const int mb = blockIdx.y * ROWS;
  const int n = n0 + fq * 4;
  // the A block first (LDS-DMA, nothing waits for it), then the wave's 32 weight fragments
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int hr = j * 4 + wave;                                     // half-row: row hr >> 1, 1-KiB half hr & 1
    const int srow = min(mb + (hr >> 1), p.M - 1);                  // rows past M re-read a valid row (never stored)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.A + (long)srow * p.lda + (hr & 1) * 512 + lane * 8),
                                     (__attribute__((address_space(3))) void*)(aimg + (hr >> 1) * AROW + (hr & 1) * 1024), 16, 0, 0);
  }
  // Weight fragments: loaded with four ADJACENT lanes on one row's 64 bytes of a k32 step (lane l: row l >> 2, 16-byte piece l & 3 - 16 requests
  // of 64 B per instruction instead of 64 of 16 B: the per-CU L1 request rate, one per clock, was what bounded the direct fragment loads:
  // 18.5 us at M = 448), then moved into MFMA layout (lane fr + 16*fq <- row fr, piece fq = loading lane 4*fr + fq) by ds_bpermute.
  const uint16_t* wrow = p.W + (long)min(n0 + (lane >> 2), p.N - 1) * p.ldw + (lane & 3) * 8;
  typedef __attribute__((ext_vector_type(4))) int i32x4_t;
  i32x4_t wraw[KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) wraw[u] = *reinterpret_cast<const i32x4_t*>(wrow + 32 * u);
  const int perm_addr = (4 * fr + fq) * 4;
  [[maybe_unused]] float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f);
  [[maybe_unused]] float4 rr[2];
  if constexpr (EPI == MODE_EPI_BIAS) e0 = *reinterpret_cast<const float4*>(p.bias + min(n, p.N - 4));
  if constexpr (EPI == MODE_EPI_RESIDUAL_NORM) {                     // residual rows + gain: requested with the weights, ahead of the MFMAs
    e0 = *reinterpret_cast<const float4*>(p.gain + min(n, p.N - 4));
#pragma unroll
    for (int i = 0; i < 2; ++i) rr[i] = *reinterpret_cast<const float4*>(p.resid + (long)min(mb + i * 16 + fr, p.M - 1) * p.ldr + min(n, p.N - 4));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the DMA'd A block has landed (the compiler does not track it)
  __syncthreads();
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const char* a0 = aimg + fr * AROW + fq * 16;
  bf16x8 wf[KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    i32x4_t t;
    t[0] = __builtin_amdgcn_ds_bpermute(perm_addr, wraw[u][0]); t[1] = __builtin_amdgcn_ds_bpermute(perm_addr, wraw[u][1]);
    t[2] = __builtin_amdgcn_ds_bpermute(perm_addr, wraw[u][2]); t[3] = __builtin_amdgcn_ds_bpermute(perm_addr, wraw[u][3]);
    wf[u] = __builtin_bit_cast(bf16x8, t);
  }
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    const bf16x8 af0 = *reinterpret_cast<const bf16x8*>(a0 + u * 64);
    const bf16x8 af1 = *reinterpret_cast<const bf16x8*>(a0 + 16 * AROW + u * 64);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af1, acc[1], 0, 0, 0);
  }
  if constexpr (EPI == MODE_EPI_RESIDUAL_NORM) {
    // x = acc + resid -> C (fp32); bf16(x * gain) -> C2; the row's sum of squares over this workgroup's 64 columns -> ss_out[row][blockIdx.x]: the 16 four-column
    // chunks of a row sit in (wave, fq) = (j / 4, j % 4); they meet in LDS and are added in the tree order of the ring kernel's epilogue (ss16_tree), so
    // both kernels publish bit-identical partial sums
    __syncthreads();                                                 // everyone is done with the A image
    float* ssb = reinterpret_cast<float*>(smem);                      // [32 rows][16 chunks]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ml = mb + i * 16 + fr;
      const bool ok = ml < p.M && n < p.N;
      const float x0 = acc[i][0] + rr[i].x, x1 = acc[i][1] + rr[i].y, x2 = acc[i][2] + rr[i].z, x3 = acc[i][3] + rr[i].w;
      if (ok) {
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(p.C) + ((long)ml * p.ldc + n) * 4) = make_float4(x0, x1, x2, x3);
        *reinterpret_cast<uint2*>(p.C2 + (long)ml * p.ldc2 + n) = make_uint2(pack_bf16x2(x0 * e0.x, x1 * e0.y), pack_bf16x2(x2 * e0.z, x3 * e0.w));
      }
      ssb[(i * 16 + fr) * 16 + wave * 4 + fq] = ok ? ss4_f(x0, x1, x2, x3) : 0.f;
    }
    __syncthreads();
    if (tid < 32 && mb + tid < p.M) p.ss_out[(long)(mb + tid) * (p.N >> 6) + blockIdx.x] = ss16_tree(ssb + tid * 16);
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ml = mb + i * 16 + fr;
    if (ml >= p.M || n >= p.N) continue;
    f32x4 o = acc[i];
    if constexpr (EPI == MODE_EPI_BIAS) { o[0] += e0.x; o[1] += e0.y; o[2] += e0.z; o[3] += e0.w; }
    if constexpr (OUT_BF16) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.C) + ((long)ml * p.ldc + n) * 2) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    else *reinterpret_cast<float4*>(reinterpret_cast<char*>(p.C) + ((long)ml * p.ldc + n) * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Medium-M launcher: ungrouped, K == 1024, no split, NONE / BIAS epilogues, N % 64 == 0.  Returns MODE_ERR_UNSUPPORTED otherwise.
int gemm_bf16_mid_launch(const ModeGemmDesc* d, const GemmParams& p, hipStream_t s) {
  if (d->expert_offsets || d->a_rows || p.koffs || p.split_k != 1 || d->K != 1024 || d->N % 64 || p.ss_in) return MODE_ERR_UNSUPPORTED;
  if (d->epilogue != MODE_EPI_NONE && d->epilogue != MODE_EPI_BIAS && d->epilogue != MODE_EPI_RESIDUAL_NORM) return MODE_ERR_UNSUPPORTED;
  if (d->epilogue == MODE_EPI_RESIDUAL_NORM && (d->out_dtype != MODE_F32 || !p.C2 || !p.gain || !p.ss_out || !p.resid)) return MODE_ERR_UNSUPPORTED;
  constexpr size_t LDS = (size_t)32 * (2048 + 16);
  const dim3 grid(d->N / 64, (d->M + 31) / 32);
  const bool ob = d->out_dtype == MODE_BF16;
#define MODE_MID(E, OB)                                                                                                                     \
  do {                                                                                                                                      \
    auto kern = gemm_bf16_mid_kernel<E, OB>;                                                                                                \
    static LdsLimitOnce lds_once;                                                                                                           \
    {                                                                                                                                       \
      const int rc = lds_once.ensure(reinterpret_cast<const void*>(kern), (int)LDS);                                                       \
      if (rc != MODE_OK) return rc;                                                                                                         \
    }                                                                                                                                       \
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, p);                                                                                   \
  } while (0)
  if (d->epilogue == MODE_EPI_RESIDUAL_NORM) MODE_MID(MODE_EPI_RESIDUAL_NORM, false);
  else if (d->epilogue == MODE_EPI_BIAS) { if (ob) MODE_MID(MODE_EPI_BIAS, true); else MODE_MID(MODE_EPI_BIAS, false); }
  else { if (ob) MODE_MID(MODE_EPI_NONE, true); else MODE_MID(MODE_EPI_NONE, false); }
#undef MODE_MID
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

template <int MT, int EPI, bool OUT_BF16>
static int launch_skinny(const GemmParams& p, int groups, hipStream_t s) {
  // 8 K-slices per workgroup when K is long and the grid is small (down-projection: K = 4096 / split), else 4
  const int kspl = p.K / p.split_k;
  const dim3 grid((p.N + 15) / 16, groups, p.split_k);
  constexpr int FNW = (EPI == MODE_EPI_SWIGLU) ? 2 : 1;
  constexpr size_t RS = MT * 16 * 4;                                 // inverse row norms behind the reduction buffer
  if constexpr (MT <= 2) {
    if (kspl == 1024) {                                              // the chain's shapes: A block through LDS
      constexpr size_t LDS = (size_t)MT * 16 * (2048 + 16) + RS;
      auto kern = gemm_bf16_stream_kernel<MT, EPI, OUT_BF16>;
      static LdsLimitOnce lds_once;
      if (LDS > 64 * 1024) {
        const int rc = lds_once.ensure(reinterpret_cast<const void*>(kern), (int)LDS);
        if (rc != MODE_OK) return rc;
      }
      hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, p);
      MODE_LAUNCH_CHECK();
      return MODE_OK;
    }
  }
  if (kspl % 256 == 0 && kspl >= 2048) {
    hipLaunchKernelGGL((gemm_bf16_skinny_kernel<MT, EPI, OUT_BF16, 8, 0>), grid, dim3(512), (size_t)8 * MT * FNW * 64 * 16 + RS, s, p);
  } else if (kspl == 1024) {                                         // every load of the wave in flight at once
    hipLaunchKernelGGL((gemm_bf16_skinny_kernel<MT, EPI, OUT_BF16, 4, 8>), grid, dim3(256), (size_t)4 * MT * FNW * 64 * 16 + RS, s, p);
  } else {
    hipLaunchKernelGGL((gemm_bf16_skinny_kernel<MT, EPI, OUT_BF16, 4, 0>), grid, dim3(256), (size_t)4 * MT * FNW * 64 * 16 + RS, s, p);
  }
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

template <int EPI, bool OUT_BF16>
static int launch_skinny_mt(const GemmParams& p, int groups, int max_rows, hipStream_t s) {
  if (max_rows <= 16) return launch_skinny<1, EPI, OUT_BF16>(p, groups, s);
  if (max_rows <= 32) return launch_skinny<2, EPI, OUT_BF16>(p, groups, s);
  return launch_skinny<4, EPI, OUT_BF16>(p, groups, s);
}

// Called by gemm_bf16_launch for M <= gemm_skinny_rows, or when the caller vouches (MODE_GEMM_SMALL_ROWS) that no GROUP has more rows than that.
// Returns MODE_ERR_UNSUPPORTED for what only the tiled kernel does (K-groups, ragged K, the 64-column fused-ln_2 partials), in which case the
// caller falls through to the tiled kernel.
int gemm_bf16_skinny_launch(const ModeGemmDesc* d, const GemmParams& p, hipStream_t s) {
  const bool small = (d->flags & MODE_GEMM_SMALL_ROWS) != 0;
  if (p.koffs) return MODE_ERR_UNSUPPORTED;
  if ((p.ss_in || d->epilogue == MODE_EPI_RESIDUAL_NORM) && !small) return MODE_ERR_UNSUPPORTED;   // 16-column partials only on request
  if (d->epilogue == MODE_EPI_RESIDUAL_NORM && d->N % 16 != 0) return MODE_ERR_UNSUPPORTED;
  if (d->K % (128 * p.split_k) != 0 || d->N % 4 != 0) return MODE_ERR_UNSUPPORTED;   // every wave slice a multiple of 32
  const int groups = d->expert_offsets ? d->num_experts : 1;
  const int max_rows = small && d->expert_offsets ? (d->M < g_gemm_skinny_rows ? d->M : g_gemm_skinny_rows) : d->M;
  const bool ob = d->out_dtype == MODE_BF16;
#define MODE_SK(E) \
  case E: return ob ? launch_skinny_mt<E, true>(p, groups, max_rows, s) : launch_skinny_mt<E, false>(p, groups, max_rows, s);
  switch (d->epilogue) {
    MODE_SK(MODE_EPI_NONE)
    MODE_SK(MODE_EPI_BIAS)
    MODE_SK(MODE_EPI_BIAS_GELU)
    MODE_SK(MODE_EPI_SWIGLU)
    case MODE_EPI_RESIDUAL: return ob ? MODE_ERR_BAD_ARG : launch_skinny_mt<MODE_EPI_RESIDUAL, false>(p, groups, max_rows, s);
    case MODE_EPI_RESIDUAL_NORM: return ob ? MODE_ERR_BAD_ARG : launch_skinny_mt<MODE_EPI_RESIDUAL_NORM, false>(p, groups, max_rows, s);
    default: return MODE_ERR_BAD_ARG;
  }
#undef MODE_SK
}

}  // namespace mode
