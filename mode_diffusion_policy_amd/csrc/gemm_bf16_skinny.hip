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
// Epilogues: NONE (+ split-K slabs), BIAS, BIAS_GELU, RESIDUAL, SWIGLU; grouped (expert_offsets) and gathered (a_rows) like the tiled kernel.
#include "mode_common.h"

namespace mode {

template <int MT, int EPI, bool OUT_BF16, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_skinny_kernel(const GemmParams p) {
  constexpr int FNW = (EPI == MODE_EPI_SWIGLU) ? 2 : 1;            // W fragments per wave: value (+ gate)
  constexpr int U = (MT * FNW >= 4) ? 2 : 4;                         // k32 steps whose loads are in flight together
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = reinterpret_cast<f32x4*>(smem);                      // [NW][MT][FNW][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int expert = blockIdx.y;
  const int n0 = blockIdx.x * 16;

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
    const uint16_t* arow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int s = min(mb + i * 16 + fr, row_end - 1);             // rows past the segment re-read a valid row (never stored)
      arow[i] = p.A + (p.a_rows ? (long)p.a_rows[s] : (long)s) * p.lda + kbeg + fq * 8;
    }
    f32x4 acc[MT][FNW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < FNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k = 0; k < kw; k += 32 * U) {
      bf16x8 wf[U][FNW], af[U][MT];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ko = min(k + 32 * u, kw - 32);                     // tail steps re-read the last block and are skipped below
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

    // ---- K-slices of the NW waves meet in LDS, summed in wave order
    if (mb != row0) __syncthreads();                                 // previous block's readers are done
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < FNW; ++j) red[((wave * MT + i) * FNW + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    for (int i = wave; i < MT; i += NW) {                            // row fragments are shared out over the waves
      f32x4 v[FNW];
#pragma unroll
      for (int j = 0; j < FNW; ++j) {
        v[j] = red[(i * FNW + j) * 64 + lane];
        for (int w = 1; w < NW; ++w) v[j] += red[((w * MT + i) * FNW + j) * 64 + lane];
      }
      const int ml = mb + i * 16 + fr;                               // sorted row; lane owns columns n .. n+3 of it
      const int n = n0 + fq * 4;
      if (ml >= row_end || n >= p.N) continue;
      f32x4 o = v[0];
      if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w;
        if constexpr (EPI == MODE_EPI_BIAS_GELU) { o[0] = gelu_erf_f(o[0]); o[1] = gelu_erf_f(o[1]); o[2] = gelu_erf_f(o[2]); o[3] = gelu_erf_f(o[3]); }
      } else if constexpr (EPI == MODE_EPI_SWIGLU) {
        const float4 bp = *reinterpret_cast<const float4*>(bias + n), bg = *reinterpret_cast<const float4*>(bias + p.N + n);
        o[0] = (v[0][0] + bp.x) * silu_f(v[1][0] + bg.x); o[1] = (v[0][1] + bp.y) * silu_f(v[1][1] + bg.y);
        o[2] = (v[0][2] + bp.z) * silu_f(v[1][2] + bg.z); o[3] = (v[0][3] + bp.w) * silu_f(v[1][3] + bg.w);
      } else if constexpr (EPI == MODE_EPI_RESIDUAL) {
        const float4 r = *reinterpret_cast<const float4*>(p.resid + (long)ml * p.ldr + n);
        o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
      }
      if constexpr (OUT_BF16) *reinterpret_cast<uint2*>(Cout + ((long)ml * p.ldc + n) * 2) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
      else *reinterpret_cast<float4*>(Cout + ((long)ml * p.ldc + n) * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

template <int MT, int EPI, bool OUT_BF16>
static int launch_skinny(const GemmParams& p, int groups, hipStream_t s) {
  // 8 K-slices per workgroup when K is long and the grid is small (down-projection: K = 4096 / split), else 4
  const int kspl = p.K / p.split_k;
  const dim3 grid((p.N + 15) / 16, groups, p.split_k);
  constexpr int FNW = (EPI == MODE_EPI_SWIGLU) ? 2 : 1;
  if (kspl % 256 == 0 && kspl >= 2048) {
    hipLaunchKernelGGL((gemm_bf16_skinny_kernel<MT, EPI, OUT_BF16, 8>), grid, dim3(512), (size_t)8 * MT * FNW * 64 * 16, s, p);
  } else {
    hipLaunchKernelGGL((gemm_bf16_skinny_kernel<MT, EPI, OUT_BF16, 4>), grid, dim3(256), (size_t)4 * MT * FNW * 64 * 16, s, p);
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

// Called by gemm_bf16_launch for M <= MODE_SKINNY_MAX_ROWS.  Returns MODE_ERR_UNSUPPORTED for what only the tiled kernel does (fused ln_2,
// K-groups, ragged K), in which case the caller falls through to the tiled kernel.
int gemm_bf16_skinny_launch(const ModeGemmDesc* d, const GemmParams& p, hipStream_t s) {
  if (p.koffs || p.ss_in || d->epilogue == MODE_EPI_RESIDUAL_NORM) return MODE_ERR_UNSUPPORTED;
  if (d->K % (128 * p.split_k) != 0 || d->N % 4 != 0) return MODE_ERR_UNSUPPORTED;   // every wave slice a multiple of 32
  const int groups = d->expert_offsets ? d->num_experts : 1;
  const bool ob = d->out_dtype == MODE_BF16;
#define MODE_SK(E) \
  case E: return ob ? launch_skinny_mt<E, true>(p, groups, d->M, s) : launch_skinny_mt<E, false>(p, groups, d->M, s);
  switch (d->epilogue) {
    MODE_SK(MODE_EPI_NONE)
    MODE_SK(MODE_EPI_BIAS)
    MODE_SK(MODE_EPI_BIAS_GELU)
    MODE_SK(MODE_EPI_SWIGLU)
    case MODE_EPI_RESIDUAL: return ob ? MODE_ERR_BAD_ARG : launch_skinny_mt<MODE_EPI_RESIDUAL, false>(p, groups, d->M, s);
    default: return MODE_ERR_BAD_ARG;
  }
#undef MODE_SK
}

}  // namespace mode
