// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T), plain / row-gathered / grouped (MoE).
//
// Design (MI355X-first):
//   * 128x128x64 workgroup tile, 256 threads = 4 wave64 in a 2x2 grid, each wave a 64x64 sub-tile held as 4x4
//     v_mfma_f32_16x16x32_bf16 accumulators (64 fp32 regs/lane).
//   * Both operands are K-contiguous ([out,in] nn.Linear weights are exactly the "B^T" layout MFMA wants), so every
//     fragment is one 16-byte ds_read_b128.
//   * LDS image: [128 rows][64 k] bf16 (128-byte rows) with the 16-byte chunk index XOR-swizzled by (row & 7): the 16 lanes of
//     a ds_read_b128 group then hit 16 distinct 16-byte slots (conflict-free), and the image is written lane-linearly so the
//     tile can be filled by `global_load_lds_dwordx4` (HBM/L2 -> LDS without VGPR staging) with the inverse swizzle applied to
//     the per-lane SOURCE address (rule: linear destination + swizzled source + swizzled read).
//   * Double-buffered LDS (2 x 32 KiB), one barrier per K-step; tile t+1 streams in while tile t feeds the MFMAs.
//   * MFMA operands are issued swapped (W fragment as "A", activation fragment as "B") so each lane owns 4 CONSECUTIVE
//     output columns of one row: bias / SwiGLU / residual epilogues run in registers and stores are 8/16 bytes per lane.
//   * SwiGLU epilogue: a workgroup takes 64 "value" rows and the matching 64 "gate" rows of W1 (rows n and 4D+n, the
//     reference's tensor_split(2)) so value*silu(gate) never leaves registers and the on-disk weight layout is untouched.
//   * Grouped mode: blockIdx -> (m-tile from the device tile table, n-tile); the expert id picks the weight slab.  No host sync.
//   * Workgroup ids are remapped per XCD and rasterised in 8-m-tile groups so the 64 tiles resident on one XCD share
//     ~4 MiB of operands (one XCD L2).
#include "mode_common.h"

namespace mode {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int GROUP_M = 8;

struct GemmParams {
  const uint16_t* A; long lda;
  const uint16_t* W; long ldw; long w_estride;
  const float* bias; long bias_estride;
  const float* resid; long ldr;
  void* C; long ldc;
  const int* a_rows; const int* tiles; const int* num_tiles;
  int M, N, K, m_tiles, n_tiles;
};

template <bool GLDS>
__device__ __forceinline__ void stage_tile(char* lds_tile, const uint16_t* const (&src)[4], int koff, int wave, int lane,
                                           uint4 (&regs)[4]) {
  // wave w fills rows [w*32, w*32+32): 4 pieces of 8 rows x 128 B = 1 KiB, lane i -> byte i*16 of the piece.
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint16_t* g = src[q] + koff;
    if constexpr (GLDS) {
      char* dst = lds_tile + (wave * 32 + q * 8) * 128;   // wave-uniform base; hardware adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    } else {
      regs[q] = *reinterpret_cast<const uint4*>(g);
    }
  }
}

__device__ __forceinline__ void commit_tile(char* lds_tile, int wave, int lane, const uint4 (&regs)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<uint4*>(lds_tile + (wave * 32 + q * 8) * 128 + lane * 16) = regs[q];
}

template <int EPI, bool OUT_BF16, bool GLDS>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- block -> (m-tile, n-tile)
  const int nblk = p.m_tiles * p.n_tiles;
  int sb = xcd_remap(blockIdx.x, nblk);
  const int per_group = GROUP_M * p.n_tiles;
  const int grp = sb / per_group, first_m = grp * GROUP_M;
  const int gsz = min(p.m_tiles - first_m, GROUP_M);
  const int rem = sb - grp * per_group;
  const int mt = first_m + rem % gsz, nt = rem / gsz;

  int row0, row_end, expert = 0;
  if (p.tiles) {
    if (mt >= *p.num_tiles) return;
    expert = p.tiles[mt * 3 + 0]; row0 = p.tiles[mt * 3 + 1]; row_end = p.tiles[mt * 3 + 2];
  } else {
    row0 = mt * BM; row_end = min(p.M, row0 + BM);
  }
  const uint16_t* W = p.W + (long)expert * p.w_estride;
  const float* bias = p.bias ? p.bias + (long)expert * p.bias_estride : nullptr;
  constexpr int NOUT = (EPI == MODE_EPI_SWIGLU) ? 64 : BN;      // output columns per n-tile
  const int n0 = nt * NOUT;

  // ---- per-thread source rows for staging (fixed over the K loop).  lane i -> row (i>>3) of an 8-row piece,
  //      physical 16-B chunk (i&7), logical chunk (i&7)^(i>>3)  [row & 7 == i>>3 because pieces start at multiples of 8].
  const int r8 = lane >> 3, lchunk = (lane & 7) ^ r8;
  const uint16_t* a_src[4];
  const uint16_t* b_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tr = wave * 32 + q * 8 + r8;                       // tile row 0..127
    int s = min(row0 + tr, row_end - 1);                         // clamp: rows past the segment re-read a valid row
    const long arow = p.a_rows ? (long)p.a_rows[s] : (long)s;
    a_src[q] = p.A + arow * p.lda + lchunk * 8;
    long brow;
    if constexpr (EPI == MODE_EPI_SWIGLU) brow = (long)min(n0 + (tr & 63), p.N - 1) + ((tr >= 64) ? p.N : 0);
    else brow = min(n0 + tr, p.N - 1);
    b_src[q] = W + brow * p.ldw + lchunk * 8;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment read offsets (bytes inside one operand tile); lane -> row (l&15), k-chunk (l>>4) [+4 for the 2nd k32 step]
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = (wm * 64 + i * 16 + fr) * 128;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int br;
    if constexpr (EPI == MODE_EPI_SWIGLU) br = (j < 2) ? (wn * 32 + j * 16) : (64 + wn * 32 + (j - 2) * 16);
    else br = wn * 64 + j * 16;
    b_off[j] = (br + fr) * 128;
  }
  const int sw = fr & 7;                                          // row & 7 for every fragment row of this lane
  const int c0 = ((fq) ^ sw) * 16, c1 = ((fq + 4) ^ sw) * 16;

  const int nk = p.K / BK;
  uint4 ra[4], rb[4];

  // prologue: stage tile 0
  stage_tile<GLDS>(smem, a_src, 0, wave, lane, ra);
  stage_tile<GLDS>(smem + TILE_BYTES, b_src, 0, wave, lane, rb);
  if constexpr (!GLDS) { commit_tile(smem, wave, lane, ra); commit_tile(smem + TILE_BYTES, wave, lane, rb); }

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * STAGE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
    if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = (kt + 1) < nk;
    if (more) {
      stage_tile<GLDS>(nxt, a_src, (kt + 1) * BK, wave, lane, ra);
      stage_tile<GLDS>(nxt + TILE_BYTES, b_src, (kt + 1) * BK, wave, lane, rb);
    }
    const char* At = cur;
    const char* Bt = cur + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = ks ? c1 : c0;
      bf16x8 af[4], bfg[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(At + a_off[i] + co);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfg[j] = *reinterpret_cast<const bf16x8*>(Bt + b_off[j] + co);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[j], af[i], acc[i][j], 0, 0, 0);   // swapped: D[n][m]
    }
    if constexpr (!GLDS) {
      if (more) { commit_tile(nxt, wave, lane, ra); commit_tile(nxt + TILE_BYTES, wave, lane, rb); }
    }
  }

  // ---- epilogue: lane owns row m = .. + (l&15), columns n = .. + (l>>4)*4 + {0..3}
  const int rows_valid = row_end - row0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ml = wm * 64 + i * 16 + fr;
    if (ml >= rows_valid) continue;
    const long m = row0 + ml;
    if constexpr (EPI == MODE_EPI_SWIGLU) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 32 + j * 16 + fq * 4;
        if (n >= p.N) continue;
        const float4 bp = *reinterpret_cast<const float4*>(bias + n);
        const float4 bg = *reinterpret_cast<const float4*>(bias + p.N + n);
        const f32x4 v = acc[i][j], g = acc[i][j + 2];
        const float o0 = (v[0] + bp.x) * silu_f(g[0] + bg.x), o1 = (v[1] + bp.y) * silu_f(g[1] + bg.y);
        const float o2 = (v[2] + bp.z) * silu_f(g[2] + bg.z), o3 = (v[3] + bp.w) * silu_f(g[3] + bg.w);
        if constexpr (OUT_BF16) {
          uint2 o; o.x = pack_bf16x2(o0, o1); o.y = pack_bf16x2(o2, o3);
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + m * p.ldc + n) = o;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + m * p.ldc + n) = make_float4(o0, o1, o2, o3);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fq * 4;
        if (n >= p.N) continue;
        f32x4 v = acc[i][j];
        if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) {
          const float4 b = *reinterpret_cast<const float4*>(bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
          if constexpr (EPI == MODE_EPI_BIAS_GELU) {
            v[0] = gelu_erf_f(v[0]); v[1] = gelu_erf_f(v[1]); v[2] = gelu_erf_f(v[2]); v[3] = gelu_erf_f(v[3]);
          }
        } else if constexpr (EPI == MODE_EPI_RESIDUAL) {
          const float4 r = *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n);
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if constexpr (OUT_BF16) {
          uint2 o; o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + m * p.ldc + n) = o;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
}

template <int EPI, bool OUT_BF16>
static int launch_epi(const GemmParams& p, int nblk, bool glds, hipStream_t s) {
  const size_t lds = 2 * STAGE_BYTES;
  if (glds) {
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, OUT_BF16, true>), dim3(nblk), dim3(NTHREADS), lds, s, p);
  } else {
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, OUT_BF16, false>), dim3(nblk), dim3(NTHREADS), lds, s, p);
  }
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

int g_use_glds = -1;

int gemm_bf16_launch(const ModeGemmDesc* d, hipStream_t s) {
  if (d->K % BK != 0 || d->K <= 0) return MODE_ERR_UNSUPPORTED;
  if (d->N % 4 != 0 || d->lda % 8 != 0 || d->ldw % 8 != 0 || d->ldc % 4 != 0) return MODE_ERR_UNSUPPORTED;
  if (d->tiles && d->tile_m != BM) return MODE_ERR_BAD_ARG;
  if ((d->epilogue == MODE_EPI_BIAS || d->epilogue == MODE_EPI_BIAS_GELU || d->epilogue == MODE_EPI_SWIGLU) && !d->bias)
    return MODE_ERR_BAD_ARG;
  if (d->epilogue == MODE_EPI_RESIDUAL && !d->resid) return MODE_ERR_BAD_ARG;
  if (d->M <= 0) return MODE_OK;
  if (g_use_glds < 0) {
    const char* e = getenv("MODE_GEMM_GLDS");
    g_use_glds = (e && e[0] == '0') ? 0 : 1;
  }
  GemmParams p;
  p.A = (const uint16_t*)d->A; p.lda = d->lda;
  p.W = (const uint16_t*)d->W; p.ldw = d->ldw; p.w_estride = d->w_expert_stride;
  p.bias = d->bias; p.bias_estride = d->bias_expert_stride;
  p.resid = d->resid; p.ldr = d->ldr; p.C = d->C; p.ldc = d->ldc;
  p.a_rows = d->a_rows; p.tiles = d->tiles; p.num_tiles = d->num_tiles;
  p.M = d->M; p.N = d->N; p.K = d->K;
  const int nout = (d->epilogue == MODE_EPI_SWIGLU) ? 64 : BN;
  p.n_tiles = (d->N + nout - 1) / nout;
  p.m_tiles = d->tiles ? d->max_tiles : (d->M + BM - 1) / BM;
  const int nblk = p.m_tiles * p.n_tiles;
  const bool ob = d->out_dtype == MODE_BF16;
  const bool glds = g_use_glds != 0;
#define MODE_CASE(E)                                                          \
  case E: return ob ? launch_epi<E, true>(p, nblk, glds, s) : launch_epi<E, false>(p, nblk, glds, s);
  switch (d->epilogue) {
    MODE_CASE(MODE_EPI_NONE)
    MODE_CASE(MODE_EPI_BIAS)
    MODE_CASE(MODE_EPI_BIAS_GELU)
    MODE_CASE(MODE_EPI_RESIDUAL)
    MODE_CASE(MODE_EPI_SWIGLU)
    default: return MODE_ERR_BAD_ARG;
  }
#undef MODE_CASE
}

}  // namespace mode
