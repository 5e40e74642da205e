// Fused QKV projection + qk-RMSNorm + causal attention for gfx950 (modedit.py:108-110, 125-127, 141-165): one launch instead of the packed-QKV GEMM
// and the attention kernel, no [N, 3D] q | k | v round trip through HBM.
//
// The sequence is T <= 16 tokens per sample, so a 64-row M-tile holds SPT = min(4, 64 / T) whole samples, and the 3 * HD weight rows
// [q_h | k_h | v_h] of ONE head are everything the attention of (those samples, head h) needs.  Workgroup = (sample group, head):
//   * GEMM: 64 x 384 output tile (HD = 128), K = D in 64-deep steps, 4 wave64s side by side (each 64 rows x 96 columns: 4 x 6 accumulators of
//     v_mfma_f32_16x16x32_bf16).  Operands arrive by LDS-DMA (global_load_lds_dwordx4) into the XOR-swizzled, lane-linear image of gemm_bf16.hip, a
//     two-slot ring with counted vmcnt waits and raw s_barriers, both k32 halves of a K-step double-buffered in registers.  The weight tile gathers its
//     rows from the three places of the packed [3D, D] matrix (row seg * D + h * HD + r).
//   * Epilogue: bias add and bf16 rounding exactly where the stand-alone GEMM does them, the tile goes to LDS as [64][q | k | v] (padded rows), and
//     wave w runs the attention of sample w of the group with the SAME wave-level body as the stand-alone kernel (attn_core.h), reading its fragments
//     from LDS instead of the qkv buffer; the output rows leave through 16-byte global stores, heads merged.
// Bit-identical to mode_gemm(MODE_EPI_BIAS) + mode_attn_block_fwd by construction: the accumulation is the same k-ordered MFMA chain (swapped
// operands, k32 halves in ascending order), the rounding points are the same, the attention arithmetic is one shared function.
// At B = 128 (32 sample groups x 8 heads = 256 workgroups, one per CU) every CU streams 16 x 56 KiB of operands - the same L2 -> LDS volume per CU as
// the busiest CU of the 128 x 64 tiling it replaces (3 tiles x 384 KiB), evenly spread.
#include "attn_core.h"
#include "mode_common.h"

namespace mode {

struct QkvAttnParams {
  const uint16_t* A; long lda;          // [B*T, D] bf16 (ln_1(x) + c)
  const uint16_t* W; long ldw;          // packed [3D, D] bf16
  const float* bias;                    // [3D]
  const float* qg; const float* kg;     // qk-norm gains [HD]
  uint16_t* y; long ldy;                // [B*T, D] bf16
  int B, T, H, D, spt, m_tiles;
  float eps;
};

namespace qa {
constexpr int BM = 64, BK = 64, WN = 4;
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ void lds_read128(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int STRIDE, int CNT, int I = 0>
__device__ __forceinline__ void lds_read_seq(bf16x8* dst, uint32_t addr) {
  if constexpr (I < CNT) {
    lds_read128<I * STRIDE>(dst[I], addr);
    lds_read_seq<STRIDE, CNT, I + 1>(dst, addr);
  }
}

// q | k | v tile of the epilogue in LDS: rows of 3 * HD bf16 + 16 bytes (the 16 token rows a ds_read_b128 group touches land on 16 distinct 4-bank groups)
// The GEMM epilogue stores 8 bytes per lane, 16 lanes = 16 consecutive tile rows per LDS cycle group: with a 196-dword pitch rows r and r + 8 share their banks
// (2-way conflict on every store: 147 K conflict cycles per launch in round 4's PMC pass).  Rows with bit 3 set therefore keep the two 8-byte halves of every
// 16-byte chunk SWAPPED (store address ^ 8); the readers below swap them back in registers.
template <int HD>
struct LdsSrc {
  static constexpr int PITCH = 3 * HD * 2 + 16;
  const char* tile;   // first token row of this sample
  int row0;           // its row index inside the tile (the half-swap goes by the TILE row)
  __device__ __forceinline__ uint4 ld(int row, int byte) const {
    const uint4 t = *reinterpret_cast<const uint4*>(tile + row * PITCH + byte);
    const bool sw = ((row0 + row) & 8) != 0;
    return make_uint4(sw ? t.z : t.x, sw ? t.w : t.y, sw ? t.x : t.z, sw ? t.y : t.w);
  }
  __device__ __forceinline__ uint4 q(int row, int d0) const { return ld(row, d0 * 2); }
  __device__ __forceinline__ uint4 k(int row, int d0) const { return ld(row, (HD + d0) * 2); }
  __device__ __forceinline__ uint4 v(int row, int d0) const { return ld(row, (2 * HD + d0) * 2); }
};
}  // namespace qa

// W3: the weight tile (6/7 of a K-step's bytes) rides a THREE-slot ring, the activation tile a two-slot one - 2 x 8 + 3 x 48 KiB = exactly the CU's 160 KiB.
// With two slots per operand the DMA engine idles from the moment a K-step's data has landed until the barrier after which the next stage is issued; with
// the weights of K-step kt+2 requested at the top of K-step kt the L2 -> LDS stream never stops (counted vmcnt: the newest 12 weight pieces stay in flight
// across the wait).  Same arithmetic, same order.
// WM: wave rows (1: four waves of 64 x 96; 2: eight waves of 32 x 96 - two per SIMD, one's MFMAs cover the other's waits).
template <int HD, bool W3, int WM>
__global__ __launch_bounds__(WM * 256, 1) void qkv_attn_kernel(const QkvAttnParams p) {
  using namespace qa;
  constexpr int NW = WM * WN, TM = BM / WM;
  constexpr int BN = 3 * HD, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;             // 1-KiB DMA pieces (8 rows x 128 B) per wave per operand tile
  constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2;
  constexpr int GROUP_M = 16;                                   // sample groups per rasterisation group: an XCD's 32 workgroups = 16 groups x 2 heads
  static_assert(TN % 16 == 0 && BN % (8 * NW) == 0, "head_dim must give whole fragments / DMA pieces per wave");
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int W_BASE = W3 ? 2 * A_BYTES : A_BYTES;               // W3: [A slot 0 | A slot 1 | W slot 0 | W slot 1 | W slot 2]; else [A | W] per stage
  static_assert(BM * LdsSrc<HD>::PITCH <= 2 * STAGE_BYTES, "the q | k | v tile reuses the operand ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- block -> (sample group, head), XCD-contiguous chunks, groups rasterised m-fastest
  const int nblk = p.m_tiles * p.H;
  const int sb = xcd_remap(blockIdx.x, nblk);
  const int per_group = GROUP_M * p.H;
  const int grp = sb / per_group, first_m = grp * GROUP_M;
  const int gsz = min(p.m_tiles - first_m, GROUP_M);
  const int rem = sb - grp * per_group;
  const int mt = first_m + rem % gsz, head = rem / gsz;
  const int s0 = mt * p.spt;                                      // first sample of the group
  const int ns = min(p.spt, p.B - s0);                            // samples in this group
  const int row0 = s0 * p.T, row_end = row0 + ns * p.T;

  // ---- per-thread DMA sources (fixed over the K loop): lane i -> row (i>>3) of an 8-row piece, physical 16-B chunk (i&7), logical chunk (i&7)^(i>>3)
  const int r8 = lane >> 3, lchunk = (lane & 7) ^ r8;
  const uint16_t* a_src[PA];
  const uint16_t* b_src[PB];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int tr = (wave * PA + q) * 8 + r8;
    a_src[q] = p.A + (long)min(row0 + tr, row_end - 1) * p.lda + lchunk * 8;      // rows past the group re-read a valid row (never used)
  }
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int tr = (wave * PB + q) * 8 + r8;                      // tile row: segment (q / k / v) * HD + r
    const long wrow = (long)(tr / HD) * p.D + head * HD + tr % HD;
    b_src[q] = p.W + wrow * p.ldw + lchunk * 8;
  }
  auto stage_a = [&](char* base, int kt) {
    const int koff = kt * BK;
#pragma unroll
    for (int q = 0; q < PA; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + koff),
                                       (__attribute__((address_space(3))) void*)(base + (wave * PA + q) * 1024), 16, 0, 0);
  };
  auto stage_w = [&](char* base, int kt) {
    const int koff = kt * BK;
#pragma unroll
    for (int q = 0; q < PB; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + koff),
                                       (__attribute__((address_space(3))) void*)(base + (wave * PB + q) * 1024), 16, 0, 0);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing: lane -> row (l&15), k-chunk (l>>4) [+4 for the 2nd k32 half]; fragment i/j = immediate offset i*16 rows
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = fr & 7;
  const int c0 = (fq ^ sw) * 16, c1 = ((fq + 4) ^ sw) * 16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t a_base = lds0 + (wm * TM + fr) * 128;
  const uint32_t b_base = lds0 + W_BASE + (wn * TN + fr) * 128;
  bf16x8 fa0[FM], fb0[FN], fa1[FM], fb1[FN];
  auto read_frags = [&](bf16x8* fa, bf16x8* fb, uint32_t a_off, uint32_t w_off, int co) {
    lds_read_seq<2048, FM>(fa, a_base + a_off + co);
    lds_read_seq<2048, FN>(fb, b_base + w_off + co);
  };
  auto mma = [&](const bf16x8(&fa)[FM], const bf16x8(&fb)[FN]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // swapped operands: D[n][m], like gemm_bf16.hip
    __builtin_amdgcn_s_setprio(0);
  };

  auto ksteps = [&](uint32_t a_off, uint32_t w_off) {
    read_frags(fa0, fb0, a_off, w_off, c0);
    read_frags(fa1, fb1, a_off, w_off, c1);
    wait_lgkmcnt<FM + FN>();                                       // first half arrived, second half still in flight
    __builtin_amdgcn_sched_barrier(0);
    mma(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    mma(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- K loop: one barrier per K-step, the following K-steps' DMA in flight under this one's MFMAs
  const int nk = p.D / BK;
  if constexpr (W3) {
    stage_a(smem, 0);
    stage_w(smem + W_BASE, 0);
    if (nk > 1) stage_w(smem + W_BASE + W_BYTES, 1);
    int ws = 0;                                                    // weight slot of K-step kt
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<PB>(); else wait_vmcnt<0>();     // A(kt), W(kt) landed; the 12 pieces of W(kt+1) stay in flight
      __builtin_amdgcn_s_barrier();                                // K-step kt visible to all waves; everyone is done with K-step kt-1
      if (kt + 1 < nk) stage_a(smem + ((kt + 1) & 1) * A_BYTES, kt + 1);                       // (issued BEFORE the weights: the counted wait above relies on it)
      if (kt + 2 < nk) stage_w(smem + W_BASE + (ws == 0 ? 2 : ws - 1) * W_BYTES, kt + 2);      // slot of K-step kt-1 = (ws + 2) % 3
      ksteps((kt & 1) * A_BYTES, ws * W_BYTES);
      ws = ws == 2 ? 0 : ws + 1;
    }
  } else {
    stage_a(smem, 0); stage_w(smem + A_BYTES, 0);
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                                // tile kt visible to all waves; everyone is done with tile kt-1
      if (kt + 1 < nk) { stage_a(smem + (slot ^ 1) * STAGE_BYTES, kt + 1); stage_w(smem + (slot ^ 1) * STAGE_BYTES + A_BYTES, kt + 1); }
      ksteps(slot * STAGE_BYTES, slot * STAGE_BYTES);
      slot ^= 1;
    }
  }

  // ---- epilogue 1: + bias, bf16, -> LDS tile [row][q | k | v]  (lane: row wm*TM + i*16 + fr, 4 consecutive columns wn*TN + j*16 + fq*4)
  constexpr int PITCH = LdsSrc<HD>::PITCH;
  __builtin_amdgcn_s_barrier();                                    // all waves are done reading operand tiles
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int nl = wn * TN + j * 16 + fq * 4;                      // tile column
    const float4 b = *reinterpret_cast<const float4*>(p.bias + (long)(nl / HD) * p.D + head * HD + nl % HD);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      f32x4 v = acc[i][j];
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      const int trow = wm * TM + i * 16 + fr;
      *reinterpret_cast<uint2*>(smem + trow * PITCH + ((nl * 2) ^ (trow & 8))) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue 2: wave w = sample w of the group: qk-RMSNorm, causal softmax, PV - the stand-alone kernel's body on LDS operands
  if (wave < ns) {
    const int b = s0 + wave;
    const LdsSrc<HD> src{smem + wave * p.T * PITCH, wave * p.T};
    attn_wave_bf16<(HD + 31) / 32>(src, p.qg, p.kg, p.y + (long)b * p.T * p.ldy + head * HD, p.ldy, p.T, HD, p.eps, 0u, 0u, 1.0f, b * p.H + head, lane);
  }
}

int g_fuse_qkv_attn = 1;           // "fuse_qkv_attn" option: 1 = the chain uses this kernel where it applies, 0 = GEMM + attention kernels
int g_qkv_attn_waves = 8;          // "qkv_attn_waves" option: 8 (default: two waves per SIMD, 18.1 us at B = 128) or 4 waves per workgroup (20.0 us)
int g_qkv_attn_w3 = 1;             // "qkv_attn_w3" option: 1 = three-slot weight ring (160 KiB of LDS), 0 = two-slot ring (112 KiB)
int g_fuse_qkv_attn_min_b = 56;    // "fuse_qkv_attn_min_b": smallest batch the chain takes it for.  A workgroup lives ~17 us whatever the batch (one per CU, 16 K-steps of 56 KiB);
                                   // measured (scripts/qkv_attn_probe.py) fused / two kernels: B = 32 16.6 / 14.6 us, B = 64 17.0 / 18.6, B = 128 18.1 / 23.6

}  // namespace mode

using namespace mode;

extern "C" int mode_qkv_attn_fwd(const ModeQkvAttnDesc* d, void* stream) {
  if (!d || !d->h || !d->wqkv || !d->bqkv || !d->q_gain || !d->k_gain || !d->y || d->B < 0 || d->T <= 0 || d->H <= 0 || d->D <= 0) return MODE_ERR_BAD_ARG;
  if (d->dtype != MODE_BF16) return MODE_ERR_UNSUPPORTED;          // fp32 parity mode keeps the two kernels
  if (d->B == 0) return MODE_OK;
  const int HD = d->D / d->H;
  if (HD * d->H != d->D || HD != 128 || d->T > 16 || d->D % 64) return MODE_ERR_UNSUPPORTED;
  if (d->ldh % 8 || d->ldw % 8 || d->ldy % 8 || d->ldh < d->D || d->ldw < d->D || d->ldy < d->D) return MODE_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(d->h) | reinterpret_cast<uintptr_t>(d->wqkv) | reinterpret_cast<uintptr_t>(d->y) | reinterpret_cast<uintptr_t>(d->bqkv) |
       reinterpret_cast<uintptr_t>(d->q_gain) | reinterpret_cast<uintptr_t>(d->k_gain)) & 15) return MODE_ERR_UNSUPPORTED;
  QkvAttnParams p;
  p.A = (const uint16_t*)d->h; p.lda = d->ldh; p.W = (const uint16_t*)d->wqkv; p.ldw = d->ldw; p.bias = d->bqkv; p.qg = d->q_gain; p.kg = d->k_gain;
  p.y = (uint16_t*)d->y; p.ldy = d->ldy; p.B = d->B; p.T = d->T; p.H = d->H; p.D = d->D; p.eps = d->eps;
  p.spt = 64 / d->T < 4 ? 64 / d->T : 4;                           // whole samples per 64-row tile, one wave each
  p.m_tiles = (d->B + p.spt - 1) / p.spt;
  constexpr int LDS2 = 2 * (qa::BM + 3 * 128) * qa::BK * 2;        // two-slot operand ring (the q | k | v tile of the epilogue reuses it)
  constexpr int LDS3 = (2 * qa::BM + 3 * 3 * 128) * qa::BK * 2;    // activations x 2 + weights x 3 = 160 KiB
  const bool w3 = g_qkv_attn_w3 != 0;
  const bool w8 = g_qkv_attn_waves == 8;
  auto kern = w8 ? (w3 ? qkv_attn_kernel<128, true, 2> : qkv_attn_kernel<128, false, 2>) : (w3 ? qkv_attn_kernel<128, true, 1> : qkv_attn_kernel<128, false, 1>);
  const int lds = w3 ? LDS3 : LDS2;
  static LdsLimitOnce lds_once[2][2];
  {
    const int rc = lds_once[w8][w3].ensure(reinterpret_cast<const void*>(kern), lds);
    if (rc != MODE_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(p.m_tiles * p.H), dim3(w8 ? 512 : 256), lds, (hipStream_t)stream, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}
