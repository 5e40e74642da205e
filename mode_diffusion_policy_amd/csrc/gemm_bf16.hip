// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T), plain / row-gathered / grouped (MoE).
//
// Design (MI355X-first):
//   * Workgroup tile BM x BN x 64 with WM x WN wave64s, each wave a (BM/WM) x (BN/WN) sub-tile of v_mfma_f32_16x16x32_bf16
//     accumulators.  Shipped geometries: 256x128 (8 waves), 128x128 (4 waves), 128x64 (4 waves; small-N / small-M problems that would
//     not fill 256 CUs with bigger tiles).
//   * Both operands are K-contiguous ([out,in] nn.Linear weights are exactly the "B^T" layout MFMA wants), so every
//     fragment is one 16-byte ds_read_b128.
//   * LDS image per operand tile: [rows][64 k] bf16 (128-byte rows) with the 16-byte chunk index XOR-swizzled by (row & 7): the 16 lanes
//     of a ds_read_b128 group hit 16 distinct 16-byte slots (measured SQ_LDS_BANK_CONFLICT = 0), and the image is written lane-linearly so
//     the tile is filled by `global_load_lds_dwordx4` (L2/HBM -> LDS DMA, no VGPR staging) with the inverse swizzle applied to the per-lane
//     SOURCE address (linear destination + swizzled source + swizzled read).
//   * NS-slot LDS ring with COUNTED `s_waitcnt vmcnt(N)` and a raw `s_barrier` (never __syncthreads, which drains vmcnt): the loads of
//     the younger tiles stay in flight across the barrier, one barrier per 64-deep K-step.
//   * Two-phase software pipeline inside a K-step: MFMA operands are double-buffered in VGPRs, so the ds_reads of one k32 half are
//     always in flight under the 16..32 MFMAs of the other half (measured: an un-pipelined read->wait->MFMA body left the matrix pipe
//     idle for the whole LDS round trip twice per K-step).
//   * MFMA operands are issued swapped (W fragment as "A", activation fragment as "B") so each lane owns 4 CONSECUTIVE
//     output columns of one row: bias / SwiGLU epilogues run in registers.
//   * SwiGLU epilogue: a workgroup takes BN/2 "value" rows and the matching BN/2 "gate" rows of W1 (rows n and 4D+n, the
//     reference's tensor_split(2)) so value*silu(gate) never leaves registers and the on-disk weight layout is untouched.
//   * Output tile is transposed through LDS (padded rows, conflict-free) so global stores — and the fp32 residual read of the
//     c_proj epilogue — are full contiguous row segments, 16 bytes per lane (measured: direct 8-byte fragment stores cost 18 us of
//     the 88 us expert up-projection).
//   * Grouped mode: blockIdx -> (expert, row range) from the device-side expert offsets; the expert id picks the weight slab.  No host sync.
//   * Workgroup ids are remapped per XCD and rasterised in m-tile groups so the tiles resident on one XCD share
//     ~4 MiB of operands (one XCD L2; measured TCC hit rate 85 %).
#include "mode_common.h"

namespace mode {

constexpr int BK = 64;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int N>
__device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// ds_read_b128 the compiler does not track: the MFMA operands are requested with hand-counted lgkmcnt waits so the second k32 half's
// LDS round trip stays in flight under the first half's MFMAs (hipcc's own scoreboard emitted lgkmcnt(0) before the first MFMA).
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

// wait until at most `tiles` K-tiles (LOADS VMEM ops each) are still in flight for this wave
template <int LOADS, int MAXT>
__device__ __forceinline__ void wait_tiles_in_flight(int tiles) {
  if (MAXT >= 3 && tiles >= 3) wait_vmcnt<3 * LOADS>();
  else if (MAXT >= 2 && tiles == 2) wait_vmcnt<2 * LOADS>();
  else if (MAXT >= 1 && tiles == 1) wait_vmcnt<LOADS>();
  else wait_vmcnt<0>();
}

template <int BM, int BN, int WM, int WN, int NS, int EPI, bool OUT_BF16, int LR = 0>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN >= 16 ? 1 : (LR ? LR : (NS == 1 ? 3 : 2)))) void gemm_bf16_kernel(const GemmParams p) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;         // 1-KiB DMA pieces (8 rows x 128 B) per wave per operand tile
  constexpr int LOADS = PA + PB;                              // VMEM ops per wave per K-tile
  constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2;
  const int GROUP_M = p.group_m > 0 ? p.group_m : ((BM >= 256) ? 4 : 8);
  constexpr int NOUT = (EPI == MODE_EPI_SWIGLU) ? BN / 2 : BN;      // output columns per n-tile
  constexpr int ESZ = OUT_BF16 ? 2 : 4;
  constexpr int CROW = NOUT * ESZ;                            // LDS row of the output tile; 16-B chunks XOR-swizzled by the row
  constexpr int CPR = CROW / 16;                              // 16-byte chunks per output row
  constexpr int CSWZ = (CPR < 16 ? CPR : 16) - 1;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into whole DMA pieces per wave");
  static_assert(EPI != MODE_EPI_SWIGLU || (FN % 2 == 0), "SwiGLU needs value+gate fragments in every wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- block -> (m-tile, n-tile): XCD-contiguous chunks, GROUP_M m-tiles rasterised m-fastest
  const int nblk = p.m_tiles * p.n_tiles;
  const int sb = xcd_remap(blockIdx.x, nblk);
  const int per_group = GROUP_M * p.n_tiles;
  const int grp = sb / per_group, first_m = grp * GROUP_M;
  const int gsz = min(p.m_tiles - first_m, GROUP_M);
  const int rem = sb - grp * per_group;
  const int mt = first_m + rem % gsz, nt = rem / gsz;

  int row0 = 0, row_end = 0, expert = 0;
  if (p.offsets) {
    // grouped (MoE): rows are sorted by expert; expert e owns sorted rows [offsets[e], offsets[e+1]) and ceil(count/BM) m-tiles.
    // The offsets are fetched with independent scalar loads (no dependent chain), then scanned in registers.
    int t = mt;
    bool found = false;
    if (p.E <= 8) {
      int o[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) o[e] = p.offsets[min(e, p.E)];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (!found && e < p.E) {
          const int nt_e = (o[e + 1] - o[e] + BM - 1) / BM;
          if (t < nt_e) { row0 = o[e] + t * BM; row_end = min(o[e + 1], row0 + BM); expert = e; found = true; }
          else t -= nt_e;
        }
      }
    } else {
      for (int e = 0; e < p.E && !found; ++e) {
        const int o0 = p.offsets[e], o1 = p.offsets[e + 1];
        const int nt_e = (o1 - o0 + BM - 1) / BM;
        if (t < nt_e) { row0 = o0 + t * BM; row_end = min(o1, row0 + BM); expert = e; found = true; }
        else t -= nt_e;
      }
    }
    if (!found) return;
  } else {
    row0 = mt * BM; row_end = min(p.M, row0 + BM);
  }
  const uint16_t* W = p.W + (long)expert * p.w_estride;
  const float* bias = p.bias ? p.bias + (long)expert * p.bias_estride : nullptr;
  const int n0 = nt * NOUT;

  // ---- per-thread DMA sources (fixed over the K loop).  lane i -> row (i>>3) of an 8-row piece, physical 16-B chunk (i&7),
  //      logical chunk (i&7)^(i>>3)   [row & 7 == i>>3 because pieces start at multiples of 8].
  const int r8 = lane >> 3, lchunk = (lane & 7) ^ r8;
  const uint16_t* a_src[PA];
  const uint16_t* b_src[PB];
  [[maybe_unused]] long a_tok[PA];                                 // token row behind each DMA row (fused ln_2 consumer)
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int tr = (wave * PA + q) * 8 + r8;
    const int s = min(row0 + tr, row_end - 1);                   // rows past the segment re-read a valid row (never stored)
    const long arow = p.a_rows ? (long)p.a_rows[s] : (long)s;
    a_src[q] = p.A + arow * p.lda + lchunk * 8;
    a_tok[q] = arow;
  }
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int tr = (wave * PB + q) * 8 + r8;
    long brow;
    if constexpr (EPI == MODE_EPI_SWIGLU) brow = (long)min(n0 + (tr % (BN / 2)), p.N - 1) + ((tr >= BN / 2) ? p.N : 0);
    else brow = min(n0 + tr, p.N - 1);
    b_src[q] = W + brow * p.ldw + lchunk * 8;
  }

  // fused ln_2 consumer: 1 / max(|x_row| K^-1/2, eps) of this tile's rows from the producer's per-64-column partial sums -> LDS strip past
  // the ring (read again in the epilogue; the loads overlap the first operand tile's fill)
  constexpr int RING_BYTES = NS * STAGE_BYTES, OUT_BYTES = (BM / WM) * NOUT * ESZ;
  float* inv_nrm = reinterpret_cast<float*>(smem + (RING_BYTES > OUT_BYTES ? RING_BYTES : OUT_BYTES));
  auto stage = [&](int slot, int kt) {
    char* base = smem + slot * STAGE_BYTES;
    const int koff = kt * BK;
#pragma unroll
    for (int q = 0; q < PA; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + koff),
                                       (__attribute__((address_space(3))) void*)(base + (wave * PA + q) * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < PB; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + koff),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave * PB + q) * 1024), 16, 0, 0);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing: lane -> row (l&15), k-chunk (l>>4) [+4 for the 2nd k32 half]; fragment i/j = immediate offset i*16 rows
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = fr & 7;                                          // row & 7 for every fragment row of this lane
  const int c0 = (fq ^ sw) * 16, c1 = ((fq + 4) ^ sw) * 16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t a_base = lds0 + (wm * TM + fr) * 128;
  constexpr int BJ = (EPI == MODE_EPI_SWIGLU) ? FN / 2 : FN;       // B fragments per contiguous run (value / gate runs for SwiGLU)
  const uint32_t b_base = lds0 + A_BYTES + ((EPI == MODE_EPI_SWIGLU ? wn * (TN / 2) : wn * TN) + fr) * 128;

  bf16x8 fa0[FM], fb0[FN], fa1[FM], fb1[FN];                     // both k32 halves of a K-step live in VGPRs
  auto read_frags = [&](bf16x8* fa, bf16x8* fb, uint32_t slot_off, int co) {
    lds_read_seq<2048, FM>(fa, a_base + slot_off + co);
    lds_read_seq<2048, BJ>(fb, b_base + slot_off + co);
    if constexpr (EPI == MODE_EPI_SWIGLU) lds_read_seq<2048, BJ>(fb + BJ, b_base + slot_off + co + (BN / 2) * 128);
  };
  auto mma = [&](const bf16x8(&fa)[FM], const bf16x8(&fb)[FN]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // swapped operands: D[n][m]
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- main loop
  int nk = p.K / BK / p.split_k;                                  // K-tiles of this slice
  int kt0 = blockIdx.y * nk;
  if (p.koffs) {                                                   // K-group mode: this workgroup reduces over one expert's (64-padded) rows
    const int kb = p.koffs[blockIdx.z], ke = p.koffs[blockIdx.z + 1];
    kt0 = kb / BK; nk = (ke - kb) / BK;
  }
  constexpr int PRE = (NS == 1) ? 1 : NS - 1;                      // tiles in flight before the loop
#pragma unroll
  for (int s = 0; s < PRE; ++s)
    if (s < nk) stage(s, kt0 + s);
  if constexpr (EPI == MODE_EPI_SWIGLU) {
    // fused ln_2 consumer: 1 / max(|x_row| K^-1/2, eps) of this tile's rows -> LDS strip past the ring (read in the epilogue).  The 8 lanes that
    // DMA one A row share its token id (a_tok): each fetches every 8th partial sum, a 3-step shuffle adds them — issued right behind the first
    // operand tile, no second dependent a_rows load.
    if (p.ss_in) {
      const int j0 = lane & 7, nss = p.ss_n;
      float part[PA];
#pragma unroll
      for (int q = 0; q < PA; ++q) {                                // all 2*PA loads in flight together (clamped index + select: no branches)
        const float* sp = p.ss_in + a_tok[q] * nss;
        const float v0 = sp[min(j0, nss - 1)], v1 = sp[min(j0 + 8, nss - 1)];
        part[q] = (j0 < nss ? v0 : 0.f) + (j0 + 8 < nss ? v1 : 0.f);
      }
      if (nss > 16) {                                                // D > 1024
#pragma unroll
        for (int q = 0; q < PA; ++q)
          for (int j = j0 + 16; j < nss; j += 8) part[q] += p.ss_in[a_tok[q] * nss + j];
      }
#pragma unroll
      for (int sh = 1; sh < 8; sh <<= 1) {
#pragma unroll
        for (int q = 0; q < PA; ++q) part[q] += __shfl_xor(part[q], sh, 64);
      }
      const float rk = rsqrtf((float)p.K);
#pragma unroll
      for (int q = 0; q < PA; ++q)
        if (j0 == 0) inv_nrm[(wave * PA + q) * 8 + r8] = __frcp_rn(fmaxf(__fsqrt_rn(part[q]) * rk, p.ss_eps));
    }
  }

  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt landed for this wave's pieces; younger tiles stay in flight across the barrier
    wait_tiles_in_flight<LOADS, (NS >= 2 ? NS - 2 : 0)>(min(NS - 2, nk - 1 - kt));
    __builtin_amdgcn_s_barrier();                                  // tile kt visible to all waves; everyone is done with tile kt-1
    if constexpr (NS >= 2) {
      if (kt + NS - 1 < nk) stage((slot + NS - 1) % NS, kt0 + kt + NS - 1);   // refill the slot tile kt-1 just vacated
    }
    const uint32_t so = slot * STAGE_BYTES;
    if constexpr (NW >= 16 || LR) {
      // 16 waves share one 256x256 tile (4 per SIMD, <= 128 VGPRs each): one k32 half of fragments at a time; the LDS round trip of a
      // wave is covered by the MFMAs of the three other waves on its SIMD
      read_frags(fa0, fb0, so, c0);
      wait_lgkmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
      mma(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(fa0, fb0, so, c1);
      wait_lgkmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
      mma(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
    read_frags(fa0, fb0, so, c0);
    read_frags(fa1, fb1, so, c1);
    wait_lgkmcnt<FM + FN>();                                       // first half arrived, second half still in flight
    __builtin_amdgcn_sched_barrier(0);
    mma(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);                             // keep the first half's MFMAs above the second wait
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    mma(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NS == 1) {
      if (kt + 1 < nk) {
        __builtin_amdgcn_s_barrier();                              // single buffer: every wave is done reading before the refill
        stage(0, kt0 + kt + 1);
      }
    } else {
      slot = (slot + 1 == NS) ? 0 : slot + 1;
    }
  }

  // ---- epilogue: bias / SwiGLU in registers (lane owns row ..+(l&15), 4 consecutive columns) -> swizzled LDS tile -> coalesced stores.
  //      One pass when the whole output tile fits the operand ring, else one pass per wave-row group (TM rows).
  char* Cout = reinterpret_cast<char*>(p.C) + ((long)blockIdx.y * p.split_stride + (long)blockIdx.z * p.c_gstride) * ESZ;
  const int rows_valid = row_end - row0;
  __builtin_amdgcn_s_barrier();                                    // all waves are done reading operand tiles
  constexpr int EPASS = (BM * CROW <= NS * STAGE_BYTES) ? 1 : WM;   // epilogue passes
  constexpr int RP = BM / EPASS;                                   // rows per pass
  [[maybe_unused]] float rs[FM];                                   // fused ln_2: inverse row norms of this lane's rows, read once (x 1.0f is exact)
  if constexpr (EPI == MODE_EPI_SWIGLU) {
#pragma unroll
    for (int i = 0; i < FM; ++i) rs[i] = p.ss_in ? inv_nrm[wm * TM + i * 16 + fr] : 1.0f;
  }
#pragma unroll 1
  for (int g = 0; g < EPASS; ++g) {
    if (EPASS == 1 || wm == g) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int rl = (EPASS == 1 ? wm * TM : 0) + i * 16 + fr;   // row inside this pass
        char* crow = smem + rl * CROW;
        const int rsw = rl & CSWZ;
        auto cpos = [&](int nl) { const int b = nl * ESZ; return crow + ((((b >> 4) ^ rsw) << 4) | (b & 15)); };
        if constexpr (EPI == MODE_EPI_SWIGLU) {
#pragma unroll
          for (int j = 0; j < FN / 2; ++j) {
            const int nl = wn * (TN / 2) + j * 16 + fq * 4;
            const int n = min(n0 + nl, p.N - 4);
            const float4 bp = *reinterpret_cast<const float4*>(bias + n);
            const float4 bg = *reinterpret_cast<const float4*>(bias + p.N + n);
            const f32x4 v = acc[i][j], gt = acc[i][j + FN / 2];
            const float o0 = swiglu_f(v[0], gt[0], rs[i], bp.x, bg.x), o1 = swiglu_f(v[1], gt[1], rs[i], bp.y, bg.y);
            const float o2 = swiglu_f(v[2], gt[2], rs[i], bp.z, bg.z), o3 = swiglu_f(v[3], gt[3], rs[i], bp.w, bg.w);
            if constexpr (OUT_BF16) *reinterpret_cast<uint2*>(cpos(nl)) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
            else *reinterpret_cast<float4*>(cpos(nl)) = make_float4(o0, o1, o2, o3);
          }
        } else {
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int nl = wn * TN + j * 16 + fq * 4;
            f32x4 v = acc[i][j];
            if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) {
              const float4 b = *reinterpret_cast<const float4*>(bias + min(n0 + nl, p.N - 4));
              v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
              if constexpr (EPI == MODE_EPI_BIAS_GELU) {
                v[0] = gelu_erf_f(v[0]); v[1] = gelu_erf_f(v[1]); v[2] = gelu_erf_f(v[2]); v[3] = gelu_erf_f(v[3]);
              }
            }
            if constexpr (OUT_BF16) *reinterpret_cast<uint2*>(cpos(nl)) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            else *reinterpret_cast<float4*>(cpos(nl)) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int EPC = 16 / ESZ;                                  // elements per 16-byte chunk
    if constexpr (EPI == MODE_EPI_RESIDUAL_NORM && !OUT_BF16) {
      // v = acc + resid -> C (fp32); bf16(v * gain) -> C2; sum of v^2 over this row's 64-column group -> ss_out (16 lanes = 64 columns, fixed
      // shuffle order: deterministic).  Four chunks per batch: the residual / gain loads of a batch are all in flight before the first use;
      // every lane takes part in the shuffles, out-of-range lanes contribute zeros.
      constexpr int ITERS = RP * CPR / NT, BATCH = ITERS % 4 == 0 ? 4 : 1;
      static_assert((RP * CPR) % NT == 0, "output tile must split evenly over the workgroup");
#pragma unroll 1
      for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
        float4 rr[BATCH], gq[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int c = tid + (it0 + u) * NT, rl = c / CPR, ch = c % CPR, ml = g * RP + rl, n = n0 + ch * 4;
          const bool ok = ml < rows_valid && n < p.N;
          rr[u] = ok ? *reinterpret_cast<const float4*>(p.resid + (long)(row0 + ml) * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          gq[u] = ok ? *reinterpret_cast<const float4*>(p.gain + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int c = tid + (it0 + u) * NT, rl = c / CPR, ch = c % CPR, ml = g * RP + rl, n = n0 + ch * 4;
          const bool ok = ml < rows_valid && n < p.N;
          const long m = row0 + (ok ? ml : 0);
          float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) {
            f = *reinterpret_cast<const float4*>(smem + rl * CROW + ((ch ^ (rl & CSWZ)) << 4));
            f.x += rr[u].x; f.y += rr[u].y; f.z += rr[u].z; f.w += rr[u].w;
            *reinterpret_cast<float4*>(Cout + (m * p.ldc + n) * 4) = f;
            *reinterpret_cast<uint2*>(p.C2 + m * p.ldc2 + n) =
                make_uint2(pack_bf16x2(f.x * gq[u].x, f.y * gq[u].y), pack_bf16x2(f.z * gq[u].z, f.w * gq[u].w));
          }
          float ss = ss4_f(f.x, f.y, f.z, f.w);                      // (shared with gemm_bf16_mid_kernel: mode_common.h)
          ss += __shfl_xor(ss, 8, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 1, 64);
          if (ok && (lane & 15) == 0) p.ss_out[m * (p.N >> 6) + (n >> 6)] = ss;
        }
      }
    } else
    for (int c = tid; c < RP * CPR; c += NT) {
      const int rl = c / CPR, ch = c % CPR;
      const int ml = g * RP + rl;
      const int n = n0 + ch * EPC;
      if (ml >= rows_valid || n >= p.N) continue;
      const long m = row0 + ml;
      uint4 v = *reinterpret_cast<const uint4*>(smem + rl * CROW + ((ch ^ (rl & CSWZ)) << 4));
      if constexpr (EPI == MODE_EPI_RESIDUAL && !OUT_BF16) {
        const float4 r = *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n);
        float4 f = *reinterpret_cast<float4*>(&v);
        f.x += r.x; f.y += r.y; f.z += r.z; f.w += r.w;
        v = *reinterpret_cast<uint4*>(&f);
      }
      if constexpr (OUT_BF16) {
        if (n + 8 <= p.N) *reinterpret_cast<uint4*>(Cout + (m * p.ldc + n) * 2) = v;
        else *reinterpret_cast<uint2*>(Cout + (m * p.ldc + n) * 2) = make_uint2(v.x, v.y);   // N % 8 == 4 tail
      } else {
        *reinterpret_cast<uint4*>(Cout + (m * p.ldc + n) * 4) = v;
      }
    }
    if (g + 1 < EPASS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ host side
// geometry ids (also the values of the "gemm_cfg" option; 0 = auto)
// (ids 2, 3, 5, 7, 9-12, 15 were measured-and-rejected geometries of round 1 - deeper rings, 256-wide one-barrier tiles, a persistent 256x256 kernel with
// a serial epilogue - removed once the ping-pong kernel superseded them; the numbers stay in LABNOTES.md)
enum { CFG_AUTO = 0, CFG_128x128_NS2 = 1, CFG_128x64_NS3 = 4, CFG_128x128_NS1 = 6, CFG_128x64_NS2 = 8, CFG_128x128_NS1_4WG = 13, CFG_64x64_NS3 = 14,
       CFG_PP224 = 17, CFG_PP256 = 18, CFG_128x128_NS3 = 20 };
int gemm_bf16_pp_launch(const ModeGemmDesc* d, const GemmParams& p, int rows256, hipStream_t s);   // gemm_bf16_pp.hip: persistent ping-pong kernel, 224 / 256 x 256 tiles
int pp_num_cus();                                                                                   // gemm_bf16_pp.hip: compute units of the current device (= its grid)
int gemm_bf16_skinny_launch(const ModeGemmDesc* d, const GemmParams& p, hipStream_t s);   // gemm_bf16_skinny.hip: weight streamer for a handful of rows
int gemm_bf16_mid_launch(const ModeGemmDesc* d, const GemmParams& p, hipStream_t s);      // gemm_bf16_skinny.hip: register-resident weights, no K loop (a few hundred rows)
int g_gemm_mid_rows_rn = 512;   // "gemm_mid_rows_rn" option: the same kernel for the c_proj (+ residual + first half of ln_2) GEMM, whose [D, D] weight is small enough to be re-read by every 32-row block (rollout batches up to 36 environments)
int g_gemm_mid_rows = 128;   // "gemm_mid_rows" option: ungrouped K = 1024 GEMMs with at most this many rows take gemm_bf16_mid_kernel (0 = off).  Measured chunk latency B = 4: 6.89 -> 6.60 ms, B = 8: 7.40 -> 7.14 ms; from ~200 rows on the M/32 re-reads of W through L2 cost more than the ring kernel (B = 16: 8.41 -> 8.50 ms, B = 32: 9.21 -> 9.48 ms)
int g_gemm_cfg = CFG_AUTO;
int g_gemm_dn_ring3 = 1;   // "gemm_dn_ring3" option: 1 = a K-sliced GEMM whose 128x128 tiles x slices fill ONE round of the part (160 .. 288 workgroups: the expert down-projection of 17 .. 36 environments) takes 128x128 tiles on a 3-slot ring, one workgroup per CU, instead of twice as many 128x64 tiles (a third less L2 -> LDS traffic; 10-step chunk B = 24: 8.29 -> 8.11 ms, B = 32: 8.90 -> 8.73 ms).  Also tried there for the up-projection: 128x256 tiles on a 3-slot ring, one 8-wave workgroup per CU - 8.93 vs 8.90 ms, removed
int g_gemm_skinny_rows = 32;   // measured (scripts/rollout_batch_probe.py): chunk latency B=1 9.4 -> 7.35 ms, B=2 8.7 -> 8.3 ms; from ~3 environments on the tiled kernel (64x64 tiles) is as fast or faster
int g_gemm_pp_min_tiles = 200;   // "gemm_pp_min_tiles" option
int g_gemm_pp_min_tiles_up = 190;   // "gemm_pp_min_tiles_up" option: the threshold for the SwiGLU epilogue (the expert up-projection): 192 tiles (B = 41 .. 48) already pay, 160 do not
int g_gemm_pp = 1;        // "gemm_pp" option: 1 = large problems go to the persistent ping-pong kernel (gemm_bf16_pp.hip), 0 = 128x128 family only
int g_gemm_group_m = 0;   // "gemm_group_m" option: m-tiles per rasterisation group (0 = default 8; >= m_tiles = n-major partition over the XCDs)

template <int BM, int BN, int WM, int WN, int NS, int EPI, bool OUT_BF16, int LR = 0>
static int launch_cfg(GemmParams p, const ModeGemmDesc* d, hipStream_t s) {
  constexpr int NOUT = (EPI == MODE_EPI_SWIGLU) ? BN / 2 : BN;
  constexpr size_t RING = (size_t)NS * (BM + BN) * BK * 2, OUTT = (size_t)(BM / WM) * NOUT * (OUT_BF16 ? 2 : 4);
  constexpr size_t LDS = (RING > OUTT ? RING : OUTT) + (EPI == MODE_EPI_SWIGLU ? BM * 4 : 0);   // whole tile in one pass when it fits the ring, else TM rows per pass; + the inverse row norms of the fused ln_2
  p.n_tiles = (d->N + NOUT - 1) / NOUT;
  p.m_tiles = (d->M + BM - 1) / BM + (d->expert_offsets ? d->num_experts : 0);
  auto kern = gemm_bf16_kernel<BM, BN, WM, WN, NS, EPI, OUT_BF16, LR>;
  static LdsLimitOnce lds_once;
  if (LDS > 64 * 1024) {
    const int rc = lds_once.ensure(reinterpret_cast<const void*>(kern), (int)LDS);
    if (rc != MODE_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(p.m_tiles * p.n_tiles, p.split_k, d->k_group_offsets ? d->num_k_groups : 1), dim3(WM * WN * 64), LDS, s, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

template <int EPI, bool OUT_BF16>
static int launch_epi(const GemmParams& p, const ModeGemmDesc* d, int cfg, hipStream_t s) {
  switch (cfg) {
    case CFG_128x128_NS2: return launch_cfg<128, 128, 2, 2, 2, EPI, OUT_BF16>(p, d, s);
    case CFG_128x64_NS3: return launch_cfg<128, 64, 2, 2, 3, EPI, OUT_BF16>(p, d, s);
    case CFG_128x128_NS1: return launch_cfg<128, 128, 2, 2, 1, EPI, OUT_BF16>(p, d, s);
    case CFG_128x64_NS2: return launch_cfg<128, 64, 2, 2, 2, EPI, OUT_BF16>(p, d, s);
    case CFG_128x128_NS1_4WG: return launch_cfg<128, 128, 2, 2, 1, EPI, OUT_BF16, 4>(p, d, s);   // <= 128 VGPRs: four 32-KiB workgroups per CU
    case CFG_64x64_NS3:
      if constexpr (EPI == MODE_EPI_SWIGLU) return MODE_ERR_UNSUPPORTED; else return launch_cfg<64, 64, 2, 2, 3, EPI, OUT_BF16>(p, d, s);
    case CFG_128x128_NS3:
      if constexpr (EPI == MODE_EPI_NONE) return launch_cfg<128, 128, 2, 2, 3, EPI, OUT_BF16, 1>(p, d, s); else return MODE_ERR_UNSUPPORTED;
    default: return MODE_ERR_BAD_ARG;
  }
}

// Tile-geometry heuristic for 256 CUs, from scripts/gemm_bench.py (--batch 32 / 64 / 128) / gemm_ksweep.py on the config-2 layer shapes:
//   >= 768 tiles of 128x128 : single-buffered 128x128, <= 128 VGPRs, 4 workgroups/CU (72 vs 76 us at B=128, 41 vs 47 us at B=64)   [expert up-projection]
//   >= 576                  : single-buffered 128x128 at 3 workgroups/CU (other workgroups hide the fill latency)
//   >= 256                  : double-buffered 128x128 ring            [QKV at B=128; up-projection at B=32: 448 tiles, 26.6 vs 29.5 us single-buffered]
//   < 256 tiles of 128x64   : 64x64 tiles, 3-slot ring: twice the workgroups, two per CU (c_proj 15.1 -> 13.0 us at B=128, 12.4 -> 9.9 us at B=32)
//   fewer                   : 128x64 tiles so that more CUs get a workgroup, 3-slot ring (two K-tiles in flight per workgroup: with
//                             <= 1 workgroup per CU nothing else hides the fill latency)                 [c_proj, expert down-proj, small batches]
static int pick_cfg(const ModeGemmDesc* d, bool allow_pp) {
  const long rows = d->M;
  // persistent ping-pong kernel (224 x 256 tiles, one workgroup per CU): whenever its tiles cover most of the chip - the expert up-projection from
  // B = 64 on (256 / 512 tiles at B = 64 / 128) and the K-sliced down-projection at B = 128 (16 x 4 tiles x 4 slices = 256).  Unsupported shapes
  // come back from its launcher and fall through to the 128x128 family below.
  if (allow_pp && (!d->expert_offsets || (d->flags & MODE_GEMM_UNIFORM_GROUPS)) &&
      (d->epilogue == MODE_EPI_SWIGLU || d->epilogue == MODE_EPI_NONE || d->epilogue == MODE_EPI_BIAS)) {
    const int nout = d->epilogue == MODE_EPI_SWIGLU ? 128 : 256;
    const long tpp = ((rows + 223) / 224) * (d->N / nout) * (d->split_k > 1 ? d->split_k : 1);
    if (d->N % nout == 0 && tpp >= (d->epilogue == MODE_EPI_SWIGLU ? g_gemm_pp_min_tiles_up : g_gemm_pp_min_tiles)) return CFG_PP224;
  }
  // RAGGED expert segments (device-side offsets, no uniformity promise: the training forward's per-token multinomial routing): 224-row tiles leave half
  // of the experts with a nearly empty fifth tile (measured equal to the ring kernels, scripts/ragged_pp_probe.py), 256-row tiles cover the same rows in
  // four - the up-projection of the training forward is 512 tiles = exactly two rounds: 88.9 -> 73.1 us.  Taken when the expected tile count (segment
  // lengths within 6 % of their mean) fills whole rounds of the part; anything else stays on the ring kernels.
  // (K-slices multiply the tiles: the training forward's down-projection in four slices is 4 experts x 4 row tiles x 4 x 4 = 256 = one tile per CU)
  if (allow_pp && d->expert_offsets && !(d->flags & MODE_GEMM_UNIFORM_GROUPS) && d->num_experts > 0 &&
      (d->epilogue == MODE_EPI_NONE || d->epilogue == MODE_EPI_BIAS) && d->N % 256 == 0) {
    const long per = (long)((double)rows / d->num_experts * 1.06) + 1;
    const long t4 = (long)d->num_experts * ((per + 255) / 256) * (d->N / 256) * (d->split_k > 1 ? d->split_k : 1);
    const long ncu = pp_num_cus();                              // one persistent workgroup per CU: a "round" is one tile on every CU of THIS part
    const long rounds = (t4 + ncu - 1) / ncu;
    if (t4 >= ncu && t4 * 10 >= rounds * ncu * 9) return CFG_PP256;
  }
  const int nout128 = (d->epilogue == MODE_EPI_SWIGLU) ? 64 : 128;
  const long t128 = ((rows + 127) / 128) * ((d->N + nout128 - 1) / nout128);
  if (g_gemm_dn_ring3 && d->split_k > 1 && d->epilogue == MODE_EPI_NONE && t128 * d->split_k >= 160 && t128 * d->split_k <= 256 + 32) return CFG_128x128_NS3;
  const long t64 = ((rows + 127) / 128) * ((d->N + 63) / 64);
  if (d->split_k > 1) {   // split-K (down-projection, dit.hip down_proj_split): slices are extra workgroups
    if (t128 * d->split_k >= 448) return CFG_128x128_NS2;
    return t64 * d->split_k >= 448 ? CFG_128x64_NS3 : CFG_64x64_NS3;      // B=32: 22.1 -> 18.2 us with 64x64 tiles
  }
  if (t128 >= 768) return CFG_128x128_NS1_4WG;   // >= 3 tiles per CU: four low-register workgroups per CU interleave fill / LDS / MFMA phases best
  if (t128 >= 576) return CFG_128x128_NS1;
  if (t128 >= 384) return CFG_128x128_NS2;
  if (t128 >= 256) return d->epilogue == MODE_EPI_SWIGLU ? CFG_128x128_NS2 : CFG_128x64_NS2;   // QKV at B=128 (336 tiles): 20.8 vs 21.7 us
  if (t64 < 256 && d->epilogue != MODE_EPI_SWIGLU) return CFG_64x64_NS3;   // c_proj at B <= 128, QKV at B <= 32: two small workgroups per CU overlap each other
  return CFG_128x64_NS3;
}

int gemm_bf16_launch(const ModeGemmDesc* d, hipStream_t s) {
  if (d->K % BK != 0 || d->K <= 0) return MODE_ERR_UNSUPPORTED;
  if (d->N % 4 != 0 || d->lda % 8 != 0 || d->ldw % 8 != 0 || d->ldc % 4 != 0) return MODE_ERR_UNSUPPORTED;
  if ((d->epilogue == MODE_EPI_BIAS || d->epilogue == MODE_EPI_BIAS_GELU || d->epilogue == MODE_EPI_SWIGLU) && !d->bias)
    return MODE_ERR_BAD_ARG;
  if ((d->epilogue == MODE_EPI_RESIDUAL || d->epilogue == MODE_EPI_RESIDUAL_NORM) && (!d->resid || d->out_dtype != MODE_F32 || d->ldr % 4))
    return MODE_ERR_BAD_ARG;
  const bool small_rows = (d->flags & MODE_GEMM_SMALL_ROWS) != 0;
  if (d->epilogue == MODE_EPI_RESIDUAL_NORM && (!d->C2 || !d->gain || !d->row_ss_out || d->N % (small_rows ? 16 : 64) || d->ldc2 % 4 || d->expert_offsets)) return MODE_ERR_BAD_ARG;
  if (d->row_ss && (d->epilogue != MODE_EPI_SWIGLU || d->row_ss_n <= 0)) return MODE_ERR_BAD_ARG;
  if (d->M <= 0) return MODE_OK;
  GemmParams p;
  p.A = (const uint16_t*)d->A; p.lda = d->lda;
  p.W = (const uint16_t*)d->W; p.ldw = d->ldw; p.w_estride = d->w_expert_stride;
  p.bias = d->bias; p.bias_estride = d->bias_expert_stride;
  p.resid = d->resid; p.ldr = d->ldr; p.C = d->C; p.ldc = d->ldc;
  p.a_rows = d->a_rows; p.offsets = d->expert_offsets; p.E = d->num_experts;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.m_tiles = p.n_tiles = 0;
  p.split_k = d->split_k > 1 ? d->split_k : 1; p.split_stride = d->split_stride;
  if (d->K % (BK * p.split_k) != 0) return MODE_ERR_UNSUPPORTED;
  if (p.split_k > 1 && d->epilogue != MODE_EPI_NONE) return MODE_ERR_UNSUPPORTED;
  // XCD partition of the tile grid (8 private 4-MiB L2s): by default groups of 8 m-tiles x all n-tiles per XCD chunk (every XCD streams its share
  // of A once and ~n_share of W); when A is small next to W (expert up-projection: 3.7 MB of tokens vs 33-67 MB of weights) partition by
  // n-tile instead, so every weight tile is fetched by exactly one XCD and A is the operand that is re-read (measured: 76.7 -> 73.8 us; the
  // down-projection with its 29 MB A operand gets 48 % slower and keeps the default).
  const long a_elems = (long)d->M * d->K, w_elems = (long)(d->epilogue == MODE_EPI_SWIGLU ? 2 : 1) * d->N * d->K * (d->expert_offsets ? d->num_experts : 1);
  p.group_m = g_gemm_group_m > 0 ? g_gemm_group_m : (2 * a_elems <= w_elems ? (1 << 20) : 0);
  p.koffs = d->k_group_offsets; p.c_gstride = d->c_group_stride;
  p.C2 = (uint16_t*)d->C2; p.ldc2 = d->ldc2; p.gain = d->gain; p.ss_out = d->row_ss_out;
  p.ss_in = d->row_ss; p.ss_n = d->row_ss_n; p.ss_eps = d->row_eps;
  p.identity_rows = (d->flags & MODE_GEMM_IDENTITY_ROWS) && d->a_rows && d->expert_offsets ? 1 : 0;
  if (p.koffs && (d->num_k_groups <= 0 || p.split_k > 1 || d->expert_offsets)) return MODE_ERR_BAD_ARG;
  if (small_rows && (g_gemm_cfg != CFG_AUTO || g_gemm_skinny_rows <= 0)) return MODE_ERR_UNSUPPORTED;   // the caller sized its row_ss buffers for the streamer
  if (g_gemm_cfg == CFG_AUTO && (d->M <= g_gemm_skinny_rows || small_rows)) {
    const int rc = gemm_bf16_skinny_launch(d, p, s);
    if (rc != MODE_ERR_UNSUPPORTED || small_rows) return rc;
  }
  if (g_gemm_cfg == CFG_AUTO && !small_rows && (d->M <= g_gemm_mid_rows || (d->epilogue == MODE_EPI_RESIDUAL_NORM && d->M <= g_gemm_mid_rows_rn))) {
    const int rc = gemm_bf16_mid_launch(d, p, s);
    if (rc != MODE_ERR_UNSUPPORTED) return rc;
  }
  int cfg = g_gemm_cfg != CFG_AUTO ? g_gemm_cfg : pick_cfg(d, g_gemm_pp != 0);
  if (cfg == CFG_PP224 || cfg == CFG_PP256) {
    const int rc = gemm_bf16_pp_launch(d, p, cfg == CFG_PP256, s);
    if (rc != MODE_ERR_UNSUPPORTED) return rc;
    cfg = pick_cfg(d, false);                                      // shapes / epilogues the ping-pong kernel does not take
  }
  const bool ob = d->out_dtype == MODE_BF16;
#define MODE_CASE(E) \
  case E: return ob ? launch_epi<E, true>(p, d, cfg, s) : launch_epi<E, false>(p, d, cfg, s);
  switch (d->epilogue) {
    MODE_CASE(MODE_EPI_NONE)
    MODE_CASE(MODE_EPI_BIAS)
    MODE_CASE(MODE_EPI_BIAS_GELU)
    MODE_CASE(MODE_EPI_RESIDUAL)
    MODE_CASE(MODE_EPI_SWIGLU)
    case MODE_EPI_RESIDUAL_NORM: return ob ? MODE_ERR_BAD_ARG : launch_epi<MODE_EPI_RESIDUAL_NORM, false>(p, d, cfg, s);
    default: return MODE_ERR_BAD_ARG;
  }
#undef MODE_CASE
}

}  // namespace mode
