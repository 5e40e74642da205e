// bf16 MFMA GEMMs for the BACKWARD pass on gfx950, fed by the LDS transpose-read `ds_read_b64_tr_b16`:
//
//   data gradient    dX[M,N] = dY[M,K] @ W[K,N]          (MODE_GEMM_W_KN)            W = nn.Linear's [out,in] weight as stored
//   weight gradient  dW[M,N] = dY[K,M]^T @ X[K,N]        (MODE_GEMM_W_KN | A_KM)     both operands row-major activations [rows, features]
//
// The forward kernel (gemm_bf16.hip) wants both operands K-contiguous.  In the backward pass the reduction index is the *row* index of
// at least one row-major operand, i.e. its MFMA fragment (8 consecutive k for one m/n) is strided in memory.  Instead of materialising
// transposed copies in HBM (2 x 1.4 GB of weight shadows per optimizer step + 8 activation transposes per layer, ~4 ms of a 23 ms
// training step on MI355X), the row-major tile [64 k][128 n] is DMA'd into LDS as it lies in memory and the fragment is gathered by
// the CDNA4 transpose read: lane (i = l&15, g = l>>4) supplies the address of 4 contiguous elements of row g*8 + (i>>2) and receives
// column i of the 4x16 block its 16-lane group addressed (semantics pinned on hardware by scripts/probe/tr_probe.hip).  Two reads
// (k..k+3, k+4..k+7) make the 8-element operand of v_mfma_f32_16x16x32_bf16.
//
// LDS image of a [64 k][128 n] tile: 256-byte rows, 32-byte column groups XOR-swizzled by f(k) = (k & 3) | ((k >> 3) & 1) << 2, so the 32
// lanes of one LDS cycle (2 groups x 4 rows x 32 B) cover all 64 banks exactly once.  The image is written lane-linearly by
// `global_load_lds_dwordx4` with the inverse swizzle applied to the per-lane SOURCE address, as in the forward kernel.
//
// Weight-gradient K ranges are arbitrary row ranges (per-expert segments of the sorted dispatch order, no padding): rows past the end of
// a range read a zero row for the A operand and a clamped (finite) row for the W operand, so they contribute exactly 0.
#include "mode_common.h"
#include "gemm_tr_common.h"
#include <type_traits>

namespace mode {

__device__ __attribute__((aligned(256))) uint16_t g_zero_row[128];      // 256 B of zeros: DMA source of masked K rows

// BN = 128 | 64 output columns per workgroup (64: twice the workgroups for problems that would not fill 256 CUs); NS = LDS ring depth
// (2: vmcnt(0) per K-step, 2 workgroups/CU hide each other's fill latency; 3: two tiles in flight under counted waits, for long-K
// problems with <= 1 workgroup per CU).  The gathered-W weight gradient (w_rows) needs NS == 2.
template <bool A_KM, bool OUT_BF16, int BN, int NS, int EPI = 0>
__global__ __launch_bounds__(256, (NS == 1 ? 3 : 2)) void gemm_tr_kernel(const TrParams p) {
  constexpr int BM = 128, BKT = 64, TM = 64, TN = BN / 2, FM = 4, FN = TN / 16;
  constexpr int W_ROW = BN * 2, W_BYTES = BKT * W_ROW;                 // bytes per k-row / per tile of the [k][n] operand
  constexpr int RPP = 1024 / W_ROW, NPW = (BKT / RPP) / 4, CHW = BN / 8; // rows per 1-KiB DMA piece, pieces per wave, 16-B chunks per row
  constexpr int A_BYTES = BM * BKT * 2, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int LOADS = 4 + NPW;
  constexpr int GROUP_M = 8;
  constexpr int ESZ = OUT_BF16 ? 2 : 4;
  constexpr int CROW = BN * ESZ, CPR = CROW / 16, CSWZ = (CPR < 16 ? CPR : 16) - 1;
  static_assert(BM * CROW <= 2 * NS * STAGE_BYTES, "half the output tile must fit the operand ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nblk = p.m_tiles * p.n_tiles;
  const int sb = xcd_remap(blockIdx.x, nblk);
  const int per_group = GROUP_M * p.n_tiles;
  const int grp = sb / per_group, first_m = grp * GROUP_M;
  const int gsz = min(p.m_tiles - first_m, GROUP_M);
  const int rem = sb - grp * per_group;
  int mt = first_m + rem % gsz, nt = rem / gsz;
  if constexpr (EPI == 2) {
    // the fused optimizer epilogue is HBM-bound: workgroups that run side by side take CONSECUTIVE column tiles of one row band, so that together they
    // stream whole parameter rows (a column of tiles shares its low address bits - with m fastest, every concurrently streaming workgroup would sit on
    // the same few memory channels)
    mt = sb / p.n_tiles; nt = sb - mt * p.n_tiles;
  }

  if constexpr (EPI == 1) {
    if (blockIdx.x == 0 && tid == 0 && p.tile_offs) {
      int t = 0;
      p.tile_offs[0] = 0;
      for (int e = 0; e < p.E; ++e) { t += (p.offsets[e + 1] - p.offsets[e] + BM - 1) / BM; p.tile_offs[e + 1] = t; }
    }
  }
  int row0 = 0, row_end = 0, expert = 0;
  if (!A_KM && p.offsets) {
    int t = mt;
    bool found = false;
    for (int e = 0; e < p.E && !found; ++e) {
      const int o0 = p.offsets[e], o1 = p.offsets[e + 1];
      const int nt_e = (o1 - o0 + BM - 1) / BM;
      if (t < nt_e) { row0 = o0 + t * BM; row_end = min(o1, row0 + BM); expert = e; found = true; }
      else t -= nt_e;
    }
    if (!found) return;
  } else {
    row0 = mt * BM; row_end = min(p.M, row0 + BM);
  }
  const int n0 = nt * BN;
  int kb = 0, ke = p.K;
  if (A_KM && p.koffs) { kb = p.koffs[blockIdx.z]; ke = p.koffs[blockIdx.z + 1]; }
  if (!A_KM && p.split_k > 1) { const int ks = p.K / p.split_k; kb = blockIdx.y * ks; ke = kb + ks; }
  const int nk = (ke - kb + BKT - 1) / BKT;
  const uint16_t* W = p.W + (long)expert * p.w_estride;
  // taps (the k x k filter positions of a convolution's weight gradient, one product): this tile's tap selects the index table and the W columns
  const int tap = (A_KM && p.tap_cols > 0) ? n0 / p.tap_cols : 0;
  const int nw0 = n0 - tap * (p.tap_cols > 0 ? p.tap_cols : 0);           // first W column of the tile
  const int nw_end = (A_KM && p.tap_cols > 0) ? p.tap_cols : p.N;
  const int* w_rows = p.w_rows ? p.w_rows + (long)tap * p.tap_stride : nullptr;

  // ---- DMA sources.  [k][cols] tiles: piece P covers RPP tile rows; lane l -> row P*RPP + l / chunks_per_row, physical chunk l % chunks_per_row
  const int kra = lane >> 4, pca = lane & 15;          // [k][128] A tile (weight gradient)
  const int krw = lane / CHW, pcw = lane % CHW;        // [k][BN] W tile
  int kn_col_a[4], kn_col_w[NPW];                      // logical column (elements) of this lane's 16-B chunk, per piece
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = pca ^ (kn_swz<128>((wave * 4 + q) * 4 + kra) << 1);
    kn_col_a[q] = min(row0 + c * 8, p.M - 8);          // (A_KM only) clamped: columns past M are never stored
  }
#pragma unroll
  for (int q = 0; q < NPW; ++q) {
    const int c = pcw ^ (kn_swz<BN>((wave * NPW + q) * RPP + krw) << 1);
    kn_col_w[q] = min(nw0 + c * 8, nw_end - 8);
  }
  // [rows][64 k] A tile of the data gradient (same image as the forward kernel): 8-row pieces, chunk ^ (row & 7)
  const uint16_t* a_src[4];
  if constexpr (!A_KM) {
    const int r8 = lane >> 3, lchunk = (lane & 7) ^ r8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tr = (wave * 4 + q) * 8 + r8;
      const int s = min(row0 + tr, row_end - 1);
      a_src[q] = p.A + (long)s * p.lda + lchunk * 8;
    }
  }

  int widx[NPW];                                       // gathered W rows of the NEXT tile to stage (weight gradient with w_rows, NS == 2)
#pragma unroll
  for (int q = 0; q < NPW; ++q) widx[q] = 0;
  auto load_widx = [&](int kt) {
    if (A_KM && NS <= 2 && w_rows) {
#pragma unroll
      for (int q = 0; q < NPW; ++q) {
        const int r = kb + kt * BKT + (wave * NPW + q) * RPP + krw;
        widx[q] = w_rows[min(r, ke - 1)];
      }
    }
  };
  auto stage = [&](int slot, int kt) {
    char* base = smem + slot * STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int P = wave * 4 + q;
      const uint16_t* src;
      if constexpr (A_KM) {
        const int r = kb + kt * BKT + P * 4 + kra;
        src = r < ke ? p.A + (long)r * p.lda + kn_col_a[q] : g_zero_row;
      } else {
        src = a_src[q] + kb + kt * BKT;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + P * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const int P = wave * NPW + q;
      long r = kb + kt * BKT + P * RPP + krw;
      if constexpr (A_KM) {
        r = min(r, (long)ke - 1);
        if (NS <= 2 && w_rows) r = widx[q];
      }
      const uint16_t* src = W + r * p.ldw + kn_col_w[q];
      if constexpr (A_KM && NS <= 2) {
        if (w_rows) src = r < 0 ? g_zero_row + (lane & 15) * 8 : src;       // a negative index = a zero row (out-of-image filter taps of a convolution)
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + P * 1024), 16, 0, 0);
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
  // transpose-read base addresses of this lane, one per 16-column tile of the wave: row fq*8 + (fr>>2), cols T*16 + (fr&3)*4
  const int fsw_a = (fr >> 2) | ((fq & 1) << 2);
  const int fsw_w = BN == 128 ? fsw_a : (((fr >> 3) & 1) | ((fq & 1) << 1));
  uint32_t tr_a[FM], tr_w[FN];
#pragma unroll
  for (int t = 0; t < FM; ++t)
    tr_a[t] = lds0 + (fq * 8 + (fr >> 2)) * 256 + (fr & 1) * 8 + ((((((wm * 4 + t) ^ fsw_a) << 1) | ((fr >> 1) & 1))) << 4);
#pragma unroll
  for (int t = 0; t < FN; ++t)
    tr_w[t] = lds0 + A_BYTES + (fq * 8 + (fr >> 2)) * W_ROW + (fr & 1) * 8 + ((((((wn * FN + t) ^ fsw_w) << 1) | ((fr >> 1) & 1))) << 4);
  // K-contiguous A tile (data gradient): lane -> row fr of fragment i, 16-B chunk (fq [+4]) ^ (fr & 7)
  const int sw = fr & 7;
  const uint32_t a_off = (wm * TM + fr) * 128;

  constexpr int PRE = (NS == 1) ? 1 : NS - 1;
  if (nk > 0) load_widx(0);
#pragma unroll
  for (int s = 0; s < PRE; ++s)
    if (s < nk) { stage(s, s); load_widx(s + 1); }
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt landed for this wave's pieces; (NS == 3) one younger tile stays in flight across the barrier
    if (NS >= 3 && kt + 1 < nk) tr_wait_vmcnt<LOADS>();
    else tr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if constexpr (NS >= 2) {
      if (kt + NS - 1 < nk) {
        stage((slot + NS - 1) % NS, kt + NS - 1);
        load_widx(kt + NS);
      }
    }
    const uint32_t so = slot * STAGE_BYTES;
    s16x4 alo[2][FM], ahi[2][FM], wlo[2][FN], whi[2][FN];
    bf16x8 fa[2][FM];
    auto read_half = [&](auto KH) {
      constexpr int kh = decltype(KH)::value;
      if constexpr (A_KM) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          lds_tr64<kh * 8192>(alo[kh][i], tr_a[i] + so);
          lds_tr64<kh * 8192 + 1024>(ahi[kh][i], tr_a[i] + so);
        }
      } else {
        const uint32_t ab = lds0 + a_off + so + (((fq + kh * 4) ^ sw) * 16);
        lds_b128<0>(fa[kh][0], ab); lds_b128<2048>(fa[kh][1], ab); lds_b128<4096>(fa[kh][2], ab); lds_b128<6144>(fa[kh][3], ab);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        lds_tr64<kh * 32 * W_ROW>(wlo[kh][j], tr_w[j] + so);
        lds_tr64<kh * 32 * W_ROW + 4 * W_ROW>(whi[kh][j], tr_w[j] + so);
      }
    };
    auto mma_half = [&](auto KH) {
      constexpr int kh = decltype(KH)::value;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        bf16x8 a;
        if constexpr (A_KM) a = join8(alo[kh][i], ahi[kh][i]);
        else a = fa[kh][i];
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(wlo[kh][j], whi[kh][j]), a, acc[i][j], 0, 0, 0);   // swapped: D[n][m]
      }
    };
    read_half(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_half(std::integral_constant<int, 1>{});                   // second half's LDS round trip runs under the first half's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    mma_half(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma_half(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NS == 1) {
      // single buffer (32-48 KiB: three workgroups per CU hide each other's fill latency): everyone must be done reading before the refill
      if (kt + 1 < nk) {
        __builtin_amdgcn_s_barrier();
        stage(0, kt + 1);
        load_widx(kt + 2);
      }
    } else {
      slot = (slot + 1 == NS) ? 0 : slot + 1;
    }
  }

  // ---- epilogue: accumulators -> swizzled LDS tile -> coalesced 16-byte stores; one pass when the tile fits the operand ring, else one
  //      pass per wave-row group (64 rows), so a single-buffered (NS == 1) workgroup keeps its small LDS footprint
  char* Cout = reinterpret_cast<char*>(p.C) + ((long)blockIdx.z * p.c_gstride + (long)blockIdx.y * p.split_stride) * ESZ;
  const int rows_valid = row_end - row0;
  constexpr int EPASS = (BM * CROW <= NS * STAGE_BYTES) ? 1 : 2;
  constexpr int RP = BM / EPASS;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll 1
  for (int g = 0; g < EPASS; ++g) {
    __builtin_amdgcn_s_barrier();
    if (EPASS == 1 || wm == g) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int rl = (EPASS == 1 ? wm * TM : 0) + i * 16 + fr;
        char* crow = smem + rl * CROW;
        const int rsw = rl & CSWZ;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int nl = wn * TN + j * 16 + fq * 4;
          const int bb = nl * ESZ;
          char* dst = crow + ((((bb >> 4) ^ rsw) << 4) | (bb & 15));
          const f32x4 v = acc[i][j];
          if constexpr (OUT_BF16) *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
          else *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int EPC = 16 / ESZ;
    if constexpr (EPI == 1) {
      // ---- dH tile (bf16, in LDS) -> dP = SwishGLU'(P) * dropout(dH), stored as value | gate halves; column sums of the rounded dP for the bias gradient.
      // Thread -> 16-byte chunk ch (8 columns) of rows rbase, rbase + 16, ...: its 8 + 8 running column sums meet their 15 partners (same chunk, other rows)
      // by two xor shuffles inside the wave and one LDS round across the four waves - fixed order, no atomics.
      static_assert(!A_KM && OUT_BF16 && BN == 128 && EPASS == 1, "fused SwishGLU backward: 128-wide bf16 data-gradient tile");
      constexpr int NR = BM / 16;
      const int ch = tid & 15, rbase = tid >> 4;
      const int n = n0 + ch * 8;
      const long ldp = 2L * p.N;
      uint4 pvq[NR], pgq[NR], dhq[NR];
#pragma unroll
      for (int k = 0; k < NR; ++k) {                                   // all operand loads of the thread in flight together (rows past the segment re-read its last row)
        const int rl = rbase + 16 * k;
        const long m = row0 + min(rl, rows_valid - 1);
        pvq[k] = *reinterpret_cast<const uint4*>(p.P + m * ldp + n); pgq[k] = *reinterpret_cast<const uint4*>(p.P + m * ldp + p.N + n);
        dhq[k] = *reinterpret_cast<const uint4*>(smem + rl * CROW + ((ch ^ (rl & CSWZ)) << 4));
      }
      float sv[8], sg[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { sv[j] = 0.f; sg[j] = 0.f; }
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int rl = rbase + 16 * k;
        if (rl >= rows_valid) continue;
        const long m = row0 + rl;
        const uint32_t wv[4] = {pvq[k].x, pvq[k].y, pvq[k].z, pvq[k].w}, wg[4] = {pgq[k].x, pgq[k].y, pgq[k].z, pgq[k].w}, wd[4] = {dhq[k].x, dhq[k].y, dhq[k].z, dhq[k].w};
        uint32_t ov[4], og[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float dv[2], dg[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float v = bf16_bits_to_f32(q ? wv[j] >> 16 : wv[j] & 0xffff), g_ = bf16_bits_to_f32(q ? wg[j] >> 16 : wg[j] & 0xffff);
            float dh = bf16_bits_to_f32(q ? wd[j] >> 16 : wd[j] & 0xffff);
            if (p.thresh) dh = drop_keep(p.seed, (uint64_t)(m * p.N + n + 2 * j + q), p.thresh) ? dh * p.inv_keep : 0.f;
            const float sgm = __builtin_amdgcn_rcpf(1.0f + __expf(-g_));
            dv[q] = dh * g_ * sgm;                                       // d/d value = silu(gate)
            dg[q] = dh * v * sgm * (1.0f + g_ * (1.0f - sgm));           // d/d gate  = value * silu'(gate)
          }
          ov[j] = pack_bf16x2(dv[0], dv[1]); og[j] = pack_bf16x2(dg[0], dg[1]);
          sv[2 * j] += bf16_bits_to_f32(ov[j] & 0xffff); sv[2 * j + 1] += bf16_bits_to_f32(ov[j] >> 16);
          sg[2 * j] += bf16_bits_to_f32(og[j] & 0xffff); sg[2 * j + 1] += bf16_bits_to_f32(og[j] >> 16);
        }
        *reinterpret_cast<uint4*>(p.dP + m * ldp + n) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
        *reinterpret_cast<uint4*>(p.dP + m * ldp + p.N + n) = make_uint4(og[0], og[1], og[2], og[3]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {                                      // lanes 16 / 32 apart hold the same chunk of other rows
        sv[j] += __shfl_xor(sv[j], 16, 64); sv[j] += __shfl_xor(sv[j], 32, 64);
        sg[j] += __shfl_xor(sg[j], 16, 64); sg[j] += __shfl_xor(sg[j], 32, 64);
      }
      __builtin_amdgcn_s_barrier();                                    // every thread has read its dH chunks: the tile's LDS can be reused
      float* red = reinterpret_cast<float*>(smem);                      // [4 waves][16 chunks][16]
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[(wave * 16 + ch) * 16 + j] = sv[j]; red[(wave * 16 + ch) * 16 + 8 + j] = sg[j]; }
      }
      __syncthreads();
      if (tid < 16) {
        float* o = p.bsum + (long)mt * ldp + n;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o[j] = ((red[(0 * 16 + tid) * 16 + j] + red[(1 * 16 + tid) * 16 + j]) + red[(2 * 16 + tid) * 16 + j]) + red[(3 * 16 + tid) * 16 + j];
          o[p.N + j] = ((red[(0 * 16 + tid) * 16 + 8 + j] + red[(1 * 16 + tid) * 16 + 8 + j]) + red[(2 * 16 + tid) * 16 + 8 + j]) + red[(3 * 16 + tid) * 16 + 8 + j];
        }
      }
      return;
    }
    if constexpr (EPI == 2) {
      // ---- fp32 gradient tile (in LDS) -> AdamW in place: p, m, v (+ bf16 shadow, + EMA) of the tile's elements; the gradient itself is never stored.
      // A thread owns 16-byte chunk `ch` of rows rl0, rl0 + 8, ...: a wave covers two whole 512-byte rows per round - 4 full cache lines of each of the
      // three arenas.  Four rounds' worth of loads (12 x 16 B per thread) are in flight before the first use; the accesses are non-temporal (every byte
      // is touched once per step), the shadow store is a normal one (the next forward reads it).  Arithmetic = adamw_kernel's, operation for operation
      // (train_ops.hip): with the same gradient bits the fused and the two-pass update produce the same parameter bits.
      // Two or three workgroups share a CU (32 KiB of LDS, <= 168 registers): while this one streams 416 KiB, the others run their K loops.
      static_assert(A_KM && !OUT_BF16 && BN == 128, "fused AdamW: 128-wide fp32 weight-gradient tile");
      typedef float f4 __attribute__((ext_vector_type(4)));
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      // chunks per thread and pass (two passes of 64 rows: 8) and chunks per batch: four - 12 loads of 16 B per thread in flight; the other wave row's
      // accumulators are still live during the first pass.  (Measured and not kept: the two-slot ring with the whole tile in LDS and all 48 loads of a
      // thread in flight, two workgroups per CU - the epilogue alone is no faster, the fused launch slower, 11.7 vs 11.3 ms per step; batches of two; plain
      // instead of non-temporal accesses - no difference.  profiles/r05_fused_adamw.txt)
      constexpr int NCH = RP * CPR / 256, UB = 4;
      static_assert(NCH % UB == 0, "chunk batches");
      const long gofs = (long)blockIdx.z * p.c_gstride;
      float gs2 = 0.f;
#pragma unroll 1
      for (int u0 = 0; u0 < NCH; u0 += UB) {
        f4 P[UB], M[UB], V[UB], G[UB];
        // element offset of chunk u0 + u of this thread (rows past the tile / columns past N clamp to the tile's first element: every load is
        // unconditional - one round trip for all of them - and recomputed rather than kept: registers)
        auto where = [&](int u, long& eo) -> bool {
          const int c = tid + (u0 + u) * 256;
          const int rl = c / CPR, ch = c % CPR;
          const int ml = g * RP + rl;
          const int n = n0 + ch * EPC;
          const bool ok = ml < rows_valid && n < p.N;
          eo = gofs + (long)(row0 + (ok ? ml : 0)) * p.ldc + (ok ? n : n0);
          return ok;
        };
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          long eo;
          where(u, eo);
          const int c = tid + (u0 + u) * 256;
          const int rl = c / CPR, ch = c % CPR;
          P[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.ad_p + eo));
          M[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.ad_m + eo));
          V[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.ad_v + eo));
          G[u] = *reinterpret_cast<const f4*>(smem + rl * CROW + ((ch ^ (rl & CSWZ)) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);                               // every global load of the batch is issued before the first chunk is touched
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          long eo_;
          if (!where(u, eo_)) continue;
          const long eo[1] = {eo_};
#define MODE_EO eo[0]
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float gr = __fmul_rn(G[u][j], p.ad_gscale);
            gs2 = __builtin_fmaf(gr, gr, gs2);
            float w = P[u][j], m_ = M[u][j], v_ = V[u][j];
            adamw_update_f(w, m_, v_, gr, p.ad_decay, p.ad_b1, p.ad_b2, p.ad_eps, p.ad_step_size, p.ad_inv_bc2_sqrt);
            P[u][j] = w; M[u][j] = m_; V[u][j] = v_;
          }
          __builtin_nontemporal_store(P[u], reinterpret_cast<f4*>(p.ad_p + MODE_EO));
          __builtin_nontemporal_store(M[u], reinterpret_cast<f4*>(p.ad_m + MODE_EO));
          __builtin_nontemporal_store(V[u], reinterpret_cast<f4*>(p.ad_v + MODE_EO));
          if (p.ad_ema) {
            f4 Ev = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.ad_ema + MODE_EO));
#pragma unroll
            for (int j = 0; j < 4; ++j) Ev[j] = Ev[j] - p.ad_ema_rate * (Ev[j] - P[u][j]);
            __builtin_nontemporal_store(Ev, reinterpret_cast<f4*>(p.ad_ema + MODE_EO));
          }
          if (p.ad_lp) {
            u2 o; o[0] = pack_bf16x2(P[u][0], P[u][1]); o[1] = pack_bf16x2(P[u][2], P[u][3]);
            *reinterpret_cast<u2*>(p.ad_lp + MODE_EO) = o;
          }
#undef MODE_EO
        }
      }
      if (p.ad_gsq) {                                                  // ||g||^2 of this tile: wave butterfly, then the four waves through LDS (fixed order)
        gs2 = wave_sum(gs2);
        __builtin_amdgcn_s_barrier();                                  // every thread has read its gradient chunks of this pass
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) red[wave] = gs2;
        __syncthreads();
        if (tid == 0) {
          float* o = p.ad_gsq + ((long)blockIdx.z * gridDim.x + blockIdx.x);
          const float t = (red[0] + red[1]) + (red[2] + red[3]);
          *o = (g == 0 ? 0.f : *o) + t;                                // pass 1 adds to what pass 0 of the same workgroup stored
        }
      }
      continue;
    }
    for (int c = tid; c < RP * CPR; c += 256) {
      const int rl = c / CPR, ch = c % CPR;
      const int ml = g * RP + rl;
      const int n = n0 + ch * EPC;
      if (ml >= rows_valid || n >= p.N) continue;
      const long m = row0 + ml;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + rl * CROW + ((ch ^ (rl & CSWZ)) << 4));
      if constexpr (OUT_BF16) *reinterpret_cast<uint4*>(Cout + (m * p.ldc + n) * 2) = v;
      else *reinterpret_cast<uint4*>(Cout + (m * p.ldc + n) * 4) = v;
    }
  }
}

template <bool KM, bool OB, int BN, int NS, int EPI = 0>
static int tr_launch(TrParams p, const ModeGemmDesc* d, hipStream_t s) {
  p.n_tiles = (d->N + BN - 1) / BN;
  p.m_tiles = (d->M + 127) / 128 + (d->expert_offsets ? d->num_experts : 0);
  const dim3 grid(p.m_tiles * p.n_tiles, p.split_k, (KM && d->k_group_offsets) ? d->num_k_groups : 1);
  constexpr size_t lds = (size_t)NS * (128 * 64 * 2 + 64 * BN * 2);       // the output tile goes through the ring (one or two passes)
  auto kern = gemm_tr_kernel<KM, OB, BN, NS, EPI>;
  static LdsLimitOnce lds_once;
  {
    const int rc = lds_once.ensure(reinterpret_cast<const void*>(kern), (int)lds);
    if (rc != MODE_OK) return rc;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

// The down-projection's data gradient fused with the SwishGLU backward (dit_train.hip): dP[M, 2N] and per-m-tile bias-gradient partial sums from
// dY[M, K], W2 (grouped by expert), the stashed pre-activations P.  dH itself is never written.  128 x 128 ring tile (two workgroups per CU: one's
// epilogue - 128 KiB of P / dP traffic per tile - runs under the other's K loop).  m-tile t of the grouped tile space = row t of `bsum`.
int gemm_bf16_tr_swiglu_bwd_launch(const ModeGemmDesc* d, const void* P, void* dP, uint32_t seed, uint32_t thresh, float inv_keep, float* bsum, int* tile_offs,
                                   hipStream_t s) {
  if (!(d->flags & MODE_GEMM_W_KN) || (d->flags & MODE_GEMM_A_KM) || d->dtype != MODE_BF16 || d->epilogue != MODE_EPI_NONE || d->a_rows || d->w_rows) return MODE_ERR_UNSUPPORTED;
  if (d->N % 128 || d->K % 64 || d->K <= 0 || d->split_k > 1 || d->k_group_offsets || d->lda % 8 || d->ldw % 8 || !P || !dP || !bsum) return MODE_ERR_UNSUPPORTED;
  if ((((uintptr_t)P | (uintptr_t)dP | (uintptr_t)bsum) & 15)) return MODE_ERR_UNSUPPORTED;
  if (d->M <= 0) return MODE_OK;
  TrParams p;
  p.A = (const uint16_t*)d->A; p.lda = d->lda; p.W = (const uint16_t*)d->W; p.ldw = d->ldw; p.w_estride = d->w_expert_stride;
  p.C = nullptr; p.ldc = 0; p.offsets = d->expert_offsets; p.E = d->num_experts;
  p.koffs = nullptr; p.c_gstride = 0; p.w_rows = nullptr; p.tap_cols = 0; p.tap_stride = 0;
  p.M = d->M; p.N = d->N; p.K = d->K; p.split_k = 1; p.split_stride = 0; p.m_tiles = p.n_tiles = 0;
  p.P = (const uint16_t*)P; p.dP = (uint16_t*)dP; p.seed = seed; p.thresh = thresh; p.inv_keep = inv_keep; p.bsum = bsum; p.tile_offs = tile_offs;
  p.ad_p = p.ad_m = p.ad_v = p.ad_ema = p.ad_gsq = nullptr; p.ad_lp = nullptr;
  if (!d->expert_offsets) return MODE_ERR_UNSUPPORTED;
  return tr_launch<false, true, 128, 2, 1>(p, d, s);
}

// Weight gradient dW[M, N] (per K-group z: + z * c_group_stride) = A^T W whose epilogue applies AdamW to the parameter tile the C pointer locates
// (ModeAdamWFuse, include/mode_hip.h).  Single-buffered 128 x 128 ring tile, three workgroups per CU: the 416 KiB of p / m / v / shadow traffic of one
// workgroup's epilogue run under the other workgroups' K loops - these launches are HBM-bound (26 B per parameter against ~7 us of MFMA work per tile).
// The scalars are derived exactly as mode_adamw_step derives them (train_ops.hip), so both paths round alike.
// Round-6 probe (scripts/probe/gemm_bf16_trws.hip, linked only by scripts/probe/build_trws_variant.sh): the same launch as ONE persistent wave-specialised
// workgroup per CU.  Measured and not shipped (profiles/r06_fused_adamw_ws.txt: 190 vs 195 us isolated, 11.12 vs 10.52 ms in the step) - a weak symbol, null here.
__attribute__((weak)) int gemm_bf16_trws_adamw_launch(TrParams p, const ModeGemmDesc* d, long gsq_slots, hipStream_t s);
static int tr_adamw_launch(const ModeGemmDesc* d, hipStream_t s) {
  const ModeAdamWFuse* z = d->adamw;
  if (!(d->flags & MODE_GEMM_A_KM) || d->out_dtype != MODE_F32 || d->split_k > 1 || d->w_tap_cols > 0 || d->N % 128 || d->ldc % 4) return MODE_ERR_UNSUPPORTED;
  if (!z->grad_base || !z->param_base || !z->exp_avg_base || !z->exp_avg_sq_base || !d->C || z->step < 1) return MODE_ERR_BAD_ARG;
  const long ofs = reinterpret_cast<const float*>(d->C) - z->grad_base;
  if (ofs < 0 || (ofs & 3) || (d->k_group_offsets && (d->c_group_stride & 3))) return MODE_ERR_BAD_ARG;
  if (((uintptr_t)z->param_base | (uintptr_t)z->exp_avg_base | (uintptr_t)z->exp_avg_sq_base | (uintptr_t)z->ema_base) & 15 || ((uintptr_t)z->lp_base & 7)) return MODE_ERR_BAD_ARG;
  TrParams p;
  p.A = (const uint16_t*)d->A; p.lda = d->lda; p.W = (const uint16_t*)d->W; p.ldw = d->ldw; p.w_estride = d->w_expert_stride;
  p.C = nullptr; p.ldc = d->ldc; p.offsets = nullptr; p.E = 0;
  p.koffs = d->k_group_offsets; p.c_gstride = d->c_group_stride; p.w_rows = d->w_rows; p.tap_cols = 0; p.tap_stride = 0;
  p.M = d->M; p.N = d->N; p.K = d->K; p.split_k = 1; p.split_stride = 0; p.m_tiles = p.n_tiles = 0;
  p.P = nullptr; p.dP = nullptr; p.seed = p.thresh = 0; p.inv_keep = 1.f; p.bsum = nullptr; p.tile_offs = nullptr;
  const double bc1 = 1.0 - pow((double)z->beta1, z->step), bc2 = 1.0 - pow((double)z->beta2, z->step);
  p.ad_p = z->param_base + ofs; p.ad_m = z->exp_avg_base + ofs; p.ad_v = z->exp_avg_sq_base + ofs;
  p.ad_lp = z->lp_base ? z->lp_base + ofs : nullptr; p.ad_ema = z->ema_base ? z->ema_base + ofs : nullptr; p.ad_ema_rate = z->ema_rate;
  p.ad_decay = 1.f - z->lr * z->weight_decay; p.ad_b1 = z->beta1; p.ad_b2 = z->beta2; p.ad_eps = z->eps;
  p.ad_step_size = (float)(z->lr / bc1); p.ad_inv_bc2_sqrt = (float)(1.0 / sqrt(bc2)); p.ad_gscale = z->grad_scale;
  const long groups = d->k_group_offsets ? d->num_k_groups : 1;
  const long wgs = groups * ((d->M + 127) / 128) * (d->N / 128);
  if (z->gsq && z->gsq_capacity < wgs) return MODE_ERR_WORKSPACE;
  p.ad_gsq = z->gsq;
  if (gemm_bf16_trws_adamw_launch) {                         // only in the probe build (see the declaration): not part of the shipped library
    const int rc = gemm_bf16_trws_adamw_launch(p, d, z->gsq ? z->gsq_capacity : 0, s);
    if (rc != MODE_ERR_UNSUPPORTED) return rc;
  }
  return tr_launch<true, false, 128, 1, 2>(p, d, s);
}

int g_tr_cfg = 0;   // "gemm_tr_cfg" option: 0 auto, 1 = 128-wide NS2, 2 = 64-wide NS3, 3 = 128-wide NS3, 4 = 64-wide NS2, 5 = 128-wide NS1,
                    // 6 = the persistent ping-pong kernel (gemm_bf16_pptr.hip) for every shape it takes, 7 = auto without it
int g_bwd_coexec = 0;   // "bwd_coexec" option: 1 = other kernels (an overlapped optimizer pass, collectives) share the CUs with the backward chain - the
                        // auto choice then keeps the ring kernels, whose small workgroups leave CU resources free (the persistent kernel owns a CU's whole
                        // register file: measured at C2 / B = 128 with the per-block AdamW overlap, 12.8 vs 12.3 ms per step)
int gemm_bf16_pptr_launch(const ModeGemmDesc* d, bool force, hipStream_t s);   // gemm_bf16_pptr.hip: 256 x 256 ping-pong tiles, large problems

int gemm_bf16_tr_launch(const ModeGemmDesc* d, hipStream_t s) {
  const bool a_km = (d->flags & MODE_GEMM_A_KM) != 0;
  if (!(d->flags & MODE_GEMM_W_KN)) return MODE_ERR_BAD_ARG;
  if (d->dtype != MODE_BF16 || d->epilogue != MODE_EPI_NONE || d->a_rows) return MODE_ERR_UNSUPPORTED;
  const int split = d->split_k > 1 ? d->split_k : 1;
  if (split > 1 && (a_km || d->K % (64 * split) != 0)) return MODE_ERR_UNSUPPORTED;      // K-slices: data gradient only
  if (d->N % 8 != 0 || d->N < 8 || d->lda % 8 != 0 || d->ldw % 8 != 0 || d->ldc % (d->out_dtype == MODE_BF16 ? 8 : 4) != 0) return MODE_ERR_UNSUPPORTED;
  if (a_km) {
    if (d->M % 8 != 0 || d->M < 8 || d->expert_offsets) return MODE_ERR_UNSUPPORTED;
    if (d->k_group_offsets && d->num_k_groups <= 0) return MODE_ERR_BAD_ARG;
  } else {
    if (d->K % 64 != 0 || d->K <= 0 || d->k_group_offsets || d->w_rows) return MODE_ERR_UNSUPPORTED;
  }
  if (d->M <= 0) return MODE_OK;
  if (d->adamw) return tr_adamw_launch(d, s);                  // weight gradient with the optimizer in its epilogue (ModeAdamWFuse)
  if ((g_tr_cfg == 0 && !g_bwd_coexec) || g_tr_cfg == 6) {     // large problems: the ping-pong structure (MODE_ERR_UNSUPPORTED = not its shape)
    const int rc = gemm_bf16_pptr_launch(d, g_tr_cfg == 6, s);
    if (rc != MODE_ERR_UNSUPPORTED) return rc;
  }
  TrParams p;
  p.A = (const uint16_t*)d->A; p.lda = d->lda; p.W = (const uint16_t*)d->W; p.ldw = d->ldw; p.w_estride = d->w_expert_stride;
  p.C = d->C; p.ldc = d->ldc; p.offsets = d->expert_offsets; p.E = d->num_experts;
  p.koffs = d->k_group_offsets; p.c_gstride = d->c_group_stride; p.w_rows = d->w_rows;
  p.tap_cols = 0; p.tap_stride = 0;
  if (d->w_tap_cols > 0) {                                      // w_rows in taps: a tile must lie inside one tap
    if (!a_km || !d->w_rows || d->w_tap_cols % 64 || d->N % d->w_tap_cols) return MODE_ERR_UNSUPPORTED;
    p.tap_cols = d->w_tap_cols; p.tap_stride = d->w_rows_tap_stride;
  }
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.split_k = split; p.split_stride = d->split_stride;
  p.m_tiles = p.n_tiles = 0;
  p.P = nullptr; p.dP = nullptr; p.seed = p.thresh = 0; p.inv_keep = 1.f; p.bsum = nullptr; p.tile_offs = nullptr;
  p.ad_p = p.ad_m = p.ad_v = p.ad_ema = p.ad_gsq = nullptr; p.ad_lp = nullptr;
  // geometry: enough 128x128 workgroups to put two on every CU -> NS2 ring (they hide each other's fill latency); otherwise 128x64
  // tiles (twice the workgroups) with a 3-slot ring so one workgroup keeps two tiles in flight
  const long groups = (a_km && d->k_group_offsets) ? d->num_k_groups : 1;
  const long t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128) * groups;
  int cfg = g_tr_cfg >= 6 ? 0 : g_tr_cfg;
  // measured (profiles/): weight gradients (short K per tile, >= 3 workgroups per CU) are 4 % faster single-buffered; data gradients are not
  if (cfg == 0) cfg = (a_km && t128 >= 768) ? 5 : ((t128 * split >= 448 || d->w_rows) ? 1 : 2);      // K-slices count as workgroups
  if (d->w_rows && (cfg == 2 || cfg == 3)) cfg = 1;
  if (p.tap_cols > 0 && p.tap_cols % 128 && (cfg == 1 || cfg == 3 || cfg == 5)) cfg = 4;      // 64-column taps: 64-wide tiles (two-slot ring: gathered rows)
  const bool ob = d->out_dtype == MODE_BF16;
#define MODE_TR_CFG(BN, NS)                                                                     \
  {                                                                                             \
    if (a_km) return ob ? tr_launch<true, true, BN, NS>(p, d, s) : tr_launch<true, false, BN, NS>(p, d, s);   \
    return ob ? tr_launch<false, true, BN, NS>(p, d, s) : tr_launch<false, false, BN, NS>(p, d, s);           \
  }
  switch (cfg) {
    case 1: MODE_TR_CFG(128, 2)
    case 2: MODE_TR_CFG(64, 3)
    case 3: MODE_TR_CFG(128, 3)
    case 4: MODE_TR_CFG(64, 2)
    case 5: MODE_TR_CFG(128, 1)
    default: return MODE_ERR_BAD_ARG;
  }
#undef MODE_TR_CFG
}

}  // namespace mode
