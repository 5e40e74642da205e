// Persistent "ping-pong" bf16 MFMA GEMM for gfx950, large-M forward shapes:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T), plain / row-gathered /
// grouped (MoE) / K-sliced.  This is the kernel of the two expert projections at B = 128 (85 % of the denoiser's FLOPs).
//
// Why a second tiled kernel: the 128x128 one-barrier-per-K-step loop of gemm_bf16.hip tops out at ~840 TF/s (33 % of the bf16 MFMA peak): a
// workgroup's fill / LDS-read / MFMA phases serialise and a 128x128x64 step moves 32 KiB through the vector-memory path per 512 MFMA cycles.
// This kernel changes the structure, not the tuning:
//   * (128 + 32*FM1) x 256 output tile; shipped: FM1 = 3 = 224 rows (3584 sorted rows = 16 x 224 gives 512 tiles = exactly two per CU),
//     BK = 64, 8 wave64 as 2 (M) x 4 (N): half the L2->LDS bytes per flop of the 128x128 tile.
//   * The operand tile of a K-step lives in LDS as four 16-KiB HALF-tiles (A rows 0-127 / 128-223, W rows 0-127 / 128-255), two K-steps
//     resident (128 KiB).  A wave's output is the 2 x 2 grid of quadrants {A half} x {W half}; one K-step = two PHASES: a phase = one A half x
//     BOTH W halves (32 / 24 MFMAs, v_mfma_f32_16x16x32_bf16): {ds_read the fragments the phase needs | issue the global_load_lds of half-tiles
//     one to two K-steps ahead} -> s_barrier -> MFMAs -> s_barrier.  A half-tile is re-filled one whole phase after its last fragment read.
//   * Counted waits only: every read section ends in ONE `s_waitcnt vmcnt(8 | 6)` that leaves the newest K-step's worth of DMA in flight, never
//     vmcnt(0); the loop body is straight-line code (first K-step pair peeled, wave role a template argument: see the comments at the loop) -
//     tests/test_boundary.py::test_pp_kernel_isa_contract pins "no scratch access, no branch but the back edge" on the compiled ISA.
//   * The two wave rows run STAGGERED by one barrier: while waves 0-3 issue their MFMA cluster, waves 4-7 (their SIMD partners) issue fragment
//     reads and DMA, and vice versa - the matrix pipe of every SIMD always has one wave feeding it (s_setprio around the MFMA cluster).
//   * PERSISTENT: one workgroup per CU walks its output tiles (consecutive n-tiles of one m-tile, so the gathered A rows and their per-lane
//     DMA sources never change) and the operand stream never stops: while the last K-steps of a tile run, the first two K-steps of the NEXT
//     tile are already landing.  K = 1024 is only 16 K-steps, so the first-fill latency of a tile would otherwise cost ~20 % of it.
//   * Weight rows are assigned to LDS rows through a permutation (free: the DMA source address is per lane) such that, with the swapped
//     MFMA operands, a lane ends up with EIGHT consecutive output columns of one token row: bias / SwiGLU run in registers and the tile is
//     stored straight from registers, 16 bytes per lane - no LDS round trip, no barrier in the epilogue.
//   * SwiGLU: W half 0 = the 128 "value" rows, W half 1 = the matching 128 "gate" rows (rows n and N+n, the reference's tensor_split(2),
//     modedit.py:89) - quadrant (a, 0) and (a, 1) of a wave hold value and gate of the same outputs.
//   * XCD-aware tile order: an XCD's 32 workgroups cover 8 m-tiles x 8 n-tiles, so the 4-MiB L2 sees each operand K-slice once.
//
// Numerics: every output element is the same k-ordered fp32 MFMA accumulation chain and the same epilogue expression as gemm_bf16.hip, so
// results are BIT-IDENTICAL to the 128x128 kernels (the batch-slice consistency tests of the sampler rely on it).
#include "mode_common.h"
#include <type_traits>

// Round-5 experiment (VERDICT r04 "next" #1a), OFF in the shipped library: an L2-only run-ahead of the operand stream.  -DPP_L2_TOUCH=d makes waves 6 and 7
// (224-row tile: the two waves with the lighter DMA load) issue ONE dword LDS-DMA load per K-step into a landing strip nobody reads, touching the cache
// lines of K-step kt + 2 + d - wave 6 lines of the W tile, wave 7 lines of the A tile - so that the operand DMA of that K-step, d steps later, finds them in
// the XCD's L2.  -DPP_L2_TOUCH_SPLIT=1 divides the lines among the workgroups of the XCD that share them (W: the 8 m-tiles of a band touch 32 lines each, A:
// the 4 n-runs 56 rows each) so that together they cover every line once; without it a workgroup touches every fourth line of its own tiles.  The touch
// is issued right behind a counted wait, so it rides INSIDE the vmcnt budget (the waits of those two waves allow one more instruction in flight).
// (First form, every wave touching 64 lines per K-step: 576-585 instead of 628-631 denoise-steps/s - the vector-memory path is the loop's bottleneck.)  scripts/build_pp_variant.sh builds libmode_hip_<tag>.so; scripts/pp_l2touch_probe.py times it against the shipped library in one
// process.  Result and reading: profiles/r05_pp_l2touch.txt, LABNOTES.md.
#ifndef PP_L2_TOUCH
#define PP_L2_TOUCH 0
#endif
#ifndef PP_L2_TOUCH_SPLIT
#define PP_L2_TOUCH_SPLIT 0
#endif

namespace mode {

namespace pp {
constexpr int BKK = 64;
constexpr int HALF_BYTES = 128 * BKK * 2;                  // one 128-row half-tile: 16 KiB
constexpr int LDS_A = 0;                                   // A[t][h] at (t*2+h) * 16 KiB
constexpr int LDS_B = 4 * HALF_BYTES;                      // W[t][h] at 64 KiB + (t*2+h) * 16 KiB
constexpr int LDS_BIAS = 8 * HALF_BYTES;                   // 8 x 1 KiB: one bias slot per wave (each wave DMAs and reads its own copy)
constexpr int LDS_NRM = LDS_BIAS + 8 * 1024;               // 2 x 256 floats: inverse row norms of the fused ln_2 (double-buffered per restart)
constexpr int LDS_SS = LDS_NRM + 2 * 1024;                 // 256 rows x 64 B: per-64-column partial sums of squares of the tile's rows (DMA'd at a restart)
#if PP_L2_TOUCH
constexpr int LDS_PF = LDS_SS + 256 * 64;                  // 8 x 256 B: landing strip of the L2 touches (never read)
constexpr int LDS_TOTAL = LDS_PF + 8 * 256;
#else
constexpr int LDS_TOTAL = LDS_SS + 256 * 64;               // 154 KiB of the CU's 160
#endif
constexpr int GM = 8;                                      // m-tiles per rasterisation band
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ void lds_read128(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int BASE, int STRIDE, int CNT, int I = 0>
__device__ __forceinline__ void lds_read_seq(bf16x8* dst, uint32_t addr) {
  if constexpr (I < CNT) {
    lds_read128<BASE + I * STRIDE>(dst[I], addr);
    lds_read_seq<BASE, STRIDE, CNT, I + 1>(dst, addr);
  }
}
__device__ __forceinline__ void lds_read_f4(float4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_read_f1(float& dst, uint32_t addr) {
  asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_write_u4(uint32_t addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read_u4(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
template <int V>
using IC = std::integral_constant<int, V>;
}  // namespace pp

template <int EPI, bool OUT_BF16, int FM1>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmParams p) {
  using namespace pp;
  constexpr int BM = 128 + 32 * FM1;                         // rows of an output tile: half 0 = 2 x 64, half 1 = 2 x 16*FM1
  constexpr bool SWI = EPI == MODE_EPI_SWIGLU;
  constexpr bool HAS_BIAS = EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU || SWI;
  constexpr int NOUT = SWI ? 128 : 256;                      // output columns of a tile
  constexpr int ESZ = OUT_BF16 ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---------------------------------------------------------------------------------------------------- tile space (all scalar)
  // grouped (MoE): rows are sorted by expert; expert e owns sorted rows [o[e], o[e+1]) and ceil(count / BM) m-tiles (device-side offsets).
  int o[9];
  int m_real;
  if (p.offsets) {
#pragma unroll
    for (int e = 0; e < 9; ++e) o[e] = p.offsets[min(e, p.E)];
    m_real = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (e < p.E) m_real += (o[e + 1] - o[e] + BM - 1) / BM;
  } else {
    m_real = (p.M + BM - 1) / BM;
  }
  const int S = p.split_k, n_tiles = p.n_tiles;
  const int T = m_real * n_tiles * S;
  const int G = gridDim.x;
  const int R = (T + G - 1) / G;                               // tiles per workgroup
  if (R == 0) return;
  const int RN = (n_tiles % R == 0) ? R : 1;                   // n-run: a workgroup's consecutive tiles share the m-tile when R | n_tiles
  const int nwg = (T + R - 1) / R;                             // workgroups that get work
  if ((int)blockIdx.x >= nwg) return;
  const int wg = xcd_remap(blockIdx.x, nwg);
  int L = wg * R;
  const int Lend = min(T, L + R);
  const int nk = p.K / BKK / S;                                // K-steps per slice (even, >= 2: checked by the launcher)

  struct Tile { int m, n, slice, row0, row_end, expert, seg0; };
  auto map_tile = [&](int l, Tile& t) {
    const int per_band = GM * n_tiles * S;
    const int band = l / per_band, first_m = band * GM;
    const int gsz = min(GM, m_real - first_m);
    const int rem = l - band * per_band;
    const int per_slice = gsz * n_tiles;
    t.slice = rem / per_slice;
    const int q = rem - t.slice * per_slice;
    const int run = gsz * RN;
    const int n_hi = q / run, r2 = q - n_hi * run;
    t.m = first_m + r2 / RN;
    t.n = n_hi * RN + r2 % RN;
    t.expert = 0; t.seg0 = 0;
    if (p.offsets) {
      int tt = t.m;
      bool found = false;
      t.row0 = 0; t.row_end = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (!found && e < p.E) {
          const int nt_e = (o[e + 1] - o[e] + BM - 1) / BM;
          if (tt < nt_e) { t.row0 = o[e] + tt * BM; t.row_end = min(o[e + 1], t.row0 + BM); t.expert = e; t.seg0 = o[e]; found = true; }
          else tt -= nt_e;
        }
      }
    } else {
      t.row0 = t.m * BM; t.row_end = min(p.M, t.row0 + BM);
    }
  };
  // byte address of a tile's first weight row at K-offset 0 of its slice; half 1 of the W tile sits `w_half` bytes further
  auto w_tile_base = [&](const Tile& t) -> const char* {
    return reinterpret_cast<const char*>(p.W) + ((long)t.expert * p.w_estride + (long)t.n * NOUT * p.ldw + (long)t.slice * nk * BKK) * 2;
  };
  const long w_half = (long)(SWI ? p.N : 128) * p.ldw * 2;

  // ---------------------------------------------------------------------------------------------------- per-lane constants
  // DMA: one global_load_lds_dwordx4 fills a 1-KiB piece = 8 rows x 128 B; lane i -> row i>>3, physical 16-B chunk i&7 which holds the
  // LOGICAL chunk (i&7)^(i>>3) (XOR swizzle on the source address; linear destination; same XOR on the fragment reads).  A half-tile is
  // 16 pieces: wave w fills pieces 2w and 2w+1.
  const int r8 = lane >> 3, lchunk = (lane & 7) ^ r8;
  uint32_t b_off[2];                                           // byte offsets from a W half-tile base (tile independent)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int lrow = (wave * 2 + q) * 8 + r8;                  // LDS row of the half-tile, 0..127
    // LDS row blk*32 + j*16 + q4*4 + r  holds weight row  blk*32 + q4*8 + j*4 + r : a lane's 2 fragments x 4 accumulator rows = 8 consecutive columns
    const int blk = lrow >> 5, rho = lrow & 31;
    const int col = blk * 32 + ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);
    b_off[q] = (uint32_t)(((long)col * p.ldw + lchunk * 8) * 2);
  }
  // fragment reads: lane -> row (l&15) of a 16-row fragment, 16-B chunk (l>>4) [+4 for the second k32 half], chunk XOR (row & 7)
  const int fr = lane & 15, fq = lane >> 4;
  const int c0 = (fq ^ (fr & 7)) * 16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t a_addr[2][2], b_addr[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int c = kh ? (c0 ^ 64) : c0;
    a_addr[0][kh] = lds0 + LDS_A + (wr * 64 + fr) * 128 + c;
    a_addr[1][kh] = lds0 + LDS_A + (wr * 16 * FM1 + fr) * 128 + c;
    b_addr[kh] = lds0 + LDS_B + (wc * 32 + fr) * 128 + c;
  }

  bf16x8 A_[8], Bf[2][4];                                      // A fragments [k-half*4 + i], W fragments [half][k-half*2 + j]
  f32x4 acc[2][2][4][2];                                       // [A half][W half][i][j]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bool fresh = true;
  // Everything the K loop calls is resolved at compile time (no run-time switches: the shipped loop is straight-line code between its barriers;
  // the timing ablations and cycle stamps that measured it in round 2 are gone from the product library - the numbers stay in LABNOTES.md).
  auto rdA = [&](auto T_, auto H_) __attribute__((always_inline)) {
    constexpr int t = decltype(T_)::value, h = decltype(H_)::value;
    constexpr int base = (t * 2 + h) * HALF_BYTES, nf = h ? FM1 : 4;
    lds_read_seq<base, 2048, nf>(&A_[0], a_addr[h][0]);
    lds_read_seq<base, 2048, nf>(&A_[4], a_addr[h][1]);
  };
  auto rdB = [&](auto T_, auto H_) __attribute__((always_inline)) {
    constexpr int t = decltype(T_)::value, h = decltype(H_)::value;
    constexpr int base = (t * 2 + h) * HALF_BYTES;
    lds_read_seq<base, 2048, 2>(&Bf[h][0], b_addr[0]);
    lds_read_seq<base, 2048, 2>(&Bf[h][2], b_addr[1]);
  };
  auto mma = [&](auto AH_, auto BH_) __attribute__((always_inline)) {
    constexpr int ah = decltype(AH_)::value, bh = decltype(BH_)::value, nf = ah ? FM1 : 4;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < nf; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)                              // swapped operands: D[weight row][token]
          acc[ah][bh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf[bh][kh * 2 + j], A_[kh * 4 + i], acc[ah][bh][i][j], 0, 0, 0);
  };
  auto stage = [&](auto OP_, auto T_, auto H_, const char* g, uint32_t o0, uint32_t o1) __attribute__((always_inline)) {
    constexpr int op = decltype(OP_)::value, t = decltype(T_)::value, h = decltype(H_)::value;
    constexpr int base = (op ? LDS_B : LDS_A) + (t * 2 + h) * HALF_BYTES;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + o0),
                                     (__attribute__((address_space(3))) void*)(smem + base + (wave * 2 + 0) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + o1),
                                     (__attribute__((address_space(3))) void*)(smem + base + (wave * 2 + 1) * 1024), 16, 0, 0);
  };
  // A half 1 of the 224-row tile has 96 rows = 12 pieces: waves 6 and 7 own pieces 12-15 and stage nothing (6 % fewer DMA bytes; the loop is
  // DMA-rate bound).  Their counted waits differ (6 instead of 8 instructions per K-step): the K loop exists in two compile-time copies, one per
  // wave role, selected ONCE per output tile - not by a branch in front of every stage / wait.
  const bool stage_a1 = FM1 == 4 || wave < 6;
  constexpr IC<0> _0{};
  constexpr IC<1> _1{};
#define PP_SB() __builtin_amdgcn_sched_barrier(0)
#define PP_BAR() __builtin_amdgcn_s_barrier()
  bool staggered = false;                                      // true while wave row 1 runs one barrier behind wave row 0

  Tile cur, nxt;
  map_tile(L, cur);
#if PP_L2_TOUCH
  // Only the two waves that stage no A half 1 (waves 6 and 7 of the 224-row tile: two DMA instructions per K-step fewer than the others) issue touches:
  // wave 6 the W lines, wave 7 the A lines - ONE extra vector-memory instruction per K-step on each (the all-waves form cost 5-8 % in the chain: the
  // vector-memory path is what the loop is bound by).  The touch is an LDS-DMA dword load into a landing strip nobody reads: a VGPR destination
  // would be written whenever the data arrives, long after the compiler considers the instruction complete.
  uint32_t pf_off = 0;                                           // this lane's line: byte offset from the tile's W base (wave 6) / from Ak (wave 7)
  auto pf_set = [&](const Tile& t) __attribute__((always_inline)) {
    if (wave == 6) {                                             // W: logical row jj of the 256-row tile (half jj >> 7)
#if PP_L2_TOUCH_SPLIT
      const int jj = (t.m & 7) * 32 + (lane & 31);               // the 8 m-tiles of a band walk the same W tiles: each touches 32 of the 256 lines
#else
      const int jj = lane * 4;                                   // every fourth line of the 256
#endif
      pf_off = (uint32_t)((long)(jj >> 7) * w_half + (long)(jj & 127) * p.ldw * 2);
    } else {                                                     // A: tile row jj (clamped to the segment's last row; identity / ungathered rows only)
#if PP_L2_TOUCH_SPLIT
      const int jj = ((t.n / RN) & 3) * 56 + min(lane, 55);      // the n-runs of a band walk the same A tile: four of them share an XCD
#else
      const int jj = min(lane * 4, BM - 1);
#endif
      int srow = min(t.row0 + jj, t.row_end - 1);
      if (p.a_rows != nullptr) srow = p.identity_rows ? srow - t.seg0 : 0;
      pf_off = (uint32_t)((long)srow * p.lda * 2);
    }
  };
  auto pf_touch = [&](const char* base) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + pf_off),
                                     (__attribute__((address_space(3))) void*)(smem + LDS_PF + wave * 256), 4, 0, 16);   // aux 16 = sc1: served by L2, not kept in L1
  };
#endif
  int nrm_par = 0;
  uint32_t a_off[2][2] = {{0, 0}, {0, 0}};                     // byte offsets of this lane's A rows (gathered) from p.A, per half / piece
  const char* Ak = nullptr;
  const char* Wc = nullptr;

  while (true) {
    if (fresh) {
      // ---- (re)start the operand stream for a new m-tile / K-slice: nothing this wave issued is in flight after the wait; other waves'
      //      DMA only ever writes their own pieces, and every fragment read of the previous tile was consumed by its MFMAs.
      // Both wave rows run the (re)start CONCURRENTLY: the stagger is taken out first (row 0 passes the barrier row 1 still owes) and put back
      // at the end.  With the stagger left in, row 1's first barrier here would pair with row 0's LAST one and row 1 would only begin its index
      // loads after row 0 had finished its whole prologue (measured: 15.5k cycles to the first MFMA instead of ~6k).
      wait_vmcnt<0>();
      if (staggered && wr == 0) PP_BAR();
      PP_BAR();
      nrm_par ^= 1;
      const int last = cur.row_end - 1;
      Ak = reinterpret_cast<const char*>(p.A) + (long)cur.slice * nk * BKK * 2;
      Wc = w_tile_base(cur);
#if PP_L2_TOUCH
      pf_set(cur);
#endif
      // ONE dependent round trip: the gathered-row indices of this lane's four A pieces and of its two norm-row pieces (fused ln_2), together
      const bool do_nrm = SWI && p.ss_in;
      int srow[2][2], nrow[2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) srow[h][q] = min(cur.row0 + h * 128 + (wave * 2 + q) * 8 + r8, last);   // rows past the segment re-read a valid row (never stored)
#pragma unroll
      for (int q = 0; q < 2; ++q) nrow[q] = min(cur.row0 + (wave * 2 + q) * 16 + (lane >> 2), last);
      // Issue order: the six index loads (hand-issued: the compiler would wait for them with vmcnt(0), i.e. behind the W DMA), then the four
      // W half-tiles - cold in HBM, the longest latency of the start-up, and independent of the indices - then a COUNTED wait that lets the
      // eight W DMA instructions stay in flight (loads retire in order), then the A half-tiles.
      const bool gather = p.a_rows != nullptr && !p.identity_rows;
      if (p.a_rows != nullptr && p.identity_rows) {               // promised: a_rows[o[e] + i] == i - no index round trip in front of the A tiles
        const int seg0 = cur.seg0;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 2; ++q) srow[h][q] -= seg0;
        nrow[0] -= seg0; nrow[1] -= seg0;
      }
      int tk[6] = {0, 0, 0, 0, 0, 0};
      if (gather) {
        const int* ap[6] = {p.a_rows + srow[0][0], p.a_rows + srow[0][1], p.a_rows + srow[1][0], p.a_rows + srow[1][1],
                            p.a_rows + (do_nrm ? nrow[0] : srow[0][0]), p.a_rows + (do_nrm ? nrow[1] : srow[0][0])};
#pragma unroll
        for (int q = 0; q < 6; ++q) asm volatile("global_load_dword %0, %1, off" : "=v"(tk[q]) : "v"(ap[q]) : "memory");
      }
      stage(_1, _0, _0, Wc, b_off[0], b_off[1]);
      stage(_1, _0, _1, Wc + w_half, b_off[0], b_off[1]);
      stage(_1, _1, _0, Wc + 128, b_off[0], b_off[1]);
      stage(_1, _1, _1, Wc + w_half + 128, b_off[0], b_off[1]);
      if (gather) {
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(tk[0]), "+v"(tk[1]), "+v"(tk[2]), "+v"(tk[3]), "+v"(tk[4]), "+v"(tk[5])::"memory");
        PP_SB();
        srow[0][0] = tk[0]; srow[0][1] = tk[1]; srow[1][0] = tk[2]; srow[1][1] = tk[3]; nrow[0] = tk[4]; nrow[1] = tk[5];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) a_off[h][q] = (uint32_t)(((long)srow[h][q] * p.lda + lchunk * 8) * 2);
      stage(_0, _0, _0, Ak, a_off[0][0], a_off[0][1]);
      if (stage_a1) stage(_0, _0, _1, Ak, a_off[1][0], a_off[1][1]);
      stage(_0, _1, _0, Ak + 128, a_off[0][0], a_off[0][1]);
      if (stage_a1) stage(_0, _1, _1, Ak + 128, a_off[1][0], a_off[1][1]);
      if constexpr (SWI) {
        // fused ln_2 consumer: the tile rows' per-64-column partial sums of squares (64 B per token row) come in by DMA as well - 16 rows per
        // instruction, 4 lanes per row - instead of 16 scattered dword loads per row (measured: those loads, 4096 cache-line requests per
        // workgroup, queue in front of the operand DMA and cost ~10k cycles of start-up)
        if (do_nrm) {
          const int chunk = (lane & 3) * 4 < p.ss_n ? (lane & 3) : 0;   // D < 1024: fewer than 16 partials per row; the reader ignores the rest
#pragma unroll
          for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ss_in + (long)nrow[q] * p.ss_n + chunk * 4),
                                             (__attribute__((address_space(3))) void*)(smem + LDS_SS + (wave * 2 + q) * 1024), 16, 0, 0);
        }
      }
      // K-steps 0 and 1 complete and waited for: the state in which every output tile begins (see the epilogue)
      wait_vmcnt<0>();
      PP_BAR();
      if constexpr (SWI) {
        // 1 / max(|x_row| K^-1/2, eps) per tile row (summation order of gemm_bf16.hip: p_j = v_j + v_{j+8}, then the xor-shuffle tree
        // ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7))) -> LDS strip read by the epilogues (any number of barriers later)
        if (do_nrm && tid < 256) {
          const float4* sp = reinterpret_cast<const float4*>(smem + LDS_SS + tid * 64);
          const float4 q0 = sp[0], q1 = sp[1], q2 = sp[2], q3 = sp[3];
          const float v[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
          const int nss = p.ss_n;
          float pj[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pj[j] = (j < nss ? v[j] : 0.f) + (j + 8 < nss ? v[j + 8] : 0.f);
          const float ssum = ((pj[0] + pj[1]) + (pj[2] + pj[3])) + ((pj[4] + pj[5]) + (pj[6] + pj[7]));
          const float rk = rsqrtf((float)p.K);
          reinterpret_cast<float*>(smem + LDS_NRM)[nrm_par * 256 + tid] = __frcp_rn(fmaxf(__fsqrt_rn(ssum) * rk, p.ss_eps));
        }
      }
      wait_lgkmcnt<0>();
      if (wr == 1) PP_BAR();                                     // stagger: wave row 1 runs one barrier behind wave row 0 from here on
      staggered = true;
      PP_SB();
      fresh = false;
    }
    const bool has_next = L + 1 < Lend;
    bool cont = false;
    if (has_next) {
      map_tile(L + 1, nxt);
      cont = nxt.m == cur.m && nxt.slice == cur.slice;
    }
    const char* Wn = cont ? w_tile_base(nxt) : Wc;             // no successor on this stream: the tail re-reads valid memory, never consumed

    // ------------------------------------------------------------------------------------------------ K loop: 4 phases = 2 K-steps
    // A phase = one A half x BOTH W halves (32 / 8*FM1 MFMAs per wave between two barriers).  A half-tile is re-staged one whole phase after the
    // phase whose read section read it last:
    //   R(P0, t): reads W0 W1 A0 of K-step t;  stages A1[t+1]           R(P1, t): reads A1[t];  stages W0 W1 A0 of K-step t+2
    // and every read section ends in ONE counted wait that leaves the newest K-step's worth of DMA (8 instructions; 6 on the waves that stage
    // no A half 1) in flight: everything a later phase reads was issued before those.  (Round-2 measurements behind this shape - the eight-phase
    // loop, barrier hand-over inside a cluster, per-phase cycle stamps - are in LABNOTES.md §4.)
    // The FIRST pair of a tile is peeled (K-steps 0 and 1 were resident before it began and the previous tile's stores may still be draining -
    // vmcnt counts them - so it takes no waits, skips the A1[1] stage and requests the bias slice), and the wave role is a template argument:
    // between two barriers there is no branch.
#define PP_COMPUTE2(AH)           \
  PP_BAR();                       \
  wait_lgkmcnt<0>();              \
  PP_SB();                        \
  __builtin_amdgcn_s_setprio(1);  \
  mma(AH, _0);                    \
  mma(AH, _1);                    \
  __builtin_amdgcn_s_setprio(0);  \
  PP_SB();                        \
  PP_BAR();                       \
  PP_SB();
    auto kpair = [&](auto A1_, auto FIRST_, int kt) __attribute__((always_inline)) {
      constexpr bool a1 = decltype(A1_)::value != 0, first = decltype(FIRST_)::value != 0;
      constexpr bool pf = PP_L2_TOUCH != 0 && !a1;               // waves 6 / 7 (the role without A half 1) carry the L2 touches
      constexpr int NW = (a1 ? 8 : 6) + (pf ? 1 : 0);            // + the touch of a K-step (issued right behind a wait: always inside the budget)
      const bool cross = kt + 2 >= nk;
      const int k2 = cross ? kt + 2 - nk : kt + 2;             // K-step (kt+2) inside its own output tile
      const char* A1 = Ak + (long)(kt + 1) * 128;
      const char* A2 = Ak + (long)k2 * 128;
      const char* W2 = (cross ? Wn : Wc) + (long)k2 * 128;
#if PP_L2_TOUCH
      // lines of K-steps kt + 2 + d and kt + 3 + d (d = PP_L2_TOUCH): W of this or the next output tile, A of the same rows (wraps to the tile's k = 0)
      const int kp0 = kt + 2 + PP_L2_TOUCH, kp1 = kp0 + 1;
      const bool x0 = kp0 >= nk, x1 = kp1 >= nk;
      const int kq0 = x0 ? kp0 - nk : kp0, kq1 = x1 ? kp1 - nk : kp1;
      const char* T0 = (wave == 6 ? (x0 ? Wn : Wc) : Ak) + (long)kq0 * 128;
      const char* T1 = (wave == 6 ? (x1 ? Wn : Wc) : Ak) + (long)kq1 * 128;
#endif
      // P0 of K-step kt [buffer 0]
      rdB(_0, _0); rdB(_0, _1);
      PP_SB();
      rdA(_0, _0);
      if constexpr (a1 && !first) stage(_0, _1, _1, A1, a_off[1][0], a_off[1][1]);
      if constexpr (!first) wait_vmcnt<NW>();
      PP_COMPUTE2(_0)
      // P1 of K-step kt
      rdA(_0, _1);
      stage(_1, _0, _0, W2, b_off[0], b_off[1]);
      stage(_1, _0, _1, W2 + w_half, b_off[0], b_off[1]);
      stage(_0, _0, _0, A2, a_off[0][0], a_off[0][1]);
      if constexpr (!first) wait_vmcnt<NW>();
      if constexpr (HAS_BIAS && first) {                         // this tile's bias slice -> the wave's own LDS slot (behind the wait: not counted by it)
        const float* bsrc = p.bias + (long)cur.expert * p.bias_estride + (long)cur.n * NOUT;
        const float* bl = SWI ? (lane < 32 ? bsrc + lane * 4 : bsrc + p.N + (lane - 32) * 4) : bsrc + lane * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bl,
                                         (__attribute__((address_space(3))) void*)(smem + LDS_BIAS + wave * 1024), 16, 0, 0);
      }
#if PP_L2_TOUCH
      if constexpr (pf) pf_touch(T0);
#endif
      PP_COMPUTE2(_1)
      // P0 of K-step kt+1 [buffer 1]
      rdB(_1, _0); rdB(_1, _1);
      PP_SB();
      rdA(_1, _0);
      if constexpr (a1) stage(_0, _0, _1, A2, a_off[1][0], a_off[1][1]);
      if constexpr (!first) wait_vmcnt<NW>();
      PP_COMPUTE2(_0)
      // P1 of K-step kt+1; K-step kt+2 [buffer 0] retired by this wait
      rdA(_1, _1);
      stage(_1, _1, _0, W2 + 128, b_off[0], b_off[1]);
      stage(_1, _1, _1, W2 + w_half + 128, b_off[0], b_off[1]);
      stage(_0, _1, _0, A2 + 128, a_off[0][0], a_off[0][1]);
      wait_vmcnt<NW>();
#if PP_L2_TOUCH
      if constexpr (pf) pf_touch(T1);
#endif
      PP_COMPUTE2(_1)
    };
    auto kloop = [&](auto A1_) __attribute__((always_inline)) {
      kpair(A1_, _1, 0);
#pragma unroll 1
      for (int kt = 2; kt < nk; kt += 2) kpair(A1_, _0, kt);
    };
    if constexpr (FM1 == 4) {
      kloop(_1);
    } else {
      if (stage_a1) kloop(_1); else kloop(_0);
    }
#undef PP_COMPUTE2

    // ------------------------------------------------------------------------------------------------ epilogue: registers -> global
    // bias / SwiGLU in registers (a lane owns 8 consecutive output columns of a token row), stored straight from registers, 16 bytes per lane.
    // What makes an epilogue expensive here is not its VALU work but its STORES: CDNA4's vmcnt counts stores, the L2 is write-through, and a
    // counted wait of the next tile's K loop that is reached before the stores have drained stalls on them (measured: 8.9 us of a 58 us launch;
    // staging through LDS for whole-row stores cost more in barriers than it saved).  So the order is: (1) the LAST half-tile of the next output
    // tile's second K-step is requested, (2) all outputs are computed and packed while every DMA still in flight lands, (3) `vmcnt(0)` - by now
    // free - so K-steps 0 and 1 of the next tile are resident, (4) the stores are issued back to back.  The next tile's first eight phases then
    // run without any vmcnt wait; the first counted wait (phase 8) comes ~4.5k cycles after the stores were issued.
    {
      const int rows_valid = cur.row_end - cur.row0;
      // store addressing = UNIFORM tile base (scalar registers) + one 32-bit per-lane offset shared by all rounds: the row / column a round adds is
      // wave-uniform, so a lane's 14 stores need ONE address VGPR (64-bit per-round addresses had pushed the kernel into spills whose reloads -
      // vmcnt counts stores too - serialised the stores)
      char* Ct = reinterpret_cast<char*>(p.C) + ((long)cur.slice * p.split_stride + (long)cur.row0 * p.ldc + (long)cur.n * NOUT + wc * 32) * ESZ;
      const uint32_t c_lane = (uint32_t)(fr * (int)p.ldc + fq * 8) * ESZ;
      if (cont && stage_a1) stage(_0, _1, _1, Ak + 128, a_off[1][0], a_off[1][1]);      // next tile, K-step 1, A half 1 (its slot was last read in phase 7)
      // The two wave rows run their epilogues CONCURRENTLY: wave row 0 passes one extra barrier here (it pairs with wave row 1's last phase
      // barrier, so row 1 is released into its epilogue one MFMA segment later instead of after row 0's whole epilogue) and wave row 1 passes one
      // after its epilogue (pairs with row 0's first barrier of the next tile), which restores the one-barrier stagger.
      if (wr == 0) PP_BAR();
      [[maybe_unused]] float4 bq[2][2];                        // bias of this lane's 8 columns: [W half][4-column group]
      [[maybe_unused]] float rs[2][4];
      if constexpr (HAS_BIAS) {
        const uint32_t ba = lds0 + LDS_BIAS + wave * 1024 + (wc * 32 + fq * 8) * 4;
        lds_read_f4(bq[0][0], ba); lds_read_f4(bq[0][1], ba + 16);
        lds_read_f4(bq[1][0], ba + 512); lds_read_f4(bq[1][1], ba + 528);
      }
      if constexpr (SWI) {
        if (p.ss_in) {
          const uint32_t na = lds0 + LDS_NRM + nrm_par * 1024;
#pragma unroll
          for (int i = 0; i < 4; ++i) lds_read_f1(rs[0][i], na + (wr * 64 + i * 16 + fr) * 4);
#pragma unroll
          for (int i = 0; i < FM1; ++i) lds_read_f1(rs[1][i], na + (128 + wr * 16 * FM1 + i * 16 + fr) * 4);
        }
      }
      wait_lgkmcnt<0>();
      PP_SB();
      if constexpr (SWI) {
        if (!p.ss_in) {
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) rs[a][i] = 1.0f;       // x 1.0f is exact
        }
      }
      // one ROUND = one fragment row (x one 128-column W half when there is no SwiGLU pairing): this lane's 8 outputs of it
      constexpr int NFR = 4 + FM1, R = SWI ? NFR : 2 * NFR;
      auto outputs = [&](int r, float (&ov)[8]) {                 // r is a compile-time constant after unrolling
        const int fi = SWI ? r : r >> 1, a = fi < 4 ? 0 : 1, i = a ? fi - 4 : fi;
        if constexpr (SWI) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4 v = acc[a][0][i][j], gt = acc[a][1][i][j];
            const float4 bp = bq[0][j], bg = bq[1][j];
            ov[j * 4 + 0] = swiglu_f(v[0], gt[0], rs[a][i], bp.x, bg.x); ov[j * 4 + 1] = swiglu_f(v[1], gt[1], rs[a][i], bp.y, bg.y);
            ov[j * 4 + 2] = swiglu_f(v[2], gt[2], rs[a][i], bp.z, bg.z); ov[j * 4 + 3] = swiglu_f(v[3], gt[3], rs[a][i], bp.w, bg.w);
          }
        } else {
          const int b = r & 1;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x4 v = acc[a][b][i][j];
            if constexpr (HAS_BIAS) { const float4 bb = bq[b][j]; v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
            if constexpr (EPI == MODE_EPI_BIAS_GELU) { v[0] = gelu_erf_f(v[0]); v[1] = gelu_erf_f(v[1]); v[2] = gelu_erf_f(v[2]); v[3] = gelu_erf_f(v[3]); }
            ov[j * 4 + 0] = v[0]; ov[j * 4 + 1] = v[1]; ov[j * 4 + 2] = v[2]; ov[j * 4 + 3] = v[3];
          }
        }
      };
      auto round_urow = [&](int r) { const int fi = SWI ? r : r >> 1, a = fi < 4 ? 0 : 1, i = a ? fi - 4 : fi;       // wave-uniform part of the row
                                     return a * 128 + (a ? wr * 16 * FM1 : wr * 64) + i * 16; };
      auto round_col = [&](int r) { return SWI ? 0 : (r & 1) * 128; };
      if constexpr (OUT_BF16) {
        u32x4 pk[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float ov[8];
          outputs(r, ov);
          pk[r] = u32x4{pack_bf16x2(ov[0], ov[1]), pack_bf16x2(ov[2], ov[3]), pack_bf16x2(ov[4], ov[5]), pack_bf16x2(ov[6], ov[7])};
        }
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" : "+v"(pk[r]));      // every output is computed before the wait below
        wait_vmcnt<0>();
        PP_SB();
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int urow = round_urow(r);
          if (fr < rows_valid - urow) *reinterpret_cast<u32x4*>(Ct + ((long)urow * p.ldc + round_col(r)) * 2 + c_lane) = pk[r];
        }
      } else {
        wait_vmcnt<0>();
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float ov[8];
          outputs(r, ov);
          const int urow = round_urow(r);
          if (fr < rows_valid - urow) {
            float* c = reinterpret_cast<float*>(Ct + ((long)urow * p.ldc + round_col(r)) * 4 + c_lane);
            *reinterpret_cast<float4*>(c) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            *reinterpret_cast<float4*>(c + 4) = make_float4(ov[4], ov[5], ov[6], ov[7]);
          }
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (wr == 1) PP_BAR();
    }
    if (!has_next) break;
    ++L;
    cur = nxt;
    fresh = !cont;
    Wc = Wn;
  }
  wait_vmcnt<0>();                                             // the tail of the operand stream must land before the LDS is released

  if (staggered && wr == 0) PP_BAR();                                       // balance the stagger barrier of wave row 1
#undef PP_BAR
#undef PP_SB
}

// ------------------------------------------------------------------------------------------------------------ host side
int pp_num_cus() {   // also used by the tile heuristic (gemm_bf16.hip: pick_cfg)
  static std::atomic<int> ncu[kMaxDevices];                   // zero-initialised; a racing first call writes the same value twice
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
  int n = ncu[dev].load(std::memory_order_relaxed);
  if (!n) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 256;
    n = v > 0 ? v : 256;
    ncu[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

template <int EPI, bool OUT_BF16, int FM1>
static int pp_launch(GemmParams p, const ModeGemmDesc* d, hipStream_t s) {
  constexpr int BM = 128 + 32 * FM1, NOUT = (EPI == MODE_EPI_SWIGLU) ? 128 : 256;
  p.n_tiles = d->N / NOUT;
  p.m_tiles = (d->M + BM - 1) / BM + (d->expert_offsets ? d->num_experts : 0);     // upper bound; the kernel counts the real m-tiles
  const long t_max = (long)p.m_tiles * p.n_tiles * p.split_k;
  const int ncu = pp_num_cus();
  const int grid = (int)(t_max < ncu ? t_max : ncu);           // one persistent workgroup per CU (140 KiB of LDS each)
  auto kern = gemm_pp_kernel<EPI, OUT_BF16, FM1>;
  static LdsLimitOnce lds_once;
  {
    const int rc = lds_once.ensure(reinterpret_cast<const void*>(kern), pp::LDS_TOTAL);
    if (rc != MODE_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), pp::LDS_TOTAL, s, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

// Entered from gemm_bf16_launch with a validated descriptor and a filled parameter block.  Returns MODE_ERR_UNSUPPORTED for shapes /
// epilogues this kernel does not take (the caller falls back to the 128x128 family).  Tiles: 224 rows (FM1 = 3, all epilogues) and 256 rows (FM1 = 4,
// NONE / BIAS: ragged expert segments).  A spill inside the K loop would break the COUNTED vmcnt waits (scratch traffic counts in vmcnt; seen once on an
// earlier 256-row SwiGLU variant) - tests/test_boundary.py::test_pp_kernel_isa_contract pins "no scratch access in an MFMA block" on the shipped ISA of
// every instantiation.
int gemm_bf16_pp_launch(const ModeGemmDesc* d, const GemmParams& p0, int rows256, hipStream_t s) {
  const int epi = d->epilogue;
  if (rows256 && epi == MODE_EPI_SWIGLU) return MODE_ERR_UNSUPPORTED;          // the 256-row tile exists for the NONE / BIAS epilogues (see below)
  if (epi != MODE_EPI_NONE && epi != MODE_EPI_BIAS && epi != MODE_EPI_SWIGLU) return MODE_ERR_UNSUPPORTED;
  const int nout = epi == MODE_EPI_SWIGLU ? 128 : 256;
  const int S = p0.split_k;
  if (d->k_group_offsets || d->N % nout || d->K % (128 * S) || d->K / S < 128) return MODE_ERR_UNSUPPORTED;
  if (d->expert_offsets && d->num_experts > 8) return MODE_ERR_UNSUPPORTED;
  if (d->ldc % 8 || (reinterpret_cast<uintptr_t>(d->C) & 15) || (S > 1 && d->split_stride % 8)) return MODE_ERR_UNSUPPORTED;
  if (d->row_ss && (d->row_ss_n > 16 || d->row_ss_n % 4 || (reinterpret_cast<uintptr_t>(d->row_ss) & 15))) return MODE_ERR_UNSUPPORTED;   // fused ln_2 partial sums: D <= 1024, D % 256 == 0
  // 32-bit per-lane byte offsets: both operands must span < 4 GiB from their bases
  const long wrows = (epi == MODE_EPI_SWIGLU ? 2L : 1L) * d->N;   // gathered A rows are token ids < M (M = tokens x top_k sorted rows)
  if (wrows * d->ldw * 2 >= (1L << 32) || (long)d->M * d->lda * 2 >= (1L << 32)) return MODE_ERR_UNSUPPORTED;
  if (d->bias && ((reinterpret_cast<uintptr_t>(d->bias) & 15) || d->bias_expert_stride % 4)) return MODE_ERR_UNSUPPORTED;
  const bool ob = d->out_dtype == MODE_BF16;
  if (rows256) {
    // 256-row tile (FM1 = 4): for RAGGED expert segments.  With the reference's per-token multinomial routing an expert owns 896 +- 30 of the 3584 sorted
    // rows: five 224-row tiles for half of the experts (608-640 tiles = three rounds of a 256-CU part), but four 256-row tiles for all of them (512 tiles
    // = two rounds).  Same loop, same numerics; the ISA contract test covers these instantiations too (no scratch access in an MFMA block).
    if (epi == MODE_EPI_NONE) return ob ? pp_launch<MODE_EPI_NONE, true, 4>(p0, d, s) : pp_launch<MODE_EPI_NONE, false, 4>(p0, d, s);
    return ob ? pp_launch<MODE_EPI_BIAS, true, 4>(p0, d, s) : pp_launch<MODE_EPI_BIAS, false, 4>(p0, d, s);
  }
#define PP_CASE(E) \
  case E: return ob ? pp_launch<E, true, 3>(p0, d, s) : pp_launch<E, false, 3>(p0, d, s);
  switch (epi) {
    PP_CASE(MODE_EPI_NONE)
    PP_CASE(MODE_EPI_BIAS)
    PP_CASE(MODE_EPI_SWIGLU)
    default: return MODE_ERR_UNSUPPORTED;
  }
#undef PP_CASE
}

}  // namespace mode
