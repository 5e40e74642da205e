// NOT PART OF THE LIBRARY (round 4, measured and rejected - profiles/r04_w4_vs_pp.txt): it was built into libmode_hip.so behind a "gemm_w4" option, passed the
// bit-identity tests against the ping-pong and ring kernels, and lost in situ (same ~2400 cycles per K-step with cold weights, larger fixed part).  Kept as
// the record of the experiment; compiles against mode_diffusion_policy_amd/csrc/mode_common.h (-I that directory -I include).
//
// Persistent FOUR-wave bf16 MFMA GEMM for gfx950 - the expert projections at large batch:
//   C[M,N] = epilogue(A[M,K] @ W[N,K]^T), row-gathered / grouped (MoE, uniform groups) / K-sliced; epilogues SWIGLU (+ fused ln_2 row scale) and NONE, bf16 out.
//
// What changed against the eight-wave ping-pong kernel, and why (measurements: scripts/probe/w3_kloop_probe.hip, w4k64_probe.hip, LABNOTES.md section 4):
//   * ONE wave per SIMD with the 512-register budget: a wave owns 112 rows x 128 weight rows of the 224 x 256 tile (7 x 8 accumulator fragments = 224 AGPRs).
//     Per K-step the four waves read 4 x (112 + 128) x 128 B = 120 KiB of fragments from LDS; eight waves of 112 x 64 read 176 KiB.  With the 60 KiB the
//     operand DMA writes, the eight-wave form keeps the LDS port busy for 1 850 cycles of a K-step whose MFMAs need 1 792 - it was LDS-port bound whatever
//     its barrier schedule.
//   * The operand ring is split BY OPERAND and k64-granular: activations 2 slots x 28 KiB + weights 3 slots x 32 KiB = 152 KiB, 128-byte rows, so a
//     global_load_lds_dwordx4 moves 8 full cache lines.  The weights of K-step s+3 and the activations of s+2 are requested during K-step s: the L2 -> LDS
//     stream never stops and has 2.5 K-steps (weights: HBM / Infinity Cache) resp. 1 K-step (activations: L2) to land.  DMA alone on this ring: 1 734
//     cycles per K-step = 35 B/clk/CU (the k32-granular five-slot ring of the round-2 probes: 2 373).
//   * ONE s_barrier per K-step, in the MIDDLE of the step: after the wave has read the step's last fragments and before the MFMAs that consume them -
//       [56 MFMAs on (s, k32 half 0) | fragment reads of (s, half 1)] -> lgkmcnt(0), vmcnt(8), s_barrier ->
//       [56 MFMAs on (s, half 1) | fragment reads of (s+1, half 0) | DMA: A(s+2) -> the slot step s vacated, W(s+3) -> the slot of W(s)]
//     The barrier says "every wave has read all of step s" and "step s+1 has landed for every wave" (counted wait: the 8 newest instructions, W(s+2), stay
//     in flight; activations are always requested BEFORE weights so that this holds).  MFMAs are inline asm on AGPR accumulators, the fragment reads and
//     the 15 DMA instructions of a step are interleaved with them by hand.
//   * Persistent like the ping-pong kernel: a workgroup walks consecutive n-tiles of one m-tile, the K-step sequence simply continues into the next tile
//     (its first steps are in flight while the epilogue runs); weight rows are permuted on their way into LDS so that a lane owns 8 consecutive output
//     columns (SwiGLU: the value AND the gate rows of a wave's 64 outputs sit in that wave's half of the tile) and the tile leaves through 16-byte stores.
// Numerics: the same k-ordered fp32 MFMA chain per output element and the same epilogue expressions (swiglu_f, the ln_2 partial-sum tree) as every other
// forward geometry: BIT-IDENTICAL results (tests/test_gpu_kernels.py, the batch-slice tests of the full model).
#include <type_traits>
#include <utility>

#include "mode_common.h"

namespace mode {

namespace w4 {
constexpr int BM = 224, BKK = 64;
constexpr int A_BYTES = BM * 128, W_BYTES = 256 * 128;          // 28 KiB, 32 KiB per K-step
constexpr int W_BASE = 2 * A_BYTES, LDS_NRM = W_BASE + 3 * W_BYTES, LDS_BIAS = LDS_NRM + 1024, LDS_TOTAL = LDS_BIAS + 4 * 1024;   // ring 152 KiB + norm strip + one bias slot per wave = 157 KiB
constexpr int GM = 8;                                           // m-tiles per rasterisation band
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ void lds_read128(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_read_f1(float& dst, uint32_t addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); }
__device__ __forceinline__ void lds_read_f4(float4& dst, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); }
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// accumulators pinned to AGPRs and updated in place; asm volatile keeps the hand-written MFMA / ds_read / DMA interleave
__device__ __forceinline__ void mfma_acc(f32x4& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
}  // namespace w4

#define W4_SB() __builtin_amdgcn_sched_barrier(0)

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const GemmParams p) {
  using namespace w4;
  constexpr bool SWI = EPI == MODE_EPI_SWIGLU;
  constexpr int NOUT = SWI ? 128 : 256;                        // output columns of a tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---------------------------------------------------------------------------------------------------- tile space (all scalar; as gemm_bf16_pp.hip)
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
  const int nwg = (T + R - 1) / R;
  if ((int)blockIdx.x >= nwg) return;
  const int wg = xcd_remap(blockIdx.x, nwg);
  int L = wg * R;
  const int Lend = min(T, L + R);
  const int nk = p.K / BKK / S;                                // K-steps per slice (>= 4: checked by the launcher)

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
  // byte address of a tile's first weight row at K-offset 0 of its slice
  auto w_tile_base = [&](const Tile& t) -> const char* {
    return reinterpret_cast<const char*>(p.W) + ((long)t.expert * p.w_estride + (long)t.n * NOUT * p.ldw + (long)t.slice * nk * BKK) * 2;
  };

  // ---------------------------------------------------------------------------------------------------- per-lane constants
  // DMA: one global_load_lds_dwordx4 fills a 1-KiB piece = 8 rows x 128 B; lane i -> row i>>3, physical 16-B chunk i&7 = LOGICAL chunk (i&7)^(i>>3) (XOR
  // swizzle on the source address, linear destination, same XOR on the fragment reads).  A: 28 pieces, 7 per wave; W: 32 pieces, 8 per wave.
  const int r8 = lane >> 3, lc = (lane & 7) ^ r8;
  uint32_t w_off[8];                                           // byte offsets from a W tile base (tile independent)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int lrow = (wave * 8 + q) * 8 + r8;                  // LDS row of the W tile, 0..255: rows 128 c .. 128 c + 127 belong to wave column c
    // inside a 32-row block: LDS row j*16 + q4*4 + r holds weight row q4*8 + j*4 + r  ->  a lane's 2 fragments x 4 accumulator rows = 8 consecutive columns
    long wrow;
    if constexpr (SWI) {
      const int c = lrow >> 7, within = lrow & 127, gate = within >> 6, r64 = within & 63;
      const int blk = r64 >> 5, rho = r64 & 31;
      const int col = blk * 32 + ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);
      wrow = (gate ? (long)p.N : 0L) + c * 64 + col;           // value row n and gate row N + n of the same output (modedit.py:89) in the same wave column
    } else {
      const int blk = lrow >> 5, rho = lrow & 31;
      wrow = blk * 32 + ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);
    }
    w_off[q] = (uint32_t)((wrow * p.ldw + lc * 8) * 2);
  }
  // fragment reads: lane -> row (l&15) of a 16-row fragment, 16-B chunk (l>>4) [+4 for the second k32 half], chunk XOR (row & 7)
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t a_rd[2], w_rd[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int c = ((fq + 4 * kh) ^ (fr & 7)) * 16;
    a_rd[kh] = lds0 + (wr * 112 + fr) * 128 + c;
    w_rd[kh] = lds0 + W_BASE + (wc * 128 + fr) * 128 + c;
  }

  f32x4 acc[7][8];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 Af[2][7], Wf[2][8];

  uint32_t a_off[7] = {0, 0, 0, 0, 0, 0, 0};                   // byte offsets of this lane's A rows (gathered) from p.A
  const char* Ak = nullptr;
  const char* Wc = nullptr;
  auto dma_a = [&](auto q_, int slot, int kstep) __attribute__((always_inline)) {
    constexpr int q = decltype(q_)::value;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ak + a_off[q] + (long)kstep * 128),
                                     (__attribute__((address_space(3))) void*)(smem + slot * A_BYTES + (wave * 7 + q) * 1024), 16, 0, 0);
  };
  auto dma_w = [&](auto q_, int slot, const char* Wt, int kstep) __attribute__((always_inline)) {
    constexpr int q = decltype(q_)::value;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wt + w_off[q] + (long)kstep * 128),
                                     (__attribute__((address_space(3))) void*)(smem + W_BASE + slot * W_BYTES + (wave * 8 + q) * 1024), 16, 0, 0);
  };

  Tile cur, nxt;
  map_tile(L, cur);
  bool fresh = true;
  bool after_store = false;                                    // the previous tile's stores are the newest vector-memory instructions: see the first barrier of a tile
  int a_slot = 0, w_slot = 0;                                  // slots of the current K-step

  while (true) {
    if (fresh) {
      // ---- (re)start the operand stream for a new m-tile / K-slice.  Nothing this wave issued is in flight after the wait; every fragment read of
      //      the previous tile was consumed by its MFMAs; after the barrier no wave still reads the ring or the norm strip.
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const int last = cur.row_end - 1;
      Ak = reinterpret_cast<const char*>(p.A) + (long)cur.slice * nk * BKK * 2;
      Wc = w_tile_base(cur);
      a_slot = 0; w_slot = 0;
      const bool do_nrm = SWI && p.ss_in;
      int srow[7], nrow;
#pragma unroll
      for (int q = 0; q < 7; ++q) srow[q] = min(cur.row0 + (wave * 7 + q) * 8 + r8, last);     // rows past the segment re-read a valid row (never stored)
      nrow = min(cur.row0 + min(tid, BM - 1), last);
      const bool gather = p.a_rows != nullptr && !p.identity_rows;
      if (p.a_rows != nullptr && p.identity_rows) {             // promised: a_rows[o[e] + i] == i - no index round trip in front of the A tiles
#pragma unroll
        for (int q = 0; q < 7; ++q) srow[q] -= cur.seg0;
        nrow -= cur.seg0;
      }
      int tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (gather) {
#pragma unroll
        for (int q = 0; q < 7; ++q) asm volatile("global_load_dword %0, %1, off" : "=v"(tk[q]) : "v"(p.a_rows + srow[q]) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(tk[7]) : "v"(p.a_rows + nrow) : "memory");
      }
      static_for<8>([&](auto q) { dma_w(q, 0, Wc, 0); });       // cold in HBM and independent of the indices: first
      if (gather) {
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(tk[0]), "+v"(tk[1]), "+v"(tk[2]), "+v"(tk[3]), "+v"(tk[4]), "+v"(tk[5]), "+v"(tk[6]), "+v"(tk[7])::"memory");
        W4_SB();
#pragma unroll
        for (int q = 0; q < 7; ++q) srow[q] = tk[q];
        nrow = tk[7];
      }
#pragma unroll
      for (int q = 0; q < 7; ++q) a_off[q] = (uint32_t)(((long)srow[q] * p.lda + lc * 8) * 2);
      // fused ln_2 consumer: this thread's tile row - its (up to 16) per-64-column partial sums of squares, four 16-byte loads issued before the operand DMA
      [[maybe_unused]] float4 sq[4];
      if constexpr (SWI) {
        if (do_nrm) {
          const float* sp = p.ss_in + (long)nrow * p.ss_n;
          const int nch = p.ss_n >> 2;
#pragma unroll
          for (int c = 0; c < 4; ++c) sq[c] = *reinterpret_cast<const float4*>(sp + min(c, nch - 1) * 4);
#pragma unroll
          for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(sq[c].x), "+v"(sq[c].y), "+v"(sq[c].z), "+v"(sq[c].w));
        }
      }
      static_for<7>([&](auto q) { dma_a(q, 0, 0); });
      static_for<8>([&](auto q) { dma_w(q, 1, Wc, 1); });
      static_for<7>([&](auto q) { dma_a(q, 1, 1); });
      static_for<8>([&](auto q) { dma_w(q, 2, Wc, 2); });
      wait_vmcnt<23>();                                          // W(0), the norm loads and A(0) landed; W(1), A(1), W(2) stay in flight
      if constexpr (SWI) {
        // 1 / max(|x_row| K^-1/2, eps) per tile row (summation order of gemm_bf16.hip: p_j = v_j + v_{j+8}, then ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)))
        if (do_nrm && tid < BM) {
          const float v[16] = {sq[0].x, sq[0].y, sq[0].z, sq[0].w, sq[1].x, sq[1].y, sq[1].z, sq[1].w,
                               sq[2].x, sq[2].y, sq[2].z, sq[2].w, sq[3].x, sq[3].y, sq[3].z, sq[3].w};
          const int nss = p.ss_n;
          float pj[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pj[j] = (j < nss ? v[j] : 0.f) + (j + 8 < nss ? v[j + 8] : 0.f);
          const float ssum = ((pj[0] + pj[1]) + (pj[2] + pj[3])) + ((pj[4] + pj[5]) + (pj[6] + pj[7]));
          const float rk = rsqrtf((float)p.K);
          reinterpret_cast<float*>(smem + LDS_NRM)[tid] = __frcp_rn(fmaxf(__fsqrt_rn(ssum) * rk, p.ss_eps));
        }
      }
      __builtin_amdgcn_s_barrier();                              // K-step 0 (and the norm strip) visible to every wave
      static_for<7>([&](auto i) { lds_read128<decltype(i)::value * 2048>(Af[0][decltype(i)::value], a_rd[0]); });
      static_for<8>([&](auto j) { lds_read128<decltype(j)::value * 2048>(Wf[0][decltype(j)::value], w_rd[0]); });
      wait_lgkmcnt<0>();
      W4_SB();
      fresh = false;
      after_store = false;
    }
    const bool has_next = L + 1 < Lend;
    bool cont = false;
    if (has_next) {
      map_tile(L + 1, nxt);
      cont = nxt.m == cur.m && nxt.slice == cur.slice;
    }
    const char* Wn = cont ? w_tile_base(nxt) : Wc;              // no successor on this stream: the tail re-reads valid memory, never consumed

    // ------------------------------------------------------------------------------------------------ K loop
    // One K-step.  The stream never changes its rhythm: past the end of a stream (no successor tile) the requests re-read valid memory that is never
    // consumed, like the ping-pong kernel's tail - one loop body, one counted wait.  BIAS (first step of a tile): this tile's bias slice goes to the
    // wave's LDS slot, requested IN FRONT of the step's operand DMA (anything issued after a step's weight DMA would be among "the 8 newest").
    auto kstep = [&](auto BIAS_, int s) __attribute__((always_inline)) {
      constexpr bool BIAS = decltype(BIAS_)::value != 0;
      // ---- half 0: 56 MFMAs on register set 0, the 15 fragments of (s, half 1) read underneath
      const uint32_t ar1 = a_rd[1] + a_slot * A_BYTES, wr1 = w_rd[1] + w_slot * W_BYTES;
      W4_SB();
      static_for<56>([&](auto m_) {
        constexpr int m = decltype(m_)::value, i = m / 8, j = m % 8;
        mfma_acc(acc[i][j], Wf[0][j], Af[0][i]);
        if constexpr (m % 2 == 1 && m / 2 < 15) {
          constexpr int r = m / 2;
          if constexpr (r < 7) lds_read128<r * 2048>(Af[1][r], ar1);
          else lds_read128<(r - 7) * 2048>(Wf[1][r - 7], wr1);
        }
      });
      W4_SB();
      wait_lgkmcnt<0>();
      // A(s+1), W(s+1) landed, W(s+2) may still be in flight.  First step after an epilogue: everything was waited for before the stores were issued, and
      // the stores are now the newest instructions - no wait (a counted wait here would stall on them: vmcnt counts stores).
      if (!(BIAS && after_store)) wait_vmcnt<8>();
      __builtin_amdgcn_s_barrier();                              // every wave has read all of step s; step s+1 is visible
      W4_SB();
      // ---- half 1: 56 MFMAs on register set 1; fragment reads of (s+1, half 0); DMA A(s+2) -> A slot of step s, W(s+3) -> W slot of step s
      const int na_slot = a_slot ^ 1, nw_slot = w_slot == 2 ? 0 : w_slot + 1;
      const uint32_t ar0 = a_rd[0] + na_slot * A_BYTES, wr0 = w_rd[0] + nw_slot * W_BYTES;
      const int my_a = a_slot, my_w = w_slot;
      int ka = s + 2, kw = s + 3;
      const char* Wt = Wc;
      if (ka >= nk) ka -= nk;                                    // (the next tile of this stream has the same activation rows)
      if (kw >= nk) { kw -= nk; Wt = Wn; }
      if constexpr (BIAS && SWI) {
        const float* bsrc = p.bias + (long)cur.expert * p.bias_estride + (long)cur.n * NOUT;
        const float* bl = lane < 32 ? bsrc + lane * 4 : bsrc + p.N + (lane - 32) * 4;        // [128 value | 128 gate] floats
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bl,
                                         (__attribute__((address_space(3))) void*)(smem + LDS_BIAS + wave * 1024), 16, 0, 0);
      }
      static_for<56>([&](auto m_) {
        constexpr int m = decltype(m_)::value, i = m / 8, j = m % 8;
        mfma_acc(acc[i][j], Wf[1][j], Af[1][i]);
        if constexpr (m % 3 == 0 && m / 3 < 15) {                // the step's 15 DMA instructions spread over its MFMAs, activations first
          constexpr int d = m / 3;
          W4_SB();
          if constexpr (d < 7) dma_a(std::integral_constant<int, d>{}, my_a, ka);
          else dma_w(std::integral_constant<int, d - 7>{}, my_w, Wt, kw);
          W4_SB();
        }
        if constexpr (m % 2 == 1 && m / 2 < 15) {
          constexpr int r = m / 2;
          if constexpr (r < 7) lds_read128<r * 2048>(Af[0][r], ar0);
          else lds_read128<(r - 7) * 2048>(Wf[0][r - 7], wr0);
        }
      });
      W4_SB();
      wait_lgkmcnt<0>();
      W4_SB();
      a_slot = na_slot; w_slot = nw_slot;
    };
    {
      constexpr std::integral_constant<int, 0> _0{};
      constexpr std::integral_constant<int, 1> _1{};
      kstep(_1, 0);
#pragma unroll 1
      for (int s = 1; s < nk; ++s) kstep(_0, s);
    }

    // ------------------------------------------------------------------------------------------------ epilogue: registers -> global
    // Every output is computed and packed while the DMA in flight (the next tile's first K-steps) lands; then vmcnt(0) - by now cheap - and the stores back
    // to back.  (vmcnt counts stores: a counted wait reached before they have drained would stall on them - see the first barrier of the K loop.)
    {
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");           // the last MFMAs retire before the accumulators are read (inline asm: no hazard tracking)
      const int rows_valid = cur.row_end - cur.row0;
      [[maybe_unused]] float rs[7];
      if constexpr (SWI) {
        if (p.ss_in) {
#pragma unroll
          for (int i = 0; i < 7; ++i) lds_read_f1(rs[i], lds0 + LDS_NRM + (wr * 112 + i * 16 + fr) * 4);
          wait_lgkmcnt<0>();
          W4_SB();
        } else {
#pragma unroll
          for (int i = 0; i < 7; ++i) rs[i] = 1.0f;             // x 1.0f is exact
        }
      }
      [[maybe_unused]] float4 bv[2][2], bg[2][2];              // bias of this lane's columns: [32-column block][4-column group], value / gate
      if constexpr (SWI) {
        const uint32_t ba = lds0 + LDS_BIAS + wave * 1024 + (wc * 64 + fq * 8) * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          lds_read_f4(bv[b][0], ba + b * 128); lds_read_f4(bv[b][1], ba + b * 128 + 16);
          lds_read_f4(bg[b][0], ba + 512 + b * 128); lds_read_f4(bg[b][1], ba + 512 + b * 128 + 16);
        }
        wait_lgkmcnt<0>();
        W4_SB();
      }
      constexpr int NP = SWI ? 2 : 4;                            // 8-column groups per fragment row of this lane
      u32x4 pk[7][NP];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
#pragma unroll
        for (int b = 0; b < NP; ++b) {
          float ov[8];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            if constexpr (SWI) {
              const f32x4 v = acc[i][b * 2 + jj], gt = acc[i][4 + b * 2 + jj];
              const float4 bp = bv[b][jj], bq = bg[b][jj];
              ov[jj * 4 + 0] = swiglu_f(v[0], gt[0], rs[i], bp.x, bq.x); ov[jj * 4 + 1] = swiglu_f(v[1], gt[1], rs[i], bp.y, bq.y);
              ov[jj * 4 + 2] = swiglu_f(v[2], gt[2], rs[i], bp.z, bq.z); ov[jj * 4 + 3] = swiglu_f(v[3], gt[3], rs[i], bp.w, bq.w);
            } else {
              const f32x4 v = acc[i][b * 2 + jj];
              ov[jj * 4 + 0] = v[0]; ov[jj * 4 + 1] = v[1]; ov[jj * 4 + 2] = v[2]; ov[jj * 4 + 3] = v[3];
            }
          }
          pk[i][b] = u32x4{pack_bf16x2(ov[0], ov[1]), pack_bf16x2(ov[2], ov[3]), pack_bf16x2(ov[4], ov[5]), pack_bf16x2(ov[6], ov[7])};
        }
      }
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int b = 0; b < NP; ++b) asm volatile("" : "+v"(pk[i][b]));      // every output is computed before the wait below
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      wait_vmcnt<0>();
      W4_SB();
      // store addressing = uniform tile base + one 32-bit per-lane offset; the row / column a store adds is wave-uniform
      char* Ct = reinterpret_cast<char*>(p.C) + ((long)cur.slice * p.split_stride + (long)cur.row0 * p.ldc + (long)cur.n * NOUT + wc * (NOUT / 2)) * 2;
      const uint32_t c_lane = (uint32_t)(fr * (int)p.ldc + fq * 8) * 2;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int urow = wr * 112 + i * 16;
#pragma unroll
        for (int b = 0; b < NP; ++b)
          if (fr < rows_valid - urow) *reinterpret_cast<u32x4*>(Ct + ((long)urow * p.ldc + b * 32) * 2 + c_lane) = pk[i][b];
      }
      after_store = true;
    }
    if (!has_next) break;
    ++L;
    cur = nxt;
    fresh = !cont;
    Wc = Wn;
  }
  wait_vmcnt<0>();                                             // the tail of the operand stream must land before the LDS is released
}
#undef W4_SB

// ------------------------------------------------------------------------------------------------------------ host side
int pp_num_cus();                                               // gemm_bf16_pp.hip

// Entered from gemm_bf16_launch with a validated descriptor and a filled parameter block.  Returns MODE_ERR_UNSUPPORTED for everything this kernel does
// not take (the caller goes on to the ping-pong kernel / the 128x128 family): it is the large-batch INFERENCE kernel of the two expert projections -
// SWIGLU (bias, optional fused ln_2 scale) and NONE epilogues, bf16 output, ungrouped or grouped with the caller's uniform-groups promise.
int gemm_bf16_w4_launch(const ModeGemmDesc* d, const GemmParams& p0, hipStream_t s) {
  const int epi = d->epilogue;
  if (epi != MODE_EPI_NONE && epi != MODE_EPI_SWIGLU) return MODE_ERR_UNSUPPORTED;
  if (d->out_dtype != MODE_BF16) return MODE_ERR_UNSUPPORTED;
  if (d->expert_offsets && !(d->flags & MODE_GEMM_UNIFORM_GROUPS)) return MODE_ERR_UNSUPPORTED;
  const int nout = epi == MODE_EPI_SWIGLU ? 128 : 256;
  const int S = p0.split_k;
  if (d->k_group_offsets || d->N % nout || d->K % (64 * S) || d->K / S < 256) return MODE_ERR_UNSUPPORTED;
  if (d->expert_offsets && d->num_experts > 8) return MODE_ERR_UNSUPPORTED;
  if (d->ldc % 8 || (reinterpret_cast<uintptr_t>(d->C) & 15) || (S > 1 && d->split_stride % 8)) return MODE_ERR_UNSUPPORTED;
  if (d->row_ss && (d->row_ss_n > 16 || d->row_ss_n % 4 || (reinterpret_cast<uintptr_t>(d->row_ss) & 15))) return MODE_ERR_UNSUPPORTED;   // fused ln_2 partial sums: D <= 1024, D % 256 == 0
  // 32-bit per-lane byte offsets: both operands must span < 4 GiB from their bases
  const long wrows = (epi == MODE_EPI_SWIGLU ? 2L : 1L) * d->N;
  if (wrows * d->ldw * 2 >= (1L << 32) || (long)d->M * d->lda * 2 >= (1L << 32)) return MODE_ERR_UNSUPPORTED;
  if (epi == MODE_EPI_SWIGLU && (!d->bias || (reinterpret_cast<uintptr_t>(d->bias) & 15) || d->bias_expert_stride % 4)) return MODE_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->W)) & 15 || d->lda % 8 || d->ldw % 8) return MODE_ERR_UNSUPPORTED;
  GemmParams p = p0;
  p.n_tiles = d->N / nout;
  p.m_tiles = (d->M + w4::BM - 1) / w4::BM + (d->expert_offsets ? d->num_experts : 0);     // upper bound; the kernel counts the real m-tiles
  const long t_max = (long)p.m_tiles * p.n_tiles * p.split_k;
  const int ncu = pp_num_cus();
  const int grid = (int)(t_max < ncu ? t_max : ncu);           // one persistent workgroup per CU (153 KiB of LDS each)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) dev = 0;
  static bool attr_set[2][16] = {{false}};
  const int ei = epi == MODE_EPI_SWIGLU;
  auto kern = ei ? gemm_w4_kernel<MODE_EPI_SWIGLU> : gemm_w4_kernel<MODE_EPI_NONE>;
  if (!attr_set[ei][dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_TOTAL);
    if (e != hipSuccess) return (int)e;
    attr_set[ei][dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), w4::LDS_TOTAL, s, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

}  // namespace mode
