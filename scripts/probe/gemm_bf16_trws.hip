// Expert weight gradient dW[M, N] = dY[K, M]^T X[K, N] with the AdamW update in its epilogue (ModeAdamWFuse, include/mode_hip.h) as ONE persistent,
// WAVE-SPECIALISED workgroup per CU - a round-6 PROBE, measured and NOT shipped (profiles/r06_fused_adamw_ws.txt; build: scripts/probe/build_trws_variant.sh,
// which links this file into a variant of libmode_hip.so through the weak hook in csrc/gemm_bf16_tr.hip):
//
//   waves 0-3  "GEMM group"    the 128 x 128 fp32 gradient tile of tile i + 1: bf16 MFMA over K-steps of 32 rows, operands as they lie in memory through a
//                              four-slot LDS-DMA ring (three K-steps in flight, counted vmcnt, running ahead ACROSS tile boundaries), fragments by the LDS
//                              transpose read; the finished accumulators go to a 64-KiB fp32 tile image in LDS, one 64-row half per wave row
//   waves 4-7  "stream group"  tile i: p / m / v of the tile's parameters (non-temporal, 12 x 16 B per thread in flight), adamw_update_f on the gradient
//                              read from the LDS image, p / m / v / bf16 shadow (/ EMA) written back - 26 B per parameter, the gradient never reaches HBM
//
// Why: the ring kernel's launch cost was "epilogue + ~65 % of the GEMM" (profiles/r05_fused_adamw.txt: 217 us for dW1 against 159 us of streaming) - three
// workgroups per CU fall into lockstep, all in their K loops or all streaming, and a K loop with one exposed round trip per step crawls while HBM is
// saturated.  Here a CU always has exactly one group of each kind at work: the launch costs max(streaming, GEMM), and the GEMM group (~7 us of work per
// 20-us tile) has the slack to absorb the loaded L2 latency.
//
// No s_barrier inside the tile loop (it would couple the two groups): the groups and the GEMM group's four waves synchronise through monotonic LDS counters
//   full        += 1 per GEMM wave and K-step: "my DMA pieces of global step G have landed AND I am done reading step G - 1" - one spin per step gives
//                  both the read-after-DMA and the slot-reuse guarantee (slot of step G - 1 is refilled right after the spin of step G)
//   c_full[h]   += 1 per GEMM wave of wave row h and tile: half h of the tile image holds tile i      (stream group waits for 2 (i + 1))
//   c_empty[h]  += 1 per stream wave and tile: half h has been consumed                               (GEMM wave row h waits for 4 i before writing tile i)
// Every LDS access of the GEMM group is inline asm: the compiler would otherwise drain the LDS-DMA queue (vmcnt(0)) in front of each visible LDS access.
// Each parameter element is updated exactly once, by one thread, with the arithmetic of adamw_kernel (adamw_update_f): results are bit-identical to the
// two-pass update and to the ring kernel whatever the schedule.  The W operand's row gather (w_rows: the dispatch permutation) is cached in LDS once per
// launch (<= 4096 rows), so index loads never enter the DMA queue's vmcnt accounting.
#include "mode_common.h"
#include "gemm_tr_common.h"
#include <algorithm>
#include <cstring>
#include <type_traits>

namespace mode {

namespace {
constexpr int WS_BK = 32;
constexpr int WS_OP = WS_BK * 256;            // 8 KiB: a [32 k][128] bf16 operand tile, 256-byte rows
constexpr int WS_STAGE = 2 * WS_OP;           // A | W
constexpr int WS_C = 128 * 512;               // 64 KiB fp32 tile image
constexpr int WS_IDX_CAP = 4032;              // cached w_rows entries (15.75 KiB: with a five-slot ring the workgroup owns 163 776 of the CU's 163 840 bytes)
constexpr int WS_FLAG_BYTES = 64;
constexpr int WS_KOFF_CAP = 32;               // cached k_group_offsets entries (groups + 1)
constexpr int ws_ring(int ns) { return ns * WS_STAGE; }
constexpr int ws_lds(int ns) { return ws_ring(ns) + WS_C + WS_IDX_CAP * 4 + WS_FLAG_BYTES + WS_KOFF_CAP * 4; }
static_assert(ws_lds(5) <= 163840, "LDS budget");
enum { F_FULL = 0, F_CFULL0 = 1, F_CFULL1 = 2, F_CEMPTY0 = 3, F_CEMPTY1 = 4, F_RED = 8 };

__device__ __attribute__((aligned(256))) uint16_t g_ws_zero_row[128];

__device__ __forceinline__ void ws_signal(uint32_t addr) {
  const uint32_t one = 1;
  asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(one) : "memory");
}
__device__ __forceinline__ uint32_t ws_peek(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
template <bool SLEEP>
__device__ __forceinline__ void ws_wait_ge(uint32_t addr, uint32_t target) {
  while ((int32_t)(ws_peek(addr) - target) < 0) {
    if (SLEEP) __builtin_amdgcn_s_sleep(4);
  }
}
__device__ __forceinline__ void ws_write128(uint32_t addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

struct WsTile { int z, mt, nt, kb, ke, nk; };
}  // namespace

template <int WS_NS, int UB>
__global__ __launch_bounds__(512, 2) void gemm_tr_adamw_ws_kernel(const TrParams p, const int tiles_per_group, const int total_tiles, const int num_groups, const int gsq_slots, const int dbg,
                                                                  const uint16_t* __restrict__ zrow) {   // zrow = &g_ws_zero_row (a kernel argument: referenced by name, the symbol's address is re-loaded through the GOT inside the K loop)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WS_RING = ws_ring(WS_NS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t c0 = lds0 + WS_RING, idx0 = c0 + WS_C, fl0 = idx0 + WS_IDX_CAP * 4;
  const int t_first = xcd_remap(blockIdx.x, gridDim.x), t_step = gridDim.x;

  // ---- launch prologue (all 512 threads): flags = 0, the W-row gather table into LDS
  if (tid < WS_FLAG_BYTES / 4) reinterpret_cast<uint32_t*>(smem + WS_RING + WS_C + WS_IDX_CAP * 4)[tid] = 0;
  if (p.w_rows) {
    int* ix = reinterpret_cast<int*>(smem + WS_RING + WS_C);
    for (int i = tid; i < p.K; i += 512) ix[i] = p.w_rows[i];
  }
  // (the K-group table too: a global load of it inside the tile loop is a VMEM operation - the compiler cannot prove the table read-only next to the
  //  kernel's stores - and waiting for it would drain the DMA queue once per tile)
  const uint32_t kof0 = fl0 + WS_FLAG_BYTES;
  if (tid <= num_groups) reinterpret_cast<int*>(smem + WS_RING + WS_C + WS_IDX_CAP * 4 + WS_FLAG_BYTES)[tid] = p.koffs ? p.koffs[tid] : (tid ? p.K : 0);
  __syncthreads();

  auto tile_of = [&](int t) {
    WsTile w;
    w.z = t / tiles_per_group;
    const int r = t - w.z * tiles_per_group;
    w.mt = r / p.n_tiles; w.nt = r - w.mt * p.n_tiles;
    w.kb = (int)ws_peek(kof0 + 4 * w.z); w.ke = (int)ws_peek(kof0 + 4 * w.z + 4);
    w.nk = (w.ke - w.kb + WS_BK - 1) / WS_BK;
    return w;
  };

  float gs2 = 0.f;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // (probe, dbg == 4) cycles per phase of the K loop, wave 0 of workgroup 0
  if (wave < 4) {
    // =========================================================================================== GEMM group
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int kra = lane >> 4, pca = lane & 15;                 // DMA: row within a 4-row piece, 16-byte chunk of the 256-byte row
    // ---- DMA cursor: (tile, K-step) of the next step to issue; runs WS_NS - 1 steps ahead of the compute cursor, across tile boundaries.  Past the last
    //      tile it issues zero-row fills into the (free) slot instead: every iteration then issues exactly four DMA instructions per wave and the
    //      landed-wait is one constant vmcnt
    int d_t = t_first, d_kt = 0;
    WsTile dT;
    dT.kb = dT.ke = dT.nk = 0; dT.z = dT.mt = dT.nt = 0;
    int col_w[2], row[2];                                       // W column of this lane's chunk; K row of this lane in piece q of the NEXT step
    const uint16_t* a_ptr[2];                                   // A source of that row (advanced by 32 rows per step)
    auto dma_enter = [&]() {                                    // first non-empty tile at or after d_t
      while (d_t < total_tiles) {
        dT = tile_of(d_t);
        if (dT.nk > 0) break;
        d_t += t_step;
      }
      if (d_t < total_tiles) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int P = wave * 2 + q;
          const int c = pca ^ (kn_swz<128>(P * 4 + kra) << 1);
          row[q] = dT.kb + P * 4 + kra;
          a_ptr[q] = p.A + (long)row[q] * p.lda + min(dT.mt * 128 + c * 8, p.M - 8);   // clamped: columns past M are never stored
          col_w[q] = min(dT.nt * 128 + c * 8, p.N - 8);
        }
      }
      d_kt = 0;
    };
    dma_enter();
    int idxr[2] = {0, 0};
    auto idx_fetch = [&]() {                                    // gathered W rows of the next step: LDS table reads WITHOUT a wait (covered by the fragment wait)
      if (p.w_rows) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t a = idx0 + 4u * (uint32_t)max(min(row[q], dT.ke - 1), 0);
          asm volatile("ds_read_b32 %0, %1" : "=v"(idxr[q]) : "v"(a) : "memory");
        }
      }
    };
    auto issue = [&](int slot) {                                // four DMA instructions, unconditionally (see above)
      char* base = smem + slot * WS_STAGE;
      if (dbg == 2) return;                                     // probe: no operand traffic at all
      const bool live = d_t < total_tiles;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int P = wave * 2 + q;
        const uint16_t* src = (live && row[q] < dT.ke) ? a_ptr[q] : zrow;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(base + P * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int P = wave * 2 + q;
        const int rw = p.w_rows ? idxr[q] : min(row[q], dT.ke - 1);
        const uint16_t* src = (!live || rw < 0) ? zrow + pca * 8 : p.W + (long)rw * p.ldw + col_w[q];   // a negative index = a zero row
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(base + WS_OP + P * 1024), 16, 0, 0);
      }
    };
    auto advance = [&]() {
      if (d_t < total_tiles) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { row[q] += WS_BK; a_ptr[q] += (long)WS_BK * p.lda; }
        if (++d_kt == dT.nk) { d_t += t_step; dma_enter(); }
      }
    };
    // ---- fragment addresses (transpose reads; see gemm_bf16_tr.hip): row fq*8 + (fr>>2) [+4], 16-column tile T of the wave
    const int fsw = (fr >> 2) | ((fq & 1) << 2);
    uint32_t tr_a[4], tr_w[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      tr_a[t] = lds0 + (fq * 8 + (fr >> 2)) * 256 + (fr & 1) * 8 + ((((((wm * 4 + t) ^ fsw) << 1) | ((fr >> 1) & 1))) << 4);
      tr_w[t] = lds0 + WS_OP + (fq * 8 + (fr >> 2)) * 256 + (fr & 1) * 8 + ((((((wn * 4 + t) ^ fsw) << 1) | ((fr >> 1) & 1))) << 4);
    }
    const uint32_t f_full = fl0 + 4 * F_FULL, f_cfull = fl0 + 4 * (F_CFULL0 + wm), f_cempty = fl0 + 4 * (F_CEMPTY0 + wm);

    // ---- the K loop, software-pipelined for ONE wave per SIMD (no second wave hides this one's latencies): the fragments of step G + 1 are read - and the
    //      flag round trip is paid - under the MFMAs of step G.  Iteration G (global step counter; step G sits in ring slot G % WS_NS; fragment buffer G & 1):
    //        wait: my DMA pieces of step G + 1 have landed; my fragment reads of step G have returned
    //        signal round G + 1 ("landed G + 1, done reading G")           round r complete <=> full == 4 (r + 1)
    //        MFMA first half of step G
    //        spin until round G + 1 is complete                            -> step G + 1 readable, slot of step G free
    //        read the fragments of step G + 1 into the other buffer (no wait), fetch the gather indices of step G + WS_NS
    //        MFMA second half of step G
    //        issue the DMA of step G + WS_NS into slot G % WS_NS (four instructions per wave, zero-row fills past the last tile: one constant vmcnt)
    s16x4 alo[2][4], ahi[2][4], wlo[2][4], whi[2][4];
    f32x4 acc[4][4];
    auto zero_acc = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    int seq = 0, ct = t_first, kt = 0;                          // compute cursor: tile ordinal, tile, K-step inside it
    WsTile cT = tile_of(ct);
    auto write_tile = [&]() {                                   // accumulators -> this wave row's half of the tile image, once the stream group has consumed the previous tile's
      ws_wait_ge<true>(f_cempty, 4u * (uint32_t)seq);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = wm * 64 + i * 16 + fr;
        const uint32_t crow = c0 + rl * 512;
        const int rsw = rl & 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nl = wn * 64 + j * 16 + fq * 4;
          ws_write128(crow + ((((nl >> 2) ^ rsw)) << 4), acc[i][j]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) ws_signal(f_cfull);
      zero_acc();
      ++seq; ct += t_step; kt = 0;
      if (ct < total_tiles) cT = tile_of(ct);
    };
    auto read_frags = [&](auto PARC, uint32_t so) {
      constexpr int b = decltype(PARC)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) { lds_tr64<0>(alo[b][i], tr_a[i] + so); lds_tr64<1024>(ahi[b][i], tr_a[i] + so); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { lds_tr64<0>(wlo[b][j], tr_w[j] + so); lds_tr64<1024>(whi[b][j], tr_w[j] + so); }
    };
    auto mma_rows = [&](auto PARC, auto I0C) {
      constexpr int b = decltype(PARC)::value, i0 = decltype(I0C)::value;
#pragma unroll
      for (int i = i0; i < i0 + 2; ++i) {
        const bf16x8 a = join8(alo[b][i], ahi[b][i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(wlo[b][j], whi[b][j]), a, acc[i][j], 0, 0, 0);   // swapped: D[n][m]
      }
    };
    int G = 0;
    // prologue: every slot filled (steps 0 .. WS_NS - 1), round 0, fragments of step 0
#pragma unroll 1
    for (int s = 0; s < WS_NS; ++s) {
      idx_fetch();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(idxr[0]), "+v"(idxr[1])::"memory");
      issue(s);
      advance();
    }
    zero_acc();
    tr_wait_vmcnt<4 * (WS_NS - 1)>();
    if (lane == 0) ws_signal(f_full);
    ws_wait_ge<false>(f_full, 4u);
    read_frags(std::integral_constant<int, 0>{}, 0u);
    while (ct < total_tiles && cT.nk == 0) write_tile();          // leading tiles with an empty K range: a zero gradient (decay and moment decay still apply)
    auto step = [&](auto PARC) {
      constexpr int par = decltype(PARC)::value;
      unsigned long long tq[8];
#define WS_STAMP(k) if (dbg == 4) { tq[k] = __builtin_amdgcn_s_memtime(); }
      WS_STAMP(0)
      tr_wait_vmcnt<4 * (WS_NS - 2)>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      WS_STAMP(1)
      if (lane == 0) ws_signal(f_full);
      mma_rows(PARC, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      WS_STAMP(2)
      ws_wait_ge<false>(f_full, 4u * (uint32_t)(G + 2));
      WS_STAMP(3)
      idx_fetch();
      read_frags(std::integral_constant<int, 1 - par>{}, (uint32_t)((G + 1) % WS_NS) * WS_STAGE);
      __builtin_amdgcn_sched_barrier(0);
      WS_STAMP(4)
      mma_rows(PARC, std::integral_constant<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
      WS_STAMP(5)
      {                                                       // every LDS read above has returned: the statement names each destination (no compiler copy of
        constexpr int nb = 1 - par;                             // a fragment register may sit between its read and this wait; cdna_hip_programming.md 5.7)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(idxr[0]), "+v"(idxr[1]), "+v"(alo[nb][0]), "+v"(alo[nb][1]), "+v"(alo[nb][2]), "+v"(alo[nb][3]), "+v"(ahi[nb][0]), "+v"(ahi[nb][1]),
                       "+v"(ahi[nb][2]), "+v"(ahi[nb][3]), "+v"(wlo[nb][0]), "+v"(wlo[nb][1]), "+v"(wlo[nb][2]), "+v"(wlo[nb][3]), "+v"(whi[nb][0]),
                       "+v"(whi[nb][1]), "+v"(whi[nb][2]), "+v"(whi[nb][3])::"memory");
      }
      WS_STAMP(6)
      issue(G % WS_NS);
      advance();
      WS_STAMP(7)
      if (dbg == 4) {
#pragma unroll
        for (int k = 0; k < 7; ++k) ph[k] += tq[k + 1] - tq[k];
        ph[7] += 1;
      }
#undef WS_STAMP
      ++G;
      if (++kt == cT.nk) {
        write_tile();
        while (ct < total_tiles && cT.nk == 0) write_tile();
      }
    };
#pragma unroll 1
    while (ct < total_tiles) {
      step(std::integral_constant<int, 0>{});
      if (ct >= total_tiles) break;
      step(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  } else {
    // =========================================================================================== stream group
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const int stid = tid - 256;
    // UB chunks per batch: 3 UB global loads of 16 B per thread in flight
    int seq = 0;
#pragma unroll 1
    for (int t = t_first; t < total_tiles; t += t_step, ++seq) {
      const int z = t / tiles_per_group;
      const int r = t - z * tiles_per_group;
      const int mt = r / p.n_tiles, nt = r - mt * p.n_tiles;
      const int row0 = mt * 128, n0 = nt * 128;
      const int rows_valid = min(p.M, row0 + 128) - row0;
      const long gofs = (long)z * p.c_gstride;
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        bool waited = false;
#pragma unroll 1
        for (int u0 = 0; u0 < 8; u0 += UB) {
          f4 P[UB], M[UB], V[UB], G[UB];
          // element offset of chunk u0 + u of this thread: a wave covers two whole 512-byte rows per chunk index (rows past the tile clamp to the tile's
          // first element: every load is unconditional - one round trip for all of them)
          auto where = [&](int u, long& eo, int& rl, int& ch) -> bool {
            const int c = stid + (u0 + u) * 256;
            rl = c >> 5; ch = c & 31;
            const int ml = h * 64 + rl;
            const bool ok = ml < rows_valid;
            eo = gofs + (long)(row0 + (ok ? ml : 0)) * p.ldc + n0 + (ok ? ch * 4 : 0);
            return ok;
          };
          if (dbg >= 1) {                                       // probe: the GEMM group alone (the stream group only hands the halves back)
            if (!waited) { ws_wait_ge<true>(fl0 + 4 * (F_CFULL0 + h), 2u * (uint32_t)(seq + 1)); waited = true; }
            continue;
          }
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            long eo; int rl, ch;
            where(u, eo, rl, ch);
            P[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.ad_p + eo));
            M[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.ad_m + eo));
            V[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.ad_v + eo));
          }
          if (!waited) {                                        // the first batch's parameter loads are in flight while this half's gradients arrive
            ws_wait_ge<true>(fl0 + 4 * (F_CFULL0 + h), 2u * (uint32_t)(seq + 1));
            waited = true;
          }
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            long eo; int rl, ch;
            where(u, eo, rl, ch);
            const int ml = h * 64 + rl;
            G[u] = *reinterpret_cast<const f4*>(smem + WS_RING + ml * 512 + ((ch ^ (ml & 15)) << 4));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            long eo; int rl, ch;
            if (!where(u, eo, rl, ch)) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float gr = __fmul_rn(G[u][j], p.ad_gscale);
              gs2 = __builtin_fmaf(gr, gr, gs2);
              float w = P[u][j], m_ = M[u][j], v_ = V[u][j];
              adamw_update_f(w, m_, v_, gr, p.ad_decay, p.ad_b1, p.ad_b2, p.ad_eps, p.ad_step_size, p.ad_inv_bc2_sqrt);
              P[u][j] = w; M[u][j] = m_; V[u][j] = v_;
            }
            __builtin_nontemporal_store(P[u], reinterpret_cast<f4*>(p.ad_p + eo));
            __builtin_nontemporal_store(M[u], reinterpret_cast<f4*>(p.ad_m + eo));
            __builtin_nontemporal_store(V[u], reinterpret_cast<f4*>(p.ad_v + eo));
            if (p.ad_ema) {
              f4 Ev = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.ad_ema + eo));
#pragma unroll
              for (int j = 0; j < 4; ++j) Ev[j] = Ev[j] - p.ad_ema_rate * (Ev[j] - P[u][j]);
              __builtin_nontemporal_store(Ev, reinterpret_cast<f4*>(p.ad_ema + eo));
            }
            if (p.ad_lp) {
              u2 o; o[0] = pack_bf16x2(P[u][0], P[u][1]); o[1] = pack_bf16x2(P[u][2], P[u][3]);
              *reinterpret_cast<u2*>(p.ad_lp + eo) = o;          // the bf16 shadow is re-read by the next forward: a normal (cached) store
            }
          }
        }
        // every gradient chunk of this half has been consumed (the update above used them): hand the half back to the GEMM group
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) ws_signal(fl0 + 4 * (F_CEMPTY0 + h));
      }
    }
  }
  // ---- ||g||^2 of everything this workgroup updated: wave butterfly, the four stream waves through LDS in a fixed order
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem + WS_RING + WS_C + WS_IDX_CAP * 4) + F_RED;
  if (p.ad_gsq) {
    if (wave >= 4) {
      gs2 = wave_sum(gs2);
      if (lane == 0) red[wave - 4] = gs2;
    }
    __syncthreads();
    if (tid == 0) p.ad_gsq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (blockIdx.x == 0)
      for (int i = gridDim.x + tid; i < gsq_slots; i += 512) p.ad_gsq[i] = 0.f;    // the slots of the ring kernel's per-tile layout that this launch does not use
    if (dbg == 4 && blockIdx.x == 0) {
      __syncthreads();
      if (tid == 0)
        for (int k = 0; k < 8; ++k) p.ad_gsq[gridDim.x + k] = (float)ph[k];
    }
  }
}

int g_adamw_ws = 1;   // "adamw_ws" option: 1 = the wave-specialised kernel for every shape it takes (default), 0 = the ring kernel gemm_tr_kernel<.., EPI = 2>
int g_adamw_ws_dbg = 0;   // "adamw_ws_dbg" option (probing only): 1 = the stream group moves no data
int g_adamw_ws_cfg = 0;   // "adamw_ws_cfg" option (probing): 10 * ring slots + chunks per stream batch; 0 = default

template <int NS, int UB>
static int ws_launch(const TrParams& p, int grid, int tpg, int total, int groups, int gsq_slots, hipStream_t s) {
  auto kern = gemm_tr_adamw_ws_kernel<NS, UB>;
  static LdsLimitOnce lds_once;
  const int rc = lds_once.ensure(reinterpret_cast<const void*>(kern), ws_lds(NS));
  if (rc != MODE_OK) return rc;
  static std::atomic<void*> zcache[kMaxDevices];            // the zero row's address on each device (looked up once)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return MODE_ERR_UNSUPPORTED;
  void* zrow = zcache[dev].load(std::memory_order_acquire);
  if (!zrow) {
    if (hipGetSymbolAddress(&zrow, HIP_SYMBOL(g_ws_zero_row)) != hipSuccess || !zrow) return MODE_ERR_UNSUPPORTED;
    zcache[dev].store(zrow, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ws_lds(NS), s, p, tpg, total, groups, gsq_slots, g_adamw_ws_dbg, (const uint16_t*)zrow);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

// `p` as tr_adamw_launch (gemm_bf16_tr.hip) fills it.  MODE_ERR_UNSUPPORTED = not this kernel's shape (the caller falls back to the ring kernel).
int gemm_bf16_trws_adamw_launch(TrParams p, const ModeGemmDesc* d, long gsq_slots, hipStream_t s) {
  if (!g_adamw_ws || d->N % 128 || d->M % 8 || d->M < 8) return MODE_ERR_UNSUPPORTED;
  if (d->w_rows && d->K > WS_IDX_CAP) return MODE_ERR_UNSUPPORTED;
  const int groups = d->k_group_offsets ? d->num_k_groups : 1;
  if (groups + 1 > WS_KOFF_CAP) return MODE_ERR_UNSUPPORTED;
  p.m_tiles = (d->M + 127) / 128; p.n_tiles = d->N / 128;
  const long tpg = (long)p.m_tiles * p.n_tiles, total = tpg * groups;
  if (total <= 0 || total > (1L << 30)) return MODE_ERR_UNSUPPORTED;
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return MODE_ERR_UNSUPPORTED;
  const int grid = (int)std::min<long>(total, ncu);
  if (p.ad_gsq && gsq_slots < grid) return MODE_ERR_WORKSPACE;
  switch (g_adamw_ws_cfg) {
    case 42: return ws_launch<4, 2>(p, grid, (int)tpg, (int)total, groups, (int)gsq_slots, s);
    case 44: return ws_launch<4, 4>(p, grid, (int)tpg, (int)total, groups, (int)gsq_slots, s);
    case 52: return ws_launch<5, 2>(p, grid, (int)tpg, (int)total, groups, (int)gsq_slots, s);
    default: return ws_launch<5, 4>(p, grid, (int)tpg, (int)total, groups, (int)gsq_slots, s);
  }
}

}  // namespace mode

extern "C" int mode_trws_set_option(const char* key, int value) {                 // reached through mode_set_option (weak hook in csrc/dit.hip)
  if (!strcmp(key, "adamw_ws")) { mode::g_adamw_ws = value != 0; return MODE_OK; }
  if (!strcmp(key, "adamw_ws_cfg")) { mode::g_adamw_ws_cfg = value; return MODE_OK; }
  if (!strcmp(key, "adamw_ws_dbg")) { mode::g_adamw_ws_dbg = value; return MODE_OK; }
  return MODE_ERR_UNSUPPORTED;
}
