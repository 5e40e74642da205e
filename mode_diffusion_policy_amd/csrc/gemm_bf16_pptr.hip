// Persistent ping-pong bf16 MFMA GEMM for the BACKWARD pass's large products on gfx950 - the loop structure of gemm_bf16_pp.hip (256 x 256 output
// tile, BK = 64, 8 wave64 as 2 x 4, two K-steps of four 16-KiB half-tiles resident in LDS, two wave rows staggered by one barrier, counted vmcnt
// waits, one workgroup per CU walking its tiles with an operand stream that never stops) fed by the operand layouts of gemm_bf16_tr.hip:
//
//   data gradient    dX[M,N] = dY[M,K] @ W[K,N]        (MODE_GEMM_W_KN)           A rows K-contiguous as in the forward, W = nn.Linear's [out,in] weight
//                                                                                as it lies in memory; rows grouped by expert (ragged segments), K-slices
//   weight gradient  dW[M,N] = dY[K,M]^T @ X[K,N]      (MODE_GEMM_W_KN | A_KM)    both operands row-major activations; K = arbitrary row ranges per group
//                                                                                (expert segments of the sorted dispatch order), rows past a range masked
//
// An operand whose reduction index is its ROW index is DMA'd as [64 k][128 cols] half-tiles (256-byte rows, 32-byte column groups XOR-swizzled by
// f(k) - the image of gemm_bf16_tr.hip) and its MFMA fragments are gathered by `ds_read_b64_tr_b16` (two reads = the 8 consecutive k of one column).
// Same bytes through LDS per flop as the forward ping-pong kernel, twice the LDS read instructions for such an operand.
//
// Why: the four expert GEMMs of the training backward (dH = dY W2, dW2 = dY^T H, dU = dP W1, dW1 = dP^T u: 180 GF per layer at C2 / B = 128) ran on
// the 128 x 128 one-barrier ring of gemm_bf16_tr.hip at 540-640 TF/s; the forward's ping-pong structure reaches 1 050-1 130 TF/s on the same shapes.
//
// Accumulators use the swapped MFMA operands of the forward kernels (D[weight column][token]): a lane owns 4 consecutive output columns of one row per
// 16-column fragment.  No bias / activation epilogues: backward GEMMs have none.
#include "mode_common.h"
#include <type_traits>

namespace mode {

__device__ __attribute__((aligned(256))) uint16_t g_pptr_zero_row[128];      // 256 B of zeros: DMA source of masked K rows (weight gradient)

struct PpTrParams {
  const uint16_t* A; long lda;
  const uint16_t* W; long ldw; long w_estride;
  void* C; long ldc;
  const int* offsets; int E;              // data gradient, grouped rows: expert e owns rows [offsets[e], offsets[e+1])
  const int* koffs; int G; long c_gstride;   // weight gradient: group z reduces over rows [koffs[z], koffs[z+1]) and writes C + z * c_gstride
  int M, N, K, n_tiles;
  int split_k; long split_stride;         // data gradient: K-slices, partial sums to C + slice * split_stride
};

namespace pptr {
constexpr int BKK = 64;
constexpr int HALF_BYTES = 128 * BKK * 2;                  // one half-tile (128 rows x 64 k, or 64 k x 128 columns): 16 KiB
constexpr int LDS_A = 0;                                   // A[t][h] at (t*2+h) * 16 KiB
constexpr int LDS_B = 4 * HALF_BYTES;                      // W[t][h] at 64 KiB + (t*2+h) * 16 KiB
constexpr int LDS_TOTAL = 8 * HALF_BYTES;                  // 128 KiB
constexpr int GM = 8;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ void lds_read128(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_tr64(s16x4& dst, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
template <int V>
using IC = std::integral_constant<int, V>;
}  // namespace pptr

template <bool A_KM, bool OUT_BF16>
__global__ __launch_bounds__(512, 2) void gemm_pptr_kernel(const PpTrParams p) {
  using namespace pptr;
  constexpr int BM = 256, NOUT = 256;
  constexpr int ESZ = OUT_BF16 ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---------------------------------------------------------------------------------------------------- tile space (all scalar)
  int o[9];
  int m_real;
  if (!A_KM && p.offsets) {
#pragma unroll
    for (int e = 0; e < 9; ++e) o[e] = p.offsets[min(e, p.E)];
    m_real = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (e < p.E) m_real += (o[e + 1] - o[e] + BM - 1) / BM;
  } else {
    m_real = (p.M + BM - 1) / BM;
  }
  const int S = A_KM ? 1 : p.split_k, n_tiles = p.n_tiles;
  const int groups = A_KM ? p.G : 1;
  const int T = m_real * n_tiles * S * groups;
  const int G = gridDim.x;
  const int R = (T + G - 1) / G;                               // tiles per workgroup
  if (R == 0) return;
  const int RN = (n_tiles % R == 0) ? R : 1;
  const int nwg = (T + R - 1) / R;
  if ((int)blockIdx.x >= nwg) return;
  const int wg = xcd_remap(blockIdx.x, nwg);
  int L = wg * R;
  const int Lend = min(T, L + R);

  struct Tile { int m, n, slice, row0, row_end, expert, kb, ke, nk; };
  auto map_tile = [&](int l, Tile& t) {
    if constexpr (A_KM) {
      // group-major, then m, then n: a workgroup's consecutive tiles share the A columns (and the K range)
      const int per_g = m_real * n_tiles;
      const int g = l / per_g, rem = l - g * per_g;
      t.m = rem / n_tiles; t.n = rem - t.m * n_tiles;
      t.slice = 0; t.expert = g; t.row0 = t.m * BM; t.row_end = t.row0 + BM;
      t.kb = p.koffs ? p.koffs[g] : 0;
      t.ke = p.koffs ? p.koffs[g + 1] : p.K;
      const int steps = (t.ke - t.kb + BKK - 1) / BKK;
      t.nk = max(2, (steps + 1) & ~1);                          // the loop runs K-step PAIRS; rows past the range are masked
    } else {
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
      t.expert = 0;
      t.nk = p.K / BKK / S;
      t.kb = t.slice * t.nk * BKK; t.ke = t.kb + t.nk * BKK;
      if (p.offsets) {
        int tt = t.m;
        bool found = false;
        t.row0 = 0; t.row_end = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (!found && e < p.E) {
            const int nt_e = (o[e + 1] - o[e] + BM - 1) / BM;
            if (tt < nt_e) { t.row0 = o[e] + tt * BM; t.row_end = min(o[e + 1], t.row0 + BM); t.expert = e; found = true; }
            else tt -= nt_e;
          }
        }
      } else {
        t.row0 = t.m * BM; t.row_end = min(p.M, t.row0 + BM);
      }
    }
  };

  // ---------------------------------------------------------------------------------------------------- per-lane constants
  // [64 k][128 cols] half-tile: a 1-KiB DMA piece = 4 k-rows x 256 B; lane -> k-row (lane >> 4) of the piece, physical 16-B chunk (lane & 15) which
  // holds the LOGICAL chunk (lane & 15) ^ (f(k) << 1), f(k) = (k & 3) | ((k >> 3) & 1) << 2.  Wave w fills pieces 2w, 2w+1: k = (2w+q)*4 + (lane >> 4),
  // so f(k) = (lane >> 4) | (w & 1) << 2 for both pieces.
  const int kra = lane >> 4;
  const int kn_cb = ((lane & 15) ^ ((kra | ((wave & 1) << 2)) << 1)) * 16;      // byte offset of this lane's logical chunk inside a 256-B row
  const int krow0 = wave * 8 + kra;                                             // k-row of piece q = krow0 + 4 q
  // W operand, data gradient: tile-independent 32-bit byte offsets from a half-tile's (k0, col0) corner
  [[maybe_unused]] uint32_t b_off[2];
  if constexpr (!A_KM) {
#pragma unroll
    for (int q = 0; q < 2; ++q) b_off[q] = (uint32_t)((long)(krow0 + 4 * q) * p.ldw * 2 + kn_cb);
  }
  // A operand of the data gradient: [128 rows][64 k] half-tiles as in gemm_bf16_pp.hip (8-row pieces, chunk ^ (row & 7))
  const int r8 = lane >> 3, lchunk = (lane & 7) ^ r8;
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // transpose-read addresses: lane (fr, fq) addresses 4 elements of k-row fq*8 + (fr >> 2), columns T*16 + (fr & 3)*4 of 16-column tile T
  const int fsw = (fr >> 2) | ((fq & 1) << 2);
  auto tr_addr = [&](int base, int T_) {
    return lds0 + base + (fq * 8 + (fr >> 2)) * 256 + (fr & 1) * 8 + ((((T_ ^ fsw) << 1) | ((fr >> 1) & 1)) << 4);
  };
  uint32_t b_tr[2];
  b_tr[0] = tr_addr(LDS_B, wc * 2 + 0);
  b_tr[1] = tr_addr(LDS_B, wc * 2 + 1);
  [[maybe_unused]] uint32_t a_tr[4];
  [[maybe_unused]] uint32_t a_addr[2];
  if constexpr (A_KM) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a_tr[i] = tr_addr(LDS_A, wr * 4 + i);
  } else {
    const int c0 = (fq ^ (fr & 7)) * 16;
    a_addr[0] = lds0 + LDS_A + (wr * 64 + fr) * 128 + c0;
    a_addr[1] = lds0 + LDS_A + (wr * 64 + fr) * 128 + (c0 ^ 64);
  }

  // fragments: W always through transpose reads (lo = k..k+3, hi = k+4..k+7); A by transpose reads (weight gradient) or ds_read_b128
  s16x4 Blo[2][4], Bhi[2][4];                                  // [W half][k-half*2 + j]
  [[maybe_unused]] s16x4 Alo[8], Ahi[8];                        // [k-half*4 + i]
  [[maybe_unused]] bf16x8 A_[8];
  f32x4 acc[2][2][4][2];                                       // [A half][W half][i][j]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto rdA = [&](auto T_, auto H_) __attribute__((always_inline)) {
    constexpr int t = decltype(T_)::value, h = decltype(H_)::value;
    constexpr int base = (t * 2 + h) * HALF_BYTES;
    if constexpr (A_KM) {
      lds_tr64<base>(Alo[0], a_tr[0]); lds_tr64<base + 1024>(Ahi[0], a_tr[0]);
      lds_tr64<base>(Alo[1], a_tr[1]); lds_tr64<base + 1024>(Ahi[1], a_tr[1]);
      lds_tr64<base>(Alo[2], a_tr[2]); lds_tr64<base + 1024>(Ahi[2], a_tr[2]);
      lds_tr64<base>(Alo[3], a_tr[3]); lds_tr64<base + 1024>(Ahi[3], a_tr[3]);
      lds_tr64<base + 8192>(Alo[4], a_tr[0]); lds_tr64<base + 9216>(Ahi[4], a_tr[0]);
      lds_tr64<base + 8192>(Alo[5], a_tr[1]); lds_tr64<base + 9216>(Ahi[5], a_tr[1]);
      lds_tr64<base + 8192>(Alo[6], a_tr[2]); lds_tr64<base + 9216>(Ahi[6], a_tr[2]);
      lds_tr64<base + 8192>(Alo[7], a_tr[3]); lds_tr64<base + 9216>(Ahi[7], a_tr[3]);
    } else {
      lds_read128<base>(A_[0], a_addr[0]); lds_read128<base + 2048>(A_[1], a_addr[0]);
      lds_read128<base + 4096>(A_[2], a_addr[0]); lds_read128<base + 6144>(A_[3], a_addr[0]);
      lds_read128<base>(A_[4], a_addr[1]); lds_read128<base + 2048>(A_[5], a_addr[1]);
      lds_read128<base + 4096>(A_[6], a_addr[1]); lds_read128<base + 6144>(A_[7], a_addr[1]);
    }
  };
  auto rdB = [&](auto T_, auto H_) __attribute__((always_inline)) {
    constexpr int t = decltype(T_)::value, h = decltype(H_)::value;
    constexpr int base = (t * 2 + h) * HALF_BYTES;
    lds_tr64<base>(Blo[h][0], b_tr[0]); lds_tr64<base + 1024>(Bhi[h][0], b_tr[0]);
    lds_tr64<base>(Blo[h][1], b_tr[1]); lds_tr64<base + 1024>(Bhi[h][1], b_tr[1]);
    lds_tr64<base + 8192>(Blo[h][2], b_tr[0]); lds_tr64<base + 9216>(Bhi[h][2], b_tr[0]);
    lds_tr64<base + 8192>(Blo[h][3], b_tr[1]); lds_tr64<base + 9216>(Bhi[h][3], b_tr[1]);
  };
  auto mma = [&](auto AH_, auto BH_) __attribute__((always_inline)) {
    constexpr int ah = decltype(AH_)::value, bh = decltype(BH_)::value;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x8 a;
        if constexpr (A_KM) a = join8(Alo[kh * 4 + i], Ahi[kh * 4 + i]);
        else a = A_[kh * 4 + i];
#pragma unroll
        for (int j = 0; j < 2; ++j)                              // swapped operands: D[weight column][token / dW row]
          acc[ah][bh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(Blo[bh][kh * 2 + j], Bhi[bh][kh * 2 + j]), a, acc[ah][bh][i][j], 0, 0, 0);
      }
  };
  // DMA of one half-tile: two 1-KiB pieces per wave (pieces 2w, 2w+1), uniform base + per-lane 32-bit offset
  auto stage = [&](auto OP_, auto T_, auto H_, const char* g, uint32_t o0, uint32_t o1) __attribute__((always_inline)) {
    constexpr int op = decltype(OP_)::value, t = decltype(T_)::value, h = decltype(H_)::value;
    constexpr int base = (op ? LDS_B : LDS_A) + (t * 2 + h) * HALF_BYTES;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + o0),
                                     (__attribute__((address_space(3))) void*)(smem + base + (wave * 2 + 0) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + o1),
                                     (__attribute__((address_space(3))) void*)(smem + base + (wave * 2 + 1) * 1024), 16, 0, 0);
  };
  // weight gradient: half-tile of a [rows][cols] operand over the K rows [r0, r0 + 64) of the reduction range [.., ke): rows past the range read a
  // zero row (A: they must contribute exactly 0) / a clamped, finite row (W)
  auto stage_kn = [&](auto OP_, auto T_, auto H_, const char* colbase, long ld2, int r0, int ke) __attribute__((always_inline)) {
    constexpr int op = decltype(OP_)::value, t = decltype(T_)::value, h = decltype(H_)::value;
    constexpr int base = (op ? LDS_B : LDS_A) + (t * 2 + h) * HALF_BYTES;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = r0 + krow0 + 4 * q;
      const int rc = max(min(r, ke - 1), 0);
      const char* src = colbase + ((long)rc * ld2 + kn_cb);
      if constexpr (op == 0) {
        const char* z = reinterpret_cast<const char*>(g_pptr_zero_row) + (lane & 15) * 16;
        src = r < ke ? src : z;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + base + (wave * 2 + q) * 1024), 16, 0, 0);
    }
  };
  constexpr IC<0> _0{};
  constexpr IC<1> _1{};
#define PT_SB() __builtin_amdgcn_sched_barrier(0)
#define PT_BAR() __builtin_amdgcn_s_barrier()
  bool staggered = false;
  bool fresh = true;

  Tile cur, nxt;
  map_tile(L, cur);
  [[maybe_unused]] uint32_t a_off[2][2] = {{0, 0}, {0, 0}};      // data gradient: byte offsets of this lane's A rows from p.A, per half / piece
  const char* Ak = nullptr;                                    // A operand at the tile's K origin (data gradient) / column origin (weight gradient)
  const char* Wc = nullptr;                                    // W operand at the tile's (K origin, first column)
  const long a_kstep = A_KM ? 64 * p.lda * 2 : 128;            // bytes per K-step
  const long w_kstep = 64 * p.ldw * 2;
  const long lda2 = p.lda * 2, ldw2 = p.ldw * 2;

  auto w_tile_base = [&](const Tile& t) -> const char* {
    if constexpr (A_KM) return reinterpret_cast<const char*>(p.W) + (long)t.n * NOUT * 2;
    else return reinterpret_cast<const char*>(p.W) + ((long)t.expert * p.w_estride + (long)t.kb * p.ldw + (long)t.n * NOUT) * 2;
  };
  // stage half h of operand op for K-step k (counted from the tile's K origin) into buffer t
  auto stA = [&](auto T_, auto H_, int k) __attribute__((always_inline)) {
    constexpr int h = decltype(H_)::value;
    if constexpr (A_KM) stage_kn(_0, T_, H_, Ak + h * 256, lda2, cur.kb + k * BKK, cur.ke);
    else stage(_0, T_, H_, Ak + (long)k * 128, a_off[h][0], a_off[h][1]);
  };
  auto stW = [&](auto T_, auto H_, const char* wbase, int k) __attribute__((always_inline)) {
    constexpr int h = decltype(H_)::value;
    if constexpr (A_KM) stage_kn(_1, T_, H_, wbase + h * 256, ldw2, cur.kb + k * BKK, cur.ke);
    else stage(_1, T_, H_, wbase + (long)k * w_kstep + h * 256, b_off[0], b_off[1]);
  };

  while (true) {
    const int nk = cur.nk;
    if (fresh) {
      // ---- (re)start the operand stream for a new m-tile / K-slice / group (see gemm_bf16_pp.hip for the barrier choreography)
      wait_vmcnt<0>();
      if (staggered && wr == 0) PT_BAR();
      PT_BAR();
      Wc = w_tile_base(cur);
      if constexpr (A_KM) {
        Ak = reinterpret_cast<const char*>(p.A) + (long)cur.m * BM * 2;
      } else {
        Ak = reinterpret_cast<const char*>(p.A) + (long)cur.kb * 2;
        const int last = cur.row_end - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int srow = min(cur.row0 + h * 128 + (wave * 2 + q) * 8 + r8, last);      // rows past the segment re-read a valid row (never stored)
            a_off[h][q] = (uint32_t)(((long)srow * p.lda + lchunk * 8) * 2);
          }
      }
      stW(_0, _0, Wc, 0); stW(_0, _1, Wc, 0);
      stW(_1, _0, Wc, 1); stW(_1, _1, Wc, 1);
      stA(_0, _0, 0); stA(_0, _1, 0);
      stA(_1, _0, 1); stA(_1, _1, 1);
      wait_vmcnt<0>();
      PT_BAR();
      if (wr == 1) PT_BAR();                                     // stagger: wave row 1 runs one barrier behind wave row 0 from here on
      staggered = true;
      PT_SB();
      fresh = false;
    }
    const bool has_next = L + 1 < Lend;
    bool cont = false;
    if (has_next) {
      map_tile(L + 1, nxt);
      cont = nxt.m == cur.m && nxt.slice == cur.slice && nxt.expert == cur.expert;
    }
    const char* Wn = cont ? w_tile_base(nxt) : Wc;             // no successor on this stream: the tail re-reads valid memory, never consumed

    // ------------------------------------------------------------------------------------------------ K loop: 4 phases = 2 K-steps (gemm_bf16_pp.hip)
    //   R(P0, t): reads W0 W1 A0 of K-step t;  stages A1[t+1]           R(P1, t): reads A1[t];  stages W0 W1 A0 of K-step t+2
#define PT_COMPUTE2(AH)           \
  PT_BAR();                       \
  wait_lgkmcnt<0>();              \
  PT_SB();                        \
  __builtin_amdgcn_s_setprio(1);  \
  mma(AH, _0);                    \
  mma(AH, _1);                    \
  __builtin_amdgcn_s_setprio(0);  \
  PT_SB();                        \
  PT_BAR();                       \
  PT_SB();
    auto kpair = [&](auto FIRST_, int kt) __attribute__((always_inline)) {
      constexpr bool first = decltype(FIRST_)::value != 0;
      const bool cross = kt + 2 >= nk;
      const int k2 = cross ? kt + 2 - nk : kt + 2;             // K-step (kt+2) inside its own output tile
      const char* W2 = cross ? Wn : Wc;
      // P0 of K-step kt [buffer 0]
      rdB(_0, _0); rdB(_0, _1);
      PT_SB();
      rdA(_0, _0);
      if constexpr (!first) { stA(_1, _1, kt + 1); wait_vmcnt<8>(); }
      PT_COMPUTE2(_0)
      // P1 of K-step kt
      rdA(_0, _1);
      stW(_0, _0, W2, k2); stW(_0, _1, W2, k2);
      stA(_0, _0, k2);
      if constexpr (!first) wait_vmcnt<8>();
      PT_COMPUTE2(_1)
      // P0 of K-step kt+1 [buffer 1]
      rdB(_1, _0); rdB(_1, _1);
      PT_SB();
      rdA(_1, _0);
      stA(_0, _1, k2);
      if constexpr (!first) wait_vmcnt<8>();
      PT_COMPUTE2(_0)
      // P1 of K-step kt+1; K-step kt+2 [buffer 0] retired by this wait
      rdA(_1, _1);
      stW(_1, _0, W2, k2 + 1); stW(_1, _1, W2, k2 + 1);
      stA(_1, _0, k2 + 1);
      wait_vmcnt<8>();
      PT_COMPUTE2(_1)
    };
    kpair(_1, 0);
#pragma unroll 1
    for (int kt = 2; kt < nk; kt += 2) kpair(_0, kt);
#undef PT_COMPUTE2

    // ------------------------------------------------------------------------------------------------ epilogue: registers -> global
    // (ordering as in gemm_bf16_pp.hip: the last half-tile of the next tile's second K-step is requested, outputs are packed while every DMA still in
    // flight lands, vmcnt(0), then the stores back to back)
    {
      const int rows_valid = cur.row_end - cur.row0;
      char* Ct = reinterpret_cast<char*>(p.C) +
                 ((A_KM ? (long)cur.expert * p.c_gstride : (long)cur.slice * p.split_stride) + (long)cur.row0 * p.ldc + (long)cur.n * NOUT + wc * 32) * ESZ;
      const uint32_t c_lane = (uint32_t)(fr * (int)p.ldc + fq * 4) * ESZ;
      if (cont) stA(_1, _1, 1);                                  // next tile, K-step 1, A half 1 (its slot was last read in phase 7)
      if (wr == 0) PT_BAR();
      // one ROUND = one fragment row x one 128-column W half: this lane's 2 x 4 outputs of it (columns j*16 + fq*4 .. +3)
      auto round_urow = [&](int r) { const int fi = r >> 1, a = fi >> 2, i = fi & 3; return a * 128 + wr * 64 + i * 16; };
      auto round_col = [&](int r) { return (r & 1) * 128; };
      if constexpr (OUT_BF16) {
        u32x2 pk[16][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int fi = r >> 1, a = fi >> 2, i = fi & 3, b = r & 1;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4 v = acc[a][b][i][j];
            pk[r][j] = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(pk[r][0]), "+v"(pk[r][1]));      // every output is packed before the wait below
        wait_vmcnt<0>();
        PT_SB();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int urow = round_urow(r);
          if (fr < rows_valid - urow) {
            char* c = Ct + ((long)urow * p.ldc + round_col(r)) * 2 + c_lane;
            *reinterpret_cast<u32x2*>(c) = pk[r][0];
            *reinterpret_cast<u32x2*>(c + 32) = pk[r][1];
          }
        }
      } else {
        wait_vmcnt<0>();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int fi = r >> 1, a = fi >> 2, i = fi & 3, b = r & 1;
          const int urow = round_urow(r);
          if (fr < rows_valid - urow) {
            float* c = reinterpret_cast<float*>(Ct + ((long)urow * p.ldc + round_col(r)) * 4 + c_lane);
            const f32x4 v0 = acc[a][b][i][0], v1 = acc[a][b][i][1];
            *reinterpret_cast<float4*>(c) = make_float4(v0[0], v0[1], v0[2], v0[3]);
            *reinterpret_cast<float4*>(c + 16) = make_float4(v1[0], v1[1], v1[2], v1[3]);
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
      if (wr == 1) PT_BAR();
    }
    if (!has_next) break;
    ++L;
    cur = nxt;
    fresh = !cont;
    Wc = Wn;
  }
  wait_vmcnt<0>();                                             // the tail of the operand stream must land before the LDS is released
  if (staggered && wr == 0) PT_BAR();                          // balance the stagger barrier of wave row 1
#undef PT_BAR
#undef PT_SB
}

// ------------------------------------------------------------------------------------------------------------ host side
int pp_num_cus();   // gemm_bf16_pp.hip

template <bool A_KM, bool OUT_BF16>
static int pptr_launch(const PpTrParams& p, long t_max, hipStream_t s) {
  const int ncu = pp_num_cus();
  const int grid = (int)(t_max < ncu ? t_max : ncu);           // one persistent workgroup per CU (128 KiB of LDS each)
  auto kern = gemm_pptr_kernel<A_KM, OUT_BF16>;
  static LdsLimitOnce lds_once;
  {
    const int rc = lds_once.ensure(reinterpret_cast<const void*>(kern), pptr::LDS_TOTAL);
    if (rc != MODE_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), pptr::LDS_TOTAL, s, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

// Shape / alignment contract and the "large problems only" heuristic.  `force`: every shape the kernel supports ("gemm_tr_cfg" 6), otherwise only
// problems whose 256 x 256 tiles cover most of the part.  MODE_ERR_UNSUPPORTED sends the caller back to the 128 x 128 ring kernels.
static int pptr_plan(const ModeGemmDesc* d, bool force, PpTrParams& p, long& t_max) {
  const bool a_km = (d->flags & MODE_GEMM_A_KM) != 0;
  const int S = d->split_k > 1 ? d->split_k : 1;
  if (d->dtype != MODE_BF16 || d->epilogue != MODE_EPI_NONE || !(d->flags & MODE_GEMM_W_KN)) return MODE_ERR_UNSUPPORTED;
  if (d->w_rows || d->a_rows || d->N % 256 || d->N <= 0) return MODE_ERR_UNSUPPORTED;
  if (d->ldc % 4 || (reinterpret_cast<uintptr_t>(d->C) & 15)) return MODE_ERR_UNSUPPORTED;
  if (d->lda % 8 || d->ldw % 8 || (reinterpret_cast<uintptr_t>(d->A) & 15) || (reinterpret_cast<uintptr_t>(d->W) & 15)) return MODE_ERR_UNSUPPORTED;
  p.A = (const uint16_t*)d->A; p.lda = d->lda; p.W = (const uint16_t*)d->W; p.ldw = d->ldw; p.w_estride = d->w_expert_stride;
  p.C = d->C; p.ldc = d->ldc; p.offsets = d->expert_offsets; p.E = d->num_experts;
  p.koffs = d->k_group_offsets; p.G = d->k_group_offsets ? d->num_k_groups : 1; p.c_gstride = d->c_group_stride;
  p.M = d->M; p.N = d->N; p.K = d->K; p.n_tiles = d->N / 256;
  p.split_k = S; p.split_stride = d->split_stride;
  if (a_km) {
    if (d->M % 256 || d->M <= 0 || S > 1 || d->expert_offsets || d->K <= 0 || p.G <= 0) return MODE_ERR_UNSUPPORTED;
    if ((long)d->K * d->lda * 2 >= (1L << 31) || (long)d->K * d->ldw * 2 >= (1L << 31)) return MODE_ERR_UNSUPPORTED;
    t_max = (long)p.G * (d->M / 256) * p.n_tiles;
    if (!force && (t_max < 192 || d->K / p.G < 512)) return MODE_ERR_UNSUPPORTED;
  } else {
    if (d->K % (128 * S) || d->K / S < 128 || d->k_group_offsets || d->M <= 0) return MODE_ERR_UNSUPPORTED;
    if (d->expert_offsets && d->num_experts > 8) return MODE_ERR_UNSUPPORTED;
    if (S > 1 && d->split_stride % 4) return MODE_ERR_UNSUPPORTED;
    const long wspan = (long)d->K * d->ldw * 2;                  // 32-bit per-lane byte offsets inside one expert's weight / the A operand
    if (wspan >= (1L << 32) || (long)d->M * d->lda * 2 >= (1L << 32)) return MODE_ERR_UNSUPPORTED;
    const long mt = (d->M + 255) / 256 + (d->expert_offsets ? d->num_experts : 0);      // upper bound; the kernel counts the real m-tiles
    t_max = mt * p.n_tiles * S;
    if (!force && (((d->M + 255) / 256) * p.n_tiles * S < 192 || d->K / S < 512)) return MODE_ERR_UNSUPPORTED;
  }
  return MODE_OK;
}

extern int g_tr_cfg;       // gemm_bf16_tr.hip ("gemm_tr_cfg" option)
extern int g_bwd_coexec;   // gemm_bf16_tr.hip ("bwd_coexec" option)

// Would mode_gemm run this descriptor on the ping-pong kernel?  The training chain asks before it shapes its operands for it (K-slice count of the
// up-projection data gradient, the pre-gathered u rows of its weight gradient).
bool gemm_bf16_pptr_accepts(const ModeGemmDesc* d) {
  if (!((g_tr_cfg == 0 && !g_bwd_coexec) || g_tr_cfg == 6)) return false;
  PpTrParams p;
  long t_max = 0;
  return pptr_plan(d, g_tr_cfg == 6, p, t_max) == MODE_OK;
}

// Entered from gemm_bf16_tr_launch with a validated descriptor.
int gemm_bf16_pptr_launch(const ModeGemmDesc* d, bool force, hipStream_t s) {
  PpTrParams p;
  long t_max = 0;
  const int rc = pptr_plan(d, force, p, t_max);
  if (rc != MODE_OK) return rc;
  const bool a_km = (d->flags & MODE_GEMM_A_KM) != 0, ob = d->out_dtype == MODE_BF16;
  if (a_km) return ob ? pptr_launch<true, true>(p, t_max, s) : pptr_launch<true, false>(p, t_max, s);
  return ob ? pptr_launch<false, true>(p, t_max, s) : pptr_launch<false, false>(p, t_max, s);
}

// out[r][:] = in[rows[r]][:]  (bf16, cols % 8 == 0): the sorted-order copy of the u rows for the up-projection weight gradient (the ping-pong kernel
// reads its K rows where they lie; the ring kernel gathers them through w_rows instead)
__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const uint16_t* __restrict__ in, long ld_in, const int* __restrict__ rows, int n, int cols,
                                                               uint16_t* __restrict__ out, long ld_out) {
  const int cpr = cols / 8;                                    // 16-byte chunks per row
  const long total = (long)n * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / cpr), c = (int)(i - (long)r * cpr);
    *reinterpret_cast<uint4*>(out + (long)r * ld_out + c * 8) = *reinterpret_cast<const uint4*>(in + (long)rows[r] * ld_in + c * 8);
  }
}
int gather_rows_bf16(const void* in, long ld_in, const int* rows, int n, int cols, void* out, long ld_out, hipStream_t s) {
  if (n <= 0) return MODE_OK;
  if (cols % 8 || ld_in % 8 || ld_out % 8) return MODE_ERR_UNSUPPORTED;
  const long total = (long)n * (cols / 8);
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(grid), dim3(256), 0, s, (const uint16_t*)in, ld_in, rows, n, cols, (uint16_t*)out, ld_out);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

}  // namespace mode
