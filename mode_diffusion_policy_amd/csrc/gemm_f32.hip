// fp32 MFMA GEMM for gfx950: same contract as the bf16 kernel (plain / gathered / grouped, same epilogues) on
// v_mfma_f32_16x16x4_f32 — bit-exactly a k-ordered fp32 fma chain, so the noise-conditioned router (whose top-k
// integers must match the fp32 reference) and the fp32 parity mode of the whole denoiser run on the matrix cores
// without any reduced-precision step.  64x64 tile (32x32 for products with few tiles), 4 wave64 (2x2), each wave 2x2 (1) accumulators of 16x16; K advances in
// LDS fills of FOUR (EIGHT) 16-wide sub-steps: one barrier pair per fill, the next fill's eight 16-byte loads per thread in flight under the MFMAs of the current one
// (with one sub-step per barrier every K-step cost a memory round trip - 0.7 us - and four dependent ds_read_b32 + MFMA groups: the whole price of the M = 128
// embedding / router products).  The MFMA sequence of an output element is the same k-ascending chain whatever the tile and the fill size
// (tests/test_gpu_kernels.py::test_gemm_f32_result_does_not_depend_on_tile_or_operand_layout).
// General in M, N, K (guarded loads/stores); float4 global loads when K % 16 == 0 and rows are 16-byte aligned.
#include <type_traits>
#include "mode_common.h"

namespace mode {

constexpr int FBK = 16, FNT = 256;

struct GemmF32Params {
  const float* A; long lda;
  const float* W; long ldw; long w_estride;
  const float* bias; long bias_estride;
  const float* resid; long ldr;
  void* C; long ldc;
  const int* a_rows; const int* offsets; int E;
  const int* koffs; long c_gstride;
  int M, N, K, m_tiles, n_tiles;
};

// LAYOUT bit 0: A given as [K][M] row-major, bit 1: W given as [K][N] row-major (backward-pass layouts, see MODE_GEMM_A_KM / W_KN).
// SMALL: 32x32 tile (one 16x16 accumulator per wave) with 128 k per LDS fill - for products with fewer 64x64 tiles than CUs, where the serial
// MFMA chain of an accumulator (K/4 instructions of 32 cycles) times the accumulators per wave is the run time.
// LDS image of a fill: [tile row][KF], the 16 k of a sub-step u stored 4x4-transposed as four 16-byte chunks c = k % 4 holding k/4 = 0..3: lane (row fr,
// k-group fq) of an MFMA operand reads ONE chunk = its values for the sub-step's four MFMAs, i.e. the same k-ascending chain as four 4-byte reads of a
// row-major image.  Chunk (u, c) of row r lies at slot (4u + c) ^ (r & 15) of the row: the 16-lane groups in which ds_read_b128 is served (MI355X_MICROARCH.md
// "LDS": each holds every fragment row once, with two different fq) then hit 16 different 16-byte slots, and the staging maps below make every 32-lane group of
// a ds_write_b32 hit 32 different banks for both operand layouts (a padded row-major image measured 50 % - [rows][K] - to 80 % - [K][cols] - of its LDS
// cycles as bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
template <int EPI, bool OUT_BF16, bool VEC, int LAYOUT, bool SMALL>
__global__ __launch_bounds__(FNT) void gemm_f32_kernel(const GemmF32Params p) {
  constexpr int TM = SMALL ? 32 : 64, TN = TM, FSUB = SMALL ? 8 : 4, KF = FSUB * FBK, FLD = KF;
  constexpr int MI = TM / 32, NJ = TN / 32, QR = KF / 4, RS = FNT / QR, NL = TM * QR / FNT;        // NL float4 per thread, operand and fill
  constexpr int CQ = TM / 4, KS = FNT / CQ;                                                        // [K][cols] staging: column quads per k row, k rows per step
  constexpr bool A_KM = (LAYOUT & 1) != 0, W_KN = (LAYOUT & 2) != 0;
  static_assert(!(SMALL && EPI == MODE_EPI_SWIGLU), "the SwiGLU epilogue pairs the two 32-row halves of a 64-row weight tile");
  __shared__ __attribute__((aligned(16))) float sA[TM * FLD];
  __shared__ __attribute__((aligned(16))) float sB[TN * FLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int mt = blockIdx.x / p.n_tiles, nt = blockIdx.x % p.n_tiles;

  int row0 = 0, row_end = 0, expert = 0;
  if (p.offsets) {
    int t = mt, e = 0;
    bool found = false;
    for (; e < p.E; ++e) {
      const int o0 = p.offsets[e], o1 = p.offsets[e + 1];
      const int nt_e = (o1 - o0 + TM - 1) / TM;
      if (t < nt_e) { row0 = o0 + t * TM; row_end = min(o1, row0 + TM); found = true; break; }
      t -= nt_e;
    }
    if (!found) return;
    expert = e;
  } else {
    row0 = mt * TM; row_end = min(p.M, row0 + TM);
  }
  const float* W = p.W + (long)expert * p.w_estride;
  const float* bias = p.bias ? p.bias + (long)expert * p.bias_estride : nullptr;
  constexpr int NOUT = (EPI == MODE_EPI_SWIGLU) ? 32 : TN;
  const int n0 = nt * NOUT;

  int kbeg = 0, kend = p.K;
  if (p.koffs) { kbeg = p.koffs[blockIdx.z]; kend = p.koffs[blockIdx.z + 1]; }
  const int nk = (kend - kbeg + FBK - 1) / FBK;                          // 16-wide sub-steps
  const int nfill = (nk + FSUB - 1) / FSUB;

  // ---- staging.  A thread moves NL float4 per operand and fill; step i covers RS tile rows ([rows][K] operand) / KS k rows ([K][cols] operand).
  // [rows][K]: a 32-lane group = 4 rows x 8 adjacent k quads (128 B of each row), lane bits {q = quad & 3, sub-step bit 0, row & 3}: the four dwords a
  // lane stores (k = 4 quad + j -> chunk j, dword q) land on banks 16 (u ^ r>>2 & 1) + 4 (j ^ r & 3) + q - all 32 different.
  // [K][cols]: a 32-lane group = 16 k rows x 2 adjacent column quads, so that dword (k>>2 & 3) of chunk (k & 3) of rows 4 cq + j covers 32 banks too
  // (lanes along the columns, as coalescing alone would have it, put 32 lanes on 4 banks).
  constexpr int UB = QR / 8;                                             // values of (sub-step >> 1)
  const int l5 = tid & 31, hi5 = tid >> 5;
  const int q_ = l5 & 3, su = ((l5 >> 2) & 1) + 2 * (hi5 % UB), qd = 4 * su + q_, r0 = (l5 >> 3) + 4 * (hi5 / UB);
  const int cq = (tid >> 4) % CQ, kl0 = (tid & 15) + 16 * (tid / (16 * CQ));
  const int rj = r0 & 3, kj = kl0 & 3;                                   // XOR term of the element-j chunk: constant over the steps
  auto chunk_base = [](int u, int x) { return 4 * (((u >> 2) << 4) | (((u & 3) ^ (x & 3)) << 2)); };   // dword offset of chunk (u, c = 0 ^ ...) before the c term
  const float* a_ptr[NL]; const float* b_ptr[NL];
  int a_lds[NL], b_lds[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    if constexpr (A_KM) {
      const int kl = kl0 + KS * i;
      int c = row0 + 4 * cq;
      if constexpr (VEC) c = max(min(c, p.M - 4), 0);                     // columns past M feed rows that are never stored: any in-bounds address will do
      a_ptr[i] = p.A + (long)kl * p.lda + c;
      a_lds[i] = 4 * cq * FLD + chunk_base(kl >> 4, cq) + ((kl >> 2) & 3);
    } else {
      const int t = r0 + RS * i, s = min(row0 + t, row_end - 1);
      const long arow = p.a_rows ? (long)p.a_rows[s] : (long)s;
      a_ptr[i] = p.A + arow * p.lda + 4 * qd;
      a_lds[i] = t * FLD + chunk_base(su, t >> 2) + q_;
    }
    if constexpr (W_KN) {
      const int kl = kl0 + KS * i;
      int c = n0 + 4 * cq;
      if constexpr (VEC) c = max(min(c, p.N - 4), 0);
      b_ptr[i] = W + (long)kl * p.ldw + c;
      b_lds[i] = 4 * cq * FLD + chunk_base(kl >> 4, cq) + ((kl >> 2) & 3);
    } else {
      const int t = r0 + RS * i;
      long brow;
      if constexpr (EPI == MODE_EPI_SWIGLU) brow = (long)min(n0 + (t & 31), p.N - 1) + ((t >= 32) ? p.N : 0);
      else brow = min(n0 + t, p.N - 1);
      b_ptr[i] = W + brow * p.ldw + 4 * qd;
      b_lds[i] = t * FLD + chunk_base(su, t >> 2) + q_;
    }
  }

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  int a_row[MI], b_row[NJ];                                              // fragment row + this lane's chunk term 4 (fq ^ (fr & 3))
#pragma unroll
  for (int i = 0; i < MI; ++i) a_row[i] = (wm * (TM / 2) + i * 16 + fr) * FLD + 4 * (fq ^ (fr & 3));
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int br;
    if constexpr (EPI == MODE_EPI_SWIGLU) br = (j == 0) ? wn * 16 : 32 + wn * 16;
    else br = wn * (TN / 2) + j * 16;
    b_row[j] = (br + fr) * FLD + 4 * (fq ^ (fr & 3));
  }
  int xoff[4];                                                           // sub-step term 16 ((u & 3) ^ (fr >> 2)); u >> 2 adds 64
#pragma unroll
  for (int v = 0; v < 4; ++v) xoff[v] = 16 * (v ^ (fr >> 2));

  float4 ra[NL], rb[NL];
  // one operand's NL float4 of fill `st`.  FULL: the fill lies inside [kbeg, kend) - every load unconditional, all of them issued back to back (a load
  // under a per-lane condition compiles to a branch + a dependent round trip); the one partial fill at the end of a K range takes the guarded form
  auto load_rows = [&](const float* const (&ptr)[NL], float4 (&r)[NL], int st, auto full_c) {    // [rows][K]: 4 consecutive k at kbeg + st KF + 4 qd
    constexpr bool FULL = decltype(full_c)::value;
    const int k = kbeg + st * KF + 4 * qd;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const float* src = ptr[i] + kbeg + st * KF;
      if constexpr (VEC && FULL) r[i] = *reinterpret_cast<const float4*>(src);
      else if (VEC && k + 3 < kend) r[i] = *reinterpret_cast<const float4*>(src);
      else {
        r[i].x = (k + 0 < kend) ? src[0] : 0.f; r[i].y = (k + 1 < kend) ? src[1] : 0.f;
        r[i].z = (k + 2 < kend) ? src[2] : 0.f; r[i].w = (k + 3 < kend) ? src[3] : 0.f;
      }
    }
  };
  auto load_cols = [&](const float* const (&ptr)[NL], float4 (&r)[NL], int st, auto full_c, long ld, int c0, int climit) {   // [K][cols]: 4 consecutive columns of one k row
    constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int kl = kl0 + KS * i, k = kbeg + st * KF + kl;
      if constexpr (VEC) {                                              // clamped row, then a select: no load under a per-lane condition
        const int kc = FULL ? k : min(k, kend - 1);
        const float4 v = *reinterpret_cast<const float4*>(ptr[i] + (long)(kc - kl) * ld);
        r[i] = (FULL || k < kend) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const float* src = ptr[i] + (long)(k - kl) * ld;
        const int c = c0 + 4 * cq;
        const bool in = k < kend;
        r[i].x = (in && c + 0 < climit) ? src[0] : 0.f; r[i].y = (in && c + 1 < climit) ? src[1] : 0.f;
        r[i].z = (in && c + 2 < climit) ? src[2] : 0.f; r[i].w = (in && c + 3 < climit) ? src[3] : 0.f;
      }
    }
  };
  auto gload_as = [&](int st, auto full_c) {
    if constexpr (A_KM) load_cols(a_ptr, ra, st, full_c, p.lda, row0, p.M); else load_rows(a_ptr, ra, st, full_c);
    if constexpr (W_KN) load_cols(b_ptr, rb, st, full_c, p.ldw, n0, p.N); else load_rows(b_ptr, rb, st, full_c);
  };
  auto gload = [&](int st) {
    if (kbeg + (st + 1) * KF <= kend) gload_as(st, std::true_type{});
    else gload_as(st, std::false_type{});
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      // element j of a float4: [K][cols] - tile row 4 cq + j, chunk (k & 3) ^ j;  [rows][K] - chunk j ^ (row & 3) of the thread's row
      if constexpr (A_KM) { float* a = &sA[a_lds[i]]; a[4 * kj] = ra[i].x; a[FLD + 4 * (kj ^ 1)] = ra[i].y; a[2 * FLD + 4 * (kj ^ 2)] = ra[i].z; a[3 * FLD + 4 * (kj ^ 3)] = ra[i].w; }
      else { float* a = &sA[a_lds[i]]; a[4 * rj] = ra[i].x; a[4 * (rj ^ 1)] = ra[i].y; a[4 * (rj ^ 2)] = ra[i].z; a[4 * (rj ^ 3)] = ra[i].w; }
      if constexpr (W_KN) { float* b = &sB[b_lds[i]]; b[4 * kj] = rb[i].x; b[FLD + 4 * (kj ^ 1)] = rb[i].y; b[2 * FLD + 4 * (kj ^ 2)] = rb[i].z; b[3 * FLD + 4 * (kj ^ 3)] = rb[i].w; }
      else { float* b = &sB[b_lds[i]]; b[4 * rj] = rb[i].x; b[4 * (rj ^ 1)] = rb[i].y; b[4 * (rj ^ 2)] = rb[i].z; b[4 * (rj ^ 3)] = rb[i].w; }
    }
  };
  auto frag = [&](int u, float4 (&a)[MI], float4 (&b)[NJ]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const float4*>(&sA[a_row[i] + (u >> 2) * 64 + xoff[u & 3]]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const float4*>(&sB[b_row[j] + (u >> 2) * 64 + xoff[u & 3]]);
  };
  auto mfmas = [&](const float4 (&a)[MI], const float4 (&b)[NJ]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float av = kk == 0 ? a[i].x : kk == 1 ? a[i].y : kk == 2 ? a[i].z : a[i].w;
          const float bv = kk == 0 ? b[j].x : kk == 1 ? b[j].y : kk == 2 ? b[j].z : b[j].w;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);           // swapped: D[n][m]
        }
  };

  if (nfill > 0) gload(0);
  for (int st = 0; st < nfill; ++st) {
    commit();
    __syncthreads();
    if (st + 1 < nfill) gload(st + 1);                                   // in flight under this fill's MFMAs
    const int nsub = min(FSUB, nk - st * FSUB);
    if (nsub == FSUB) {                                                  // the next sub-step's fragments are read ahead of the current one's MFMAs
      float4 ac[MI], bc[NJ], an[MI], bn[NJ];
      frag(0, ac, bc);
#pragma unroll
      for (int u = 0; u < FSUB; ++u) {
        if (u + 1 < FSUB) frag(u + 1, an, bn);
        mfmas(ac, bc);
#pragma unroll
        for (int i = 0; i < MI; ++i) ac[i] = an[i];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bc[j] = bn[j];
      }
    } else {
#pragma unroll
      for (int u = 0; u < FSUB; ++u) {                                   // (u stays a compile-time index of the chunk offsets)
        if (u < nsub) {
          float4 ac[MI], bc[NJ];
          frag(u, ac, bc);
          mfmas(ac, bc);
        }
      }
    }
    __syncthreads();
  }

  const int rows_valid = row_end - row0;
  auto store = [&](long m, int n, float v) {
    if (n < p.N) {
      const long o = (long)blockIdx.z * p.c_gstride + m * p.ldc + n;
      if constexpr (OUT_BF16) reinterpret_cast<uint16_t*>(p.C)[o] = f32_to_bf16_bits(v);
      else reinterpret_cast<float*>(p.C)[o] = v;
    }
  };
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ml = wm * (TM / 2) + i * 16 + fr;
    if (ml >= rows_valid) continue;
    const long m = row0 + ml;
    if constexpr (EPI == MODE_EPI_SWIGLU) {
      const int n = n0 + wn * 16 + fq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r < p.N) {
          const float v = acc[i][0][r] + bias[n + r], g = acc[i][NJ - 1][r] + bias[p.N + n + r];
          store(m, n + r, v * silu_f(g));
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * (TN / 2) + j * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r < p.N) {
            float v = acc[i][j][r];
            if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) v += bias[n + r];
            if constexpr (EPI == MODE_EPI_BIAS_GELU) v = gelu_erf_f(v);
            if constexpr (EPI == MODE_EPI_RESIDUAL) v += p.resid[m * p.ldr + n + r];
            store(m, n + r, v);
          }
        }
      }
    }
  }
}

// ---- skinny path: M <= 16 rows (the noise-conditioned router on the sampler's distinct sigma rows, sigma_linear on R rows).
// HBM-bound weight stream: one wave per output feature reads its weight row once (float4 per lane, coalesced) and reuses it
// for all R activation rows (L1/L2-resident), fixed-order per-lane partial sums + xor-butterfly -> deterministic fp32.
template <int EPI>
__global__ __launch_bounds__(256) void linear_f32_skinny_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W, long ldw,
                                                                 const float* __restrict__ bias, float* __restrict__ Y, long ldy, int R,
                                                                 int N, int K) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* w = W + (long)n * ldw;
  // four K-chunks per trip: their weight loads (the HBM stream) are requested together instead of one round trip per chunk; the per-lane
  // fma order (k ascending) is unchanged
  for (int k0 = lane * 4; k0 < K; k0 += 1024) {
    float4 wq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wq[u] = *reinterpret_cast<const float4*>(w + min(k0 + u * 256, K - 4));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
    const int k = k0 + u * 256;
    if (k >= K) break;
    const float4 wv = wq[u];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < R) {
        const float4 xv = *reinterpret_cast<const float4*>(X + (long)r * ldx + k);
        acc[r] = fmaf(xv.x, wv.x, acc[r]); acc[r] = fmaf(xv.y, wv.y, acc[r]);
        acc[r] = fmaf(xv.z, wv.z, acc[r]); acc[r] = fmaf(xv.w, wv.w, acc[r]);
      }
    }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < R) {
      float v = wave_sum(acc[r]);
      if (lane == 0) {
        if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) v += bias[n];
        if constexpr (EPI == MODE_EPI_BIAS_GELU) v = gelu_erf_f(v);
        Y[(long)r * ldy + n] = v;
      }
    }
  }
}

// ---- small-N path (N <= 16: router logits [B, E], output head [R, A]): one wave per output element, both rows streamed with float4
// loads, xor-butterfly reduction.  Chosen by N only (never by the batch size), so results do not depend on how many rows are batched.
template <int EPI>
__global__ __launch_bounds__(256) void linear_f32_dot_kernel(const float* __restrict__ X, long ldx, const int* __restrict__ x_rows,
                                                              const float* __restrict__ W, long ldw, const float* __restrict__ bias,
                                                              const float* resid, long ldr, float* Y, long ldy, int M, int N, int K) {
  const int lane = threadIdx.x & 63;
  const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (long)M * N) return;
  const int m = o / N, n = o % N;
  const float* x = X + (x_rows ? (long)x_rows[m] : (long)m) * ldx;
  const float* w = W + (long)n * ldw;
  float acc = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 xv = *reinterpret_cast<const float4*>(x + k);
    const float4 wv = *reinterpret_cast<const float4*>(w + k);
    acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    if constexpr (EPI == MODE_EPI_BIAS || EPI == MODE_EPI_BIAS_GELU) acc += bias[n];
    if constexpr (EPI == MODE_EPI_BIAS_GELU) acc = gelu_erf_f(acc);
    if constexpr (EPI == MODE_EPI_RESIDUAL) acc += resid[(long)m * ldr + n];
    Y[(long)m * ldy + n] = acc;
  }
}

static int launch_dot(const ModeGemmDesc* d, hipStream_t s) {
  const long outs = (long)d->M * d->N;
  const dim3 grid((outs + 3) / 4), blk(256);
  const float* X = (const float*)d->A; const float* W = (const float*)d->W; float* Y = (float*)d->C;
#define MODE_DOT(E) hipLaunchKernelGGL(linear_f32_dot_kernel<E>, grid, blk, 0, s, X, d->lda, d->a_rows, W, d->ldw, d->bias, d->resid, d->ldr, Y, d->ldc, d->M, d->N, d->K)
  switch (d->epilogue) {
    case MODE_EPI_NONE: MODE_DOT(MODE_EPI_NONE); break;
    case MODE_EPI_BIAS: MODE_DOT(MODE_EPI_BIAS); break;
    case MODE_EPI_BIAS_GELU: MODE_DOT(MODE_EPI_BIAS_GELU); break;
    case MODE_EPI_RESIDUAL: MODE_DOT(MODE_EPI_RESIDUAL); break;
    default: return MODE_ERR_BAD_ARG;
  }
#undef MODE_DOT
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

static int launch_skinny(const ModeGemmDesc* d, hipStream_t s) {
  const dim3 grid((d->N + 3) / 4), blk(256);
  const float* X = (const float*)d->A; const float* W = (const float*)d->W; float* Y = (float*)d->C;
  switch (d->epilogue) {
    case MODE_EPI_NONE: hipLaunchKernelGGL(linear_f32_skinny_kernel<MODE_EPI_NONE>, grid, blk, 0, s, X, d->lda, W, d->ldw, d->bias, Y, d->ldc, d->M, d->N, d->K); break;
    case MODE_EPI_BIAS: hipLaunchKernelGGL(linear_f32_skinny_kernel<MODE_EPI_BIAS>, grid, blk, 0, s, X, d->lda, W, d->ldw, d->bias, Y, d->ldc, d->M, d->N, d->K); break;
    case MODE_EPI_BIAS_GELU: hipLaunchKernelGGL(linear_f32_skinny_kernel<MODE_EPI_BIAS_GELU>, grid, blk, 0, s, X, d->lda, W, d->ldw, d->bias, Y, d->ldc, d->M, d->N, d->K); break;
    default: return MODE_ERR_BAD_ARG;
  }
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

template <int EPI, bool OUT_BF16, int LAYOUT>
static int launch_f32(const GemmF32Params& p, int nblk, bool vec, bool small, int ngroups, hipStream_t s) {
  const dim3 grid(nblk, 1, ngroups), blk(FNT);
  if constexpr (EPI == MODE_EPI_SWIGLU) {
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<EPI, OUT_BF16, true, LAYOUT, false>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<EPI, OUT_BF16, false, LAYOUT, false>), grid, blk, 0, s, p);
  } else {
    if (vec && small) hipLaunchKernelGGL((gemm_f32_kernel<EPI, OUT_BF16, true, LAYOUT, true>), grid, blk, 0, s, p);
    else if (vec) hipLaunchKernelGGL((gemm_f32_kernel<EPI, OUT_BF16, true, LAYOUT, false>), grid, blk, 0, s, p);
    else if (small) hipLaunchKernelGGL((gemm_f32_kernel<EPI, OUT_BF16, false, LAYOUT, true>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<EPI, OUT_BF16, false, LAYOUT, false>), grid, blk, 0, s, p);
  }
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

int gemm_f32_launch(const ModeGemmDesc* d, hipStream_t s) {
  if (d->K <= 0 || d->split_k > 1 || d->row_ss || d->epilogue == MODE_EPI_RESIDUAL_NORM) return MODE_ERR_UNSUPPORTED;
  if ((d->epilogue == MODE_EPI_BIAS || d->epilogue == MODE_EPI_BIAS_GELU || d->epilogue == MODE_EPI_SWIGLU) && !d->bias)
    return MODE_ERR_BAD_ARG;
  if (d->epilogue == MODE_EPI_RESIDUAL && !d->resid) return MODE_ERR_BAD_ARG;
  if (d->M <= 0) return MODE_OK;
  if (!(d->flags & (MODE_GEMM_A_KM | MODE_GEMM_W_KN)) && d->N <= 16 && d->K >= 64 && d->K % 4 == 0 && !d->expert_offsets && !d->k_group_offsets && d->out_dtype == MODE_F32 && d->lda % 4 == 0 &&
      d->ldw % 4 == 0 && d->epilogue != MODE_EPI_SWIGLU && (((uintptr_t)d->A | (uintptr_t)d->W) % 16 == 0))
    return launch_dot(d, s);
  if ((d->flags & MODE_GEMM_SKINNY_OK) && !(d->flags & (MODE_GEMM_A_KM | MODE_GEMM_W_KN)) && d->M <= 16 && !d->a_rows && !d->expert_offsets && !d->k_group_offsets && d->out_dtype == MODE_F32 && d->K % 4 == 0 && d->lda % 4 == 0 && d->ldw % 4 == 0 &&
      (d->epilogue == MODE_EPI_NONE || d->epilogue == MODE_EPI_BIAS || d->epilogue == MODE_EPI_BIAS_GELU) &&
      (((uintptr_t)d->A | (uintptr_t)d->W) % 16 == 0))
    return launch_skinny(d, s);
  const bool a_km = (d->flags & MODE_GEMM_A_KM) != 0, w_kn = (d->flags & MODE_GEMM_W_KN) != 0;
  if (a_km || w_kn) {      // backward-pass operand layouts: plain or K-grouped only; the "a_src/b_src + k" fast path needs 16-B aligned rows
    if (d->a_rows || d->expert_offsets || d->epilogue != MODE_EPI_NONE || d->out_dtype != MODE_F32) return MODE_ERR_UNSUPPORTED;
    if (d->lda % 4 || d->ldw % 4 || (((uintptr_t)d->A | (uintptr_t)d->W) % 16)) return MODE_ERR_UNSUPPORTED;
    if (!a_km && d->K % 4) return MODE_ERR_UNSUPPORTED;
  }
  GemmF32Params p;
  p.A = (const float*)d->A; p.lda = d->lda;
  p.W = (const float*)d->W; p.ldw = d->ldw; p.w_estride = d->w_expert_stride;
  p.bias = d->bias; p.bias_estride = d->bias_expert_stride;
  p.resid = d->resid; p.ldr = d->ldr; p.C = d->C; p.ldc = d->ldc;
  p.a_rows = d->a_rows; p.offsets = d->expert_offsets; p.E = d->num_experts;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.koffs = d->k_group_offsets; p.c_gstride = d->c_group_stride;
  const int ng = d->k_group_offsets ? d->num_k_groups : 1;
  if (d->k_group_offsets && (d->num_k_groups <= 0 || d->expert_offsets)) return MODE_ERR_BAD_ARG;
  const bool swiglu = d->epilogue == MODE_EPI_SWIGLU;
  // 32x32 tiles when the 64x64 cover leaves most CUs without a workgroup (bit-identical: an element's MFMA chain does not depend on the tile)
  const long tiles64 = ((long)(d->M + 63) / 64 + (d->expert_offsets ? d->num_experts : 0)) * ((d->N + 63) / 64) * ng;
  const bool small = !swiglu && tiles64 <= 160;
  const int tm = small ? 32 : 64, nout = swiglu ? 32 : tm;
  p.n_tiles = (d->N + nout - 1) / nout;
  p.m_tiles = (d->M + tm - 1) / tm + (d->expert_offsets ? d->num_experts : 0);
  const int nblk = p.m_tiles * p.n_tiles;
  // float4 global loads: [rows][K] operands need K % 16 == 0 and 16-byte aligned rows; [K][cols] operands a column count that is a multiple of 4
  const bool vec = (a_km || w_kn) ? ((!a_km || d->M % 4 == 0) && (!w_kn || d->N % 4 == 0))
                                  : (d->K % FBK == 0) && (d->lda % 4 == 0) && (d->ldw % 4 == 0) && (d->w_expert_stride % 4 == 0) &&
                                        (((uintptr_t)d->A | (uintptr_t)d->W) % 16 == 0);
  const bool ob = d->out_dtype == MODE_BF16;
  if (a_km || w_kn) {
    const int lay = (a_km ? 1 : 0) | (w_kn ? 2 : 0);
    if (lay == 1) return launch_f32<MODE_EPI_NONE, false, 1>(p, nblk, vec, small, ng, s);
    if (lay == 2) return launch_f32<MODE_EPI_NONE, false, 2>(p, nblk, vec, small, ng, s);
    return launch_f32<MODE_EPI_NONE, false, 3>(p, nblk, vec, small, ng, s);
  }
#define MODE_CASE(E) \
  case E: return ob ? launch_f32<E, true, 0>(p, nblk, vec, small, ng, s) : launch_f32<E, false, 0>(p, nblk, vec, small, ng, s);
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
