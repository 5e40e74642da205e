// fp32 MFMA GEMM for gfx950: same contract as the bf16 kernel (plain / gathered / grouped, same epilogues) on
// v_mfma_f32_16x16x4_f32 — bit-exactly a k-ordered fp32 fma chain, so the noise-conditioned router (whose top-k
// integers must match the fp32 reference) and the fp32 parity mode of the whole denoiser run on the matrix cores
// without any reduced-precision step.  64x64x16 tile, 4 wave64 (2x2), each wave 2x2 accumulators of 16x16.
// General in M, N, K (guarded loads/stores); float4 global loads when K % 16 == 0 and rows are 16-byte aligned.
#include "mode_common.h"

namespace mode {

constexpr int FBM = 64, FBN = 64, FBK = 16, FNT = 256, FLD = FBK + 1;

struct GemmF32Params {
  const float* A; long lda;
  const float* W; long ldw; long w_estride;
  const float* bias; long bias_estride;
  const float* resid; long ldr;
  void* C; long ldc;
  const int* a_rows; const int* offsets; int E;
  const int* koffs; long c_gstride;
  int M, N, K, m_tiles, n_tiles;
  int a_km, w_kn;                         // operand given as [K][M] / [K][N] row-major (backward-pass layouts, see MODE_GEMM_A_KM / W_KN)
};

template <int EPI, bool OUT_BF16, bool VEC>
__global__ __launch_bounds__(FNT) void gemm_f32_kernel(const GemmF32Params p) {
  __shared__ float sA[2][FBM * FLD];
  __shared__ float sB[2][FBN * FLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int mt = blockIdx.x / p.n_tiles, nt = blockIdx.x % p.n_tiles;

  int row0 = 0, row_end = 0, expert = 0;
  if (p.offsets) {
    int t = mt, e = 0;
    bool found = false;
    for (; e < p.E; ++e) {
      const int o0 = p.offsets[e], o1 = p.offsets[e + 1];
      const int nt_e = (o1 - o0 + FBM - 1) / FBM;
      if (t < nt_e) { row0 = o0 + t * FBM; row_end = min(o1, row0 + FBM); found = true; break; }
      t -= nt_e;
    }
    if (!found) return;
    expert = e;
  } else {
    row0 = mt * FBM; row_end = min(p.M, row0 + FBM);
  }
  const float* W = p.W + (long)expert * p.w_estride;
  const float* bias = p.bias ? p.bias + (long)expert * p.bias_estride : nullptr;
  constexpr int NOUT = (EPI == MODE_EPI_SWIGLU) ? 32 : FBN;
  const int n0 = nt * NOUT;

  // staging: thread -> tile row (tid>>2), k quad (tid&3)
  const int tr = tid >> 2, kq = (tid & 3) * 4;
  const int s = min(row0 + tr, row_end - 1);
  const long arow = p.a_rows ? (long)p.a_rows[s] : (long)s;
  const float* a_src = p.A + arow * p.lda + kq;
  long brow;
  if constexpr (EPI == MODE_EPI_SWIGLU) brow = (long)min(n0 + (tr & 31), p.N - 1) + ((tr >= 32) ? p.N : 0);
  else brow = min(n0 + tr, p.N - 1);
  const float* b_src = W + brow * p.ldw + kq;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  int a_row[2], b_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_row[i] = (wm * 32 + i * 16 + fr) * FLD;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int br;
    if constexpr (EPI == MODE_EPI_SWIGLU) br = (j == 0) ? wn * 16 : 32 + wn * 16;
    else br = wn * 32 + j * 16;
    b_row[j] = (br + fr) * FLD;
  }

  int kbeg = 0, kend = p.K;
  if (p.koffs) { kbeg = p.koffs[blockIdx.z]; kend = p.koffs[blockIdx.z + 1]; }
  const int nk = (kend - kbeg + FBK - 1) / FBK;
  float4 ra, rb;
  // [K][cols] operands: thread -> k row (tid>>4), column quad (tid&15)*4; stored transposed into the same [row][k] LDS image
  const int krow = tid >> 4, q4 = (tid & 15) * 4;
  auto kn_load = [&](const float* base, long ld, int k, int c, int climit) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kend) {
      const float* src = base + (long)k * ld + c;
      if (VEC && c + 3 < climit) v = *reinterpret_cast<const float4*>(src);
      else {
        if (c + 0 < climit) v.x = src[0];
        if (c + 1 < climit) v.y = src[1];
        if (c + 2 < climit) v.z = src[2];
        if (c + 3 < climit) v.w = src[3];
      }
    }
    return v;
  };
  auto gload = [&](int kt) {
    const int k = kbeg + kt * FBK + kq;
    if (p.a_km || p.w_kn) {
      const int kr = kbeg + kt * FBK + krow;
      if (p.a_km) ra = kn_load(p.A, p.lda, kr, row0 + q4, p.M);
      else if (k + 3 < kend) ra = *reinterpret_cast<const float4*>(a_src + kbeg + kt * FBK);
      else {
        const float* a = a_src + kbeg + kt * FBK;
        ra.x = (k + 0 < kend) ? a[0] : 0.f; ra.y = (k + 1 < kend) ? a[1] : 0.f; ra.z = (k + 2 < kend) ? a[2] : 0.f; ra.w = (k + 3 < kend) ? a[3] : 0.f;
      }
      if (p.w_kn) rb = kn_load(W, p.ldw, kr, n0 + q4, p.N);
      else if (k + 3 < kend) rb = *reinterpret_cast<const float4*>(b_src + kbeg + kt * FBK);
      else {
        const float* b = b_src + kbeg + kt * FBK;
        rb.x = (k + 0 < kend) ? b[0] : 0.f; rb.y = (k + 1 < kend) ? b[1] : 0.f; rb.z = (k + 2 < kend) ? b[2] : 0.f; rb.w = (k + 3 < kend) ? b[3] : 0.f;
      }
    } else if constexpr (VEC) {
      ra = *reinterpret_cast<const float4*>(a_src + kbeg + kt * FBK);
      rb = *reinterpret_cast<const float4*>(b_src + kbeg + kt * FBK);
    } else {
      const float* a = a_src + kbeg + kt * FBK; const float* b = b_src + kbeg + kt * FBK;
      ra.x = (k + 0 < kend) ? a[0] : 0.f; ra.y = (k + 1 < kend) ? a[1] : 0.f;
      ra.z = (k + 2 < kend) ? a[2] : 0.f; ra.w = (k + 3 < kend) ? a[3] : 0.f;
      rb.x = (k + 0 < kend) ? b[0] : 0.f; rb.y = (k + 1 < kend) ? b[1] : 0.f;
      rb.z = (k + 2 < kend) ? b[2] : 0.f; rb.w = (k + 3 < kend) ? b[3] : 0.f;
    }
  };
  auto commit = [&](int buf) {
    if (p.a_km) {
      float* a = &sA[buf][q4 * FLD + krow];
      a[0] = ra.x; a[FLD] = ra.y; a[2 * FLD] = ra.z; a[3 * FLD] = ra.w;
    } else {
      float* a = &sA[buf][tr * FLD + kq];
      a[0] = ra.x; a[1] = ra.y; a[2] = ra.z; a[3] = ra.w;
    }
    if (p.w_kn) {
      float* b = &sB[buf][q4 * FLD + krow];
      b[0] = rb.x; b[FLD] = rb.y; b[2 * FLD] = rb.z; b[3 * FLD] = rb.w;
    } else {
      float* b = &sB[buf][tr * FLD + kq];
      b[0] = rb.x; b[1] = rb.y; b[2] = rb.z; b[3] = rb.w;
    }
  };

  if (nk > 0) { gload(0); commit(0); }
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    const bool more = (kt + 1) < nk;
    if (more) gload(kt + 1);
    const float* At = sA[kt & 1]; const float* Bt = sB[kt & 1];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = At[a_row[i] + kk * 4 + fq];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = Bt[b_row[j] + kk * 4 + fq];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);   // swapped: D[n][m]
    }
    if (more) commit((kt + 1) & 1);
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
  for (int i = 0; i < 2; ++i) {
    const int ml = wm * 32 + i * 16 + fr;
    if (ml >= rows_valid) continue;
    const long m = row0 + ml;
    if constexpr (EPI == MODE_EPI_SWIGLU) {
      const int n = n0 + wn * 16 + fq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r < p.N) {
          const float v = acc[i][0][r] + bias[n + r], g = acc[i][1][r] + bias[p.N + n + r];
          store(m, n + r, v * silu_f(g));
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 32 + j * 16 + fq * 4;
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

template <int EPI, bool OUT_BF16>
static int launch_f32(const GemmF32Params& p, int nblk, bool vec, int ngroups, hipStream_t s) {
  if (vec) hipLaunchKernelGGL((gemm_f32_kernel<EPI, OUT_BF16, true>), dim3(nblk, 1, ngroups), dim3(FNT), 0, s, p);
  else hipLaunchKernelGGL((gemm_f32_kernel<EPI, OUT_BF16, false>), dim3(nblk, 1, ngroups), dim3(FNT), 0, s, p);
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
    if (d->a_rows || d->expert_offsets || d->epilogue == MODE_EPI_SWIGLU || d->out_dtype != MODE_F32) return MODE_ERR_UNSUPPORTED;
    if (d->lda % 4 || d->ldw % 4 || (((uintptr_t)d->A | (uintptr_t)d->W) % 16)) return MODE_ERR_UNSUPPORTED;
    if (!a_km && d->K % 4) return MODE_ERR_UNSUPPORTED;
  }
  GemmF32Params p;
  p.a_km = a_km; p.w_kn = w_kn;
  p.A = (const float*)d->A; p.lda = d->lda;
  p.W = (const float*)d->W; p.ldw = d->ldw; p.w_estride = d->w_expert_stride;
  p.bias = d->bias; p.bias_estride = d->bias_expert_stride;
  p.resid = d->resid; p.ldr = d->ldr; p.C = d->C; p.ldc = d->ldc;
  p.a_rows = d->a_rows; p.offsets = d->expert_offsets; p.E = d->num_experts;
  p.M = d->M; p.N = d->N; p.K = d->K;
  const int nout = (d->epilogue == MODE_EPI_SWIGLU) ? 32 : FBN;
  p.n_tiles = (d->N + nout - 1) / nout;
  p.m_tiles = (d->M + FBM - 1) / FBM + (d->expert_offsets ? d->num_experts : 0);
  const int nblk = p.m_tiles * p.n_tiles;
  const bool vec = (a_km || w_kn) ? true
                                  : (d->K % FBK == 0) && (d->lda % 4 == 0) && (d->ldw % 4 == 0) && (d->w_expert_stride % 4 == 0) &&
                                        (((uintptr_t)d->A | (uintptr_t)d->W) % 16 == 0);
  const bool ob = d->out_dtype == MODE_BF16;
  p.koffs = d->k_group_offsets; p.c_gstride = d->c_group_stride;
  const int ng = d->k_group_offsets ? d->num_k_groups : 1;
  if (d->k_group_offsets && (d->num_k_groups <= 0 || d->expert_offsets)) return MODE_ERR_BAD_ARG;
#define MODE_CASE(E) \
  case E: return ob ? launch_f32<E, true>(p, nblk, vec, ng, s) : launch_f32<E, false>(p, nblk, vec, ng, s);
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
