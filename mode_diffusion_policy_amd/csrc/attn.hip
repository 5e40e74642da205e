// Tiny-T causal attention for the 14-token MoDE sequence (modedit.py:125-127, 145-165): B*H independent problems of
// T <= 16 tokens, head_dim 32..128.  One wave64 per (sample, head), four per workgroup, no LDS at all:
//   * q/k rows are read straight into MFMA fragments (16 B per lane), qk-RMSNorm is a 4-lane-group shuffle reduction on those
//     fragments, S^T = K Q^T runs on v_mfma_f32_16x16x32_bf16 (T padded to 16) so that each lane ends up owning one
//     QUERY column and 4 key rows: the causal softmax is 4 local values + two xor-shuffles,
//   * the probabilities are already laid out as the B operand of  O^T = V^T P^T  (k-slots 4..7 of each lane group are zero
//     padding), so P never moves between lanes; each lane then owns 4 consecutive head-dim outputs of one query -> 8-byte stores.
// fp32 parity mode: a plain VALU kernel with the same math (one wave per (sample, head), LDS-staged q/k/v).
#include <type_traits>

#include "mode_common.h"
#include "attn_core.h"

namespace mode {

template <int NKS>   // 32*(NKS-1) < head_dim <= 32*NKS, head_dim % 16 == 0 (dims past head_dim are zero k-slots)
__global__ __launch_bounds__(256) void attn_bf16_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ qg,
                                                        const float* __restrict__ kg, uint16_t* __restrict__ y, int B, int T,
                                                        int H, int HD, float eps, uint32_t seed, uint32_t thresh, float inv_keep) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int prob = blockIdx.x * 4 + wave;
  if (prob >= B * H) return;
  const int b = prob / H, h = prob % H;
  const int D = H * HD;
  const long ld = 3L * D;
  const AttnGlobalSrc src{qkv + (long)b * T * ld + h * HD, ld, D};
  attn_wave_bf16<NKS>(src, qg, kg, y + (long)b * T * D + h * HD, D, T, HD, eps, seed, thresh, inv_keep, prob, lane);   // attn_core.h
}

// fp32 parity kernel: one wave per (b,h); q,k,v rows staged in LDS; plain VALU math in the reference's order.
__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ qg,
                                                      const float* __restrict__ kg, float* __restrict__ y, int B, int T, int H,
                                                      int HD, float eps, uint32_t seed, uint32_t thresh, float inv_keep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sq = reinterpret_cast<float*>(smem);          // [T][HD]
  float* sk = sq + T * HD;
  float* sv = sk + T * HD;
  float* sp = sv + T * HD;                             // [T][T]
  const int lane = threadIdx.x, b = blockIdx.x / H, h = blockIdx.x % H, D = H * HD;
  const long ld = 3L * D;
  for (int i = lane; i < T * HD; i += 64) {
    const int t = i / HD, d = i % HD;
    const float* r = qkv + ((long)b * T + t) * ld + h * HD + d;
    sq[i] = r[0]; sk[i] = r[D]; sv[i] = r[2 * D];
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {                        // qk-RMSNorm (modedit.py:126-127, 145-146)
    float a = 0.f, c = 0.f;
    for (int d = lane; d < HD; d += 64) { a += sq[t * HD + d] * sq[t * HD + d]; c += sk[t * HD + d] * sk[t * HD + d]; }
    a = wave_sum(a); c = wave_sum(c);
    const float qn = fmaxf(sqrtf(a) * rsqrtf((float)HD), eps), kn = fmaxf(sqrtf(c) * rsqrtf((float)HD), eps);
    for (int d = lane; d < HD; d += 64) { sq[t * HD + d] = sq[t * HD + d] / qn * qg[d]; sk[t * HD + d] = sk[t * HD + d] / kn * kg[d]; }
  }
  __syncthreads();
  const float scale = rsqrtf((float)HD);
  for (int i = lane; i < T * T; i += 64) {
    const int qi = i / T, ki = i % T;
    float a = -INFINITY;
    if (ki <= qi) {
      a = 0.f;
      for (int d = 0; d < HD; ++d) a = fmaf(sq[qi * HD + d], sk[ki * HD + d], a);
      a *= scale;
    }
    sp[i] = a;
  }
  __syncthreads();
  if (lane < T) {
    float mx = -INFINITY;
    for (int ki = 0; ki <= lane; ++ki) mx = fmaxf(mx, sp[lane * T + ki]);
    float sum = 0.f;
    for (int ki = 0; ki < T; ++ki) { const float e = (ki <= lane) ? expf(sp[lane * T + ki] - mx) : 0.f; sp[lane * T + ki] = e; sum += e; }
    for (int ki = 0; ki < T; ++ki) {
      float pv = sp[lane * T + ki] / sum;
      if (thresh) pv = attn_keep(seed, blockIdx.x, T, lane, ki, thresh) ? pv * inv_keep : 0.f;
      sp[lane * T + ki] = pv;
    }
  }
  __syncthreads();
  for (int i = lane; i < T * HD; i += 64) {
    const int qi = i / HD, d = i % HD;
    float a = 0.f;
    for (int ki = 0; ki <= qi; ++ki) a = fmaf(sp[qi * T + ki], sv[ki * HD + d], a);
    y[((long)b * T + qi) * D + h * HD + d] = a;
  }
}

// ---- backward (training): one 256-thread workgroup per (sample, head), everything in LDS, fp32 VALU in the forward's order.
// dY [B*T, D] (T2 = bf16/f32) -> dqkv [B*T, 3D]; per-workgroup partial gradients of the qk-norm gains: dgq/dgk [B*H, HD].
// MF: the five small matrix products (S = q_hat k_hat^T and dPd = dO V^T over the head dim; dq_hat = dS k_hat, dk_hat = dS^T q_hat, dV = Pd^T dO over the tokens) on
// v_mfma_f32_16x16x4_f32 - exact fp32 like the VALU form, but an operand element is read from LDS once per 16 outputs: the VALU form moved ~0.8 MB of LDS
// reads per (sample, head) - with four workgroups per CU these two phases were LDS-bandwidth-bound and two thirds of the kernel's 38 us.  head_dim % 16 == 0.
template <typename T2, bool MF>
__global__ __launch_bounds__(256, 4) void attn_bwd_kernel(const T2* __restrict__ qkv, const float* __restrict__ qg, const float* __restrict__ kg,
                                                       const T2* __restrict__ dY, T2* __restrict__ dqkv, float* __restrict__ dgq_part,
                                                       float* __restrict__ dgk_part, int B, int T, int H, int HD, float eps, uint32_t seed,
                                                       uint32_t thresh, float inv_keep, float* __restrict__ dbias_part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HP = HD + 1;                               // padded row (bank-conflict-free column walks)
  // v / dO (and dV, which replaces v) are kept in the INPUT dtype: for bf16 that is exact (the inputs are bf16, dV is rounded to bf16 on
  // its way out anyway) and brings the workgroup under 40 KB of LDS — four workgroups per CU, the whole grid of B*H = 1024 in one round
  typedef typename std::conditional<sizeof(T2) == 2, uint16_t, float>::type TV;
  const int HV = sizeof(T2) == 2 ? HD + 8 : HD + 1;     // row stride of the TV tiles (bf16: 16-byte aligned rows, 4 banks of skew)
  float* sq = reinterpret_cast<float*>(smem);          // raw q  [T][HP]
  float* sk = sq + T * HP;                             // raw k
  float* sqh = sk + T * HP;                            // q_hat, later d q_hat
  float* skh = sqh + T * HP;                           // k_hat, later d k_hat
  // probability-sized tiles are zero-padded to 16 x 16 (row stride 16): branch-free 16-term dot products, float4 broadcast reads
  float* sp = skh + T * HP + ((4 - ((4 * T * HP) & 3)) & 3);   // P [query][key]                    (16-byte aligned)
  float* sds = sp + 256;                               // dPd, then dS   [query][key]
  float* spdT = sds + 256;                             // dropped probabilities, transposed [key][query]   (VALU form only: the MFMA form keeps them in sp)
  float* sdsT = MF ? sds + 256 : spdT + 256;           // dS transposed                     [key][query]
  float* srq = sdsT + 256;                             // 1/norm per token (q)              [T]
  float* srk = srq + T;                                // (k)
  float* sgq = srk + T + ((4 - ((2 * T) & 3)) & 3);    // qk-norm gains [HD] (read per element in three phases: from LDS, not through the vector cache)
  float* sgk = sgq + HD;
  TV* sv = reinterpret_cast<TV*>(sgk + HD);            // v, later dV   [T][HV]   (16-byte aligned: HD % 4 == 0)
  TV* sdo = sv + T * HV;                               // dO
  auto tvf = [](TV x) -> float { if constexpr (sizeof(TV) == 2) return bf16_bits_to_f32(x); else return x; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int prob = blockIdx.x, b = prob / H, h = prob % H, D = H * HD;
  const long ld = 3L * D;
  // 16-byte global loads, all four tensors of a thread's chunk requested before the first use (one memory round trip per workgroup —
  // a per-element loop serialises T*HD/256 dependent round trips and was 80 % of this kernel's time)
  constexpr int VE = 16 / (int)sizeof(T2);              // elements per 16-byte chunk
  const int cpr = HD / VE;                              // chunks per token row
  auto unpack = [&](const uint4& u, float* dst) {
    if constexpr (sizeof(T2) == 2) {
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { dst[2 * j] = bf16_bits_to_f32(w[j] & 0xffff); dst[2 * j + 1] = bf16_bits_to_f32(w[j] >> 16); }
    } else {
      dst[0] = __uint_as_float(u.x); dst[1] = __uint_as_float(u.y); dst[2] = __uint_as_float(u.z); dst[3] = __uint_as_float(u.w);
    }
  };
  for (int i = tid; i < (MF ? 3 : 4) * 256; i += 256) sp[i] = 0.f;   // zero padding of the 16x16 tiles (written sparsely below)
  for (int d = tid; d < HD; d += 256) { sgq[d] = qg[d]; sgk[d] = kg[d]; }
  for (int i = tid; i < T * cpr; i += 256) {
    const int t = i / cpr, d = (i % cpr) * VE;
    const T2* r = qkv + ((long)b * T + t) * ld + h * HD + d;
    const uint4 uq = *reinterpret_cast<const uint4*>(r), uk = *reinterpret_cast<const uint4*>(r + D), uv = *reinterpret_cast<const uint4*>(r + 2 * D);
    const uint4 ud = *reinterpret_cast<const uint4*>(dY + ((long)b * T + t) * D + h * HD + d);
    float fq_[VE], fk_[VE];
    unpack(uq, fq_); unpack(uk, fk_);
#pragma unroll
    for (int j = 0; j < VE; ++j) { sq[t * HP + d + j] = fq_[j]; sk[t * HP + d + j] = fk_[j]; }
    if constexpr (sizeof(T2) == 2) {                    // bf16: the 16-byte chunks go to LDS as they are
      *reinterpret_cast<uint4*>(sv + t * HV + d) = uv; *reinterpret_cast<uint4*>(sdo + t * HV + d) = ud;
    } else {
      float fv_[VE], fd_[VE];
      unpack(uv, fv_); unpack(ud, fd_);
#pragma unroll
      for (int j = 0; j < VE; ++j) { sv[t * HV + d + j] = fv_[j]; sdo[t * HV + d + j] = fd_[j]; }
    }
  }
  __syncthreads();
  for (int t = wave; t < T; t += 4) {                  // one wave per token: row norms
    float a = 0.f, c = 0.f;
    for (int d = lane; d < HD; d += 64) { a += sq[t * HP + d] * sq[t * HP + d]; c += sk[t * HP + d] * sk[t * HP + d]; }
    a = wave_sum(a); c = wave_sum(c);
    if (lane == 0) {
      srq[t] = 1.0f / fmaxf(sqrtf(a) * rsqrtf((float)HD), eps);
      srk[t] = 1.0f / fmaxf(sqrtf(c) * rsqrtf((float)HD), eps);
    }
  }
  __syncthreads();
  for (int i = tid; i < T * HD; i += 256) {
    const int t = i / HD, d = i % HD;
    sqh[t * HP + d] = sq[t * HP + d] * srq[t] * sgq[d];
    skh[t * HP + d] = sk[t * HP + d] * srk[t] * sgk[d];
  }
  __syncthreads();
  const float scale = rsqrtf((float)HD);
  if constexpr (MF) {
    // wave 0: S = q_hat k_hat^T * scale, wave 1: dPd = dO V^T - D[i][j] += A[i][d] B[d][j], four head dims per instruction; a lane supplies A[i = l & 15][d0 + (l >> 4)]
    // and B[..][j = l & 15] and ends up with rows i = 4 (l >> 4) + r of column j.  Rows / columns >= T read a clamped (finite) row: their outputs are not stored.
    if (wave < 2) {
      const int fr = min(lane & 15, T - 1), fq = lane >> 4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (wave == 0) {
        const float* qa = sqh + fr * HP + fq; const float* kb = skh + fr * HP + fq;
        for (int d0 = 0; d0 < HD; d0 += 16) {
          float a[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { a[u] = qa[d0 + 4 * u]; bv[u] = kb[d0 + 4 * u]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[u], acc, 0, 0, 0);
        }
      } else {
        const TV* oa = sdo + fr * HV + fq; const TV* vb = sv + fr * HV + fq;
        for (int d0 = 0; d0 < HD; d0 += 16) {
          float a[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { a[u] = tvf(oa[d0 + 4 * u]); bv[u] = tvf(vb[d0 + 4 * u]); }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[u], acc, 0, 0, 0);
        }
      }
      const int ki = lane & 15;
      float* dst = wave == 0 ? sp : sds;
      const float mul = wave == 0 ? scale : 1.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = 4 * (lane >> 4) + r;
        if (qi < T && ki <= qi) dst[qi * 16 + ki] = acc[r] * mul;
      }
    }
  } else
  // S = q_hat k_hat^T * scale (causal) and dPd = dO V^T: four lanes per (query, key) pair, each a quarter of the head dim with two
  // independent accumulator pairs (the one-thread-per-pair loop was a 128-deep chain of dependent LDS reads), xor-shuffle combine
  {
    // thread = (query qi, group of 4 keys, quarter of the head dim): the q / dO values are loaded once per 4 keys (10 LDS reads per 8 FMAs),
    // 14 x 4 x 4 = 224 threads cover the whole [T][T] tile in one pass; quarter sums are combined by xor-shuffles
    const int part = tid & 3, kgrp = (tid >> 2) & 3, qi = tid >> 4;
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, pa[4] = {0.f, 0.f, 0.f, 0.f};
    const bool act = qi < T && kgrp * 4 <= qi;
    if (act) {
      const float* qr = sqh + qi * HP; const TV* orow = sdo + qi * HV;
      const int k0 = kgrp * 4;
      const float* kr[4]; const TV* vr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int kk = min(k0 + j, T - 1); kr[j] = skh + kk * HP; vr[j] = sv + kk * HV; }
#pragma unroll 4
      for (int d = part; d < HD; d += 4) {                  // lanes interleave d: consecutive banks; unrolled: 40 LDS reads in flight
        const float qv = qr[d], ov = tvf(orow[d]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { sa[j] = fmaf(qv, kr[j][d], sa[j]); pa[j] = fmaf(ov, tvf(vr[j][d]), pa[j]); }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sa[j] += __shfl_xor(sa[j], 1, 64); sa[j] += __shfl_xor(sa[j], 2, 64);
      pa[j] += __shfl_xor(pa[j], 1, 64); pa[j] += __shfl_xor(pa[j], 2, 64);
    }
    if (act && part == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ki = kgrp * 4 + j;
        if (ki <= qi) { sp[qi * 16 + ki] = sa[j] * scale; sds[qi * 16 + ki] = pa[j]; }
      }
    }
  }
  __syncthreads();
  {                                                     // softmax rows, dropout, dS = P * (dP - sum(dP*P)) * scale: 16 lanes per query row
    const int qi = wave * 4 + (lane >> 4), ki = lane & 15;
    const bool valid = qi < T && ki <= qi;
    const float sv_ = valid ? sp[qi * 16 + ki] : -INFINITY;
    float mx = sv_;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e = valid ? expf(sv_ - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float pv = valid ? e / sum : 0.f;
    float m = 1.0f;
    if (thresh && valid) m = attn_keep(seed, prob, T, qi, ki, thresh) ? inv_keep : 0.f;
    const float dP = valid ? sds[qi * 16 + ki] * m : 0.f;   // gradient wrt the un-dropped probability
    float rs = dP * pv;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) rs += __shfl_xor(rs, o, 64);
    const float dS = pv * (dP - rs) * scale;
    if (qi < T && ki < T) {
      if constexpr (MF) sp[qi * 16 + ki] = pv * m;               // the MFMA form reads the dropped probabilities [query][key]
      else { sp[qi * 16 + ki] = pv; spdT[ki * 16 + qi] = pv * m; }
      sds[qi * 16 + ki] = dS; sdsT[ki * 16 + qi] = dS;
    }
  }
  __syncthreads();
  // dV[j][d] = sum_{i>=j} Pd[i][j] dO[i][d];  dq_hat[i][d] = sum_{j<=i} dS[i][j] k_hat[j][d];  dk_hat[j][d] = sum_{i>=j} dS[i][j] q_hat[i][d].
  // One thread per (role, head-dim column) — role 0: dq_hat, role 1: dV and dk_hat.  The T column values a thread needs are read from
  // LDS once into registers, the P / dS coefficients are wave-uniform broadcast reads, so the FMAs are independent of LDS latency.
  // Results stay in registers across the barrier: they overwrite q_hat / k_hat / v in place.   (host side guarantees 2*HD <= 256, T <= 16)
  if constexpr (MF) {
    // 16-column tiles of the head dim, tile t on wave t % 4 (head_dim <= 128: at most two per wave); per tile three products over the token index k (four per
    // instruction): dq_hat[i][d] = dS[i][k] k_hat[k][d], dk_hat[j][d] = dS[k][j] q_hat[k][d], dV[j][d] = Pd[k][j] dO[k][d].  The coefficient tiles are zero beyond
    // T and above the diagonal; the other operand's rows k >= T are clamped (0 x finite).  Results wait in registers for the barrier, then replace q_hat / k_hat / v.
    const int fr = lane & 15, fq = lane >> 4, nt = HD / 16;
    f32x4 rq[2], rk[2], rv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rq[u] = f32x4{0.f, 0.f, 0.f, 0.f}; rk[u] = rq[u]; rv[u] = rq[u];
      const int t = wave + 4 * u;
      if (t < nt) {
        const int d = t * 16 + fr;
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
          const int k = k0 + fq, kc = min(k, T - 1);
          const float a_dq = sdsT[k * 16 + fr], a_dk = sds[k * 16 + fr], a_dv = sp[k * 16 + fr];
          const float b_k = skh[kc * HP + d], b_q = sqh[kc * HP + d], b_o = tvf(sdo[kc * HV + d]);
          rq[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_dq, b_k, rq[u], 0, 0, 0);
          rk[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_dk, b_q, rk[u], 0, 0, 0);
          rv[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_dv, b_o, rv[u], 0, 0, 0);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = wave + 4 * u;
      if (t < nt) {
        const int d = t * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * fq + r;
          if (row < T) {
            sqh[row * HP + d] = rq[u][r]; skh[row * HP + d] = rk[u][r];
            if constexpr (sizeof(TV) == 2) sv[row * HV + d] = f32_to_bf16_bits(rv[u][r]); else sv[row * HV + d] = rv[u][r];
          }
        }
      }
    }
  } else
  {
    constexpr int TMAX = 16;
    const int role = tid / HD, d = tid % HD;
    const bool act = tid < 2 * HD;
    float o0[TMAX], o1[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
    auto dot16 = [](const float* row, const float* col) {   // row: 16 floats in LDS (wave-uniform address -> broadcast), col: registers
      const float4 a = *reinterpret_cast<const float4*>(row), b4 = *reinterpret_cast<const float4*>(row + 4);
      const float4 c = *reinterpret_cast<const float4*>(row + 8), e = *reinterpret_cast<const float4*>(row + 12);
      float r0 = a.x * col[0], r1 = a.y * col[1], r2 = a.z * col[2], r3 = a.w * col[3];
      r0 = fmaf(b4.x, col[4], r0); r1 = fmaf(b4.y, col[5], r1); r2 = fmaf(b4.z, col[6], r2); r3 = fmaf(b4.w, col[7], r3);
      r0 = fmaf(c.x, col[8], r0); r1 = fmaf(c.y, col[9], r1); r2 = fmaf(c.z, col[10], r2); r3 = fmaf(c.w, col[11], r3);
      r0 = fmaf(e.x, col[12], r0); r1 = fmaf(e.y, col[13], r1); r2 = fmaf(e.z, col[14], r2); r3 = fmaf(e.w, col[15], r3);
      return (r0 + r1) + (r2 + r3);
    };
    if (act && role == 0) {
      float kc[TMAX];
#pragma unroll
      for (int u = 0; u < TMAX; ++u) kc[u] = u < T ? skh[u * HP + d] : 0.f;
#pragma unroll
      for (int t = 0; t < TMAX; ++t) o0[t] = dot16(sds + t * 16, kc);          // causal zeros are in the tile
    } else if (act) {
      float oc[TMAX], qc[TMAX];
#pragma unroll
      for (int u = 0; u < TMAX; ++u) { oc[u] = u < T ? tvf(sdo[u * HV + d]) : 0.f; qc[u] = u < T ? sqh[u * HP + d] : 0.f; }
#pragma unroll
      for (int j = 0; j < TMAX; ++j) { o0[j] = dot16(spdT + j * 16, oc); o1[j] = dot16(sdsT + j * 16, qc); }
    }
    __syncthreads();
    if (act && role == 0) {
#pragma unroll
      for (int t = 0; t < TMAX; ++t) if (t < T) sqh[t * HP + d] = o0[t];
    } else if (act) {
#pragma unroll
      for (int j = 0; j < TMAX; ++j) if (j < T) {
        if constexpr (sizeof(TV) == 2) sv[j * HV + d] = f32_to_bf16_bits(o0[j]); else sv[j * HV + d] = o0[j];
        skh[j * HP + d] = o1[j];
      }
    }
  }
  __syncthreads();
  for (int d = tid; d < HD; d += 256) {                 // gain-gradient partials of this (sample, head)
    float a = 0.f, c = 0.f;
    for (int t = 0; t < T; ++t) { a += sqh[t * HP + d] * sq[t * HP + d] * srq[t]; c += skh[t * HP + d] * sk[t * HP + d] * srk[t]; }
    dgq_part[(long)prob * HD + d] = a; dgk_part[(long)prob * HD + d] = c;
  }
  __syncthreads();
  // qk-RMSNorm backward (x_hat = x * r * g): dx = g*dxh*r - x * <g*dxh, x> * r^3 / HD  (clamped rows: dx = g*dxh/eps); in place over d x_hat
  {                                                     // 16 lanes per token: all T tokens in one pass, 4-step xor-shuffle reductions
    const int t = tid >> 4, l16 = tid & 15;
    float cq = 0.f, ck = 0.f;
    if (t < T) {
#pragma unroll 4
      for (int d = l16; d < HD; d += 16) { cq += sgq[d] * sqh[t * HP + d] * sq[t * HP + d]; ck += sgk[d] * skh[t * HP + d] * sk[t * HP + d]; }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { cq += __shfl_xor(cq, o, 64); ck += __shfl_xor(ck, o, 64); }
    if (t < T) {
      const float rq = srq[t], rk = srk[t];
      const bool clq = rq >= 1.0f / eps, clk = rk >= 1.0f / eps;
      const float cqs = clq ? 0.f : cq * rq * rq * rq / (float)HD, cks = clk ? 0.f : ck * rk * rk * rk / (float)HD;   // per token, not per element (a division each)
#pragma unroll 4
      for (int d = l16; d < HD; d += 16) {
        sqh[t * HP + d] = sgq[d] * sqh[t * HP + d] * rq - sq[t * HP + d] * cqs;
        skh[t * HP + d] = sgk[d] * skh[t * HP + d] * rk - sk[t * HP + d] * cks;
      }
    }
  }
  __syncthreads();
  auto pack = [&](const float* src) -> uint4 {
    uint4 u;
    if constexpr (sizeof(T2) == 2) {
      u.x = pack_bf16x2(src[0], src[1]); u.y = pack_bf16x2(src[2], src[3]); u.z = pack_bf16x2(src[4], src[5]); u.w = pack_bf16x2(src[6], src[7]);
    } else {
      u.x = __float_as_uint(src[0]); u.y = __float_as_uint(src[1]); u.z = __float_as_uint(src[2]); u.w = __float_as_uint(src[3]);
    }
    return u;
  };
  if (dbias_part) {
    // bias gradient of the packed QKV projection = column sums of dqkv: this sample's share [3 D] (summed over its T tokens, from the values AS STORED - bf16-
    // rounded on the bf16 path, like the separate column sum over dqkv did); the caller adds the B rows with one small column sum instead of re-reading dqkv
    for (int c = tid; c < 3 * HD; c += 256) {
      const int which = c / HD, d = c - which * HD;
      float sacc = 0.f;
      for (int t = 0; t < T; ++t) {
        float v = which == 0 ? sqh[t * HP + d] : which == 1 ? skh[t * HP + d] : tvf(sv[t * HV + d]);
        if constexpr (sizeof(T2) == 2) v = bf16_bits_to_f32(f32_to_bf16_bits(v));
        sacc += v;
      }
      dbias_part[(long)b * 3 * D + (long)which * D + h * HD + d] = sacc;
    }
  }
  for (int i = tid; i < T * cpr; i += 256) {            // 16-byte stores of dq | dk | dv
    const int t = i / cpr, d = (i % cpr) * VE;
    T2* o = dqkv + ((long)b * T + t) * ld + h * HD + d;
    *reinterpret_cast<uint4*>(o) = pack(sqh + t * HP + d);
    *reinterpret_cast<uint4*>(o + D) = pack(skh + t * HP + d);
    if constexpr (sizeof(T2) == 2) *reinterpret_cast<uint4*>(o + 2 * D) = *reinterpret_cast<const uint4*>(sv + t * HV + d);
    else *reinterpret_cast<uint4*>(o + 2 * D) = pack(reinterpret_cast<const float*>(sv) + t * HV + d);
  }
}

}  // namespace mode

using namespace mode;

static inline uint32_t attn_thresh(float p) { return p <= 0.f ? 0u : (uint32_t)((double)p * 4294967296.0); }

extern "C" int mode_attn_block_fwd(const void* qkv, const float* q_gain, const float* k_gain, void* y, int dtype, int B, int T, int H,
                                   int head_dim, float eps, uint32_t seed, float p_drop, void* stream) {
  if (!qkv || !q_gain || !k_gain || !y || B < 0 || T <= 0 || H <= 0) return MODE_ERR_BAD_ARG;
  if (B == 0) return MODE_OK;
  if (p_drop < 0.f || p_drop >= 1.f) return MODE_ERR_BAD_ARG;
  const uint32_t th = attn_thresh(p_drop); const float ik = 1.0f / (1.0f - p_drop);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MODE_BF16) {
    if (T > 16 || head_dim % 16 != 0 || head_dim > 128) return MODE_ERR_UNSUPPORTED;
    const dim3 grid((B * H + 3) / 4), blk(256);
    const uint16_t* in = (const uint16_t*)qkv; uint16_t* out = (uint16_t*)y;
    switch ((head_dim + 31) / 32) {
      case 1: hipLaunchKernelGGL(attn_bf16_kernel<1>, grid, blk, 0, s, in, q_gain, k_gain, out, B, T, H, head_dim, eps, seed, th, ik); break;
      case 2: hipLaunchKernelGGL(attn_bf16_kernel<2>, grid, blk, 0, s, in, q_gain, k_gain, out, B, T, H, head_dim, eps, seed, th, ik); break;
      case 3: hipLaunchKernelGGL(attn_bf16_kernel<3>, grid, blk, 0, s, in, q_gain, k_gain, out, B, T, H, head_dim, eps, seed, th, ik); break;
      case 4: hipLaunchKernelGGL(attn_bf16_kernel<4>, grid, blk, 0, s, in, q_gain, k_gain, out, B, T, H, head_dim, eps, seed, th, ik); break;
      default: return MODE_ERR_UNSUPPORTED;
    }
  } else {
    const size_t lds = ((size_t)3 * T * head_dim + (size_t)T * T) * 4;
    if (lds > 64 * 1024) return MODE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(attn_f32_kernel, dim3(B * H), dim3(64), lds, s, (const float*)qkv, q_gain, k_gain, (float*)y, B, T, H, head_dim, eps, seed, th, ik);
  }
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

namespace mode {
int g_attn_bwd_mfma = 1;   // "attn_bwd_mfma" option: 1 = the attention backward's matrix products on fp32 MFMA (head_dim % 16 == 0), 0 = the VALU form
// dbias_partial (optional): [B][3 D] fp32, row b = sample b's share of the packed QKV bias gradient (dit_train.hip adds the rows)
int attn_block_bwd_launch(const void* qkv, const float* q_gain, const float* k_gain, const void* dy, void* dqkv, float* dgq_partial,
                          float* dgk_partial, int dtype, int B, int T, int H, int head_dim, float eps, uint32_t seed, float p_drop, float* dbias_partial,
                          void* stream) {
  if (!qkv || !q_gain || !k_gain || !dy || !dqkv || !dgq_partial || !dgk_partial || B < 0 || T <= 0 || H <= 0) return MODE_ERR_BAD_ARG;
  if (p_drop < 0.f || p_drop >= 1.f) return MODE_ERR_BAD_ARG;
  if (B == 0) return MODE_OK;
  const bool mf = head_dim % 16 == 0 && g_attn_bwd_mfma;
  const size_t tiles = mf ? 3 : 4;                             // 16 x 16 coefficient tiles (the MFMA form needs no transposed copy of the dropped probabilities)
  const size_t lds = dtype == MODE_BF16 ? ((size_t)4 * T * (head_dim + 1) + 4 + tiles * 256 + 2 * T + 4 + 2 * head_dim) * 4 + (size_t)2 * T * (head_dim + 8) * 2
                                        : ((size_t)6 * T * (head_dim + 1) + 4 + tiles * 256 + 2 * T + 4 + 2 * head_dim) * 4;
  if (lds > 64 * 1024 || head_dim > 128 || T > 16 || head_dim % (dtype == MODE_BF16 ? 8 : 4)) return MODE_ERR_UNSUPPORTED;
  const uint32_t th = attn_thresh(p_drop); const float ik = 1.0f / (1.0f - p_drop);
  hipStream_t s = (hipStream_t)stream;
#define MODE_ATTN_BWD(T2, MF) hipLaunchKernelGGL((attn_bwd_kernel<T2, MF>), dim3(B * H), dim3(256), lds, s, (const T2*)qkv, q_gain, k_gain, (const T2*)dy, (T2*)dqkv, \
                                                 dgq_partial, dgk_partial, B, T, H, head_dim, eps, seed, th, ik, dbias_partial)
  if (dtype == MODE_BF16) { if (mf) MODE_ATTN_BWD(uint16_t, true); else MODE_ATTN_BWD(uint16_t, false); }
  else { if (mf) MODE_ATTN_BWD(float, true); else MODE_ATTN_BWD(float, false); }
#undef MODE_ATTN_BWD
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}
}  // namespace mode

extern "C" int mode_attn_block_bwd(const void* qkv, const float* q_gain, const float* k_gain, const void* dy, void* dqkv, float* dgq_partial,
                                   float* dgk_partial, int dtype, int B, int T, int H, int head_dim, float eps, uint32_t seed, float p_drop,
                                   void* stream) {
  return mode::attn_block_bwd_launch(qkv, q_gain, k_gain, dy, dqkv, dgq_partial, dgk_partial, dtype, B, T, H, head_dim, eps, seed, p_drop, nullptr, stream);
}
