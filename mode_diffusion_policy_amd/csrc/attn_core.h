// The wave-level body of the tiny-T causal attention (see attn.hip for the layout story), shared by the stand-alone kernel (operands from the
// packed qkv buffer in global memory) and the fused QKV-projection + attention kernel (qkv_attn.hip: operands from the bf16 q | k | v tile the GEMM
// epilogue left in LDS).  ONE definition so that both produce the same bits: a sample's result must not depend on which chain its batch size selects
// (tests/test_gpu_model.py::test_c2_full_size_properties).
#pragma once
#include "mode_common.h"

namespace mode {

__device__ __forceinline__ uint32_t ahash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// attention-dropout keep mask (SDPA dropout_p, modedit.py:149): element (problem, query, key) of stream `seed`
__device__ __forceinline__ bool attn_keep(uint32_t seed, int prob, int T, int q, int k, uint32_t thresh) {
  return ahash_u32(ahash_u32((uint32_t)((prob * T + q) * T + k) ^ seed) + 0x9e3779b9U) >= thresh;
}


// Operand sources: q / k / v (token row, first dim) -> the 8 bf16 at dims d0 .. d0+7 of that row of this (sample, head).
struct AttnGlobalSrc {
  const uint16_t* base;   // qkv + (b * T) * ld + h * HD
  long ld; int D;
  __device__ __forceinline__ uint4 q(int row, int d0) const { return *reinterpret_cast<const uint4*>(base + (long)row * ld + d0); }
  __device__ __forceinline__ uint4 k(int row, int d0) const { return *reinterpret_cast<const uint4*>(base + (long)row * ld + D + d0); }
  __device__ __forceinline__ uint4 v(int row, int d0) const { return *reinterpret_cast<const uint4*>(base + (long)row * ld + 2L * D + d0); }
};

// One wave64 = one (sample, head): T <= 16 tokens, 32*(NKS-1) < HD <= 32*NKS, HD % 16 == 0 (dims past HD are zero k-slots).  `prob` only feeds the
// dropout hash.  y0 = output row of token 0 of this sample at this head's first column, ldy its row stride (elements).
template <int NKS, class Src>
__device__ __forceinline__ void attn_wave_bf16(const Src& src, const float* __restrict__ qg, const float* __restrict__ kg, uint16_t* __restrict__ y0,
                                               long ldy, int T, int HD, float eps, uint32_t seed, uint32_t thresh, float inv_keep, int prob, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  const bool tv = fr < T;
  const int rrow = tv ? fr : 0;

  // ---- V rows first (requested before anything else: the global latency hides under QK^T / softmax): FOUR 16-byte loads per lane, keys
  // fq*4 + j, dims fr*8 .. fr*8+7.  Branch-free (clamped address + select): as predicated 2-byte loads this block compiled into 16 serial
  // branch + s_waitcnt vmcnt(0) pairs, 7.4 us of dependent round trips per launch whatever the batch.
  uint4 vr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = fq * 4 + j;
    vr[j] = src.v(min(key, T - 1), min(fr * 8, HD - 8));   // masked where it is used
  }

  // ---- qk-norm gains: independent of everything else, fetched up front (their round trip used to follow the row-norm reduction)
  float4 gq4[NKS][2], gk4[NKS][2];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int dg = (ks * 32 + fq * 8 < HD) ? ks * 32 + fq * 8 : 0;      // padded k-slots carry zeros; any valid gain address works
    gq4[ks][0] = *reinterpret_cast<const float4*>(qg + dg); gq4[ks][1] = *reinterpret_cast<const float4*>(qg + dg + 4);
    gk4[ks][0] = *reinterpret_cast<const float4*>(kg + dg); gk4[ks][1] = *reinterpret_cast<const float4*>(kg + dg + 4);
  }

  // ---- load q / k fragments: token row fr, dims ks*32 + fq*8 + [0,8).  All loads of the kernel are requested here, ahead of any use; the
  // scheduling barrier keeps the compiler from sinking them to their use sites (it did: ~20 waits on partial results in a row)
  uint4 tq_[NKS], tk_[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int dq = (ks * 32 + fq * 8 < HD) ? ks * 32 : 0;     // clamped address + select below: no branch around the loads
    tq_[ks] = src.q(rrow, fq * 8 + dq); tk_[ks] = src.k(rrow, fq * 8 + dq);
  }
  // (an empty asm that "modifies" every loaded register: the loads cannot move below it, so the kernel has ONE wait for one round trip)
#define MODE_PIN4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#pragma unroll
  for (int j = 0; j < 4; ++j) MODE_PIN4(vr[j]);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    MODE_PIN4(gq4[ks][0]); MODE_PIN4(gq4[ks][1]); MODE_PIN4(gk4[ks][0]); MODE_PIN4(gk4[ks][1]); MODE_PIN4(tq_[ks]); MODE_PIN4(tk_[ks]);
  }
#undef MODE_PIN4
  float qf[NKS][8], kf[NKS][8];
  float qss = 0.f, kss = 0.f;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const bool okq = tv && ks * 32 + fq * 8 < HD;
    const uint4 rq = make_uint4(okq ? tq_[ks].x : 0u, okq ? tq_[ks].y : 0u, okq ? tq_[ks].z : 0u, okq ? tq_[ks].w : 0u);
    const uint4 rk = make_uint4(okq ? tk_[ks].x : 0u, okq ? tk_[ks].y : 0u, okq ? tk_[ks].z : 0u, okq ? tk_[ks].w : 0u);
    const uint32_t uq[4] = {rq.x, rq.y, rq.z, rq.w}, uk[4] = {rk.x, rk.y, rk.z, rk.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qf[ks][2 * i] = bf16_bits_to_f32(uq[i] & 0xffff); qf[ks][2 * i + 1] = bf16_bits_to_f32(uq[i] >> 16);
      kf[ks][2 * i] = bf16_bits_to_f32(uk[i] & 0xffff); kf[ks][2 * i + 1] = bf16_bits_to_f32(uk[i] >> 16);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { qss += qf[ks][i] * qf[ks][i]; kss += kf[ks][i] * kf[ks][i]; }
  }
  // row norm: the 4 lane groups (fq) of a token hold disjoint dims -> xor 16, 32
  qss += __shfl_xor(qss, 16, 64); qss += __shfl_xor(qss, 32, 64);
  kss += __shfl_xor(kss, 16, 64); kss += __shfl_xor(kss, 32, 64);
  const float qn = fmaxf(sqrtf(qss) * rsqrtf((float)HD), eps), kn = fmaxf(sqrtf(kss) * rsqrtf((float)HD), eps);
  const float rqn = __frcp_rn(qn), rkn = __frcp_rn(kn);      // one reciprocal per row (64 per-element fp32 divisions per lane were ~2 us of this kernel)

  bf16x8 qfrag[NKS], kfrag[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const float4 g0 = gq4[ks][0], g1 = gq4[ks][1], h0 = gk4[ks][0], h1 = gk4[ks][1];
    const float gq[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, gk[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    uint32_t pq[4], pk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pq[i] = pack_bf16x2(qf[ks][2 * i] * rqn * gq[2 * i], qf[ks][2 * i + 1] * rqn * gq[2 * i + 1]);
      pk[i] = pack_bf16x2(kf[ks][2 * i] * rkn * gk[2 * i], kf[ks][2 * i + 1] * rkn * gk[2 * i + 1]);
    }
    uint4 tq = make_uint4(pq[0], pq[1], pq[2], pq[3]), tk = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    qfrag[ks] = *reinterpret_cast<bf16x8*>(&tq);
    kfrag[ks] = *reinterpret_cast<bf16x8*>(&tk);
  }

  // ---- S^T[key][query] = sum_d K[key][d] Q[query][d]; lane: query = fr, keys = fq*4 + r
  f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag[ks], qfrag[ks], st, 0, 0, 0);
  const float scale = rsqrtf((float)HD);
  float s[4], mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = fq * 4 + r;
    s[r] = (key <= fr && key < T) ? st[r] * scale : -INFINITY;     // is_causal=True (modedit.py:149)
    mx = fmaxf(mx, s[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float p[4], sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { p[r] = (s[r] == -INFINITY) ? 0.f : __expf(s[r] - mx); sum += p[r]; }
  sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p[r] *= inv;
    if (thresh) p[r] = attn_keep(seed, prob, T, fr, fq * 4 + r, thresh) ? p[r] * inv_keep : 0.f;
  }
  uint4 tp = make_uint4(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), 0u, 0u);
  const bf16x8 pfrag = *reinterpret_cast<bf16x8*>(&tp);          // B operand: slots 0..3 = keys fq*4+r, slots 4..7 = 0

  // ---- O^T[row][query] = sum_key V[key][dim(row)] P[query][key].  MFMA i (0..7) takes as its 16 rows the dims fr*8 + i - element i of each
  // lane's four V vectors is its A operand (k-slots j < 4 = keys fq*4+j, slots 4..7 zero) - so that a lane ends up with 8 CONSECUTIVE dims
  // (fq*4+r)*8 .. +7 of its query per accumulator register r: four 16-byte stores.
  uint32_t vd[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool okv = fq * 4 + j < T && fr * 8 < HD;               // keys past T / dims past head_dim (clamped loads above) are zero operands
    vd[j][0] = okv ? vr[j].x : 0u; vd[j][1] = okv ? vr[j].y : 0u; vd[j][2] = okv ? vr[j].z : 0u; vd[j][3] = okv ? vr[j].w : 0u;
  }
  float ov[4][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int dw = i >> 1, sh = (i & 1) * 16;
    const uint32_t e0 = (vd[0][dw] >> sh) & 0xffffu, e1 = (vd[1][dw] >> sh) & 0xffffu, e2 = (vd[2][dw] >> sh) & 0xffffu, e3 = (vd[3][dw] >> sh) & 0xffffu;
    uint4 tvv = make_uint4(e0 | (e1 << 16), e2 | (e3 << 16), 0u, 0u);
    const bf16x8 vfrag = *reinterpret_cast<bf16x8*>(&tvv);
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pfrag, o, 0, 0, 0);     // D[row = fq*4 + r][col = query fr]
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r][i] = o[r];
  }
  uint16_t* yrow = y0 + (long)fr * ldy;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int d0 = (fq * 4 + r) * 8;
    if (tv && d0 < HD)
      *reinterpret_cast<uint4*>(yrow + d0) = make_uint4(pack_bf16x2(ov[r][0], ov[r][1]), pack_bf16x2(ov[r][2], ov[r][3]),
                                                        pack_bf16x2(ov[r][4], ov[r][5]), pack_bf16x2(ov[r][6], ov[r][7]));
  }
}

}  // namespace mode
