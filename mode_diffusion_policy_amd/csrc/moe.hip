// MoE routing tail and token-dispatch metadata (integer work; outputs must be bit-exact against the fp32 reference).
//   route_topk   : softmax / clamp / top-k / renormalise with wave-shuffle reductions over EP = pow2(E) lanes per row
//                  (modedit.py:345-349, 392, 398-399, 418-419)
//   dispatch_meta: canonical permutation of the reference's boolean-mask loop — experts ascending, token ids ascending
//                  inside an expert (modedit.py:561-566) — via ballot/popcount block scans; also emits the inverse map
//                  (token, ascending-expert slot) -> sorted row used by the combine kernels; the grouped GEMMs read `offsets`.
#include "mode_common.h"

namespace mode {

// ---- route: EP lanes cooperate on one row (EP = power of two >= E, <= 64)
__global__ __launch_bounds__(256) void route_topk_kernel(const float* __restrict__ logits, int R, int E, int EP, int k,
                                                         int normalize, float* shifted, float* probs, int* topk_idx,
                                                         float* topk_w) {
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = gtid / EP, e = gtid % EP;
  const bool rv = row < R;                      // whole EP-lane group is uniform in rv (EP divides 64)
  const bool ev = rv && e < E;
  float lg = ev ? logits[(long)row * E + e] : -INFINITY;
  float mx = lg;
  for (int o = EP >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float sh = lg - mx;                     // logits - rowmax   (temperature 1.0, modedit.py:345)
  const float ex = ev ? expf(sh) : 0.f;
  float sum = ex;
  for (int o = EP >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  float p = ex / sum;
  p = fminf(fmaxf(p, 1e-9f), 1.0f - 1e-9f);     // clamp(1e-9, 1-1e-9)  (:349)
  if (ev) {
    if (shifted) shifted[(long)row * E + e] = sh;
    if (probs) probs[(long)row * E + e] = p;
  }
  // k rounds of arg-max; ties -> lower expert id (torch.topk tie order is unspecified; exact ties are outside the contract)
  float cand = ev ? p : -1.f;
  float wsum = 0.f;
  float myw = 0.f; int myslot = -1;
  for (int j = 0; j < k; ++j) {
    float bv = cand; int bi = e;
    for (int o = EP >> 1; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    wsum += bv;
    if (bi == e) { myslot = j; myw = bv; cand = -2.f; }
  }
  if (ev && myslot >= 0) {
    if (topk_idx) topk_idx[(long)row * k + myslot] = e;
    if (topk_w) topk_w[(long)row * k + myslot] = normalize ? myw / wsum : myw;
  }
}

// ---- combine weights for host-chosen expert ids (training multinomial): w[n,j] = probs[row(n), idx[n,j]] (/ sum_j)
__global__ void weights_from_idx_kernel(const float* __restrict__ probs, const int* __restrict__ idx, int N, int tpr, int E, int k, int normalize,
                                        float* __restrict__ w) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* p = probs + (long)(n / tpr) * E;
  float s = 0.f;
  for (int j = 0; j < k; ++j) s += p[idx[(long)n * k + j]];
  for (int j = 0; j < k; ++j) { const float v = p[idx[(long)n * k + j]]; w[(long)n * k + j] = normalize ? v / s : v; }
}

// ---- training draw: k expert ids per token row WITHOUT replacement, distributed like torch.multinomial(probs, k, replacement=False)
// (modedit.py:390).  torch's own algorithm for that case is the exponential race: keys p_e / q_e with q ~ Exp(1) i.i.d., the k largest keys in
// order - the caller supplies q (torch.empty(N, E).exponential_(): the draw stays on torch's generator), this kernel does the race, the
// top-k and the combine weights of weights_from_idx_kernel in one pass (torch spends ~20 launches on validation + topk + sort there).
__global__ void sample_experts_kernel(const float* __restrict__ probs, const float* __restrict__ expo, int N, int tpr, int E, int k, int normalize,
                                      int* __restrict__ idx, float* __restrict__ w) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* p = probs + (long)(n / tpr) * E;
  const float* q = expo + (long)n * E;
  unsigned long long taken = 0ull;
  float s = 0.f;
  for (int j = 0; j < k; ++j) {
    float best = -1.f; int bi = 0;
    for (int e = 0; e < E; ++e) {
      if ((taken >> e) & 1ull) continue;
      const float key = p[e] / q[e];                                   // q == 0 -> +inf: wins, as it should; ties -> lower expert id
      if (key > best) { best = key; bi = e; }
    }
    taken |= 1ull << bi;
    idx[(long)n * k + j] = bi;
    s += p[bi];
  }
  for (int j = 0; j < k; ++j) { const float v = p[idx[(long)n * k + j]]; w[(long)n * k + j] = normalize ? v / s : v; }
}

// ---- training side channels of the routers (modedit.py:584-593, 816-820, 930-969): one workgroup per layer + a one-wave pass for the layer means:
//   frac[l,e]  = share of token rows routed to expert e            lb[l] = E * sum_e mean_n(rp[l,n,e]) * frac[l,e]   (rp = combine weights scattered)
//   zl[l]      = mean_r (log(sum_e exp(shifted[l,r,e]) + 1e-6))^2  lb_mean / zl_mean = means over the layers
//   mask[l,n,e] (optional) = 1 where token n uses expert e         usage[l,e] (optional, int64) += token rows per expert
// Fixed summation order (per-thread strided partials -> xor butterfly -> waves in order): deterministic, no atomics.  E <= 16.
__global__ __launch_bounds__(1024) void moe_aux_stats_kernel(const int* __restrict__ idx, const float* __restrict__ w, int L, int R, int tpr, int E, int k,
                                                             const float* __restrict__ shifted, int Rs, float* __restrict__ frac, float* __restrict__ lb,
                                                             float* __restrict__ zl, float* __restrict__ mask, long long* __restrict__ usage) {
  __shared__ float s_part[16][33];
  __shared__ float s_tot[33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long N = (long)R * tpr;
  {
    const int l = blockIdx.x;
    float cnt[16], ws[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { cnt[e] = 0.f; ws[e] = 0.f; }
    float z = 0.f;
    for (int r = tid; r < R; r += 1024) {
      const int* ir = idx + ((long)l * R + r) * k;
      const float* wr = w + ((long)l * R + r) * k;
      unsigned m = 0u;
      for (int j = 0; j < k; ++j) {
        const int e = ir[j];
        const float wv = wr[j];
        m |= 1u << e;
#pragma unroll
        for (int c = 0; c < 16; ++c) if (c == e) { cnt[c] += 1.f; ws[c] += wv; }
      }
      if (mask) {
        for (int t = 0; t < tpr; ++t) {
          float* mr = mask + ((long)l * N + (long)r * tpr + t) * E;
          for (int e = 0; e < E; ++e) mr[e] = ((m >> e) & 1u) ? 1.f : 0.f;
        }
      }
    }
    for (int r = tid; r < Rs; r += 1024) {
      const float* sr = shifted + ((long)l * Rs + r) * E;
      float se = 0.f;
      for (int e = 0; e < E; ++e) se += expf(sr[e]);
      const float lg = logf(se + 1e-6f);
      z += lg * lg;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) { cnt[e] = wave_sum(cnt[e]); ws[e] = wave_sum(ws[e]); }
    z = wave_sum(z);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { s_part[wave][e] = cnt[e]; s_part[wave][16 + e] = ws[e]; }
      s_part[wave][32] = z;
    }
    __syncthreads();
    if (tid < 33) {
      float t = 0.f;
      for (int wv = 0; wv < 16; ++wv) t += s_part[wv][tid];
      s_tot[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
      float acc = 0.f;
      for (int e = 0; e < E; ++e) {
        const float f = s_tot[e] / (float)R;                           // every routing row stands for tpr token rows: the ratio is the same
        frac[(long)l * E + e] = f;
        acc += (s_tot[16 + e] / (float)R) * f;
        if (usage) usage[(long)l * E + e] += (long long)(s_tot[e] + 0.5f) * tpr;
      }
      lb[l] = (float)E * acc; zl[l] = s_tot[32] / (float)Rs;
    }
  }
}

// means over the layers, summed in layer order
__global__ void moe_aux_means_kernel(const float* __restrict__ lb, const float* __restrict__ zl, int L, float* __restrict__ lb_mean,
                                     float* __restrict__ zl_mean) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float lb_acc = 0.f, zl_acc = 0.f;
  for (int l = 0; l < L; ++l) { lb_acc += lb[l]; zl_acc += zl[l]; }
  *lb_mean = lb_acc / (float)L; *zl_mean = zl_acc / (float)L;
}

// ---- dispatch metadata: one workgroup per problem (layer); blockDim = 1024
struct MetaBatch {
  const int* idx; const float* w; long idx_bstride;          // [R,k] per problem
  int* counts; int* offsets; int* perm; int* pos; float* posw; int* poffsets; int* prow; long out_bstride;  // strides in 4-byte words
};

__global__ __launch_bounds__(1024) void dispatch_meta_kernel(MetaBatch mb, int R, int tpr, int N, int E, int k) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  __shared__ int s_offsets[65];
  const long bo = (long)blockIdx.x * mb.out_bstride;
  const int* idx = mb.idx + (long)blockIdx.x * mb.idx_bstride;
  const float* w = mb.w + (long)blockIdx.x * mb.idx_bstride;
  int* counts = mb.counts + bo; int* offsets = mb.offsets + bo; int* perm = mb.perm + bo; int* pos = mb.pos + bo;
  float* posw = mb.posw + bo;
  int* poffsets = mb.poffsets ? mb.poffsets + bo : nullptr; int* prow = mb.prow ? mb.prow + bo : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;

  if (tid == 0) s_offsets[0] = 0;
  for (int e = 0; e < E; ++e) {
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < N; c0 += blockDim.x) {
      const int n = c0 + tid;
      int slot = -1, jasc = 0;
      if (n < N) {
        const int* ri = idx + (long)(n / tpr) * k;
        for (int j = 0; j < k; ++j) {
          const int v = ri[j];
          if (v == e) slot = j;
          jasc += (v < e) ? 1 : 0;                        // how many of this token's experts precede e (ascending order)
        }
      }
      const bool f = slot >= 0;
      const unsigned long long bal = __ballot(f);
      const int wrank = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave[wave] = __popcll(bal);
      __syncthreads();
      int wbase = 0, total = 0;
      for (int i = 0; i < nw; ++i) { const int c = s_wave[i]; if (i < wave) wbase += c; total += c; }
      const int base = s_base;
      if (f) {
        const int srow = s_offsets[e] + base + wbase + wrank;
        perm[srow] = n;
        pos[(long)n * k + jasc] = srow;
        posw[(long)n * k + jasc] = w[(long)(n / tpr) * k + slot];
      }
      __syncthreads();
      if (tid == 0) s_base = base + total;
    }
    __syncthreads();
    if (tid == 0) { counts[e] = s_base; s_offsets[e + 1] = s_offsets[e] + s_base; }
    __syncthreads();
  }
  if (tid == 0)
    for (int e = 0; e <= E; ++e) offsets[e] = s_offsets[e];
  if (poffsets) {
    // 64-padded layout for the weight-gradient GEMMs: expert e's rows start at poffsets[e] (multiple of 64); prow[p] = padded position
    __shared__ int s_poff[65];
    if (tid == 0) {
      int a = 0;
      for (int e = 0; e <= E; ++e) { s_poff[e] = a; poffsets[e] = a; if (e < E) a += (s_offsets[e + 1] - s_offsets[e] + 63) / 64 * 64; }
    }
    __syncthreads();
    for (int e = 0; e < E; ++e)
      for (int r = s_offsets[e] + tid; r < s_offsets[e + 1]; r += blockDim.x) prow[r] = s_poff[e] + (r - s_offsets[e]);
  }
}

// Router output layer for all layers at once: logits[l][r][e] = b3[l][e] + sum_n hid[r][l*K + n] * W3[l][e][n]  (Linear(2D, E), modedit.py:199).
// One wave per output element, float4 loads, fixed per-lane partial order + xor butterfly: deterministic and independent of R.
__global__ __launch_bounds__(256) void router_logits_kernel(const float* __restrict__ hid, long ld_hid, const float* __restrict__ w3, long w3_ls,
                                                            const float* __restrict__ b3, long b3_ls, int L, int R, int E, int K,
                                                            float* __restrict__ logits) {
  const int lane = threadIdx.x & 63;
  const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (long)L * R * E) return;
  const int e = (int)(o % E), r = (int)((o / E) % R), l = (int)(o / ((long)E * R));
  const float* x = hid + (long)r * ld_hid + (long)l * K;
  const float* w = w3 + (long)l * w3_ls + (long)e * K;
  float acc = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 xv = *reinterpret_cast<const float4*>(x + k);
    const float4 wv = *reinterpret_cast<const float4*>(w + k);
    acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) logits[o] = acc + b3[(long)l * b3_ls + e];
}

}  // namespace mode

using namespace mode;

extern "C" int mode_router_logits(const float* hid, int64_t ld_hid, const float* w3, int64_t w3_layer_stride, const float* b3, int64_t b3_layer_stride,
                                  int L, int R, int E, int K, float* logits, void* stream) {
  if (!hid || !w3 || !b3 || !logits || L <= 0 || R < 0 || E <= 0 || K <= 0) return MODE_ERR_BAD_ARG;
  if (K % 4 || ld_hid % 4 || w3_layer_stride % 4 || (((uintptr_t)hid | (uintptr_t)w3) & 15)) return MODE_ERR_UNSUPPORTED;
  if (R == 0) return MODE_OK;
  const long outs = (long)L * R * E;
  hipLaunchKernelGGL(router_logits_kernel, dim3((outs + 3) / 4), dim3(256), 0, (hipStream_t)stream, hid, (long)ld_hid, w3, (long)w3_layer_stride, b3,
                     (long)b3_layer_stride, L, R, E, K, logits);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_moe_route_topk_f32(const float* logits, int R, int E, int k, int normalize, float* shifted, float* probs,
                                       int32_t* topk_idx, float* topk_w, void* stream) {
  if (!logits || R < 0 || E <= 0 || k <= 0 || k > E) return MODE_ERR_BAD_ARG;
  if (E > 64) return MODE_ERR_UNSUPPORTED;
  if (R == 0) return MODE_OK;
  int EP = 1;
  while (EP < E) EP <<= 1;
  const long threads = (long)R * EP;
  hipLaunchKernelGGL(route_topk_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, R, E, EP, k, normalize,
                     shifted, probs, topk_idx, topk_w);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

namespace mode {
int dispatch_meta_batched(const MetaBatch& mb, int nbatch, int R, int tpr, int N, int E, int k, hipStream_t s) {
  if (E > 64 || k > E || tpr <= 0 || R * (long)tpr < N) return MODE_ERR_BAD_ARG;
  if (nbatch == 0) return MODE_OK;
  hipLaunchKernelGGL(dispatch_meta_kernel, dim3(nbatch), dim3(1024), 0, s, mb, R, tpr, N, E, k);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}
}  // namespace mode

extern "C" int mode_moe_dispatch_meta(const int32_t* idx, const float* w, int R, int tokens_per_row, int N, int E, int k, int32_t* counts,
                                      int32_t* offsets, int32_t* perm, int32_t* pos, float* posw, int32_t* poffsets, int32_t* prow,
                                      void* stream) {
  if (!idx || !w || !counts || !offsets || !perm || !pos || !posw || (!poffsets != !prow)) return MODE_ERR_BAD_ARG;
  MetaBatch mb{idx, w, 0, counts, offsets, perm, pos, posw, poffsets, prow, 0};
  return dispatch_meta_batched(mb, 1, R, tokens_per_row, N, E, k, (hipStream_t)stream);
}

extern "C" int mode_moe_sample_experts(const float* probs, const float* expo, int N, int tokens_per_row, int E, int k, int normalize, int32_t* idx,
                                       float* w, void* stream) {
  if (!probs || !expo || !idx || !w || N < 0 || tokens_per_row <= 0 || E <= 0 || k <= 0 || k > E) return MODE_ERR_BAD_ARG;
  if (E > 64) return MODE_ERR_UNSUPPORTED;
  if (N == 0) return MODE_OK;
  hipLaunchKernelGGL(sample_experts_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, probs, expo, N, tokens_per_row, E, k, normalize, idx, w);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_moe_aux_stats(const int32_t* idx, const float* w, int L, int R, int tokens_per_row, int E, int k, const float* shifted, int Rs,
                                  float* frac, float* lb, float* zl, float* lb_mean, float* zl_mean, float* mask, int64_t* usage, void* stream) {
  if (!idx || !w || !shifted || !frac || !lb || !zl || !lb_mean || !zl_mean || L <= 0 || R <= 0 || Rs <= 0 || tokens_per_row <= 0 || E <= 0 || k <= 0 || k > E)
    return MODE_ERR_BAD_ARG;
  if (E > 16) return MODE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(moe_aux_stats_kernel, dim3(L), dim3(1024), 0, (hipStream_t)stream, idx, w, L, R, tokens_per_row, E, k, shifted, Rs, frac, lb, zl, mask,
                     (long long*)usage);
  MODE_LAUNCH_CHECK();
  hipLaunchKernelGGL(moe_aux_means_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lb, zl, L, lb_mean, zl_mean);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_moe_weights_from_idx(const float* probs, const int32_t* idx, int N, int tokens_per_row, int E, int k, int normalize, float* w,
                                         void* stream) {
  if (!probs || !idx || !w || N < 0 || tokens_per_row <= 0 || E <= 0 || k <= 0 || k > E) return MODE_ERR_BAD_ARG;
  if (N == 0) return MODE_OK;
  hipLaunchKernelGGL(weights_from_idx_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, probs, idx, N, tokens_per_row, E, k, normalize, w);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}
