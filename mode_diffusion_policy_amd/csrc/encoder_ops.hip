// FiLM-ResNet perceptual encoders: the producer of `state_images`.  The convolutions are conv_gemm.hip / mode_gemm
// (round 4: every one but the 3-channel 7 x 7 stem, which stays a MIOpen call through PyTorch's conv2d); this file is everything BETWEEN two convolutions, which the reference
// runs as 3-6 separate elementwise / reduction launches per block:
//
//   y = post_film( relu( pre_film( batch_norm(x) ) + residual ) )
//
//   batch_norm : nn.BatchNorm2d after every conv                      (resnets.py:41-45, timm / torchvision BasicBlock / Bottleneck)
//   pre_film   : v = gamma[n,c] * v + beta[n,c] after bn2              (BasicBlockWithModulation.forward, resnets.py:64-71)
//   residual   : out += identity ; relu                                (resnets.py:76-77)
//   post_film  : v = (1 + gamma[n,c]) * v + beta[n,c] after a stage    (FiLMLayer.forward, pretrained_resnets.py:19-23)
//
// Activations are NCHW (what conv2d produces): a (sample, channel) pair is one contiguous row of HW elements, FiLM and BN parameters are
// constant along it.  Every kernel gives one WAVE a row (lanes stride over HW: coalesced 256-byte segments), so the per-row reductions of
// the backward (FiLM gradients are sums over HW per (n, c)) are wave shuffles - no atomics, deterministic.  All of it is HBM-bound streaming:
// forward reads x (+ residual) and writes y once; backward = one reduction pass (6 sums per row) + one pass that writes dx / d residual.
#include "mode_common.h"
#include <algorithm>

using namespace mode;

namespace mode {

template <typename T>
__device__ __forceinline__ float ld(const T* p, long i);
template <>
__device__ __forceinline__ float ld<float>(const float* p, long i) { return p[i]; }
template <>
__device__ __forceinline__ float ld<uint16_t>(const uint16_t* p, long i) { return bf16_bits_to_f32(p[i]); }
template <typename T>
__device__ __forceinline__ void st(T* p, long i, float v);
template <>
__device__ __forceinline__ void st<float>(float* p, long i, float v) { p[i] = v; }
template <>
__device__ __forceinline__ void st<uint16_t>(uint16_t* p, long i, float v) { p[i] = f32_to_bf16_bits(v); }

// VEC consecutive elements of a row per lane and iteration (16-byte accesses: 8 bf16 / 4 fp32; 8-byte for rows whose length is only a multiple of 4;
// scalar otherwise - 7 x 7 maps).  One wave covers VEC * 64 elements per iteration.
template <typename T, int VEC>
__device__ __forceinline__ void ldv(const T* __restrict__ p, long i, float (&v)[VEC]) {
  if constexpr (VEC == 1) { v[0] = ld(p, i); }
  else if constexpr (sizeof(T) == 4) {
    static_assert(VEC == 4, "fp32 rows: 4 elements per access");
    const float4 t = *reinterpret_cast<const float4*>(p + i);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (VEC == 8) {
    const uint4 t = *reinterpret_cast<const uint4*>(p + i);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
  } else {
    const uint2 t = *reinterpret_cast<const uint2*>(p + i);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
}
template <typename T, int VEC>
__device__ __forceinline__ void stv(T* __restrict__ p, long i, const float (&v)[VEC]) {
  if constexpr (VEC == 1) { st(p, i, v[0]); }
  else if constexpr (sizeof(T) == 4) { *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]); }
  else if constexpr (VEC == 8) { *reinterpret_cast<uint4*>(p + i) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])); }
  else { *reinterpret_cast<uint2*>(p + i) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }
}

struct RowParams {            // per (n, c) constants of the fused chain
  float scale, shift, pg, pb, qg, qb;
  bool pre, post;
};
__device__ __forceinline__ RowParams row_params(const ModeBnFilmDesc& d, int n, int c) {
  RowParams r;
  if (d.scale) { r.scale = d.scale[c]; r.shift = d.shift[c]; }
  else {                                                        // eval-mode BatchNorm folded here: no separate launch for scale / shift
    const float is = 1.0f / sqrtf(d.bn_var[c] + d.bn_eps);
    r.scale = (d.bn_weight ? d.bn_weight[c] : 1.f) * is;
    r.shift = (d.bn_bias ? d.bn_bias[c] : 0.f) - d.bn_mean[c] * r.scale;
  }
  r.pre = d.pre_gamma != nullptr; r.post = d.post_gamma != nullptr;
  const long nc = (long)n * d.C + c;
  r.pg = r.pre ? d.pre_gamma[nc] : 1.f; r.pb = r.pre ? d.pre_beta[nc] : 0.f;
  r.qg = r.post ? d.post_gamma[nc] : 0.f; r.qb = r.post ? d.post_beta[nc] : 0.f;
  return r;
}

// ---- forward
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_film_act_fwd_kernel(const ModeBnFilmDesc d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave, rows = (long)d.N * d.C;
  if (row >= rows) return;
  const int n = (int)(row / d.C), c = (int)(row % d.C);
  const RowParams r = row_params(d, n, c);
  const T* x = reinterpret_cast<const T*>(d.x) + row * d.HW;
  const T* res = d.residual ? reinterpret_cast<const T*>(d.residual) + row * d.HW : nullptr;
  T* y = reinterpret_cast<T*>(d.y) + row * d.HW;
  for (int i = lane * VEC; i < d.HW; i += 64 * VEC) {
    float v[VEC], rv[VEC];
    ldv<T, VEC>(x, i, v);
    if (res) ldv<T, VEC>(res, i, rv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t = __builtin_fmaf(v[j], r.scale, r.shift);
      if (r.pre) t = __builtin_fmaf(r.pg, t, r.pb);
      if (res) t += rv[j];
      if (d.relu) t = fmaxf(t, 0.f);
      if (r.post) t = __builtin_fmaf(1.f + r.qg, t, r.qb);
      v[j] = t;
    }
    stv<T, VEC>(y, i, v);
  }
}

// ---- batch statistics (training-mode BatchNorm): per-row partial sums, then one thread per channel folds the N rows in double
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_row_sums_kernel(const T* __restrict__ x, long rows, int HW, float* __restrict__ psum, float* __restrict__ psq) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const T* p = x + row * HW;
  float s = 0.f, q = 0.f;
  for (int i = lane * VEC; i < HW; i += 64 * VEC) {
    float v[VEC];
    ldv<T, VEC>(p, i, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s += v[j]; q = __builtin_fmaf(v[j], v[j], q); }
  }
  s = wave_sum(s); q = wave_sum(q);
  if (lane == 0) { psum[row] = s; psq[row] = q; }
}
// Per-channel fold of per-row partials: a 1024-thread workgroup owns 64 channels, its 16 waves split the N rows (coalesced 256-byte reads of 64
// neighbouring channels), partials meet in LDS as doubles and are added in wave order: deterministic.  (One thread per channel walking all N rows,
// the first version, took 17 us per BatchNorm - a serial chain of N strided loads - and there are 106 BatchNorms per ResNet-50 pair.)
// per-channel folds of [rows][C] partials: a 1024-thread workgroup owns FC = 16 channels, its 64 row lanes split the rows (a wave reads 4 rows x 64 bytes),
// partials meet in LDS as doubles and are added in row-lane order: deterministic; C / 16 workgroups keep even a 64-channel layer on four CUs' worth of
// memory pipes (64 channels per workgroup - the previous shape - left a 256-channel fold on 4 workgroups: 9 us per launch, 212 launches per step)
constexpr int FC = 16, FR = 64;
__device__ __forceinline__ bool fold_pair(double& a, double& b, double* lds) {
  const int cl = threadIdx.x & (FC - 1), rl = threadIdx.x / FC;
  lds[(rl * FC + cl) * 2] = a; lds[(rl * FC + cl) * 2 + 1] = b;
  __syncthreads();
  if (rl != 0) return false;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < FR; ++k) { s += lds[(k * FC + cl) * 2]; q += lds[(k * FC + cl) * 2 + 1]; }
  a = s; b = q;
  return true;
}
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ psum, const float* __restrict__ psq, int N, int C, double count,
                                                           float* __restrict__ mean, float* __restrict__ var) {
  __shared__ double lds[2 * FC * FR];
  const int cl = threadIdx.x & (FC - 1), rl = threadIdx.x / FC;
  const int c = blockIdx.x * FC + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int n = rl; n < N; n += FR) { a += (double)psum[(long)n * C + c]; b += (double)psq[(long)n * C + c]; }
  if (!fold_pair(a, b, lds) || c >= C) return;
  const double m = a / count;
  mean[c] = (float)m;
  var[c] = (float)fmax(b / count - m * m, 0.0);                          // biased variance (what normalises the batch; nn.BatchNorm2d)
}

// Everything a BatchNorm needs before the fused pass, in one launch after the row sums: batch mean / biased variance (training) or the running
// statistics (eval), invstd, the folded scale = weight * invstd and shift = bias - mean * scale, and nn.BatchNorm2d's bookkeeping - running_mean /
// running_var (unbiased variance, momentum or cumulative average) and num_batches_tracked - updated in place.
__global__ __launch_bounds__(1024) void bn_prepare_kernel(const float* __restrict__ psum, const float* __restrict__ psq, int N, int C, double count, int training,
                                                          const float* __restrict__ weight, const float* __restrict__ bias, float eps, float momentum,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt,
                                                          float* __restrict__ mean, float* __restrict__ var, float* __restrict__ invstd, float* __restrict__ scale,
                                                          float* __restrict__ shift) {
  __shared__ double lds[2 * FC * FR];
  const int cl = threadIdx.x & (FC - 1), rl = threadIdx.x / FC;
  const int c = blockIdx.x * FC + cl;
  const bool valid = c < C;
  double m, v;
  if (training) {
    double s = 0.0, q = 0.0;
    if (valid)
      for (int n = rl; n < N; n += FR) { s += (double)psum[(long)n * C + c]; q += (double)psq[(long)n * C + c]; }
    if (!fold_pair(s, q, lds)) return;
    m = s / count;
    v = fmax(q / count - m * m, 0.0);                                      // biased variance (what normalises the batch; nn.BatchNorm2d)
  } else {
    if (rl != 0) return;
    m = valid ? (double)running_mean[c] : 0.0; v = valid ? (double)running_var[c] : 1.0;
  }
  if (!valid) return;
  const float mf = (float)m, vf = (float)v;
  const float is = 1.0f / sqrtf(vf + eps);
  const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
  const float sc = w * is;
  mean[c] = mf; var[c] = vf; invstd[c] = is; scale[c] = sc; shift[c] = b - mf * sc;
  if (training && running_mean) {
    // exponential moving average with `momentum`, or the cumulative average when momentum < 0 (nn.BatchNorm2d(momentum=None)): factor = 1 / batches seen
    const float f = momentum >= 0.f ? momentum : 1.0f / (float)((nbt ? *nbt : 0) + 1);
    const float unbias = (float)(count / fmax(count - 1.0, 1.0));
    running_mean[c] = running_mean[c] * (1.f - f) + mf * f;
    running_var[c] = running_var[c] * (1.f - f) + vf * unbias * f;
  }
  // num_batches_tracked += 1: here when nobody reads it (momentum given); the cumulative-average case counts in a launch of its own, after all reads
  if (training && nbt && momentum >= 0.f && blockIdx.x == 0 && cl == 0) *nbt += 1;
}
__global__ void bn_count_step_kernel(long long* nbt) { *nbt += 1; }          // num_batches_tracked += 1, after every channel block read it

// ---- backward, pass 1: six sums per row
//   sums[row] = { S dy*v4, S dy, S dv2*v1, S dv2, S dv1, S dv1*xhat }
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_film_act_bwd_sums_kernel(const ModeBnFilmDesc d, const T* __restrict__ dy, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, float* __restrict__ sums) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave, rows = (long)d.N * d.C;
  if (row >= rows) return;
  const int n = (int)(row / d.C), c = (int)(row % d.C);
  const RowParams r = row_params(d, n, c);
  const float mu = mean[c], is = invstd[c];
  const T* x = reinterpret_cast<const T*>(d.x) + row * d.HW;
  const T* res = d.residual ? reinterpret_cast<const T*>(d.residual) + row * d.HW : nullptr;
  const T* g = dy + row * d.HW;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f;
  for (int i = lane * VEC; i < d.HW; i += 64 * VEC) {
    float xv[VEC], gv[VEC], rv[VEC];
    ldv<T, VEC>(x, i, xv); ldv<T, VEC>(g, i, gv);
    if (res) ldv<T, VEC>(res, i, rv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float gy = gv[j];
      const float v1 = __builtin_fmaf(xv[j], r.scale, r.shift);
      const float v2 = r.pre ? __builtin_fmaf(r.pg, v1, r.pb) : v1;
      const float v3 = res ? v2 + rv[j] : v2;
      const float v4 = d.relu ? fmaxf(v3, 0.f) : v3;
      const float dv4 = r.post ? gy * (1.f + r.qg) : gy;
      const float dv2 = (d.relu && v3 <= 0.f) ? 0.f : dv4;
      const float dv1 = r.pre ? dv2 * r.pg : dv2;
      a0 = __builtin_fmaf(gy, v4, a0); a1 += gy; a2 = __builtin_fmaf(dv2, v1, a2); a3 += dv2; a4 += dv1;
      a5 = __builtin_fmaf(dv1, (xv[j] - mu) * is, a5);
    }
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3); a4 = wave_sum(a4); a5 = wave_sum(a5);
  if (lane == 0) {
    float* o = sums + row * 6;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5;
  }
}
// per-channel fold of the row sums (double; 64 channels per 1024-thread workgroup, the 16 waves split the rows), FiLM gradients are the row sums themselves
__global__ __launch_bounds__(1024) void bn_film_bwd_fold_kernel(const float* __restrict__ sums, int NS, int S, int C, float* __restrict__ dweight,
                                                                float* __restrict__ dbias, float* __restrict__ dpg, float* __restrict__ dpb, float* __restrict__ dqg,
                                                                float* __restrict__ dqb) {
  // sums: [NS = samples * S pixel splits][C][6]; FiLM gradients per (sample, channel) = the S splits added in order, BatchNorm affine gradients = all NS rows
  __shared__ double lds[2 * FC * FR];
  const int N = NS;
  const long total = (long)(NS / S) * C;
  if (dqg || dpg)
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < total; i += (long)gridDim.x * 1024) {
      const long smp = i / C, c_ = i % C;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
      for (int k = 0; k < S; ++k) {
        const float* s = sums + ((smp * S + k) * C + c_) * 6;
        t0 += s[0]; t1 += s[1]; t2 += s[2]; t3 += s[3];
      }
      if (dqg) { dqg[i] = t0; dqb[i] = t1; }
      if (dpg) { dpg[i] = t2; dpb[i] = t3; }
    }
  const int cl = threadIdx.x & (FC - 1), rl = threadIdx.x / FC;
  const int c = blockIdx.x * FC + cl;
  if (blockIdx.x * FC >= C) return;                                        // (whole workgroup: no barrier is skipped by part of it)
  double b = 0.0, w = 0.0;
  if (c < C)
    for (int n = rl; n < N; n += FR) { const float* s = sums + ((long)n * C + c) * 6; b += (double)s[4]; w += (double)s[5]; }
  if (fold_pair(b, w, lds) && c < C) { dbias[c] = (float)b; dweight[c] = (float)w; }
}
// pass 2: dx (training: through the batch statistics; eval: dv1 * scale) and d residual
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_film_act_bwd_dx_kernel(const ModeBnFilmDesc d, const T* __restrict__ dy, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ dweight,
                                                                 const float* __restrict__ dbias, int training, float inv_m, T* __restrict__ dx, T* __restrict__ dres) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave, rows = (long)d.N * d.C;
  if (row >= rows) return;
  const int n = (int)(row / d.C), c = (int)(row % d.C);
  const RowParams r = row_params(d, n, c);
  const float mu = mean[c], is = invstd[c];
  const float mb = training ? dbias[c] * inv_m : 0.f, mw = training ? dweight[c] * inv_m : 0.f;
  const T* x = reinterpret_cast<const T*>(d.x) + row * d.HW;
  const T* res = d.residual ? reinterpret_cast<const T*>(d.residual) + row * d.HW : nullptr;
  const T* g = dy + row * d.HW;
  T* ox = dx + row * d.HW;
  T* orr = dres ? dres + row * d.HW : nullptr;
  for (int i = lane * VEC; i < d.HW; i += 64 * VEC) {
    float xv[VEC], gv[VEC], rv[VEC], o1[VEC], o2[VEC];
    ldv<T, VEC>(x, i, xv); ldv<T, VEC>(g, i, gv);
    if (res) ldv<T, VEC>(res, i, rv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float v1 = __builtin_fmaf(xv[j], r.scale, r.shift);
      const float v2 = r.pre ? __builtin_fmaf(r.pg, v1, r.pb) : v1;
      const float v3 = res ? v2 + rv[j] : v2;
      const float dv4 = r.post ? gv[j] * (1.f + r.qg) : gv[j];
      const float dv2 = (d.relu && v3 <= 0.f) ? 0.f : dv4;
      const float dv1 = r.pre ? dv2 * r.pg : dv2;
      o2[j] = dv2;
      o1[j] = r.scale * (dv1 - mb - (xv[j] - mu) * is * mw);       // scale = weight * invstd
    }
    if (orr) stv<T, VEC>(orr, i, o2);
    stv<T, VEC>(ox, i, o1);
  }
}

// =====================================================================================================================
// channels_last (NHWC) activations: x[n][p][c] - what MIOpen's fast implicit-GEMM convolutions read and write natively (from NCHW tensors it wraps
// them in layout transposes: 10 % of the agent's training step).  BN / FiLM parameters now vary along the FASTEST axis: a thread owns W = 16 bytes of
// consecutive channels (8 bf16 / 4 fp32) of one pixel, a 256-thread workgroup = 8 such channel vectors (CT = 8 W channels: one 128-byte segment per
// pixel) x 32 pixel lanes, and walks the pixels [p0, p1) of ONE sample n (FiLM parameters are per sample).  grid = (channel tiles, N, S pixel
// splits).  The reductions (statistics, the six backward sums) fold the 32 pixel lanes by xor shuffles (lanes 8 apart hold the same channels) and
// LDS across the four waves, and write the SAME per-(row, channel) layout the NCHW kernels write with rows = N * S - the per-channel folds
// (bn_prepare_kernel, bn_film_bwd_fold_kernel) are shared.  C % W == 0 required.
template <typename T>
struct Nhwc { static constexpr int W = 16 / sizeof(T); static constexpr int CT = 8 * W; };

template <int W>
__device__ __forceinline__ void ldc(const float* __restrict__ p, long i, float (&v)[W]) {
#pragma unroll
  for (int j = 0; j < W; j += 4) { const float4 t = *reinterpret_cast<const float4*>(p + i + j); v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w; }
}
// fold K * W per-thread sums over the 32 pixel lanes of the workgroup; the result lands in threads 0..7 (pixel lane 0 of wave 0), returns true there
template <int KW>
__device__ __forceinline__ bool nhwc_fold(float (&a)[KW], float* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, oct = threadIdx.x & 7;
#pragma unroll
  for (int j = 0; j < KW; ++j) { a[j] += __shfl_xor(a[j], 8, 64); a[j] += __shfl_xor(a[j], 16, 64); a[j] += __shfl_xor(a[j], 32, 64); }
  if (lane < 8) {
#pragma unroll
    for (int j = 0; j < KW; ++j) lds[(wave * 8 + oct) * KW + j] = a[j];
  }
  __syncthreads();
  if (threadIdx.x >= 8) return false;
#pragma unroll
  for (int j = 0; j < KW; ++j) a[j] = lds[(0 * 8 + oct) * KW + j] + lds[(1 * 8 + oct) * KW + j] + lds[(2 * 8 + oct) * KW + j] + lds[(3 * 8 + oct) * KW + j];
  return true;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_nhwc_stats_kernel(const T* __restrict__ x, int C, int HW, int S, float* __restrict__ psum, float* __restrict__ psq) {
  constexpr int W = Nhwc<T>::W, CT = Nhwc<T>::CT;
  __shared__ float lds[4 * 8 * 2 * W];
  const int oct = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * CT + oct * W, n = blockIdx.y, sp = blockIdx.z;
  const int p0 = (int)((long)HW * sp / S), p1 = (int)((long)HW * (sp + 1) / S);
  float a[2 * W];
#pragma unroll
  for (int j = 0; j < 2 * W; ++j) a[j] = 0.f;
  if (c0 < C) {
    int p = p0 + pl;
    for (; p + 32 < p1; p += 64) {                                       // two pixels in flight per thread (the loop is one memory round trip per pass)
      float v[W], u[W];
      ldv<T, W>(x, ((long)n * HW + p) * C + c0, v); ldv<T, W>(x, ((long)n * HW + p + 32) * C + c0, u);
#pragma unroll
      for (int j = 0; j < W; ++j) { a[j] += v[j]; a[W + j] = __builtin_fmaf(v[j], v[j], a[W + j]); }
#pragma unroll
      for (int j = 0; j < W; ++j) { a[j] += u[j]; a[W + j] = __builtin_fmaf(u[j], u[j], a[W + j]); }
    }
    for (; p < p1; p += 32) {
      float v[W];
      ldv<T, W>(x, ((long)n * HW + p) * C + c0, v);
#pragma unroll
      for (int j = 0; j < W; ++j) { a[j] += v[j]; a[W + j] = __builtin_fmaf(v[j], v[j], a[W + j]); }
    }
  }
  if (nhwc_fold<2 * W>(a, lds) && c0 < C) {
    const long o = ((long)n * S + sp) * C + c0;
#pragma unroll
    for (int j = 0; j < W; ++j) { psum[o + j] = a[j]; psq[o + j] = a[W + j]; }
  }
}

template <int W>
struct ChanParams { float sc[W], sh[W], pg[W], pb[W], qg[W], qb[W]; bool pre, post; };
template <int W>
__device__ __forceinline__ void chan_params(const ModeBnFilmDesc& d, int n, int c0, ChanParams<W>& r) {
  if (d.scale) { ldc<W>(d.scale, c0, r.sc); ldc<W>(d.shift, c0, r.sh); }
  else {                                                        // eval-mode BatchNorm folded here (same expressions as bn_prepare_kernel)
    float mu[W], va[W];
    ldc<W>(d.bn_mean, c0, mu); ldc<W>(d.bn_var, c0, va);
#pragma unroll
    for (int j = 0; j < W; ++j) { r.sc[j] = 1.0f / sqrtf(va[j] + d.bn_eps); r.sh[j] = 0.f; }
    if (d.bn_weight) { float w[W]; ldc<W>(d.bn_weight, c0, w);
#pragma unroll
      for (int j = 0; j < W; ++j) r.sc[j] *= w[j]; }
    if (d.bn_bias) ldc<W>(d.bn_bias, c0, r.sh);
#pragma unroll
    for (int j = 0; j < W; ++j) r.sh[j] -= mu[j] * r.sc[j];
  }
  r.pre = d.pre_gamma != nullptr; r.post = d.post_gamma != nullptr;
  const long nc = (long)n * d.C + c0;
  if (r.pre) { ldc<W>(d.pre_gamma, nc, r.pg); ldc<W>(d.pre_beta, nc, r.pb); }
  if (r.post) { ldc<W>(d.post_gamma, nc, r.qg); ldc<W>(d.post_beta, nc, r.qb); }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_film_act_fwd_nhwc_kernel(const ModeBnFilmDesc d, int S) {
  constexpr int W = Nhwc<T>::W, CT = Nhwc<T>::CT;
  const int oct = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * CT + oct * W, n = blockIdx.y;
  if (c0 >= d.C) return;
  const int p0 = (int)((long)d.HW * blockIdx.z / S), p1 = (int)((long)d.HW * (blockIdx.z + 1) / S);
  ChanParams<W> r;
  chan_params<W>(d, n, c0, r);
  const T* x = reinterpret_cast<const T*>(d.x);
  const T* res = reinterpret_cast<const T*>(d.residual);
  T* y = reinterpret_cast<T*>(d.y);
  for (int p = p0 + pl; p < p1; p += 32) {
    const long o = ((long)n * d.HW + p) * d.C + c0;
    float v[W], rv[W];
    ldv<T, W>(x, o, v);
    if (res) ldv<T, W>(res, o, rv);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      float t = __builtin_fmaf(v[j], r.sc[j], r.sh[j]);
      if (r.pre) t = __builtin_fmaf(r.pg[j], t, r.pb[j]);
      if (res) t += rv[j];
      if (d.relu) t = fmaxf(t, 0.f);
      if (r.post) t = __builtin_fmaf(1.f + r.qg[j], t, r.qb[j]);
      v[j] = t;
    }
    stv<T, W>(y, o, v);
  }
}

// PLAIN = no FiLM on either side (49 of a FiLM-ResNet-50's 53 BatchNorms): four of the six sums are FiLM gradients nobody reads - only sum(dv) and
// sum(dv * xhat) are accumulated (zeros are written for the rest), and the pass is no longer VALU-bound (per 16 bytes of each operand: ~90 instead of ~200
// vector instructions; measured inside the agent step: see LABNOTES.md section 4 "Round 4").
template <typename T, bool PLAIN>
__global__ __launch_bounds__(256) void bn_film_act_bwd_sums_nhwc_kernel(const ModeBnFilmDesc d, const T* __restrict__ dy, const float* __restrict__ mean,
                                                                        const float* __restrict__ invstd, int S, float* __restrict__ sums) {
  constexpr int W = Nhwc<T>::W, CT = Nhwc<T>::CT;
  __shared__ float lds[4 * 8 * 6 * W];
  const int oct = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * CT + oct * W, n = blockIdx.y, sp = blockIdx.z;
  const int p0 = (int)((long)d.HW * sp / S), p1 = (int)((long)d.HW * (sp + 1) / S);
  float a[6 * W];
#pragma unroll
  for (int j = 0; j < 6 * W; ++j) a[j] = 0.f;
  if (c0 < d.C) {
    ChanParams<W> r;
    chan_params<W>(d, n, c0, r);
    float mu[W], is[W];
    ldc<W>(mean, c0, mu); ldc<W>(invstd, c0, is);
    const T* x = reinterpret_cast<const T*>(d.x);
    const T* res = reinterpret_cast<const T*>(d.residual);
    float xn[W], gn[W], rn[W];
    int p = p0 + pl;
    if (p < p1) {
      const long o = ((long)n * d.HW + p) * d.C + c0;
      ldv<T, W>(x, o, xn); ldv<T, W>(dy, o, gn);
      if (res) ldv<T, W>(res, o, rn);
    }
    for (; p < p1; p += 32) {
      float xv[W], gv[W], rv[W];
#pragma unroll
      for (int j = 0; j < W; ++j) { xv[j] = xn[j]; gv[j] = gn[j]; rv[j] = rn[j]; }
      if (p + 32 < p1) {                                                 // the next pixel's operands are requested before this pixel's arithmetic
        const long o = ((long)n * d.HW + p + 32) * d.C + c0;
        ldv<T, W>(x, o, xn); ldv<T, W>(dy, o, gn);
        if (res) ldv<T, W>(res, o, rn);
      }
      if constexpr (PLAIN) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const float v1 = __builtin_fmaf(xv[j], r.sc[j], r.sh[j]);
          const float v3 = res ? v1 + rv[j] : v1;
          const float dv = (d.relu && v3 <= 0.f) ? 0.f : gv[j];
          a[4 * W + j] += dv; a[5 * W + j] = __builtin_fmaf(dv, (xv[j] - mu[j]) * is[j], a[5 * W + j]);
        }
      } else {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const float gy = gv[j];
        const float v1 = __builtin_fmaf(xv[j], r.sc[j], r.sh[j]);
        const float v2 = r.pre ? __builtin_fmaf(r.pg[j], v1, r.pb[j]) : v1;
        const float v3 = res ? v2 + rv[j] : v2;
        const float v4 = d.relu ? fmaxf(v3, 0.f) : v3;
        const float dv4 = r.post ? gy * (1.f + r.qg[j]) : gy;
        const float dv2 = (d.relu && v3 <= 0.f) ? 0.f : dv4;
        const float dv1 = r.pre ? dv2 * r.pg[j] : dv2;
        a[0 * W + j] = __builtin_fmaf(gy, v4, a[0 * W + j]); a[1 * W + j] += gy; a[2 * W + j] = __builtin_fmaf(dv2, v1, a[2 * W + j]); a[3 * W + j] += dv2;
        a[4 * W + j] += dv1; a[5 * W + j] = __builtin_fmaf(dv1, (xv[j] - mu[j]) * is[j], a[5 * W + j]);
      }
      }
    }
  }
  if (nhwc_fold<6 * W>(a, lds) && c0 < d.C) {
    float* o = sums + (((long)n * S + sp) * d.C + c0) * 6;
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int k = 0; k < 6; ++k) o[j * 6 + k] = a[k * W + j];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_film_act_bwd_dx_nhwc_kernel(const ModeBnFilmDesc d, const T* __restrict__ dy, const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd, const float* __restrict__ dweight,
                                                                      const float* __restrict__ dbias, int training, float inv_m, int S, T* __restrict__ dx,
                                                                      T* __restrict__ dres) {
  constexpr int W = Nhwc<T>::W, CT = Nhwc<T>::CT;
  const int oct = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * CT + oct * W, n = blockIdx.y;
  if (c0 >= d.C) return;
  const int p0 = (int)((long)d.HW * blockIdx.z / S), p1 = (int)((long)d.HW * (blockIdx.z + 1) / S);
  ChanParams<W> r;
  chan_params<W>(d, n, c0, r);
  float mu[W], is[W], mb[W], mw[W];
  ldc<W>(mean, c0, mu); ldc<W>(invstd, c0, is);
  if (training) { ldc<W>(dbias, c0, mb); ldc<W>(dweight, c0, mw); }
#pragma unroll
  for (int j = 0; j < W; ++j) { mb[j] = training ? mb[j] * inv_m : 0.f; mw[j] = training ? mw[j] * inv_m : 0.f; }
  const T* x = reinterpret_cast<const T*>(d.x);
  const T* res = reinterpret_cast<const T*>(d.residual);
  for (int p = p0 + pl; p < p1; p += 32) {
    const long o = ((long)n * d.HW + p) * d.C + c0;
    float xv[W], gv[W], rv[W], o1[W], o2[W];
    ldv<T, W>(x, o, xv); ldv<T, W>(dy, o, gv);
    if (res) ldv<T, W>(res, o, rv);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const float v1 = __builtin_fmaf(xv[j], r.sc[j], r.sh[j]);
      const float v2 = r.pre ? __builtin_fmaf(r.pg[j], v1, r.pb[j]) : v1;
      const float v3 = res ? v2 + rv[j] : v2;
      const float dv4 = r.post ? gv[j] * (1.f + r.qg[j]) : gv[j];
      const float dv2 = (d.relu && v3 <= 0.f) ? 0.f : dv4;
      const float dv1 = r.pre ? dv2 * r.pg[j] : dv2;
      o2[j] = dv2;
      o1[j] = r.sc[j] * (dv1 - mb[j] - (xv[j] - mu[j]) * is[j] * mw[j]);       // scale = weight * invstd
    }
    if (dres) stv<T, W>(dres, o, o2);
    stv<T, W>(dx, o, o1);
  }
}

// pixel splits of a sample: enough workgroups to fill the part (early layers have few channel tiles), chunks of >= 64 pixels
static int nhwc_splits(int N, int C, int HW, int CT) {
  const long wg = (long)N * ((C + CT - 1) / CT);
  long S = std::max<long>(1, 768 / std::max<long>(wg, 1));
  S = std::min<long>(S, std::max(1, HW / 128));
  return (int)std::min<long>(S, 32);
}
static int nhwc_ct(int dtype) { return dtype == MODE_BF16 ? 64 : 32; }
static bool nhwc_ok(int dtype, int C, const void* a, const void* b, const void* c, const void* e, const void* f) {
  const uintptr_t al = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)e | (uintptr_t)f;
  return !(al & 15) && C % (dtype == MODE_BF16 ? 8 : 4) == 0;
}

// widest access the rows allow: every row starts at row * HW elements, so HW must be a multiple of the vector length (and the base 16-byte aligned)
static int bn_vec(int dtype, int HW, const void* a, const void* b, const void* c, const void* e, const void* f) {
  const uintptr_t al = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)e | (uintptr_t)f;
  if (al & 15) return 1;
  if (dtype == MODE_BF16 && HW % 8 == 0) return 8;
  if (HW % 4 == 0) return 4;
  return 1;
}
#define BN_DISPATCH(KERNEL, dt, vec, ...)                                                              \
  do {                                                                                                  \
    if ((dt) == MODE_F32) { if ((vec) >= 4) { KERNEL(float, 4, __VA_ARGS__); } else { KERNEL(float, 1, __VA_ARGS__); } }          \
    else if ((vec) == 8) { KERNEL(uint16_t, 8, __VA_ARGS__); }                                         \
    else if ((vec) == 4) { KERNEL(uint16_t, 4, __VA_ARGS__); }                                         \
    else { KERNEL(uint16_t, 1, __VA_ARGS__); }                                                         \
  } while (0)

static bool bn_desc_ok(const ModeBnFilmDesc* d) {
  if (!d || !d->x || !d->y || d->N < 0 || d->C <= 0 || d->HW <= 0) return false;
  if ((d->scale == nullptr) != (d->shift == nullptr)) return false;
  if (!d->scale && (!d->bn_mean || !d->bn_var)) return false;                      // either folded scale / shift, or the raw eval-mode BatchNorm
  if ((d->pre_gamma == nullptr) != (d->pre_beta == nullptr) || (d->post_gamma == nullptr) != (d->post_beta == nullptr)) return false;
  return d->dtype == MODE_F32 || d->dtype == MODE_BF16;
}

}  // namespace mode

extern "C" int mode_bn_film_act_fwd(const ModeBnFilmDesc* d, void* stream) {
  if (!bn_desc_ok(d)) return MODE_ERR_BAD_ARG;
  const long rows = (long)d->N * d->C;
  if (rows == 0) return MODE_OK;
  if (d->channels_last) {
    if (!nhwc_ok(d->dtype, d->C, d->x, d->residual, d->y, d->scale ? d->scale : d->bn_mean, d->scale ? d->shift : d->bn_var)) return MODE_ERR_UNSUPPORTED;
    const int CT = nhwc_ct(d->dtype), S = nhwc_splits(d->N, d->C, d->HW, CT);
    const dim3 g((d->C + CT - 1) / CT, d->N, S);
    if (d->dtype == MODE_BF16) hipLaunchKernelGGL(bn_film_act_fwd_nhwc_kernel<uint16_t>, g, dim3(256), 0, (hipStream_t)stream, *d, S);
    else hipLaunchKernelGGL(bn_film_act_fwd_nhwc_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, *d, S);
    MODE_LAUNCH_CHECK();
    return MODE_OK;
  }
  const dim3 grid((unsigned)((rows + 3) / 4));
  const int vec = bn_vec(d->dtype, d->HW, d->x, d->residual, d->y, nullptr, nullptr);
#define K_FWD(T, V, s_) hipLaunchKernelGGL((bn_film_act_fwd_kernel<T, V>), grid, dim3(256), 0, s_, *d)
  BN_DISPATCH(K_FWD, d->dtype, vec, (hipStream_t)stream);
#undef K_FWD
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

// rows of per-(row, channel) partials: N for NCHW (one wave per (n, c) row), N * pixel splits for channels_last
static long bn_rows(int N, int C, int HW, int dtype, int channels_last) { return channels_last ? (long)N * nhwc_splits(N, C, HW, nhwc_ct(dtype)) : (long)N; }
extern "C" size_t mode_bn_workspace_bytes(int N, int C, int HW, int dtype, int channels_last) {
  return N < 0 || C <= 0 || HW <= 0 ? 0 : (size_t)bn_rows(N, C, HW, dtype, channels_last) * C * 6 * 4 + 256;
}

// row sums (NCHW) / per-split channel sums (channels_last) into psum / psq [NS][C]
static int bn_partial_sums(const void* x, int dtype, int N, int C, int HW, int channels_last, float* psum_base, hipStream_t s, float** psum, float** psq, long* NS_out) {
  if (dtype != MODE_F32 && dtype != MODE_BF16) return MODE_ERR_BAD_ARG;
  const long NS = bn_rows(N, C, HW, dtype, channels_last);
  *NS_out = NS;
  *psum = psum_base; *psq = psum_base + NS * C;
  if (channels_last) {
    if (!nhwc_ok(dtype, C, x, nullptr, nullptr, nullptr, nullptr)) return MODE_ERR_UNSUPPORTED;
    const int CT = nhwc_ct(dtype), S = (int)(NS / N);
    const dim3 g((C + CT - 1) / CT, N, S);
    if (dtype == MODE_BF16) hipLaunchKernelGGL(bn_nhwc_stats_kernel<uint16_t>, g, dim3(256), 0, s, (const uint16_t*)x, C, HW, S, *psum, *psq);
    else hipLaunchKernelGGL(bn_nhwc_stats_kernel<float>, g, dim3(256), 0, s, (const float*)x, C, HW, S, *psum, *psq);
  } else {
    const long rows = (long)N * C;
    const dim3 grid((unsigned)((rows + 3) / 4));
    const int vec = bn_vec(dtype, HW, x, nullptr, nullptr, nullptr, nullptr);
#define K_SUM(T, V, s_) hipLaunchKernelGGL((bn_row_sums_kernel<T, V>), grid, dim3(256), 0, s_, (const T*)x, rows, HW, *psum, *psq)
    BN_DISPATCH(K_SUM, dtype, vec, s);
#undef K_SUM
  }
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_bn_stats(const void* x, int dtype, int N, int C, int HW, int channels_last, float* mean, float* var, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (!x || !mean || !var || !workspace || N <= 0 || C <= 0 || HW <= 0) return MODE_ERR_BAD_ARG;
  if (workspace_bytes < mode_bn_workspace_bytes(N, C, HW, dtype, channels_last)) return MODE_ERR_WORKSPACE;
  float *psum, *psq;
  long NS = N;
  if (int rc = bn_partial_sums(x, dtype, N, C, HW, channels_last, (float*)workspace, (hipStream_t)stream, &psum, &psq, &NS)) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, psum, psq, (int)NS, C, (double)N * (double)HW, mean, var);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_bn_prepare(const void* x, int dtype, int N, int C, int HW, int channels_last, const float* weight, const float* bias, float eps, float momentum,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* var, float* invstd, float* scale,
                               float* shift, void* workspace, size_t workspace_bytes, void* stream) {
  if (!mean || !var || !invstd || !scale || !shift || N <= 0 || C <= 0 || HW <= 0) return MODE_ERR_BAD_ARG;
  const int training = x != nullptr;
  if (!training && (!running_mean || !running_var)) return MODE_ERR_BAD_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return MODE_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* psum = nullptr; float* psq = nullptr;
  long NS = N;
  if (training) {
    if (!workspace || workspace_bytes < mode_bn_workspace_bytes(N, C, HW, dtype, channels_last)) return MODE_ERR_WORKSPACE;
    if (int rc = bn_partial_sums(x, dtype, N, C, HW, channels_last, (float*)workspace, s, &psum, &psq, &NS)) return rc;
  }
  hipLaunchKernelGGL(bn_prepare_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, psum, psq, (int)NS, C, (double)N * (double)HW, training, weight, bias, eps, momentum,
                     running_mean, running_var, (long long*)num_batches_tracked, mean, var, invstd, scale, shift);
  MODE_LAUNCH_CHECK();
  if (training && num_batches_tracked && momentum < 0.f) {
    hipLaunchKernelGGL(bn_count_step_kernel, dim3(1), dim3(1), 0, s, (long long*)num_batches_tracked);
    MODE_LAUNCH_CHECK();
  }
  return MODE_OK;
}

// Training statistics from partial sums somebody else produced (the convolution kernel's epilogue, mode_conv_bn_act_fwd: one row per 128-row output tile):
// the per-channel fold + bookkeeping of mode_bn_prepare without its pass over the activation.
extern "C" int mode_bn_prepare_partials(const float* psum, const float* psq, int rows, double count, int C, const float* weight, const float* bias, float eps,
                                        float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* var, float* invstd,
                                        float* scale, float* shift, void* stream) {
  if (!psum || !psq || !mean || !var || !invstd || !scale || !shift || rows <= 0 || C <= 0 || count <= 0.0) return MODE_ERR_BAD_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return MODE_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_prepare_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, psum, psq, rows, C, count, 1, weight, bias, eps, momentum, running_mean, running_var,
                     (long long*)num_batches_tracked, mean, var, invstd, scale, shift);
  MODE_LAUNCH_CHECK();
  if (num_batches_tracked && momentum < 0.f) {
    hipLaunchKernelGGL(bn_count_step_kernel, dim3(1), dim3(1), 0, s, (long long*)num_batches_tracked);
    MODE_LAUNCH_CHECK();
  }
  return MODE_OK;
}

extern "C" int mode_bn_film_act_bwd(const ModeBnFilmDesc* d, const void* dy, const float* mean, const float* invstd, int training, int phase, float inv_count,
                                    void* dx, void* dresidual, float* dweight, float* dbias, float* d_pre_gamma, float* d_pre_beta, float* d_post_gamma,
                                    float* d_post_beta, void* workspace, size_t workspace_bytes, void* stream) {
  if (!bn_desc_ok(d) || !d->scale || !dy || !mean || !invstd || !dx || !dweight || !dbias || !workspace) return MODE_ERR_BAD_ARG;
  if ((d->pre_gamma != nullptr) != (d_pre_gamma != nullptr && d_pre_beta != nullptr)) return MODE_ERR_BAD_ARG;
  if ((d->post_gamma != nullptr) != (d_post_gamma != nullptr && d_post_beta != nullptr)) return MODE_ERR_BAD_ARG;
  if ((d->residual != nullptr) != (dresidual != nullptr)) return MODE_ERR_BAD_ARG;
  if (workspace_bytes < mode_bn_workspace_bytes(d->N, d->C, d->HW, d->dtype, d->channels_last)) return MODE_ERR_WORKSPACE;
  const long rows = (long)d->N * d->C;
  if (rows == 0) return MODE_OK;
  if (phase < 0 || phase > 2) return MODE_ERR_BAD_ARG;
  float* sums = (float*)workspace;
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  const int vec = bn_vec(d->dtype, d->HW, d->x, d->residual, dy, dx, dresidual);
  const int cl = d->channels_last;
  if (cl && !nhwc_ok(d->dtype, d->C, d->x, d->residual, dy, dx, dresidual)) return MODE_ERR_UNSUPPORTED;
  const int CT = nhwc_ct(d->dtype), S = cl ? nhwc_splits(d->N, d->C, d->HW, CT) : 1;
  const dim3 g((d->C + CT - 1) / CT, d->N, S);
  if (phase != 2) {                                            // reductions: FiLM gradients per (n, c), BatchNorm affine gradients per channel
    if (cl) {
      const bool plain = !d->pre_gamma && !d->post_gamma;
      if (d->dtype == MODE_BF16) {
        if (plain) hipLaunchKernelGGL((bn_film_act_bwd_sums_nhwc_kernel<uint16_t, true>), g, dim3(256), 0, s, *d, (const uint16_t*)dy, mean, invstd, S, sums);
        else hipLaunchKernelGGL((bn_film_act_bwd_sums_nhwc_kernel<uint16_t, false>), g, dim3(256), 0, s, *d, (const uint16_t*)dy, mean, invstd, S, sums);
      } else {
        if (plain) hipLaunchKernelGGL((bn_film_act_bwd_sums_nhwc_kernel<float, true>), g, dim3(256), 0, s, *d, (const float*)dy, mean, invstd, S, sums);
        else hipLaunchKernelGGL((bn_film_act_bwd_sums_nhwc_kernel<float, false>), g, dim3(256), 0, s, *d, (const float*)dy, mean, invstd, S, sums);
      }
    } else {
#define K_BS(T, V, s_) hipLaunchKernelGGL((bn_film_act_bwd_sums_kernel<T, V>), grid, dim3(256), 0, s_, *d, (const T*)dy, mean, invstd, sums)
      BN_DISPATCH(K_BS, d->dtype, vec, s);
#undef K_BS
    }
    MODE_LAUNCH_CHECK();
    const long cblocks = (d->C + 15) / 16, eblocks = (d_pre_gamma || d_post_gamma) ? std::min<long>((rows + 1023) / 1024, 256) : 0;
    hipLaunchKernelGGL(bn_film_bwd_fold_kernel, dim3((unsigned)std::max(cblocks, eblocks)), dim3(1024), 0, s, sums, d->N * S, S, d->C, dweight, dbias, d_pre_gamma,
                       d_pre_beta, d_post_gamma, d_post_beta);
    MODE_LAUNCH_CHECK();
  }
  if (phase != 1) {                                            // dx / d residual from the (possibly cross-rank summed) channel sums
    const float inv_m = inv_count > 0.f ? inv_count : 1.f / ((float)d->N * (float)d->HW);
    if (cl) {
      if (d->dtype == MODE_BF16)
        hipLaunchKernelGGL(bn_film_act_bwd_dx_nhwc_kernel<uint16_t>, g, dim3(256), 0, s, *d, (const uint16_t*)dy, mean, invstd, dweight, dbias, training, inv_m, S,
                           (uint16_t*)dx, (uint16_t*)dresidual);
      else
        hipLaunchKernelGGL(bn_film_act_bwd_dx_nhwc_kernel<float>, g, dim3(256), 0, s, *d, (const float*)dy, mean, invstd, dweight, dbias, training, inv_m, S, (float*)dx,
                           (float*)dresidual);
    } else {
#define K_DX(T, V, s_) hipLaunchKernelGGL((bn_film_act_bwd_dx_kernel<T, V>), grid, dim3(256), 0, s_, *d, (const T*)dy, mean, invstd, dweight, dbias, training, inv_m, (T*)dx, (T*)dresidual)
      BN_DISPATCH(K_DX, d->dtype, vec, s);
#undef K_DX
    }
    MODE_LAUNCH_CHECK();
  }
  return MODE_OK;
}
