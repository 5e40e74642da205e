// Backward / training-side row kernels of the MoDE denoiser (HBM-bound, deterministic: no float atomics anywhere).
//   transpose          : [R,C] -> [C,R'] with optional row gather and per-row destination column (per-expert 64-padding), feeds the
//                        weight-gradient GEMMs (dW = dY^T X) in the K-contiguous layout the MFMA kernel wants
//   colsum             : bias / gain / conditioning gradients (segmented column sums), two deterministic stages
//   swiglu_fwd / _bwd  : SwishGLU + expert dropout (modedit.py:83-90, 254); mask is a counter-based hash of (seed, element) so the
//                        backward regenerates it instead of storing it
//   rmsnorm_bwd        : backward of x / max(rms, eps) * g (+ the MoE gather-sum of the expert input gradients fused in front)
//   combine_bwd        : backward of next += w * expert(x): dY (sorted rows) and the router-weight gradients <dy, E_e(u)>
#include "mode_common.h"

namespace mode {

// (hash_u32 / drop_keep - the dropout keep-mask - live in mode_common.h: the fused data-gradient epilogue of gemm_bf16_tr.hip uses them too)
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<uint16_t>(const uint16_t* p) { return bf16_bits_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<uint16_t>(uint16_t* p, float v) { *p = f32_to_bf16_bits(v); }

// -------------------------------------------------------------------------------------------------------------- transpose
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, long ld_src, int rows, int cols, T* __restrict__ dst,
                                                        long ld_dst, const int* __restrict__ src_rows, const int* __restrict__ dst_cols) {
  __shared__ T tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    if (r < rows && c < cols) {
      const long sr = src_rows ? src_rows[r] : r;
      tile[i][tx] = src[sr * ld_src + c];
    }
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) {
      const long dc = dst_cols ? dst_cols[r] : r;
      dst[(long)c * ld_dst + dc] = tile[tx][i];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ colsum
// stage 1: block (x = 64-column group, y = row split, z = segment) -> partial[z][y][col]
template <typename T>
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const T* __restrict__ X, long ld, int rows, int cols, const int* __restrict__ seg_off,
                                                            int seg_len, int nsplit, float* __restrict__ partial, float* out, int accumulate) {
  __shared__ float red[4][64];
  const int seg = blockIdx.z, sp = blockIdx.y;
  int o0, o1;
  if (seg_off) { o0 = seg_off[seg]; o1 = seg_off[seg + 1]; }
  else if (seg_len > 0) { o0 = seg * seg_len; o1 = min(rows, o0 + seg_len); }
  else { o0 = 0; o1 = rows; }
  const int chunk = (o1 - o0 + nsplit - 1) / nsplit;
  const int ra = o0 + sp * chunk, rb = min(o1, ra + chunk);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < cols) {
    // eight rows per trip, all loads requested before the first add (a serial "acc += load" loop is one memory round trip per row); the adds keep
    // the row order
    for (int r = ra + ty; r < rb; r += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld1<T>(X + (long)min(r + 4 * u, rb - 1) * ld + c);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r + 4 * u < rb) acc += v[u];
    }
  }
  red[ty][threadIdx.x & 63] = acc;
  __syncthreads();
  if (ty == 0 && c < cols) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (out) { float* o = out + (long)seg * cols + c; *o = accumulate ? *o + v : v; }            // single stage (nsplit == 1)
    else partial[((long)seg * nsplit + sp) * cols + c] = v;
  }
}
// Short fixed-length segments (the per-sample token sums of the conditioning gradient: 14 rows): one thread per (segment, 4 columns), every row of the
// segment requested before the first add; the additions follow stage 1's order (rows r % 4 = 0..3 in four chains, then (s0 + s1) + (s2 + s3)) - same bits.
template <int MAXR>
__global__ __launch_bounds__(256) void colsum_shortseg_kernel(const float* __restrict__ X, long ld, int rows, int cols, int seg_len, float* __restrict__ out,
                                                              int accumulate) {
  const int c4 = blockIdx.x * 256 + threadIdx.x, seg = blockIdx.y;
  if (c4 * 4 >= cols) return;
  const int o0 = seg * seg_len, n = min(rows, o0 + seg_len) - o0;
  float4 v[MAXR];
#pragma unroll
  for (int r = 0; r < MAXR; ++r) v[r] = *reinterpret_cast<const float4*>(X + (long)(o0 + min(r, n - 1)) * ld + c4 * 4);
  float4 a[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) a[t] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int r = 0; r < MAXR; ++r)
    if (r < n) { a[r & 3].x += v[r].x; a[r & 3].y += v[r].y; a[r & 3].z += v[r].z; a[r & 3].w += v[r].w; }
  float4 s4 = make_float4((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y), (a[0].z + a[1].z) + (a[2].z + a[3].z),
                          (a[0].w + a[1].w) + (a[2].w + a[3].w));
  float4* o = reinterpret_cast<float4*>(out + (long)seg * cols + c4 * 4);
  if (accumulate) { const float4 p = *o; s4.x = p.x + s4.x; s4.y = p.y + s4.y; s4.z = p.z + s4.z; s4.w = p.w + s4.w; }
  *o = s4;
}
// stage 2: out[seg][col] (+)= sum over splits in order
__global__ void colsum_stage2_kernel(const float* __restrict__ partial, int nseg, int nsplit, int cols, float* __restrict__ out, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nseg * cols) return;
  const int seg = i / cols, c = i % cols;
  float s = 0.f;
  for (int k0 = 0; k0 < nsplit; k0 += 8) {                               // eight partials requested per trip, added in split order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[((long)seg * nsplit + min(k0 + u, nsplit - 1)) * cols + c];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < nsplit) s += v[u];
  }
  out[i] = accumulate ? out[i] + s : s;
}

// ------------------------------------------------------------------------------------------------------------------ swiglu
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* __restrict__ P, T* __restrict__ Hd, long rows, int Hdim, uint32_t seed,
                                                         uint32_t thresh, float inv_keep) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;           // one thread per 4 output elements
  const long n4 = rows * Hdim / 4;
  if (i >= n4) return;
  const long r = (i * 4) / Hdim; const int c = (i * 4) % Hdim;
  const T* pr = P + r * 2 * Hdim;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = ld1<T>(pr + c + j), g = ld1<T>(pr + Hdim + c + j);
    float h = v * (g / (1.0f + __expf(-g)));
    if (thresh) h = drop_keep(seed, (uint64_t)(r * Hdim + c + j), thresh) ? h * inv_keep : 0.f;
    st1<T>(Hd + r * Hdim + c + j, h);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ P, const T* __restrict__ dHd, T* __restrict__ dP, long rows, int Hdim,
                                                         uint32_t seed, uint32_t thresh, float inv_keep) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n4 = rows * Hdim / 4;
  if (i >= n4) return;
  const long r = (i * 4) / Hdim; const int c = (i * 4) % Hdim;
  const T* pr = P + r * 2 * Hdim;
  T* dr = dP + r * 2 * Hdim;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = ld1<T>(pr + c + j), g = ld1<T>(pr + Hdim + c + j);
    float dh = ld1<T>(dHd + r * Hdim + c + j);
    if (thresh) dh = drop_keep(seed, (uint64_t)(r * Hdim + c + j), thresh) ? dh * inv_keep : 0.f;
    const float sg = 1.0f / (1.0f + __expf(-g));
    st1<T>(dr + c + j, dh * g * sg);                                     // d/d value = silu(gate)
    st1<T>(dr + Hdim + c + j, dh * v * sg * (1.0f + g * (1.0f - sg)));    // d/d gate  = value * silu'(gate)
  }
}

// ---- bf16 fast paths: 16-byte accesses (8 columns per thread).  The element loops above move 2 bytes per load and reach ~3 TB/s.
__global__ __launch_bounds__(256) void swiglu_fwd_bf16x8_kernel(const uint16_t* __restrict__ P, uint16_t* __restrict__ Hd, long rows, int Hdim,
                                                                uint32_t seed, uint32_t thresh, float inv_keep) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;                  // one thread per 8 output columns
  const int cpr = Hdim / 8;
  if (i >= rows * cpr) return;
  const long r = i / cpr; const int c = (int)(i % cpr) * 8;
  const uint4 pv = *reinterpret_cast<const uint4*>(P + r * 2 * Hdim + c), pg = *reinterpret_cast<const uint4*>(P + r * 2 * Hdim + Hdim + c);
  const uint32_t wv[4] = {pv.x, pv.y, pv.z, pv.w}, wg[4] = {pg.x, pg.y, pg.z, pg.w};
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float h[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float v = bf16_bits_to_f32(q ? wv[j] >> 16 : wv[j] & 0xffff), g = bf16_bits_to_f32(q ? wg[j] >> 16 : wg[j] & 0xffff);
      h[q] = v * (g / (1.0f + __expf(-g)));
      if (thresh) h[q] = drop_keep(seed, (uint64_t)(r * Hdim + c + 2 * j + q), thresh) ? h[q] * inv_keep : 0.f;
    }
    o[j] = pack_bf16x2(h[0], h[1]);
  }
  *reinterpret_cast<uint4*>(Hd + r * Hdim + c) = make_uint4(o[0], o[1], o[2], o[3]);
}

// SwishGLU(+dropout) backward fused with the per-expert bias gradient: a workgroup owns RB consecutive SORTED rows x 2048 value columns
// (+ their gate columns); every thread keeps running column sums of its 8 + 8 dP columns and flushes them to
// partial[row block][expert][2*Hdim] whenever the expert of its rows changes (rows are sorted by expert, so at most E flushes).  The
// caller reduces the [row blocks] axis with one small column sum: db1 costs 7 MB of traffic instead of re-reading the 59 MB dP.
template <int EMAX>
__global__ __launch_bounds__(256) void swiglu_bwd_bias_bf16x8_kernel(const uint16_t* __restrict__ P, const uint16_t* __restrict__ dHd,
                                                                     uint16_t* __restrict__ dP, long rows, int Hdim, uint32_t seed, uint32_t thresh,
                                                                     float inv_keep, const int* __restrict__ offsets, int E, int RB,
                                                                     float* __restrict__ partial) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= Hdim) return;
  const long r0 = (long)blockIdx.y * RB, r1 = min(rows, r0 + RB);
  int off[EMAX + 1];
#pragma unroll
  for (int e = 0; e <= EMAX; ++e) off[e] = offsets[min(e, E)];
  float sv[8], sg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sv[j] = 0.f; sg[j] = 0.f; }
  float* pbase = partial + (long)blockIdx.y * E * 2 * Hdim;
  auto flush = [&](int e, bool zero) {
    float* o = pbase + (long)e * 2 * Hdim;
    *reinterpret_cast<float4*>(o + c) = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(sv[0], sv[1], sv[2], sv[3]);
    *reinterpret_cast<float4*>(o + c + 4) = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(sv[4], sv[5], sv[6], sv[7]);
    *reinterpret_cast<float4*>(o + Hdim + c) = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(sg[0], sg[1], sg[2], sg[3]);
    *reinterpret_cast<float4*>(o + Hdim + c + 4) = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(sg[4], sg[5], sg[6], sg[7]);
  };
  int e = 0;
  while (e < E - 1 && r0 >= off[e + 1]) ++e;                             // expert of the first row
  for (int z = 0; z < e; ++z) flush(z, true);
  // rows go through in groups of four (eight is slower: 49 vs 45 us): the 12 operand loads of a group are requested before the first row is processed (a thread's serial
  // row loop was one memory round trip per row: 55 us per launch); rows past r1 re-read the last row and are skipped
  constexpr int RG = 4;
  for (long rb = r0; rb < r1; rb += RG) {
    uint4 pvq[RG], pgq[RG], dhq[RG];
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const long r = min(rb + u, r1 - 1);
      pvq[u] = *reinterpret_cast<const uint4*>(P + r * 2 * Hdim + c); pgq[u] = *reinterpret_cast<const uint4*>(P + r * 2 * Hdim + Hdim + c);
      dhq[u] = *reinterpret_cast<const uint4*>(dHd + r * Hdim + c);
    }
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const long r = rb + u;
      if (r >= r1) break;
      while (e < E - 1 && r >= off[e + 1]) {                              // segment boundary: hand the sums over, start the next expert
        flush(e, false);
#pragma unroll
        for (int j = 0; j < 8; ++j) { sv[j] = 0.f; sg[j] = 0.f; }
        ++e;
      }
      const uint4 pv = pvq[u], pg = pgq[u], dh4 = dhq[u];
      const uint32_t wv[4] = {pv.x, pv.y, pv.z, pv.w}, wg[4] = {pg.x, pg.y, pg.z, pg.w}, wd[4] = {dh4.x, dh4.y, dh4.z, dh4.w};
      uint32_t ov[4], og[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float dv[2], dg[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float v = bf16_bits_to_f32(q ? wv[j] >> 16 : wv[j] & 0xffff), g = bf16_bits_to_f32(q ? wg[j] >> 16 : wg[j] & 0xffff);
          float dh = bf16_bits_to_f32(q ? wd[j] >> 16 : wd[j] & 0xffff);
          if (thresh) dh = drop_keep(seed, (uint64_t)(r * Hdim + c + 2 * j + q), thresh) ? dh * inv_keep : 0.f;
          const float sgm = __builtin_amdgcn_rcpf(1.0f + __expf(-g));       // hardware reciprocal (1 ulp) like the forward's silu_f: `1.0f / x` is a ~10-instruction IEEE sequence
          dv[q] = dh * g * sgm;                                             // d/d value = silu(gate)
          dg[q] = dh * v * sgm * (1.0f + g * (1.0f - sgm));                  // d/d gate  = value * silu'(gate)
        }
        ov[j] = pack_bf16x2(dv[0], dv[1]); og[j] = pack_bf16x2(dg[0], dg[1]);
        // the bias gradient sums what the weight-gradient GEMM will read: the bf16-ROUNDED dP (as the separate column sum did)
        sv[2 * j] += bf16_bits_to_f32(ov[j] & 0xffff); sv[2 * j + 1] += bf16_bits_to_f32(ov[j] >> 16);
        sg[2 * j] += bf16_bits_to_f32(og[j] & 0xffff); sg[2 * j + 1] += bf16_bits_to_f32(og[j] >> 16);
      }
      *reinterpret_cast<uint4*>(dP + r * 2 * Hdim + c) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
      *reinterpret_cast<uint4*>(dP + r * 2 * Hdim + Hdim + c) = make_uint4(og[0], og[1], og[2], og[3]);
    }
  }
  flush(e, false);
  for (int z = e + 1; z < E; ++z) flush(z, true);
}

// ------------------------------------------------------------------------------------------------------------ rmsnorm bwd
// One wave per row.  dy[row] = (dy_a ? dy_a[row] : 0) + (dy_b ? dy_b[row] : 0) + sum_j G[pos[row*k+j]]   (all fp32)
// dx[row] (+)= (g*dy)/n - x * <g*dy, x> / (D n^3)   with n = max(||x||/sqrt(D), eps)  [clamped branch: dx = g*dy/eps]
// dg partial: dgp[block][d] = sum over the block's 4 rows of dy_d * x_d / n  (reduced by colsum stage 2)
// KK >= 0: top-k and the slab count GS of G are compile-time and every load of a row is unconditional (absent dy_a / dy_b read a dummy row and
// are masked): with run-time conditions around the loads the kernel compiled into 85 dependent load + s_waitcnt vmcnt(0) rounds (14.7 us per
// launch, 25 launches per training step).  KK = -1: any k / g_splits.
template <int NCH, int KK = -1, int GS = 1>   // NCH > 0: D == 256*NCH, chunk loops fully unrolled so that a phase's loads are all in flight together (else generic D % 4 == 0)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* dy_a,
                                                          const float* dy_b, const float* __restrict__ G, const int* __restrict__ pos, int k,
                                                          int rows, int D, float eps, float* dx, int accumulate, float* __restrict__ dgp,
                                                          float* __restrict__ dy_out, void* __restrict__ dx_lp, int lp_bf16, int g_splits,
                                                          long g_split_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = reinterpret_cast<float*>(smem);                       // [4][D] x rows
  float* sd = sx + 4 * D;                                            // [4][D] dy rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  const bool valid = row < rows;
  float ssq = 0.f, dot = 0.f;
  long prow[8];                                                      // gathered rows of G (the k expert copies), fetched once per row
#pragma unroll
  for (int j = 0; j < 8; ++j) prow[j] = (valid && j < k) ? (long)pos[(long)row * k + j] * D : 0;
  auto chunks = [&](auto&& f) {
    if constexpr (NCH > 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) f(lane * 4 + c * 256);
    } else {
      for (int d = lane * 4; d < D; d += 256) f(d);
    }
  };
  if (valid) {
    chunks([&](int d) {
      const float4 xv = *reinterpret_cast<const float4*>(x + (long)row * D + d);
      float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (KK >= 0) {
        const bool ha = dy_a != nullptr, hb = dy_b != nullptr;
        const float4 ta = *reinterpret_cast<const float4*>((ha ? dy_a : x) + (long)row * D + d);
        const float4 tb = *reinterpret_cast<const float4*>((hb ? dy_b : x) + (long)row * D + d);
        float4 tg[KK > 0 ? KK : 1][GS];
#pragma unroll
        for (int j = 0; j < KK; ++j)
#pragma unroll
          for (int z = 0; z < GS; ++z) tg[j][z] = *reinterpret_cast<const float4*>(G + (z < g_splits ? z : 0) * g_split_stride + prow[j] + d);
        dv.x += ha ? ta.x : 0.f; dv.y += ha ? ta.y : 0.f; dv.z += ha ? ta.z : 0.f; dv.w += ha ? ta.w : 0.f;
        dv.x += hb ? tb.x : 0.f; dv.y += hb ? tb.y : 0.f; dv.z += hb ? tb.z : 0.f; dv.w += hb ? tb.w : 0.f;
#pragma unroll
        for (int j = 0; j < KK; ++j)
#pragma unroll
          for (int z = 0; z < GS; ++z) {
            const bool on = z < g_splits;
            dv.x += on ? tg[j][z].x : 0.f; dv.y += on ? tg[j][z].y : 0.f; dv.z += on ? tg[j][z].z : 0.f; dv.w += on ? tg[j][z].w : 0.f;
          }
      } else {
      if (dy_a) { const float4 t = *reinterpret_cast<const float4*>(dy_a + (long)row * D + d); dv.x += t.x; dv.y += t.y; dv.z += t.z; dv.w += t.w; }
      if (dy_b) { const float4 t = *reinterpret_cast<const float4*>(dy_b + (long)row * D + d); dv.x += t.x; dv.y += t.y; dv.z += t.z; dv.w += t.w; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < k) {
          const float4 t = *reinterpret_cast<const float4*>(G + prow[j] + d);
          dv.x += t.x; dv.y += t.y; dv.z += t.z; dv.w += t.w;
          for (int z = 1; z < g_splits; ++z) {                       // K-slice slabs of the producing data-gradient GEMM
            const float4 t2 = *reinterpret_cast<const float4*>(G + z * g_split_stride + prow[j] + d);
            dv.x += t2.x; dv.y += t2.y; dv.z += t2.z; dv.w += t2.w;
          }
        }
      }
      }
      const float4 gv = *reinterpret_cast<const float4*>(g + d);
      *reinterpret_cast<float4*>(sx + wave * D + d) = xv;
      *reinterpret_cast<float4*>(sd + wave * D + d) = dv;
      if (dy_out) *reinterpret_cast<float4*>(dy_out + (long)row * D + d) = dv;
      ssq += xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
      dot += gv.x * dv.x * xv.x + gv.y * dv.y * xv.y + gv.z * dv.z * xv.z + gv.w * dv.w * xv.w;
    });
  }
  ssq = wave_sum(ssq); dot = wave_sum(dot);
  const float rms = sqrtf(ssq) * rsqrtf((float)D);
  const bool clamped = rms < eps;
  const float n = clamped ? eps : rms;
  const float rn = 1.0f / n;
  const float coef = clamped ? 0.f : dot * rn * rn * rn / (float)D;
  if (valid) {
    chunks([&](int d) {
      const float4 xv = *reinterpret_cast<const float4*>(sx + wave * D + d);
      const float4 dv = *reinterpret_cast<const float4*>(sd + wave * D + d);
      const float4 gv = *reinterpret_cast<const float4*>(g + d);
      float4 o = make_float4(gv.x * dv.x * rn - xv.x * coef, gv.y * dv.y * rn - xv.y * coef, gv.z * dv.z * rn - xv.z * coef,
                             gv.w * dv.w * rn - xv.w * coef);
      float* dst = dx + (long)row * D + d;
      if constexpr (KK >= 0) {
        const float4 t = *reinterpret_cast<const float4*>(dst);      // (dx is always valid memory; masked when the call overwrites)
        if (accumulate) { o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
      } else {
        if (accumulate) { const float4 t = *reinterpret_cast<const float4*>(dst); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
      }
      *reinterpret_cast<float4*>(dst) = o;
      if (dx_lp) {
        if (lp_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(dx_lp) + (long)row * D + d) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
        else *reinterpret_cast<float4*>(reinterpret_cast<float*>(dx_lp) + (long)row * D + d) = o;
      }
      // stash dy * x / n for the gain gradient in place of dy
      *reinterpret_cast<float4*>(sd + wave * D + d) = make_float4(dv.x * xv.x * rn, dv.y * xv.y * rn, dv.z * xv.z * rn, dv.w * xv.w * rn);
    });
  } else {
    for (int d = lane * 4; d < D; d += 256) *reinterpret_cast<float4*>(sd + wave * D + d) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  if (dgp) {
    for (int d = threadIdx.x; d < D; d += 256)
      dgp[(long)blockIdx.x * D + d] = (sd[d] + sd[D + d]) + (sd[2 * D + d] + sd[3 * D + d]);
  }
}

// ------------------------------------------------------------------------------------------------------------ combine bwd
// one wave per token: dYs[pos[t,j]] = posw[t,j] * dy[t]  (lp), dw[t,j] = <dy[t], Y[pos[t,j]]>.  Y may come as nsp split-K slabs (sp_stride elements
// apart) of the down-projection: added in slice order in fp32, like the forward combine adds them.
template <typename T>
__global__ __launch_bounds__(256) void combine_bwd_kernel(const float* __restrict__ dy, const T* __restrict__ Y, int nsp, long sp_stride,
                                                          const int* __restrict__ pos, const float* __restrict__ posw, int N, int D, int k,
                                                          T* __restrict__ dYs, float* __restrict__ dw) {
  const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= N) return;
  for (int j = 0; j < k; ++j) {
    const long p = pos[(long)t * k + j];
    const float w = posw[(long)t * k + j];
    float acc = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
      const float4 g = *reinterpret_cast<const float4*>(dy + (long)t * D + d);
      if constexpr (sizeof(T) == 2) {                               // bf16: one 8-byte load / store per lane instead of four 2-byte ones
        uint2 y2 = *reinterpret_cast<const uint2*>(Y + p * D + d);
        float y0 = bf16_bits_to_f32(y2.x & 0xffff), y1 = bf16_bits_to_f32(y2.x >> 16), y2f = bf16_bits_to_f32(y2.y & 0xffff), y3 = bf16_bits_to_f32(y2.y >> 16);
        for (int sp = 1; sp < nsp; ++sp) {
          y2 = *reinterpret_cast<const uint2*>(Y + sp * sp_stride + p * D + d);
          y0 += bf16_bits_to_f32(y2.x & 0xffff); y1 += bf16_bits_to_f32(y2.x >> 16); y2f += bf16_bits_to_f32(y2.y & 0xffff); y3 += bf16_bits_to_f32(y2.y >> 16);
        }
        acc += g.x * y0; acc += g.y * y1; acc += g.z * y2f; acc += g.w * y3;
        *reinterpret_cast<uint2*>(dYs + p * D + d) = make_uint2(pack_bf16x2(w * g.x, w * g.y), pack_bf16x2(w * g.z, w * g.w));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float gv = (&g.x)[c];
          float yv = ld1<T>(Y + p * D + d + c);
          for (int sp = 1; sp < nsp; ++sp) yv += ld1<T>(Y + sp * sp_stride + p * D + d + c);
          acc += gv * yv;
          st1<T>(dYs + p * D + d + c, w * gv);
        }
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) dw[(long)t * k + j] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------- small helpers
// dst[d0 + i*dstride (or didx[i])] (+)= src[s0 + i*sstride (or sidx[i])]  — strided / indexed fp32 row moves (token-type slices of the
// sequence gradient, scatter of the action-row gradients)
__global__ void rowcopy_f32_kernel(const float* __restrict__ src, long ld_s, int s0, int sstride, const int* __restrict__ sidx, float* dst,
                                   long ld_d, int d0, int dstride, const int* __restrict__ didx, const float* __restrict__ add, long ld_a, int n,
                                   int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * D) return;
  const int r = i / D, c = i % D;
  const long sr = sidx ? sidx[r] : (long)s0 + (long)r * sstride;
  const long dr = didx ? didx[r] : (long)d0 + (long)r * dstride;
  float v = src[sr * ld_s + c];
  if (add) v += add[(long)r * ld_a + c];
  dst[dr * ld_d + c] = v;
}

__global__ void gelu_fwd_kernel(const float* __restrict__ pre, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = gelu_erf_f(pre[i]);
}
__global__ void gelu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dout, float* __restrict__ dpre, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = pre[i];
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  dpre[i] = dout[i] * (cdf + x * pdf);
}

// ---------------------------------------------------------------------------------------------------------------------
// Router MLP backward for ALL layers at once (RouterCond.router = Linear(D,2D) -> GELU -> Dropout(0) -> Linear(2D,E), modedit.py:190-200):
//   dpre[b][l][n] = (sum_e dlog[l][b][e] * W3[l][e][n]) * gelu'(pre[b][l][n])          (one thread per 4 n)
//   dW3[l][e][n]  = sum_b dlog[l][b][e] * gelu(pre[b][l][n])                           (one thread per n, fixed b order -> deterministic)
// pre / dpre are [B][L][2D] so that dpre is directly the A operand of the batched weight/data-gradient GEMMs of the first Linear.
__global__ __launch_bounds__(256) void router_mlp_dpre_kernel(const float* __restrict__ dlog, const float* __restrict__ pre, const float* __restrict__ w3,
                                                              int L, int B, int E, int H2, float* __restrict__ dpre) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;              // index of a 4-element group in [B][L][H2]
  const long total4 = (long)B * L * H2 / 4;
  if (i4 >= total4) return;
  const long i = i4 * 4;
  const int n = (int)(i % H2);
  const int l = (int)((i / H2) % L);
  const int b = (int)(i / ((long)H2 * L));
  const float4 x = *reinterpret_cast<const float4*>(pre + i);
  float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = 0; e < E; ++e) {
    const float dl = dlog[((long)l * B + b) * E + e];
    const float4 w = *reinterpret_cast<const float4*>(w3 + ((long)l * E + e) * H2 + n);
    dh.x = fmaf(dl, w.x, dh.x); dh.y = fmaf(dl, w.y, dh.y); dh.z = fmaf(dl, w.z, dh.z); dh.w = fmaf(dl, w.w, dh.w);
  }
  auto gp = [](float v) {
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
    return cdf + v * pdf;
  };
  *reinterpret_cast<float4*>(dpre + i) = make_float4(dh.x * gp(x.x), dh.y * gp(x.y), dh.z * gp(x.z), dh.w * gp(x.w));
}

template <int EMAX>
__global__ __launch_bounds__(256) void router_w3_grad_kernel(const float* __restrict__ dlog, const float* __restrict__ pre, int L, int B, int E, int H2,
                                                             float* __restrict__ dw3) {
  const int n = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
  if (n >= H2) return;
  float acc[EMAX];
#pragma unroll
  for (int e = 0; e < EMAX; ++e) acc[e] = 0.f;
  for (int b0 = 0; b0 < B; b0 += 8) {                                    // eight samples' loads in flight at once; fma order = sample order
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = pre[((long)min(b0 + u, B - 1) * L + l) * H2 + n];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (b0 + u >= B) break;
      const float h = gelu_erf_f(x[u]);
      const float* dl = dlog + ((long)l * B + b0 + u) * E;
#pragma unroll
      for (int e = 0; e < EMAX; ++e)
        if (e < E) acc[e] = fmaf(dl[e], h, acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < EMAX; ++e)
    if (e < E) dw3[((long)l * E + e) * H2 + n] = acc[e];
}

__global__ void iota_scale_kernel(int* out, int n, int step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i * step;
}

// Router backward on distinct conditioning rows (one thread per sample): dw[t,j] (j in ascending expert id) -> dlogits[b,:]
// w = p[S]/sum(p[S]) (router_normalize) or p[S]; p = clamp(softmax(l), 1e-9, 1-1e-9)   (modedit.py:345-349, 398-399, 418-419)
template <int EMAX>
__global__ void router_bwd_kernel(const float* __restrict__ dw, const int* __restrict__ idx, const float* __restrict__ probs, int B, int T, int E,
                                  int k, int normalize, int idx_per_token, float* __restrict__ dlogits, const float* __restrict__ shifted,
                                  const float* __restrict__ lb_coef, const float* __restrict__ z_coef, int rows_per_layer) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float lbc[EMAX];                                                       // load-balancing coefficient of this row's layer, per expert
#pragma unroll
  for (int e = 0; e < EMAX; ++e) lbc[e] = (lb_coef && e < E) ? lb_coef[(long)(b / rows_per_layer) * E + e] : 0.f;
  float dp[EMAX], p[EMAX];                                             // compile-time indexed only: stays in registers
#pragma unroll
  for (int e = 0; e < EMAX; ++e) { dp[e] = 0.f; p[e] = e < E ? probs[(long)b * E + e] : 0.f; }
  auto pick = [&](const float* arr, int e) { float r = 0.f;
#pragma unroll
    for (int q = 0; q < EMAX; ++q) r = (q == e) ? arr[q] : r;
    return r; };
  for (int t = 0; t < T; ++t) {
    const long tok = (long)b * T + t;
    const int* id = idx + (idx_per_token ? tok : (long)b) * k;
    int es[8];
    float pe[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) es[j] = j < k ? id[j] : 0x7fffffff;
    // ascending expert id: slot j of dw belongs to the j-th smallest id (combine_bwd's convention); k <= 8 -> fixed compare-exchange network
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int c = 0; c + 1 < 8 - a; ++c) { const int lo = min(es[c], es[c + 1]), hi = max(es[c], es[c + 1]); es[c] = lo; es[c + 1] = hi; }
    float s = 0.f, wd = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { pe[j] = j < k ? pick(p, es[j]) : 0.f; s += pe[j]; }
    float add[8], dwj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dwj[j] = j < k ? dw[tok * k + j] + pick(lbc, es[j]) : 0.f;     // + d(gamma * LB) / d(router_probs[tok, e_j])
    if (normalize) {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < k) wd += (pe[j] / s) * dwj[j];
#pragma unroll
      for (int j = 0; j < 8; ++j) add[j] = j < k ? (dwj[j] - wd) / s : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) add[j] = dwj[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < EMAX; ++e) dp[e] += (j < k && es[j] == e) ? add[j] : 0.f;
  }
  float dot = 0.f;
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    if (e < E) {
      if (p[e] <= 1e-9f || p[e] >= 1.0f - 1e-9f) dp[e] = 0.f;          // clamp passes gradient only strictly inside
      dot += dp[e] * p[e];
    }
  }
  float dl[EMAX];
#pragma unroll
  for (int e = 0; e < EMAX; ++e) dl[e] = e < E ? p[e] * (dp[e] - dot) : 0.f;
  if (shifted && z_coef) {
    // router z-loss (modedit.py:930-969) on the max-shifted logits l: Z_row = log(sum_e exp(l_e) + 1e-6)^2;  the shift l = logits - max(logits)
    // is differentiated like autograd does: the arg-max column (first one on a tie, torch.max) receives minus the row sum
    float l[EMAX], ex[EMAX], S = 0.f;
    int amax = 0;
    float lmax = -3.4e38f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      l[e] = e < E ? shifted[(long)b * E + e] : -3.4e38f;
      ex[e] = e < E ? __expf(l[e]) : 0.f;
      S += ex[e];
      if (e < E && l[e] > lmax) { lmax = l[e]; amax = e; }
    }
    const float zc = z_coef[0] * __logf(S + 1e-6f) / (S + 1e-6f);
    float gsum = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) gsum += zc * ex[e];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) dl[e] += zc * ex[e] - (e == amax ? gsum : 0.f);
  }
#pragma unroll
  for (int e = 0; e < EMAX; ++e)
    if (e < E) dlogits[(long)b * E + e] = dl[e];
}

// ---- EDM preconditioning of the score-matching loss (score_wrappers.py:31-63)
__global__ __launch_bounds__(256) void edm_noise_scale_kernel(const float* __restrict__ action, const float* __restrict__ noise, const float* __restrict__ sigma,
                                                              float sd, int B, int n, float* __restrict__ xs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * n) return;
  const float s = sigma[i / n];
  const float c_in = 1.0f / __fsqrt_rn(s * s + sd * sd);
  xs[i] = (action[i] + noise[i] * s) * c_in;
}
// one workgroup, fixed summation order (deterministic): B * n is a few thousand elements
__global__ __launch_bounds__(1024) void edm_loss_kernel(const float* __restrict__ F, const float* __restrict__ action, const float* __restrict__ noise,
                                                        const float* __restrict__ sigma, float sd, int B, int n, float* __restrict__ loss, float* __restrict__ dF) {
  __shared__ float red[16];
  const long tot = (long)B * n;
  const float inv = 1.0f / (float)tot;
  float acc = 0.f;
  for (long i = threadIdx.x; i < tot; i += 1024) {
    const float s = sigma[i / n];
    const float s2 = s * s + sd * sd;
    const float c_skip = sd * sd / s2, c_out = s * sd / __fsqrt_rn(s2);
    const float noised = action[i] + noise[i] * s;
    const float target = (action[i] - c_skip * noised) / c_out;
    const float d = F[i] - target;
    acc += d * d;
    dF[i] = 2.0f * d * inv;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    loss[0] = t * inv;
  }
}

// pos_emb backward (modedit.py:760-790: row 0 is added to the goal token, row 1 to BOTH image tokens and the first action token, row 1+a
// to action token a): dpos[r][d] = sum over samples of the token gradients that row fed.  One thread per (row, d), fixed b order.
__global__ __launch_bounds__(256) void pos_emb_bwd_kernel(const float* __restrict__ dx0, int B, int T, int D, int t0, int n_img, int A_len,
                                                          float* __restrict__ dpos) {
  const int d = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (d >= D) return;
  const int t_img = t0 + 1, t_act = t_img + n_img;
  float acc = 0.f;
  const int tok = r == 0 ? t0 : (r == 1 ? t_act : t_act + r - 1), extra = r == 1 ? n_img : 0;
  for (int b0 = 0; b0 < B; b0 += 16) {                                   // sixteen samples' loads in flight at once; the adds keep the sample order
    float v[16], e0[16], e1[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = dx0[((long)min(b0 + u, B - 1) * T + tok) * D + d];
    if (extra > 0) {                                                     // (workgroup-uniform: whole blocks of loads, no load behind a per-lane condition)
#pragma unroll
      for (int u = 0; u < 16; ++u) e0[u] = dx0[((long)min(b0 + u, B - 1) * T + t_img) * D + d];
    }
    if (extra > 1) {
#pragma unroll
      for (int u = 0; u < 16; ++u) e1[u] = dx0[((long)min(b0 + u, B - 1) * T + t_img + 1) * D + d];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (b0 + u >= B) break;
      float x = v[u];
      if (extra > 0) x += e0[u];
      if (extra > 1) x += e1[u];
      for (int i = 2; i < extra; ++i) x += dx0[((long)(b0 + u) * T + t_img + i) * D + d];
      acc += x;
    }
  }
  dpos[(long)r * D + d] = acc;
}

// sigma_emb backward: e1[b,d] = s_b * w[d] + bias[d], s_b = ln(sigma_b)/4:  dw[d] = sum_b de1[b,d] s_b, dbias[d] = sum_b de1[b,d]
__global__ void sigma_embed_bwd_kernel(const float* __restrict__ de1, const float* __restrict__ sigma, int B, int D, float* __restrict__ dw,
                                       float* __restrict__ db) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float a = 0.f, c = 0.f;
  for (int b0 = 0; b0 < B; b0 += 16) {                                    // sixteen samples' loads in flight at once; the adds keep the sample order
    float g[16], sg[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int b = min(b0 + u, B - 1); g[u] = de1[(long)b * D + d]; sg[u] = sigma[b]; }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (b0 + u < B) { a += g[u] * (logf(sg[u]) / 4.0f); c += g[u]; }
  }
  dw[d] = a; db[d] = c;
}

}  // namespace mode

using namespace mode;

extern "C" int mode_transpose(const void* src, int64_t ld_src, int rows, int cols, void* dst, int64_t ld_dst, const int32_t* src_rows,
                              const int32_t* dst_cols, int dtype, void* stream) {
  if (!src || !dst || rows < 0 || cols < 0) return MODE_ERR_BAD_ARG;
  if (rows == 0 || cols == 0) return MODE_OK;
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (dtype == MODE_BF16)
    hipLaunchKernelGGL(transpose_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, (long)ld_src, rows, cols,
                       (uint16_t*)dst, (long)ld_dst, src_rows, dst_cols);
  else
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, (long)ld_src, rows, cols,
                       (float*)dst, (long)ld_dst, src_rows, dst_cols);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

static int colsum_nsplit(int rows, int cols, int nseg) {
  if (rows <= 512) return 1;                                      // one stage: every column group walks its rows itself
  int nsplit = 1;
  const long per = (long)((cols + 63) / 64) * nseg;
  while (per * nsplit < 1024 && nsplit * 64 < rows) nsplit *= 2;
  return nsplit;
}

extern "C" size_t mode_colsum_workspace_bytes(int rows, int cols, int nseg) {
  if (nseg < 1) nseg = 1;
  return (size_t)nseg * colsum_nsplit(rows, cols, nseg) * cols * 4 + 256;
}

extern "C" int mode_colsum(const void* X, int64_t ld, int rows, int cols, int dtype, const int32_t* seg_offsets, int seg_len, int nseg,
                           float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (!X || !out || rows < 0 || cols <= 0) return MODE_ERR_BAD_ARG;
  if (nseg < 1) nseg = 1;
  const int nsplit = colsum_nsplit(rows, cols, nseg);
  if (nsplit == 1 && dtype == MODE_F32 && !seg_offsets && seg_len > 0 && seg_len <= 16 && nseg > 1 && cols % 4 == 0 && ld % 4 == 0 &&
      (((uintptr_t)X | (uintptr_t)out) & 15) == 0 && (long)nseg * seg_len <= (long)rows + seg_len - 1) {
    hipLaunchKernelGGL(colsum_shortseg_kernel<16>, dim3((cols / 4 + 255) / 256, nseg), dim3(256), 0, (hipStream_t)stream, (const float*)X, (long)ld, rows, cols, seg_len, out,
                       accumulate);
    MODE_LAUNCH_CHECK();
    return MODE_OK;
  }
  if (nsplit > 1 && (!workspace || workspace_bytes < (size_t)nseg * nsplit * cols * 4)) return MODE_ERR_WORKSPACE;   // single-stage sums need none
  float* partial = (float*)workspace;
  const dim3 grid((cols + 63) / 64, nsplit, nseg);
  hipStream_t s = (hipStream_t)stream;
  float* direct = nsplit == 1 ? out : nullptr;
  if (dtype == MODE_BF16)
    hipLaunchKernelGGL(colsum_stage1_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)X, (long)ld, rows, cols, seg_offsets, seg_len, nsplit, partial, direct, accumulate);
  else
    hipLaunchKernelGGL(colsum_stage1_kernel<float>, grid, dim3(256), 0, s, (const float*)X, (long)ld, rows, cols, seg_offsets, seg_len, nsplit, partial, direct, accumulate);
  MODE_LAUNCH_CHECK();
  if (nsplit == 1) return MODE_OK;
  const long n = (long)nseg * cols;
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3((n + 255) / 256), dim3(256), 0, s, partial, nseg, nsplit, cols, out, accumulate);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

static inline uint32_t drop_thresh(float p) { return p <= 0.f ? 0u : (uint32_t)((double)p * 4294967296.0); }

extern "C" int mode_swiglu_fwd(const void* P, void* Hd, int64_t rows, int Hdim, int dtype, uint32_t seed, float p_drop, void* stream) {
  if (!P || !Hd || rows < 0 || Hdim <= 0 || (Hdim & 3) || p_drop < 0.f || p_drop >= 1.f) return MODE_ERR_BAD_ARG;
  const long n4 = rows * Hdim / 4;
  if (n4 == 0) return MODE_OK;
  const float ik = 1.0f / (1.0f - p_drop);
  if (dtype == MODE_BF16)
    if (Hdim % 8 == 0 && (((uintptr_t)P | (uintptr_t)Hd) & 15) == 0)
      hipLaunchKernelGGL(swiglu_fwd_bf16x8_kernel, dim3((unsigned)((rows * (Hdim / 8) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)P, (uint16_t*)Hd, (long)rows, Hdim, seed, drop_thresh(p_drop), ik);
    else
      hipLaunchKernelGGL(swiglu_fwd_kernel<uint16_t>, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)P, (uint16_t*)Hd, (long)rows, Hdim, seed, drop_thresh(p_drop), ik);
  else
    hipLaunchKernelGGL(swiglu_fwd_kernel<float>, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)P, (float*)Hd, (long)rows, Hdim, seed, drop_thresh(p_drop), ik);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_swiglu_bwd(const void* P, const void* dHd, void* dP, int64_t rows, int Hdim, int dtype, uint32_t seed, float p_drop,
                               void* stream) {
  if (!P || !dHd || !dP || rows < 0 || Hdim <= 0 || (Hdim & 3) || p_drop < 0.f || p_drop >= 1.f) return MODE_ERR_BAD_ARG;
  const long n4 = rows * Hdim / 4;
  if (n4 == 0) return MODE_OK;
  const float ik = 1.0f / (1.0f - p_drop);
  if (dtype == MODE_BF16)
    hipLaunchKernelGGL(swiglu_bwd_kernel<uint16_t>, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)P, (const uint16_t*)dHd, (uint16_t*)dP, (long)rows, Hdim, seed, drop_thresh(p_drop), ik);
  else
    hipLaunchKernelGGL(swiglu_bwd_kernel<float>, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)P, (const float*)dHd, (float*)dP, (long)rows, Hdim, seed, drop_thresh(p_drop), ik);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

namespace mode {
// G may come as g_splits K-slice slabs (g_split_stride elements apart) of the data-gradient GEMM that produced it (dit_train.hip)
int rmsnorm_bwd_launch(const float* x, const float* g, const float* dy_a, const float* dy_b, const float* G, int g_splits, long g_split_stride,
                       const int32_t* pos, int k, int rows, int D, float eps, float* dx, int accumulate, float* dg_partial, float* dy_out, void* dx_lp,
                       int lp_dtype, hipStream_t stream) {
  if (!x || !g || !dx || rows < 0 || D <= 0 || (D & 3) || (k > 0 && (!G || !pos)) || g_splits < 1) return MODE_ERR_BAD_ARG;
  if (rows == 0) return MODE_OK;
  const size_t lds = (size_t)8 * D * 4;
  if (lds > 64 * 1024) return MODE_ERR_UNSUPPORTED;
  if (k > 8) return MODE_ERR_UNSUPPORTED;
#define MODE_RB(KK, GS) hipLaunchKernelGGL((rmsnorm_bwd_kernel<4, KK, GS>), dim3((rows + 3) / 4), dim3(256), lds, stream, x, g, dy_a, dy_b, G, pos, k, rows, D, eps, dx, \
                                        accumulate, dg_partial, dy_out, dx_lp, lp_dtype == MODE_BF16 ? 1 : 0, g_splits, g_split_stride)
  if (D == 1024 && k <= 2 && g_splits <= 4) {                      // the chain's shapes: branch-free loads
    if (k == 0) MODE_RB(0, 1);
    else if (k == 1) { if (g_splits == 1) MODE_RB(1, 1); else if (g_splits == 2) MODE_RB(1, 2); else MODE_RB(1, 4); }
    else { if (g_splits == 1) MODE_RB(2, 1); else if (g_splits == 2) MODE_RB(2, 2); else MODE_RB(2, 4); }
  } else if (D == 1024)
#undef MODE_RB
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<4>, dim3((rows + 3) / 4), dim3(256), lds, stream, x, g, dy_a, dy_b, G, pos, k, rows, D, eps, dx,
                       accumulate, dg_partial, dy_out, dx_lp, lp_dtype == MODE_BF16 ? 1 : 0, g_splits, g_split_stride);
  else
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<0>, dim3((rows + 3) / 4), dim3(256), lds, stream, x, g, dy_a, dy_b, G, pos, k, rows, D, eps, dx,
                       accumulate, dg_partial, dy_out, dx_lp, lp_dtype == MODE_BF16 ? 1 : 0, g_splits, g_split_stride);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}
}  // namespace mode

extern "C" int mode_rmsnorm_bwd(const float* x, const float* g, const float* dy_a, const float* dy_b, const float* G, const int32_t* pos, int k,
                                int rows, int D, float eps, float* dx, int accumulate, float* dg_partial, float* dy_out, void* dx_lp,
                                int lp_dtype, void* stream) {
  return mode::rmsnorm_bwd_launch(x, g, dy_a, dy_b, G, 1, 0, pos, k, rows, D, eps, dx, accumulate, dg_partial, dy_out, dx_lp, lp_dtype, (hipStream_t)stream);
}

// bf16, D = 256 NIT, k <= KMAX, NSP split-K slabs of Y: the slot indices, then every Y / dy load of the token are requested before the first use (the generic kernel
// walks j and d serially: k x D/256 dependent round trips, 9 us for 7 MB at C2); per (t, j) the same lane-strided fma chain + butterfly as above, slabs added in
// slice order - same bits.
template <int NIT, int KMAX, int NSP>
__global__ __launch_bounds__(256) void combine_bwd_bf16_kernel(const float* __restrict__ dy, const uint16_t* __restrict__ Y, long sp_stride, const int* __restrict__ pos,
                                                               const float* __restrict__ posw, int N, int k, uint16_t* __restrict__ dYs, float* __restrict__ dw) {
  constexpr int D = 256 * NIT;
  const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= N) return;
  long pj[KMAX]; float wj[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { const int jj = min(j, k - 1); pj[j] = pos[(long)t * k + jj]; wj[j] = posw[(long)t * k + jj]; }
  float4 g[NIT]; uint2 y[KMAX][NSP][NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) g[it] = *reinterpret_cast<const float4*>(dy + (long)t * D + it * 256 + lane * 4);
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
#pragma unroll
    for (int sp = 0; sp < NSP; ++sp)
#pragma unroll
      for (int it = 0; it < NIT; ++it) y[j][sp][it] = *reinterpret_cast<const uint2*>(Y + sp * sp_stride + pj[j] * D + it * 256 + lane * 4);
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j >= k) break;
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const float4 gv = g[it];
      float y0 = bf16_bits_to_f32(y[j][0][it].x & 0xffff), y1 = bf16_bits_to_f32(y[j][0][it].x >> 16);
      float y2 = bf16_bits_to_f32(y[j][0][it].y & 0xffff), y3 = bf16_bits_to_f32(y[j][0][it].y >> 16);
#pragma unroll
      for (int sp = 1; sp < NSP; ++sp) {
        y0 += bf16_bits_to_f32(y[j][sp][it].x & 0xffff); y1 += bf16_bits_to_f32(y[j][sp][it].x >> 16);
        y2 += bf16_bits_to_f32(y[j][sp][it].y & 0xffff); y3 += bf16_bits_to_f32(y[j][sp][it].y >> 16);
      }
      acc += gv.x * y0; acc += gv.y * y1; acc += gv.z * y2; acc += gv.w * y3;
      *reinterpret_cast<uint2*>(dYs + pj[j] * D + it * 256 + lane * 4) = make_uint2(pack_bf16x2(wj[j] * gv.x, wj[j] * gv.y), pack_bf16x2(wj[j] * gv.z, wj[j] * gv.w));
    }
    acc = wave_sum(acc);
    if (lane == 0) dw[(long)t * k + j] = acc;
  }
}

namespace mode {
int combine_bwd_launch(const float* dy, const void* Y, int y_dtype, int y_splits, long y_split_stride, const int32_t* pos, const float* posw, int N, int D, int k,
                       void* dYs, float* dw, void* stream) {
  if (!dy || !Y || !pos || !posw || !dYs || !dw || N < 0 || D <= 0 || (D & 3) || k <= 0 || y_splits < 1) return MODE_ERR_BAD_ARG;
  if (N == 0) return MODE_OK;
  if (y_dtype == MODE_BF16 && (y_splits == 1 || y_splits == 4) && k <= 2 && (D == 1024 || D == 512 || D == 256) &&
      ((((uintptr_t)dy) & 15) | (((uintptr_t)Y | (uintptr_t)dYs) & 7)) == 0 && y_split_stride % 4 == 0) {
    const dim3 grid((N + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define MODE_CB(NIT, NSP) hipLaunchKernelGGL((combine_bwd_bf16_kernel<NIT, 2, NSP>), grid, blk, 0, s, dy, (const uint16_t*)Y, (long)y_split_stride, pos, posw, N, k, (uint16_t*)dYs, dw)
    if (y_splits == 1) { if (D == 1024) MODE_CB(4, 1); else if (D == 512) MODE_CB(2, 1); else MODE_CB(1, 1); }
    else { if (D == 1024) MODE_CB(4, 4); else if (D == 512) MODE_CB(2, 4); else MODE_CB(1, 4); }
#undef MODE_CB
    MODE_LAUNCH_CHECK();
    return MODE_OK;
  }
  if (y_dtype == MODE_BF16)
    hipLaunchKernelGGL(combine_bwd_kernel<uint16_t>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, (const uint16_t*)Y, y_splits, y_split_stride, pos, posw, N, D,
                       k, (uint16_t*)dYs, dw);
  else
    hipLaunchKernelGGL(combine_bwd_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, (const float*)Y, y_splits, y_split_stride, pos, posw, N, D, k,
                       (float*)dYs, dw);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}
}  // namespace mode

extern "C" int mode_moe_combine_bwd(const float* dy, const void* Y, int y_dtype, const int32_t* pos, const float* posw, int N, int D, int k,
                                    void* dYs, float* dw, void* stream) {
  return mode::combine_bwd_launch(dy, Y, y_dtype, 1, 0, pos, posw, N, D, k, dYs, dw, stream);
}

extern "C" int mode_rowcopy_f32(const float* src, int64_t ld_src, int s0, int sstride, const int32_t* sidx, float* dst, int64_t ld_dst, int d0,
                                int dstride, const int32_t* didx, const float* add, int64_t ld_add, int n, int D, void* stream) {
  if (!src || !dst || n < 0 || D <= 0) return MODE_ERR_BAD_ARG;
  const long tot = (long)n * D;
  if (tot == 0) return MODE_OK;
  hipLaunchKernelGGL(rowcopy_f32_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, (long)ld_src, s0, sstride, sidx, dst,
                     (long)ld_dst, d0, dstride, didx, add, (long)ld_add, n, D);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_gelu_fwd(const float* pre, float* out, int64_t n, void* stream) {
  if (!pre || !out || n < 0) return MODE_ERR_BAD_ARG;
  if (n == 0) return MODE_OK;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pre, out, (long)n);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_gelu_bwd(const float* pre, const float* dout, float* dpre, int64_t n, void* stream) {
  if (!pre || !dout || !dpre || n < 0) return MODE_ERR_BAD_ARG;
  if (n == 0) return MODE_OK;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pre, dout, dpre, (long)n);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_moe_router_bwd(const float* dw, const int32_t* idx, const float* probs, int B, int T, int E, int k, int normalize,
                                   int idx_per_token, float* dlogits, void* stream) {
  if (!dw || !idx || !probs || !dlogits || B < 0 || E <= 0 || E > 64 || k <= 0 || k > 8 || k > E) return MODE_ERR_BAD_ARG;
  if (B == 0) return MODE_OK;
  return mode_moe_router_bwd_aux(dw, idx, probs, nullptr, nullptr, nullptr, B, B, T, E, k, normalize, idx_per_token, dlogits, stream);
}

extern "C" int mode_moe_router_bwd_aux(const float* dw, const int32_t* idx, const float* probs, const float* shifted, const float* lb_coef,
                                       const float* z_coef, int B, int rows_per_layer, int T, int E, int k, int normalize, int idx_per_token,
                                       float* dlogits, void* stream) {
  if (!dw || !idx || !probs || !dlogits || B < 0 || E <= 0 || E > 64 || k <= 0 || k > 8 || k > E) return MODE_ERR_BAD_ARG;
  if ((lb_coef || z_coef) && (rows_per_layer <= 0 || B % rows_per_layer)) return MODE_ERR_BAD_ARG;
  if (z_coef && !shifted) return MODE_ERR_BAD_ARG;
  if (rows_per_layer <= 0) rows_per_layer = B > 0 ? B : 1;
  if (B == 0) return MODE_OK;
#define MODE_RB(EM) hipLaunchKernelGGL(router_bwd_kernel<EM>, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, dw, idx, probs, B, T, E, k, normalize, idx_per_token, dlogits, shifted, lb_coef, z_coef, rows_per_layer)
  if (E <= 4) MODE_RB(4); else if (E <= 8) MODE_RB(8); else if (E <= 16) MODE_RB(16); else MODE_RB(64);
#undef MODE_RB
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_edm_noise_scale(const float* action, const float* noise, const float* sigma, float sigma_data, int B, int n, float* x_scaled, void* stream) {
  if (!action || !noise || !sigma || !x_scaled || B < 0 || n <= 0) return MODE_ERR_BAD_ARG;
  if (B == 0) return MODE_OK;
  hipLaunchKernelGGL(edm_noise_scale_kernel, dim3((unsigned)(((long)B * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, action, noise, sigma, sigma_data, B, n, x_scaled);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_edm_loss(const float* F, const float* action, const float* noise, const float* sigma, float sigma_data, int B, int n, float* loss,
                             float* dF, void* stream) {
  if (!F || !action || !noise || !sigma || !loss || !dF || B <= 0 || n <= 0) return MODE_ERR_BAD_ARG;
  hipLaunchKernelGGL(edm_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, F, action, noise, sigma, sigma_data, B, n, loss, dF);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_sigma_embed_bwd(const float* de1, const float* sigma, int B, int D, float* dw, float* db, void* stream) {
  if (!de1 || !sigma || !dw || !db) return MODE_ERR_BAD_ARG;
  if (D == 0) return MODE_OK;
  hipLaunchKernelGGL(sigma_embed_bwd_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, de1, sigma, B, D, dw, db);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused AdamW over a flat fp32 arena slice (replaces torch.optim.AdamW's multi-tensor loop for mode_agent.py:365-384's two groups).
// One pass: reads p, g, m, v (16 B/elem), writes p, m, v (12 B/elem) and, when asked, the bf16 compute shadow (2 B/elem) — so the
// low-precision weights the next forward reads never need a separate cast pass.  Arithmetic follows torch's single-tensor AdamW order.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    long n4, float decay, float b1, float b2, float eps, float step_size, float inv_bc2_sqrt,
                                                    float gscale, uint16_t* __restrict__ lp, float* __restrict__ ema, float ema_rate) {
  // pure stream: every byte is touched once per step -> non-temporal accesses, two float4 groups in flight per thread and stream
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const long stride = (long)gridDim.x * 256;
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 2 * stride) {
    f4 P[2], G[2], M[2], V[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long i = i0 + u * stride;
      if (i < n4) {
        P[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p) + i);
        G[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(g) + i);
        M[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(m) + i);
        V[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(v) + i);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long i = i0 + u * stride;
      if (i >= n4) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gr = __fmul_rn(G[u][j], gscale);
        float w = P[u][j], m_ = M[u][j], v_ = V[u][j];
        mode::adamw_update_f(w, m_, v_, gr, decay, b1, b2, eps, step_size, inv_bc2_sqrt);
        P[u][j] = w; M[u][j] = m_; V[u][j] = v_;
      }
      __builtin_nontemporal_store(P[u], reinterpret_cast<f4*>(p) + i);
      __builtin_nontemporal_store(M[u], reinterpret_cast<f4*>(m) + i);
      __builtin_nontemporal_store(V[u], reinterpret_cast<f4*>(v) + i);
      if (ema) {                                       // EMA of the weights (mode/callbacks/ema.py:119-126): e -= (1 - decay) * (e - w)
        f4 Ev = __builtin_nontemporal_load(reinterpret_cast<const f4*>(ema) + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) Ev[j] = Ev[j] - ema_rate * (Ev[j] - P[u][j]);
        __builtin_nontemporal_store(Ev, reinterpret_cast<f4*>(ema) + i);
      }
      if (lp) {
        u2 o; o[0] = pack_bf16x2(P[u][0], P[u][1]); o[1] = pack_bf16x2(P[u][2], P[u][3]);
        *(reinterpret_cast<u2*>(lp) + i) = o;           // the bf16 shadow is re-read by the next forward: normal (cached) store
      }
    }
  }
}

namespace mode { int g_adamw_blocks = 0; }   // "adamw_blocks" option: cap on the workgroups of one AdamW launch (0 = 256, one per CU: 15.4 vs 15.8 ms per step with 2048)

extern "C" int mode_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, float grad_scale, void* lp_bf16, float* ema, float ema_rate, void* stream) {
  if (n == 0) return MODE_OK;
  if (!p || !g || !m || !v || n < 0 || n % 4 || step < 1 || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) || ((uintptr_t)lp_bf16 & 7))
    return MODE_ERR_BAD_ARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const long n4 = n / 4;
  const int blocks = (int)std::min<long>((n4 + 511) / 512, g_adamw_blocks > 0 ? g_adamw_blocks : 256);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, 1.f - lr * weight_decay, beta1, beta2, eps,
                     (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, (uint16_t*)lp_bf16, ema, ema_rate);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_router_mlp_bwd(const float* dlog, const float* r_pre, const float* w3, int L, int B, int E, int H2, float* dpre, float* dw3,
                                   void* stream) {
  if (!dlog || !r_pre || !w3 || !dpre || !dw3 || L <= 0 || B <= 0 || E <= 0 || H2 <= 0) return MODE_ERR_BAD_ARG;
  if (H2 % 4 || E > 16) return MODE_ERR_UNSUPPORTED;
  const long total4 = (long)B * L * H2 / 4;
  hipLaunchKernelGGL(router_mlp_dpre_kernel, dim3((total4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, dlog, r_pre, w3, L, B, E, H2, dpre);
  MODE_LAUNCH_CHECK();
  const dim3 grid((H2 + 255) / 256, L);
  if (E <= 4) hipLaunchKernelGGL(router_w3_grad_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dlog, r_pre, L, B, E, H2, dw3);
  else if (E <= 8) hipLaunchKernelGGL(router_w3_grad_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, dlog, r_pre, L, B, E, H2, dw3);
  else hipLaunchKernelGGL(router_w3_grad_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, dlog, r_pre, L, B, E, H2, dw3);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_iota_i32(int32_t* out, int n, int step, void* stream) {
  if (!out || n < 0) return MODE_ERR_BAD_ARG;
  if (n == 0) return MODE_OK;
  hipLaunchKernelGGL(iota_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, n, step);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

namespace mode {
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long n4, float rate) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 e = reinterpret_cast<float4*>(ema)[i];
    const float4 w = reinterpret_cast<const float4*>(p)[i];
    e.x -= rate * (e.x - w.x); e.y -= rate * (e.y - w.y); e.z -= rate * (e.z - w.z); e.w -= rate * (e.w - w.w);
    reinterpret_cast<float4*>(ema)[i] = e;
  }
}
}  // namespace mode

extern "C" int mode_ema_update(float* ema, const float* p, int64_t n, float rate, void* stream) {
  if (n == 0) return MODE_OK;
  if (!ema || !p || n < 0 || n % 4 || (((uintptr_t)ema | (uintptr_t)p) & 15)) return MODE_ERR_BAD_ARG;
  const long n4 = n / 4;
  hipLaunchKernelGGL(mode::ema_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, ema, p, n4, rate);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" size_t mode_swiglu_bwd_bias_workspace_bytes(int64_t rows, int Hdim, int E) {
  if (rows < 0 || Hdim <= 0 || E <= 0) return 0;
  const long nrb = (rows + 31) / 32;
  return (size_t)nrb * E * 2 * Hdim * 4 + 256;
}

extern "C" int mode_swiglu_bwd_bias(const void* P, const void* dHd, void* dP, int64_t rows, int Hdim, int dtype, uint32_t seed, float p_drop,
                                    const int32_t* expert_offsets, int E, float* db, void* workspace, size_t workspace_bytes, void* stream) {
  if (!P || !dHd || !dP || !expert_offsets || !db || !workspace || rows < 0 || Hdim <= 0 || E <= 0 || p_drop < 0.f || p_drop >= 1.f) return MODE_ERR_BAD_ARG;
  if (dtype != MODE_BF16 || Hdim % 8 || E > 16 || (((uintptr_t)P | (uintptr_t)dHd | (uintptr_t)dP) & 15)) return MODE_ERR_UNSUPPORTED;
  if (workspace_bytes < mode_swiglu_bwd_bias_workspace_bytes(rows, Hdim, E)) return MODE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (rows == 0) return hipMemsetAsync(db, 0, (size_t)E * 2 * Hdim * 4, s) == hipSuccess ? MODE_OK : (int)hipGetLastError();
  const int RB = 32;
  const int nrb = (int)((rows + RB - 1) / RB);
  float* partial = (float*)workspace;
  const dim3 grid((Hdim / 8 + 255) / 256, nrb);
  const float ik = 1.0f / (1.0f - p_drop);
  if (E <= 4)
    hipLaunchKernelGGL(swiglu_bwd_bias_bf16x8_kernel<4>, grid, dim3(256), 0, s, (const uint16_t*)P, (const uint16_t*)dHd, (uint16_t*)dP, (long)rows, Hdim, seed,
                       drop_thresh(p_drop), ik, expert_offsets, E, RB, partial);
  else
    hipLaunchKernelGGL(swiglu_bwd_bias_bf16x8_kernel<16>, grid, dim3(256), 0, s, (const uint16_t*)P, (const uint16_t*)dHd, (uint16_t*)dP, (long)rows, Hdim, seed,
                       drop_thresh(p_drop), ik, expert_offsets, E, RB, partial);
  MODE_LAUNCH_CHECK();
  // db[e][c] = sum over row blocks of partial[rb][e][c]: rows = nrb, cols = E * 2 * Hdim
  return mode_colsum(partial, (int64_t)E * 2 * Hdim, nrb, E * 2 * Hdim, MODE_F32, nullptr, 0, 1, db, 0, (char*)workspace + (size_t)nrb * E * 2 * Hdim * 4, 0, stream);
}

extern "C" int mode_pos_emb_bwd(const float* dx0, int B, int T, int D, int t0, int n_img, int A_len, float* dpos, void* stream) {
  if (!dx0 || !dpos || B < 0 || T <= 0 || D <= 0 || t0 < 0 || n_img < 0 || A_len <= 0 || t0 + 1 + n_img + A_len > T) return MODE_ERR_BAD_ARG;
  hipLaunchKernelGGL(pos_emb_bwd_kernel, dim3((D + 255) / 256, 1 + A_len), dim3(256), 0, (hipStream_t)stream, dx0, B, T, D, t0, n_img, A_len, dpos);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}
