// The encoders' entry: the small-channel k x k convolution (ResNet stem: 3 -> 64 channels, 7 x 7, stride 2 - reference
// mode/models/perceptual_encoders/resnets.py:96 `models.resnet18` / pretrained_resnets.py:29 timm `resnet50`, both conv1) and the 3 x 3 / stride-2
// max-pool behind it, on gfx950.
//
// The implicit-GEMM convolutions of conv_gemm.hip step K in 64-channel pieces of ONE filter tap; with 3 input channels a tap is 3 elements, so the
// stem went to MIOpen (igemm_fwd 122 us + 24 us of layout / cast copies, igemm_wrw 103 us + helpers per tower at B = 64).  Here a workgroup walks tiles of
// 8 x 16 output pixels:
//   patch   the tile's input window [Cin][(8-1) sh + kh][(16-1) sw + kw] (21 x 37 x 3 for the stem) is fetched ONE TILE AHEAD into registers straight from the
//           caller's image tensor - any strides, fp32 or bf16: the fp32 -> bf16 rounding and the NCHW -> channels_last change that torch did in two copy
//           kernels happen here - with unconditional loads (out-of-image elements are masked when the patch is written to LDS)
//   im2col  P[pixel][k = (a, b, c)] (K = kh * kw * Cin = 147, padded with zero columns to a multiple of 32) is built from the patch inside LDS, eight
//           pixel rows per batch (all reads before the first write); it never exists in memory
//   forward          Y[pixel][cout]  = P[pixel][k] W[cout][k]^T        K-contiguous fragments (ds_read_b128), swapped operands, output tile through LDS
//   weight gradient  dW[cout][k]     = dY[pixel][cout]^T P[pixel][k]   reduction over pixels: both fragments by the LDS transpose read (ds_read_b64_tr_b16)
// HBM traffic is the image (once per tile, the overlap of neighbouring windows from L2) and Y / dY; the weight gradient keeps its [cout][k] accumulators
// in registers over all the tiles of a workgroup and writes one partial slab per workgroup (summed by the caller in slab order: deterministic).
// No data gradient: the stem's input is the camera image.  Measured (scripts/stem_probe.py, B = 64): forward 104 us, of which the LDS im2col build 38
// (2-byte LDS gathers at ~4 clk per wave instruction), the products 14, the prefetch 17, the stores 11 - profiles/r05_stem.txt.
//
// Max-pool (k x k / stride s / padding p, channels_last): the forward keeps the window position of each maximum (first maximum in scan order, NaN
// wins - aten's rule), the backward GATHERS: an input pixel sums the dy of the (at most ceil(k/s)^2) windows whose maximum it was - no atomics.
#include "mode_common.h"

namespace mode {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct StemParams {
  const void* x; int x_f32; long sxn, sxc, sxh, sxw;     // image [N, Cin, H, W] by element strides
  int N, H, W, Cin, kh, kw, sh, sw, ph, pw, ho, wo, Cout, K;
  int tiles_h, tiles_w, tiles;                           // 8 x 16-pixel output tiles per image (rows, columns), tiles in all
  int PH, PW, PE;                                        // input patch of a tile: rows, columns, elements (Cin * PH * PW)
  const uint16_t* w;                                     // [Cout][K] bf16, k = (a * kw + b) * Cin + c (the channels_last storage of [Cout, Cin, kh, kw])
  uint16_t* y;                                           // [N * ho * wo][Cout] bf16
  const float* bn_mean; const float* bn_var; const float* bn_w; const float* bn_b; float bn_eps; int relu;   // optional epilogue: eval-mode BatchNorm (folded as in conv_gemm.hip) + ReLU
  const uint16_t* dy; float* part;                       // weight gradient: dY [N * ho * wo][Cout] bf16, partial slabs [gridDim.x][Cout][K] fp32
};

constexpr int ST_TH = 8, ST_TW = 16;                     // output pixels of a tile: 8 rows x 16 columns
constexpr int ST_BM = ST_TH * ST_TW, ST_BN = 64, ST_CP = 72;   // pixels per tile, output channels, pitch (elements) of the [pixel][channel] LDS tiles
constexpr int ST_MAXR = 32;                              // patch elements per thread (registers of the prefetch): PE <= 256 * ST_MAXR; kernels come in MR = 12 | 32

__device__ __forceinline__ void stem_tile_origin(const StemParams& p, int tile, int& n, int& oh0, int& ow0) {
  const int per = p.tiles_h * p.tiles_w;
  n = tile / per;
  const int r = tile - n * per, th = r / p.tiles_w;
  oh0 = th * ST_TH; ow0 = (r - th * p.tiles_w) * ST_TW;
}

// The tile's input patch, planar [c][PH][PW] (zero outside the image), into registers: thread t owns elements t, t + 256, ... - consecutive lanes read
// consecutive image columns.  Which image element a thread's i-th patch element is does not depend on the tile: offset (relative to the patch origin) and
// (row, col) are computed once (stem_patch_map), a tile adds its origin.  Issued one tile ahead; the loads land under the previous tile's products.
template <int MR>
__device__ __forceinline__ void stem_patch_map(const StemParams& p, int tid, int (&rel)[MR], uint32_t (&rc)[MR]) {
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const int idx = i * 256 + tid;
    rel[i] = 0; rc[i] = 0xffffffffu;                       // row 65535: never inside an image
    if (idx < p.PE) {
      const int t = idx / p.PW, col = idx - t * p.PW, c = t / p.PH, row = t - c * p.PH;
      rel[i] = (int)((long)c * p.sxc + (long)row * p.sxh + (long)col * p.sxw);
      rc[i] = ((uint32_t)row << 16) | (uint32_t)col;
    }
  }
}
// The loads are unconditional (outside the image: element 0 of the tensor, masked when the patch is written) and their values stay raw until then - a
// bounds branch or the fp32 -> bf16 rounding next to the load would make every load wait for itself, and the prefetch would hide nothing.
template <int MR>
__device__ __forceinline__ uint32_t stem_load_patch(const StemParams& p, int tile, const int (&rel)[MR], const uint32_t (&rc)[MR], uint32_t (&pr)[MR]) {
  int n, oh0, ow0;
  stem_tile_origin(p, tile, n, oh0, ow0);
  const int ih0 = oh0 * p.sh - p.ph, iw0 = ow0 * p.sw - p.pw;
  const long base = (long)n * p.sxn + (long)ih0 * p.sxh + (long)iw0 * p.sxw;
  uint32_t ok = 0;
  if (p.x_f32) {
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int ih = ih0 + (int)(rc[i] >> 16), iw = iw0 + (int)(rc[i] & 0xffffu);
      const bool in = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      ok |= (uint32_t)in << i;
      pr[i] = __float_as_uint(((const float*)p.x)[in ? base + rel[i] : 0l]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int ih = ih0 + (int)(rc[i] >> 16), iw = iw0 + (int)(rc[i] & 0xffffu);
      const bool in = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      ok |= (uint32_t)in << i;
      pr[i] = ((const uint16_t*)p.x)[in ? base + rel[i] : 0l];
    }
  }
  return ok;
}
template <int MR>
__device__ __forceinline__ void stem_store_patch(const StemParams& p, int tid, const uint32_t (&pr)[MR], uint32_t ok, uint16_t* patch) {
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const int idx = i * 256 + tid;
    const uint16_t v = p.x_f32 ? f32_to_bf16_bits(__uint_as_float(pr[i])) : (uint16_t)pr[i];
    if (idx < p.PE) patch[idx] = ((ok >> i) & 1u) ? v : (uint16_t)0;
  }
}

// lut[k] = patch offset of im2col column k = (a, b, c) relative to the pixel's window origin; -1 = zero padding column
template <int KP32>
__device__ __forceinline__ void stem_fill_lut(const StemParams& p, int* lut, int tid) {
  constexpr int KP = KP32 * 32;
  for (int k = tid; k < KP; k += 256) {
    int v = -1;
    if (k < p.K) { const int c = k % p.Cin, t = k / p.Cin, a = t / p.kw, b = t - a * p.kw; v = (c * p.PH + a) * p.PW + b; }
    lut[k] = v;
  }
}

// rows `wave, wave + 4, ...` of the tile's im2col image P[pixel][k] from the patch; lane -> columns lane, lane + 64, ...  Eight rows per batch: all reads
// of a batch are issued before its first write (patch and P are both LDS: left to the compiler, every write would wait for its own read).  Padding columns
// read the zero element kept at patch[PE].
template <int KP32>
__device__ __forceinline__ void stem_build_tile(const StemParams& p, const uint16_t* __restrict__ patch, uint16_t* __restrict__ P, const int* lut, int wave, int lane) {
  constexpr int KP = KP32 * 32, PITCH = KP + 8, KJ = (KP + 63) / 64;
  int e[KJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j) { const int k = lane + 64 * j; e[j] = k < KP ? lut[k] : -1; }
  const bool last_live = lane + 64 * (KJ - 1) < KP;        // (KP % 64 == 32: the upper half of the lanes has no column in the last group)
#pragma unroll 1
  for (int r0 = wave; r0 < ST_BM; r0 += 32) {
    uint16_t v[8][KJ];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int rl = r0 + 4 * r;
      const int org = (rl >> 4) * p.sh * p.PW + (rl & 15) * p.sw;
#pragma unroll
      for (int j = 0; j < KJ; ++j) v[r][j] = patch[e[j] >= 0 ? org + e[j] : p.PE];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int rl = r0 + 4 * r;
#pragma unroll
      for (int j = 0; j < KJ; ++j)
        if (j + 1 < KJ || last_live) P[rl * PITCH + lane + 64 * j] = v[r][j];
    }
  }
}

// global row of tile pixel `px` (-1: outside the image)
__device__ __forceinline__ long stem_pixel_row(const StemParams& p, int n, int oh0, int ow0, int px) {
  const int oh = oh0 + (px >> 4), ow = ow0 + (px & 15);
  return (oh < p.ho && ow < p.wo) ? ((long)n * p.ho + oh) * p.wo + ow : -1;
}

template <int KP32, int MR>
__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(const StemParams p) {
  constexpr int KP = KP32 * 32, PITCH = KP + 8, PROWS = PITCH > ST_CP ? PITCH : ST_CP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* lut = (int*)smem;
  float* bnp = (float*)(smem + KP * 4);                  // [2][64] folded eval-mode BatchNorm: scale | shift
  uint16_t* Wt = (uint16_t*)(smem + KP * 4 + 2 * ST_BN * 4);   // [64][PITCH]
  uint16_t* P = Wt + ST_BN * PITCH;                      // [128][PITCH]; reused as the output tile [128][ST_CP]
  uint16_t* patch = P + ST_BM * PROWS;                   // [Cin][PH][PW]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4;
  int rel[MR]; uint32_t rc[MR], pr[MR];
  stem_patch_map<MR>(p, tid, rel, rc);
  if (tid == 0) patch[p.PE] = 0;                         // the source of the im2col padding columns
  int tile = blockIdx.x;
  uint32_t ok = 0;
  if (tile < p.tiles) ok = stem_load_patch<MR>(p, tile, rel, rc, pr);
  stem_fill_lut<KP32>(p, lut, tid);
  // weight tile [64][PITCH]: zero (padding columns / rows past Cout), then the [Cout][K] matrix as it lies - unconditional loads in batches of eight
  for (int c = tid; c < ST_BN * PITCH / 8; c += 256) *(uint4*)&Wt[c * 8] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  {
    const int total = p.Cout * p.K;
    for (int i0 = tid; i0 < total; i0 += 8 * 256) {
      uint16_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p.w[min(i0 + u * 256, total - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + u * 256, n = idx / p.K;
        if (idx < total) Wt[n * PITCH + idx - n * p.K] = v[u];
      }
    }
  }
  if (tid < ST_BN) {                                      // the expressions of conv_gemm.hip's fused epilogue / bn_prepare_kernel
    float sc = 1.f, sf = 0.f;
    if (p.bn_mean && tid < p.Cout) {
      sc = 1.0f / sqrtf(p.bn_var[tid] + p.bn_eps);
      if (p.bn_w) sc *= p.bn_w[tid];
      if (p.bn_b) sf = p.bn_b[tid];
      sf -= p.bn_mean[tid] * sc;
    }
    bnp[tid] = sc; bnp[ST_BN + tid] = sf;
  }
  for (; tile < p.tiles; tile += gridDim.x) {
    int n, oh0, ow0;
    stem_tile_origin(p, tile, n, oh0, ow0);
    stem_store_patch<MR>(p, tid, pr, ok, patch);
    __syncthreads();                                      // (first pass: also the lut / weight tile / BatchNorm fold)
    if (tile + (int)gridDim.x < p.tiles) ok = stem_load_patch<MR>(p, tile + gridDim.x, rel, rc, pr);
    stem_build_tile<KP32>(p, patch, P, lut, wave, lane);
    __syncthreads();
    f32x4 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = acc[j][0]; }
#pragma unroll
    for (int ks = 0; ks < KP32; ++ks) {
      bf16x8 fb[2], fa[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) fb[i] = *(const bf16x8*)&P[(32 * wave + 16 * i + fi) * PITCH + 32 * ks + 8 * fg];
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[j] = *(const bf16x8*)&Wt[(16 * j + fi) * PITCH + 32 * ks + 8 * fg];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j], fb[i], acc[j][i], 0, 0, 0);   // D[cout 16 j + 4 fg + t][pixel 16 i + fi]
    }
    __syncthreads();                                      // every wave is done with P: it becomes the output tile
    uint16_t* Ct = P;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = 16 * j + 4 * fg;
      const float4 sc4 = *(const float4*)&bnp[co], sf4 = *(const float4*)&bnp[ST_BN + co];
      const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sf[4] = {sf4.x, sf4.y, sf4.z, sf4.w};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          v[t] = p.bn_mean ? __builtin_fmaf(acc[j][i][t], sc[t], sf[t]) : acc[j][i][t];
          if (p.relu) v[t] = fmaxf(v[t], 0.f);
        }
        uint2 o; o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)&Ct[(32 * wave + 16 * i + fi) * ST_CP + co] = o;
      }
    }
    __syncthreads();
    const int cpr = p.Cout >> 3;                          // 16-byte chunks per output row
    for (int c = tid; c < ST_BM * 8; c += 256) {
      const int px = c >> 3, ch = c & 7;
      const long r = stem_pixel_row(p, n, oh0, ow0, px);
      if (r >= 0 && ch < cpr) *(uint4*)&p.y[r * p.Cout + ch * 8] = *(const uint4*)&Ct[px * ST_CP + ch * 8];
    }
    __syncthreads();
  }
}

__device__ __forceinline__ bf16x8 tr8(const uint16_t* lo, int pitch) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo + 4 * pitch));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int KP32, int MR>
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const StemParams p) {
  constexpr int KP = KP32 * 32, PITCH = KP + 8, NT = KP / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* lut = (int*)smem;
  uint16_t* Dy = (uint16_t*)(smem + KP * 4);             // [128 pixels][ST_CP]
  uint16_t* P = Dy + ST_BM * ST_CP;                      // [128 pixels][PITCH]
  uint16_t* patch = P + ST_BM * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4;
  int rel[MR]; uint32_t rc[MR], pr[MR];
  stem_patch_map<MR>(p, tid, rel, rc);
  if (tid == 0) patch[p.PE] = 0;                         // the source of the im2col padding columns
  int tile = blockIdx.x;
  uint32_t ok = 0;
  if (tile < p.tiles) ok = stem_load_patch<MR>(p, tile, rel, rc, pr);
  stem_fill_lut<KP32>(p, lut, tid);
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int cpr = p.Cout >> 3;
  for (; tile < p.tiles; tile += gridDim.x) {
    int n, oh0, ow0;
    stem_tile_origin(p, tile, n, oh0, ow0);
    stem_store_patch<MR>(p, tid, pr, ok, patch);
    for (int c = tid; c < ST_BM * 8; c += 256) {
      const int px = c >> 3, ch = c & 7;
      const long r = stem_pixel_row(p, n, oh0, ow0, px);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r >= 0 && ch < cpr) v = *(const uint4*)&p.dy[r * p.Cout + ch * 8];
      *(uint4*)&Dy[px * ST_CP + ch * 8] = v;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < p.tiles) ok = stem_load_patch<MR>(p, tile + gridDim.x, rel, rc, pr);
    stem_build_tile<KP32>(p, patch, P, lut, wave, lane);
    __syncthreads();
    // this wave: output channels 16 wave .. + 15 (rows of dW), all NT column tiles; the transpose read hands lane (fi, fg) the 8 pixels 8 fg .. 8 fg + 7 of column fi
#pragma unroll
    for (int k0 = 0; k0 < ST_BM; k0 += 32) {
      const int row = k0 + 8 * fg + (fi >> 2), col = 4 * (fi & 3);
      const bf16x8 a = tr8(&Dy[row * ST_CP + 16 * wave + col], ST_CP);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 b = tr8(&P[row * PITCH + 16 * t + col], PITCH);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);        // D[cout 16 wave + 4 fg + j][k 16 t + fi]
      }
    }
    __syncthreads();
  }
  float* out = p.part + (long)blockIdx.x * p.Cout * p.K;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = 16 * t + fi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = 16 * wave + 4 * fg + j;
      if (k < p.K && co < p.Cout) out[(long)co * p.K + k] = acc[t][j];
    }
  }
}

// forward: the output tile [128][ST_CP] reuses the im2col region; the patch comes last
static int stem_lds_bytes(int KP, int PE, bool wgrad) {
  const int pitch = KP + 8, patch = (((PE + 1) * 2 + 15) / 16) * 16;
  return KP * 4 + (wgrad ? ST_BM * ST_CP * 2 + ST_BM * pitch * 2 : 2 * ST_BN * 4 + ST_BN * pitch * 2 + ST_BM * (pitch > ST_CP ? pitch : ST_CP) * 2) + patch;
}

template <int KP32, int MR>
static int stem_launch_mr(const StemParams& p, bool wgrad, int grid, hipStream_t s) {
  const int lds = stem_lds_bytes(KP32 * 32, p.PE, wgrad);
  if (wgrad) {
    static LdsLimitOnce once;
    if (int rc = once.ensure((const void*)stem_wgrad_kernel<KP32, MR>, lds)) return rc;
    hipLaunchKernelGGL((stem_wgrad_kernel<KP32, MR>), dim3(grid), dim3(256), lds, s, p);
  } else {
    static LdsLimitOnce once;
    if (int rc = once.ensure((const void*)stem_fwd_kernel<KP32, MR>, lds)) return rc;
    hipLaunchKernelGGL((stem_fwd_kernel<KP32, MR>), dim3(grid), dim3(256), lds, s, p);
  }
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}
template <int KP32>
static int stem_launch(const StemParams& p, bool wgrad, int grid, hipStream_t s) {
  return p.PE <= 256 * 12 ? stem_launch_mr<KP32, 12>(p, wgrad, grid, s) : stem_launch_mr<KP32, ST_MAXR>(p, wgrad, grid, s);
}

static int stem_dispatch(const StemParams& p, bool wgrad, int grid, hipStream_t s) {
  switch ((p.K + 31) / 32) {
    case 1: return stem_launch<1>(p, wgrad, grid, s);
    case 2: return stem_launch<2>(p, wgrad, grid, s);
    case 3: return stem_launch<3>(p, wgrad, grid, s);
    case 4: return stem_launch<4>(p, wgrad, grid, s);
    case 5: return stem_launch<5>(p, wgrad, grid, s);
    case 6: return stem_launch<6>(p, wgrad, grid, s);
    case 7: return stem_launch<7>(p, wgrad, grid, s);
    case 8: return stem_launch<8>(p, wgrad, grid, s);
    default: return MODE_ERR_UNSUPPORTED;
  }
}

static int stem_params(const ModeStemConvDesc* d, StemParams& p) {
  if (!d || (!d->x && d->N > 0) || !d->w || d->N < 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->kh <= 0 || d->kw <= 0 || d->sh <= 0 || d->sw <= 0 || d->ph < 0 || d->pw < 0)
    return MODE_ERR_BAD_ARG;
  if (d->x_dtype != MODE_F32 && d->x_dtype != MODE_BF16) return MODE_ERR_BAD_ARG;
  if (d->Cout <= 0 || d->Cout > ST_BN || d->Cout % 16 != 0 || d->kh > 255 || d->kw > 255) return MODE_ERR_UNSUPPORTED;
  const long K = (long)d->kh * d->kw * d->Cin;
  if (K > 256) return MODE_ERR_UNSUPPORTED;
  const int ho = (d->H + 2 * d->ph - d->kh) / d->sh + 1, wo = (d->W + 2 * d->pw - d->kw) / d->sw + 1;
  if (d->H + 2 * d->ph < d->kh || d->W + 2 * d->pw < d->kw) return MODE_ERR_BAD_ARG;
  const long R = (long)d->N * ho * wo;
  if (R >= (1l << 31)) return MODE_ERR_UNSUPPORTED;
  const long PH = (long)(ST_TH - 1) * d->sh + d->kh, PW = (long)(ST_TW - 1) * d->sw + d->kw;
  if (PH * PW * d->Cin > 256l * ST_MAXR || PH > 65535 || PW > 65535) return MODE_ERR_UNSUPPORTED;   // the tile's input patch is staged in registers + LDS
  const auto mag = [](int64_t v) { return v < 0 ? -v : v; };
  if ((double)(d->Cin - 1) * mag(d->sxc) + (double)(PH + d->H) * mag(d->sxh) + (double)(PW + d->W) * mag(d->sxw) >= 2147483647.0) return MODE_ERR_UNSUPPORTED;   // 32-bit offsets inside one image
  p = StemParams{};
  p.x = d->x; p.x_f32 = d->x_dtype == MODE_F32; p.sxn = d->sxn; p.sxc = d->sxc; p.sxh = d->sxh; p.sxw = d->sxw;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.kh = d->kh; p.kw = d->kw; p.sh = d->sh; p.sw = d->sw; p.ph = d->ph; p.pw = d->pw;
  p.ho = ho; p.wo = wo; p.Cout = d->Cout; p.K = (int)K;
  p.tiles_h = (ho + ST_TH - 1) / ST_TH; p.tiles_w = (wo + ST_TW - 1) / ST_TW;
  const long tiles = (long)d->N * p.tiles_h * p.tiles_w;
  if (tiles >= (1l << 31)) return MODE_ERR_UNSUPPORTED;
  p.tiles = (int)tiles; p.PH = (int)PH; p.PW = (int)PW; p.PE = (int)(PH * PW * d->Cin);
  p.w = (const uint16_t*)d->w;
  return MODE_OK;
}

// ---- max-pool, channels_last ------------------------------------------------------------------------------------------------------------
struct PoolParams { const void* x; void* y; uint8_t* arg; int N, H, W, C, k, s, p, ho, wo; };

template <typename T> struct Vec8;
template <> struct Vec8<uint16_t> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void load(const void* p, long off, float* f) {
    const uint4 u = *(const uint4*)((const uint16_t*)p + off);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(void* p, long off, const float* f) {
    uint4 u; u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]); u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
    *(uint4*)((uint16_t*)p + off) = u;
  }
};
template <> struct Vec8<float> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void load(const void* p, long off, float* f) {
    const float4 a = *(const float4*)((const float*)p + off), b = *(const float4*)((const float*)p + off + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
  static __device__ __forceinline__ void store(void* p, long off, const float* f) {
    *(float4*)((float*)p + off) = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)((float*)p + off + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
};

// one thread = 8 channels of one output pixel
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_nhwc_kernel(const PoolParams p) {
  const int cv = p.C >> 3;
  const long total = (long)p.N * p.ho * p.wo * cv;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8 = (int)(idx % cv);
  long pix = idx / cv;
  const int ow = (int)(pix % p.wo); pix /= p.wo;
  const int oh = (int)(pix % p.ho);
  const int n = (int)(pix / p.ho);
  float best[8]; int bi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bi[i] = -1; }
  for (int a = 0; a < p.k; ++a) {
    const int ih = oh * p.s - p.p + a;
    if ((unsigned)ih >= (unsigned)p.H) continue;
    for (int b = 0; b < p.k; ++b) {
      const int iw = ow * p.s - p.p + b;
      if ((unsigned)iw >= (unsigned)p.W) continue;
      float v[8];
      Vec8<T>::load(p.x, (((long)n * p.H + ih) * p.W + iw) * p.C + c8 * 8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (v[i] > best[i] || v[i] != v[i] || bi[i] < 0) { best[i] = v[i]; bi[i] = a * p.k + b; }       // aten: (val > max) || isnan(val)
    }
  }
  const long o = (((long)n * p.ho + oh) * p.wo + ow) * p.C + c8 * 8;
  Vec8<T>::store(p.y, o, best);
  if (p.arg) {
    uint2 u;
    u.x = (uint32_t)(bi[0] & 255) | ((uint32_t)(bi[1] & 255) << 8) | ((uint32_t)(bi[2] & 255) << 16) | ((uint32_t)(bi[3] & 255) << 24);
    u.y = (uint32_t)(bi[4] & 255) | ((uint32_t)(bi[5] & 255) << 8) | ((uint32_t)(bi[6] & 255) << 16) | ((uint32_t)(bi[7] & 255) << 24);
    *(uint2*)(p.arg + o) = u;
  }
}

// one thread = 8 channels of one INPUT pixel: dx = sum of dy over the windows whose recorded maximum is this pixel (windows in ascending (oh, ow) order)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_nhwc_kernel(const PoolParams p) {       // x = dy, y = dx
  const int cv = p.C >> 3;
  const long total = (long)p.N * p.H * p.W * cv;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8 = (int)(idx % cv);
  long pix = idx / cv;
  const int iw = (int)(pix % p.W); pix /= p.W;
  const int ih = (int)(pix % p.H);
  const int n = (int)(pix / p.H);
  float g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) g[i] = 0.f;
  // windows oh with oh * s - pad <= ih <= oh * s - pad + k - 1
  const int oh_lo = max(0, (ih + p.p - p.k + p.s) / p.s), oh_hi = min(p.ho - 1, (ih + p.p) / p.s);
  const int ow_lo = max(0, (iw + p.p - p.k + p.s) / p.s), ow_hi = min(p.wo - 1, (iw + p.p) / p.s);
  for (int oh = oh_lo; oh <= oh_hi; ++oh) {
    const int a = ih - (oh * p.s - p.p);
    if (a < 0 || a >= p.k) continue;
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      const int b = iw - (ow * p.s - p.p);
      if (b < 0 || b >= p.k) continue;
      const long o = (((long)n * p.ho + oh) * p.wo + ow) * p.C + c8 * 8;
      const uint2 u = *(const uint2*)(p.arg + o);
      float v[8];
      Vec8<T>::load(p.x, o, v);
      const uint32_t want = (uint32_t)(a * p.k + b);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t w = ((i < 4 ? u.x : u.y) >> (8 * (i & 3))) & 255u;
        if (w == want) g[i] += v[i];
      }
    }
  }
  Vec8<T>::store(p.y, (((long)n * p.H + ih) * p.W + iw) * p.C + c8 * 8, g);
}

}  // namespace mode

using namespace mode;

extern "C" int mode_stem_conv_fwd(const ModeStemConvDesc* d, void* stream) {
  StemParams p;
  if (int rc = stem_params(d, p)) return rc;
  if ((d->bn_mean == nullptr) != (d->bn_var == nullptr)) return MODE_ERR_BAD_ARG;
  if (p.tiles == 0) return MODE_OK;
  if (!d->y) return MODE_ERR_BAD_ARG;
  if (reinterpret_cast<uintptr_t>(d->y) & 15) return MODE_ERR_UNSUPPORTED;                 // 16-byte row chunks
  p.y = (uint16_t*)d->y; p.bn_mean = d->bn_mean; p.bn_var = d->bn_var; p.bn_w = d->bn_weight; p.bn_b = d->bn_bias; p.bn_eps = d->bn_eps; p.relu = d->relu;
  return stem_dispatch(p, false, p.tiles < 512 ? p.tiles : 512, (hipStream_t)stream);      // one resident round (2 workgroups per CU)
}

extern "C" int mode_stem_conv_wgrad_slabs(const ModeStemConvDesc* d) {
  StemParams p;
  if (stem_params(d, p)) return 0;
  return p.tiles < 512 ? (p.tiles > 0 ? p.tiles : 1) : 512;
}

extern "C" int mode_stem_conv_wgrad(const ModeStemConvDesc* d, void* stream) {
  StemParams p;
  if (int rc = stem_params(d, p)) return rc;
  if ((!d->dy && p.tiles > 0) || !d->dw_part) return MODE_ERR_BAD_ARG;
  if (reinterpret_cast<uintptr_t>(d->dy) & 15) return MODE_ERR_UNSUPPORTED;
  p.dy = (const uint16_t*)d->dy; p.part = d->dw_part;
  return stem_dispatch(p, true, mode_stem_conv_wgrad_slabs(d), (hipStream_t)stream);      // no pixels: one workgroup writes a zero slab
}

static int pool_params(int dtype, int N, int H, int W, int C, int k, int s, int pad, PoolParams& p) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || k > 15 || s <= 0 || pad < 0 || 2 * pad > k) return MODE_ERR_BAD_ARG;
  if (dtype != MODE_F32 && dtype != MODE_BF16) return MODE_ERR_BAD_ARG;
  if (C % 8 != 0) return MODE_ERR_UNSUPPORTED;
  if (H + 2 * pad < k || W + 2 * pad < k) return MODE_ERR_BAD_ARG;
  p.N = N; p.H = H; p.W = W; p.C = C; p.k = k; p.s = s; p.p = pad;
  p.ho = (H + 2 * pad - k) / s + 1; p.wo = (W + 2 * pad - k) / s + 1;
  return MODE_OK;
}

extern "C" int mode_maxpool_nhwc_fwd(const void* x, int dtype, int N, int H, int W, int C, int k, int s, int pad, void* y, uint8_t* argmax, void* stream) {
  PoolParams p{};
  if (int rc = pool_params(dtype, N, H, W, C, k, s, pad, p)) return rc;
  if (!x || !y) return MODE_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15 || (reinterpret_cast<uintptr_t>(argmax) & 7)) return MODE_ERR_UNSUPPORTED;
  p.x = x; p.y = y; p.arg = argmax;
  const long total = (long)N * p.ho * p.wo * (C >> 3);
  if (total == 0) return MODE_OK;
  const dim3 g((unsigned)((total + 255) / 256));
  if (dtype == MODE_BF16) hipLaunchKernelGGL(maxpool_fwd_nhwc_kernel<uint16_t>, g, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(maxpool_fwd_nhwc_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}

extern "C" int mode_maxpool_nhwc_bwd(const void* dy, const uint8_t* argmax, int dtype, int N, int H, int W, int C, int k, int s, int pad, void* dx, void* stream) {
  PoolParams p{};
  if (int rc = pool_params(dtype, N, H, W, C, k, s, pad, p)) return rc;
  if (!dy || !argmax || !dx) return MODE_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15 || (reinterpret_cast<uintptr_t>(argmax) & 7)) return MODE_ERR_UNSUPPORTED;
  p.x = dy; p.y = dx; p.arg = const_cast<uint8_t*>(argmax);
  const long total = (long)N * H * W * (C >> 3);
  if (total == 0) return MODE_OK;
  const dim3 g((unsigned)((total + 255) / 256));
  if (dtype == MODE_BF16) hipLaunchKernelGGL(maxpool_bwd_nhwc_kernel<uint16_t>, g, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(maxpool_bwd_nhwc_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, p);
  MODE_LAUNCH_CHECK();
  return MODE_OK;
}
