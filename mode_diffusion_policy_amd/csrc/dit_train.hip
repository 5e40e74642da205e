// Training launch chains of the MoDE denoiser: forward with an activation stash (per-token routing indices
// from the host-side multinomial, attention / expert dropout by counter-based hash masks) and the full backward — every gradient of
// MoDeDiT's parameters — as one C++ launch chain of HIP kernels: no host sync, no floating-point atomics (deterministic).
//
// Backward of a Linear y = x W^T reuses the MFMA GEMM twice:  dx = dy @ (W^T)^T  with a pre-transposed weight shadow, and
// dW = dy^T x as GEMM(A = dy^T, "W" = x^T) on operands produced by mode_transpose (per-expert 64-padded for the grouped GEMMs,
// contracted with the K-group mode so one launch covers all experts; experts without tokens get exact zeros).
#include "mode_common.h"

#include <string.h>

using namespace mode;

namespace mode {
int rmsnorm_bwd_launch(const float* x, const float* g, const float* dy_a, const float* dy_b, const float* G, int g_splits, long g_split_stride,
                       const int32_t* pos, int k, int rows, int D, float eps, float* dx, int accumulate, float* dg_partial, float* dy_out, void* dx_lp,
                       int lp_dtype, hipStream_t stream);   // train_ops.hip
int gemm_bf16_tr_swiglu_bwd_launch(const ModeGemmDesc* d, const void* P, void* dP, uint32_t seed, uint32_t thresh, float inv_keep, float* bsum, int* tile_offs,
                                   hipStream_t s);                                       // gemm_bf16_tr.hip
extern int g_fuse_swiglu_bwd;                                                            // "fuse_swiglu_bwd" option (dit.hip)
int attn_block_bwd_launch(const void* qkv, const float* q_gain, const float* k_gain, const void* dy, void* dqkv, float* dgq_partial, float* dgk_partial, int dtype, int B,
                          int T, int H, int head_dim, float eps, uint32_t seed, float p_drop, float* dbias_partial, void* stream);   // attn.hip
bool gemm_bf16_pptr_accepts(const ModeGemmDesc* d);                                      // gemm_bf16_pptr.hip: would mode_gemm take the ping-pong kernel?
int gather_rows_bf16(const void* in, long ld_in, const int* rows, int n, int cols, void* out, long ld_out, hipStream_t s);   // gemm_bf16_pptr.hip
int down_proj_split(int dt, int K);                                                      // dit.hip: K-slices of the expert down-projection (bf16 slabs)
extern int g_train_dn_split;                                                             // "train_dn_split" option (dit.hip)
// K-slices of the training forward's expert down-projection (bf16 slabs, added in slice order by the forward and the backward combine): -1 = auto - the
// inference chain's slices (then the training forward with dropouts off IS the inference forward, bit for bit) from 2048 sorted rows on, where the four
// slices give the persistent ping-pong kernel exactly one 256 x 256 tile per CU (round 6: 53 -> ~33 us per block, 10.35-10.52 -> 10.23-10.25 ms per step);
// one fp32-accumulated slab below that (small batches: the ring kernel has enough tiles, and one rounding less).  0 / 1 force either form.
static inline int train_dn_split(int dt, int K, long NK) {
  const bool on = g_train_dn_split < 0 ? NK >= 2048 : g_train_dn_split != 0;
  return on ? down_proj_split(dt, K) : 1;
}
int combine_bwd_launch(const float* dy, const void* Y, int y_dtype, int y_splits, long y_split_stride, const int32_t* pos, const float* posw, int N, int D, int k,
                       void* dYs, float* dw, void* stream);                               // train_ops.hip
}

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Take {
  size_t o = 0;
  size_t operator()(size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; }
};

ModeGemmDesc gdesc(int dtype, int epi, int out_dtype, int M, int N, int K, const void* A, long lda, const void* W, long ldw, void* C, long ldc) {
  ModeGemmDesc g;
  memset(&g, 0, sizeof(g));
  g.dtype = dtype; g.epilogue = epi; g.out_dtype = out_dtype; g.M = M; g.N = N; g.K = K;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc;
  return g;
}

struct TrainWs {   // backward workspace offsets
  size_t dxa, dxb, dyl, dys, dhd, dp, dus, dwt, t_big, t_mid, t_d, t_d2, dx1lp, dyattn, dqkv, dh1, dgp, apq, apk, csw, dcond, dlog, dhid, dpre,
      st1, st2, tmp_rd, demb, de1, dimg, dgoal, tr_dlog, tr_dpre, tr_u, tr_du, dys2, dp2, total;
};

TrainWs train_ws(const ModeDims& d, int B, int dtype) {
  const size_t esz = dtype == MODE_BF16 ? 2 : 4;
  const size_t N = (size_t)B * d.T, NK = N * d.k, D = d.D;
  const size_t NKp = (NK + 63) / 64 * 64 + 64 * (size_t)d.E, Np = (N + 63) / 64 * 64;
  const size_t R = (size_t)B * d.A_len;
  TrainWs w{};
  Take t;
  w.dxa = t(N * D * 4); w.dxb = t(N * D * 4); w.dyl = t(N * D * 4);
  w.dys = t(NK * D * esz); w.dhd = t(NK * 4 * D * esz); w.dp = t(NK * 8 * D * esz); w.dus = t(NK * D * 4 * 4);   /* dU: up to four K-slice slabs of the up-projection data gradient */ w.dwt = t(NK * 4 * (size_t)d.L);      // router-weight gradients of ALL layers [L][N*k]
  w.t_big = t(8 * D * NKp * esz);            // dP^T  [8D, NKp]   (also dqkv^T [3D, Np])
  w.t_mid = t(4 * D * NKp * esz);            // Hd^T  [4D, NKp]
  w.t_d = t(D * NKp * esz);                  // dY^T / u^T / dx1^T / h1^T  [D, NKp]
  w.t_d2 = t(D * NKp * esz);
  w.dx1lp = t(N * D * esz); w.dyattn = t(N * D * esz); w.dqkv = t(N * 3 * D * esz); w.dh1 = t(N * D * 4);
  w.dgp = t(((N + 3) / 4) * D * 4 * (2 * (size_t)d.L + 1));            // per-block gain-gradient partials: ln_1 / ln_2 of every layer + final ln
  w.apq = t((size_t)d.L * B * d.H * (D / d.H) * 4); w.apk = t((size_t)d.L * B * d.H * (D / d.H) * 4);   // q/k-norm gain partials of every layer
  size_t cs = mode_colsum_workspace_bytes((int)NK, 8 * d.D, d.E);
  const size_t c2 = mode_colsum_workspace_bytes((int)N, 3 * d.D, 1), c3 = mode_colsum_workspace_bytes((int)N, d.D, B);
  if (c2 > cs) cs = c2;
  if (c3 > cs) cs = c3;
  const size_t c4 = mode_swiglu_bwd_bias_workspace_bytes((int64_t)NK, 4 * d.D, d.E);     // fused SwishGLU backward + bias sums
  if (c4 > cs) cs = c4;
  w.csw = t(cs + 4096);
  w.dcond = t((size_t)B * D * 4); w.dlog = t((size_t)d.L * B * d.E * 4);           // dlogits of ALL layers [L][B][E]
  w.dhid = t((size_t)d.L * 4 * B * D * 4 + 256 * 4);                               // partial dcond [4L][B][D] (4 K-slices per layer) + K-group offsets
  w.dpre = t((size_t)B * d.L * 2 * D * 4);                                         // dpre of ALL layers [B][L][2D]
  const size_t smallT = ((size_t)2 * D > (size_t)d.O ? 2 * D : d.O) * (R > 2 * (size_t)B ? R : 2 * B) * 4 + 4096;
  w.st1 = t(smallT); w.st2 = t(smallT);
  w.tmp_rd = t(R * D * 4);
  w.demb = t((size_t)B * D * 4); w.de1 = t((size_t)B * D * 4); w.dimg = t((size_t)B * d.n_img * D * 4); w.dgoal = t((size_t)B * D * 4);
  // token routing (cond_router=False): one layer's router backward - dlogits [N,E], dpre [N,2D], recomputed ln_2 output [N,D], d u from the router [N,D]
  w.tr_dlog = t(N * (size_t)d.E * 4); w.tr_dpre = t(N * 2 * D * 4); w.tr_u = t(N * D * 4); w.tr_du = t(N * D * 4);
  // second dY / dP pair (bf16 only): with the fused weight-gradient + optimizer launches on a side stream (ModeAdamWFuse.side_stream) odd blocks use these, so
  // that the side stream may still read block l's operands while the chain writes block l-1's
  w.dys2 = t(dtype == MODE_BF16 ? NK * D * esz : 0); w.dp2 = t(dtype == MODE_BF16 ? NK * 8 * D * esz : 0);
  w.total = t.o;
  return w;
}

int check_train_dims(const ModeDims* d) {
  if (!d) return MODE_ERR_BAD_ARG;
  if (d->D <= 0 || d->H <= 0 || d->D % d->H || d->L <= 0 || d->E <= 0 || d->k <= 0 || d->k > d->E || d->k > 8) return MODE_ERR_BAD_ARG;
  if (d->T != (d->use_noise_token ? 1 : 0) + 1 + d->n_img + d->A_len) return MODE_ERR_UNSUPPORTED;
  if (d->D % 4 || d->A_dim > 8 || d->T > 16) return MODE_ERR_UNSUPPORTED;
  return MODE_OK;
}

}  // namespace

extern "C" int mode_dit_train_stash_layout(const ModeDims* dims, int B, int dtype, ModeStashLayout* out) {
  int rc = check_train_dims(dims);
  if (rc) return rc;
  if (!out || B <= 0) return MODE_ERR_BAD_ARG;
  const size_t esz = dtype == MODE_BF16 ? 2 : 4;
  const size_t N = (size_t)B * dims->T, NK = N * dims->k, D = dims->D;
  Take t;
  out->x0 = t(N * D * 4); out->h1 = t(N * D * esz); out->qkv = t(N * 3 * D * esz); out->yattn = t(N * D * esz); out->x1 = t(N * D * 4);
  // Y: the down-projection's split-K slabs (the inference chain's tiling: one 256-row x 256 x 1024 tile per CU; the combine kernels add the slabs)
  out->ub = t(N * D * esz); out->P = t(NK * 8 * D * esz); out->Hd = t(NK * 4 * D * esz); out->Y = t(NK * D * esz * (size_t)train_dn_split(dtype, 4 * (int)D, (long)NK));
  out->layer_stride = t.o;
  Take g;
  out->xL = g(N * D * 4); out->yL = g(N * D * 4); out->u_tmp = g(N * D * 4);
  out->tr_hid = g(N * 2 * D * 4); out->tr_logits = g(N * (size_t)dims->E * 4);      /* token routing (cond_router=False): one layer's router scratch */
  out->global_bytes = g.o;
  out->total_bytes = out->global_bytes + out->layer_stride * dims->L;
  return MODE_OK;
}

extern "C" size_t mode_dit_train_workspace_bytes(const ModeDims* dims, int B, int dtype) {
  if (check_train_dims(dims) != MODE_OK || B <= 0) return 0;
  return train_ws(*dims, B, dtype).total;
}

// ------------------------------------------------------------------------------------------------------------------ forward
// Layers [l0, l1) of the training forward; phases bit 0 = everything up to and including ln_2 (+ the token router of cond_router=False), bit 1 =
// experts + combine (+ the output head behind the last layer).  Conditioning-row routing runs all layers with both phases in one call; token
// routing is driven layer by layer from the host, which draws the expert ids between the two phases (modedit.py:390: torch.multinomial per token).
static int forward_train_impl(const ModeDims* dims, const ModeModelWeights* w, const ModeTrainArgs* a, void* stash, size_t stash_bytes, int l0, int l1,
                              int phases, void* stream) {
  int rc = check_train_dims(dims);
  if (rc) return rc;
  if (!w || !w->layers || !a || !stash || !a->meta || !a->goal_e || !a->img_e || !a->actions || !a->emb_t || !a->cond || !a->act_rows || !a->F)
    return MODE_ERR_BAD_ARG;
  const ModeDims& d = *dims;
  if (l0 < 0 || l1 > d.L || l0 >= l1) return MODE_ERR_BAD_ARG;
  const int dt = a->dtype, B = a->B, T = d.T, D = d.D, N = B * T, NK = N * d.k;
  if (dt == MODE_BF16 && (D % 64 || (D / d.H) % 16 || (D / d.H) > 128)) return MODE_ERR_UNSUPPORTED;
  if (a->token_routing && (!a->tr_pre || !a->probs || !a->tr_shifted || !a->tr_topk_idx || !a->tr_topk_w)) return MODE_ERR_BAD_ARG;
  ModeStashLayout sl;
  rc = mode_dit_train_stash_layout(dims, B, dt, &sl);
  if (rc) return rc;
  if (stash_bytes < sl.total_bytes) return MODE_ERR_WORKSPACE;
  const int ysplit = train_dn_split(dt, 4 * D, NK);
  char* sg = (char*)stash;
  auto L_ = [&](int l) { return sg + sl.global_bytes + (size_t)l * sl.layer_stride; };
  float* u_tmp = (float*)(sg + sl.u_tmp);
  ModeMetaLayout ml;
  mode_moe_meta_layout(N, d.E, d.k, &ml);

  if (l0 == 0 && (phases & 1)) {
    ModeEmbedDesc e;
    memset(&e, 0, sizeof(e));
    e.B = B; e.T = T; e.D = D; e.A_len = d.A_len; e.A_dim = d.A_dim; e.n_img = d.n_img; e.use_noise_token = d.use_noise_token;
    e.emb_t = a->emb_t; e.emb_row_stride = D; e.goal_e = a->goal_e; e.img_e = a->img_e; e.actions = a->actions;
    e.c_in = a->c_in; e.c_in_stride = a->c_in_stride; e.w_act = w->w_act; e.pos = w->pos; e.g = w->layers[0].ln1_g;
    e.cond = a->cond; e.cond_row_stride = D; e.eps = d.eps; e.x = (float*)(L_(0) + sl.x0); e.h = L_(0) + sl.h1; e.h_dtype = dt;
    rc = mode_embed_tokens_fwd(&e, stream);
    if (rc) return rc;
  }

  for (int l = l0; l < l1; ++l) {
    const ModeLayerWeights& lw = w->layers[l];
    char* S = L_(l);
    const int32_t* meta = a->meta + (long)l * a->meta_layer_stride;
    float* x0 = (float*)(S + sl.x0); float* x1 = (float*)(S + sl.x1);
    ModeGemmDesc g;
    if (phases & 1) {
      g = gdesc(dt, MODE_EPI_BIAS, dt, N, 3 * D, D, S + sl.h1, D, lw.wqkv, D, S + sl.qkv, 3 * D);
      g.bias = lw.bqkv;
      if ((rc = mode_gemm(&g, stream))) return rc;
      if ((rc = mode_attn_block_fwd(S + sl.qkv, lw.qn_g, lw.kn_g, S + sl.yattn, dt, B, T, d.H, D / d.H, d.eps, mode_stream_seed(a->seed, 2 * l), a->attn_pdrop, stream)))
        return rc;
      g = gdesc(dt, MODE_EPI_RESIDUAL, MODE_F32, N, D, D, S + sl.yattn, D, lw.wo, D, x1, D);
      g.resid = x0; g.ldr = D;
      if ((rc = mode_gemm(&g, stream))) return rc;
      if ((rc = mode_rmsnorm_cond_fwd(x1, lw.ln2_g, nullptr, N, D, 1, d.eps, u_tmp, S + sl.ub, dt, stream))) return rc;
      if (a->token_routing) {
        // router(x, None) on the ln_2-normalised TOKEN states (modedit.py:296-301, 322-325, 553): fp32 like every router of this library; the
        // pre-GELU activations stay for the backward; probabilities / shifted logits / top-k of THIS layer go out to the host
        float* pre = a->tr_pre + (size_t)l * N * 2 * D;
        float* hid = (float*)(sg + sl.tr_hid); float* logits = (float*)(sg + sl.tr_logits);
        ModeGemmDesc rg = gdesc(MODE_F32, MODE_EPI_BIAS, MODE_F32, N, 2 * D, D, u_tmp, D, lw.r_w0, D, pre, 2L * D);
        rg.bias = lw.r_b0;
        if ((rc = mode_gemm(&rg, stream))) return rc;
        if ((rc = mode_gelu_fwd(pre, hid, (int64_t)N * 2 * D, stream))) return rc;
        if ((rc = mode_router_logits(hid, 2L * D, lw.r_w3, 0, lw.r_b3, 0, 1, N, d.E, 2 * D, logits, stream))) return rc;
        if ((rc = mode_moe_route_topk_f32(logits, N, d.E, d.k, d.router_normalize, a->tr_shifted + (size_t)l * N * d.E,
                                          const_cast<float*>(a->probs) + (size_t)l * N * d.E, a->tr_topk_idx, a->tr_topk_w, stream))) return rc;
      }
    }
    if (phases & 2) {
      g = gdesc(dt, MODE_EPI_BIAS, dt, NK, 8 * D, D, S + sl.ub, D, lw.w1, D, S + sl.P, 8 * D);          // pre-activation [value | gate] kept for backward
      g.bias = lw.b1; g.w_expert_stride = 8L * D * D; g.bias_expert_stride = 8L * D;
      g.a_rows = meta + ml.perm; g.expert_offsets = meta + ml.offsets; g.num_experts = d.E;
      if ((rc = mode_gemm(&g, stream))) return rc;
      if ((rc = mode_swiglu_fwd(S + sl.P, S + sl.Hd, NK, 4 * D, dt, mode_stream_seed(a->seed, 2 * l + 1), a->mlp_pdrop, stream))) return rc;
      g = gdesc(dt, MODE_EPI_NONE, dt, NK, D, 4 * D, S + sl.Hd, 4 * D, lw.w2, 4 * D, S + sl.Y, D);
      g.w_expert_stride = 4L * D * D; g.expert_offsets = meta + ml.offsets; g.num_experts = d.E;
      g.split_k = ysplit; g.split_stride = (long)NK * D;
      if ((rc = mode_gemm(&g, stream))) return rc;
      const bool last = l + 1 == d.L;
      float* xn = last ? (float*)(sg + sl.xL) : (float*)(L_(l + 1) + sl.x0);
      rc = mode_moe_combine_norm_fwd(u_tmp, S + sl.Y, dt, ysplit, (long)NK * D, meta + ml.pos, reinterpret_cast<const float*>(meta + ml.posw), N, D, d.k,
                                     last ? nullptr : w->layers[l + 1].ln1_g, last ? nullptr : a->cond, T, d.eps, xn,
                                     last ? nullptr : (void*)(L_(l + 1) + sl.h1), dt, stream);
      if (rc) return rc;
    }
  }
  if (l1 == d.L && (phases & 2)) {
    float* yL = (float*)(sg + sl.yL);
    if ((rc = mode_rmsnorm_cond_fwd((const float*)(sg + sl.xL), w->ln_g, nullptr, N, D, 1, d.eps, yL, nullptr, MODE_F32, stream))) return rc;
    ModeGemmDesc g = gdesc(MODE_F32, MODE_EPI_BIAS, MODE_F32, B * d.A_len, d.A_dim, D, yL, D, w->w_out, D, a->F, d.A_dim);
    g.bias = w->b_out; g.a_rows = a->act_rows;
    return mode_gemm(&g, stream);
  }
  return MODE_OK;
}

extern "C" int mode_dit_forward_train(const ModeDims* dims, const ModeModelWeights* w, const ModeTrainArgs* a, void* stash, size_t stash_bytes,
                                      void* stream) {
  if (a && a->token_routing) return MODE_ERR_BAD_ARG;           // token routing is driven layer by layer: mode_dit_forward_train_layer
  return forward_train_impl(dims, w, a, stash, stash_bytes, 0, dims ? dims->L : 0, 3, stream);
}

extern "C" int mode_dit_forward_train_layer(const ModeDims* dims, const ModeModelWeights* w, const ModeTrainArgs* a, void* stash, size_t stash_bytes,
                                            int layer, int phase, void* stream) {
  if (phase != 0 && phase != 1) return MODE_ERR_BAD_ARG;
  return forward_train_impl(dims, w, a, stash, stash_bytes, layer, layer + 1, phase == 0 ? 1 : 2, stream);
}

// ----------------------------------------------------------------------------------------------------------------- backward
extern "C" int64_t mode_adamw_fuse_gsq_floats(const ModeDims* dims) {
  if (!dims || dims->D <= 0 || dims->E <= 0 || dims->L <= 0) return 0;
  const long t = (dims->D + 127) / 128;
  return (long)dims->L * dims->E * 12 * t * t;
}

extern "C" int mode_dit_backward(const ModeDims* dims, const ModeModelWeights* w, const ModeModelWeightsT* wt, const ModeTrainArgs* a,
                                 const void* stash, const float* dF, const ModeModelGrads* gr, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  int rc = check_train_dims(dims);
  if (rc) return rc;
  if (!w || !w->layers || !wt || !wt->layers || !a || !stash || !dF || !gr || !gr->layers || !workspace) return MODE_ERR_BAD_ARG;
  if (!a->meta || !a->act_rows || !a->probs || !a->topk_idx || !a->sigma || !a->state_images || !a->goals || !a->e1) return MODE_ERR_BAD_ARG;
  const bool tokr = a->token_routing != 0;
  if (tokr ? (!a->tr_pre || !a->idx_per_token) : !a->r_pre) return MODE_ERR_BAD_ARG;
  const ModeDims& d = *dims;
  const int dt = a->dtype, B = a->B, T = d.T, D = d.D, N = B * T, NK = N * d.k, E = d.E, R = B * d.A_len, A = d.A_dim, hd = D / d.H;
  for (int l = 1; l < d.L; ++l) {                      // the batched router backward needs layer-contiguous router weights / gradients
    const ModeLayerWeights& wl = w->layers[l]; const ModeLayerWeights& w0 = w->layers[0];
    const ModeLayerGrads& gl = gr->layers[l]; const ModeLayerGrads& g0 = gr->layers[0];
    if (wl.r_w0 != w0.r_w0 + (long)l * 2 * D * D || wl.r_w3 != w0.r_w3 + (long)l * E * 2 * D || gl.r_w0 != g0.r_w0 + (long)l * 2 * D * D ||
        gl.r_b0 != g0.r_b0 + (long)l * 2 * D || gl.r_w3 != g0.r_w3 + (long)l * E * 2 * D || gl.r_b3 != g0.r_b3 + (long)l * E)
      return MODE_ERR_UNSUPPORTED;
  }
  const bool tr = dt == MODE_BF16 && D % 8 == 0;      // bf16: backward GEMMs read row-major operands directly (gemm_bf16_tr.hip); no transposed copies
  int du_split = (tr && (8 * D) % 128 == 0) ? 2 : 1;   // K-slices of the up-projection data gradient
  const bool bf = dt == MODE_BF16;
  const long NKp = ((long)NK + 63) / 64 * 64 + 64L * E, Np = ((long)N + 63) / 64 * 64;
  const int Ktok = bf ? (int)Np : N;                       // token-dim contraction length (bf16 kernel needs a multiple of 64: zero padded)
  const size_t esz = bf ? 2 : 4;
  ModeStashLayout sl;
  if ((rc = mode_dit_train_stash_layout(dims, B, dt, &sl))) return rc;
  const TrainWs W = train_ws(d, B, dt);
  if (workspace_bytes < W.total) return MODE_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  const char* sg = (const char*)stash;
  auto L_ = [&](int l) { return sg + sl.global_bytes + (size_t)l * sl.layer_stride; };
  hipStream_t hs = (hipStream_t)stream;
  ModeMetaLayout ml;
  mode_moe_meta_layout(N, E, d.k, &ml);
  float* DXa = (float*)(ws + W.dxa); float* DXb = (float*)(ws + W.dxb);
  float* dcond = (float*)(ws + W.dcond);
  void* csw = ws + W.csw; const size_t cswb = W.dcond - W.csw;        // everything up to the next carve
  float* dgp = (float*)(ws + W.dgp);
  const int nblk4 = (N + 3) / 4;
  auto colsum = [&](const void* X, long ld, int rows, int cols, int xdt, const int32_t* segoff, int seglen, int nseg, float* out, int acc) {
    return mode_colsum(X, ld, rows, cols, xdt, segoff, seglen, nseg, out, acc, csw, cswb, stream);
  };
  if (hipMemsetAsync(dcond, 0, (size_t)B * D * 4, hs) != hipSuccess) return (int)hipGetLastError();

  // ---- output head: F = yL[act_rows] @ w_out^T + b_out ;  yL = RMSNorm(xL; ln.g)
  {
    float* tmp = (float*)(ws + W.tmp_rd);                   // d yL on the action rows  [R, D]
    ModeGemmDesc g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, R, D, A, dF, A, wt->w_outT, A, tmp, D);
    if ((rc = mode_gemm(&g, stream))) return rc;
    float* dFT = (float*)(ws + W.st1); float* yT = (float*)(ws + W.st2);
    if ((rc = mode_transpose(dF, A, R, A, dFT, R, nullptr, nullptr, MODE_F32, stream))) return rc;
    if ((rc = mode_transpose(sg + sl.yL, D, R, D, yT, R, a->act_rows, nullptr, MODE_F32, stream))) return rc;
    g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, A, D, R, dFT, R, yT, R, gr->w_out, D);
    g.flags = MODE_GEMM_SKINNY_OK;
    if ((rc = mode_gemm(&g, stream))) return rc;
    if ((rc = colsum(dF, A, R, A, MODE_F32, nullptr, 0, 1, gr->b_out, 0))) return rc;
    float* dyl = (float*)(ws + W.dyl);
    if (hipMemsetAsync(dyl, 0, (size_t)N * D * 4, hs) != hipSuccess) return (int)hipGetLastError();
    if ((rc = mode_rowcopy_f32(tmp, D, 0, 1, nullptr, dyl, D, 0, 0, a->act_rows, nullptr, 0, R, D, stream))) return rc;
    if ((rc = mode_rmsnorm_bwd((const float*)(sg + sl.xL), w->ln_g, dyl, nullptr, nullptr, nullptr, 0, N, D, d.eps, DXa, 0, dgp, nullptr, nullptr,
                               MODE_F32, stream))) return rc;
    if ((rc = colsum(dgp, D, nblk4, D, MODE_F32, nullptr, 0, 1, gr->ln_g, 0))) return rc;
  }

  void* dYs = ws + W.dys; void* dHd = ws + W.dhd; void* dP = ws + W.dp; float* dUs = (float*)(ws + W.dus); float* dwt = (float*)(ws + W.dwt);
  void* Tbig = ws + W.t_big; void* Tmid = ws + W.t_mid; void* Td = ws + W.t_d; void* Td2 = ws + W.t_d2;
  void* dx1lp = ws + W.dx1lp; void* dyattn = ws + W.dyattn; void* dqkv = ws + W.dqkv; float* dh1 = (float*)(ws + W.dh1);
  float* apq = (float*)(ws + W.apq); float* apk = (float*)(ws + W.apk);
  float* dlog = (float*)(ws + W.dlog); float* dhid = (float*)(ws + W.dhid); float* dpre = (float*)(ws + W.dpre);
  float* st1 = (float*)(ws + W.st1); float* st2 = (float*)(ws + W.st2);
#define ZERO(ptr, bytes) if (hipMemsetAsync((ptr), 0, (bytes), hs) != hipSuccess) return (int)hipGetLastError();
#define MODE_HIP_OK(call) if ((call) != hipSuccess) return (int)hipGetLastError();

  // (ABI 11) AdamW in the epilogue of the expert weight-gradient GEMMs: bf16 transpose-read path only
  ModeAdamWFuse fz;
  const long t128 = (D + 127) / 128;
  const long fz_w2 = (long)E * t128 * (4 * t128), fz_w1 = (long)E * (8 * t128) * t128, fz_per_layer = fz_w2 + fz_w1;
  if (a->fuse_adamw) {
    if (!tr || (4 * D) % 128 || D % 128) return MODE_ERR_UNSUPPORTED;
    if (a->fuse_adamw->gsq && a->fuse_adamw->gsq_capacity < fz_per_layer * d.L) return MODE_ERR_WORKSPACE;
  }
  // Second stream for the fused weight-gradient + optimizer launches (ModeAdamWFuse.side_stream): they are HBM-bound, the data-gradient chain MFMA-bound.
  // Block parity selects the dY / dP buffers (the transposed-operand scratch of the fp32 path, unused here, is the second pair), so that the side stream may
  // lag one block behind the chain.
  const bool side = a->fuse_adamw && a->fuse_adamw->side_stream && a->fuse_adamw->side_events;
  hipStream_t ss = side ? (hipStream_t)a->fuse_adamw->side_stream : hs;
  hipEvent_t ev_w2 = nullptr, ev_w1 = nullptr, ev_done[2] = {nullptr, nullptr};
  if (side) {
    ev_w2 = (hipEvent_t)a->fuse_adamw->side_events[0]; ev_w1 = (hipEvent_t)a->fuse_adamw->side_events[1];
    ev_done[0] = (hipEvent_t)a->fuse_adamw->side_events[2]; ev_done[1] = (hipEvent_t)a->fuse_adamw->side_events[3];
    if (!ev_w2 || !ev_w1 || !ev_done[0] || !ev_done[1]) return MODE_ERR_BAD_ARG;
  }
  bool done_rec[2] = {false, false};
  for (int l = d.L - 1; l >= 0; --l) {
    const ModeLayerWeights& lw = w->layers[l];
    const ModeLayerWeightsT* lt = &wt->layers[l];
    const ModeLayerGrads& lg = gr->layers[l];
    const char* S = L_(l);
    const int32_t* meta = a->meta + (long)l * a->meta_layer_stride;
    const int32_t* offsets = meta + ml.offsets; const int32_t* poff = meta + ml.poffsets; const int32_t* prow = meta + ml.prow;
    const int32_t* pos = meta + ml.pos; const float* posw = reinterpret_cast<const float*>(meta + ml.posw);
    const int par = l & 1;
    if (side) {                                                  // this block's dY / dP pair; the side stream must be done with the block that used it last
      dYs = par ? ws + W.dys2 : ws + W.dys; dP = par ? ws + W.dp2 : ws + W.dp;
      if (done_rec[par]) MODE_HIP_OK(hipStreamWaitEvent(hs, ev_done[par], 0));
    }
    // (1) combine backward: dY (sorted rows) and router-weight gradients
    if ((rc = combine_bwd_launch(DXa, S + sl.Y, dt, train_dn_split(dt, 4 * D, NK), (long)NK * D, pos, posw, N, D, d.k, dYs, dwt + (size_t)l * NK, stream))) return rc;
    // (1b) token routing: this block's router, back-propagated in place - its input is the block's own ln_2 output, so d u gets a second term
    const float* du_router = nullptr;
    if (tokr) {
      float* dlogl = (float*)(ws + W.tr_dlog); float* dprel = (float*)(ws + W.tr_dpre); float* u_re = (float*)(ws + W.tr_u); float* dur = (float*)(ws + W.tr_du);
      const float* probs_l = a->probs + (size_t)l * N * E;
      const float* shifted_l = a->shifted ? a->shifted + (size_t)l * N * E : nullptr;
      const float* pre_l = a->tr_pre + (size_t)l * N * 2 * D;
      if ((rc = mode_moe_router_bwd_aux(dwt + (size_t)l * NK, a->topk_idx + (long)l * a->topk_layer_stride, probs_l, shifted_l,
                                        a->aux_lb_coef ? a->aux_lb_coef + (long)l * E : nullptr, shifted_l ? a->aux_z_coef : nullptr, N, N, 1, E, d.k,
                                        d.router_normalize, 1, dlogl, stream))) return rc;
      if ((rc = colsum(dlogl, E, N, E, MODE_F32, nullptr, 0, 1, lg.r_b3, 0))) return rc;
      if ((rc = mode_router_mlp_bwd(dlogl, pre_l, lw.r_w3, 1, N, E, 2 * D, dprel, lg.r_w3, stream))) return rc;
      if ((rc = colsum(dprel, 2L * D, N, 2 * D, MODE_F32, nullptr, 0, 1, lg.r_b0, 0))) return rc;
      // the router read the fp32 ln_2 output (not its bf16 copy): recompute it from the stashed pre-norm stream
      if ((rc = mode_rmsnorm_cond_fwd((const float*)(S + sl.x1), lw.ln2_g, nullptr, N, D, 1, d.eps, u_re, nullptr, MODE_F32, stream))) return rc;
      ModeGemmDesc rg = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, 2 * D, D, N, dprel, 2L * D, u_re, D, lg.r_w0, D);      // dW0 = dpre^T u
      rg.flags = MODE_GEMM_A_KM | MODE_GEMM_W_KN;
      if ((rc = mode_gemm(&rg, stream))) return rc;
      rg = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, N, D, 2 * D, dprel, 2L * D, lw.r_w0, D, dur, D);                     // d u (router) = dpre W0
      rg.flags = MODE_GEMM_W_KN;
      if ((rc = mode_gemm(&rg, stream))) return rc;
      du_router = dur;
    }
    // (2) expert down-projection: dH = dY W2 ; dW2_e = dY_e^T H_e
    ModeGemmDesc g;
    // (2) + (3) as ONE launch (bf16): the data gradient's 128 x 128 tile never leaves the chip - its epilogue reads the stashed pre-activations, applies the
    // SwishGLU (+ dropout) backward and writes dP and the bias-gradient partial sums (one row per m-tile, summed per expert below).  Saves the 29 MB dH round
    // trip and a 146-MB elementwise pass per layer (measured in the step: profiles/r04_train_step_kernel_stats.txt).
    const size_t bsum_bytes = ((size_t)NK / 128 + E + 1) * 8 * D * 4;
    const bool fuse_sb = tr && mode::g_fuse_swiglu_bwd && (4 * D) % 128 == 0 && D % 64 == 0 && E <= 16 && cswb >= bsum_bytes + 4096 + 256;
    if (fuse_sb) {
      float* bsum = (float*)csw;
      int* toff = (int*)((char*)csw + bsum_bytes);
      g = gdesc(dt, MODE_EPI_NONE, dt, NK, 4 * D, D, dYs, D, lw.w2, 4 * D, nullptr, 4 * D);
      g.w_expert_stride = 4L * D * D; g.expert_offsets = offsets; g.num_experts = E; g.flags = MODE_GEMM_W_KN;
      const float pd = a->mlp_pdrop;
      const uint32_t th = pd <= 0.f ? 0u : (uint32_t)((double)pd * 4294967296.0);
      if ((rc = mode::gemm_bf16_tr_swiglu_bwd_launch(&g, S + sl.P, dP, mode_stream_seed(a->seed, 2 * l + 1), th, 1.0f / (1.0f - pd), bsum, toff, hs))) return rc;
      if ((rc = mode_colsum(bsum, 8L * D, NK / 128 + E, 8 * D, MODE_F32, toff, 0, E, lg.b1, 0, (char*)csw + bsum_bytes + 4096, cswb - bsum_bytes - 4096, stream))) return rc;
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, D, 4 * D, NK, dYs, D, S + sl.Hd, 4 * D, lg.w2, 4 * D);
      g.k_group_offsets = offsets; g.num_k_groups = E; g.c_group_stride = 4L * D * D; g.flags = MODE_GEMM_W_KN | MODE_GEMM_A_KM;
      if (a->fuse_adamw) { fz = *a->fuse_adamw; if (fz.gsq) { fz.gsq += (long)l * fz_per_layer; fz.gsq_capacity = fz_w2; } g.adamw = &fz; }   // W2 is updated here: no gradient is written
      if (side) { MODE_HIP_OK(hipEventRecord(ev_w2, hs)); MODE_HIP_OK(hipStreamWaitEvent(ss, ev_w2, 0)); }   // ... after the data gradient above has read it
      if ((rc = mode_gemm(&g, side ? (void*)ss : stream))) return rc;
    } else if (tr) {                                 // bf16: operands as they lie in memory, fragments by LDS transpose reads
      g = gdesc(dt, MODE_EPI_NONE, dt, NK, 4 * D, D, dYs, D, lw.w2, 4 * D, dHd, 4 * D);
      g.w_expert_stride = 4L * D * D; g.expert_offsets = offsets; g.num_experts = E; g.flags = MODE_GEMM_W_KN;
      if ((rc = mode_gemm(&g, stream))) return rc;
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, D, 4 * D, NK, dYs, D, S + sl.Hd, 4 * D, lg.w2, 4 * D);
      g.k_group_offsets = offsets; g.num_k_groups = E; g.c_group_stride = 4L * D * D; g.flags = MODE_GEMM_W_KN | MODE_GEMM_A_KM;
      if (a->fuse_adamw) { fz = *a->fuse_adamw; if (fz.gsq) { fz.gsq += (long)l * fz_per_layer; fz.gsq_capacity = fz_w2; } g.adamw = &fz; }   // W2 is updated here: no gradient is written
      if (side) { MODE_HIP_OK(hipEventRecord(ev_w2, hs)); MODE_HIP_OK(hipStreamWaitEvent(ss, ev_w2, 0)); }   // ... after the data gradient above has read it
      if ((rc = mode_gemm(&g, side ? (void*)ss : stream))) return rc;
    } else {
      g = gdesc(dt, MODE_EPI_NONE, dt, NK, 4 * D, D, dYs, D, lt->w2T, D, dHd, 4 * D);
      g.w_expert_stride = 4L * D * D; g.expert_offsets = offsets; g.num_experts = E;
      if ((rc = mode_gemm(&g, stream))) return rc;
      ZERO(Td, (size_t)D * NKp * esz); ZERO(Tmid, (size_t)4 * D * NKp * esz);
      if ((rc = mode_transpose(dYs, D, NK, D, Td, NKp, nullptr, prow, dt, stream))) return rc;
      if ((rc = mode_transpose(S + sl.Hd, 4 * D, NK, 4 * D, Tmid, NKp, nullptr, prow, dt, stream))) return rc;
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, D, 4 * D, (int)NKp, Td, NKp, Tmid, NKp, lg.w2, 4 * D);
      g.k_group_offsets = poff; g.num_k_groups = E; g.c_group_stride = 4L * D * D;
      if ((rc = mode_gemm(&g, stream))) return rc;
    }
    // (3) SwishGLU (+ dropout) backward, bias gradient
    if (fuse_sb) {
      // done inside the data-gradient GEMM above
    } else if (dt == MODE_BF16 && D % 2 == 0 && E <= 16) {                  // one pass: dP and the per-expert column sums of dP
      if ((rc = mode_swiglu_bwd_bias(S + sl.P, dHd, dP, NK, 4 * D, dt, mode_stream_seed(a->seed, 2 * l + 1), a->mlp_pdrop, offsets, E, lg.b1, csw, cswb, stream))) return rc;
    } else {
      if ((rc = mode_swiglu_bwd(S + sl.P, dHd, dP, NK, 4 * D, dt, mode_stream_seed(a->seed, 2 * l + 1), a->mlp_pdrop, stream))) return rc;
      if ((rc = colsum(dP, 8 * D, NK, 8 * D, dt, offsets, 0, E, lg.b1, 0))) return rc;
    }
    // (4) expert up-projection: dU (sorted rows, fp32) = dP W1 ; dW1_e = dP_e^T U_e
    if (tr) {
      // K = 8D with only NK*D/128^2 = 224 output tiles: two K-slices (fp32 slabs, added by the ln_2 backward's gather-sum) double the workgroups
      // and allow 128-wide tiles, as for the inference down-projection (112 -> ~80 us)
      // large batches: the persistent ping-pong kernel (gemm_bf16_pptr.hip) takes 256 x 256 tiles - FOUR K-slices give it one tile per CU
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, NK, D, 8 * D, dP, 8 * D, lw.w1, D, dUs, D);
      g.w_expert_stride = 8L * D * D; g.expert_offsets = offsets; g.num_experts = E; g.flags = MODE_GEMM_W_KN;
      g.split_k = 4; g.split_stride = (long)NK * D;
      du_split = ((8 * D) % 512 == 0 && mode::gemm_bf16_pptr_accepts(&g)) ? 4 : du_split;
      g.split_k = du_split;
      if ((rc = mode_gemm(&g, stream))) return rc;
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, 8 * D, D, NK, dP, 8 * D, Td2, D, lg.w1, D);
      g.k_group_offsets = offsets; g.num_k_groups = E; g.c_group_stride = 8L * D * D;
      g.flags = MODE_GEMM_W_KN | MODE_GEMM_A_KM;
      if (a->fuse_adamw) { fz = *a->fuse_adamw; if (fz.gsq) { fz.gsq += (long)l * fz_per_layer + fz_w2; fz.gsq_capacity = fz_w1; } g.adamw = &fz; }   // W1 likewise (ring kernel: gathers u through perm)
      if (!a->fuse_adamw && mode::gemm_bf16_pptr_accepts(&g)) { // it reads its K rows where they lie: sorted-order copy of the u rows first (7 MB)
        if ((rc = mode::gather_rows_bf16(S + sl.ub, D, meta + ml.perm, NK, D, Td2, D, hs))) return rc;
      } else {
        g.W = S + sl.ub; g.w_rows = meta + ml.perm;             // ring kernel: u rows gathered through perm inside the GEMM
      }
      if (side) { MODE_HIP_OK(hipEventRecord(ev_w1, hs)); MODE_HIP_OK(hipStreamWaitEvent(ss, ev_w1, 0)); }   // dU (reads W1) is on the chain: W1 may be updated behind it
      if ((rc = mode_gemm(&g, side ? (void*)ss : stream))) return rc;
      if (side) { MODE_HIP_OK(hipEventRecord(ev_done[par], ss)); done_rec[par] = true; }
    } else {
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, NK, D, 8 * D, dP, 8 * D, lt->w1T, 8 * D, dUs, D);
      g.w_expert_stride = 8L * D * D; g.expert_offsets = offsets; g.num_experts = E;
      if ((rc = mode_gemm(&g, stream))) return rc;
      ZERO(Tbig, (size_t)8 * D * NKp * esz); ZERO(Td2, (size_t)D * NKp * esz);
      if ((rc = mode_transpose(dP, 8 * D, NK, 8 * D, Tbig, NKp, nullptr, prow, dt, stream))) return rc;
      if ((rc = mode_transpose(S + sl.ub, D, NK, D, Td2, NKp, meta + ml.perm, prow, dt, stream))) return rc;     // gathered u rows
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, 8 * D, D, (int)NKp, Tbig, NKp, Td2, NKp, lg.w1, D);
      g.k_group_offsets = poff; g.num_k_groups = E; g.c_group_stride = 8L * D * D;
      if ((rc = mode_gemm(&g, stream))) return rc;
    }
    // (5) ln_2 backward: du = dx_out (residual from the normalised stream) + gather-sum of dU ; -> d x1
    float* dgp2 = dgp + (size_t)(1 + d.L + l) * nblk4 * D;                  // gain partials are reduced for all layers after the loop
    float* dgp1 = dgp + (size_t)(1 + l) * nblk4 * D;
    float* apq_l = apq + (size_t)l * B * D; float* apk_l = apk + (size_t)l * B * D;
    if ((rc = rmsnorm_bwd_launch((const float*)(S + sl.x1), lw.ln2_g, DXa, du_router, dUs, tr ? du_split : 1, (long)NK * D, pos, d.k, N, D, d.eps, DXb, 0, dgp2,
                                 nullptr, dx1lp, dt, (hipStream_t)stream)))
      return rc;
    // (6) c_proj: d yattn = dx1 Wo ; dWo = dx1^T yattn
    if (tr) {
      g = gdesc(dt, MODE_EPI_NONE, dt, N, D, D, dx1lp, D, lw.wo, D, dyattn, D);
      g.flags = MODE_GEMM_W_KN;
      if ((rc = mode_gemm(&g, stream))) return rc;
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, D, D, N, dx1lp, D, S + sl.yattn, D, lg.wo, D);
      g.flags = MODE_GEMM_W_KN | MODE_GEMM_A_KM;
      if ((rc = mode_gemm(&g, stream))) return rc;
    } else {
      g = gdesc(dt, MODE_EPI_NONE, dt, N, D, D, dx1lp, D, lt->woT, D, dyattn, D);
      if ((rc = mode_gemm(&g, stream))) return rc;
      ZERO(Td, (size_t)D * Np * esz); ZERO(Td2, (size_t)D * Np * esz);
      if ((rc = mode_transpose(dx1lp, D, N, D, Td, Np, nullptr, nullptr, dt, stream))) return rc;
      if ((rc = mode_transpose(S + sl.yattn, D, N, D, Td2, Np, nullptr, nullptr, dt, stream))) return rc;
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, D, D, Ktok, Td, Np, Td2, Np, lg.wo, D);
      if ((rc = mode_gemm(&g, stream))) return rc;
    }
    // (7) attention backward
    // (bf16: the kernel also emits each sample's share of the QKV bias gradient - the [B, 3D] partial lives in the transpose scratch this path does not use)
    float* dbq_part = tr ? (float*)Tbig : nullptr;
    if ((rc = mode::attn_block_bwd_launch(S + sl.qkv, lw.qn_g, lw.kn_g, dyattn, dqkv, apq_l, apk_l, dt, B, T, d.H, hd, d.eps, mode_stream_seed(a->seed, 2 * l), a->attn_pdrop,
                                          dbq_part, stream)))
      return rc;
    // (8) QKV projection: dh1 = dqkv Wqkv ; dWqkv = dqkv^T h1 ; db = colsum(dqkv)
    if (tr) {
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, N, D, 3 * D, dqkv, 3 * D, lw.wqkv, D, dh1, D);
      g.flags = MODE_GEMM_W_KN;
      if ((rc = mode_gemm(&g, stream))) return rc;
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, 3 * D, D, N, dqkv, 3 * D, S + sl.h1, D, lg.wqkv, D);
      g.flags = MODE_GEMM_W_KN | MODE_GEMM_A_KM;
      if ((rc = mode_gemm(&g, stream))) return rc;
    } else {
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, N, D, 3 * D, dqkv, 3 * D, lt->wqkvT, 3 * D, dh1, D);
      if ((rc = mode_gemm(&g, stream))) return rc;
      ZERO(Tbig, (size_t)3 * D * Np * esz); ZERO(Td, (size_t)D * Np * esz);
      if ((rc = mode_transpose(dqkv, 3 * D, N, 3 * D, Tbig, Np, nullptr, nullptr, dt, stream))) return rc;
      if ((rc = mode_transpose(S + sl.h1, D, N, D, Td, Np, nullptr, nullptr, dt, stream))) return rc;
      g = gdesc(dt, MODE_EPI_NONE, MODE_F32, 3 * D, D, Ktok, Tbig, Np, Td, Np, lg.wqkv, D);
      if ((rc = mode_gemm(&g, stream))) return rc;
    }
    if (dbq_part) { if ((rc = colsum(dbq_part, 3 * D, B, 3 * D, MODE_F32, nullptr, 0, 1, lg.bqkv, 0))) return rc; }
    else if ((rc = colsum(dqkv, 3 * D, N, 3 * D, dt, nullptr, 0, 1, lg.bqkv, 0))) return rc;
    // (9) ln_1 (+c) backward: d x0 = d x1 (residual) + RMSNorm'(dh1); dc_b += sum_t dh1[b,t]
    if ((rc = mode_rmsnorm_bwd((const float*)(S + sl.x0), lw.ln1_g, dh1, nullptr, nullptr, nullptr, 0, N, D, d.eps, DXb, 1, dgp1, nullptr, nullptr, MODE_F32,
                               stream))) return rc;
    if ((rc = colsum(dh1, D, N, D, MODE_F32, nullptr, T, B, dcond, 1))) return rc;
    if (a->layer_events && a->layer_events[l] && hipEventRecord((hipEvent_t)a->layer_events[l], hs) != hipSuccess) return (int)hipGetLastError();
    float* t_ = DXa; DXa = DXb; DXb = t_;                    // DXa now holds d x_l
  }

  if (side) {                                                    // join: the chain's stream owns the parameters again when the call returns
    for (int q = 0; q < 2; ++q)
      if (done_rec[q]) MODE_HIP_OK(hipStreamWaitEvent(hs, ev_done[q], 0));
  }
  // ---- RMSNorm gain gradients of all layers: one segmented column sum per kind when the gradient slots are layer-contiguous
  {
    const int Ly = d.L;
    const ModeLayerGrads& g0 = gr->layers[0];
    bool packed = true;
    for (int l = 1; l < Ly; ++l) {
      const ModeLayerGrads& gl = gr->layers[l];
      packed = packed && gl.ln1_g == g0.ln1_g + (long)l * D && gl.ln2_g == g0.ln2_g + (long)l * D && gl.qn_g == g0.qn_g + (long)l * hd &&
               gl.kn_g == g0.kn_g + (long)l * hd;
    }
    const int nb = packed ? 1 : Ly, lb = packed ? Ly : 1;
    for (int i = 0; i < nb; ++i) {
      const ModeLayerGrads& gi = gr->layers[i];
      if ((rc = colsum(dgp + (size_t)(1 + i) * nblk4 * D, D, lb * nblk4, D, MODE_F32, nullptr, nblk4, lb, gi.ln1_g, 0))) return rc;
      if ((rc = colsum(dgp + (size_t)(1 + Ly + i) * nblk4 * D, D, lb * nblk4, D, MODE_F32, nullptr, nblk4, lb, gi.ln2_g, 0))) return rc;
      if ((rc = colsum(apq + (size_t)i * B * D, hd, lb * B * d.H, hd, MODE_F32, nullptr, B * d.H, lb, gi.qn_g, 0))) return rc;
      if ((rc = colsum(apk + (size_t)i * B * D, hd, lb * B * d.H, hd, MODE_F32, nullptr, B * d.H, lb, gi.kn_g, 0))) return rc;
    }
  }

  // ---- router MLPs of all layers in one batch (fp32).  r_pre / dpre are [B][L][2D]; weights and gradients [L][...] contiguous.
  if (!tokr) {
    const int Ly = d.L, H2 = 2 * D;
    const ModeLayerWeights& w0 = w->layers[0];
    const ModeLayerGrads& g0 = gr->layers[0];
    const int KS = (H2 % 64 == 0) ? 4 : 1;                                        // K-slices per layer: 4x the workgroups, 4x shorter serial K loops
    float* cpart = dhid;                                                          // [KS*L][B][D] partial dcond
    int32_t* koffs = reinterpret_cast<int32_t*>(dhid + (size_t)Ly * 4 * B * D);   // [KS*L+1] K-group offsets 0, 2D/KS, ...
    // dlogits through renormalisation / clamp / softmax: all layers in one launch when the expert ids of the layers are adjacent
    const long idx_rows = a->idx_per_token ? (long)N : (long)B;
    if (a->topk_layer_stride == idx_rows * d.k) {
      if ((rc = mode_moe_router_bwd_aux(dwt, a->topk_idx, a->probs, a->shifted, a->aux_lb_coef, a->shifted ? a->aux_z_coef : nullptr, Ly * B, B, T, E, d.k,
                                        d.router_normalize, a->idx_per_token, dlog, stream))) return rc;
    } else {
      for (int l = 0; l < Ly; ++l)
        if ((rc = mode_moe_router_bwd_aux(dwt + (size_t)l * NK, a->topk_idx + (long)l * a->topk_layer_stride, a->probs + (long)l * B * E,
                                          a->shifted ? a->shifted + (long)l * B * E : nullptr, a->aux_lb_coef ? a->aux_lb_coef + (long)l * E : nullptr,
                                          a->shifted ? a->aux_z_coef : nullptr, B, B, T, E, d.k, d.router_normalize, a->idx_per_token,
                                          dlog + (long)l * B * E, stream))) return rc;
    }
    if ((rc = colsum(dlog, E, Ly * B, E, MODE_F32, nullptr, B, Ly, g0.r_b3, 0))) return rc;                    // db3 [L][E]
    if ((rc = mode_router_mlp_bwd(dlog, a->r_pre, w0.r_w3, Ly, B, E, H2, dpre, g0.r_w3, stream))) return rc;  // dpre, dW3
    if ((rc = colsum(dpre, (long)Ly * H2, B, Ly * H2, MODE_F32, nullptr, 0, 1, g0.r_b0, 0))) return rc;        // db0 [L][2D]
    // dW0 [L*2D, D] = dpre^T [L*2D x B] cond [B x D]
    ModeGemmDesc g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, Ly * H2, D, B, dpre, (long)Ly * H2, a->cond, D, g0.r_w0, D);
    g.flags = MODE_GEMM_A_KM | MODE_GEMM_W_KN;
    if ((rc = mode_gemm(&g, stream))) return rc;
    // dcond += dpre [B, L*2D] W0 [L*2D, D]: KS K-groups per layer (parallelism), partial results summed in a fixed order
    if ((rc = mode_iota_i32(koffs, KS * Ly + 1, H2 / KS, stream))) return rc;
    g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, B, D, Ly * H2, dpre, (long)Ly * H2, w0.r_w0, D, cpart, D);
    g.flags = MODE_GEMM_W_KN; g.k_group_offsets = koffs; g.num_k_groups = KS * Ly; g.c_group_stride = (long)B * D;
    if ((rc = mode_gemm(&g, stream))) return rc;
    if ((rc = colsum(cpart, (long)B * D, KS * Ly, B * D, MODE_F32, nullptr, 0, 1, dcond, 1))) return rc;
  }

  // ---- embeddings: x_0 = [emb_t | goal_e + pos0 | img_e + pos1 | act_e + pos(1..A)]
  {
    const int t0 = d.use_noise_token ? 1 : 0, t_img = t0 + 1, t_act = t_img + d.n_img;
    float* demb = (float*)(ws + W.demb); float* de1 = (float*)(ws + W.de1); float* dimg = (float*)(ws + W.dimg); float* dgoal = (float*)(ws + W.dgoal);
    // conditioning: cond = emb_t (+ goal_e): d emb_t = dcond (+ d x0[:, noise token])
    if (d.use_noise_token) { if ((rc = mode_rowcopy_f32(DXa, D, 0, T, nullptr, demb, D, 0, 1, nullptr, dcond, D, B, D, stream))) return rc; }
    else { if ((rc = mode_rowcopy_f32(dcond, D, 0, 1, nullptr, demb, D, 0, 1, nullptr, nullptr, 0, B, D, stream))) return rc; }
    // goal token (+ routing contribution when the goal embedding feeds the conditioning)
    if ((rc = mode_rowcopy_f32(DXa, D, t0, T, nullptr, dgoal, D, 0, 1, nullptr, a->goal_in_cond ? dcond : nullptr, D, B, D, stream))) return rc;
    for (int i = 0; i < d.n_img; ++i)
      if ((rc = mode_rowcopy_f32(DXa, D, t_img + i, T, nullptr, dimg, D, i, d.n_img, nullptr, nullptr, 0, B, D, stream))) return rc;
    // pos_emb: row 0 <- goal token; row 1 <- both image tokens + first action; row 1+a <- action a
    if ((rc = mode_pos_emb_bwd(DXa, B, T, D, t0, d.n_img, d.A_len, gr->pos, stream))) return rc;
    // action_emb: dW_act [D, A] = dX_act^T [D, R] (actions*c_in) [R, A]
    if ((rc = mode_transpose(DXa, D, R, D, st1, R, a->act_rows, nullptr, MODE_F32, stream))) return rc;
    if ((rc = mode_transpose(a->actions_scaled, A, R, A, st2, R, nullptr, nullptr, MODE_F32, stream))) return rc;
    ModeGemmDesc g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, D, A, R, st1, R, st2, R, gr->w_act, A);
    if ((rc = mode_gemm(&g, stream))) return rc;
    // tok_emb / goal_emb: dW = d e^T input
    const int RI = B * d.n_img;
    const bool kn_ok = d.O % 4 == 0 && d.G % 4 == 0;               // [K][M] / [K][N] operand layouts need 16-byte aligned rows
    if (kn_ok) {
      g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, D, d.O, RI, dimg, D, a->state_images, d.O, gr->w_tok, d.O);
      g.flags = MODE_GEMM_A_KM | MODE_GEMM_W_KN;
      if ((rc = mode_gemm(&g, stream))) return rc;
      g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, D, d.G, B, dgoal, D, a->goals, d.G, gr->w_goal, d.G);
      g.flags = MODE_GEMM_A_KM | MODE_GEMM_W_KN;
      if ((rc = mode_gemm(&g, stream))) return rc;
    } else {
      if ((rc = mode_transpose(dimg, D, RI, D, st1, RI, nullptr, nullptr, MODE_F32, stream))) return rc;
      if ((rc = mode_transpose(a->state_images, d.O, RI, d.O, st2, RI, nullptr, nullptr, MODE_F32, stream))) return rc;
      g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, D, d.O, RI, st1, RI, st2, RI, gr->w_tok, d.O);
      if ((rc = mode_gemm(&g, stream))) return rc;
      if ((rc = mode_transpose(dgoal, D, B, D, st1, B, nullptr, nullptr, MODE_F32, stream))) return rc;
      if ((rc = mode_transpose(a->goals, d.G, B, d.G, st2, B, nullptr, nullptr, MODE_F32, stream))) return rc;
      g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, D, d.G, B, st1, B, st2, B, gr->w_goal, d.G);
      if ((rc = mode_gemm(&g, stream))) return rc;
    }
    // input gradients (ABI 5): the upstream encoders train through these (mode_agent.py:404-411, 548-567)
    if (a->d_state_images) {
      if (d.O % 4) return MODE_ERR_UNSUPPORTED;
      g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, RI, d.O, D, dimg, D, w->w_tok, d.O, a->d_state_images, d.O);
      g.flags = MODE_GEMM_W_KN;
      if ((rc = mode_gemm(&g, stream))) return rc;
    }
    if (a->d_goals) {
      if (d.G % 4) return MODE_ERR_UNSUPPORTED;
      g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, B, d.G, D, dgoal, D, w->w_goal, d.G, a->d_goals, d.G);
      g.flags = MODE_GEMM_W_KN;
      if ((rc = mode_gemm(&g, stream))) return rc;
    }
    // sigma path: emb_t = e1 W_sl^T, e1 = s * w_se + b_se
    g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, D, D, B, demb, D, a->e1, D, gr->w_sl, D);
    g.flags = MODE_GEMM_A_KM | MODE_GEMM_W_KN;
    if ((rc = mode_gemm(&g, stream))) return rc;
    g = gdesc(MODE_F32, MODE_EPI_NONE, MODE_F32, B, D, D, demb, D, w->w_sl, D, de1, D);
    g.flags = MODE_GEMM_W_KN;
    if ((rc = mode_gemm(&g, stream))) return rc;
    if ((rc = mode_sigma_embed_bwd(de1, a->sigma, B, D, gr->w_se, gr->b_se, stream))) return rc;
  }
#undef ZERO
  return MODE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stand-alone composites named in the scope contract (SURVEY §8b "minimum exports"): the same kernels the whole-model chains use,
// callable (and testable against torch.autograd) on one MoE block / one norm.
namespace {
struct MlpWs { size_t dh, dp, csw, total; size_t csw_bytes; };
MlpWs mlp_ws(int N, int D, int E, int k, int dtype) {
  const size_t esz = dtype == MODE_BF16 ? 2 : 4, NK = (size_t)N * k;
  MlpWs w{};
  Take t;
  w.dh = t(NK * 4 * D * esz); w.dp = t(NK * 8 * D * esz);
  w.csw_bytes = mode_colsum_workspace_bytes((int)NK, 8 * D, E) + 4096;
  const size_t fb = mode_swiglu_bwd_bias_workspace_bytes((int64_t)NK, 4 * D, E);
  if (fb > w.csw_bytes) w.csw_bytes = fb;
  w.csw = t(w.csw_bytes);
  w.total = t.o;
  return w;
}
int mlp_check(const ModeGroupedMlpDesc* d) {
  if (!d || !d->x || !d->perm || !d->offsets || !d->w1 || !d->b1 || !d->w2 || !d->h) return MODE_ERR_BAD_ARG;
  if (d->N < 0 || d->D <= 0 || d->E <= 0 || d->k <= 0 || d->k > d->E) return MODE_ERR_BAD_ARG;
  if (d->dtype != MODE_BF16 && d->dtype != MODE_F32) return MODE_ERR_BAD_ARG;
  return MODE_OK;
}
}  // namespace

extern "C" size_t mode_moe_grouped_mlp_workspace_bytes(int N, int D, int E, int k, int dtype) {
  if (N < 0 || D <= 0 || E <= 0 || k <= 0) return 0;
  return mlp_ws(N, D, E, k, dtype).total;
}

extern "C" int mode_moe_grouped_mlp_fwd(const ModeGroupedMlpDesc* d, void* stream) {
  int rc = mlp_check(d);
  if (rc) return rc;
  if (!d->y) return MODE_ERR_BAD_ARG;
  const int D = d->D, NK = d->N * d->k, dt = d->dtype;
  if (NK == 0) return MODE_OK;
  ModeGemmDesc g;
  if (d->p) {                                          // training: keep the pre-activation, SwishGLU (+dropout) as its own pass
    g = gdesc(dt, MODE_EPI_BIAS, dt, NK, 8 * D, D, d->x, D, d->w1, D, d->p, 8 * D);
    g.bias = d->b1; g.bias_expert_stride = 8L * D; g.w_expert_stride = 8L * D * D; g.a_rows = d->perm; g.expert_offsets = d->offsets;
    g.num_experts = d->E;
    if ((rc = mode_gemm(&g, stream))) return rc;
    if ((rc = mode_swiglu_fwd(d->p, d->h, NK, 4 * D, dt, d->seed, d->p_drop, stream))) return rc;
  } else {                                             // inference: bias + SwishGLU fused into the GEMM epilogue
    if (d->p_drop != 0.f) return MODE_ERR_BAD_ARG;
    g = gdesc(dt, MODE_EPI_SWIGLU, dt, NK, 4 * D, D, d->x, D, d->w1, D, d->h, 4 * D);
    g.bias = d->b1; g.bias_expert_stride = 8L * D; g.w_expert_stride = 8L * D * D; g.a_rows = d->perm; g.expert_offsets = d->offsets;
    g.num_experts = d->E;
    if ((rc = mode_gemm(&g, stream))) return rc;
  }
  g = gdesc(dt, MODE_EPI_NONE, d->y_dtype, NK, D, 4 * D, d->h, 4 * D, d->w2, 4 * D, d->y, D);
  g.w_expert_stride = 4L * D * D; g.expert_offsets = d->offsets; g.num_experts = d->E;
  return mode_gemm(&g, stream);
}

extern "C" int mode_moe_grouped_mlp_bwd(const ModeGroupedMlpDesc* d, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = mlp_check(d);
  if (rc) return rc;
  if (!d->p || !d->dy || !d->dxs || !d->dw1 || !d->db1 || !d->dw2 || !workspace) return MODE_ERR_BAD_ARG;
  if (d->dtype != MODE_BF16 || d->D % 8) return MODE_ERR_UNSUPPORTED;       // fp32 parity mode goes through mode_dit_backward's transpose path
  const int D = d->D, E = d->E, NK = d->N * d->k, dt = d->dtype;
  const MlpWs W = mlp_ws(d->N, D, E, d->k, dt);
  if (workspace_bytes < W.total) return MODE_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  void* dH = ws + W.dh; void* dP = ws + W.dp;
  // down-projection: dH = dY W2 ; dW2_e = dY_e^T H_e
  ModeGemmDesc g = gdesc(dt, MODE_EPI_NONE, dt, NK, 4 * D, D, d->dy, D, d->w2, 4 * D, dH, 4 * D);
  g.w_expert_stride = 4L * D * D; g.expert_offsets = d->offsets; g.num_experts = E; g.flags = MODE_GEMM_W_KN;
  if ((rc = mode_gemm(&g, stream))) return rc;
  g = gdesc(dt, MODE_EPI_NONE, MODE_F32, D, 4 * D, NK, d->dy, D, d->h, 4 * D, d->dw2, 4 * D);
  g.k_group_offsets = d->offsets; g.num_k_groups = E; g.c_group_stride = 4L * D * D; g.flags = MODE_GEMM_W_KN | MODE_GEMM_A_KM;
  if ((rc = mode_gemm(&g, stream))) return rc;
  // SwishGLU (+dropout) backward and the bias gradient
  if ((rc = mode_swiglu_bwd_bias(d->p, dH, dP, NK, 4 * D, dt, d->seed, d->p_drop, d->offsets, E, d->db1, ws + W.csw, W.csw_bytes, stream))) return rc;
  // up-projection: dX (sorted rows) = dP W1 ; dW1_e = dP_e^T X_e (rows gathered through perm)
  g = gdesc(dt, MODE_EPI_NONE, MODE_F32, NK, D, 8 * D, dP, 8 * D, d->w1, D, d->dxs, D);
  g.w_expert_stride = 8L * D * D; g.expert_offsets = d->offsets; g.num_experts = E; g.flags = MODE_GEMM_W_KN;
  if ((rc = mode_gemm(&g, stream))) return rc;
  g = gdesc(dt, MODE_EPI_NONE, MODE_F32, 8 * D, D, NK, dP, 8 * D, d->x, D, d->dw1, D);
  g.k_group_offsets = d->offsets; g.num_k_groups = E; g.c_group_stride = 8L * D * D; g.w_rows = d->perm;
  g.flags = MODE_GEMM_W_KN | MODE_GEMM_A_KM;
  return mode_gemm(&g, stream);
}

extern "C" size_t mode_rmsnorm_cond_bwd_workspace_bytes(int rows, int D, int rows_per_cond) {
  if (rows < 0 || D <= 0) return 0;
  const size_t nblk4 = ((size_t)rows + 3) / 4;
  size_t cs = mode_colsum_workspace_bytes((int)nblk4, D, 1);
  const size_t c2 = rows_per_cond > 0 ? mode_colsum_workspace_bytes(rows, D, (rows + rows_per_cond - 1) / rows_per_cond) : 0;
  if (c2 > cs) cs = c2;
  return align_up(nblk4 * D * 4, 256) + cs + 4096;
}

extern "C" int mode_rmsnorm_cond_bwd(const float* x, const float* g, const float* dy, int rows, int D, int rows_per_cond, float eps, float* dx,
                                     float* dg, float* dcond, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !g || !dy || !dx || !dg || !workspace || rows < 0 || D <= 0) return MODE_ERR_BAD_ARG;
  if (dcond && rows_per_cond <= 0) return MODE_ERR_BAD_ARG;
  if (workspace_bytes < mode_rmsnorm_cond_bwd_workspace_bytes(rows, D, rows_per_cond)) return MODE_ERR_WORKSPACE;
  if (rows == 0) return MODE_OK;
  const int nblk4 = (rows + 3) / 4;
  float* dgp = (float*)workspace;
  char* csw = (char*)workspace + align_up((size_t)nblk4 * D * 4, 256);
  const size_t cswb = workspace_bytes - align_up((size_t)nblk4 * D * 4, 256);
  int rc = mode_rmsnorm_bwd(x, g, dy, nullptr, nullptr, nullptr, 0, rows, D, eps, dx, 0, dgp, nullptr, nullptr, MODE_F32, stream);
  if (rc) return rc;
  if ((rc = mode_colsum(dgp, D, nblk4, D, MODE_F32, nullptr, 0, 1, dg, 0, csw, cswb, stream))) return rc;
  if (dcond) {
    const int nseg = (rows + rows_per_cond - 1) / rows_per_cond;
    if ((rc = mode_colsum(dy, D, rows, D, MODE_F32, nullptr, rows_per_cond, nseg, dcond, 0, csw, cswb, stream))) return rc;
  }
  return MODE_OK;
}
