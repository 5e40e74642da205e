// C-ABI glue + the whole-denoiser launch chain (MoDeDiT.forward, modedit.py:741-821) issued from C++:
// 1 + 7*L kernels per denoise step, zero host synchronisation (the reference crosses device->host ~120 times per forward,
// SURVEY.md §3.1), hipGraph-capture safe.
#include "mode_common.h"

#include <string.h>

namespace mode {
int gemm_bf16_launch(const ModeGemmDesc* d, hipStream_t s);
int gemm_f32_launch(const ModeGemmDesc* d, hipStream_t s);
int gemm_bf16_tr_launch(const ModeGemmDesc* d, hipStream_t s);
int gemm_bf16_conv_launch(const ModeGemmDesc* d, hipStream_t s);   // conv_gemm.hip: a_rows in taps (implicit-GEMM convolution forward / data gradient)
struct MetaBatch {
  const int* idx; const float* w; long idx_bstride;
  int* counts; int* offsets; int* perm; int* pos; float* posw; int* poffsets; int* prow; long out_bstride;
};
int dispatch_meta_batched(const MetaBatch& mb, int nbatch, int R, int tpr, int N, int E, int k, hipStream_t s);

extern int g_gemm_cfg;
extern int g_gemm_pp;
extern int g_gemm_pp_min_tiles;
extern int g_gemm_pp_min_tiles_up;
extern int g_gemm_dn_ring3;
extern int g_fuse_qkv_attn, g_fuse_qkv_attn_min_b, g_qkv_attn_w3, g_qkv_attn_waves;   // qkv_attn.hip
extern int g_combine_row_max;
extern int g_gemm_mid_rows;
extern int g_gemm_mid_rows_rn;
extern int g_tr_cfg;
extern int g_bwd_coexec;
extern int g_attn_bwd_mfma;   // attn.hip
int g_train_dn_split = -1;   // "train_dn_split" option: 1 = the training forward cuts the expert down-projection into the inference chain's K-slices (bf16 slabs), 0 = one slab, -1 = by batch size (dit_train.hip: train_dn_split)
int g_fuse_swiglu_bwd = 1;   // "fuse_swiglu_bwd" option: 1 = the training backward runs dH = dY W2 and the SwishGLU backward as one launch
extern int g_conv_ns;
extern int g_gemm_group_m;
extern int g_adamw_blocks;
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct WsLayout {
  size_t x, h, qkv, y, hbuf, ybuf, rowss, meta, e1, hid, logits, tr_hid, tr_logits, tr_idx, tr_w, tr_meta, total;
};

static WsLayout ws_layout(const ModeDims& d, int B, int R, int dtype) {
  const size_t esz = dtype == MODE_BF16 ? 2 : 4;
  const size_t N = (size_t)B * d.T, NK = N * d.k, D = d.D;
  ModeMetaLayout ml;
  mode_moe_meta_layout((int)N, d.E, d.k, &ml);
  WsLayout w{};
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
  w.x = take(N * D * 4);
  w.h = take(N * D * esz);
  w.qkv = take(N * 3 * D * esz);
  w.y = take(N * D * esz);
  w.hbuf = take(NK * 4 * D * esz);
  w.ybuf = take(NK * D * 16);       // expert outputs (compute dtype): up to 8 bf16 split-K slabs of the down-projection (or one fp32 slab)
  w.rowss = take(N * ((D + 15) / 16) * 4);   // fused ln_2: per-64-column sums of squares of the residual stream (per 16 columns in the small-batch chain)
  w.meta = take((size_t)d.L * ml.total_words * 4);
  const size_t Rr = R > 0 ? R : 1;
  w.e1 = take(Rr * D * 4);
  w.hid = take(Rr * 2 * D * 4 * d.L);          // router hidden activations of all layers [R][L][2D]
  w.logits = take(Rr * d.E * 4 * d.L);
  // token routing (cond_router=False): one layer's router activations / decisions / dispatch record at a time, N routing rows
  w.tr_hid = take(N * 2 * D * 4); w.tr_logits = take(N * d.E * 4); w.tr_idx = take(NK * 4); w.tr_w = take(NK * 4); w.tr_meta = take((size_t)ml.total_words * 4);
  w.total = o;
  return w;
}

int g_fuse_ln2 = 1;     // "fuse_ln2" option: 1 = ln_2 folded into the c_proj epilogue / up-projection epilogue / combine (bf16 path), 0 = its own kernel
int g_dn_split_k = 0;   // "dn_split_k" option: K-slices of the inference-path expert down-projection (0 = default 4, 1 = off, <= 8)

// The expert down-projection [NK, 4D] x [D, 4D]^T has few output tiles (16 x 4 tiles of 224 x 256 at B=128) and a long K, so the 256 CUs are
// not covered by whole-K tiles.  K is cut into FOUR slices whose bf16 partial slabs the combine / head kernel adds in slice order: at B=128 that
// is exactly one 224x256x1024 tile per CU for the persistent ping-pong kernel (scripts/pp_probe.py: 30.2 us against 33.7 us for two slices of
// 128x128 tiles; 540 -> 549 denoise-steps/s in the chain), at B=32 twice the workgroups of two slices (19.0 vs 21.9 us).  The slice count is
// the same for EVERY batch size so that a sample's result does not depend on how many samples share its batch (bit-exact batch-slice
// consistency, tests/test_gpu_model.py::test_c2_full_size_properties).
int down_proj_split(int dt, int K) {          // also the training forward (dit_train.hip): same slices, same combine
  if (dt != MODE_BF16) return 1;
  int s = g_dn_split_k > 0 ? g_dn_split_k : 4;
  while (s > 1 && K % (64 * s)) s /= 2;
  return s;
}

static int check_dims(const ModeDims* d) {
  if (!d) return MODE_ERR_BAD_ARG;
  if (d->D <= 0 || d->H <= 0 || d->D % d->H || d->L <= 0 || d->E <= 0 || d->k <= 0 || d->k > d->E) return MODE_ERR_BAD_ARG;
  if (d->T != (d->use_noise_token ? 1 : 0) + 1 + d->n_img + d->A_len) return MODE_ERR_UNSUPPORTED;
  if (d->D % 4 || d->A_dim > 8 || d->T > 16) return MODE_ERR_UNSUPPORTED;
  return MODE_OK;
}

}  // namespace mode

using namespace mode;

extern "C" int mode_hip_version(void) { return MODE_HIP_ABI_VERSION; }

extern "C" size_t mode_hip_sizeof(const char* n) {
  if (!n) return 0;
#define MODE_SZ(T) if (!strcmp(n, #T)) return sizeof(T);
  MODE_SZ(ModeGemmDesc) MODE_SZ(ModeEmbedDesc) MODE_SZ(ModeHeadDesc) MODE_SZ(ModeGroupedMlpDesc) MODE_SZ(ModeDims) MODE_SZ(ModeLayerWeights)
  MODE_SZ(ModeModelWeights) MODE_SZ(ModeMetaLayout) MODE_SZ(ModeForwardArgs) MODE_SZ(ModeStashLayout) MODE_SZ(ModeTrainArgs) MODE_SZ(ModeLayerGrads)
  MODE_SZ(ModeModelGrads) MODE_SZ(ModeLayerWeightsT) MODE_SZ(ModeModelWeightsT) MODE_SZ(ModeBnFilmDesc) MODE_SZ(ModeQkvAttnDesc) MODE_SZ(ModeConvBnDesc)
  MODE_SZ(ModeAdamWFuse) MODE_SZ(ModeStemConvDesc)
#undef MODE_SZ
  return 0;
}

extern "C" const char* mode_hip_status_string(int status) {
  switch (status) {
    case MODE_OK: return "ok";
    case MODE_ERR_BAD_ARG: return "bad argument";
    case MODE_ERR_UNSUPPORTED: return "unsupported shape/flag combination";
    case MODE_ERR_WORKSPACE: return "workspace too small";
    default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown status";
  }
}

extern "C" __attribute__((weak)) int mode_trws_set_option(const char* key, int value);

extern "C" int mode_set_option(const char* key, int value) {
  if (!key) return MODE_ERR_BAD_ARG;
  if (!strcmp(key, "gemm_cfg")) { g_gemm_cfg = value; return MODE_OK; }
  if (!strcmp(key, "gemm_pp")) { g_gemm_pp = value != 0; return MODE_OK; }
  if (!strcmp(key, "gemm_pp_min_tiles")) { g_gemm_pp_min_tiles = value; return MODE_OK; }
  if (!strcmp(key, "gemm_pp_min_tiles_up")) { g_gemm_pp_min_tiles_up = value; return MODE_OK; }
  if (!strcmp(key, "gemm_dn_ring3")) { g_gemm_dn_ring3 = value != 0; return MODE_OK; }
  if (!strcmp(key, "gemm_tr_cfg")) { g_tr_cfg = value; return MODE_OK; }
  if (!strcmp(key, "bwd_coexec")) { g_bwd_coexec = value != 0; return MODE_OK; }
  if (!strcmp(key, "fuse_swiglu_bwd")) { g_fuse_swiglu_bwd = value != 0; return MODE_OK; }
  if (!strcmp(key, "attn_bwd_mfma")) { g_attn_bwd_mfma = value != 0; return MODE_OK; }
  if (!strcmp(key, "train_dn_split")) { g_train_dn_split = value < 0 ? -1 : (value != 0); return MODE_OK; }
  if (!strcmp(key, "conv_ns")) { if (value != 0 && value != 2 && value != 3) return MODE_ERR_BAD_ARG; g_conv_ns = value; return MODE_OK; }
  if (!strcmp(key, "gemm_group_m")) { g_gemm_group_m = value; return MODE_OK; }
  if (!strcmp(key, "adamw_blocks")) { g_adamw_blocks = value; return MODE_OK; }
  if (mode_trws_set_option) { const int rc = mode_trws_set_option(key, value); if (rc != MODE_ERR_UNSUPPORTED) return rc; }   // probe build only (scripts/probe/gemm_bf16_trws.hip)
  if (!strcmp(key, "gemm_skinny_rows")) { if (value < 0) return MODE_ERR_BAD_ARG; g_gemm_skinny_rows = value; return MODE_OK; }
  if (!strcmp(key, "fuse_ln2")) { g_fuse_ln2 = value != 0; return MODE_OK; }
  if (!strcmp(key, "gemm_mid_rows_rn")) { if (value < 0) return MODE_ERR_BAD_ARG; g_gemm_mid_rows_rn = value; return MODE_OK; }
  if (!strcmp(key, "gemm_mid_rows")) { if (value < 0) return MODE_ERR_BAD_ARG; g_gemm_mid_rows = value; return MODE_OK; }
  if (!strcmp(key, "combine_row_max")) { if (value < 0) return MODE_ERR_BAD_ARG; g_combine_row_max = value; return MODE_OK; }
  if (!strcmp(key, "fuse_qkv_attn")) { g_fuse_qkv_attn = value != 0; return MODE_OK; }
  if (!strcmp(key, "qkv_attn_waves")) { if (value != 4 && value != 8) return MODE_ERR_BAD_ARG; g_qkv_attn_waves = value; return MODE_OK; }
  if (!strcmp(key, "qkv_attn_w3")) { g_qkv_attn_w3 = value != 0; return MODE_OK; }
  if (!strcmp(key, "fuse_qkv_attn_min_b")) { if (value < 0) return MODE_ERR_BAD_ARG; g_fuse_qkv_attn_min_b = value; return MODE_OK; }
  if (!strcmp(key, "dn_split_k")) { if (value < 0 || value > 8) return MODE_ERR_BAD_ARG; g_dn_split_k = value; return MODE_OK; }
  return MODE_ERR_UNSUPPORTED;
}

extern "C" int mode_gemm(const ModeGemmDesc* d, void* stream) {
  if (!d || !d->A || !d->W || !d->C || d->M < 0 || d->N <= 0) return MODE_ERR_BAD_ARG;
  if (d->expert_offsets && d->num_experts <= 0) return MODE_ERR_BAD_ARG;
  if (d->adamw && !(d->dtype == MODE_BF16 && d->a_tap_cols <= 0 && (d->flags & MODE_GEMM_W_KN) && (d->flags & MODE_GEMM_A_KM)))
    return MODE_ERR_UNSUPPORTED;                               // the fused optimizer epilogue exists for the bf16 row-major weight gradient only
  if (d->a_tap_cols > 0) return d->dtype == MODE_BF16 ? gemm_bf16_conv_launch(d, (hipStream_t)stream) : MODE_ERR_UNSUPPORTED;
  if (d->dtype == MODE_BF16 && (d->flags & (MODE_GEMM_W_KN | MODE_GEMM_A_KM))) return gemm_bf16_tr_launch(d, (hipStream_t)stream);
  if (d->dtype == MODE_BF16) return gemm_bf16_launch(d, (hipStream_t)stream);
  if (d->dtype == MODE_F32) return gemm_f32_launch(d, (hipStream_t)stream);
  return MODE_ERR_BAD_ARG;
}

extern "C" int mode_moe_meta_layout(int N, int E, int k, ModeMetaLayout* out) {
  if (!out || N < 0 || E <= 0 || k <= 0) return MODE_ERR_BAD_ARG;
  const int NK = N * k;
  int o = 0;
  auto take = [&](int words) { int r = o; o = (o + words + 3) & ~3; return r; };
  out->counts = take(E);
  out->offsets = take(E + 1);
  out->perm = take(NK);
  out->pos = take(NK);
  out->posw = take(NK);
  out->poffsets = take(E + 1);
  out->prow = take(NK);
  out->total_words = o;
  out->padded_rows = (NK + 63) / 64 * 64 + 64 * E;
  return MODE_OK;
}

extern "C" int mode_dit_dispatch(const int32_t* topk_idx, const float* topk_w, int nbatch, int64_t idx_bstride, int R, int tokens_per_row,
                                 int N, int E, int k, int32_t* meta, void* stream) {
  if (!topk_idx || !topk_w || !meta || nbatch < 0) return MODE_ERR_BAD_ARG;
  ModeMetaLayout ml;
  int rc = mode_moe_meta_layout(N, E, k, &ml);
  if (rc) return rc;
  MetaBatch mb{topk_idx, topk_w, (long)idx_bstride, meta + ml.counts, meta + ml.offsets, meta + ml.perm, meta + ml.pos,
               reinterpret_cast<float*>(meta + ml.posw), meta + ml.poffsets, meta + ml.prow, (long)ml.total_words};
  return dispatch_meta_batched(mb, nbatch, R, tokens_per_row, N, E, k, (hipStream_t)stream);
}

extern "C" size_t mode_dit_workspace_bytes(const ModeDims* dims, int B, int R, int dtype) {
  if (check_dims(dims) != MODE_OK || B < 0) return 0;
  return ws_layout(*dims, B, R, dtype).total;
}

static ModeGemmDesc gemm_desc(int dtype, int epi, int out_dtype, int M, int N, int K, const void* A, long lda, const void* W, long ldw,
                              void* C, long ldc) {
  ModeGemmDesc g;
  memset(&g, 0, sizeof(g));
  g.dtype = dtype; g.epilogue = epi; g.out_dtype = out_dtype; g.M = M; g.N = N; g.K = K;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc;
  return g;
}

extern "C" int mode_dit_sigma_embed(const ModeDims* dims, const ModeModelWeights* w, const float* sigma, int R, float* emb_t,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if (!w || !sigma || !emb_t || !workspace || R <= 0) return MODE_ERR_BAD_ARG;
  const WsLayout L = ws_layout(*dims, 0, R, MODE_F32);
  if (workspace_bytes < L.total) return MODE_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  float* e1 = (float*)(ws + L.e1);
  rc = mode_sigma_embed(sigma, w->w_se, w->b_se, e1, R, dims->D, stream);
  if (rc) return rc;
  ModeGemmDesc g = gemm_desc(MODE_F32, MODE_EPI_NONE, MODE_F32, R, dims->D, dims->D, e1, dims->D, w->w_sl, dims->D, emb_t, dims->D);
  return mode_gemm(&g, stream);
}

extern "C" int mode_dit_embed_obs(const ModeDims* dims, const ModeModelWeights* w, const float* state_images, const float* goals, int B,
                                  float* img_e, float* goal_e, void* stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if (!w || !state_images || !goals || !img_e || !goal_e || B <= 0) return MODE_ERR_BAD_ARG;
  ModeGemmDesc g = gemm_desc(MODE_F32, MODE_EPI_NONE, MODE_F32, B * dims->n_img, dims->D, dims->O, state_images, dims->O, w->w_tok,
                             dims->O, img_e, dims->D);
  rc = mode_gemm(&g, stream);
  if (rc) return rc;
  g = gemm_desc(MODE_F32, MODE_EPI_NONE, MODE_F32, B, dims->D, dims->G, goals, dims->G, w->w_goal, dims->G, goal_e, dims->D);
  return mode_gemm(&g, stream);
}

extern "C" int mode_dit_route(const ModeDims* dims, const ModeModelWeights* w, const float* cond, int R, int32_t* topk_idx, float* topk_w,
                              float* probs, float* shifted, float* r_pre, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if (!w || !w->layers || !cond || !topk_idx || !topk_w || !workspace || R <= 0) return MODE_ERR_BAD_ARG;
  const WsLayout L = ws_layout(*dims, 0, R, MODE_F32);
  if (workspace_bytes < L.total) return MODE_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  float* hid = (float*)(ws + L.hid);                 // [R][L][2D]
  float* logits = (float*)(ws + L.logits);           // [L][R][E]
  const int D = dims->D, E = dims->E, k = dims->k, Ly = dims->L, H2 = 2 * D;
  // The router only sees the conditioning rows, so ALL layers are routed up front.  When the router weights of the layers are adjacent
  // in memory (the parameter arena lays them out as r_w0 [L,2D,D], r_b0 [L,2D], r_w3 [L,E,2D], r_b3 [L,E]) the first Linear of every
  // layer is ONE GEMM with N = L*2D, the second one launch of router_logits, the softmax/top-k one launch over L*R rows.
  bool packed = true;
  for (int l = 1; l < Ly; ++l) {
    const ModeLayerWeights& a = w->layers[l]; const ModeLayerWeights& z = w->layers[0];
    packed = packed && a.r_w0 == z.r_w0 + (long)l * H2 * D && a.r_b0 == z.r_b0 + (long)l * H2 && a.r_w3 == z.r_w3 + (long)l * E * H2 &&
             a.r_b3 == z.r_b3 + (long)l * E;
  }
  const int nb = packed ? 1 : Ly, lb = packed ? Ly : 1;      // launches x layers per launch
  for (int i = 0; i < nb; ++i) {
    const ModeLayerWeights& lw = w->layers[i];
    const long col = (long)i * H2;
    ModeGemmDesc g;
    if (r_pre) {                                              // training: keep the pre-GELU activations [R][L][2D] for the router backward
      g = gemm_desc(MODE_F32, MODE_EPI_BIAS, MODE_F32, R, lb * H2, D, cond, D, lw.r_w0, D, r_pre + col, (long)Ly * H2);
      g.bias = lw.r_b0;
      if ((rc = mode_gemm(&g, stream))) return rc;
    } else {
      g = gemm_desc(MODE_F32, MODE_EPI_BIAS_GELU, MODE_F32, R, lb * H2, D, cond, D, lw.r_w0, D, hid + col, (long)Ly * H2);
      g.bias = lw.r_b0; g.flags = MODE_GEMM_SKINNY_OK;     // R <= 16 distinct sigma rows (sampler): stream the router weights once
      if ((rc = mode_gemm(&g, stream))) return rc;
    }
  }
  if (r_pre && (rc = mode_gelu_fwd(r_pre, hid, (long)R * Ly * H2, stream))) return rc;
  for (int i = 0; i < nb; ++i) {
    const ModeLayerWeights& lw = w->layers[i];
    if ((rc = mode_router_logits(hid + (long)i * H2, (long)Ly * H2, lw.r_w3, (long)E * H2, lw.r_b3, E, lb, R, E, H2, logits + (long)i * R * E, stream)))
      return rc;
  }
  return mode_moe_route_topk_f32(logits, Ly * R, E, k, dims->router_normalize, shifted, probs, topk_idx, topk_w, stream);
}

extern "C" int mode_dit_forward(const ModeDims* dims, const ModeModelWeights* w, const ModeForwardArgs* a, void* workspace,
                                size_t workspace_bytes, void* stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if (!w || !w->layers || !a || !workspace) return MODE_ERR_BAD_ARG;
  if (!a->goal_e || !a->img_e || !a->actions || !a->cond) return MODE_ERR_BAD_ARG;
  const bool tok_route = a->meta == nullptr;                   // cond_router=False: routing on the token states, inside the chain
  if (a->B <= 0) return MODE_OK;
  const ModeDims& d = *dims;
  const int dt = a->dtype, B = a->B, T = d.T, D = d.D, N = B * T, NK = N * d.k;
  if (dt == MODE_BF16 && (D % 64 || (D / d.H) % 16 || (D / d.H) > 128)) return MODE_ERR_UNSUPPORTED;
  const WsLayout L = ws_layout(d, B, 0, dt);
  if (workspace_bytes < L.total) return MODE_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  float* x = (float*)(ws + L.x);
  void* h = ws + L.h; void* qkv = ws + L.qkv; void* yat = ws + L.y; void* hbuf = ws + L.hbuf;
  void* ybuf = ws + L.ybuf;
  float* rowss = (float*)(ws + L.rowss);
  // Small-batch chain (B <= 2 environments: N <= "gemm_skinny_rows" token rows): every GEMM of the layer is a weight stream
  // (MODE_GEMM_SMALL_ROWS: also the grouped ones, whose segments have at most N rows), the fused ln_2 works on 16-column partials.
  const bool small = !tok_route && dt == MODE_BF16 && g_gemm_cfg == 0 && g_gemm_skinny_rows > 0 && N <= g_gemm_skinny_rows && D % 128 == 0;
  const bool fuse = !tok_route && g_fuse_ln2 && dt == MODE_BF16 && D % 64 == 0;     // (token routing reads the normalised fp32 stream: ln_2 stays a kernel)
  const int ssn = small ? D / 16 : D / 64;
  const int small_flag = small ? MODE_GEMM_SMALL_ROWS : 0;
  const bool qa_fused = g_fuse_qkv_attn && !small && dt == MODE_BF16 && g_gemm_cfg == 0 && B >= g_fuse_qkv_attn_min_b;
  ModeMetaLayout ml;
  mode_moe_meta_layout(N, d.E, d.k, &ml);
  const int ysplit = down_proj_split(dt, 4 * D);
  const bool uniform = !tok_route && a->uniform_routing != 0;   // the caller's promise (ModeForwardArgs::uniform_routing): one routing row for the whole batch (the sampler)
  const int cond_rpc = T;   // one conditioning row per sample
  // cond addressing: row b at cond + b*cond_row_stride.  rmsnorm/combine kernels index cond by (row / rows_per_cond) * D, so a
  // shared row (stride 0) is expressed as rows_per_cond = N (every token maps to row 0).
  const int rpc = a->cond_row_stride == 0 ? N : cond_rpc;
  if (a->cond_row_stride != 0 && a->cond_row_stride != D) return MODE_ERR_UNSUPPORTED;

  // ---- sequence assembly + block 0's ln_1 + c
  ModeEmbedDesc e;
  memset(&e, 0, sizeof(e));
  e.B = B; e.T = T; e.D = D; e.A_len = d.A_len; e.A_dim = d.A_dim; e.n_img = d.n_img; e.use_noise_token = d.use_noise_token;
  e.emb_t = a->emb_t; e.emb_row_stride = a->emb_row_stride; e.goal_e = a->goal_e; e.img_e = a->img_e; e.actions = a->actions;
  e.c_in = a->c_in; e.c_in_stride = a->c_in_stride; e.w_act = w->w_act; e.pos = w->pos; e.g = w->layers[0].ln1_g;
  e.cond = a->cond; e.cond_row_stride = a->cond_row_stride; e.eps = d.eps; e.x = x; e.h = h; e.h_dtype = dt;
  rc = mode_embed_tokens_fwd(&e, stream);
  if (rc) return rc;

  for (int l = 0; l < d.L; ++l) {
    const ModeLayerWeights& lw = w->layers[l];
    const int32_t* meta = tok_route ? reinterpret_cast<const int32_t*>(ws + L.tr_meta) : a->meta + (long)l * a->meta_layer_stride;
    // q,k,v as ONE GEMM [N,D] x [3D,D]^T + bias   (modedit.py:108-110, 141-143)
    // ... and the attention behind it (modedit.py:125-127, 145-165).  Large batches: ONE launch per block (qkv_attn.hip, bit-identical to the two
    // kernels); shapes it does not take come back MODE_ERR_UNSUPPORTED.
    ModeGemmDesc g;
    rc = MODE_ERR_UNSUPPORTED;
    if (qa_fused) {
      ModeQkvAttnDesc qa;
      memset(&qa, 0, sizeof(qa));
      qa.dtype = dt; qa.B = B; qa.T = T; qa.H = d.H; qa.D = D; qa.h = h; qa.ldh = D; qa.wqkv = lw.wqkv; qa.ldw = D; qa.bqkv = lw.bqkv;
      qa.q_gain = lw.qn_g; qa.k_gain = lw.kn_g; qa.eps = d.eps; qa.y = yat; qa.ldy = D;
      rc = mode_qkv_attn_fwd(&qa, stream);
      if (rc && rc != MODE_ERR_UNSUPPORTED) return rc;
    }
    if (rc == MODE_ERR_UNSUPPORTED) {
      g = gemm_desc(dt, MODE_EPI_BIAS, dt, N, 3 * D, D, h, D, lw.wqkv, D, qkv, 3 * D);
      g.bias = lw.bqkv; g.flags = small_flag;
      rc = mode_gemm(&g, stream);
      if (rc) return rc;
      rc = mode_attn_block_fwd(qkv, lw.qn_g, lw.kn_g, yat, dt, B, T, d.H, D / d.H, d.eps, 0u, 0.0f, stream);
      if (rc) return rc;
    }
    // c_proj (no bias) + residual, in place on the fp32 stream   (modedit.py:111, 166, 532)
    // ln_2 (modedit.py:539) has no kernel of its own on the bf16 path: the c_proj epilogue also writes bf16(x * g) and per-64-column sums of
    // squares of x, the up-projection scales its accumulator rows by 1 / max(|x| D^-1/2, eps) — (x g / n) W^T == ((x g) W^T) / n — and the
    // combine / head kernel rebuilds the normalised fp32 residual from x, the sums and g.
    g = gemm_desc(dt, fuse ? MODE_EPI_RESIDUAL_NORM : MODE_EPI_RESIDUAL, MODE_F32, N, D, D, yat, D, lw.wo, D, x, D);
    g.resid = x; g.ldr = D; g.flags = small_flag;
    if (fuse) { g.C2 = h; g.ldc2 = D; g.gain = lw.ln2_g; g.row_ss_out = rowss; }
    rc = mode_gemm(&g, stream);
    if (rc) return rc;
    if (!fuse) {
      // x = ln_2(x): overwrites the stream (modedit.py:539); low-precision copy feeds the experts
      rc = mode_rmsnorm_cond_fwd(x, lw.ln2_g, nullptr, N, D, 1, d.eps, x, h, dt, stream);
      if (rc) return rc;
    }
    if (tok_route) {
      // router(x, None) on the ln_2-normalised token states (modedit.py:553, 322-325): Linear(D,2D) + GELU + Linear(2D,E) in fp32, softmax /
      // clamp / top-k per token, dispatch record of this layer - the same kernels the conditioning-row router runs once per sampler schedule
      float* r_hid = (float*)(ws + L.tr_hid); float* r_logits = (float*)(ws + L.tr_logits);
      int32_t* r_idx = (int32_t*)(ws + L.tr_idx); float* r_w = (float*)(ws + L.tr_w); int32_t* m = (int32_t*)(ws + L.tr_meta);
      ModeGemmDesc rg = gemm_desc(MODE_F32, MODE_EPI_BIAS_GELU, MODE_F32, N, 2 * D, D, x, D, lw.r_w0, D, r_hid, 2L * D);
      rg.bias = lw.r_b0;
      if ((rc = mode_gemm(&rg, stream))) return rc;
      if ((rc = mode_router_logits(r_hid, 2L * D, lw.r_w3, 0, lw.r_b3, 0, 1, N, d.E, 2 * D, r_logits, stream))) return rc;
      int32_t* idx_out = a->topk_idx_out ? a->topk_idx_out + (long)l * NK : r_idx;
      if ((rc = mode_moe_route_topk_f32(r_logits, N, d.E, d.k, d.router_normalize, nullptr, nullptr, idx_out, r_w, stream))) return rc;
      if ((rc = mode_moe_dispatch_meta(idx_out, r_w, N, 1, N, d.E, d.k, m + ml.counts, m + ml.offsets, m + ml.perm, m + ml.pos,
                                       reinterpret_cast<float*>(m + ml.posw), nullptr, nullptr, stream))) return rc;
    }
    // experts: gather -> grouped GEMM (SwishGLU epilogue) -> grouped GEMM   (modedit.py:561-566, 83-90, 247-255)
    g = gemm_desc(dt, MODE_EPI_SWIGLU, dt, NK, 4 * D, D, h, D, lw.w1, D, hbuf, 4 * D);
    g.bias = lw.b1; g.w_expert_stride = 8L * D * D; g.bias_expert_stride = 8L * D;
    g.a_rows = meta + ml.perm; g.expert_offsets = meta + ml.offsets; g.num_experts = d.E;
    if (fuse) { g.row_ss = rowss; g.row_ss_n = ssn; g.row_eps = d.eps; }
    g.flags = small_flag;
    if (uniform) g.flags |= MODE_GEMM_UNIFORM_GROUPS | MODE_GEMM_IDENTITY_ROWS;   // one routing row: every token goes to the same experts, segments in token order
    rc = mode_gemm(&g, stream);
    if (rc) return rc;
    g = gemm_desc(dt, MODE_EPI_NONE, dt, NK, D, 4 * D, hbuf, 4 * D, lw.w2, 4 * D, ybuf, D);   // bf16 Y like the reference's autocast Linear
    g.w_expert_stride = 4L * D * D;
    g.expert_offsets = meta + ml.offsets; g.num_experts = d.E;
    g.split_k = ysplit; g.split_stride = (long)NK * D;
    g.flags = small_flag;
    if (uniform) g.flags |= MODE_GEMM_UNIFORM_GROUPS;
    rc = mode_gemm(&g, stream);
    if (rc) return rc;
    if (l + 1 < d.L) {
      // weighted combine + residual (from the normalised stream) + next block's ln_1 + c
      rc = mode_moe_combine_norm_fused_fwd(x, fuse ? rowss : nullptr, ssn, lw.ln2_g, ybuf, dt, ysplit, (long)NK * D, meta + ml.pos,
                                           reinterpret_cast<const float*>(meta + ml.posw), N, D, d.k, w->layers[l + 1].ln1_g, a->cond, rpc, d.eps,
                                           x, h, dt, stream);
      if (rc) return rc;
    } else {
      ModeHeadDesc hd;
      memset(&hd, 0, sizeof(hd));
      hd.B = B; hd.T = T; hd.D = D; hd.A_len = d.A_len; hd.A_dim = d.A_dim; hd.k = d.k;
      hd.u = x; hd.Y = ybuf; hd.y_dtype = dt; hd.y_splits = ysplit; hd.y_split_stride = (long)NK * D; hd.pos = meta + ml.pos; hd.posw = reinterpret_cast<const float*>(meta + ml.posw);
      hd.g = w->ln_g; hd.eps = d.eps; hd.u_ss = fuse ? rowss : nullptr; hd.u_ss_n = ssn; hd.u_gain = lw.ln2_g; hd.w_out = w->w_out; hd.b_out = w->b_out;
      hd.x_a = a->actions; hd.scal = a->scal; hd.scal_stride = a->scal_stride;
      hd.F = a->F; hd.denoised = a->denoised; hd.x_next = a->x_next; hd.den_prev = a->den_prev; hd.lin = a->lin; hd.aux1 = a->aux1; hd.aux2 = a->aux2;
      rc = mode_head_ddim_fwd(&hd, stream);
      if (rc) return rc;
    }
  }
  return MODE_OK;
}
