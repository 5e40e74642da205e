/*
 * mode_hip.h — C-ABI of libmode_hip.so: the MI355X (gfx950 / CDNA4) implementation of the MoDE denoising hot path.
 *
 * The reference (intuitive-robots/MoDE_Diffusion_Policy) has NO FFI / operator-plugin boundary for this path: the seam
 * is Hydra `_target_` instantiation of Python classes.  This header therefore defines the C-ABI that
 * sits UNDER the Python mirror of those classes (the .py files of mode_diffusion_policy_amd); each entry point cites the reference
 * code it replaces (paths relative to the reference checkout).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no torch / C++ types in signatures.
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocations in practice) unless it says "host".
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - functions never allocate, never synchronise, never throw; they are hipGraph-capture safe.
 *   - return 0 on success, a negative ModeStatus for argument errors, or a positive hipError_t from the launch.
 *   - all matrices are row-major; "weight" matrices are stored exactly like torch.nn.Linear.weight: [out, in].
 */
#ifndef MODE_HIP_H
#define MODE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODE_HIP_ABI_VERSION 12

typedef enum ModeStatus {
  MODE_OK = 0,
  MODE_ERR_BAD_ARG = -1,       /* null pointer / inconsistent sizes */
  MODE_ERR_UNSUPPORTED = -2,   /* shape or flag combination the kernels do not implement */
  MODE_ERR_WORKSPACE = -3      /* workspace too small */
} ModeStatus;

typedef enum ModeDType { MODE_BF16 = 0, MODE_F32 = 1 } ModeDType;

/* GEMM epilogues (applied to acc = A @ W^T, fp32 accumulators) */
typedef enum ModeEpilogue {
  MODE_EPI_NONE = 0,      /* C = acc                                                     */
  MODE_EPI_BIAS = 1,      /* C = acc + bias[n]                (nn.Linear with bias)      */
  MODE_EPI_BIAS_GELU = 2, /* C = gelu_erf(acc + bias[n])      (router mlp.0 + GELU)      */
  MODE_EPI_RESIDUAL = 3,  /* C = acc + resid[m, n] (fp32)     (c_proj + residual)        */
  MODE_EPI_SWIGLU = 4,    /* W is [2*N, K]; C[m,n] = (acc[m,n]+b[n]) * silu(acc[m,N+n]+b[N+n])   (SwishGLU) */
  MODE_EPI_RESIDUAL_NORM = 5 /* RESIDUAL plus the first half of the NEXT RMSNorm (ln_2, modedit.py:539): v = acc + resid goes to C (fp32),
                                bf16(v * gain[n]) to C2 and sum_n v^2 over each 64-column group to row_ss_out[m][n/64]; the consumer
                                (MODE_EPI_SWIGLU with row_ss, the combine / head kernels with u_ss) applies 1/max(|v| D^-1/2, eps) per row */
} ModeEpilogue;

int mode_hip_version(void);
/* sizeof() of an ABI struct by name ("ModeGemmDesc", "ModeHeadDesc", ...), 0 for an unknown name: lets a binding verify its mirror of the
 * structs against the library it actually loaded (tests/test_boundary.py does, for every struct of this header). */
size_t mode_hip_sizeof(const char* struct_name);
const char* mode_hip_status_string(int status);
/* Tuning knobs.  PROCESS-WIDE state read by the launchers at launch time: set them before issuing work (or through
 * MODE_HIP_OPTS="key=value,..." when the Python binding loads the library), never concurrently with launches from another thread - this
 * one entry point is NOT re-entrant; every other entry point is (no other global state, no allocation, no synchronisation).  Every knob only
 * selects between implementations that produce the SAME result (geometries / fusion on-off / slice counts); the library ships no switch
 * that changes results (the round-2 timing ablations and cycle-stamp buffers of the GEMM kernels were removed from the product).
 * "gemm_cfg": bf16 forward GEMM tile geometry, 0 = auto (default), 1 = 128x128 ring-2, 4 = 128x64 ring-3, 6 = 128x128 single-buffered
 *   (3 workgroups/CU), 8 = 128x64 ring-2, 13 = 128x128 single-buffered with <= 128 VGPRs (4 workgroups/CU), 14 = 64x64 ring-3 (no SwiGLU),
 *   20 = 128x128 ring-3 (one workgroup per CU; NONE epilogue only),
 *   17 = persistent ping-pong kernel with 224-row x 256-column tiles (gemm_bf16_pp.hip; epilogues NONE / BIAS / SWIGLU).
 *   18 = the same kernel with 256-row tiles (epilogues NONE / BIAS; the heuristic takes it for RAGGED expert segments whose expected tiles fill whole
 *   rounds of the part: the training forward's up-projection).  A forced geometry that does not take a shape falls back to the heuristic's choice.
 *   Every forward geometry produces bit-identical results (k-ordered fp32 MFMA chain, explicit-fma epilogues).
 * "gemm_pp": 1 (default) = the heuristic may pick geometries 17 / 18; "gemm_pp_min_tiles": tile count from which it does (default 200);
 * "gemm_pp_min_tiles_up": the same for the SwiGLU epilogue, i.e. the expert up-projection (default 190: 10-step chunk at B = 44 / 48 10.8 / 11.3 -> 10.5 / 10.9 ms).
 * "gemm_dn_ring3": 1 (default) = a K-sliced GEMM that fills one round of the part with 128x128 tiles (160 .. 288 workgroups) takes them on a 3-slot ring
 *   (geometry 20) instead of twice as many 128x64 tiles; bit-identical.
 * "gemm_group_m": m-tiles per XCD rasterisation group of the ring kernels (0 = default).
 * "gemm_tr_cfg": backward (transpose-read) GEMM geometry, 0 = auto, 1 = 128-wide ring-2, 2 = 64-wide ring-3, 3 = 128-wide ring-3,
 *   4 = 64-wide ring-2, 5 = 128-wide single-buffered, 6 = the persistent ping-pong kernel of gemm_bf16_pptr.hip (256 x 256 tiles, one workgroup per CU) for
 *   every shape it takes, 7 = auto without it.  Auto takes it for the large expert GEMMs of the training backward unless "bwd_coexec" is 1.
 * "attn_bwd_mfma": 1 (default) = the attention backward's five small matrix products run on v_mfma_f32_16x16x4_f32 (head_dim % 16 == 0; exact fp32 like
 *   the VALU form, another summation order), 0 = the VALU form.
 * "train_dn_split": 1 = the training forward cuts the expert down-projection into the inference chain's K-slices (bf16 slabs added by the combine kernels,
 *   forward and backward; 22 MB more stash per layer at C2 / B = 128), 0 = one fp32-accumulated slab, -1 (default) = by batch size: slices from 2048 sorted rows
 *   on (the four slices then give the 256-row ping-pong kernel one tile per CU, and a training forward with dropouts off equals the inference forward bit for
 *   bit).  Round 4 measured 11.37 -> 11.33 ms per step and left it off; with the optimizer inside the backward (round 5) the forward's time is no longer
 *   hidden behind an optimizer pass: 10.35-10.52 -> 10.23-10.25 ms (round 6).
 * "fuse_swiglu_bwd": 1 (default) = mode_dit_backward runs the down-projection's data gradient and the SwishGLU (+ dropout) backward + bias-gradient sums as
 *   ONE launch (the dH tile never leaves the chip), 0 = GEMM + mode_swiglu_bwd_bias.
 * "conv_ns": LDS ring depth of the implicit-GEMM convolution kernel (csrc/conv_gemm.hip): 0 = auto (3 below two workgroups per CU), 2, 3.
 * "bwd_coexec": 1 = the caller runs other kernels beside the backward chain (FusedAdamW.step(overlap=True), ArenaGradReducer's collectives set it):
 *   the backward's large GEMMs keep the ring kernels, whose small workgroups leave CU resources to the co-running work; 0 (default) = the backward has
 *   the GPU to itself.  Same results either way (bit-identical kernels).  "adamw_blocks": workgroup cap of one AdamW launch (0 = 256, one streaming workgroup per CU).
 * "gemm_skinny_rows": bf16 GEMMs with M <= this many rows use the weight-streaming kernels, and mode_dit_forward runs a batch of at most this
 *   many TOKEN rows as the small-batch chain (MODE_GEMM_SMALL_ROWS) (default 32 = two environments, 0 = off).
 * "gemm_mid_rows": ungrouped bf16 GEMMs with K = 1024 and at most this many rows (more than "gemm_skinny_rows") keep their weights in registers
 *   and their A block in LDS - no K loop (default 128 = up to nine environments, 0 = off).
 * "gemm_mid_rows_rn": the same kernel for MODE_EPI_RESIDUAL_NORM GEMMs (the c_proj of a block: [D, D] weight) up to this many rows (default 512 = 36 environments).
 * "fuse_ln2": 1 (default) = ln_2 folded into the c_proj / up-projection / combine kernels on the bf16 path, 0 = its own kernel.
 * "dn_split_k": K-slices of the inference path's expert down-projection, 0 = default (4, for every batch size), 1 = off, <= 8.
 * "combine_row_max": token rows up to which the MoE combine runs one workgroup per row (default: always), 0 = one wave per row.
 * "fuse_qkv_attn": 1 (default) = mode_dit_forward runs the QKV projection and the attention of a block as ONE launch (mode_qkv_attn_fwd) where that
 *   kernel applies, 0 = mode_gemm + mode_attn_block_fwd.  "fuse_qkv_attn_min_b": smallest batch it is taken for (default 56); "qkv_attn_w3": 1 (default) = its weight tiles ride a three-slot LDS ring
 *   (160 KiB per workgroup), 0 = two slots; "qkv_attn_waves": 8 (default) or 4 waves per workgroup.  Same results either way.
 * Unknown keys return MODE_ERR_BAD_ARG. */
int mode_set_option(const char* key, int value);

/* Measurement aid (not on the denoising path; nothing in the reference corresponds to it): one launch of `workgroups` x 8 waves issuing
 * `iters` x 16 register-resident v_mfma_f32_16x16x32_bf16 each, operands derived from *seed (device).  bench.py times a burst of these to
 * report the bf16 MFMA rate the socket SUSTAINS at its power cap beside the datasheet peak.  *flop_per_launch (host, optional) receives the
 * FLOP count of the launch; out (device, optional) workgroups * 512 floats. */
int mode_probe_mfma_burn(const uint32_t* seed, float* out, int workgroups, int iters, double* flop_per_launch, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_gemm — C[M,N] = epilogue(A[M,K] @ W[N,K]^T), optionally grouped (MoE) and row-gathered.
 * Replaces every nn.Linear on the path (modedit.py:108-111 q/k/v/c_proj, :194-202 router, :86/:255 expert MLPs,
 * :683-686 embeddings) and the per-expert Python loop's GEMMs (modedit.py:561-566).
 *   dtype MODE_BF16: A, W bf16; MFMA 16x16x32 bf16, fp32 accumulate. Requires K % 64 == 0.
 *   dtype MODE_F32 : A, W fp32; MFMA 16x16x4 f32 (bit-exact fp32 fma chain).  Any K.
 * Grouped mode (expert_offsets != NULL): rows are sorted by expert and expert e owns the SORTED rows
 *   [expert_offsets[e], expert_offsets[e+1]) (device int32[num_experts+1], written by mode_moe_dispatch_meta); the expert id
 *   selects W + expert*w_expert_stride and bias + expert*bias_expert_stride (strides in ELEMENTS).  M is then the
 *   total number of sorted rows (N_tokens * top_k).  Workgroups derive their (expert, row range) on the device: no host sync.
 * a_rows (optional): int32[M]; logical row m of A is read from A[a_rows[m]] (MoE gather by the dispatch permutation).
 * ------------------------------------------------------------------------------------------------------------------ */
/* (ABI 11) AdamW applied in the EPILOGUE of a weight-gradient GEMM (MODE_GEMM_A_KM | MODE_GEMM_W_KN, bf16 operands, fp32 output, no split-K):
 * the gradient tile never goes to memory - every output element g = grad_scale * acc updates its parameter, its two moments and the bf16 compute
 * shadow in place, with exactly the arithmetic (and the operation order) of mode_adamw_step, so the result is bit-identical to "write dW, then
 * run the optimizer pass".  26 instead of 34 bytes of HBM traffic per parameter, and no optimizer pass left to compete with the backward for the
 * expert matrices (88 % of the denoiser's parameters).  All arenas share the gradient arena's element offsets: the GEMM's C pointer (which is NOT
 * written) locates the tile: offset = C - grad_base.  Replaces, for those tensors, torch.optim.AdamW.step as configured by
 * mode/models/mode_agent.py:365-392 (the reference neither clips nor accumulates gradients: conf/config_libero.yaml:45). */
typedef struct ModeAdamWFuse {
  const float* grad_base;                 /* first element of the gradient arena (address arithmetic only: never read, never written)   */
  float* param_base;                      /* fp32 master weights                                                                        */
  float* exp_avg_base; float* exp_avg_sq_base;
  uint16_t* lp_base;                      /* bf16 compute shadow, or NULL                                                               */
  float* ema_base; float ema_rate;        /* optional EMA of the weights: e -= ema_rate * (e - w) (mode/callbacks/ema.py:119-126); NULL = off */
  float lr, beta1, beta2, eps, weight_decay;
  int32_t step;                           /* 1-based optimizer step this update belongs to (bias correction)                            */
  float grad_scale;
  float* gsq;                             /* optional: one float per workgroup of the launch (grid order: z-major, then tile) = sum of squares of the   */
  int64_t gsq_capacity;                   /* scaled gradient over the workgroup's tile, so ||g||^2 stays observable (mode_agent.py:304-363); floats available at gsq */
  /* mode_dit_backward only (ignored by mode_gemm): run the HBM-bound weight-gradient + optimizer launches on a SECOND stream, beside the MFMA-bound
   * data-gradient chain of the same and the following blocks.  side_events = hipEvent_t[4] owned by the caller ([0] / [1]: "the data gradient that reads
   * W2 / W1 of this block has been issued" - main -> side; [2] / [3]: "both weight-gradient launches of an even / odd block are done" - side -> main, which
   * waits for them before it reuses that block parity's dY / dP buffers and once more before it returns).  NULL side_stream = everything on the one stream. */
  void* side_stream; void* const* side_events;
} ModeAdamWFuse;
typedef struct ModeGemmDesc {
  int32_t dtype;              /* ModeDType of A and W                                     */
  int32_t epilogue;           /* ModeEpilogue                                             */
  int32_t out_dtype;          /* ModeDType of C                                           */
  int32_t M, N, K;            /* N = number of OUTPUT columns                             */
  const void* A;  int64_t lda;
  const void* W;  int64_t ldw;  int64_t w_expert_stride;
  const float* bias;          int64_t bias_expert_stride;
  const float* resid;         int64_t ldr;
  void* C;        int64_t ldc;
  const int32_t* a_rows;          /* optional gather                                      */
  const int32_t* expert_offsets;  /* optional grouped mode: device int32[num_experts + 1]  */
  int32_t num_experts;
  int32_t split_k;                /* bf16 only, epilogue NONE: K is cut into split_k slices; slice z writes its partial sums to   */
  int64_t split_stride;           /* C + z*split_stride (elements).  0/1 = no split.  The consumer adds the slabs in slice order. */
  const int32_t* k_group_offsets; /* optional K-group mode (per-expert weight gradients dW_e = dY_e^T X_e): device int32[num_k_groups+1],  */
  int32_t num_k_groups;           /* group z contracts over K in [off[z], off[z+1]) (multiples of 64 for bf16) and writes                 */
  int64_t c_group_stride;         /* C + z*c_group_stride (elements); an empty range yields an all-zero C_z                                */
  int32_t flags;                  /* MODE_GEMM_SKINNY_OK: fp32, M <= 16 may use the weight-streaming GEMV kernel (wave-tree reduction
                                     instead of the MFMA k-ordered chain: same fp32 accuracy, different rounding)                   */
  const int32_t* w_rows;          /* MODE_GEMM_A_KM only: optional gather of W's K rows (row r of the reduction reads W[w_rows[r]]; ABI 10: a negative index reads a zero row) */
  /* fused ln_2 (bf16 only).  Producer, MODE_EPI_RESIDUAL_NORM (N % 64 == 0): */
  void* C2; int64_t ldc2;         /* bf16 [M, N]: (acc + resid) * gain[n] — the un-normalised, gain-scaled input of the next GEMM       */
  const float* gain;              /* fp32 [N]: the RMSNorm gain                                                                          */
  float* row_ss_out;              /* fp32 [M, N/64]: per-row sums of squares of (acc + resid), one per 64 output columns (fixed order)     */
  /* Consumer, MODE_EPI_SWIGLU: when row_ss != NULL every accumulator row m (token a_rows[m], or m) is multiplied by                      */
  const float* row_ss;            /* 1 / max(sqrt(sum_j row_ss[token][j]) * K^-1/2, row_eps) before bias and activation, i.e. the GEMM   */
  int32_t row_ss_n;               /* reads A = x*gain and produces RMSNorm(x) @ W^T (K = the normalised width D)                          */
  float row_eps;
  /* (ABI 10) w_rows in TAPS - the weight gradient of a k x k convolution as ONE product over its k*k filter taps (MODE_GEMM_A_KM | W_KN, bf16):  */
  int32_t w_tap_cols;             /* > 0: the N output columns are N / w_tap_cols taps of w_tap_cols columns each; tap t reads columns              */
  int64_t w_rows_tap_stride;      /* [0, w_tap_cols) of W through the index table w_rows + t * w_rows_tap_stride (elements).  w_tap_cols % 64 == 0,  */
                                  /* N % w_tap_cols == 0.  0 = one table for all columns.  (perceptual_encoders.py: dW[Cout][tap][Cin] of a 3 x 3 conv) */
  /* (ABI 10) a_rows in TAPS - implicit-GEMM convolution, forward and data gradient (bf16 in / out, epilogue NONE; csrc/conv_gemm.hip):             */
  int32_t a_tap_cols;             /* > 0: K = taps * a_tap_cols; the reduction index k = t * a_tap_cols + c reads A[a_rows[t * a_rows_tap_stride + m]][c] */
  int64_t a_rows_tap_stride;      /* (a negative index = a zero row: the tap falls outside the image).  W: forward layout [N][K] K-contiguous (a          */
                                  /* channels_last conv weight [Cout][taps][Cin]); with MODE_GEMM_W_KN element (k, n) = W[c * ldw + t * N + n] - the SAME     */
                                  /* weight memory read for the data gradient (ldw = taps * Cin, N = Cin).  a_tap_cols % 64 == 0.                            */
  const ModeAdamWFuse* adamw;     /* (ABI 11) non-NULL: weight-gradient GEMM with the AdamW update in its epilogue (see ModeAdamWFuse); C only locates the tile. */
                                  /* MODE_ERR_UNSUPPORTED unless bf16 operands, MODE_GEMM_A_KM | W_KN, fp32 out, epilogue NONE, N % 128 == 0, ldc == N row pitch of the parameter */
} ModeGemmDesc;
#define MODE_GEMM_SKINNY_OK 1
/* Backward-pass operand layouts (bf16, epilogue NONE; replace autograd's mm_backward for nn.Linear, i.e. the `grad @ W` and
 * `grad.T @ x` GEMMs PyTorch launches for every Linear of modedit.py in loss.backward()).  No transposed copies are made: the row-major
 * tiles are gathered into MFMA fragments by the LDS transpose read ds_read_b64_tr_b16.
 *   MODE_GEMM_W_KN            W is [K, N] row-major (ldw = row stride): C[M,N] = A[M,K] @ W   — data gradient dX = dY @ W_linear.
 *                             K % 64 == 0; expert_offsets / w_expert_stride group the rows as in the forward; split_k K-slices
 *                             (K % (64*split_k) == 0) write partial slabs split_stride elements apart that the consumer adds.
 *   MODE_GEMM_W_KN | A_KM     A is [K, M] row-major too: C[M,N] = A^T @ W — weight gradient dW = dY^T @ X from row-major activations.
 *                             k_group_offsets are then ARBITRARY row ranges [off[z], off[z+1]) (per-expert segments, no padding), K is
 *                             the row count when no groups are given; M % 8 == 0. */
#define MODE_GEMM_W_KN 2
#define MODE_GEMM_A_KM 4
/*   MODE_GEMM_UNIFORM_GROUPS  hint (bf16 forward layout, grouped): the expert segments are whole multiples of large tiles - the uniform-sigma
 *                             sampler, where every sample routes alike.  Lets the heuristic pick the persistent 224x256 ping-pong kernel, whose
 *                             fixed tiles-per-workgroup split loses to the 128x128 family when ragged segments add partial tiles (measured at
 *                             four ragged experts: 81 vs 74 us).  Never changes results (all forward kernels are bit-identical). */
#define MODE_GEMM_UNIFORM_GROUPS 8
/*   MODE_GEMM_SMALL_ROWS      bf16 forward layout: the caller vouches that no group (expert segment; the whole problem when ungrouped) has more
 *                             than "gemm_skinny_rows" rows - the small-batch chain (B <= 2 environments).  The GEMM then always takes the
 *                             weight-streaming kernel, and the fused ln_2 works on 16-column partials: MODE_EPI_RESIDUAL_NORM (N % 16 == 0)
 *                             writes row_ss_out as [M, N/16], MODE_EPI_SWIGLU accepts any row_ss_n.  MODE_ERR_UNSUPPORTED when the streamer is
 *                             switched off ("gemm_skinny_rows" = 0 / a forced "gemm_cfg") or does not take the shape. */
#define MODE_GEMM_SMALL_ROWS 16
/*   MODE_GEMM_IDENTITY_ROWS   grouped + gathered, a PROMISE (unlike the hints above it changes what is read): inside every group the gather is
 *                             the ascending identity, a_rows[expert_offsets[e] + i] == i - every token is routed to every active expert, in
 *                             token order, which is what the dispatch permutation of a uniform-sigma sampler step looks like.  Kernels may then
 *                             compute the row index instead of loading it (one dependent memory round trip less at the start of a workgroup). */
#define MODE_GEMM_IDENTITY_ROWS 32
int mode_gemm(const ModeGemmDesc* desc, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_conv_bn_act_fwd (ABI 10) - INFERENCE: convolution + eval-mode BatchNorm + FiLM + residual + ReLU as ONE launch (csrc/conv_gemm.hip).
 * Replaces, in the perceptual encoders' eval path, F.conv2d followed by the fused BatchNorm pass - i.e. conv -> bn -> (FiLM) -> (+ identity) -> relu of
 * a ResNet block (mode/models/perceptual_encoders/pretrained_resnets.py:52-64, resnets.py:58-76): the normalised activation is computed from the fp32
 * accumulators and written once.  channels_last bf16: x [rows, Cin], w [Cout][taps][Cin] (a channels_last conv weight), y [M, Cout].
 *   idx            int32 [taps][M] (tap stride idx_tap_stride): input row that tap t pairs with output row m, -1 = outside the image; NULL = 1 x 1 / stride 1
 *   y = post(relu(pre(acc * scale + shift) + residual)),  scale = bn_weight / sqrt(bn_var + bn_eps), shift = bn_bias - bn_mean * scale (bn_mean NULL: 1, 0),
 *   pre: gamma * v + beta, post: (1 + gamma) * v + beta with gamma / beta fp32 [samples][Cout], sample = m / rows_per_sample.  Cin % 64 == 0, Cout % 8 == 0.
 * With every epilogue term absent it is the plain convolution; the training forward uses it that way with stat_sum / stat_sq.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ModeConvBnDesc {
  const void* x; int64_t ldx;
  const int32_t* idx; int64_t idx_tap_stride; int32_t taps;
  const void* w; int64_t ldw;
  void* y; int64_t ldy;
  int32_t M, Cin, Cout;
  const float* bn_mean; const float* bn_var; const float* bn_weight; const float* bn_bias; float bn_eps;
  const void* residual; int64_t ldr;
  int32_t relu;
  const float* pre_gamma; const float* pre_beta; const float* post_gamma; const float* post_beta; int32_t rows_per_sample;
  float* stat_sum; float* stat_sq;   /* optional, TRAINING forward (no epilogue terms): fp32 [ceil(M / 128)][Cout] column sums / sums of squares of y as stored, one row
                                        per 128-row tile - the partial statistics mode_bn_prepare_partials folds for the BatchNorm that follows (one pass over y less) */
} ModeConvBnDesc;
int mode_conv_bn_act_fwd(const ModeConvBnDesc* desc, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_rmsnorm_cond_fwd — y = x / max(||x||_2 * D^-1/2, eps) * g  (+ cond[row / rows_per_cond])
 * Replaces RMSNorm (modedit.py:72-80) and the additive conditioning `ln_1(x) + c` (modedit.py:532); with cond == NULL
 * it is the plain ln_2 / final ln (modedit.py:539, 818).  x fp32 [rows, D]; writes any of y_f32 / y_lp (may be NULL).
 * ------------------------------------------------------------------------------------------------------------------ */
int mode_rmsnorm_cond_fwd(const float* x, const float* g, const float* cond, int rows, int D, int rows_per_cond,
                          float eps, float* y_f32, void* y_lp, int lp_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_attn_block_fwd — per (sample, head): qk-RMSNorm over head_dim (learned gains, eps), causal softmax(QK^T/sqrt(hd)) V.
 * Replaces Attention.forward minus the four Linears (modedit.py:125-127, 145-165; SDPA is_causal=True at :149).
 * qkv is the packed [B*T, 3*D] output of the fused QKV GEMM ([q | k | v] along columns); y is [B*T, D] (heads merged).
 * T <= 16 (the path's sequence is 14 tokens), head_dim % 16 == 0 and <= 128 for bf16.  p_drop > 0 (training only) drops
 * probabilities after the softmax exactly where SDPA does.
 * ------------------------------------------------------------------------------------------------------------------ */
int mode_attn_block_fwd(const void* qkv, const float* q_gain, const float* k_gain, void* y, int dtype,
                        int B, int T, int H, int head_dim, float eps, uint32_t seed, float p_drop, void* stream);
/* ------------------------------------------------------------------------------------------------------------------
 * mode_qkv_attn_fwd (ABI 9) — Attention.forward up to c_proj as ONE launch (modedit.py:108-110, 125-127, 141-165): q, k, v = Linear(h) with the
 * packed [3D, D] weight / [3D] bias ([q | k | v] rows), qk-RMSNorm, causal softmax(QK^T / sqrt(hd)) V, heads merged into y [B*T, D].  A workgroup owns
 * (a group of whole samples, one head): the 3 * head_dim weight rows of that head are everything its attention needs, so q | k | v never go to HBM.
 * Results are BIT-IDENTICAL to mode_gemm(MODE_EPI_BIAS, out bf16) followed by mode_attn_block_fwd (same accumulation order, same rounding points, one
 * shared attention body).  bf16 only, inference only (no dropout, nothing stashed); head_dim == 128, T <= 16, D % 64 == 0, 16-byte aligned operands -
 * anything else returns MODE_ERR_UNSUPPORTED and the caller runs the two kernels.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ModeQkvAttnDesc {
  int32_t dtype;                 /* MODE_BF16 */
  int32_t B, T, H, D;            /* samples, tokens per sample, heads, model width (head_dim = D / H) */
  const void* h; int64_t ldh;    /* [B*T, D] bf16: ln_1(x) + c */
  const void* wqkv; int64_t ldw; /* [3D, D] bf16 */
  const float* bqkv;             /* [3D] */
  const float* q_gain; const float* k_gain;   /* [head_dim] */
  float eps;
  void* y; int64_t ldy;          /* [B*T, D] bf16 */
} ModeQkvAttnDesc;
int mode_qkv_attn_fwd(const ModeQkvAttnDesc* desc, void* stream);

/* backward of the above (training): dy [B*T, D] -> dqkv [B*T, 3D]; dgq_partial / dgk_partial [B*H, head_dim] are per-(sample, head)
 * partial gradients of the qk-norm gains (reduce with mode_colsum).  Attention dropout (SDPA dropout_p, modedit.py:149) is a
 * counter-based hash mask keyed by (seed, sample, head, query, key), regenerated here. */
int mode_attn_block_bwd(const void* qkv, const float* q_gain, const float* k_gain, const void* dy, void* dqkv,
                        float* dgq_partial, float* dgk_partial, int dtype, int B, int T, int H, int head_dim, float eps,
                        uint32_t seed, float p_drop, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_sigma_embed — e1[r, :] = (ln(sigma[r]) / 4) * w[:, 0] + b        (modedit.py:823-828, Linear(1, D))
 * (the following Linear(D, D) `sigma_linear` is a mode_gemm in fp32).
 * ------------------------------------------------------------------------------------------------------------------ */
int mode_sigma_embed(const float* sigma, const float* w, const float* b, float* e1, int R, int D, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_moe_route_topk_f32 — router tail on R distinct conditioning rows, fp32 end to end (bit-exact integers):
 *   shifted = logits - rowmax ; probs = clamp(softmax(shifted), 1e-9, 1-1e-9) ; top-k by prob (descending, ties ->
 *   lower expert id) ; w = probs[topk] (/ sum when normalize).      (modedit.py:345-349, 392, 398-399, 418-419)
 * logits [R, E] fp32 in; outputs (any may be NULL): shifted [R,E], probs [R,E], topk_idx int32 [R,k], topk_w [R,k].
 * ------------------------------------------------------------------------------------------------------------------ */
int mode_moe_route_topk_f32(const float* logits, int R, int E, int k, int normalize, float* shifted, float* probs,
                            int32_t* topk_idx, float* topk_w, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_moe_dispatch_meta — token-dispatch metadata for one layer; replaces the boolean-mask gather / index_put loop
 * (modedit.py:555-572) with a canonical permutation: experts ascending, token ids ascending within an expert.
 * Inputs: idx int32 [R, k] and w fp32 [R, k] on R distinct routing rows; token n uses row n / tokens_per_row
 *         (tokens_per_row = T for the noise-conditioned router, 1 for per-token routing such as training multinomial).
 * Outputs: counts int32 [E]; offsets int32 [E+1]; perm int32 [N*k] sorted row -> token id;
 *          pos int32 [N*k]: (token, j) -> sorted row, j enumerating the token's experts in ASCENDING expert id;
 *          posw fp32 [N*k]: combine weight for (token, j);
 *          (optional, both or neither) poffsets int32 [E+1]: expert starts when every expert's rows are padded to a multiple of 64,
 *          prow int32 [N*k]: sorted row -> padded position  (layout of the transposed operands of the weight-gradient GEMMs).
 * ------------------------------------------------------------------------------------------------------------------ */
int mode_moe_dispatch_meta(const int32_t* idx, const float* w, int R, int tokens_per_row, int N, int E, int k,
                           int32_t* counts, int32_t* offsets, int32_t* perm, int32_t* pos, float* posw,
                           int32_t* poffsets, int32_t* prow, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_moe_combine_norm_fwd — x_next[t] = u[t] + sum_j posw[t,j] * Y[pos[t,j]]  (ascending expert order; the residual is
 * the NORMALISED stream u, modedit.py:539/595), then optionally the next block's  h = RMSNorm(x_next; g) + cond
 * (modedit.py:532).  u fp32 [N,D]; Y [N*k, D] (y_dtype), optionally y_splits split-K slabs y_split_stride elements apart that are
 * added in slice order; writes x_next fp32 (may alias u) and h (lp dtype; may be NULL).
 * ------------------------------------------------------------------------------------------------------------------ */
int mode_moe_combine_norm_fwd(const float* u, const void* Y, int y_dtype, int y_splits, int64_t y_split_stride,
                              const int32_t* pos, const float* posw, int N, int D, int k, const float* g, const float* cond,
                              int rows_per_cond, float eps, float* x_next, void* h, int h_dtype, void* stream);
/* Same with ln_2 fused in: u is the UN-normalised residual stream written by a MODE_EPI_RESIDUAL_NORM GEMM and u_ss [N, u_ss_n] its
 * per-64-column sums of squares; the kernel uses u / max(sqrt(sum u_ss[row]) * D^-1/2, eps) * u_gain in place of u (modedit.py:539 moved
 * out of a kernel of its own).  u_ss == NULL: identical to mode_moe_combine_norm_fwd. */
int mode_moe_combine_norm_fused_fwd(const float* u, const float* u_ss, int u_ss_n, const float* u_gain, const void* Y, int y_dtype,
                                    int y_splits, int64_t y_split_stride, const int32_t* pos, const float* posw, int N, int D, int k,
                                    const float* g, const float* cond, int rows_per_cond, float eps, float* x_next, void* h, int h_dtype,
                                    void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_embed_tokens_fwd — builds the input sequence and the first block's conditioned norm in one pass:
 *   [ emb_t | goal_e + pos[0] | img_e[0..n_img) + pos[1] | (actions * c_in) @ Wact^T + pos[1..A] ]
 * (modedit.py:760-790, 847-860; c_in from score_wrappers.py:42).  emb_t / goal_e / img_e are fp32 outputs of mode_gemm.
 *   emb_row_stride: 0 when one sigma row is shared by the whole batch (sampler), D otherwise; same for cond.
 * Writes x fp32 [B*T, D] and h = RMSNorm(x; g)+cond in h_dtype.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ModeEmbedDesc {
  int32_t B, T, D, A_len, A_dim, n_img, use_noise_token;
  const float* emb_t;   int64_t emb_row_stride;
  const float* goal_e;  /* [B, D]         */
  const float* img_e;   /* [B, n_img, D]  */
  const float* actions; /* [B, A_len, A_dim] */
  const float* c_in;    /* [B] or NULL (=1); c_in_stride 0 => shared scalar */
  int64_t c_in_stride;
  const float* w_act;   /* [D, A_dim]     */
  const float* pos;     /* [1 + A_len, D] */
  const float* g;       /* ln_1 gain of block 0 */
  const float* cond;    int64_t cond_row_stride;
  float eps;
  float* x; void* h; int32_t h_dtype;
} ModeEmbedDesc;
int mode_embed_tokens_fwd(const ModeEmbedDesc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * mode_head_ddim_fwd — last block's MoE combine, final RMSNorm, output head and the EDM / DDIM epilogue on the action
 * tokens only:  F = Linear(D, A)(RMSNorm(u + sum_j w Y))  (modedit.py:807-808, 818);
 *   denoised = F * c_out + x_a * c_skip (score_wrappers.py:79-80);  x_next = r * x_a + (1 - r) * denoised,
 *   r = sigma_next / sigma (gc_sampling.py:948-950).
 * scal: fp32 [B or 1][4] = {c_skip, c_out, r, c (multistep weight, see den_prev; 0 for DDIM)}; scal == NULL => write F only (training / raw forward).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ModeHeadDesc {
  int32_t B, T, D, A_len, A_dim, k;
  const float* u; const void* Y; int32_t y_dtype; int32_t y_splits; int64_t y_split_stride;
  const int32_t* pos; const float* posw;
  const float* g; float eps;
  const float* w_out; const float* b_out;     /* [A_dim, D], [A_dim] */
  const float* x_a;                            /* current noisy actions [B, A_len, A_dim] (un-scaled) or NULL */
  const float* scal; int64_t scal_stride;      /* 0 => shared by the batch */
  float* F;                                    /* [B, A_len, A_dim] raw network output (may be NULL) */
  float* denoised;                             /* may be NULL */
  float* x_next;                               /* may be NULL; may alias x_a */
  const float* u_ss; int32_t u_ss_n; const float* u_gain;   /* fused ln_2: u is un-normalised, see mode_moe_combine_norm_fused_fwd (NULL = u as is) */
  /* ABI 12.  Two-point multistep update (DPM-Solver++(2M), gc_sampling.py:700-734): with den_prev != NULL and c = scal[3] != 0 the update uses
   * denoised_d = (1 + c) * denoised - c * den_prev instead of denoised, c = 1 / (2 r), r = h_last / h; x_next = r_sigma * x_a + (1 - r_sigma) * denoised_d is
   * the same exponential-integrator step as DDIM's.  den_prev = the `denoised` output of the previous step (the caller ping-pongs two buffers).
   * NULL (or c == 0: the first step, the step to sigma = 0) = the DDIM update, bit for bit. */
  const float* den_prev;
  /* ABI 12.  General linear update for the two-stage solvers (Heun, DPM-Solver-2, DPM-Solver++(2S): gc_sampling.py:257-373, 956-994), whose every stage is a
   * linear combination of the stage's input, its denoised prediction and at most two earlier tensors:
   *   x_next = lin[0] * x_a + lin[1] * denoised + lin[2] * aux1 + lin[3] * aux2        (lin: fp32 [4], shared by the batch; aux1 / aux2 may be NULL)
   * lin != NULL replaces the DDIM / multistep update above (scal still supplies c_skip / c_out).  x_next may alias x_a, aux1 or aux2 (element-wise). */
  const float* lin; const float* aux1; const float* aux2;
} ModeHeadDesc;
int mode_head_ddim_fwd(const ModeHeadDesc* d, void* stream);

/* mode_ddim_edm_step — the same elementwise epilogue standalone (used by the generic, un-fused sampler path). */
int mode_ddim_edm_step(const float* F, const float* x_a, const float* scal, int64_t scal_stride, int B, int per_sample,
                       float* denoised, float* x_next, void* stream);

/* ==================================================================================================================
 * Training-side operators (score-matching step: SURVEY.md §8 rows 13-17).  Deterministic: no floating-point atomics.
 * ================================================================================================================== */

/* mode_transpose — dst[c, dcol(r)] = src[srow(r), c]: feeds the weight-gradient GEMMs (dW = dY^T X, autograd of nn.Linear) with
 * K-contiguous operands.  src_rows (optional) gathers rows (MoE permutation); dst_cols (optional) places row r at an arbitrary
 * destination column (per-expert 64-padded positions from mode_moe_dispatch_meta's `prow`); untouched columns keep their content
 * (zero the buffer first for padding). */
int mode_transpose(const void* src, int64_t ld_src, int rows, int cols, void* dst, int64_t ld_dst, const int32_t* src_rows,
                   const int32_t* dst_cols, int dtype, void* stream);

/* mode_colsum — out[s, c] (+)= sum_{r in segment s} X[r, c]  (bias / RMSNorm-gain / conditioning gradients).
 * Segments: seg_offsets (device int32[nseg+1]) or uniform seg_len or the whole matrix (nseg = 1).  Two deterministic stages. */
size_t mode_colsum_workspace_bytes(int rows, int cols, int nseg);
int mode_colsum(const void* X, int64_t ld, int rows, int cols, int dtype, const int32_t* seg_offsets, int seg_len, int nseg,
                float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* mode_swiglu_fwd/_bwd — H = value * silu(gate) * dropout  on P = [value | gate] (SwishGLU + nn.Dropout, modedit.py:83-90, 254).
 * The keep-mask is hash(seed, element index): the backward regenerates it.  p_drop = 0 disables dropout. */
int mode_swiglu_fwd(const void* P, void* Hd, int64_t rows, int Hdim, int dtype, uint32_t seed, float p_drop, void* stream);
int mode_swiglu_bwd(const void* P, const void* dHd, void* dP, int64_t rows, int Hdim, int dtype, uint32_t seed, float p_drop,
                    void* stream);

/* mode_swiglu_bwd_bias — mode_swiglu_bwd fused with the per-expert bias gradient of the up-projection (bf16 only): db[e, 0:2*Hdim] =
 * column sums of dP over expert e's sorted rows [expert_offsets[e], expert_offsets[e+1]) (autograd's `grad.sum(0)` of
 * FusedMLPV2's first Linear, modedit.py:52-56), accumulated in registers while dP is produced; db is overwritten. */
size_t mode_swiglu_bwd_bias_workspace_bytes(int64_t rows, int Hdim, int E);
int mode_swiglu_bwd_bias(const void* P, const void* dHd, void* dP, int64_t rows, int Hdim, int dtype, uint32_t seed, float p_drop,
                         const int32_t* expert_offsets, int E, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* mode_rmsnorm_bwd — backward of RMSNorm (modedit.py:72-80).  dy[row] = dy_a[row] + dy_b[row] + sum_j G[pos[row*k + j]] (any may be
 * NULL / k = 0; the gather-sum is the MoE dispatch backward); dx (+)= d/dx; dg_partial [ceil(rows/4), D] per-workgroup partial gain
 * gradients (reduce with mode_colsum); dy_out (optional) receives the assembled dy (the conditioning gradient needs it). */
int mode_rmsnorm_bwd(const float* x, const float* g, const float* dy_a, const float* dy_b, const float* G, const int32_t* pos, int k,
                     int rows, int D, float eps, float* dx, int accumulate, float* dg_partial, float* dy_out, void* dx_lp,
                     int lp_dtype, void* stream);   /* dx_lp (optional): compute-dtype copy of the final dx (next GEMM's operand) */

/* mode_moe_combine_bwd — backward of next[t] += w[t,e] * expert_e(u[t]) (modedit.py:566): dYs[pos[t,j]] = posw[t,j] * dy[t] (sorted
 * rows, compute dtype) and dw[t,j] = <dy[t], Y[pos[t,j]]> (router-weight gradient, SURVEY §8 a-bis). */
int mode_moe_combine_bwd(const float* dy, const void* Y, int y_dtype, const int32_t* pos, const float* posw, int N, int D, int k,
                         void* dYs, float* dw, void* stream);

/* Small fp32 helpers of the backward chain. */
int mode_rowcopy_f32(const float* src, int64_t ld_src, int s0, int sstride, const int32_t* sidx, float* dst, int64_t ld_dst, int d0,
                     int dstride, const int32_t* didx, const float* add, int64_t ld_add, int n, int D, void* stream);
int mode_gelu_fwd(const float* pre, float* out, int64_t n, void* stream);                       /* router GELU (modedit.py:198)      */
int mode_gelu_bwd(const float* pre, const float* dout, float* dpre, int64_t n, void* stream);
/* Router backward on distinct conditioning rows: dw [B*T, k] (slot j = j-th expert in ASCENDING id, as mode_moe_combine_bwd writes it),
 * idx [B or B*T, k] top-k ids, probs [B, E] -> dlogits [B, E]; through renormalisation, clamp and softmax (SURVEY §8 a-bis). */
int mode_moe_router_bwd(const float* dw, const int32_t* idx, const float* probs, int B, int T, int E, int k, int normalize,
                        int idx_per_token, float* dlogits, void* stream);
/* Same, plus the gradients of the two auxiliary router losses of the reference (MoDeDiT.load_balancing_loss modedit.py:898-928 with the
 * per-block term of :586-593, and compute_router_z_loss :930-969; composed into the training loss at mode_agent.py:413-419).  The B rows
 * are `B / rows_per_layer` layers of `rows_per_layer` conditioning rows each.
 *   lb_coef [layers, E] (device, NULL = off): d(loss)/d(router_probs[n, e]) for every token n that selected expert e - the load-balancing
 *     term is linear in the (renormalised) combine weights: coefficient = dLB * E * f_e / (layers * N), f_e = fraction of tokens on expert e;
 *     it is added to dw before the renormalisation / clamp / softmax backward.
 *   shifted [B, E] max-shifted logits + z_coef (device scalar, NULL = off) = dZ * 2 / (layers * rows_per_layer):
 *     dlogits += z_coef * z * exp(l) / (sum exp(l) + 1e-6), z = log(sum exp(l) + 1e-6), pushed through the max-shift (the arg-max column
 *     receives minus the row sum, like autograd through `logits - logits.max()`). */
int mode_moe_router_bwd_aux(const float* dw, const int32_t* idx, const float* probs, const float* shifted, const float* lb_coef,
                            const float* z_coef, int B, int rows_per_layer, int T, int E, int k, int normalize, int idx_per_token,
                            float* dlogits, void* stream);
/* EDM preconditioning of the score-matching loss (GCDenoiser.loss, score_wrappers.py:45-63) around the training chain, one launch each:
 *   mode_edm_noise_scale : x_scaled = (action + noise * sigma_b) * c_in(sigma_b)                       [B, n] (n = A_len * A_dim)
 *   mode_edm_loss        : target = (action - c_skip * noised) / c_out; loss = mean((F - target)^2) (fixed-order reduction, one workgroup);
 *                          dF = 2 (F - target) / (B n)  (gradient of the loss for a unit upstream gradient) */
int mode_edm_noise_scale(const float* action, const float* noise, const float* sigma, float sigma_data, int B, int n, float* x_scaled, void* stream);
int mode_edm_loss(const float* F, const float* action, const float* noise, const float* sigma, float sigma_data, int B, int n, float* loss,
                  float* dF, void* stream);
/* pos_emb gradient (modedit.py:760-790): dx0 [B, T, D] gradient of the embedded token sequence -> dpos [1 + A_len, D]; row 0 <- goal token,
 * row 1 <- the n_img image tokens + first action token, row 1+a <- action token a.  t0 = 1 when the sigma token is part of the sequence. */
int mode_pos_emb_bwd(const float* dx0, int B, int T, int D, int t0, int n_img, int A_len, float* dpos, void* stream);
int mode_sigma_embed_bwd(const float* de1, const float* sigma, int B, int D, float* dw, float* db, void* stream);
/* ------------------------------------------------------------------------------------------------------------------
 * mode_moe_grouped_mlp_fwd / _bwd — the expert MLP of one MoE block on the SORTED dispatch order (SURVEY §8b minimum exports).
 * Replaces NoiseBlockMoE's per-expert Python loop `for idx in range(E): x[token_indices] -> expert(...)` (modedit.py:557-566) with
 * FusedMLPV2 = Linear(D,8D) -> SwishGLU -> Dropout -> Linear(4D,D) (modedit.py:21-60, 83-90), and autograd's backward of it.
 *   fwd:  y[s] = W2_e ( swiglu(W1_e x[perm[s]] + b1_e) * dropmask ),  s in expert e's segment [offsets[e], offsets[e+1])
 *         p (optional, training) receives the pre-activation [NK, 8D] = [value | gate]; h the post-SwishGLU(+dropout) [NK, 4D].
 *   bwd:  dy [NK, D] (compute dtype) -> dxs [NK, D] fp32 (gradient wrt the GATHERED rows: the caller scatters/gather-sums it back
 *         to tokens, e.g. mode_rmsnorm_bwd's G/pos input), dw1 [E,8D,D], db1 [E,8D], dw2 [E,D,4D] fp32 (overwritten).
 * x [N, D], w1 [E,8D,D], w2 [E,D,4D], p, h, dy in the compute dtype (bf16 | fp32); perm / offsets from mode_moe_dispatch_meta.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ModeGroupedMlpDesc {
  int32_t dtype; int32_t N, D, E, k;
  const void* x; const int32_t* perm; const int32_t* offsets;
  const void* w1; const float* b1; const void* w2;
  void* p;                        /* [NK, 8D] pre-activation: out of fwd when non-NULL (then h = swiglu(p)), in of bwd              */
  void* h;                        /* [NK, 4D] out of fwd, in of bwd                                                                  */
  void* y; int32_t y_dtype;       /* [NK, D]  out of fwd (bf16 | fp32)                                                               */
  uint32_t seed; float p_drop;    /* expert dropout (hash mask; 0 = off)                                                             */
  const void* dy;                 /* bwd: [NK, D] compute dtype                                                                      */
  float* dxs; float* dw1; float* db1; float* dw2;      /* bwd outputs                                                               */
} ModeGroupedMlpDesc;
size_t mode_moe_grouped_mlp_workspace_bytes(int N, int D, int E, int k, int dtype);          /* backward scratch (dH, dP, column sums) */
int mode_moe_grouped_mlp_fwd(const ModeGroupedMlpDesc* d, void* stream);
int mode_moe_grouped_mlp_bwd(const ModeGroupedMlpDesc* d, void* workspace, size_t workspace_bytes, void* stream);

/* mode_rmsnorm_cond_bwd — complete backward of y = RMSNorm(x; g) (+ cond[row / rows_per_cond]) (modedit.py:72-80, 532): dx [rows, D],
 * dg [D] and (when dcond != NULL) dcond [rows / rows_per_cond, D] = sum of dy over the rows sharing a conditioning vector.  Thin
 * composite of mode_rmsnorm_bwd + mode_colsum (both deterministic). */
size_t mode_rmsnorm_cond_bwd_workspace_bytes(int rows, int D, int rows_per_cond);
int mode_rmsnorm_cond_bwd(const float* x, const float* g, const float* dy, int rows, int D, int rows_per_cond, float eps, float* dx,
                          float* dg, float* dcond, void* workspace, size_t workspace_bytes, void* stream);

/* Router MLP (RouterCond.router: Linear(D,2D) -> GELU -> Linear(2D,E), modedit.py:190-200) for ALL L layers in one launch each:
 * mode_router_logits:  logits[l][r][e] = b3[l][e] + sum_n hid[r][l*K + n] * w3[l][e][n]   (hid row stride ld_hid, K = 2D);
 * mode_router_mlp_bwd: dpre[b][l][n] = (sum_e dlog[l][b][e] w3[l][e][n]) * gelu'(pre[b][l][n]);  dw3[l][e][n] = sum_b dlog[l][b][e] gelu(pre[b][l][n]).
 * pre / dpre are [B][L][2D] (what mode_dit_route writes to r_pre), dlog [L][B][E], w3 / dw3 [L][E][2D] contiguous over layers. */
int mode_router_logits(const float* hid, int64_t ld_hid, const float* w3, int64_t w3_layer_stride, const float* b3, int64_t b3_layer_stride,
                       int L, int R, int E, int K, float* logits, void* stream);
int mode_router_mlp_bwd(const float* dlog, const float* r_pre, const float* w3, int L, int B, int E, int H2, float* dpre, float* dw3, void* stream);
int mode_iota_i32(int32_t* out, int n, int step, void* stream);                               /* out[i] = i * step (K-group offsets) */
/* Fused AdamW over a flat fp32 slice (p, g, m, v: n elements, n % 4 == 0, 16-byte aligned).  Replaces torch.optim.AdamW as configured by
 * MoDEAgent.configure_optimizers (mode/models/mode_agent.py:365-392): decoupled weight decay, bias correction by `step` (1-based),
 * g is scaled by grad_scale first (1/world for a summed data-parallel gradient).  lp_bf16 (nullable) receives the updated weights
 * rounded to bf16 — the compute shadow of the next forward.  ema (nullable): exponential moving average of the weights, updated in the
 * same pass with the updated weight, e -= ema_rate * (e - w), ema_rate = 1 - decay (EMA callback, mode/callbacks/ema.py:83-126). */
int mode_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, float grad_scale, void* lp_bf16, float* ema, float ema_rate, void* stream);
int mode_ema_update(float* ema, const float* p, int64_t n, float rate, void* stream);      /* stand-alone EMA pass (foreign optimizers) */

/* ------------------------------------------------------------------------------------------------------------------
 * Whole-denoiser forward: the launch chain of one MoDeDiT.forward (modedit.py:741-821) [+ GCDenoiser.forward scalings
 * + one sample_ddim update], issued from C++ so a 10-step sampler is ~100 back-to-back launches per step with no host
 * logic in between (and can be captured into a hipGraph by the caller).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ModeDims {
  int32_t D, H, L, E, k, T, A_len, A_dim, O, G, n_img, use_noise_token, router_normalize;
  float eps;
} ModeDims;

/* Per-layer weight table (device pointers).  lp = low-precision compute dtype (bf16 or f32), same layout as torch. */
typedef struct ModeLayerWeights {
  const float* ln1_g; const float* ln2_g; const float* qn_g; const float* kn_g;
  const void* wqkv;  const float* bqkv;     /* [3D, D] rows = [query; key; value], [3D]          */
  const void* wo;                            /* [D, D]                                            */
  const float* r_w0; const float* r_b0;      /* router mlp.0 [2D, D], [2D]   (always fp32)        */
  const float* r_w3; const float* r_b3;      /* router mlp.3 [E, 2D], [E]                         */
  const void* w1;    const float* b1;        /* [E][8D, D], [E][8D]   expert SwishGLU projection  */
  const void* w2;                            /* [E][D, 4D]                                        */
} ModeLayerWeights;

typedef struct ModeModelWeights {
  const float* pos; const float* w_se; const float* b_se; const float* w_sl;   /* fp32 */
  const float* w_tok; const float* w_goal; const float* w_act;                /* fp32 [D,O] [D,G] [D,A] */
  const float* ln_g; const float* w_out; const float* b_out;
  const ModeLayerWeights* layers;                                             /* host array of L entries */
} ModeModelWeights;

/* Dispatch-metadata record of one layer (4-byte word offsets inside the record), see mode_moe_dispatch_meta. */
typedef struct ModeMetaLayout {
  int32_t counts, offsets, perm, pos, posw, poffsets, prow;   /* word offsets */
  int32_t total_words;                        /* record size (multiple of 4 words) */
  int32_t padded_rows;                        /* static bound on poffsets[E]: ceil64(N*k) + 64*E */
} ModeMetaLayout;
int mode_moe_meta_layout(int N, int E, int k, ModeMetaLayout* out);

/* Batched dispatch: nbatch records (e.g. L layers, or steps*L for a whole sampler run) in one launch.
 * topk_idx/topk_w: [nbatch][R][k] (idx_bstride elements apart); meta: [nbatch][layout.total_words]. */
int mode_dit_dispatch(const int32_t* topk_idx, const float* topk_w, int nbatch, int64_t idx_bstride, int R,
                      int tokens_per_row, int N, int E, int k, int32_t* meta, void* stream);

size_t mode_dit_workspace_bytes(const ModeDims* dims, int B, int R, int dtype);

/* sigma [R] -> emb_t [R, D]:  Linear(1,D)(ln(sigma)/4) -> Linear(D,D)   (modedit.py:823-832).  fp32. */
int mode_dit_sigma_embed(const ModeDims* dims, const ModeModelWeights* w, const float* sigma, int R, float* emb_t,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Hoisted, step-invariant embeddings (modedit.py:760, 765): img_e [B*n_img, D] = state_images @ tok_emb^T,
 * goal_e [B, D] = goals @ goal_emb^T.  fp32 MFMA. */
int mode_dit_embed_obs(const ModeDims* dims, const ModeModelWeights* w, const float* state_images, const float* goals,
                       int B, float* img_e, float* goal_e, void* stream);

/* Routing for all L layers on R distinct conditioning rows (cond = emb_t, or emb_t + goal_e with use_goal_in_routing):
 * per layer Linear(D,2D)+GELU -> Linear(2D,E) -> softmax/clamp/top-k.   (modedit.py:194-202, 336, 345-349, 392)
 * Outputs (device, caller-owned): topk_idx int32 [L, R, k]; topk_w fp32 [L, R, k]; probs / shifted fp32 [L, R, E] or NULL. */
int mode_dit_route(const ModeDims* dims, const ModeModelWeights* w, const float* cond, int R,
                   int32_t* topk_idx, float* topk_w, float* probs, float* shifted, float* r_pre /* [R,L,2D] pre-GELU, training */,
                   void* workspace, size_t workspace_bytes, void* stream);
/* combine weights for HOST-chosen expert ids (training: torch.multinomial per token row, modedit.py:390):
 * w[n, j] = probs[n / tokens_per_row, idx[n, j]] (/ their sum when normalize). */
int mode_moe_weights_from_idx(const float* probs, const int32_t* idx, int N, int tokens_per_row, int E, int k, int normalize, float* w,
                              void* stream);

/* Training draw of the expert ids: k per token row WITHOUT replacement, distributed like torch.multinomial(probs, k, replacement=False)
 * (modedit.py:390) - the exponential race torch itself uses for that case: keys probs[n / tokens_per_row, e] / expo[n, e], the k largest in order.
 * expo [N, E]: i.i.d. Exp(1) variates drawn by the caller (the randomness stays on the caller's generator).  Also writes the combine weights
 * of mode_moe_weights_from_idx.  idx int32 [N, k], w fp32 [N, k].  E <= 64. */
int mode_moe_sample_experts(const float* probs, const float* expo, int N, int tokens_per_row, int E, int k, int normalize, int32_t* idx, float* w,
                            void* stream);
/* Router side channels of a training forward for all L layers in one launch (modedit.py:584-593 load-balancing term, 816-820 expert usage,
 * 930-969 router z-loss).  idx / w [L, R, k]: every routing row stands for tokens_per_row token rows; shifted [L, Rs, E] (logits - rowmax).
 * Outputs: frac [L, E] share of token rows per expert; lb [L] = E * sum_e mean_n(scattered combine weights) * frac; zl [L] = mean_r
 * (log(sum_e exp(shifted) + 1e-6))^2; lb_mean / zl_mean: their means over the layers; mask [L, R * tokens_per_row, E] one-hot-of-k rows or NULL;
 * usage int64 [L, E] or NULL: token rows per expert are ADDED.  Deterministic (fixed summation order, no atomics).  E <= 16. */
int mode_moe_aux_stats(const int32_t* idx, const float* w, int L, int R, int tokens_per_row, int E, int k, const float* shifted, int Rs,
                       float* frac, float* lb, float* zl, float* lb_mean, float* zl_mean, float* mask, int64_t* usage, void* stream);

typedef struct ModeForwardArgs {
  int32_t B; int32_t dtype;
  const float* emb_t; int64_t emb_row_stride;        /* sigma token rows; stride 0 => one row shared by the whole batch      */
  const float* cond;  int64_t cond_row_stride;       /* additive conditioning c (normally == emb_t)                          */
  const int32_t* meta; int64_t meta_layer_stride;    /* L dispatch records (mode_dit_dispatch); stride in 4-byte words; NULL: see below */
  const float* goal_e; const float* img_e;           /* hoisted embeddings fp32 [B,D], [B,n_img,D]                           */
  const float* actions;                              /* [B, A_len, A_dim] un-scaled noisy actions                            */
  const float* c_in; int64_t c_in_stride;            /* NULL => 1                                                            */
  const float* scal; int64_t scal_stride;            /* {c_skip,c_out,r,_} per sample (stride 4) / shared (0); NULL => F only */
  float* F; float* denoised; float* x_next;
  /* ABI 4.  meta == NULL selects TOKEN routing (the reference's cond_router=False, modedit.py:296-301, 322-325, 550-553): every block routes
   * each token on its own ln_2-normalised state through the block's router MLP (fp32), top-k and dispatch happen inside the chain, per layer.
   * topk_idx_out (optional): int32 [L][B*T][k] - the experts every token took (for the caller's side channels / tests). */
  int32_t* topk_idx_out;
  /* ABI 5.  PROMISE by the caller: every dispatch record in `meta` was built from ONE routing row for the whole batch (mode_dit_dispatch with
   * R = 1: the sampler's uniform sigma), i.e. every expert segment holds all N tokens in token order.  The chain then passes
   * MODE_GEMM_UNIFORM_GROUPS | MODE_GEMM_IDENTITY_ROWS to the expert GEMMs (gather indices computed, not loaded).  0 = no promise (always
   * correct).  It is NOT inferred from cond_row_stride: a shared conditioning row with per-sample routing is a legal call. */
  int32_t uniform_routing;
  const float* den_prev;                             /* ABI 12: ModeHeadDesc.den_prev of the chain's head (two-point multistep samplers); NULL = none */
  const float* lin; const float* aux1; const float* aux2;   /* ABI 12: ModeHeadDesc.lin / aux1 / aux2 (two-stage solvers); NULL = none */
} ModeForwardArgs;
int mode_dit_forward(const ModeDims* dims, const ModeModelWeights* w, const ModeForwardArgs* a,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training chains (score-matching step).  mode_dit_forward_train = MoDeDiT.forward in .train() mode (per-token routing ids chosen by
 * the host, attention / expert dropout) that keeps every activation the backward needs in `stash`; mode_dit_backward turns dF into
 * the gradient of every parameter (written, not accumulated) — the HIP counterpart of autograd over modedit.py:741-821.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ModeStashLayout {
  uint64_t x0, h1, qkv, yattn, x1, ub, P, Hd, Y;     /* byte offsets inside one layer record */
  uint64_t layer_stride;
  uint64_t xL, yL, u_tmp, tr_hid, tr_logits, global_bytes;   /* global part (precedes the layer records); tr_*: token-router scratch */
  uint64_t total_bytes;
} ModeStashLayout;
int mode_dit_train_stash_layout(const ModeDims* dims, int B, int dtype, ModeStashLayout* out);
size_t mode_dit_train_workspace_bytes(const ModeDims* dims, int B, int dtype);

typedef struct ModeTrainArgs {
  int32_t B; int32_t dtype;
  uint32_t seed; float attn_pdrop; float mlp_pdrop;
  const float* sigma;            /* [B]                                                     */
  const float* e1;               /* [B, D]  Linear(1,D)(ln sigma / 4)  (mode_sigma_embed)    */
  const float* emb_t;            /* [B, D]                                                   */
  const float* cond;             /* [B, D]  additive conditioning = router input             */
  int32_t goal_in_cond;          /* use_goal_in_routing: cond = emb_t + goal_e               */
  const float* state_images; const float* goals;          /* raw inputs [B*n_img, O], [B, G] (embedding weight gradients) */
  const float* goal_e; const float* img_e;                /* hoisted embeddings                                          */
  const float* actions; const float* c_in; int64_t c_in_stride;   /* un-scaled noisy actions [B, A_len, A]; c_in [B] or NULL      */
  const float* actions_scaled;                            /* actions * c_in (backward of action_emb)                     */
  const int32_t* act_rows;                                /* [B*A_len] token row of every action token                    */
  const int32_t* meta; int64_t meta_layer_stride;         /* L per-token dispatch records                                 */
  const int32_t* topk_idx; int64_t topk_layer_stride; int32_t idx_per_token;   /* [L][B*T or B][k] expert ids            */
  const float* probs;            /* [L, B, E] clamped softmax of the router ([L, N, E] under token routing: written by phase 0) */
  const float* r_pre;            /* [B, L, 2D] router pre-GELU activations (mode_dit_route)  */
  float* F;                      /* out: [B, A_len, A]                                       */
  void* const* layer_events;     /* optional hipEvent_t[L] (backward only): event l is recorded on the stream as soon as ALL weight
                                    gradients of block l (wqkv, bqkv, wo, w1, b1, w2) are written — blocks finish in the order L-1 … 0,
                                    so a data-parallel reducer can exchange block l's gradient slice while earlier blocks are still
                                    back-propagating (replaces DDP's autograd-hook bucketing, mode/training_calvin.py:92-103) */
  /* auxiliary router losses (backward only; all NULL = off): see mode_moe_router_bwd_aux */
  const float* shifted;          /* [L, B, E] max-shifted router logits (mode_dit_route)     */
  const float* aux_lb_coef;      /* [L, E] device                                            */
  const float* aux_z_coef;       /* device scalar                                            */
  /* ABI 5: gradients with respect to the INPUTS (backward only, each optional).  The reference trains its perceptual encoders through
   * `state_images` (mode/models/mode_agent.py:404-411, 548-567: perceptual_emb comes straight from the FiLM-ResNets), so the backward chain
   * must hand d state_images to whatever produced them:  d state_images = d img_e · W_tok,  d goals = d goal_e · W_goal  (d goal_e includes
   * the conditioning/router path when goal_in_cond).  Two fp32 GEMMs on quantities the chain already holds. */
  float* d_state_images;         /* out: [B*n_img, O] or NULL                                */
  float* d_goals;                /* out: [B, G] or NULL (gradient w.r.t. the goals AFTER preprocess_goals / the Bernoulli mask) */
  /* ABI 5: TOKEN routing in training (the reference's cond_router=False, modedit.py:296-301, 322-325, 550-553: every block routes each token on
   * its own ln_2-normalised state).  The forward is driven layer by layer (mode_dit_forward_train_layer): phase 0 of layer l runs attention +
   * ln_2 + the block's fp32 router and writes, for THAT layer, probs[l] / tr_shifted[l] ([L, N, E], N = B*T routing rows), the pre-GELU
   * activations tr_pre[l] ([L, N, 2D], kept for the backward) and the top-k choice into tr_topk_idx / tr_topk_w ([N, k], one layer's worth);
   * the caller then fixes the experts of the layer (top-k, or torch.multinomial on probs[l] like modedit.py:390), fills topk_idx[l]
   * (idx_per_token = 1) and meta[l], and runs phase 1 (experts + combine).  The backward (one call) back-propagates every block's router in
   * place: d u gets the router's term, r_pre / the batched conditioning-row router backward are not used. */
  int32_t token_routing;
  float* tr_pre; float* tr_shifted; int32_t* tr_topk_idx; float* tr_topk_w;
  /* (ABI 11, backward only) non-NULL: the expert matrices w1 / w2 of every block are UPDATED by their weight-gradient GEMMs (ModeAdamWFuse) instead of
   * receiving a gradient: grads->layers[l].w1 / .w2 only locate the tensors inside the arenas and are not written.  bf16 compute mode only
   * (MODE_ERR_UNSUPPORTED otherwise).  gsq (optional) receives mode_adamw_fuse_gsq_floats(dims) floats: per block [w2 tiles | w1 tiles], block l at
   * l * (that count / L); their sum is the squared norm of the scaled expert-matrix gradients of the step. */
  const ModeAdamWFuse* fuse_adamw;
} ModeTrainArgs;
int64_t mode_adamw_fuse_gsq_floats(const ModeDims* dims);   /* L * E * 12 * ceil(D/128)^2 */
int mode_dit_forward_train(const ModeDims* dims, const ModeModelWeights* w, const ModeTrainArgs* a, void* stash, size_t stash_bytes,
                           void* stream);
/* One layer of the training forward in two phases (see ModeTrainArgs.token_routing): phase 0 = [token embedding if layer == 0,] QKV, attention,
 * c_proj + residual, ln_2 [, token router]; phase 1 = experts, combine [, output head if layer == L - 1].  Phases must be called in order. */
int mode_dit_forward_train_layer(const ModeDims* dims, const ModeModelWeights* w, const ModeTrainArgs* a, void* stash, size_t stash_bytes,
                                 int layer, int phase, void* stream);

typedef struct ModeLayerGrads {           /* fp32 gradients, same shapes as ModeLayerWeights (packed qkv / stacked experts) */
  float* ln1_g; float* ln2_g; float* qn_g; float* kn_g; float* wqkv; float* bqkv; float* wo;
  float* r_w0; float* r_b0; float* r_w3; float* r_b3; float* w1; float* b1; float* w2;
} ModeLayerGrads;
typedef struct ModeModelGrads {
  float* pos; float* w_se; float* b_se; float* w_sl; float* w_tok; float* w_goal; float* w_act; float* ln_g; float* w_out; float* b_out;
  const ModeLayerGrads* layers;
} ModeModelGrads;
typedef struct ModeLayerWeightsT {        /* transposed shadows for the data-gradient GEMMs of the fp32 compute mode only: in bf16 the backward
                                             GEMMs read the [out,in] weights directly (MODE_GEMM_W_KN) and these may be NULL */
  const void* wqkvT;  /* [D, 3D] */  const void* woT;   /* [D, D] */
  const void* w1T;    /* [E][D, 8D] */ const void* w2T;  /* [E][4D, D] */
} ModeLayerWeightsT;
typedef struct ModeModelWeightsT { const float* w_outT; /* [D, A] */ const ModeLayerWeightsT* layers; } ModeModelWeightsT;
/* Router weights AND router gradients of the L layers must be contiguous ([L,2D,D], [L,2D], [L,E,2D], [L,E]: layers[l].r_w0 ==
 * layers[0].r_w0 + l*2D*D, ...): the router MLPs of all layers are back-propagated as one batch (MODE_ERR_UNSUPPORTED otherwise). */
int mode_dit_backward(const ModeDims* dims, const ModeModelWeights* w, const ModeModelWeightsT* wt, const ModeTrainArgs* a,
                      const void* stash, const float* dF, const ModeModelGrads* grads, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * FiLM-ResNet perceptual encoders (mode/models/perceptual_encoders/pretrained_resnets.py:5-60, resnets.py:27-200,
 * mode_agent.py:548-567): the producer of `state_images`.  The convolutions themselves are mode_gemm calls (1 x 1; a_rows / w_rows in taps for k x k: csrc/conv_gemm.hip) except the 3-channel stem, which the caller leaves
 * to MIOpen; these entry points are
 * everything between two convolutions as ONE pass over the activation (NCHW, or channels_last - ModeBnFilmDesc.channels_last):
 *   y = post_film( relu( pre_film( x * scale[c] + shift[c] ) + residual ) )
 *   pre_film : v = pre_gamma[n,c] * v + pre_beta[n,c]          (BasicBlockWithModulation, resnets.py:64-71: after bn2, before the skip add)
 *   post_film: v = (1 + post_gamma[n,c]) * v + post_beta[n,c]  (FiLMLayer after a whole stage, pretrained_resnets.py:19-23)
 * scale / shift are the folded BatchNorm (eval: running statistics; training: the batch statistics of mode_bn_stats).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ModeBnFilmDesc {
  int32_t N, C, HW; int32_t dtype;                 /* activation dtype of x / residual / y (and dy / dx): MODE_F32 or MODE_BF16 */
  const void* x;                                   /* [N, C, HW] convolution output                                    */
  const float* scale; const float* shift;          /* [C]                                                              */
  const float* pre_gamma; const float* pre_beta;   /* [N, C] or both NULL                                              */
  const void* residual;                            /* [N, C, HW] or NULL                                               */
  int32_t relu;
  const float* post_gamma; const float* post_beta; /* [N, C] or both NULL                                              */
  void* y;                                         /* [N, C, HW] (unused by the backward)                              */
  /* ABI 8.  Forward only, when scale == shift == NULL: the eval-mode BatchNorm is folded inside the pass - scale = bn_weight / sqrt(bn_var + bn_eps),
   * shift = bn_bias - bn_mean * scale ([C] fp32 each; bn_weight / bn_bias may be NULL = 1 / 0): one launch per BatchNorm in the rollout. */
  const float* bn_weight; const float* bn_bias; const float* bn_mean; const float* bn_var; float bn_eps;
  int32_t channels_last;                           /* 0: x / residual / y (dy, dx) are [N][C][HW] (NCHW); 1: [N][HW][C] (torch.channels_last: what MIOpen's
                                                      implicit-GEMM convolutions read and write without layout transposes); needs C % 8 == 0 (bf16) / 4 (fp32) */
} ModeBnFilmDesc;
int mode_bn_film_act_fwd(const ModeBnFilmDesc* d, void* stream);
size_t mode_bn_workspace_bytes(int N, int C, int HW, int dtype, int channels_last);
/* per-channel mean and BIASED variance over (N, HW) (training-mode nn.BatchNorm2d); fixed summation order, no atomics */
int mode_bn_stats(const void* x, int dtype, int N, int C, int HW, int channels_last, float* mean, float* var, void* workspace, size_t workspace_bytes,
                  void* stream);
/* Everything a BatchNorm2d contributes before the fused pass, in two launches: x != NULL (training) - batch mean / biased variance over (N, HW) as
 * mode_bn_stats, x == NULL (eval) - the statistics are running_mean / running_var; then invstd = 1 / sqrt(var + eps), scale = weight * invstd, shift =
 * bias - mean * scale (weight / bias NULL = 1 / 0), and nn.BatchNorm2d's bookkeeping IN PLACE when training and the pointers are given: running_mean /
 * running_var (unbiased variance; momentum >= 0: exponential average, < 0: cumulative average 1 / num_batches_tracked as with momentum=None),
 * num_batches_tracked += 1.  All outputs [C] fp32. */
int mode_bn_prepare(const void* x, int dtype, int N, int C, int HW, int channels_last, const float* weight, const float* bias, float eps, float momentum,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* var, float* invstd, float* scale,
                    float* shift, void* workspace, size_t workspace_bytes, void* stream);
/* (ABI 10) The same fold + bookkeeping from PARTIAL sums [rows][C] (sum and sum of squares per row block) that the producing convolution wrote in its epilogue
 * (ModeConvBnDesc.stat_sum / stat_sq): no pass over the activation.  count = elements per channel (N * HW). */
int mode_bn_prepare_partials(const float* psum, const float* psq, int rows, double count, int C, const float* weight, const float* bias, float eps, float momentum,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* var, float* invstd, float* scale,
                             float* shift, void* stream);
/* Backward of the fused chain.  mean / invstd [C]: the statistics scale / shift were folded from (scale = weight * invstd).  training != 0:
 * gradient through the batch statistics; 0: dx = d * scale.  Outputs: dx, dresidual (iff d->residual), dweight / dbias [C] (BatchNorm affine),
 * d_pre_* / d_post_* [N, C] (iff the corresponding FiLM is present).
 * phase 0 = everything; for nn.SyncBatchNorm (mode/training_calvin.py:102 sync_batchnorm=True) the caller runs phase 1 (reductions only: dweight /
 * dbias are THIS rank's sums), sums dweight / dbias over the ranks, and runs phase 2 (dx / dresidual only) with those sums and inv_count =
 * 1 / (global N * HW); inv_count <= 0 means 1 / (N * HW). */
int mode_bn_film_act_bwd(const ModeBnFilmDesc* d, const void* dy, const float* mean, const float* invstd, int training, int phase, float inv_count,
                         void* dx, void* dresidual, float* dweight, float* dbias, float* d_pre_gamma, float* d_pre_beta, float* d_post_gamma,
                         float* d_post_beta, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * (ABI 12) The encoders' entry (csrc/conv_stem.hip): the small-channel k x k convolution whose input is the camera image (ResNet conv1: 3 -> 64
 * channels, 7 x 7, stride 2, padding 3 - reference mode/models/perceptual_encoders/resnets.py:96 torchvision `resnet18`, pretrained_resnets.py:29 timm
 * `resnet50`) and the max-pool behind it (`maxpool`, 3 x 3 / stride 2 / padding 1).  Replaces the last MIOpen / aten kernels of the encoders.
 *   x        the image [N, Cin, H, W] addressed by ELEMENT strides (NCHW, channels_last, a slice ...), fp32 or bf16; fp32 is rounded to bf16
 *            (nearest-even) as it is read, i.e. the result equals the convolution of x.to(bf16)
 *   w        [Cout][kh][kw][Cin] bf16 = the channels_last storage of the [Cout, Cin, kh, kw] weight; Cout % 16 == 0, Cout <= 64, kh * kw * Cin <= 256
 *   forward  y [N * ho * wo][Cout] bf16 (channels_last), fp32 accumulation; optional epilogue on the fp32 sums (inference): eval-mode BatchNorm
 *            (bn_mean / bn_var both NULL or both [Cout]; bn_weight / bn_bias may be NULL = 1 / 0; folded as in mode_conv_bn_act_fwd) and ReLU
 *   wgrad    dw_part [mode_stem_conv_wgrad_slabs()][Cout][kh][kw][Cin] fp32 partial sums, one slab per workgroup; dW = their sum in slab order
 *            (dy [N * ho * wo][Cout] bf16).  There is no data gradient: nothing upstream of the image is trained.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ModeStemConvDesc {
  const void* x; int32_t x_dtype;                  /* MODE_F32 | MODE_BF16 */
  int64_t sxn, sxc, sxh, sxw;                      /* element strides of x */
  int32_t N, H, W, Cin, kh, kw, sh, sw, ph, pw, Cout;
  const void* w;
  void* y;                                                            /* forward */
  const float* bn_mean; const float* bn_var; const float* bn_weight; const float* bn_bias; float bn_eps; int32_t relu;
  const void* dy; float* dw_part;                                     /* weight gradient */
} ModeStemConvDesc;
int mode_stem_conv_fwd(const ModeStemConvDesc* d, void* stream);
int mode_stem_conv_wgrad_slabs(const ModeStemConvDesc* d);            /* number of partial slabs mode_stem_conv_wgrad writes (0: bad descriptor) */
int mode_stem_conv_wgrad(const ModeStemConvDesc* d, void* stream);
/* Max-pool k x k / stride s / padding pad (2 * pad <= k) on channels_last data [N][H][W][C] (C % 8 == 0; MODE_F32 | MODE_BF16) with aten's selection
 * rule (a later element replaces the maximum iff it is greater or NaN).  argmax (may be NULL in inference) [N][ho][wo][C] uint8 = window position
 * a * k + b of the selected element; the backward sums, per INPUT pixel, the dy of the windows that selected it (fp32 sum in ascending window
 * order, no atomics). */
int mode_maxpool_nhwc_fwd(const void* x, int dtype, int N, int H, int W, int C, int k, int s, int pad, void* y, uint8_t* argmax, void* stream);
int mode_maxpool_nhwc_bwd(const void* dy, const uint8_t* argmax, int dtype, int N, int H, int W, int C, int k, int s, int pad, void* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MODE_HIP_H */
