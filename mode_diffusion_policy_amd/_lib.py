"""ctypes binding of libmode_hip.so (C-ABI declared in include/mode_hip.h).

The HIP library IS the product: there is no CPU / PyTorch fallback.  If the shared object is missing or a symbol is
absent, loading fails loudly (``ModeHipUnavailable``) and every operator that needs it raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MODE_HIP_LIB", os.path.join(_HERE, "libmode_hip.so"))

MODE_BF16, MODE_F32 = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_RESIDUAL, EPI_SWIGLU, EPI_RESIDUAL_NORM = 0, 1, 2, 3, 4, 5
ABI_VERSION = 12
GEMM_SKINNY_OK, GEMM_W_KN, GEMM_A_KM, GEMM_UNIFORM_GROUPS, GEMM_SMALL_ROWS, GEMM_IDENTITY_ROWS = 1, 2, 4, 8, 16, 32

c_i32, c_i64, c_f32, c_vp, c_sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class ModeHipUnavailable(RuntimeError):
    pass


class ModeAdamWFuse(C.Structure):
    """AdamW in the epilogue of a weight-gradient GEMM (include/mode_hip.h, ABI 11): arena base pointers (same element offsets as the gradient
    arena), the step's hyper-parameters, optional per-workgroup sums of squares of the gradient."""
    _fields_ = [("grad_base", c_vp), ("param_base", c_vp), ("exp_avg_base", c_vp), ("exp_avg_sq_base", c_vp), ("lp_base", c_vp), ("ema_base", c_vp),
                ("ema_rate", c_f32), ("lr", c_f32), ("beta1", c_f32), ("beta2", c_f32), ("eps", c_f32), ("weight_decay", c_f32), ("step", c_i32),
                ("grad_scale", c_f32), ("gsq", c_vp), ("gsq_capacity", c_i64), ("side_stream", c_vp), ("side_events", c_vp)]


class ModeGemmDesc(C.Structure):
    _fields_ = [("dtype", c_i32), ("epilogue", c_i32), ("out_dtype", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("A", c_vp), ("lda", c_i64), ("W", c_vp), ("ldw", c_i64), ("w_expert_stride", c_i64),
                ("bias", c_vp), ("bias_expert_stride", c_i64), ("resid", c_vp), ("ldr", c_i64), ("C", c_vp), ("ldc", c_i64),
                ("a_rows", c_vp), ("expert_offsets", c_vp), ("num_experts", c_i32), ("split_k", c_i32), ("split_stride", c_i64), ("k_group_offsets", c_vp), ("num_k_groups", c_i32),
                ("c_group_stride", c_i64), ("flags", c_i32), ("w_rows", c_vp),
                ("C2", c_vp), ("ldc2", c_i64), ("gain", c_vp), ("row_ss_out", c_vp), ("row_ss", c_vp), ("row_ss_n", c_i32), ("row_eps", c_f32),
                ("w_tap_cols", c_i32), ("w_rows_tap_stride", c_i64), ("a_tap_cols", c_i32), ("a_rows_tap_stride", c_i64),
                ("adamw", C.POINTER(ModeAdamWFuse))]


class ModeEmbedDesc(C.Structure):
    _fields_ = [("B", c_i32), ("T", c_i32), ("D", c_i32), ("A_len", c_i32), ("A_dim", c_i32), ("n_img", c_i32),
                ("use_noise_token", c_i32), ("emb_t", c_vp), ("emb_row_stride", c_i64), ("goal_e", c_vp), ("img_e", c_vp),
                ("actions", c_vp), ("c_in", c_vp), ("c_in_stride", c_i64), ("w_act", c_vp), ("pos", c_vp), ("g", c_vp),
                ("cond", c_vp), ("cond_row_stride", c_i64), ("eps", c_f32), ("x", c_vp), ("h", c_vp), ("h_dtype", c_i32)]


class ModeHeadDesc(C.Structure):
    _fields_ = [("B", c_i32), ("T", c_i32), ("D", c_i32), ("A_len", c_i32), ("A_dim", c_i32), ("k", c_i32),
                ("u", c_vp), ("Y", c_vp), ("y_dtype", c_i32), ("y_splits", c_i32), ("y_split_stride", c_i64), ("pos", c_vp), ("posw", c_vp), ("g", c_vp), ("eps", c_f32),
                ("w_out", c_vp), ("b_out", c_vp), ("x_a", c_vp), ("scal", c_vp), ("scal_stride", c_i64),
                ("F", c_vp), ("denoised", c_vp), ("x_next", c_vp), ("u_ss", c_vp), ("u_ss_n", c_i32), ("u_gain", c_vp), ("den_prev", c_vp), ("lin", c_vp), ("aux1", c_vp), ("aux2", c_vp)]


class ModeDims(C.Structure):
    _fields_ = [("D", c_i32), ("H", c_i32), ("L", c_i32), ("E", c_i32), ("k", c_i32), ("T", c_i32), ("A_len", c_i32),
                ("A_dim", c_i32), ("O", c_i32), ("G", c_i32), ("n_img", c_i32), ("use_noise_token", c_i32),
                ("router_normalize", c_i32), ("eps", c_f32)]


class ModeLayerWeights(C.Structure):
    _fields_ = [("ln1_g", c_vp), ("ln2_g", c_vp), ("qn_g", c_vp), ("kn_g", c_vp), ("wqkv", c_vp), ("bqkv", c_vp),
                ("wo", c_vp), ("r_w0", c_vp), ("r_b0", c_vp), ("r_w3", c_vp), ("r_b3", c_vp), ("w1", c_vp), ("b1", c_vp),
                ("w2", c_vp)]


class ModeModelWeights(C.Structure):
    _fields_ = [("pos", c_vp), ("w_se", c_vp), ("b_se", c_vp), ("w_sl", c_vp), ("w_tok", c_vp), ("w_goal", c_vp),
                ("w_act", c_vp), ("ln_g", c_vp), ("w_out", c_vp), ("b_out", c_vp), ("layers", C.POINTER(ModeLayerWeights))]


class ModeMetaLayout(C.Structure):
    _fields_ = [("counts", c_i32), ("offsets", c_i32), ("perm", c_i32), ("pos", c_i32), ("posw", c_i32), ("poffsets", c_i32), ("prow", c_i32),
                ("total_words", c_i32), ("padded_rows", c_i32)]


c_u64, c_u32 = C.c_uint64, C.c_uint32


class ModeStashLayout(C.Structure):
    _fields_ = [(n, c_u64) for n in ("x0", "h1", "qkv", "yattn", "x1", "ub", "P", "Hd", "Y", "layer_stride", "xL", "yL", "u_tmp", "tr_hid", "tr_logits", "global_bytes",
                                     "total_bytes")]


class ModeGroupedMlpDesc(C.Structure):
    _fields_ = [("dtype", c_i32), ("N", c_i32), ("D", c_i32), ("E", c_i32), ("k", c_i32), ("x", c_vp), ("perm", c_vp), ("offsets", c_vp),
                ("w1", c_vp), ("b1", c_vp), ("w2", c_vp), ("p", c_vp), ("h", c_vp), ("y", c_vp), ("y_dtype", c_i32), ("seed", c_u32),
                ("p_drop", c_f32), ("dy", c_vp), ("dxs", c_vp), ("dw1", c_vp), ("db1", c_vp), ("dw2", c_vp)]


class ModeTrainArgs(C.Structure):
    _fields_ = [("B", c_i32), ("dtype", c_i32), ("seed", c_u32), ("attn_pdrop", c_f32), ("mlp_pdrop", c_f32), ("sigma", c_vp), ("e1", c_vp),
                ("emb_t", c_vp), ("cond", c_vp), ("goal_in_cond", c_i32), ("state_images", c_vp), ("goals", c_vp), ("goal_e", c_vp),
                ("img_e", c_vp), ("actions", c_vp), ("c_in", c_vp), ("c_in_stride", c_i64), ("actions_scaled", c_vp), ("act_rows", c_vp),
                ("meta", c_vp), ("meta_layer_stride", c_i64), ("topk_idx", c_vp), ("topk_layer_stride", c_i64), ("idx_per_token", c_i32),
                ("probs", c_vp), ("r_pre", c_vp), ("F", c_vp), ("layer_events", c_vp), ("shifted", c_vp), ("aux_lb_coef", c_vp), ("aux_z_coef", c_vp),
                ("d_state_images", c_vp), ("d_goals", c_vp), ("token_routing", c_i32), ("tr_pre", c_vp), ("tr_shifted", c_vp), ("tr_topk_idx", c_vp),
                ("tr_topk_w", c_vp), ("fuse_adamw", C.POINTER(ModeAdamWFuse))]


class ModeLayerGrads(C.Structure):
    _fields_ = [(n, c_vp) for n in ("ln1_g", "ln2_g", "qn_g", "kn_g", "wqkv", "bqkv", "wo", "r_w0", "r_b0", "r_w3", "r_b3", "w1", "b1", "w2")]


class ModeModelGrads(C.Structure):
    _fields_ = [(n, c_vp) for n in ("pos", "w_se", "b_se", "w_sl", "w_tok", "w_goal", "w_act", "ln_g", "w_out", "b_out")] + \
               [("layers", C.POINTER(ModeLayerGrads))]


class ModeLayerWeightsT(C.Structure):
    _fields_ = [(n, c_vp) for n in ("wqkvT", "woT", "w1T", "w2T")]


class ModeModelWeightsT(C.Structure):
    _fields_ = [("w_outT", c_vp), ("layers", C.POINTER(ModeLayerWeightsT))]


class ModeForwardArgs(C.Structure):
    _fields_ = [("B", c_i32), ("dtype", c_i32), ("emb_t", c_vp), ("emb_row_stride", c_i64), ("cond", c_vp),
                ("cond_row_stride", c_i64), ("meta", c_vp), ("meta_layer_stride", c_i64), ("goal_e", c_vp), ("img_e", c_vp),
                ("actions", c_vp), ("c_in", c_vp), ("c_in_stride", c_i64), ("scal", c_vp), ("scal_stride", c_i64),
                ("F", c_vp), ("denoised", c_vp), ("x_next", c_vp), ("topk_idx_out", c_vp), ("uniform_routing", c_i32), ("den_prev", c_vp), ("lin", c_vp), ("aux1", c_vp), ("aux2", c_vp)]


class ModeQkvAttnDesc(C.Structure):
    _fields_ = [("dtype", c_i32), ("B", c_i32), ("T", c_i32), ("H", c_i32), ("D", c_i32), ("h", c_vp), ("ldh", c_i64), ("wqkv", c_vp), ("ldw", c_i64),
                ("bqkv", c_vp), ("q_gain", c_vp), ("k_gain", c_vp), ("eps", c_f32), ("y", c_vp), ("ldy", c_i64)]


class ModeConvBnDesc(C.Structure):
    _fields_ = [("x", c_vp), ("ldx", c_i64), ("idx", c_vp), ("idx_tap_stride", c_i64), ("taps", c_i32), ("w", c_vp), ("ldw", c_i64), ("y", c_vp), ("ldy", c_i64),
                ("M", c_i32), ("Cin", c_i32), ("Cout", c_i32), ("bn_mean", c_vp), ("bn_var", c_vp), ("bn_weight", c_vp), ("bn_bias", c_vp), ("bn_eps", c_f32),
                ("residual", c_vp), ("ldr", c_i64), ("relu", c_i32), ("pre_gamma", c_vp), ("pre_beta", c_vp), ("post_gamma", c_vp), ("post_beta", c_vp),
                ("rows_per_sample", c_i32), ("stat_sum", c_vp), ("stat_sq", c_vp)]


class ModeBnFilmDesc(C.Structure):
    _fields_ = [("N", c_i32), ("C", c_i32), ("HW", c_i32), ("dtype", c_i32), ("x", c_vp), ("scale", c_vp), ("shift", c_vp), ("pre_gamma", c_vp),
                ("pre_beta", c_vp), ("residual", c_vp), ("relu", c_i32), ("post_gamma", c_vp), ("post_beta", c_vp), ("y", c_vp),
                ("bn_weight", c_vp), ("bn_bias", c_vp), ("bn_mean", c_vp), ("bn_var", c_vp), ("bn_eps", c_f32), ("channels_last", c_i32)]


class ModeStemConvDesc(C.Structure):
    _fields_ = [("x", c_vp), ("x_dtype", c_i32), ("sxn", c_i64), ("sxc", c_i64), ("sxh", c_i64), ("sxw", c_i64), ("N", c_i32), ("H", c_i32), ("W", c_i32),
                ("Cin", c_i32), ("kh", c_i32), ("kw", c_i32), ("sh", c_i32), ("sw", c_i32), ("ph", c_i32), ("pw", c_i32), ("Cout", c_i32), ("w", c_vp),
                ("y", c_vp), ("bn_mean", c_vp), ("bn_var", c_vp), ("bn_weight", c_vp), ("bn_bias", c_vp), ("bn_eps", c_f32), ("relu", c_i32), ("dy", c_vp),
                ("dw_part", c_vp)]


P = C.POINTER
# name -> (restype, argtypes): every symbol include/mode_hip.h declares
PROTOTYPES = {
    "mode_hip_version": (C.c_int, []),
    "mode_hip_status_string": (C.c_char_p, [C.c_int]),
    "mode_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "mode_probe_mfma_burn": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, P(C.c_double), c_vp]),
    "mode_hip_sizeof": (c_sz, [C.c_char_p]),
    "mode_gemm": (C.c_int, [P(ModeGemmDesc), c_vp]),
    "mode_rmsnorm_cond_fwd": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_f32, c_vp, c_vp, C.c_int, c_vp]),
    "mode_attn_block_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32, C.c_uint32, c_f32, c_vp]),
    "mode_qkv_attn_fwd": (C.c_int, [C.POINTER(ModeQkvAttnDesc), c_vp]),
    "mode_conv_bn_act_fwd": (C.c_int, [C.POINTER(ModeConvBnDesc), c_vp]),
    "mode_attn_block_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32, C.c_uint32,
                                      c_f32, c_vp]),
    "mode_sigma_embed": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, c_vp]),
    "mode_moe_route_topk_f32": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mode_moe_dispatch_meta": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mode_moe_combine_norm_fwd": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, c_i64, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp, c_vp, C.c_int,
                                            c_f32, c_vp, c_vp, C.c_int, c_vp]),
    "mode_moe_combine_norm_fused_fwd": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp, C.c_int, C.c_int, c_i64, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp,
                                                  c_vp, C.c_int, c_f32, c_vp, c_vp, C.c_int, c_vp]),
    "mode_embed_tokens_fwd": (C.c_int, [P(ModeEmbedDesc), c_vp]),
    "mode_head_ddim_fwd": (C.c_int, [P(ModeHeadDesc), c_vp]),
    "mode_ddim_edm_step": (C.c_int, [c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp, c_vp, c_vp]),
    "mode_transpose": (C.c_int, [c_vp, c_i64, C.c_int, C.c_int, c_vp, c_i64, c_vp, c_vp, C.c_int, c_vp]),
    "mode_colsum_workspace_bytes": (c_sz, [C.c_int, C.c_int, C.c_int]),
    "mode_colsum": (C.c_int, [c_vp, c_i64, C.c_int, C.c_int, C.c_int, c_vp, C.c_int, C.c_int, c_vp, C.c_int, c_vp, c_sz, c_vp]),
    "mode_swiglu_fwd": (C.c_int, [c_vp, c_vp, c_i64, C.c_int, C.c_int, C.c_uint32, c_f32, c_vp]),
    "mode_swiglu_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_int, C.c_uint32, c_f32, c_vp]),
    "mode_rmsnorm_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_f32, c_vp, C.c_int, c_vp, c_vp, c_vp, C.c_int, c_vp]),
    "mode_rowcopy_f32": (C.c_int, [c_vp, c_i64, C.c_int, C.c_int, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "mode_gelu_fwd": (C.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "mode_gelu_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mode_moe_router_bwd": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp]),
    "mode_moe_router_bwd_aux": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp]),
    "mode_edm_noise_scale": (C.c_int, [c_vp, c_vp, c_vp, c_f32, C.c_int, C.c_int, c_vp, c_vp]),
    "mode_edm_loss": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, C.c_int, C.c_int, c_vp, c_vp, c_vp]),
    "mode_sigma_embed_bwd": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, c_vp, c_vp, c_vp]),
    "mode_moe_combine_bwd": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp, c_vp, c_vp]),
    "mode_moe_meta_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, P(ModeMetaLayout)]),
    "mode_dit_dispatch": (C.c_int, [c_vp, c_vp, C.c_int, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp]),
    "mode_dit_workspace_bytes": (c_sz, [P(ModeDims), C.c_int, C.c_int, C.c_int]),
    "mode_dit_sigma_embed": (C.c_int, [P(ModeDims), P(ModeModelWeights), c_vp, C.c_int, c_vp, c_vp, c_sz, c_vp]),
    "mode_dit_embed_obs": (C.c_int, [P(ModeDims), P(ModeModelWeights), c_vp, c_vp, C.c_int, c_vp, c_vp, c_vp]),
    "mode_dit_route": (C.c_int, [P(ModeDims), P(ModeModelWeights), c_vp, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mode_moe_weights_from_idx": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp]),
    "mode_moe_sample_experts": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp, c_vp]),
    "mode_moe_aux_stats": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mode_dit_train_stash_layout": (C.c_int, [P(ModeDims), C.c_int, C.c_int, P(ModeStashLayout)]),
    "mode_dit_train_workspace_bytes": (c_sz, [P(ModeDims), C.c_int, C.c_int]),
    "mode_dit_forward_train": (C.c_int, [P(ModeDims), P(ModeModelWeights), P(ModeTrainArgs), c_vp, c_sz, c_vp]),
    "mode_dit_forward_train_layer": (C.c_int, [P(ModeDims), P(ModeModelWeights), P(ModeTrainArgs), c_vp, c_sz, c_i32, c_i32, c_vp]),
    "mode_moe_grouped_mlp_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "mode_moe_grouped_mlp_fwd": (C.c_int, [P(ModeGroupedMlpDesc), c_vp]),
    "mode_moe_grouped_mlp_bwd": (C.c_int, [P(ModeGroupedMlpDesc), c_vp, c_sz, c_vp]),
    "mode_rmsnorm_cond_bwd_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32]),
    "mode_rmsnorm_cond_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mode_swiglu_bwd_bias_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32]),
    "mode_swiglu_bwd_bias": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_u32, c_f32, c_vp, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "mode_pos_emb_bwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mode_router_logits": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mode_router_mlp_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "mode_iota_i32": (C.c_int, [c_vp, c_i32, c_i32, c_vp]),
    "mode_adamw_step": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, c_i32, C.c_float, c_vp, c_vp,
                                  C.c_float, c_vp]),
    "mode_ema_update": (C.c_int, [c_vp, c_vp, c_i64, C.c_float, c_vp]),
    "mode_adamw_fuse_gsq_floats": (c_i64, [P(ModeDims)]),
    "mode_dit_backward": (C.c_int, [P(ModeDims), P(ModeModelWeights), P(ModeModelWeightsT), P(ModeTrainArgs), c_vp, c_vp, P(ModeModelGrads),
                                    c_vp, c_sz, c_vp]),
    "mode_dit_forward": (C.c_int, [P(ModeDims), P(ModeModelWeights), P(ModeForwardArgs), c_vp, c_sz, c_vp]),
    "mode_bn_film_act_fwd": (C.c_int, [P(ModeBnFilmDesc), c_vp]),
    "mode_bn_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "mode_bn_stats": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mode_bn_prepare": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, C.c_float, C.c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz,
                                  c_vp]),
    "mode_bn_prepare_partials": (C.c_int, [c_vp, c_vp, c_i32, C.c_double, c_i32, c_vp, c_vp, C.c_float, C.c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mode_bn_film_act_bwd": (C.c_int, [P(ModeBnFilmDesc), c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mode_stem_conv_fwd": (C.c_int, [P(ModeStemConvDesc), c_vp]),
    "mode_stem_conv_wgrad_slabs": (C.c_int, [P(ModeStemConvDesc)]),
    "mode_stem_conv_wgrad": (C.c_int, [P(ModeStemConvDesc), c_vp]),
    "mode_maxpool_nhwc_fwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "mode_maxpool_nhwc_bwd": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load (once) and type every entry point.  Raises ModeHipUnavailable if the library or any symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ModeHipUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C mode_diffusion_policy_amd/csrc`. There is no CPU fallback for the MoDE denoising path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise ModeHipUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ModeHipUnavailable(f"{LIB_PATH} lacks symbol {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.mode_hip_version() != ABI_VERSION:
        raise ModeHipUnavailable(f"ABI version mismatch: library {lib.mode_hip_version()} != binding {ABI_VERSION}")
    for kv in filter(None, os.environ.get("MODE_HIP_OPTS", "").split(",")):     # tuning knobs of mode_set_option, e.g. "fuse_ln2=0,dn_split_k=1"
        key, _, val = kv.partition("=")
        if lib.mode_set_option(key.strip().encode(), int(val)) != 0:
            raise ValueError(f"MODE_HIP_OPTS: unknown or invalid option {kv!r}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mode_hip_status_string(rc).decode()
        raise RuntimeError(f"libmode_hip {what} failed: {msg} (status {rc})")
