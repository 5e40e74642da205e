"""FiLM-ResNet perceptual encoders on MI355X — the producers of ``perceptual_emb['state_images']``:

* ``FiLMResNet18Policy / 34 / 50``  = mode/models/perceptual_encoders/pretrained_resnets.py:25-137 (a timm ResNet trunk, ``FiLMLayer`` after
  every stage, global average pool -> (B, 512 | 2048) tokens; what ``MoDEAgent`` instantiates twice, mode_agent.py:82-91, 548-567);
* ``ResNetEncoderWithFiLM``         = mode/models/perceptual_encoders/resnets.py:81-200 (torchvision ResNet-18 trunk, ``FilmModule`` ->
  per-block gamma / beta applied after ``bn2``, ``fc`` head).

Same constructor arguments, forward signatures and ``state_dict`` keys (``resnet.conv1.weight``, ``resnet.layer3.1.bn2.running_var``,
``film2.gamma.weight`` ... — the agent's checkpoint loaders match by key, mode_agent.py:132-251), so published weights load unchanged; the
trunks are built here from ``nn.Conv2d`` / ``nn.BatchNorm2d`` holders (neither ``timm`` nor ``torchvision`` is needed at run time —
``pretrained=True`` therefore means "load a checkpoint": there is no network on the box).

Compute: convolutions run on the library's GEMM / implicit-GEMM kernels in bf16 on channels_last data (``_ConvFn`` below; the 3-channel stem and the
max-pool on csrc/conv_stem.hip, ``stem_conv_bn_pool``; an fp32 compute dtype goes through ``torch.nn.functional.conv2d`` = MIOpen - SURVEY: "convs via MIOpen
first" was round 3); everything BETWEEN two
convolutions — BatchNorm (eval or training statistics), the FiLM modulations, the residual add and the ReLU — is ONE hand-written HIP pass over
the activation (``mode_bn_film_act_fwd``; the reference launches 3-6 elementwise kernels there), with a HIP backward
(``mode_bn_film_act_bwd``: one reduction pass + one dx pass, deterministic) behind ``torch.autograd`` so the encoders train through the
denoiser's input gradients (training.py).  No CPU path: the fused op raises off-device.

Parity status: the FiLM wiring is pinned against the REFERENCE classes (oracle/gen_golden_encoders.py runs them on top of a stand-in trunk with
the attribute names their constructors read, fixture F15); the trunk itself is the textbook ResNet restated — timm / torchvision are absent
from the build image, so "trunk == timm's resnet50" rests on the key / shape contract only.  ``nn.SyncBatchNorm`` holders (what Lightning's
``sync_batchnorm=True`` converts the BatchNorm2d modules into, mode/training_calvin.py:102) are honoured: statistics and the backward's channel
sums are all-reduced over the module's process group (two small collectives per BatchNorm and direction).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from .engine import _ptr, _stream


# ------------------------------------------------------------------------------------------------------------------ fused op
def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.MODE_F32
    if t.dtype == torch.bfloat16:
        return L.MODE_BF16
    raise TypeError(f"encoder activations must be float32 or bfloat16, got {t.dtype}")


def _is_cl(x: torch.Tensor) -> bool:
    """torch.channels_last storage of a 4-D activation (and not also plain-contiguous, as 1 x 1 maps / single channels are)."""
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


def _desc(x, scale, shift, pre, res, relu, post, y, cl=False, raw_bn=None) -> L.ModeBnFilmDesc:
    """``raw_bn`` = (weight, bias, running_mean, running_var, eps) instead of scale / shift: the eval-mode BatchNorm is folded inside the pass."""
    N, Cc = x.shape[0], x.shape[1]
    d = L.ModeBnFilmDesc(N=N, C=Cc, HW=x[0, 0].numel(), dtype=_dt(x), x=_ptr(x), scale=_ptr(scale), shift=_ptr(shift),
                         pre_gamma=_ptr(pre[0]) if pre else None, pre_beta=_ptr(pre[1]) if pre else None, residual=_ptr(res), relu=int(relu),
                         post_gamma=_ptr(post[0]) if post else None, post_beta=_ptr(post[1]) if post else None, y=_ptr(y), channels_last=int(cl))
    if raw_bn is not None:
        d.bn_weight, d.bn_bias, d.bn_mean, d.bn_var, d.bn_eps = _ptr(raw_bn[0]), _ptr(raw_bn[1]), _ptr(raw_bn[2]), _ptr(raw_bn[3]), float(raw_bn[4])
    return d


_BN_WS = {}


def _bn_ws_bytes(lib, N, Cc, HW, dt, cl) -> int:
    """mode_bn_workspace_bytes, memoised per geometry (a ctypes call per BatchNorm pass otherwise)."""
    key = (N, Cc, HW, dt, cl)
    v = _BN_WS.get(key)
    if v is None:
        v = _BN_WS[key] = lib.mode_bn_workspace_bytes(N, Cc, HW, dt, cl)
    return v


class _BnFilmAct(torch.autograd.Function):
    """y = post_film(relu(pre_film(batch_norm(x)) + residual)) as one HIP launch (two for training statistics); see csrc/encoder_ops.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, pre_g, pre_b, post_g, post_b, cfg):
        # cfg: everything that is not differentiable, as ONE argument (Function.apply walks its arguments several times: 19 of them were ~2 ms of host
        # time per agent step over the 106 BatchNorms) - (running_mean, running_var, training, momentum, eps, relu, sync_group, num_batches_tracked,
        # grad_enabled, stat_sum, stat_sq)
        running_mean, running_var, training, momentum, eps, relu, sync_group, num_batches_tracked, grad_enabled, stat_sum, stat_sq = cfg
        # ``stat_sum`` / ``stat_sq``: [row blocks, C] partial sums of x that the producing convolution wrote in its epilogue (conv_bn_act): the training
        # statistics are folded from them instead of from another pass over x
        # ``grad_enabled``: torch.is_grad_enabled() at the CALL site (inside Function.forward it is always False, and ctx.needs_input_grad reflects the
        # inputs' requires_grad even under torch.no_grad()): under no_grad nothing will be back-propagated, whatever the parameters say
        wants_grad = grad_enabled and any(ctx.needs_input_grad)
        if x.device.type != "cuda":
            raise L.ModeHipUnavailable("FiLM-ResNet encoders run through the HIP library only: inputs must live on a ROCm device")
        lib = L.load()
        # channels_last activations stay channels_last (the NHWC kernels of encoder_ops.hip; needs whole 16-byte channel vectors); everything else NCHW
        cl = _is_cl(x) and x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4) == 0
        fmt = torch.channels_last if cl else torch.contiguous_format
        x = x.contiguous(memory_format=fmt)
        N, Cc = x.shape[0], x.shape[1]
        HW = x.shape[2] * x.shape[3] if x.dim() == 4 else x[0, 0].numel()
        dev = x.device
        f32 = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()
        fp32_buf = lambda t: t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.device == dev)
        w = weight if fp32_buf(weight) else f32(weight)                   # (only their pointers are used)
        b = bias if fp32_buf(bias) else f32(bias)
        m = N * HW
        if (not training and not wants_grad and running_mean is not None and fp32_buf(running_mean) and fp32_buf(running_var)
                and Cc % 4 == 0):
            # inference (the rollout): the eval-mode BatchNorm is folded INSIDE the pass from the module's own buffers - one launch per BatchNorm
            pre = (f32(pre_g).reshape(N, Cc), f32(pre_b).reshape(N, Cc)) if pre_g is not None else None
            post = (f32(post_g).reshape(N, Cc), f32(post_b).reshape(N, Cc)) if post_g is not None else None
            res = None if residual is None else residual.contiguous(memory_format=fmt)
            if res is not None and (res.shape != x.shape or res.dtype != x.dtype):
                raise ValueError("residual must match the activation's shape and dtype")
            y = torch.empty_like(x, memory_format=fmt)
            L.check(lib.mode_bn_film_act_fwd(C.byref(_desc(x, None, None, pre, res, relu, post, y, cl, raw_bn=(w, b, running_mean, running_var, eps))),
                                             _stream()), "bn_film_act_fwd")
            return y
        if sync_group is None and fp32_buf(running_mean) and fp32_buf(running_var) and (training or running_mean is not None):
            # one call: (row sums +) per-channel statistics, invstd, folded scale / shift, and nn.BatchNorm2d's bookkeeping in place (running
            # statistics with the unbiased variance, num_batches_tracked) - a dozen torch launches per BatchNorm otherwise, 106 BatchNorms per
            # pair of ResNet-50s
            mean, var, invstd, scale, shift = torch.empty(5, Cc, device=dev).unbind(0)
            ws = torch.empty(_bn_ws_bytes(lib, N, Cc, HW, _dt(x), int(cl)), dtype=torch.uint8, device=dev) if (training and stat_sum is None) else None
            nbt = num_batches_tracked if (training and num_batches_tracked is not None and num_batches_tracked.dtype == torch.int64
                                          and num_batches_tracked.device == dev) else None
            if training and stat_sum is not None:                              # (Function.forward runs with grad mode off: no no_grad() context needed)
                L.check(lib.mode_bn_prepare_partials(stat_sum.data_ptr(), stat_sq.data_ptr(), stat_sum.shape[0], float(m), Cc, _ptr(w), _ptr(b), float(eps),
                                                     -1.0 if momentum is None else float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                                     mean.data_ptr(), var.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), _stream()),
                        "bn_prepare_partials")
            else:
                L.check(lib.mode_bn_prepare(x.data_ptr() if training else None, _dt(x), N, Cc, HW, int(cl), _ptr(w), _ptr(b), float(eps),
                                            -1.0 if momentum is None else float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(nbt), mean.data_ptr(),
                                            var.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), _ptr(ws), 0 if ws is None else ws.numel(),
                                            _stream()), "bn_prepare")
        else:
            ws = torch.empty(_bn_ws_bytes(lib, N, Cc, HW, _dt(x), int(cl)), dtype=torch.uint8, device=dev)
            if training:
                mean = torch.empty(Cc, device=dev); var = torch.empty(Cc, device=dev)
                L.check(lib.mode_bn_stats(x.data_ptr(), _dt(x), N, Cc, HW, int(cl), mean.data_ptr(), var.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "bn_stats")
                if sync_group is not None:
                    # nn.SyncBatchNorm (Lightning's sync_batchnorm=True, mode/training_calvin.py:102): the statistics of the GLOBAL batch - one small
                    # all-reduce of [sum, sum of squares, count] per BatchNorm (RCCL; 2C + 1 floats)
                    import torch.distributed as dist
                    pack = torch.cat([mean * m, (var + mean * mean) * m, torch.full((1,), float(m), device=dev)])
                    dist.all_reduce(pack, group=sync_group if sync_group is not True else None)
                    m = pack[-1]                                               # global element count per channel: stays on the device (no host sync)
                    mean = pack[:Cc] / m
                    var = (pack[Cc:2 * Cc] / m - mean * mean).clamp_min_(0.0)
                if running_mean is not None:                                   # nn.BatchNorm2d bookkeeping: unbiased variance into the running estimate
                    with torch.no_grad():
                        if num_batches_tracked is not None:
                            num_batches_tracked.add_(1)
                        f = (1.0 / num_batches_tracked.to(torch.float32)) if momentum is None else momentum
                        unbias = (m / (m - 1).clamp_min(1.0)) if torch.is_tensor(m) else (m / max(m - 1, 1))
                        running_mean.mul_(1 - f).add_((mean * f).to(running_mean.dtype))
                        running_var.mul_(1 - f).add_((var * unbias * f).to(running_var.dtype))
            else:
                mean, var = f32(running_mean), f32(running_var)
            invstd = torch.rsqrt(var + eps)
            w1 = w if w is not None else torch.ones(Cc, device=dev)
            scale = w1 * invstd
            shift = (b if b is not None else 0.0) - mean * scale
        pre = (f32(pre_g).reshape(N, Cc), f32(pre_b).reshape(N, Cc)) if pre_g is not None else None
        post = (f32(post_g).reshape(N, Cc), f32(post_b).reshape(N, Cc)) if post_g is not None else None
        res = None if residual is None else residual.contiguous(memory_format=fmt)
        if res is not None and (res.shape != x.shape or res.dtype != x.dtype):
            raise ValueError("residual must match the activation's shape and dtype")
        y = torch.empty_like(x, memory_format=fmt)
        L.check(lib.mode_bn_film_act_fwd(C.byref(_desc(x, scale, shift, pre, res, relu, post, y, cl)), _stream()), "bn_film_act_fwd")
        ctx.save_for_backward(x, res, scale, shift, mean, invstd, *(pre or ()), *(post or ()))
        ctx.cfg = (bool(training), bool(relu), pre is not None, post is not None, res is not None, cl)
        ctx.sync = (sync_group, m) if (training and sync_group is not None) else None
        ctx.shapes = (None if pre_g is None else pre_g.shape, None if post_g is None else post_g.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        training, relu, has_pre, has_post, has_res, cl = ctx.cfg
        fmt = torch.channels_last if cl else torch.contiguous_format
        sv = list(ctx.saved_tensors)
        x, res, scale, shift, mean, invstd = sv[:6]
        rest = sv[6:]
        pre = (rest[0], rest[1]) if has_pre else None
        post = (rest[2 if has_pre else 0], rest[3 if has_pre else 1]) if has_post else None
        N, Cc = x.shape[0], x.shape[1]
        dev = x.device
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if not dy.is_contiguous(memory_format=fmt):
            dy = dy.contiguous(memory_format=fmt)
        dx = torch.empty_like(x, memory_format=fmt)
        dres = torch.empty_like(x, memory_format=fmt) if has_res else None
        dw = torch.empty(Cc, device=dev); db = torch.empty(Cc, device=dev)      # (leaf gradients: own tensors, so that AccumulateGrad can take them without a copy)
        dpg, dpb = torch.empty(2, N, Cc, device=dev).unbind(0) if has_pre else (None, None)
        dqg, dqb = torch.empty(2, N, Cc, device=dev).unbind(0) if has_post else (None, None)
        HW = x.shape[2] * x.shape[3] if x.dim() == 4 else x[0, 0].numel()
        ws = torch.empty(_bn_ws_bytes(lib, N, Cc, HW, _dt(x), int(cl)), dtype=torch.uint8, device=dev)
        d = _desc(x, scale, shift, pre, res if has_res else None, relu, post, None, cl)
        d.y = x.data_ptr()                                                     # unused by the backward; the descriptor check wants a pointer
        def call(phase, inv_count, dw_, db_):
            L.check(lib.mode_bn_film_act_bwd(C.byref(d), dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), int(training), phase, float(inv_count), dx.data_ptr(),
                                             _ptr(dres), dw_.data_ptr(), db_.data_ptr(), _ptr(dpg), _ptr(dpb), _ptr(dqg), _ptr(dqb), ws.data_ptr(), ws.numel(),
                                             _stream()), "bn_film_act_bwd")
        if ctx.sync is None:
            call(0, 0.0, dw, db)
        else:                                                                  # SyncBatchNorm: channel sums over ALL ranks feed dx; dweight / dbias stay this rank's
            import torch.distributed as dist
            group, m_global = ctx.sync                                      # m_global: device scalar
            call(1, 0.0, dw, db)
            tot = torch.stack([dw, db])
            dist.all_reduce(tot, group=group if group is not True else None)
            tot = tot / m_global                                            # the global MEANS, so that phase 2 runs with inv_count = 1
            call(2, 1.0, tot[0].contiguous(), tot[1].contiguous())
        ps, qs = ctx.shapes
        rs = lambda t, shp: None if t is None else t.reshape(shp)
        # inputs: x, weight, bias, residual, pre_g, pre_b, post_g, post_b, cfg
        return dx, dw, db, dres, rs(dpg, ps), rs(dpb, ps), rs(dqg, qs), rs(dqb, qs), None


# Activation layout inside the encoders.  True: torch.channels_last - MIOpen's implicit-GEMM convolutions run on NHWC data and wrap NCHW tensors in
# layout transposes (10 % of the agent's training step, scripts/step_kernel_profile.sh); the fused pass has kernels for both layouts.  The encoders'
# inputs and outputs are ordinary tensors either way (the layout is a stride pattern, not a shape).
CHANNELS_LAST = True


def _to_layout(x: torch.Tensor) -> torch.Tensor:
    return x.contiguous(memory_format=torch.channels_last) if (CHANNELS_LAST and x.dim() == 4) else x


def _store_channels_last(module: nn.Module) -> None:
    """Convolution weights of ``module`` into channels_last STORAGE (same values, shapes and state_dict).  Done at construction, i.e. before anything
    (torch DDP's bucket views, an optimizer's state) has looked at the parameters' strides."""
    if not CHANNELS_LAST:
        return
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, nn.Conv2d) and not m.weight.is_contiguous(memory_format=torch.channels_last):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)


# {id(conv): weight already in the compute dtype}: set by GraphedVisualEncoder around its warm-up and capture, so that the per-call weight casts of
# autocast (one launch per convolution) are not part of the replayed graph; it refreshes the copies in place when a weight's version moves.
# Process-wide, set and restored around one call (not re-entrant: stream capture is not either).
_W_OVERRIDE: Optional[dict] = None


def _compute_dtype(x: torch.Tensor) -> torch.dtype:
    return torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype


# ---- convolutions on the library's own kernels (round 4; LABNOTES.md section 4, profiles/r04_conv_probes.txt).  On channels_last bf16 activations every convolution
# is a GEMM over rows = pixels, and the library already had the GEMMs; MIOpen (round 3: "convs via MIOpen first") cost 37 % of the agent's step - weight-gradient
# kernels at ~100 TF/s wrapped in zero / cast helper launches, split-K forward / data-gradient kernels with the same helpers, two autocast casts per convolution.
#   1 x 1 / stride 1   forward = mode_gemm, data gradient = MODE_GEMM_W_KN, weight gradient dW[Cout, Cin] = dY[R, Cout]^T X[R, Cin] = MODE_GEMM_A_KM | W_KN with
#                      K = pixels cut into groups (fp32 partial sums added in group order: deterministic)
#   k x k / strided    forward and data gradient as implicit GEMM (csrc/conv_gemm.hip: `a_rows` in taps - the A tile of a K-step is gathered from the rows its
#                      filter tap pairs with the output pixels, -1 = outside the image), weight gradient as ONE product over the taps (`w_rows` in taps)
#   3-channel stem     csrc/conv_stem.hip (`stem_conv_bn_pool` below): the im2col tile of 8 x 16 output pixels is built on chip from the image as it lies
# `_ConvFn` hands autograd fp32 weight gradients directly and reads a compute-dtype SHADOW of the weight that is refreshed when the parameter's version moves (all
# stale shadows of an encoder in one `_foreach_copy_`).  Inference takes the same forward kernels, and `conv_bn_act` folds the eval-mode BatchNorm / FiLM / residual
# / ReLU into the convolution's epilogue (mode_conv_bn_act_fwd).  MODE_ENC_HIPCONV=0 restores F.conv2d with per-call casts everywhere (A/B runs).
USE_HIP_CONV_WGRAD = __import__("os").environ.get("MODE_ENC_HIPCONV", "1") == "1"     # MODE_ENC_HIPCONV=0: A/B runs


def _is_1x1(wshape, stride, padding, cin_mult: int = 64) -> bool:
    return wshape[2] == 1 and wshape[3] == 1 and tuple(stride) == (1, 1) and tuple(padding) == (0, 0) and wshape[1] % cin_mult == 0 and wshape[0] % 64 == 0


def _gemm_1x1_fwd(x: torch.Tensor, w_lp: torch.Tensor) -> torch.Tensor:
    """1 x 1 / stride-1 convolution on channels_last bf16 data = Y[R, Cout] = X[R, Cin] W[Cout, Cin]^T: the library's forward GEMM (scripts/conv1x1_probe.py:
    12-27 us where MIOpen's implicit-GEMM kernels + their split-K helpers take 20-100 us at the ResNet-50 shapes)."""
    B, cin, H, W_ = x.shape
    cout = w_lp.shape[0]
    y = torch.empty((B, cout, H, W_), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=B * H * W_, N=cout, K=cin, A=x.data_ptr(), lda=cin, W=w_lp.data_ptr(), ldw=cin,
                       C=y.data_ptr(), ldc=cout)
    L.check(L.load().mode_gemm(C.byref(d), _stream()), "conv 1x1 forward")
    return y


def _gemm_1x1_dgrad(dy: torch.Tensor, w_lp: torch.Tensor, xshape) -> torch.Tensor:
    """dX[R, Cin] = dY[R, Cout] W[Cout, Cin]: the data-gradient GEMM on the [out, in] weight where it lies (MODE_GEMM_W_KN)."""
    B, cin, H, W_ = xshape
    cout = w_lp.shape[0]
    dx = torch.empty((B, cin, H, W_), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=B * H * W_, N=cin, K=cout, A=dy.data_ptr(), lda=cout, W=w_lp.data_ptr(), ldw=cin,
                       C=dx.data_ptr(), ldc=cin, flags=L.GEMM_W_KN)
    L.check(L.load().mode_gemm(C.byref(d), _stream()), "conv 1x1 data gradient")
    return dx


def _wgrad_1x1(dy: torch.Tensor, x: torch.Tensor, wshape) -> torch.Tensor:
    """dW of a 1 x 1 / stride-1 convolution from channels_last bf16 activations: [Cout, Cin, 1, 1] fp32."""
    cout, cin = wshape[0], wshape[1]
    R = dy.shape[0] * dy.shape[2] * dy.shape[3]
    tiles = ((cout + 127) // 128) * ((cin + 127) // 128)
    G = max(1, min(R // 256, 768 // tiles, max(4, (8 << 20) // (cout * cin * 4))))      # ~3 workgroups per CU, >= 4 K-steps per group, <= 8 MiB of partial sums
    key = ("koffs", R, G, dy.device)
    offs = _TABLES.get(key)
    if offs is None:
        offs = _TABLES.put(key, torch.tensor([(R * i) // G for i in range(G + 1)], dtype=torch.int32, device=dy.device))
    part = torch.empty((G, cout, cin), dtype=torch.float32, device=dy.device)
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=cout, N=cin, K=R, A=dy.data_ptr(), lda=cout, W=x.data_ptr(), ldw=cin,
                       C=part.data_ptr(), ldc=cin, k_group_offsets=offs.data_ptr(), num_k_groups=G, c_group_stride=cout * cin,
                       flags=L.GEMM_W_KN | L.GEMM_A_KM)
    L.check(L.load().mode_gemm(C.byref(d), _stream()), "conv 1x1 weight gradient")
    return (part.sum(0) if G > 1 else part[0]).view(cout, cin, 1, 1)


class _TableCache:
    """Byte-bounded LRU for the int32 index tables of the implicit-GEMM convolutions (tap tables: ~0.35 MB per sample and distinct batch size for a
    ResNet-50 at 224 x 224; K-group offsets: bytes).  A ragged last batch or a varying environment count would otherwise grow device memory without
    bound.  Eviction only drops the CACHE's reference: a table in use by an in-flight autograd graph stays alive through `ctx`, and a captured
    hipGraph - which holds raw pointers - pins its tables through :attr:`sink` (GraphedVisualEncoder sets it around warm-up + capture and keeps the
    list next to the graph).  MODE_ENC_TABLE_CACHE_MB (default 256) sets the bound."""

    def __init__(self):
        import collections
        import os
        self.d = collections.OrderedDict()
        self.bytes = 0
        self.limit = int(float(os.environ.get("MODE_ENC_TABLE_CACHE_MB", "256")) * (1 << 20))
        self.sink = None                                                       # a list while a GraphedVisualEncoder warms up / captures

    def get(self, key):
        t = self.d.get(key)
        if t is not None:
            self.d.move_to_end(key)
            if self.sink is not None:
                self.sink.append(t)
        return t

    def put(self, key, t):
        if TWO_TOWER_STREAMS and t.is_cuda and not torch.cuda.is_current_stream_capturing():
            # the other tower's stream may use the table right away: built before it is published.  (Under hipGraph capture a host synchronize is illegal - and
            # not needed: both towers' launches are captured in fork / join order behind the build.)
            torch.cuda.current_stream(t.device).synchronize()
        self.d[key] = t
        self.bytes += t.numel() * t.element_size()
        if self.sink is not None:
            self.sink.append(t)
        while self.bytes > self.limit and len(self.d) > 1:
            _, old = self.d.popitem(last=False)
            self.bytes -= old.numel() * old.element_size()
        return t

    def __len__(self):
        return len(self.d)


_TABLES = _TableCache()


def _tap_table(n, H, W_, ho, wo, kh_, kw_, sh, sw, ph, pw, dev, transposed: bool = False) -> torch.Tensor:
    """int32 [kh*kw, rows]: forward table (rows = output pixels): the INPUT row that tap (a, b) pairs with output pixel (n, h, w); transposed (rows = input
    pixels): the OUTPUT row whose tap (a, b) read input pixel (n, h, w).  -1 where there is none (outside the image / between the strides).  Cached
    (LRU, `_TableCache`)."""
    key = ("taps", n, H, W_, kh_, kw_, sh, sw, ph, pw, dev, transposed)
    idx = _TABLES.get(key)
    if idx is None:
        # one entry per (batch, image size, filter geometry): a run with one batch size has ~20
        nn_ = torch.arange(n, device=dev).view(n, 1, 1)
        tabs = []
        neg = torch.full((), -1, device=dev)
        if not transposed:
            hh = torch.arange(ho, device=dev).view(1, ho, 1); ww = torch.arange(wo, device=dev).view(1, 1, wo)
            for a in range(kh_):
                for b in range(kw_):
                    hi, wi = hh * sh + a - ph, ww * sw + b - pw
                    ok = (hi >= 0) & (hi < H) & (wi >= 0) & (wi < W_)
                    tabs.append(torch.where(ok, (nn_ * H + hi) * W_ + wi, neg).reshape(-1))
        else:
            hh = torch.arange(H, device=dev).view(1, H, 1); ww = torch.arange(W_, device=dev).view(1, 1, W_)
            for a in range(kh_):
                for b in range(kw_):
                    hn, wn_ = hh + ph - a, ww + pw - b
                    h2, w2 = torch.div(hn, sh, rounding_mode="floor"), torch.div(wn_, sw, rounding_mode="floor")
                    ok = (hn >= 0) & (wn_ >= 0) & (hn % sh == 0) & (wn_ % sw == 0) & (h2 < ho) & (w2 < wo)
                    tabs.append(torch.where(ok, (nn_ * ho + h2) * wo + w2, neg).reshape(-1))
        idx = _TABLES.put(key, torch.stack(tabs).to(torch.int32).contiguous())
    return idx


def _conv_taps_ok(wshape, k_per_tap: int, w_lp: Optional[torch.Tensor] = None) -> bool:
    """Implicit-GEMM path: 64-channel K-steps inside one tap, and the weight must lie as [Cout][kh][kw][Cin] (channels_last storage)."""
    return k_per_tap % 64 == 0 and wshape[0] % 8 == 0 and wshape[1] % 8 == 0 and (w_lp is None or w_lp.is_contiguous(memory_format=torch.channels_last))


def _conv_fwd_taps(x: torch.Tensor, w_lp: torch.Tensor, stride, padding) -> torch.Tensor:
    """k x k convolution forward as one implicit-GEMM launch (mode_gemm with a_rows in taps, csrc/conv_gemm.hip): Y[R_out, Cout] = sum_t X[idx[t], :] W[:, t, :]^T."""
    n, cin, H, W_ = x.shape
    cout, _, kh_, kw_ = w_lp.shape
    ho = (H + 2 * padding[0] - kh_) // stride[0] + 1; wo = (W_ + 2 * padding[1] - kw_) // stride[1] + 1
    idx = _tap_table(n, H, W_, ho, wo, kh_, kw_, stride[0], stride[1], padding[0], padding[1], x.device)
    y = torch.empty((n, cout, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    R = n * ho * wo
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=R, N=cout, K=kh_ * kw_ * cin, A=x.data_ptr(), lda=cin, W=w_lp.data_ptr(),
                       ldw=kh_ * kw_ * cin, C=y.data_ptr(), ldc=cout, a_rows=idx.data_ptr(), a_tap_cols=cin, a_rows_tap_stride=R)
    L.check(L.load().mode_gemm(C.byref(d), _stream()), "conv forward (taps)")
    return y


def _conv_dgrad_taps(dy: torch.Tensor, w_lp: torch.Tensor, xshape, stride, padding) -> torch.Tensor:
    """dX[R_in, Cin] = sum_t dY[idx_T[t], :] W[:, t, :] - the same weight memory read as [k = (t, cout)][cin] (MODE_GEMM_W_KN with a_rows in taps)."""
    n, cin, H, W_ = xshape
    cout, _, kh_, kw_ = w_lp.shape
    ho, wo = dy.shape[2], dy.shape[3]
    idx = _tap_table(n, H, W_, ho, wo, kh_, kw_, stride[0], stride[1], padding[0], padding[1], dy.device, transposed=True)
    dx = torch.empty((n, cin, H, W_), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    R = n * H * W_
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=R, N=cin, K=kh_ * kw_ * cout, A=dy.data_ptr(), lda=cout, W=w_lp.data_ptr(),
                       ldw=kh_ * kw_ * cin, C=dx.data_ptr(), ldc=cin, a_rows=idx.data_ptr(), a_tap_cols=cout, a_rows_tap_stride=R, flags=L.GEMM_W_KN)
    L.check(L.load().mode_gemm(C.byref(d), _stream()), "conv data gradient (taps)")
    return dx


def _conv_fwd_stats(x: torch.Tensor, w_lp: torch.Tensor, stride, padding):
    """Training forward of a convolution that feeds a BatchNorm: y and, from the same launch's epilogue, the per-128-row-tile column sums / sums of squares of y as
    stored (mode_conv_bn_act_fwd without epilogue terms + stat_sum / stat_sq) - the BatchNorm folds its batch statistics from them instead of re-reading y."""
    n, cin, H, W_ = x.shape
    cout, _, kh_, kw_ = w_lp.shape
    sh, sw = stride; ph, pw = padding
    ho = (H + 2 * ph - kh_) // sh + 1; wo = (W_ + 2 * pw - kw_) // sw + 1
    one = kh_ == 1 and kw_ == 1 and (sh, sw) == (1, 1) and (ph, pw) == (0, 0)
    idx = None if one else _tap_table(n, H, W_, ho, wo, kh_, kw_, sh, sw, ph, pw, x.device)
    R = n * ho * wo
    y = torch.empty((n, cout, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    st = torch.empty((2, (R + 127) // 128, cout), dtype=torch.float32, device=x.device)
    d = L.ModeConvBnDesc(x=x.data_ptr(), ldx=cin, idx=_ptr(idx), idx_tap_stride=R, taps=kh_ * kw_, w=w_lp.data_ptr(), ldw=kh_ * kw_ * cin, y=y.data_ptr(), ldy=cout,
                         M=R, Cin=cin, Cout=cout, relu=0, rows_per_sample=ho * wo, stat_sum=st[0].data_ptr(), stat_sq=st[1].data_ptr())
    L.check(L.load().mode_conv_bn_act_fwd(C.byref(d), _stream()), "conv forward + BatchNorm partial statistics")
    return y, st[0], st[1]


def _wgrad_taps(dy: torch.Tensor, x: torch.Tensor, wshape, stride, padding) -> torch.Tensor:
    """dW of a k x k convolution (any stride, zero padding) from channels_last bf16 activations as k*k weight-gradient GEMMs, one per filter tap:
    dW[:, :, kh, kw] = dY[R_out, Cout]^T X[rows(kh, kw), Cin] - the input rows that tap (kh, kw) pairs with the output pixels (a zero row where the tap falls
    outside the image), gathered inside the GEMM's DMA through an index table (mode_gemm `w_rows`, -1 = zero row; cached per geometry).  Each tap writes its [Cout, Cin] slice of the channels_last
    gradient ([Cout][kh][kw][Cin] in memory) directly: C = base + tap * Cin, ldc = k*k*Cin.  fp32 [Cout, Cin, k, k], channels_last."""
    cout, cin, kh_, kw_ = wshape
    n, _, H, W_ = x.shape
    ho, wo = dy.shape[2], dy.shape[3]
    ph, pw = padding
    sh, sw = stride
    R = n * ho * wo
    idx = _tap_table(n, H, W_, ho, wo, kh_, kw_, sh, sw, ph, pw, x.device)
    xp = x
    taps = kh_ * kw_
    one_launch = cin % 64 == 0                                   # all taps as ONE product (N = taps * Cin, ABI 10 `w_tap_cols`): the dY tiles are shared through L2
    tiles = ((cout + 127) // 128) * ((cin + 127) // 128) * taps
    G = max(1, min(R // 256, 512 // tiles, max(4, (8 << 20) // (cout * cin * taps * 4))))      # scripts/conv_wgrad_probe.py: 40-80 groups at 64 / 128 channels, 10-20 at 256, ~4 at 512; <= 8 MiB of partial sums
    okey = ("koffs", R, G, dy.device)
    offs = _TABLES.get(okey)
    if offs is None:
        offs = _TABLES.put(okey, torch.tensor([(R * i) // G for i in range(G + 1)], dtype=torch.int32, device=dy.device))
    part = torch.empty((G, cout, kh_, kw_, cin), dtype=torch.float32, device=dy.device)       # channels_last order of [Cout, Cin, kh, kw]
    lib = L.load(); st = _stream()
    if one_launch:
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=cout, N=taps * cin, K=R, A=dy.data_ptr(), lda=cout, W=xp.data_ptr(), ldw=cin,
                           C=part.data_ptr(), ldc=taps * cin, k_group_offsets=offs.data_ptr(), num_k_groups=G, c_group_stride=cout * taps * cin,
                           w_rows=idx.data_ptr(), w_tap_cols=cin, w_rows_tap_stride=R, flags=L.GEMM_W_KN | L.GEMM_A_KM)
        L.check(lib.mode_gemm(C.byref(d), st), "conv weight gradient (taps)")
    else:
        for t in range(taps):
            d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=cout, N=cin, K=R, A=dy.data_ptr(), lda=cout, W=xp.data_ptr(), ldw=cin,
                               C=part.data_ptr() + t * cin * 4, ldc=taps * cin, k_group_offsets=offs.data_ptr(), num_k_groups=G, c_group_stride=cout * taps * cin,
                               w_rows=idx[t].data_ptr(), flags=L.GEMM_W_KN | L.GEMM_A_KM)
            L.check(lib.mode_gemm(C.byref(d), st), "conv weight gradient (tap GEMM)")
    dw = part.sum(0) if G > 1 else part[0]
    return dw.permute(0, 3, 1, 2)                                  # [Cout, Cin, kh, kw] view with channels_last strides


def _aten_conv(x, w, stride, padding, why: str):
    """The ONLY way into ``F.conv2d`` (MIOpen on a ROCm device).  CPU tensors (parameter-holder use, the CPU tests' reference) and the fp32 compute dtype (the
    reference-precision / debugging mode: the library's convolutions are bf16 MFMA kernels) take it by design; the explicit A/B switches MODE_ENC_HIPCONV=0 /
    MODE_ENC_HIPSTEM=0 too.  A bf16 convolution on a ROCm device that the library's kernels do not cover (groups, dilation, a bias, channel counts that are not
    multiples of 8, a layout other than channels_last) is an ERROR, not a silent change of backend - unless MODE_ENC_ATEN_FALLBACK=1 says so."""
    import os
    if x.is_cuda and x.dtype == torch.bfloat16 and USE_HIP_CONV_WGRAD and USE_HIP_STEM and os.environ.get("MODE_ENC_ATEN_FALLBACK", "0") != "1":
        raise RuntimeError(f"perceptual_encoders: bf16 convolution outside the library's kernels ({why}; weight {tuple(w.shape)}, input {tuple(x.shape)}, "
                           f"channels_last={x.is_contiguous(memory_format=torch.channels_last)}): set MODE_ENC_ATEN_FALLBACK=1 to run it through aten / MIOpen")
    return F.conv2d(x, w, None, stride, padding)


class _ConvFn(torch.autograd.Function):
    """y = conv2d(x, w) computed with `w_lp` (w in the compute dtype); differentiable in x and in the fp32 PARAMETER w."""

    @staticmethod
    def forward(ctx, x, w, w_lp, stride, padding, want_stats=False):
        ctx.save_for_backward(x, w_lp)
        ctx.conf = (tuple(stride), tuple(padding), tuple(w.shape), w.dtype)
        if want_stats:                                            # (the caller checked the kernel's contract) y + the partial BatchNorm statistics of y
            y, ps, pq = _conv_fwd_stats(x, w_lp, stride, padding)
            ctx.mark_non_differentiable(ps, pq)
            return y, ps, pq
        if x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last):
            if _is_1x1(w.shape, stride, padding):
                return _gemm_1x1_fwd(x, w_lp)
            if _conv_taps_ok(w.shape, w.shape[1], w_lp):
                return _conv_fwd_taps(x, w_lp, stride, padding)
        return _aten_conv(x, w_lp, stride, padding, "training forward")

    @staticmethod
    def backward(ctx, dy, *_unused):
        x, w_lp = ctx.saved_tensors
        stride, padding, wshape, wdtype = ctx.conf
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dw = None
        if (need_w and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and wshape[0] % 8 == 0 and wshape[1] % 8 == 0
                and x.is_contiguous(memory_format=torch.channels_last)):
            dyc = dy.contiguous(memory_format=torch.channels_last)
            if wshape[2] == 1 and wshape[3] == 1 and stride == (1, 1) and padding == (0, 0):
                dw = _wgrad_1x1(dyc, x, wshape)
            else:
                dw = _wgrad_taps(dyc, x, wshape, stride, padding)
            need_w = False
        dx = None
        if need_x and dy.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last):
            if _is_1x1(wshape, stride, padding, 8):
                dx = _gemm_1x1_dgrad(dy.contiguous(memory_format=torch.channels_last), w_lp, x.shape)
                need_x = False
            elif _conv_taps_ok(wshape, wshape[0], w_lp):
                dx = _conv_dgrad_taps(dy.contiguous(memory_format=torch.channels_last), w_lp, x.shape, stride, padding)
                need_x = False
        if need_x or need_w:
            dxl, dwl, _ = torch.ops.aten.convolution_backward(dy, x, w_lp, None, stride, padding, (1, 1), False, (0, 0), 1, (need_x, need_w, False))
            if need_x:
                dx = dxl
            if need_w:
                dw = dwl.to(wdtype)
        return dx, dw, None, None, None, None


def _wver(conv: nn.Conv2d):
    """What a cached copy of `conv.weight` is valid for: torch's version counter + the module's invalidation generation (invalidate_conv_shadows)."""
    return (conv.weight._version, conv.__dict__.get("_mode_wgen", 0))


def _shadow(conv: nn.Conv2d, dtype: torch.dtype) -> torch.Tensor:
    """The convolution's weight in the compute dtype, cached on the module (not a buffer: never in a state_dict) and refreshed in place when the
    parameter changes (its version counter / storage moved).  Staleness rests on torch's version counter: optimizers, `load_state_dict`, `copy_` and every
    other in-place torch op bump it; code that writes a parameter through a raw pointer must bump it too (`torch.autograd.graph.increment_version`)."""
    w = conv.weight
    ent = conv.__dict__.get("_mode_lp")
    if ent is None or ent[2].dtype != dtype or ent[2].device != w.device or ent[2].shape != w.shape:
        ent = conv.__dict__["_mode_lp"] = [-1, 0, torch.empty_like(w, dtype=dtype)]
    if ent[0] != _wver(conv) or ent[1] != w.data_ptr():
        with torch.no_grad():
            ent[2].copy_(w)
        ent[0], ent[1] = _wver(conv), w.data_ptr()
    return ent[2]


def refresh_conv_shadows(module: nn.Module, dtype: torch.dtype, force: bool = False) -> None:
    """The compute-dtype weight shadows of `module`'s convolutions in ONE multi-tensor copy (instead of one cast launch per convolution).
    ``force=False`` copies only the shadows whose parameter's version counter / storage moved.  ``force=True`` copies all of them: the encoders'
    TRAINING forward (grad mode) does that once per call, because a write through ``p.data`` (``p.data.copy_(ema)``, ``p.data.mul_()``,
    ``w.data.normal_()``) does NOT bump ``p._version`` and a version-gated cache would silently keep computing with the old weights in both the
    forward and the backward (53 tensors, one `_foreach_copy_`: ~20 us beside a 30-ms step).  Inference (no-grad, the captured graphs) stays
    version-gated - code that writes weights through ``.data`` there calls :func:`invalidate_conv_shadows` (the checkpoint loader does)."""
    convs = module.__dict__.get("_mode_convs")
    if convs is None:
        convs = module.__dict__["_mode_convs"] = [m for m in module.modules() if isinstance(m, nn.Conv2d)]
    src, dst = [], []
    for c in convs:
        w = c.weight
        ent = c.__dict__.get("_mode_lp")
        if ent is None or ent[2].dtype != dtype or ent[2].device != w.device or ent[2].shape != w.shape:
            ent = c.__dict__["_mode_lp"] = [-1, 0, torch.empty_like(w, dtype=dtype)]
        if force or ent[0] != _wver(c) or ent[1] != w.data_ptr():
            src.append(w.detach()); dst.append(ent[2])
            ent[0], ent[1] = _wver(c), w.data_ptr()
    if dst:
        with torch.no_grad():
            torch._foreach_copy_(dst, src)


def invalidate_conv_shadows(module: nn.Module) -> None:
    """Mark every cached compute-dtype convolution weight under `module` stale (eager shadows AND the copies `GraphedVisualEncoder` replays
    against): the next forward re-casts them in place.  Call it after writing weights in a way torch's version counter does not see -
    ``p.data.copy_() / p.data.mul_()`` (EMA swaps), raw-pointer writes.  In-place ops on the Parameter itself, optimizers and
    ``load_state_dict`` bump the counter and need no call."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            m.__dict__["_mode_wgen"] = m.__dict__.get("_mode_wgen", 0) + 1


def _conv2d(conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """conv2d through MIOpen with the module's weight in the activations' dtype and layout.  The PARAMETER's storage is converted to channels_last
    once (values, shape and state_dict unchanged), so no per-call weight transposes are left."""
    w = conv.weight
    if CHANNELS_LAST and w.dim() == 4 and not w.is_contiguous(memory_format=torch.channels_last):
        with torch.no_grad():
            w.data = w.data.contiguous(memory_format=torch.channels_last)
    cd = _compute_dtype(x)
    hip_ok = (USE_HIP_CONV_WGRAD and x.is_cuda and cd == torch.bfloat16 and conv.groups == 1 and conv.dilation == (1, 1) and conv.bias is None
              and isinstance(conv.padding, tuple))
    hip_1x1 = hip_ok and _is_1x1(w.shape, conv.stride, conv.padding)
    hip_taps = hip_ok and not hip_1x1 and _conv_taps_ok(w.shape, w.shape[1])
    if _W_OVERRIDE is not None:
        hit = _W_OVERRIDE.get(id(conv))
        if hit is not None and hit.dtype == cd:
            xc = x.to(hit.dtype)
            if hip_1x1 and xc.is_contiguous(memory_format=torch.channels_last):
                return _gemm_1x1_fwd(xc, hit)
            if hip_taps and xc.is_contiguous(memory_format=torch.channels_last) and hit.is_contiguous(memory_format=torch.channels_last):
                return _conv_fwd_taps(xc, hit, conv.stride, conv.padding)
            return _aten_conv(xc, hit, conv.stride, conv.padding, "captured weight shadow")
    if (hip_1x1 or hip_taps) and not (torch.is_grad_enabled() and (w.requires_grad or x.requires_grad)):
        xc = x.to(cd)
        if xc.is_contiguous(memory_format=torch.channels_last):                    # inference: the shadow also saves the per-call weight cast
            return _gemm_1x1_fwd(xc, _shadow(conv, cd)) if hip_1x1 else _conv_fwd_taps(xc, _shadow(conv, cd), conv.stride, conv.padding)
    if (USE_HIP_CONV_WGRAD and x.is_cuda and cd == torch.bfloat16 and torch.is_grad_enabled() and (w.requires_grad or x.requires_grad) and conv.groups == 1
            and conv.dilation == (1, 1) and conv.bias is None and isinstance(conv.padding, tuple)):
        return _ConvFn.apply(x.to(cd), w, _shadow(conv, cd), conv.stride, conv.padding)
    return _aten_conv(x, w.to(x.dtype), conv.stride, conv.padding, "unsupported geometry / layout")


def bn_film_act(x, bn: nn.BatchNorm2d, relu: bool = True, residual=None, pre_film=None, post_film=None, stats=None):
    """``post_film(relu(pre_film(bn(x)) + residual))``; ``pre_film`` / ``post_film`` = (gamma, beta), each (N, C) (or broadcastable views of it)."""
    pg, pb = pre_film if pre_film is not None else (None, None)
    qg, qb = post_film if post_film is not None else (None, None)
    sync = None
    if isinstance(bn, nn.SyncBatchNorm) and bn.training:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(bn.process_group) > 1:
            sync = bn.process_group if bn.process_group is not None else True
    training = bn.training or bn.running_mean is None                       # no running statistics -> batch statistics also in eval (nn.BatchNorm2d)
    s0, s1 = (stats[0], stats[1]) if (stats is not None and sync is None) else (None, None)
    return _BnFilmAct.apply(x, bn.weight, bn.bias, residual, pg, pb, qg, qb,
                            (bn.running_mean, bn.running_var, training, bn.momentum, bn.eps, relu, sync, bn.num_batches_tracked if bn.training else None,
                             torch.is_grad_enabled(), s0, s1))


# Inference: convolution + eval-mode BatchNorm + FiLM + residual + ReLU as ONE launch (mode_conv_bn_act_fwd, csrc/conv_gemm.hip) - the rollout's encoders run
# half the kernels and the convolution output never goes to memory un-normalised.  MODE_ENC_FUSE_CONV_BN=0: the two launches (A/B runs, tests).
FUSE_CONV_BN = __import__("os").environ.get("MODE_ENC_FUSE_CONV_BN", "1") == "1"
# training: BatchNorm partial statistics from the convolution's epilogue (ModeConvBnDesc.stat_sum / stat_sq + mode_bn_prepare_partials).  OFF by default: measured in
# the agent's step it does not pay - 33.2-33.4 (k x k convolutions only) / 33.9-34.2 ms (all) against 33.1 without: the sums in the epilogue of a latency-bound ring
# kernel cost what the separate, HBM-efficient statistics pass costs, and 1 x 1 convolutions would leave their tuned GEMM kernels for it.
FUSE_CONV_STATS = __import__("os").environ.get("MODE_ENC_FUSE_CONV_STATS", "0") == "1"
FUSE_CONV_STATS_1X1 = __import__("os").environ.get("MODE_ENC_FUSE_CONV_STATS_1X1", "0") == "1"   # ... also for 1 x 1 / stride-1 convolutions (they leave the tuned GEMM kernels for it)


def _film_arg(t, n: int, c: int):
    return None if t is None else t.reshape(n, c).to(torch.float32).contiguous()


def conv_bn_act(conv: nn.Conv2d, bn: nn.BatchNorm2d, x, relu: bool = True, residual=None, pre_film=None, post_film=None):
    """``bn_film_act(_conv2d(conv, x), bn, ...)``; on the inference path (no grad, eval-mode BatchNorm with running statistics, bf16 compute, a channel count
    the implicit-GEMM kernel takes) as one launch."""
    w = conv.weight
    cd = _compute_dtype(x)
    if (FUSE_CONV_BN and USE_HIP_CONV_WGRAD and x.is_cuda and cd == torch.bfloat16 and not torch.is_grad_enabled() and not bn.training and bn.running_mean is not None
            and conv.groups == 1 and conv.dilation == (1, 1) and conv.bias is None and isinstance(conv.padding, tuple) and w.shape[1] % 64 == 0 and w.shape[0] % 8 == 0):
        w_lp = None
        if _W_OVERRIDE is not None:
            hit = _W_OVERRIDE.get(id(conv))
            if hit is not None and hit.dtype == cd:
                w_lp = hit
        if w_lp is None:
            if CHANNELS_LAST and not w.is_contiguous(memory_format=torch.channels_last):
                with torch.no_grad():
                    w.data = w.data.contiguous(memory_format=torch.channels_last)
            w_lp = _shadow(conv, cd)
        xc = x.to(cd)
        n, cin, H, W_ = xc.shape
        cout, _, kh_, kw_ = w.shape
        sh, sw = conv.stride; ph, pw = conv.padding
        ho = (H + 2 * ph - kh_) // sh + 1; wo = (W_ + 2 * pw - kw_) // sw + 1
        one = kh_ == 1 and kw_ == 1 and (sh, sw) == (1, 1) and (ph, pw) == (0, 0)
        res = None if residual is None else residual.to(cd)
        if (xc.is_contiguous(memory_format=torch.channels_last) and (one or w_lp.is_contiguous(memory_format=torch.channels_last))
                and (res is None or (res.shape == (n, cout, ho, wo) and res.is_contiguous(memory_format=torch.channels_last)))):
            idx = None if one else _tap_table(n, H, W_, ho, wo, kh_, kw_, sh, sw, ph, pw, xc.device)
            pg, pb = (_film_arg(pre_film[0], n, cout), _film_arg(pre_film[1], n, cout)) if pre_film is not None else (None, None)
            qg, qb = (_film_arg(post_film[0], n, cout), _film_arg(post_film[1], n, cout)) if post_film is not None else (None, None)
            y = torch.empty((n, cout, ho, wo), dtype=cd, device=xc.device, memory_format=torch.channels_last)
            R = n * ho * wo
            d = L.ModeConvBnDesc(x=xc.data_ptr(), ldx=cin, idx=_ptr(idx), idx_tap_stride=R, taps=kh_ * kw_, w=w_lp.data_ptr(), ldw=kh_ * kw_ * cin, y=y.data_ptr(), ldy=cout,
                                 M=R, Cin=cin, Cout=cout, bn_mean=_ptr(bn.running_mean), bn_var=_ptr(bn.running_var), bn_weight=_ptr(bn.weight), bn_bias=_ptr(bn.bias),
                                 bn_eps=bn.eps, residual=_ptr(res), ldr=cout, relu=int(relu), pre_gamma=_ptr(pg), pre_beta=_ptr(pb), post_gamma=_ptr(qg), post_beta=_ptr(qb),
                                 rows_per_sample=ho * wo)
            L.check(L.load().mode_conv_bn_act_fwd(C.byref(d), _stream()), "mode_conv_bn_act_fwd")
            return y
    if (FUSE_CONV_STATS and USE_HIP_CONV_WGRAD and x.is_cuda and cd == torch.bfloat16 and torch.is_grad_enabled() and bn.training and not isinstance(bn, nn.SyncBatchNorm)
            and (w.requires_grad or x.requires_grad) and conv.groups == 1 and conv.dilation == (1, 1) and conv.bias is None and isinstance(conv.padding, tuple)
            and w.shape[1] % 64 == 0 and w.shape[0] % 8 == 0 and (FUSE_CONV_STATS_1X1 or not _is_1x1(w.shape, conv.stride, conv.padding))):
        # training: the convolution's epilogue also writes the partial batch statistics of its output (one pass over y less per BatchNorm)
        if CHANNELS_LAST and not w.is_contiguous(memory_format=torch.channels_last):
            with torch.no_grad():
                w.data = w.data.contiguous(memory_format=torch.channels_last)
        xc = x.to(cd)
        w_lp = _shadow(conv, cd)
        if xc.is_contiguous(memory_format=torch.channels_last) and w_lp.is_contiguous(memory_format=torch.channels_last):
            y, ps, pq = _ConvFn.apply(xc, w, w_lp, conv.stride, conv.padding, True)
            return bn_film_act(y, bn, relu=relu, residual=residual, pre_film=pre_film, post_film=post_film, stats=(ps, pq))
    return bn_film_act(_conv2d(conv, x), bn, relu=relu, residual=residual, pre_film=pre_film, post_film=post_film)


# ---- the encoders' entry on the library's kernels (round 5, csrc/conv_stem.hip, ABI 12): conv1 (3 -> 64 channels, 7 x 7 / 2: too few channels for the implicit-GEMM
# path, MIOpen until now) gathers its im2col tile from the image as it lies (fp32 NCHW: the bf16 rounding and the layout change happen in the gather), the max-pool
# keeps window positions for a gather-form backward.  With them no MIOpen / aten compute kernel is left in the encoders.  MODE_ENC_HIPSTEM=0: F.conv2d / F.max_pool2d.
USE_HIP_STEM = __import__("os").environ.get("MODE_ENC_HIPSTEM", "1") == "1"
_DT = {torch.float32: L.MODE_F32, torch.bfloat16: L.MODE_BF16}


def _stem_ok(conv: nn.Conv2d, x: torch.Tensor) -> bool:
    w = conv.weight
    return (USE_HIP_STEM and x.is_cuda and x.dim() == 4 and x.dtype in _DT and _compute_dtype(x) == torch.bfloat16 and conv.groups == 1 and conv.dilation == (1, 1)
            and conv.bias is None and isinstance(conv.padding, tuple) and w.shape[1] % 64 != 0 and w.shape[0] % 16 == 0 and w.shape[0] <= 64
            and w.shape[1] * w.shape[2] * w.shape[3] <= 256 and x.shape[1] == w.shape[1]
            and (7 * conv.stride[0] + w.shape[2]) * (15 * conv.stride[1] + w.shape[3]) * w.shape[1] <= 8192)      # the input patch of an 8 x 16-pixel tile is staged on chip


def _stem_desc(x: torch.Tensor, w_lp: torch.Tensor, stride, padding, **kw):
    n, cin, H, W_ = x.shape
    cout, _, kh_, kw_ = w_lp.shape
    return L.ModeStemConvDesc(x=x.data_ptr(), x_dtype=_DT[x.dtype], sxn=x.stride(0), sxc=x.stride(1), sxh=x.stride(2), sxw=x.stride(3), N=n, H=H, W=W_, Cin=cin,
                              kh=kh_, kw=kw_, sh=stride[0], sw=stride[1], ph=padding[0], pw=padding[1], Cout=cout, w=w_lp.data_ptr(), **kw)


def _stem_fwd(x: torch.Tensor, w_lp: torch.Tensor, stride, padding, bn: Optional[nn.BatchNorm2d] = None, relu: bool = False) -> torch.Tensor:
    """conv2d(x.to(bf16), w_lp) as channels_last bf16 (one launch; `bn`: + eval-mode BatchNorm on the fp32 sums, `relu`)."""
    assert w_lp.dtype == torch.bfloat16 and w_lp.is_contiguous(memory_format=torch.channels_last)
    n, _, H, W_ = x.shape
    cout, _, kh_, kw_ = w_lp.shape
    ho = (H + 2 * padding[0] - kh_) // stride[0] + 1; wo = (W_ + 2 * padding[1] - kw_) // stride[1] + 1
    y = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    ep = {} if bn is None else dict(bn_mean=_ptr(bn.running_mean), bn_var=_ptr(bn.running_var), bn_weight=_ptr(bn.weight), bn_bias=_ptr(bn.bias), bn_eps=bn.eps)
    d = _stem_desc(x, w_lp, stride, padding, y=y.data_ptr(), relu=int(relu), **ep)
    L.check(L.load().mode_stem_conv_fwd(C.byref(d), _stream()), "stem convolution forward")
    return y


class _StemConvFn(torch.autograd.Function):
    """y = conv2d(image, w) for the small-channel stem; differentiable in the fp32 PARAMETER w (the image needs no gradient)."""

    @staticmethod
    def forward(ctx, x, w, w_lp, stride, padding):
        ctx.save_for_backward(x, w_lp)
        ctx.conf = (tuple(stride), tuple(padding), w.dtype)
        return _stem_fwd(x, w_lp, stride, padding)

    @staticmethod
    def backward(ctx, dy):
        x, w_lp = ctx.saved_tensors
        stride, padding, wdtype = ctx.conf
        dx = dw = None
        if ctx.needs_input_grad[1]:
            cout, cin, kh_, kw_ = w_lp.shape
            dyc = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            if dyc.data_ptr() % 16:                                   # (a view at an odd offset: the kernel reads 16-byte row chunks)
                dyc = dyc.clone(memory_format=torch.channels_last)
            d = _stem_desc(x, w_lp, stride, padding, dy=dyc.data_ptr())
            lib = L.load()
            slabs = lib.mode_stem_conv_wgrad_slabs(C.byref(d))
            part = torch.empty((slabs, cout, kh_, kw_, cin), dtype=torch.float32, device=dy.device)      # channels_last order of [Cout, Cin, kh, kw]
            d.dw_part = part.data_ptr()
            L.check(lib.mode_stem_conv_wgrad(C.byref(d), _stream()), "stem convolution weight gradient")
            dw = (part.sum(0) if slabs > 1 else part[0]).permute(0, 3, 1, 2).to(wdtype)
        if ctx.needs_input_grad[0]:                                  # (not a path the encoders take: the input is the camera image)
            dx = torch.ops.aten.convolution_backward(dy, x.to(dy.dtype), w_lp.to(dy.dtype), None, stride, padding, (1, 1), False, (0, 0), 1, (True, False, False))[0]
        return dx, dw, None, None, None


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        n, c, H, W_ = x.shape
        ho = (H + 2 * p - k) // s + 1; wo = (W_ + 2 * p - k) // s + 1
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        arg = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device) if ctx.needs_input_grad[0] else None     # window positions only when a backward will follow
        L.check(L.load().mode_maxpool_nhwc_fwd(x.data_ptr(), _DT[x.dtype], n, H, W_, c, k, s, p, y.data_ptr(), _ptr(arg), _stream()), "max-pool forward")
        ctx.conf = (k, s, p, tuple(x.shape))
        if arg is not None:
            ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        k, s, p, (n, c, H, W_) = ctx.conf
        dyc = dy.contiguous(memory_format=torch.channels_last)
        if dyc.data_ptr() % 16:
            dyc = dyc.clone(memory_format=torch.channels_last)
        dx = torch.empty((n, c, H, W_), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        L.check(L.load().mode_maxpool_nhwc_bwd(dyc.data_ptr(), arg.data_ptr(), _DT[dy.dtype], n, H, W_, c, k, s, p, dx.data_ptr(), _stream()), "max-pool backward")
        return dx, None, None, None


def max_pool(x: torch.Tensor, k: int = 3, s: int = 2, p: int = 1) -> torch.Tensor:
    """F.max_pool2d(x, k, s, p) on channels_last activations through the library (forward + gather-form backward)."""
    if (USE_HIP_STEM and x.is_cuda and x.dim() == 4 and x.dtype in _DT and x.shape[1] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and 2 * p <= k <= 15 and x.shape[2] + 2 * p >= k and x.shape[3] + 2 * p >= k and x.data_ptr() % 16 == 0):
        return _MaxPoolFn.apply(x, k, s, p)
    return F.max_pool2d(x, k, s, p)


def stem_conv_bn_pool(conv: nn.Conv2d, bn: nn.BatchNorm2d, x: torch.Tensor) -> torch.Tensor:
    """``max_pool2d(relu(bn(conv(x))), 3, 2, 1)`` - the trunk's entry (timm / torchvision ResNet: conv1, bn1, act1 / relu, maxpool)."""
    if not _stem_ok(conv, x):
        return max_pool(bn_film_act(_conv2d(conv, _to_layout(x)), bn, relu=True))
    w = conv.weight
    if not w.is_contiguous(memory_format=torch.channels_last):
        with torch.no_grad():
            w.data = w.data.contiguous(memory_format=torch.channels_last)
    w_lp = None
    if _W_OVERRIDE is not None:
        hit = _W_OVERRIDE.get(id(conv))
        if hit is not None and hit.dtype == torch.bfloat16 and hit.is_contiguous(memory_format=torch.channels_last):
            w_lp = hit
    train = torch.is_grad_enabled() and (w.requires_grad or x.requires_grad)
    if w_lp is None:
        w_lp = _shadow(conv, torch.bfloat16)
    if train:
        y = bn_film_act(_StemConvFn.apply(x, w, w_lp, conv.stride, conv.padding), bn, relu=True)
    elif FUSE_CONV_BN and not bn.training and bn.running_mean is not None and _bn_fusable(bn):
        y = _stem_fwd(x, w_lp, conv.stride, conv.padding, bn=bn, relu=True)          # inference: convolution + BatchNorm + ReLU in one launch
    else:
        y = bn_film_act(_stem_fwd(x, w_lp, conv.stride, conv.padding), bn, relu=True)
    return max_pool(y)


def _bn_fusable(bn) -> bool:
    """The folded conv + BatchNorm inference launch bypasses autograd and reads the BatchNorm statistics / affine terms as raw fp32 pointers: only when nothing of
    the BatchNorm can want a gradient (a frozen convolution in front of a trainable eval-mode BatchNorm must keep the autograd path) and every tensor is fp32
    and contiguous (an encoder cast to bfloat16 would otherwise be read as garbage)."""
    ts = [t for t in (bn.running_mean, bn.running_var, bn.weight, bn.bias) if t is not None]
    if torch.is_grad_enabled() and any(getattr(t, "requires_grad", False) for t in (bn.weight, bn.bias) if t is not None):
        return False
    return all(t.dtype == torch.float32 and t.is_contiguous() for t in ts)


# ------------------------------------------------------------------------------------------------------------------ trunk (parameter holders)
class _Block(nn.Module):
    """BasicBlock (expansion 1) / Bottleneck (expansion 4) holder with the timm / torchvision attribute names."""

    def __init__(self, inplanes: int, planes: int, stride: int, bottleneck: bool):
        super().__init__()
        self.bottleneck, self.stride = bottleneck, stride
        out = planes * (4 if bottleneck else 1)
        if bottleneck:
            self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, out, 1, bias=False); self.bn3 = nn.BatchNorm2d(out)
        else:
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != out:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, out, 1, stride=stride, bias=False), nn.BatchNorm2d(out))
        self.out_channels = out

    def _conv(self, conv: nn.Conv2d, x):
        return _conv2d(conv, x)

    def forward(self, x, pre_film=None, post_film=None):
        """``pre_film``: FiLM on the last BatchNorm's output before the skip add (resnets.py:64-71); ``post_film``: FiLM on the block output (the
        stage-level FiLMLayer of pretrained_resnets.py fused into the stage's last block)."""
        identity = x
        if self.downsample is not None:
            identity = conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False)
        out = conv_bn_act(self.conv1, self.bn1, x, relu=True)
        if self.bottleneck:
            out = conv_bn_act(self.conv2, self.bn2, out, relu=True)
            return conv_bn_act(self.conv3, self.bn3, out, relu=True, residual=identity, pre_film=pre_film, post_film=post_film)
        return conv_bn_act(self.conv2, self.bn2, out, relu=True, residual=identity, pre_film=pre_film, post_film=post_film)


_ARCH = {"18": (False, (2, 2, 2, 2)), "34": (False, (3, 4, 6, 3)), "50": (True, (3, 4, 6, 3))}


class _Trunk(nn.Module):
    """conv1 / bn1 / maxpool / layer1..4 with the standard ResNet widths (He et al. 2016); attribute names as in timm (``act1``, ``global_pool``)
    and torchvision (``relu``, ``avgpool``) so that either family's keys load."""

    def __init__(self, arch: str):
        super().__init__()
        bottleneck, depths = _ARCH[arch]
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), depths)):
            blocks: List[_Block] = []
            for j in range(n):
                blocks.append(_Block(inplanes, planes, stride=(1 if i == 0 else 2) if j == 0 else 1, bottleneck=bottleneck))
                inplanes = blocks[-1].out_channels
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.num_features = inplanes
        _store_channels_last(self)

    def stem(self, x):
        return stem_conv_bn_pool(self.conv1, self.bn1, x)


# ---- the encoders' small Linears (FiLM gamma / beta, FilmModule, fc) on the library's fp32 MFMA GEMM instead of hipBLASLt: y = x W^T + b, dx = dy W, dW = dy^T x from
#      row-major operands where they lie (exact fp32 fma chains; conditioning vectors are 512 wide, the products tiny).  MODE_ENC_HIPLINEAR=0: nn.Linear (A/B runs).
USE_HIP_LINEAR = __import__("os").environ.get("MODE_ENC_HIPLINEAR", "1") == "1"


class _LinearFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w, b):
        x = x.contiguous()
        M_, K_ = x.shape
        N_ = w.shape[0]
        y = torch.empty(M_, N_, dtype=torch.float32, device=x.device)
        d = L.ModeGemmDesc(dtype=L.MODE_F32, epilogue=L.EPI_BIAS if b is not None else L.EPI_NONE, out_dtype=L.MODE_F32, M=M_, N=N_, K=K_, A=x.data_ptr(), lda=K_,
                           W=w.data_ptr(), ldw=w.stride(0), bias=_ptr(b), C=y.data_ptr(), ldc=N_, flags=L.GEMM_SKINNY_OK)
        L.check(L.load().mode_gemm(C.byref(d), _stream()), "linear fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous().float()
        M_, K_ = x.shape
        N_ = w.shape[0]
        lib, st = L.load(), _stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:                                            # dx[M, K] = dy[M, N] @ W[N, K]  (W as [K_red = N][cols = K]: MODE_GEMM_W_KN)
            dx = torch.empty(M_, K_, dtype=torch.float32, device=x.device)
            d = L.ModeGemmDesc(dtype=L.MODE_F32, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=M_, N=K_, K=N_, A=dy.data_ptr(), lda=N_, W=w.data_ptr(), ldw=w.stride(0),
                               C=dx.data_ptr(), ldc=K_, flags=L.GEMM_W_KN)
            L.check(lib.mode_gemm(C.byref(d), st), "linear dx")
        if ctx.needs_input_grad[1]:                                            # dW[N, K] = dy[M, N]^T @ x[M, K]  (both operands [K_red = M][cols])
            dw = torch.empty(N_, K_, dtype=torch.float32, device=x.device)
            d = L.ModeGemmDesc(dtype=L.MODE_F32, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=N_, N=K_, K=M_, A=dy.data_ptr(), lda=N_, W=x.data_ptr(), ldw=K_,
                               C=dw.data_ptr(), ldc=K_, flags=L.GEMM_W_KN | L.GEMM_A_KM)
            L.check(lib.mode_gemm(C.byref(d), st), "linear dW")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def _linear(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """``lin(x)`` for 2-d fp32-parameter Linears on the device: the library's GEMM; anything else: nn.Linear."""
    w = lin.weight
    if (USE_HIP_LINEAR and x.is_cuda and x.dim() == 2 and w.dtype == torch.float32 and w.is_contiguous() and x.shape[1] % 4 == 0 and w.shape[0] % 4 == 0
            and x.shape[0] > 0):
        return _LinearFn.apply(x, w, lin.bias)
    return lin(x)


class FiLMLayer(nn.Module):
    """gamma / beta = Linear(condition): x <- (1 + gamma) x + beta, zero-initialised (pretrained_resnets.py:5-23).  Holds parameters; the modulation
    itself is fused into the preceding BatchNorm pass."""

    def __init__(self, num_features: int, condition_dim: int):
        super().__init__()
        self.num_features, self.condition_dim = num_features, condition_dim
        self.gamma = nn.Linear(condition_dim, num_features)
        self.beta = nn.Linear(condition_dim, num_features)
        for lin in (self.gamma, self.beta):
            nn.init.zeros_(lin.weight); nn.init.zeros_(lin.bias)

    def params(self, condition):
        return _linear(self.gamma, condition), _linear(self.beta, condition)


class _FiLMResNetPolicy(nn.Module):
    arch = "50"

    def __init__(self, condition_dim: int):
        super().__init__()
        self.resnet = _Trunk(self.arch)
        widths = [getattr(self.resnet, f"layer{i}")[-1].out_channels for i in range(1, 5)]
        self.film1, self.film2, self.film3, self.film4 = (FiLMLayer(w, condition_dim) for w in widths)

    def forward(self, x, condition):
        if condition.dim() == 3:
            condition = condition.squeeze(1)
        if USE_HIP_CONV_WGRAD and x.is_cuda and torch.is_grad_enabled() and _compute_dtype(x) == torch.bfloat16:
            refresh_conv_shadows(self, torch.bfloat16, force=True)                  # one multi-tensor cast for all 53 weights; unconditional: `.data` writes bump no version
        x = self.resnet.stem(x)
        for i in range(1, 5):
            film = getattr(self, f"film{i}").params(condition.to(torch.float32))
            layer = getattr(self.resnet, f"layer{i}")
            for j, blk in enumerate(layer):
                x = blk(x, post_film=film if j == len(layer) - 1 else None)     # FiLM after the stage = epilogue of its last block
        return x.mean(dim=(2, 3))                                             # global_pool + flatten(1)


class FiLMResNet50Policy(_FiLMResNetPolicy):
    arch = "50"


class FiLMResNet34Policy(_FiLMResNetPolicy):
    arch = "34"


class FiLMResNet18Policy(_FiLMResNetPolicy):
    arch = "18"


class FilmModule(nn.Module):
    """SiLU -> Linear(cond, 4 * hidden): ((gamma_0, beta_0), (gamma_1, beta_1)) for the two blocks of a ResNet-18 stage (resnets.py:27-44)."""

    def __init__(self, input_size: int, hidden_size: int):
        super().__init__()
        self.modulation = nn.Sequential(nn.SiLU(), nn.Linear(input_size, 4 * hidden_size, bias=True))

    def forward(self, c):
        x = _linear(self.modulation[1], self.modulation[0](c)).chunk(2, dim=-1)
        return x[0].chunk(2, dim=-1), x[1].chunk(2, dim=-1)


class ResNetEncoderWithFiLM(nn.Module):
    """resnets.py:81-200: ResNet-18 trunk, per-block FiLM ``gamma * bn2(.) + beta`` before the skip add, average pool, ``fc`` to ``latent_dim``.
    Accepts (B, C, H, W) or (B, T, C, H, W) like the reference."""

    def __init__(self, cond_dim: int, latent_dim: int = 512, pretrained: bool = False, hidden_size: int = 512):
        super().__init__()
        if pretrained:
            raise NotImplementedError("pretrained=True downloads ImageNet weights in the reference; load a checkpoint with load_state_dict instead")
        self.latent_dim = latent_dim
        t = _Trunk("18")
        self.conv1, self.bn1 = t.conv1, t.bn1
        self.film_module1, self.film_module2 = FilmModule(cond_dim, 64), FilmModule(cond_dim, 128)
        self.film_module3, self.film_module4 = FilmModule(cond_dim, 256), FilmModule(cond_dim, 512)
        self.layer1, self.layer2, self.layer3, self.layer4 = t.layer1, t.layer2, t.layer3, t.layer4
        self.fc = nn.Linear(512, latent_dim)

    def forward(self, x, conditioning_vector: Optional[torch.Tensor] = None):
        B, t_steps, series = len(x), 1, False
        if x.dim() == 5:
            t_steps, series = x.shape[1], True
            x = x.reshape(B * t_steps, *x.shape[2:])
            if conditioning_vector is not None:
                conditioning_vector = torch.cat([conditioning_vector for _ in range(t_steps)], dim=0)     # the reference's order (resnets.py:129)
        if USE_HIP_CONV_WGRAD and x.is_cuda and torch.is_grad_enabled() and _compute_dtype(x) == torch.bfloat16:
            refresh_conv_shadows(self, torch.bfloat16, force=True)
        x = stem_conv_bn_pool(self.conv1, self.bn1, x)
        for i in range(1, 5):
            mods = getattr(self, f"film_module{i}")(conditioning_vector.to(torch.float32)) if conditioning_vector is not None else (None, None)
            for j, blk in enumerate(getattr(self, f"layer{i}")):
                x = blk(x, pre_film=mods[j])
        x = _linear(self.fc, x.mean(dim=(2, 3)).to(self.fc.weight.dtype))
        if series:
            x = x.reshape(B, t_steps, self.latent_dim)
        return x


# the two camera towers on two streams (agent training step B = 64, same box alternating: 30.5 / 30.2 -> 27.0 / 27.8 ms, then host-bound: 26.7-27.3 ms of enqueue);
# MODE_ENC_TWO_STREAMS=0: one stream (A/B runs)
TWO_TOWER_STREAMS = __import__("os").environ.get("MODE_ENC_TWO_STREAMS", "1") == "1"
TWO_TOWER_CAPTURE = __import__("os").environ.get("MODE_ENC_TWO_STREAMS_CAPTURE", "1") == "1"   # also inside GraphedVisualEncoder's captures (two graph branches)
_TOWER_STREAMS: dict = {}


def embed_visual_obs(static_resnet, gripper_resnet, rgb_static, rgb_gripper, latent_goal=None):
    """``MoDEAgent.embed_visual_obs`` (mode_agent.py:548-567): (B, T, C, H, W) camera streams -> ``{'state_images': (B, 2 T, obs_dim)}``, the
    ``perceptual_emb`` the denoiser consumes (one token per camera and frame; T = 1 in every shipped config)."""
    B, T = rgb_static.shape[0], rgb_static.shape[1]
    s = rgb_static.reshape(B * T, *rgb_static.shape[2:]); g = rgb_gripper.reshape(B * T, *rgb_gripper.shape[2:])
    args = (latent_goal,) if latent_goal is not None else ()
    capturing = s.is_cuda and torch.cuda.is_current_stream_capturing()
    if TWO_TOWER_STREAMS and s.is_cuda and (TWO_TOWER_CAPTURE or not capturing):
        # The two camera towers are independent: the gripper tower runs on a second stream (its backward too - autograd replays a node on the stream its
        # forward ran on), so that one tower's HBM-bound BatchNorm passes overlap the other's MFMA-bound convolution GEMMs.
        cur = torch.cuda.current_stream(s.device)
        side = _TOWER_STREAMS.get(s.device)
        if side is None:
            side = _TOWER_STREAMS[s.device] = torch.cuda.Stream(device=s.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gt = gripper_resnet(g, *args)
        st = static_resnet(s, *args)
        cur.wait_stream(side)                                                # (under capture: the side stream's work becomes a second branch of the graph)
        if not capturing:
            gt.record_stream(cur)
    else:
        st, gt = static_resnet(s, *args), gripper_resnet(g, *args)
    return {"state_images": torch.cat([st.reshape(B, T, -1), gt.reshape(B, T, -1)], dim=1)}


class GraphedVisualEncoder:
    """``embed_visual_obs`` for the rollout as ONE hipGraph replay per call.

    In the rollout the two encoders run in eval mode on a handful of frames (the reference agent: one environment, B = 1, mode_agent.py:584-637):
    ~640 launches of microseconds each - 10.2 ms of host time at B = 1 for two FiLM-ResNet-50s, more than twice the 10-step denoising chunk they
    feed.  The launch sequence does not depend on the data, so it is captured once per (batch, image shapes, dtype) and replayed: 2.8 ms at B = 1
    (scripts/encoder_eval_latency.py).  The graph reads parameters and BatchNorm buffers where they live: in-place updates (``load_state_dict``,
    an EMA swap) are seen by the next replay; re-allocated parameters (``.to()``, ``.half()``) or a switch to training mode re-capture.

    ``autocast_dtype``: run under ``torch.autocast`` like the reference's trainer / evaluation precision (None = the tensors' own dtype)."""

    def __init__(self, static_resnet: nn.Module, gripper_resnet: nn.Module, autocast_dtype: Optional[torch.dtype] = torch.bfloat16, max_graphs: int = 4):
        self.static_resnet, self.gripper_resnet = static_resnet, gripper_resnet
        self.autocast_dtype, self.max_graphs = autocast_dtype, max_graphs
        self._graphs = {}
        self._fork = None

    def _eager(self, rgb_static, rgb_gripper, latent_goal):
        """embed_visual_obs with the two cameras on two streams (fork / join): at rollout batch sizes every kernel is a few microseconds of a
        dependent chain, and the two encoders do not depend on each other - captured, they become two parallel branches of the graph."""
        import contextlib
        ac = (lambda: torch.autocast("cuda", dtype=self.autocast_dtype)) if self.autocast_dtype is not None else contextlib.nullcontext
        B, T = rgb_static.shape[0], rgb_static.shape[1]
        s = rgb_static.reshape(B * T, *rgb_static.shape[2:]); g = rgb_gripper.reshape(B * T, *rgb_gripper.shape[2:])
        cur = torch.cuda.current_stream(rgb_static.device)
        if self._fork is None or self._fork.device != rgb_static.device:
            self._fork = torch.cuda.Stream(device=rgb_static.device)
        self._fork.wait_stream(cur)
        with torch.cuda.stream(self._fork), ac():
            gt = self.gripper_resnet(g, latent_goal) if latent_goal is not None else self.gripper_resnet(g)
        with ac():
            st = self.static_resnet(s, latent_goal) if latent_goal is not None else self.static_resnet(s)
        cur.wait_stream(self._fork)
        return torch.cat([st.reshape(B, T, -1), gt.reshape(B, T, -1)], dim=1)

    def _weights(self, dtype: torch.dtype) -> dict:
        """Convolution weights of both encoders in the compute dtype, refreshed IN PLACE when a Parameter's version (or storage) changed - the graph
        reads these copies, so an in-place weight update is seen by the next replay."""
        cache = self.__dict__.setdefault("_wcache", {})
        convs = self.__dict__.get("_convs")
        if convs is None:                                                        # the module tree is fixed: walk it once
            convs = self._convs = [m for enc in (self.static_resnet, self.gripper_resnet) for m in enc.modules() if isinstance(m, nn.Conv2d)]
        out = {}
        for m in convs:
            w = m.weight
            # one copy per (convolution, dtype, device): a graph captured for another input dtype keeps reading ITS copies - nothing a captured graph
            # points at is ever re-allocated while the parameter itself stays where it is (a moved parameter changes _param_key -> new graphs)
            ck = (id(m), dtype, str(w.device))
            ent = cache.get(ck)
            if ent is None:
                ent = cache[ck] = [None, None, torch.empty_like(w, dtype=dtype)]
            if ent[0] != _wver(m) or ent[1] != w.data_ptr():
                ent[2].copy_(w)
                ent[0], ent[1] = _wver(m), w.data_ptr()
            out[id(m)] = ent[2]
        return out

    def _param_key(self):
        """Fingerprint of where EVERY tensor the captured graphs read lives (convolution / BatchNorm / FiLM parameters and the BatchNorm buffers of both
        encoders): a re-allocated interior tensor (``p.data = ...``, ``.to()``, ``.half()``) must re-capture, not replay against freed memory."""
        return hash(tuple(t.data_ptr() for m in (self.static_resnet, self.gripper_resnet) for t in list(m.parameters()) + list(m.buffers())))

    @torch.no_grad()
    def __call__(self, rgb_static: torch.Tensor, rgb_gripper: torch.Tensor, latent_goal: Optional[torch.Tensor] = None):
        import os
        from .engine import capture_graph
        if self.static_resnet.training or self.gripper_resnet.training or rgb_static.device.type != "cuda" or os.environ.get("MODE_HIP_GRAPH", "1") == "0":
            return {"state_images": self._eager(rgb_static, rgb_gripper, latent_goal)}       # batch statistics / no device: nothing to replay
        wdt = self.autocast_dtype if self.autocast_dtype is not None else rgb_static.dtype
        key = (tuple(rgb_static.shape), rgb_static.dtype, tuple(rgb_gripper.shape), rgb_gripper.dtype,
               None if latent_goal is None else (tuple(latent_goal.shape), latent_goal.dtype), str(rgb_static.device), self._param_key())
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            ent = dict(s=rgb_static.clone(), g=rgb_gripper.clone(), c=None if latent_goal is None else latent_goal.clone())
            dev = rgb_static.device
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            global _W_OVERRIDE
            saved, _W_OVERRIDE = _W_OVERRIDE, self._weights(wdt)
            saved_sink, _TABLES.sink = _TABLES.sink, []                          # every index table the warm-up / capture touches: pinned next to the graph
            ent["tables"] = _TABLES.sink
            try:
                with torch.cuda.stream(side):                                    # outside the capture: MIOpen's algorithm search, code-object loads
                    for _ in range(2):
                        self._eager(ent["s"], ent["g"], ent["c"])
                torch.cuda.current_stream(dev).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with capture_graph(graph):
                    ent["out"] = self._eager(ent["s"], ent["g"], ent["c"])
            finally:
                _W_OVERRIDE = saved
                _TABLES.sink = saved_sink
            ent["graph"] = graph
            self._graphs[key] = ent
        self._weights(wdt)                                                       # weights whose version moved since the last call: re-cast in place
        ent["s"].copy_(rgb_static); ent["g"].copy_(rgb_gripper)
        if latent_goal is not None:
            ent["c"].copy_(latent_goal)
        ent["graph"].replay()
        return {"state_images": ent["out"].clone()}
