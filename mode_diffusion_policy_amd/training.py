"""Training-mode forward + backward of the MoDE denoiser through the HIP chains.

``MoDeDiT.forward`` in ``.train()`` mode lands here.  The forward runs ``mode_dit_forward_train`` (activation stash, per-token
expert ids drawn by ``torch.multinomial`` exactly where the reference draws them — modedit.py:390 — or top-k when
``use_argmax``; attention / expert dropout as counter-based hash masks), and a ``torch.autograd.Function`` hands ``dF`` to
``mode_dit_backward`` which writes the gradient of every parameter.  PyTorch only wires tensors together; no FLOP of the
denoiser runs in eager PyTorch.

The auxiliary router losses (``MoDeDiT.load_balancing_loss`` / ``compute_router_z_loss``, modedit.py:898-969, per-block term :586-593) are
outputs of the same autograd node: adding ``entropy_gamma * LB + router_z_delta * Z`` to the loss (mode_agent.py:413-419) trains the routers
through ``mode_moe_router_bwd_aux``.  ``edm_loss`` is ``GCDenoiser.loss`` with its scalings, target and MSE (and their backward) as two HIP
launches; ``diffusion_loss`` / ``training_step`` restate ``MoDEAgent.diffusion_loss`` / ``training_step`` (mode_agent.py:659-672, 386-440).
"""
from __future__ import annotations

import os
import ctypes as C
from typing import Dict

import torch

from . import _lib as L
from .engine import DitEngine, _ptr, _stream


class TrainState:
    """Per-engine training resources that follow the weights: transposed shadows for the data-gradient GEMMs (allocated once,
    re-filled in place whenever the weights change) and the pointer tables of the gradient arena."""

    def __init__(self, eng: DitEngine):
        self.eng = eng
        self.key = None
        self.keep: Dict[str, torch.Tensor] = {}
        self.built_for = None
        self.grads_for = None
        self._ws = None
        self.events = None                 # one event per block: "all weight gradients of block l are written" (data-parallel overlap)
        self._ev_arr = None
        self._act_rows: Dict[tuple, torch.Tensor] = {}

    def _build(self) -> None:
        eng = self.eng
        m, dev, tdt = eng.model, eng.device, eng.tdt
        D, E, A, Ly = m.embed_dim, m.num_experts, m.action_dim, m.num_layers
        keep: Dict[str, torch.Tensor] = {}
        layers = (L.ModeLayerWeightsT * Ly)()
        lp = eng.compute_dtype == "bf16"                        # bf16 backward GEMMs read the [out,in] weights directly (MODE_GEMM_W_KN)
        for i in range(Ly):
            k = f"l{i}."
            lt = layers[i]
            if not lp:
                keep[k + "wqkvT"] = torch.empty(D, 3 * D, dtype=tdt, device=dev)
                keep[k + "woT"] = torch.empty(D, D, dtype=tdt, device=dev)
                keep[k + "w1T"] = torch.empty(E, D, 8 * D, dtype=tdt, device=dev)
                keep[k + "w2T"] = torch.empty(E, 4 * D, D, dtype=tdt, device=dev)
                lt.wqkvT, lt.woT, lt.w1T, lt.w2T = (_ptr(keep[k + n]) for n in ("wqkvT", "woT", "w1T", "w2T"))
        keep["w_outT"] = torch.empty(D, A, device=dev)
        wt = L.ModeModelWeightsT()
        wt.w_outT = _ptr(keep["w_outT"])
        wt.layers = C.cast(layers, C.POINTER(L.ModeLayerWeightsT))
        self.keep, self.layersT, self.wt = keep, layers, wt
        self.built_for = eng._structs_for

    def ensure(self) -> None:
        eng = self.eng
        if self.built_for != eng._structs_for:
            self._build()
            self.key = None
        if self.key == eng._wkey:
            return
        m, lib, dt = eng.model, eng.lib, eng.dt
        ar = eng.arena
        mat = ar.wl if eng.compute_dtype == "bf16" else ar.w
        D, E, A = m.embed_dim, m.num_experts, m.action_dim
        keep = self.keep

        def tr(src: torch.Tensor, dst: torch.Tensor, rows: int, cols: int, dtype_code: int) -> None:
            L.check(lib.mode_transpose(src.data_ptr(), cols, rows, cols, dst.data_ptr(), rows, None, None, dtype_code, _stream()), "transpose")
        for i in range(m.num_layers):
            k = f"l{i}."
            if eng.compute_dtype != "bf16":
                tr(mat[k + "wqkv"], keep[k + "wqkvT"], 3 * D, D, dt)
                tr(mat[k + "wo"], keep[k + "woT"], D, D, dt)
                for e in range(E):
                    tr(mat[k + "w1"][e], keep[k + "w1T"][e], 8 * D, D, dt)
                    tr(mat[k + "w2"][e], keep[k + "w2T"][e], D, 4 * D, dt)
        tr(ar.w["w_out"], keep["w_outT"], A, D, L.MODE_F32)
        self.key = eng._wkey

    def layer_events(self):
        """hipEvent handles the backward chain records after each block (created once; see ModeTrainArgs.layer_events)."""
        if self.events is None:
            n = self.eng.model.num_layers
            self.events = [torch.cuda.Event() for _ in range(n)]
            for ev in self.events:
                ev.record()                                                     # forces creation of the underlying hipEvent
            self._ev_arr = (C.c_void_p * n)(*[ev.cuda_event for ev in self.events])
        return C.cast(self._ev_arr, C.c_void_p)

    def act_rows(self, B: int, T: int, A_len: int) -> torch.Tensor:
        """Token rows of the action positions (the last A_len of every sample's T), int32 [B * A_len]; built once per batch size."""
        key = (B, T, A_len)
        hit = self._act_rows.get(key)
        if hit is None:
            dev = self.eng.device
            hit = (torch.arange(B, device=dev).repeat_interleave(A_len) * T + (T - A_len) + torch.arange(A_len, device=dev).repeat(B)).to(torch.int32)
            if len(self._act_rows) >= 8:
                self._act_rows.clear()
            self._act_rows[key] = hit
        return hit

    def grad_tables(self, flat=None):
        """ctypes tables pointing the backward chain at a gradient buffer with the arena's layout: the gradient arena itself (built once,
        cached) or ``flat`` - a per-backward buffer (autograd-visible gradients / accumulation).  Returns (tables, views-by-parameter-name)."""
        eng = self.eng
        ar = eng.arena
        m = eng.model
        if flat is None:
            ar.ensure_grad(m)
            if self.grads_for != id(ar.grad):
                self._mg, self._lgr = self._tables(ar.g)
                self.grads_for = id(ar.grad)
            return self._mg, ar.g_by_name
        from .arena import param_views
        g = ar._views(flat)
        # the ctypes tables hold absolute addresses: cached per base pointer (the caching allocator hands a freed 2.7-GB block straight back, so a
        # training loop in autograd mode builds them once); the parameter views must be views of THIS tensor object and are rebuilt
        cache = self.__dict__.setdefault("_flat_tables", {})
        hit = cache.get(flat.data_ptr())
        if hit is None:
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            hit = cache[flat.data_ptr()] = self._tables(g)
        return hit[0], param_views(m, g)

    def _tables(self, g):
        m = self.eng.model
        lgr = (L.ModeLayerGrads * m.num_layers)()
        for i in range(m.num_layers):
            for fld, _ in L.ModeLayerGrads._fields_:
                t = g[fld][i] if fld in g else g[f"l{i}.{fld}"]              # routers / norm gains are stacked over layers
                setattr(lgr[i], fld, t.data_ptr())
        mg = L.ModeModelGrads()
        for fld in ("pos", "w_se", "b_se", "w_sl", "w_tok", "w_goal", "w_act", "ln_g", "w_out", "b_out"):
            setattr(mg, fld, g[fld].data_ptr())
        mg.layers = C.cast(lgr, C.POINTER(L.ModeLayerGrads))
        return mg, lgr


class _Run:
    """Everything one training forward leaves behind for its backward."""
    pass


class _DitTrainFn(torch.autograd.Function):
    """Outputs: F (B, A_len, A), the load-balancing loss and the router z-loss of this forward (scalars).  One backward serves all three.
    Inputs: the (fp32, on-device) state_images and preprocessed goals - so that an upstream encoder trains through the denoiser like it does
    in the reference (mode_agent.py:404-411, 548-567) - and every parameter that can receive a gradient."""

    @staticmethod
    def forward(ctx, run, img, goals, *params):
        # The outputs must NOT stay reachable from ctx: output -> grad_fn -> ctx -> run -> output is a cycle through the C++ autograd node that
        # Python's collector cannot even see - it would pin the activation stash (1.5 GiB at C2 / B = 128) of every step for the life of the process.
        outs, run.outs = run.outs, None
        ctx.run = run
        ctx.set_materialize_grads(False)                         # unused outputs hand None to backward, not zero tensors
        return outs

    @staticmethod
    def backward(ctx, dF, dlb, dz):
        run = ctx.run
        if dF is None:
            dF = torch.zeros(run.F_shape, device=run.device)
        d_img, d_goal, grads = run.backward(dF.contiguous().float(), dlb, dz, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return (None, d_img, d_goal, *grads)


class _EdmLossFn(torch.autograd.Function):
    """loss = mean((F - target)^2) computed by mode_edm_loss together with dF for a unit upstream gradient."""

    @staticmethod
    def forward(ctx, F, loss, dF_unit):
        ctx.save_for_backward(dF_unit)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        (dF_unit,) = ctx.saved_tensors
        return dF_unit * dloss, None, None


def dit_forward_train(model, states, actions, goals, sigma, uncond=False):
    tokr = not getattr(model, "cond_router", True)             # token routing: every block routes each token on its own ln_2 state (modedit.py:296-301)
    eng: DitEngine = model.engine
    if not hasattr(eng, "_train") or eng._train is None:
        eng._train = TrainState(eng)
    ts: TrainState = eng._train
    ts.ensure()
    lib, dev, d = eng.lib, eng.device, eng.dims
    B, T, D, E, k, Ly = actions.shape[0], model.seq_len, model.embed_dim, model.num_experts, model.top_k, model.num_layers
    if B == 0:
        raise ValueError("training forward needs at least one sample")
    N, A_len, A = B * T, model.action_seq_len, model.action_dim
    if torch.is_grad_enabled() and torch.is_tensor(actions) and actions.requires_grad:
        # d state_images and d goals are outputs of the backward chain (the reference trains its FiLM-ResNets through state_images,
        # mode_agent.py:548-567); nothing upstream of the ACTIONS is trainable in the reference (they come from the dataset).
        raise NotImplementedError("MoDeDiT (HIP) training forward: actions require grad, but the backward chain returns gradients for "
                                  "state_images, goals and the parameters only (detach the actions)")
    f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    img_in = states["state_images"].to(device=dev, dtype=torch.float32)       # autograd-tracked casts: the node below returns d img_in / d goal_in
    img = img_in.detach().contiguous()
    if img.dim() != 3 or img.shape[1] != model.n_img_tokens or img.shape[2] != model.obs_dim:
        raise ValueError(f"state_images must be (B, {model.n_img_tokens}, {model.obs_dim}), got {tuple(img.shape)}")
    goal_in = model.preprocess_goals(goals.to(device=dev, dtype=torch.float32), 1, uncond=bool(uncond))      # incl. the Bernoulli goal mask
    gl = goal_in.detach().reshape(B, -1).contiguous()
    acts = f(actions)
    model._check_batch(B, img, gl, acts)
    sig = f(sigma).reshape(-1)
    if sig.numel() == 1:
        sig = sig.expand(B).contiguous()
    if sig.numel() != B:
        raise ValueError("sigma must be a scalar or have one entry per sample")

    run = _Run()
    # sigma embedding in two visible steps (e1 is needed by the backward)
    e1 = torch.empty(B, D, device=dev)
    L.check(lib.mode_sigma_embed(sig.data_ptr(), eng.arena.w["w_se"].data_ptr(), eng.arena.w["b_se"].data_ptr(), e1.data_ptr(), B, D, _stream()), "sigma_embed")
    emb_t = torch.empty(B, D, device=dev)
    g = L.ModeGemmDesc(dtype=L.MODE_F32, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=B, N=D, K=D, A=e1.data_ptr(), lda=D,
                       W=eng.arena.w["w_sl"].data_ptr(), ldw=D, C=emb_t.data_ptr(), ldc=D)
    L.check(lib.mode_gemm(C.byref(g), _stream()), "sigma_linear")
    img_e, goal_e = eng.embed_obs(img, gl)
    cond = (emb_t + goal_e).contiguous() if model.use_goal_in_routing else emb_t
    r_pre = None
    if tokr:
        # routing depends on the layer's own token states: resolved layer by layer inside the forward below (N routing rows per layer)
        probs = torch.empty(Ly, N, E, device=dev); shifted = torch.empty(Ly, N, E, device=dev)
        tr_pre = torch.empty(Ly, N, 2 * D, device=dev)
        idx = torch.empty(Ly, N, k, dtype=torch.int32, device=dev); w = torch.empty(Ly, N, k, device=dev)
        tr_idx = torch.empty(N, k, dtype=torch.int32, device=dev); tr_w = torch.empty(N, k, device=dev)
        tr_expo = None if model.use_argmax else torch.empty(Ly, N, E, device=dev).exponential_()     # the per-layer draws' Exp(1) variates, one launch
        per_tok, tpr, Rr = 1, 1, N
    else:
        idx_top, w_top, probs, shifted, r_pre = eng.route(cond, want_probs=True, want_pre=True)      # [L,B,*]
    if tokr:
        pass
    elif model.use_argmax:
        idx, w, per_tok, tpr, Rr = idx_top, w_top, 0, T, B                                           # top-k also in training (modedit.py:389)
    else:
        # expert ids are SAMPLED per token row without replacement (modedit.py:390); the draw stays on the host side of the ABI
        # torch.multinomial(p, k, replacement=False) IS an exponential race (keys p / q, q ~ Exp(1), the k largest): the variates come from
        # torch's generator exactly as multinomial draws them, so the ids equal torch.multinomial's for the same generator state
        # (tests/test_gpu_train_ops.py); race + top-k + combine weights are one launch for all layers (token row n of layer l reads probs row
        # (l*N + n) / T = l*B + b) - torch.multinomial itself costs ~20 launches per step (input validation, topk, sort, copies)
        expo = torch.empty(Ly * N, E, device=dev).exponential_()
        idx = torch.empty(Ly, N, k, dtype=torch.int32, device=dev)
        w = torch.empty(Ly, N, k, device=dev)
        L.check(lib.mode_moe_sample_experts(probs.data_ptr(), expo.data_ptr(), Ly * N, T, E, k, int(model.router_normalize), idx.data_ptr(),
                                            w.data_ptr(), _stream()), "sample_experts")
        per_tok, tpr, Rr = 1, 1, N
    ml = eng.meta_layout(N)
    meta = torch.empty(Ly, ml.total_words, dtype=torch.int32, device=dev) if tokr else eng.dispatch(idx, w, Ly, Rr, tpr, N)
    act_rows = ts.act_rows(B, T, A_len)
    sl = L.ModeStashLayout()
    L.check(lib.mode_dit_train_stash_layout(C.byref(d), B, eng.dt, C.byref(sl)), "stash_layout")
    stash = torch.empty(sl.total_bytes, dtype=torch.uint8, device=dev)
    F = torch.empty(B, A_len, A, device=dev)
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    model._last_seed = seed                                   # step seed of the hash dropout streams (tests reproduce the masks from it)
    args = L.ModeTrainArgs(B=B, dtype=eng.dt, seed=seed, attn_pdrop=float(model.attn_pdrop), mlp_pdrop=float(model.mlp_pdrop),
                           sigma=sig.data_ptr(), e1=e1.data_ptr(), emb_t=emb_t.data_ptr(), cond=cond.data_ptr(),
                           goal_in_cond=int(model.use_goal_in_routing), state_images=img.data_ptr(), goals=gl.data_ptr(),
                           goal_e=goal_e.data_ptr(), img_e=img_e.data_ptr(), actions=acts.data_ptr(), c_in=None, c_in_stride=0,
                           actions_scaled=acts.data_ptr(), act_rows=act_rows.data_ptr(), meta=meta.data_ptr(), meta_layer_stride=ml.total_words,
                           topk_idx=idx.data_ptr(), topk_layer_stride=idx.stride(0), idx_per_token=per_tok, probs=probs.data_ptr(),
                           r_pre=_ptr(r_pre), F=F.data_ptr(), layer_events=ts.layer_events())
    if not tokr:
        L.check(lib.mode_dit_forward_train(C.byref(d), C.byref(eng._mw), C.byref(args), stash.data_ptr(), stash.numel(), _stream()), "forward_train")
    else:
        args.token_routing, args.tr_pre, args.tr_shifted = 1, tr_pre.data_ptr(), shifted.data_ptr()
        args.tr_topk_idx, args.tr_topk_w = tr_idx.data_ptr(), tr_w.data_ptr()
        for l in range(Ly):
            L.check(lib.mode_dit_forward_train_layer(C.byref(d), C.byref(eng._mw), C.byref(args), stash.data_ptr(), stash.numel(), l, 0, _stream()), "forward_train_layer/0")
            if model.use_argmax:                                                                    # top-k also in training (modedit.py:389)
                idx[l].copy_(tr_idx); w[l].copy_(tr_w)
            else:                                                                                   # sampled per token without replacement (modedit.py:390)
                L.check(lib.mode_moe_sample_experts(probs[l].data_ptr(), tr_expo[l].data_ptr(), N, 1, E, k, int(model.router_normalize), idx[l].data_ptr(),
                                                    w[l].data_ptr(), _stream()), "sample_experts")
            L.check(lib.mode_dit_dispatch(idx[l].data_ptr(), w[l].data_ptr(), 1, N * k, N, 1, N, E, k, meta[l].data_ptr(), _stream()), "dispatch")
            L.check(lib.mode_dit_forward_train_layer(C.byref(d), C.byref(eng._mw), C.byref(args), stash.data_ptr(), stash.numel(), l, 1, _stream()), "forward_train_layer/1")

    # ---- reference side channels (training only, modedit.py:584-593, 816-820, 930-969): load-balancing term, z-loss, expert usage, the per-block
    # views the agent's logging reads - one launch for all layers (E <= 16; wider MoEs take the torch expressions below)
    with torch.no_grad():
        counts_l = meta[:, ml.counts: ml.counts + E]
        if getattr(model, "_train_usage_dev", None) is None or model._train_usage_dev.device != counts_l.device:
            model._train_usage_dev = torch.zeros(Ly, E, dtype=torch.int64, device=dev)
        want_views = bool(getattr(model, "log_router_stats", True))     # `model.log_router_stats = False` skips the per-block views (mode_agent.py:470-511)
        Rs = N if tokr else B
        if E <= 16:
            idx, w = idx.contiguous(), w.contiguous()
            st = torch.empty(Ly * E + 2 * Ly + 2, device=dev)
            frac, lb, zl = st[:Ly * E].view(Ly, E), st[Ly * E: Ly * E + Ly], st[Ly * E + Ly: Ly * E + 2 * Ly]
            lb_mean, zl_mean = st[Ly * E + 2 * Ly], st[Ly * E + 2 * Ly + 1]
            mask = torch.empty(Ly, N, E, device=dev) if want_views else None
            L.check(lib.mode_moe_aux_stats(idx.data_ptr(), w.data_ptr(), Ly, Rr, tpr, E, k, shifted.data_ptr(), Rs, frac.data_ptr(), lb.data_ptr(),
                                           zl.data_ptr(), lb_mean.data_ptr(), zl_mean.data_ptr(), _ptr(mask), model._train_usage_dev.data_ptr(), _stream()),
                    "aux_stats")
        else:
            idx64 = (idx if per_tok else idx.unsqueeze(2).expand(Ly, B, T, k).reshape(Ly, N, k)).long()
            wtok = w if per_tok else w.unsqueeze(2).expand(Ly, B, T, k).reshape(Ly, N, k)
            mask = torch.zeros(Ly, N, E, device=dev).scatter_(2, idx64, 1.0)
            rp = torch.zeros(Ly, N, E, device=dev).scatter_(2, idx64, wtok)
            frac = mask.sum(1) / N                                                                  # [L, E] fraction of tokens per expert
            lb = E * (rp.mean(1) * frac).sum(-1)                                                    # [L]
            zl = torch.log(torch.exp(shifted).sum(-1) + 1e-6).pow(2).mean(-1)                       # [L] router z-loss per layer
            lb_mean, zl_mean = lb.mean(), zl.mean()
            model._train_usage_dev += counts_l
        model.logits_per_layer, model.probs_per_layer = [], []
        if want_views:
            logits_tok = shifted if tokr else shifted.unsqueeze(2).expand(Ly, B, T, E).reshape(Ly, N, E)
            for l, blk in enumerate(model.blocks):
                blk.logits = logits_tok[l]
                blk.probs = {"probs": probs[l].view(B, T, E) if tokr else probs[l].unsqueeze(1).expand(B, T, E), "top_k_hot": mask[l].view(B, T, E),
                             "load_balancing_term": lb[l]}
                model.logits_per_layer.append(blk.logits); model.probs_per_layer.append(blk.probs)
        for blk in model.blocks:
            blk.total_tokens_processed += N
    model._last_topk = idx

    keep_alive = (img, gl, acts, sig, e1, emb_t, img_e, goal_e, cond, idx, w, probs, shifted, r_pre, meta, act_rows, stash) + ((tr_pre, tr_idx, tr_w) if tokr else ())
    # Function inputs = the parameters that can receive a gradient.  gripper_embed.weight is dead in the reference too (modedit.py:684): left out,
    # so that DistributedDataParallel(find_unused_parameters=True) - how the reference trains, training_calvin.py:92-103 - sees it as unused.
    named = [(n, p) for n, p in eng.named_params() if p.requires_grad and n != "gripper_embed.weight"]
    names, params = [n for n, _ in named], [p for _, p in named]

    def backward(dF: torch.Tensor, dlb=None, dz=None, want_img=False, want_goal=False):
        """Runs the backward chain.  ``model.grad_mode``:

        * ``"autograd"`` (default; what a foreign trainer - Lightning, torch DDP, hook-driven clipping - needs): every parameter gradient is
          handed to autograd (views of one flat per-backward buffer with the arena's layout), so AccumulateGrad accumulates and its hooks fire
          exactly as for the reference module.
        * ``"arena"`` (set by ``FusedAdamW`` / ``ArenaGradReducer.for_model``): gradients are written straight into the gradient arena and
          ``p.grad`` points at its slices; autograd sees None.  The first backward after an optimizer step overwrites, further ones accumulate.

        ``dlb`` / ``dz``: upstream gradients of the two auxiliary router losses (None = not part of the loss).  Returns
        (d state_images, d goals, [parameter gradients])."""
        grad_mode = getattr(model, "grad_mode", "autograd")
        if grad_mode not in ("autograd", "arena"):
            raise ValueError(f"MoDeDiT.grad_mode must be 'autograd' or 'arena', got {grad_mode!r}")
        keep_aux = []
        args.shifted = args.aux_lb_coef = args.aux_z_coef = None
        if dlb is not None:                                                       # LB = mean_l E sum_e mean_n(rp[n,e]) f_{l,e}: linear in the combine weights
            coef = (frac * (dlb.reshape(()).float() * (E / (Ly * N)))).contiguous()
            keep_aux.append(coef); args.aux_lb_coef = coef.data_ptr()
        if dz is not None:
            zc = (dz.reshape(1).float() * (2.0 / (Ly * (N if tokr else B)))).contiguous()          # mean over the routing rows of a layer, mean over layers
            keep_aux.append(zc); args.aux_z_coef = zc.data_ptr(); args.shifted = shifted.data_ptr()
        d_img = torch.empty(B, model.n_img_tokens, model.obs_dim, device=dev) if want_img else None
        d_goal = torch.empty(tuple(goal_in.shape), device=dev) if want_goal else None
        args.d_state_images, args.d_goals = _ptr(d_img), _ptr(d_goal)
        ts.ensure()
        ar = eng.arena
        wsb = lib.mode_dit_train_workspace_bytes(C.byref(d), B, eng.dt)
        if ts._ws is None or ts._ws.numel() < wsb:
            ts._ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        nred = ar.bounds["no_decay"]
        all_params = [p for _, p in eng.named_params()]
        # The chain OVERWRITES its gradient buffer.  Straight into the arena when that is what the step wants (arena mode, first backward after
        # the optimizer consumed the previous sum); otherwise into a buffer of its own: handed to autograd (autograd mode), or ADDED to the
        # arena - a second backward before the optimizer step must accumulate like autograd does (the reference's training_step sums the losses
        # of several modalities, mode_agent.py:386-440).
        accumulate = grad_mode == "arena" and getattr(ar, "grad_pending", False) and any(p.grad is not None for p in all_params if p.requires_grad)
        own = grad_mode != "arena" or accumulate
        # (accumulate: zero-filled - the chain does not write the 256-byte alignment gaps between tensors, and whatever sits there would be added
        # into the arena's gaps; autograd mode only ever exposes per-tensor views)
        # Memory: in autograd mode every backward allocates one buffer of the arena's size (2.7 GB at C2) and the p.grad views AccumulateGrad keeps
        # pin it until the gradients are dropped; a two-modality training_step holds two plus the accumulated one.  The buffer is NOT zero-filled
        # (0.5 ms per backward at C2): the chain writes every element of every tensor it returns - MODE_DEBUG_GRAD_COVERAGE=1 checks exactly that
        # (NaN pre-fill, every returned view must come back finite; tests/test_gpu_train_dropin.py runs it over the shipped layouts).
        debug_cov = own and not accumulate and os.environ.get("MODE_DEBUG_GRAD_COVERAGE", "0") == "1"
        flat = (torch.zeros if accumulate else torch.empty)(ar.bounds["total"], device=dev) if own else None
        if debug_cov:
            flat.fill_(float("nan"))
        mg, gv = ts.grad_tables(flat)
        # (ABI 11) FusedAdamW(fuse_expert_step=True): the expert matrices are UPDATED by their weight-gradient GEMMs - arena mode, first backward of the
        # step only (the optimizer object refuses accumulation); their slices of the gradient arena are not written
        fopt = getattr(model, "_fused_optimizer", None)
        fz = None
        if fopt is not None and fopt.fuse_expert_step:
            if grad_mode != "arena":
                raise RuntimeError("FusedAdamW(fuse_expert_step=True) needs model.grad_mode == 'arena' (it was changed after the optimizer was built)")
            fz = fopt.fused_step_struct(accumulate)
        args.fuse_adamw = C.pointer(fz) if fz is not None else None
        L.check(lib.mode_dit_backward(C.byref(d), C.byref(eng._mw), C.byref(ts.wt), C.byref(args), stash.data_ptr(), dF.data_ptr(), C.byref(mg),
                                      ts._ws.data_ptr(), ts._ws.numel(), _stream()), "backward")
        if fz is not None:
            fopt.fused_backward_done()
        if debug_cov:
            holes = [n for n in names if not bool(torch.isfinite(gv[n]).all())]
            if holes:
                raise RuntimeError(f"backward chain left gradient elements unwritten (or non-finite) in: {holes[:8]}")
        if grad_mode != "arena":
            return d_img, d_goal, [gv[n].view(p.shape) for n, p in zip(names, params)]
        if accumulate:
            ar.grad[:nred].add_(flat[:nred])
            # the per-block events the chain recorded mid-way now precede this add: re-record them behind it, or an overlapped optimizer /
            # reducer (FusedAdamW.step(overlap=True), ArenaGradReducer) gated on events[l] would read block l before its sum is complete
            if ts.events is not None:
                for ev in ts.events:
                    ev.record()
            gv = ar.g_by_name
        ar.grad_pending = True                                                  # cleared by FusedAdamW.step()/zero_grad() or p.grad = None
        for n, p in zip(names, params):
            if p.grad is None or p.grad.data_ptr() != gv[n].data_ptr():
                p.grad = gv[n].view(p.shape)
        return d_img, d_goal, [None] * len(params)

    run.backward, run.keep = backward, keep_alive
    run.F_shape, run.device = tuple(F.shape), dev
    run.outs = (F, lb_mean, zl_mean)                                        # taken out of `run` by the node (see _DitTrainFn.forward)
    Fo, lb_o, z_o = _DitTrainFn.apply(run, img_in, goal_in, *params)
    model._aux_losses = (lb_o, z_o)                                             # what load_balancing_loss() / compute_router_z_loss() return in training
    return Fo


def edm_loss(denoiser, state, action, goal, noise, sigma):
    """``GCDenoiser.loss`` (score_wrappers.py:45-63) on the HIP chain: noised input and c_in scaling in one launch, target / MSE / dF in
    another; returns (loss, model_output) like the reference."""
    m = denoiser.inner_model
    eng: DitEngine = m.engine
    lib, dev = eng.lib, eng.device
    for nm, t in (("action", action), ("noise", noise)):
        if torch.is_grad_enabled() and torch.is_tensor(t) and t.requires_grad:
            raise NotImplementedError(f"GCDenoiser.loss (HIP): {nm} requires grad, but the chain returns no input gradients")
    f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    a, nz = f(action), f(noise)
    B = a.shape[0]
    n = a[0].numel()
    sig = f(sigma).reshape(-1)
    if sig.numel() == 1:
        sig = sig.expand(B).contiguous()
    xs = torch.empty_like(a)
    sd = float(denoiser.sigma_data)
    L.check(lib.mode_edm_noise_scale(a.data_ptr(), nz.data_ptr(), sig.data_ptr(), sd, B, n, xs.data_ptr(), _stream()), "edm_noise_scale")
    F = m(state, xs, goal, sig)
    loss = torch.empty(1, device=dev)
    dF = torch.empty_like(a)
    L.check(lib.mode_edm_loss(F.detach().contiguous().data_ptr(), a.data_ptr(), nz.data_ptr(), sig.data_ptr(), sd, B, n, loss.data_ptr(), dF.data_ptr(), _stream()),
            "edm_loss")
    return _EdmLossFn.apply(F, loss, dF), F


def diffusion_loss(denoiser, perceptual_emb, latent_goal, actions, sample_density=None):
    """``MoDEAgent.diffusion_loss`` (mode_agent.py:659-672): train mode, sigma ~ density (log-logistic by default), eps ~ N(0, 1)."""
    from .utils import make_sample_density
    denoiser.train()
    density = sample_density or make_sample_density("loglogistic", sigma_data=float(denoiser.sigma_data))
    sigmas = density(shape=(len(actions),), device=actions.device).to(actions.device)
    noise = torch.randn_like(actions)
    loss, _ = denoiser.loss(perceptual_emb, actions, latent_goal, noise, sigmas)
    return loss


def training_step(denoiser, batch, entropy_gamma: float = 0.0, router_z_delta: float = 0.0, sample_density=None):
    """``MoDEAgent.training_step`` (mode_agent.py:386-440) for precomputed embeddings: ``batch`` maps a modality name to
    ``{"perceptual_emb": {"state_images": ...}, "latent_goal": ..., "actions": ...}`` (what ``compute_input_embeddings`` hands over);
    per modality ``act_loss (+ entropy_gamma * LB) (+ router_z_delta * Z)``, summed, divided by the number of modalities.
    Returns (total_loss, action_loss, aux) with aux = the last modality's LB / Z values (what the reference logs)."""
    inner = denoiser.inner_model
    total, action_loss, aux = None, None, {}
    for _, db in batch.items():
        act = diffusion_loss(denoiser, db["perceptual_emb"], db["latent_goal"], db["actions"], sample_density)
        t = act
        if entropy_gamma > 0:
            aux["load_balancing_loss"] = inner.load_balancing_loss()
            t = t + aux["load_balancing_loss"] * entropy_gamma
        if router_z_delta > 0:
            aux["router_z_loss"] = inner.compute_router_z_loss()
            t = t + router_z_delta * aux["router_z_loss"]
        total = t if total is None else total + t
        action_loss = act if action_loss is None else action_loss + act
    n = len(batch)
    return total / n, action_loss / n, aux
