"""Fused AdamW over the flat parameter arena (replaces ``torch.optim.AdamW`` as configured by MoDEAgent.configure_optimizers,
mode/models/mode_agent.py:365-392: two groups — decayed weights / undecayed biases — betas and lr from the Hydra config).

One HIP kernel per group walks ``(p, g, m, v)`` of the arena region (16 B read + 12 B written per element, HBM-bound: ~20 GB per step
for the 685 M-parameter denoiser) and also emits the bf16 compute shadow, so the next forward needs no separate cast pass.
``gripper_embed.weight`` (never receives a gradient in the reference, hence never updated by its optimizer) lives in the arena's
``dead`` region and is not touched.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Tuple

import torch

from . import _lib as L
from .engine import _stream


class FusedAdamW:
    """``fuse_expert_step=True`` (opt-in; single process, no gradient accumulation - what the reference's training loop is,
    conf/config_libero.yaml:45): the expert matrices ``w1`` / ``w2`` of every block - 604 M of the 686 M parameters - are updated INSIDE their
    weight-gradient GEMMs (``ModeAdamWFuse``, include/mode_hip.h; ``csrc/gemm_bf16_tr.hip`` EPI = 2): the backward chain reads p / m / v, applies
    this class's arithmetic to the fp32 accumulators and writes p / m / v and the bf16 shadow back - the gradients of those tensors are never
    stored or re-read (26 instead of 34 bytes per parameter) and no optimizer pass is left to compete with the backward for them.  ``step()``
    then covers the remaining 12 %.  The backward picks the hyper-parameters up from this object when it runs, so set the step's learning rate
    BEFORE ``loss.backward()`` (a scheduler stepped after ``optimizer.step()``, as Lightning does, satisfies that) and pass a gradient scale via
    ``opt.fused_grad_scale``.  Results are bit-identical to the two-pass update (same gradient bits, same expression order).
    ``p.grad`` of the expert matrices is not produced in this mode; ``opt.fused_grad_sq()`` returns their squared gradient norm for logging
    (mode_agent.py:304-363).

    The expert update is IRREVERSIBLE once ``loss.backward()`` has run, so everything that can refuse the step is checked before the chain launches
    (``fused_step_struct``: arena identity, world size, accumulation, frozen experts; a registered ``fused_ema`` is allocated there, i.e. from the
    pre-update weights).  ``step()`` never leaves a half-applied step behind: when its arguments disagree with what the backward applied (a different
    ``grad_scale``, a multi-rank reducer, an EMA that did not exist before the backward) it first COMPLETES the step with the backward's settings -
    remaining 12 % of the parameters, ``step_count``, shadow versions - and only then raises.  Skipping ``step()`` after a fused backward (e.g. a
    non-finite-loss guard) is unsupported - the experts have already moved; call ``finish_fused_step()`` to bring the rest of the arena to the same
    step before continuing."""

    def __init__(self, model, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05,
                 fuse_expert_step: bool = False, fused_side_stream: bool = True):
        self.model = model
        eng = model.engine                                   # adopts the parameters into the arena
        self.eng, self.arena = eng, eng.arena
        model.grad_mode = "arena"                            # this optimizer reads the flat gradient arena: the backward chain writes there directly
        n = self.arena.bounds["no_decay"]                    # decay + no_decay regions (dead region excluded)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=eng.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=eng.device)
        self.step_count = 0
        self._side = None
        self._ema_now = (None, 0.0)
        self._frozen = []
        self._master_sharded = self._state_sharded = False
        self._sharded_emas = []                                  # ArenaEMAs updated inside ZeRO-1 steps since the last gather_state(): current on each rank's shards only
        self.param_groups = [dict(name="decay", lr=lr, betas=betas, eps=eps, weight_decay=weight_decay),
                             dict(name="no_decay", lr=lr, betas=betas, eps=eps, weight_decay=0.0)]
        # ---- expert matrices updated by their weight-gradient GEMMs (see the class docstring)
        self.fuse_expert_step = bool(fuse_expert_step)
        self.fused_grad_scale = 1.0
        self.fused_ema = None                                    # an ArenaEMA whose schedule the fused update honours (else pass ema= to step(): a separate pass over the expert ranges)
        self._fused_pending = False                              # a backward has applied the expert update of step step_count + 1
        self._fused_struct = self._fused_gsq = None
        # the fused weight-gradient + optimizer launches are HBM-bound, the data-gradient chain MFMA-bound: the backward runs them on a second stream
        # (ModeAdamWFuse.side_stream; joined before mode_dit_backward returns).  False = everything on the one stream.
        self.fused_side_stream = bool(fused_side_stream) and os.environ.get("MODE_FUSED_SIDE_STREAM", "1") == "1"
        self._fused_side = None                                  # (stream, [4 events], ctypes array of their handles)
        prev = getattr(model, "_fused_optimizer", None)
        if prev is not None and prev is not self:
            # a NEW optimizer takes the model over: an earlier fuse_expert_step optimizer must not keep updating the experts inside the backward
            if prev._fused_pending:
                raise RuntimeError("another FusedAdamW of this model has a fused backward pending: call its step() / finish_fused_step() first")
            model._fused_optimizer = None
        if self.fuse_expert_step:
            self.fuse_expert_step = False
            self.set_fuse_expert_step(True)                      # model._fused_optimizer = self: training.py's backward asks this object for the ModeAdamWFuse of the step

    def set_fuse_expert_step(self, on: bool) -> None:
        """Switch the fused expert step on / off between optimizer steps (same object, same moment buffers: `bench.py` times both modes on the state it
        allocated at start-up).  Refused while a fused backward is pending."""
        if self._fused_pending:
            raise RuntimeError("set_fuse_expert_step(): a fused backward is pending - call step() / finish_fused_step() first")
        on = bool(on)
        if on:
            if self.eng.compute_dtype != "bf16":
                raise ValueError("fuse_expert_step needs the bf16 compute mode (the fused epilogue lives in the bf16 weight-gradient GEMM)")
            if self.model.embed_dim % 128:
                raise ValueError("fuse_expert_step needs embed_dim % 128 == 0 (128-column tiles of the fused weight-gradient launches)")
            self.model._fused_optimizer = self
        elif getattr(self.model, "_fused_optimizer", None) is self:
            self.model._fused_optimizer = None
        self.fuse_expert_step = on

    def reset_state(self) -> None:
        """Back to a freshly constructed optimizer WITHOUT re-allocating: moments zeroed in place, step count 0."""
        if self._fused_pending:
            raise RuntimeError("reset_state(): a fused backward is pending - call step() / finish_fused_step() first")
        if self._state_sharded or self._master_sharded:
            raise RuntimeError("reset_state(): ZeRO-1 state is sharded - call gather_state(reducer) first")
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        self.step_count = 0
        self.arena.grad_pending = False

    def zero_grad(self, set_to_none: bool = False) -> None:
        """The next backward overwrites the gradient arena instead of accumulating into it (no memset pass: the backward chain writes every
        element of the reducible region, exact zeros for un-routed experts included)."""
        self.arena.grad_pending = False

    # ---- fused expert step ---------------------------------------------------------------------------------------------------
    def _expert_ranges(self):
        """Arena element ranges [lo, hi) of every block's w1 and w2 (decay region), ascending."""
        ar, Ly = self.arena, self.model.num_layers
        out = []
        for i in range(Ly):
            for nm in (f"l{i}.w1", f"l{i}.w2"):
                lo = ar.offset(nm)
                out.append((lo, lo + ar.w[nm].numel()))
        return sorted(out)

    def fused_step_struct(self, accumulate: bool):
        """Called by the backward (training.py) right before the chain runs: the ModeAdamWFuse of optimizer step ``step_count + 1``, or None when
        this backward must write gradients as usual.  Raises where a fused update would be silently wrong."""
        if not self.fuse_expert_step:
            return None
        eng, ar = self.eng, self.arena
        if eng.arena is not ar:
            raise RuntimeError("the parameter arena was rebuilt (model.to()/half()?): create a new FusedAdamW")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise NotImplementedError("fuse_expert_step is a single-process mode (the expert matrices would be updated from this rank's gradients only): "
                                      "construct FusedAdamW(fuse_expert_step=False) under torch.distributed with more than one rank")
        if accumulate or self._fused_pending:
            raise RuntimeError("fuse_expert_step: a second backward before optimizer.step() (gradient accumulation) cannot be fused - the first "
                               "backward has already updated the expert matrices; construct FusedAdamW(fuse_expert_step=False)")
        if any(not p.requires_grad for n, p in eng.named_params() if ".experts." in n and n.endswith("weight")):
            raise NotImplementedError("fuse_expert_step with frozen expert matrices")
        ar.ensure_grad(self.model)
        gd = self.param_groups[0]
        ema_base, rate = None, 0.0
        if self.fused_ema is not None:
            self.fused_ema.ensure(ar)                            # (first call: a copy of the PRE-update weights, as the callback's on_train_start makes it)
        if self.fused_ema is not None and self.fused_ema.should_apply(self.step_count + 1):
            ema_base, rate = self.fused_ema.flat.data_ptr(), 1.0 - self.fused_ema.get_decay(self.step_count + 1)
        if self._fused_gsq is None:
            n = int(eng.lib.mode_adamw_fuse_gsq_floats(C.byref(eng.dims)))
            self._fused_gsq = torch.zeros(n, dtype=torch.float32, device=eng.device)
        side_stream, side_events = None, None
        if self.fused_side_stream and not torch.cuda.is_current_stream_capturing():
            if self._fused_side is None:
                stream = torch.cuda.Stream(device=eng.device)
                evs = [torch.cuda.Event() for _ in range(4)]
                for ev in evs:
                    ev.record()                                                 # forces creation of the underlying hipEvent
                self._fused_side = (stream, evs, (C.c_void_p * 4)(*[ev.cuda_event for ev in evs]))
            side_stream, side_events = self._fused_side[0].cuda_stream, C.cast(self._fused_side[2], C.c_void_p)
        st = L.ModeAdamWFuse(side_stream=side_stream, side_events=side_events, grad_base=ar.grad.data_ptr(), param_base=ar.flat.data_ptr(), exp_avg_base=self.exp_avg.data_ptr(),
                             exp_avg_sq_base=self.exp_avg_sq.data_ptr(), lp_base=ar.lp.data_ptr() if ar.lp is not None else None, ema_base=ema_base,
                             ema_rate=float(rate), lr=float(gd["lr"]), beta1=float(gd["betas"][0]), beta2=float(gd["betas"][1]), eps=float(gd["eps"]),
                             weight_decay=float(gd["weight_decay"]), step=self.step_count + 1, grad_scale=float(self.fused_grad_scale),
                             gsq=self._fused_gsq.data_ptr(), gsq_capacity=self._fused_gsq.numel())
        self._fused_struct = st                                  # keeps the ctypes object alive across the call
        return st

    def fused_backward_done(self) -> None:
        """The backward chain has updated the expert matrices (masters AND bf16 shadow, in step with each other).  The arena's version moves in
        ``step()`` (``weights_updated``), as for every update this optimizer makes through raw pointers."""
        self._fused_pending = True

    def fused_grad_sq(self) -> torch.Tensor:
        """Squared L2 norm of the (scaled) expert-matrix gradients of the last fused backward (device scalar)."""
        if self._fused_gsq is None:
            raise RuntimeError("no fused backward has run yet")
        return self._fused_gsq.double().sum()

    def finish_fused_step(self) -> None:
        """Completes a fused step whose expert update already ran inside ``loss.backward()``: the remaining parameters take the same optimizer step
        (the backward's ``fused_grad_scale``, the registered ``fused_ema``), ``step_count`` advances, the pending flag clears.  This is what
        ``step()`` does in fused mode; call it directly where a training loop would otherwise SKIP ``step()`` (a skipped step is not supported in
        this mode - see the class docstring).  No-op when nothing is pending."""
        if self._fused_pending:
            self.step()

    def _frozen_ranges(self):
        """Arena element ranges of parameters with ``requires_grad == False`` (``freeze_router()`` for fine-tuning, mode_agent.py:762-766):
        torch's AdamW skips tensors without a gradient, so these slices are left untouched."""
        ar = self.arena
        base = ar.flat.data_ptr()
        spans = sorted(((p.data_ptr() - base) // 4, (p.data_ptr() - base) // 4 + p.numel()) for _, p in self.eng.named_params() if not p.requires_grad)
        merged = []
        for lo, hi in spans:
            if merged and lo <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], hi)
            else:
                merged.append([lo, hi])
        for lo, hi in merged:
            if lo % 4 or hi % 4:
                raise NotImplementedError("frozen parameter ranges must start and end on 16-byte boundaries of the arena")
        return merged

    def _launch(self, lo: int, hi: int, gp, lp, grad_scale: float) -> None:
        for flo, fhi in self._frozen:                          # carve frozen slices (and, after a fused backward, the expert matrices) out of [lo, hi)
            if flo < hi and fhi > lo:
                if lo < flo:
                    self._launch_raw(lo, flo, gp, lp, grad_scale)
                lo = max(lo, fhi)
                if lo >= hi:
                    return
        if lo < hi:
            self._launch_raw(lo, hi, gp, lp, grad_scale)

    def _launch_raw(self, lo: int, hi: int, gp, lp, grad_scale: float) -> None:
        ar = self.arena
        ema, rate = self._ema_now
        L.check(self.eng.lib.mode_adamw_step(ar.flat[lo:hi].data_ptr(), ar.grad[lo:hi].data_ptr(), self.exp_avg[lo:hi].data_ptr(),
                                             self.exp_avg_sq[lo:hi].data_ptr(), hi - lo, float(gp["lr"]), float(gp["betas"][0]),
                                             float(gp["betas"][1]), float(gp["eps"]), float(gp["weight_decay"]), self.step_count,
                                             float(grad_scale), None if lp is None else lp[lo:hi].data_ptr(),
                                             None if ema is None else ema.flat[lo:hi].data_ptr(), float(rate), _stream()), "adamw_step")

    def _block_slices(self):
        """[(lo, hi, layer)] of the per-block weight slices (arena order = backward order) and the remaining decay-region ranges."""
        ar, Ly = self.arena, self.model.num_layers
        starts = [ar.offset(f"l{i}.wqkv") for i in range(Ly)]
        bounds = sorted(starts) + [ar.offset("pos")]
        end = {lo: hi for lo, hi in zip(bounds[:-1], bounds[1:])}
        blocks = [(starts[i], end[starts[i]], i) for i in reversed(range(Ly))]
        rest = [(0, starts[Ly - 1]), (ar.offset("pos"), ar.bounds["decay"])]
        return blocks, rest

    @torch.no_grad()
    def _step_zero1(self, grad_scale: float, reducer, gather: str, lp, gd, gn) -> None:
        """ZeRO-1 over the data-parallel ranks: per block slice {reduce-scatter of the gradients (behind the block's backward event) -> AdamW on
        THIS rank's 1/world shard only -> all-gather of the updated weights}, everything chained per slice on side streams.  The HBM-bound
        optimizer pass shrinks by the world size (20.5 GB -> 2.6 GB per rank at 8 GPUs: 3.4 -> ~0.45 ms) and the wire carries a reduce-scatter +
        an all-gather (each half an all-reduce's bytes).  ``gather='fp32'``: the fp32 weights are gathered (every rank keeps exact masters,
        the bf16 shadow is re-cast locally); ``gather='bf16'``: only the bf16 compute shadow is gathered (half the bytes again) - the fp32
        masters of the OTHER ranks' shards are then refreshed from it (bf16-rounded) and ``gather_master()`` restores exact masters everywhere
        before a checkpoint."""
        eng, ar = self.eng, self.arena
        if self._side is None:
            self._side = torch.cuda.Stream(device=eng.device, priority=int(os.environ.get("MODE_OPT_PRIO", "0")))
        cur = torch.cuda.current_stream()
        n_red = ar.bounds["no_decay"]
        dec = ar.bounds["decay"]
        parts, done = [], []
        with torch.cuda.stream(self._side):
            # collectives run in issue order: per slice reduce-scatter -> (shard update) -> all-gather, so that the all-gather of block l travels
            # while the earlier blocks are still back-propagating instead of queueing behind every reduce-scatter
            for si in range(len(reducer.slices)):
                with torch.cuda.stream(cur):
                    (lo, hi, slo, shi, ev), = reducer.reduce_scatter_async(only=si)
                parts.append((lo, hi, slo, shi, ev))
                if ev is not None:
                    self._side.wait_event(ev)
                # the shard may straddle the decay / no-decay boundary
                if slo < min(shi, dec):
                    self._launch(slo, min(shi, dec), gd, lp, grad_scale)
                if max(slo, dec) < shi:
                    self._launch(max(slo, dec), shi, gn, lp, grad_scale)
                upd = torch.cuda.Event(); upd.record(self._side)
                if gather == "bf16" and lp is not None:
                    done.append((lo, hi, slo, shi, reducer.all_gather_async(lp, lo, hi, after=upd)))
                else:
                    done.append((lo, hi, slo, shi, reducer.all_gather_async(ar.flat, lo, hi, after=upd)))
            for lo, hi, slo, shi, e in done:                           # refresh what was not gathered
                if e is not None:
                    self._side.wait_event(e)
                if gather == "bf16" and lp is not None:
                    if slo > lo:
                        ar.flat[lo:slo].copy_(lp[lo:slo])
                    if shi < hi:
                        ar.flat[shi:hi].copy_(lp[shi:hi])
                elif lp is not None:
                    if slo > lo:
                        lp[lo:slo].copy_(ar.flat[lo:slo])
                    if shi < hi:
                        lp[shi:hi].copy_(ar.flat[shi:hi])
        cur.wait_stream(self._side)
        self._master_sharded = gather == "bf16" and lp is not None and reducer.world > 1
        self._state_sharded = reducer.world > 1                  # exp_avg / exp_avg_sq are current on this rank's shards only

    def _validate_zero1(self, reducer, zero1, ema) -> None:
        """Everything that can refuse a ZeRO-1 step, checked BEFORE any state changes or any collective is issued (a refusal half-way would leave
        the ranks with different step counts / partially updated slices)."""
        if zero1 not in ("fp32", "bf16"):
            raise ValueError("zero1 must be None, 'fp32' or 'bf16'")
        # (an `ema=` is sharded like the moments: each rank averages its own shard - where it holds exact fp32 masters in both gather modes - and
        #  gather_state() completes it; mode/callbacks/ema.py:101-126 runs the callback on every rank after every optimizer step)
        ar = self.arena
        n_red, dec, world = ar.bounds["no_decay"], ar.bounds["decay"], reducer.world
        spans = sorted((sl[0], sl[1]) for sl in reducer.slices)
        if not spans or spans[0][0] != 0 or spans[-1][1] != n_red or any(a[1] != b[0] for a, b in zip(spans[:-1], spans[1:])):
            raise RuntimeError("ZeRO-1 step: the reducer's slices do not tile the optimised arena exactly once")
        for lo, hi in spans:
            if (hi - lo) % (4 * world) or lo % 4:
                raise ValueError(f"ZeRO-1 step: slice [{lo}, {hi}) does not split into {world} shards of whole 16-byte groups (mode_adamw_step needs "
                                 "n % 4 == 0 and 16-byte aligned pointers)")
        if dec % 4:
            raise ValueError("ZeRO-1 step: the decay / no-decay boundary of the arena is not 16-byte aligned")

    @torch.no_grad()
    def gather_state(self, reducer) -> None:
        """After ZeRO-1 steps every rank holds current Adam moments (and, with ``zero1='bf16'``, exact fp32 masters) for its OWN shards only.
        All-gathers ``exp_avg`` / ``exp_avg_sq`` (and every EMA that was passed to those steps) per slice and the masters, so that ``state_dict()`` / a checkpoint written by any rank - or a
        stand-alone ``ArenaEMA.update`` - sees the complete, exact state.  Collective: all ranks call it."""
        self.gather_master(reducer)
        if getattr(self, "_state_sharded", False):
            evs = []
            for sl in reducer.slices:
                for buf in (self.exp_avg, self.exp_avg_sq):
                    evs.append(reducer.all_gather_async(buf, sl[0], sl[1], after=None))
            for em in self._sharded_emas:                             # EMAs updated inside ZeRO-1 steps: every rank averaged its own shards
                for sl in reducer.slices:
                    evs.append(reducer.all_gather_async(em.flat, sl[0], sl[1], after=None))
            for e in evs:
                if e is not None:
                    torch.cuda.current_stream().wait_event(e)
            for em in self._sharded_emas:
                em._sharded = False
            self._sharded_emas = []
            self._state_sharded = False

    @torch.no_grad()
    def gather_master(self, reducer) -> None:
        """After ``zero1='bf16'`` steps: all-gather the exact fp32 masters of every shard (call before ``state_dict()`` / a checkpoint)."""
        if not getattr(self, "_master_sharded", False):
            return
        ev = [reducer.all_gather_async(self.arena.flat, lo, hi, after=None) for lo, hi, *_ in [(s[0], s[1]) for s in reducer.slices]]
        for e in ev:
            if e is not None:
                torch.cuda.current_stream().wait_event(e)
        self._master_sharded = False

    @torch.no_grad()
    def step(self, grad_scale=None, overlap: bool = False, reducer=None, ema=None, zero1=None) -> None:
        """One AdamW update of the whole arena.

        Default: (exchange gradients through ``reducer`` — its collectives overlap the backward —, then) two launches over the decay / no-decay
        regions on the current stream.

        ``overlap=True``: call right after ``loss.backward()`` returned and do not touch the gradients in between (clipping goes through
        ``grad_scale``).  The backward kernels are still executing; the update of block l's 228 MB weight slice is queued on a side stream
        behind the event the backward chain records when block l's gradients are complete, so the HBM-bound optimizer pass (a quarter of a
        serial step) runs underneath the backward of the earlier blocks (block l's backward only reads block l's weights, so updating later
        blocks early is safe).  Measured on MI355X inside one process, interleaved rounds (scripts/train_overlap_probe.py): 14.65-14.75 ms
        per step against 15.35-15.45 ms serial with one AdamW workgroup per CU (``adamw_blocks`` = 256, the default), 15.2-15.7 ms
        overlapped / 15.8 ms serial with 2048 workgroups — a streaming pass that oversubscribes the CUs raises the memory latency the
        fill-bound backward GEMMs are sensitive to.  ``bench.py --mode train`` uses it; the keyword default stays False because an
        overlapped update must not be preceded by in-place gradient edits.  ``reducer`` (an ``ArenaGradReducer``) chains the data-parallel
        exchange in front of each slice's update on the same events; its 1/world scale is applied here.

        ``ema`` (an ``ArenaEMA``): the moving average of the weights is updated in the same pass whenever the callback's schedule says so
        for this step (the reference's EMA callback runs right after every optimizer step, mode/callbacks/ema.py:128-142)."""
        eng = self.model.engine
        ar = eng.arena
        if ar is not self.arena:
            raise RuntimeError("the parameter arena was rebuilt (model.to()/half()?): create a new FusedAdamW")
        if ar.grad is None:
            raise RuntimeError("no gradients: run a training forward + backward first")
        use_zero1 = bool(zero1) and reducer is not None and eng.device.type == "cuda"
        fused = self._fused_pending
        if self.fuse_expert_step and not fused:
            raise RuntimeError("fuse_expert_step: optimizer.step() without a fused backward since the last step (was the loss back-propagated "
                               "through MoDeDiT in training mode?)")
        deferred = None                                         # fused mode: a disagreement with what the backward applied - the step is COMPLETED first (class docstring)
        if fused:
            done_msg = "; the step was completed with the backward's settings (parameters, moments and step count are consistent)"
            if use_zero1 or (reducer is not None and reducer.world > 1):
                deferred = NotImplementedError("fuse_expert_step is a single-process mode: the expert matrices were updated from this rank's gradients only" + done_msg)
                reducer, use_zero1 = None, False
            if grad_scale is not None and float(grad_scale) != float(self.fused_grad_scale):
                deferred = ValueError(f"step(grad_scale={grad_scale}) differs from the scale the fused backward applied ({self.fused_grad_scale}): set "
                                      "opt.fused_grad_scale before loss.backward()" + done_msg)
            grad_scale = self.fused_grad_scale
            if ema is not None and ema is not self.fused_ema and ema.flat is None and ema.should_apply(self.step_count + 1) \
                    and ema.get_decay(self.step_count + 1) != 0.0:
                # this EMA did not exist when the backward updated the experts: its first copy would start from post-update expert weights (with a zero
                # decay - the schedule's first step - the average equals the new weights either way and nothing is lost)
                deferred = RuntimeError("fuse_expert_step: register the EMA as opt.fused_ema (or call ema.ensure(arena)) BEFORE the first backward - it is "
                                        "initialised from the pre-update weights" + done_msg)
        if grad_scale is None:
            grad_scale = 1.0
        if use_zero1:
            self._validate_zero1(reducer, zero1, ema)
        self.step_count += 1
        self._frozen = self._frozen_ranges()
        if fused:                                               # the expert matrices are done: the passes below skip them exactly like frozen tensors
            spans = sorted([list(x) for x in self._frozen] + [list(x) for x in self._expert_ranges()])
            merged = []
            for lo_, hi_ in spans:
                if merged and lo_ <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], hi_)
                else:
                    merged.append([lo_, hi_])
            self._frozen = merged
            self._fused_pending = False
        self._ema_now = (None, 0.0)
        fe = self.fused_ema if fused else None                   # the backward applied this EMA to the expert matrices iff its schedule says so for this step
        fe_rest = fe is not None and fe is not ema and fe.should_apply(self.step_count)
        if ema is not None and ema.should_apply(self.step_count):
            ema.ensure(ar)
            self._ema_now = (ema, 1.0 - ema.get_decay(self.step_count))
            ema.mark_applied(self.step_count)
            if use_zero1 and reducer.world > 1:
                ema._sharded = True                              # current on this rank's shards only until gather_state()
                self._sharded_emas = [e for e in self._sharded_emas if e is not ema] + [ema]
        lp = ar.lp if eng.compute_dtype == "bf16" else None
        gd, gn = self.param_groups
        train = getattr(eng, "_train", None)
        events = train.events if (train is not None and train.events is not None) else None
        if reducer is not None and reducer.world > 1:
            grad_scale = grad_scale * (1.0 if reducer.average else 1.0 / reducer.world)
        if use_zero1:
            self._step_zero1(grad_scale, reducer, zero1, lp, gd, gn)
        elif not overlap or events is None:
            if reducer is not None:
                reducer.reduce()
            self._launch(0, ar.bounds["decay"], gd, lp, grad_scale)
            self._launch(ar.bounds["decay"], ar.bounds["no_decay"], gn, lp, grad_scale)
        else:
            if self._side is None:
                self._side = torch.cuda.Stream(device=eng.device, priority=int(os.environ.get("MODE_OPT_PRIO", "0")))
                # the per-block passes run BESIDE the next backward chains: tell the library, so that its large backward GEMMs keep the ring
                # kernels that leave CU resources free (include/mode_hip.h "bwd_coexec"; process-wide like every option).  Not with the expert
                # matrices updated inside the backward: what is left per block is a 4-M-parameter pass (22 us) that fits between the kernels
                if not self.fuse_expert_step:
                    eng.lib.mode_set_option(b"bwd_coexec", 1)
            cur = torch.cuda.current_stream()
            blocks, rest = self._block_slices()
            done = reducer.reduce_async() if (reducer is not None and reducer.world > 1) else None    # {(lo, hi): event after the exchange}
            with torch.cuda.stream(self._side):
                for lo, hi, i in blocks:
                    self._side.wait_event(done[(lo, hi)] if done is not None else events[i])
                    self._launch(lo, hi, gd, lp, grad_scale)
                self._side.wait_stream(cur)                                    # everything else needs the whole backward
                if done is not None:
                    for ev in done.values():
                        self._side.wait_event(ev)
                for lo, hi in rest:
                    self._launch(lo, hi, gd, lp, grad_scale)
                self._launch(ar.bounds["decay"], ar.bounds["no_decay"], gn, lp, grad_scale)
            cur.wait_stream(self._side)
        if fused and self._ema_now[0] is not None and self.fused_ema is not self._ema_now[0]:
            ema_, rate_ = self._ema_now                          # the fused epilogue did not know about this EMA: one stand-alone pass over the expert ranges
            for lo_, hi_ in self._expert_ranges():
                L.check(eng.lib.mode_ema_update(ema_.flat[lo_:hi_].data_ptr(), ar.flat[lo_:hi_].data_ptr(), hi_ - lo_, float(rate_), _stream()), "ema_update")
        if fe_rest:
            # the fused epilogue applied the registered EMA to the expert matrices; the rest of the arena follows here (step() got no ema=, or another one)
            ema_ = fe
            rate_ = 1.0 - ema_.get_decay(self.step_count)
            n_all = ar.bounds["total"]
            prev = 0
            for lo_, hi_ in self._expert_ranges() + [(n_all, n_all)]:
                if prev < lo_:
                    L.check(eng.lib.mode_ema_update(ema_.flat[prev:lo_].data_ptr(), ar.flat[prev:lo_].data_ptr(), lo_ - prev, float(rate_), _stream()), "ema_update")
                prev = hi_
            ema_.mark_applied(self.step_count)
        eng.weights_updated(lp_synced=lp is not None)
        ar.grad_pending = False                                  # gradients consumed: the next backward starts a fresh sum
        if deferred is not None:
            raise deferred

    # ---- checkpointing (same information as torch's optimizer state, flat)
    def state_dict(self) -> Dict:
        if getattr(self, "_state_sharded", False) or getattr(self, "_master_sharded", False):
            raise RuntimeError("FusedAdamW.state_dict(): after ZeRO-1 steps the moments (and with zero1='bf16' the fp32 masters) are current on "
                               "each rank's own shards only - call opt.gather_state(reducer) on every rank first")
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd: Dict) -> None:
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


class FlatAdamW:
    """``torch.optim.AdamW`` for an ARBITRARY list of fp32 device parameters - the perceptual encoders next to the denoiser (``mode_agent.py:289``: the
    reference hands every parameter of the agent to one AdamW) - as ONE ``mode_adamw_step`` launch per parameter group instead of torch's multi-tensor
    loop (~37 launches / 1.6 ms per step for two FiLM-ResNet-50s).  The parameters are re-pointed at views of one flat buffer per group (names, shapes,
    strides incl. channels_last, ``state_dict`` untouched - what ``arena.py`` does for the denoiser) and their ``.grad`` at views of a flat gradient
    buffer that autograd accumulates into in place.  Same arithmetic as ``FusedAdamW`` (``adamw_update_f``), i.e. torch's single-tensor order.

    Like torch, a parameter whose ``.grad`` is None in a step is SKIPPED (no decay, moments untouched: FiLM modules that saw no conditioning vector,
    parameters frozen later): the flat launch is cut into the runs of tensors that do have a gradient.  The update writes through raw pointers, so the
    parameters' version counters are bumped explicitly (cached bf16 weight shadows must notice); a parameter whose storage was re-pointed after
    construction (``module.to()`` / ``.half()``) is refused in ``step()`` rather than silently left behind."""

    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [dict(params=groups)]
        self.param_groups, self._flat = [], []
        self.lib = L.load()
        self.step_count = 0
        for g in groups:
            ps = [p for p in g["params"] if p.requires_grad]
            if not ps:
                continue
            dev = ps[0].device
            for p in ps:
                if p.dtype != torch.float32 or p.device != dev or dev.type != "cuda":
                    raise ValueError("FlatAdamW: fp32 parameters on one ROCm device per group")
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if not dense:
                    raise ValueError("FlatAdamW: parameters must be dense (contiguous or channels_last)")
            sizes = [(p.numel() + 3) // 4 * 4 for p in ps]                        # every tensor starts on a 16-byte boundary
            n = sum(sizes)
            flat, grad = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
            views, o = [], 0
            with torch.no_grad():
                for p, sz in zip(ps, sizes):
                    w = torch.as_strided(flat, p.shape, p.stride(), o)
                    w.copy_(p)
                    p.data = w
                    views.append(torch.as_strided(grad, p.shape, p.stride(), o))
                    o += sz
            grp = {k: v for k, v in g.items() if k != "params"}
            grp.setdefault("lr", lr); grp.setdefault("betas", betas); grp.setdefault("eps", eps); grp.setdefault("weight_decay", weight_decay)
            grp["params"] = ps
            self.param_groups.append(grp)
            offs, o = [], 0
            for sz in sizes:
                offs.append((o, o + sz)); o += sz
            self._flat.append(dict(flat=flat, grad=grad, views=views, exp_avg=torch.zeros(n, device=dev), exp_avg_sq=torch.zeros(n, device=dev), n=n,
                                   spans=offs, ptrs=[p.data_ptr() for p in ps]))

    def zero_grad(self, set_to_none: bool = True) -> None:
        """``set_to_none=True`` (default, like torch): drop the gradients - the next backward hands autograd-owned tensors to ``.grad`` (no accumulation
        kernel per parameter) and ``step()`` gathers them into the flat buffer with ONE multi-tensor copy.  ``False``: ``.grad`` become zeroed views of
        the flat buffer and autograd accumulates into them in place (one small add per parameter and backward: 350 launches for two ResNet-50s)."""
        for grp, f in zip(self.param_groups, self._flat):
            if set_to_none:
                for p in grp["params"]:
                    p.grad = None
            else:
                f["grad"].zero_()
                for p, gv in zip(grp["params"], f["views"]):
                    if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                        p.grad = gv

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0) -> None:
        self.step_count += 1
        for grp, f in zip(self.param_groups, self._flat):
            src, dst, runs = [], [], []
            for p, gv, (lo, hi), ptr in zip(grp["params"], f["views"], f["spans"], f["ptrs"]):
                if p.data_ptr() != ptr:
                    raise RuntimeError("FlatAdamW: a parameter's storage moved after construction (module.to() / .half()?): create a new FlatAdamW")
                if p.grad is None:
                    continue                                                    # torch skips a tensor without a gradient: no decay, moments untouched
                if p.grad.data_ptr() != gv.data_ptr():
                    src.append(p.grad); dst.append(gv)
                if runs and runs[-1][1] == lo:
                    runs[-1][1] = hi
                else:
                    runs.append([lo, hi])
            if dst:
                torch._foreach_copy_(dst, src)                                  # all gradients into the flat buffer: one multi-tensor launch
            for lo, hi in runs:                                                 # ONE launch when every tensor has a gradient (the usual step)
                L.check(self.lib.mode_adamw_step(f["flat"][lo:hi].data_ptr(), f["grad"][lo:hi].data_ptr(), f["exp_avg"][lo:hi].data_ptr(),
                                                 f["exp_avg_sq"][lo:hi].data_ptr(), hi - lo, float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]),
                                                 float(grp["eps"]), float(grp["weight_decay"]), self.step_count, float(grad_scale), None, None, 0.0, _stream()),
                        "adamw_step")
            torch.autograd.graph.increment_version(grp["params"])                 # raw-pointer write: version-gated caches (conv weight shadows) must see it

    def state_dict(self) -> Dict:
        return {"step": self.step_count, "exp_avg": [f["exp_avg"] for f in self._flat], "exp_avg_sq": [f["exp_avg_sq"] for f in self._flat],
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd: Dict) -> None:
        self.step_count = int(sd["step"])
        for f, m_, v_ in zip(self._flat, sd["exp_avg"], sd["exp_avg_sq"]):
            f["exp_avg"].copy_(m_); f["exp_avg_sq"].copy_(v_)
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            g.update(s_)


class ArenaEMA:
    """Exponential moving average of the denoiser weights over the flat arena (replaces the ``EMA`` Lightning callback,
    mode/callbacks/ema.py:36-141, for the denoiser's parameters): ``e -= (1 - decay_t) * (e - w)`` with the callback's warm-up schedule
    ``decay_t = clamp(1 - (1 + max(0, t - start_step - 1) / inv_gamma) ** -power, min_value, max_value)`` (its non-apex path — the apex
    multi-tensor path with a constant decay does not exist on ROCm).  Fused into ``FusedAdamW.step(ema=...)`` (+8 B/element on the
    optimizer stream) or applied on its own with ``update(step)`` after a foreign optimizer.  ``swap()`` exchanges live and averaged
    weights in place (the callback's ``swap_model_weights`` around validation) — both directions keep the bf16 shadow coherent."""

    def __init__(self, model, decay: float = 0.999, apply_ema_every_n_steps: int = 1, start_step: int = 0, inv_gamma: float = 1.0,
                 power: float = 2 / 3, min_value: float = 0.0, max_value: float = 0.9999):
        if not 0.0 <= decay <= 1.0:
            raise ValueError("EMA decay value must be between 0 and 1")
        self.model = model
        self.decay, self.every, self.start_step = decay, apply_ema_every_n_steps, start_step
        self.inv_gamma, self.power, self.min_value, self.max_value = inv_gamma, power, min_value, max_value
        self.flat = None
        self._cur_step = None
        self._sharded = False                                    # True between a ZeRO-1 step(ema=self) and FusedAdamW.gather_state(): own shards only

    def _need_complete(self, what: str) -> None:
        if self._sharded:
            raise RuntimeError(f"ArenaEMA.{what}: the average is current on each rank's own shards only after ZeRO-1 steps - call "
                               "optimizer.gather_state(reducer) on every rank first")

    def get_decay(self, optimization_step: int) -> float:
        step = max(0, optimization_step - self.start_step - 1)
        value = 1 - (1 + step / self.inv_gamma) ** -self.power
        return max(min(value, self.max_value), self.min_value)

    def should_apply(self, step: int) -> bool:
        return step != self._cur_step and step >= self.start_step and step % self.every == 0

    def mark_applied(self, step: int) -> None:
        self._cur_step = step

    def ensure(self, arena) -> None:
        if self.flat is None or self.flat.numel() != arena.flat.numel() or self.flat.device != arena.flat.device:
            self.flat = arena.flat.detach().clone()                       # starts as a copy of the weights (on_train_start)

    @torch.no_grad()
    def update(self, step: int, optimizer=None) -> None:
        """Stand-alone EMA pass (when the optimizer is not FusedAdamW, or after ZeRO-1 steps).  Pass the ``FusedAdamW`` as ``optimizer`` to have
        sharded (bf16-rounded) masters refused instead of averaged."""
        if not self.should_apply(step):
            return
        self._need_complete("update()")
        if optimizer is not None and (getattr(optimizer, "_master_sharded", False)):
            raise RuntimeError("ArenaEMA.update(): the fp32 masters of the other ranks' shards are bf16-rounded after zero1='bf16' steps - call "
                               "optimizer.gather_state(reducer) first")
        eng = self.model.engine
        self.ensure(eng.arena)
        n = eng.arena.bounds["total"]
        L.check(eng.lib.mode_ema_update(self.flat.data_ptr(), eng.arena.flat.data_ptr(), n, 1.0 - self.get_decay(step), _stream()), "ema_update")
        self.mark_applied(step)

    @torch.no_grad()
    def swap(self) -> None:
        """Exchange the live weights with the averaged ones (call again to swap back)."""
        self._need_complete("swap()")
        eng = self.model.engine
        self.ensure(eng.arena)
        tmp = eng.arena.flat.clone()
        eng.arena.flat.copy_(self.flat)
        self.flat.copy_(tmp)
        eng.weights_updated(lp_synced=False)


class TriStageLR:
    """Warm-up / hold / cosine-decay learning-rate schedule of the reference (mode/utils/lr_schedulers/tri_stage_scheduler.py:69-147, as
    configured by conf/model/mode_agent.yaml lr_scheduler): linear from ``init_lr_scale * lr`` to ``lr`` over ``phase_ratio[0] * total_steps``,
    hold for ``phase_ratio[1]``, cosine to ``final_lr_scale * lr`` over ``phase_ratio[2]``, then constant.  Works on any optimizer exposing
    ``param_groups`` (``FusedAdamW`` or torch's)."""

    def __init__(self, optimizer, lr: float, init_lr_scale: float = 0.01, final_lr_scale: float = 0.01, phase_ratio=(0.1, 0.4, 0.5),
                 total_steps: int = 400000):
        if isinstance(phase_ratio, str):
            phase_ratio = tuple(float(x) for x in phase_ratio.strip("()[] ").split(","))
        self.optimizer = optimizer
        self.warmup_steps, self.hold_steps, self.decay_steps = (int(total_steps * r) for r in phase_ratio)
        self.peak_lr, self.init_lr, self.final_lr = lr, init_lr_scale * lr, final_lr_scale * lr
        self.update_step = 0
        self.lr = self.init_lr

    def lr_at(self, t: int) -> float:
        import math
        if t < self.warmup_steps:
            return self.init_lr + (self.peak_lr - self.init_lr) / self.warmup_steps * t
        t -= self.warmup_steps
        if t < self.hold_steps:
            return self.peak_lr
        t -= self.hold_steps
        if t <= self.decay_steps:
            return self.final_lr + 0.5 * (self.peak_lr - self.final_lr) * (1 + math.cos(t / self.decay_steps * math.pi))
        return self.final_lr

    def step(self, val_loss=None) -> float:
        self.lr = self.lr_at(self.update_step)
        for g in self.optimizer.param_groups:
            g["lr"] = self.lr
        self.update_step += 1
        return self.lr

    def get_lr(self) -> float:
        return self.lr
