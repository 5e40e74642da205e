"""Fused AdamW over the flat parameter arena (replaces ``torch.optim.AdamW`` as configured by MoDEAgent.configure_optimizers,
mode/models/mode_agent.py:365-392: two groups — decayed weights / undecayed biases — betas and lr from the Hydra config).

One HIP kernel per group walks ``(p, g, m, v)`` of the arena region (16 B read + 12 B written per element, HBM-bound: ~20 GB per step
for the 685 M-parameter denoiser) and also emits the bf16 compute shadow, so the next forward needs no separate cast pass.
``gripper_embed.weight`` (never receives a gradient in the reference, hence never updated by its optimizer) lives in the arena's
``dead`` region and is not touched.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch

from . import _lib as L
from .engine import _stream


class FusedAdamW:
    def __init__(self, model, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05):
        frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
        if frozen:
            raise NotImplementedError(f"FusedAdamW updates the whole arena; frozen parameters are not supported: {frozen[:3]}")
        self.model = model
        eng = model.engine                                   # adopts the parameters into the arena
        self.eng, self.arena = eng, eng.arena
        n = self.arena.bounds["no_decay"]                    # decay + no_decay regions (dead region excluded)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=eng.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=eng.device)
        self.step_count = 0
        self._side = None
        self.param_groups = [dict(name="decay", lr=lr, betas=betas, eps=eps, weight_decay=weight_decay),
                             dict(name="no_decay", lr=lr, betas=betas, eps=eps, weight_decay=0.0)]

    def zero_grad(self, set_to_none: bool = False) -> None:
        """Gradients are overwritten by every backward; nothing to clear (kept for optimizer-API compatibility)."""

    def _launch(self, lo: int, hi: int, gp, lp, grad_scale: float) -> None:
        ar = self.arena
        L.check(self.eng.lib.mode_adamw_step(ar.flat[lo:hi].data_ptr(), ar.grad[lo:hi].data_ptr(), self.exp_avg[lo:hi].data_ptr(),
                                             self.exp_avg_sq[lo:hi].data_ptr(), hi - lo, float(gp["lr"]), float(gp["betas"][0]),
                                             float(gp["betas"][1]), float(gp["eps"]), float(gp["weight_decay"]), self.step_count,
                                             float(grad_scale), None if lp is None else lp[lo:hi].data_ptr(), _stream()), "adamw_step")

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
    def step(self, grad_scale: float = 1.0, overlap: bool = False, reducer=None) -> None:
        """One AdamW update of the whole arena.

        Default: (exchange gradients through ``reducer`` — its collectives overlap the backward —, then) two launches over the decay / no-decay
        regions on the current stream.

        ``overlap=True`` (experimental): call right after ``loss.backward()`` returned.  The backward kernels are still executing; the update
        of block l's 228 MB weight slice is queued on a side stream behind the event the backward chain records when block l's gradients are
        complete, so the HBM-bound optimizer pass (25 % of a serial step) runs underneath the backward of the earlier blocks (block l's
        backward only reads block l's weights, so updating later blocks early is safe).  Measured on MI355X: 16.1-16.7 ms per step at best
        (AdamW capped to 128-256 workgroups, ``mode_set_option("adamw_blocks", n)``) against 17.0-18.0 ms serial, but run-to-run spread up to
        21 ms — the streaming pass raises the memory latency the fill-bound GEMMs are sensitive to — hence not the default.  ``reducer`` (an ``ArenaGradReducer``)
        chains the data-parallel exchange in front of each slice's update on the same events; its 1/world scale is applied here."""
        eng = self.model.engine
        ar = eng.arena
        if ar is not self.arena:
            raise RuntimeError("the parameter arena was rebuilt (model.to()/half()?): create a new FusedAdamW")
        if ar.grad is None:
            raise RuntimeError("no gradients: run a training forward + backward first")
        self.step_count += 1
        lp = ar.lp if eng.compute_dtype == "bf16" else None
        gd, gn = self.param_groups
        train = getattr(eng, "_train", None)
        events = train.events if (train is not None and train.events is not None) else None
        if reducer is not None and reducer.world > 1:
            grad_scale = grad_scale * (1.0 if reducer.average else 1.0 / reducer.world)
        if not overlap or events is None:
            if reducer is not None:
                reducer.reduce()
            self._launch(0, ar.bounds["decay"], gd, lp, grad_scale)
            self._launch(ar.bounds["decay"], ar.bounds["no_decay"], gn, lp, grad_scale)
        else:
            if self._side is None:
                self._side = torch.cuda.Stream(device=eng.device, priority=int(os.environ.get("MODE_OPT_PRIO", "0")))
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
        eng.weights_updated(lp_synced=lp is not None)

    # ---- checkpointing (same information as torch's optimizer state, flat)
    def state_dict(self) -> Dict:
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd: Dict) -> None:
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
