"""Fused AdamW over the flat parameter arena (replaces ``torch.optim.AdamW`` as configured by MoDEAgent.configure_optimizers,
mode/models/mode_agent.py:365-392: two groups — decayed weights / undecayed biases — betas and lr from the Hydra config).

One HIP kernel per group walks ``(p, g, m, v)`` of the arena region (16 B read + 12 B written per element, HBM-bound: ~20 GB per step
for the 685 M-parameter denoiser) and also emits the bf16 compute shadow, so the next forward needs no separate cast pass.
``gripper_embed.weight`` (never receives a gradient in the reference, hence never updated by its optimizer) lives in the arena's
``dead`` region and is not touched.
"""
from __future__ import annotations

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
        self.param_groups = [dict(name="decay", lr=lr, betas=betas, eps=eps, weight_decay=weight_decay),
                             dict(name="no_decay", lr=lr, betas=betas, eps=eps, weight_decay=0.0)]

    def zero_grad(self, set_to_none: bool = False) -> None:
        """Gradients are overwritten by every backward; nothing to clear (kept for optimizer-API compatibility)."""

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0) -> None:
        eng = self.model.engine
        ar = eng.arena
        if ar is not self.arena:
            raise RuntimeError("the parameter arena was rebuilt (model.to()/half()?): create a new FusedAdamW")
        if ar.grad is None:
            raise RuntimeError("no gradients: run a training forward + backward first")
        self.step_count += 1
        lp = ar.lp if eng.compute_dtype == "bf16" else None
        bounds = ((0, ar.bounds["decay"]), (ar.bounds["decay"], ar.bounds["no_decay"]))
        for (lo, hi), gp in zip(bounds, self.param_groups):
            L.check(eng.lib.mode_adamw_step(ar.flat[lo:hi].data_ptr(), ar.grad[lo:hi].data_ptr(), self.exp_avg[lo:hi].data_ptr(),
                                            self.exp_avg_sq[lo:hi].data_ptr(), hi - lo, float(gp["lr"]), float(gp["betas"][0]),
                                            float(gp["betas"][1]), float(gp["eps"]), float(gp["weight_decay"]), self.step_count,
                                            float(grad_scale), None if lp is None else lp[lo:hi].data_ptr(), _stream()), "adamw_step")
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
