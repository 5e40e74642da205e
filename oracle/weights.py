"""Seeded weight recipe shared by the golden-fixture generator and the tests.  TEST INFRASTRUCTURE ONLY.

Only seeds ship in ``tests/golden``: weights are regenerated from
``numpy.random.RandomState(seed).standard_normal`` in the reference's ``state_dict`` key order
(SURVEY.md §8b / §8c), scaled per tensor so that activations stay O(1), RMSNorm gains and
``pos_emb`` are non-trivial (they are ones/zeros at reference init) and router top-k margins are healthy.
``oracle/gen_golden.py`` asserts that :func:`param_spec` equals the imported reference's
``state_dict()`` keys and shapes.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from .mode_oracle import DiTConfig


def param_spec(cfg: DiTConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) in the reference's registration order (modedit.py:680-725, 447-506)."""
    D, E = cfg.embed_dim, cfg.num_experts
    hd = D // cfg.n_heads
    seq = cfg.goal_seq_len + cfg.obs_seq_len - 1 + cfg.action_seq_len
    spec: List[Tuple[str, Tuple[int, ...]]] = [
        ("pos_emb", (1, seq, D)),
        ("sigma_emb.weight", (D, 1)), ("sigma_emb.bias", (D,)),
        ("sigma_linear.weight", (D, D)),
        ("tok_emb.weight", (D, cfg.obs_dim)),
        ("gripper_embed.weight", (D, cfg.obs_dim)),
        ("goal_emb.weight", (D, cfg.goal_dim)),
        ("action_emb.weight", (D, cfg.action_dim)),
    ]
    for i in range(cfg.n_layers):
        p = f"blocks.{i}."
        spec += [(p + "ln_1.g", (D,))]
        for n in ("key", "query", "value"):
            spec += [(p + f"attn.{n}.weight", (D, D)), (p + f"attn.{n}.bias", (D,))]
        spec += [(p + "attn.c_proj.weight", (D, D)), (p + "attn.q_norm.g", (hd,)), (p + "attn.k_norm.g", (hd,)),
                 (p + "ln_2.g", (D,)),
                 (p + "router.router.mlp.0.weight", (2 * D, D)), (p + "router.router.mlp.0.bias", (2 * D,)),
                 (p + "router.router.mlp.3.weight", (E, 2 * D)), (p + "router.router.mlp.3.bias", (E,))]
        for e in range(E):
            q = p + f"experts.expert_{e}.mlp."
            spec += [(q + "0.project.weight", (8 * D, D)), (q + "0.project.bias", (8 * D,)), (q + "2.weight", (D, 4 * D))]
    spec += [("ln.g", (D,)), ("out.weight", (cfg.action_dim, D)), ("out.bias", (cfg.action_dim,))]
    return spec


def _scale(name: str, shape: Tuple[int, ...], router_gain: float) -> Tuple[float, float]:
    """(mean, std) of the seeded fill for one tensor."""
    if name.endswith(".g"):
        return 1.0, 0.1
    if name == "pos_emb":
        return 0.0, 0.1
    if name.endswith("bias"):
        return 0.0, 0.1
    if name == "sigma_emb.weight":
        return 0.0, 1.0
    fan_in = shape[-1]
    std = fan_in ** -0.5
    if "router.router.mlp.3" in name:
        std *= router_gain
    return 0.0, std


def make_state_dict(cfg: DiTConfig, seed: int, router_gain: float = 4.0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    rs = np.random.RandomState(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_spec(cfg):
        mean, std = _scale(name, shape, router_gain)
        a = rs.standard_normal(size=shape).astype(np.float32) * np.float32(std) + np.float32(mean)
        sd[name] = torch.from_numpy(a).to(dtype)
    return sd


def make_inputs(cfg: DiTConfig, B: int, seed: int):
    """Synthetic CALVIN-shaped inputs: N(0,1) image/goal tokens, N(0,1) actions."""
    rs = np.random.RandomState(seed)
    f = lambda *s: torch.from_numpy(rs.standard_normal(size=s).astype(np.float32))
    return dict(state_images=f(B, cfg.n_img_tokens, cfg.obs_dim), goals=f(B, 1, cfg.goal_dim),
                actions=f(B, cfg.action_seq_len, cfg.action_dim), noise=f(B, cfg.action_seq_len, cfg.action_dim),
                x0=f(B, cfg.action_seq_len, cfg.action_dim) * 80.0, u=torch.from_numpy(rs.uniform(size=(B,))))


# Named configurations used across fixtures/tests (C1, tiny; C2 is the benchmark model).
CONFIGS = {
    "tiny": dict(obs_dim=32, goal_dim=16, embed_dim=64, n_layers=2, n_heads=4, num_experts=4, top_k=2),
    "c1": dict(obs_dim=512, goal_dim=512, embed_dim=256, n_layers=2, n_heads=8, num_experts=2, top_k=1),
    "c1e4": dict(obs_dim=512, goal_dim=512, embed_dim=256, n_layers=2, n_heads=8, num_experts=4, top_k=2),
    "c2block": dict(obs_dim=64, goal_dim=512, embed_dim=1024, n_layers=1, n_heads=8, num_experts=4, top_k=2),
    "c2": dict(obs_dim=2048, goal_dim=512, embed_dim=1024, n_layers=12, n_heads=8, num_experts=4, top_k=2),
}


def get_config(name: str) -> DiTConfig:
    return DiTConfig(**CONFIGS[name])
