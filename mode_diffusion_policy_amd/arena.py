"""Flat parameter / gradient arenas for the MoDE denoiser (HBM layout of the training + inference hot path).

All ~685 M fp32 parameters of ``MoDeDiT`` live in ONE contiguous HBM buffer; the module's ``nn.Parameter`` objects are *views* into
it (their names, shapes and ``state_dict`` layout stay exactly the reference's — mode/models/networks/modedit.py:598-704).  What this
buys on MI355X (288 GB HBM, ~8 TB/s):

* the q/k/v projections and the E experts of a block are adjacent, so the kernels read them as packed ``[3D, D]`` / ``[E, 8D, D]``
  operands without any per-step ``cat`` / ``stack`` copies;
* the router MLP weights of all L blocks are adjacent (``r_w0 [L, 2D, D]``), so routing every layer is one grouped GEMM;
* the low-precision compute shadow is one cast of the arena (or is written by the fused optimizer as a by-product);
* the gradient arena has the same layout: the backward chain writes straight into ``p.grad`` views, the optimizer is one
  elementwise pass over ``(p, g, m, v)`` and the data-parallel exchange is a handful of large collectives over flat slices.

Regions (each tensor 256-byte aligned): ``[decay | no_decay | dead]`` following the AdamW grouping of
MoDEAgent.get_optim_groups (mode/models/mode_agent.py:365-384); ``dead`` holds ``gripper_embed.weight`` which never receives a gradient
in the reference (modedit.py:684) and therefore is never touched by its optimizer either.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

_ALIGN = 64                                   # elements (256 B of fp32, 128 B of bf16)


def _layer_keys(D: int, E: int, hd: int):
    decay = [("wqkv", (3 * D, D)), ("wo", (D, D)), ("w1", (E, 8 * D, D)), ("w2", (E, D, 4 * D))]
    no_decay = [("bqkv", (3 * D,)), ("b1", (E, 8 * D))]
    return decay, no_decay


def arena_layout(m) -> Tuple[List[Tuple[str, tuple, int]], Dict[str, int]]:
    """[(key, shape, offset)], region bounds {"decay": n0, "no_decay": n1, "total": n2} (element counts, aligned)."""
    D, E, L, A = m.embed_dim, m.num_experts, m.num_layers, m.action_dim
    hd = D // m.n_heads
    ld, lnd = _layer_keys(D, E, hd)
    # small per-layer tensors are stacked over layers ([L, ...]): the router of every layer is routed / back-propagated as one batch and the
    # norm-gain gradients of all layers are reduced by one segmented column sum each
    decay = [("r_w0", (L, 2 * D, D)), ("r_w3", (L, E, 2 * D)), ("ln1_g", (L, D)), ("ln2_g", (L, D)), ("qn_g", (L, hd)), ("kn_g", (L, hd))]
    # per-layer blocks in BACKWARD order (last layer first) so gradient slices complete front-to-back during the backward pass
    for i in reversed(range(L)):
        decay += [(f"l{i}.{k}", s) for k, s in ld]
    decay += [("pos", (m.pos_emb.shape[1], D)), ("w_se", (D,)), ("w_sl", (D, D)), ("w_tok", (D, m.obs_dim)),
              ("w_goal", (D, m.goal_dim)), ("w_act", (D, A)), ("ln_g", (D,)), ("w_out", (A, D))]
    no_decay = [("r_b0", (L, 2 * D)), ("r_b3", (L, E))]
    for i in reversed(range(L)):
        no_decay += [(f"l{i}.{k}", s) for k, s in lnd]
    no_decay += [("b_se", (D,)), ("b_out", (A,))]
    dead = [("gripper", tuple(m.gripper_embed.weight.shape))]
    out, off, bounds = [], 0, {}
    for name, region in (("decay", decay), ("no_decay", no_decay), ("dead", dead)):
        for key, shp in region:
            out.append((key, shp, off))
            n = int(torch.Size(shp).numel())
            off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        bounds[name] = off
    bounds["total"] = off
    return out, bounds


def param_views(m, g: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Map arena keys onto the reference's parameter names (state_dict layout).  Works for weights and gradients."""
    D = m.embed_dim
    v: Dict[str, torch.Tensor] = {
        "pos_emb": g["pos"].unsqueeze(0), "sigma_emb.weight": g["w_se"].unsqueeze(1), "sigma_emb.bias": g["b_se"],
        "sigma_linear.weight": g["w_sl"], "tok_emb.weight": g["w_tok"], "gripper_embed.weight": g["gripper"], "goal_emb.weight": g["w_goal"],
        "action_emb.weight": g["w_act"], "ln.g": g["ln_g"], "out.weight": g["w_out"], "out.bias": g["b_out"]}
    for i in range(m.num_layers):
        k, p = f"l{i}.", f"blocks.{i}."
        v[p + "ln_1.g"], v[p + "ln_2.g"] = g["ln1_g"][i], g["ln2_g"][i]
        v[p + "attn.q_norm.g"], v[p + "attn.k_norm.g"] = g["qn_g"][i], g["kn_g"][i]
        for j, nm in enumerate(("query", "key", "value")):                       # packed rows = [query; key; value]
            v[p + f"attn.{nm}.weight"] = g[k + "wqkv"][j * D:(j + 1) * D]
            v[p + f"attn.{nm}.bias"] = g[k + "bqkv"][j * D:(j + 1) * D]
        v[p + "attn.c_proj.weight"] = g[k + "wo"]
        v[p + "router.router.mlp.0.weight"], v[p + "router.router.mlp.0.bias"] = g["r_w0"][i], g["r_b0"][i]
        v[p + "router.router.mlp.3.weight"], v[p + "router.router.mlp.3.bias"] = g["r_w3"][i], g["r_b3"][i]
        for e in range(m.num_experts):
            q = p + f"experts.expert_{e}.mlp."
            v[q + "0.project.weight"], v[q + "0.project.bias"], v[q + "2.weight"] = g[k + "w1"][e], g[k + "b1"][e], g[k + "w2"][e]
    return v


class ParamArena:
    def __init__(self, model, device: torch.device):
        self.layout, self.bounds = arena_layout(model)
        self.device = device
        n = self.bounds["total"]
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.w: Dict[str, torch.Tensor] = self._views(self.flat)
        self.by_name = param_views(model, self.w)
        self.names = [nm for nm, _ in model.named_parameters()]
        missing = set(self.names) - set(self.by_name)
        if missing:
            raise RuntimeError(f"parameters without an arena slot: {sorted(missing)[:4]}")
        with torch.no_grad():
            for nm, p in model.named_parameters():
                dst = self.by_name[nm]
                dst.copy_(p.detach().to(device=device, dtype=torch.float32).reshape(dst.shape))
                p.data = dst.view(p.shape)                                      # the parameter IS the arena slice from now on
        self._ptrs = {nm: self.by_name[nm].data_ptr() for nm in self.names}
        self.lp = None                                                          # bf16 compute shadow (same offsets)
        self.wl: Dict[str, torch.Tensor] = {}
        self.grad = None                                                        # fp32 gradient arena (same offsets)
        self.g: Dict[str, torch.Tensor] = {}
        self.g_by_name: Dict[str, torch.Tensor] = {}
        self.lp_synced = False
        self.version = 0                                                        # bumped whenever the weights change

    def _views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {key: flat[off: off + int(torch.Size(shp).numel())].view(shp) for key, shp, off in self.layout}

    def owns(self, model, named=None) -> bool:
        """True while every parameter still aliases its arena slice (``.to()`` / ``.half()`` / re-assignment breaks that).  ``named``: a cached
        [(name, parameter)] list (``DitEngine.named_params``) instead of a walk over the module tree."""
        ptrs = self._ptrs
        for nm, p in (named if named is not None else model.named_parameters()):
            if p.data_ptr() != ptrs.get(nm) or p.dtype != torch.float32:
                return False
        return True

    def ensure_lp(self) -> None:
        if self.lp is None:
            self.lp = torch.empty(self.bounds["total"], dtype=torch.bfloat16, device=self.device)
            self.wl = self._views(self.lp)
            self.lp_synced = False
        if not self.lp_synced:
            self.lp.copy_(self.flat)                                            # one cast pass over the arena
            self.lp_synced = True

    def ensure_grad(self, model) -> None:
        if self.grad is None:
            self.grad = torch.zeros(self.bounds["total"], dtype=torch.float32, device=self.device)
            self.g = self._views(self.grad)
            self.g_by_name = param_views(model, self.g)

    def offset(self, key: str) -> int:
        for k, _, off in self.layout:
            if k == key:
                return off
        raise KeyError(key)
