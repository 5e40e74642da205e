"""CPU oracle for the MoDE denoising hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, pure-PyTorch fp32 *functional* restatement of the
reference algorithm (it consumes a plain ``state_dict`` — no nn.Module graph).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product path (``mode_diffusion_policy_amd``) never does.

Parity pin: the reference ships NO golden vectors/tests for this path
(SURVEY.md §4), so this oracle is pinned against outputs of the reference
itself, generated in the build container by ``oracle/gen_golden.py`` (which
imports /root/reference) and committed as ``tests/golden/*.npz``.

Every function cites the reference file:line it restates (paths relative to
the reference checkout).  The restatement deliberately uses the *kernel-side*
formulation (packed QKV, routing on distinct rows only, sorted dispatch
permutation, fused DDIM/EDM update) so it doubles as the spec of the HIP path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class DiTConfig:
    """Dimension/flag bundle; names follow the Hydra keys of
    conf/model/mode_agent.yaml:46-76."""
    obs_dim: int = 2048
    goal_dim: int = 512
    action_dim: int = 7
    embed_dim: int = 1024
    n_layers: int = 12
    n_heads: int = 8
    goal_seq_len: int = 1
    obs_seq_len: int = 1
    action_seq_len: int = 10
    num_experts: int = 4
    top_k: int = 2
    router_normalize: bool = True
    use_goal_in_routing: bool = False
    use_noise_token_as_input: bool = True
    cond_router: bool = True
    n_img_tokens: int = 2          # 'state_images' carries 2 camera tokens

    @property
    def seq_len(self) -> int:
        return (1 if self.use_noise_token_as_input else 0) + self.goal_seq_len + self.n_img_tokens + self.action_seq_len

    @staticmethod
    def from_state_dict(sd: Dict[str, Tensor], n_heads: int, top_k: int, **kw) -> "DiTConfig":
        D = sd["sigma_linear.weight"].shape[0]
        L = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        E = sd["blocks.0.router.router.mlp.3.weight"].shape[0]
        return DiTConfig(obs_dim=sd["tok_emb.weight"].shape[1], goal_dim=sd["goal_emb.weight"].shape[1],
                         action_dim=sd["action_emb.weight"].shape[1], embed_dim=D, n_layers=L, n_heads=n_heads,
                         action_seq_len=sd["pos_emb"].shape[1] - 1, num_experts=E, top_k=top_k, **kw)


# --------------------------------------------------------------------------- schedules / EDM
def get_sigmas_exponential(n: int, sigma_min: float, sigma_max: float) -> Tensor:
    """exp(linspace(ln smax, ln smin, n)) ++ [0]   (gc_sampling.py:35-38, append_zero :22)."""
    s = torch.linspace(math.log(sigma_max), math.log(sigma_min), n, dtype=torch.float32).exp()
    return torch.cat([s, s.new_zeros(1)])


def edm_scalings(sigma: Tensor, sigma_data: float) -> Tuple[Tensor, Tensor, Tensor]:
    """c_skip, c_out, c_in of Karras et al.   (score_wrappers.py:31-43)."""
    s2 = sigma ** 2 + sigma_data ** 2
    return sigma_data ** 2 / s2, sigma * sigma_data / s2 ** 0.5, 1 / s2 ** 0.5


def rand_log_logistic(shape, loc, scale, min_value, max_value, generator=None) -> Tensor:
    """Truncated log-logistic sigma draw, fp64 internally  (edm_diffusion/utils.py:159-166)."""
    lo = torch.as_tensor(min_value, dtype=torch.float64).log().sub(loc).div(scale).sigmoid()
    hi = torch.as_tensor(max_value, dtype=torch.float64).log().sub(loc).div(scale).sigmoid()
    u = torch.rand(shape, dtype=torch.float64, generator=generator) * (hi - lo) + lo
    return u.logit().mul(scale).add(loc).exp().to(torch.float32)


# --------------------------------------------------------------------------- primitives
def rmsnorm(x: Tensor, g: Tensor, eps: float = 1e-6) -> Tensor:
    """x / clamp(||x||_2 * dim^-1/2, eps) * g   (modedit.py:72-80)."""
    n = torch.linalg.vector_norm(x, dim=-1, keepdim=True) * (x.shape[-1] ** -0.5)
    return x / n.clamp(min=eps) * g


def sigma_embedding(sd, sigma: Tensor) -> Tensor:
    """ln(sigma)/4 -> Linear(1,D)+b -> Linear(D,D)   (modedit.py:823-832).  Returns (R, D)."""
    s = (sigma.reshape(-1, 1).log() / 4)
    e = s * sd["sigma_emb.weight"].t() + sd["sigma_emb.bias"]          # (R,1)*(1,D)
    return e @ sd["sigma_linear.weight"].t()


def router_probs(sd, layer: int, cond: Tensor) -> Tuple[Tensor, Tensor]:
    """Router on DISTINCT conditioning rows.  cond (R,D) -> (shifted_logits, probs) each (R,E).
    Linear(D,2D)+b -> GELU(erf) -> Linear(2D,E)+b ; logits -= rowmax ; softmax ; clamp(1e-9, 1-1e-9)
    (modedit.py:194-202, 336, 345-349)."""
    p = f"blocks.{layer}.router.router.mlp."
    h = F.gelu(cond @ sd[p + "0.weight"].t() + sd[p + "0.bias"])
    logits = h @ sd[p + "3.weight"].t() + sd[p + "3.bias"]
    logits = logits - logits.max(dim=-1, keepdim=True).values
    probs = torch.softmax(logits, dim=-1).clamp(min=1e-9, max=1 - 1e-9)
    return logits, probs


def topk_route(probs: Tensor, k: int, normalize: bool) -> Tuple[Tensor, Tensor]:
    """Eval / use_argmax selection: top-k by prob (descending; ties -> lower expert id) and combine
    weights (probs at chosen experts, renormalised to sum 1 when router_normalize).
    (modedit.py:392, 398-399, 418-419).  Returns idx (R,k) int64, w (R,k) aligned with idx."""
    # stable descending sort == "ties -> lower index first" (torch.topk tie order is unspecified;
    # exact ties are outside the parity contract, SURVEY §8 a-bis)
    order = torch.sort(probs, dim=-1, descending=True, stable=True).indices[:, :k]
    w = probs.gather(1, order)
    if normalize:
        w = w / w.sum(dim=-1, keepdim=True)
    return order, w


def dispatch_permutation(idx: Tensor, num_experts: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Canonical MoE dispatch implied by the reference's boolean-mask loop (modedit.py:561-566):
    experts ascending, within an expert token ids ascending.
    idx (N,k) -> counts (E,), perm (N*k,) token id per sorted row, slot (N*k,) which top-k slot."""
    N, k = idx.shape
    tok = torch.arange(N).repeat_interleave(k)
    slot = torch.arange(k).repeat(N)
    key = idx.reshape(-1) * N + tok                       # sort by (expert, token)
    order = torch.argsort(key, stable=True)
    counts = torch.bincount(idx.reshape(-1), minlength=num_experts)
    return counts, tok[order], slot[order]


# --------------------------------------------------------------------------- counter-based dropout streams of the HIP training chain
# The reference draws its dropout masks from torch's RNG (SDPA dropout_p modedit.py:149, nn.Dropout in the expert MLP modedit.py:254); those
# draws cannot be bit-matched by another implementation.  The HIP chain uses counter-based hash masks instead (a pure function of
# (step seed, stream, element index), so the backward regenerates them).  Restated here bit for bit so that the STOCHASTIC training path can be
# checked against this oracle's autograd with shared randomness (mode_common.h: mode_stream_seed; attn.hip: attn_keep; train_ops.hip: drop_keep).
def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x7feb352d)
        x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x846ca68b)
        x = x ^ (x >> np.uint32(16))
    return x


def stream_seed(seed: int, stream: int) -> int:
    """mode_stream_seed: stream 2l = attention dropout of layer l, 2l+1 = expert dropout of layer l."""
    x = np.array([(seed ^ ((0x9e3779b9 * (stream + 1)) & 0xffffffff)) & 0xffffffff], dtype=np.uint32)
    return int(_lowbias32(x)[0])


def _thresh(p: float) -> int:
    return 0 if p <= 0.0 else int(float(p) * 4294967296.0)


def attn_keep_scale(seed: int, B: int, H: int, T: int, p: float) -> Tensor:
    """(B,H,T,T) multiplier of the softmax probabilities: 1/(1-p) where kept, 0 where dropped (attn.hip attn_keep)."""
    e = np.arange(B * H * T * T, dtype=np.uint64).astype(np.uint32)             # ((b*H+h)*T + q)*T + k
    with np.errstate(over="ignore"):
        hsh = _lowbias32(_lowbias32(e ^ np.uint32(seed)) + np.uint32(0x9e3779b9))
    keep = hsh >= np.uint32(_thresh(p))
    return torch.from_numpy(keep.astype(np.float32) / np.float32(1.0 - p)).view(B, H, T, T)


def mlp_keep_scale(seed: int, row0: int, rows: int, hdim: int, p: float) -> Tensor:
    """(rows, hdim) multiplier of the SwishGLU output of SORTED rows row0.. (train_ops.hip drop_keep, element index r*hdim + c)."""
    idx = (np.arange(row0, row0 + rows, dtype=np.uint64)[:, None] * np.uint64(hdim) + np.arange(hdim, dtype=np.uint64)[None, :])
    lo, hi = (idx & np.uint64(0xffffffff)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        hsh = _lowbias32(_lowbias32(lo ^ np.uint32(seed)) + hi * np.uint32(0x9e3779b9))
    keep = hsh >= np.uint32(_thresh(p))
    return torch.from_numpy(keep.astype(np.float32) / np.float32(1.0 - p))


def expert_mlp(sd, layer: int, e: int, x: Tensor, keep_scale: Optional[Tensor] = None) -> Tensor:
    """SwishGLU(D,4D) -> Dropout -> Linear(4D,D, no bias)   (modedit.py:83-90, 247-255); dropout = identity in eval, or the given
    keep/scale multiplier (mlp_keep_scale)."""
    p = f"blocks.{layer}.experts.expert_{e}.mlp."
    h = x @ sd[p + "0.project.weight"].t() + sd[p + "0.project.bias"]
    proj, gate = h.tensor_split(2, dim=-1)
    hh = proj * F.silu(gate)
    if keep_scale is not None:
        hh = hh * keep_scale
    return hh @ sd[p + "2.weight"].t()


def causal_attention(sd, layer: int, h: Tensor, n_heads: int, keep_scale: Optional[Tensor] = None) -> Tensor:
    """q,k,v Linear(+bias) -> per-head qk-RMSNorm(eps 1e-6) -> causal softmax(QK^T/sqrt(hd)) V -> c_proj (no bias)
    (modedit.py:108-111, 125-127, 141-166).  h (B,T,D) -> (B,T,D)."""
    p = f"blocks.{layer}.attn."
    B, T, D = h.shape
    hd = D // n_heads
    wqkv = torch.cat([sd[p + "query.weight"], sd[p + "key.weight"], sd[p + "value.weight"]], 0)
    bqkv = torch.cat([sd[p + "query.bias"], sd[p + "key.bias"], sd[p + "value.bias"]], 0)
    qkv = h @ wqkv.t() + bqkv                                           # packed QKV, one GEMM
    q, k, v = (t.view(B, T, n_heads, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    q = rmsnorm(q, sd[p + "q_norm.g"])
    k = rmsnorm(k, sd[p + "k_norm.g"])
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
    mask = torch.ones(T, T, dtype=torch.bool).tril()
    att = att.masked_fill(~mask, float("-inf")).softmax(dim=-1)
    if keep_scale is not None:                                           # SDPA dropout_p on the attention weights (modedit.py:149)
        att = att * keep_scale
    y = (att @ v).transpose(1, 2).reshape(B, T, D)
    return y @ sd[p + "c_proj.weight"].t()


# --------------------------------------------------------------------------- denoiser network
@dataclass
class Aux:
    """Side outputs used by parity tests (router integers must be bit-exact)."""
    topk_idx: List[Tensor] = field(default_factory=list)       # per layer (B,T,k) int64
    combine_w: List[Tensor] = field(default_factory=list)      # per layer (B,T,k)
    probs: List[Tensor] = field(default_factory=list)          # per layer (B,E) on distinct rows
    shifted_logits: List[Tensor] = field(default_factory=list)  # per layer (B,E)
    perm: List[Tensor] = field(default_factory=list)           # per layer (N*k,)
    counts: List[Tensor] = field(default_factory=list)         # per layer (E,)
    block_out: List[Tensor] = field(default_factory=list)      # per layer (B,T,D)
    cond: Optional[Tensor] = None


def embed_sequence(sd, cfg: DiTConfig, state_images: Tensor, actions: Tensor, goals: Tensor, emb_t: Tensor) -> Tensor:
    """[sigma tok | goal+pos0 | img+pos1 (both) | act+pos1..A]   (modedit.py:760-790, 847-860)."""
    B = actions.shape[0]
    pos = sd["pos_emb"][0]
    goals = goals.reshape(B, 1, -1)
    goal_x = goals @ sd["goal_emb.weight"].t() + pos[: cfg.goal_seq_len]
    img_x = state_images @ sd["tok_emb.weight"].t() + pos[cfg.goal_seq_len: cfg.goal_seq_len + 1]
    act_x = actions @ sd["action_emb.weight"].t() + pos[cfg.goal_seq_len:]
    seq = ([emb_t.reshape(B, 1, -1)] if cfg.use_noise_token_as_input else []) + [goal_x, img_x, act_x]
    return torch.cat(seq, dim=1)


def dit_forward(sd: Dict[str, Tensor], cfg: DiTConfig, state_images: Tensor, actions: Tensor, goals: Tensor,
                sigma: Tensor, topk_idx: Optional[List[Tensor]] = None, return_aux: bool = False, dropout: Optional[dict] = None):
    """MoDeDiT.forward in eval mode (modedit.py:741-821, 530-595).

    ``topk_idx``: optional per-layer (B,T,k) expert ids (training: the reference draws them with
    torch.multinomial per token row, modedit.py:390 — the draw stays on the host side of the ABI).
    ``dropout``: optional {"seed": step seed, "attn_p": p, "mlp_p": p} — the HIP chain's counter-based masks (see attn_keep_scale).
    """
    B = actions.shape[0]
    T, D, E, k = cfg.seq_len, cfg.embed_dim, cfg.num_experts, cfg.top_k
    sigma = sigma.reshape(-1).expand(B) if sigma.numel() == 1 else sigma.reshape(B)
    emb_t = sigma_embedding(sd, sigma)                                   # (B,D)
    x = embed_sequence(sd, cfg, state_images, actions, goals, emb_t)     # (B,T,D)
    cond = emb_t
    if cfg.use_goal_in_routing:                                          # modedit.py:801-802
        cond = cond + goals.reshape(B, -1) @ sd["goal_emb.weight"].t()
    c = cond.reshape(B, 1, D)
    aux = Aux(cond=cond)
    for l in range(cfg.n_layers):
        p = f"blocks.{l}."
        ak = None
        if dropout is not None and dropout.get("attn_p", 0.0) > 0.0:
            ak = attn_keep_scale(stream_seed(int(dropout["seed"]), 2 * l), B, cfg.n_heads, T, float(dropout["attn_p"]))
        x = x + causal_attention(sd, l, rmsnorm(x, sd[p + "ln_1.g"]) + c, cfg.n_heads, ak)   # :532
        u = rmsnorm(x, sd[p + "ln_2.g"])                                                # :539 (overwrites stream)
        if not cfg.cond_router:
            # token routing (modedit.py:296-301, 322-325, 553): router(x, None) on the ln_2-normalised token states, one decision per token
            logits, probs = router_probs(sd, l, u.reshape(B * T, D))
            if topk_idx is None:
                idx, w = topk_route(probs, k, cfg.router_normalize)
            else:                                                            # training: ids drawn per token by the caller (modedit.py:390)
                idx = topk_idx[l].reshape(B * T, k)
                pr = probs.gather(1, idx)
                w = pr / pr.sum(-1, keepdim=True) if cfg.router_normalize else pr
        elif topk_idx is None:
            logits, probs = router_probs(sd, l, cond)                                   # distinct rows only
            idx_b, w_b = topk_route(probs, k, cfg.router_normalize)
            idx = idx_b[:, None, :].expand(B, T, k).reshape(B * T, k)
            w = w_b[:, None, :].expand(B, T, k).reshape(B * T, k)
        else:
            logits, probs = router_probs(sd, l, cond)
            idx = topk_idx[l].reshape(B * T, k)
            pr = probs[:, None, :].expand(B, T, E).reshape(B * T, E).gather(1, idx)
            w = pr / pr.sum(-1, keepdim=True) if cfg.router_normalize else pr
        counts, perm, slot = dispatch_permutation(idx, E)
        uf = u.reshape(B * T, D)
        nxt = torch.zeros_like(uf)
        off = 0
        for e in range(E):                                               # ascending expert id (:561)
            n_e = int(counts[e])
            if n_e:
                rows = perm[off: off + n_e]
                we = w[rows, slot[off: off + n_e]].unsqueeze(-1)
                mk = None
                if dropout is not None and dropout.get("mlp_p", 0.0) > 0.0:
                    mk = mlp_keep_scale(stream_seed(int(dropout["seed"]), 2 * l + 1), off, n_e, 4 * D, float(dropout["mlp_p"]))
                nxt[rows] += we * expert_mlp(sd, l, e, uf[rows], mk)
            off += n_e
        x = (uf + nxt).reshape(B, T, D)                                  # residual from the NORMALISED stream (:595)
        if return_aux:
            aux.topk_idx.append(idx.reshape(B, T, k)); aux.combine_w.append(w.reshape(B, T, k))
            aux.probs.append(probs); aux.shifted_logits.append(logits)
            aux.perm.append(perm); aux.counts.append(counts); aux.block_out.append(x)
    x = rmsnorm(x, sd["ln.g"])                                           # :818
    out = x[:, -cfg.action_seq_len:, :] @ sd["out.weight"].t() + sd["out.bias"]   # :807-808
    return (out, aux) if return_aux else out


# --------------------------------------------------------------------------- EDM wrapper, loss, sampler
def denoiser_forward(sd, cfg, sigma_data, state_images, action, goal, sigma, **kw):
    """F(x*c_in)*c_out + x*c_skip   (score_wrappers.py:65-80)."""
    c_skip, c_out, c_in = (t.reshape(-1, 1, 1) for t in edm_scalings(sigma.reshape(-1), sigma_data))
    r = dit_forward(sd, cfg, state_images, action * c_in, goal, sigma, **kw)
    if isinstance(r, tuple):
        return r[0] * c_out + action * c_skip, r[1]
    return r * c_out + action * c_skip


def denoiser_loss(sd, cfg, sigma_data, state_images, action, goal, noise, sigma, **kw):
    """Score-matching loss (score_wrappers.py:45-63): returns (loss, model_output)."""
    c_skip, c_out, c_in = (t.reshape(-1, 1, 1) for t in edm_scalings(sigma.reshape(-1), sigma_data))
    noised = action + noise * sigma.reshape(-1, 1, 1)
    out = dit_forward(sd, cfg, state_images, noised * c_in, goal, sigma, **kw)
    target = (action - c_skip * noised) / c_out
    return (out - target).pow(2).flatten(1).mean(), out


def ddim_update(x: Tensor, denoised: Tensor, sigma: float, sigma_next: float) -> Tensor:
    """x <- r*x + (1-r)*denoised, r = sigma_next/sigma.  Algebraically identical to
    (sigma_fn(t_next)/sigma_fn(t))*x - expm1(-h)*denoised  (gc_sampling.py:948-950)."""
    r = sigma_next / sigma
    return r * x + (1.0 - r) * denoised


def sample_ddim(sd, cfg, sigma_data, state_images, x: Tensor, goal: Tensor, sigmas: Tensor, trace: bool = False):
    """10-step DDIM / DPM-Solver-1 loop (gc_sampling.py:922-951)."""
    xs = []
    for i in range(len(sigmas) - 1):
        s = sigmas[i] * x.new_ones(x.shape[0])
        den = denoiser_forward(sd, cfg, sigma_data, state_images, x, goal, s)
        x = ddim_update(x, den, float(sigmas[i]), float(sigmas[i + 1]))
        if trace:
            xs.append(x.clone())
    return (x, xs) if trace else x


# --------------------------------------------------------------------------- aux losses / optimizer grouping
def load_balancing_term(probs_b: Tensor, idx: Tensor, w: Tensor, T: int, E: int) -> Tensor:
    """Per-block E * sum_e mean_{b,t}(router_probs[...,e]) * (sum_{b,t} mask[...,e] / N)   (modedit.py:586-593)."""
    N = idx.shape[0]
    mask = torch.zeros(N, E).scatter_(1, idx, 1.0)
    rp = torch.zeros(N, E).scatter_(1, idx, w)
    return E * (rp.mean(0) * (mask.sum(0) / N)).sum()


def router_z_loss(shifted_logits_per_layer: List[Tensor], T: int, eps: float = 1e-6) -> Tensor:
    """mean over layers of mean_rows(log(sum exp(shifted logits) + eps)^2)   (modedit.py:930-969).
    Rows are repeated T times in the reference; the mean over repeated rows equals the mean over distinct rows."""
    z = [torch.log(torch.exp(lg).sum(-1) + eps).pow(2).mean() for lg in shifted_logits_per_layer]
    return sum(z) / len(z)


def training_total_loss(sd, cfg: DiTConfig, sigma_data: float, state_images, action, goal, noise, sigma, entropy_gamma: float = 0.0,
                        router_z_delta: float = 0.0, **kw):
    """One modality of ``MoDEAgent.training_step`` (mode_agent.py:399-419): ``act_loss + entropy_gamma * load_balancing_loss() +
    router_z_delta * compute_router_z_loss()`` with every term attached to the autograd graph.  Returns (total, act, lb, z)."""
    c_skip, c_out, c_in = (t.reshape(-1, 1, 1) for t in edm_scalings(sigma.reshape(-1), sigma_data))
    noised = action + noise * sigma.reshape(-1, 1, 1)
    out, aux = dit_forward(sd, cfg, state_images, noised * c_in, goal, sigma, return_aux=True, **kw)
    target = (action - c_skip * noised) / c_out
    act = (out - target).pow(2).flatten(1).mean()
    T, E, k = cfg.seq_len, cfg.num_experts, cfg.top_k
    lb = sum(load_balancing_term(aux.probs[l], aux.topk_idx[l].reshape(-1, k), aux.combine_w[l].reshape(-1, k), T, E)
             for l in range(cfg.n_layers)) / cfg.n_layers
    z = router_z_loss(aux.shifted_logits, T)
    return act + entropy_gamma * lb + router_z_delta * z, act, lb, z


def uses_weight_decay(param_name: str) -> bool:
    """AdamW grouping rule (mode_agent.py:365-384): decay unless the NAME contains one of these."""
    return all(s not in param_name for s in ("bias", "LayerNorm", "embedding"))
