"""Inference-side harness around the HIP sampler: what ``MoDEAgent`` does between "perceptual
embeddings + goal embedding" and "action for this control step" — noise-schedule / sampler dispatch, the initial noise draw, action
chunking with replanning every ``multistep`` steps, routing pre-cache per noise level, and loading the denoiser's weights from a
published checkpoint.  The reference agent (mode/models/mode_agent.py) is a LightningModule that also owns the ResNet / CLIP encoders;
those producers are out of scope here, so this harness starts at their outputs and — unlike the reference's ``step`` which is B = 1
(``pred_action_seq[0, ...]``, mode_agent.py:630) — serves a BATCH of environments per call (BASELINE configs[4]: 32 envs).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import gc_sampling as gs


def get_noise_schedule(n_sampling_steps: int, noise_schedule_type: str, sigma_min: float, sigma_max: float, device="cpu") -> torch.Tensor:
    """``MoDEAgent.get_noise_schedule`` (mode_agent.py:841-861): same names, same defaults (Karras rho = 7), same error."""
    if noise_schedule_type == "karras":
        return gs.get_sigmas_karras(n_sampling_steps, sigma_min, sigma_max, 7, device)
    if noise_schedule_type == "exponential":
        return gs.get_sigmas_exponential(n_sampling_steps, sigma_min, sigma_max, device)
    if noise_schedule_type == "vp":
        return gs.get_sigmas_vp(n_sampling_steps, device=device)
    if noise_schedule_type == "linear":
        return gs.get_sigmas_linear(n_sampling_steps, sigma_min, sigma_max, device=device)
    if noise_schedule_type == "cosine_beta":
        return gs.cosine_beta_schedule(n_sampling_steps, device=device)
    if noise_schedule_type == "ve":
        return gs.get_sigmas_ve(n_sampling_steps, sigma_min, sigma_max, device=device)
    if noise_schedule_type == "iddpm":
        return gs.get_iddpm_sigmas(n_sampling_steps, sigma_min, sigma_max, device=device)
    raise ValueError("Unknown noise schedule type")


def sample_loop(model, sigmas, x_t, state, goal, sampler_type: str = "ddim", extra_args: Optional[dict] = None, scaler=None):
    """``MoDEAgent.sample_loop`` (mode_agent.py:779-839): sampler names -> functions, ``s_churn`` / ``s_min`` / ``use_scaler`` taken from
    ``extra_args`` the way the agent does, ``ValueError`` for an unknown name."""
    extra_args = extra_args or {}
    s_churn = extra_args.get("s_churn", 0)
    s_min = extra_args.get("s_min", 0)
    sc = scaler if extra_args.get("use_scaler", False) else None
    reduced = {k: extra_args[k] for k in ("s_churn", "keep_last_actions")} if extra_args else {}
    table = {
        "lms": lambda: gs.sample_lms(model, state, x_t, goal, sigmas, scaler=sc, disable=True, extra_args=reduced),
        "heun": lambda: gs.sample_heun(model, state, x_t, goal, sigmas, scaler=sc, s_churn=s_churn, s_tmin=s_min, disable=True),
        "euler": lambda: gs.sample_euler(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
        "ancestral": lambda: gs.sample_dpm_2_ancestral(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
        "euler_ancestral": lambda: gs.sample_euler_ancestral(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
        "dpm": lambda: gs.sample_dpm_2(model, state, x_t, goal, sigmas, disable=True),
        "dpm_adaptive": lambda: gs.sample_dpm_adaptive(model, state, x_t, goal, sigmas[-2].item(), sigmas[0].item(), disable=True),
        "dpm_fast": lambda: gs.sample_dpm_fast(model, state, x_t, goal, sigmas[-2].item(), sigmas[0].item(), len(sigmas), disable=True),
        "dpmpp_2s_ancestral": lambda: gs.sample_dpmpp_2s_ancestral(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
        "dpmpp_2m": lambda: gs.sample_dpmpp_2m(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
        "dpmpp_2m_sde": lambda: gs.sample_dpmpp_sde(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
        "ddim": lambda: gs.sample_ddim(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
        "dpmpp_2s": lambda: gs.sample_dpmpp_2s(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
        "debugging": lambda: gs.sample_dpmpp_2_with_lms(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
        "dpmpp_2_with_lms": lambda: gs.sample_dpmpp_2_with_lms(model, state, x_t, goal, sigmas, scaler=sc, disable=True),
    }
    if sampler_type not in table:
        raise ValueError("desired sampler type not found!")
    return table[sampler_type]()


# Samplers whose step loop depends on the schedule only (no data-dependent control flow, no host-side noise source): whole call capturable in a hipGraph.
# Not here: "ddim" and "euler" (without churn the same update: both take MoDeDiT's fused graph, samplers.sample_euler) and "dpmpp_2m" (the same chain with the
# two-point extrapolation inside the head kernel, samplers.sample_dpmpp_2m -> GCDenoiser.dpmpp_2m_fused), "heun" / "dpm" / "dpmpp_2s" (two-stage solvers: every
# stage's linear update inside the head kernel, GCDenoiser.two_stage_fused), "lms" / "dpmpp_2_with_lms"
# (host-side quadrature of the schedule), "dpmpp_2m_sde" (torchsde Brownian tree on the host), "dpm_adaptive" / "dpm_fast" (step sizes from error
# norms / host floats).
_GRAPHABLE_SAMPLERS = ("euler_ancestral", "ancestral", "dpmpp_2s_ancestral")


class ChunkedRolloutPolicy:
    """Action-chunking policy for a batch of environments: plan ``act_window_size`` actions with the sampler, emit one per control step,
    replan every ``multistep`` steps (``MoDEAgent.forward`` / ``step`` / ``denoise_actions`` / ``precompute_expert_for_inference``,
    mode_agent.py:584-644, 733-760).

    ``step(perceptual_emb, latent_goal)``: ``perceptual_emb = {'state_images': (B, 2, obs_dim)}`` (the encoders' output), ``latent_goal``
    (B, G) or (B, 1, G); returns the actions of this control step, (B, action_dim).  With the default DDIM sampler a replanning call is one
    hipGraph replay; the routing decisions of every noise level are resolved once (``precompute_experts_for_inference``) like the agent does
    on its first inference call.

    With ``static_resnet`` / ``gripper_resnet`` (the agent's perceptual encoders, mode_agent.py:132-160) ``step`` also takes the environment's
    observation as the agent does - ``{'rgb_obs': {'rgb_static': (B, T, 3, H, W), 'rgb_gripper': ...}}`` - and embeds it on replanning steps
    (``MoDEAgent.forward``, mode_agent.py:598-603) through ``GraphedVisualEncoder``: one more hipGraph replay instead of ~640 eager launches."""

    def __init__(self, denoiser, num_sampling_steps: int = 10, sigma_min: float = 0.001, sigma_max: float = 80.0,
                 noise_scheduler: str = "exponential", sampler_type: str = "ddim", act_window_size: int = 10, multistep: int = 10,
                 action_dim: int = 7, generator: Optional[torch.Generator] = None, static_resnet=None, gripper_resnet=None,
                 encoder_autocast: Optional[torch.dtype] = torch.bfloat16):
        if multistep > act_window_size:
            raise ValueError("multistep cannot exceed the planned window")
        if (static_resnet is None) != (gripper_resnet is None):
            raise ValueError("give both perceptual encoders or neither")
        self.encoders = None
        if static_resnet is not None:
            from .perceptual_encoders import GraphedVisualEncoder
            self.encoders = GraphedVisualEncoder(static_resnet.eval(), gripper_resnet.eval(), encoder_autocast)
        self.model = denoiser
        self.num_sampling_steps, self.sigma_min, self.sigma_max = num_sampling_steps, sigma_min, sigma_max
        self.noise_scheduler, self.sampler_type = noise_scheduler, sampler_type
        self.act_window_size, self.multistep, self.action_dim = act_window_size, multistep, action_dim
        self.generator = generator
        self.need_precompute_experts_for_inference = True
        self.reset()

    def reset(self) -> None:
        """Start of an episode (the reference agent's ``reset``): replan at the next ``step``."""
        self.rollout_step_counter = 0
        self.pred_action_seq: Optional[torch.Tensor] = None

    def _schedule(self, dev) -> torch.Tensor:
        """The noise schedule of this policy, built once per device: handing the sampler the SAME tensor every call lets it recognise the
        schedule (pointer + version) without comparing values."""
        cached = getattr(self, "_sigmas", None)
        if cached is None or cached.device != torch.device(dev):
            cached = self._sigmas = get_noise_schedule(self.num_sampling_steps, self.noise_scheduler, self.sigma_min, self.sigma_max, dev)
        return cached

    def precompute_expert_for_inference(self, goal=None) -> None:
        inner = self.model.inner_model
        dev = next(inner.parameters()).device
        for sigma in self._schedule(dev)[:-1]:
            inner.precompute_experts_for_inference(sigma, goal)

    @torch.no_grad()
    def denoise_actions(self, perceptual_emb: Dict[str, torch.Tensor], latent_goal: torch.Tensor, extra_args: Optional[dict] = None) -> torch.Tensor:
        self.model.eval()
        dev = perceptual_emb["state_images"].device
        if latent_goal.dim() < perceptual_emb["state_images"].dim():
            latent_goal = latent_goal.unsqueeze(1)
        if self.need_precompute_experts_for_inference:
            self.precompute_expert_for_inference(latent_goal[:1] if self.model.inner_model.use_goal_in_routing else None)
            self.need_precompute_experts_for_inference = False
        sigmas = self._schedule(dev)
        x = torch.randn((len(latent_goal), self.act_window_size, self.action_dim), device=dev, generator=self.generator) * self.sigma_max
        graphable = _GRAPHABLE_SAMPLERS + (("heun", "dpm", "dpmpp_2s") if os.environ.get("MODE_TWO_STAGE_FUSED", "1") == "0" else ())
        if self.sampler_type in graphable and not extra_args:
            out = self._sample_graphed(sigmas, x, perceptual_emb, latent_goal)
            if out is not None:
                return out
        return sample_loop(self.model, sigmas, x, perceptual_emb, latent_goal, self.sampler_type, extra_args)

    def _sample_graphed(self, sigmas, x, perceptual_emb, latent_goal):
        """The whole sampler call (every denoiser call, every update of the recurrence, the samplers' own noise draws) as ONE hipGraph replay - what the
        fused DDIM path does for ``ddim`` and ``euler``, for the samplers whose control flow does not depend on the data (heun, dpm-solver(++) ...: the step
        loop only branches on which levels of the SCHEDULE are zero).  Captured once per (sampler, batch, weights storage); the ancestral samplers' noise
        comes from torch's default generator, which hipGraph capture advances per replay.  None = not applicable (goal / token routing, training mode,
        MODE_HIP_GRAPH=0): the caller takes the step-by-step path."""
        import os
        from . import samplers as S
        from .engine import capture_graph
        from .modedit import MoDeDiT
        den = self.model
        inner = getattr(den, "inner_model", None)
        if (not isinstance(inner, MoDeDiT) or inner.training or inner.use_goal_in_routing or not inner.cond_router or len(x) == 0
                or os.environ.get("MODE_HIP_GRAPH", "1") == "0"):
            return None
        eng = inner.engine
        dev, B = eng.device, x.shape[0]
        img, gl = inner._prep_obs(eng, perceptual_emb, latent_goal)
        inner._check_batch(B, img, gl, x)
        cache = self.__dict__.setdefault("_chunk_graphs", {})
        key = (self.sampler_type, B, eng.compute_dtype, eng._structs_for, str(dev), id(sigmas), sigmas._version, float(den.sigma_data))
        ent = cache.get(key)
        if ent is None:
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            ent = dict(x=x.clone(), img=img.clone(), goals=gl.clone(), sig=sigmas,
                       img_e=torch.empty(B * inner.n_img_tokens, inner.embed_dim, device=dev), goal_e=torch.empty(B, inner.embed_dim, device=dev))
            ent["ws"] = torch.empty(max(eng.workspace_bytes(B, 0), eng.workspace_bytes(0, 1)), dtype=torch.uint8, device=dev)
            state = {"state_images": ent["img"].view(B, inner.n_img_tokens, -1)}
            goal3 = ent["goals"].view(B, 1, -1)

            def chunk():
                eng.embed_obs(ent["img"], ent["goals"], out=(ent["img_e"], ent["goal_e"]))
                cc = dict(inner=inner, sigma_data=float(den.sigma_data), obs_emb=(ent["img_e"], ent["goal_e"]), metas=[])
                S._set_chunk_capture(cc)
                try:
                    out = sample_loop(den, sigmas, ent["x"], state, goal3, self.sampler_type, None)
                    return out, cc["metas"]
                finally:
                    S._set_chunk_capture(None)
            with eng.pinned_workspace(ent["ws"]):
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):                                # warm-up outside the capture (code objects, the schedule's host-side reads)
                    chunk()
                torch.cuda.current_stream(dev).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with capture_graph(graph):
                    ent["out"], ent["metas"] = chunk()
            ent["graph"] = graph
            cache[key] = ent
        ent["x"].copy_(x); ent["img"].copy_(img); ent["goals"].copy_(gl)
        ent["graph"].replay()
        ml = eng.meta_layout(B * inner.seq_len)
        for meta in ent["metas"]:                                             # expert-usage counters, as the step-by-step path keeps them
            inner._account_usage(meta, ml, B * inner.seq_len)
        return ent["out"].clone()

    @torch.no_grad()
    def embed(self, obs: Dict, latent_goal: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Raw camera observation -> ``perceptual_emb`` (``MoDEAgent.embed_visual_obs``, mode_agent.py:548-567); embedded observations pass through."""
        if "state_images" in obs:
            return obs
        if self.encoders is None:
            raise ValueError("raw observations ('rgb_obs') need the policy's perceptual encoders: pass static_resnet / gripper_resnet")
        rgb = obs["rgb_obs"]
        goal = latent_goal.reshape(latent_goal.shape[0], -1)
        emb = self.encoders(rgb["rgb_static"], rgb["rgb_gripper"], goal)
        return {"state_images": emb["state_images"].to(torch.float32)}

    @torch.no_grad()
    def step(self, perceptual_emb: Dict, latent_goal: torch.Tensor) -> torch.Tensor:
        if self.rollout_step_counter % self.multistep == 0:
            self.pred_action_seq = self.denoise_actions(self.embed(perceptual_emb, latent_goal), latent_goal)
        current = self.pred_action_seq[:, self.rollout_step_counter]
        self.rollout_step_counter += 1
        if self.rollout_step_counter == self.multistep:
            self.rollout_step_counter = 0
        return current


def _read_checkpoint_file(path: str, trust_pickle: bool = False) -> Dict[str, torch.Tensor]:
    """The file forms `MoDEAgent.load_pretrained_parameters` accepts (mode_agent.py:141-158): a checkpoint DIRECTORY holding
    ``model_cleaned.safetensors`` (preferred) or ``model_cleaned.pt``; a ``.safetensors`` file; a Lightning ``.ckpt`` / ``torch.save``d dict whose
    weights sit under ``'state_dict'`` (a bare state_dict is accepted too).  Lightning checkpoints carry hyper-parameter objects, which torch >= 2.6
    refuses under its ``weights_only=True`` default: the safe load is tried first and the full unpickler only after it fails - the reference
    (torch 2.2) always unpickles; load only checkpoints you trust."""
    if os.path.isdir(path):
        st, pt = os.path.join(path, "model_cleaned.safetensors"), os.path.join(path, "model_cleaned.pt")
        if os.path.exists(st):
            path = st
        elif os.path.exists(pt):
            path = pt
        else:
            raise FileNotFoundError(f"No cleaned weights found in {path}")     # the reference's message (mode_agent.py:155)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    try:
        blob = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as safe_err:                                               # pickle.UnpicklingError and friends: non-tensor objects in the file
        # Full unpickling executes whatever the file says: only on the caller's explicit say-so (Lightning checkpoints carry omegaconf objects and need it;
        # the reference's own loader, mode_agent.py:135-160, reads the cleaned tensor files above, which never get here)
        if not (trust_pickle or os.environ.get("MODE_TRUST_CKPT", "0") == "1"):
            raise RuntimeError(f"cannot read checkpoint {path} with weights_only=True ({safe_err!r}); pass trust_pickle=True / set MODE_TRUST_CKPT=1 to allow "
                               "full unpickling of a file you trust") from safe_err
        import warnings
        warnings.warn(f"loading {path} with the full unpickler (trusted by the caller)")
        try:
            blob = torch.load(path, map_location="cpu", weights_only=False)
        except Exception as e:
            raise RuntimeError(f"cannot read checkpoint {path}: weights_only load failed with {safe_err!r}, full unpickling with {e!r}") from e
    return blob.get("state_dict", blob) if isinstance(blob, dict) else blob


def load_denoiser_checkpoint(model, source, prefix: str = "model.inner_model.", strict: bool = False, trust_pickle: bool = False):
    """Load the denoiser's tensors from an agent checkpoint: a ``.safetensors`` file (the published HF weights), a ``torch.save``d
    ``state_dict`` / Lightning checkpoint, or an in-memory mapping.  Keys are matched by name after stripping the agent's prefix
    (``model.inner_model.`` — mode_agent.py:209-251 loads by key and skips the CLIP / ResNet tensors, which belong to the out-of-scope
    encoders); the kernel-side layout is untouched because the Parameters are arena views.  Returns (missing, unexpected, skipped_shape)."""
    if isinstance(source, (str, bytes, os.PathLike)):
        sd = _read_checkpoint_file(os.fsdecode(source), trust_pickle)
    else:
        sd = dict(source)
    own = model.state_dict()
    picked, skipped = {}, []
    for key, t in sd.items():
        if "visual" in key or "clip" in key.lower():
            continue
        name = key[len(prefix):] if key.startswith(prefix) else key
        if name not in own:
            continue
        if tuple(own[name].shape) != tuple(t.shape):
            if own[name].numel() == t.numel():
                t = t.reshape(own[name].shape)
            else:
                skipped.append(name)
                continue
        picked[name] = t
    res = model.load_state_dict(picked, strict=strict)
    return list(res.missing_keys), list(res.unexpected_keys), skipped


# Key prefixes of older published checkpoints -> the agent's current attribute names (mode_agent.py:216-226).  Tried in this order, first match wins,
# and only for keys the agent does not have under their own name.
_AGENT_KEY_REMAP = (
    ("img_encoder_image_wrist.", "gripper_resnet."),
    ("img_encoder_image_secondary.", "static_resnet."),
    ("img_encoder_image_primary.", "static_resnet."),
    ("net.", "gripper_resnet.resnet."),
)


def _fit_checkpoint_tensor(key: str, t: torch.Tensor, shape) -> Optional[torch.Tensor]:
    """The reference loader's shape rule (mode_agent.py:163-200): equal shapes pass; a 0-d entry becomes zeros; 1-d -> 1-d of another length is tiled
    (BatchNorm vectors); 1-d or 2-d -> 4-d with the same element count is viewed as the convolution weight; ``running_*`` buffers of another rank
    with the same element count are viewed; anything else is incompatible (None)."""
    shape = tuple(shape)
    if tuple(t.shape) == shape:
        return t
    want = 1
    for d in shape:
        want *= d
    if t.dim() == 0:
        return torch.zeros(shape, device=t.device)
    if t.dim() == 1 and len(shape) == 1:
        return t.repeat(shape[0] // t.shape[0]) if t.shape[0] != shape[0] else t
    if t.dim() == 1 and len(shape) == 4:
        return t.view(shape)                                                   # raises on an element-count mismatch, like the reference
    if t.dim() == 2 and len(shape) == 4 and t.numel() == want:
        return t.view(shape)
    if "running_" in key and t.dim() != len(shape) and t.numel() == want:
        return t.view(shape)
    return None


def _agent_parts(target) -> Dict[str, torch.nn.Module]:
    """{'model': GCDenoiser, 'static_resnet': ..., 'gripper_resnet': ...} from a mapping, a ChunkedRolloutPolicy, or any object with those attributes
    (the attribute names ARE the key prefixes of the agent's state_dict, mode_agent.py:79, 90-91)."""
    if isinstance(target, dict):
        parts = dict(target)
    else:
        parts = {"model": getattr(target, "model", None)}
        enc = getattr(target, "encoders", None)
        for name in ("static_resnet", "gripper_resnet"):
            parts[name] = getattr(target, name, None) if getattr(target, name, None) is not None else getattr(enc, name, None)
    parts = {k: v for k, v in parts.items() if v is not None}
    if not parts:
        raise ValueError("nothing to load into: expected 'model' and / or 'static_resnet' / 'gripper_resnet'")
    return parts


def load_agent_checkpoint(target, source, strict: bool = False, verbose: bool = False, trust_pickle: bool = False) -> Dict[str, object]:
    """Load ONE agent checkpoint - the published HF ``.safetensors``, a Lightning ``.ckpt`` / ``torch.save``d ``state_dict`` or a mapping - into the
    denoiser AND both perceptual encoders, as ``MoDEAgent.load_pretrained_parameters`` does (mode_agent.py:135-251): CLIP tensors are skipped
    (``'visual'`` / ``'clip'`` in the key), keys the agent does not know are retried under the prefix table of older releases
    (``img_encoder_image_wrist.`` -> ``gripper_resnet.``, ``img_encoder_image_primary.`` / ``secondary.`` -> ``static_resnet.``, ``net.`` ->
    ``gripper_resnet.resnet.``), tensors are fitted by the reference's reshape rule, incompatible ones are skipped and reported, and the result is
    loaded with ``load_state_dict(strict=strict)``.

    ``target``: ``{'model': GCDenoiser, 'static_resnet': m, 'gripper_resnet': m}`` (any subset), a ``ChunkedRolloutPolicy`` built with encoders, or an
    object with those attributes.  Returns ``{'direct', 'reshaped', 'skipped', 'missing', 'unexpected'}`` (counts for the first two, key lists for the
    rest; keys carry the agent-level prefix).  Parameters are written in place (arena views, the graphs' static pointers stay valid).

    Files are read with ``weights_only=True``; a file that needs the full unpickler (a Lightning ``.ckpt`` with omegaconf / argparse objects next to the weights)
    is only read on the caller's say-so - ``trust_pickle=True`` or ``MODE_TRUST_CKPT=1`` - because unpickling executes what the file says."""
    if isinstance(source, (str, bytes, os.PathLike)):
        sd = _read_checkpoint_file(os.fsdecode(source), trust_pickle)
    else:
        sd = dict(source)
    parts = _agent_parts(target)
    current = {f"{name}.{k}": v for name, mod in parts.items() for k, v in mod.state_dict().items()}
    picked: Dict[str, Dict[str, torch.Tensor]] = {name: {} for name in parts}
    direct, reshaped, skipped = 0, 0, []
    for key, t in sd.items():
        if "visual" in key or "clip" in key.lower():
            continue
        tkey = key
        if key not in current:
            for old, new in _AGENT_KEY_REMAP:
                if key.startswith(old):
                    tkey = key.replace(old, new)
                    break
        if tkey not in current:
            continue                                                           # the reference drops unknown keys before load_state_dict too
        fitted = _fit_checkpoint_tensor(tkey, t, current[tkey].shape)
        if fitted is None:
            skipped.append(tkey)
            if verbose:
                print(f"Skipping incompatible tensor {tkey}: checkpoint {tuple(t.shape)} vs {tuple(current[tkey].shape)}")
            continue
        if tuple(fitted.shape) != tuple(t.shape):
            reshaped += 1
        else:
            direct += 1
        name, sub = tkey.split(".", 1)
        picked[name][sub] = fitted
    missing, unexpected = [], []
    from .perceptual_encoders import invalidate_conv_shadows
    for name, mod in parts.items():
        res = mod.load_state_dict(picked[name], strict=strict)
        invalidate_conv_shadows(mod)                                             # cached compute-dtype conv weights are re-cast on next use, whatever the loader's write path
        missing += [f"{name}.{k}" for k in res.missing_keys]
        unexpected += [f"{name}.{k}" for k in res.unexpected_keys]
    if verbose:
        print(f"Direct copies: {direct}  reshaped: {reshaped}  skipped: {len(skipped)}  missing: {len(missing)}")
    return {"direct": direct, "reshaped": reshaped, "skipped": skipped, "missing": missing, "unexpected": unexpected}
