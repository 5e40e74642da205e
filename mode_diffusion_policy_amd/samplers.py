"""The other score-based samplers and noise schedules of ``mode.models.edm_diffusion.gc_sampling`` behind the same signatures
(``MoDEAgent.sample_loop`` dispatches on ``sampler_type``, mode_agent.py:798-839; ``get_noise_schedule`` on
``noise_scheduler``, :841-861).

Every sampler is a host-side recurrence over tiny (B, 10, 7) tensors around 1-2 denoiser calls per step; the denoiser call is the HIP
launch chain (``GCDenoiser.forward`` -> ``MoDeDiT.denoise``: EDM scalings fused, routing resolved per noise level), so these need no
kernels of their own.  They are written against two primitives:

* the Karras ODE derivative  d = (x - D(x; sigma)) / sigma                              (Karras et al. 2022, eq. 3 / Alg. 1-2)
* the exponential-integrator step of DPM-Solver++  x' = (s'/s) x - expm1(-h) D,  h = ln(s/s')       (Lu et al. 2022, eq. 8)

and keep the reference's observable behaviour: argument names and defaults, the callback payload keys of each sampler (``'x'`` vs
``'action'``), the ``scaler.clip_output`` hook, Euler fallback on the final ``sigma = 0`` step, and the noise-"churn" of Algorithm 2.
``sample_dpmpp_sde`` runs with any ``noise_sampler=`` callable; its default Brownian-tree sampler needs the optional ``torchsde``.
``sample_dpm_adaptive`` raises ``NotImplementedError``: the reference function itself fails on every call (see its docstring).
"""
from __future__ import annotations

import threading

import numpy as np
import torch
from scipy import integrate


# ------------------------------------------------------------------------------------------------------------------ schedules
def _with_zero(s: torch.Tensor) -> torch.Tensor:
    return torch.cat([s, s.new_zeros([1])])


def tag_schedule(sigmas: torch.Tensor, *key) -> torch.Tensor:
    """Attach the host-side identity of a schedule's VALUES (generator name + arguments) to the tensor object.  ``MoDEAgent.denoise_actions``
    builds a fresh schedule tensor for every chunk (mode_agent.py:752, 842-861); the fused sampler recognises an unchanged schedule by this tag
    instead of comparing device values (a host sync) or trusting a recyclable data pointer."""
    sigmas._mode_sched = (key, str(sigmas.device), str(sigmas.dtype), sigmas._version)     # the tag describes the values AT THIS VERSION only
    return sigmas


def get_sigmas_vp(n, beta_d=19.9, beta_min=0.1, eps_s=1e-3, device="cpu"):
    """Continuous VP schedule sigma(t) = sqrt(exp(beta_d t^2 / 2 + beta_min t) - 1), t from 1 to eps_s (gc_sampling.py:84-88)."""
    t = torch.linspace(1, eps_s, n, device=device)
    return tag_schedule(_with_zero(torch.sqrt(torch.exp(beta_d * t ** 2 / 2 + beta_min * t) - 1)), "vp", n, beta_d, beta_min, eps_s)


def get_sigmas_ve(n, sigma_min=0.02, sigma_max=100, device="cpu"):
    """Geometric variance schedule as the reference computes it (gc_sampling.py:61-68; note its index ramp runs to n+1, not n-1)."""
    t = torch.linspace(0, n + 1, n, device=device)
    var = (sigma_max ** 2) * ((sigma_min ** 2 / sigma_max ** 2) ** (t / (n - 1)))
    return tag_schedule(_with_zero(torch.sqrt(var)), "ve", n, sigma_min, sigma_max)


def cosine_beta_schedule(n, s=0.008, device="cpu"):
    """Nichol & Dhariwal cosine schedule, returned as the reversed clipped betas (gc_sampling.py:47-58)."""
    grid = np.linspace(0, n + 1, n + 1)
    acp = np.cos(((grid / (n + 1)) + s) / (1 + s) * np.pi * 0.5) ** 2
    acp = acp / acp[0]
    betas = np.clip(1 - acp[1:] / acp[:-1], 0, 0.999)
    return tag_schedule(_with_zero(torch.tensor(betas[::-1].copy(), device=device, dtype=torch.float32)), "cosine_beta", n, s)


def get_iddpm_sigmas(n, sigma_min=0.02, sigma_max=100, M=1000, j_0=0, C_1=0.001, C_2=0.008, device="cpu"):
    """iDDPM discretisation (Karras et al. 2022, Table 1; gc_sampling.py:71-81): u_{j-1} = sqrt((u_j^2 + 1) / max(ab(j-1)/ab(j), C_1) - 1),
    keep the levels inside [sigma_min, sigma_max], pick n of them at equal index spacing."""
    # alpha_bar is evaluated in fp32 (an int64 index tensor times a Python float promotes to the default dtype) while the recurrence
    # runs in fp64: the mixed precision is part of the reference's observable output, so it is reproduced here
    idx = torch.arange(0, M + 1, dtype=torch.int64)
    ab = (0.5 * np.pi * idx / M / (C_2 + 1)).sin() ** 2                       # fp32 [M+1]
    u = torch.zeros(M + 1, dtype=torch.float64)
    for j in range(M, j_0, -1):
        u[j - 1] = ((u[j] ** 2 + 1) / (ab[j - 1] / ab[j]).clip(min=C_1) - 1).sqrt()
    kept = u[torch.logical_and(u >= sigma_min, u <= sigma_max)].numpy()
    pick = np.round((len(kept) - 1) / (n - 1) * np.arange(n, dtype=np.float64)).astype(np.int64)
    return tag_schedule(_with_zero(torch.tensor(kept[pick], dtype=torch.float64, device=device)).to(torch.float32), "iddpm", n, sigma_min, sigma_max, M, j_0, C_1, C_2)


# ------------------------------------------------------------------------------------------------------------------ primitives
def to_d(action, sigma, denoised):
    """Karras ODE derivative (gc_sampling.py:91-93)."""
    sigma = torch.as_tensor(sigma, device=action.device, dtype=action.dtype)
    return (action - denoised) / sigma.reshape(sigma.shape + (1,) * (action.ndim - sigma.ndim))


def default_noise_sampler(x):
    return lambda sigma, sigma_next: torch.randn_like(x)


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """(sigma_down, sigma_up) of an ancestral step (gc_sampling.py:102-109)."""
    if not eta:
        return sigma_to, 0.0
    cand = eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5
    # (python's min() on device tensors is a host comparison = one sync per step; torch.minimum is the same number without it)
    sigma_up = torch.minimum(sigma_to, cand) if torch.is_tensor(sigma_to) and torch.is_tensor(cand) else min(sigma_to, cand)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def _ancestral_plan(sigmas, eta):
    """(sigma_down [n], sigma_up [n], [sigma_down_i == 0 on the HOST]) of all ancestral steps of a schedule: the step loops branch on "is sigma_down
    zero", which depends on the schedule alone - read once per schedule tensor (kept on the object like ``_zero_levels``) instead of once per step."""
    hit = getattr(sigmas, "_mode_anc", None)
    if hit is not None and hit[0] == (sigmas._version, float(eta)):
        return hit[1]
    down, up = get_ancestral_step(sigmas[:-1], sigmas[1:], eta=eta)
    if not torch.is_tensor(up):
        up = torch.zeros_like(down)
    plan = (down, up, (down == 0).tolist())
    try:
        sigmas._mode_anc = ((sigmas._version, float(eta)), plan)
    except Exception:                                                        # noqa: BLE001
        pass
    return plan


def _exp_step(x, denoised, s_from, s_to):
    """x(s_to) of the data-prediction exponential integrator with D frozen: (s_to/s_from) x - expm1(-h) D, h = ln(s_from/s_to)."""
    h = s_from.log() - s_to.log()
    return (s_to / s_from) * x - (-h).expm1() * denoised


def _churn(x, sigmas, i, s_churn, s_tmin, s_tmax, s_noise):
    """Algorithm 2's stochastic 'churn': raise the noise level to sigma_hat = sigma_i (1 + gamma) by adding fresh noise."""
    # (s_churn = 0 - the agent's default - makes gamma 0 whatever sigma is: no tensor-vs-float comparison, i.e. no host sync per step)
    gamma = 0.0 if s_churn <= 0 else (min(s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= sigmas[i] <= s_tmax else 0.0)
    eps = torch.randn_like(x) * s_noise
    sigma_hat = sigmas[i] * (gamma + 1)
    if gamma > 0:
        x = x + eps * (sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5
    return x, sigma_hat


def _zero_levels(sigmas):
    """[sigma_i == 0 for every level] on the HOST: the step loops branch on "is the next level zero" (Euler fallback on the final step) - asking the
    device tensor inside the loop is one host sync per step, which keeps the host from running ahead of the GPU.  Read once per schedule TENSOR (kept
    on the object with its in-place version): a policy that hands the sampler the same schedule every chunk pays no host sync at all, and the whole
    sampler call becomes capturable in a hipGraph (rollout.ChunkedRolloutPolicy)."""
    hit = getattr(sigmas, "_mode_zero", None)
    if hit is not None and hit[0] == sigmas._version:
        return hit[1]
    z = (sigmas == 0).tolist()
    try:
        sigmas._mode_zero = (sigmas._version, z)
    except Exception:                                                        # noqa: BLE001  (tensor subclasses without a __dict__)
        pass
    return z


# Set by rollout.ChunkedRolloutPolicy while it captures a whole sampler call in ONE hipGraph: {"inner": MoDeDiT, "sigma_data": float, "obs_emb":
# (img_e, goal_e), "metas": []}.  A hipGraph cannot be replayed inside a capture, so _Run.denoise then issues the eager launch chain of
# MoDeDiT.denoise (device-scalar sigma, observation embeddings computed once at the top of the captured chunk) instead of denoise_graphed's replay.
# Per THREAD (hipGraph stream capture is a per-thread state too: one capture per thread at a time; the policy sets and clears it around its own call),
# so two policies capturing on two threads do not see each other's hook.
_CAPTURE_TLS = threading.local()


def _set_chunk_capture(cc):
    _CAPTURE_TLS.cc = cc


def _chunk_capture():
    return getattr(_CAPTURE_TLS, "cc", None)


class _Run:
    """Shared per-call plumbing: the batched sigma vector, extra args, callback and output clipping."""

    def __init__(self, model, state, goal, action, extra_args, callback, scaler, x_key):
        self.model, self.state, self.goal = model, state, goal
        self.kw = {} if extra_args is None else extra_args
        self.ones = action.new_ones([action.shape[0]])
        self.callback, self.scaler, self.x_key = callback, scaler, x_key

    def denoise(self, x, sigma):
        cc = _chunk_capture()
        if cc is not None and not self.kw and torch.is_tensor(sigma) and sigma.numel() == 1:
            inner = cc["inner"]
            out = inner.denoise(None, x, None, sigma.reshape(1), cc["sigma_data"], _account=False, _obs_emb=cc["obs_emb"])
            cc["metas"].append(inner._last_meta)
            return out
        fast = getattr(self.model, "denoise_uniform", None)              # GCDenoiser over the HIP MoDeDiT: one hipGraph replay per call
        if fast is not None and not self.kw and torch.is_tensor(sigma) and sigma.numel() == 1:
            out = fast(self.state, x, self.goal, sigma)
            if out is not None:
                return out
        return self.model(self.state, x, self.goal, sigma * self.ones, **self.kw)

    def report(self, x, i, sigma, sigma_hat, denoised):
        if self.callback is not None:
            self.callback({self.x_key: x, "i": i, "sigma": sigma, "sigma_hat": sigma_hat, "denoised": denoised})

    def clip(self, x):
        return x if self.scaler is None else self.scaler.clip_output(x)


# ------------------------------------------------------------------------------------------------------------------ samplers
@torch.no_grad()
def sample_euler(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0,
                 s_tmax=float("inf"), s_noise=1.0):
    """Euler steps of Karras et al. (2022) Algorithm 2 (gc_sampling.py:165-211); an ODE solver for s_churn = 0.  Without churn, clipping, callback
    and extra arguments the step  x + (x - D)/s (s' - s)  is  (s'/s) x + (1 - s'/s) D  - the fused DDIM update -, so a GCDenoiser over the HIP
    MoDeDiT takes the whole call as one hipGraph replay (same result to fp32 rounding of the rearranged update)."""
    if s_churn <= 0 and scaler is None and callback is None and not extra_args and _chunk_capture() is None:
        fused = getattr(model, "first_order_ode_fused", None)
        out = fused(state, action, goal, sigmas) if fused is not None else None
        if out is not None:
            # The reference draws `eps = randn_like(action)` on EVERY step, churn or not (gc_sampling.py:196), and so does the step loop below: the
            # fused route makes the same draws and discards them, so the generator leaves this call in the state the reference leaves it in (the
            # next chunk's initial latent for a given seed is the reference's).  n tiny launches, ~3 us of host each.
            for _ in range(len(sigmas) - 1):
                torch.randn_like(action)
            return out
    run = _Run(model, state, goal, action, extra_args, callback, scaler, "x")
    for i in range(len(sigmas) - 1):
        action, sigma_hat = _churn(action, sigmas, i, s_churn, s_tmin, s_tmax, s_noise)
        denoised = run.denoise(action, sigma_hat)
        d = to_d(action, sigma_hat, denoised)
        run.report(action, i, sigmas[i], sigma_hat, denoised)
        action = run.clip(action + d * (sigmas[i + 1] - sigma_hat))
    return action


@torch.no_grad()
def sample_euler_ancestral(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, eta=1.0):
    """Euler step to sigma_down, then fresh noise of scale sigma_up (gc_sampling.py:214-254)."""
    run = _Run(model, state, goal, action, extra_args, callback, scaler, "x")
    downs, ups, down_zero = _ancestral_plan(sigmas, eta)
    for i in range(len(sigmas) - 1):
        denoised = run.denoise(action, sigmas[i])
        sigma_down, sigma_up = downs[i], ups[i]
        run.report(action, i, sigmas[i], sigmas[i], denoised)
        action = action + to_d(action, sigmas[i], denoised) * (sigma_down - sigmas[i])
        if not down_zero[i]:
            action = action + torch.randn_like(action) * sigma_up
        action = run.clip(action)
    return action


def _two_stage_fused(model, state, action, goal, sigmas, solver, draws: bool):
    """The fused route of the deterministic two-stage solvers (GCDenoiser.two_stage_fused); ``draws``: the reference draws ``eps = randn_like(action)``
    on every step of this sampler, churn or not - made and discarded here too, so the generator leaves the call as the reference leaves it."""
    import os
    if os.environ.get("MODE_TWO_STAGE_FUSED", "1") == "0":               # A/B runs: the step loop (captured whole by the rollout policy, as before)
        return None
    fused = getattr(model, "two_stage_fused", None)
    out = fused(state, action, goal, sigmas, solver) if fused is not None else None
    if out is not None and draws:
        for _ in range(len(sigmas) - 1):
            torch.randn_like(action)
    return out


@torch.no_grad()
def sample_heun(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0,
                s_tmax=float("inf"), s_noise=1.0):
    """Heun (2nd-order) steps of Algorithm 2: Euler predictor, trapezoidal corrector, plain Euler into sigma = 0 (gc_sampling.py:257-312).  Without
    churn, clipping, callback and extra arguments a GCDenoiser over the HIP MoDeDiT takes the whole call as one hipGraph replay (both stages' updates
    inside the head kernel; same result to fp32 rounding of the multiplied-out recurrence)."""
    if s_churn <= 0 and scaler is None and callback is None and not extra_args and _chunk_capture() is None:
        out = _two_stage_fused(model, state, action, goal, sigmas, "heun", draws=True)
        if out is not None:
            return out
    run = _Run(model, state, goal, action, extra_args, callback, scaler, "x")
    zeros = _zero_levels(sigmas)
    for i in range(len(sigmas) - 1):
        action, sigma_hat = _churn(action, sigmas, i, s_churn, s_tmin, s_tmax, s_noise)
        denoised = run.denoise(action, sigma_hat)
        d = to_d(action, sigma_hat, denoised)
        run.report(action, i, sigmas[i], sigma_hat, denoised)
        dt = sigmas[i + 1] - sigma_hat
        if zeros[i + 1]:
            action = action + d * dt
        else:
            probe = action + d * dt
            d_probe = to_d(probe, sigmas[i + 1], run.denoise(probe, sigmas[i + 1]))
            action = action + (d + d_probe) / 2 * dt
        action = run.clip(action)
    return action


@torch.no_grad()
def sample_dpm_2(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0,
                 s_tmax=float("inf"), s_noise=1.0):
    """Midpoint method in log-sigma (DPM-Solver-2 flavoured), Euler into sigma = 0 (gc_sampling.py:315-373).  Fused route as in sample_heun."""
    if s_churn <= 0 and scaler is None and callback is None and not extra_args and _chunk_capture() is None:
        out = _two_stage_fused(model, state, action, goal, sigmas, "dpm_2", draws=True)
        if out is not None:
            return out
    run = _Run(model, state, goal, action, extra_args, callback, scaler, "action")
    zeros = _zero_levels(sigmas)
    for i in range(len(sigmas) - 1):
        action, sigma_hat = _churn(action, sigmas, i, s_churn, s_tmin, s_tmax, s_noise)
        denoised = run.denoise(action, sigma_hat)
        d = to_d(action, sigma_hat, denoised)
        run.report(action, i, sigmas[i], sigma_hat, denoised)
        if zeros[i + 1]:
            action = action + d * (sigmas[i + 1] - sigma_hat)
        else:
            sigma_mid = sigma_hat.log().lerp(sigmas[i + 1].log(), 0.5).exp()
            mid = action + d * (sigma_mid - sigma_hat)
            d_mid = to_d(mid, sigma_mid, run.denoise(mid, sigma_mid))
            action = action + d_mid * (sigmas[i + 1] - sigma_hat)
        action = run.clip(action)
    return action


@torch.no_grad()
def sample_dpm_2_ancestral(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, eta=1.0):
    """Midpoint step to sigma_down plus ancestral noise (gc_sampling.py:376-410)."""
    run = _Run(model, state, goal, action, extra_args, callback, scaler, "x")
    downs, ups, down_zero = _ancestral_plan(sigmas, eta)
    for i in range(len(sigmas) - 1):
        denoised = run.denoise(action, sigmas[i])
        sigma_down, sigma_up = downs[i], ups[i]
        run.report(action, i, sigmas[i], sigmas[i], denoised)
        d = to_d(action, sigmas[i], denoised)
        if down_zero[i]:
            action = action + d * (sigma_down - sigmas[i])
        else:
            sigma_mid = sigmas[i].log().lerp(sigma_down.log(), 0.5).exp()
            mid = action + d * (sigma_mid - sigmas[i])
            d_mid = to_d(mid, sigma_mid, run.denoise(mid, sigma_mid))
            action = action + d_mid * (sigma_down - sigmas[i])
            action = action + torch.randn_like(action) * sigma_up
        action = run.clip(action)
    return action


def linear_multistep_coeff(order, t, i, j):
    """Integral over [t_i, t_{i+1}] of the j-th Lagrange basis polynomial through the last `order` nodes (gc_sampling.py:413-427)."""
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")

    def basis(tau):
        prod = 1.0
        for k in range(order):
            if k != j:
                prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return prod
    return integrate.quad(basis, t[i], t[i + 1], epsrel=1e-4)[0]


@torch.no_grad()
def sample_lms(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, order=4):
    """Adams-Bashforth style linear multistep sampler on the Karras ODE (gc_sampling.py:430-466)."""
    run = _Run(model, state, goal, action, extra_args, callback, scaler, "x")
    nodes = sigmas.detach().cpu().numpy()
    history = []
    for i in range(len(sigmas) - 1):
        denoised = run.denoise(action, sigmas[i])
        history.append(to_d(action, sigmas[i], denoised))
        if len(history) > order:
            history.pop(0)
        run.report(action, i, sigmas[i], sigmas[i], denoised)
        cur = min(i + 1, order)
        weights = [linear_multistep_coeff(cur, nodes, i, j) for j in range(cur)]
        action = run.clip(action + sum(w * d for w, d in zip(weights, reversed(history))))
    return action


@torch.no_grad()
def sample_dpmpp_2m(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None):
    """DPM-Solver++(2M): exponential-integrator steps with a two-point extrapolation of the denoised prediction (gc_sampling.py:700-734).  Without
    callback and extra arguments a GCDenoiser over the HIP MoDeDiT takes the whole call as one hipGraph replay (the extrapolation inside the head kernel)."""
    if callback is None and not extra_args and _chunk_capture() is None:
        fused = getattr(model, "dpmpp_2m_fused", None)
        out = fused(state, action, goal, sigmas) if fused is not None else None
        if out is not None:
            return out
    run = _Run(model, state, goal, action, extra_args, callback, None, "action")
    zeros = _zero_levels(sigmas)
    previous = None
    for i in range(len(sigmas) - 1):
        denoised = run.denoise(action, sigmas[i])
        run.report(action, i, sigmas[i], sigmas[i], denoised)
        if previous is None or zeros[i + 1]:
            action = _exp_step(action, denoised, sigmas[i], sigmas[i + 1])
        else:
            h = sigmas[i].log() - sigmas[i + 1].log()
            h_last = sigmas[i - 1].log() - sigmas[i].log()
            r = h_last / h
            action = _exp_step(action, (1 + 1 / (2 * r)) * denoised - (1 / (2 * r)) * previous, sigmas[i], sigmas[i + 1])
        previous = denoised
    return action


def _dpmpp_2s_core(run, action, denoised, s_from, s_to):
    """One DPM-Solver++(2S) step s_from -> s_to (s_to > 0): half step in log-sigma with D(x), full step with D at the midpoint."""
    s_mid = (0.5 * (s_from.log() + s_to.log())).exp()                      # t + h/2 in t = -ln sigma
    h = s_from.log() - s_to.log()
    x_mid = (s_mid / s_from) * action - (-h * 0.5).expm1() * denoised
    return (s_to / s_from) * action - (-h).expm1() * run.denoise(x_mid, s_mid)


@torch.no_grad()
def sample_dpmpp_2s_ancestral(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, eta=1.0,
                              s_noise=1.0, noise_sampler=None):
    """DPM-Solver++(2S) steps to sigma_down plus ancestral noise (gc_sampling.py:874-920)."""
    run = _Run(model, state, goal, action, extra_args, callback, scaler, "action")
    noise_sampler = default_noise_sampler(action) if noise_sampler is None else noise_sampler
    downs, ups, down_zero = _ancestral_plan(sigmas, eta)
    for i in range(len(sigmas) - 1):
        denoised = run.denoise(action, sigmas[i])
        sigma_down, sigma_up = downs[i], ups[i]
        run.report(action, i, sigmas[i], sigmas[i], denoised)
        if down_zero[i]:
            action = action + to_d(action, sigmas[i], denoised) * (sigma_down - sigmas[i])
        else:
            action = _dpmpp_2s_core(run, action, denoised, sigmas[i], sigma_down)
        action = run.clip(action + noise_sampler(sigmas[i], sigmas[i + 1]) * s_noise * sigma_up)
    return action


@torch.no_grad()
def sample_dpmpp_2s(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, eta=1.0):
    """Deterministic DPM-Solver++(2S) (gc_sampling.py:956-994).  Fused route as in sample_heun (this sampler draws no noise)."""
    if scaler is None and callback is None and not extra_args and _chunk_capture() is None:
        out = _two_stage_fused(model, state, action, goal, sigmas, "dpmpp_2s", draws=False)
        if out is not None:
            return out
    run = _Run(model, state, goal, action, extra_args, callback, scaler, "action")
    zeros = _zero_levels(sigmas)
    for i in range(len(sigmas) - 1):
        denoised = run.denoise(action, sigmas[i])
        run.report(action, i, sigmas[i], sigmas[i], denoised)
        if zeros[i + 1]:
            action = action + to_d(action, sigmas[i], denoised) * (sigmas[i + 1] - sigmas[i])
        else:
            action = _dpmpp_2s_core(run, action, denoised, sigmas[i], sigmas[i + 1])
        action = run.clip(action)
    return action


@torch.no_grad()
def sample_dpmpp_2_with_lms(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None):
    """The agent's ``'debugging'`` / ``'dpmpp_2_with_lms'`` sampler (gc_sampling.py:797-830): the DPM-Solver++(2M) recurrence again —
    the reference carries the same update under two names; ``scaler`` is accepted and unused there as well."""
    return sample_dpmpp_2m(model, state, action, goal, sigmas, scaler=scaler, extra_args=extra_args, callback=callback, disable=disable)


class BrownianTreeNoiseSampler:
    """Noise for the SDE sampler drawn from one Brownian path per call site, so that refining the step grid refines the SAME sample
    path: ``W(t0, t1) / sqrt(|t1 - t0|)`` is a unit normal for every interval (gc_sampling.py:112-162).  Needs ``torchsde`` (imported
    here, not at module import: the package is optional).  ``seed`` may be a list with one entry per batch item."""

    def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda v: v):
        try:
            import torchsde
        except ImportError as e:  # pragma: no cover - depends on the image
            raise ImportError("sample_dpmpp_sde's default noise sampler needs the optional package 'torchsde'; pass noise_sampler= "
                              "(a callable (sigma, sigma_next) -> noise like x) to run without it") from e
        self.transform = transform
        lo, hi = transform(torch.as_tensor(sigma_min)), transform(torch.as_tensor(sigma_max))
        self.flip = 1.0
        if not lo < hi:
            lo, hi, self.flip = hi, lo, -1.0
        if seed is None:
            seed = torch.randint(0, 2 ** 63 - 1, []).item()
        self.per_item = isinstance(seed, (list, tuple))
        seeds = list(seed) if self.per_item else [seed]
        if self.per_item and len(seeds) != x.shape[0]:
            raise ValueError("one seed per batch item expected")
        w0 = torch.zeros_like(x[0] if self.per_item else x)
        self.trees = [torchsde.BrownianTree(lo, w0, hi, entropy=sd) for sd in seeds]

    def __call__(self, sigma, sigma_next):
        t0, t1 = self.transform(torch.as_tensor(sigma)), self.transform(torch.as_tensor(sigma_next))
        a, b, sign = (t0, t1, 1.0) if t0 < t1 else (t1, t0, -1.0)
        w = torch.stack([tree(a, b) for tree in self.trees]) * (self.flip * sign)
        return (w if self.per_item else w[0]) / (t1 - t0).abs().sqrt()


@torch.no_grad()
def sample_dpmpp_sde(model, state, action, goal, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0, scaler=None,
                     noise_sampler=None, r=1 / 2):
    """Stochastic DPM-Solver++ (the agent's ``'dpmpp_2m_sde'``, gc_sampling.py:737-793): per step a 2S-style pair of exponential-integrator
    moves (to the intermediate level t + r h, then to t_next with the two predictions blended by 1/(2r)), each split into a deterministic
    move to sigma_down and fresh noise of scale sigma_up (``get_ancestral_step``); plain Euler on the final sigma = 0 step."""
    run = _Run(model, state, goal, action, extra_args, callback, scaler, "x")
    zeros = _zero_levels(sigmas)
    if noise_sampler is None:
        noise_sampler = BrownianTreeNoiseSampler(action, sigmas[sigmas > 0].min(), sigmas.max())
    x = action
    for i in range(len(sigmas) - 1):
        denoised = run.denoise(x, sigmas[i])
        run.report(x, i, sigmas[i], sigmas[i], denoised)
        if zeros[i + 1]:
            x = x + to_d(x, sigmas[i], denoised) * (sigmas[i + 1] - sigmas[i])
            continue
        t, t_next = sigmas[i].log().neg(), sigmas[i + 1].log().neg()
        s_mid = (t + (t_next - t) * r).neg().exp()                       # sigma at the intermediate level
        w = 1 / (2 * r)
        down, up = get_ancestral_step(sigmas[i], s_mid, eta)
        x_mid = (down / sigmas[i]) * x - (down.log() - sigmas[i].log()).expm1() * denoised
        x_mid = x_mid + noise_sampler(sigmas[i], s_mid) * s_noise * up
        denoised_mid = run.denoise(x_mid, s_mid)
        down, up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta)
        blend = (1 - w) * denoised + w * denoised_mid
        x = (down / sigmas[i]) * x - (down.log() - sigmas[i].log()).expm1() * blend
        x = run.clip(x + noise_sampler(sigmas[i], sigmas[i + 1]) * s_noise * up)
    return x


class _EpsSolver:
    """DPM-Solver (Lu et al. 2022, Alg. 1-2) single steps of order 1-3 in the noise-prediction view eps = (x - D(x; sigma)) / sigma on the
    half-log-SNR axis t = -ln sigma (gc_sampling.py:524-588).  ``e`` is eps at the start of the step, evaluated once by the caller."""

    def __init__(self, run):
        self.run = run

    @staticmethod
    def sigma(t):
        return t.neg().exp()

    def eps(self, x, t):
        s = self.sigma(t)
        return (x - self.run.denoise(x, s)) / s

    def step1(self, x, t, t_next, e):
        return x - self.sigma(t_next) * (t_next - t).expm1() * e

    def step2(self, x, t, t_next, e, r1=1 / 2):
        h = t_next - t
        s1 = t + r1 * h
        e1 = self.eps(x - self.sigma(s1) * (r1 * h).expm1() * e, s1)
        return x - self.sigma(t_next) * h.expm1() * e - self.sigma(t_next) / (2 * r1) * h.expm1() * (e1 - e)

    def step3(self, x, t, t_next, e, r1=1 / 3, r2=2 / 3):
        h = t_next - t
        s1, s2 = t + r1 * h, t + r2 * h
        e1 = self.eps(x - self.sigma(s1) * (r1 * h).expm1() * e, s1)
        u2 = x - self.sigma(s2) * (r2 * h).expm1() * e - self.sigma(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1) * (e1 - e)
        e2 = self.eps(u2, s2)
        return x - self.sigma(t_next) * h.expm1() * e - self.sigma(t_next) / r2 * (h.expm1() / h - 1) * (e2 - e)


@torch.no_grad()
def sample_dpm_fast(model, state, action, goal, sigma_min, sigma_max, n, scaler=None, extra_args=None, callback=None, disable=None, eta=0.0,
                    s_noise=1.0, noise_sampler=None):
    """DPM-Solver-fast: ``n`` denoiser evaluations spent on a uniform grid in t = -ln sigma from sigma_max to sigma_min, third-order steps
    with a lower-order tail so that the budget is met exactly (gc_sampling.py:589-627, 673-697).  Callback payload: ``x, i, t, t_up,
    denoised, sigma, sigma_hat``.

    Deviation, stated: with ``noise_sampler=None`` the reference dereferences an undefined name (``default_noise_sampler(x)``,
    gc_sampling.py:590) and raises ``NameError`` — so does ``MoDEAgent.sample_loop('dpm_fast')``.  Here None means
    ``default_noise_sampler(action)``, the evident intent; with an explicit sampler both agree (tests/golden/F11)."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    run = _Run(model, state, goal, action, extra_args, None, scaler, "x")
    solver = _EpsSolver(run)
    noise_sampler = default_noise_sampler(action) if noise_sampler is None else noise_sampler
    t_start, t_end = -torch.tensor(sigma_max).log(), -torch.tensor(sigma_min, device=action.device).log()
    if not t_end > t_start and eta:                                     # gc_sampling.py:591-592
        raise ValueError("eta must be 0 for reverse sampling")
    m = n // 3 + 1
    ts = torch.linspace(t_start, t_end.cpu(), m + 1, device=action.device)
    orders = [3] * (m - 2) + [2, 1] if n % 3 == 0 else [3] * (m - 1) + [n % 3]
    x = action
    for i, order in enumerate(orders):
        t, t_next = ts[i], ts[i + 1]
        if eta:
            down, _ = get_ancestral_step(solver.sigma(t), solver.sigma(t_next), eta)
            t_to = torch.minimum(t_end, -down.log())
            up = (solver.sigma(t_next) ** 2 - solver.sigma(t_to) ** 2) ** 0.5
        else:
            t_to, up = t_next, 0.0
        e = solver.eps(x, t)
        if callback is not None:
            callback({"sigma": solver.sigma(t), "sigma_hat": solver.sigma(t), "x": x, "i": i, "t": t, "t_up": t, "denoised": x - solver.sigma(t) * e})
        x = (solver.step1, solver.step2, solver.step3)[order - 1](x, t, t_to, e)
        x = x + up * s_noise * noise_sampler(solver.sigma(t), solver.sigma(t_next))
    return x


def sample_dpm_adaptive(model, state, action, goal, sigma_min, sigma_max, extra_args=None, callback=None, disable=None, order=3, rtol=0.05,
                        atol=0.0078, h_init=0.05, pcoeff=0.0, icoeff=1.0, dcoeff=0.0, accept_safety=0.81, eta=0.0, s_noise=1.0,
                        return_info=False):
    """Not provided.  In the reference every call fails before the first denoiser evaluation: ``dpm_solver_adaptive`` reads its local
    ``noise_sampler`` before assigning it (gc_sampling.py:630, ``UnboundLocalError``), so ``MoDEAgent.sample_loop('dpm_adaptive')`` has no
    behaviour to match and no output to pin an implementation against."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    raise NotImplementedError("sample_dpm_adaptive: the reference implementation raises UnboundLocalError on every call "
                              "(gc_sampling.py:630); there is no reference behaviour to reproduce")


__all__ = ["get_sigmas_vp", "get_sigmas_ve", "cosine_beta_schedule", "get_iddpm_sigmas", "to_d", "default_noise_sampler", "get_ancestral_step",
           "sample_euler", "sample_euler_ancestral", "sample_heun", "sample_dpm_2", "sample_dpm_2_ancestral", "linear_multistep_coeff",
           "sample_lms", "sample_dpmpp_2m", "sample_dpmpp_2s_ancestral", "sample_dpmpp_2s", "sample_dpmpp_sde", "sample_dpm_fast",
           "sample_dpm_adaptive", "sample_dpmpp_2_with_lms", "BrownianTreeNoiseSampler"]
