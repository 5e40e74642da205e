"""Mirror of ``mode.models.edm_diffusion.gc_sampling`` on the HIP denoiser: schedules, the default DDIM sampler (fused + hipGraph) and, from
``samplers.py``, the other samplers behind the reference signatures.

``sample_ddim`` keeps the reference signature (gc_sampling.py:922-951).  When the model is our ``GCDenoiser(MoDeDiT)`` the whole
sampler runs as one launch chain: observation embeddings hoisted out of the step loop, routing for ALL steps resolved up front
(the router only sees the noise level, so this is the reference's "pre-cache per noise level" without duplicating weights),
``x <- r*x + (1-r)*denoised`` fused into the output-head kernel, and the chain captured once into a hipGraph and replayed.
The agent star-imports the reference module and relies on it for ``np``, ``plt`` (optional) and ``math`` (mode_agent.py:16):
those names stay exported.
"""
from __future__ import annotations

import math

import numpy as np
import torch

try:  # the reference module exports matplotlib's pyplot as `plt`; optional here
    from matplotlib import pyplot as plt  # noqa: F401
except Exception:  # pragma: no cover
    plt = None

from .score_wrappers import GCDenoiser
from .modedit import MoDeDiT
from .samplers import *  # noqa: F401,F403  the other samplers / schedules MoDEAgent.sample_loop and get_noise_schedule dispatch to
from .samplers import tag_schedule


def append_zero(action):
    return torch.cat([action, action.new_zeros([1])])


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0, device="cpu"):
    """Karras et al. (2022) schedule (gc_sampling.py:26-32)."""
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return tag_schedule(append_zero(sigmas).to(device), "karras", n, sigma_min, sigma_max, rho)


def get_sigmas_exponential(n, sigma_min, sigma_max, device="cpu"):
    """exp(linspace(ln smax, ln smin, n)) ++ [0]   (gc_sampling.py:35-38) — the default ('exponential', mode_agent.yaml:14)."""
    sigmas = torch.linspace(math.log(sigma_max), math.log(sigma_min), n, device=device).exp()
    return tag_schedule(append_zero(sigmas), "exponential", n, sigma_min, sigma_max)


def get_sigmas_linear(n, sigma_min, sigma_max, device="cpu"):
    """gc_sampling.py:41-44."""
    return tag_schedule(append_zero(torch.linspace(sigma_max, sigma_min, n, device=device)), "linear", n, sigma_min, sigma_max)


@torch.no_grad()
def sample_ddim(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, eta=1.0):
    """DPM-Solver-1 / DDIM sampler (gc_sampling.py:922-951)."""
    extra_args = {} if extra_args is None else extra_args
    if (isinstance(model, GCDenoiser) and isinstance(model.inner_model, MoDeDiT) and not model.inner_model.training
            and callback is None and not extra_args):
        return model.inner_model.sample_ddim_fused(state, action, goal, sigmas, model.sigma_data)
    # generic path: any callable denoiser, reference step order
    s_in = action.new_ones([action.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(state, action, goal, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({"action": action, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        t, t_next = sigmas[i].log().neg(), sigmas[i + 1].log().neg()
        h = t_next - t
        action = (t_next.neg().exp() / t.neg().exp()) * action - (-h).expm1() * denoised
    return action
