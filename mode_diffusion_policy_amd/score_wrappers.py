"""Mirror of ``mode.models.edm_diffusion.score_wrappers.GCDenoiser`` (Karras/EDM preconditioner) on the HIP denoiser.

``forward`` fuses the three scalings into the HIP launch chain: ``c_in`` is applied while the action tokens are embedded and
``F*c_out + x*c_skip`` is the epilogue of the output-head kernel (reference: score_wrappers.py:65-80 = 3 extra elementwise
launches + temporaries per call).
"""
from __future__ import annotations

import torch
from torch import nn

from .modedit import MoDeDiT
from .utils import append_dims


def _instantiate(cfg):
    """Accept a ready module, or a Hydra/OmegaConf node / plain dict with ``_target_`` (the agent passes a DictConfig because
    it is built with ``_recursive_: false``; score_wrappers.py:28)."""
    if isinstance(cfg, nn.Module):
        return cfg
    try:
        import hydra
        return hydra.utils.instantiate(cfg)
    except ImportError:
        kw = {k: v for k, v in dict(cfg).items() if k not in ("_target_", "_recursive_")}
        return MoDeDiT(**kw)


class GCDenoiser(nn.Module):
    def __init__(self, inner_model, sigma_data=1.0):
        super().__init__()
        self.inner_model = _instantiate(inner_model)
        self.sigma_data = sigma_data

    def get_scalings(self, sigma):
        """c_skip, c_out, c_in   (score_wrappers.py:31-43)."""
        c_skip = self.sigma_data ** 2 / (sigma ** 2 + self.sigma_data ** 2)
        c_out = sigma * self.sigma_data / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        c_in = 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out, c_in

    def loss(self, state, action, goal, noise, sigma, **kwargs):
        """Score-matching loss (score_wrappers.py:45-63) -> (loss, model_output).  On a HIP MoDeDiT in training mode the scalings, the
        target, the MSE and their backward are two HIP launches around the training chain (training.edm_loss); the expression below is
        the generic path (eval-mode loss, extra kwargs)."""
        m = self.inner_model
        if isinstance(m, MoDeDiT) and m.training and not kwargs and torch.is_grad_enabled():
            from .training import edm_loss
            return edm_loss(self, state, action, goal, noise, sigma)
        c_skip, c_out, c_in = [append_dims(x, action.ndim) for x in self.get_scalings(sigma)]
        noised_input = action + noise * append_dims(sigma, action.ndim)
        model_output = self.inner_model(state, noised_input * c_in, goal, sigma, **kwargs)
        target = (action - c_skip * noised_input) / c_out
        return (model_output - target).pow(2).flatten(1).mean(), model_output

    def forward(self, state, action, goal, sigma, **kwargs):
        """D(x; sigma) = F(x*c_in)*c_out + x*c_skip   (score_wrappers.py:65-80)."""
        m = self.inner_model
        if isinstance(m, MoDeDiT) and not m.training and not kwargs:
            return m.denoise(state, action, goal, sigma, self.sigma_data)
        c_skip, c_out, c_in = [append_dims(x, action.ndim) for x in self.get_scalings(sigma)]
        return m(state, action * c_in, goal, sigma, **kwargs) * c_out + action * c_skip

    def denoise_uniform(self, state, action, goal, sigma):
        """D(x; sigma) for ONE noise level shared by the whole batch (what every k-diffusion style sampler asks for: ``sigma * ones``), as one
        hipGraph replay of the HIP chain (``MoDeDiT.denoise_graphed``).  Returns None when the fast path does not apply (training mode, token /
        goal routing, a foreign inner model) - the caller then takes ``forward``."""
        m = self.inner_model
        if isinstance(m, MoDeDiT) and not m.training and not torch.is_grad_enabled():
            return m.denoise_graphed(state, action, goal, sigma, self.sigma_data)
        return None

    def first_order_ode_fused(self, state, action, goal, sigmas):
        """The whole deterministic first-order solve  x <- (s'/s) x + (1 - s'/s) D(x; s)  over ``sigmas`` as one hipGraph replay
        (``MoDeDiT.sample_ddim_fused``).  That update is both sample_ddim's (gc_sampling.py:922-951: exp(-t')/exp(-t) = s'/s, -expm1(-h) = 1 - s'/s)
        and, multiplied out, sample_euler's without churn (gc_sampling.py:165-211: x + (x - D)/s (s' - s)).  None when the fast path does not apply."""
        m = self.inner_model
        if isinstance(m, MoDeDiT) and not m.training and not torch.is_grad_enabled() and torch.is_tensor(sigmas) and sigmas.dim() == 1:
            return m.sample_ddim_fused(state, action, goal, sigmas, self.sigma_data)
        return None

    def dpmpp_2m_fused(self, state, action, goal, sigmas):
        """sample_dpmpp_2m (gc_sampling.py:700-734) as one hipGraph replay: the same chain as the first-order solve, the head kernel applying the step to the
        two-point extrapolation (1 + 1/(2r)) D - (1/(2r)) D_old of the denoised prediction.  None when the fast path does not apply."""
        m = self.inner_model
        if isinstance(m, MoDeDiT) and not m.training and not torch.is_grad_enabled() and torch.is_tensor(sigmas) and sigmas.dim() == 1:
            return m.sample_ddim_fused(state, action, goal, sigmas, self.sigma_data, solver="dpmpp_2m")
        return None

    def two_stage_fused(self, state, action, goal, sigmas, solver: str):
        """sample_heun / sample_dpm_2 / sample_dpmpp_2s without churn, clipping or callback as one hipGraph replay (``MoDeDiT.sample_two_stage_fused``:
        every stage's update is linear and runs inside the head kernel).  None when the fast path does not apply."""
        m = self.inner_model
        if isinstance(m, MoDeDiT) and not m.training and not torch.is_grad_enabled() and torch.is_tensor(sigmas) and sigmas.dim() == 1:
            return m.sample_two_stage_fused(state, action, goal, sigmas, self.sigma_data, solver)
        return None

    def get_params(self):
        return self.inner_model.parameters()
