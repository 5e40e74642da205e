"""Mirror of the pieces of ``mode.models.edm_diffusion.utils`` the denoising path uses."""
from __future__ import annotations

import torch


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    """Append trailing singleton dims until ``x.ndim == target_dims`` (edm_diffusion/utils.py:146-151)."""
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


def rand_log_logistic(shape, loc=0.0, scale=1.0, min_value=0.0, max_value=float("inf"), device="cpu", dtype=torch.float32):
    """Truncated log-logistic draw, fp64 internally (edm_diffusion/utils.py:159-166) — the default sigma density
    (mode_agent.py:703-708: loc = ln(sigma_data), scale = 0.5, [sigma_min, sigma_max])."""
    min_value = torch.as_tensor(min_value, device=device, dtype=torch.float64)
    max_value = torch.as_tensor(max_value, device=device, dtype=torch.float64)
    min_cdf = min_value.log().sub(loc).div(scale).sigmoid()
    max_cdf = max_value.log().sub(loc).div(scale).sigmoid()
    u = torch.rand(shape, device=device, dtype=torch.float64) * (max_cdf - min_cdf) + min_cdf
    return u.logit().mul(scale).add(loc).exp().to(dtype)
