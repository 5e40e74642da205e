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
    # The two truncation bounds are scalars: evaluated on the host (fp64, the same torch kernels the reference's CPU path runs) and folded in as
    # Python floats.  `torch.as_tensor(scalar, device=cuda)` - the reference's form - is a blocking host-to-device copy: it drains the stream once
    # per training step, the GPU then idles while the host enqueues the next forward (measured: 7 ms of host wait per 14.5-ms step, and
    # 22-42 ms steps on a box with a slower launch path).
    min_cdf = float(torch.as_tensor(min_value, dtype=torch.float64).log().sub(loc).div(scale).sigmoid())
    max_cdf = float(torch.as_tensor(max_value, dtype=torch.float64).log().sub(loc).div(scale).sigmoid())
    u = torch.rand(shape, device=device, dtype=torch.float64) * (max_cdf - min_cdf) + min_cdf
    return u.logit().mul(scale).add(loc).exp().to(dtype)


# ---- the other training noise-level densities `MoDEAgent.make_sample_density` can select (mode_agent.py:691-730; edm_diffusion/utils.py:154-203).
#      Each consumes the torch RNG stream exactly like its reference counterpart (same draws, same order, same dtype), so a seeded run
#      reproduces the reference's sigma sequence.
def rand_log_normal(shape, loc=0.0, scale=1.0, device="cpu", dtype=torch.float32):
    """exp(N(loc, scale^2))."""
    return torch.randn(shape, device=device, dtype=dtype).mul_(scale).add_(loc).exp_()


def rand_log_uniform(shape, min_value, max_value, device="cpu", dtype=torch.float32):
    """exp(U(ln min, ln max))."""
    import math
    lo, hi = math.log(min_value), math.log(max_value)
    return (torch.rand(shape, device=device, dtype=dtype) * (hi - lo) + lo).exp()


def rand_uniform(shape, min_value, max_value, device="cpu", dtype=torch.float32):
    return torch.rand(shape, device=device, dtype=dtype) * (max_value - min_value) + min_value


def rand_v_diffusion(shape, sigma_data=1.0, min_value=0.0, max_value=float("inf"), device="cpu", dtype=torch.float32):
    """sigma = sigma_data * tan(pi/2 * u), u uniform between the CDF values of the truncation bounds (v-diffusion timestep density)."""
    import math
    lo = math.atan(min_value / sigma_data) * 2 / math.pi
    hi = math.atan(max_value / sigma_data) * 2 / math.pi
    u = torch.rand(shape, device=device, dtype=dtype) * (hi - lo) + lo
    return torch.tan(u * math.pi / 2) * sigma_data


def rand_split_log_normal(shape, loc, scale_1, scale_2, device="cpu", dtype=torch.float32):
    """Two half log-normals glued at exp(loc): the left half (scale_1) with probability scale_1 / (scale_1 + scale_2)."""
    half = torch.randn(shape, device=device, dtype=dtype).abs()
    u = torch.rand(shape, device=device, dtype=dtype)
    return torch.where(u < scale_1 / (scale_1 + scale_2), half * -scale_1 + loc, half * scale_2 + loc).exp()


def rand_discrete(shape, values, device="cpu", dtype=torch.float32):
    """Uniform draw from a table of noise levels."""
    idx = torch.randint(0, len(values), shape, device=device)
    return torch.index_select(values, 0, idx).to(dtype)


def make_sample_density(kind: str, sigma_data: float = 0.5, sigma_min: float = 0.001, sigma_max: float = 80.0, mean: float = -1.2, std: float = 1.2,
                        num_sampling_steps: int = 10, **cfg):
    """``MoDEAgent.make_sample_density`` (mode_agent.py:691-730): density name -> ``fn(shape, device=...)``.  'loglogistic' (the shipped
    default, mode_agent.yaml:15) needs no extra hyper-parameters; the reference's 'split-lognormal' and parts of 'v-diffusion' read keys of
    an always-empty config list and cannot run there — here they take ``loc/scale_1/scale_2`` via keyword arguments."""
    import math
    from functools import partial
    if kind == "lognormal":
        return partial(rand_log_normal, loc=mean, scale=std)
    if kind == "loglogistic":
        return partial(rand_log_logistic, loc=cfg.get("loc", math.log(sigma_data)), scale=cfg.get("scale", 0.5),
                       min_value=cfg.get("min_value", sigma_min), max_value=cfg.get("max_value", sigma_max))
    if kind == "loguniform":
        return partial(rand_log_uniform, min_value=cfg.get("min_value", sigma_min), max_value=cfg.get("max_value", sigma_max))
    if kind == "uniform":
        return partial(rand_uniform, min_value=sigma_min, max_value=sigma_max)
    if kind == "v-diffusion":
        return partial(rand_v_diffusion, sigma_data=sigma_data, min_value=cfg.get("min_value", sigma_min), max_value=cfg.get("max_value", sigma_max))
    if kind == "discrete":
        from .gc_sampling import get_sigmas_exponential
        return partial(rand_discrete, values=get_sigmas_exponential(int(num_sampling_steps * 1e5), sigma_min, sigma_max))
    if kind == "split-lognormal":
        return partial(rand_split_log_normal, loc=cfg["loc"], scale_1=cfg["scale_1"], scale_2=cfg["scale_2"])
    raise ValueError("Unknown sample density type")
