"""The non-DDIM samplers and noise schedules (mode_diffusion_policy_amd/samplers.py) against golden vectors produced by
the reference functions (oracle/gen_golden_samplers.py -> tests/golden/F10_*.npz).

CPU: the sampler recurrences (host logic) are driven by the ORACLE's denoiser, so only the sampler code itself is under test.
GPU: the same samplers drive the HIP GCDenoiser (fp32 parity mode and bf16)."""
import inspect

import numpy as np
import pytest
import torch

import mode_diffusion_policy_amd as M
from mode_diffusion_policy_amd import gc_sampling, samplers
from oracle import mode_oracle as O
from oracle.weights import get_config, make_inputs, make_state_dict

RUNS = {
    "euler": lambda den, st, x0, g, s: samplers.sample_euler(den, st, x0, g, s, disable=True),
    "heun": lambda den, st, x0, g, s: samplers.sample_heun(den, st, x0, g, s, disable=True),
    "dpm_2": lambda den, st, x0, g, s: samplers.sample_dpm_2(den, st, x0, g, s, disable=True),
    "lms": lambda den, st, x0, g, s: samplers.sample_lms(den, st, x0, g, s, disable=True),
    "dpmpp_2m": lambda den, st, x0, g, s: samplers.sample_dpmpp_2m(den, st, x0, g, s, disable=True),
    "dpmpp_2s": lambda den, st, x0, g, s: samplers.sample_dpmpp_2s(den, st, x0, g, s, disable=True),
    "euler_ancestral_eta0": lambda den, st, x0, g, s: samplers.sample_euler_ancestral(den, st, x0, g, s, disable=True, eta=0.0),
    "dpm_2_ancestral_eta0": lambda den, st, x0, g, s: samplers.sample_dpm_2_ancestral(den, st, x0, g, s, disable=True, eta=0.0),
    "dpmpp_2s_ancestral_eta0": lambda den, st, x0, g, s: samplers.sample_dpmpp_2s_ancestral(den, st, x0, g, s, disable=True, eta=0.0),
}


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm())


def test_schedules_match_reference(golden):
    g = golden("F10_schedules")
    for n in (5, 10):
        for name, fn in (("vp", lambda: gc_sampling.get_sigmas_vp(n)), ("ve", lambda: gc_sampling.get_sigmas_ve(n, 1e-3, 80.0)),
                         ("cosine_beta", lambda: gc_sampling.cosine_beta_schedule(n)), ("iddpm", lambda: gc_sampling.get_iddpm_sigmas(n, 1e-3, 80.0)),
                         ("karras", lambda: gc_sampling.get_sigmas_karras(n, 1e-3, 80.0, 7)), ("linear", lambda: gc_sampling.get_sigmas_linear(n, 1e-3, 80.0))):
            got = fn().numpy()
            assert got.shape == g[f"{name}_n{n}"].shape and got.dtype == np.float32, name
            np.testing.assert_allclose(got, g[f"{name}_n{n}"], rtol=2e-6, atol=1e-9, err_msg=f"{name} n={n}")


def test_samplers_host_logic_vs_reference_on_oracle_denoiser(golden):
    g = golden("F10_samplers")
    cfg = get_config("c1e4"); sd = make_state_dict(cfg, int(g["seed"])); inp = make_inputs(cfg, 8, int(g["seed"]) + 1)

    def den(state, action, goal, sigma, **kw):                    # the oracle's GCDenoiser.forward (CPU fp32)
        return O.denoiser_forward(sd, cfg, 0.5, state["state_images"], action, goal, sigma)
    state = {"state_images": inp["state_images"]}
    for key in g.files:
        if ":" not in key:
            continue
        name, sched = key.split(":")
        x = RUNS[name](den, state, inp["x0"], inp["goals"], torch.from_numpy(g[f"sigmas_{sched}"]))
        assert rel(x, g[key]) < 2e-5, key
    # observable behaviour: callback payload keys and signatures follow the reference, sampler by sampler
    want = dict(kv.split("=") for kv in g["callback_keys"].tolist())
    fns = {"euler": samplers.sample_euler, "heun": samplers.sample_heun, "dpm_2": samplers.sample_dpm_2, "lms": samplers.sample_lms,
           "dpmpp_2m": samplers.sample_dpmpp_2m, "dpmpp_2s": samplers.sample_dpmpp_2s, "euler_ancestral": samplers.sample_euler_ancestral,
           "dpm_2_ancestral": samplers.sample_dpm_2_ancestral, "dpmpp_2s_ancestral": samplers.sample_dpmpp_2s_ancestral}
    sig3 = torch.from_numpy(g["sigmas_exponential"])[-3:]
    for name, fn in fns.items():
        seen = []
        out = fn(den, state, inp["x0"], inp["goals"], sig3, disable=True, callback=lambda d: seen.append(sorted(d.keys())))
        assert ",".join(seen[0]) == want[name] and len(seen) == 2 and torch.isfinite(out).all(), name
        assert list(inspect.signature(fn).parameters)[:5] == ["model", "state", "action", "goal", "sigmas"]
    # stochastic variants (eta = 1 / churn) run and stay finite; the scaler hook is honoured
    class Clip:
        def clip_output(self, x):
            return x.clamp(-1, 1)
    torch.manual_seed(0)
    for fn, kw in ((samplers.sample_euler_ancestral, {}), (samplers.sample_dpm_2_ancestral, {}), (samplers.sample_dpmpp_2s_ancestral, {}),
                   (samplers.sample_heun, {"s_churn": 5.0}), (samplers.sample_euler, {"s_churn": 5.0})):
        out = fn(den, state, inp["x0"], inp["goals"], sig3, scaler=Clip(), disable=True, **kw)
        assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0
    for name in ("sample_dpmpp_sde", "sample_dpm_fast", "sample_dpm_adaptive"):
        with pytest.raises(NotImplementedError):
            getattr(gc_sampling, name)(den, state, inp["x0"], inp["goals"], sig3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 2e-2)])
def test_samplers_on_hip_denoiser(golden, dtype, tol):
    g = golden("F10_samplers")
    cfg = get_config("c1e4"); sd = make_state_dict(cfg, int(g["seed"])); inp = {k: v.cuda() for k, v in make_inputs(cfg, 8, int(g["seed"]) + 1).items()}
    m = M.MoDeDiT(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cuda", goal_conditioned=True, action_dim=7, embed_dim=cfg.embed_dim,
                  embed_pdrob=0, attn_pdrop=0.3, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=1, obs_seq_len=1, action_seq_len=10,
                  num_experts=cfg.num_experts, top_k=cfg.top_k, compute_dtype=dtype)
    m.load_state_dict(sd)
    den = M.GCDenoiser(m.cuda().eval(), 0.5).eval()
    state = {"state_images": inp["state_images"]}
    for key in g.files:
        if ":" not in key:
            continue
        name, sched = key.split(":")
        x = RUNS[name](den, state, inp["x0"], inp["goals"], torch.from_numpy(g[f"sigmas_{sched}"]).cuda())
        assert rel(x, g[key]) < tol, (key, rel(x, g[key]))
