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
from tolerances import BF16_OUT, BF16_OUT_FUZZ, FP32_OUT

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
    with pytest.raises(NotImplementedError):       # fails on every call in the reference too (F11 reference_errors): nothing to match
        gc_sampling.sample_dpm_adaptive(den, state, inp["x0"], inp["goals"], 1e-3, 80.0)


def _noise_table(shape, seed, calls=64):
    """The deterministic noise sampler of oracle/gen_golden_samplers2.py: call k returns slab k of a seeded normal table."""
    tab = torch.randn((calls,) + tuple(shape), generator=torch.Generator().manual_seed(seed))
    k = [0]

    def sampler(sigma, sigma_next):
        k[0] += 1
        return tab[k[0] - 1].to(torch.as_tensor(sigma).device)
    return sampler


def _f11_runs(g, den, state, x0, goals, sig):
    """Every vector of F11_samplers.npz re-computed with samplers.py; yields (key, result)."""
    seed = int(g["noise_seed"])
    yield "dpmpp_2_with_lms", samplers.sample_dpmpp_2_with_lms(den, state, x0, goals, sig, disable=True)
    for eta in (0.0, 1.0):
        yield f"dpmpp_sde_eta{eta:g}", samplers.sample_dpmpp_sde(den, state, x0, goals, sig, disable=True, eta=eta, noise_sampler=_noise_table(x0.shape, seed))
    yield "dpmpp_sde_eta1_r0.25_snoise0.5", samplers.sample_dpmpp_sde(den, state, x0, goals, sig, disable=True, eta=1.0, s_noise=0.5, r=0.25,
                                                                      noise_sampler=_noise_table(x0.shape, seed))
    smin, smax = sig[-2].item(), sig[0].item()
    for n in (len(sig), 9, 10, 4):
        for eta in (0.0, 0.6):
            yield f"dpm_fast_n{n}_eta{eta:g}", samplers.sample_dpm_fast(den, state, x0, goals, smin, smax, n, disable=True, eta=eta,
                                                                        noise_sampler=_noise_table(x0.shape, seed))


def test_remaining_samplers_vs_reference_on_oracle_denoiser(golden):
    """sample_dpmpp_2_with_lms / sample_dpm_fast / sample_dpmpp_sde against reference runs with an explicit deterministic noise sampler
    (tests/golden/F11_samplers.npz): the noise call order, its scale and the ancestral split are pinned, not only the eta = 0 path."""
    g = golden("F11_samplers")
    cfg = get_config("c1e4"); sd = make_state_dict(cfg, int(g["seed"])); inp = make_inputs(cfg, 8, int(g["seed"]) + 1)

    def den(state, action, goal, sigma, **kw):
        return O.denoiser_forward(sd, cfg, 0.5, state["state_images"], action, goal, sigma)
    state = {"state_images": inp["state_images"]}
    sig = torch.from_numpy(g["sigmas"])
    seen = set()
    for key, x in _f11_runs(g, den, state, inp["x0"], inp["goals"], sig):
        assert rel(x, g[key]) < 2e-5, (key, rel(x, g[key]))
        seen.add(key)
    assert seen == {k for k in g.files if g[k].dtype == np.float32 and g[k].ndim == 3}
    # callback payloads
    got = []
    samplers.sample_dpm_fast(den, state, inp["x0"], inp["goals"], sig[-2].item(), sig[0].item(), 4, disable=True,
                             noise_sampler=_noise_table(inp["x0"].shape, 7), callback=lambda d: got.append(sorted(d.keys())))
    assert ",".join(got[0]) == str(g["dpm_fast_callback_keys"]) and len(got) == int(g["dpm_fast_callback_calls_n4"])
    got = []
    samplers.sample_dpmpp_sde(den, state, inp["x0"], inp["goals"], sig[-3:], disable=True, noise_sampler=_noise_table(inp["x0"].shape, 7),
                              callback=lambda d: got.append(sorted(d.keys())))
    assert ",".join(got[0]) == str(g["dpmpp_sde_callback_keys"]) and len(got) == 2
    # recorded reference failures: dpm_fast without a noise sampler (NameError there; default_noise_sampler here, documented) and dpm_adaptive
    assert dict(kv.split("=") for kv in g["reference_errors"].tolist()) == {"dpm_fast_default_noise": "NameError", "dpm_adaptive": "UnboundLocalError"}
    torch.manual_seed(0)
    out = samplers.sample_dpm_fast(den, state, inp["x0"], inp["goals"], sig[-2].item(), sig[0].item(), 4, disable=True, eta=0.5)
    assert torch.isfinite(out).all()
    with pytest.raises(ValueError):
        samplers.sample_dpm_fast(den, state, inp["x0"], inp["goals"], 0.0, 80.0, 4)
    try:
        import torchsde  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="torchsde"):       # default Brownian-tree noise needs the optional package
            samplers.sample_dpmpp_sde(den, state, inp["x0"], inp["goals"], sig[-3:], disable=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("fp32", FP32_OUT), ("bf16", BF16_OUT)])
def test_samplers_on_hip_denoiser(golden, dtype, tol):
    g = golden("F10_samplers")
    cfg = get_config("c1e4"); sd = make_state_dict(cfg, int(g["seed"])); inp = {k: v.cuda() for k, v in make_inputs(cfg, 8, int(g["seed"]) + 1).items()}
    m = M.MoDeDiT(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cuda", goal_conditioned=True, action_dim=7, embed_dim=cfg.embed_dim,
                  embed_pdrob=0, attn_pdrop=0.3, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=1, obs_seq_len=1, action_seq_len=10,
                  num_experts=cfg.num_experts, top_k=cfg.top_k, compute_dtype=dtype)
    m.load_state_dict(sd)
    den = M.GCDenoiser(m.cuda().eval(), 0.5).eval()
    state = {"state_images": inp["state_images"]}
    errs = {}
    for key in g.files:
        if ":" not in key:
            continue
        name, sched = key.split(":")
        x = RUNS[name](den, state, inp["x0"], inp["goals"], torch.from_numpy(g[f"sigmas_{sched}"]).cuda())
        errs[key] = rel(x, g[key])
    g2 = golden("F11_samplers")
    for key, x in _f11_runs(g2, den, state, inp["x0"], inp["goals"], torch.from_numpy(g2["sigmas"]).cuda()):
        if key.startswith("dpm_fast_n4"):
            continue                                  # 2 coarse steps from sigma = 80: |x| ~ 75-94, an ill-conditioned solve, CPU-checked only
        errs[key] = rel(x, g2[key])
    worst = sorted(errs.items(), key=lambda kv: -kv[1])
    print(f"samplers on the HIP denoiser, {dtype}: " + ", ".join(f"{k} {v:.2e}" for k, v in worst[:6]) + f" ... ({len(errs)} runs, tol {tol:g})")
    # dpm_fast (DPM-Solver-3 in a handful of large steps) amplifies the denoiser's error the most - in fp32 too it is the worst run by 2.5x (1.7e-5
    # against <= 6.7e-6 for every other sampler) -: its six runs measure 0.96-1.09e-2 in bf16 and are held to the wider bf16 envelope; every other
    # sampler stays under the fixture tolerance
    for k, v in errs.items():
        assert v < (BF16_OUT_FUZZ if (dtype == "bf16" and k.startswith("dpm_fast")) else tol), (k, v)

@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_euler_without_churn_is_the_fused_first_order_solve(dtype):
    """sample_euler with s_churn = 0 multiplies out to the DDIM update: on the HIP denoiser it takes the fused one-replay path - the same tensor as
    sample_ddim, bit for bit - and agrees with its own step-by-step loop (forced by a callback / a scaler / churn arguments that change nothing) to
    the rounding of the rearranged update."""
    from test_gpu_model import build
    cfg, sd, m = build("c1e4", 77, dtype)
    den = M.GCDenoiser(m, 0.5).eval()
    inp = {k: v.cuda() for k, v in make_inputs(cfg, 6, 5).items()}
    state = {"state_images": inp["state_images"]}
    sig = gc_sampling.get_sigmas_exponential(10, 0.001, 80.0, "cuda")
    torch.cuda.manual_seed(123)
    fused = samplers.sample_euler(den, state, inp["x0"], inp["goals"], sig, disable=True)
    after_fused = torch.randn(4, device="cuda")
    ddim = gc_sampling.sample_ddim(den, state, inp["x0"], inp["goals"], sig, disable=True)
    assert torch.equal(fused, ddim)
    steps = []
    torch.cuda.manual_seed(123)
    loop = samplers.sample_euler(den, state, inp["x0"], inp["goals"], sig, disable=True, callback=lambda d: steps.append(d["i"]))
    after_loop = torch.randn(4, device="cuda")
    assert steps == list(range(10))
    # the reference draws eps on every step even without churn (gc_sampling.py:196): both routes leave the generator in the same state
    assert torch.equal(after_fused, after_loop)
    r = rel(fused, loop)
    print(f"euler fused vs step loop, {dtype}: {r:.2e}")
    assert r < (2e-6 if dtype == "fp32" else BF16_OUT), r

    class Same:
        def clip_output(self, x):
            return x
    assert torch.equal(samplers.sample_euler(den, state, inp["x0"], inp["goals"], sig, scaler=Same(), disable=True), loop)   # a scaler keeps the step loop
    m.train()
    try:                                                  # training mode (dropout live in the reference): never the fused path
        assert den.first_order_ode_fused(state, inp["x0"], inp["goals"], sig) is None
    finally:
        m.eval()


@pytest.mark.gpu
def test_graphed_denoise_observation_cache_is_never_stale():
    """``denoise_graphed`` keeps the observation embeddings of one sampler run beside its graph.  Whatever changes between two calls - the tensors'
    contents in place, new tensor objects (also ones the allocator puts at a recycled address), the weights - the next call must see it: every call is
    compared with the eager chain on the same inputs."""
    from test_gpu_model import build
    cfg, sd, m = build("c1e4", 210, "fp32")
    den = M.GCDenoiser(m, 0.5).eval()
    B = 5
    inp = {k: v.cuda() for k, v in make_inputs(cfg, B, 3).items()}
    state = {"state_images": inp["state_images"].clone()}
    goal = inp["goals"].clone()
    x = inp["actions"].clone()
    sig = torch.tensor(1.7, device="cuda")

    def check(tag):
        with torch.no_grad():
            fast = den.denoise_uniform(state, x, goal, sig)
            ref = den(state, x, goal, sig * torch.ones(B, device="cuda"))
        assert fast is not None
        r = float((fast - ref).norm() / ref.norm())
        assert r < 1e-5, (tag, r)
    check("first"); check("same tensors: cached")
    state["state_images"].mul_(1.5); check("state changed in place")
    goal.add_(0.3); check("goal changed in place")
    for i in range(4):                                                        # fresh objects every call (what a rollout loop does); old ones die -> addresses recycle
        state = {"state_images": torch.randn_like(inp["state_images"]) * (1 + i)}
        goal = torch.randn_like(inp["goals"])
        check(f"new tensors {i}")
    with torch.no_grad():
        m.tok_emb.weight.mul_(1.1); m.goal_emb.weight.add_(0.01)
    check("weights changed in place")
    sig = torch.tensor(0.2, device="cuda"); check("another sigma, same observations")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_dpmpp_2m_takes_the_fused_multistep_chain(dtype):
    """sample_dpmpp_2m on the HIP denoiser is one hipGraph replay whose head kernel forms the two-point extrapolation (ModeHeadDesc.den_prev): equal to its own
    step loop (forced by a callback) to fp32 rounding, for exponential / Karras schedules, 1- and 2-level schedules (no multistep step at all), repeated
    and interleaved with the DDIM graph (one captured chain per solver), and not in training mode."""
    from test_gpu_model import build
    cfg, sd, m = build("c1e4", 78, dtype)
    den = M.GCDenoiser(m, 0.5).eval()
    inp = {k: v.cuda() for k, v in make_inputs(cfg, 6, 9).items()}
    state = {"state_images": inp["state_images"]}
    tol = 3e-6 if dtype == "fp32" else BF16_OUT
    for sig in (gc_sampling.get_sigmas_exponential(10, 0.001, 80.0, "cuda"), gc_sampling.get_sigmas_karras(7, 0.01, 40.0, 7.0, "cuda"),
                gc_sampling.get_sigmas_exponential(2, 0.01, 10.0, "cuda"), gc_sampling.get_sigmas_exponential(1, 0.5, 0.5, "cuda")):
        steps = []
        loop = samplers.sample_dpmpp_2m(den, state, inp["x0"], inp["goals"], sig, disable=True, callback=lambda d: steps.append(d["i"]))
        assert steps == list(range(len(sig) - 1))
        fused = samplers.sample_dpmpp_2m(den, state, inp["x0"], inp["goals"], sig, disable=True)
        assert rel(fused, loop) < tol, (len(sig), rel(fused, loop))
        ddim = gc_sampling.sample_ddim(den, state, inp["x0"], inp["goals"], sig, disable=True)           # the other solver's graph in between
        again = samplers.sample_dpmpp_2m(den, state, inp["x0"], inp["goals"], sig, disable=True)
        assert torch.equal(fused, again)
        if len(sig) <= 3:
            assert torch.equal(fused, ddim)                       # no step has both a predecessor and a non-zero target: the plain update, bit for bit
        else:
            assert rel(fused, ddim) > 10 * tol                    # a different solver, not the DDIM chain under another name
    # a second input on the cached graph
    inp2 = {k: v.cuda() for k, v in make_inputs(cfg, 6, 10).items()}
    sig = gc_sampling.get_sigmas_exponential(10, 0.001, 80.0, "cuda")
    f2 = samplers.sample_dpmpp_2m(den, {"state_images": inp2["state_images"]}, inp2["x0"], inp2["goals"], sig, disable=True)
    l2 = samplers.sample_dpmpp_2m(den, {"state_images": inp2["state_images"]}, inp2["x0"], inp2["goals"], sig, disable=True, callback=lambda d: None)
    assert rel(f2, l2) < tol
    m.train()
    try:
        assert den.dpmpp_2m_fused(state, inp["x0"], inp["goals"], sig) is None
    finally:
        m.eval()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_two_stage_solvers_take_the_fused_chain(dtype):
    """sample_heun / sample_dpm_2 / sample_dpmpp_2s (no churn, clipping, callback) on the HIP denoiser: one hipGraph replay with both stages' linear updates
    inside the head kernel (ModeHeadDesc.lin) - equal to their own step loops (forced by a callback) to fp32 rounding of the multiplied-out recurrence,
    for two schedules and a two-level schedule (a single Euler step into sigma = 0), repeatable, a new schedule on the cached graph, the generator left
    where the step loop leaves it, and never in training mode."""
    from test_gpu_model import build
    cfg, sd, m = build("c1e4", 79, dtype)
    den = M.GCDenoiser(m, 0.5).eval()
    inp = {k: v.cuda() for k, v in make_inputs(cfg, 6, 12).items()}
    state = {"state_images": inp["state_images"]}
    tol = 1e-5 if dtype == "fp32" else BF16_OUT
    fns = {"heun": samplers.sample_heun, "dpm_2": samplers.sample_dpm_2, "dpmpp_2s": samplers.sample_dpmpp_2s}
    for name, fn in fns.items():
        for sig in (gc_sampling.get_sigmas_exponential(10, 0.001, 80.0, "cuda"), gc_sampling.get_sigmas_karras(6, 0.01, 40.0, 7.0, "cuda"),
                    gc_sampling.get_sigmas_exponential(1, 0.5, 0.5, "cuda")):
            steps = []
            torch.cuda.manual_seed(5)
            loop = fn(den, state, inp["x0"], inp["goals"], sig, disable=True, callback=lambda d: steps.append(d["i"]))
            after_loop = torch.randn(3, device="cuda")
            assert steps == list(range(len(sig) - 1))
            torch.cuda.manual_seed(5)
            fused = fn(den, state, inp["x0"], inp["goals"], sig, disable=True)
            after_fused = torch.randn(3, device="cuda")
            assert torch.equal(after_loop, after_fused), name
            assert rel(fused, loop) < tol, (name, len(sig), rel(fused, loop))
            assert torch.equal(fused, fn(den, state, inp["x0"], inp["goals"], sig, disable=True))
        assert "graph:" + name in m._route_cache
    m.train()
    try:
        assert den.two_stage_fused(state, inp["x0"], inp["goals"], gc_sampling.get_sigmas_exponential(4, 0.01, 10.0, "cuda"), "heun") is None
    finally:
        m.eval()


def test_two_stage_plans_reproduce_the_reference_samplers_on_the_oracle_denoiser(golden):
    """Host logic of the fused two-stage solvers (MoDeDiT._two_stage_plan: which sigma each denoiser evaluation sees, which buffers it reads / writes, the four
    coefficients of its linear update) executed on the CPU with the ORACLE's denoiser: the multiplied-out recurrences reproduce the reference's heun /
    dpm_2 / dpmpp_2s outputs (F10) for every schedule of the fixture - the same plan drives the head kernel on the GPU."""
    from mode_diffusion_policy_amd.modedit import MoDeDiT
    g = golden("F10_samplers")
    cfg = get_config("c1e4"); sd = make_state_dict(cfg, int(g["seed"])); inp = make_inputs(cfg, 8, int(g["seed"]) + 1)
    seen = 0
    for key in g.files:
        if ":" not in key:
            continue
        name, sched = key.split(":")
        if name not in ("heun", "dpm_2", "dpmpp_2s"):
            continue
        sig = torch.from_numpy(g[f"sigmas_{sched}"])
        plan = MoDeDiT._two_stage_plan(name, [float(v) for v in sig.tolist()])
        assert len(plan) == 2 * (len(sig) - 1) - int(float(sig[-1]) == 0.0)
        bufs = [inp["x0"].clone().double(), torch.zeros_like(inp["x0"]).double(), torch.zeros_like(inp["x0"]).double()]
        for sigma, xin, xout, lin, a1, a2, dout in plan:
            x = bufs[xin]
            den = O.denoiser_forward(sd, cfg, 0.5, inp["state_images"], x.float(), inp["goals"], torch.full((8,), sigma)).double()
            v = lin[0] * x + lin[1] * den
            if a1 is not None:
                v = v + lin[2] * bufs[a1]
            if a2 is not None:
                v = v + lin[3] * bufs[a2]
            if dout is not None:
                bufs[dout] = den
            bufs[xout] = v
        assert rel(bufs[0], g[key]) < 2e-5, (key, rel(bufs[0], g[key]))
        seen += 1
    assert seen >= 3


def test_dpmpp_2m_multistep_weights_reproduce_the_reference_on_the_oracle_denoiser(golden):
    """Host logic of the fused DPM-Solver++(2M) chain (MoDeDiT._dpmpp_2m_weights -> scal[:, 3]) with the head kernel's update written out on the CPU:
    x <- r x + (1 - r) ((1 + c) D - c D_old), c = 0 on the first step and into sigma = 0 - the reference's outputs (F10) on the oracle's denoiser."""
    from mode_diffusion_policy_amd.modedit import MoDeDiT
    g = golden("F10_samplers")
    cfg = get_config("c1e4"); sd = make_state_dict(cfg, int(g["seed"])); inp = make_inputs(cfg, 8, int(g["seed"]) + 1)
    seen = 0
    for key in g.files:
        if not key.startswith("dpmpp_2m:"):
            continue
        sig = torch.from_numpy(g["sigmas_" + key.split(":")[1]])
        c = MoDeDiT._dpmpp_2m_weights(sig)
        assert float(c[0]) == 0.0 and (float(sig[-1]) != 0.0 or float(c[-1]) == 0.0) and bool((c[1:-1] > 0).all())
        x = inp["x0"].clone(); prev = None
        for i in range(len(sig) - 1):
            den = O.denoiser_forward(sd, cfg, 0.5, inp["state_images"], x, inp["goals"], sig[i] * torch.ones(8))
            dd = den if (prev is None or float(c[i]) == 0.0) else (1.0 + c[i]) * den - c[i] * prev
            r = sig[i + 1] / sig[i]
            x = r * x + (1.0 - r) * dd
            prev = den
        assert rel(x, g[key]) < 2e-5, (key, rel(x, g[key]))
        seen += 1
    assert seen >= 1
