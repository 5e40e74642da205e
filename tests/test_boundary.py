"""CPU: drop-in boundary of the reference's Python seam + the C-ABI library's exported surface.
No compute kernels are launched here (there is no GPU in the build container)."""
import os
import re

import numpy as np
import pytest
import torch

import mode_diffusion_policy_amd as M
from mode_diffusion_policy_amd import _lib as L
from mode_diffusion_policy_amd import gc_sampling, score_wrappers, utils
from oracle import mode_oracle as O
from oracle.weights import get_config, make_state_dict, param_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# conf/model/mode_agent.yaml:46-76 (every key the Hydra node carries besides _target_)
HYDRA_KEYS = dict(action_dim=7, goal_dim=512, obs_dim=2048, goal_conditioned=True, causal=True, use_custom_attn_mask=False,
                  use_proprio=False, state_dim=8, embed_dim=256, n_layers=2, goal_seq_len=1, obs_seq_len=1, action_seq_len=10,
                  embed_pdrob=0, goal_drop=0.1, attn_pdrop=0.3, mlp_pdrop=0.1, n_heads=8, device="cpu", linear_output=True,
                  cond_router=True, num_experts=4, top_k=2, router_normalize=True, use_goal_in_routing=False, use_argmax=False,
                  use_shared_expert=False, use_noise_token_as_input=True, init_style="olmoe")


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "mode_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mode_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    lib = L.load()                                   # loads libmode_hip.so built by __graft_entry__.build()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mode_hip_version() == L.ABI_VERSION
    assert lib.mode_hip_status_string(-2).decode().startswith("unsupported")
    assert lib.mode_set_option(b"gemm_cfg", 0) == 0 and lib.mode_set_option(b"nope", 1) == -2
    for key, default, bad in ((b"dn_split_k", 0, 9), (b"gemm_skinny_rows", 32, -1), (b"fuse_ln2", 1, None), (b"adamw_blocks", 0, None),
                              (b"combine_row_max", 0x7fffffff, -1), (b"gemm_pp", 1, None), (b"gemm_pp_min_tiles", 200, None), (b"gemm_pp_min_tiles_up", 190, None), (b"gemm_dn_ring3", 1, None),
                              (b"gemm_group_m", 0, None), (b"gemm_tr_cfg", 0, None), (b"gemm_mid_rows", 128, -1),
                              (b"fuse_qkv_attn", 1, None), (b"fuse_qkv_attn_min_b", 56, -1), (b"qkv_attn_w3", 1, None), (b"qkv_attn_waves", 8, 5)):
        assert lib.mode_set_option(key, default) == 0, key                                  # documented knobs exist (include/mode_hip.h)
        if bad is not None:
            assert lib.mode_set_option(key, bad) != 0, key
    for key in (b"pp_flags", b"pp_trace_lo", b"pp_trace_hi", b"attn_bwd_stop", b"gemm_setprio"):
        assert lib.mode_set_option(key, 0) == -2, key                                       # result-changing ablation switches / trace buffers do not ship


def test_integration_doc_abi_matches_header():
    """The ctypes stub printed in INTEGRATION.md asserts the ABI version the header defines (a maintainer pasting it must not hit the assert)."""
    hdr = open(os.path.join(ROOT, "include", "mode_hip.h")).read()
    abi = int(re.search(r"#define\s+MODE_HIP_ABI_VERSION\s+(\d+)", hdr).group(1))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stated = [int(v) for v in re.findall(r"mode_hip_version\(\)\s*==\s*(\d+)", doc)]
    assert stated and all(v == abi for v in stated), (stated, abi)
    assert abi == L.ABI_VERSION


def test_mode_hip_opts_env(monkeypatch):
    """MODE_HIP_OPTS="key=value,..." is applied when the library is loaded; unknown keys fail loudly."""
    lib = L.load()
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setenv("MODE_HIP_OPTS", "fuse_ln2=1,dn_split_k=0")
    assert L.load() is not None
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setenv("MODE_HIP_OPTS", "no_such_knob=1")
    with pytest.raises(ValueError, match="MODE_HIP_OPTS"):
        L.load()
    monkeypatch.setattr(L, "_lib", lib)


def test_ctypes_struct_sizes_match_header_layout():
    """Host-only ABI sanity: every ctypes mirror has the size the loaded library compiled the struct with; layout queries round-trip."""
    import ctypes as C
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "mode_hip.h")).read()
    names = set(re.findall(r"typedef struct (Mode\w+)", hdr))
    assert names and lib.mode_hip_sizeof(b"NoSuchStruct") == 0
    for n in sorted(names):
        assert hasattr(L, n), f"{n} has no ctypes mirror in _lib.py"
        assert lib.mode_hip_sizeof(n.encode()) == C.sizeof(getattr(L, n)), n
    ml = L.ModeMetaLayout()
    assert lib.mode_moe_meta_layout(1792, 4, 2, C.byref(ml)) == 0
    assert ml.perm % 4 == 0 and ml.total_words >= 3 * 3584 + 9
    assert ml.counts < ml.offsets < ml.perm < ml.pos < ml.posw < ml.poffsets < ml.prow < ml.total_words
    assert ml.padded_rows == 3584 + 256
    dims = L.ModeDims(D=1024, H=8, L=12, E=4, k=2, T=14, A_len=10, A_dim=7, O=2048, G=512, n_img=2, use_noise_token=1,
                      router_normalize=1, eps=1e-6)
    nb = lib.mode_dit_workspace_bytes(C.byref(dims), 128, 10, L.MODE_BF16)
    N = 128 * 14
    assert nb >= N * 1024 * 4 + N * 1024 * 2 * 5 + 2 * N * 4096 * 2 + 2 * N * 1024 * 4
    bad = L.ModeDims(D=1024, H=8, L=12, E=4, k=2, T=15, A_len=10, A_dim=7, O=2048, G=512, n_img=2, use_noise_token=1,
                     router_normalize=1, eps=1e-6)
    assert lib.mode_dit_workspace_bytes(C.byref(bad), 128, 10, L.MODE_BF16) == 0        # inconsistent T -> rejected
    assert lib.mode_gemm(None, None) == -1                                                # bad-arg path, no launch


@pytest.mark.parametrize("cfgname", ["tiny", "c1", "c1e4"])
def test_state_dict_keys_and_shapes(cfgname):
    cfg = get_config(cfgname)
    kw = dict(HYDRA_KEYS, obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, embed_dim=cfg.embed_dim, n_layers=cfg.n_layers,
              n_heads=cfg.n_heads, num_experts=cfg.num_experts, top_k=cfg.top_k)
    m = M.MoDeDiT(**kw)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == param_spec(cfg)
    sd = make_state_dict(cfg, 1)
    m.load_state_dict(sd)                                                                  # strict
    names = [n for n, _ in m.named_parameters()]
    assert all(n.startswith("blocks.") for n in names if "experts" in n)                   # mode_agent.py:319-330
    assert all(isinstance(b, M.NoiseBlockMoE) for b in m.blocks)                           # mode_agent.py:470-476
    for n in names:
        assert O.uses_weight_decay(n) == all(x not in n for x in ["bias", "LayerNorm", "embedding"])


def test_default_init_matches_reference_effective_init():
    """pos_emb = 0, RMSNorm g = 1, router N(0, 0.02)/zero bias, everything else torch default."""
    torch.manual_seed(0)
    m = M.MoDeDiT(**HYDRA_KEYS)
    assert float(m.pos_emb.abs().max()) == 0 and bool((m.ln.g == 1).all())
    r = m.blocks[0].router.router.mlp
    assert abs(float(r[0].weight.std()) - 0.02) < 2e-3 and float(r[0].bias.abs().max()) == 0


@pytest.mark.parametrize("flag", ["use_proprio", "use_custom_attn_mask", "use_shared_expert"])
def test_unsupported_flags_raise(flag):
    with pytest.raises(NotImplementedError):
        M.MoDeDiT(**dict(HYDRA_KEYS, **{flag: True}))
    with pytest.raises(NotImplementedError):
        M.MoDeDiT(**dict(HYDRA_KEYS, goal_conditioned=False))


def test_side_channel_api():
    m = M.MoDeDiT(**HYDRA_KEYS)
    for attr in ("load_balancing_loss", "compute_router_z_loss", "precompute_experts_for_inference", "reset_all_caches",
                 "freeze_router", "unfreeze_router", "get_params", "forward"):
        assert callable(getattr(m, attr))
    assert m.logits_per_layer is None and m.probs_per_layer is None
    m.freeze_router()
    assert not any(p.requires_grad for b in m.blocks for p in b.router.parameters())
    assert not m.blocks[0].router.training
    m.unfreeze_router()
    assert all(p.requires_grad for b in m.blocks for p in b.router.parameters())
    b = m.blocks[0]
    b.inference_expert_usage += 1; b.total_tokens_processed = 5; b.reset_expert_usage()
    assert float(b.get_expert_usage().sum()) == 0 and b.total_tokens_processed == 0
    assert len(list(m.get_params())) == len(list(m.parameters()))


def test_gcdenoiser_wrapper_contract():
    den = M.GCDenoiser(dict(HYDRA_KEYS, _target_="mode.models.networks.modedit.MoDeDiT"), sigma_data=0.5)   # config node -> module
    assert isinstance(den.inner_model, M.MoDeDiT) and den.sigma_data == 0.5
    den2 = M.GCDenoiser(den.inner_model, 0.5)
    assert den2.inner_model is den.inner_model
    sig = torch.tensor([0.001, 0.5, 80.0])
    for a, b in zip(den.get_scalings(sig), O.edm_scalings(sig, 0.5)):
        assert torch.allclose(a, b, rtol=1e-6)
    assert len(list(den.get_params())) == len(list(den.inner_model.parameters()))
    assert utils.append_dims(sig, 3).shape == (3, 1, 1)
    with pytest.raises(ValueError):
        utils.append_dims(torch.zeros(2, 2), 1)


def test_sampler_module_contract(golden):
    g = golden("F1_schedule")
    for n in (1, 5, 10):
        assert np.array_equal(gc_sampling.get_sigmas_exponential(n, 1e-3, 80.0).numpy(), g[f"n{n}"])
    assert gc_sampling.np is np and gc_sampling.math is not None and hasattr(gc_sampling, "plt")    # mode_agent.py:16 star-import
    import inspect
    sig = inspect.signature(gc_sampling.sample_ddim)
    assert list(sig.parameters) == ["model", "state", "action", "goal", "sigmas", "scaler", "extra_args", "callback", "disable", "eta"]

    # generic path on an arbitrary callable denoiser (CPU): reference step order, callback dict keys (gc_sampling.py:946-947)
    def toy(state, action, goal, sigma, **kw):
        return 0.5 * action + sigma.reshape(-1, 1, 1)
    x0 = torch.randn(3, 10, 7, generator=torch.Generator().manual_seed(0)) * 80
    sigs = gc_sampling.get_sigmas_exponential(10, 1e-3, 80.0)
    seen = []
    x = gc_sampling.sample_ddim(toy, None, x0, None, sigs, callback=lambda d: seen.append(sorted(d)))
    ref = x0
    for i in range(10):
        ref = O.ddim_update(ref, toy(None, ref, None, sigs[i] * torch.ones(3)), float(sigs[i]), float(sigs[i + 1]))
    assert torch.allclose(x, ref, rtol=1e-5, atol=1e-5)
    assert seen[0] == ["action", "denoised", "i", "sigma", "sigma_hat"] and len(seen) == 10


def test_rand_log_logistic_matches_oracle_stream():
    torch.manual_seed(5)
    a = utils.rand_log_logistic((64,), loc=float(np.log(0.5)), scale=0.5, min_value=1e-3, max_value=80.0)
    torch.manual_seed(5)
    b = O.rand_log_logistic((64,), float(np.log(0.5)), 0.5, 1e-3, 80.0)
    assert torch.equal(a, b) and float(a.min()) >= 1e-3 and float(a.max()) <= 80.0


def test_goal_preprocessing_host_logic():
    m = M.MoDeDiT(**HYDRA_KEYS).eval()
    g2 = torch.randn(4, 512)
    assert m.preprocess_goals(g2, 1).shape == (4, 1, 512)
    assert float(m.preprocess_goals(g2, 1, uncond=True).abs().max()) == 0          # uncond -> zeros (modedit.py:878-879)
    m.train()
    torch.manual_seed(0)
    masked = m.preprocess_goals(torch.ones(64, 1, 512), 1)
    frac = float((masked == 0).float().mean())
    assert 0.07 < frac < 0.13                                                        # element-wise Bernoulli(goal_drop=0.1) (:888)


def test_forward_has_no_cpu_fallback():
    """Product path must fail loudly off-device: no eager / oracle fallback exists."""
    m = M.MoDeDiT(**dict(HYDRA_KEYS, obs_dim=64)).eval()
    with pytest.raises(Exception) as ei:
        m({"state_images": torch.zeros(2, 2, 64)}, torch.zeros(2, 10, 7), torch.zeros(2, 1, 512), torch.ones(2))
    assert "CPU" in str(ei.value) or "ROCm" in str(ei.value) or "HIP" in str(ei.value)
    src = "".join(open(os.path.join(ROOT, "mode_diffusion_policy_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "mode_diffusion_policy_amd")) if f.endswith(".py"))
    assert "oracle" not in src.replace("oracle/", "")                               # product never imports the oracle


def test_arena_layout_regions_and_names():
    """Flat parameter arena: every reference parameter has a slot, slots are 256-byte aligned, the decay / no-decay split is
    MoDEAgent.get_optim_groups' name rule (mode_agent.py:365-384) and adoption keeps values + state_dict order (CPU tensors suffice)."""
    from mode_diffusion_policy_amd.arena import ParamArena, arena_layout
    from mode_diffusion_policy_amd.ddp import optimizer_param_groups
    cfg = get_config("tiny")
    m = M.MoDeDiT(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cpu", goal_conditioned=True, action_dim=7, embed_dim=cfg.embed_dim,
                  embed_pdrob=0, attn_pdrop=0.0, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=1, obs_seq_len=1, action_seq_len=10,
                  mlp_pdrop=0.0, goal_drop=0.0, num_experts=cfg.num_experts, top_k=cfg.top_k)
    sd = make_state_dict(cfg, 3)
    m.load_state_dict(sd)
    layout, bounds = arena_layout(m)
    assert all(off % 64 == 0 for _, _, off in layout) and bounds["decay"] < bounds["no_decay"] < bounds["total"]
    ar = ParamArena(m, torch.device("cpu"))
    assert ar.owns(m) and list(m.state_dict().keys()) == list(sd.keys())
    base = ar.flat.data_ptr()
    groups = optimizer_param_groups(m, 0.05)
    decay_ids = {id(p) for p in groups[0]["params"]}
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), sd[n]), n
        off = (p.data_ptr() - base) // 4
        if n == "gripper_embed.weight":
            assert off >= bounds["no_decay"]
        elif id(p) in decay_ids:
            assert off < bounds["decay"], n
        else:
            assert bounds["decay"] <= off < bounds["no_decay"], n
    total = sum(p.numel() for p in m.parameters())
    assert total <= bounds["total"] < total + 64 * len(layout)
    with torch.no_grad():
        m.out.bias.add_(1.0)
    assert abs(float(ar.w["b_out"][0]) - (float(sd["out.bias"][0]) + 1.0)) < 1e-6            # Parameters ARE arena views
    m.double()
    assert not ar.owns(m)


def test_tri_stage_lr_schedule():
    """TriStageLR == the reference's warm-up / hold / cosine schedule (tri_stage_scheduler.py:100-147) at its stage boundaries."""
    import math
    from mode_diffusion_policy_amd.optim import TriStageLR

    class _Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0}]
    sch = TriStageLR(_Opt(), lr=1e-4, init_lr_scale=0.1, final_lr_scale=0.5, phase_ratio="(0.02, 0.08, 0.9)", total_steps=1000)
    assert (sch.warmup_steps, sch.hold_steps, sch.decay_steps) == (20, 80, 900)
    lrs = [sch.step() for _ in range(1100)]
    assert abs(lrs[0] - 1e-5) < 1e-15 and abs(lrs[10] - (1e-5 + 9e-5 / 20 * 10)) < 1e-15        # linear warm-up from init_lr_scale * lr
    assert lrs[20] == 1e-4 and lrs[99] == 1e-4                                                      # hold
    assert abs(lrs[100] - 1e-4) < 1e-15                                                             # cosine starts at the peak
    assert abs(lrs[100 + 450] - (5e-5 + 0.5 * 5e-5 * (1 + math.cos(0.5 * math.pi)))) < 1e-15
    assert abs(lrs[1000] - 5e-5) < 1e-15 and lrs[1001] == 5e-5 and lrs[-1] == 5e-5                  # floor at final_lr_scale * lr
    assert _Opt.param_groups[0]["lr"] == lrs[-1] == _Opt.param_groups[1]["lr"]


def test_noise_level_densities_reproduce_reference_rng_stream():
    """The sigma densities of MoDEAgent.make_sample_density (mode_agent.py:691-730) against closed-form restatements driven by the same seeded
    torch RNG stream: identical draws in identical order -> bit-identical samples (the reference functions are edm_diffusion/utils.py:154-203)."""
    import math
    shape = (257,)

    def stream(seed):
        torch.manual_seed(seed)

    stream(1); got = utils.rand_log_normal(shape, loc=-1.2, scale=1.2)
    stream(1); assert torch.equal(got, (torch.randn(shape) * 1.2 + -1.2).exp())
    stream(2); got = utils.rand_log_uniform(shape, 1e-3, 80.0)
    stream(2); assert torch.equal(got, (torch.rand(shape) * (math.log(80.0) - math.log(1e-3)) + math.log(1e-3)).exp())
    stream(3); got = utils.rand_uniform(shape, 1e-3, 80.0)
    stream(3); assert torch.equal(got, torch.rand(shape) * (80.0 - 1e-3) + 1e-3)
    stream(4); got = utils.rand_v_diffusion(shape, 0.5, 1e-3, 80.0)
    lo, hi = math.atan(1e-3 / 0.5) * 2 / math.pi, math.atan(80.0 / 0.5) * 2 / math.pi
    stream(4); assert torch.equal(got, torch.tan((torch.rand(shape) * (hi - lo) + lo) * math.pi / 2) * 0.5)
    assert float(got.min()) >= 1e-3 * 0.999 and float(got.max()) <= 80.0 * 1.001
    stream(5); got = utils.rand_split_log_normal(shape, -1.0, 0.5, 1.5)
    stream(5); n = torch.randn(shape).abs(); u = torch.rand(shape)
    assert torch.equal(got, torch.where(u < 0.25, n * -0.5 - 1.0, n * 1.5 - 1.0).exp())
    vals = torch.tensor([0.1, 0.2, 0.3])
    stream(6); got = utils.rand_discrete(shape, vals)
    stream(6); assert torch.equal(got, vals[torch.randint(0, 3, shape)])
    # factory: names, defaults (loglogistic = the shipped default) and the agent's error
    stream(7); a = utils.make_sample_density("loglogistic", sigma_data=0.5, sigma_min=1e-3, sigma_max=80.0)(shape=shape, device="cpu")
    stream(7); b = utils.rand_log_logistic(shape, loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0)
    assert torch.equal(a, b)
    for kind in ("lognormal", "loguniform", "uniform", "v-diffusion"):
        s = utils.make_sample_density(kind)(shape=(16,), device="cpu")
        assert s.shape == (16,) and torch.isfinite(s).all() and float(s.min()) > 0
    assert utils.make_sample_density("split-lognormal", loc=-1.0, scale_1=0.5, scale_2=1.5)(shape=(4,)).shape == (4,)
    with pytest.raises(ValueError, match="Unknown sample density type"):
        utils.make_sample_density("nope")


def test_binding_constants_match_header():
    """The enum / flag values the ctypes binding hard-codes are the ones include/mode_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "mode_hip.h")).read()
    vals = {k: int(v) for k, v in re.findall(r"\b(MODE_[A-Z0-9_]+)\s*=\s*(-?\d+)", hdr)}
    vals.update({k: int(v) for k, v in re.findall(r"#define\s+(MODE_[A-Z0-9_]+)\s+(-?\d+)\b", hdr)})
    assert vals["MODE_BF16"] == L.MODE_BF16 and vals["MODE_F32"] == L.MODE_F32
    for name in ("NONE", "BIAS", "BIAS_GELU", "RESIDUAL", "SWIGLU", "RESIDUAL_NORM"):
        assert vals[f"MODE_EPI_{name}"] == getattr(L, f"EPI_{name}"), name
    for name in ("SKINNY_OK", "W_KN", "A_KM"):
        assert vals[f"MODE_GEMM_{name}"] == getattr(L, f"GEMM_{name}"), name
    assert vals["MODE_HIP_ABI_VERSION"] == L.ABI_VERSION


def test_input_shape_contract_is_checked_before_any_launch():
    """The chain reads raw pointers: a goal of the wrong width (incl. the reference's `goal_dim == 2 * obs_dim` slice, modedit.py:862-880, which makes
    its own goal_emb raise), a batch mismatch or a mis-shaped action chunk must raise, not read out of bounds.  (CPU: the checks run before the
    engine is touched.)"""
    import torch
    import mode_diffusion_policy_amd as M
    m = M.MoDeDiT(obs_dim=32, goal_dim=64, device="cpu", goal_conditioned=True, action_dim=2, embed_dim=64, embed_pdrob=0, attn_pdrop=0.0,
                  n_layers=1, n_heads=2, goal_seq_len=1, obs_seq_len=1, action_seq_len=4, state_dim=None, num_experts=2, top_k=1)
    with pytest.raises(ValueError, match="goals must be"):
        m.preprocess_goals(torch.zeros(3, 1, 64), 1)                 # 64 == 2 * obs_dim: sliced to 32 like the reference, then refused
    m2 = M.MoDeDiT(obs_dim=32, goal_dim=16, device="cpu", goal_conditioned=True, action_dim=2, embed_dim=64, embed_pdrob=0, attn_pdrop=0.0,
                   n_layers=1, n_heads=2, goal_seq_len=1, obs_seq_len=1, action_seq_len=4, state_dim=None, num_experts=2, top_k=1)
    g = m2.preprocess_goals(torch.zeros(3, 16), 1)
    assert g.shape == (3, 1, 16)
    with pytest.raises(ValueError, match="goals must be"):
        m2.preprocess_goals(torch.zeros(3, 1, 20), 1)
    with pytest.raises(ValueError, match="batch mismatch"):
        m2._check_batch(3, torch.zeros(2, 2, 32), g, torch.zeros(3, 4, 2))
    with pytest.raises(ValueError, match="actions must be"):
        m2._check_batch(3, torch.zeros(3, 2, 32), g, torch.zeros(3, 5, 2))
    m2._check_batch(3, torch.zeros(3, 2, 32), g, torch.zeros(3, 4, 2))


def test_pp_kernel_isa_contract(tmp_path):
    """The persistent ping-pong GEMM (85 % of the benchmark's FLOPs) relies on two properties of its COMPILED K loop that no numerics test on
    small shapes would catch reliably: (1) no scratch (spill) access inside a block that issues MFMAs - scratch traffic counts in vmcnt and would
    make the loop's counted `s_waitcnt vmcnt(N)` release LDS tiles that are still being filled; (2) the loop is straight-line code between its
    barriers: one basic block per K-step pair, whose only branch is the back edge (plus the loop-skip test of the peeled first pair).  Checked on
    the gfx950 ISA hipcc produces from the shipped source (cross-compiles without a GPU)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "mode_diffusion_policy_amd", "csrc")
    out = tmp_path / "pp.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{os.path.join(ROOT, 'include')}", f"-I{csrc}", "-S",
                    "--cuda-device-only", os.path.join(csrc, "gemm_bf16_pp.hip"), "-o", str(out)], check=True, capture_output=True)
    asm = out.read_text()
    kernels = [(m.group(1), m.start()) for m in re.finditer(r"^(_ZN4mode14gemm_pp_kernel\w+):", asm, flags=re.M)]
    assert len(kernels) == 10                                        # 224-row tile: {NONE, BIAS, SWIGLU} x {bf16, fp32 out}; 256-row tile: {NONE, BIAS} x {bf16, fp32 out}
    for name, start in kernels:
        body = asm[start: asm.index(".end_amdhsa_kernel", start)]
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)) <= 256
        blocks, cur = [], []
        for line in body.split("\n"):
            if re.match(r"^\.LBB\w+:", line):
                blocks.append(cur); cur = []
            else:
                cur.append(line)
        blocks.append(cur)
        mf = [b for b in blocks if any("v_mfma" in x for x in b)]
        big = "Li4EEEv" in name                                      # the 256-row tile (FM1 = 4): every wave stages both A halves - one role
        assert len(mf) == (2 if big else 4), (name, len(mf))         # wave roles x {peeled first K-step pair, loop body}
        for b in mf:
            assert sum("v_mfma" in x for x in b) == (128 if big else 112)    # 2 K-steps x (32 + 8 FM1) MFMAs per wave
            assert sum("s_barrier" in x for x in b) == 8
            assert not any("scratch_" in x for x in b), name          # (1)
            assert sum("s_cbranch" in x for x in b) <= 2, name        # (2)


def test_pptr_kernel_isa_contract(tmp_path):
    """The backward's persistent ping-pong GEMM (gemm_bf16_pptr.hip) has the same compiled-loop contract as the forward one: counted `s_waitcnt vmcnt(N)`
    release LDS half-tiles, so a scratch (spill) access inside an MFMA block would be a correctness bug; the K loop is straight-line code between its
    barriers (peeled first K-step pair + loop body, 128 MFMAs / 8 barriers each, the only branch is the back edge)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "mode_diffusion_policy_amd", "csrc")
    out = tmp_path / "pptr.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{os.path.join(ROOT, 'include')}", f"-I{csrc}", "-S",
                    "--cuda-device-only", os.path.join(csrc, "gemm_bf16_pptr.hip"), "-o", str(out)], check=True, capture_output=True)
    asm = out.read_text()
    kernels = [(m.group(1), m.start()) for m in re.finditer(r"^(_ZN4mode16gemm_pptr_kernel\w+):", asm, flags=re.M)]
    assert len(kernels) == 4                                         # {data gradient, weight gradient} x {bf16, fp32 out}
    for name, start in kernels:
        body = asm[start: asm.index(".end_amdhsa_kernel", start)]
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)) <= 256
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)) == 0, name      # no scratch at all
        blocks, cur = [], []
        for line in body.split("\n"):
            if re.match(r"^\.LBB\w+:", line):
                blocks.append(cur); cur = []
            else:
                cur.append(line)
        blocks.append(cur)
        mf = [b for b in blocks if any("v_mfma" in x for x in b)]
        assert len(mf) == 2, (name, len(mf))
        for b in mf:
            assert sum("v_mfma" in x for x in b) == 128
            assert sum("s_barrier" in x for x in b) == 8
            assert not any("scratch_" in x for x in b), name
            assert sum("s_cbranch" in x for x in b) <= 2, name
            assert sum("ds_read_b64_tr_b16" in x for x in b) in (32, 96), name        # per K-step pair: 32 for W, + 64 for A in the weight gradient


def test_training_side_api_surface():
    """Host-side contract of the round-6 training / loader additions (no GPU): the optimizer exposes the calls INTEGRATION.md section 3 names, the
    EMA's schedule is the callback's (mode/callbacks/ema.py:101-126) and a sharded EMA refuses to be read, the checkpoint readers take the explicit
    trust switch and never unpickle without it."""
    import inspect
    import pickle
    from mode_diffusion_policy_amd import optim, rollout
    for name in ("finish_fused_step", "set_fuse_expert_step", "reset_state", "gather_state", "fused_grad_sq"):
        assert callable(getattr(optim.FusedAdamW, name)), name
    assert inspect.signature(optim.FusedAdamW.step).parameters["grad_scale"].default is None      # fused mode: the backward's scale unless the caller insists
    for fn in (rollout.load_denoiser_checkpoint, rollout.load_agent_checkpoint, rollout._read_checkpoint_file):
        assert inspect.signature(fn).parameters["trust_pickle"].default is False, fn.__name__

    class _M:                                                                   # ArenaEMA only keeps the reference until it is used
        pass
    ema = optim.ArenaEMA(_M(), decay=0.999)
    # warm-up schedule of the callback: decay_t = 1 - (1 + t')^-2/3 with t' = max(0, t - start - 1), clamped to [min_value, max_value]
    assert ema.get_decay(1) == 0.0 and abs(ema.get_decay(2) - (1 - 2 ** (-2 / 3))) < 1e-12 and ema.get_decay(10 ** 9) == 0.9999
    assert ema.should_apply(1) and not optim.ArenaEMA(_M(), start_step=5).should_apply(3)
    ema._sharded = True
    for call in (lambda: ema.swap(), lambda: ema.update(3)):
        with pytest.raises(RuntimeError, match="gather_state"):
            call()
    # a file that needs the full unpickler is refused without the switch (and says how to allow it)
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "x.ckpt")
        with open(path, "wb") as f:
            pickle.dump({"state_dict": {}, "obj": inspect.Signature()}, f)
        os.environ.pop("MODE_TRUST_CKPT", None)
        with pytest.raises(RuntimeError, match="trust_pickle=True"):
            rollout._read_checkpoint_file(path)
