"""Inference harness (mode_diffusion_policy_amd/rollout.py): schedule / sampler dispatch with the agent's names and errors (mode_agent.py:779-861),
checkpoint loading by key, and the batched action-chunking policy (mode_agent.py:584-637)."""
import os

import numpy as np
import pytest
import torch

import mode_diffusion_policy_amd as M
from mode_diffusion_policy_amd import gc_sampling, rollout
from oracle.weights import get_config, make_inputs, make_state_dict


def _model(cfg, device, dtype="fp32"):
    return M.MoDeDiT(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device=device, goal_conditioned=True, action_dim=7, embed_dim=cfg.embed_dim,
                     embed_pdrob=0, attn_pdrop=0.3, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=1, obs_seq_len=1, action_seq_len=10,
                     num_experts=cfg.num_experts, top_k=cfg.top_k, compute_dtype=dtype)


def test_dispatch_names_and_errors(golden):
    g = golden("F10_schedules")
    for name in ("karras", "linear", "vp", "cosine_beta", "iddpm"):
        got = rollout.get_noise_schedule(10, name, 1e-3, 80.0)
        np.testing.assert_allclose(got.numpy(), g[f"{name}_n10"], rtol=2e-6, atol=1e-9)
    assert torch.equal(rollout.get_noise_schedule(10, "exponential", 1e-3, 80.0), gc_sampling.get_sigmas_exponential(10, 1e-3, 80.0))
    with pytest.raises(ValueError, match="Unknown noise schedule type"):
        rollout.get_noise_schedule(10, "nope", 1e-3, 80.0)

    def toy(state, action, goal, sigma, **kw):
        return 0.5 * action
    x0 = torch.randn(2, 10, 7, generator=torch.Generator().manual_seed(0))
    sig = gc_sampling.get_sigmas_exponential(5, 1e-3, 80.0)
    for name in ("lms", "heun", "euler", "ancestral", "euler_ancestral", "dpm", "dpmpp_2s_ancestral", "dpmpp_2m", "ddim", "dpmpp_2s", "dpm_fast",
                 "debugging", "dpmpp_2_with_lms"):
        out = rollout.sample_loop(toy, sig, x0, None, None, name)
        assert out.shape == x0.shape and torch.isfinite(out).all(), name
    with pytest.raises(ValueError, match="desired sampler type not found"):
        rollout.sample_loop(toy, sig, x0, None, None, "nope")
    with pytest.raises(NotImplementedError):                       # fails on every call in the reference as well (gc_sampling.py:630)
        rollout.sample_loop(toy, sig, x0, None, None, "dpm_adaptive")
    try:
        import torchsde  # noqa: F401
        assert torch.isfinite(rollout.sample_loop(toy, sig, x0, None, None, "dpmpp_2m_sde")).all()
    except ImportError:
        with pytest.raises(ImportError, match="torchsde"):
            rollout.sample_loop(toy, sig, x0, None, None, "dpmpp_2m_sde")


def test_checkpoint_loader_by_key(tmp_path):
    from safetensors.torch import save_file
    cfg = get_config("tiny")
    sd = make_state_dict(cfg, 5)
    ck = {"model.inner_model." + k: v.clone() for k, v in sd.items()}
    ck["model.inner_model.sigma_emb.weight"] = ck["model.inner_model.sigma_emb.weight"].reshape(-1)     # same numel, other shape: reshaped on load
    ck["static_resnet.resnet.conv1.weight"] = torch.zeros(4, 3, 3, 3)                                  # encoder tensors: not ours
    ck["clip_model.visual.proj"] = torch.zeros(4, 4)
    path = os.path.join(tmp_path, "agent.safetensors")
    save_file(ck, path)
    m = _model(cfg, "cpu")
    missing, unexpected, skipped = rollout.load_denoiser_checkpoint(m, path)
    assert not missing and not unexpected and not skipped
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # a plain mapping without the prefix, one tensor absent, one incompatible
    part = {k: v for k, v in sd.items() if k != "out.bias"}
    part["pos_emb"] = torch.zeros(3)
    m2 = _model(cfg, "cpu")
    missing, unexpected, skipped = rollout.load_denoiser_checkpoint(m2, part)
    assert "out.bias" in missing and "pos_emb" in missing and skipped == ["pos_emb"] and not unexpected


def test_agent_checkpoint_loader_reference_key_names(tmp_path):
    """One agent checkpoint written under the key names of the reference's releases (mode_agent.py:209-251): the denoiser under
    ``model.inner_model.``, the static camera under the old ``img_encoder_image_primary.`` prefix, the gripper camera partly under
    ``img_encoder_image_wrist.`` and partly under ``net.`` (-> ``gripper_resnet.resnet.``), CLIP tensors to be ignored, a flattened convolution
    weight, a tiled BatchNorm vector, a 0-d entry, an incompatible tensor.  Every tensor must land where the reference's loader puts it."""
    from safetensors.torch import save_file
    from mode_diffusion_policy_amd import perceptual_encoders as E
    from oracle import resnet_oracle as R
    cfg = get_config("tiny")
    sd = make_state_dict(cfg, 7)
    den = M.GCDenoiser(_model(cfg, "cpu"), 0.5)
    stat, grip = E.FiLMResNet18Policy(32), E.FiLMResNet18Policy(32)
    s_sd = R.fill_encoder_state_dict(stat.state_dict(), 11)
    g_sd = R.fill_encoder_state_dict(grip.state_dict(), 12)
    ck = {"model.inner_model." + k: v.clone() for k, v in sd.items()}
    ck.update({"img_encoder_image_primary." + k: v.clone() for k, v in s_sd.items()})
    for k, v in g_sd.items():
        if k.startswith("resnet.layer1."):
            ck["net." + k[len("resnet."):]] = v.clone()
        else:
            ck["img_encoder_image_wrist." + k] = v.clone()
    ck["clip_model.visual.proj"] = torch.zeros(4, 4)
    ck["model.visual_stub"] = torch.zeros(2)
    ck["img_encoder_image_primary.resnet.conv1.weight"] = s_sd["resnet.conv1.weight"].reshape(-1).clone()          # conv weight stored 1-d
    ck["img_encoder_image_wrist.resnet.layer2.0.conv1.weight"] = g_sd["resnet.layer2.0.conv1.weight"].reshape(128, -1).clone()   # ... stored 2-d
    half = g_sd["resnet.bn1.weight"][:32].clone()
    ck["img_encoder_image_wrist.resnet.bn1.weight"] = half                                                            # shorter vector: tiled
    ck["img_encoder_image_wrist.resnet.bn1.num_batches_tracked"] = torch.tensor(5)
    ck["img_encoder_image_primary.resnet.bn1.bias"] = torch.tensor(0.0)                                                # 0-d: zeros of the target shape
    ck["img_encoder_image_primary.resnet.layer1.0.conv1.weight"] = torch.zeros(3, 5, 7)                                # incompatible: skipped
    path = os.path.join(tmp_path, "model_cleaned.safetensors")
    save_file({k: v.contiguous() for k, v in ck.items()}, path)
    before = stat.state_dict()["resnet.layer1.0.conv1.weight"].clone()
    res = rollout.load_agent_checkpoint({"model": den, "static_resnet": stat, "gripper_resnet": grip}, path)
    assert res["skipped"] == ["static_resnet.resnet.layer1.0.conv1.weight"] and res["missing"] == res["skipped"] and not res["unexpected"]
    assert res["reshaped"] == 4
    for k, v in den.inner_model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    for k, v in stat.state_dict().items():
        want = s_sd[k]
        if k == "resnet.bn1.bias":
            want = torch.zeros_like(want)
        if k == "resnet.layer1.0.conv1.weight":
            want = before
        assert torch.equal(v, want), k
    for k, v in grip.state_dict().items():
        want = g_sd[k]
        if k == "resnet.bn1.weight":
            want = half.repeat(2)
        if k == "resnet.bn1.num_batches_tracked":
            want = torch.tensor(5)
        assert torch.equal(v, want), k
    # strict=True surfaces what is absent (the reference passes `strict` through to load_state_dict)
    with pytest.raises(RuntimeError, match="Missing key"):
        rollout.load_agent_checkpoint({"model": den, "static_resnet": stat}, {"model.inner_model.pos_emb": sd["pos_emb"]}, strict=True)
    # an object with the agent's attribute names works as well as the mapping
    class _Agent:
        pass
    a = _Agent(); a.model, a.static_resnet, a.gripper_resnet = den, stat, grip
    assert rollout.load_agent_checkpoint(a, path)["direct"] == res["direct"]
    # the reference also takes the checkpoint DIRECTORY (model_cleaned.safetensors inside, mode_agent.py:143-155) ...
    assert rollout.load_agent_checkpoint(a, str(tmp_path))["direct"] == res["direct"]
    assert rollout.load_agent_checkpoint(a, tmp_path)["direct"] == res["direct"]                   # os.PathLike
    # ... a directory with model_cleaned.pt only, and an empty one (its FileNotFoundError)
    d2 = tmp_path / "pt_only"; d2.mkdir()
    torch.save({"state_dict": {k: v.contiguous() for k, v in ck.items()}}, d2 / "model_cleaned.pt")
    assert rollout.load_agent_checkpoint(a, str(d2))["direct"] == res["direct"]
    d3 = tmp_path / "empty"; d3.mkdir()
    with pytest.raises(FileNotFoundError, match="No cleaned weights"):
        rollout.load_agent_checkpoint(a, str(d3))
    # a Lightning-style .ckpt carries non-tensor objects next to the weights: torch >= 2.6 refuses it under weights_only=True, the loader retries
    import argparse
    lightning = {"state_dict": {k: v.contiguous() for k, v in ck.items()}, "hyper_parameters": argparse.Namespace(lr=1e-4), "epoch": 3}
    torch.save(lightning, tmp_path / "last.ckpt")
    # ... but only when the caller says the file is trusted (full unpickling executes what the file says)
    with pytest.raises(RuntimeError, match="trust_pickle=True"):
        rollout.load_agent_checkpoint(a, str(tmp_path / "last.ckpt"))
    with pytest.warns(UserWarning, match="full unpickler"):
        assert rollout.load_agent_checkpoint(a, str(tmp_path / "last.ckpt"), trust_pickle=True)["direct"] == res["direct"]
    (tmp_path / "garbage.ckpt").write_bytes(b"not a checkpoint")
    with pytest.raises(RuntimeError, match="cannot read checkpoint"):
        rollout.load_agent_checkpoint(a, str(tmp_path / "garbage.ckpt"))
    with pytest.raises(RuntimeError, match="cannot read checkpoint"):
        rollout.load_agent_checkpoint(a, str(tmp_path / "garbage.ckpt"), trust_pickle=True)


@pytest.mark.gpu
def test_chunked_rollout_policy_batch_of_envs():
    cfg = get_config("c1e4")
    sd = make_state_dict(cfg, 210)
    m = _model(cfg, "cuda", "bf16")
    m.load_state_dict(sd)
    den = M.GCDenoiser(m.cuda().eval(), 0.5).eval()
    B = 6
    inp = {k: v.cuda() for k, v in make_inputs(cfg, B, 3).items()}
    obs = {"state_images": inp["state_images"]}
    pol = rollout.ChunkedRolloutPolicy(den, num_sampling_steps=10, multistep=4, act_window_size=10,
                                       generator=torch.Generator(device="cuda").manual_seed(7))
    acts = [pol.step(obs, inp["goals"].squeeze(1)).clone() for _ in range(9)]            # replans at control steps 0, 4, 8
    assert all(a.shape == (B, 7) and torch.isfinite(a).all() for a in acts)
    assert pol.rollout_step_counter == 1 and not pol.need_precompute_experts_for_inference
    # the emitted actions are consecutive rows of the planned chunk; the plan is what sample_ddim produces from the same noise
    ref = rollout.ChunkedRolloutPolicy(den, multistep=4, generator=torch.Generator(device="cuda").manual_seed(7))
    plan0 = ref.denoise_actions(obs, inp["goals"])
    for t in range(4):
        assert torch.equal(acts[t], plan0[:, t])
    plan1 = ref.denoise_actions(obs, inp["goals"])
    assert torch.equal(acts[4], plan1[:, 0]) and not torch.equal(plan1, plan0)             # fresh noise per replanning call
    sig = rollout.get_noise_schedule(10, "exponential", 1e-3, 80.0, "cuda")
    x0 = torch.randn((B, 10, 7), device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)) * 80.0
    assert torch.equal(plan0, M.sample_ddim(den, obs, x0, inp["goals"], sig, disable=True))
    # environments are independent: the plan of env 2 does not depend on who else is in the batch
    solo = rollout.ChunkedRolloutPolicy(den, multistep=4)
    sub = {"state_images": inp["state_images"][2:3]}
    x_solo = M.sample_ddim(den, sub, x0[2:3], inp["goals"][2:3], sig, disable=True)
    assert float((x_solo - plan0[2:3]).abs().max()) < 2e-2 * float(plan0.abs().max())
    pol.reset()
    assert pol.rollout_step_counter == 0 and pol.pred_action_seq is None and solo.rollout_step_counter == 0
    # another sampler through the same policy
    pol2 = rollout.ChunkedRolloutPolicy(den, sampler_type="dpmpp_2m", noise_scheduler="karras", multistep=10)
    a = pol2.step(obs, inp["goals"])
    assert a.shape == (B, 7) and torch.isfinite(a).all()


@pytest.mark.gpu
def test_policy_with_graphed_perceptual_encoders():
    """The rollout as the agent runs it (mode_agent.py:584-637): raw camera frames -> two FiLM-ResNets -> 10-step DDIM chunk.  The encoders go
    through GraphedVisualEncoder (one hipGraph replay): same tokens as the eager ``embed_visual_obs`` (the same launches),
    the plan matches the plan from the eager tokens, in-place weight changes are seen, training mode falls back to the eager path."""
    from oracle import resnet_oracle as R
    cfg = get_config("c1e4")
    m = _model(cfg, "cuda", "bf16")
    m.load_state_dict(make_state_dict(cfg, 210))
    den = M.GCDenoiser(m.cuda().eval(), 0.5).eval()
    encs = []
    for seed in (1, 2):
        e = M.FiLMResNet18Policy(cfg.goal_dim).cuda().eval()
        e.load_state_dict({k: v.cuda() for k, v in R.fill_encoder_state_dict(e.state_dict(), seed).items()})
        encs.append(e)
    B = 3
    g = torch.Generator(device="cuda").manual_seed(5)
    frames = lambda: {"rgb_obs": {"rgb_static": torch.randn(B, 1, 3, 64, 64, device="cuda", generator=g), "rgb_gripper": torch.randn(B, 1, 3, 64, 64, device="cuda", generator=g)}}
    goal = torch.randn(B, cfg.goal_dim, device="cuda", generator=g)
    if encs[0].resnet.num_features != cfg.obs_dim:
        pytest.skip("fixture geometry: encoder width != obs_dim")
    pol = rollout.ChunkedRolloutPolicy(den, multistep=2, static_resnet=encs[0], gripper_resnet=encs[1], encoder_autocast=None,
                                       generator=torch.Generator(device="cuda").manual_seed(9))
    ref = rollout.ChunkedRolloutPolicy(den, multistep=2, generator=torch.Generator(device="cuda").manual_seed(9))
    for it in range(3):
        obs = frames()
        with torch.no_grad():
            tok = M.embed_visual_obs(encs[0], encs[1], obs["rgb_obs"]["rgb_static"], obs["rgb_obs"]["rgb_gripper"], goal)
        got = pol.embed(obs, goal)["state_images"]
        assert torch.equal(got, tok["state_images"].float()) or float((got - tok["state_images"]).abs().max()) < 1e-5 * float(tok["state_images"].abs().max())
        a = [pol.step(obs, goal).clone() for _ in range(2)]
        b = [ref.step(tok, goal).clone() for _ in range(2)]
        # (MIOpen's split-K convolutions add with atomics: two runs of the SAME encoder launches differ in the last fp32 bits, ~3e-7 of the token scale,
        # which the bf16 denoiser turns into a few 1e-3 of the plan - so the plans are compared at the bf16 output tolerance, the tokens at 1e-5)
        for x, y in zip(a, b):
            assert float((x - y).norm() / y.norm()) < 2e-2, it
        if it == 1:                                                          # in-place weight change: the graph reads the parameters where they live
            with torch.no_grad():
                encs[0].resnet.conv1.weight.mul_(1.05); encs[1].film4.gamma.weight.add_(0.01)
    assert len(pol.encoders._graphs) == 1
    encs[0].train()
    out = pol.encoders(obs["rgb_obs"]["rgb_static"], obs["rgb_obs"]["rgb_gripper"], goal)["state_images"]     # batch statistics: eager path
    assert out.shape == (B, 2, encs[0].resnet.num_features) and torch.isfinite(out).all()
    encs[0].eval()
    # (round-3 advisor finding) an INTERIOR tensor re-allocated behind the graph's back - a BatchNorm weight in the middle of the trunk replaced through
    # ``.data`` - must re-capture instead of replaying against the freed storage; a second input dtype gets its own weight copies and graph, and the
    # first dtype's graph still replays correctly afterwards
    obs = frames()
    with torch.no_grad():
        bn = encs[1].resnet.layer2[0].bn1
        bn.weight.data = bn.weight.data.clone() * 1.5
        tok = M.embed_visual_obs(encs[0], encs[1], obs["rgb_obs"]["rgb_static"], obs["rgb_obs"]["rgb_gripper"], goal)["state_images"]
    got = pol.encoders(obs["rgb_obs"]["rgb_static"], obs["rgb_obs"]["rgb_gripper"], goal)["state_images"]
    assert float((got - tok).abs().max()) < 1e-5 * float(tok.abs().max()) and len(pol.encoders._graphs) == 2
    half = pol.encoders(obs["rgb_obs"]["rgb_static"].bfloat16(), obs["rgb_obs"]["rgb_gripper"].bfloat16(), goal)["state_images"]
    assert float((half.float() - tok).norm() / tok.norm()) < 5e-2 and len(pol.encoders._graphs) == 3
    again = pol.encoders(obs["rgb_obs"]["rgb_static"], obs["rgb_obs"]["rgb_gripper"], goal)["state_images"]
    assert float((again - tok).abs().max()) < 1e-5 * float(tok.abs().max()) and len(pol.encoders._graphs) == 3
    # folded-BatchNorm inference path (one launch per BatchNorm) is taken under no_grad even though the BatchNorm weights require grad
    assert all(p.requires_grad for p in encs[0].parameters())
    with pytest.raises(ValueError):
        ref.step(frames(), goal)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_whole_sampler_call_as_one_graph(dtype, monkeypatch):
    """Non-DDIM samplers through the policy: the whole call is one hipGraph replay.  Deterministic samplers must reproduce the step-by-step path
    (MODE_HIP_GRAPH=0: the reference-shaped host loop over the same kernels); the ancestral ones draw their noise inside the graph - finite, fresh per
    replay, and the usage counters advance like on the step-by-step path."""
    cfg = get_config("c1e4")
    m = _model(cfg, "cuda", dtype)
    m.load_state_dict(make_state_dict(cfg, 210))
    den = M.GCDenoiser(m.cuda().eval(), 0.5).eval()
    B = 5
    inp = {k: v.cuda() for k, v in make_inputs(cfg, B, 3).items()}
    obs = {"state_images": inp["state_images"]}
    tol = 1e-5 if dtype == "fp32" else 2e-2
    for name in rollout._GRAPHABLE_SAMPLERS:
        mk = lambda: rollout.ChunkedRolloutPolicy(den, sampler_type=name, noise_scheduler="karras", multistep=10, generator=torch.Generator(device="cuda").manual_seed(11))
        pol, ref = mk(), mk()
        plans = []
        for it in range(3):
            o = {"state_images": inp["state_images"] * (1 + 0.1 * it)}
            tokens0 = [int(b.total_tokens_processed) for b in m.blocks]
            p = pol.denoise_actions(o, inp["goals"])
            calls = (int(m.blocks[0].total_tokens_processed) - tokens0[0]) // (B * m.seq_len)
            monkeypatch.setenv("MODE_HIP_GRAPH", "0")
            tokens1 = int(m.blocks[0].total_tokens_processed)
            r = ref.denoise_actions(o, inp["goals"])
            calls_ref = (int(m.blocks[0].total_tokens_processed) - tokens1) // (B * m.seq_len)
            monkeypatch.delenv("MODE_HIP_GRAPH")
            assert p.shape == r.shape == (B, 10, 7) and torch.isfinite(p).all()
            assert calls == calls_ref and calls >= 10, (name, calls, calls_ref)           # same number of denoiser calls accounted
            if "ancestral" not in name:
                assert float((p - r).norm() / r.norm()) < tol, (name, it, float((p - r).norm() / r.norm()))
            plans.append(p)
        assert len(pol._chunk_graphs) == 1                                            # one capture, three replays
        assert not torch.equal(plans[0], plans[1])
    # euler (no churn) and ddim are one update: through the policy both take the fused DDIM graph - the same plan from the same noise
    plan = {}
    for name in ("euler", "ddim"):
        pol = rollout.ChunkedRolloutPolicy(den, sampler_type=name, noise_scheduler="karras", multistep=10, generator=torch.Generator(device="cuda").manual_seed(11))
        plan[name] = pol.denoise_actions(obs, inp["goals"])
        assert not getattr(pol, "_chunk_graphs", None)
    assert torch.equal(plan["euler"], plan["ddim"])
    # dpmpp_2m (two-point extrapolation in the head kernel) and the two-stage solvers (both stages' linear updates in the head kernel): the fused chain,
    # no policy-level capture, the same number of denoiser calls accounted, equal to the step-by-step path
    for name, want_calls in (("dpmpp_2m", 10), ("heun", 19), ("dpm", 19), ("dpmpp_2s", 19)):
        pol = rollout.ChunkedRolloutPolicy(den, sampler_type=name, noise_scheduler="karras", multistep=10, generator=torch.Generator(device="cuda").manual_seed(11))
        ref = rollout.ChunkedRolloutPolicy(den, sampler_type=name, noise_scheduler="karras", multistep=10, generator=torch.Generator(device="cuda").manual_seed(11))
        for it in range(2):
            o = {"state_images": inp["state_images"] * (1 + 0.1 * it)}
            tokens0 = int(m.blocks[0].total_tokens_processed)
            p = pol.denoise_actions(o, inp["goals"])
            calls = (int(m.blocks[0].total_tokens_processed) - tokens0) // (B * m.seq_len)
            monkeypatch.setenv("MODE_HIP_GRAPH", "0")
            r = ref.denoise_actions(o, inp["goals"])
            monkeypatch.delenv("MODE_HIP_GRAPH")
            assert not getattr(pol, "_chunk_graphs", None) and calls == want_calls, (name, calls)
            assert float((p - r).norm() / r.norm()) < tol, (name, it, float((p - r).norm() / r.norm()))
        assert not torch.equal(p, plan["ddim"])
