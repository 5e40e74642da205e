"""CPU: the oracle (oracle/mode_oracle.py) against the golden vectors generated from the REAL reference
(oracle/gen_golden.py).  fp32 tolerance 1e-5 rel-L2; router integers bit-exact."""
import math

import numpy as np
import pytest
import torch

from oracle import mode_oracle as O
from oracle.weights import get_config, make_inputs, make_state_dict, param_spec

TOL = 1e-5


def rel(a, b):
    a = torch.as_tensor(a, dtype=torch.float64); b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_schedule(golden):
    g = golden("F1_schedule")
    for n in (1, 5, 10):
        np.testing.assert_array_equal(O.get_sigmas_exponential(n, 1e-3, 80.0).numpy(), g[f"n{n}"])
    s = g["n10"]
    assert s[0] == pytest.approx(80.0) and s[-2] == pytest.approx(1e-3, rel=1e-5) and s[-1] == 0.0


@pytest.mark.parametrize("name", ["F2_blocks_tiny", "F3_c1_forward_uniform", "F3_c1_forward_persample",
                                  "F3_c1e4_forward_persample", "F7_c2block"])
def test_forward_fixture(golden, name):
    g = golden(name)
    cfg = get_config(str(g["cfg"])); B = int(g["B"]); seed = int(g["seed"])
    sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, B, seed + 1)
    out, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], torch.from_numpy(g["sigma"]),
                             return_aux=True)
    assert float(g["margin"]) > 1e-5, "fixture too close to a top-k tie"
    assert np.array_equal(torch.stack(aux.topk_idx).numpy(), g["topk_idx"])          # bit-exact integers
    assert np.array_equal(torch.stack(aux.perm).numpy(), g["perm"])
    assert np.array_equal(torch.stack(aux.counts).numpy(), g["counts"])
    assert rel(out, g["out"]) < TOL
    # combine weights == reference router_probs at the chosen experts
    rp = torch.from_numpy(g["router_probs"])                                          # (L,B,T,E)
    w = torch.stack(aux.combine_w)                                                    # (L,B,T,k)
    assert rel(w, rp.gather(-1, torch.from_numpy(g["topk_idx"]))) < TOL
    if "block_out" in g.files:
        assert rel(torch.stack(aux.block_out), g["block_out"]) < TOL


@pytest.mark.parametrize("tag", ["nonoise", "goalroute", "nonorm", "b1"])
def test_flag_variant_fixture(golden, tag):
    """Constructor-flag variants the reference can run: T=13, goal-conditioned routing, raw router weights, B=1."""
    import dataclasses
    g = golden(f"F9_{tag}")
    over = {k: bool(g[k]) for k in ("use_noise_token_as_input", "use_goal_in_routing", "router_normalize") if k in g.files}
    cfg = dataclasses.replace(get_config("c1e4"), **over)
    B = int(g["B"]); sd = make_state_dict(cfg, int(g["seed"])); inp = make_inputs(cfg, B, int(g["seed"]) + 1)
    out, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], torch.from_numpy(g["sigma"]), return_aux=True)
    assert np.array_equal(torch.stack(aux.topk_idx).numpy(), g["topk_idx"])
    assert rel(out, g["out"]) < TOL


@pytest.mark.parametrize("cfgname", ["c1", "c1e4"])
def test_ddim_fixture(golden, cfgname):
    g = golden(f"F4_{cfgname}_ddim")
    cfg = get_config(cfgname); seed = int(g["seed"])
    sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, 8, seed + 1)
    sig = torch.from_numpy(g["sigmas"])
    x, xs = O.sample_ddim(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sig, trace=True)
    assert rel(x, g["x_final"]) < TOL
    # last step has sigma_next = 0 => r = 0 => x == denoised
    assert rel(xs[-1], g["denoised"][-1]) < TOL
    # per-step router indices
    xx = inp["x0"]
    for i in range(10):
        den, aux = O.denoiser_forward(sd, cfg, 0.5, inp["state_images"], xx, inp["goals"], sig[i] * torch.ones(8),
                                      return_aux=True)
        assert np.array_equal(torch.stack(aux.topk_idx).numpy(), g["topk_idx"][i])
        assert rel(den, g["denoised"][i]) < TOL
        xx = O.ddim_update(xx, den, float(sig[i]), float(sig[i + 1]))


@pytest.mark.parametrize("cfgname", ["c1", "c1e4"])
def test_loss_grad_fixture(golden, cfgname):
    g = golden(f"F5_{cfgname}_loss_grad")
    cfg = get_config(cfgname); seed = int(g["seed"])
    sd = {k: v.requires_grad_(True) for k, v in make_state_dict(cfg, seed).items()}
    inp = make_inputs(cfg, 8, seed + 1)
    sig = torch.from_numpy(g["sigma"])
    c_skip, c_out, c_in = (t.reshape(-1, 1, 1) for t in O.edm_scalings(sig, 0.5))
    noised = inp["actions"] + inp["noise"] * sig.reshape(-1, 1, 1)
    F, aux = O.dit_forward(sd, cfg, inp["state_images"], noised * c_in, inp["goals"], sig, return_aux=True)
    loss = (F - (inp["actions"] - c_skip * noised) / c_out).pow(2).flatten(1).mean()
    assert abs(float(loss) - float(g["loss"])) / float(g["loss"]) < TOL
    assert rel(F.detach(), g["F"]) < TOL
    # aux losses (modedit.py:898-969)
    T, E = cfg.seq_len, cfg.num_experts
    lb = sum(O.load_balancing_term(None, aux.topk_idx[l].reshape(-1, cfg.top_k), aux.combine_w[l].reshape(-1, cfg.top_k), T, E)
             for l in range(cfg.n_layers)) / cfg.n_layers
    assert abs(float(lb) - float(g["lb"])) < 1e-5 * max(1.0, abs(float(g["lb"])))
    z = O.router_z_loss(aux.shifted_logits, T)
    assert abs(float(z) - float(g["z"])) < 1e-5 * max(1.0, abs(float(g["z"])))
    loss.backward()
    none = set(g["none"].tolist())
    assert "gripper_embed.weight" in none                       # dead parameter
    for k, v in sd.items():
        if k in none:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    gn = dict(zip(g["gn_keys"].tolist(), g["gn_vals"].tolist()))
    for k, ref_norm in gn.items():
        if ref_norm <= 1e-6:                                     # numerically-zero router grads at k=1 (see gen_golden)
            continue
        assert abs(float(sd[k].grad.norm()) - ref_norm) / ref_norm < 1e-4, k
    for key in g.files:
        if key.startswith("g:") and gn[key[2:]] > 1e-6:
            assert rel(sd[key[2:]].grad, g[key]) < 1e-4, key
        if key.startswith("gs:") and gn[key[3:]] > 1e-6:
            got = sd[key[3:]].grad.reshape(-1)[:2048]
            assert float((got - torch.from_numpy(g[key])).norm()) < 1e-4 * gn[key[3:]], key


def test_fused_cache_fixture(golden):
    """Reference's per-sigma fused-expert cache (modedit.py:607-633): cached experts/probs == oracle routing on R=1 row."""
    g = golden("F6_fused_cache")
    cfg = get_config("c1e4"); sd = make_state_dict(cfg, int(g["seed"]))
    for si, s in enumerate(g["sigmas"]):
        cond = O.sigma_embedding(sd, torch.tensor([s]))
        for l in range(cfg.n_layers):
            _, probs = O.router_probs(sd, l, cond)
            idx, w = O.topk_route(probs, 2, True)
            assert idx[0].tolist() == g["idx"][si, l].tolist()
            assert np.allclose(w[0].numpy(), g["p"][si, l], rtol=1e-5)
    for B in (1, 8):
        inp = make_inputs(cfg, B, 777)
        out = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], torch.tensor(g["sigmas"][2]) * torch.ones(B))
        assert rel(out, g[f"loop_B{B}"]) < TOL


def test_optimizer_groups(golden):
    g = golden("F8_optimizer_groups")
    for n, d in zip(g["names"].tolist(), g["decay"].tolist()):
        assert O.uses_weight_decay(n) == bool(d), n
    # RMSNorm gains, pos_emb and *_emb.weight ARE decayed
    assert O.uses_weight_decay("blocks.0.ln_1.g") and O.uses_weight_decay("pos_emb") and O.uses_weight_decay("tok_emb.weight")
    assert not O.uses_weight_decay("out.bias")


def test_param_spec_counts():
    n = lambda c: sum(int(np.prod(s)) for _, s in param_spec(get_config(c)))
    assert n("c2") == 685_850_631 or abs(n("c2") - 685.85e6) < 0.01e6     # 685.85 M
    assert abs(n("c1") - 4.41e6) < 0.01e6


def test_ddim_update_equals_reference_form():
    """r*x+(1-r)*den == (sigma_fn(t_next)/sigma_fn(t))*x - expm1(-h)*den   (gc_sampling.py:948-950)."""
    x = torch.randn(4, 10, 7, dtype=torch.float64); d = torch.randn(4, 10, 7, dtype=torch.float64)
    s, sn = torch.tensor(6.509, dtype=torch.float64), torch.tensor(1.857, dtype=torch.float64)
    t, tn = -s.log(), -sn.log()
    ref = ((-tn).exp() / (-t).exp()) * x - (-(tn - t)).expm1() * d
    assert torch.allclose(O.ddim_update(x, d, float(s), float(sn)), ref, rtol=1e-12)


def test_oracle_aux_loss_composition_vs_reference(golden):
    """F12: act_loss + 0.01 * load_balancing_loss + 0.001 * router_z_loss composed on the real reference modules (mode_agent.py:399-419) and
    differentiated by autograd: the oracle's restatement reproduces the losses and every gradient."""
    g = golden("F12_c1e4_aux_loss_grad")
    cfg = get_config(str(g["cfg"])); seed, B = int(g["seed"]), int(g["B"])
    sd = {k: v.clone().requires_grad_(True) for k, v in make_state_dict(cfg, seed).items()}
    inp = make_inputs(cfg, B, seed + 1)
    tot, act, lb, z = O.training_total_loss(sd, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], inp["noise"], torch.from_numpy(g["sigma"]),
                                            float(g["gamma"]), float(g["delta"]))
    tot.backward()
    assert abs(float(tot) - float(g["total"])) < 1e-5 * abs(float(g["total"]))
    assert abs(float(lb) - float(g["lb"])) < 1e-5 and abs(float(z) - float(g["z"])) < 1e-5
    gn = dict(zip(g["gn_keys"].tolist(), g["gn_vals"].tolist()))
    for n, ref in gn.items():
        if ref > 1e-6:
            assert abs(float(sd[n].grad.norm()) - ref) / ref < 1e-4, n


def test_oracle_token_routing_vs_reference(golden):
    """Oracle branch for cond_router=False (router on the ln_2-normalised token states) against fixture F14 (the reference with that flag)."""
    import dataclasses
    g = golden("F14_c1e4_token_routing")
    cfg = dataclasses.replace(get_config(str(g["cfg"])), cond_router=False)
    sd = make_state_dict(cfg, int(g["seed"])); inp = make_inputs(cfg, int(g["B"]), int(g["seed"]) + 1)
    for tag in ("uniform", "persample"):
        out, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], torch.from_numpy(g[f"{tag}_sigma"]), return_aux=True)
        assert torch.equal(torch.stack(aux.topk_idx), torch.from_numpy(g[f"{tag}_idx"]).long())
        assert float((out - torch.from_numpy(g[f"{tag}_out"])).norm() / torch.from_numpy(g[f"{tag}_out"]).norm()) < 2e-5
    x = O.sample_ddim(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], torch.from_numpy(g["sigmas"]))
    assert float((x - torch.from_numpy(g["x_final"])).norm() / torch.from_numpy(g["x_final"]).norm()) < 2e-5


def test_oracle_full_size_benchmark_workload_vs_reference(golden):
    """F17 (oracle/gen_golden_c2_full.py): the BENCHMARKED workload - 12 layers, d = 1024, 4 experts top-2, obs 2048, B = 128, seed 400 - produced by the
    reference's own `MoDeDiT` + `GCDenoiser` + `sample_ddim` (modedit.py:741-809, gc_sampling.py:922-951).  The oracle reproduces the forward at
    sigma = sched[3], the 10-step DDIM chunk, the sampler's intermediate actions and the expert ids of every (step, layer): twelve layers of depth
    pinned to the reference directly.  ~25 s of CPU."""
    g = golden("F17_c2_full")
    cfg = get_config(str(g["cfg"])); B, seed = int(g["B"]), int(g["seed"])
    assert (cfg.n_layers, cfg.embed_dim, cfg.num_experts, cfg.top_k, cfg.obs_dim, B) == (12, 1024, 4, 2, 2048, 128)
    assert float(g["margin"]) > 1e-5, "fixture too close to a top-k tie"
    sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, B, seed + 1)
    sched = torch.from_numpy(g["sigmas"])
    assert torch.equal(sched, O.get_sigmas_exponential(10, 1e-3, 80.0))
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    with torch.no_grad():
        f, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], float(g["sigma_fwd"]) * torch.ones(B), return_aux=True)
        x, xs = O.sample_ddim(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sched, trace=True)
    assert np.array_equal(torch.stack(aux.topk_idx)[:, 0, 0, :].numpy(), g["fwd_topk_idx"])
    emb = O.sigma_embedding(sd, sched[:-1])
    for l in range(cfg.n_layers):
        _, p = O.router_probs(sd, l, emb)
        wi, _ = O.topk_route(p, cfg.top_k, cfg.router_normalize)
        assert np.array_equal(wi.numpy(), g["topk_idx"][:, l, :]), l
    assert rel(f, g["forward"]) < TOL and rel(x, g["x_final"]) < TOL
    for i in range(1, 10):                                                       # the reference's callback `action` at step i = the update of step i-1
        assert rel(xs[i - 1][:4], g["action_in"][i]) < TOL, i
    assert np.array_equal(inp["x0"][:4].numpy(), g["action_in"][0])


def test_oracle_full_size_training_step_vs_reference(golden):
    """F18 (oracle/gen_golden_c2_train.py): the deterministic training step of the full-size model (12 layers, d = 1024, B = 16, both auxiliary router losses)
    run by the reference's own modules + autograd: the oracle's autograd reproduces the losses, the expert ids and every gradient - twelve layers of BACKWARD
    depth pinned to the reference directly.  ~40 s of CPU."""
    g = golden("F18_c2_train")
    cfg = get_config(str(g["cfg"])); B, seed = int(g["B"]), int(g["seed"])
    assert (cfg.n_layers, cfg.embed_dim, cfg.num_experts, cfg.top_k) == (12, 1024, 4, 2) and float(g["margin"]) > 1e-5
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    sd = {k: v.clone().requires_grad_(True) for k, v in make_state_dict(cfg, seed).items()}
    inp = make_inputs(cfg, B, seed + 1)
    tot, act, lb, z = O.training_total_loss(sd, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], inp["noise"], torch.from_numpy(g["sigma"]),
                                            float(g["gamma"]), float(g["delta"]))
    tot.backward()
    assert abs(float(tot) - float(g["total"])) < 1e-5 * abs(float(g["total"])) and abs(float(act) - float(g["act"])) < 1e-5 * abs(float(g["act"]))
    assert abs(float(lb) - float(g["lb"])) < 1e-5 and abs(float(z) - float(g["z"])) < 1e-5
    gn = dict(zip(g["gn_keys"].tolist(), g["gn_vals"].tolist()))
    for n, ref in gn.items():
        if ref > 1e-6:
            assert abs(float(sd[n].grad.norm()) - ref) / ref < 1e-4, n
    for key in g.files:
        if key.startswith("g:") and gn[key[2:]] > 1e-6:
            assert rel(sd[key[2:]].grad, g[key]) < 1e-4, key
        if key.startswith("gs:") and gn[key[3:]] > 1e-6:
            assert float((sd[key[3:]].grad.reshape(-1)[:len(g[key])] - torch.from_numpy(g[key])).norm()) < 1e-4 * gn[key[3:]], key
