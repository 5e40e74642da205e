"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference) on CPU fp32.

Run in the build container only (the reference never travels to the GPU box):

    python -m oracle.gen_golden            # writes tests/golden/F*.npz and prints oracle-vs-reference errors

Three absent third-party modules are stubbed exactly as SURVEY.md §8c describes (hydra.utils.instantiate = identity,
empty torchsde / torchdiffeq).  Weights/inputs come from oracle/weights.py seeds, so fixtures hold seeds + outputs only.
Golden vectors are "reference code on torch 2.10 CPU fp32" (the reference pins torch 2.2.2).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("MODE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    hydra = types.ModuleType("hydra"); hydra.utils = types.ModuleType("hydra.utils")
    hydra.utils.instantiate = lambda x, *a, **k: x
    sys.modules.setdefault("hydra", hydra); sys.modules.setdefault("hydra.utils", hydra.utils)
    sys.modules.setdefault("torchsde", types.ModuleType("torchsde"))
    tde = types.ModuleType("torchdiffeq"); tde.odeint = None
    sys.modules.setdefault("torchdiffeq", tde)
    sys.path.insert(0, REF)
    import importlib
    modedit = importlib.import_module("mode.models.networks.modedit")
    sw = importlib.import_module("mode.models.edm_diffusion.score_wrappers")
    gs = importlib.import_module("mode.models.edm_diffusion.gc_sampling")
    ut = importlib.import_module("mode.models.edm_diffusion.utils")
    return modedit, sw, gs, ut


def _ref_model(modedit, cfg, sd, train=False, **over):
    kw = dict(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cpu", goal_conditioned=True, action_dim=cfg.action_dim,
              embed_dim=cfg.embed_dim, embed_pdrob=0, attn_pdrop=0.3, n_layers=cfg.n_layers, n_heads=cfg.n_heads,
              goal_seq_len=1, obs_seq_len=1, action_seq_len=cfg.action_seq_len, state_dim=None, mlp_pdrop=0.1,
              goal_drop=0.1, linear_output=True, use_proprio=False, cond_router=True, num_experts=cfg.num_experts,
              top_k=cfg.top_k, router_normalize=True, use_goal_in_routing=False, use_argmax=False, causal=True,
              use_shared_expert=False, use_noise_token_as_input=True, use_custom_attn_mask=False, init_style="olmoe")
    kw.update(over)
    m = modedit.MoDeDiT(**kw)
    from oracle.weights import param_spec
    ref_spec = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert ref_spec == param_spec(cfg), "param_spec drifted from the reference state_dict"
    m.load_state_dict(sd)
    return m.train(train)


def _hook_router(model):
    """Capture per-layer top-k indices / router probs from the reference RouterCond."""
    cap = {"idx": [], "rp": [], "probs": []}

    def hook(_m, _inp, out):
        cap["idx"].append(out[1].clone()); cap["rp"].append(out[2].clone()); cap["probs"].append(out[3].clone())
    hs = [b.router.register_forward_hook(hook) for b in model.blocks]
    return cap, hs


def _margin(probs_list, k):
    m = 1e9
    for p in probs_list:
        s = torch.sort(p.reshape(-1, p.shape[-1]), dim=-1, descending=True).values
        if s.shape[-1] > k:
            m = min(m, float((s[:, k - 1] - s[:, k]).min()))
    return m


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    modedit, sw, gs, ut = _import_reference()
    from oracle import mode_oracle as O
    from oracle.weights import get_config, make_state_dict, make_inputs

    report = {}

    # ---- F1 schedule (gc_sampling.py:35-38)
    f1 = {f"n{n}": gs.get_sigmas_exponential(n, 1e-3, 80.0).numpy() for n in (1, 5, 10)}
    np.savez(os.path.join(OUT, "F1_schedule.npz"), **f1)
    report["F1"] = max(float(np.abs(f1[f"n{n}"] - O.get_sigmas_exponential(n, 1e-3, 80.0).numpy()).max()) for n in (1, 5, 10))

    def forward_fixture(name, cfgname, B, seed, sigma, with_blocks=False):
        cfg = get_config(cfgname)
        sd = make_state_dict(cfg, seed)
        inp = make_inputs(cfg, B, seed + 1)
        m = _ref_model(modedit, cfg, sd)
        cap, hs = _hook_router(m)
        blocks = []
        if with_blocks:
            hs += [b.register_forward_hook(lambda _m, _i, o: blocks.append(o.clone())) for b in m.blocks]
        with torch.no_grad():
            out = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], sigma)
        for h in hs:
            h.remove()
        o_out, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], sigma, return_aux=True)
        idx = torch.stack(cap["idx"]); o_idx = torch.stack(aux.topk_idx)
        assert torch.equal(idx, o_idx), f"{name}: oracle router indices differ from reference"
        d = dict(cfg=cfgname, B=B, seed=seed, sigma=sigma.numpy(), out=out.numpy(), topk_idx=idx.numpy(),
                 router_probs=torch.stack(cap["rp"]).numpy(), probs=torch.stack(cap["probs"]).numpy(),
                 perm=torch.stack(aux.perm).numpy(), counts=torch.stack(aux.counts).numpy(),
                 margin=_margin(cap["probs"], cfg.top_k))
        if with_blocks:
            d["block_out"] = torch.stack(blocks).numpy()
            report[name + ".blocks"] = _rel(torch.stack(aux.block_out), torch.stack(blocks))
        np.savez(os.path.join(OUT, name + ".npz"), **d)
        report[name] = _rel(o_out, out)
        report[name + ".margin"] = d["margin"]
        return cfg, sd, inp, m

    # ---- F2 tiny blocks, per-sample sigma
    cfg_t = get_config("tiny")
    sig_t = O.rand_log_logistic((6,), math_log(0.5), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(7))
    forward_fixture("F2_blocks_tiny", "tiny", 6, 100, sig_t, with_blocks=True)

    # ---- F3 C1 forward: (a) schedule sigma uniform over the batch, (b) per-sample log-logistic sigma
    sched = gs.get_sigmas_exponential(10, 1e-3, 80.0)
    forward_fixture("F3_c1_forward_uniform", "c1", 8, 200, sched[3] * torch.ones(8))
    sig_c1 = O.rand_log_logistic((8,), math_log(0.5), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(8))
    cfg1, sd1, inp1, m1 = forward_fixture("F3_c1_forward_persample", "c1", 8, 200, sig_c1)
    forward_fixture("F3_c1e4_forward_persample", "c1e4", 8, 210, sig_c1)

    # ---- F4 C1 10-step DDIM through GCDenoiser(sigma_data=0.5)   (gc_sampling.py:922-951)
    for cfgname, seed in (("c1", 200), ("c1e4", 210)):
        cfg = get_config(cfgname); sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, 8, seed + 1)
        den = sw.GCDenoiser(_ref_model(modedit, cfg, sd), 0.5).eval()
        cap, hs = _hook_router(den.inner_model)
        trace = []
        x = gs.sample_ddim(den, {"state_images": inp["state_images"]}, inp["x0"], inp["goals"], sched, disable=True,
                           callback=lambda d: trace.append(d["denoised"].clone()))
        for h in hs:
            h.remove()
        ox, oxs = O.sample_ddim(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sched, trace=True)
        L = cfg.n_layers
        idx = torch.stack(cap["idx"]).reshape(10, L, 8, cfg.seq_len, cfg.top_k)
        np.savez(os.path.join(OUT, f"F4_{cfgname}_ddim.npz"), cfg=cfgname, B=8, seed=seed, sigmas=sched.numpy(), x_final=x.numpy(),
                 denoised=torch.stack(trace).numpy(), topk_idx=idx.numpy(), margin=_margin(cap["probs"], cfg.top_k))
        report[f"F4_{cfgname}"] = _rel(ox, x)

    # ---- F5 C1(E4) loss + grads, deterministic training config (SURVEY §7 hard parts: dropout parity impossible)
    for cfgname, seed in (("c1e4", 210), ("c1", 200)):
        cfg = get_config(cfgname); sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, 8, seed + 1)
        m = _ref_model(modedit, cfg, sd, train=True, attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True)
        den = sw.GCDenoiser(m, 0.5).train()
        loss, F_out = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig_c1)
        lb = m.load_balancing_loss(); zl = m.compute_router_z_loss()
        loss.backward()
        grads = {k: p.grad for k, p in m.named_parameters()}
        none = sorted(k for k, g in grads.items() if g is None)
        gn = {k: float(g.norm()) for k, g in grads.items() if g is not None}
        # oracle side
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ol, oF = O.denoiser_loss(sdg, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], inp["noise"], sig_c1)
        ol.backward()
        # k=1 + router_normalize => w==1 and the router's gradient is exactly zero in theory (SURVEY §8 a-bis);
        # what autograd returns there is ~1e-9 rounding noise, excluded from the comparison.
        worst = max(_rel(sdg[k].grad, grads[k]) for k in gn if gn[k] > 1e-6)
        report[f"F5_{cfgname}.loss"] = abs(float(ol) - float(loss)) / abs(float(loss))
        report[f"F5_{cfgname}.grad_worst_rel"] = worst
        keep = {}
        for k, g in grads.items():
            if g is None:
                continue
            if g.numel() <= 4096:
                keep["g:" + k] = g.numpy()
            else:                                   # leading slice of big tensors
                keep["gs:" + k] = g.reshape(-1)[:2048].numpy()
        np.savez(os.path.join(OUT, f"F5_{cfgname}_loss_grad.npz"), cfg=cfgname, B=8, seed=seed, sigma=sig_c1.numpy(),
                 loss=float(loss), F=F_out.detach().numpy(), lb=float(lb), z=float(zl), none=np.array(none),
                 gn_keys=np.array(list(gn.keys())), gn_vals=np.array(list(gn.values()), dtype=np.float64), **keep)

    # ---- F6 fused-expert cache (modedit.py:607-633, 971-992): cached (e0,e1,p0,p1) per layer per sigma; fused == loop
    cfg = get_config("c1e4"); sd = make_state_dict(cfg, 210)
    m = _ref_model(modedit, cfg, sd)
    e_idx, e_p = [], []
    for s in sched[:-1]:
        m.reset_all_caches()
        m.precompute_experts_for_inference(s.reshape(1))
        for b in m.blocks:
            (info,) = b.routing_info.values()
            e_idx.append(info["indices"]); e_p.append(info["probs"]); b.routing_info = {}
    outs = {}
    for B in (1, 8):
        inp = make_inputs(cfg, B, 777)
        m.reset_all_caches()
        with torch.no_grad():
            outs[f"loop_B{B}"] = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], sched[2] * torch.ones(B)).numpy()
    np.savez(os.path.join(OUT, "F6_fused_cache.npz"), cfg="c1e4", seed=210, sigmas=sched[:-1].numpy(),
             idx=np.array(e_idx).reshape(10, cfg.n_layers, 2), p=np.array(e_p).reshape(10, cfg.n_layers, 2), **outs)

    # ---- F7 one C2-sized block (D=1024, E=4, k=2)
    sig_c2 = O.rand_log_logistic((16,), math_log(0.5), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(9))
    forward_fixture("F7_c2block", "c2block", 16, 300, sig_c2)

    # ---- F9 constructor-flag variants the reference can actually run: T=13 without the noise token,
    #      goal-conditioned routing, un-normalised router weights, B=1 (what the reference's rollouts use)
    for tag, over, B in (("nonoise", dict(use_noise_token_as_input=False), 5), ("goalroute", dict(use_goal_in_routing=True), 5),
                         ("nonorm", dict(router_normalize=False), 5), ("b1", dict(), 1)):
        cfg = get_config("c1e4"); sd = make_state_dict(cfg, 220); inp = make_inputs(cfg, B, 221)
        import dataclasses
        ocfg = dataclasses.replace(cfg, **{k: v for k, v in over.items()})
        m = _ref_model(modedit, cfg, sd, **over)
        cap, hs = _hook_router(m)
        sig = sig_c1[:B]
        with torch.no_grad():
            out = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], sig)
        for h in hs:
            h.remove()
        o_out, aux = O.dit_forward(sd, ocfg, inp["state_images"], inp["actions"], inp["goals"], sig, return_aux=True)
        assert torch.equal(torch.stack(cap["idx"]), torch.stack(aux.topk_idx)), tag
        np.savez(os.path.join(OUT, f"F9_{tag}.npz"), cfg="c1e4", seed=220, B=B, sigma=sig.numpy(), out=out.numpy(),
                 topk_idx=torch.stack(cap["idx"]).numpy(), margin=_margin(cap["probs"], cfg.top_k), **{k: np.array(v) for k, v in over.items()})
        report[f"F9_{tag}"] = _rel(o_out, out)
        report[f"F9_{tag}.margin"] = _margin(cap["probs"], cfg.top_k)

    # ---- F8 optimizer groups (mode_agent.py:365-384) — rule restated from the reference text (agent not importable)
    cfg = get_config("c1e4")
    m = _ref_model(modedit, cfg, make_state_dict(cfg, 210))
    names = [k for k, _ in m.named_parameters()]
    decay = [all(x not in n for x in ["bias", "LayerNorm", "embedding"]) for n in names]
    np.savez(os.path.join(OUT, "F8_optimizer_groups.npz"), names=np.array(names), decay=np.array(decay))

    print("oracle-vs-reference (rel-L2 unless noted):")
    for k, v in report.items():
        print(f"  {k:36s} {v:.3e}")


def math_log(x):
    import math
    return math.log(x)


if __name__ == "__main__":
    main()
